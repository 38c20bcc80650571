#!/bin/bash
# Time the all-vs-all bench with several variant builds on the same box:
#   tools/ab_overlap.sh "h16 h8" READS [TARGETS_PER_CALL]      (variants from tools/build_variant.py)
for v in $1; do
  PBA_LIB_PATH=pacbioassembly_amd/lib/variants/$v/libpba.so timeout -k 10 600 python tools/bench_overlap.py --reads $2 --reps 2 --targets-per-call ${3:-25000} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'overlap', d['seconds'], 'scan', round(d['scan_ms'],1), 'sort', round(d['sort_ms'],1), 'walk', round(d['walk_ms'],1), 'listed', d.get('n_listed'), 'pairs', d['pairs'], 'overlaps', d['overlaps'], flush=True)" || exit 1
done
