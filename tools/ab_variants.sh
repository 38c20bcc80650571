#!/bin/bash
# Time bench.py (and optionally the all-vs-all bench) with several variant builds on the same box:
#   tools/ab_variants.sh "head cur" [overlap-reads]      (variants from tools/build_variant.py)
for rep in 1 2; do
for v in $1; do
  PBA_LIB_PATH=pacbioassembly_amd/lib/variants/$v/libpba.so timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --overlap-reads 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', d['ms_per_step'])" || exit 1
done; done
if [ -n "$2" ]; then for v in $1; do
  PBA_LIB_PATH=pacbioassembly_amd/lib/variants/$v/libpba.so timeout -k 10 300 python tools/bench_overlap.py --reads $2 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'overlap', d['seconds'], d['scan_ms'], d['sort_ms'], d['walk_ms'])" || exit 1
done; fi
