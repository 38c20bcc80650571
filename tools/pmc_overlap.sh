#!/bin/bash
# PMC passes over tools/bench_overlap.py (run on the GPU box through gpurun): usage tools/pmc_overlap.sh <tag> [bench_overlap args]
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/bench_overlap.py "$@" > $out/line_trace.json 2> $out/trace.err || echo "trace rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $R/tools/bench_overlap.py "$@" > $out/line_sq.json 2> $out/sq.err || echo "sq rc=$?"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/pmc_sq2 -- python3 $R/tools/bench_overlap.py "$@" > $out/line_sq2.json 2> $out/sq2.err || echo "sq2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $R/tools/bench_overlap.py "$@" > $out/line_fetch.json 2> $out/fetch.err || echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $R/tools/bench_overlap.py "$@" > $out/line_write.json 2> $out/write.err || echo "write rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/pmc_tcc -- python3 $R/tools/bench_overlap.py "$@" > $out/line_tcc.json 2> $out/tcc.err || echo "tcc rc=$?"
find $out -name "*.csv" | wc -l
