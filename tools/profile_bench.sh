#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + two separate PMC passes for the bench command.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cat /sys/fs/cgroup/cpu.max > $out/cpu_max.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --cpu-sample 0 --overlap-reads 0 "$@" > $out/bench_trace.json 2> $out/trace.err || echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --cpu-sample 0 --overlap-reads 0 "$@" > $out/bench_pmc_fetch.json 2> $out/pmc_fetch.err || echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $R/bench.py --cpu-sample 0 --overlap-reads 0 "$@" > $out/bench_pmc_write.json 2> $out/pmc_write.err || echo "pmc write rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $R/bench.py --cpu-sample 0 --overlap-reads 0 "$@" > $out/bench_pmc_sq.json 2> $out/pmc_sq.err || echo "pmc sq rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_grbm -- python3 $R/bench.py --cpu-sample 0 --overlap-reads 0 "$@" > $out/bench_pmc_grbm.json 2> $out/pmc_grbm.err || echo "pmc grbm rc=$?"
find $out -name "*.csv" | head -30
