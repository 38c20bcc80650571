#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate PMC passes for tools/bench_trace.py.
# usage: tools/profile_trace.sh <tag> [bench_trace args...]
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_trace.py "$@" > $out/bench_plain.json 2> $out/plain.err || echo "plain rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/bench_trace.py "$@" > $out/bench_trace.json 2> $out/trace.err || echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $R/tools/bench_trace.py "$@" > $out/bench_pmc_fetch.json 2> $out/pmc_fetch.err || echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $R/tools/bench_trace.py "$@" > $out/bench_pmc_write.json 2> $out/pmc_write.err || echo "pmc write rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $R/tools/bench_trace.py "$@" > $out/bench_pmc_sq.json 2> $out/pmc_sq.err || echo "pmc sq rc=$?"
find $out -name "*.csv" | head -30
