cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_ovl2
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/bench_overlap.py --reads 200000 --coverage 20 --targets-per-call 50000 > $out/bench.json 2> $out/err.txt || echo rc=$?
