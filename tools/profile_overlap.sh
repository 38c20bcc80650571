cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_ovl
mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $R/tools/bench_overlap.py > $out/bench.json 2> $out/err.txt || echo rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/bench_overlap.py > $out/bench2.json 2> $out/err2.txt || echo rc=$?
