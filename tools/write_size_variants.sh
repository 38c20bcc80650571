#!/bin/bash
# WRITE_SIZE of k_locate for variant builds: tools/write_size_variants.sh "head p1 p2"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  rm -rf $R/gpurun_out/ws_$v
  PBA_LIB_PATH=$R/pacbioassembly_amd/lib/variants/$v/libpba.so timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ws_$v -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --overlap-reads 0 > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/ws_$v/*/*_counter_collection.csv")[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_locate" in r["Kernel_Name"]]
print("$v", "k_locate WRITE_SIZE MB per launch", round(sum(v)/len(v)/1024,1))
PY
done
