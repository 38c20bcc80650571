#!/bin/bash
# PMC counters of k_ovl_scan for a variant build: tools/pmc_scan.sh <variant> [bench_overlap args]
v=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_scan_$v
mkdir -p $out
export PBA_LIB_PATH=$R/pacbioassembly_amd/lib/variants/$v/libpba.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/pmc -- python3 $R/tools/bench_overlap.py "$@" > $out/line.json 2> $out/err.txt || echo "rc=$?"
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "k_ovl_scan" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print("$v",k,{c:f"{max(v):.3g}" for c,v in sorted(d.items())})
PY
