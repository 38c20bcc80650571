#!/bin/bash
# usage: tools/pmc_bench.sh <tag> "<CTR1 CTR2 ...>" [bench args]   (one rocprofv3 --pmc pass, on the GPU box)
tag=$1; ctrs=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --cpu-sample 0 "$@" > $out/bench.json 2> $out/err.log || echo "rc=$?"
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/*/*_counter_collection.csv")
if not f: print("no counter file"); raise SystemExit
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_locate" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, len(v), "avg %.4g"%(sum(v)/len(v)))
PY
