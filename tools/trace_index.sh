#!/bin/bash
# Kernel timeline of one seed-index build (the last of 20) under rocprofv3: tools/trace_index.sh SIZE
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_ixt
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/tools/bench_index.py --sizes $1 --reps 20 > $out/line.json 2>/dev/null
t=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$t")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_seed_count" in r["Kernel_Name"]][-1]
t0=int(rows[idx]["Start_Timestamp"])
for r in rows[idx-2:idx+34]:
    print(r["Kernel_Name"].split("(")[0][:34].ljust(34), round((int(r["Start_Timestamp"])-t0)/1e3,1), round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,1))
PY
