#!/bin/bash
# kernel trace of a few hundred drop-in ticks (N = ${1:-20}): which kernels a tick is made of, how long each runs, how far apart they start
#   bash tools/facade_kt.sh 20
n=${1:-20}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/facade_kt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 $root/tools/facade_trace.py $n > $out.log 2>&1 || exit 1
python3 - $out/k_kernel_stats.csv $out/k_kernel_trace.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:6.2f} min {float(r['MinNs'])/1e3:6.2f}")
rows=list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-90:]
t0=int(rows[0]["Start_Timestamp"])
print("the last ticks, start (us after the first shown) and duration of each launch:")
for r in rows[:12]:
    print(f"  {r['Kernel_Name'][:60]:60s} start {(int(r['Start_Timestamp'])-t0)/1e3:8.2f}  runs {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:6.2f}")
PY
