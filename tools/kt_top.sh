#!/bin/bash
# kernel-trace stats of a few ticks of the given workloads, top kernels printed:   bash tools/kt_top.sh <tag> c3 c4 c5
tag=${1:-kt}; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  steps=400; [ $w = c4 ] && steps=200; [ $w = c5 ] && steps=60
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt_$w -o k -- python3 $root/bench.py --workload $w --steps $steps --warmup 20 --windows 2 --min-seconds 0.05 --no-cpu-baseline > $out/${tag}_kt_$w.log 2>&1 || exit 1
  echo "== $w"
  python3 - $out/${tag}_kt_$w/k_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f} pct {r['Percentage']}")
PY
done
