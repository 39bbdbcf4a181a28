#!/bin/bash
# kernel trace of one replayed rank of c5 at G = 8 (plain and split tick):  bash tools/r04_shard_kt.sh <tag> [rank]
tag=${1:-r04}; r=${2:-1}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in plain split; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_shardkt_$mode -o k -- python3 $root/tools/shard_trace.py $r $mode > $out/${tag}_shardkt_$mode.log 2>&1 || exit 1
  echo "== rank $r, $mode tick (40 ticks)"
  python3 - $out/${tag}_shardkt_$mode/k_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_us {float(r['TotalDurationNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
done
