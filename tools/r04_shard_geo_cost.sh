#!/bin/bash
# what the border / obstacle forces cost one replayed rank of c5 at G = 8: kernel trace of the plain tick with all forces and with the
# pedestrian force alone:  bash tools/r04_shard_geo_cost.sh <tag> [rank]
tag=${1:-r04}; r=${2:-1}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for f in all ped; do
  export PROBE_FORCES=$f
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_shardgeo_$f -o k -- python3 $root/tools/shard_trace.py $r plain > $out/${tag}_shardgeo_$f.log 2>&1 || exit 1
  echo "== rank $r, forces: $f (40 ticks)"
  python3 - $out/${tag}_shardgeo_$f/k_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_us {float(r['TotalDurationNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
done
