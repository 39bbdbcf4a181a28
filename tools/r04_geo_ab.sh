#!/bin/bash
# A/B of library builds on everything the border / obstacle code touches: a replayed rank of c5 / c4, c1 / c3 / c5 bench lines, mid-size
# crowds with all forces, the drop-in tick.   bash tools/r04_geo_ab.sh "" variants/libsfm_old.so
root=$GRAFT_REPO_ROOT
for l in "$@"; do
  echo "== lib ${l:-product}"
  if [ -n "$l" ]; then export SFM_LIB_PATH=$root/$l; else unset SFM_LIB_PATH; fi
  python tools/shard_rank_time.py 1 c5 2>/dev/null
  python tools/shard_rank_time.py 1 c4 2>/dev/null
  for w in c1 c3 c5; do python bench.py --workload $w --steps 300 --warmup 50 --min-seconds 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print(\"$w us/tick %.2f\" % (1e6/d[\"value\"]))"; done
  python tools/mid_crowd_probe.py 2>/dev/null | tail -4
  python tools/facade_latency.py 2>/dev/null | grep "N = "
done
