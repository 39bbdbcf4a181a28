#!/bin/bash
# Round-4 A/B of c2 between library builds, interleaved ROUNDS times on one box:   bash tools/r04_ab.sh <out-name> lib1 lib2 ...   ("" = in-tree)
cd $GRAFT_REPO_ROOT
name=$1; shift
out=gpurun_out/$name.txt
: > $out
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    echo "== ${WL:-c2} lib=${lib:-in-tree}" >> $out
    SFM_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} python bench.py --workload ${WL:-c2} --steps ${STEPS:-2000} --warmup ${WARM:-300} --min-seconds ${SECS:-1} --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('ticks/s %.0f  us/tick %.3f  (min %.0f max %.0f, %d win)  kernel_us %.3f tick_us %.3f launches %.2f' % (d['value'], 1e6/d['value'], d['min'], d['max'], d['windows'], r['kernel_us'], r['tick_us'], r['launches_per_tick']))" >> $out
  done
done
cat $out
