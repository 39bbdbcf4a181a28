#!/bin/bash
# A/B of bench lines between settings of ONE environment variable:  WLS="c3 c4" bash tools/ab_env.sh SFM_LIST_AHEAD 0 1
cd $GRAFT_REPO_ROOT
var=$1; shift
for w in ${WLS:-c3 c4 c5}; do
  for val in "$@"; do
    echo "== $w $var=$val"
    env $var=$val python bench.py --workload $w --steps ${STEPS:-200} --warmup ${WARM:-100} --min-seconds 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('ticks/s %.0f  us/tick %.2f  (min %.0f max %.0f, %d win)  kernel_us %.2f tick_us %.2f launches %.2f  terms %s' % (d['value'], 1e6/d['value'], d['min'], d['max'], d['windows'], r['kernel_us'], r['tick_us'], r['launches_per_tick'], d.get('evaluated_pair_terms_per_tick')))"
  done
done
