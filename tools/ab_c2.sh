#!/bin/bash
# A/B of the c2 tick under different SFM_* settings:  bash tools/ab_c2.sh "SFM_FUSED=0" "SFM_FUSED=1" ...
cd $GRAFT_REPO_ROOT
for env in "$@"; do
  echo "== $env"
  env $env python bench.py --workload ${WL:-c2} --steps ${STEPS:-2000} --warmup ${WARM:-300} --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('ticks/s %.0f  us/tick %.2f  (min %.0f max %.0f, %d win)  kernel_us %.2f tick_us %.2f launches %.2f  terms %s' % (d['value'], 1e6/d['value'], d['min'], d['max'], d['windows'], r['kernel_us'], r['tick_us'], r['launches_per_tick'], d.get('evaluated_pair_terms_per_tick')))"
done
