#!/bin/bash
# bench lines (no CPU baseline) for the given configs: "cN ticks/s ms/step pair_kernel_us tick_us"
cd $GRAFT_REPO_ROOT
for w in "$@"; do
  steps=2000; [ "$w" = "c5" ] && steps=300
  python bench.py --workload $w --steps $steps --warmup 50 --no-cpu-baseline > gpurun_out/qb_$w.json 2> gpurun_out/qb_$w.err || { tail -3 gpurun_out/qb_$w.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/qb_$w.json')); r=d['roofline']
print('$w', 'ticks/s %.1f ms/step %.4f pair_us %.1f tick_us %.1f launches %.2f' % (d['value'], d['ms_per_step'], r['kernel_us'], r['tick_us'], r['launches_per_tick']))"
done
