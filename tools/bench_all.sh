#!/bin/bash
# the five BASELINE configs through bench.py (default protocol), lines into gpurun_out/<tag>_bench_<w>.json
tag=${1:-run}
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for w in c1 c2 c3 c4 c5; do
  extra=""; [ $w = c4 ] && extra="--no-cpu-baseline"      # (c1, c2, c3, c5 carry cpu_baseline; c4's crowd disperses, its line is a state-dependent curiosity)
  steps=2000; [ $w = c4 ] && steps=1000; [ $w = c5 ] && steps=300
  timeout -k 10 600 python bench.py --workload $w --steps $steps --warmup 100 $extra > $out/${tag}_bench_$w.json 2> $out/${tag}_bench_$w.err || exit 1
  python3 -c "
import json,sys
d=json.load(open('$out/${tag}_bench_$w.json')); r=d['roofline']
print('$w ticks/s %.0f  us/tick %.2f (min %.0f max %.0f, %d windows)  pair kernel us %.2f  tick us %.2f  launches %.2f  bound %s frac %.3f' % (d['value'], 1e6/d['value'], d['min'], d['max'], d['windows'], r['kernel_us'], r['tick_us'], r['launches_per_tick'], r['bound'], r['frac']))"
done
