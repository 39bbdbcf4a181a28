#!/bin/bash
# second half of tools/final_round.sh as a call of its own (the counter summary must already be in profiles/): bench lines, the driver's
# command line, kernel-trace stats.   bash tools/final_round_b.sh r04
tag=${1:-r04}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
bash $root/tools/bench_all.sh $tag || exit 1
cd $root && timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_c2_driver.json 2> $out/${tag}_bench_c2_driver.err
python3 -c "import json; d=json.load(open('$out/${tag}_bench_c2_driver.json')); print('driver command: %.0f ticks/s, %.2f us/tick, frac %s stale %s' % (d['value'], 1e6/d['value'], d['roofline']['frac'], d['roofline'].get('counters_stale')))"
bash $root/tools/kt_top.sh $tag c2 c3 c5
