#!/bin/bash
# One gpurun call that produces the round's judged artefacts with ONE build: counter passes -> profiles/<tag>_pmc_summary.csv (on the
# box, so that the bench lines read it), the five bench lines, kernel-trace stats of the same bench command.
#   bash tools/final_round.sh r04      (inside gpurun; copy gpurun_out/<tag>_* into profiles/ afterwards)
tag=${1:-r04}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
bash $root/tools/pmc_run.sh $tag c1 c2 c3 c4 c5 || exit 1
cp $out/${tag}_pmc_summary.csv $root/profiles/${tag}_pmc_summary.csv
bash $root/tools/bench_all.sh $tag || exit 1
# the driver's own command line as well (20-step windows)
cd $root && timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_c2_driver.json 2> $out/${tag}_bench_c2_driver.err
python3 -c "import json; d=json.load(open('$out/${tag}_bench_c2_driver.json')); print('driver command: %.0f ticks/s, %.2f us/tick, frac %s stale %s' % (d['value'], 1e6/d['value'], d['roofline']['frac'], d['roofline'].get('counters_stale')))"
bash $root/tools/kt_top.sh $tag c2 c3 c5
