#!/bin/bash
# one gpurun call: GPU tests, c2 bench line, kernel-trace stats of the same command, PMC passes.   bash tools/gpu_round.sh <tag> [workloads...]
tag=${1:-r03}; shift
wl=${@:-c2}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd $root
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest.log 2>&1; rc=$?
tail -3 $out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
for w in $wl; do
  timeout -k 10 300 python bench.py --workload $w > $out/${tag}_bench_$w.json 2> $out/${tag}_bench_$w.err || exit 1
  cat $out/${tag}_bench_$w.json | python3 -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$w', 'ticks/s %.0f (min %.0f max %.0f, %d windows) kernel_us %.2f tick_us %.2f frac %s' % (d['value'], d['min'], d['max'], d['windows'], r['kernel_us'], r['tick_us'], r['frac']))"
done
cd /tmp && export TMPDIR=/tmp
for w in $wl; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt_$w -o k -- python3 $root/bench.py --workload $w --steps ${KT_STEPS:-300} --warmup 50 --no-cpu-baseline > $out/${tag}_kt_$w.log 2>&1 || exit 1
done
bash $root/tools/pmc_run.sh $tag $wl
