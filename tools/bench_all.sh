#!/bin/bash
# all five BASELINE configs: bench lines into gpurun_out/bench_<tag>_cN.json, rocprofv3 kernel stats for c2/c3/c5
tag=${1:-run}
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python bench.py --workload c2 > $out/bench_${tag}_c2.json 2> $out/bench_${tag}_c2.err || exit 1
for w in c1 c3 c4; do python bench.py --workload $w --steps 2000 --warmup 50 --no-cpu-baseline > $out/bench_${tag}_$w.json 2> $out/bench_${tag}_$w.err || exit 1; done
python bench.py --workload c5 --steps 300 --warmup 50 --no-cpu-baseline > $out/bench_${tag}_c5.json 2> $out/bench_${tag}_c5.err || exit 1
cd /tmp && export TMPDIR=/tmp
for w in c2 c3 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_$w -o k -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 300 --warmup 50 --no-cpu-baseline > $out/prof_${tag}_$w.log 2>&1 || exit 1
done
python3 - "$tag" <<'PY'
import json, os, sys
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
for w in ("c1", "c2", "c3", "c4", "c5"):
    d = json.load(open(os.path.join(out, f"bench_{sys.argv[1]}_{w}.json")))
    print(w, "ticks/s %.1f  ms/step %.4f  pair kernel us %.1f  tick us %.1f  launches %.2f" % (
        d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["tick_us"], d["roofline"]["launches_per_tick"]))
PY
