#!/bin/bash
# geometry-kernel duration per BASELINE config (fresh state, 10 ticks, pedestrian_force off): rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_$w -o k -- python $GRAFT_REPO_ROOT/tools/geo_pmc.py $w > $GRAFT_REPO_ROOT/gpurun_out/kt_$w.log 2>&1
  python3 - "$w" <<'PY'
import csv, os, sys
w = sys.argv[1]
for r in csv.DictReader(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"kt_{w}", "k_kernel_stats.csv"))):
    if "geometry" in r["Name"] or "tick_kernel" in r["Name"]:
        print(w, r["Name"][:48], "calls", r["Calls"], "avg_us %.1f min %.1f max %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
