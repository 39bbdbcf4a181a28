#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for m in device none host; do
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rs_$m -o k -- python $GRAFT_REPO_ROOT/tools/resort_probe.py $m > $GRAFT_REPO_ROOT/gpurun_out/rs_$m.log 2>&1
  python3 - "$m" <<'PY'
import csv, os, sys
m = sys.argv[1]
rows = list(csv.DictReader(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"rs_{m}", "k_kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for pat in ("geometry", "pair_sym"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
    print(m, pat, len(d), [round(x) for x in d[::16]])
PY
done
