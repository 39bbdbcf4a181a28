"""Condense rocprofv3 --pmc passes into the tracked per-kernel summary bench.py reads.

    python3 tools/pmc_summarize.py OUT.csv WORKLOAD=DIR [WORKLOAD=DIR ...]

Every *_counter_collection.csv below DIR is read (one pass per counter group: SQ counters, FETCH_SIZE, WRITE_SIZE are
separate runs -- MI355X_MICROARCH.md, rocprofv3 PMC slots); per (kernel, counter) the mean over the dispatches is written,
skipping each kernel's first `SKIP` dispatches (warm-up).  Kernel names are cut to the bare function name."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

SKIP = 3
TAIL_SKIP = {"sfm_fused_tick_kernel": 1}      # the last launch of a fused run only integrates and stores (16 workgroups)


def short(name):
    m = re.search(r"(sfm_[a-z0-9_]+)", name)
    return m.group(1) if m else name


def main():
    out, specs = sys.argv[1], sys.argv[2:]
    rows = []
    for spec in specs:
        workload, d = spec.split("=", 1)
        vals = defaultdict(list)               # (kernel, counter) -> per-dispatch values in dispatch order
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            per = defaultdict(list)
            with open(path) as f:
                for r in csv.DictReader(f):
                    per[(short(r["Kernel_Name"]), r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            for key, lst in per.items():
                lst.sort()
                lst = lst[SKIP:len(lst) - TAIL_SKIP.get(key[0], 0)]
                vals[key].extend(v for _, v in lst)
        for (kernel, counter), v in sorted(vals.items()):
            if v:
                rows.append((workload, kernel, counter, sum(v) / len(v), len(v)))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["workload", "kernel", "counter", "mean_per_launch", "launches"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.3f}", r[4]])
    print(f"wrote {out}: {len(rows)} rows")


if __name__ == "__main__":
    main()
