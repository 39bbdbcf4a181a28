"""Condense rocprofv3 --pmc passes into the tracked per-kernel summary bench.py reads.

    python3 tools/pmc_summarize.py OUT.csv WORKLOAD=DIR [WORKLOAD=DIR ...]

Every *_counter_collection.csv below DIR is read (one pass per counter group: SQ counters, FETCH_SIZE, WRITE_SIZE are
separate runs -- MI355X_MICROARCH.md, rocprofv3 PMC slots); per (kernel, counter) the mean over the dispatches is written,
skipping each kernel's first `SKIP` dispatches (warm-up).  Kernel names are cut to the bare function name.
The header comment carries the git blob id of csrc/sfm_kernels.hip as it stands beside this tool (bench.py compares it with the
source beside the library it loads: `counters_stale`).  Where DIR/meta.json (tools/pmc_ticks.py) says how many Moussaid terms
the last tick's pair kernel evaluated, a derived row VALU_PER_64_PAIR_STEP = SQ_INSTS_VALU / (terms / 64) is added for the
pair kernels: what bench.py prices a rank of a sharded run with."""
import csv
import glob
import hashlib
import json
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
        try:
            meta = json.load(open(os.path.join(d, "meta.json")))
        except OSError:
            meta = None
        if meta and meta.get("pair_terms"):
            for kernel in ("sfm_fused_tick_kernel", "sfm_pair_sym_kernel", "sfm_pair_geo_kernel"):
                v = vals.get((kernel, "SQ_INSTS_VALU"))
                if v:      # (counters of the LAST launches and the terms of the last tick: the same state to within a tick)
                    rows.append((workload, kernel, "VALU_PER_64_PAIR_STEP", v[-1] / (meta["pair_terms"] / 64.0), 1))
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "carla-social-force-model_amd", "csrc", "sfm_kernels.hip")
    data = open(src, "rb").read()
    blob = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    with open(out, "w", newline="") as f:
        f.write(f"# rocprofv3 --pmc means per launch (tools/pmc_run.sh); sfm_kernels.hip git-blob {blob}\n")
        w = csv.writer(f)
        w.writerow(["workload", "kernel", "counter", "mean_per_launch", "launches"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.3f}", r[4]])
    print(f"wrote {out}: {len(rows)} rows")


if __name__ == "__main__":
    main()
