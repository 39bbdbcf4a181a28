"""Dev probe (GPU, experiments build): phase stamps of every workgroup of the LAST launch of a fused run.
    make -C carla-social-force-model_amd/csrc EXPERIMENTS=1 OUT=$PWD/variants/libsfm_exp.so
    python tools/fused_census.py [c2|N] [geo]
Prints, per phase, the median / 10th / 90th percentile over workgroups in microseconds (s_memrealtime: 100 MHz):
entry -> column sums in registers -> new state in LDS -> systolic steps done -> slab rows stored; and the launch's span."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SFM_LIB_PATH", os.path.join(ROOT, "variants", "libsfm_exp.so"))
out = os.path.join(ROOT, "gpurun_out", "fused_stamps.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["SFM_FUSED_STAMPS"] = out
import numpy as np
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine
what = sys.argv[1] if len(sys.argv) > 1 else "c2"
geo = len(sys.argv) > 2
if what == "c2":
    sc, forces = scenarios.baseline_scenario("c2")
else:
    n = int(what)
    sc = scenarios.make_scenario(n, 500 + n, n_borders=max(40, n // 8) if geo else 0, n_static=max(16, n // 64) if geo else 0, n_dynamic=8 if geo else 0)
    forces = scenarios.ALL_FORCES if geo else ("acceleration_force", "pedestrian_force")
eng = HipShardEngine(default_sfm_config(forces), 0.05)
eng.load(sc)
eng.run(300)
eng.synchronize()
print(eng.engine.kernel_variant())
eng.close()
d = np.loadtxt(out, dtype=np.uint64)
b, t = d[:, 0].astype(int), d[:, 1:6].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) * 0.01
names = ["entry", "column sums in", "state in LDS", "steps done", "rows stored"]
print(f"workgroups {len(b)}, launch span {us[:, 4].max():.2f} us (first entry -> last store); entries spread {us[:, 0].max():.2f} us")
for k in range(5):
    print(f"  {names[k]:16s} at  median {np.median(us[:, k]):6.2f}  p10 {np.percentile(us[:, k], 10):6.2f}  p90 {np.percentile(us[:, k], 90):6.2f}  max {us[:, k].max():6.2f} us")
for k in range(1, 5):
    dt = us[:, k] - us[:, k - 1]
    print(f"  phase -> {names[k]:16s} median {np.median(dt):6.2f}  p10 {np.percentile(dt, 10):6.2f}  p90 {np.percentile(dt, 90):6.2f} us")
if d.shape[1] >= 6 + 32:
    # per WAVE: end of its last systolic step, and the moment it is past the barrier that follows (round 4)
    w = (d[:, 6:6 + 32].astype(np.int64) - t0) * 0.01
    nw = int((d[0, 6:6 + 16] != 0).sum())
    done, past = w[:, :nw], w[:, 16:16 + nw]
    first, last = done.min(axis=1), done.max(axis=1)
    def line(name, v):
        print(f"  {name:58s} median {np.median(v):6.2f}  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
    print(f"per wave ({nw} waves per workgroup):")
    line("first wave of the workgroup done with its steps at", first)
    line("last wave of the workgroup done with its steps at", last)
    line("skew inside a workgroup (last - first wave done)", last - first)
    line("thread 0 'steps done' -> LAST wave of its workgroup done", last - us[:, 3])
    line("last wave done -> all past the barrier", past.max(axis=1) - last)
    line("last wave done -> thread 0's row stored (combine + store)", us[:, 4] - last)
    order = np.argsort(np.argsort(done, axis=1), axis=1)            # rank of each wave's finishing time inside its workgroup
    print("  mean finishing rank of wave w (0 = first):", " ".join(f"{order[:, k].mean():.1f}" for k in range(nw)))
    print(f"  the launch's last systolic step ends at {last.max():.2f} us; last row stored at {us[:, 4].max():.2f} us")
