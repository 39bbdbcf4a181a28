"""Dev probe (GPU): per-workgroup start/end stamps of the symmetric pair kernel -> residency picture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "stamps.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["SFM_STAMPS"] = out
os.environ["SFM_CUTOFF"] = os.environ.get("CENSUS_CUT", "0")
import numpy as np
from carla_social_force_model_amd.engine import SfmEngine
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
cfg = default_sfm_config(("acceleration_force", "pedestrian_force"))
sc, _ = scenarios.baseline_scenario("c2")
eng = SfmEngine(cfg, 0.05)
eng.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
eng.run(200)
eng.close()
d = np.loadtxt(out, dtype=np.uint64)
d = d[d[:, 1] > 0]
skipped = d[:, 2] == np.uint64(0xFFFFFFFFFFFFFFFF)
print('skipped workgroups', int(skipped.sum()))
d = d[~skipped]
t0 = d[:, 0].min()
s = (d[:, 0] - t0).astype(float) * 10.0   # 100 MHz -> ns
e = (d[:, 1] - t0).astype(float) * 10.0
print("workgroups stamped", len(d), "kernel span ns", e.max(), "median lifetime ns", np.median(e - s))
print("start times ns percentiles", np.percentile(s, [0, 25, 50, 75, 90, 99, 100]))
print("end times ns percentiles", np.percentile(e, [0, 25, 50, 75, 90, 99, 100]))
hw = d[:, 2]
cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(int); sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
xcc = (hw >> np.uint64(32)).astype(int) & 0xf
key = xcc * 10000 + se * 1000 + sh * 100 + cu
u, c = np.unique(key, return_counts=True)
print("distinct CUs", len(u), "workgroups per CU min/median/max", c.min(), np.median(c), c.max())
for t in (2000, 5000, 8000, 11000, 14000):
    print(f"resident workgroups at t={t} ns:", int(((s <= t) & (e > t)).sum()))
# per-CU finishing time and lifetime spread (load balance picture)
life = e - s
last = np.array([e[key == k].max() for k in u])
work = np.array([life[key == k].sum() for k in u])
print("per-CU last end ns: min/median/max", last.min(), np.median(last), last.max())
print("per-CU sum of lifetimes ns: min/median/max", work.min(), np.median(work), work.max())
print("lifetime ns percentiles", np.percentile(life, [0, 10, 25, 50, 75, 90, 100]))
