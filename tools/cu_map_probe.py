"""Dev probe (GPU): is the workgroup -> CU placement of the symmetric pair kernel's grid the same from launch to launch?
Dumps the HW id stamps of the LAST pair-kernel launch of runs of different lengths and compares the maps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from carla_social_force_model_amd.engine import SfmEngine
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config

cfg = default_sfm_config(("acceleration_force", "pedestrian_force"))
sc, _ = scenarios.baseline_scenario("c2")
maps = []
for k, ticks in enumerate((1, 2, 7, 50, 51, 200)):
    out = os.path.join(ROOT, "gpurun_out", f"stamps_{k}.txt")
    os.environ["SFM_STAMPS"] = out
    eng = SfmEngine(cfg, 0.05)
    eng.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
    eng.run(ticks)
    eng.close()
    d = np.loadtxt(out, dtype=np.uint64)
    hw = d[:, 2]
    cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(int); sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
    xcc = (hw >> np.uint64(32)).astype(int) & 0xf
    key = np.where(d[:, 1] > 0, xcc * 10000 + se * 1000 + sh * 100 + cu, -1)
    maps.append(key)
    live = key >= 0
    print(f"run of {ticks} ticks: stamped {live.sum()} workgroups, distinct CUs {len(np.unique(key[live]))}")
ref = maps[0]
for k, m in enumerate(maps[1:], 1):
    both = (ref >= 0) & (m >= 0)
    print(f"map {k} vs map 0: same CU for {int((ref[both] == m[both]).sum())} of {int(both.sum())} workgroups")
# structure of map 0: xcc by index, and how the workgroups of one CU are spaced
live = np.flatnonzero(ref >= 0)
print("first 24 live workgroup indices and their CU keys:", list(zip(live[:24].tolist(), ref[live[:24]].tolist())))
xcc0 = ref[live] // 10000
print("xcc == index % 8 for", int((xcc0 == live % 8).sum()), "of", len(live))
u = np.unique(ref[live])
gaps = [np.diff(np.sort(live[ref[live] == c])) for c in u[:6]]
print("index gaps between the workgroups of the first CUs:", [g.tolist() for g in gaps])
