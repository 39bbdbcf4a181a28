"""Dev probe (GPU): c2's one-launch tick on a freshly packed crowd (ticks 1-64 after an upload) and on the mixed crowd a long run
ends in, for the library named by SFM_LIB_PATH:    python tools/fused_cut_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine
sc = scenarios.make_scenario(4096, 2)
eng = HipShardEngine(default_sfm_config(("acceleration_force", "pedestrian_force")), 0.05)
def timed(k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.run(k); torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
eng.load(sc); eng.run(200); eng.load(sc)
fresh = []
for rep in range(30):
    eng.load(sc); eng.run(200)                       # velocities settle (the scenario starts from rest or not -- either way)
    loc, vel, wp = eng.engine.state()
    sc2 = scenarios.make_scenario(4096, 2); sc2.loc[:, :2] = loc[:, :2]; sc2.vel[:, :2] = vel[:, :2]
    eng.load(sc2)                                     # re-upload = fresh spatial packing of a walking crowd
    eng.run(2)
    fresh.append(timed(64))
eng.load(sc); eng.run(20000)
mixed = [timed(2000) for _ in range(5)]
print(f"{os.environ.get('SFM_LIB_PATH', 'in-tree')}: fresh-packed 64 ticks {np.median(fresh):.2f} us/tick (min {min(fresh):.2f}), mixed {np.median(mixed):.2f} us/tick  [{eng.engine.kernel_variant()}]")
