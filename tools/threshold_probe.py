"""Dev probe (GPU): where does the list cutoff (two- / three-launch tick) overtake the one-launch fused tick?  Per-tick time of
device-resident runs of plain crowds (acceleration + pedestrian force, 0.25 ped/m2) and of crowds with all forces, by N, both ways:
    python tools/threshold_probe.py [geo]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine
geo = len(sys.argv) > 1
for n in (3072, 4096, 5120, 6144, 7168, 8000):
    row = []
    for cut in ("0", "1"):
        os.environ["SFM_CUTOFF"] = cut
        if geo:
            sc = scenarios.make_scenario(n, 500 + n, n_borders=max(40, n // 8), n_static=max(16, n // 64), n_dynamic=8)
            forces = scenarios.ALL_FORCES
        else:
            sc = scenarios.make_scenario(n, 500 + n)
            forces = ("acceleration_force", "pedestrian_force")
        eng = HipShardEngine(default_sfm_config(forces), 0.05)
        eng.load(sc)
        eng.engine.set_timing(False)
        eng.run(300); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.run(1500); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 1500
        row.append((dt * 1e6, eng.engine.kernel_variant()))
        eng.close()
    print(f"N={n}: fused {row[0][0]:.1f} us ({row[0][1]})   list cutoff {row[1][0]:.1f} us ({row[1][1]})", flush=True)
