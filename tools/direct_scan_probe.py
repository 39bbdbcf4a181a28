"""Dev probe (GPU): device-resident ticks of small crowds with MANY polylines per pedestrian tile, for builds with different
GEO_DIRECT_PER_WAVE (make EXTRA=-DGEO_DIRECT_PER_WAVE=n OUT=...):    SFM_LIB_PATH=... python tools/direct_scan_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine
row = []
for n, nb, ns, nd in ((64, 40, 16, 4), (64, 200, 60, 8), (64, 400, 100, 8), (256, 400, 100, 8), (256, 1500, 200, 8), (1024, 1000, 100, 8), (2048, 1800, 200, 8)):
    sc = scenarios.make_scenario(n, 900 + n + nb, n_borders=nb, n_static=ns, n_dynamic=nd, border_len=(5.0, 30.0))
    eng = HipShardEngine(default_sfm_config(scenarios.ALL_FORCES), 0.05)
    eng.load(sc); eng.engine.set_timing(False)
    eng.run(200); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.run(2000); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2000
    row.append(f"N={n} K={nb + ns + nd}: {dt * 1e6:.1f}")
    eng.close()
print(os.environ.get("SFM_LIB_PATH", "in-tree").split("/")[-1], " | ".join(row), flush=True)
