"""Dev probe (GPU): per-tick time of mid-sized crowds with all forces (no list cutoff below 8192 pedestrians):
    python tools/mid_crowd_probe.py [N ...]        flat crowds (symmetric path);  Z=1 python tools/mid_crowd_probe.py: 3-D crowds (ordered kernel)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine
z = float(os.environ.get("Z", "0"))
for n in [int(a) for a in sys.argv[1:]] or [512, 2048, 4096]:
    sc = scenarios.make_scenario(n, 500 + n, n_borders=max(40, n // 8), n_static=max(16, n // 64), n_dynamic=8, z_spread=z)
    eng = HipShardEngine(default_sfm_config(scenarios.ALL_FORCES), 0.05)
    eng.load(sc)
    eng.engine.set_timing(False)
    eng.run(200); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.run(2000); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2000
    eng.engine.set_timing(True); eng.run(50)
    print(f"{'3-D ' if z else ''}N={n}: {dt * 1e6:.1f} us per tick, {eng.engine.timing()[2] / 50:.1f} launches per tick, {eng.engine.kernel_variant()}", flush=True)
    eng.close()
