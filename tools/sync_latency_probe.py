"""Dev probe (GPU): what a 20-tick timing window of c2 pays around its ticks -- host clock with torch.cuda.synchronize() against the
GPU-side elapsed time of the same run, and against a busy poll of the stream:    python tools/sync_latency_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.engine import SfmEngine
sc, forces = scenarios.baseline_scenario("c2")
eng = SfmEngine(default_sfm_config(forces), 0.05)
stream = torch.cuda.Stream()
eng.set_stream(stream.cuda_stream)
eng.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
eng.set_waypoint_stream(sc.seed, sc.world_side, 2.0)
eng.set_timing(True)
eng.run(2000, redraw=True); torch.cuda.synchronize()
for mode in ("synchronize", "poll", "synchronize", "poll"):
    host, gpu = [], []
    for _ in range(400):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(20, redraw=True)
        if mode == "poll":
            while not stream.query():
                pass
        torch.cuda.synchronize()
        host.append((time.perf_counter() - t0) * 1e6)
        gpu.append(eng.timing()[0] * 1e3)
    print(f"{mode:12s}: host window {np.median(host):7.1f} us   GPU events {np.median(gpu):7.1f} us   = {np.median(host) / 20:.2f} / {np.median(gpu) / 20:.2f} us per tick")
eng.close()
