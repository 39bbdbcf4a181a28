"""Dev probe (GPU): long device-resident runs with the state checked at the end (finite, inside the speed cap, still inside the world's
reach) -- the one-launch tick planar / 3-D / with radii, and the list-cutoff tick:   python tools/soak.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine

def soak(label, sc, forces, ticks, radius=False):
    cfg = default_sfm_config(forces)
    cfg["use_ped_radius"] = radius
    eng = HipShardEngine(cfg, 0.05)
    eng.load(sc)
    t0 = time.perf_counter()
    done = 0
    while done < ticks:
        k = min(50000, ticks - done)
        eng.run(k); done += k
    eng.synchronize()
    dt = time.perf_counter() - t0
    loc, vel, _ = eng.engine.state()
    speed = np.linalg.norm(vel, axis=1)
    ok = np.isfinite(loc).all() and np.isfinite(vel).all() and speed.max() <= 1.3 * float(sc.target_speed.max()) * 1.001 \
        and np.abs(loc[:, :2]).max() < 4.0 * sc.world_side
    print(f"{label:46s} {ticks:8d} ticks in {dt:6.1f} s ({dt / ticks * 1e6:6.1f} us per tick)  max speed {speed.max():.3f}  "
          f"|x|max {np.abs(loc[:, :2]).max():8.1f} (world {sc.world_side:.0f})  {eng.engine.kernel_variant()}  {'OK' if ok else 'FAILED'}", flush=True)
    eng.close()
    return ok

ok = True
sc, f = scenarios.baseline_scenario("c2"); ok &= soak("c2 (N=4096, pedestrian + acceleration)", sc, f, 1000000)
sc, f = scenarios.baseline_scenario("c1"); ok &= soak("c1 (N=64, all forces)", sc, f, 500000)
sc = scenarios.make_scenario(4096, 77, n_borders=512, n_static=64, n_dynamic=8, z_spread=1.5); ok &= soak("N=4096 3-D, all forces", sc, scenarios.ALL_FORCES, 200000)
sc = scenarios.make_scenario(2048, 78, z_spread=1.0); ok &= soak("N=2048 3-D, pedestrian + acceleration", sc, ("acceleration_force", "pedestrian_force"), 300000)
sc, f = scenarios.baseline_scenario("c2"); ok &= soak("c2 with use_ped_radius", sc, f, 300000, radius=True)
sc, f = scenarios.baseline_scenario("c3"); ok &= soak("c3 (N=16384, all forces, list cutoff)", sc, f, 100000)
sys.exit(0 if ok else 1)
