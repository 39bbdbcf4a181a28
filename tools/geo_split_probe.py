"""Dev probe (GPU): geometry-only tick time (pedestrian force off) of c3 / c5 with each geometry family alone -- how the
geometry kernel's time splits over borders / static / dynamic obstacles (find cost ~ number of polylines, scan cost ~ kept items)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine

for name in sys.argv[1:] or ("c3", "c5"):
    sc, forces = scenarios.baseline_scenario(name)
    reps = 100 if sc.n <= 16384 else 20
    sets = {"all geometry": ("border_force", "static_obstacle_force", "dynamic_obstacle_force"), "borders": ("border_force",),
            "static": ("static_obstacle_force",), "dynamic": ("dynamic_obstacle_force",), "none (epilogue only)": ()}
    for label, fs in sets.items():
        eng = HipShardEngine(default_sfm_config(("acceleration_force",) + fs), 0.05)
        eng.load(sc)
        eng.run(5)
        eng.synchronize()
        eng.engine.run(reps, redraw=True)
        ms, t, l = eng.engine.timing()
        print(f"{name} {label:22s} tick us {ms / t * 1e3:9.1f}  launches/tick {l / t:.2f}  {eng.engine.kernel_variant()}", flush=True)
        eng.close()
