"""Dev probe: c2 with use_ped_radius on -- symmetric kernel (default) vs ordered kernel (SFM_SYM=0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.engine import SfmEngine
sc, forces = scenarios.baseline_scenario("c2")
cfg = default_sfm_config(forces)
cfg["use_ped_radius"] = True
for sym in ("1", "0"):
    os.environ["SFM_SYM"] = sym
    eng = SfmEngine(cfg, 0.05)
    eng.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
    eng.set_waypoint_stream(sc.seed, sc.world_side, 2.0)
    eng.run(200, redraw=True)
    eng.run(1000, redraw=True)
    ms, t, l = eng.timing()
    print(f"use_ped_radius, SFM_SYM={sym}: {eng.kernel_variant()}  tick us {ms / t * 1e3:.1f}", flush=True)
    eng.close()
