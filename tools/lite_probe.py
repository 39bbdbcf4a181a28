import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd.engine import SfmEngine
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
cfg = default_sfm_config(("acceleration_force", "pedestrian_force"))
sc, _ = scenarios.baseline_scenario("c2")
for cut in ("0", "2", "1"):
    os.environ["SFM_CUTOFF"] = cut
    eng = SfmEngine(cfg, 0.05)
    eng.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
    eng.set_waypoint_stream(sc.seed, sc.world_side, 2.0)
    print("cut", cut, "fresh pair kernel us", eng.profile_dominant_kernel(200), flush=True)
    eng.run(300, redraw=True)
    print("cut", cut, "after 300 ticks pair kernel us", eng.profile_dominant_kernel(200), flush=True)
    eng.run(1000, redraw=True); ms, t, l = eng.timing()
    print("cut", cut, "tick us", ms / t * 1e3, "launches/tick", l / t, flush=True)
    eng.close()
