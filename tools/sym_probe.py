"""Dev probe (GPU): per-kernel time of the symmetric path as a function of systolic steps per wave."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from carla_social_force_model_amd.engine import SfmEngine
    from carla_social_force_model_amd import scenarios
    from carla_social_force_model_amd.config import default_sfm_config
    cfg = default_sfm_config(("acceleration_force", "pedestrian_force"))
    sc = scenarios.make_scenario(4096, 1002)
    eng = SfmEngine(cfg, 0.05)
    eng.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
    eng.set_waypoint_stream(sc.seed, sc.world_side, 2.0)
    eng.run(100, redraw=True); eng.run(1000, redraw=True)
    ms, t, l = eng.timing()
    print(f"steps={os.environ.get('SFM_DEBUG_STEPS','-')} {ms/t*1e3:.2f} us/tick")
else:
    for st in ("0", "1", "4", "8", "16"):
        env = dict(os.environ, SFM_DEBUG_STEPS=st, SFM_SYM="1")
        print(subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
