"""Dev script (GPU box): time kernel shapes (SFM_IPW x SFM_TEAM) on ped+accel crowds of several sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd.engine import SfmEngine
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
cfg = default_sfm_config(("acceleration_force", "pedestrian_force"))
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 16384, 65536]
for n in sizes:
    sc = scenarios.make_scenario(n, 1002)
    combos = [(0, 4, 2), (0, 4, 4), (0, 1, 4), (1, 4, 2)] if len(sys.argv) > 1 and sys.argv[1] == "short" else \
             [(0, t, w) for t in (1, 4) for w in (1, 2, 4, 8)] + [(1, 4, 2)]
    for sym, team, ipw in combos:
        if True:
            os.environ["SFM_IPW"] = str(ipw); os.environ["SFM_TEAM"] = str(team); os.environ["SFM_SYM"] = str(sym)
            eng = SfmEngine(cfg, 0.05)
            eng.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
            eng.set_waypoint_stream(sc.seed, sc.world_side, 2.0)
            reps = max(3, min(1000, int(2e10 / (n * n))))
            eng.run(max(2, reps // 10), redraw=True); eng.run(reps, redraw=True)
            ms, t, l = eng.timing()
            print(f"N={n:6d} sym={sym} {eng.kernel_variant():44s} {ms/t*1e3:9.2f} us/tick {t/ms*1e3:9.0f} ticks/s {ms/t*1e9/(n*(n-1.0)):.3f} ps/pair", flush=True)
            eng.close()
