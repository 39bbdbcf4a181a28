"""Where does a tick with geometry go?  For c3 / c5: whole tick with the geometry kernel beside the pair kernel
(default), serialised (SFM_NO_OVERLAP=1), and with pedestrian_force off (geometry + epilogue only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine

for name in sys.argv[1:] or ("c3", "c5"):
    sc, forces = scenarios.baseline_scenario(name)
    reps = 200 if sc.n <= 16384 else 30
    for label, env, fs in (("overlap", {}, forces), ("serial", {"SFM_NO_OVERLAP": "1"}, forces),
                           ("geometry only", {}, tuple(f for f in forces if f != "pedestrian_force")),
                           ("pairs only", {}, ("acceleration_force", "pedestrian_force"))):
        for k in ("SFM_NO_OVERLAP",):
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = HipShardEngine(default_sfm_config(fs), 0.05)
        eng.load(sc)
        eng.run(20)
        eng.synchronize()
        eng.engine.run(reps, redraw=True)
        ms, t, l = eng.engine.timing()
        print(f"{name} {label:14s} tick us {ms / t * 1e3:9.1f}  launches/tick {l / t:.2f}", flush=True)
        eng.close()
