"""Dev probe (GPU): the whole-crowd tick of a BASELINE workload on the SAME state every time (state re-uploaded before each timed tick), for
A/B builds whose results differ on purpose (probe builds):   python tools/same_state_tick.py c5 [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sc, forces = scenarios.baseline_scenario(name)
eng = HipShardEngine(default_sfm_config(forces), 0.05)
t = []
for k in range(reps):
    eng.load(sc)
    eng.engine.tick(); eng.synchronize()            # settles lists / launch shape on this state (velocities move, positions do not)
    eng.load(sc)
    eng.engine.set_timing(True)
    eng.engine.tick()
    ms, ticks, launches = eng.engine.timing()
    t.append(ms * 1e3)
print(f"{name}: tick on the uploaded state {np.median(t[1:]):.1f} us (median of {reps - 1}; all: {' '.join('%.0f' % v for v in t)})  {eng.engine.kernel_variant()}  {os.environ.get('SFM_LIB_PATH', '')[-24:]}", flush=True)
eng.close()
