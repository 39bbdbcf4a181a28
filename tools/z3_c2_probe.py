"""Dev probe (GPU): c2's crowd with a z spread (3-D bodies), pedestrian + acceleration force, device-resident: us per tick.
    python tools/z3_c2_probe.py [N ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine
for n in [int(v) for v in sys.argv[1:]] or [1024, 4096]:
    sc = scenarios.make_scenario(n, 1002, z_spread=1.5)
    eng = HipShardEngine(default_sfm_config(("acceleration_force", "pedestrian_force")), 0.05)
    eng.load(sc)
    eng.run(300)
    eng.synchronize()
    eng.engine.run(2000, redraw=True)
    ms, t, l = eng.engine.timing()
    print(f"3-D N={n} pedestrian + acceleration force: {ms / t * 1e3:.2f} us per tick, {l / t:.2f} launches per tick, {eng.engine.kernel_variant()}", flush=True)
    eng.close()
