"""A few ticks of one BASELINE config with pedestrian_force off (geometry + epilogue only), for rocprofv3 --pmc runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc, forces = scenarios.baseline_scenario(name)
eng = HipShardEngine(default_sfm_config(tuple(f for f in forces if f != "pedestrian_force")), 0.05)
eng.load(sc)
eng.run(10)
eng.synchronize()
eng.close()
