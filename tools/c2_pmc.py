"""A few c2 ticks for rocprofv3 --pmc runs on the symmetric pair kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine
sc, forces = scenarios.baseline_scenario("c2")
eng = HipShardEngine(default_sfm_config(forces), 0.05)
eng.load(sc)
eng.run(20)
eng.synchronize()
eng.close()
