"""A few ticks of ONE replayed rank of c5 at G = 8 (rank blocks 2x4, equal shares) for `rocprofv3 --kernel-trace --stats`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine, block_layout, equal_bounds
sc, forces = scenarios.baseline_scenario("c5")
G, r = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 3
if os.environ.get("PROBE_FORCES") == "ped":     # the same crowd without border / obstacle forces
    forces = ("acceleration_force", "pedestrian_force")
elif os.environ.get("PROBE_FORCES", "all") != "all":          # ... or with some of them: border_force,static_obstacle_force,...
    forces = ("acceleration_force", "pedestrian_force") + tuple(os.environ["PROBE_FORCES"].split(","))
eng = HipShardEngine(default_sfm_config(forces), 0.05)
if os.environ.get("PROBE_LAYOUT") != "plain":   # plain: the whole crowd's strip packing, the rank a vertical slab of it
    eng.set_partition(*block_layout(G))
n, n_pad = eng.load(sc)
b = equal_bounds(n, n_pad, G)
eng.engine.tick()
if os.environ.get("PROBE_WHOLE") != "1":        # PROBE_WHOLE=1: no shard, the whole crowd's tick
    eng.engine.set_shard(b[r], b[r + 1])
split = len(sys.argv) > 2 and sys.argv[2] == "split"
for _ in range(40):
    if split:
        eng.begin(); eng.end()
    else:
        eng.engine.tick()
eng.synchronize()
eng.close()
