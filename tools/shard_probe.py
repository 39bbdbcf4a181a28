"""Per-rank compute time of the sharded (ordered-kernel) path, emulated on one GPU: rank 0 of G owns rows [0, N/G).
No collective here -- this is the compute a rank does between two all-gathers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine, shard_bounds

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
sc, forces = scenarios.baseline_scenario(name)
for G in (1, 2, 4, 8):
    for rank in sorted({0, G // 2, G - 1}):
        eng = HipShardEngine(default_sfm_config(forces), 0.05)
        n, n_pad = eng.load(sc)
        lo, hi, _ = shard_bounds(n, n_pad, rank, G)
        eng.set_shard(lo, hi)
        eng.run(3); eng.synchronize()
        eng.engine.run(10, redraw=True)
        ms, t, l = eng.engine.timing()
        print(f"{name} G={G} rank {rank} rows [{lo},{hi}) tick us {ms / t * 1e3:9.1f} launches/tick {l / t:.2f} kernel {eng.engine.kernel_variant()}", flush=True)
        eng.close()
