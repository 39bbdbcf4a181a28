"""Dev probe (GPU): wall-clock tick of ONE replayed rank of c5 at G = 8 (c4: G = 4; rank blocks, equal shares), plain and split tick,
for the library named by SFM_LIB_PATH and whatever SFM_* knobs are set:   python tools/shard_rank_time.py [rank] [c5|c4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine, block_layout, equal_bounds
r = int(sys.argv[1]) if len(sys.argv) > 1 else 1
name = sys.argv[2] if len(sys.argv) > 2 else "c5"
G = 8 if name == "c5" else 4
sc, forces = scenarios.baseline_scenario(name)
if os.environ.get("PROBE_FORCES") == "ped":     # the same crowd without border / obstacle forces
    forces = ("acceleration_force", "pedestrian_force")
elif os.environ.get("PROBE_FORCES"):            # ... or with one of them: border_force / static_obstacle_force / dynamic_obstacle_force
    forces = ("acceleration_force", "pedestrian_force") + tuple(os.environ["PROBE_FORCES"].split(","))
eng = HipShardEngine(default_sfm_config(forces), 0.05)
eng.set_partition(*block_layout(G))
n, n_pad = eng.load(sc)
b = equal_bounds(n, n_pad, G)
e = eng.engine
e.tick()
e.set_shard(b[r], b[r + 1])
e.set_timing(False)
INTEGRATE = os.environ.get("PROBE_INTEGRATE") == "1"     # the rank's own rows walk on (the other ranks' stay: nothing exchanges here)
if os.environ.get("PROBE_WHOLE") == "1":                  # no shard: the whole crowd's tick, timed the same way
    e.set_shard(0, n)
out = []
for mode in ("plain", "split"):
    def go(k):
        for _ in range(k):
            if mode == "split":
                eng.begin(); eng.end()
            else:
                e.tick(integrate=INTEGRATE)
    go(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(200); torch.cuda.synchronize()
    out.append(f"{mode} {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us")
print(f"{name} rank {r} of {G}: " + "   ".join(out) + f"   terms {e.pair_work()[1] / 1e6:.1f} M   [{e.kernel_variant()}]  " +
      " ".join(f"{k}={v}" for k, v in os.environ.items() if (k.startswith("SFM_") or k.startswith("PROBE_")) and k != "SFM_LIB_PATH"), flush=True)
eng.close()
