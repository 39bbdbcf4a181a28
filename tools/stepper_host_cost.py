"""Dev probe (GPU): host time per tick of the sharded stepper's loop (begin / wait / end / start the all-gather) with a
single-rank RCCL group on a workload whose GPU time is small (c1): what the Python + ctypes + torch.distributed layer costs."""
import os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine, ShardedStepper
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
sc, forces = scenarios.baseline_scenario(sys.argv[1] if len(sys.argv) > 1 else "c1")
eng = HipShardEngine(default_sfm_config(forces), 0.05)
st = ShardedStepper(eng, sc)
st.world = 1
def loop(n, collective=True):
    p = None
    for _ in range(n):
        eng.begin(); st._gather_finish(p); eng.end()
        p = st._gather_start(eng.packed()) if collective else None
    st._gather_finish(p)
    torch.cuda.synchronize()
for coll in (False, True):
    loop(50, coll)
    t0 = time.perf_counter(); loop(500, coll); dt = (time.perf_counter() - t0) / 500
    print(f"collective={coll}: {dt * 1e6:.1f} us per tick (host loop incl. GPU time)")
t0 = time.perf_counter(); eng.run(500); torch.cuda.synchronize(); print(f"engine.run(500): {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per tick")
eng.close(); dist.destroy_process_group()
