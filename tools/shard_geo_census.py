"""Dev probe (GPU, experiments build with SFM_GEO_STAMPS_MERGED): phase stamps of the border / obstacle workgroups INSIDE the pair launch of
one replayed rank of c5 at G = 8 -- when they start, how long they live, when the last one is done.   python tools/shard_geo_census.py [rank]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
out = os.path.join(ROOT, "gpurun_out", "shard_geo_stamps.txt")
os.environ["SFM_GEO_STAMPS"] = out
os.environ["SFM_GEO_STAMPS_MERGED"] = "1"
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine, block_layout, equal_bounds
r = int(sys.argv[1]) if len(sys.argv) > 1 else 1
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc, forces = scenarios.baseline_scenario("c5")
eng = HipShardEngine(default_sfm_config(forces), 0.05)
if G > 1:
    eng.set_partition(*block_layout(G))
n, n_pad = eng.load(sc)
e = eng.engine
e.tick()
if G > 1:
    b = equal_bounds(n, n_pad, G)
    e.set_shard(b[r], b[r + 1])
for _ in range(5):
    e.tick()
eng.synchronize()
print(e.kernel_variant())
eng.close()
d = np.loadtxt(out, dtype=np.uint64)
d = d[d[:, 3] > 0].astype(np.float64) * 10.0          # 100 MHz -> ns
d = d[d[:, 0] > d[:, 0].max() - (2.0e5 if G > 1 else 9.0e5)]          # the last tick's launch only (the buffer keeps older stamps)
t0 = d[:, 0].min()
find, scan, tail, life = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 3] - d[:, 0]
pc = lambda v: (np.percentile(v, [10, 50, 90, 100]) / 1e3).round(1).tolist()
print(f"rank {r} of {G}: {len(d)} geometry workgroups; starts p50/p90/max {pc(d[:, 0] - t0)[1:]} us after the first; ends p50/p90/max {pc(d[:, 3] - t0)[1:]} us")
print(f"   us p10/p50/p90/max: find {pc(find)}  scan {pc(scan)}  tail {pc(tail)}  lifetime {pc(life)}   sum of lifetimes {life.sum() / 1e3:.0f} us")
h, _ = np.histogram((d[:, 0] - t0) / 1e3, bins=np.arange(0, 170, 10))
print("   starts per 10 us:", h.tolist())
h, _ = np.histogram((d[:, 3] - t0) / 1e3, bins=np.arange(0, 170, 10))
print("   ends   per 10 us:", h.tolist())
