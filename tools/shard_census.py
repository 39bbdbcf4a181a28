"""Dev probe (GPU, experiments build): per-workgroup stamps of the symmetric pair kernel of ONE replayed rank (c5 at G = 8, c4 at G = 4;
plain tick, geometry kernel on its own launch because stamps are on) -> how long workgroups live, how the launch drains.
    make -C carla-social-force-model_amd/csrc EXPERIMENTS=1 OUT=$PWD/variants/libsfm_exp.so;  python tools/shard_census.py [rank] [c5|c4] [whole]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SFM_LIB_PATH", os.path.join(ROOT, "variants", "libsfm_exp.so"))
out = os.path.join(ROOT, "gpurun_out", "shard_stamps.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["SFM_STAMPS"] = out
import numpy as np
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine, block_layout, equal_bounds
r = int(sys.argv[1]) if len(sys.argv) > 1 else 1
name = sys.argv[2] if len(sys.argv) > 2 else "c5"
whole = len(sys.argv) > 3
G = 8 if name == "c5" else 4
sc, forces = scenarios.baseline_scenario(name)
eng = HipShardEngine(default_sfm_config(forces), 0.05)
eng.set_partition(*block_layout(G))
n, n_pad = eng.load(sc)
b = equal_bounds(n, n_pad, G)
e = eng.engine
e.tick()
if not whole:
    e.set_shard(b[r], b[r + 1])
for _ in range(4):
    e.tick()
eng.synchronize()
items = e.pair_work()[0]
print(f"{name} {'whole crowd' if whole else f'rank {r} of {G}'}: {items} tile-pair items, {e.pair_work()[1] / 1e6:.1f} M terms, {e.kernel_variant()}")
eng.close()
d = np.loadtxt(out, dtype=np.uint64)
s = (d[:, 0] - d[:, 0].min()).astype(float) * 0.01
t = (d[:, 1] - d[:, 0].min()).astype(float) * 0.01
life = t - s
print(f"workgroups stamped {len(d)}; launch span {t.max():.1f} us; items per workgroup {items / len(d):.2f}")
print("lifetime us percentiles 0/10/50/90/99/100:", " ".join(f"{v:.2f}" for v in np.percentile(life, [0, 10, 50, 90, 99, 100])))
print("start    us percentiles 0/10/50/90/99/100:", " ".join(f"{v:.1f}" for v in np.percentile(s, [0, 10, 50, 90, 99, 100])))
print("end      us percentiles 0/10/50/90/99/100:", " ".join(f"{v:.1f}" for v in np.percentile(t, [0, 10, 50, 90, 99, 100])))
for q in np.linspace(0.1, 1.0, 10) * t.max():
    print(f"  resident workgroups at {q:6.1f} us: {int(((s <= q) & (t > q)).sum())}")
print(f"sum of lifetimes / (span x 2048 slots) = {life.sum() / (t.max() * 2048):.2f}")
