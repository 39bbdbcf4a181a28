"""Per-rank compute time of the sharded path, emulated on one GPU: the whole crowd is resident on one handle and rank r of G
is replayed by sfm_set_shard(rows of r).  No collective here -- this is the compute a rank does between two all-gathers.

    python tools/shard_probe.py c5 [G ...]          default G = 1 2 4 8

For every G: the equal split, then `ITER` rounds of stepper.balanced_bounds on the pair kernel's evaluated terms (what
ShardedStepper does at each re-pack).  Prints per-rank tick time (HIP events, mean of 3 ticks), the max over ranks and
the strong-scaling bound  t(G=1) / max_r t_r(G)  that the compute alone allows."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios                                     # noqa: E402
from carla_social_force_model_amd.config import default_sfm_config                     # noqa: E402
from carla_social_force_model_amd.stepper import HipShardEngine, balanced_bounds, equal_bounds   # noqa: E402

ITER = int(os.environ.get("PROBE_ITER", "4"))
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
Gs = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
sc, forces = scenarios.baseline_scenario(name)
eng = HipShardEngine(default_sfm_config(forces), 0.05)
n, n_pad = eng.load(sc)
e = eng.engine


def measure(lo, hi):
    e.set_shard(lo, hi)
    e.tick()                                  # settles the launch shape / lists for this shard
    e.run(3, redraw=False)
    eng.synchronize()
    ms, t, l = e.timing()
    return ms / t * 1e3, l / t, eng.work()


t1 = None
for G in Gs:
    b = equal_bounds(n, n_pad, G)
    for it in range(ITER + 1 if G > 1 else 1):
        res = [measure(b[r], b[r + 1]) for r in range(G)]
        us = [x[0] for x in res]
        tag = "equal split" if it == 0 else f"balanced, round {it}"
        if G == 1:
            t1 = us[0]
        print(f"{name} G={G} {tag:20s} max {max(us):8.1f} us  mean {sum(us) / G:8.1f} us  speed-up bound {t1 / max(us) if t1 else float('nan'):5.2f}x  "
              f"launches/tick {res[0][1]:.1f}  kernel {e.kernel_variant()}", flush=True)
        print("      per rank us: " + " ".join(f"{u:7.1f}" for u in us), flush=True)
        print("      rows       : " + " ".join(f"{b[r + 1] - b[r]:7d}" for r in range(G)), flush=True)
        if G > 1 and it < ITER:
            b = balanced_bounds(b, [x[2] for x in res], n)
eng.close()
