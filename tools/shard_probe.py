"""Per-rank compute time of the sharded path, replayed on one GPU: the whole crowd is resident on one handle and rank r of G
is replayed by sfm_set_shard(rows of r).  No collective here -- this is the compute a rank does between two all-gathers.

    python tools/shard_probe.py c5 [G ...]          default G = 1 2 4 8;  LAYOUT=gx,gy picks the rank blocks (default: stepper.block_layout)

For every G (a fresh handle each): the equal split, then `PROBE_ITER` rounds of stepper.balanced_bounds on the engine's work
measure, each followed by the re-pack that cuts the blocks at the new boundaries (what ShardedStepper does at each re-pack).
Prints per-rank tick time (wall clock over PROBE_REPS = 40 ticks, the library's per-call event bracket off), the max over ranks and the strong-scaling bound
t(G=1) / max_r t_r(G) that the compute alone allows."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from carla_social_force_model_amd import scenarios                                     # noqa: E402
from carla_social_force_model_amd.config import default_sfm_config                     # noqa: E402
from carla_social_force_model_amd.stepper import HipShardEngine, balanced_bounds, block_layout, equal_bounds   # noqa: E402

ITER = int(os.environ.get("PROBE_ITER", "3"))
REPS = int(os.environ.get("PROBE_REPS", "40"))
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
Gs = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
sc, forces = scenarios.baseline_scenario(name)
t1 = None
for G in Gs:
    layout = block_layout(G, [int(v) for v in os.environ["LAYOUT"].split(",")] if os.environ.get("LAYOUT") else None) if G > 1 else None
    eng = HipShardEngine(default_sfm_config(forces), 0.05)
    if layout:
        eng.set_partition(*layout)
    n, n_pad = eng.load(sc)
    e = eng.engine

    def measure(lo, hi):
        e.set_shard(lo, hi)
        e.set_timing(True)
        e.tick()                                  # settles the launch shape / lists for this shard (state not advanced)
        e.tick(); e.tick(); e.tick()
        eng.synchronize()
        _, k, l = e.timing()
        # wall clock over REPS ticks with the library's own HIP-event bracket of every call switched off (round 4: the bracket is two event
        # records, ~10 us per tick -- 15 % of a c4 rank's tick -- and no real run has it on: bench.py and the stepper switch it off too)
        e.set_timing(False)
        e.tick(); eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(REPS):
            e.tick()
        eng.synchronize()
        us = (time.perf_counter() - t0) / REPS * 1e6
        return us, l / k, eng.work(), e.pair_work()[1]

    b = equal_bounds(n, n_pad, G)
    for it in range(ITER + 1 if G > 1 else 1):
        if layout:
            e.set_shard(0, n)
            eng.set_partition(*layout, b)
            e.resort()                            # the blocks are cut at the current boundaries
            e.tick()                              # whole crowd once: both halves of the ping-pong state are in the new row order
                                                  # (a real run's exchange rewrites the other ranks' rows every tick; here nothing does)
        res = [measure(b[r], b[r + 1]) for r in range(G)]
        us = [x[0] for x in res]
        tag = "equal split" if it == 0 else f"balanced, round {it}"
        if G == 1:
            t1 = us[0]
        print(f"{name} G={G} layout {layout} {tag:20s} max {max(us):8.1f} us  mean {sum(us) / G:8.1f} us  speed-up bound "
              f"{t1 / max(us) if t1 else float('nan'):5.2f}x  launches/tick {res[0][1]:.1f}  kernel {e.kernel_variant()}", flush=True)
        print("      per rank us: " + " ".join(f"{u:7.1f}" for u in us), flush=True)
        print("      rows       : " + " ".join(f"{b[r + 1] - b[r]:7d}" for r in range(G)), flush=True)
        # Moussaid terms the rank's pair kernel evaluated (own-own pairs once, pairs with another rank's tile one-sided: the other
        # rank evaluates them too) against an eighth of what the whole crowd on one GPU evaluates
        terms = [x[3] for x in res]
        if G == 1:
            terms1 = terms[0]
        print("      pair terms : " + " ".join(f"{v / 1e6:7.1f}" for v in terms) + f"  M   (whole crowd / G = {terms1 / G / 1e6:.1f} M)", flush=True)
        if G > 1 and it < ITER:
            b = balanced_bounds(b, [x[2] for x in res], n)
    eng.close()
