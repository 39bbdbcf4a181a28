#!/usr/bin/env python3
"""Benchmark of the SFM stepper hot path on MI355X.  One JSON line on rank 0.

  python bench.py --gpus 1 --steps K --warmup W          # 1 GPU: BASELINE config 2 (N = 4096, ped + accel)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node G ... bench.py --gpus G ...

A "step" is one SFM tick (one pass of the hot path over the whole crowd): forces -> capped velocity ->
position update -> arrival / next waypoint, all on the device, state resident in HBM before the timed region.
Metric: SFM ticks/s (BASELINE.json "SFM ticks/s and ns/pedestrian-pair"); ns/pair is reported beside it.

Workloads (--workload overrides the default):
  G = 1  -> c2: N = 4 096, pedestrian_force + acceleration_force        (the config the metric is quoted on)
  G > 1  -> c5: N = 262 144 + 2 000 borders + 256 static + 512 dynamic obstacles, all forces, STRONG scaling:
            pedestrians sharded by row (whole 64-row tiles of a spatial packing), one RCCL all-gather of the packed
            state per tick, plus a gather of the waypoint arrays and an identical re-pack on every rank each 64 ticks.
            Rank 0 also times 40 ticks of the SAME workload on one GPU before the run ("single_gpu_same_workload")
            so the strong-scaling speed-up can be read off one line.
  SFM_BENCH_REHEARSAL=1 (development only): every rank on cuda:0, collectives over gloo -- runs the multi-rank flow end to
            end on a one-GPU box; its timings mean nothing.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def algorithmic_bytes(sc, forces, n_local, sample=2048, seed=0):
    """Algorithmic bytes of one launch over n_local pedestrian rows:
    16 B per ordered pair (the {x,y,vx,vy} operand stream) + 44 B per own row (28 read + 16 written)
    + 12 B per (pedestrian, border) cull test + 8 B per scanned border point
    + 8 B / 16 B per (pedestrian, static / dynamic obstacle) cull test + 8 B per scanned ring point.
    The scanned-point terms are estimated on a sample of pedestrians."""
    n = sc.n
    b = 0.0
    if "pedestrian_force" in forces:
        b += 16.0 * n_local * (n - 1)
    b += 44.0 * n_local
    rng = np.random.default_rng(seed)
    idx = rng.choice(n, size=min(sample, n), replace=False)
    xy = sc.loc[idx, :2]

    def scanned(centers, radii, counts):
        if len(counts) == 0:
            return 0.0
        tot = 0.0
        for s in range(0, len(idx), 256):
            d = np.linalg.norm(xy[s:s + 256, None, :] - centers[None, :, :], axis=-1)
            tot += float(((d < radii[None, :]) * counts[None, :]).sum())
        return tot / len(idx)

    if "border_force" in forces and len(sc.borders):
        cnt = np.array([len(p) for p in sc.borders], dtype=np.float64)
        b += n_local * (12.0 * len(sc.borders) + 8.0 * scanned(sc.border_centers, sc.border_lengths, cnt))
    if "static_obstacle_force" in forces and len(sc.static_obstacles):
        c = np.array([o[0] for o in sc.static_obstacles])
        cnt = np.array([len(o[1]) for o in sc.static_obstacles], dtype=np.float64)
        b += n_local * (8.0 * len(cnt) + 8.0 * scanned(c, np.full(len(cnt), 20.0), cnt))
    if "dynamic_obstacle_force" in forces and len(sc.dynamic_obstacles):
        c = np.array([o[0] for o in sc.dynamic_obstacles])
        cnt = np.array([len(o[1]) for o in sc.dynamic_obstacles], dtype=np.float64)
        b += n_local * (16.0 * len(cnt) + 8.0 * scanned(c, np.full(len(cnt), 50.0), cnt))
    return b


def cpu_baseline(sc, forces, cfg, budget_s=15.0):
    """The oracle's C/OpenMP port timed on this node's host cores, on a bounded sample of the workload:
    whole ticks when one fits the budget, otherwise a contiguous block of pedestrian rows of ONE tick
    (all j are still visited) scaled to a full tick."""
    from oracle import c_oracle
    from oracle import sfm_oracle as O
    prm = O.OracleParams.from_config(cfg)
    geom = O.Geometry(sc.borders, sc.border_centers, sc.border_lengths, sc.static_obstacles, sc.dynamic_obstacles,
                      sc.dynamic_vel)
    cores = min(os.cpu_count() or 1, c_oracle.max_threads())
    crossing = np.zeros(sc.n, bool)
    args = (sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, crossing, geom, prm, 0.05)
    probe = min(sc.n, 64 * cores)
    c_oracle.tick(*args, rows=(0, min(sc.n, cores)), nthreads=cores)      # spin the thread pool up
    t0 = time.perf_counter()
    c_oracle.tick(*args, rows=(0, probe), nthreads=cores)
    t_probe = time.perf_counter() - t0
    est_tick = t_probe * sc.n / probe
    if est_tick <= budget_s:
        reps, t0 = 0, time.perf_counter()
        while reps == 0 or time.perf_counter() - t0 < budget_s:      # whole ticks until the budget is used
            c_oracle.tick(*args, nthreads=cores)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        sample = f"{reps} full ticks of {len(forces)} forces at N={sc.n}"
    else:
        rows = max(probe, int(sc.n * budget_s / est_tick))
        t0 = time.perf_counter()
        c_oracle.tick(*args, rows=(0, rows), nthreads=cores)
        dt = (time.perf_counter() - t0) * sc.n / rows
        sample = f"rows 0..{rows} of one tick at N={sc.n} (all j visited), scaled to a full tick"
    return {"value": 1.0 / dt, "unit": "ticks/s", "cores": cores, "kind": "port",
            "sample": sample + "; oracle/sfm_oracle.c, float64, OpenMP", "ns_per_pair": dt * 1e9 / (sc.n * (sc.n - 1.0))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="auto", choices=["auto", "c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from carla_social_force_model_amd import scenarios
    from carla_social_force_model_amd.config import default_sfm_config
    from carla_social_force_model_amd.stepper import HipShardEngine, ShardedStepper

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # rehearsal on a one-GPU box (never used by the driver): SFM_BENCH_REHEARSAL=1 puts every rank on cuda:0 and moves the
    # collectives over gloo, so that the multi-rank flow (shards, exchange, re-pack protocol) runs end to end
    rehearsal = os.environ.get("SFM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))

    name = args.workload if args.workload != "auto" else ("c2" if world == 1 else "c5")
    sc, forces = scenarios.baseline_scenario(name)
    cfg = default_sfm_config(forces)
    dt = 0.05

    single_ref = None
    if world > 1 and rank == 0:
        # the same workload on ONE GPU, a few ticks, so the strong-scaling speed-up is on this line
        e1 = HipShardEngine(cfg, dt, device=local)
        s1 = ShardedStepper(e1, sc)
        s1.step(10)
        e1.synchronize()
        t0 = time.perf_counter()
        s1.step(40)
        e1.synchronize()
        single_ref = 40.0 / (time.perf_counter() - t0)
        e1.close()

    eng = HipShardEngine(cfg, dt, device=local)
    st = ShardedStepper(eng, sc, rank=rank, world=world)
    lo, hi = st.local_rows()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    st.exchange()                 # brings the communicator up outside the timed region (idempotent on a fresh state)
    st.step(args.warmup)
    barrier()
    t0 = time.perf_counter()
    st.step(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # whole tick on the launch stream (HIP events, no collective in between) ...
    reps = min(max(args.steps, 1), 200 if sc.n <= 16384 else 10)
    eng.engine.run(reps, redraw=True)
    ev_ms, ev_ticks, ev_launches = eng.engine.timing()
    tick_us = ev_ms * 1e3 / max(ev_ticks, 1)
    # ... and the dominant kernel (the pedestrian-pair kernel) on its own, same stream, HIP events
    kernel_us = eng.engine.profile_dominant_kernel(reps)
    variant = eng.engine.kernel_variant()
    dominant = "sfm_pair_sym_kernel" if "sym" in variant else variant
    alg = 16.0 * (hi - lo) * (sc.n - 1.0) if "pedestrian_force" in forces else 0.0
    if "sym" not in variant:
        alg += 44.0 * (hi - lo)                      # the ordered kernel also reads / writes the own rows
    alg_tick = algorithmic_bytes(sc, forces, hi - lo)
    achieved = alg / (kernel_us * 1e-6) / 1e9
    traffic = None
    try:                                              # HBM-side bytes per launch from the committed PMC passes
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            traffic = json.load(f).get(name, {}).get(dominant, {}).get("hbm_bytes")
    except (OSError, ValueError):
        pass
    # The binding resource of the pair kernel is VALU issue, not HBM: price it too.  One systolic
    # step of a wave = 64 pairs = 47 plain + 6 DPP + 2 v_bfi + v_cmp/v_cndmask + 5 transcendental ops; with the issue
    # costs measured on MI355X (tools/valu_microbench.hip: 2.2 / 4.4 / 4.4 / 8.7 / 9.3 cycles) that is ~200 cycles.
    valu = None
    if "sym" in variant and "pedestrian_force" in forces and world == 1 and sc.n < 8192:      # no tile cutoff: every pair is evaluated
        n_t = (sc.n + 63) // 64
        wave_steps = 64.0 * n_t * (n_t - 1) / 2 + 32.0 * n_t
        simds, clock_hz, model = 1024, 2.4e9, 200.0
        spent = kernel_us * 1e-6 * clock_hz * simds / wave_steps
        valu = {"bound": "valu_issue", "model_cycles_per_wave_step": model, "achieved_cycles_per_wave_step": spent,
                "frac": model / spent, "wave_steps_per_launch": wave_steps, "simds": simds, "clock_hz": clock_hz,
                "note": "cycles at the 2.4 GHz peak engine clock (the in-kernel clock read 2.08 GHz under this load: frac ~0.75 at that clock)"}
    barrier()

    if rank == 0:
        n = sc.n
        ticks_s = args.steps / elapsed
        out = {
            "metric": "sfm_ticks_per_s", "value": ticks_s, "unit": "ticks/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name}: N={n} pedestrians, forces={'+'.join(f.replace('_force', '') for f in forces)}, "
                                   f"borders={len(sc.borders)}, static={len(sc.static_obstacles)}, dynamic={len(sc.dynamic_obstacles)}, "
                                   f"dt={dt}", "n_pedestrians": n, "sharding": f"rows/{world}" + (", 1 all-gather/tick" if world > 1 else ""),
                       "kernel": variant},
            "ns_per_pair": elapsed / args.steps * 1e9 / (n * (n - 1.0)),
            "pairs_per_s": n * (n - 1.0) * ticks_s,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": dominant, "kernel_us": kernel_us, "algorithmic_bytes_per_launch": alg,
                         "tick_us": tick_us, "launches_per_tick": ev_launches / max(ev_ticks, 1),
                         "algorithmic_bytes_per_tick": alg_tick,
                         "note": "dominant kernel = the pedestrian-pair kernel: 16 B per ordered pair x the pairs one launch "
                                 "covers / its HIP-event time; traffic = HBM bytes per launch from rocprofv3 PMC passes "
                                 "(profiles/). The operand stream is served on-chip; the binding resource is VALU issue (DESIGN.md 3.4)"},
        }
        if valu is not None:
            out["roofline"]["valu_issue"] = valu
        if "pedestrian_force" in forces:
            # SURVEY.md section 8d, "algorithmic flops": ~65 fp32 VALU ops per ORDERED pedestrian pair against the
            # 7.9e13 lane-ops/s of the chip (256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz); the symmetric kernel evaluates an
            # unordered pair once, which this definition credits as two
            ops = 65.0 * n * (n - 1.0) * ticks_s
            out["roofline"]["valu_algorithmic"] = {"ops_per_ordered_pair": 65.0, "achieved_ops_per_s": ops, "peak_ops_per_s": 7.9e13,
                                                   "frac": ops / 7.9e13 / world}
        if single_ref is not None:
            out["single_gpu_same_workload"] = {"value": single_ref, "unit": "ticks/s", "speedup": ticks_s / single_ref}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, forces, cfg, args.cpu_budget)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
