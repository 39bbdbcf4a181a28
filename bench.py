#!/usr/bin/env python3
"""Benchmark of the SFM stepper hot path on MI355X.  One JSON line on rank 0.

  python bench.py --gpus 1 --steps K --warmup W          # 1 GPU: BASELINE config 2 (N = 4096, ped + accel)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node G ... bench.py --gpus G ...

A "step" is one SFM tick (one pass of the hot path over the whole crowd): forces -> capped velocity ->
position update -> arrival / next waypoint, all on the device, state resident in HBM before the timed region.
Metric: SFM ticks/s (BASELINE.json "SFM ticks/s and ns/pedestrian-pair"); ns/pair is reported beside it.

Timing protocol.  After W untimed warm-up steps a WINDOW is exactly K steps bracketed by a
barrier + torch.cuda.synchronize() on both sides, its duration the MAX over ranks.  At least 5 windows are run, and
more until they add up to >= 3 s (--min-seconds); `value` = K / the MEDIAN window; `windows`, `min`, `max` and the
mean over all windows are on the line.  `steps` / `warmup` echo the arguments.

Workloads (--workload overrides the default, and the SAME workload can be run at every G):
  G = 1  -> c2: N = 4 096, pedestrian_force + acceleration_force        (the config the metric is quoted on)
  G > 1  -> c5: N = 262 144 + 2 000 borders + 256 static + 512 dynamic obstacles, all forces, STRONG scaling:
            pedestrians sharded by row (whole 64-row tiles of a spatial packing, shard sizes balanced by pair work),
            one RCCL all-gather of the packed state per tick, plus a gather of the waypoint arrays and an identical
            re-pack on every rank each 64 ticks.  Rank 0 first times the SAME workload on ONE GPU with the same
            protocol ("single_gpu_same_workload"), so the strong-scaling speed-up is on the line;
            `--gpus 1 --workload c5` prints that N = 1 point as a line of its own.
  --rehearsal (development only): every rank on cuda:0, collectives over gloo -- runs the multi-rank flow end to
            end on a one-GPU box; its timings mean nothing.

Roofline record.  The dominant kernel is the pedestrian-pair kernel.  Its binding resource is VALU issue (DESIGN.md
3.6): `roofline.bound = "valu_issue"`, achieved = VALU wave-instructions per launch (SQ_INSTS_VALU, mean per launch from
the rocprofv3 --pmc passes summarised in profiles/r04_pmc_summary.csv, a tracked file whose header carries the git blob id of
sfm_kernels.hip at collection time: `roofline.counters_stale` says whether the loaded build differs) / the kernel's own launch
duration measured live with HIP events on the launch stream, against 1024 SIMDs x 0.5 wave-instr/clk x 2.4 GHz.
`roofline.hbm_algorithmic` keeps SURVEY.md 8d's yardstick (16 B per ordered pair / launch time against 8 TB/s) and
`roofline.traffic` the HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (same file).
"""
import argparse
import csv
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
SIMDS, CLOCK_HZ = 1024, 2.4e9
VALU_PEAK_GINSTR = SIMDS * 0.5 * CLOCK_HZ / 1e9     # v_fma_f32 (wave64): 2 cycles per SIMD (MI355X_MICROARCH.md, cycle constants)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r04_pmc_summary.csv")
KERNEL_SOURCE = os.path.join(ROOT, "carla-social-force-model_amd", "csrc", "sfm_kernels.hip")   # beside the loaded libsfm_hip.so
# The counter passes (tools/pmc_run.sh -> tools/pmc_ticks.py) run this many ticks from the uploaded scenario.  With a cutoff the
# pair kernel's instruction count depends on where the crowd has got to, so the launch duration that goes with those
# counters is measured at the same point (a second handle, same ticks), not at the end of the timed run.
PMC_STATE_TICKS = {"c1": 40, "c2": 40, "c3": 40, "c4": 40, "c5": 12}


def algorithmic_bytes(sc, forces, n_local, sample=2048, seed=0):
    """Algorithmic bytes of one tick over n_local pedestrian rows:
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
    """Two CPU legs timed on this node's host cores, on bounded samples of the workload:
      * the oracle's C/OpenMP float64 port on all cores: whole ticks when one fits the budget, otherwise a contiguous
        block of pedestrian rows of ONE tick (all j are still visited) scaled to a full tick;
      * the oracle's NumPy float64 restatement with DENSE (N, N, 3) temporaries on one core -- algorithm-equivalent to the
        reference's forces.py / stateutils.py (which cannot travel to this box) -- on the first min(N, 2048) pedestrians."""
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
    out = {"value": 1.0 / dt, "unit": "ticks/s", "cores": cores, "kind": "port",
           "sample": sample + "; oracle/sfm_oracle.c, float64, OpenMP", "ns_per_pair": dt * 1e9 / (sc.n * (sc.n - 1.0))}
    # NumPy float64, dense temporaries, one core: the reference's own algorithm shape
    m = min(sc.n, 2048)
    sub = np.arange(m)
    try:
        import threadpoolctl
        limiter = threadpoolctl.threadpool_limits(1)
    except Exception:                                                # pragma: no cover
        limiter = None
    with np.errstate(all="ignore"):
        reps, t0 = 0, time.perf_counter()
        while reps == 0 or (time.perf_counter() - t0 < min(budget_s, 6.0) and reps < 5):
            f = np.zeros((m, 3))
            if "acceleration_force" in forces:
                f = f + O.acceleration_force(sc.loc[sub], sc.vel[sub], sc.waypoint[sub], sc.target_speed[sub], prm.tau)
            if "pedestrian_force" in forces:
                f = f + O.pedestrian_force(sc.loc[sub], sc.vel[sub], sc.radius[sub], prm.ped, prm.use_ped_radius, chunk=m,
                                           diagnostics=False)[0]
            O.new_velocities(sc.vel[sub], f, sc.target_speed[sub], 0.05)
            reps += 1
        dn = (time.perf_counter() - t0) / reps
    if limiter is not None:
        limiter.restore_original_limits()
    out["numpy_dense_1core"] = {"value": 1.0 / dn, "unit": "ticks/s", "cores": 1, "kind": "port", "n_pedestrians": m,
                                "ns_per_pair": dn * 1e9 / (m * (m - 1.0)),
                                "sample": f"{reps} ticks of acceleration + pedestrian force + cap on the first {m} pedestrians; "
                                          "oracle/sfm_oracle.py, float64, dense (N,N,3) temporaries like forces.py:74-117"}
    return out


def git_blob_sha1(path):
    """The git blob id of a file's content (sha1 of "blob <size>\\0" + bytes): what `git hash-object` prints, computed here
    because the GPU box has no .git directory."""
    import hashlib
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_counters(workload, kernel):
    """Mean per-launch counter values of `kernel` on `workload` from the tracked PMC summary (tools/pmc_summarize.py), and
    whether that summary was collected on another version of the kernels than the one next to the loaded library: its
    header comment carries the git blob id of sfm_kernels.hip at collection time (`counters_stale`)."""
    out, blob = {}, None
    try:
        with open(PMC_SUMMARY) as f:
            lines = f.readlines()
        for ln in lines:
            if ln.startswith("#") and "sfm_kernels.hip" in ln and "git-blob" in ln:
                blob = ln.split("git-blob")[1].split()[0]
        for row in csv.DictReader(ln for ln in lines if not ln.startswith("#")):
            if row["workload"] == workload and row["kernel"] == kernel:
                out[row["counter"]] = float(row["mean_per_launch"])
    except OSError:
        pass
    here = git_blob_sha1(KERNEL_SOURCE)
    stale = None if (blob is None or here is None) else (blob != here)
    return out, {"counters_kernel_source_blob": blob, "loaded_kernel_source_blob": here, "counters_stale": stale}


def timed_windows(step, barrier, reduce_max, steps, min_windows, min_seconds, max_windows=1000000, wall_factor=8.0):
    """Windows of exactly `steps` steps, each bracketed by barrier() on both sides; stops after >= min_windows windows
    adding up to >= min_seconds of TIMED time (round 4: the driver's --steps 20 windows of 0.3 ms need ~10 000 of them; the old cap of
    5000 windows stopped that run at 1.7 s).  reduce_max makes the duration (and therefore the stop decision) identical on every rank.
    The only other stop is a wall-clock guard -- wall_factor x min_seconds spent in this loop, barriers included -- which is decided
    from the reduced times plus a per-window allowance, so every rank takes it at the same window."""
    times = []
    total = 0.0
    while len(times) < min_windows or (total < min_seconds and len(times) < max_windows
                                       and total + 2.0e-4 * len(times) < wall_factor * min_seconds):
        barrier()
        t0 = time.perf_counter()
        step(steps)
        barrier()
        times.append(reduce_max(time.perf_counter() - t0))
        total += times[-1]
    return times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="auto", choices=["auto", "c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--windows", type=int, default=5, help="minimum number of timed windows of --steps steps")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="keep adding windows until they total this long (SURVEY.md 8d: >= 3 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-ref", action="store_true", help="G > 1: skip the one-GPU run of the same workload")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    ap.add_argument("--rehearsal", action="store_true", help="development: all ranks on cuda:0, collectives over gloo (timings mean nothing)")
    args = ap.parse_args()
    # stdout carries exactly ONE line, the JSON record: libraries that print at start-up (RCCL: "Librccl path ...", gloo's rank
    # banner in rehearsals) write to file descriptor 1 from C, so it is pointed at stderr for the run and the record goes to a
    # duplicate of the original
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL between processes); before HIP comes up
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
    # rehearsal on a one-GPU box (never used by the driver): --rehearsal puts every rank on cuda:0 and moves the
    # collectives over gloo, so that the multi-rank flow (shards, exchange, re-pack protocol) runs end to end
    rehearsal = args.rehearsal
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

    def local_barrier():
        torch.cuda.synchronize()

    single_ref = None
    if world > 1 and rank == 0 and not args.no_single_ref:
        # the same workload on ONE GPU, same protocol (K-step windows incl. the periodic re-pack), so the strong-scaling
        # speed-up is on this line
        e1 = HipShardEngine(cfg, dt, device=local)
        s1 = ShardedStepper(e1, sc)
        e1.engine.set_timing(False)
        s1.step(args.warmup)
        t1 = timed_windows(s1.step, local_barrier, lambda v: v, args.steps, max(3, args.windows), args.min_seconds)
        single_ref = {"value": args.steps / statistics.median(t1), "unit": "ticks/s", "windows": len(t1),
                      "min": args.steps / max(t1), "max": args.steps / min(t1)}
        e1.close()

    eng = HipShardEngine(cfg, dt, device=local)
    st = ShardedStepper(eng, sc, rank=rank, world=world)
    lo, hi = st.local_rows()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if rehearsal else f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    st.exchange()                 # brings the communicator up outside the timed region (idempotent on a fresh state)
    eng.engine.set_timing(False)  # the library's own HIP-event bracket of every call (sfm_get_timing) stays out of the timed windows:
    st.step(args.warmup)          # two event records per call, ~11 us -- 3 % of a 20-tick window of c2
    if world > 1 and st.split_mode == "auto":
        st.step(24)               # (a short warm-up: let the stepper finish choosing between the split and the plain tick)
    times = timed_windows(st.step, barrier, reduce_max, args.steps, args.windows, args.min_seconds)
    elapsed = statistics.median(times)

    # the crowd must still be a crowd: a NaN or a runaway pedestrian anywhere in this rank's rows fails the run loudly
    _loc, _vel, _ = eng.engine.state()
    _own = ~np.isnan(_vel[:, 0]) if world > 1 else np.ones(sc.n, bool)
    _speed = np.linalg.norm(_vel[_own, :2], axis=1)
    if not (np.isfinite(_loc[_own, :2]).all() and np.isfinite(_speed).all() and _speed.max() <= 1.3 * float(sc.target_speed.max()) * 1.001):
        raise SystemExit(f"bench.py: state check failed after the timed run on rank {rank} (non-finite or over-speed pedestrians)")

    # whole tick on the launch stream (HIP events, no collective in between) ...
    reps = min(max(args.steps, 1), 200 if sc.n <= 16384 else 10)
    if world == 1:
        # (the state check above left the GPU idle for a moment: a few hundred untimed ticks first, so that the event-timed run and the
        #  profiled launches behind `roofline` see the clocks the timed windows saw -- round 4: without them a 200-tick burst read 7 % slow)
        eng.engine.run(max(3 * reps, 600 if sc.n <= 16384 else 0), redraw=True)
    eng.engine.set_timing(True)
    # (a rank of a sharded run gets no exchange here: one tick only, so that the other ranks' rows it reads are one tick old at most)
    eng.engine.run(reps if world == 1 else 1, redraw=True)
    ev_ms, ev_ticks, ev_launches = eng.engine.timing()
    tick_us = ev_ms * 1e3 / max(ev_ticks, 1)
    variant = eng.engine.kernel_variant()
    fused = "fused" in variant                   # one launch per tick: the pair kernel with the integration as its prologue
    sym = "sym" in variant or fused
    pair_items = pair_terms = None
    if sym and "pedestrian_force" in forces:
        pair_items, pair_terms = eng.engine.pair_work()             # of the last tick above
    # ... and the dominant kernel (the pedestrian-pair kernel) ALONE, same stream, HIP events (the tile-pair list of the
    # cutoff configs is built once, outside the timed launches)
    kernel_us = eng.engine.profile_dominant_kernel(reps)
    kernel_us_end = kernel_us
    state_note = "end of the timed run"
    if world == 1 and sc.n > 4096 and "pedestrian_force" in forces:
        # a cutoff is on: the launch that goes with the committed counters is the one PMC_STATE_TICKS ticks after the upload
        e2 = HipShardEngine(cfg, dt, device=local)
        e2.load(sc)
        e2.run(PMC_STATE_TICKS[name])
        kernel_us = e2.engine.profile_dominant_kernel(reps)
        if sym:
            pair_items, pair_terms = e2.engine.pair_work()
        e2.close()
        state_note = f"{PMC_STATE_TICKS[name]} ticks after the upload (the state of the committed counter passes)"
    dominant = "sfm_fused_tick_kernel" if fused else "sfm_pair_sym_kernel" if sym else "sfm_tick_kernel"
    alg = 16.0 * (hi - lo) * (sc.n - 1.0) if "pedestrian_force" in forces else 0.0
    if not sym or fused:
        alg += 44.0 * (hi - lo)                      # the ordered kernel also reads / writes the own rows
    alg_tick = algorithmic_bytes(sc, forces, hi - lo)
    alg_gbs = alg / (kernel_us * 1e-6) / 1e9
    # counters of this kernel on this workload (whole crowd on one GPU), mean per launch, from the tracked summary
    pmc, stamp = pmc_counters(name, dominant)
    src = os.path.relpath(PMC_SUMMARY, ROOT)
    traffic = None
    if world == 1 and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:   # KiB; FETCH_SIZE doubled for 16-B/lane reads on gfx950 (MI355X_MICROARCH.md, HBM)
        traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
    hbm_alg = {"bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS,
               "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (traffic / alg) if (traffic and alg) else None,
               "note": "SURVEY.md 8d yardstick: 16 B per ORDERED pair x the pairs one launch covers / its launch time. The operand "
                       "stream is served from registers / L2, so this exceeds the HBM peak as soon as the kernel is fast: it does "
                       "not bind (traffic_over_algorithmic says how little of it reaches HBM)"}
    peak_def = "1024 SIMDs x 0.5 wave64 VALU instr/clk (v_fma_f32 = 2 cycles) x 2.4 GHz"
    if world == 1 and "SQ_INSTS_VALU" in pmc:
        insts = pmc["SQ_INSTS_VALU"]
        ach = insts / (kernel_us * 1e-6) / 1e9
        roof = {"bound": "valu_issue", "achieved": ach, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s",
                "frac": ach / VALU_PEAK_GINSTR, "traffic": traffic,
                "valu_wave_instructions_per_launch": insts,
                "counters": {k: pmc[k] for k in sorted(pmc)}, "counters_source": src, "peak_definition": peak_def}
        if pair_terms:
            roof["valu_instructions_per_64_pair_step"] = insts / (pair_terms / 64.0)
        if "SQ_WAIT_INST_ANY" in pmc and "SQ_WAVE_CYCLES" in pmc and pmc["SQ_WAVE_CYCLES"] > 0:
            roof["wait_inst_any_over_wave_cycles"] = pmc["SQ_WAIT_INST_ANY"] / pmc["SQ_WAVE_CYCLES"]
        mix = {k: pmc[k] for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32") if k in pmc}
        if mix:
            roof["fp32_instruction_mix_per_launch"] = mix
        if fused and pair_terms and name == "c2":
            # Round 4: the packed step makes ONE instruction of two, so the plain count above FALLS while the kernel gets faster (8.66 M
            # wave-instructions per launch in round 3, 5.10 M now).  The pipes are busy for PASSES of 2 cycles: 1 per full-rate fp32
            # instruction, 2 per packed (v_pk_*_f32) or "other" (v_bfi, v_add_f32_dpp) instruction, 4 per transcendental (MI355X_MICROARCH.md
            # cycle table; tools/valu_microbench.hip).  Packed and other instructions of the loop are counted from its ISA (41 and 10 per
            # 128-pair double step, `make asm`), transcendentals by SQ_INSTS_VALU_TRANS_F32 (10 per double step if that pass is missing).
            dsteps = pair_terms / 128.0
            trans = pmc.get("SQ_INSTS_VALU_TRANS_F32", 10.0 * dsteps)
            # `frac` itself: fp32 wave-OPERATIONS per second against the 2-cycle rate -- a packed instruction is two operations (and takes
            # the pipe for two passes), so this is the round-3 definition carried over (every instruction was one operation then: 0.43)
            ops = insts + 41.0 * dsteps
            roof.update({"frac_plain_instructions": roof["frac"], "achieved_plain_instructions": roof["achieved"],
                         "achieved": ops / (kernel_us * 1e-6) / 1e9, "frac": ops / (kernel_us * 1e-6) / 1e9 / VALU_PEAK_GINSTR,
                         "unit": "G fp32 wave-operations/s", "valu_wave_operations_per_launch": ops,
                         "achieved_definition": "(SQ_INSTS_VALU + packed instructions: each v_pk_*_f32 is two wave-wide fp32 operations; 41 per 128-pair "
                                                "double step of the loop, from its ISA) per launch / kernel_us"})
            passes = insts + 41.0 * dsteps + 10.0 * dsteps + 3.0 * trans
            roof["class_weighted"] = {"issue_passes_per_launch": passes, "achieved": passes / (kernel_us * 1e-6) / 1e9, "peak": VALU_PEAK_GINSTR,
                                      "unit": "G two-cycle passes/s", "frac": passes / (kernel_us * 1e-6) / 1e9 / VALU_PEAK_GINSTR,
                                      "definition": "SQ_INSTS_VALU + packed + other + 3 x transcendental instructions per launch / kernel_us "
                                                    "against 1024 SIMDs x 2.4 GHz / 2"}
        if sc.n < 256:
            # a crowd of one tile: ONE launch per tick of a handful of workgroups (the fused tick: a pair workgroup, its geometry
            # workgroups, the vehicles).  Nothing on the chip is busy; what the tick costs is the launch and the geometry workgroup's
            # chain of dependent memory round trips, so the VALU fraction above says little.  The latency floor goes under keys of
            # its own (achieved / peak / frac keep their VALU meaning: higher is better on every workload's line)
            launches = ev_launches / max(ev_ticks, 1)
            floor_us = launches * 1.45               # MI355X_MICROARCH.md price list, row "boundary": a dependent trivial launch
            roof.update({"bound_in_practice": "latency", "latency_floor_us": floor_us, "latency_frac": floor_us / tick_us,
                         "latency_definition": "dependent kernel launches per tick x 1.45 us (the guide's kernel-boundary price); latency_frac = that floor / the measured tick",
                         "dependent_chain": f"{launches:.2f} launches per tick; inside the launch: kernel start -> own rows + previous partial sums -> "
                                            "integrate -> tile box -> polyline records -> points -> LDS combine -> store"})
    elif world > 1 and pair_terms and "VALU_PER_64_PAIR_STEP" in pmc:
        # a rank of a sharded run: no counter pass of its own.  Its VALU work is the Moussaid terms its pair kernel evaluated in the
        # last tick (sfm_get_pair_work: a count kept by the library) x the instructions per 64-term step the committed
        # whole-crowd counter passes measured for this kernel on this workload
        insts = pair_terms / 64.0 * pmc["VALU_PER_64_PAIR_STEP"]
        ach = insts / (kernel_us * 1e-6) / 1e9
        roof = {"bound": "valu_issue", "achieved": ach, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s", "frac": ach / VALU_PEAK_GINSTR,
                "traffic": None, "valu_wave_instructions_per_launch": insts, "counters_source": src, "peak_definition": peak_def,
                "note": "rank 0 of a sharded run: instructions = its evaluated pair terms / 64 x VALU_PER_64_PAIR_STEP of the whole-crowd "
                        "counter passes (no PMC pass is run per rank); kernel_us is this rank's pair kernel alone"}
    else:                                              # no counter pass committed for this workload / rank layout
        roof = {"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s", "frac": None, "traffic": None,
                "peak_definition": peak_def,
                "note": f"no committed counter pass for {dominant} on {name} in {src}: the fraction is not derivable here; "
                        "hbm_algorithmic below is SURVEY.md 8d's yardstick, which does not bind"}
    roof.update(stamp)
    if roof.get("frac") is not None and roof["bound"] == "valu_issue" and roof["frac"] > 1.0:      # never publish a non-physical fraction
        roof.update({"frac": None, "note": (roof.get("note", "") + " [fraction above 1 withheld: counters do not belong to this build]").strip()})
    if world == 1 and sc.n > 4096 and roof.get("frac") is not None:
        # (state-dependent workloads: say what the fraction is a fraction OF -- round-3 verdict, bench hygiene)
        roof["frac_scope"] = ("pair kernel ALONE (no border / obstacle workgroups in its launch: the counter passes run with SFM_PAIR_GEO=0 and "
                              "sfm_profile_dominant_kernel launches sfm_pair_sym_kernel by itself), at the state of the counter passes -- "
                              f"{PMC_STATE_TICKS[name]} ticks after the upload; ms_per_step is the median window of the whole timed run, whose "
                              "crowd moves on (kernel_us_end_of_run)")
    roof.update({"kernel": dominant, "kernel_variant": variant, "kernel_us": kernel_us, "kernel_us_state": state_note,
                 "kernel_us_end_of_run": kernel_us_end, "tick_us": tick_us,
                 "launches_per_tick": ev_launches / max(ev_ticks, 1), "algorithmic_bytes_per_tick": alg_tick})
    roof["hbm_algorithmic"] = hbm_alg
    barrier()

    if rank == 0:
        n = sc.n
        ticks_s = args.steps / elapsed
        total_t = sum(times)
        out = {
            "metric": "sfm_ticks_per_s", "value": ticks_s, "unit": "ticks/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "windows": len(times), "min": args.steps / max(times), "max": args.steps / min(times),
            "mean_over_windows": args.steps * len(times) / total_t, "timed_seconds": total_t,
            "config": {"workload": f"{name}: N={n} pedestrians, forces={'+'.join(f.replace('_force', '') for f in forces)}, "
                                   f"borders={len(sc.borders)}, static={len(sc.static_obstacles)}, dynamic={len(sc.dynamic_obstacles)}, "
                                   f"dt={dt}", "n_pedestrians": n,
                       "sharding": (f"rows/{world}, tile-aligned, balanced by pair work, 1 all-gather/tick" if world > 1 else "none"),
                       "kernel": variant},
            "ns_per_pair": elapsed / args.steps * 1e9 / (n * (n - 1.0)),
            "pairs_per_s": n * (n - 1.0) * ticks_s,
            "roofline": roof,
        }
        if pair_terms:
            # the provably-negligible tile pairs are not evaluated: what the pair kernel actually did
            out["evaluated_pair_terms_per_tick"] = pair_terms
            out["evaluated_tile_pair_items_per_tick"] = pair_items
            out["evaluated_fraction_of_unordered_pairs"] = pair_terms / (n * (n - 1.0) / 2.0) if world == 1 else None
            out["ns_per_evaluated_pair_term"] = kernel_us * 1e3 / pair_terms
        if "pedestrian_force" in forces:
            # SURVEY.md section 8d, "algorithmic flops": ~65 fp32 VALU ops per ORDERED pedestrian pair against the
            # 7.9e13 lane-ops/s of the chip (256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz); the symmetric kernel evaluates an
            # unordered pair once, which this definition credits as two
            ops = 65.0 * n * (n - 1.0) * ticks_s
            out["roofline"]["valu_algorithmic"] = {"ops_per_ordered_pair": 65.0, "achieved_ops_per_s": ops, "peak_ops_per_s": 7.9e13,
                                                   "frac": ops / 7.9e13 / world}
        if world > 1:
            out["config"]["tick_form"] = getattr(st, "split_probe", {"chosen": "split" if st.split_mode else "plain"})
        if single_ref is not None:
            single_ref["speedup"] = ticks_s / single_ref["value"]
            out["single_gpu_same_workload"] = single_ref
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, forces, cfg, args.cpu_budget)
        os.write(record_fd, (json.dumps(out) + "\n").encode())
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
