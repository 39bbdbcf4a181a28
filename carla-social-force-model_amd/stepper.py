"""Device-resident, index-sharded SFM stepping.

One process per GPU.  Every rank keeps the whole packed j-operand state {x,y,vx,vy} (16 B per pedestrian)
resident, computes and integrates only its contiguous block of pedestrian rows, and after each tick ONE
all-gather of the packed records over RCCL/xGMI (``torch.distributed``, backend "nccl") makes the next
tick's j-set complete again.  Geometry is replicated.  There is no other data-path collective.

The compute engine is injected: the product engine is ``HipShardEngine`` (libsfm_hip on ``cuda:<rank>``);
the CPU ``gloo`` tests drive the same partition / collective logic with a host engine of their own.
"""
from __future__ import annotations

import os
import time

import numpy as np

from .engine import SfmEngine


def shard_bounds(n, n_pad, rank, world):
    """Contiguous row block of ``rank`` under the EQUAL split: the padded record count is split evenly (it is a
    multiple of 256, hence of any world size up to 8 ranks x 32 lanes); rows at or beyond ``n`` are padding.
    Returns (lo, hi, chunk).  The starting point of ``ShardedStepper``; ``balanced_bounds`` moves it from there."""
    if n_pad % world:
        raise ValueError(f"padded size {n_pad} not divisible by world size {world}")
    chunk = n_pad // world
    lo = min(rank * chunk, n)
    hi = min((rank + 1) * chunk, n)
    return lo, hi, chunk


TILE = 64
ROW_COST_TERMS = 600             # HipShardEngine.work(): per-row cost of a tick outside the pair kernel, in pair terms
GEOMETRY_ROW_COST_TERMS = 1000   # ... and of the border / obstacle forces


def block_layout(world, layout=None):
    """(gx, gy) of the rank blocks, gx columns by x times gy blocks by y: as square as the rank count allows
    (2 -> 1x2, 4 -> 2x2, 8 -> 2x4).  ``layout=(gx, gy)`` overrides (e.g. (8, 1) = the slabs of round 1)."""
    if layout:
        gx, gy = (int(v) for v in layout)
        if gx * gy != world:
            raise ValueError(f"layout {gx}x{gy} does not make {world} blocks")
        return gx, gy
    gy = 1
    while gy * gy * 2 <= world and world % (gy * 2) == 0:
        gy *= 2
    return world // gy, gy


def equal_bounds(n, n_pad, world):
    """[b_0 = 0, b_1, ..., b_world = n]: the equal split as a bounds list.  Every interior bound is a whole number of 64-row
    tiles (sfm_set_partition and the symmetric kernel's shards need that) -- the library's own default split,
    ``n_pad * r / world`` rounded down to a tile -- and never beyond the last whole tile of the real rows.  When
    ``n_pad / world`` is itself a multiple of 64 (every BASELINE config) this is ``shard_bounds``' split and the exchange is
    one in-place all-gather; otherwise the shares differ by a tile and the exchange goes through the padded staging buffer."""
    if n_pad % world:
        raise ValueError(f"padded size {n_pad} not divisible by world size {world}")
    top = (n // TILE) * TILE
    return [min((n_pad * r // world) // TILE * TILE, top) for r in range(world)] + [n]


def balanced_bounds(bounds, costs, n, damping=0.7, min_rows=TILE):
    """Move the shard boundaries so that every rank carries the same cost (a rank in the middle of
    the map has two neighbours whose boundary pairs it evaluates one-sided, a rank at the edge one).

    ``costs[r]`` is what rank r spent on rows [bounds[r], bounds[r+1]) (any positive measure; the stepper uses the pair
    kernel's evaluated terms, a deterministic function of state and partition).  The cost is taken as uniform over a
    rank's rows; the new boundary k sits where the cumulative cost reaches k/G of the total, moved only ``damping`` of the way
    (the one-sided work travels with the boundary, so the estimate is re-measured after every move), rounded to whole
    64-row tiles and kept at least ``min_rows`` apart.  Pure function of its arguments: every rank computes the same list."""
    world = len(bounds) - 1
    if world == 1:
        return [0, n]
    rows = [bounds[r + 1] - bounds[r] for r in range(world)]
    total = float(sum(costs))
    if total <= 0.0 or any(c < 0 for c in costs):
        return list(bounds)
    cum = [0.0]
    for c in costs:
        cum.append(cum[-1] + float(c))
    new = [0]
    r = 0
    for k in range(1, world):
        target = total * k / world
        while r < world - 1 and cum[r + 1] < target:
            r += 1
        frac = (target - cum[r]) / max(cum[r + 1] - cum[r], 1e-300)
        ideal = bounds[r] + frac * rows[r]
        moved = bounds[k] + damping * (ideal - bounds[k])
        b = int(round(moved / TILE)) * TILE
        # whole tiles, at least ``min_rows`` per rank where the crowd has that many tiles (a crowd of fewer tiles than ranks
        # leaves some ranks without rows: their bounds coincide, still on tile edges)
        step = max(TILE, (min_rows + TILE - 1) // TILE * TILE)
        b = max(b, new[-1] + step)
        top = (n // TILE) * TILE                 # the last rank can live on the partial tile [top, n) if there is one
        b = min(b, top - (world - 1 - k) * step - (0 if n > top else step))
        b = max(b, new[-1])
        new.append(b)
    new.append(n)
    return new


class _DevSpan:
    """Minimal __cuda_array_interface__ carrier so torch can alias a raw device pointer (no copy)."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2}


class HipShardEngine:
    """libsfm_hip handle + torch views of its packed buffers (the plumbing the collective needs)."""

    def __init__(self, sfm_config, step_length, device=0):
        import torch
        self.torch = torch
        self.device = int(device)
        torch.cuda.set_device(self.device)
        self.engine = SfmEngine(sfm_config, step_length, device=self.device,
                                stream=torch.cuda.current_stream().cuda_stream)
        self._views = {}

    def load(self, sc, redraw=True):
        e = self.engine
        if sc.borders:
            e.set_borders(sc.borders, sc.border_centers, sc.border_lengths)
        e.set_static_obstacles(sc.static_obstacles)
        if len(sc.dynamic_obstacles):     # vehicles live on the device and move between ticks
            e.set_dynamic_boxes([c for c, _ in sc.dynamic_obstacles], sc.dynamic_yaw, sc.dynamic_extent, sc.dynamic_vel)
        crossing = (sc.mode == 2) | (sc.mode == 3)
        e.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, crossing)
        e.set_waypoint_stream(sc.seed, sc.world_side, 2.0)
        en = e.params.enabled
        self._has_geometry = bool((en[2] and len(sc.borders)) or (en[3] and len(sc.static_obstacles)) or (en[4] and len(sc.dynamic_obstacles)))
        self.n = sc.n
        _, self.n_pad = e.packed_state_ptr()
        return self.n, self.n_pad

    def set_shard(self, lo, hi):
        self.engine.set_shard(lo, hi)

    def set_partition(self, gx, gy, bounds=None):
        self.engine.set_partition(gx, gy, bounds)

    def run(self, ticks, redraw=True):
        self.engine.run(ticks, redraw=redraw)

    def begin(self, redraw=True):
        """The part of the next tick that only reads this rank's own rows (runs beside the exchange of the previous tick)."""
        self.engine.tick_begin(redraw=redraw)

    def end(self, redraw=True):
        """The rest of the tick; every other rank's rows must be current."""
        self.engine.tick_end(redraw=redraw)

    def _view(self, ptr, width):
        key = (ptr, width)
        if key not in self._views:
            self._views[key] = self.torch.as_tensor(_DevSpan(ptr, self.n_pad * width), device=f"cuda:{self.device}")
        return self._views[key]

    def packed(self):
        """[(flat fp32 tensor aliasing the buffer the next tick reads, floats per record), ...]"""
        ptr, _ = self.engine.packed_state_ptr()
        out = [(self._view(ptr, 4), 4)]
        zptr = self.engine.packed_z_ptr()
        if zptr:
            out.append((self._view(zptr, 2), 2))
        return out

    def row_data(self):
        """[(flat fp32 tensor aliasing a per-row array only its owner keeps current, floats per row), ...]: what a
        sharded driver all-gathers before ``resort`` (waypoints + target speed + radius; waypoint draw counters,
        moved as raw 32-bit words)."""
        out = []
        for which in (0, 1):
            ptr, nbytes = self.engine.row_data_ptr(which)
            out.append((self._view(ptr, nbytes // 4), nbytes // 4))
        return out

    def resort(self):
        self.engine.resort()

    def work(self):
        """What this rank's last tick cost, for the boundary balancing: the Moussaid terms the pair kernel evaluated
        (a deterministic function of state and partition); 0 when the ordered kernel ran (no measure: boundaries stay)."""
        try:
            terms = self.engine.pair_work()[1]
        except Exception:
            return 0
        # ... plus what scales with the rows themselves (tile boxes, list building, epilogue; the geometry kernel when a
        # border / obstacle force is on), in units of pair terms: fitted on c5 at G = 8 (tools/shard_probe.py:
        # ~1.5 ns per row against ~1 ps per term)
        lo, hi = self.engine.shard
        return terms + (hi - lo) * (ROW_COST_TERMS + (GEOMETRY_ROW_COST_TERMS if self._has_geometry else 0))

    auto_resort = True          # a whole-crowd handle re-packs its rows by itself (SFM_RESORT_EVERY)

    def state(self):
        return self.engine.state()

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    def close(self):
        self.engine.close()


class ShardedStepper:
    """K ticks of the CARLA-free loop across ``world`` ranks.

    Rank r owns rows [bounds[r], bounds[r+1]) of the (spatially packed) row order: whole 64-row tiles, equal shares to
    start with.  Every ``resort_every`` ticks the rows are re-packed (all ranks identically) and -- ``balance`` -- the
    boundaries move so that every rank carries the same pair work (``balanced_bounds``; the measure is the engine's
    ``work()``, all-gathered, so every rank computes the same new boundaries)."""

    def __init__(self, engine, scenario, rank=0, world=1, group=None, redraw=True, resort_every=None, balance=True, layout=None,
                 split="auto"):
        self.engine, self.rank, self.world, self.group, self.redraw = engine, rank, world, group, redraw
        # The split tick (engine.begin beside the exchange, engine.end after it) hides the all-gather behind the own-own pair work at
        # the price of three more launches per tick (~15-20 us on one replayed rank of c5, tools/shard_rank_time.py); whether that
        # pays depends on what the collective costs on the node at hand.  "auto": the first ticks alternate between the two forms in
        # blocks of three, timed with a device synchronisation, and after eight samples of each the faster one (max over ranks of the
        # medians, so that every rank decides alike) is kept.  Both forms compute the same bits.
        self.split_mode = split if (world > 1 and hasattr(engine, "begin")) else False
        self._split_samples = {True: [], False: []}
        # rank blocks instead of slabs: every rank's rows are the pedestrians of one rectangle of a gx x gy grid over the map
        self.layout = block_layout(world, layout) if (world > 1 and hasattr(engine, "set_partition")) else None
        if self.layout:
            engine.set_partition(*self.layout)
        self.n, self.n_pad = engine.load(scenario, redraw=redraw)
        self.bounds = equal_bounds(self.n, self.n_pad, world)
        self.lo, self.hi = self.bounds[rank], self.bounds[rank + 1]
        engine.set_shard(self.lo, self.hi)
        if self.layout:
            engine.set_partition(*self.layout, self.bounds)
        self.ticks_done = 0
        # rows are kept spatially packed (compact 64-row tiles); pedestrians walk, so the packing is renewed every
        # `resort_every` ticks.  One rank: the engine does it itself.  Shards: gather the owner-only per-row arrays,
        # then every rank re-packs identically and carries on with its rows of the new order.
        if resort_every is None:
            resort_every = int(os.environ.get("SFM_RESORT_EVERY", "64"))
        self.resort_every = resort_every
        self.since_resort = 0
        self.balance = balance and world > 1 and hasattr(engine, "work")
        self._plans = {}

    # ---- the exchange ---------------------------------------------------------------------------------------------
    def exchange(self, force=False):
        """The one collective of a tick: all-gather of the packed records.  ``force`` issues it on a single-rank group
        too (used to exercise the RCCL path on a one-GPU box)."""
        if self.world == 1 and not force:
            return
        self._gather(self.engine.packed())

    def _gather(self, buffers):
        self._gather_finish(self._gather_start(buffers))

    def _gather_start(self, buffers):
        """Every rank's own rows of each buffer reach every other rank.  Equal shares: ONE in-place
        all_gather_into_tensor on the buffer itself.  Unequal shares (balanced boundaries): the own rows go into this
        rank's slot of a staging tensor padded to the largest share, one in-place all_gather_into_tensor on it, and one
        indexed copy spreads the other ranks' rows over the buffer (two small kernels beside the collective).
        The collectives are issued asynchronously; ``_gather_finish`` waits for them (and does the indexed copies), so
        work that only touches this rank's own rows can be queued in between."""
        import torch.distributed as dist
        b = self.bounds
        sizes = [b[r + 1] - b[r] for r in range(self.world)]
        chunk = sizes[0]
        even = all(sz == chunk for sz in sizes[:-1]) and sizes[-1] <= chunk and all(b[r] == r * chunk for r in range(self.world))
        pending = []
        for buf, width in buffers:
            if even and buf.numel() >= self.world * chunk * width:
                mine = buf[self.rank * chunk * width:(self.rank + 1) * chunk * width]
                w = dist.all_gather_into_tensor(buf[:self.world * chunk * width], mine, group=self.group, async_op=True)
                pending.append((w, None))
                continue
            stage, src, dst = self._plan(buf, width, sizes)
            rows = buf.view(-1, width)
            stage[self.rank, :sizes[self.rank]].copy_(rows[self.lo:self.hi])
            w = dist.all_gather_into_tensor(stage.view(-1), stage[self.rank].reshape(-1), group=self.group, async_op=True)
            pending.append((w, (rows, stage, src, dst, width)))
        return pending

    def _gather_finish(self, pending):
        for w, unpack in pending or ():
            w.wait()
            if unpack is not None:
                rows, stage, src, dst, width = unpack
                rows.index_copy_(0, dst, stage.view(-1, width).index_select(0, src))

    def _plan(self, buf, width, sizes):
        """(staging tensor [world, largest share, width], source rows in it, destination rows in the buffer) for the
        current boundaries; cached per (buffer, boundaries)."""
        import torch
        key = (buf.data_ptr(), width, tuple(self.bounds))
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 16:
                self._plans.clear()
            cmax = max(sizes)
            stage = torch.zeros((self.world, cmax, width), dtype=buf.dtype, device=buf.device)
            src, dst = [], []
            for r in range(self.world):
                if r == self.rank or sizes[r] == 0:
                    continue
                k = np.arange(sizes[r], dtype=np.int64)
                src.append(r * cmax + k)
                dst.append(self.bounds[r] + k)
            cat = lambda parts: torch.as_tensor(np.concatenate(parts) if parts else np.zeros(0, np.int64), device=buf.device)
            plan = (stage, cat(src), cat(dst))
            self._plans[key] = plan
        return plan

    # ---- stepping -------------------------------------------------------------------------------------------------
    def _repack(self):
        """All ranks: bring every row's owner-only data up to date everywhere, re-pack identically, then (balance) move
        the boundaries by the work each rank measured in its last tick."""
        if self.world > 1:
            self._gather(self.engine.row_data())
        self.engine.resort()
        if self.balance:
            import torch
            import torch.distributed as dist
            mine = float(self.engine.work())
            dev = self.engine.packed()[0][0].device
            t = torch.zeros(self.world, dtype=torch.float64, device=dev)
            t[self.rank] = mine
            dist.all_reduce(t, group=self.group)
            self.set_bounds(balanced_bounds(self.bounds, [float(v) for v in t.cpu().tolist()], self.n))
        self.since_resort = 0

    def set_bounds(self, bounds):
        assert len(bounds) == self.world + 1 and bounds[0] == 0 and bounds[-1] == self.n
        self.bounds = list(bounds)
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.engine.set_shard(self.lo, self.hi)
        if self.layout:
            self.engine.set_partition(*self.layout, self.bounds)      # the next re-pack cuts the blocks at the new boundaries

    def step(self, ticks=1):
        """``ticks`` ticks.  Sharded: the exchange of tick t's rows is started right after tick t and only waited for
        after the part of tick t+1 that reads nothing but this rank's own rows (``engine.begin``: tile boxes, border /
        obstacle forces, own-own tile pairs) has been queued -- the collective runs beside it.
        On return the exchange of the last tick has completed: every rank holds the whole state."""
        driver_resorts = self.resort_every > 0 and not (self.world == 1 and getattr(self.engine, "auto_resort", False))
        if self.world == 1 and not driver_resorts:
            self.engine.run(ticks, redraw=self.redraw)      # no exchange needed: launch back to back
        else:
            pending = None
            for _ in range(ticks):
                if driver_resorts and self.since_resort >= self.resort_every:
                    self._gather_finish(pending)
                    pending = None
                    self._repack()
                probing = self.split_mode == "auto"
                if probing:                                  # blocks of three ticks of either form, each tick timed on its own
                    k = len(self._split_samples[True]) + len(self._split_samples[False])
                    split = (k // 3) % 2 == 0
                    self._gather_finish(pending)
                    pending = None
                    self.engine.synchronize()
                    t0 = time.perf_counter()
                else:
                    split = bool(self.split_mode)
                if split:
                    self.engine.begin(redraw=self.redraw)
                    self._gather_finish(pending)
                    self.engine.end(redraw=self.redraw)
                else:
                    self._gather_finish(pending)
                    self.engine.run(1, redraw=self.redraw)
                pending = self._gather_start(self.engine.packed()) if self.world > 1 else None
                if probing:
                    self._gather_finish(pending)             # a probed tick pays for its own exchange
                    pending = None
                    self.engine.synchronize()
                    self._split_samples[split].append(time.perf_counter() - t0)
                    self._decide_split()
                self.since_resort += 1
            self._gather_finish(pending)
        self.ticks_done += ticks

    def _decide_split(self):
        a, b = self._split_samples[True], self._split_samples[False]
        if len(a) < 8 or len(b) < 8:
            return
        import torch
        import torch.distributed as dist
        dev = self.engine.packed()[0][0].device
        t = torch.tensor([float(np.median(a[2:])), float(np.median(b[2:]))], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        split_s, plain_s = (float(v) for v in t.cpu().tolist())
        self.split_mode = split_s < plain_s
        self.split_probe = {"split_tick_us": split_s * 1e6, "plain_tick_us": plain_s * 1e6, "chosen": "split" if self.split_mode else "plain"}

    def local_rows(self):
        return self.lo, self.hi

    def gather_state(self):
        """(loc, vel, wp) of all pedestrians on every rank (host side; for tests and reporting)."""
        loc, vel, wp = self.engine.state()
        if self.world == 1:
            return loc, vel, wp
        import torch.distributed as dist
        own = np.flatnonzero(~np.isnan(vel[:, 0]))          # this rank's pedestrians (rows are an internal order)
        mine = np.concatenate([loc[own], vel[own], wp[own]], axis=1)
        parts = [None] * self.world
        dist.all_gather_object(parts, (own, mine), group=self.group)
        for idx, blk in parts:
            loc[idx], vel[idx], wp[idx] = blk[:, 0:3], blk[:, 3:6], blk[:, 6:8]
        return loc, vel, wp
