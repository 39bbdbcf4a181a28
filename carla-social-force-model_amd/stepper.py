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

import numpy as np

from .engine import SfmEngine


def shard_bounds(n, n_pad, rank, world):
    """Contiguous row block of ``rank``: the padded record count is split evenly (it is a multiple of 256,
    hence of any world size up to 8 ranks x 32 lanes), so every rank contributes an equal-sized chunk to
    the all-gather; rows at or beyond ``n`` are padding."""
    if n_pad % world:
        raise ValueError(f"padded size {n_pad} not divisible by world size {world}")
    chunk = n_pad // world
    lo = min(rank * chunk, n)
    hi = min((rank + 1) * chunk, n)
    return lo, hi, chunk


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
        self.n = sc.n
        _, self.n_pad = e.packed_state_ptr()
        return self.n, self.n_pad

    def set_shard(self, lo, hi):
        self.engine.set_shard(lo, hi)

    def run(self, ticks, redraw=True):
        self.engine.run(ticks, redraw=redraw)

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

    auto_resort = True          # a whole-crowd handle re-packs its rows by itself (SFM_RESORT_EVERY)

    def state(self):
        return self.engine.state()

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    def close(self):
        self.engine.close()


class ShardedStepper:
    """K ticks of the CARLA-free loop across ``world`` ranks."""

    def __init__(self, engine, scenario, rank=0, world=1, group=None, redraw=True, resort_every=None):
        self.engine, self.rank, self.world, self.group, self.redraw = engine, rank, world, group, redraw
        self.n, self.n_pad = engine.load(scenario, redraw=redraw)
        self.lo, self.hi, self.chunk = shard_bounds(self.n, self.n_pad, rank, world)
        engine.set_shard(self.lo, self.hi)
        self.ticks_done = 0
        # rows are kept spatially packed (compact 64-row tiles); pedestrians walk, so the packing is renewed every
        # `resort_every` ticks.  One rank: the engine does it itself.  Shards: gather the owner-only per-row arrays,
        # then every rank re-packs identically and carries on with rows [lo, hi) of the new order.
        if resort_every is None:
            resort_every = int(os.environ.get("SFM_RESORT_EVERY", "64"))
        self.resort_every = resort_every
        self.since_resort = 0

    def exchange(self, force=False):
        """The one collective of a tick: in-place all-gather of the packed records.  ``force`` issues it on a
        single-rank group too (used to exercise the RCCL path on a one-GPU box)."""
        if self.world == 1 and not force:
            return
        self._gather(self.engine.packed())

    def _gather(self, buffers):
        import torch.distributed as dist
        for buf, width in buffers:
            mine = buf[self.rank * self.chunk * width:(self.rank + 1) * self.chunk * width]
            dist.all_gather_into_tensor(buf, mine, group=self.group)

    def step(self, ticks=1):
        driver_resorts = self.resort_every > 0 and not (self.world == 1 and getattr(self.engine, "auto_resort", False))
        if self.world == 1 and not driver_resorts:
            self.engine.run(ticks, redraw=self.redraw)      # no exchange needed: launch back to back
        else:
            for _ in range(ticks):
                if driver_resorts and self.since_resort >= self.resort_every:
                    if self.world > 1:
                        self._gather(self.engine.row_data())
                    self.engine.resort()
                    self.since_resort = 0
                self.engine.run(1, redraw=self.redraw)
                self.exchange()
                self.since_resort += 1
        self.ticks_done += ticks

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
