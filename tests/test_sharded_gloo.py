"""world_size-2 (and 4) CPU test of the multi-GPU path over gloo: the partition + per-tick all-gather logic of
ShardedStepper, driven by a host engine (the oracle, used here as the checker's stand-in for the HIP engine,
which needs a GPU).  The sharded run must reproduce the single-rank run bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import ShardedStepper
from oracle import c_oracle
from oracle import sfm_oracle as O


class OracleShardEngine:
    """Same interface as stepper.HipShardEngine, computing on the host in float64 and keeping the packed
    {x,y,vx,vy} records in a CPU fp32 tensor (what the device buffer holds).  Like the device engine it keeps its
    rows in an internal order (``ids``: row -> caller's index), only keeps the waypoints / draw counters of its own
    rows current, and re-packs the rows on ``resort`` -- here simply by x, which is enough to move pedestrians
    between ranks."""

    auto_resort = False

    def __init__(self, cfg, dt):
        self.prm = O.OracleParams.from_config(cfg)
        self.dt = dt

    def load(self, sc, redraw=True):
        self.sc = sc
        self.n = sc.n
        self.n_pad = -(-sc.n // 256) * 256
        self.buf = torch.zeros(self.n_pad * 4, dtype=torch.float32)
        v = self.buf.view(self.n_pad, 4).numpy()
        v[:sc.n, 0:2] = sc.loc[:, :2]
        v[:sc.n, 2:4] = sc.vel[:, :2]
        self.own = torch.zeros(self.n_pad * 4, dtype=torch.float64)          # {wx, wy, target speed, radius} per row
        o = self.own.view(self.n_pad, 4).numpy()
        o[:sc.n, 0:2] = sc.waypoint[:, :2]; o[:sc.n, 2] = sc.target_speed; o[:sc.n, 3] = sc.radius
        self.draws = torch.zeros(self.n_pad, dtype=torch.int64)
        self.ids = np.arange(sc.n)
        self.geom = O.Geometry(sc.borders, sc.border_centers, sc.border_lengths, sc.static_obstacles,
                               sc.dynamic_obstacles, sc.dynamic_vel)
        return self.n, self.n_pad

    def set_shard(self, lo, hi):
        self.lo, self.hi = lo, hi

    def run(self, ticks, redraw=True):
        sc, lo, hi, n = self.sc, self.lo, self.hi, self.n
        for _ in range(ticks):
            v = self.buf.view(self.n_pad, 4).numpy()
            o = self.own.view(self.n_pad, 4).numpy()
            d = self.draws.numpy()
            loc = np.zeros((n, 3)); vel = np.zeros((n, 3)); wp = np.zeros((n, 3))
            loc[:, :2] = v[:n, 0:2]; vel[:, :2] = v[:n, 2:4]; wp[:, :2] = o[:n, 0:2]
            _, _, v_new, _, _ = c_oracle.tick(loc, vel, wp, o[:n, 2].copy(), o[:n, 3].copy(), np.zeros(n, bool),
                                              self.geom, self.prm, self.dt, rows=(lo, hi), nthreads=1)
            if redraw:
                hit = np.nonzero(O.arrived(loc[lo:hi], wp[lo:hi], 2.0))[0] + lo
                d[hit] += 1
                o[hit, 0:2] = O.redraw_waypoint(self.ids[hit], d[hit], sc.seed, sc.world_side)   # keyed by the caller's index
            v[lo:hi, 0:2] = (loc[lo:hi] + self.dt * v_new)[:, :2]
            v[lo:hi, 2:4] = v_new[:, :2]

    def packed(self):
        return [(self.buf, 4)]

    def row_data(self):
        return [(self.own, 4), (self.draws, 1)]

    def resort(self):
        n = self.n
        v = self.buf.view(self.n_pad, 4).numpy()
        order = np.argsort(v[:n, 0], kind="stable")
        v[:n] = v[:n][order]
        o = self.own.view(self.n_pad, 4).numpy()
        o[:n] = o[:n][order]
        d = self.draws.numpy()
        d[:n] = d[:n][order]
        self.ids = self.ids[order]

    def work(self):
        """A deliberately lopsided cost measure (rows x (1 + first own row / n)): the balancing then moves every
        boundary, shares become unequal and the padded exchange path of ShardedStepper._gather is what runs."""
        return (self.hi - self.lo) * (1.0 + 2.0 * self.lo / max(self.n, 1))

    def state(self):
        v = self.buf.view(self.n_pad, 4).numpy()
        o = self.own.view(self.n_pad, 4).numpy()
        loc = np.full((self.n, 3), np.nan); vel = np.full((self.n, 3), np.nan); wp = np.full((self.n, 2), np.nan)
        lo, hi = self.lo, self.hi                     # like the HIP engine: only this rank's pedestrians
        who = self.ids[lo:hi]
        loc[who, :2] = v[lo:hi, 0:2]; vel[who, :2] = v[lo:hi, 2:4]; loc[who, 2] = 0.0; vel[who, 2] = 0.0
        wp[who] = o[lo:hi, 0:2]
        return loc, vel, wp


CFG = default_sfm_config(("acceleration_force", "pedestrian_force", "border_force"))
N, TICKS, RESORT_EVERY = 700, 7, 3      # two re-packs inside the run: rows change rank twice


def _scenario():
    return scenarios.make_scenario(N, 2024, n_borders=12, border_len=(5.0, 20.0))


def _worker(rank, world, port, out, balance):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = ShardedStepper(OracleShardEngine(CFG, 0.05), _scenario(), rank=rank, world=world, resort_every=RESORT_EVERY,
                            balance=balance)
        st.step(TICKS)
        loc, vel, wp = st.gather_state()
        if rank == 0:
            np.savez(out, loc=loc, vel=vel, wp=wp, bounds=np.array(st.bounds))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,balance", [(2, False), (4, False), (2, True), (4, True)])
def test_sharded_equals_single_rank(world, balance, tmp_path):
    single = ShardedStepper(OracleShardEngine(CFG, 0.05), _scenario(), resort_every=RESORT_EVERY)
    single.step(TICKS)
    loc1, vel1, wp1 = single.gather_state()
    out = str(tmp_path / "sharded.npz")
    mp.spawn(_worker, args=(world, _free_port(), out, balance), nprocs=world, join=True)
    z = np.load(out)
    assert np.array_equal(z["loc"], loc1) and np.array_equal(z["vel"], vel1) and np.array_equal(z["wp"], wp1)
    shares = np.diff(z["bounds"])
    assert z["bounds"][0] == 0 and z["bounds"][-1] == N and (shares > 0).all()
    if balance:                                      # the boundaries moved: unequal shares of whole tiles, padded exchange
        assert len(set(shares[:-1].tolist())) > 1 or shares[0] != -(-N // 256) * 256 // world
        assert (z["bounds"][1:-1] % 64 == 0).all()
    # and the crowd actually moved / interacted
    assert np.linalg.norm(loc1 - _scenario().loc) > 1.0
