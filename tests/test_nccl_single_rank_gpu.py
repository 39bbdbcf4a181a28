"""GPU test of the multi-GPU plumbing on a one-GPU box: a world-size-1 RCCL ("nccl") process group, the torch
view aliasing libsfm_hip's packed device buffer (no copy), the in-place all_gather_into_tensor on it, and
kernel launches on torch's stream.  The sharded result must equal the plain single-engine run bit for bit."""
import os
import socket

import numpy as np
import pytest

from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config

pytestmark = pytest.mark.gpu


def test_world1_rccl_allgather_on_library_buffer():
    import torch
    import torch.distributed as dist
    from carla_social_force_model_amd.engine import SfmEngine
    from carla_social_force_model_amd.stepper import HipShardEngine, ShardedStepper

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        cfg = default_sfm_config(("acceleration_force", "pedestrian_force", "border_force"))
        sc = scenarios.make_scenario(1500, 31, n_borders=8, border_len=(5.0, 20.0))
        os.environ["SFM_SYM"] = "0"                     # ordered kernel on both sides: bit-identical
        os.environ["SFM_REORDER"] = "1"                 # spatial packing on at this size, so that resort() does something
        os.environ["SFM_RESORT_EVERY"] = "0"            # ... and only when this test says so
        eng = HipShardEngine(cfg, 0.05, device=0)
        st = ShardedStepper(eng, sc, rank=0, world=1)
        (buf, width), = eng.packed()
        assert buf.is_cuda and buf.numel() == eng.n_pad * 4 and buf.data_ptr() == eng.engine.packed_state_ptr()[0]
        before = buf[:4 * sc.n].clone()
        for _ in range(5):
            eng.run(1, redraw=True)
            st.exchange(force=True)                     # all_gather_into_tensor(buf, buf[chunk]) in place
        eng.synchronize()
        assert not torch.equal(eng.packed()[0][0][:4 * sc.n], before)
        # the re-pack protocol of a sharded run: all-gather the owner-only per-row arrays (RCCL on the library's
        # buffers again), re-pack, carry on
        st._gather(eng.row_data())
        eng.resort()
        for _ in range(2):
            eng.run(1, redraw=True)
            st.exchange(force=True)
        eng.synchronize()
        loc, vel, wp = st.gather_state()
        plain = SfmEngine(cfg, 0.05)
        plain.set_borders(sc.borders, sc.border_centers, sc.border_lengths)
        plain.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
        plain.set_waypoint_stream(sc.seed, sc.world_side, 2.0)
        plain.run(5, redraw=True)
        plain.resort()
        plain.run(2, redraw=True)
        loc2, vel2, wp2 = plain.state()
        assert np.array_equal(loc, loc2) and np.array_equal(vel, vel2) and np.array_equal(wp, wp2)
        plain.close()
        eng.close()
    finally:
        for k in ("SFM_SYM", "SFM_REORDER", "SFM_RESORT_EVERY"):
            os.environ.pop(k, None)
        dist.destroy_process_group()


def test_row_shards_reassemble_the_full_tick():
    """Two handles on one GPU, each computing half of the rows of the same state (what two ranks do before
    the all-gather): concatenated they equal the unsharded tick bit for bit."""
    from carla_social_force_model_amd.engine import SfmEngine
    os.environ["SFM_SYM"] = "0"
    try:
        cfg = default_sfm_config()
        sc = scenarios.make_scenario(1000, 8, n_borders=10, n_static=5, n_dynamic=3, border_len=(5.0, 20.0))
        outs = []
        for rows in ((0, 1000), (0, 512), (512, 1000)):
            e = SfmEngine(cfg, 0.05)
            e.set_borders(sc.borders, sc.border_centers, sc.border_lengths)
            e.set_static_obstacles(sc.static_obstacles)
            e.set_dynamic_obstacles(sc.dynamic_obstacles, sc.dynamic_vel)
            e.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
            e.set_shard(*rows)
            e.tick(integrate=True, record=True)
            outs.append((e.velocities(), e.forces("total"), e.state()[0]))
            e.close()
        full, lo, hi = outs
        for k in range(3):
            assert np.array_equal(full[k][:512], lo[k][:512]) and np.array_equal(full[k][512:], hi[k][512:])
            assert np.isnan(lo[k][512:]).all() and np.isnan(hi[k][:512]).all()     # rows outside the shard untouched
    finally:
        os.environ.pop("SFM_SYM", None)


@pytest.mark.parametrize("path", ["ordered", "symmetric", "symmetric-split"])
def test_two_shards_with_row_repacking_match_the_whole_crowd_run(path, monkeypatch):
    """What two ranks do, replayed on one GPU with two handles: a tick on the own rows, the exchange of the packed
    records (here a device copy instead of the all-gather), and every 4 ticks the re-pack protocol of
    sfm_resort -- exchange the owner-only per-row arrays, re-pack identically on both handles, carry on with the same
    row ranges (now other pedestrians).  Must equal the whole-crowd handle, which re-packs by itself, bit for bit."""
    import torch
    from carla_social_force_model_amd.stepper import HipShardEngine, shard_bounds
    # ordered kernel on every handle: bit-identical.  Symmetric kernel (tile-aligned shards + tile-pair list): a pair
    # across the shard boundary is evaluated one-sided by both handles, so sums are ordered differently -> rounding only.
    # "symmetric-split": the tick in two halves (sfm_tick_begin / sfm_tick_end) with the exchange of the PREVIOUS tick's rows in
    # between -- what ShardedStepper.step does to run the collective beside the own-own tile pairs.
    split = path == "symmetric-split"
    if split:
        path = "symmetric"
    env = {"SFM_CUTOFF": "1", "SFM_RESORT_EVERY": "4"}
    env.update({"SFM_SYM": "0", "SFM_IPW": "4", "SFM_TEAM": "1"} if path == "ordered" else {"SFM_SYM": "1"})
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = default_sfm_config(("acceleration_force", "pedestrian_force", "border_force", "static_obstacle_force"))
    sc = scenarios.make_scenario(3000, 97, n_borders=20, n_static=10, border_len=(5.0, 25.0))
    ticks = 11
    whole = HipShardEngine(cfg, 0.05)
    whole.load(sc)
    whole.run(ticks)
    ref = whole.engine.state()
    whole_variant = whole.engine.kernel_variant()
    whole.close()

    assert ("sym" in whole_variant) == (path == "symmetric")
    ranks = [HipShardEngine(cfg, 0.05) for _ in range(2)]
    bounds = []
    for r, e in enumerate(ranks):
        n, n_pad = e.load(sc)
        lo, hi, chunk = shard_bounds(n, n_pad, r, 2)
        e.set_shard(lo, hi)
        bounds.append((lo, hi, chunk))

    def exchange(buffers_of):
        """every handle receives the other handle's own chunk of each buffer"""
        bufs = [buffers_of(e) for e in ranks]
        for r in range(2):
            chunk = bounds[r][2]
            for (src, width), (dst, _) in zip(bufs[r], bufs[1 - r]):
                dst[r * chunk * width:(r + 1) * chunk * width].copy_(src[r * chunk * width:(r + 1) * chunk * width])
        torch.cuda.synchronize()

    moved = 0
    for t in range(ticks):
        if t and t % 4 == 0:
            exchange(lambda e: e.row_data())
            before = ranks[0].engine.state()[0]
            for e in ranks:
                e.resort()
            after = ranks[0].engine.state()[0]
            moved += int((np.isnan(before[:, 0]) != np.isnan(after[:, 0])).sum())   # pedestrians that changed rank
        if split:
            for e in ranks:
                e.begin()                                  # own rows only: the other handle's rows are still one tick old here
            exchange(lambda e: e.packed())
            for e in ranks:
                e.end()
                e.synchronize()
            if (t + 1) % 4 == 0 or t + 1 == ticks:         # the re-pack (and the final comparison) needs the whole state
                exchange(lambda e: e.packed())
        else:
            for e in ranks:
                e.run(1)
                e.synchronize()
            exchange(lambda e: e.packed())
    assert moved > 0
    got = [e.engine.state() for e in ranks]
    for k in range(3):
        mine0, mine1 = ~np.isnan(got[0][k][:, 0]), ~np.isnan(got[1][k][:, 0])
        assert (mine0 ^ mine1).all()                       # every pedestrian is owned by exactly one handle
        merged = np.where(mine0[:, None], got[0][k], got[1][k])
        if path == "ordered":
            assert np.array_equal(merged, ref[k])
        else:
            assert np.allclose(merged, ref[k], rtol=2e-5, atol=2e-5)
    assert all(("sym" in e.engine.kernel_variant()) == (path == "symmetric") for e in ranks)
    assert all(("own|remote" in e.engine.kernel_variant()) == split for e in ranks)      # the split tick really ran in two halves
    for e in ranks:
        e.close()


def _two_rank_worker(rank, world, port, out, ticks, every):
    import torch.distributed as dist
    from carla_social_force_model_amd.stepper import HipShardEngine, ShardedStepper
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SFM_CUTOFF="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = default_sfm_config(("acceleration_force", "pedestrian_force", "border_force"))
        sc = scenarios.make_scenario(6000, 4711, n_borders=25, border_len=(5.0, 25.0))
        eng = HipShardEngine(cfg, 0.05, device=0)
        st = ShardedStepper(eng, sc, rank=rank, world=world, resort_every=every)
        st.step(ticks)
        eng.synchronize()
        variant = eng.engine.kernel_variant()
        loc, vel, wp = st.gather_state()
        if rank == 0:
            np.savez(out, loc=loc, vel=vel, wp=wp, sym=np.array("sym" in variant))
        eng.close()
    finally:
        dist.destroy_process_group()


def test_two_processes_share_the_gpu_and_match_the_single_process_run(tmp_path, monkeypatch):
    """The real multi-rank flow -- two processes, ShardedStepper + HipShardEngine, symmetric kernel on tile-aligned
    shards, the per-tick exchange and two re-packs (rows change rank) -- with both ranks on the one GPU of this box and
    the collectives over gloo.  Against one process stepping the whole crowd: rounding-level agreement."""
    import torch.multiprocessing as mp
    from carla_social_force_model_amd.stepper import HipShardEngine, ShardedStepper
    ticks, every = 20, 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "two_ranks.npz")
    mp.spawn(_two_rank_worker, args=(2, port, out, ticks, every), nprocs=2, join=True)
    z = np.load(out)
    assert bool(z["sym"])
    monkeypatch.setenv("SFM_CUTOFF", "1")
    monkeypatch.setenv("SFM_RESORT_EVERY", str(every))
    cfg = default_sfm_config(("acceleration_force", "pedestrian_force", "border_force"))
    sc = scenarios.make_scenario(6000, 4711, n_borders=25, border_len=(5.0, 25.0))
    eng = HipShardEngine(cfg, 0.05, device=0)
    st = ShardedStepper(eng, sc)
    st.step(ticks)
    loc, vel, wp = st.gather_state()
    eng.close()
    assert np.allclose(z["loc"], loc, rtol=2e-5, atol=2e-5) and np.allclose(z["vel"], vel, rtol=2e-5, atol=2e-5)
    assert np.array_equal(z["wp"], wp)


def test_antipodal_tile_pairs_across_the_shard_boundary_are_evaluated(monkeypatch):
    """Even tile count, two shards, a crowd narrower than the cutoff reach: the pair (tile t, tile t + n_t/2) has its two
    tiles on different ranks and is NOT negligible, so each rank must evaluate its own side of it (a one-sided list item).
    Round 1 dropped those items for the upper-half tiles (and with them the rest of the workgroup's run of the list)."""
    import torch
    from carla_social_force_model_amd.stepper import HipShardEngine, shard_bounds
    for k, v in {"SFM_CUTOFF": "1", "SFM_SYM": "1", "SFM_RESORT_EVERY": "0"}.items():
        monkeypatch.setenv(k, v)
    cfg = default_sfm_config(("acceleration_force", "pedestrian_force"))
    sc = scenarios.make_scenario(8192, 424242, density=1.0)           # 90 m square, 128 tiles
    whole = HipShardEngine(cfg, 0.05)
    whole.load(sc)
    whole.run(1)
    ref = whole.engine.state()
    assert "sym" in whole.engine.kernel_variant()
    whole.close()
    got = []
    for r in range(2):
        e = HipShardEngine(cfg, 0.05)
        n, n_pad = e.load(sc)
        lo, hi, _ = shard_bounds(n, n_pad, r, 2)
        e.set_shard(lo, hi)
        e.run(1)
        e.synchronize()
        assert "sym" in e.engine.kernel_variant()
        got.append(e.engine.state())
        e.close()
    torch.cuda.synchronize()
    for k in range(2):                                                  # loc, vel
        mine0, mine1 = ~np.isnan(got[0][k][:, 0]), ~np.isnan(got[1][k][:, 0])
        assert (mine0 ^ mine1).all()
        merged = np.where(mine0[:, None], got[0][k], got[1][k])
        assert np.isfinite(merged).all()
        assert np.allclose(merged, ref[k], rtol=2e-5, atol=2e-5)


def test_rows_written_through_the_packed_pointer_invalidate_the_carried_tile_boxes():
    """A whole crowd at N >= 8192 carries the NEXT tick's tile boxes from its epilogue (no sfm_tile_bounds launch).  A caller that
    writes the packed rows through sfm_packed_state_ptr in between must get a tile-pair list built from fresh boxes (round-2
    advisor finding: the getters only reset the fused tick's carry): moved crowd through the pointer == fresh upload of it."""
    import torch
    from carla_social_force_model_amd.engine import SfmEngine
    from carla_social_force_model_amd.stepper import HipShardEngine

    n = 8192
    sc = scenarios.make_scenario(n, 77)
    cfg = default_sfm_config(("acceleration_force", "pedestrian_force"))
    eng = HipShardEngine(cfg, 0.05, device=0)
    other = SfmEngine(cfg, 0.05)
    try:
        eng.load(sc)
        eng.run(3)                                      # the last epilogue left boxes for tick 4
        (buf, width), = eng.packed()                    # the pointer is handed out: the caller may write
        rows = buf.view(-1, 4)[:n]
        moved = rows.clone()
        far = moved[:, 0] > moved[:, 0].median()        # half the crowd goes 300 m away: every tile's box changes or moves
        moved[far, 0] += 300.0
        rows.copy_(moved)
        eng.synchronize()
        eng.engine.tick(record=True)
        assert "sym" in eng.engine.kernel_variant()
        got_f, got_v = eng.engine.forces("pedestrian_force"), eng.engine.velocities()
        loc, vel, wp = eng.engine.state()               # (velocities: those the tick left in the state's place are read above)
        loc0, vel0, wp0 = loc.copy(), vel.copy(), wp.copy()
        # the same state uploaded afresh (the tick above did not integrate positions; its input velocities are `moved`'s)
        perm_state = moved.cpu().numpy().astype(np.float64)
        # rows are in the library's order: map back through positions (unique) to the caller's order
        key = {tuple(np.float32(p)): i for i, p in enumerate(loc0[:, :2])}
        vel_in = np.zeros_like(vel0)
        for r in perm_state:
            vel_in[key[(np.float32(r[0]), np.float32(r[1]))], :2] = r[2:4]
        wp3 = np.zeros_like(loc0); wp3[:, :2] = wp0
        other.upload_state(loc0, vel_in, wp3, sc.target_speed, sc.radius, None)
        other.tick(record=True)
        want_f, want_v = other.forces("pedestrian_force"), other.velocities()
        # another packing order sums in another order: rounding only -- a stale list drops whole tile pairs (errors of order one)
        scale = np.abs(want_f).max()
        assert np.abs(got_f - want_f).max() <= 2e-5 * scale, np.abs(got_f - want_f).max() / scale
        assert np.abs(got_v - want_v).max() <= 1e-5
    finally:
        eng.close()
        other.close()
