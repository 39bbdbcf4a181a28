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
        os.environ["SFM_SYM"] = "0"                     # the ordered kernel is what sharded runs use
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
        loc, vel, wp = st.gather_state()
        plain = SfmEngine(cfg, 0.05)
        plain.set_borders(sc.borders, sc.border_centers, sc.border_lengths)
        plain.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
        plain.set_waypoint_stream(sc.seed, sc.world_side, 2.0)
        plain.run(5, redraw=True)
        loc2, vel2, wp2 = plain.state()
        assert np.array_equal(loc, loc2) and np.array_equal(vel, vel2) and np.array_equal(wp, wp2)
        plain.close()
        eng.close()
    finally:
        os.environ.pop("SFM_SYM", None)
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
