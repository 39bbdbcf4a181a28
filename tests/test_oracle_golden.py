"""Pins the CPU oracle (oracle/sfm_oracle.py) against the golden vectors produced by the reference's own
NumPy modules (tests/golden/make_golden.py).  Gate: 1e-12 relative."""
import numpy as np
import pytest

import _golden_io as gio
from oracle import sfm_oracle as O

CASES = gio.list_cases()
RTOL = 1e-12


def _close(a, b, what, rtol=RTOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), f"{what}: NaN pattern differs"
    scale = max(1.0, np.nanmax(np.abs(b))) if b.size else 1.0
    err = np.nanmax(np.abs(a - b)) / scale if b.size and (~nan_b).any() else 0.0
    assert err <= rtol, f"{what}: rel err {err:.3e}"


def _geom(c):
    return O.Geometry(borders=c.borders, border_centers=c.border_centers, border_lengths=c.border_lengths,
                      static_obstacles=c.static_obstacles, dynamic_obstacles=c.dynamic_obstacles,
                      dynamic_vel=c.dynamic_vel)


def test_fixtures_present():
    assert len(CASES) >= 25


@pytest.mark.parametrize("path", CASES, ids=[p.split("/")[-1][:-4] for p in CASES])
def test_forces_and_velocities(path):
    c = gio.Case(path)
    prm = O.OracleParams.from_config(c.cfg)
    tspeed = c.z["mode_target_speed"]
    with np.errstate(all="ignore"):
        per, total, _ = O.tick_forces(c.loc, c.vel, c.waypoint, tspeed, c.radius, c.crossing, _geom(c), prm)
        v_new = O.new_velocities(c.vel, total, tspeed, c.dt, prm.max_speed_factor)
    for name in O.FORCE_NAMES:
        assert (name in per) == c.has(name)
        if name in per:
            _close(per[name], c.ref(name), f"{c.name}/{name}")
    _close(total, c.ref("total"), f"{c.name}/total")
    _close(v_new, c.ref("new_vel"), f"{c.name}/new_vel")
    assert np.array_equal(O.arrived(c.loc, c.waypoint, 2.0), c.ref("arrived"))


@pytest.mark.parametrize("path", [p for p in CASES if "ref_traj_loc" in np.load(p).files],
                         ids=lambda p: p.split("/")[-1][:-4])
def test_trajectories(path):
    """20 free-running ticks reproduce the reference trajectory (f64 both sides, so no divergence yet)."""
    c = gio.Case(path)
    prm = O.OracleParams.from_config(c.cfg)
    tspeed = c.z["mode_target_speed"]
    queue = c.z["wp_queue"]
    loc, vel, wp = c.loc.copy(), c.vel.copy(), c.waypoint.copy()
    draws = np.zeros(c.n, dtype=np.int64)
    src = lambda ids, d: queue[ids, (d - 1) % queue.shape[1]]   # noqa: E731
    for k in range(c.ref("traj_loc").shape[0]):
        with np.errstate(all="ignore"):
            loc, vel, wp, draws = O.free_step(loc, vel, wp, tspeed, c.radius, c.crossing, draws, _geom(c), prm,
                                              c.dt, 2.0, waypoint_source=src)
        _close(loc, c.ref("traj_loc")[k], f"{c.name}/loc@{k}", 1e-10)
        _close(vel, c.ref("traj_vel")[k], f"{c.name}/vel@{k}", 1e-10)
        _close(wp, c.ref("traj_wp")[k], f"{c.name}/wp@{k}", 1e-12)


def test_coincident_pair_semantics():
    """forces.py:97,105: equal position + equal velocity -> NaN for both; equal position, different
    velocity -> finite (theta = -angle(t))."""
    c = gio.Case([p for p in CASES if p.endswith("coincident_n8.npz")][0])
    ref = c.ref("pedestrian_force")
    assert np.isnan(ref[0]).any() and np.isnan(ref[1]).any()
    assert np.isfinite(ref[4]).all() and np.isfinite(ref[5]).all()


def test_config_bug_compat():
    """tau / max speed factor come from keys the stock file does not define."""
    cfg = {"forces": {"acceleration_force": True}, "acceleration_force": {"tau": 9.0},
           "max_speed_multiplier": 7.0}
    p = O.OracleParams.from_config(cfg)
    assert p.tau == 0.5 and p.max_speed_factor == 1.3
    cfg["goal_force"] = {"tau": 0.25}
    cfg["max_speed_factor"] = 2.0
    p = O.OracleParams.from_config(cfg)
    assert p.tau == 0.25 and p.max_speed_factor == 2.0
    with pytest.raises(KeyError):
        O.OracleParams.from_config({"forces": {"pedestrian_force": True}})


@pytest.mark.parametrize("path", CASES, ids=[p.split("/")[-1][:-4] for p in CASES])
def test_c_oracle_matches_golden(path):
    """The C/OpenMP restatement (large-N oracle and CPU baseline) is gated on the same vectors."""
    from oracle import c_oracle
    c = gio.Case(path)
    prm = O.OracleParams.from_config(c.cfg)
    tspeed = c.z["mode_target_speed"]
    per, total, v_new, _, _ = c_oracle.tick(c.loc, c.vel, c.waypoint, tspeed, c.radius, c.crossing, _geom(c), prm, c.dt,
                                         nthreads=1)
    for name in per:
        _close(per[name], c.ref(name), f"{c.name}/{name}")
    _close(total, c.ref("total"), f"{c.name}/total")
    _close(v_new, c.ref("new_vel"), f"{c.name}/new_vel")
