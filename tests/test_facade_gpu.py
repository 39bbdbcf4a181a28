"""GPU tests of the drop-in boundary: the PedestrianSimulation / Force facade driven the way
run_simulation.py drives the reference (spawn tuples, update_dynamic_obstacles, tick, get_new_velocities,
get_arrived_peds, update_next_waypoint), checked against the reference's golden outputs."""
import numpy as np
import pytest

import _golden_io as gio
import _parity as P
from carla_social_force_model_amd import forces as hip_forces
from carla_social_force_model_amd.ped_mode_manager import PedMode, PedModeManager
from carla_social_force_model_amd.pedestrian_simulation import PedestrianSimulation
from oracle import sfm_oracle as O

pytestmark = pytest.mark.gpu
CASES = {p.split("/")[-1][:-4]: p for p in gio.list_cases()}


def _mode_manager(name, speed, m):
    mm = PedModeManager(name, float(speed), PedMode.WALKING_SIDEWALK, 1.5, 1.5)
    steps = {0: [PedMode.IDLE], 2: [PedMode.CROSSING_ROAD] * 2, 3: [PedMode.CROSSING_ROAD] * 2 + [PedMode.WALKING_SIDEWALK],
             4: [PedMode.CROSSING_ROAD]}.get(int(m), [])
    for s in steps:
        mm.set_mode(s)
    assert int(mm.current_mode) == int(m)
    return mm


def _build(c):
    info = [[c.border_centers[k], float(c.border_lengths[k])] for k in range(len(c.borders))]
    sim = PedestrianSimulation(c.borders, info, c.static_obstacles, c.cfg, c.dt)
    for i in range(c.n):
        name = f"ped_{i}"
        sim.spawn_pedestrian((name, 100 + i, c.loc[i], c.vel[i], c.waypoint[i],
                              _mode_manager(name, c.target_speed[i], c.mode[i]), float(c.radius[i]), float(c.target_speed[i])))
    if len(c.dynamic_obstacles):
        m = len(c.dynamic_obstacles)
        sim.update_dynamic_obstacles((list(range(m)), [o[0] for o in c.dynamic_obstacles], [0.0] * m,
                                      [v for v in c.dynamic_vel], [np.array([2.4, 1.0])] * m,
                                      [o[1] for o in c.dynamic_obstacles]))
    return sim


def _diag(c):
    prm = O.OracleParams.from_config(c.cfg)
    geom = O.Geometry(c.borders, c.border_centers, c.border_lengths, c.static_obstacles, c.dynamic_obstacles, c.dynamic_vel)
    d = {}
    with np.errstate(all="ignore"):
        O.tick_forces(c.loc, c.vel, c.waypoint, c.z["mode_target_speed"], c.radius, c.crossing, geom, prm,
                      theta_tol=P.THETA_TOL, tie_rel=P.TIE_REL, diag=d)
    return d


@pytest.mark.parametrize("name", ["c1_n64", "modes_n32", "all_s1_n256", "zspread_n64", "radius_n64", "single_ped"])
def test_tick_through_facade(name):
    c = gio.Case(CASES[name])
    sim = _build(c)
    try:
        assert sim.get_new_velocities() is None
        # per-force view of the fused engine (Force.get_force contract), on the untouched spawn state
        sim.peds.apply_current_mode()
        d = _diag(c)
        for fname, f in sim.forces.items():
            P.check_force(fname, f.get_force(sim.peds), c.ref(fname), d[fname][1], d[fname][0])
        sim.tick(0.0)
        nv = sim.get_new_velocities()
        assert nv.dtype.names == ('id', 'vel') and nv['vel'].shape == (c.n, 3)
        assert np.shares_memory(nv, sim.peds.state)                       # the reference's view semantics
        assert np.array_equal(sim.peds.vel(), nv['vel'])
        P.check_velocity(nv['vel'], c.ref("new_vel"), _diag(c)["total"][0], c.dt)
        assert np.array_equal(sim.peds.target_speed(), c.z["mode_target_speed"])
        assert list(sim.get_arrived_peds(2.0)) == [f"ped_{i}" for i in np.nonzero(c.ref("arrived"))[0]]
        assert 0.0 in sim.get_states() and sim.get_states()[0.0]['mode'][0] == int(c.mode[0])
    finally:
        sim.close()


def test_empty_simulation_and_despawn():
    c = gio.Case(CASES["all_s0_n16"])
    info = [[c.border_centers[k], float(c.border_lengths[k])] for k in range(len(c.borders))]
    sim = PedestrianSimulation(c.borders, info, c.static_obstacles, c.cfg, c.dt)
    try:
        sim.tick(0.0)                                                      # no pedestrians: no-op (:60-61)
        assert sim.get_new_velocities() is None and sim.get_arrived_peds(2.0) == []
        for i in range(3):
            sim.spawn_pedestrian((f"ped_{i}", i, c.loc[i], c.vel[i], c.waypoint[i],
                                  _mode_manager(f"ped_{i}", 1.2, 1), 0.3, 1.2))
        sim.tick(0.05)
        sim.destroy_pedestrian("ped_1")
        sim.tick(0.10)
        assert sim.peds.size() == 2 and list(sim.peds.name()) == ["ped_0", "ped_2"]
        assert np.isfinite(sim.get_new_velocities()['vel']).all()
        for i in (0, 2):
            sim.destroy_pedestrian(f"ped_{i}")
        sim.tick(0.15)                                                     # empty again
    finally:
        sim.close()


def test_standalone_force_classes_match_golden():
    """forces.AccelerationForce / PedestrianForce / BorderForce / ObstacleForce slotted in one by one."""
    c = gio.Case(CASES["c1_n64"])
    sim = _build(c)                                                        # only used as a PedState factory here
    sim.peds.apply_current_mode()
    d = _diag(c)
    info = [[c.border_centers[k], float(c.border_lengths[k])] for k in range(len(c.borders))]
    fs = {"acceleration_force": hip_forces.AccelerationForce(c.dt, c.cfg),
          "pedestrian_force": hip_forces.PedestrianForce(c.dt, c.cfg),
          "border_force": hip_forces.BorderForce(c.dt, c.cfg, c.borders, info),
          "static_obstacle_force": hip_forces.ObstacleForce(c.dt, c.cfg),
          "dynamic_obstacle_force": hip_forces.ObstacleForce(c.dt, c.cfg, True)}
    try:
        assert np.array_equal(fs["dynamic_obstacle_force"].get_force(sim.peds), np.zeros((c.n, 3)))   # before any update
        fs["static_obstacle_force"].update_obstacles(c.static_obstacles)
        fs["dynamic_obstacle_force"].update_obstacles(c.dynamic_obstacles)
        fs["dynamic_obstacle_force"].update_obstacle_velocities(c.dynamic_vel)
        for name, f in fs.items():
            got = f.get_force(sim.peds, debug=True)
            assert got.shape == (c.n, 3) and got.dtype == np.float64
            P.check_force(name, got, c.ref(name), d[name][1], d[name][0])
    finally:
        for f in fs.values():
            f.close()
        sim.close()


def test_host_loop_trajectory_tracks_reference():
    """20 ticks of the CARLA-free host loop through the facade (tick -> arrival -> waypoint swap -> move)
    against the reference trajectory; fp32 forces vs float64, so a small, growth-aware bound."""
    c = gio.Case(CASES["all_s2_n64"])
    sim = _build(c)
    queue, used = c.z["wp_queue"], np.zeros(c.n, dtype=int)
    try:
        sim.record_states = False
        for k in range(c.ref("traj_loc").shape[0]):
            sim.tick(k * c.dt)
            for name in sim.get_arrived_peds(2.0):
                i = int(name.split("_")[1])
                used[i] += 1
                wp = np.array([*queue[i, (used[i] - 1) % queue.shape[1]], 0.0])
                sim.peds.update_next_waypoint(name, (wp, False))
            sim.peds.state['loc'] = sim.peds.state['loc'] + c.dt * sim.peds.state['vel']
            err = np.max(np.linalg.norm(sim.peds.loc() - c.ref("traj_loc")[k], axis=1))
            assert err <= 2e-5 * (k + 1), f"tick {k}: {err:.2e}"
    finally:
        sim.close()


def test_planar_tolerance_is_an_opt_in_deviation():
    """Default: any spread in z keeps the exact 3-D kernel (zspread_n64 goes through test_tick_through_facade that way).
    With ``planar_tolerance`` a crowd on nearly flat ground (z within 2 cm, |v_z| ~ 1e-3) is handed to the planar path --
    the symmetric kernel at this size -- and agrees with the reference evaluated on the exactly flat crowd."""
    from carla_social_force_model_amd import scenarios
    from carla_social_force_model_amd.config import default_sfm_config
    cfg = default_sfm_config(("acceleration_force", "pedestrian_force"))
    sc = scenarios.make_scenario(512, 909)
    rng = np.random.default_rng(5)
    bumpy_loc, bumpy_vel = sc.loc.copy(), sc.vel.copy()
    bumpy_loc[:, 2] = np.float32(0.2) + rng.uniform(-0.01, 0.01, sc.n).astype(np.float32)
    bumpy_vel[:, 2] = rng.uniform(-1e-3, 1e-3, sc.n).astype(np.float32)
    prm = O.OracleParams.from_config(cfg)
    with np.errstate(all="ignore"):
        _, flat_total, _ = O.tick_forces(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, np.zeros(sc.n, bool),
                                         O.Geometry(), prm)
    flat_v = O.new_velocities(sc.vel, flat_total, sc.target_speed, 0.05)
    variants = {}
    for tol in (None, 0.05):
        sim = PedestrianSimulation([], [], [], cfg, 0.05, planar_tolerance=tol)
        try:
            for i in range(sc.n):
                sim.spawn_pedestrian((f"ped_{i}", i, bumpy_loc[i], bumpy_vel[i], sc.waypoint[i],
                                      _mode_manager(f"ped_{i}", sc.target_speed[i], 1), 0.3, float(sc.target_speed[i])))
            sim.tick(0.0)
            variants[tol] = (sim.engine.kernel_variant(), sim.get_new_velocities()['vel'].copy())
        finally:
            sim.close()
    # default: the exact 3-D evaluation (since round 3 on the symmetric kernel's 3-D body at this size)
    with np.errstate(all="ignore"):
        _, bumpy_total, _ = O.tick_forces(bumpy_loc, bumpy_vel, sc.waypoint, sc.target_speed, sc.radius, np.zeros(sc.n, bool),
                                          O.Geometry(), prm)
    bumpy_v = O.new_velocities(bumpy_vel, bumpy_total, sc.target_speed, 0.05)
    assert np.allclose(variants[None][1], bumpy_v, rtol=1e-5, atol=1e-7)
    assert "sym" in variants[0.05][0]
    v = variants[0.05][1]
    assert np.all(v[:, 2] == 0.0)
    assert np.allclose(v[:, :2], flat_v[:, :2], rtol=1e-5, atol=1e-7)
    # and the deviation is small but real: the exact 3-D evaluation of the bumpy crowd is not the flat one
    assert not np.array_equal(variants[None][1][:, :2], v[:, :2])


def test_mode_changes_between_ticks_are_honoured():
    """Round 4: the facade keeps target speeds and the border-force mask as cached arrays (host_state.ModeWatch) and ticks only IDLE
    pedestrians' FSMs.  Whatever a caller does to a mode between two ticks -- update_next_waypoint onto a crossing (run_simulation.py:
    124-126), a mode object's target_speed written directly, an IDLE pedestrian whose waiting time runs out during the run -- must
    reach the device at the next tick: v' against the oracle fed with the target speeds / mask the reference's tick would have
    used (pedestrian_simulation.py:63-65, forces.py:176-177)."""
    c = gio.Case(CASES["all_s1_n256"])
    sim = _build(c)
    prm = O.OracleParams.from_config(c.cfg)
    geom = O.Geometry(c.borders, c.border_centers, c.border_lengths, c.static_obstacles, c.dynamic_obstacles, c.dynamic_vel)
    sim.record_states = False
    sim.dyn_obstacles = []                 # (no gap acceptance in this test: a pedestrian sent to CHECKING_TRAFFIC crosses at once, :70)
    try:
        mgr = sim.peds.mode()
        t = 0.0
        for step in range(4):
            if step == 1:
                sim.peds.update_next_waypoint("ped_3", (np.array([1.0, 2.0, 0.0]), True))     # -> CHECKING -> (tick) CROSSING_ROAD
                mgr[7].target_speed = 0.5                                                       # a direct write to the mode object
                mgr[11].set_mode(PedMode.IDLE)                                                  # stands still for waiting_time
            if step == 3:
                t = 6.0                                                                         # ped_11 wakes up: 0.05 + 5 <= 6
            loc, vel, wp = sim.peds.loc().copy(), sim.peds.vel().copy(), sim.peds.next_waypoint().copy()
            sim.tick(t)
            modes = np.array([int(m.current_mode) for m in mgr])
            ts = np.array([float(m.target_speed) for m in mgr])                                 # what the NEXT tick will apply ...
            used_ts = sim.peds.target_speed().copy()                                            # ... and what THIS tick applied
            if step == 1:
                assert modes[3] == PedMode.CROSSING_ROAD and used_ts[3] == 0 and used_ts[7] == 0.5 and used_ts[11] == 0
            if step == 2:
                assert used_ts[3] == pytest.approx(1.5 * c.target_speed[3]) and modes[11] == PedMode.IDLE
            if step == 3:
                assert modes[11] == PedMode.WALKING_SIDEWALK and used_ts[11] == 0 and ts[11] > 0   # the tick reads speeds BEFORE mode.tick (:63-65)
            crossing = (modes == 2) | (modes == 3)
            d = {}
            with np.errstate(all="ignore"):
                _, total, _ = O.tick_forces(loc, vel, wp, used_ts, c.radius, crossing, geom, prm, theta_tol=P.THETA_TOL, tie_rel=P.TIE_REL, diag=d)
            P.check_velocity(sim.get_new_velocities()['vel'], O.new_velocities(vel, total, used_ts, c.dt), d["total"][0], c.dt)
            t += c.dt
    finally:
        sim.close()


def test_step_from_the_records_is_the_packed_step(monkeypatch):
    """sfm_step_records (ABI 5) gathers loc / vel / next_waypoint / radius / target_speed out of PedestrianState's 132-byte records
    (pedestrian_state.py:17-23) inside the library; sfm_step_packed gets the same numbers from six NumPy assignments.  Same v', bit
    for bit: a flat crowd, walkers on uneven ground, the planar_tolerance deviation, border-force-off rows -- through the engine, and the
    facade with either path."""
    from carla_social_force_model_amd import scenarios
    from carla_social_force_model_amd.config import default_sfm_config
    from carla_social_force_model_amd.engine import SfmEngine
    from carla_social_force_model_amd.host_state import PED_STATE_DTYPE

    for n, z_spread, tol in ((37, 0.0, None), (200, 0.4, None), (200, 0.004, 0.01), (130, 0.004, 0.001)):
        sc = scenarios.make_scenario(n, 900 + n, n_borders=12, n_static=6, n_dynamic=3, z_spread=z_spread)
        rec = np.zeros(n + 5, dtype=PED_STATE_DTYPE)              # (capacity beyond n, like the facade's growable buffer)
        rec['loc'][:n], rec['vel'][:n], rec['next_waypoint'][:n] = sc.loc, sc.vel, sc.waypoint
        rec['radius'][:n], rec['target_speed'][:n] = sc.radius, sc.target_speed
        off = np.zeros(n, dtype=bool)
        off[::7] = True
        cfg = default_sfm_config(scenarios.ALL_FORCES)
        a, b = SfmEngine(cfg, 0.05), SfmEngine(cfg, 0.05)
        try:
            for e in (a, b):
                e.set_borders(sc.borders, sc.border_centers, sc.border_lengths)
                e.set_static_obstacles(sc.static_obstacles)
                e.set_dynamic_obstacles(sc.dynamic_obstacles, sc.dynamic_vel)
            va, vb = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
            planar = a.step_records(rec, n, off, va, tol)
            z, vz = sc.loc[:, 2], sc.vel[:, 2]
            flat = bool((z == z[0]).all()) and not bool(vz.any())
            if not flat and tol is not None:
                flat = bool(np.max(np.abs(z - np.median(z))) <= tol and np.max(np.abs(vz)) <= tol)
            assert planar == flat, (n, z_spread, tol)
            rows = np.zeros((n, 9), np.float32)
            rows[:, 0:2], rows[:, 2:4], rows[:, 4:6] = sc.loc[:, :2], sc.vel[:, :2], sc.waypoint[:, :2]
            rows[:, 6], rows[:, 7], rows[:, 8] = sc.target_speed, sc.radius, off
            zvz = None if flat else np.ascontiguousarray(np.stack([z, vz], axis=1), dtype=np.float32)
            b.step_packed(rows, zvz, vb)
            b._z0 = float(z[0])                                       # (what the facade does beside sfm_step_packed)
            assert np.array_equal(va, vb), (n, z_spread, tol)
            la, _, _ = a.state()
            assert np.array_equal(la, b.state()[0]) or not flat       # (a flat crowd's z comes back from the records)
        finally:
            a.close(); b.close()

    c = gio.Case(CASES["all_s0_n64"])
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SFM_FACADE_RECORDS", flag)
        sim = _build(c)
        try:
            assert sim._use_records == (flag == "1")
            for k in range(3):
                sim.tick(0.05 * k)
            out[flag] = np.array(sim.get_new_velocities()['vel'])
        finally:
            sim.close()
    assert np.array_equal(out["1"], out["0"])
