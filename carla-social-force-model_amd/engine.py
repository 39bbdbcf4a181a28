"""NumPy-level wrapper of one libsfm_hip handle (one GPU).

Converts the reference's host formats -- float64 (N,3) columns of the PedState record, lists of (P_k,2)
polylines, ``(center, ring)`` obstacle tuples -- to the fp32 SoA / CSR arrays of the C ABI and back.
Used by the drop-in facade (pedestrian_simulation.py, forces.py) and by the device-resident stepper.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FORCE_NAMES, FORCE_TOTAL, SfmLibraryError, f32, fptr, iptr, u8ptr


def interaction_from_table(tbl):
    """Same keys and code-side defaults as the reference getters (forces.py:66-72, :196-206)."""
    return _lib.SfmInteraction(tbl.get("lambda", 2.0), tbl.get("A", 4.5), tbl.get("gamma", 0.35),
                               tbl.get("n", 2.0), tbl.get("n_prime", 3.0), tbl.get("epsilon", 0.005),
                               tbl.get("perception_threshold", 20))


def params_from_config(sfm_config, step_length, honour_file_keys=False):
    """Build the C parameter block from an sfm_config dict, reading it exactly as the reference does.

    Bug-compatible by default: tau is looked up under ``[goal_force]``
    (forces.py:44) and the speed factor under ``max_speed_factor`` (pedestrian_state.py:15), so the stock
    file's ``[acceleration_force] tau`` and ``max_speed_multiplier`` are ignored and 0.5 / 1.3 apply.
    ``honour_file_keys=True`` is the documented deviation that reads those two file keys instead.
    The parameter table of an enabled force is mandatory (KeyError, forces.py:66,134,197-199)."""
    act = sfm_config["forces"]
    p = _lib.SfmParamsC()
    p.use_ped_radius = int(bool(sfm_config.get("use_ped_radius", False)))
    p.max_speed_factor = sfm_config.get("max_speed_factor", 1.3)
    p.tau = sfm_config.get("goal_force", {}).get("tau", 0.5)
    if honour_file_keys:
        p.max_speed_factor = sfm_config.get("max_speed_multiplier", p.max_speed_factor)
        p.tau = sfm_config.get("acceleration_force", {}).get("tau", p.tau)
    p.step_length = float(step_length)
    for k, name in enumerate(FORCE_NAMES):
        p.enabled[k] = int(bool(act.get(name, False)))
    for bad in ("ped_repulsive_force", "space_repulsive_force"):
        if act.get(bad, False):   # pedestrian_simulation.py:49-53 reference classes that do not exist
            raise AttributeError(f"module 'forces' has no attribute for '{bad}'")
    dflt = _lib.SfmInteraction(2.0, 4.5, 0.35, 2.0, 3.0, 0.005, 20.0)
    p.pedestrian = interaction_from_table(sfm_config["pedestrian_force"]) if p.enabled[1] else dflt
    if p.enabled[2]:
        tbl = sfm_config["border_force"]
        p.border_a, p.border_b = tbl.get("a", 3.0), tbl.get("b", 0.1)
    else:
        p.border_a, p.border_b = 3.0, 0.1
    p.static_obstacle = interaction_from_table(sfm_config["static_obstacle_force"]) if p.enabled[3] else dflt
    p.dynamic_obstacle = interaction_from_table(sfm_config["dynamic_obstacle_force"]) if p.enabled[4] else dflt
    return p


def _csr(polys):
    off = np.zeros(len(polys) + 1, dtype=np.int32)
    for k, pl in enumerate(polys):
        off[k + 1] = off[k] + len(pl)
    if len(polys) and off[-1] > 0:
        pts = np.concatenate([np.asarray(pl, dtype=np.float64).reshape(-1, 2) for pl in polys], axis=0)
    else:
        pts = np.zeros((0, 2))
    return off, f32(pts[:, 0]), f32(pts[:, 1])


class SfmEngine:
    """One libsfm_hip handle.  Raises SfmLibraryError on any failure; never falls back to the CPU."""

    def __init__(self, sfm_config, step_length, device=0, honour_file_keys=False, stream=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.params = params_from_config(sfm_config, step_length, honour_file_keys)
        rc = self._lib.sfm_create(C.byref(self.params), int(device), C.byref(self._h))
        if rc != 0:
            msg = self._lib.sfm_last_error(None)
            self._h = C.c_void_p()
            raise SfmLibraryError(f"sfm_create failed ({rc}): {msg.decode() if msg else '?'}")
        self.n = 0
        self.shard = (0, 0)
        self.planar = True
        self._z0 = 0.0            # a flat crowd's common z (the device only holds x / y of a planar crowd)
        if stream is not None:
            self.set_stream(stream)

    # ---- plumbing ---------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.sfm_last_error(self._h)
            raise SfmLibraryError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.sfm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        self._check(self._lib.sfm_set_stream(self._h, C.c_void_p(int(stream_ptr))), "sfm_set_stream")

    def set_params(self, sfm_config, step_length, honour_file_keys=False):
        self.params = params_from_config(sfm_config, step_length, honour_file_keys)
        self._check(self._lib.sfm_set_params(self._h, C.byref(self.params)), "sfm_set_params")

    # ---- geometry ---------------------------------------------------------------------------------
    def set_borders(self, borders, centers, lengths):
        """borders: list of (P_k,2); centers (K,2); lengths (K,)  (forces.py:127-132)."""
        K = len(borders)
        if K == 0:
            self._check(self._lib.sfm_set_borders(self._h, 0, None, None, None, None, None, None), "sfm_set_borders")
            return
        off, px, py = _csr(borders)
        c = np.asarray(centers, dtype=np.float64).reshape(K, 2)
        cx, cy, ln = f32(c[:, 0]), f32(c[:, 1]), f32(np.asarray(lengths, dtype=np.float64).reshape(K))
        self._check(self._lib.sfm_set_borders(self._h, K, iptr(off), fptr(px), fptr(py), fptr(cx), fptr(cy), fptr(ln)),
                    "sfm_set_borders")

    def _obstacles(self, obstacles):
        cs = np.array([np.asarray(c, dtype=np.float64)[:2] for c, _ in obstacles], dtype=np.float64).reshape(-1, 2)
        off, px, py = _csr([r for _, r in obstacles])
        return off, px, py, f32(cs[:, 0]), f32(cs[:, 1])

    def set_static_obstacles(self, obstacles):
        """obstacles: list of (center(2), ring(P,2))  (forces.py:285-288)."""
        if obstacles is None or len(obstacles) == 0:
            self._check(self._lib.sfm_set_static_obstacles(self._h, 0, None, None, None, None, None),
                        "sfm_set_static_obstacles")
            return
        off, px, py, cx, cy = self._obstacles(obstacles)
        self._check(self._lib.sfm_set_static_obstacles(self._h, len(obstacles), iptr(off), fptr(px), fptr(py),
                                                       fptr(cx), fptr(cy)), "sfm_set_static_obstacles")

    def set_dynamic_obstacles(self, obstacles, velocities=None):
        """obstacles as above + velocities (M,2) or None (zeros, forces.py:212-213).  The simulator reports the vehicles every tick
        (run_simulation.py:95, obstacles.py:297-329) with rings of unchanged sizes: offsets and fp32 buffers are kept across calls
        and only refilled."""
        if obstacles is None or len(obstacles) == 0:
            self._check(self._lib.sfm_set_dynamic_obstacles(self._h, 0, None, None, None, None, None, None, None),
                        "sfm_set_dynamic_obstacles")
            return
        M = len(obstacles)
        rings = [r for _, r in obstacles]
        lens = tuple(len(r) for r in rings)
        c = getattr(self, "_dyn_bufs", None)
        if c is None or c[0] != lens:
            off = np.zeros(M + 1, dtype=np.int32)
            off[1:] = np.cumsum(lens)
            P = int(off[-1])
            c = self._dyn_bufs = (lens, off, np.empty((max(P, 1), 2)), np.empty(max(P, 1), np.float32), np.empty(max(P, 1), np.float32),
                                  *(np.empty(M, np.float32) for _ in range(4)))
        _, off, pts, px, py, cx, cy, vx, vy = c
        P = int(off[-1])
        if P:
            np.concatenate([np.reshape(r, (-1, 2)) for r in rings], axis=0, out=pts[:P])
            px[:P] = pts[:P, 0]
            py[:P] = pts[:P, 1]
        cs = np.array([np.asarray(ctr, dtype=np.float64)[:2] for ctr, _ in obstacles], dtype=np.float64).reshape(M, 2)
        cx[:] = cs[:, 0]
        cy[:] = cs[:, 1]
        if velocities is not None:
            v = np.asarray(velocities, dtype=np.float64).reshape(M, 2)
            vx[:] = v[:, 0]
            vy[:] = v[:, 1]
        self._check(self._lib.sfm_set_dynamic_obstacles(self._h, M, iptr(off), fptr(px), fptr(py), fptr(cx), fptr(cy),
                                                        fptr(vx) if velocities is not None else None,
                                                        fptr(vy) if velocities is not None else None), "sfm_set_dynamic_obstacles")

    def set_dynamic_vehicles(self, positions, rings, velocities):
        """The simulator's per-tick vehicle report (obstacles.py:297-329: centres, ring point arrays, velocities) through
        sfm_set_dynamic_obstacles_packed: one concatenate into a kept fp32 buffer, two array writes, one call with kept addresses.
        The library keeps a small report staged and lets the next state upload's launch spread it over the device arrays (or the
        next call that reads them, whichever comes first)."""
        M = len(rings)
        if M == 0:
            return self.set_dynamic_obstacles(None)
        lens = tuple(map(len, rings))
        c = getattr(self, "_veh_bufs", None)
        if c is None or c[0] != lens:
            off = np.zeros(M + 1, dtype=np.int32)
            off[1:] = np.cumsum(lens)
            pts, cv = np.empty((max(int(off[-1]), 1), 2), np.float32), np.zeros((M, 4), np.float32)
            c = self._veh_bufs = (lens, off, pts, cv, off.ctypes.data, pts.ctypes.data, cv.ctypes.data)
            self._dyn_shape = (M, int(off[-1]), off)
        _, off, pts, cv, p_off, p_pts, p_cv = c
        if off[-1]:
            np.concatenate(rings, axis=0, out=pts[:off[-1]], casting="same_kind")
        cv[:, 0:2] = np.asarray(positions, dtype=np.float64).reshape(M, -1)[:, :2]
        cv[:, 2:4] = 0.0 if velocities is None else np.asarray(velocities, dtype=np.float64).reshape(M, 2)
        self._check(self._lib.sfm_set_dynamic_obstacles_packed(self._h, M, p_off, p_pts, p_cv), "sfm_set_dynamic_obstacles_packed")

    def set_dynamic_boxes(self, centers, yaws, extents, velocities, resolution=0.1):
        """Vehicles as oriented boxes (centre (M,2), yaw rad (M,), half-extents (M,2), velocity (M,2)); ring
        points are generated and advanced on the device (CARLA-free runs).  Ring-local offsets follow
        generate_ellipse_border (obstacles.py:269-281)."""
        from .scenarios import ring_local_offsets
        M = len(centers)
        if M == 0:
            self._check(self._lib.sfm_set_dynamic_boxes(self._h, 0, *([None] * 9)), "sfm_set_dynamic_boxes")
            return
        locs = [ring_local_offsets(ex, ey, resolution) for ex, ey in np.asarray(extents, dtype=np.float64).reshape(M, 2)]
        off, ux, uy = _csr(locs)
        c = np.asarray(centers, dtype=np.float64).reshape(M, 2)
        v = np.asarray(velocities, dtype=np.float64).reshape(M, 2)
        yaw = np.asarray(yaws, dtype=np.float64).reshape(M)
        arrs = [f32(c[:, 0]), f32(c[:, 1]), f32(np.cos(yaw)), f32(np.sin(yaw)), f32(v[:, 0]), f32(v[:, 1])]
        self._dyn_shape = (M, int(off[-1]), off)
        self._check(self._lib.sfm_set_dynamic_boxes(self._h, M, iptr(off), fptr(ux), fptr(uy), *(fptr(a) for a in arrs)),
                    "sfm_set_dynamic_boxes")

    def dynamic_obstacles(self):
        """Current device-side vehicles as the reference's list of (center, ring) tuples."""
        M, P, off = self._dyn_shape
        cx, cy, px, py = np.zeros(M, np.float32), np.zeros(M, np.float32), np.zeros(P, np.float32), np.zeros(P, np.float32)
        self._check(self._lib.sfm_download_dynamic_obstacles(self._h, fptr(cx), fptr(cy), fptr(px), fptr(py)),
                    "sfm_download_dynamic_obstacles")
        pts = np.stack([px, py], axis=1).astype(np.float64)
        return [(np.array([cx[k], cy[k]], dtype=np.float64), pts[off[k]:off[k + 1]]) for k in range(M)]

    # ---- state ------------------------------------------------------------------------------------
    def upload_state(self, loc, vel, waypoint, target_speed, radius=None, crossing=None, planar=None):
        """loc, vel, waypoint: (N,3) float; target_speed, radius: (N,); crossing: (N,) bool.
        ``planar`` None = auto: the 2-D kernel is used iff all z are equal and all v_z are 0 (then the
        3-component formulas of forces.py:74-117 / stateutils.py:18-23 reduce to it exactly)."""
        loc = np.asarray(loc, dtype=np.float64).reshape(-1, 3)
        vel = np.asarray(vel, dtype=np.float64).reshape(-1, 3)
        wp = np.asarray(waypoint, dtype=np.float64).reshape(-1, 3)
        n = loc.shape[0]
        if planar is None:
            planar = n == 0 or (bool(np.all(loc[:, 2] == loc[0, 2])) and not bool(np.any(vel[:, 2] != 0.0)))
        self.planar = planar
        self._z0 = float(loc[0, 2]) if n else 0.0
        x, y, vx, vy = f32(loc[:, 0]), f32(loc[:, 1]), f32(vel[:, 0]), f32(vel[:, 1])
        z, vz = (None, None) if planar else (f32(loc[:, 2]), f32(vel[:, 2]))
        wx, wy = f32(wp[:, 0]), f32(wp[:, 1])
        ts = f32(target_speed)
        rr = None if radius is None else f32(radius)
        cm = None if crossing is None else np.ascontiguousarray(crossing, dtype=np.uint8)
        self._check(self._lib.sfm_upload_state(self._h, n, fptr(x), fptr(y), fptr(z), fptr(vx), fptr(vy), fptr(vz),
                                               fptr(wx), fptr(wy), fptr(ts), fptr(rr), u8ptr(cm)), "sfm_upload_state")
        self.n = n
        self.shard = (0, n)

    def step_packed(self, rows, zvz, v_out, integrate=False, redraw=False):
        """One host-in-the-loop tick in one call (sfm_step_packed): ``rows`` float32 (N, 9) = {x, y, vx, vy, waypoint x, waypoint y,
        target_speed, radius, border-force-off flag}, ``zvz`` float32 (N, 2) = {z, vz} or None (planar crowd), v' into ``v_out``
        float32 (N, 3).  All three C-contiguous; nothing is allocated or converted here."""
        n = rows.shape[0]
        self.planar = zvz is None
        # (the three buffers are prefixes of arrays the caller keeps: their addresses are looked up once per array, not per tick)
        key = (id(rows.base), id(v_out.base))
        if getattr(self, "_step_key", None) != key:
            self._step_key, self._step_ptrs = key, (fptr(rows), fptr(v_out))
        p_rows, p_out = self._step_ptrs
        rc = self._lib.sfm_step_packed(self._h, n, p_rows, None if zvz is None else fptr(zvz),
                                       (_lib.TICK_INTEGRATE if integrate else 0) | (_lib.TICK_REDRAW_WAYPOINTS if redraw else 0), p_out)
        if rc != 0:
            self._check(rc, "sfm_step_packed")
        self.n = n
        self.shard = (0, n)

    def step_records(self, records, n, border_off, v_out, planar_tolerance=None, integrate=False, redraw=False):
        """One host-in-the-loop tick straight from the pedestrian records (sfm_step_records, ABI 5): ``records`` the structured array
        PedestrianState keeps (fields loc, vel, next_waypoint float64 (3,), radius, target_speed float64; any stride), its first ``n`` rows;
        ``border_off`` bool / uint8 (n,) or None; v' into ``v_out`` float32 (n, 3).  Returns whether the planar bodies ran."""
        key = (id(records), id(v_out.base))
        if getattr(self, "_rec_key", None) != key:
            f = records.dtype.fields
            off = np.array([f[k][1] for k in ("loc", "vel", "next_waypoint", "radius", "target_speed")], dtype=np.int32)
            for k in ("loc", "vel", "next_waypoint", "radius", "target_speed"):
                if f[k][0].base != np.dtype(np.float64):
                    raise TypeError(f"record field {k!r} must be float64")
            # (the arrays themselves are kept: an id can only be trusted while its object is alive)
            self._rec_key, self._rec_ptrs = key, (records.ctypes.data, int(records.strides[0]), off, off.ctypes.data, fptr(v_out), C.c_int32(0), records, v_out.base)
        p_rec, stride, _off, p_off, p_out, flag = self._rec_ptrs[:6]
        rc = self._lib.sfm_step_records(self._h, n, p_rec, stride, p_off, None if border_off is None else border_off.ctypes.data,
                                        -1.0 if planar_tolerance is None else float(planar_tolerance),
                                        (_lib.TICK_INTEGRATE if integrate else 0) | (_lib.TICK_REDRAW_WAYPOINTS if redraw else 0), p_out, C.byref(flag))
        if rc != 0:
            self._check(rc, "sfm_step_records")
        self.n = n
        self.shard = (0, n)
        self.planar = bool(flag.value)
        self._z0 = None
        return self.planar

    def set_shard(self, i_begin, i_end):
        self._check(self._lib.sfm_set_shard(self._h, int(i_begin), int(i_end)), "sfm_set_shard")
        self.shard = (int(i_begin), int(i_end))

    def set_partition(self, gx, gy, bounds=None):
        """Block-major row packing for sharded runs (sfm_set_partition): gx x gy blocks, block b = rows [bounds[b], bounds[b+1])."""
        b = None if bounds is None else np.ascontiguousarray(bounds, dtype=np.int32)
        self._check(self._lib.sfm_set_partition(self._h, int(gx), int(gy), iptr(b)), "sfm_set_partition")

    def set_waypoint_stream(self, seed, world_side, arrive_threshold=2.0):
        self._check(self._lib.sfm_set_waypoint_stream(self._h, int(seed) & 0xFFFFFFFF, float(world_side),
                                                      float(arrive_threshold)), "sfm_set_waypoint_stream")

    def set_mode_fsm(self, mode_managers, waypoint_lists, despawn_on_arrival=True, sim_time0=0.0, first_vehicle_extent=None):
        """Hand the per-pedestrian mode objects (PedModeManager: current_mode, target_speed, initial_target_speed,
        crossing_speed, crossing_safety_margin, next_mode_time) and the remaining waypoints of each pedestrian --
        ``waypoint_lists[i]`` = [(waypoint(2 or 3), crossing_road), ...] like waypoint_dict (run_simulation.py:120-126) --
        to the device, so ``run`` executes modes, gap acceptance, arrivals and despawns without the host."""
        n = len(mode_managers)
        f = lambda attr: f32([float(getattr(m, attr)) for m in mode_managers])
        mode = np.ascontiguousarray([int(m.current_mode) for m in mode_managers], dtype=np.uint8)
        off = np.zeros(n + 1, dtype=np.int32)
        for i, lst in enumerate(waypoint_lists):
            off[i + 1] = off[i] + len(lst)
        flat = [w for lst in waypoint_lists for w in lst]
        wx = f32([float(w[0][0]) for w in flat]); wy = f32([float(w[0][1]) for w in flat])
        wc = np.ascontiguousarray([1 if w[1] else 0 for w in flat], dtype=np.uint8)
        ext = None if first_vehicle_extent is None else f32(np.asarray(first_vehicle_extent, dtype=np.float64)[:2])
        arrs = [f("target_speed"), f("initial_target_speed"), f("crossing_speed"), f("crossing_safety_margin"), f("next_mode_time")]
        self._check(self._lib.sfm_set_mode_fsm(self._h, n, u8ptr(mode), *(fptr(a) for a in arrs), iptr(off),
                                               fptr(wx) if len(flat) else None, fptr(wy) if len(flat) else None,
                                               u8ptr(wc) if len(flat) else None, int(bool(despawn_on_arrival)),
                                               float(sim_time0), fptr(ext)), "sfm_set_mode_fsm")

    def modes(self):
        """(mode uint8[N] with 255 = despawned, mode target speed float32[N], queue cursor int32[N])."""
        m, t, c = np.zeros(self.n, np.uint8), np.zeros(self.n, np.float32), np.zeros(self.n, np.int32)
        self._check(self._lib.sfm_download_modes(self._h, u8ptr(m), fptr(t), iptr(c)), "sfm_download_modes")
        return m, t, c

    # ---- stepping ---------------------------------------------------------------------------------
    @staticmethod
    def _flags(integrate, redraw, record):
        return ((_lib.TICK_INTEGRATE if integrate else 0) | (_lib.TICK_REDRAW_WAYPOINTS if redraw else 0)
                | (_lib.TICK_RECORD_FORCES if record else 0))

    def tick(self, integrate=False, redraw=False, record=False):
        self._check(self._lib.sfm_tick(self._h, self._flags(integrate, redraw, record)), "sfm_tick")

    def run(self, ticks, redraw=False, record=False):
        self._check(self._lib.sfm_run(self._h, int(ticks), self._flags(True, redraw, record)), "sfm_run")

    def tick_begin(self, redraw=False):
        """First half of an integrating tick of a shard: everything that only needs this handle's own rows (sfm_tick_begin)."""
        self._check(self._lib.sfm_tick_begin(self._h, self._flags(True, redraw, False)), "sfm_tick_begin")

    def tick_end(self, redraw=False):
        """Second half, once the other ranks' rows are current (sfm_tick_end)."""
        self._check(self._lib.sfm_tick_end(self._h, self._flags(True, redraw, False)), "sfm_tick_end")

    def run_recorded(self, ticks, stride=1, redraw=False):
        """``run`` that records {x, y, vx, vy} of every pedestrian before tick 0, stride, 2*stride, ...
        Returns (frames[F, N, 4] float32, tick_index[F])."""
        n_frames = (int(ticks) + int(stride) - 1) // int(stride) if ticks > 0 else 0
        frames = np.zeros((n_frames, self.n, 4), dtype=np.float32)
        got = C.c_int(0)
        self._check(self._lib.sfm_run_recorded(self._h, int(ticks), self._flags(True, redraw, False), int(stride),
                                               fptr(frames.reshape(-1)) if frames.size else None, n_frames, C.byref(got)),
                    "sfm_run_recorded")
        return frames[:got.value], np.arange(got.value) * int(stride)

    # ---- results ----------------------------------------------------------------------------------
    def velocities(self):
        """(N,3) float64; rows outside this handle's shard are NaN."""
        n = self.n
        vx, vy, vz = (np.full(n, np.nan, np.float32) for _ in range(3))
        self._check(self._lib.sfm_download_velocities(self._h, fptr(vx), fptr(vy), fptr(vz)), "sfm_download_velocities")
        return np.stack([vx, vy, vz], axis=1).astype(np.float64)

    def state(self):
        n = self.n
        a = {k: np.full(n, np.nan, np.float32) for k in ("x", "y", "z", "vx", "vy", "vz", "wx", "wy")}
        self._check(self._lib.sfm_download_state(self._h, *(fptr(a[k]) for k in ("x", "y", "z", "vx", "vy", "vz", "wx", "wy"))),
                    "sfm_download_state")
        if self.planar:      # rows are in the library's own order: the owned pedestrians are the ones a value came back for
            if self._z0 is None:      # (last upload was sfm_step_records: the flat crowd's z is in the caller's records)
                self._z0 = float(self._rec_ptrs[6]["loc"][0, 2]) if self.n else 0.0
            a["z"][~np.isnan(a["x"])] = np.float32(self._z0)
        loc = np.stack([a["x"], a["y"], a["z"]], axis=1).astype(np.float64)
        vel = np.stack([a["vx"], a["vy"], a["vz"]], axis=1).astype(np.float64)
        wp = np.stack([a["wx"], a["wy"]], axis=1).astype(np.float64)
        return loc, vel, wp

    def forces(self, which):
        """(N,3) float64 of one force (name or index; 'total' = 5) from the last recorded tick."""
        if isinstance(which, str):
            which = FORCE_TOTAL if which == "total" else FORCE_NAMES.index(which)
        n = self.n
        fx, fy, fz = (np.full(n, np.nan, np.float32) for _ in range(3))
        self._check(self._lib.sfm_download_forces(self._h, int(which), fptr(fx), fptr(fy), fptr(fz)), "sfm_download_forces")
        return np.stack([fx, fy, fz], axis=1).astype(np.float64)

    def arrived(self, threshold):
        m = np.zeros(self.n, dtype=np.uint8)
        self._check(self._lib.sfm_get_arrived(self._h, float(threshold), u8ptr(m)), "sfm_get_arrived")
        return m.astype(bool)

    def draw_counts(self):
        c = np.zeros(self.n, dtype=np.uint32)
        self._check(self._lib.sfm_download_draw_counts(self._h, c.ctypes.data_as(_lib._U32)), "sfm_download_draw_counts")
        return c

    def packed_state_ptr(self):
        npad = C.c_int(0)
        p = self._lib.sfm_packed_state_ptr(self._h, C.byref(npad))
        return p, npad.value

    def packed_z_ptr(self):
        return self._lib.sfm_packed_z_ptr(self._h)

    def row_data_ptr(self, which):
        """(device pointer, bytes per row) of a per-row array a sharded driver gathers before ``resort``:
        0 = {waypoint x, waypoint y, target speed, radius}, 1 = waypoint draw counters."""
        b = C.c_int(0)
        p = self._lib.sfm_row_data_ptr(self._h, int(which), C.byref(b))
        return p, b.value

    def resort(self):
        """Re-pack the rows spatially now (sfm_resort)."""
        self._check(self._lib.sfm_resort(self._h), "sfm_resort")

    def set_timing(self, enable):
        """HIP-event bracket of tick()/run() on or off (sfm_set_timing): off saves ~11 us per call; ``timing()`` then raises."""
        self._check(self._lib.sfm_set_timing(self._h, 1 if enable else 0), "sfm_set_timing")

    def timing(self):
        """(elapsed_ms, ticks, launches) of the last tick()/run(), HIP events on the handle's stream."""
        ms, t, l = C.c_float(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.sfm_get_timing(self._h, C.byref(ms), C.byref(t), C.byref(l)), "sfm_get_timing")
        return ms.value, t.value, l.value

    def profile_dominant_kernel(self, reps=100):
        """Average microseconds per launch of the pedestrian-pair kernel alone (HIP events, state not advanced)."""
        us = C.c_float(0)
        self._check(self._lib.sfm_profile_dominant_kernel(self._h, int(reps), C.byref(us)), "sfm_profile_dominant_kernel")
        return us.value

    def kernel_variant(self):
        return self._lib.sfm_kernel_variant(self._h).decode()

    def pair_work(self):
        """(tile-pair work items, Moussaid terms evaluated) of the symmetric pair kernel in the last tick."""
        items, terms = C.c_longlong(0), C.c_longlong(0)
        self._check(self._lib.sfm_get_pair_work(self._h, C.byref(items), C.byref(terms)), "sfm_get_pair_work")
        return items.value, terms.value
