"""Deterministic synthetic crowds + geometry in the reference's own input formats.

The reference needs a CARLA server to produce its inputs (pedestrian_spawner.py, obstacles.py); the
BASELINE configs are CARLA-free, so this module generates inputs of the same *shape*:

* pedestrians: the 8-field spawn tuple of pedestrian_spawner.py:230-243
  (name, id, loc3, vel3, first_waypoint3, mode, radius, target_speed); target speed 1.2 +- 0.1
  (pedestrian_spawner.py:73,146-147);
* borders: straight polylines sampled every 0.1 m with ``center = line[len//2]`` and
  ``section_length = len(line) * resolution`` (obstacles.py:344-355);
* static / dynamic obstacles: ``(center, ring)`` with an ellipse ring scaled by sqrt(2) and
  ``max(6, int((2*ex + 2*ey) / resolution))`` samples (obstacles.py:269-281, 297-329).

Every coordinate is rounded to fp32 before it is stored in float64: the
device state is fp32, so parity scenarios must be fp32-representable.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

RESOLUTION = 0.1  # m, obstacles.py:25 default


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


@dataclass
class Scenario:
    loc: np.ndarray            # (N,3) f64, fp32-representable
    vel: np.ndarray            # (N,3)
    waypoint: np.ndarray       # (N,3)
    target_speed: np.ndarray   # (N,)
    radius: np.ndarray         # (N,)
    mode: np.ndarray           # (N,) int, PedMode values
    borders: list = field(default_factory=list)            # list of (P_k,2)
    border_centers: np.ndarray = None                      # (K,2)
    border_lengths: np.ndarray = None                      # (K,)
    static_obstacles: list = field(default_factory=list)   # list of (center(2), ring(P,2))
    dynamic_obstacles: list = field(default_factory=list)
    dynamic_vel: np.ndarray = None                         # (M,2)
    dynamic_yaw: np.ndarray = None                         # (M,)
    dynamic_extent: np.ndarray = None                      # (M,2)
    world_side: float = 0.0
    seed: int = 0

    @property
    def n(self):
        return self.loc.shape[0]

    def section_info(self):
        """border_section_info in the form BorderForce indexes it (forces.py:130-132): an object array
        K x [center(2), length] (a ragged list only converts on NumPy < 1.24)."""
        info = np.empty((len(self.borders), 2), dtype=object)
        for k in range(len(self.borders)):
            info[k, 0] = self.border_centers[k]
            info[k, 1] = float(self.border_lengths[k])
        return info


def straight_border(start, end, resolution=RESOLUTION):
    """One config-style border (obstacles.py:344-355)."""
    start = np.asarray(start, dtype=np.float64)
    end = np.asarray(end, dtype=np.float64)
    samples = int(np.linalg.norm(end - start) / resolution)
    line = _f32(np.column_stack((np.linspace(start[0], end[0], samples),
                                 np.linspace(start[1], end[1], samples))))
    center = line[len(line) // 2]
    return line, center, len(line) * resolution


def ellipse_ring(center, yaw, ex, ey, resolution=RESOLUTION, scale=np.sqrt(2.0)):
    """Ring of border points around an oriented box (obstacles.py:269-281), yaw in radians."""
    samples = max(6, int((2.0 * ex + 2.0 * ey) / resolution))
    th = 2.0 * np.pi * np.arange(samples) / samples
    px, py = ex * np.cos(th) * scale, ey * np.sin(th) * scale
    c, s = np.cos(yaw), np.sin(yaw)
    ring = np.column_stack((center[0] + c * px - s * py, center[1] + s * px + c * py))
    return _f32(ring)


def ring_local_offsets(ex, ey, resolution=RESOLUTION, scale=np.sqrt(2.0)):
    """The ellipse of generate_ellipse_border before the vehicle transform (obstacles.py:269-281), fp32."""
    samples = max(6, int((2.0 * ex + 2.0 * ey) / resolution))
    th = 2.0 * np.pi * np.arange(samples) / samples
    return _f32(np.column_stack((ex * np.cos(th) * scale, ey * np.sin(th) * scale)))


def place_ring_f32(center, yaw, local):
    """p = c + R(yaw) u evaluated the way the device does (fp32, fused multiply-adds): the host-side twin of
    sfm_dynamic_boxes_kernel, so CPU checks see bit-identical ring points."""
    c32 = np.asarray(center, dtype=np.float32).astype(np.float64)
    cs, sn = np.float64(np.float32(np.cos(yaw))), np.float64(np.float32(np.sin(yaw)))
    u = np.asarray(local, dtype=np.float32).astype(np.float64)
    fma = lambda a, b, c: np.float32(a * b + c).astype(np.float64)      # product of two fp32 is exact in f64
    px = fma(cs, u[:, 0], fma(-sn, u[:, 1], c32[0]))
    py = fma(sn, u[:, 0], fma(cs, u[:, 1], c32[1]))
    return np.column_stack((px, py))


def advance_center_f32(center, vel, dt):
    c = np.asarray(center, dtype=np.float32).astype(np.float64)
    v = np.asarray(vel, dtype=np.float32).astype(np.float64)
    return np.float32(np.float64(np.float32(dt)) * v + c).astype(np.float64)


def make_crowd(n, seed, density=0.25, jitter=0.8, z_spread=0.0):
    """Jittered-grid crowd.  Returns (loc, vel, waypoint, target_speed, radius,
    world_side)."""
    rng = np.random.default_rng(seed)
    g = int(np.ceil(np.sqrt(n)))
    cell = 1.0 / np.sqrt(density)
    side = g * cell
    k = np.arange(n)
    cx = (k % g + 0.5) * cell
    cy = (k // g + 0.5) * cell
    loc = np.zeros((n, 3))
    loc[:, 0] = cx + rng.uniform(-jitter, jitter, n)
    loc[:, 1] = cy + rng.uniform(-jitter, jitter, n)
    if z_spread:
        loc[:, 2] = rng.uniform(0.0, z_spread, n)
    wp = np.zeros((n, 3))
    wp[:, :2] = rng.uniform(0.0, side, (n, 2))
    to = wp[:, :2] - loc[:, :2]
    ang = np.arctan2(to[:, 1], to[:, 0]) + rng.normal(0.0, 0.3, n)
    speed = rng.uniform(0.6, 1.4, n)
    vel = np.zeros((n, 3))
    vel[:, 0] = speed * np.cos(ang)
    vel[:, 1] = speed * np.sin(ang)
    if z_spread:
        vel[:, 2] = rng.normal(0.0, 0.05, n)
    tspeed = 1.2 + rng.uniform(-0.1, 0.1, n)
    radius = np.full(n, 0.3)
    return _f32(loc), _f32(vel), _f32(wp), _f32(tspeed), _f32(radius), float(np.float32(side))


def make_scenario(n, seed, n_borders=0, n_static=0, n_dynamic=0, z_spread=0.0, density=0.25,
                  border_len=(10.0, 40.0), modes=None):
    """Full synthetic scenario: crowd + K_b borders + M_s static + M_d dynamic obstacles."""
    loc, vel, wp, tspeed, radius, side = make_crowd(n, seed, density, z_spread=z_spread)
    rng = np.random.default_rng(seed + 7919)
    sc = Scenario(loc=loc, vel=vel, waypoint=wp, target_speed=tspeed, radius=radius,
                  mode=np.ones(n, dtype=np.int64) if modes is None else np.asarray(modes, dtype=np.int64),
                  world_side=side, seed=seed)
    cents, lens = [], []
    for _ in range(n_borders):
        o = rng.uniform(0.0, side, 2)
        h = rng.uniform(0.0, 2.0 * np.pi)
        ln = rng.uniform(*border_len)
        line, c, sl = straight_border(o, o + ln * np.array([np.cos(h), np.sin(h)]))
        sc.borders.append(line)
        cents.append(c)
        lens.append(sl)
    sc.border_centers = np.array(cents).reshape(-1, 2)
    sc.border_lengths = np.array(lens, dtype=np.float64)
    for _ in range(n_static):
        c = _f32(rng.uniform(0.0, side, 2))
        ex, ey = rng.uniform(0.2, 1.5, 2)
        sc.static_obstacles.append((c, ellipse_ring(c, rng.uniform(0.0, 2.0 * np.pi), ex, ey)))
    yaws, vels, exts = [], [], []
    for _ in range(n_dynamic):
        c = _f32(rng.uniform(0.0, side, 2))
        yaw = rng.uniform(0.0, 2.0 * np.pi)
        sp = rng.uniform(0.0, 14.0)
        sc.dynamic_obstacles.append((c, place_ring_f32(c, yaw, ring_local_offsets(2.4, 1.0))))
        yaws.append(yaw)
        vels.append(_f32([sp * np.cos(yaw), sp * np.sin(yaw)]))
        exts.append([2.4, 1.0])
    sc.dynamic_vel = np.array(vels, dtype=np.float64).reshape(-1, 2)
    sc.dynamic_yaw = np.array(yaws, dtype=np.float64)
    sc.dynamic_extent = np.array(exts, dtype=np.float64).reshape(-1, 2)
    return sc


def advance_dynamic(sc: Scenario, dt):
    """Move the vehicles one step and rebuild their rings (what the simulator + get_dynamic_obstacles do per
    tick, obstacles.py:297-329), in the device's fp32 arithmetic."""
    new = []
    for k, (c, _) in enumerate(sc.dynamic_obstacles):
        c2 = advance_center_f32(c, sc.dynamic_vel[k], dt)
        new.append((c2, place_ring_f32(c2, sc.dynamic_yaw[k], ring_local_offsets(*sc.dynamic_extent[k]))))
    sc.dynamic_obstacles = new
    return sc


# BASELINE.json configs: name -> (kwargs for make_scenario, enabled forces)
ALL_FORCES = ("acceleration_force", "pedestrian_force", "border_force",
              "static_obstacle_force", "dynamic_obstacle_force")
BASELINE_CONFIGS = {
    "c1": dict(n=64, seed=1001, n_borders=40, n_static=16, n_dynamic=4, forces=ALL_FORCES),
    "c2": dict(n=4096, seed=1002, forces=("acceleration_force", "pedestrian_force")),
    "c3": dict(n=16384, seed=1003, n_borders=2000, n_static=256, forces=ALL_FORCES),
    "c4": dict(n=65536, seed=1004, forces=("pedestrian_force",)),
    "c5": dict(n=262144, seed=1005, n_borders=2000, n_static=256, n_dynamic=512, forces=ALL_FORCES),
}


def baseline_scenario(name):
    kw = dict(BASELINE_CONFIGS[name])
    forces = kw.pop("forces")
    return make_scenario(**kw), forces
