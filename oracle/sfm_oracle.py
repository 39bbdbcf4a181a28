"""CPU oracle for the Social-Force-Model tick  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker.  The product path (``carla_social_force_model_amd``) never
imports it and fails loudly when the HIP library is missing.

What it is: an independent float64 NumPy restatement of the reference's hot path, written from the
math (streamed over i-chunks, O(chunk*N) memory instead of the reference's dense (N,N-1,3) temporaries).
Every function cites the reference file:line it follows (paths relative to the upstream repository
felixlutz/carla-social-force-model).

Parity pin: the reference ships no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, generated in the build container by
``tests/golden/make_golden.py`` (imports the reference's NumPy modules) and committed as
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` gates this file against those vectors at 1e-12.

Besides the forces the oracle reports, per pedestrian, a *discontinuity exposure*: the size of the jump
the reference function itself makes if a decision that is within fp32 noise of its threshold flips
(sign(theta) at theta=0, the +-pi wrap, argmin ties, strict-< culls).  Parity tests allow
``tol*scale + exposure`` so a legitimate fp32 flip is classified instead of failing blindly
(SURVEY.md section 7.3 item 2).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

TWO_PI = 2.0 * np.pi
FP32_ANGLE_RESOLUTION = 2.0 ** -22   # rad; see moussaid_term

# PedMode values that switch the border force off (ped_mode_manager.py:4-9, forces.py:176-177)
MODE_IDLE, MODE_WALKING_SIDEWALK, MODE_CROSSING_ROAD, MODE_ROAD_TO_SIDEWALK, MODE_CHECKING_TRAFFIC = range(5)

FORCE_NAMES = ("acceleration_force", "pedestrian_force", "border_force",
               "static_obstacle_force", "dynamic_obstacle_force")


@dataclass
class Interaction:
    """Moussaid interaction parameter set (forces.py:66-72 / forces.py:196-206)."""
    lam: float = 2.0
    A: float = 4.5
    gamma: float = 0.35
    n: float = 2.0
    n_prime: float = 3.0
    epsilon: float = 0.005
    perception_threshold: float = 20.0

    @classmethod
    def from_table(cls, tbl):
        # same keys and same code-side defaults as the reference getters
        return cls(lam=tbl.get("lambda", 2.0), A=tbl.get("A", 4.5), gamma=tbl.get("gamma", 0.35),
                   n=tbl.get("n", 2.0), n_prime=tbl.get("n_prime", 3.0), epsilon=tbl.get("epsilon", 0.005),
                   perception_threshold=tbl.get("perception_threshold", 20))


@dataclass
class OracleParams:
    use_ped_radius: bool = False
    max_speed_factor: float = 1.3
    tau: float = 0.5
    enabled: dict = field(default_factory=lambda: {k: True for k in FORCE_NAMES})
    ped: Interaction = field(default_factory=Interaction)
    border_a: float = 3.0
    border_b: float = 0.1
    static: Interaction = field(default_factory=Interaction)
    dynamic: Interaction = field(default_factory=Interaction)

    @classmethod
    def from_config(cls, cfg):
        """Read an sfm_config dict the way the reference does, including its two key mismatches:
        tau comes from ``[goal_force]`` (forces.py:44), the speed factor from ``max_speed_factor``
        (pedestrian_state.py:15); the stock file defines neither, so 0.5 / 1.3 always apply."""
        act = cfg["forces"]                                             # pedestrian_simulation.py:33
        en = {k: bool(act.get(k, False)) for k in FORCE_NAMES}          # :37-48
        p = cls(use_ped_radius=cfg.get("use_ped_radius", False),        # forces.py:18
                max_speed_factor=cfg.get("max_speed_factor", 1.3),
                tau=cfg.get("goal_force", {}).get("tau", 0.5), enabled=en)
        if en["pedestrian_force"]:
            p.ped = Interaction.from_table(cfg["pedestrian_force"])     # forces.py:66 (KeyError if absent)
        if en["border_force"]:
            tbl = cfg["border_force"]                                   # forces.py:134
            p.border_a, p.border_b = tbl.get("a", 3.0), tbl.get("b", 0.1)
        if en["static_obstacle_force"]:
            p.static = Interaction.from_table(cfg["static_obstacle_force"])    # forces.py:199
        if en["dynamic_obstacle_force"]:
            p.dynamic = Interaction.from_table(cfg["dynamic_obstacle_force"])  # forces.py:197
        return p


# --------------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------------
def unit_and_norm(a):
    """stateutils.normalize (stateutils.py:78-92): L2 norm over the last axis; a zero vector is divided
    by 1 (stays zero) and its reported norm stays 0."""
    nrm = np.sqrt(np.sum(a * a, axis=-1))
    safe = np.where(nrm == 0.0, 1.0, nrm)
    return a / safe[..., None], nrm


def wrapped_angle_diff(u, w):
    """stateutils.angle_diff_2d (stateutils.py:95-128): atan2(u_y,u_x) - atan2(w_y,w_x) using components
    0,1 only, wrapped once: > pi -> -2pi, < -pi -> +2pi.  Returns (wrapped, raw)."""
    raw = np.arctan2(u[..., 1], u[..., 0]) - np.arctan2(w[..., 1], w[..., 0])
    out = np.where(raw > np.pi, raw - TWO_PI, raw)
    out = np.where(out < -np.pi, out + TWO_PI, out)
    return out, raw


def moussaid_term(e, dist, dv, p: Interaction, theta_tol=0.0, diagnostics=True):
    """Angular interaction of Moussaid et al. 2009 as coded in forces.py:85-115 (ped-ped) and
    forces.py:241-270 (ped-obstacle).  ``e`` unit direction towards the other body (k components),
    ``dist`` the (possibly radius-reduced) distance, ``dv`` = v_self - v_other.
    Returns (force[..., k], exposure[...], magnitude[...]): exposure is the jump of the lateral term if
    the sign of theta (or the +-pi wrap) flipped and |theta| (or |raw|-pi) is within ``theta_tol``;
    magnitude = |f_v| + |f_theta| is the size of the term, the natural scale for the rounding error of a
    sum of such terms."""
    D = p.lam * dv + e                                       # :85
    t, Dn = unit_and_norm(D)                                 # :86
    ang, raw = wrapped_angle_diff(e, t)                      # :94
    B = p.gamma * Dn                                         # :97
    theta = ang - p.epsilon * B                              # :101  (not re-wrapped)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore", under="ignore"):
        a = -dist / B
        f_v = -p.A * np.exp(a - np.square(p.n_prime * B * theta))                 # :104-105
        f_th = -p.A * np.sign(theta) * np.exp(a - np.square(p.n * B * theta))     # :108-109
    F = f_v[..., None] * t                                   # :112
    F[..., 0] += f_th * (-t[..., 1])                         # :89-91,113  n = (-t_y, t_x, 0)
    F[..., 1] += f_th * t[..., 0]
    if not diagnostics:                                      # plain forces only (the CPU timing leg of bench.py)
        return F, None, None
    if theta_tol > 0.0:
        near = (np.abs(theta) < theta_tol) | (np.abs(np.abs(raw) - np.pi) < theta_tol)
        with np.errstate(invalid="ignore"):
            expo = np.where(near, 2.0 * np.abs(f_th), 0.0)
    else:
        expo = np.zeros_like(dist)
    # Conditioning: d f/d theta = -2 (n B)^2 theta f.  fp32 cannot resolve theta finer than ~2^-22 rad (theta is
    # the angle between two fp32-rounded directions), so ANY fp32 evaluation of this term carries a relative
    # error of 2 (n B)^2 |theta| * 2^-22 on top of ordinary rounding; fast vehicles (B ~ 10) make that ~2e-5.
    # The scale reported for a term is its magnitude times (1 + that amplification / 1e-5), so "1e-5 of the
    # scale" means 1e-5 plus the unavoidable first-order fp32 angle noise (parity section).
    with np.errstate(invalid="ignore", over="ignore"):
        # ... and exp(x) carries |x| times the rounding error of x (three fp32 roundings ~ 2^-22 relative)
        x_v = np.abs(a) + np.square(p.n_prime * B * theta)
        x_t = np.abs(a) + np.square(p.n * B * theta)
        amp_v = (2.0 * np.square(p.n_prime * B) * np.abs(theta) + x_v) * FP32_ANGLE_RESOLUTION / 1e-5
        amp_t = (2.0 * np.square(p.n * B) * np.abs(theta) + x_t) * FP32_ANGLE_RESOLUTION / 1e-5
        mag = np.abs(f_v) * (1.0 + amp_v) + np.abs(f_th) * (1.0 + amp_t)
    return F, expo, mag


# --------------------------------------------------------------------------------------------------
# A6  acceleration force
# --------------------------------------------------------------------------------------------------
def acceleration_force(loc, vel, waypoint, target_speed, tau):
    """AccelerationForce._get_force (forces.py:46-53) with stateutils.desired_directions
    (stateutils.py:7-15): F = (v_target * e_wp - v) / tau; e_wp is the 2-D unit vector to the waypoint
    with z = 0, so the z component is -v_z / tau."""
    to_wp = waypoint[:, :2] - loc[:, :2]
    e2, _ = unit_and_norm(to_wp)
    e3 = np.concatenate([e2, np.zeros((len(e2), 1))], axis=1)
    return (1.0 / tau) * (target_speed[:, None] * e3 - vel)


# --------------------------------------------------------------------------------------------------
# A3  pedestrian force (N x N)
# --------------------------------------------------------------------------------------------------
def pedestrian_force(loc, vel, radius, p: Interaction, use_ped_radius=False, chunk=256,
                     theta_tol=0.0, rows=None, diagnostics=True):
    """PedestrianForce._get_force (forces.py:74-117) streamed over i-chunks.

    Pair (i,j), j != i (stateutils.all_diffs removes the diagonal, stateutils.py:41-49):
      diff = loc_j - loc_i (3 components), e = diff/|diff| (zero-safe), dist = |diff| [- r_i - r_j],
      dv = v_i - v_j (forces.py:77), then ``moussaid_term``; F_i = sum_j (forces.py:117).
    ``rows`` = (i0, i1) restricts the output to a shard of pedestrians (all j are still visited).
    Returns (F[(i1-i0),3], exposure[(i1-i0)], absum[(i1-i0)]) with absum_i = sum_j |f_ij|."""
    N = loc.shape[0]
    i0, i1 = (0, N) if rows is None else rows
    F = np.zeros((i1 - i0, 3))
    expo = np.zeros(i1 - i0)
    absum = np.zeros(i1 - i0)
    idx = np.arange(N)
    for s in range(i0, i1, chunk):
        e_ = min(i1, s + chunk)
        diff = loc[None, :, :] - loc[s:e_, None, :]               # (c, N, 3)
        e, dist = unit_and_norm(diff)
        dv = vel[s:e_, None, :] - vel[None, :, :]
        if use_ped_radius:
            dist = dist - (radius[s:e_, None] + radius[None, :])   # forces.py:80-82
        f, ex, mg = moussaid_term(e, dist, dv, p, theta_tol, diagnostics)
        off_diag = (idx[None, :] != np.arange(s, e_)[:, None])
        f = np.where(off_diag[..., None], f, 0.0)                  # drop j == i (select, so NaN-safe)
        if not diagnostics:
            F[s - i0:e_ - i0] = f.sum(axis=1)
            continue
        ex = np.where(off_diag, ex, 0.0)
        mg = np.where(off_diag, mg, 0.0)
        F[s - i0:e_ - i0] = f.sum(axis=1)
        expo[s - i0:e_ - i0] = np.nansum(ex, axis=1)
        absum[s - i0:e_ - i0] = np.nansum(mg, axis=1)
    return F, expo, absum


# --------------------------------------------------------------------------------------------------
# nearest sampled point of a polyline, first-minimum rule
# --------------------------------------------------------------------------------------------------
def _nearest_points(xy, pts, tie_rel=0.0):
    """argmin_k |xy - pts[k]| with np.argmin's first-minimum rule (forces.py:154,228).
    Returns (nearest[(n,2)], tie[(n,)], alt[(n,2)]): ``tie`` marks rows where a *different* point lies
    within ``tie_rel`` (relative) of the minimum distance, ``alt`` is that runner-up."""
    d = np.sqrt(np.sum((xy[:, None, :] - pts[None, :, :]) ** 2, axis=-1))       # (n, P)
    k = np.argmin(d, axis=1)
    best = pts[k]
    if tie_rel > 0.0 and pts.shape[0] > 1:
        dmin = d[np.arange(len(k)), k]
        d2 = d.copy()
        d2[np.arange(len(k)), k] = np.inf
        k2 = np.argmin(d2, axis=1)
        dsec = d2[np.arange(len(k)), k2]
        tie = (dsec - dmin) <= tie_rel * np.maximum(dmin, 1e-30)
        return best, tie, pts[k2]
    return best, np.zeros(len(k), bool), best


# --------------------------------------------------------------------------------------------------
# A4  border force
# --------------------------------------------------------------------------------------------------
def border_force(loc, radius, crossing, borders, centers, lengths, a, b, use_ped_radius=False,
                 tie_rel=0.0, rows=None):
    """BorderForce._get_force (forces.py:138-179).  Per pedestrian, 2-D:
      keep border k iff |x - center_k| < section_length_k (strict, :149-150);
      p = first nearest sampled point of each kept border (:154-155);
      F += (x-p)/|x-p| * a * exp(-(|x-p| [- r_i]) / b)  (zero-safe normalise, :158-165);
      z = 0 (:171-173); whole force zeroed where ``crossing`` (modes 2,3; :176-177);
      no borders -> zeros (:140-141).
    Looped over borders (vectorised over the pedestrians that keep each border).
    Returns (F[n,3], exposure[n], absum[n])."""
    N = loc.shape[0]
    i0, i1 = (0, N) if rows is None else rows
    n = i1 - i0
    F = np.zeros((n, 3))
    expo = np.zeros(n)
    absum = np.zeros(n)
    if len(borders) == 0:
        return F, expo, absum
    xy = loc[i0:i1, :2]
    rad = radius[i0:i1]
    centers = np.asarray(centers, dtype=np.float64).reshape(-1, 2)
    lengths = np.asarray(lengths, dtype=np.float64)
    for k, pts in enumerate(borders):
        dc = np.sqrt(np.sum((xy - centers[k]) ** 2, axis=-1))
        keep = np.nonzero(dc < lengths[k])[0]
        if tie_rel > 0.0:
            edge = np.nonzero(np.abs(dc - lengths[k]) <= tie_rel * lengths[k])[0]
        else:
            edge = np.zeros(0, int)
        both = np.union1d(keep, edge) if len(edge) else keep
        if len(both) == 0:
            continue
        pts = np.asarray(pts, dtype=np.float64)
        best, tie, alt = _nearest_points(xy[both], pts, tie_rel)

        def f_of(p):
            dirn, dist = unit_and_norm(xy[both] - p)
            if use_ped_radius:
                dist = dist - rad[both]                              # :160-161
            return dirn * (a * np.exp(-dist / b))[:, None]           # :163
        f = f_of(best)
        is_kept = np.isin(both, keep)
        F[both[is_kept], :2] += f[is_kept]
        absum[both[is_kept]] += np.linalg.norm(f[is_kept], axis=1)
        if tie_rel > 0.0:
            on_edge = np.isin(both, edge)
            jump = np.where(on_edge, np.linalg.norm(f, axis=1), 0.0)
            jump = jump + np.where(tie, np.linalg.norm(f_of(alt) - f, axis=1), 0.0)
            expo[both] += jump
    mask = np.asarray(crossing[i0:i1], dtype=bool)
    F[mask] *= 0.0
    expo[mask] = 0.0
    absum[mask] = 0.0
    return F, expo, absum


# --------------------------------------------------------------------------------------------------
# A5  obstacle force (static or dynamic)
# --------------------------------------------------------------------------------------------------
def obstacle_force(loc, vel, radius, obstacles, obstacle_vel, p: Interaction, use_ped_radius=False,
                   tie_rel=0.0, theta_tol=0.0, rows=None):
    """ObstacleForce._get_force (forces.py:208-283).  Per pedestrian, 2-D:
      keep obstacle k iff |x - c_k| < perception_threshold (strict, :222-223; the *centre* is tested);
      p = first nearest ring point of each kept obstacle (:228-229);
      e = (p-x)/|p-x|, dist = |p-x| [- r_i] (:233,237-238), dv = v_i - v_obs,k (:234; static: 0, :212-213);
      then ``moussaid_term`` with this force's own parameter set; sum; z = 0 (:279-281).
      ``obstacles`` None/empty -> zeros (:209-210).  No crossing mask.
    Returns (F[n,3], exposure[n], absum[n])."""
    N = loc.shape[0]
    i0, i1 = (0, N) if rows is None else rows
    n = i1 - i0
    F = np.zeros((n, 3))
    expo = np.zeros(n)
    absum = np.zeros(n)
    if obstacles is None or len(obstacles) == 0:
        return F, expo, absum
    xy = loc[i0:i1, :2]
    v2 = vel[i0:i1, :2]
    rad = radius[i0:i1]
    if obstacle_vel is None:
        obstacle_vel = np.zeros((len(obstacles), 2))
    obstacle_vel = np.asarray(obstacle_vel, dtype=np.float64).reshape(-1, 2)
    thr = float(p.perception_threshold)
    for k, (c, ring) in enumerate(obstacles):
        c = np.asarray(c, dtype=np.float64)[:2]
        dc = np.sqrt(np.sum((xy - c) ** 2, axis=-1))
        keep = np.nonzero(dc < thr)[0]
        edge = np.nonzero(np.abs(dc - thr) <= tie_rel * thr)[0] if tie_rel > 0.0 else np.zeros(0, int)
        both = np.union1d(keep, edge) if len(edge) else keep
        if len(both) == 0:
            continue
        ring = np.asarray(ring, dtype=np.float64)
        best, tie, alt = _nearest_points(xy[both], ring, tie_rel)

        def f_of(pt):
            e, dist = unit_and_norm(pt - xy[both])
            if use_ped_radius:
                dist = dist - rad[both]
            return moussaid_term(e, dist, v2[both] - obstacle_vel[k], p, theta_tol)
        f, ex, mg = f_of(best)
        is_kept = np.isin(both, keep)
        F[both[is_kept], :2] += f[is_kept]
        expo[both[is_kept]] += np.nan_to_num(ex[is_kept])
        absum[both[is_kept]] += np.nan_to_num(mg[is_kept])
        if tie_rel > 0.0:
            on_edge = np.isin(both, edge)
            jump = np.where(on_edge, np.linalg.norm(f, axis=1), 0.0)
            if tie.any():
                jump = jump + np.where(tie, np.linalg.norm(f_of(alt)[0] - f, axis=1), 0.0)
            expo[both] += np.nan_to_num(jump)
    return F, expo, absum


# --------------------------------------------------------------------------------------------------
# A8  velocity update
# --------------------------------------------------------------------------------------------------
def cap_velocity(desired, max_speed):
    """stateutils.cap_velocity (stateutils.py:18-23): 3-component speed, zero speed treated as 1,
    factor = min(1, max_speed / speed)."""
    s = np.sqrt(np.sum(desired * desired, axis=-1))
    s = np.where(s == 0.0, 1.0, s)
    return desired * np.minimum(1.0, max_speed / s)[:, None]


def new_velocities(vel, force, target_speed, dt, max_speed_factor=1.3):
    """PedestrianSimulation.calculate_new_velocities (pedestrian_simulation.py:117-124) with
    PedState.max_speed (pedestrian_state.py:72-73)."""
    return cap_velocity(vel + dt * force, target_speed * max_speed_factor)


def arrived(loc, waypoint, threshold):
    """PedestrianSimulation.get_arrived_peds (pedestrian_simulation.py:88-97): 2-D, strict <."""
    d = waypoint[:, :2] - loc[:, :2]
    return np.sqrt(np.sum(d * d, axis=-1)) < threshold


# --------------------------------------------------------------------------------------------------
# A7  one tick (numeric part)
# --------------------------------------------------------------------------------------------------
@dataclass
class Geometry:
    """Host-side geometry in the reference's own formats."""
    borders: list = field(default_factory=list)            # list of (P_k,2)
    border_centers: np.ndarray = None                      # (K,2)
    border_lengths: np.ndarray = None                      # (K,)
    static_obstacles: list = field(default_factory=list)   # list of (center(2), ring(P,2))
    dynamic_obstacles: list = field(default_factory=list)  # list of (center(2), ring(P,2))
    dynamic_vel: np.ndarray = None                         # (M,2)


def tick_forces(loc, vel, waypoint, target_speed, radius, crossing, geom: Geometry, prm: OracleParams,
                theta_tol=0.0, tie_rel=0.0, rows=None, chunk=256, diag=None):
    """Force part of PedestrianSimulation.tick (pedestrian_simulation.py:81): the enabled forces in the
    dict order acceleration, pedestrian, border, static, dynamic (:37-48).  Returns
    (dict name -> (n,3), total (n,3), exposure (n,)).  ``diag`` (a dict, optional) receives per force the
    exposure and the absolute sum of the terms (the scale of the fp32 rounding error of that force)."""
    N = loc.shape[0]
    i0, i1 = (0, N) if rows is None else rows
    out = {}
    expo = np.zeros(i1 - i0)
    en = prm.enabled
    if diag is None:
        diag = {}
    if en["acceleration_force"]:
        out["acceleration_force"] = acceleration_force(loc, vel, waypoint, target_speed, prm.tau)[i0:i1]
        mag = (np.abs(target_speed[i0:i1]) + np.linalg.norm(vel[i0:i1], axis=1)) / prm.tau
        diag["acceleration_force"] = (np.zeros(i1 - i0), mag)
    if en["pedestrian_force"]:
        f, ex, ab = pedestrian_force(loc, vel, radius, prm.ped, prm.use_ped_radius, chunk, theta_tol, (i0, i1))
        out["pedestrian_force"] = f
        diag["pedestrian_force"] = (ex, ab)
    if en["border_force"]:
        f, ex, ab = border_force(loc, radius, crossing, geom.borders, geom.border_centers, geom.border_lengths,
                                 prm.border_a, prm.border_b, prm.use_ped_radius, tie_rel, (i0, i1))
        out["border_force"] = f
        diag["border_force"] = (ex, ab)
    if en["static_obstacle_force"]:
        f, ex, ab = obstacle_force(loc, vel, radius, geom.static_obstacles, None, prm.static,
                                   prm.use_ped_radius, tie_rel, theta_tol, (i0, i1))
        out["static_obstacle_force"] = f
        diag["static_obstacle_force"] = (ex, ab)
    if en["dynamic_obstacle_force"]:
        f, ex, ab = obstacle_force(loc, vel, radius, geom.dynamic_obstacles, geom.dynamic_vel, prm.dynamic,
                                   prm.use_ped_radius, tie_rel, theta_tol, (i0, i1))
        out["dynamic_obstacle_force"] = f
        diag["dynamic_obstacle_force"] = (ex, ab)
    for ex, _ in diag.values():
        expo += ex
    diag["total"] = (expo.copy(), sum(ab for _, ab in diag.values()))
    total = np.zeros((i1 - i0, 3))
    for name in FORCE_NAMES:           # same accumulation order as sum(map(...)) over the dict
        if name in out:
            total = total + out[name]
    return out, total, expo


# --------------------------------------------------------------------------------------------------
# CARLA-free stepping (the build's stand-in for the simulator)
# --------------------------------------------------------------------------------------------------
def _mix32(a):
    """lowbias32 integer hash on uint32 arrays (wrap-around arithmetic)."""
    a = np.asarray(a, dtype=np.uint64) & 0xFFFFFFFF
    a ^= a >> 16
    a = (a * 0x7FEB352D) & 0xFFFFFFFF
    a ^= a >> 15
    a = (a * 0x846CA68B) & 0xFFFFFFFF
    a ^= a >> 16
    return a


def redraw_waypoint(ped_index, draw_count, seed, world_side):
    """Counter-based waypoint stream of the synthetic scenarios (not in the reference; SURVEY.md
    section 8d): coordinate c in {0,1} of draw number ``draw_count`` of pedestrian ``ped_index`` is
    world_side * (mix32(seed ^ mix32(2*ped+c + 0x9E3779B9*draw)) >> 8) * 2^-24, evaluated in fp32."""
    ped_index = np.asarray(ped_index, dtype=np.uint64)
    draw_count = np.asarray(draw_count, dtype=np.uint64)
    out = np.empty((len(ped_index), 2), dtype=np.float32)
    for c in (0, 1):
        key = (2 * ped_index + c + 0x9E3779B9 * draw_count) & 0xFFFFFFFF
        h = _mix32(np.uint64(seed & 0xFFFFFFFF) ^ _mix32(key))
        u = (h >> 8).astype(np.float32) * np.float32(2.0 ** -24)
        out[:, c] = u * np.float32(world_side)
    return out


def free_step(loc, vel, waypoint, target_speed, radius, crossing, draws, geom, prm, dt,
              arrive_threshold=2.0, seed=0, world_side=0.0, redraw=True, round_f32=False,
              waypoint_source=None):
    """One tick of the CARLA-free loop, in the order of run_simulation.py:77-132 with x <- x + dt*v'
    standing in for the simulator: forces on (x, v) -> v' (tick) -> arrival test on the *pre-move* x
    (:118) -> waypoint swap -> x <- x + dt*v' (what CARLA does before the next tick, :77-87).
    ``round_f32`` rounds the new state to fp32 after the step (the device keeps fp32 state), which makes
    this the re-sync reference for multi-tick device runs.  ``waypoint_source(ids, draws) -> (k,2)``
    overrides the hash stream (the golden trajectories use a pre-drawn queue, like waypoint_dict in
    run_simulation.py:120-126).  Returns (loc', vel', waypoint', draws')."""
    _, F, _ = tick_forces(loc, vel, waypoint, target_speed, radius, crossing, geom, prm)
    v_new = new_velocities(vel, F, target_speed, dt, prm.max_speed_factor)
    wp = waypoint.copy()
    draws = draws.copy()
    if redraw:
        hit = arrived(loc, waypoint, arrive_threshold)
        if hit.any():
            ids = np.nonzero(hit)[0]
            draws[ids] += 1
            if waypoint_source is not None:
                wp[ids, :2] = waypoint_source(ids, draws[ids])
            else:
                wp[ids, :2] = redraw_waypoint(ids, draws[ids], seed, world_side)
    x_new = loc + dt * v_new
    if round_f32:
        x_new = x_new.astype(np.float32).astype(np.float64)
        v_new = v_new.astype(np.float32).astype(np.float64)
    return x_new, v_new, wp, draws


# --------------------------------------------------------------------------------------------------
# independent float64 restatements used as checkers of the product's host twins and
# device code.  PARITY UNPINNED: obstacles.py needs carla and check_traffic.py needs shapely, neither is installable here, and
# the reference holds no fixtures for them -- these follow the source text and the documented semantics of the two libraries.
# --------------------------------------------------------------------------------------------------
def ellipse_ring(center, yaw, extent_x, extent_y, resolution=0.1, size_factor=np.sqrt(2.0)):
    """generate_ellipse_border (obstacles.py:269-281): ``samples = max(6, int((2 ex + 2 ey) / resolution))`` points
    (ex cos t, ey sin t) * size_factor for t = 2 pi i / samples, each moved by the vehicle transform -- with pitch = roll = 0,
    carla.Transform.transform rotates by yaw about z and translates by the location.  yaw in radians.  Returns (samples, 2)."""
    samples = max(6, int((2.0 * extent_x + 2.0 * extent_y) / resolution))
    out = np.empty((samples, 2))
    c, s = np.cos(yaw), np.sin(yaw)
    for i in range(samples):
        theta = np.pi * 2 * i / samples
        px, py = extent_x * np.cos(theta) * size_factor, extent_y * np.sin(theta) * size_factor
        out[i] = (center[0] + c * px - s * py, center[1] + s * px + c * py)
    return out


def _cross2(a, b):
    return a[0] * b[1] - a[1] * b[0]


def _segments_meet(p0, p1, q0, q1):
    """What shapely's LineString([p0,p1]).intersection(LineString([q0,q1])) is, as a list of 0, 1 (point) or 2 (overlap
    segment end points) points.  Written from the definitions: a point X is on both segments."""
    d1, d2 = p1 - p0, q1 - q0
    den = _cross2(d1, d2)
    if den != 0.0:                                           # lines cross in exactly one point
        t = _cross2(q0 - p0, d2) / den
        u = _cross2(q0 - p0, d1) / den
        return [p0 + t * d1] if (0.0 <= t <= 1.0 and 0.0 <= u <= 1.0) else []
    if _cross2(q0 - p0, d1) != 0.0:                          # parallel, distinct lines
        return []
    L = float(d1 @ d1)
    if L == 0.0:                                             # the pedestrian does not move: a point against a segment
        M = float(d2 @ d2)
        if M == 0.0:
            return [p0] if (p0 == q0).all() else []
        u = float((p0 - q0) @ d2) / M
        return [p0] if 0.0 <= u <= 1.0 else []
    a, b = sorted((float((q0 - p0) @ d1) / L, float((q1 - p0) @ d1) / L))   # q's extent along p's parameter
    lo, hi = max(a, 0.0), min(b, 1.0)
    if lo > hi:
        return []
    return [p0 + lo * d1] if lo == hi else [p0 + lo * d1, p0 + hi * d1]


def _dist_to(points, x):
    """shapely geometry.distance(Point(x)) for a point (1 entry) or a segment (2 entries)."""
    if len(points) == 1:
        return float(np.hypot(*(points[0] - x)))
    a, b = points
    ab = b - a
    t = min(1.0, max(0.0, float((x - a) @ ab) / float(ab @ ab)))
    return float(np.hypot(*(a + t * ab - x)))


def gap_accepted(ped_loc, ped_goal, ped_speed, safety_margin, vehicle_locs, vehicle_velocities, vehicle_extents):
    """check_traffic (check_traffic.py:7-61): may the pedestrian at ``ped_loc`` walk to ``ped_goal`` at ``ped_speed``?
    Every vehicle's front / back is its position +- direction * vehicle_extents[:][0] -- the FIRST vehicle's (x, y) extent,
    element-wise (:35-36, kept as written); its path runs from the back to front + v (time_ped + margin) (:42-43); if the paths
    meet and the vehicle moves, the pedestrian waits when tti_front - margin < tti_ped < tti_back + margin (:52-58)."""
    ped_loc, ped_goal = np.asarray(ped_loc, dtype=np.float64)[:2], np.asarray(ped_goal, dtype=np.float64)[:2]
    if not safety_margin >= 0:                                                   # :24
        return True
    time_ped = np.linalg.norm(ped_goal - ped_loc) / ped_speed                    # :27-28
    locs = np.asarray(vehicle_locs, dtype=np.float64).reshape(-1, 2)
    vels = np.asarray(vehicle_velocities, dtype=np.float64).reshape(-1, 2)
    dirs, _ = unit_and_norm(vels)                                                # :34
    first_extent = np.asarray(vehicle_extents, dtype=np.float64).reshape(-1, 2)[0]
    for loc, vel, d in zip(locs, vels, dirs):
        front, back = loc + d * first_extent, loc - d * first_extent             # :35-36
        meet = _segments_meet(ped_loc, ped_goal, back, front + vel * (time_ped + safety_margin))   # :42-46
        if not meet:
            continue
        speed = np.linalg.norm(vel)
        if speed != 0:                                                           # :49-50
            tti_ped = _dist_to(meet, ped_loc) / ped_speed
            if _dist_to(meet, front) / speed - safety_margin < tti_ped < _dist_to(meet, back) / speed + safety_margin:
                return False
    return True
