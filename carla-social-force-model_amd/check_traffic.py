"""Gap-acceptance check without shapely.

Closed-form restatement of the reference's check_traffic (check_traffic.py:7-61), which leans on
shapely's LineString.intersection / distance.  Scalar host logic on the handful of pedestrians in
CHECKING_TRAFFIC mode; it is not on the device path.  Parity status: *unpinned* -- shapely is absent from
this image, so no golden vectors could be produced from the reference for this function; the tests pin it
against hand-derived cases instead.

Kept bug-compatible: the reference offsets every vehicle's front/back by ``vehicle_extents[:][0]``
(check_traffic.py:35-36), i.e. by the *first* vehicle's (x,y) extent, element-wise on the direction.
"""
import numpy as np


def _segment_intersection(p0, p1, q0, q1):
    """Intersection of segments p0-p1 and q0-q1 as shapely's LineString.intersection reports it:
    None (empty), a point (2,), or for collinear overlap a segment (2,2)."""
    r = p1 - p0
    s = q1 - q0
    rxs = r[0] * s[1] - r[1] * s[0]
    qp = q0 - p0
    qpxr = qp[0] * r[1] - qp[1] * r[0]
    if rxs != 0.0:
        t = (qp[0] * s[1] - qp[1] * s[0]) / rxs
        u = qpxr / rxs
        if 0.0 <= t <= 1.0 and 0.0 <= u <= 1.0:
            return p0 + t * r
        return None
    if qpxr != 0.0:
        return None                                    # parallel, not collinear
    rr = float(r @ r)
    if rr == 0.0:                                      # p is a point
        ss = float(s @ s)
        if ss == 0.0:
            return p0 if np.array_equal(p0, q0) else None
        u = float((p0 - q0) @ s) / ss
        return p0 if 0.0 <= u <= 1.0 else None
    t0 = float(qp @ r) / rr
    t1 = t0 + float(s @ r) / rr
    lo, hi = max(0.0, min(t0, t1)), min(1.0, max(t0, t1))
    if lo > hi:
        return None
    if lo == hi:
        return p0 + lo * r
    return np.stack((p0 + lo * r, p0 + hi * r))


def _distance(geom, pt):
    """shapely ``geom.distance(Point(pt))`` for a point or a segment."""
    if geom.ndim == 1:
        return float(np.linalg.norm(geom - pt))
    a, b = geom
    ab = b - a
    t = np.clip(float((pt - a) @ ab) / float(ab @ ab), 0.0, 1.0)
    return float(np.linalg.norm(a + t * ab - pt))


def check_traffic(ped, vehicles, vehicle_velocities, vehicle_extents):
    """True if the pedestrian can cross safely.  Same arguments as the reference: ``ped`` is one PedState
    record, ``vehicles`` the dynamic-obstacle tuples (position, ring), then per-vehicle velocities and
    extents."""
    ped_loc = np.asarray(ped['loc'][:2], dtype=np.float64)
    ped_goal = np.asarray(ped['next_waypoint'][:2], dtype=np.float64)
    ped_speed = ped['mode'].crossing_speed
    margin = ped['mode'].crossing_safety_margin
    if margin < 0:                                     # negative margin: cross without looking (:24)
        return True
    time_ped = np.linalg.norm(ped_goal - ped_loc) / ped_speed
    locs = np.array([np.asarray(v[0], dtype=np.float64)[:2] for v in vehicles]).reshape(-1, 2)
    vels = np.asarray(vehicle_velocities, dtype=np.float64).reshape(-1, 2)
    speed = np.linalg.norm(vels, axis=-1)
    dirs = vels / np.where(speed == 0.0, 1.0, speed)[:, None]
    ext0 = np.asarray(vehicle_extents, dtype=np.float64).reshape(-1, 2)[0]    # (sic) first vehicle's extent
    fronts = locs + dirs * ext0
    backs = locs - dirs * ext0
    for front, back, vel, sp in zip(fronts, backs, vels, speed):
        goal = front + vel * (time_ped + margin)
        hit = _segment_intersection(ped_loc, ped_goal, back, goal)
        if hit is None or sp == 0:
            continue
        tti_ped = _distance(hit, ped_loc) / ped_speed
        tti_front = _distance(hit, front) / sp
        tti_back = _distance(hit, back) / sp
        if tti_front - margin < tti_ped < tti_back + margin:
            return False
    return True
