"""Parity metric shared by the GPU tests.

Tolerance (stated here once; BASELINE.json north_star: "<= 1e-5 relative in fp32"):

  * new velocities v' -- the tick's output -- per pedestrian:  |dv'_i| <= RTOL * |v'_i| + RTOL * dt * exposure_i
    (+ a 1e-12 floor for pedestrians whose v' is exactly 0);
  * a force that is a SUM of terms (pedestrian, border, obstacle, total) per pedestrian:
        |dF_i| <= RTOL * max(|F_i|, A_i) + exposure_i,      A_i = sum of the magnitudes of the summed terms,
    i.e. relative to the quantity fp32 rounding of a sum is relative to (a pedestrian whose terms cancel
    to ~0 cannot be matched to 1e-5 of ~0 by ANY fp32 evaluation order);
  * exposure_i is the jump the reference function itself makes at a decision within fp32 noise of its
    threshold (sign(theta) at 0, the +-pi wrap, argmin ties, strict-< culls), computed by the oracle in
    float64 (oracle/sfm_oracle.py).  Pedestrians with non-zero exposure are counted and reported.

NaN rows (coincident pedestrians, forces.py:97,105) must be NaN on both sides.
"""
import numpy as np

RTOL = 1e-5
THETA_TOL = 2e-5      # |theta| below this may flip sign in fp32 (angle error ~1e-7 rad, scaled by B)
TIE_REL = 2e-6        # relative distance gap below which an argmin / cull decision may flip
ATOL = 1e-9           # m/s^2: border terms below 2^-40 * a are not evaluated; a force of
                      # 1e-9 moves v' by 5e-11 m/s per tick, four orders below the fp32 resolution of v'


def check_force(name, got, ref, absum, expo, rtol=RTOL):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    nan_g, nan_r = np.isnan(got).any(axis=1), np.isnan(ref).any(axis=1)
    assert np.array_equal(nan_g, nan_r), f"{name}: NaN rows differ (got {nan_g.sum()}, ref {nan_r.sum()})"
    ok = ~nan_r
    err = np.linalg.norm(got[ok] - ref[ok], axis=1)
    scale = np.maximum(np.linalg.norm(ref[ok], axis=1), np.nan_to_num(absum[ok]))
    allow = rtol * scale + np.nan_to_num(expo[ok]) * 1.001 + ATOL
    bad = err > allow
    worst = float(np.max(err / np.maximum(scale, 1e-300))) if err.size else 0.0
    assert not bad.any(), (f"{name}: {bad.sum()} of {ok.sum()} pedestrians out of tolerance; worst "
                           f"err/scale {np.max((err / np.maximum(scale, 1e-300))[bad]):.3e} (rtol {rtol})")
    return worst, int((np.nan_to_num(expo[ok]) > 0).sum())


def check_velocity(got, ref, expo, dt, rtol=RTOL):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    nan_g, nan_r = np.isnan(got).any(axis=1), np.isnan(ref).any(axis=1)
    assert np.array_equal(nan_g, nan_r), "v': NaN rows differ"
    ok = ~nan_r
    err = np.linalg.norm(got[ok] - ref[ok], axis=1)
    nrm = np.linalg.norm(ref[ok], axis=1)
    allow = rtol * nrm + dt * np.nan_to_num(expo[ok]) * 1.001 + 1e-12
    bad = err > allow
    assert not bad.any(), f"v': {bad.sum()} pedestrians out of tolerance, worst rel {np.max(err[bad] / np.maximum(nrm[bad], 1e-300)):.3e}"
    return float(np.max(err / np.maximum(nrm, 1e-12))) if err.size else 0.0


def check_velocity_conditioned(got, ref, expo, summed, dt, rtol=RTOL):
    """v' for stress scenarios in which a pedestrian's force is a small difference of large terms (hundreds of obstacle
    terms, vehicles at 14 m/s whose exp(-(n B theta)^2) amplifies the fp32 resolution of theta): v' = cap(v + dt F)
    inherits dt * rtol * (sum of the conditioning-weighted term magnitudes, ``summed``), not rtol * |v'|.
    Used only by the tests that say so; the plain ``check_velocity`` is the bound everywhere else."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    nan_g, nan_r = np.isnan(got).any(axis=1), np.isnan(ref).any(axis=1)
    assert np.array_equal(nan_g, nan_r), "v': NaN rows differ"
    ok = ~nan_r
    err = np.linalg.norm(got[ok] - ref[ok], axis=1)
    allow = rtol * (np.linalg.norm(ref[ok], axis=1) + dt * np.nan_to_num(summed[ok])) + dt * np.nan_to_num(expo[ok]) * 1.001 + 1e-12
    bad = err > allow
    assert not bad.any(), f"v': {bad.sum()} pedestrians out of the conditioned tolerance"
    strict = rtol * np.linalg.norm(ref[ok], axis=1) + dt * np.nan_to_num(expo[ok]) * 1.001 + 1e-12
    return int((err > strict).sum())          # how many needed the conditioning term



def diagnostics(got, ref, absum, plain, expo):
    """What a passing check_force leaned on, so that the conditioned tolerance cannot hide a regression:
      plain_rel  = max_i |dF_i| / max_i |F_i|            (no conditioning, no exposure: the headline 'relative error')
      row_rel_p99 = 99th percentile of |dF_i| / |F_i|
      max_amp    = max_i (A_i / sum_j |f_ij| - 1)         (mean conditioning weight of pedestrian i's terms)
      max_expo_rel = max_i exposure_i / max(|F_i|, A_i)   (how large a discontinuity allowance was, relative to the scale)
      n_expo     = pedestrians with a non-zero exposure."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    ok = ~(np.isnan(ref).any(axis=1))
    err = np.linalg.norm(got[ok] - ref[ok], axis=1)
    nrm = np.linalg.norm(ref[ok], axis=1)
    scale = np.maximum(nrm, np.nan_to_num(absum[ok]))
    pl = np.nan_to_num(plain[ok])
    return {"plain_rel": float(err.max() / max(nrm.max(), 1e-300)) if err.size else 0.0,
            "row_rel_p99": float(np.percentile(err / np.maximum(nrm, 1e-300), 99)) if err.size else 0.0,
            "max_amp": float(np.max(np.nan_to_num(absum[ok]) / np.maximum(pl, 1e-300) - 1.0)) if err.size else 0.0,
            "max_expo_rel": float(np.max(np.nan_to_num(expo[ok]) / np.maximum(scale, 1e-300))) if err.size else 0.0,
            "n_expo": int((np.nan_to_num(expo[ok]) > 0).sum()),
            # per pedestrian against the UNWEIGHTED sum of its term magnitudes (what fp32 rounding of the sum is relative to),
            # discontinuity allowance taken off: stays meaningful when |F_i| itself is a small difference of large terms
            "row_over_terms_max": float(np.max(np.maximum(err - np.nan_to_num(expo[ok]) * 1.001 - ATOL, 0.0) / np.maximum(pl, 1e-300))) if err.size else 0.0}


def geometry_tie_exposure(O, loc, vel, waypoint, target_speed, radius, crossing, geom, prm):
    """Exposure of the border / obstacle forces to decisions within fp32 noise of their thresholds -- argmin ties between two
    sampled points, strict-< culls (forces.py:149-155, 222-229) -- from the NumPy oracle, which models them (``tie_rel``); the C
    oracle that the large tests use for speed only models the sign(theta) / wrap exposures.  Cheap: O(N x polylines), the
    pedestrian force is switched off for this call.  Add it to the C oracle's exposure."""
    import copy
    geo_only = copy.copy(prm)
    geo_only.enabled = dict(prm.enabled, pedestrian_force=False, acceleration_force=False)
    if not any(geo_only.enabled.values()):
        return np.zeros(len(loc))
    diag = {}
    with np.errstate(all="ignore"):
        O.tick_forces(loc, vel, waypoint, target_speed, radius, crossing, geom, geo_only, theta_tol=THETA_TOL, tie_rel=TIE_REL, diag=diag)
    return np.nan_to_num(diag["total"][0])
