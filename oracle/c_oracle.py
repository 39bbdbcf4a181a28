"""ctypes wrapper of oracle/libsfm_oracle.so (the C/OpenMP restatement).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import sfm_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsfm_oracle.so")


class _Ix(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("lam", "A", "gamma", "n", "n_prime", "epsilon", "thr")]


class _Params(C.Structure):
    _fields_ = [("use_ped_radius", C.c_int), ("max_speed_factor", C.c_double), ("tau", C.c_double),
                ("dt", C.c_double), ("enabled", C.c_int * 5), ("ped", _Ix), ("stat", _Ix), ("dyn", _Ix),
                ("border_a", C.c_double), ("border_b", C.c_double)]


class _Geo(C.Structure):
    _fields_ = [("K", C.c_int), ("off", C.c_void_p), ("pts", C.c_void_p), ("ctr", C.c_void_p), ("extra", C.c_void_p)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oracle_tick.restype = C.c_int
        _lib.oracle_max_threads.restype = C.c_int
    return _lib


def _ix(p: O.Interaction):
    return _Ix(p.lam, p.A, p.gamma, p.n, p.n_prime, p.epsilon, float(p.perception_threshold))


def _geo(polys, centers, extra, keep):
    K = len(polys)
    g = _Geo()
    g.K = K
    if K == 0:
        return g
    off = np.zeros(K + 1, dtype=np.int32)
    for k, p in enumerate(polys):
        off[k + 1] = off[k] + len(p)
    pts = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1, 2) for p in polys]))
    ctr = np.ascontiguousarray(np.asarray(centers, dtype=np.float64).reshape(K, 2))
    keep += [off, pts, ctr]
    g.off, g.pts, g.ctr = off.ctypes.data, pts.ctypes.data, ctr.ctypes.data
    if extra is not None:
        ex = np.ascontiguousarray(np.asarray(extra, dtype=np.float64))
        keep.append(ex)
        g.extra = ex.ctypes.data
    return g


def max_threads():
    return load().oracle_max_threads()


def tick(loc, vel, waypoint, target_speed, radius, crossing, geom: O.Geometry, prm: O.OracleParams, dt,
         rows=None, theta_tol=0.0, nthreads=0, plain=None):
    """Returns (dict name -> (n,3), total (n,3), v_new (n,3), exposure (n,), absum (n,)) for rows [i0,i1).
    ``plain``: optional float64 (n,) array that receives the unweighted sum of term magnitudes (absum without the fp32
    conditioning weights)."""
    lib = load()
    N = loc.shape[0]
    i0, i1 = (0, N) if rows is None else rows
    n = i1 - i0
    keep = []
    P = _Params()
    P.use_ped_radius = int(prm.use_ped_radius)
    P.max_speed_factor, P.tau, P.dt = prm.max_speed_factor, prm.tau, dt
    for k, name in enumerate(O.FORCE_NAMES):
        P.enabled[k] = int(prm.enabled[name])
    P.ped, P.stat, P.dyn = _ix(prm.ped), _ix(prm.static), _ix(prm.dynamic)
    P.border_a, P.border_b = prm.border_a, prm.border_b
    gb = _geo(geom.borders, geom.border_centers, geom.border_lengths, keep)
    gs = _geo([r for _, r in geom.static_obstacles], [c for c, _ in geom.static_obstacles], None, keep)
    gd = _geo([r for _, r in geom.dynamic_obstacles], [c for c, _ in geom.dynamic_obstacles],
              geom.dynamic_vel if len(geom.dynamic_obstacles) else None, keep)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (loc, vel, waypoint, target_speed, radius)]
    cm = np.ascontiguousarray(crossing, dtype=np.uint8)
    forces = np.zeros((6, n, 3))
    v_new = np.zeros((n, 3))
    expo = np.zeros(n)
    absum = np.zeros(n)
    rc = lib.oracle_tick(C.c_int(N), C.c_int(i0), C.c_int(i1), *(C.c_void_p(a.ctypes.data) for a in arrs),
                         C.c_void_p(cm.ctypes.data), C.byref(P), C.byref(gb), C.byref(gs), C.byref(gd),
                         C.c_void_p(forces.ctypes.data), C.c_void_p(v_new.ctypes.data), C.c_void_p(expo.ctypes.data),
                         C.c_void_p(absum.ctypes.data), C.c_double(theta_tol), C.c_int(nthreads),
                         C.c_void_p(plain.ctypes.data if plain is not None else None))
    assert rc == 0
    per = {name: forces[k] for k, name in enumerate(O.FORCE_NAMES) if prm.enabled[name]}
    return per, forces[5], v_new, expo, absum
