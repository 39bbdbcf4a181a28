"""Flat (npz-friendly) encoding of one golden case: inputs in fp32-representable float64, ragged
geometry as CSR (offsets + concatenated points).  Shared by tests/golden/make_golden.py (writer) and the
tests (reader).  Data only -- no reference code."""
from __future__ import annotations

import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def csr(polys):
    off = np.zeros(len(polys) + 1, dtype=np.int64)
    for k, p in enumerate(polys):
        off[k + 1] = off[k] + len(p)
    pts = np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1, 2) for p in polys], axis=0) \
        if len(polys) else np.zeros((0, 2))
    return off, pts


def uncsr(off, pts):
    return [pts[off[k]:off[k + 1]] for k in range(len(off) - 1)]


def encode_inputs(sc, cfg, dt):
    d = dict(loc=sc.loc, vel=sc.vel, waypoint=sc.waypoint, target_speed=sc.target_speed, radius=sc.radius,
             mode=sc.mode.astype(np.int64), dt=np.float64(dt), cfg=np.array(json.dumps(cfg)),
             world_side=np.float64(sc.world_side), seed=np.int64(sc.seed))
    d["b_off"], d["b_pts"] = csr(sc.borders)
    d["b_center"] = np.asarray(sc.border_centers, dtype=np.float64).reshape(-1, 2)
    d["b_len"] = np.asarray(sc.border_lengths, dtype=np.float64).reshape(-1)
    d["s_off"], d["s_pts"] = csr([r for _, r in sc.static_obstacles])
    d["s_center"] = np.array([c for c, _ in sc.static_obstacles], dtype=np.float64).reshape(-1, 2)
    d["d_off"], d["d_pts"] = csr([r for _, r in sc.dynamic_obstacles])
    d["d_center"] = np.array([c for c, _ in sc.dynamic_obstacles], dtype=np.float64).reshape(-1, 2)
    d["d_vel"] = np.asarray(sc.dynamic_vel if sc.dynamic_vel is not None else np.zeros((0, 2)),
                            dtype=np.float64).reshape(-1, 2)
    return d


class Case:
    """One golden case loaded from an npz (inputs + the reference's outputs)."""

    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.name = os.path.splitext(os.path.basename(path))[0]
        self.z = {k: z[k] for k in z.files}
        self.cfg = json.loads(str(self.z["cfg"]))
        self.dt = float(self.z["dt"])
        for k in ("loc", "vel", "waypoint", "target_speed", "radius", "mode"):
            setattr(self, k, self.z[k])
        self.borders = uncsr(self.z["b_off"], self.z["b_pts"])
        self.border_centers = self.z["b_center"]
        self.border_lengths = self.z["b_len"]
        self.static_obstacles = list(zip(self.z["s_center"], uncsr(self.z["s_off"], self.z["s_pts"])))
        self.dynamic_obstacles = list(zip(self.z["d_center"], uncsr(self.z["d_off"], self.z["d_pts"])))
        self.dynamic_vel = self.z["d_vel"]
        self.world_side = float(self.z["world_side"])
        self.seed = int(self.z["seed"])

    @property
    def n(self):
        return self.loc.shape[0]

    @property
    def crossing(self):
        return (self.mode == 2) | (self.mode == 3)

    def ref(self, key):
        return self.z["ref_" + key]

    def has(self, key):
        return ("ref_" + key) in self.z


def list_cases():
    return sorted(os.path.join(GOLDEN_DIR, f) for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f not in ("csv_case.npz", "fsm_trace.npz"))
