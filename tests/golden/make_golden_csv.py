#!/usr/bin/env python3
"""Golden CSV files from the REFERENCE's OutputGenerator (output_generator.py imports only csv/os/time/numpy).
Build-container only.  Writes tests/golden/csv_case.npz: the inputs (a small fake scene) and the four CSV texts."""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SFM_REFERENCE", "/root/reference")
if not os.path.isdir(REF):
    print("reference checkout not present -- nothing to do")
    sys.exit(0)
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from output_generator import OutputGenerator          # noqa: E402  (reference)
from pedestrian_state import PedState                 # noqa: E402  (reference: for the record dtype)

rng = np.random.default_rng(12)
dtype = PedState({}).ped_state_dtype
veh_dtype = [('id', 'i4'), ('loc', 'f8', (2,)), ('heading', 'f8'), ('vel', 'f8', (2,)), ('extent', 'f8', (2,))]
n, frames, m = 5, 3, 2
all_states, veh_states = {}, {}
for f in range(frames):
    s = np.zeros(n, dtype=dtype)
    s['name'] = [f"ped_{i}" for i in range(n)]
    s['id'] = 100 + np.arange(n)
    s['loc'] = rng.normal(size=(n, 3)).astype(np.float32)
    s['vel'] = rng.normal(size=(n, 3)).astype(np.float32)
    s['mode'] = [int(k % 5) for k in range(n)]
    all_states[0.05 * f] = s
    v = np.zeros(m, dtype=veh_dtype)
    v['id'] = [7, 9]
    v['loc'] = rng.normal(size=(m, 2)); v['heading'] = [30.0, -75.5]; v['vel'] = rng.normal(size=(m, 2)); v['extent'] = [[2.4, 1.0]] * m
    veh_states[0.05 * f] = v
borders = [rng.normal(size=(4, 2)), rng.normal(size=(3, 2))]
static = [(rng.normal(size=2), rng.normal(size=(6, 2))), (rng.normal(size=2), rng.normal(size=(7, 2)))]
scene = types.SimpleNamespace(peds=types.SimpleNamespace(all_states=all_states), all_dyn_obs_states=veh_states,
                              static_obstacles=static, borders=borders)
with tempfile.TemporaryDirectory() as tmp:
    og = OutputGenerator(scene, tmp, "golden")
    og.generate_ped_csv(); og.generate_veh_csv(); og.generate_borders_csv(); og.generate_obstacles_csv()
    texts = {k: open(os.path.join(og.output_dir, k + ".csv"), encoding="UTF8", newline="").read()
             for k in ("pedestrian", "vehicle", "borders", "obstacles")}
np.savez_compressed(os.path.join(HERE, "csv_case.npz"),
                    times=np.array(list(all_states)), ped=np.stack([all_states[t][['id', 'loc', 'vel']] for t in all_states]).view(np.uint8),
                    ped_loc=np.stack([all_states[t]['loc'] for t in all_states]), ped_vel=np.stack([all_states[t]['vel'] for t in all_states]),
                    veh_loc=np.stack([veh_states[t]['loc'] for t in veh_states]), veh_vel=np.stack([veh_states[t]['vel'] for t in veh_states]),
                    veh_heading=np.array([30.0, -75.5]), veh_id=np.array([7, 9]),
                    b0=borders[0], b1=borders[1], s0c=static[0][0], s0r=static[0][1], s1c=static[1][0], s1r=static[1][1],
                    **{"csv_" + k: np.array(v) for k, v in texts.items()})
print("wrote csv_case.npz", {k: len(v) for k, v in texts.items()})
