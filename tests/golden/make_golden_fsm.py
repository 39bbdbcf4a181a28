#!/usr/bin/env python3
"""Golden mode-machine traces from the REFERENCE's PedModeManager (ped_mode_manager.py imports nothing but enum).
Build-container only.  A seeded random script of tick / set_mode calls is replayed on reference objects and the
state after every call is stored in tests/golden/fsm_trace.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SFM_REFERENCE", "/root/reference")
if not os.path.isdir(REF):
    print("reference checkout not present -- nothing to do")
    sys.exit(0)
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from ped_mode_manager import PedMode, PedModeManager      # noqa: E402  (reference)

rng = np.random.default_rng(2024)
n_obj, n_calls = 12, 400
init = np.array([[rng.uniform(0.8, 1.6), rng.integers(0, 5), rng.uniform(1.0, 2.0), rng.uniform(-1.0, 2.0)] for _ in range(n_obj)])
objs = [PedModeManager(f"p{k}", float(a), PedMode(int(b)), float(c), float(d)) for k, (a, b, c, d) in enumerate(init)]
script = np.zeros((n_calls, 3))      # object, op (0 = tick, 1 = set_mode), argument
trace = np.zeros((n_calls, 3))       # current_mode, target_speed, next_mode_time after the call
t = 0.0
for c in range(n_calls):
    k = int(rng.integers(0, n_obj))
    if rng.random() < 0.5:
        t += float(rng.uniform(0.0, 1.5))
        objs[k].tick(t)
        script[c] = (k, 0, t)
    else:
        m = int(rng.integers(0, 5))
        objs[k].set_mode(PedMode(m))
        script[c] = (k, 1, m)
    trace[c] = (int(objs[k].current_mode), objs[k].target_speed, objs[k].next_mode_time)
np.savez_compressed(os.path.join(HERE, "fsm_trace.npz"), init=init, script=script, trace=trace)
print("wrote fsm_trace.npz", n_calls, "calls; modes visited:", sorted(set(trace[:, 0].astype(int))))
