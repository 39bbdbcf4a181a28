"""Dev probe (GPU): tile compactness of the row packing after a device re-pack, plain and block-major with unequal bounds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine

sc, forces = scenarios.baseline_scenario("c3")
for label, layout, bounds in (("plain", None, None), ("2x2 equal", (2, 2), [0, 4096, 8192, 12288, 16384]),
                              ("2x2 unequal", (2, 2), [0, 4480, 8192, 12672, 16384]), ("4x1 unequal", (4, 1), [0, 4480, 8192, 12672, 16384])):
    eng = HipShardEngine(default_sfm_config(forces), 0.05)
    if layout:
        eng.set_partition(*layout)
    n, n_pad = eng.load(sc)
    for stage in ("upload", "resort"):
        if stage == "resort":
            if layout:
                eng.set_partition(*layout, bounds)
            eng.engine.resort()
        eng.synchronize()
        pk = eng.packed()[0][0].view(-1, 4)[:n].cpu().numpy()
        t = pk.reshape(-1, 64, 4)
        w = t[:, :, 0].max(1) - t[:, :, 0].min(1); hgt = t[:, :, 1].max(1) - t[:, :, 1].min(1)
        print(f"{label:12s} {stage:7s} tile box width  median {np.median(w):7.2f} max {w.max():7.2f}   height median {np.median(hgt):7.2f} max {hgt.max():7.2f}")
        if layout and stage == "resort":
            for b in range(len(bounds) - 1):
                blk = pk[bounds[b]:bounds[b + 1]]
                print(f"      block {b}: x [{blk[:,0].min():7.1f},{blk[:,0].max():7.1f}]  y [{blk[:,1].min():7.1f},{blk[:,1].max():7.1f}]  rows {len(blk)}")
    eng.close()
