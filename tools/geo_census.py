"""Dev probe (GPU): phase stamps of the geometry kernel's workgroups (SFM_GEO_STAMPS): start -> find -> scan -> end."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.stepper import HipShardEngine
for name in sys.argv[1:] or ("c3", "c5"):
    out = os.path.join(ROOT, "gpurun_out", f"geo_stamps_{name}.txt")
    os.environ["SFM_GEO_STAMPS"] = out
    sc, forces = scenarios.baseline_scenario(name)
    eng = HipShardEngine(default_sfm_config(tuple(f for f in forces if f != "pedestrian_force")), 0.05)
    eng.load(sc)
    eng.run(5)
    eng.synchronize()
    eng.close()
    d = np.loadtxt(out, dtype=np.uint64)
    d = d[d[:, 3] > 0].astype(np.float64) * 10.0          # 100 MHz -> ns
    t0 = d[:, 0].min()
    find, scan, tail, life = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 3] - d[:, 0]
    pc = lambda v: np.percentile(v, [10, 50, 90, 100]).round(0).tolist()
    print(f"{name}: {len(d)} workgroups, kernel span {(d[:, 3].max() - t0) / 1e3:.1f} us; start spread p50/p100 {np.percentile(d[:, 0] - t0, 50) / 1e3:.1f}/{(d[:, 0].max() - t0) / 1e3:.1f} us")
    print(f"   ns p10/p50/p90/max: find {pc(find)}  scan {pc(scan)}  tail {pc(tail)}  lifetime {pc(life)}")
