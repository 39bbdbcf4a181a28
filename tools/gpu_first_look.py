"""Dev script (GPU box): run every golden case through libsfm_hip and print error statistics vs the
reference outputs, then time the c2 workload for each IPW variant."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _golden_io as gio
from carla_social_force_model_amd.engine import SfmEngine
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from oracle import sfm_oracle as O

def relerr(a, b):
    na = np.linalg.norm(a - b, axis=1); nb = np.linalg.norm(b, axis=1)
    with np.errstate(all="ignore"):
        return np.nanmax(na / np.maximum(nb, 1e-30)), np.nanmax(na) / max(np.nanmax(nb), 1e-30)

for path in gio.list_cases():
    c = gio.Case(path)
    eng = SfmEngine(c.cfg, c.dt)
    if c.borders: eng.set_borders(c.borders, c.border_centers, c.border_lengths)
    eng.set_static_obstacles(c.static_obstacles)
    eng.set_dynamic_obstacles(c.dynamic_obstacles, c.dynamic_vel)
    eng.upload_state(c.loc, c.vel, c.waypoint, c.z["mode_target_speed"], c.radius, c.crossing)
    eng.tick(record=True)
    line = f"{c.name:16s} N={c.n:4d} {eng.kernel_variant():32s}"
    for k in list(O.FORCE_NAMES) + ["total"]:
        if c.has(k):
            f = eng.forces(k); r = c.ref(k)
            nanok = np.array_equal(np.isnan(f).any(1), np.isnan(r).any(1))
            e1, e2 = relerr(f, r)
            line += f" | {k[:5]} {e1:.1e}/{e2:.1e}{'' if nanok else ' NANMISMATCH'}"
    v = eng.velocities(); e1, e2 = relerr(v, c.ref("new_vel"))
    line += f" | v' {e1:.1e}/{e2:.1e}"
    print(line, flush=True)
    eng.close()

print("--- timing c2 (N=4096 ped+acc) ---")
sc, forces = scenarios.baseline_scenario("c2")
cfg = default_sfm_config(forces)
for ipw in (1, 2, 4, 8):
    os.environ["SFM_IPW"] = str(ipw)
    eng = SfmEngine(cfg, 0.05)
    eng.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
    eng.set_waypoint_stream(sc.seed, sc.world_side, 2.0)
    eng.run(50, redraw=True)
    eng.run(500, redraw=True)
    ms, t, l = eng.timing()
    print(f"IPW={ipw} {eng.kernel_variant()} {ms/t*1e3:.2f} us/tick  {t/ms*1e3:.0f} ticks/s  {ms/t*1e6/(4096*4095):.4f} ns/pair", flush=True)
    eng.close()
for n in (16384, 65536):
    sc = scenarios.make_scenario(n, 5)
    for ipw in (4, 8):
        os.environ["SFM_IPW"] = str(ipw)
        eng = SfmEngine(cfg, 0.05)
        eng.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
        eng.set_waypoint_stream(sc.seed, sc.world_side, 2.0)
        eng.run(3, redraw=True); eng.run(10, redraw=True)
        ms, t, l = eng.timing()
        print(f"N={n} IPW={ipw} {ms/t:.3f} ms/tick {ms/t*1e6/(n*(n-1)):.4f} ns/pair", flush=True)
        eng.close()
