"""Dev probe (GPU): wall time of one drop-in PedestrianSimulation.tick -- the loop body of run_simulation.py:87-114: update_dynamic_obstacles,
tick (host record array in, v' written back), get_new_velocities -- at CARLA crowd sizes, with a host-side breakdown of where a tick goes.
    python tools/facade_latency.py > profiles/r04_facade_latency.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from carla_social_force_model_amd import scenarios
from carla_social_force_model_amd.config import default_sfm_config
from carla_social_force_model_amd.ped_mode_manager import PedMode, PedModeManager
from carla_social_force_model_amd.pedestrian_simulation import PedestrianSimulation


def timed(fn, reps=300, spread=None):
    for _ in range(20):
        fn()
    t = np.empty(reps)
    for k in range(reps):
        t0 = time.perf_counter()
        fn()
        t[k] = time.perf_counter() - t0
    if spread is not None:
        spread.extend([np.percentile(t, 90) * 1e6, t.max() * 1e6, float(np.argmax(t))])
    return float(np.median(t)) * 1e6


print("one drop-in tick = sim.update_dynamic_obstacles(...) + sim.tick(t) + sim.get_new_velocities(), all five forces, 4 (2 at N = 20) vehicles whose rings")
print("are re-uploaded every tick, record_states off; microseconds, MEDIAN of 300 ticks after 20 (reference NumPy tick: ~25 ms at N = 64)")
for quiet, (n, nb, ns, nd) in enumerate(((64, 40, 16, 4), (20, 8, 4, 2), (64, 40, 16, 4), (512, 40, 16, 4))):      # (the first pass only warms the process up: its numbers are dropped)
    sc = scenarios.make_scenario(n, 1, nb, ns, nd, z_spread=float(os.environ.get("Z", "0")))     # Z=0.3: walkers on uneven ground (the 3-D bodies)
    info = [[sc.border_centers[k], float(sc.border_lengths[k])] for k in range(nb)]
    sim = PedestrianSimulation(sc.borders, info, sc.static_obstacles, default_sfm_config(), 0.05, record_states=False)
    for i in range(n):
        nm = f"ped_{i}"
        sim.spawn_pedestrian((nm, i, sc.loc[i], sc.vel[i], sc.waypoint[i], PedModeManager(nm, 1.2, PedMode.WALKING_SIDEWALK, 1.5, 1.5), 0.3, 1.2))
    dyn = (list(range(nd)), [c for c, _ in sc.dynamic_obstacles], [0.0] * nd, list(sc.dynamic_vel), [np.array([2.4, 1.0])] * nd, [r for _, r in sc.dynamic_obstacles])
    clock = [0.0]

    def whole():
        sim.update_dynamic_obstacles(dyn); sim.tick(clock[0]); sim.get_new_velocities(); clock[0] += 0.05
    sp = []
    total = timed(whole, spread=sp)
    peds, eng = sim.peds, sim.engine
    parts = {}
    parts["update_dynamic_obstacles (staging only)"] = timed(lambda: sim.update_dynamic_obstacles(dyn))
    sim._staged.clear()
    parts["apply_current_mode + idle FSMs + kerb list"] = timed(lambda: (peds.apply_current_mode(), [f.tick(0.0) for f in tuple(peds.watch.idle)]))
    if not sim._use_records:
        parts["pack_rows (132-B records -> one fp32 block) + flatness test"] = timed(lambda: (peds.pack_rows(sim._rows), bool((peds.state['loc'][:, 2] == 0).all()), bool(peds.state['vel'][:, 2].any())))
    parts["engine.set_dynamic_vehicles (1 concatenate, 2 writes, 1 call; staged for the upload's launch)"] = timed(lambda: eng.set_dynamic_vehicles(dyn[1], dyn[5], dyn[3]))
    rows = peds.pack_rows(sim._rows)
    vout = sim._vout[:n]
    if sim._use_records:
        mask = peds.crossing_mask()
        parts["engine.step_records (fields gathered from the 132-B records, upload + tick + v' download, 1 call)"] = timed(lambda: eng.step_records(peds._buf, n, mask, vout, None))
    else:
        parts["engine.step_packed (upload + tick + v' download, 1 call)"] = timed(lambda: eng.step_packed(rows, None, vout))
    parts["publish v' into state['vel'] (the [['id','vel']] view)"] = timed(lambda: sim._publish(vout))
    if quiet == 0:
        sim.close()
        continue
    print(f"\nN = {n}: {total:6.1f} us per drop-in tick (90th percentile {sp[0]:.1f}, slowest {sp[1]:.0f} at tick {int(sp[2])})   [{eng.kernel_variant()}]")
    for k, v in parts.items():
        print(f"    {v:6.1f}  {k}")
    print(f"    {total - sum(parts.values()):6.1f}  (rest: call overheads, loop)")
    sim.close()
