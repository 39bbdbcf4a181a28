#!/usr/bin/env python3
"""CARLA-free run of the pedestrian simulation, two ways (needs an MI355X):

  host loop     -- the reference's SimulationRunner.tick order (run_simulation.py:47-132) with the simulator
                   replaced by x += v*dt: the drop-in PedestrianSimulation facade is called once per tick;
  device loop   -- the same scenario handed to the GPU once (modes, waypoint queues, vehicles as boxes) and
                   stepped there with sfm_run_recorded; the trajectory comes back as frames.

Both write the reference's four CSV files (output_generator.py) into ./out/.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carla_social_force_model_amd import scenarios                                    # noqa: E402
from carla_social_force_model_amd.config import load_sfm_config                       # noqa: E402
from carla_social_force_model_amd.engine import SfmEngine                             # noqa: E402
from carla_social_force_model_amd.output_generator import OutputGenerator, states_from_frames   # noqa: E402
from carla_social_force_model_amd.ped_mode_manager import PedMode, PedModeManager     # noqa: E402
from carla_social_force_model_amd.pedestrian_simulation import PedestrianSimulation   # noqa: E402


def build(n, seed):
    sc = scenarios.make_scenario(n, seed, n_borders=12, n_static=6, n_dynamic=3, border_len=(5.0, 25.0))
    rng = np.random.default_rng(seed + 1)
    queues = [[(np.array([*rng.uniform(0, sc.world_side, 2), 0.0]), bool(k % 2)) for k in range(3)] for _ in range(n)]
    modes = [PedModeManager(f"ped_{i}", float(sc.target_speed[i]), PedMode.WALKING_SIDEWALK, 1.5, 1.0) for i in range(n)]
    return sc, queues, modes


def host_loop(sc, queues, modes, cfg, ticks, dt):
    info = [[sc.border_centers[k], float(sc.border_lengths[k])] for k in range(len(sc.borders))]
    sim = PedestrianSimulation(sc.borders, info, sc.static_obstacles, cfg, dt)
    for i in range(sc.n):
        sim.spawn_pedestrian((f"ped_{i}", 1000 + i, sc.loc[i], sc.vel[i], sc.waypoint[i], modes[i], float(sc.radius[i]),
                              float(sc.target_speed[i])))
    waypoint_dict = {f"ped_{i}": list(q) for i, q in enumerate(queues)}
    m = len(sc.dynamic_obstacles)
    for k in range(ticks):
        t = k * dt
        sim.update_dynamic_obstacles((list(range(m)), [c for c, _ in sc.dynamic_obstacles], list(np.rad2deg(sc.dynamic_yaw)),
                                      list(sc.dynamic_vel), list(sc.dynamic_extent), [r for _, r in sc.dynamic_obstacles]))
        sim.tick(t)
        for name in sim.get_arrived_peds(2.0):
            if waypoint_dict[name]:
                sim.peds.update_next_waypoint(name, waypoint_dict[name].pop(0))
            else:
                sim.destroy_pedestrian(name)
        if sim.peds.size():
            sim.peds.state['loc'] = sim.peds.state['loc'] + dt * sim.peds.state['vel']        # the simulator's job
        scenarios.advance_dynamic(sc, dt)
    return sim


def device_loop(sc, queues, modes, cfg, ticks, dt, stride):
    eng = SfmEngine(cfg, dt)
    eng.set_borders(sc.borders, sc.border_centers, sc.border_lengths)
    eng.set_static_obstacles(sc.static_obstacles)
    eng.set_dynamic_boxes([c for c, _ in sc.dynamic_obstacles], sc.dynamic_yaw, sc.dynamic_extent, sc.dynamic_vel)
    eng.upload_state(sc.loc, sc.vel, sc.waypoint, sc.target_speed, sc.radius, None)
    eng.set_waypoint_stream(0, 0.0, 2.0)
    eng.set_mode_fsm(modes, queues, despawn_on_arrival=True, sim_time0=0.0, first_vehicle_extent=sc.dynamic_extent[0])
    frames, tick_index = eng.run_recorded(ticks, stride)
    mode, _, _ = eng.modes()
    eng.close()
    return frames, tick_index, mode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--ticks", type=int, default=400)
    ap.add_argument("--out", default="out")
    a = ap.parse_args()
    cfg, dt = load_sfm_config(), 0.05

    sc, queues, modes = build(a.n, 5)
    sim = host_loop(sc, queues, modes, cfg, a.ticks, dt)
    og = OutputGenerator(sim, a.out, "host_loop")
    og.generate_ped_csv(); og.generate_veh_csv(); og.generate_borders_csv(); og.generate_obstacles_csv()
    print(f"host loop: {a.ticks} ticks, {sim.peds.size()} of {a.n} pedestrians left -> {og.output_dir}")
    sim.close()

    sc, queues, modes = build(a.n, 5)
    frames, tick_index, mode = device_loop(sc, queues, modes, cfg, a.ticks, dt, stride=10)

    class Scene:                               # what OutputGenerator reads (output_generator.py:13-16)
        pass
    scene = Scene()
    scene.peds = Scene()
    scene.peds.all_states = states_from_frames(frames, tick_index, dt, [f"ped_{i}" for i in range(a.n)], 1000 + np.arange(a.n),
                                               np.where(mode == 255, 0, mode))
    scene.all_dyn_obs_states, scene.static_obstacles, scene.borders = {}, sc.static_obstacles, sc.borders
    og = OutputGenerator(scene, a.out, "device_loop")
    og.generate_ped_csv(); og.generate_borders_csv(); og.generate_obstacles_csv()
    print(f"device loop: {a.ticks} ticks, {(mode != 255).sum()} of {a.n} pedestrians left, {len(frames)} frames -> {og.output_dir}")


if __name__ == "__main__":
    main()
