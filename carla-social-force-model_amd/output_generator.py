"""CSV output -- mirror of the reference's output_generator.py.

``OutputGenerator(ped_sim, output_path, scenario_name)`` with the four ``generate_*_csv`` methods, the same
file names, headers and column order as output_generator.py:32-110, reading the same attributes of the
simulation (``peds.all_states``, ``all_dyn_obs_states``, ``static_obstacles``, ``borders``;
output_generator.py:13-16) -- so it works on the drop-in ``PedestrianSimulation`` and on the reference's own.
``states_from_frames`` turns a device-recorded trajectory (``SfmEngine.run_recorded``) into the ``all_states``
dict the writer consumes, replacing the per-tick full-state Python copy of the reference.
"""
import csv
import os
import time

import numpy as np

from .host_state import PED_STATE_DTYPE


def states_from_frames(frames, tick_index, step_length, names, ids, modes, t0=0.0):
    """frames[F, N, 4] = {x, y, vx, vy} -> {sim_time: snapshot}, snapshots in the PedState record layout with
    ``mode`` already reduced to its integer value (what record_current_state stores, pedestrian_state.py:100-104)."""
    out = {}
    n = frames.shape[1]
    for f, k in enumerate(tick_index):
        snap = np.zeros(n, dtype=PED_STATE_DTYPE)
        snap['name'], snap['id'] = names, ids
        snap['loc'][:, :2] = frames[f, :, 0:2]
        snap['vel'][:, :2] = frames[f, :, 2:4]
        snap['mode'] = [int(m) for m in modes]
        out[t0 + float(k) * step_length] = snap
    return out


class OutputGenerator:
    """Writes pedestrian.csv, vehicle.csv, borders.csv and obstacles.csv into ``<output_path>/<timestamp>[-name]/``."""

    def __init__(self, ped_sim, output_path, scenario_name):
        self.scene = ped_sim
        self.ped_states = ped_sim.peds.all_states
        self.veh_states = ped_sim.all_dyn_obs_states
        self.static_obstacles = ped_sim.static_obstacles
        self.borders = ped_sim.borders
        self.output_path = output_path
        stamp = time.strftime('%Y%m%d-%H%M%S')
        self.output_dir = os.path.join(output_path, f"{stamp}-{scenario_name}" if scenario_name else stamp)
        os.makedirs(self.output_dir, exist_ok=True)

    def _write(self, file_name, header, rows):
        with open(os.path.join(self.output_dir, file_name), 'w', encoding='UTF8') as f:
            w = csv.writer(f)
            w.writerow(header)
            w.writerows(rows)

    def generate_ped_csv(self):
        def rows():
            for frame, (sim_time, state) in enumerate(self.ped_states.items()):
                for ped in state:
                    yield [int(ped['name'].split('_')[-1]), frame, sim_time, ped['loc'][0], ped['loc'][1],
                           ped['vel'][0], ped['vel'][1], ped['mode']]
        self._write('pedestrian.csv', ['ped_id', 'frame', 'time', 'x', 'y', 'v_x', 'v_y', 'mode'], rows())

    def generate_veh_csv(self):
        def rows():
            for frame, (sim_time, state) in enumerate(self.veh_states.items()):
                for veh in state:
                    yield [veh['id'], frame, sim_time, veh['loc'][0], veh['loc'][1], np.deg2rad(veh['heading']),
                           np.linalg.norm(veh['vel']), veh['extent'][0], veh['extent'][1]]
        self._write('vehicle.csv', ['veh_id', 'frame', 'time', 'x', 'y', 'heading', 'vel', 'ext_x', 'ext_y'], rows())

    def generate_borders_csv(self):
        self._write('borders.csv', ['x', 'y'], ([p[0], p[1]] for border in self.borders for p in border))

    def generate_obstacles_csv(self):
        self._write('obstacles.csv', ['obs_id', 'obs_pos_x', 'obs_pos_y', 'x', 'y'],
                    ([k, pos[0], pos[1], p[0], p[1]] for k, (pos, ring) in enumerate(self.static_obstacles) for p in ring))
