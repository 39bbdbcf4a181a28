"""Force classes -- host-side mirror of the reference's forces.py, computed on the MI355X.

Same class names, constructor signatures and ``get_force(ped_state, debug=False) -> (N,3) float64``
contract as forces.py:11-291, so an instance can be slotted into an unmodified
``PedestrianSimulation.forces`` dict (pedestrian_simulation.py:81).  Each instance owns one libsfm_hip
handle configured with only its own force switched on; ``_get_force`` uploads the numeric columns of the
PedState, runs one recorded tick and returns that force.  (The fused path that evaluates all forces in one
launch is ``pedestrian_simulation.PedestrianSimulation.tick``; this per-force API exists for drop-in use
and for the per-force parity tests.)  No CPU fallback: without libsfm_hip / a GPU these raise.
"""
import logging
from abc import ABC, abstractmethod

import numpy as np

from .engine import SfmEngine


def _only(sfm_config, name):
    cfg = dict(sfm_config)
    cfg['forces'] = {name: True}
    return cfg


def _columns(peds):
    """Numeric columns of a PedState (ours or the reference's: only .state / .mode() are used)."""
    s = peds.state
    crossing = np.fromiter((int(m.current_mode) in (2, 3) for m in s['mode']), dtype=bool, count=len(s))
    return (np.ascontiguousarray(s['loc']), np.ascontiguousarray(s['vel']), np.ascontiguousarray(s['next_waypoint']),
            np.ascontiguousarray(s['target_speed']), np.ascontiguousarray(s['radius']), crossing)


class Force(ABC):
    """Force base class (forces.py:11-32)."""
    _name = None

    def __init__(self, step_length, sfm_config):
        super().__init__()
        self.step_length = step_length
        self.sfm_config = sfm_config
        self.use_ped_radius = sfm_config.get('use_ped_radius', False)
        self._engine = None

    def _make_engine(self):
        return SfmEngine(_only(self.sfm_config, self._name), self.step_length)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = self._make_engine()
            self._push_geometry()
        return self._engine

    def _push_geometry(self):
        pass

    @abstractmethod
    def _get_force(self, ped_state):
        raise NotImplementedError

    def _evaluate(self, peds):
        loc, vel, wp, ts, rad, crossing = _columns(peds)
        eng = self.engine
        eng.upload_state(loc, vel, wp, ts, rad, crossing)
        eng.tick(record=True)
        return eng.forces(self._name)

    def get_force(self, ped_state, debug=False):
        force = self._get_force(ped_state)
        if debug:
            logging.debug(f"{type(self).__name__}:\n {repr(force)}")
        return force

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None


class AccelerationForce(Force):
    """Helbing-Molnar relaxation towards the desired velocity (forces.py:35-53)."""
    _name = 'acceleration_force'

    def __init__(self, step_length, sfm_config):
        super().__init__(step_length, sfm_config)
        self.tau = self.sfm_config.get('goal_force', {}).get('tau', 0.5)

    def _get_force(self, peds):
        return self._evaluate(peds)


class PedestrianForce(Force):
    """Moussaid et al. 2009 pedestrian interaction, N x N (forces.py:56-117)."""
    _name = 'pedestrian_force'

    def __init__(self, step_length, sfm_config):
        super().__init__(step_length, sfm_config)
        self.ped_force_config = self.sfm_config['pedestrian_force']
        self.lambda_weight = self.ped_force_config.get('lambda', 2.0)
        self.A = self.ped_force_config.get('A', 4.5)
        self.gamma = self.ped_force_config.get('gamma', 0.35)
        self.n = self.ped_force_config.get('n', 2.0)
        self.n_prime = self.ped_force_config.get('n_prime', 3.0)
        self.epsilon = self.ped_force_config.get('epsilon', 0.005)

    def _get_force(self, peds):
        return self._evaluate(peds)


class BorderForce(Force):
    """Exponential repulsion from the nearest sampled point of each nearby border (forces.py:120-179)."""
    _name = 'border_force'

    def __init__(self, step_length, sfm_config, borders, section_info):
        super().__init__(step_length, sfm_config)
        self.borders = borders
        if len(borders):
            self.section_center = np.vstack([np.asarray(row[0], dtype=np.float64) for row in section_info])
            self.section_length = np.array([float(row[1]) for row in section_info])
        else:       # the reference cannot be constructed without borders (np.vstack([]), forces.py:131); we allow it
            self.section_center = np.zeros((0, 2))
            self.section_length = np.zeros(0)
        self.border_force_config = self.sfm_config['border_force']
        self.a = self.border_force_config.get('a', 3.0)
        self.b = self.border_force_config.get('b', 0.1)

    def _push_geometry(self):
        self._engine.set_borders(self.borders, self.section_center, self.section_length)

    def _get_force(self, peds):
        if not len(self.borders):
            return np.zeros((peds.size(), 3))          # forces.py:140-141
        return self._evaluate(peds)


class ObstacleForce(Force):
    """Moussaid interaction with the nearest ring point of each perceived obstacle (forces.py:182-291)."""

    def __init__(self, step_length, sfm_config, dynamic=False):
        super().__init__(step_length, sfm_config)
        self._name = 'dynamic_obstacle_force' if dynamic else 'static_obstacle_force'
        self.dynamic = dynamic
        self.obstacle_locs = None
        self.obstacle_borders = None
        self.obstacle_velocities = None
        self.evasion_force_config = self.sfm_config[self._name]
        c = self.evasion_force_config
        self.lambda_weight = c.get('lambda', 2.0)
        self.A = c.get('A', 4.5)
        self.gamma = c.get('gamma', 0.35)
        self.n = c.get('n', 2.0)
        self.n_prime = c.get('n_prime', 3.0)
        self.epsilon = c.get('epsilon', 0.005)
        self.perception_threshold = c.get('perception_threshold', 20)
        self._dirty = True

    def _push_geometry(self):
        obstacles = [] if self.obstacle_locs is None else list(zip(self.obstacle_locs, self.obstacle_borders))
        # both kinds go through the "dynamic" slot semantics: velocities default to zero (forces.py:212-213)
        if self.dynamic:
            self._engine.set_dynamic_obstacles(obstacles, self.obstacle_velocities)
        else:
            self._engine.set_static_obstacles(obstacles)
        self._dirty = False

    def _get_force(self, peds):
        if self.obstacle_locs is None or self.obstacle_locs.size == 0:
            return np.zeros((peds.size(), 3))          # forces.py:209-210
        eng = self.engine
        if self._dirty:
            self._push_geometry()
        return self._evaluate(peds)

    def update_obstacles(self, obstacles):
        obstacle_locs, obstacle_borders = zip(*obstacles)
        self.obstacle_locs = np.array(obstacle_locs)
        self.obstacle_borders = obstacle_borders
        self._dirty = True

    def update_obstacle_velocities(self, velocities):
        self.obstacle_velocities = np.array(velocities)
        self._dirty = True
