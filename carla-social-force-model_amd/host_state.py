"""Host-side pedestrian bookkeeping: mode state machine + the PedState record array.

API-compatible with the reference's ped_mode_manager.py (PedMode, PedModeManager) and pedestrian_state.py
(PedState): same names, constructor arguments, accessors and record layout, because spawner / main-loop
code outside the hot path builds and mutates these objects (pedestrian_spawner.py:230-243,
run_simulation.py:79-87,118-132).  The implementation is this build's own:

* the FSM is table-driven (transition diversions and entry actions are data);
* PedState keeps the records in a capacity-doubling buffer and exposes ``state`` as a view of the live
  rows, so spawning N pedestrians is O(N) instead of the reference's O(N^2) np.append chain
  (pedestrian_state.py:26-36), and hands the device its numeric columns through ``numeric_columns``.

The record is byte-identical to the reference's (pedestrian_state.py:17-19), 132 bytes packed:
name <U8 @0, id <i4 @32, loc 3f8 @36, vel 3f8 @60, next_waypoint 3f8 @84, mode O @108, radius f8 @116,
target_speed f8 @124.
"""
from enum import IntEnum

import numpy as np


class PedMode(IntEnum):
    IDLE = 0
    WALKING_SIDEWALK = 1
    CROSSING_ROAD = 2
    ROAD_TO_SIDEWALK = 3
    CHECKING_TRAFFIC = 4


#: modes in which BorderForce is switched off for the pedestrian (forces.py:176-177)
BORDER_FREE_MODES = frozenset((PedMode.CROSSING_ROAD, PedMode.ROAD_TO_SIDEWALK))

# (current, requested) -> mode actually entered (ped_mode_manager.py:37-47)
_DIVERSIONS = {
    (PedMode.WALKING_SIDEWALK, PedMode.CROSSING_ROAD): PedMode.CHECKING_TRAFFIC,   # stop and look first
    (PedMode.CROSSING_ROAD, PedMode.WALKING_SIDEWALK): PedMode.ROAD_TO_SIDEWALK,   # still on the road
}
# mode -> attribute holding the target speed on entry; None = stand still; missing = keep (ped_mode_manager.py:49-70)
_ENTRY_SPEED = {
    PedMode.IDLE: None,
    PedMode.CHECKING_TRAFFIC: None,
    PedMode.WALKING_SIDEWALK: "initial_target_speed",
    PedMode.CROSSING_ROAD: "crossing_speed",
}


class PedModeManager:
    """Finite state machine of one pedestrian (ped_mode_manager.py:12-70)."""
    waiting_time = 5    # seconds an IDLE pedestrian waits before walking again

    def __init__(self, ped_name, target_speed, initial_mode, crossing_speed_factor, crossing_safety_margin):
        self.ped_name = ped_name
        self.initial_target_speed = target_speed
        self.crossing_speed = crossing_speed_factor * target_speed
        self.crossing_safety_margin = crossing_safety_margin
        self.sim_time = 0
        self.next_mode_time = -1
        # NB the initial mode's entry action is not run: an initially IDLE pedestrian keeps its speed
        self.current_mode = initial_mode
        self.target_speed = target_speed

    def tick(self, sim_time):
        self.sim_time = sim_time
        if self.current_mode == PedMode.IDLE and self.next_mode_time <= sim_time:
            self._activate_mode(PedMode.WALKING_SIDEWALK)

    def set_mode(self, new_mode):
        self._activate_mode(_DIVERSIONS.get((self.current_mode, new_mode), new_mode))

    def _activate_mode(self, mode):
        if mode not in PedMode._value2member_map_:
            return                                       # the reference's if/elif chain ignores unknown modes
        mode = PedMode(mode)
        if mode in _ENTRY_SPEED:
            attr = _ENTRY_SPEED[mode]
            self.target_speed = 0 if attr is None else getattr(self, attr)
        if mode == PedMode.IDLE:
            self.next_mode_time = self.sim_time + self.waiting_time
        self.current_mode = mode


PED_STATE_DTYPE = [('name', 'U8'), ('id', 'i4'), ('loc', 'f8', (3,)), ('vel', 'f8', (3,)),
                   ('next_waypoint', 'f8', (3,)), ('mode', 'O'), ('radius', 'f8'), ('target_speed', 'f8')]


def _column(field):
    return lambda self: self.state[field]


class PedState:
    """Growable structured array of pedestrians with the reference's accessor surface."""

    def __init__(self, sfm_config):
        # the stock TOML spells the key max_speed_multiplier, so 1.3 always applies (pedestrian_state.py:15)
        self.max_speed_factor = sfm_config.get('max_speed_factor', 1.3)
        self.ped_state_dtype = list(PED_STATE_DTYPE)
        self._buf = None
        self._n = 0
        self.all_states = {}        # sim_time -> snapshot, filled by record_current_state

    # ``state`` is None until the first spawn, then a view of the live rows (callers write through it)
    @property
    def state(self):
        return None if self._buf is None else self._buf[:self._n]

    @state.setter
    def state(self, arr):
        self._buf = None if arr is None else np.array(arr, dtype=self.ped_state_dtype, ndmin=1)
        self._n = 0 if arr is None else len(self._buf)

    def add_pedestrian(self, initial_ped_state):
        """(name, id, loc3, vel3, first_waypoint3, PedModeManager, radius, target_speed)"""
        if self._buf is None:
            self._buf = np.empty(16, dtype=self.ped_state_dtype)
        elif self._n == len(self._buf):
            grown = np.empty(2 * len(self._buf), dtype=self.ped_state_dtype)
            grown[:self._n] = self._buf
            self._buf = grown
        self._buf[self._n] = tuple(initial_ped_state)
        self._n += 1

    def remove_pedestrian(self, ped_name):
        keep = self.state['name'] != ped_name
        k = int(keep.sum())
        self._buf[:k] = self._buf[:self._n][keep]
        self._buf[k:self._n]['mode'] = None            # drop references to the removed mode objects
        self._n = k

    def size(self):
        return self._n

    name = _column('name')
    walker_id = _column('id')
    loc = _column('loc')
    vel = _column('vel')
    next_waypoint = _column('next_waypoint')
    mode = _column('mode')
    radius = _column('radius')
    target_speed = _column('target_speed')

    def max_speed(self):
        return self.max_speed_factor * self.state['target_speed']

    def speeds(self):
        return np.sqrt((self.state['vel'] ** 2).sum(axis=1))

    def update_state(self, walker_id, location, velocity):
        rows = self.state['id'] == walker_id
        self.state['loc'][rows] = location
        self.state['vel'][rows] = velocity

    def update_next_waypoint(self, ped_name, next_waypoint_tuple):
        waypoint, is_crossing = next_waypoint_tuple
        rows = np.flatnonzero(self.state['name'] == ped_name)
        self.state['next_waypoint'][rows] = waypoint
        # IndexError for an unknown name, as in the reference (pedestrian_state.py:92)
        self.state['mode'][rows[0]].set_mode(PedMode.CROSSING_ROAD if is_crossing else PedMode.WALKING_SIDEWALK)

    def apply_current_mode(self):
        self.state['target_speed'] = np.fromiter((m.target_speed for m in self.state['mode']),
                                                 dtype=np.float64, count=self._n)

    def desired_directions(self):
        """Unit xy direction to the waypoint, z = 0 (stateutils.py:7-15)."""
        to_wp = self.state['next_waypoint'][:, :2] - self.state['loc'][:, :2]
        length = np.hypot(to_wp[:, 0], to_wp[:, 1])
        out = np.zeros((self._n, 3))
        out[:, :2] = to_wp / np.where(length == 0.0, 1.0, length)[:, None]
        return out

    def record_current_state(self, sim_time):
        snapshot = self.state.copy()
        snapshot['mode'] = [m.current_mode for m in self.state['mode']]
        self.all_states[sim_time] = snapshot

    def get_all_states(self):
        return self.all_states

    # ---- what the device consumes -------------------------------------------------------------------
    def crossing_mask(self):
        return np.fromiter((m.current_mode in BORDER_FREE_MODES for m in self.state['mode']),
                           dtype=bool, count=self._n)

    def numeric_columns(self):
        """(loc, vel, next_waypoint, target_speed, radius, crossing_mask), contiguous copies gathered
        through field views -- the packed 132-byte record never goes to the device as-is."""
        s = self.state
        return tuple(np.ascontiguousarray(s[f]) for f in ('loc', 'vel', 'next_waypoint', 'target_speed', 'radius')) \
            + (self.crossing_mask(),)
