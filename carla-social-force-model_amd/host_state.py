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


class ModeWatch:
    """What the PedModeManagers of ONE PedState share (round 4): a version counter that every change of a mode or a mode's target
    speed bumps, the simulation clock, and the managers currently in the two modes the per-tick host loop cares about.  With it
    ``PedState.apply_current_mode`` / ``crossing_mask`` are cached arrays instead of a Python loop over N objects every tick, and
    ``PedestrianSimulation.tick`` visits only the IDLE pedestrians (the only ones whose ``tick`` can do anything,
    ped_mode_manager.py:30-35) and the ones waiting at a kerb."""
    __slots__ = ("version", "sim_time", "idle", "checking")

    def __init__(self):
        self.version, self.sim_time = 0, 0
        self.idle, self.checking = set(), set()


class PedModeManager:
    """Finite state machine of one pedestrian (ped_mode_manager.py:12-70).  ``current_mode`` and ``target_speed`` are plain
    attributes to every caller; behind them a setter tells the owning PedState's ModeWatch (if any) that something changed."""
    waiting_time = 5    # seconds an IDLE pedestrian waits before walking again
    _watch = None       # set by PedState.add_pedestrian

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

    @property
    def current_mode(self):
        return self._mode

    @current_mode.setter
    def current_mode(self, mode):
        self._mode = mode
        w = self._watch
        if w is not None:
            w.version += 1
            w.idle.discard(self); w.checking.discard(self)
            if mode == PedMode.IDLE:
                w.idle.add(self)
            elif mode == PedMode.CHECKING_TRAFFIC:
                w.checking.add(self)

    @property
    def target_speed(self):
        return self._target_speed

    @target_speed.setter
    def target_speed(self, v):
        self._target_speed = v
        if self._watch is not None:
            self._watch.version += 1

    def _attach(self, watch):
        """Join (or, watch None, leave) a PedState's ModeWatch."""
        old = self._watch
        if old is not None:
            old.idle.discard(self); old.checking.discard(self); old.version += 1
        self._watch = watch
        if watch is not None:
            self.current_mode = self._mode           # registers the mode with the new watch

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
            # the time of the last tick (ped_mode_manager.py:52: self.sim_time) -- a watched manager that is not IDLE is not
            # ticked one by one any more, the watch's clock is what its own sim_time would have been
            now = self.sim_time if self._watch is None else max(self.sim_time, self._watch.sim_time)
            self.next_mode_time = now + self.waiting_time
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
        self.watch = ModeWatch()    # shared by the mode managers of this crowd
        self._unwatched = 0         # mode objects of another class (e.g. the reference's own): no caching while there are any
        self._cache_version = -1    # watch.version the cached arrays below belong to
        self._cache_n = -1
        self._ts_cache = self._cross_cache = self._mode_cache = None

    # ``state`` is None until the first spawn, then a view of the live rows (callers write through it)
    @property
    def state(self):
        return None if self._buf is None else self._buf[:self._n]

    @state.setter
    def state(self, arr):
        self._watch_rows(False)
        self._buf = None if arr is None else np.array(arr, dtype=self.ped_state_dtype, ndmin=1)
        self._n = 0 if arr is None else len(self._buf)
        self._watch_rows(True)

    def _watch_rows(self, attach):
        """(Un)register the mode objects of the live rows with this crowd's ModeWatch."""
        self._unwatched = 0
        for m in ([] if self._buf is None else self._buf[:self._n]['mode']):
            if isinstance(m, PedModeManager):
                m._attach(self.watch if attach else None)
            elif attach:
                self._unwatched += 1
        self.invalidate_modes()

    def invalidate_modes(self):
        """Forget the cached target speeds / crossing mask.  Spawning, removing, ``set_mode`` and assignments to a manager's
        ``current_mode`` / ``target_speed`` do it by themselves; call it after putting ANOTHER mode object into ``state['mode']``
        by hand (the reference's callers never do: pedestrian_spawner.py:230-243 builds the object once per pedestrian)."""
        self._cache_version = -1

    def add_pedestrian(self, initial_ped_state):
        """(name, id, loc3, vel3, first_waypoint3, PedModeManager, radius, target_speed)"""
        if self._buf is None:
            self._buf = np.empty(16, dtype=self.ped_state_dtype)
        elif self._n == len(self._buf):
            grown = np.empty(2 * len(self._buf), dtype=self.ped_state_dtype)
            grown[:self._n] = self._buf
            self._buf = grown
        self._buf[self._n] = tuple(initial_ped_state)
        m = self._buf[self._n]['mode']
        if isinstance(m, PedModeManager):
            m._attach(self.watch)
        else:
            self._unwatched += 1
        self._n += 1
        self.invalidate_modes()

    def remove_pedestrian(self, ped_name):
        keep = self.state['name'] != ped_name
        for m in self.state['mode'][~keep]:
            if isinstance(m, PedModeManager):
                m._attach(None)
            else:
                self._unwatched -= 1
        k = int(keep.sum())
        self._buf[:k] = self._buf[:self._n][keep]
        self._buf[k:self._n]['mode'] = None            # drop references to the removed mode objects
        self._n = k
        self.invalidate_modes()

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

    def _refresh_mode_caches(self):
        """Target speeds, border-force mask and modes of the live rows as arrays -- rebuilt only when a mode object said that
        something changed (ModeWatch.version), a row came or went, or some mode object is not one of ours."""
        w = self.watch
        if self._unwatched == 0 and self._cache_version == w.version and self._cache_n == self._n:
            return
        version = w.version
        modes = self.state['mode']
        self._ts_cache = np.fromiter((m.target_speed for m in modes), dtype=np.float64, count=self._n)
        self._mode_cache = [m.current_mode for m in modes]
        self._cross_cache = np.fromiter((c in BORDER_FREE_MODES for c in self._mode_cache), dtype=bool, count=self._n)
        self._cache_version, self._cache_n = version, self._n

    def apply_current_mode(self):
        """state['target_speed'] <- the mode objects' target speeds (pedestrian_state.py:94-95), every tick: whatever a caller
        wrote into the column in between is overwritten, as in the reference."""
        self._refresh_mode_caches()
        self.state['target_speed'] = self._ts_cache

    def desired_directions(self):
        """Unit xy direction to the waypoint, z = 0 (stateutils.py:7-15)."""
        to_wp = self.state['next_waypoint'][:, :2] - self.state['loc'][:, :2]
        length = np.hypot(to_wp[:, 0], to_wp[:, 1])
        out = np.zeros((self._n, 3))
        out[:, :2] = to_wp / np.where(length == 0.0, 1.0, length)[:, None]
        return out

    def record_current_state(self, sim_time):
        snapshot = self.state.copy()
        self._refresh_mode_caches()
        snapshot['mode'] = self._mode_cache
        self.all_states[sim_time] = snapshot

    def get_all_states(self):
        return self.all_states

    # ---- what the device consumes -------------------------------------------------------------------
    def crossing_mask(self):
        self._refresh_mode_caches()
        return self._cross_cache

    def pack_rows(self, out):
        """The numeric columns the device consumes as ONE fp32 block, written into ``out[:n]`` (float32, shape (cap, 9)):
        {x, y, vx, vy, waypoint x, waypoint y, target_speed, radius, border-force-off flag}.  Six strided assignments straight out
        of the 132-byte records (no per-column copies, no float64 intermediates)."""
        s, n = self.state, self._n
        o = out[:n]
        o[:, 0:2] = s['loc'][:, :2]
        o[:, 2:4] = s['vel'][:, :2]
        o[:, 4:6] = s['next_waypoint'][:, :2]
        o[:, 6] = s['target_speed']
        o[:, 7] = s['radius']
        o[:, 8] = self.crossing_mask()
        return o

    def numeric_columns(self):
        """(loc, vel, next_waypoint, target_speed, radius, crossing_mask), contiguous copies gathered
        through field views -- the packed 132-byte record never goes to the device as-is."""
        s = self.state
        return tuple(np.ascontiguousarray(s[f]) for f in ('loc', 'vel', 'next_waypoint', 'target_speed', 'radius')) \
            + (self.crossing_mask(),)
