"""PedestrianSimulation on the MI355X -- the drop-in boundary of the hot path.

Call-compatible with the reference class (pedestrian_simulation.py:10-143) as its callers use it:
run_simulation.py:87-128 (update_ped_info, update_dynamic_obstacles, tick, get_new_velocities,
get_arrived_peds, peds.update_next_waypoint, destroy_pedestrian), pedestrian_spawner.py:159
(spawn_pedestrian) and output_generator.py:13-16 (peds.all_states, all_dyn_obs_states, static_obstacles,
borders).

``tick`` keeps the reference's host-side sequence -- refresh target speeds from the modes, advance the
mode FSMs, gap acceptance for pedestrians waiting to cross, optional recording -- and then replaces the
numeric part (pedestrian_simulation.py:81-83: sum of the enabled forces, v + dt*F, speed cap) by ONE fused
HIP launch through libsfm_hip.  The host record array stays the source of truth: its numeric columns are
packed to fp32 SoA and uploaded every tick, and v' is written back into ``state['vel']`` in place, so
``get_new_velocities()`` is the same ``[['id','vel']]`` view the reference returns.
"""
import numpy as np

from ._lib import FORCE_NAMES
from .check_traffic import check_traffic
from .engine import SfmEngine
from .host_state import PedMode, PedState

VEHICLE_RECORD = np.dtype([('id', 'i4'), ('loc', 'f8', (2,)), ('heading', 'f8'), ('vel', 'f8', (2,)),
                           ('extent', 'f8', (2,))])      # layout of all_dyn_obs_states entries (:131-132)


class FusedForce:
    """Value type of ``PedestrianSimulation.forces``: the Force.get_force contract (forces.py:28-32) for one
    force name, served by the simulation's fused engine; obstacle forces also take the two update calls
    (forces.py:285-291), which are deferred until the next upload."""

    def __init__(self, sim, name):
        self._sim, self.name = sim, name

    def get_force(self, ped_state=None, debug=False):
        sim = self._sim
        sim._sync_device(sim.peds if ped_state is None else ped_state)
        sim.engine.tick(record=True)
        return sim.engine.forces(self.name)

    def update_obstacles(self, obstacles):
        self._sim._stage(self.name, obstacles=list(obstacles))

    def update_obstacle_velocities(self, velocities):
        self._sim._stage(self.name, velocities=np.array(velocities))


class PedestrianSimulation:
    """``PedestrianSimulation(borders, border_section_info, obstacles, sfm_config, step_length)``.

    borders: list of (P_k,2) point arrays; border_section_info: K rows of [center(2), section_length];
    obstacles: list of (center(2), ring(P,2)); sfm_config: the parsed sfm_config.toml; step_length: s.
    Keyword-only additions: ``device`` (GPU ordinal), ``record_states`` (the reference records the whole
    state every tick, pedestrian_state.py:100-104; switch off for long runs) and ``honour_file_keys``
    (read ``[acceleration_force] tau`` / ``max_speed_multiplier`` instead of the keys the reference looks
    up) and ``planar_tolerance``.

    ``planar_tolerance`` (metres and m/s; default None = off) is a documented DEVIATION for large crowds on nearly flat
    ground: the reference's pedestrian force and speed cap use 3-component vectors (pedestrian_state.py:17-19,
    forces.py:74-117, stateutils.py:18-23), so by default any spread in z or any v_z sends the crowd through the 3-D
    ordered kernel (exact).  With a tolerance, a crowd whose z stay within it of their median and whose |v_z| stay
    below it is treated as planar -- z differences and v_z are dropped from the norms, v_z' comes back as 0 -- which
    makes the symmetric pair kernel reachable from a CARLA client (walkers never have exactly equal z)."""

    def __init__(self, borders, border_section_info, obstacles, sfm_config, step_length, *, device=0,
                 record_states=True, honour_file_keys=False, planar_tolerance=None):
        self.sfm_config, self.step_length = sfm_config, step_length
        self.borders, self.section_info = borders, border_section_info
        self.static_obstacles = obstacles
        self.record_states = record_states
        self.planar_tolerance = planar_tolerance
        # dynamic obstacles as last reported by the simulator (update_dynamic_obstacles)
        self.dyn_obs_ids, self.dyn_obstacles, self.dyn_obs_heading = [], [], []
        self.dyn_obs_vel, self.dyn_obs_extent = [], []
        self.all_dyn_obs_states = {}
        self.new_velocities = None

        self.peds = PedState(sfm_config)
        if honour_file_keys:
            self.peds.max_speed_factor = sfm_config.get('max_speed_multiplier', self.peds.max_speed_factor)
        self._staged = {}
        self.engine = SfmEngine(sfm_config, step_length, device=device, honour_file_keys=honour_file_keys)
        self.engine.set_timing(False)        # no HIP-event bracket around every tick (~11 us): nobody reads it through this class
        self.forces = self.init_forces()
        # the per-tick exchange with the device: one packed fp32 block up, v' down (sfm_step_packed), buffers kept across ticks
        self._rows = np.empty((16, 9), dtype=np.float32)
        self._zvz = np.empty((16, 2), dtype=np.float32)
        self._vout = np.empty((16, 3), dtype=np.float32)
        # ... or, with a library of ABI >= 5, straight from the records (sfm_step_records); SFM_FACADE_RECORDS=0 keeps the packed block
        import os
        self._use_records = self.engine._lib.sfm_abi_version() >= 5 and os.environ.get("SFM_FACADE_RECORDS", "1") != "0"

    def init_forces(self):
        """dict force name -> FusedForce in the reference's order (pedestrian_simulation.py:37-48); uploads
        the geometry that never changes (borders, static obstacles)."""
        wanted = self.sfm_config['forces']
        table = {name: FusedForce(self, name) for name in FORCE_NAMES if wanted.get(name, False)}
        if 'border_force' in table and len(self.borders):
            rows = list(self.section_info)
            self.engine.set_borders(self.borders,
                                    np.array([np.asarray(r[0], dtype=np.float64)[:2] for r in rows]),
                                    np.array([float(r[1]) for r in rows]))
        if 'static_obstacle_force' in table and self.static_obstacles is not None and len(self.static_obstacles):
            self.engine.set_static_obstacles(self.static_obstacles)
        return table

    # ---- device synchronisation ---------------------------------------------------------------------------
    def _stage(self, force_name, **kw):
        d = self._staged.setdefault(force_name, {})
        d.pop('vehicles', None)              # an explicit update_obstacles / update_obstacle_velocities overrides the tick's vehicle report
        d.update(kw)

    def _sync_device(self, peds):
        for name, what in self._staged.items():
            if name == 'dynamic_obstacle_force' and 'vehicles' in what:
                self.engine.set_dynamic_vehicles(*what['vehicles'])
            elif name == 'dynamic_obstacle_force':
                self.engine.set_dynamic_obstacles(what.get('obstacles', self.dyn_obstacles), what.get('velocities'))
            elif name == 'static_obstacle_force' and 'obstacles' in what:
                self.engine.set_static_obstacles(what['obstacles'])
        self._staged.clear()
        cols = peds.numeric_columns()
        planar = None                                            # None: the engine decides (exactly flat crowds only)
        if self.planar_tolerance is not None and len(cols[0]):
            z, vz = np.asarray(cols[0])[:, 2], np.asarray(cols[1])[:, 2]
            if np.max(np.abs(z - np.median(z))) <= self.planar_tolerance and np.max(np.abs(vz)) <= self.planar_tolerance:
                planar = True
        self.engine.upload_state(*cols, planar=planar)

    def _step_device(self, peds):
        """pedestrian_simulation.py:81-83 on the GPU, host record array in, v' out: staged geometry updates, then ONE library call
        (sfm_step_packed) on one packed fp32 block written straight out of the 132-byte records."""
        for name, what in self._staged.items():
            if name == 'dynamic_obstacle_force' and 'vehicles' in what:
                self.engine.set_dynamic_vehicles(*what['vehicles'])
            elif name == 'dynamic_obstacle_force':
                self.engine.set_dynamic_obstacles(what.get('obstacles', self.dyn_obstacles), what.get('velocities'))
            elif name == 'static_obstacle_force' and 'obstacles' in what:
                self.engine.set_static_obstacles(what['obstacles'])
        self._staged.clear()
        n = peds.size()
        if n > len(self._rows):
            cap = max(n, 2 * len(self._rows))
            self._rows, self._zvz, self._vout = (np.empty((cap, w), dtype=np.float32) for w in (9, 2, 3))
        if self._use_records:
            # ABI 5: the library gathers the five fields out of the records itself (and decides planar / 3-D the same way as below)
            vout = self._vout[:n]
            self.engine.step_records(peds._buf, n, peds.crossing_mask(), vout, self.planar_tolerance)
            return vout
        rows = peds.pack_rows(self._rows)
        s = peds.state
        z, vz = s['loc'][:, 2], s['vel'][:, 2]
        # the 2-D kernels iff all z are equal and no pedestrian has a v_z (then the 3-component formulas of forces.py:74-117 and
        # stateutils.py:18-23 reduce to them exactly), or -- planar_tolerance -- nearly so (the documented deviation above)
        z0 = z[0]
        flat = bool((z == z0).all()) and not bool(vz.any())
        if not flat and self.planar_tolerance is not None:
            flat = bool(np.max(np.abs(z - np.median(z))) <= self.planar_tolerance and np.max(np.abs(vz)) <= self.planar_tolerance)
        zvz = None
        if not flat:
            zvz = self._zvz[:n]
            zvz[:, 0] = z
            zvz[:, 1] = vz
        vout = self._vout[:n]
        self.engine.step_packed(rows, zvz, vout)
        self.engine._z0 = float(z0)
        return vout

    # ---- one simulation step --------------------------------------------------------------------------------
    def tick(self, sim_time):
        peds = self.peds
        if peds.state is None or peds.size() == 0:              # :60-61
            return
        peds.apply_current_mode()                                # :63 (cached arrays, rebuilt when a mode object changed)
        watch = peds.watch
        watch.sim_time = sim_time
        if peds._unwatched:                                      # mode objects of another class: the reference's own loop, :64-67
            waiting = []
            for k, fsm in enumerate(peds.mode()):
                fsm.tick(sim_time)
                if fsm.current_mode == PedMode.CHECKING_TRAFFIC:
                    waiting.append(k)
        else:
            # PedModeManager.tick only ever acts on an IDLE pedestrian (ped_mode_manager.py:30-35): those are the ones visited; the
            # others' clock is the watch's.  Pedestrians waiting at a kerb are known to the watch as well.
            for fsm in tuple(watch.idle):
                fsm.tick(sim_time)
            waiting = [k for k, fsm in enumerate(peds.mode()) if fsm in watch.checking] if watch.checking else ()
        for k in waiting:                                        # :67-73 gap acceptance
            row = peds.state[k]
            if not self.dyn_obstacles or check_traffic(row, self.dyn_obstacles, self.dyn_obs_vel, self.dyn_obs_extent):
                row['mode'].set_mode(PedMode.CROSSING_ROAD)
        if self.record_states:                                   # :76-79
            peds.record_current_state(sim_time)
            if self.dyn_obstacles:
                self.record_dyn_obstacle_states(sim_time)
        self._publish(self._step_device(peds))                   # :81-83, fused on the GPU

    def _publish(self, new_vel):
        view = self.peds.state[['id', 'vel']]                    # a view: the write lands in state['vel'] (:123-124)
        view['vel'] = new_vel
        self.new_velocities = view

    def calculate_new_velocities(self, force):
        """Host-side form of pedestrian_simulation.py:117-124 for a caller-supplied (N,3) force array."""
        v = self.peds.vel() + self.step_length * np.asarray(force)
        speed = np.sqrt((v * v).sum(axis=-1))
        scale = np.minimum(1.0, self.peds.max_speed() / np.where(speed == 0.0, 1.0, speed))
        self._publish(v * scale[:, None])

    def get_new_velocities(self):
        return self.new_velocities

    def close(self):
        self.engine.close()

    # ---- calls made by the CARLA loop -------------------------------------------------------------------------
    def get_arrived_peds(self, distance_threshold):
        if self.peds.state is None:
            return []
        gap = self.peds.next_waypoint()[:, :2] - self.peds.loc()[:, :2]
        return self.peds.name()[np.hypot(gap[:, 0], gap[:, 1]) < distance_threshold]

    def spawn_pedestrian(self, initial_ped_state):
        self.peds.add_pedestrian(initial_ped_state)

    def destroy_pedestrian(self, ped_name):
        self.peds.remove_pedestrian(ped_name)

    def update_ped_info(self, walker_id, location, velocity):
        self.peds.update_state(walker_id, location, velocity)

    def update_dynamic_obstacles(self, dynamic_obstacles):
        """``(ids, positions, headings, velocities, extents, rings)`` as produced by get_dynamic_obstacles
        (obstacles.py:297-329)."""
        ids, positions, headings, velocities, extents, rings = dynamic_obstacles
        self.dyn_obs_ids, self.dyn_obs_heading = ids, headings
        self.dyn_obs_vel, self.dyn_obs_extent = velocities, extents
        self.dyn_obstacles = list(zip(positions, rings))
        moving = self.forces.get('dynamic_obstacle_force')
        if moving is not None and self.dyn_obstacles:
            # (staged like ObstacleForce.update_obstacles / update_obstacle_velocities, forces.py:285-291, but as the arrays they came in)
            self._staged['dynamic_obstacle_force'] = {'vehicles': (positions, rings, velocities)}

    def record_dyn_obstacle_states(self, sim_time):
        rec = np.empty(len(self.dyn_obs_ids), dtype=VEHICLE_RECORD)
        rec['id'] = self.dyn_obs_ids
        rec['loc'] = [c for c, _ in self.dyn_obstacles]
        rec['heading'], rec['vel'], rec['extent'] = self.dyn_obs_heading, self.dyn_obs_vel, self.dyn_obs_extent
        self.all_dyn_obs_states[sim_time] = rec

    def get_states(self):
        return self.peds.get_all_states()
