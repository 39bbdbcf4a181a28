/*
 * sfm_hip.h -- C ABI of libsfm_hip.so: the MI355X (gfx950) Social-Force-Model stepper.
 *
 * Drop-in boundary for ONE hot path of felixlutz/carla-social-force-model: forces.py + stateutils.py +
 * the numeric half of pedestrian_simulation.py.  The reference is pure Python and has no FFI layer of
 * its own, so each entry point below names the reference interface it replaces (file:line, relative to
 * the upstream repository); INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * SfmStatus; sfm_last_error() gives the message of the last failure on a handle (or of the last failed
 * sfm_create when called with NULL).  The caller owns every host buffer; the library owns all device
 * memory behind the opaque handle.  A handle drives ONE GPU and is not thread-safe; distinct handles are
 * independent.  Arithmetic is fp32 on the device; host arrays are fp32 SoA (one array per component).
 * All work is issued on the handle's stream (sfm_set_stream; default: the null stream); the download /
 * query calls synchronise that stream.
 */
#ifndef SFM_HIP_H
#define SFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_ABI_VERSION 5   /* 2: + sfm_tick_begin / sfm_tick_end, sfm_set_partition, sfm_get_pair_work; 3: + sfm_set_timing; 4: + sfm_step_packed, sfm_set_dynamic_obstacles_packed; 5: + sfm_step_records (additions only) */

typedef struct SfmHandle SfmHandle;

typedef enum SfmStatus {
    SFM_OK = 0,
    SFM_ERR_INVALID = -1,   /* bad argument (NULL pointer, negative size, inconsistent CSR offsets ...) */
    SFM_ERR_HIP = -2,       /* a HIP runtime call failed; message in sfm_last_error */
    SFM_ERR_STATE = -3,     /* call order: e.g. sfm_tick before sfm_upload_state */
    SFM_ERR_NO_DEVICE = -4  /* no usable gfx950 device */
} SfmStatus;

/* Index of each force: the dict order of PedestrianSimulation.init_forces (pedestrian_simulation.py:37-48),
 * which is also the summation order of tick() (:81). */
enum { SFM_FORCE_ACCELERATION = 0, SFM_FORCE_PEDESTRIAN = 1, SFM_FORCE_BORDER = 2,
       SFM_FORCE_STATIC_OBSTACLE = 3, SFM_FORCE_DYNAMIC_OBSTACLE = 4, SFM_NUM_FORCES = 5,
       SFM_FORCE_TOTAL = 5 };

/* One Moussaid parameter table: [pedestrian_force] (forces.py:66-72), [static_obstacle_force] /
 * [dynamic_obstacle_force] (forces.py:196-206). perception_threshold is unused for pedestrians. */
typedef struct SfmInteraction {
    float lambda, A, gamma, n, n_prime, epsilon, perception_threshold;
} SfmInteraction;

/* Mirrors config/sfm_config.toml as the reference *reads* it. */
typedef struct SfmParams {
    int32_t use_ped_radius;            /* forces.py:18 */
    float   max_speed_factor;          /* pedestrian_state.py:15 (1.3) */
    float   tau;                       /* forces.py:44 (0.5) */
    float   step_length;               /* run_simulation.py:168 (0.05 s) */
    int32_t enabled[SFM_NUM_FORCES];   /* [forces] switches, pedestrian_simulation.py:33-48 */
    SfmInteraction pedestrian;
    float   border_a, border_b;        /* forces.py:134-136 */
    SfmInteraction static_obstacle;
    SfmInteraction dynamic_obstacle;
} SfmParams;

/* sfm_tick / sfm_run flags */
enum {
    SFM_TICK_INTEGRATE = 1u,        /* also x <- x + dt*v' : the CARLA-free stand-in for the simulator that moves
                                       the walkers between ticks (run_simulation.py:77-87, carla_simulation.py:126-129) */
    SFM_TICK_REDRAW_WAYPOINTS = 2u, /* on arrival (pedestrian_simulation.py:92-95) draw the next waypoint from the
                                       counter-based stream set by sfm_set_waypoint_stream (run_simulation.py:118-126) */
    SFM_TICK_RECORD_FORCES = 4u     /* keep the per-force arrays for sfm_download_forces (Force.get_force, forces.py:28-32) */
};

/* ---- lifetime ------------------------------------------------------------------------------------ */

/* PedestrianSimulation.__init__ + init_forces (pedestrian_simulation.py:11-55): capture the parameters,
 * create the device context on `device_id`. */
int sfm_create(const SfmParams* params, int device_id, SfmHandle** out);
int sfm_destroy(SfmHandle* h);                                   /* PedestrianSimulation.close (:85) */
int sfm_set_params(SfmHandle* h, const SfmParams* params);       /* Force.__init__ captures (forces.py:14-18) */
int sfm_set_stream(SfmHandle* h, void* hip_stream);              /* hipStream_t; NULL = null stream */

/* ---- geometry (CSR: offsets[K+1], points SoA) ------------------------------------------------------ */

/* BorderForce.__init__ (forces.py:127-136): K polylines, centre and full section_length as cull radius
 * (forces.py:149-150; obstacles.py:129-131,352-355). K = 0 clears. */
int sfm_set_borders(SfmHandle* h, int K, const int32_t* offsets, const float* px, const float* py,
                    const float* cx, const float* cy, const float* cull_len);
/* ObstacleForce.update_obstacles for the static force (forces.py:285-288; pedestrian_simulation.py:45-46). */
int sfm_set_static_obstacles(SfmHandle* h, int M, const int32_t* offsets, const float* px, const float* py,
                             const float* cx, const float* cy);
/* update_obstacles + update_obstacle_velocities for the dynamic force, once per tick
 * (pedestrian_simulation.py:108-115; forces.py:285-291). */
int sfm_set_dynamic_obstacles(SfmHandle* h, int M, const int32_t* offsets, const float* px, const float* py,
                              const float* cx, const float* cy, const float* vx, const float* vy);
/* The same from packed arrays (ABI 4): pts [P][2] {x, y}, cv [M][4] {cx, cy, vx, vy}; replaces the same two reference calls
 * (forces.py:285-291) for a caller that is handed the vehicles every tick (run_simulation.py:95, obstacles.py:297-329).
 * The arrays are copied before the call returns.  A small report (<= 64 KiB) is only STAGED: the next sfm_upload_state /
 * sfm_step_packed spreads it over the device arrays in its own launch, and any other call that reads them (a tick without an upload,
 * sfm_download_dynamic_obstacles, ...) does so first -- no caller-visible difference, one launch less per host-in-the-loop tick. */
int sfm_set_dynamic_obstacles_packed(SfmHandle* h, int M, const int32_t* offsets, const float* pts, const float* cv);

/* Device-side form of get_dynamic_obstacles (obstacles.py:297-329) for CARLA-free runs:
 * the vehicles are given once as oriented boxes -- ring-local offsets (the ellipse of obstacles.py:269-281 before
 * the transform; CSR offsets[M+1], ux, uy), centre, cos/sin of the yaw, velocity.  Ring points
 * p = c + R(yaw) u are generated on the device, and after every tick of sfm_run the centres advance by
 * step_length * v and the rings are regenerated (what the simulator + get_dynamic_obstacles do between ticks). */
int sfm_set_dynamic_boxes(SfmHandle* h, int M, const int32_t* offsets, const float* ux, const float* uy,
                          const float* cx, const float* cy, const float* yaw_cos, const float* yaw_sin,
                          const float* vx, const float* vy);
/* Current dynamic-obstacle centres (M) and ring points (P) as the next tick will see them; NULL skips. */
int sfm_download_dynamic_obstacles(SfmHandle* h, float* cx, float* cy, float* px, float* py);

/* ---- state ----------------------------------------------------------------------------------------- */

/* The numeric columns of PedState.state (pedestrian_state.py:17-19) as fp32 SoA: loc, vel, next_waypoint,
 * target_speed (already refreshed from the modes, :94-95), radius, and the border-force mask
 * mode in {CROSSING_ROAD, ROAD_TO_SIDEWALK} (forces.py:176-177).
 * z / vz may both be NULL (planar crowd: the 2-D kernel variant is used); radius may be NULL when
 * use_ped_radius is 0; crossing_mask may be NULL (all zero).  N = 0 is allowed (tick is a no-op, :60-61). */
int sfm_upload_state(SfmHandle* h, int N,
                     const float* x, const float* y, const float* z,
                     const float* vx, const float* vy, const float* vz,
                     const float* wx, const float* wy,
                     const float* target_speed, const float* radius, const uint8_t* crossing_mask);

/* Rows [i_begin, i_end) this handle computes and integrates (pedestrian index sharding across GPUs,
 * SURVEY.md section 8e).  Default after sfm_upload_state: [0, N).  All N pedestrians stay resident as the
 * j-operand set; after each tick the caller all-gathers the packed state (below) across ranks.
 * Rows are the library's internal order: for N >= 2048 sfm_upload_state packs the pedestrians spatially (sorted by x
 * into strips, each strip sorted by y: every 64-row tile is one rectangle of the map; a pure function of the uploaded
 * state, so every rank derives the same order), which makes a row block a vertical slab of the map.  A shard whose
 * bounds are multiples of 64 can use the symmetric pair kernel.  Every download writes a row's result at the CALLER's
 * index of that pedestrian; entries of pedestrians outside the shard are left untouched. */
int sfm_set_shard(SfmHandle* h, int i_begin, int i_end);
/* Row packing for sharded runs (no reference counterpart): cut the spatial packing into gx columns by x
 * and each column into gy blocks by y, block b = column * gy + position holding rows [bounds[b], bounds[b+1]) (multiples of
 * 64; NULL = equal split), every block strip-packed on its own.  A rank that owns the rows of one block then holds a compact
 * rectangle of the map instead of a slab across it: less boundary, fewer tile pairs evaluated one-sided by two ranks.  Takes
 * effect at the next sfm_upload_state / sfm_resort; gx = gy = 0 switches it off.  Results never depend on the packing. */
int sfm_set_partition(SfmHandle* h, int gx, int gy, const int32_t* bounds);

/* Counter-based waypoint stream of the synthetic scenarios: on arrival within `arrive_threshold`
 * (run_simulation.py:39) pedestrian i draws waypoint number k from hash(seed, i, k) in [0, world_side)^2. */
int sfm_set_waypoint_stream(SfmHandle* h, uint32_t seed, float world_side, float arrive_threshold);

/* Device-side pedestrian modes + waypoint queues + gap acceptance for device-resident runs (SURVEY.md section 8f
 * rows 1 and 3): what PedModeManager (ped_mode_manager.py:12-70), PedState.apply_current_mode /
 * update_next_waypoint (pedestrian_state.py:83-95), the CHECKING_TRAFFIC branch of PedestrianSimulation.tick
 * (pedestrian_simulation.py:63-73, check_traffic.py:7-61) and the arrival loop of SimulationRunner.tick
 * (run_simulation.py:118-132) do per pedestrian on the host.  All arrays are length N in the caller's index:
 * mode (PedMode values 0..4), target_speed (the mode objects' current target_speed), initial_speed,
 * crossing_speed, safety_margin, next_mode_time; wp_offsets[N+1] + wp_x/wp_y/wp_crossing: the remaining
 * waypoints of each pedestrian (waypoint_dict) with the crossing flag of the leg; despawn_on_arrival as in
 * run_simulation.py:41,127; sim_time0 = simulation time of the next tick; first_vehicle_extent = {ex, ey} of
 * vehicle 0 (check_traffic.py:35-36 uses it for every vehicle) or NULL.  While set, every tick starts with the
 * mode pass (target speeds, IDLE timers, gap acceptance, border-force mask) and arrivals pop the queue instead of
 * the hash stream; exhausted pedestrians are despawned (parked as ghosts, mode 255).  N = 0 switches it off.
 * Call after sfm_upload_state. */
int sfm_set_mode_fsm(SfmHandle* h, int N, const uint8_t* mode, const float* target_speed, const float* initial_speed,
                     const float* crossing_speed, const float* safety_margin, const float* next_mode_time,
                     const int32_t* wp_offsets, const float* wp_x, const float* wp_y, const uint8_t* wp_crossing,
                     int despawn_on_arrival, float sim_time0, const float* first_vehicle_extent);
/* Current mode (255 = despawned), mode target speed and queue cursor of every pedestrian; NULL skips. */
int sfm_download_modes(SfmHandle* h, uint8_t* mode, float* target_speed, int32_t* cursor);

/* ---- stepping -------------------------------------------------------------------------------------- */

/* One fused tick: F = sum of the enabled forces (pedestrian_simulation.py:81), v' = cap(v + dt*F,
 * max_speed_factor*target_speed) (:117-124; stateutils.py:18-23) written in place of v. */
int sfm_tick(SfmHandle* h, uint32_t flags);
/* `ticks` ticks back to back without host intervention (device-resident loop for benchmarks and the
 * CARLA-free harness; the loop of run_simulation.py:212-221 without the simulator); flags as above,
 * SFM_TICK_INTEGRATE is implied.  A whole crowd while the tile-pair list cutoff is off (default: up to 4096
 * pedestrians; SFM_CUTOFF=0: any size) -- planar or 3-D, with or
 * without border / obstacle forces and vehicles that move on the device -- takes ONE launch per tick here
 * (sfm_fused_tick_kernel) whatever `ticks` is, so what a run computes does not depend on how
 * the caller cuts it into calls; consecutive sfm_run / sfm_tick calls with no other call on the handle in
 * between carry on without the launch in front.  Results are those of `ticks` calls of sfm_tick up to the
 * order of the fp32 sums (each tick checked against the oracle to <= 1e-5). */
int sfm_run(SfmHandle* h, int ticks, uint32_t flags);
/* One integrating tick of a SHARD in two halves, so that the exchange of the previous tick's rows (the one all-gather per tick
 * of SURVEY.md section 8e; no reference counterpart, the reference is single-process) can run beside the part that does not
 * need it.  sfm_tick_begin: what only reads this handle's own rows -- their tile boxes, the border / obstacle forces, the tile
 * pairs whose two tiles are both own.  sfm_tick_end: once every other rank's rows are current in the packed state -- the pairs
 * with the other ranks' tiles, the epilogue (pedestrian_simulation.py:81-83, :117-124).  Results are those of sfm_tick with
 * SFM_TICK_INTEGRATE (bit-identical: the same terms reach the same slab rows).  When the handle is not a tile-aligned shard
 * on the symmetric path with the tile-pair list, sfm_tick_begin does nothing and sfm_tick_end runs the whole tick. */
int sfm_tick_begin(SfmHandle* h, uint32_t flags);
int sfm_tick_end(SfmHandle* h, uint32_t flags);

/* sfm_run that also records the trajectory (replaces the per-tick full-state Python
 * copy of PedState.record_current_state, pedestrian_state.py:100-104): frame f holds {x, y, vx, vy} of every
 * pedestrian (caller's index order) BEFORE tick f*stride, like the reference records before computing forces
 * (pedestrian_simulation.py:76).  frames: [max_frames][N][4] floats; *n_frames receives ceil(ticks/stride)
 * (clamped to max_frames).  Copies are asynchronous device->pinned-host between the ticks. */
int sfm_run_recorded(SfmHandle* h, int ticks, uint32_t flags, int stride, float* frames, int max_frames, int* n_frames);

/* ---- results --------------------------------------------------------------------------------------- */

/* get_new_velocities (pedestrian_simulation.py:126-127): v' of this handle's shard rows, written at
 * [i_begin, i_end) of the caller's length-N arrays.  vz may be NULL. */
int sfm_download_velocities(SfmHandle* h, float* vx, float* vy, float* vz);
/* One host-in-the-loop tick in one call (ABI 4): PedestrianSimulation.tick's numeric part as the reference's main loop drives it
 * (run_simulation.py:87-114: update_ped_info -> tick -> get_new_velocities; pedestrian_simulation.py:57-83, pedestrian_state.py:79-95)
 * = sfm_upload_state + sfm_tick + sfm_download_velocities on one packed block.
 *   rows  [N][9]  {x, y, vx, vy, waypoint x, waypoint y, target_speed, radius, border-force-off flag (0 / 1)}
 *   zvz   [N][2]  {z, vz}, or NULL: planar crowd (all z equal, no v_z)
 *   v_out [N][3]  {vx', vy', vz'}, the caller's index order.
 * Errors as the three calls it stands for. */
int sfm_step_packed(SfmHandle* h, int N, const float* rows, const float* zvz, uint32_t flags, float* v_out);
/* The same straight from the caller's pedestrian RECORDS (ABI 5): PedestrianState's structured-array rows (pedestrian_state.py:17-23),
 * `stride` bytes apart, float64 fields at byte offsets field_offsets[5] = {loc[3], vel[3], next_waypoint[3], radius, target_speed} (they
 * need not be aligned); border_off [N] = 1 where the border force is off (modes CROSSING_ROAD / ROAD_TO_SIDEWALK, forces.py:176-177),
 * or NULL.  The crowd is taken as planar iff all z are equal and every v_z is 0 -- or, planar_tolerance >= 0, iff the spread of z and
 * every |v_z| are within it; *was_planar (may be NULL) says which bodies ran.  v_out [N][3] as above. */
int sfm_step_records(SfmHandle* h, int N, const void* records, int64_t stride, const int32_t* field_offsets, const uint8_t* border_off,
                     float planar_tolerance, uint32_t flags, float* v_out, int32_t* was_planar);

/* Whole numeric state of the shard rows (any pointer may be NULL to skip that column). */
int sfm_download_state(SfmHandle* h, float* x, float* y, float* z, float* vx, float* vy, float* vz,
                       float* wx, float* wy);
/* Force.get_force (forces.py:28-32) of force `which` (SFM_FORCE_*, or SFM_FORCE_TOTAL) for the shard rows,
 * from the last tick run with SFM_TICK_RECORD_FORCES.  fz may be NULL. */
int sfm_download_forces(SfmHandle* h, int which, float* fx, float* fy, float* fz);
/* get_arrived_peds (pedestrian_simulation.py:88-97): mask[i] = |wp_xy - x_xy| < threshold, current state. */
int sfm_get_arrived(SfmHandle* h, float threshold, uint8_t* mask);
/* Number of waypoint draws per pedestrian so far (shard rows). */
int sfm_download_draw_counts(SfmHandle* h, uint32_t* counts);

/* ---- multi-GPU plumbing ---------------------------------------------------------------------------- */

/* Device pointer to the packed j-operand state the NEXT tick will read: N_pad records of 4 floats
 * {x, y, vx, vy}.  After a tick the shard rows are fresh; the caller all-gathers the other rows into
 * this buffer in place (one RCCL all-gather per tick).  *n_pad receives the padded record count. */
void* sfm_packed_state_ptr(SfmHandle* h, int* n_pad);
/* Same for the {z, vz} records (2 floats) of the 3-D variant; NULL when the crowd is planar. */
void* sfm_packed_z_ptr(SfmHandle* h);
/* Rows are kept in an internal spatial order (compact 64-row tiles); a device-resident whole-crowd run re-packs them
 * every SFM_RESORT_EVERY ticks by itself.  A sharded run cannot: a rank only keeps its own rows' waypoints and draw
 * counters current.  Its driver therefore, every few dozen ticks, all-gathers the two per-row arrays below (row-indexed
 * like the packed state, N_pad rows; which = 0: {waypoint x, waypoint y, target speed, radius}, 16 bytes per row;
 * which = 1: waypoint draw counter, 4 bytes per row) and then calls sfm_resort on every rank: the same state gives the
 * same order everywhere, and rank r goes on with rows [lo, hi) of the new order.  No-op when the order is off (N < 2048).
 * SFM_ERR_STATE on a shard whose mode state machine lives on the device (that state is keyed by pedestrian, not gathered). */
void* sfm_row_data_ptr(SfmHandle* h, int which, int* bytes_per_row);
int sfm_resort(SfmHandle* h);

/* ---- diagnostics ----------------------------------------------------------------------------------- */

const char* sfm_last_error(const SfmHandle* h);
/* HIP-event time of the last sfm_tick / sfm_run on its stream: total ms, ticks it covered and kernel
 * launches it issued. */
int sfm_get_timing(SfmHandle* h, float* elapsed_ms, int* ticks, int* launches);
/* Switches the HIP-event bracket of sfm_tick / sfm_run off (enable = 0) or back on (default): the two event records
 * cost ~11 us per call, which shows when a call is a few hundred microseconds of work (no reference counterpart: the
 * reference times nothing).  While off, sfm_get_timing fails with SFM_ERR_STATE. */
int sfm_set_timing(SfmHandle* h, int enable);
/* Times the DOMINANT kernel of a tick on its own: `reps` back-to-back launches of the pedestrian-pair kernel the
 * current state would use (the symmetric tile-pair kernel, or the ordered fused tick kernel with flags = 0),
 * bracketed by HIP events on the handle's stream.  The state is not advanced.  For roofline accounting. */
int sfm_profile_dominant_kernel(SfmHandle* h, int reps, float* avg_us);
/* Name of the pair kernel variant the last tick used (for profiles), static storage. */
const char* sfm_kernel_variant(const SfmHandle* h);
/* Work the symmetric pair kernel did in the last tick (measurement only; no reference counterpart -- the reference
 * always evaluates all N(N-1) ordered pairs, forces.py:74-117): tile-pair work items (whole 2-D grid, or the compacted
 * list when the provably-negligible tile pairs are cut) and the Moussaid terms evaluated for them (one per unordered
 * pair of a two-sided item).  SFM_ERR_STATE if the last tick used the ordered kernel. */
int sfm_get_pair_work(SfmHandle* h, long long* tile_pair_items, long long* pair_terms);
int sfm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SFM_HIP_H */
