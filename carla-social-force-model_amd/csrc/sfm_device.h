// sfm_device.h -- device-side data layout and folded constants shared by the kernels and the C ABI.
// gfx950 (MI355X) only.  See DESIGN.md for the layout rationale.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfm {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;            // 4 waves per workgroup
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int TILE_J = BLOCK;         // j-records staged per LDS tile (one float4 per thread)
constexpr float TINY = 1e-30f;        // added under every rsq: zero vectors normalise to zero without a guard op
constexpr float COINCIDENT_RINV = 1e14f;  // rsq(d2) above this <=> a coincident (or < 1e-14 m) valid pair

// Moussaid interaction constants, folded on the host in double and rounded once
// (forces.py:85-109 for pedestrians, :241-264 for obstacles).
struct IxConst {
    float lam;    // lambda                                   D = lam*(v_i - v_j) + e
    float eg;     // epsilon*gamma                            theta = angle - eg*|D|          (B = gamma*|D|)
    float c1;     // -log2(e)/gamma                           -d/B * log2(e) = d * (1/|D|) * c1
    float k1;     // -(n_prime*gamma)^2 * log2(e)             -(n' B theta)^2 * log2(e) = (|D| theta)^2 * k1
    float k2;     // -(n*gamma)^2 * log2(e)
    float negA;   // -A
    float thr2;   // perception_threshold^2 (obstacles only)
    float pad;
};

struct Geo {                  // CSR polylines / rings
    const int* off;           // [K+1]
    const float2* pts;        // [P] {x, y}
    const float4* ctr;        // [K] borders: {cx, cy, section_length^2, 0}; obstacles: {cx, cy, vx, vy}
    const float4* seg;        // [2K] borders only: {ax, ay, abx, aby}, {1/|ab|^2, max deviation from segment ab, P-1 if the border is a
                              // straight uniformly sampled line (verified on the host) else 0, 0}
    int K;
};

// Device-side pedestrian mode state machine + waypoint queues: the host loop of
// ped_mode_manager.py:30-70, pedestrian_state.py:83-95, pedestrian_simulation.py:63-73 and
// run_simulation.py:118-132 for device-resident runs.  mode == null: off.
constexpr uint8_t MODE_IDLE = 0, MODE_WALKING = 1, MODE_CROSSING = 2, MODE_ROAD_TO_SIDEWALK = 3, MODE_CHECKING = 4,
                  MODE_DESPAWNED = 255;
struct FsmArgs {
    uint8_t* mode;            // PedMode per pedestrian (MODE_DESPAWNED once removed)
    float* target;            // the mode object's target_speed (applied to the state one tick later, like the reference)
    const float* initial_speed;
    const float* crossing_speed;
    const float* safety_margin;
    const float* next_mode_time;
    const int* wp_off;        // [N+1] CSR of the remaining-waypoint lists (waypoint_dict, run_simulation.py:120-126)
    const float2* wp_xy;
    const uint8_t* wp_cross;  // 1: the leg towards this waypoint crosses a road
    int* cursor;              // next unused entry of each list
    float sim_time;
    float veh_ext_x, veh_ext_y;   // extent of the FIRST vehicle (check_traffic.py:35-36 offsets every vehicle by it)
    int despawn_on_arrival;
};

// Device-side vehicles (sfm_set_dynamic_boxes): after a tick's forces are computed the centres move by dt * v and the rings
// are regenerated.  The extra workgroups of the tick's last kernel do it (block index >= block0); M = 0: nothing to do.
struct DynAdvance {
    float4* ctr;              // [M] {cx, cy, vx, vy}
    const int* off;           // [M+1]
    const float2* local;      // [P] ring points in the vehicle frame
    const float2* rot;        // [M] {cos yaw, sin yaw}
    float2* pts;              // [P] ring points in the world frame (what the geometry kernel reads)
    int M;
    float dt;
    int block0;
    float4* ctr_out;          // fused tick with border / obstacle forces: the moved centres and rings go to the other half of a
    float2* pts_out;          // ping-pong (this launch's geometry workgroups still read ctr / pts); null: in place
};

struct TickArgs {
    // packed j-operand state, N_pad records (padding rows are never selected)
    const float4* pk_cur;     // {x, y, vx, vy}
    float4* pk_next;
    const float2* zv_cur;     // {z, vz}, 3-D variant only
    float2* zv_next;
    float4* own;              // {wx, wy, target_speed, radius}
    const float* radius;      // [N_pad] radius stream for the j side (use_ped_radius only)
    uint8_t* crossing;        // border-force mask (rewritten every tick by sfm_mode_kernel when the FSM is on)
    uint32_t* draws;          // waypoint draw counters
    const uint32_t* ids;      // caller's index of the pedestrian in each row (spatial reordering); null = identity
    float* rec;               // optional per-force record, layout [6][3][N]
    float4* host_pk;          // sfm_step_packed: the new rows are ALSO written here -- pinned host memory the device reaches by itself --
    float2* host_zv;          // so the caller's v' needs no copy after the tick (null: off)
    float* geo;               // geometry forces of this tick, layout [geo_slices][6][N_pad]: {fbx,fby,fsx,fsy,fdx,fdy}; null = none
    DynAdvance adv;
    int geo_slices;           // few tiles (small crowds): the polylines of a tile are split over this many workgroups, each
                              // leaving a partial sum; the consumer adds them in slice order
    int N, N_pad, i_begin, i_end;
    uint32_t flags;
    // parameters
    int en_acc, en_ped, en_border, en_static, en_dynamic;
    IxConst ped, stat, dyn;
    float border_a, border_nlb;   // a, -log2(e)/b
    float border_skip;            // distance beyond which a border term is < 2^-40 a (40 ln2 * b), <= 0: never skip
    float inv_tau, dt, max_speed_factor;
    // waypoint stream
    uint32_t seed;
    float world_side, arrive_thr2;
    Geo borders, statics, dynamics;
    // pedestrian-force cutoff at tile granularity (null = off): see tiles_negligible()
    const float4* tile_box;   // [n_t] {xmin, ymin, xmax, ymax} of the real pedestrians of each 64-tile
    const float* tile_vmax;   // [n_t] largest speed in the tile
    float cut_scale;          // gamma * 41 ln 2 (with margin): distance per unit of (lambda*(va+vb)+1)
    float cut_pad;            // 2 * largest radius when use_ped_radius, else 0
    float4* tile_box_out;     // the list cutoff of a whole crowd: the symmetric epilogue writes the boxes / speeds of
    float* tile_vmax_out;     // the NEXT tick's state here (saves the next tick's sfm_tile_bounds_kernel launch)
    FsmArgs fsm;
    // Flat tile-pair list of a whole crowd built by extra workgroups of the geometry launch (block index >= list_block0) instead of
    // a launch of its own: the boxes were left by the previous epilogue, so nothing stands in front of it.  list_work == null: off.
    uint32_t* list_work;
    int* list_count;
    int list_n_t, list_block0;
    uint32_t* list_idx;       // ... and the pool's index table (SymArgs::idx), null: dense slab
    uint32_t list_cap;        // ... and its capacity in row pairs
    unsigned long long* geo_stamps;   // diagnostic runs only (SFM_GEO_STAMPS): per geometry workgroup {start, after find, after scan, end} of s_memrealtime
};

// Symmetric (antisymmetry-exploiting) pedestrian-force path, single shard only.
struct SymArgs {
    float2* slab;        // [n_t][stride]: slab[u][i] = -A-less force on pedestrian i from all pedestrians of tile u
    float* slabz;        // 3-D crowds: its z component, same indexing (null: planar)
    int n_t;             // number of 64-pedestrian tiles
    int stride;          // n_t * 64
    int debug_steps;     // < 0: normal; >= 0: run only this many systolic steps per wave (timing probe, wrong results)
    const uint32_t* work;    // cutoff on: compacted list of (bx | shift << 16) tile-pair items, else null
    const int* work_count;
    // vmax (largest speed per tile, this tick's input state) is set whenever the list cutoff is on: the pair kernel then also
    // skips the systolic steps whose 64 pairs are all beyond the reach of the two tiles' speeds, or below the exponent bound
    const float* vmax;
    float cut_scale, cut_pad;
    unsigned long long* stamps;   // diagnostic builds of a run only (SFM_STAMPS): per workgroup {start, end} of s_memrealtime + HW id
    // two-level list building (large crowds): boxes / largest speeds of runs of `tps` consecutive tiles (= the x-strips of
    // the spatial packing); n_strips = 0: flat
    const float4* sbox;
    const float* svmax;
    int tps, n_strips;
    int t_lo, t_hi;      // tiles of this handle's rows (a shard; whole crowd: 0, n_t).  Pairs with a tile outside are
                         // evaluated one-sided: the other side belongs to another rank, which evaluates it itself
    int zero_count;      // epilogue launch: 1 = leave *work_count at 0 for the next tick's list kernel (saves its memset)
    // Round 4 -- list mode can keep its partial forces in a POOL of row pairs instead of the dense slab (8.6 GB at N = 262 144:
    // O(N^2 / 64); the pool is O(kept tile pairs)).  idx != null: `slab` / `slabz` ARE the pool, [2 * pairs][64]: list item p of this
    // launch owns row pair q = row_base + p -- row 2q the force on the travelling tile's pedestrians (a diagonal item: on tile bx's),
    // row 2q + 1 on the resident tile's (tile bx + half_up's) -- and the list kernels leave q in idx[(bx - t_lo) * n_t + shift], where
    // the epilogue looks it up for the partner tiles its predicate keeps.  An item beyond the list's share of the pool (p >= cap_pairs: a
    // crowd whose tiles overlap keeps everything) adds its sums to spill->ovf in 2^-36 fixed point by integer atomics -- exact,
    // order-independent, hence still deterministic -- and stamps spill->tick.  (Few fields on purpose: every one is two more SGPRs in
    // kernels that sit at the occupancy limit.)
    uint32_t* idx;       // [t_hi - t_lo][n_t]; null: dense slab
    uint32_t row_base;   // first row pair of this launch's list (the split tick's own-own list sits behind the main one)
    uint32_t cap_pairs;  // row pairs a list may use; item positions beyond spill
    struct SpillArgs* spill;
    int tick_serial;     // of the tick being computed (the two halves of a split tick share it)
};

struct SpillArgs {       // (device memory; read only by an item that spills and by the epilogue)
    long long* ovf;      // [4][N_pad]: x, y, z sums in 2^-36 fixed point, count of non-finite sums
    int n_pad;
    int tick;            // serial of the last tick in which an item spilled
};

// Fused tick of a whole planar crowd with the acceleration and pedestrian forces only (sfm_fused_tick_kernel):
// ONE launch per tick inside sfm_run.  Tiles go in groups of two; a workgroup owns one unordered group pair, first integrates its own
// four tiles from the previous launch's partial forces (every workgroup that needs a tile recomputes it -- the same arithmetic on
// the same operands, so all agree bit for bit; the workgroup of the group's diagonal item is the one that stores it), then
// evaluates its tile pairs of the NEW state.  State, waypoints and partial forces ping-pong between launches.
constexpr int PAIR_STAMP_WGS = 65536;   // experiments build: workgroups of the symmetric pair kernel whose {start, end, HW id} stamps are kept
constexpr int FUSED_STAMP_STRIDE = 40, FUSED_STAMP_WGS = 4096;   // experiments build: phase stamps of the fused tick (words per workgroup, workgroups kept)
constexpr int FUSED_GEO_SLICES_MAX = 8;  // geometry workgroups per tile of the fused tick (each leaves one partial sum per pedestrian)
struct FusedArgs {
    const float2* slab_prev;  // [n_g][N_pad]: slab[r][i] = -A-less force on pedestrian i from the pedestrians of group r, previous state
    float2* slab_next;        // the same for the state this launch integrates to
    const float* slabz_prev;  // 3-D crowds: the z components, same indexing (null: planar)
    float* slabz_next;
    const float4* own_cur;    // {wx, wy, target_speed, radius} going with pk_cur
    float4* own_next;
    int n_g;                  // groups of two tiles
    int n_t;
    int blocked;              // 1: XCD-aware order of the work items (n_g a multiple of 8)
    const float2* geo_prev;   // [geo_slices][N_pad]: border + obstacle forces on the stored state, one partial sum per slice of the polylines
    float2* geo_next;         // the same for the state this launch integrates to
    int geo_slices;           // <= FUSED_GEO_SLICES_MAX
    int n_geo_wg;             // geometry workgroups in front of the grid = n_t * geo_slices (0: the crowd has no such forces)
    int n_pair_wg;            // pair workgroups behind them
    unsigned long long* stamps;   // experiments build only (SFM_FUSED_STAMPS): per workgroup FUSED_STAMP_STRIDE x s_memrealtime -- entry, column sums in, state in LDS, pairs done, end; per wave: steps done, past the barrier
    int mode;                 // 0: the stored state is the state (nothing to integrate, nothing stored but slab_next): the launch in
                              //    front of a run; 1: integrate by one tick, store, then the pairs of the new state
};

// Block-major packing for sharded runs (sfm_set_partition, sfm_reorder.hip): the row order is cut into gx columns by x, each
// column into gy blocks by y -- block b = column * gy + position holds rows [bound[b], bound[b+1]) -- and every block is
// strip-packed on its own, so a rank's contiguous row range is a compact rectangle of the map instead of a slab across it.
constexpr int MAX_BLOCKS = 16;
struct BlockPlan {
    int n_blocks, gy;                 // n_blocks = gx * gy (1: plain strip packing of the whole crowd)
    int bound[MAX_BLOCKS + 1];        // row bounds of the blocks, multiples of 64, bound[n_blocks] >= N
    int strip_rows[MAX_BLOCKS];       // rows per strip inside each block
};

}  // namespace sfm
