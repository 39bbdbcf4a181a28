// sfm_capi.hip -- host side of libsfm_hip.so: the C ABI declared in include/sfm_hip.h.
// Owns the device buffers (fp32, packed for 16-B/lane streaming), folds the TOML parameters into kernel
// constants, launches the fused tick, and moves results back.  No torch types; a hipStream_t comes in as
// a void*.
#include "sfm_device.h"
#include "sfm_hip.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace sfm {
hipError_t launch_tick(int ipw, int team, bool z3, bool rad, const TickArgs& a, hipStream_t st);
hipError_t launch_arrived(const float4* pk, const float4* own, int N, float thr2, uint8_t* mask, hipStream_t st);
hipError_t launch_sym_list(const TickArgs& a, const SymArgs& sa, hipStream_t st, int partners = 0, bool count_is_zero = false);
hipError_t launch_sym_pair(bool rad, const TickArgs& a, const SymArgs& sa, hipStream_t st);
hipError_t launch_sym_epilogue(bool rad, const TickArgs& a, const SymArgs& sa, hipStream_t st);
hipError_t launch_fused_tick(bool rad, const TickArgs& a, const FusedArgs& f, hipStream_t st, int waves);
int fused_pair_workgroups(int n_g);
hipError_t launch_sym_pair_geo(bool rad, const TickArgs& a, const SymArgs& sa, hipStream_t st);
int sym_item_count(int n_t);
hipError_t launch_strip_bounds(const float4* box, const float* vmax, int n_t, int tps, int n_strips, float4* sbox, float* svmax,
                               hipStream_t st);
hipError_t launch_tile_strip_bounds(const float4* pk, int N, int n_t, int tps, int n_strips, float4* box, float* vmax, float4* sbox,
                                    float* svmax, hipStream_t st);
hipError_t launch_geometry(bool rad, const TickArgs& a, hipStream_t st);
hipError_t launch_modes(const TickArgs& a, hipStream_t st);
struct ReorderBufs { unsigned long long *key64_in, *key64_out; uint32_t *row_a, *row_b, *key32_in, *key32_out; void* temp; size_t temp_bytes; };
size_t reorder_temp_bytes(int N);
hipError_t launch_resort(const float4* pk, int N, int strip_rows, const ReorderBufs& b, hipStream_t st);
hipError_t launch_resort_blocks(const float4* pk, int N, const BlockPlan& pl, const ReorderBufs& b, hipStream_t st);
hipError_t launch_unpack_geo(const char* block, float4* ctr, int K, float2* pts, int P, int* off, bool with_off, hipStream_t st);
hipError_t launch_unpack_rows(const char* block, size_t b_own, size_t b_zv, size_t b_rr, size_t b_cm, int n_pad, float4* pk0,
                              float4* pk1, float4* own, float2* zv0, float2* zv1, float* radius, uint8_t* crossing, uint32_t* draws,
                              const char* geo_block, float4* ctr, int K, float2* pts, int P, int* off, bool with_off, hipStream_t st);
hipError_t launch_gather(const uint32_t* src, int N, const float4* pk_in, float4* pk_out, const float2* zv_in, float2* zv_out,
                         const float4* own_in, float4* own_out, const float* rad_in, float* rad_out, const uint8_t* cr_in,
                         uint8_t* cr_out, const uint32_t* dr_in, uint32_t* dr_out, const uint32_t* id_in, uint32_t* id_out,
                         hipStream_t st);
hipError_t launch_tile_bounds(const float4* pk, const float2* zv, int N, float4* box, float* vmax, hipStream_t st, int t_lo = 0,
                              int t_hi = -1);
int probe_dpp_direction(hipStream_t st);
hipError_t launch_dynamic_boxes(float4* ctr, const int* off, const float2* local, const float2* rot, float2* pts, int M,
                                float dt, int advance, hipStream_t st);
}  // namespace sfm

using namespace sfm;

static thread_local std::string g_create_error;

constexpr int GEO_SLICES_MAX = 16;

struct DevGeo {
    int* off = nullptr;
    float2* pts = nullptr;
    float4* ctr = nullptr;
    float4* seg = nullptr;
    int K = 0;
    int P = 0;
    size_t off_cap = 0, pts_cap = 0, ctr_cap = 0;      // grow-only (the dynamic obstacles arrive every tick)
    char* stage = nullptr;                              // pinned host block the three arrays are copied from, asynchronously
    size_t stage_cap = 0;
    std::vector<int> off_host;                          // the offsets the device holds (vehicles arrive every tick with the same ring sizes: not copied again)
    bool lazy = false;                                  // the staged block has not been spread over the device arrays yet: the next state upload's
    bool lazy_off = false;                              // unpack launch takes it along (flush_lazy_geo for everything else that reads the arrays)
};

struct SfmHandle {
    int device = 0;
    hipStream_t stream = nullptr;
    SfmParams prm{};
    std::string err;

    int N = 0, N_pad = 0, cap = 0;        // cap = allocated records
    bool z3 = false, rad = false;
    int i_begin = 0, i_end = 0;
    int cur = 0;
    float4* pk[2] = {nullptr, nullptr};
    float2* zv[2] = {nullptr, nullptr};
    float4* own = nullptr;
    float* radius = nullptr;
    uint8_t* crossing = nullptr;
    uint8_t* arrived = nullptr;
    uint32_t* draws = nullptr;
    float* rec = nullptr;                 // [6][3][N]
    float* geo = nullptr;                 // [6][N_pad] geometry forces of the current tick
    bool rec_valid = false;
    DevGeo borders, statics, dynamics;
    float2* dyn_local = nullptr;          // device-side vehicles: ring-local offsets [P] and {cos,sin} yaw [M]
    float2* dyn_rot = nullptr;
    bool dyn_boxes = false;

    // symmetric pedestrian-force path (single shard, planar, no radius)
    float2* slab = nullptr;
    float* slabz = nullptr;                // 3-D crowds: the z components (same indexing)
    int n_t = 0;
    size_t slab_cap = 0, slabz_cap = 0;
    // list mode (round 4): row pairs of the pool instead of the dense slab -- O(kept tile pairs) instead of O(N^2 / 64)
    float2* pool = nullptr;                // [2 * pool_pairs][64]
    float* poolz = nullptr;
    size_t pool_pairs = 0, poolz_pairs = 0;
    uint32_t pool_main = 0;                // row pairs of the main list; the split tick's own-own list owns the ones behind
    bool pooled = false;                   // this tick's list-mode launches use the pool (else the dense slab)
    uint32_t* pair_idx = nullptr;          // [own tiles][n_t] -> row pair (SymArgs::idx)
    size_t pair_idx_cap = 0;
    long long* ovf = nullptr;              // [4][N_pad] overflow sums (pairs beyond the pool's capacity)
    size_t ovf_cap = 0;
    SpillArgs* spill = nullptr;            // device twin of {ovf, N_pad, serial of the last spilling tick}
    long long* spill_ovf = nullptr;        // what the device twin holds
    int spill_n_pad = 0;
    int tick_serial = 0;
    int dpp_dir = 0;
    int sym_mode = -1;                     // SFM_SYM: 0 off, 1 on when eligible, -1 auto
    // tile-granular cutoff of provably negligible pedestrian pairs
    float4* tile_box = nullptr;            // [2][n_t]: ping-pong (the epilogue of tick k writes the boxes of tick k+1)
    float* tile_vmax = nullptr;
    char* up_stage = nullptr;              // pinned host block sfm_upload_state assembles the rows in: one async copy to its
    size_t up_stage_cap = 0;               // device twin, one kernel spreads it over the arrays
    char* up_block = nullptr;
    size_t up_block_cap = 0;
    size_t box_cap = 0, strip_cap = 0;
    float4* strip_box = nullptr;           // [n_t]: boxes / speeds of runs of tiles (two-level list building, n_t >= 1024)
    float* strip_vmax = nullptr;
    int box_cur = 0;
    bool boxes_valid = false;
    bool count_zeroed = false;             // the last epilogue left the list counter at 0
    int carry_mode = -1;                   // SFM_CARRY=0: boxes / list counter are rebuilt every tick (A/B)
    int geo_slices_override = 0;           // SFM_GEO_SLICES / SFM_STRIPS / SFM_DEBUG_STEPS, read once at sfm_create (tests, probes)
    int strips_override = -1;
    int debug_steps = -1;
    uint32_t* work = nullptr;
    int* work_count = nullptr;
    size_t work_cap = 0;
    int cut_mode = -1;                     // SFM_CUTOFF: 0 off, 1 on, -1 auto (on above AUTO_CUTOFF_N pedestrians)
    unsigned long long* stamps = nullptr;  // SFM_STAMPS diagnostic: per-workgroup timestamps of the symmetric pair kernel
    unsigned long long* geo_stamps = nullptr;   // SFM_GEO_STAMPS diagnostic: per-workgroup phase stamps of the geometry kernel
    // spatial reordering: row s holds the caller's pedestrian perm[s] (strips in x, each sorted by y: sfm_reorder.hip), so the 64-tiles
    // are compact squares; every download translates back.  Identity when off.
    std::vector<uint32_t> perm;
    uint32_t* ids = nullptr;
    bool reordered = false;
    int reorder_mode = -1;                 // SFM_REORDER: 0 off, 1 on, -1 auto (N >= 2048)
    // periodic device re-sort (sfm_reorder.hip): alternate copies of the per-row arrays + sort scratch
    float4* own2 = nullptr;
    float* radius2 = nullptr;
    uint8_t* crossing2 = nullptr;
    uint32_t *draws2 = nullptr, *ids2 = nullptr, *sort_buf = nullptr;
    void* sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    int sort_cap = 0;
    int strip_rows = WAVE;                      // rows per x-strip of the spatial packing (multiple of 64)
    // sharded runs (sfm_set_partition): gx x gy blocks, one per rank, each strip-packed on its own; part_gx = 0: off
    double pack_aspect = 1.0;                   // x extent / y extent of the crowd at upload
    int part_gx = 0, part_gy = 0;
    std::vector<int> part_bounds;               // [gx*gy + 1] row bounds (empty: equal split of the padded row count)
    int resort_every = 64, ticks_since_sort = 0;
    bool resort_every_set = false;              // SFM_RESORT_EVERY given: every path re-packs at exactly that period
    bool perm_stale = false;
    float r_max = 0.f;
    bool used_sym = false;
    // fused tick (sfm_fused_tick_kernel): one launch per tick inside sfm_run for a whole crowd while the list cutoff is off (default: N <= 4096) --
    // planar or 3-D, with or without border / obstacle forces and device-side vehicles
    float2* fslab = nullptr;               // [2][n_g][N_pad] partial forces, ping-pong across launches
    size_t fslab_cap = 0;
    float* fslabz = nullptr;               // 3-D crowds: their z components
    size_t fslabz_cap = 0;
    float4* own_alt = nullptr;             // the waypoints ping-pong with the state
    int own_alt_cap = 0;
    int pair_geo_mode = -1;                // SFM_PAIR_GEO=0: the geometry kernel always gets a launch of its own (A/B, tests)
    int list_merge_mode = -1;              // SFM_LIST_MERGE=0: the flat tile-pair list always gets a launch of its own (A/B, tests)
    int fused_mode = -1;                   // SFM_FUSED=0: always the two-kernel tick (A/B, tests)
    int fused_waves = 16, fused_blocked = 1;   // SFM_FUSED_WAVES=8 / SFM_FUSED_BLOCKED=0: its A/B variants (tests)
    float2* fgeo = nullptr;                // [2][FUSED_GEO_SLICES_MAX][N_pad] border + obstacle forces of the fused tick, one float2 per pedestrian and slice, ping-pong
    size_t fgeo_cap = 0;
    float4* dyn_ctr_alt = nullptr;         // device-side vehicles in the fused tick: the NEXT tick's centres / rings (ping-pong with dynamics.ctr / .pts)
    float2* dyn_pts_alt = nullptr;
    size_t dyn_ctr_alt_cap = 0, dyn_pts_alt_cap = 0;
    bool hosted = false;                   // inside sfm_step_packed: the tick's last kernel also writes the new rows to down_stage
    char* down_stage = nullptr;            // pinned host block [N_pad float4 | N_pad float2] the device writes v' into
    size_t down_stage_cap = 0;
    std::vector<float> step_cols;          // sfm_step_packed: the packed block taken apart into the columns sfm_upload_state consumes
    std::vector<uint8_t> step_mask;
    unsigned long long* fused_stamps = nullptr;   // experiments build, SFM_FUSED_STAMPS=<file>: phase stamps of the last fused launch
    int fused_geo_slices = 0;              // SFM_FUSED_GEO_SLICES: A/B
    int fused_geo_mode = -1;               // SFM_FUSED_GEO=0: crowds with border / obstacle forces keep the two-launch tick (A/B, tests)
    bool used_fused = false;
    // A fused run ends with the partial forces of its final state already in fslab: the next sfm_run / sfm_tick carries on from
    // there (one launch per tick, no start-up launch) provided NOTHING else was called on the handle in between -- api_seq counts
    // the entry points that went through bind(), the few that do not reset carry_ok themselves.
    unsigned long long api_seq = 0, carry_seq = 0;
    bool carry_ok = false;
    int carry_sl = 0;
    bool last_list = false;                // the last symmetric tick ran from the tile-pair list (cutoff on)

    uint32_t seed = 0;
    float world_side = 0.f, arrive_thr = 2.0f;

    // device-side mode FSM + waypoint queues (caller's index space)
    bool fsm_on = false;
    int fsm_n = 0;
    uint8_t* f_mode = nullptr;
    float *f_target = nullptr, *f_initial = nullptr, *f_crossing = nullptr, *f_margin = nullptr, *f_next = nullptr;
    int *f_off = nullptr, *f_cursor = nullptr;
    float2* f_xy = nullptr;
    uint8_t* f_cross = nullptr;
    float sim_time = 0.f, veh_ext[2] = {0.f, 0.f};
    int despawn = 0;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // geometry forces run on a side stream beside the pair kernel (they only need the tick's input state)
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap_geo = true;
    // sharded runs: the geometry forces of tick t+1 only need this rank's own rows of the new state, so they are launched on the
    // side stream right after tick t's epilogue -- beside the all-gather the caller issues next -- and tick t+1 only joins them
    bool geo_ahead = false;
    int geo_ahead_mode = -1;               // SFM_GEO_AHEAD=0 switches it off
    // split tick of a shard (sfm_tick_begin / sfm_tick_end): the own-own tile pairs are listed and evaluated before the other
    // ranks' rows have arrived; their list lives in work2 / work_count[1]
    uint32_t* work2 = nullptr;
    size_t work2_cap = 0;
    bool begin_done = false, begin_forked = false, last_split = false;
    int begin_geo_slices = 0;              // > 0: sfm_tick_begin put the geometry workgroups into its pair launch, in this many slices per tile
    uint32_t begin_flags = 0;
    int split_mode = -1;                   // SFM_SPLIT=0: sfm_tick_begin never does anything (A/B, tests)
    int timed_ticks = 0, timed_launches = 0;
    bool timing_valid = false;
    bool timing_on = true;                 // sfm_set_timing: the two event records per call cost ~11 us (a 20-tick sfm_run of c2: 3 %)
    int ipw_last = 0;
    int ipw_override = 0, team_override = 0;
    char variant[64] = "none";
};

#define HIP_TRY(h, call)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return SFM_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

static int fail(SfmHandle* h, int code, const char* msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

static int bind(SfmHandle* h) {
    if (!h) return SFM_ERR_INVALID;
    ++h->api_seq;
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) { h->err = std::string("hipSetDevice: ") + hipGetErrorString(e); return SFM_ERR_HIP; }
    return SFM_OK;
}

// a geometry kernel launched ahead of its tick (SfmHandle::geo_ahead) reads state the caller is about to change: let it finish,
// forget its result
static void drop_geo_ahead(SfmHandle* h) {
    if (h && h->geo_ahead) {
        hipStreamSynchronize(h->aux);
        h->geo_ahead = false;
    }
    if (h) h->begin_done = false;          // (a half-done split tick is simply redone in full)
}

template <typename T>
static hipError_t dev_realloc(T*& p, size_t count) {
    if (p) { hipError_t e = hipFree(p); p = nullptr; if (e != hipSuccess) return e; }
    if (count == 0) return hipSuccess;
    return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
}

// grow-only variant for buffers that are re-filled every tick by a host-in-the-loop caller: hipMalloc / hipFree cost
// tens of microseconds and synchronise the device
template <typename T>
static hipError_t dev_reserve(T*& p, size_t& cap, size_t count) {
    if (count <= cap && p) return hipSuccess;
    const size_t want = count + count / 2 + 16;
    hipError_t e = dev_realloc(p, want);
    cap = (e == hipSuccess) ? want : 0;
    return e;
}

static IxConst fold(const SfmInteraction& s) {
    const double log2e = 1.4426950408889634;
    IxConst c{};
    c.lam = (float)s.lambda;
    c.eg = (float)((double)s.epsilon * (double)s.gamma);
    c.c1 = (float)(-log2e / (double)s.gamma);
    c.k1 = (float)(-((double)s.n_prime * s.gamma) * ((double)s.n_prime * s.gamma) * log2e);
    c.k2 = (float)(-((double)s.n * s.gamma) * ((double)s.n * s.gamma) * log2e);
    c.negA = (float)(-(double)s.A);
    c.thr2 = (float)((double)s.perception_threshold * (double)s.perception_threshold);
    return c;
}

// Environment knobs.  The product build reads eleven (SFM_SYM, SFM_IPW, SFM_TEAM, SFM_CUTOFF, SFM_REORDER,
// SFM_RESORT_EVERY, SFM_FUSED, SFM_STRIPS, SFM_PAIR_GEO, SFM_GEO_SLICES, SFM_NO_STRAIGHT -- each selects a SHIPPING path the tests must be able to
// reach at a small size).  Everything else -- A/B arrangements measured and dropped, diagnostic stamps, timing probes -- only exists
// in a build with -DSFM_EXPERIMENTS (make EXPERIMENTS=1).
static inline const char* exp_env(const char* name) {
#ifdef SFM_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

static int check_params(const SfmParams* p, const char** why) {
    if (!p) { *why = "params is NULL"; return 0; }
    if (!(p->step_length > 0.f)) { *why = "step_length must be > 0"; return 0; }
    if (!(p->tau > 0.f)) { *why = "tau must be > 0"; return 0; }
    if (p->enabled[SFM_FORCE_PEDESTRIAN] && !(p->pedestrian.gamma != 0.f)) { *why = "pedestrian.gamma is 0"; return 0; }
    if (p->enabled[SFM_FORCE_BORDER] && !(p->border_b != 0.f)) { *why = "border_force.b is 0"; return 0; }
    return 1;
}

extern "C" {

int sfm_abi_version(void) { return SFM_ABI_VERSION; }

int sfm_create(const SfmParams* params, int device_id, SfmHandle** out) {
    if (!out) return fail(nullptr, SFM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    const char* why = nullptr;
    if (!check_params(params, &why)) return fail(nullptr, SFM_ERR_INVALID, why);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, SFM_ERR_NO_DEVICE, "no HIP device visible (libsfm_hip needs an MI355X)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, SFM_ERR_INVALID, "device_id out of range");
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return fail(nullptr, SFM_ERR_HIP, hipGetErrorString(e));
    SfmHandle* h = new SfmHandle();
    h->device = device_id;
    h->prm = *params;
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        delete h;
        return fail(nullptr, SFM_ERR_HIP, "hipEventCreate failed");
    }
    if (hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        h->aux = nullptr;
        h->overlap_geo = false;
    }
    if (exp_env("SFM_NO_OVERLAP")) h->overlap_geo = false;
    const char* ov = getenv("SFM_IPW");
    if (ov) h->ipw_override = atoi(ov);
    ov = getenv("SFM_TEAM");
    if (ov) h->team_override = atoi(ov);
    ov = getenv("SFM_SYM");
    if (ov) h->sym_mode = atoi(ov);
    ov = getenv("SFM_CUTOFF");
    if (ov) h->cut_mode = atoi(ov);
    ov = exp_env("SFM_CARRY");
    if (ov) h->carry_mode = atoi(ov);
    ov = exp_env("SFM_SPLIT");
    if (ov) h->split_mode = atoi(ov);
    ov = exp_env("SFM_GEO_AHEAD");
    if (ov) h->geo_ahead_mode = atoi(ov);
    ov = getenv("SFM_PAIR_GEO");
    if (ov) h->pair_geo_mode = atoi(ov);
    ov = exp_env("SFM_LIST_MERGE");
    if (ov) h->list_merge_mode = atoi(ov);
    ov = getenv("SFM_FUSED");
    if (ov) h->fused_mode = atoi(ov);
    ov = exp_env("SFM_FUSED_WAVES");
    if (ov && atoi(ov) == 8) h->fused_waves = 8;
    ov = exp_env("SFM_FUSED_BLOCKED");
    if (ov) h->fused_blocked = atoi(ov);
    ov = exp_env("SFM_FUSED_GEO");
    if (ov) h->fused_geo_mode = atoi(ov);
    ov = exp_env("SFM_FUSED_GEO_SLICES");
    if (ov) h->fused_geo_slices = atoi(ov);
    ov = getenv("SFM_REORDER");
    if (ov) h->reorder_mode = atoi(ov);
    ov = getenv("SFM_GEO_SLICES");
    if (ov) h->geo_slices_override = std::min(GEO_SLICES_MAX, std::max(1, atoi(ov)));
    ov = getenv("SFM_STRIPS");
    if (ov) h->strips_override = atoi(ov) != 0 ? 1 : 0;
    ov = exp_env("SFM_DEBUG_STEPS");
    if (ov) h->debug_steps = atoi(ov);
    ov = getenv("SFM_RESORT_EVERY");
    if (ov) { h->resort_every = atoi(ov); h->resort_every_set = true; }
    if (exp_env("SFM_STAMPS")) {
        if (hipMalloc(reinterpret_cast<void**>(&h->stamps), sizeof(unsigned long long) * 3 * PAIR_STAMP_WGS) != hipSuccess) h->stamps = nullptr;
        else {
            hipMemset(h->stamps, 0, sizeof(unsigned long long) * 3 * PAIR_STAMP_WGS);
            FILE* f = fopen(exp_env("SFM_STAMPS"), "w");   // the address is read back by the diagnostic script through the dump below
            if (f) fclose(f);
        }
    }
    if (exp_env("SFM_FUSED_STAMPS")) {
        const size_t words = (size_t)FUSED_STAMP_STRIDE * FUSED_STAMP_WGS;
        if (hipMalloc(reinterpret_cast<void**>(&h->fused_stamps), sizeof(unsigned long long) * words) != hipSuccess) h->fused_stamps = nullptr;
        else hipMemset(h->fused_stamps, 0, sizeof(unsigned long long) * words);
    }
    if (exp_env("SFM_GEO_STAMPS")) {
        if (hipMalloc(reinterpret_cast<void**>(&h->geo_stamps), sizeof(unsigned long long) * 4 * 8192) != hipSuccess) h->geo_stamps = nullptr;
        else hipMemset(h->geo_stamps, 0, sizeof(unsigned long long) * 4 * 8192);
    }
    h->dpp_dir = probe_dpp_direction(nullptr);
    *out = h;
    return SFM_OK;
}

static void free_geo(DevGeo& g) {
    if (g.off) hipFree(g.off);
    if (g.pts) hipFree(g.pts);
    if (g.ctr) hipFree(g.ctr);
    if (g.seg) hipFree(g.seg);
    if (g.stage) hipHostFree(g.stage);
    g = DevGeo();
}

int sfm_destroy(SfmHandle* h) {
    if (!h) return SFM_OK;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (h->stamps && exp_env("SFM_STAMPS")) {       // diagnostic: dump the last launch's per-workgroup stamps
        std::vector<unsigned long long> st((size_t)3 * PAIR_STAMP_WGS);
        if (hipMemcpy(st.data(), h->stamps, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            FILE* f = fopen(exp_env("SFM_STAMPS"), "w");
            if (f) { for (size_t b = 0; b < (size_t)PAIR_STAMP_WGS; ++b) if (st[3 * b + 1]) fprintf(f, "%llu %llu %llu\n", st[3 * b], st[3 * b + 1], st[3 * b + 2]); fclose(f); }
        }
        hipFree(h->stamps);
    }
    if (h->fused_stamps && exp_env("SFM_FUSED_STAMPS")) {   // diagnostic: the last fused launch's per-workgroup phase stamps
        std::vector<unsigned long long> st((size_t)FUSED_STAMP_STRIDE * FUSED_STAMP_WGS);
        if (hipMemcpy(st.data(), h->fused_stamps, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            FILE* f = fopen(exp_env("SFM_FUSED_STAMPS"), "w");
            if (f) {
                for (size_t b = 0; b < (size_t)FUSED_STAMP_WGS; ++b) {
                    if (!st[FUSED_STAMP_STRIDE * b]) continue;
                    fprintf(f, "%zu", b);
                    for (int k = 0; k < 5 + 32; ++k) fprintf(f, " %llu", st[FUSED_STAMP_STRIDE * b + k]);
                    fprintf(f, "\n");
                }
                fclose(f);
            }
        }
        hipFree(h->fused_stamps);
    }
    if (h->geo_stamps && exp_env("SFM_GEO_STAMPS")) {   // diagnostic: dump the last launch's per-workgroup phase stamps
        std::vector<unsigned long long> st(4 * 8192);
        if (hipMemcpy(st.data(), h->geo_stamps, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            FILE* f = fopen(exp_env("SFM_GEO_STAMPS"), "w");
            if (f) { for (size_t b = 0; b < 8192; ++b) fprintf(f, "%llu %llu %llu %llu\n", st[4 * b], st[4 * b + 1], st[4 * b + 2], st[4 * b + 3]); fclose(f); }
        }
        hipFree(h->geo_stamps);
    }
    for (int b = 0; b < 2; ++b) { if (h->pk[b]) hipFree(h->pk[b]); if (h->zv[b]) hipFree(h->zv[b]); }
    if (h->own) hipFree(h->own);
    if (h->radius) hipFree(h->radius);
    if (h->crossing) hipFree(h->crossing);
    if (h->arrived) hipFree(h->arrived);
    if (h->draws) hipFree(h->draws);
    if (h->rec) hipFree(h->rec);
    if (h->geo) hipFree(h->geo);
    if (h->dyn_local) hipFree(h->dyn_local);
    if (h->dyn_rot) hipFree(h->dyn_rot);
    if (h->slab) hipFree(h->slab);
    if (h->slabz) hipFree(h->slabz);
    if (h->pool) hipFree(h->pool);
    if (h->poolz) hipFree(h->poolz);
    if (h->pair_idx) hipFree(h->pair_idx);
    if (h->ovf) hipFree(h->ovf);
    if (h->spill) hipFree(h->spill);
    if (h->fslab) hipFree(h->fslab);
    if (h->fgeo) hipFree(h->fgeo);
    if (h->fslabz) hipFree(h->fslabz);
    if (h->dyn_ctr_alt) hipFree(h->dyn_ctr_alt);
    if (h->dyn_pts_alt) hipFree(h->dyn_pts_alt);
    if (h->own_alt) hipFree(h->own_alt);
    for (void* q : {(void*)h->f_mode, (void*)h->f_target, (void*)h->f_initial, (void*)h->f_crossing, (void*)h->f_margin,
                    (void*)h->f_next, (void*)h->f_off, (void*)h->f_cursor, (void*)h->f_xy, (void*)h->f_cross})
        if (q) hipFree(q);
    for (void* q : {(void*)h->own2, (void*)h->radius2, (void*)h->crossing2, (void*)h->draws2, (void*)h->ids2, (void*)h->sort_buf, h->sort_temp})
        if (q) hipFree(q);
    if (h->ids) hipFree(h->ids);
    if (h->tile_box) hipFree(h->tile_box);
    if (h->tile_vmax) hipFree(h->tile_vmax);
    if (h->strip_box) hipFree(h->strip_box);
    if (h->up_stage) hipHostFree(h->up_stage);
    if (h->down_stage) hipHostFree(h->down_stage);
    if (h->up_block) hipFree(h->up_block);
    if (h->strip_vmax) hipFree(h->strip_vmax);
    if (h->work) hipFree(h->work);
    if (h->work_count) hipFree(h->work_count);
    if (h->work2) hipFree(h->work2);
    free_geo(h->borders); free_geo(h->statics); free_geo(h->dynamics);
    if (h->aux) { hipStreamSynchronize(h->aux); hipStreamDestroy(h->aux); }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    delete h;
    return SFM_OK;
}

int sfm_set_params(SfmHandle* h, const SfmParams* params) {
    if (!h) return SFM_ERR_INVALID;
    const char* why = nullptr;
    if (!check_params(params, &why)) return fail(h, SFM_ERR_INVALID, why);
    drop_geo_ahead(h);
    h->carry_ok = false;
    h->prm = *params;
    return SFM_OK;
}

int sfm_set_stream(SfmHandle* h, void* hip_stream) {
    if (!h) return SFM_ERR_INVALID;
    h->carry_ok = false;
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return SFM_OK;
}

// Upload one CSR geometry set. ctr4 holds the 4 floats per polyline the kernels want.
static int set_geo(SfmHandle* h, DevGeo& g, int K, const int32_t* offsets, const float* px, const float* py,
                   const std::vector<float4>& ctr4, bool may_defer = false) {
    int rc = bind(h);
    if (rc) return rc;
    if (K < 0) return fail(h, SFM_ERR_INVALID, "negative polyline count");
    drop_geo_ahead(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (K == 0) { free_geo(g); g.lazy = g.lazy_off = false; return SFM_OK; }
    if (!offsets) return fail(h, SFM_ERR_INVALID, "offsets is NULL");
    if (offsets[0] != 0) return fail(h, SFM_ERR_INVALID, "offsets[0] must be 0");
    for (int k = 0; k < K; ++k)
        if (offsets[k + 1] < offsets[k]) return fail(h, SFM_ERR_INVALID, "offsets must be non-decreasing");
    const int P = offsets[K];
    if (P > 0 && (!px || !py)) return fail(h, SFM_ERR_INVALID, "point arrays are NULL");
    const int* off_before = g.off;
    HIP_TRY(h, dev_reserve(g.off, g.off_cap, (size_t)K + 1));
    const bool off_same = g.off == off_before && g.K == K && g.off_host.size() == (size_t)K + 1 &&
                          memcmp(g.off_host.data(), offsets, sizeof(int) * ((size_t)K + 1)) == 0;
    HIP_TRY(h, dev_reserve(g.pts, g.pts_cap, (size_t)(P > 0 ? P : 1)));
    HIP_TRY(h, dev_reserve(g.ctr, g.ctr_cap, (size_t)K));
    // [ctr | pts | off] assembled in one pinned block, copied asynchronously (the stream was drained above, so the block
    // is free to overwrite)
    const size_t b_ctr = 0, b_pts = sizeof(float4) * (size_t)K, b_off = b_pts + sizeof(float2) * (size_t)P;
    const size_t bytes = b_off + sizeof(int) * ((size_t)K + 1);
    if (bytes > g.stage_cap) {
        if (g.stage) { hipHostFree(g.stage); g.stage = nullptr; }
        const size_t want = bytes + bytes / 2 + 256;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&g.stage), want, 0));
        g.stage_cap = want;
    }
    memcpy(g.stage + b_ctr, ctr4.data(), sizeof(float4) * (size_t)K);
    float2* sp = reinterpret_cast<float2*>(g.stage + b_pts);
    for (int p = 0; p < P; ++p) sp[p] = make_float2(px[p], py[p]);
    memcpy(g.stage + b_off, offsets, sizeof(int) * ((size_t)K + 1));
    const bool off_due = !off_same || g.lazy_off;         // (a deferred block that was never spread may have carried new offsets)
    g.lazy = g.lazy_off = false;
    if (bytes <= ((size_t)64 << 10) && may_defer) {
        // the vehicles the simulator reports every tick, in front of a state upload (sfm_step_packed): the block stays staged and the
        // upload's unpack launch spreads it; anything else that reads the arrays first calls flush_lazy_geo
        g.lazy = true; g.lazy_off = off_due;
    } else if (bytes <= ((size_t)64 << 10)) {
        // a handful of polylines: one launch reads the pinned block over the bus
        HIP_TRY(h, launch_unpack_geo(g.stage, g.ctr, K, g.pts, P, g.off, off_due, h->stream));
    } else {
        HIP_TRY(h, hipMemcpyAsync(g.ctr, g.stage + b_ctr, sizeof(float4) * (size_t)K, hipMemcpyHostToDevice, h->stream));
        if (P > 0) HIP_TRY(h, hipMemcpyAsync(g.pts, g.stage + b_pts, sizeof(float2) * (size_t)P, hipMemcpyHostToDevice, h->stream));
        if (off_due) HIP_TRY(h, hipMemcpyAsync(g.off, g.stage + b_off, sizeof(int) * ((size_t)K + 1), hipMemcpyHostToDevice, h->stream));
    }
    if (!off_same) g.off_host.assign(offsets, offsets + K + 1);
    g.K = K;
    g.P = P;
    return SFM_OK;
}

// vehicles staged by sfm_set_dynamic_obstacles_packed and not yet taken along by a state upload: spread them now
static int flush_lazy_geo(SfmHandle* h) {
    DevGeo& g = h->dynamics;
    if (!g.lazy) return SFM_OK;
    HIP_TRY(h, launch_unpack_geo(g.stage, g.ctr, g.K, g.pts, g.P, g.off, g.lazy_off, h->stream));
    g.lazy = g.lazy_off = false;
    return SFM_OK;
}

int sfm_set_borders(SfmHandle* h, int K, const int32_t* offsets, const float* px, const float* py,
                    const float* cx, const float* cy, const float* cull_len) {
    if (!h) return SFM_ERR_INVALID;
    if (K > 0 && (!cx || !cy || !cull_len)) return fail(h, SFM_ERR_INVALID, "border centre/length arrays are NULL");
    std::vector<float4> c4((size_t)(K > 0 ? K : 0));
    for (int k = 0; k < K; ++k)
        c4[k] = make_float4(cx[k], cy[k], (float)((double)cull_len[k] * (double)cull_len[k]), 0.f);
    int rc = set_geo(h, h->borders, K, offsets, px, py, c4);
    if (rc || K == 0) return rc;
    // chord first->last point and the largest deviation of the polyline from it: a distance lower bound that
    // lets the kernel skip borders whose term is below 2^-40 a (rounded up so the bound stays a bound)
    std::vector<float4> seg((size_t)2 * K);
    for (int k = 0; k < K; ++k) {
        const int o0 = offsets[k], o1 = offsets[k + 1];
        double ax = 0, ay = 0, abx = 0, aby = 0, inv = 0, dev = 0;
        if (o1 > o0) {
            ax = px[o0]; ay = py[o0];
            abx = (double)px[o1 - 1] - ax; aby = (double)py[o1 - 1] - ay;
            const double ab2 = abx * abx + aby * aby;
            inv = ab2 > 0 ? 1.0 / ab2 : 0.0;
            for (int p = o0; p < o1; ++p) {
                const double qx = px[p] - ax, qy = py[p] - ay;
                double t = (qx * abx + qy * aby) * inv;
                t = t < 0 ? 0 : (t > 1 ? 1 : t);
                const double ex_ = qx - t * abx, ey_ = qy - t * aby;
                const double d = std::sqrt(ex_ * ex_ + ey_ * ey_);
                if (d > dev) dev = d;
            }
        }
        // Straight and uniformly sampled (np.linspace between two points, obstacles.py:344-355)?  Then point j sits at
        // a + (b - a) j / (P - 1); verified here point by point (2 % of the spacing), so the kernel may replace the scan
        // over all P points by the few samples around the foot of the perpendicular (geo_item).  Curved or unevenly sampled
        // polylines (the sidewalk borders of obstacles.py:72-166) fail the check and keep the full scan.
        float n_seg = 0.f;
        const int P = o1 - o0;
        if (P >= 8 && !getenv("SFM_NO_STRAIGHT")) {
            const double len = std::sqrt(abx * abx + aby * aby), spacing = len / (P - 1);
            bool ok = spacing > 1e-6;
            for (int j = 0; j < P && ok; ++j) {
                const double ix = ax + abx * j / (P - 1), iy = ay + aby * j / (P - 1);
                const double ex_ = px[o0 + j] - ix, ey_ = py[o0 + j] - iy;
                ok = std::sqrt(ex_ * ex_ + ey_ * ey_) <= 0.02 * spacing;
            }
            if (ok) n_seg = (float)(P - 1);
        }
        seg[2 * k] = make_float4((float)ax, (float)ay, (float)abx, (float)aby);
        seg[2 * k + 1] = make_float4((float)inv, (float)(dev * 1.0001 + 1e-4), n_seg, 0.f);
    }
    HIP_TRY(h, dev_realloc(h->borders.seg, (size_t)2 * K));
    HIP_TRY(h, hipMemcpy(h->borders.seg, seg.data(), sizeof(float4) * (size_t)2 * K, hipMemcpyHostToDevice));
    return SFM_OK;
}

int sfm_set_static_obstacles(SfmHandle* h, int M, const int32_t* offsets, const float* px, const float* py,
                             const float* cx, const float* cy) {
    if (!h) return SFM_ERR_INVALID;
    if (M > 0 && (!cx || !cy)) return fail(h, SFM_ERR_INVALID, "obstacle centre arrays are NULL");
    std::vector<float4> c4((size_t)(M > 0 ? M : 0));
    for (int k = 0; k < M; ++k) c4[k] = make_float4(cx[k], cy[k], 0.f, 0.f);
    return set_geo(h, h->statics, M, offsets, px, py, c4);
}

int sfm_set_dynamic_obstacles(SfmHandle* h, int M, const int32_t* offsets, const float* px, const float* py,
                              const float* cx, const float* cy, const float* vx, const float* vy) {
    if (!h) return SFM_ERR_INVALID;
    if (M > 0 && (!cx || !cy)) return fail(h, SFM_ERR_INVALID, "obstacle centre arrays are NULL");
    std::vector<float4> c4((size_t)(M > 0 ? M : 0));
    for (int k = 0; k < M; ++k)       // velocities default to 0 like ObstacleForce (forces.py:212-213)
        c4[k] = make_float4(cx[k], cy[k], vx ? vx[k] : 0.f, vy ? vy[k] : 0.f);
    h->dyn_boxes = false;
    return set_geo(h, h->dynamics, M, offsets, px, py, c4);
}

// The same from packed arrays (ABI 4, additions only): pts [P][2] {x, y}, cv [M][4] {cx, cy, vx, vy} -- what a caller that receives the
// vehicles every tick (run_simulation.py:95) can fill with two array writes.
int sfm_set_dynamic_obstacles_packed(SfmHandle* h, int M, const int32_t* offsets, const float* pts, const float* cv) {
    if (!h) return SFM_ERR_INVALID;
    if (M > 0 && (!offsets || !cv)) return fail(h, SFM_ERR_INVALID, "offsets / cv is NULL");
    const int P = M > 0 ? offsets[M] : 0;
    if (P > 0 && !pts) return fail(h, SFM_ERR_INVALID, "pts is NULL");
    std::vector<float>& c = h->step_cols;          // (scratch shared with sfm_step_packed: both run on the caller's thread, one after the other)
    c.resize((size_t)2 * (size_t)(P > 0 ? P : 0) + 1);
    float *px = c.data(), *py = px + (P > 0 ? P : 0);
    for (int p = 0; p < P; ++p) { px[p] = pts[2 * p]; py[p] = pts[2 * p + 1]; }
    std::vector<float4> c4((size_t)(M > 0 ? M : 0));
    for (int k = 0; k < M; ++k) c4[k] = make_float4(cv[4 * k], cv[4 * k + 1], cv[4 * k + 2], cv[4 * k + 3]);
    h->dyn_boxes = false;
    return set_geo(h, h->dynamics, M, offsets, px, py, c4, true);
}

int sfm_set_dynamic_boxes(SfmHandle* h, int M, const int32_t* offsets, const float* ux, const float* uy,
                          const float* cx, const float* cy, const float* yaw_cos, const float* yaw_sin,
                          const float* vx, const float* vy) {
    if (!h) return SFM_ERR_INVALID;
    if (M > 0 && (!cx || !cy || !yaw_cos || !yaw_sin)) return fail(h, SFM_ERR_INVALID, "box centre / yaw arrays are NULL");
    std::vector<float4> c4((size_t)(M > 0 ? M : 0));
    for (int k = 0; k < M; ++k) c4[k] = make_float4(cx[k], cy[k], vx ? vx[k] : 0.f, vy ? vy[k] : 0.f);
    h->dyn_boxes = false;
    int rc = set_geo(h, h->dynamics, M, offsets, ux, uy, c4);     // pts temporarily holds the local offsets
    if (rc || M == 0) return rc;
    const int P = h->dynamics.P;
    std::vector<float2> rot((size_t)M);
    for (int k = 0; k < M; ++k) rot[k] = make_float2(yaw_cos[k], yaw_sin[k]);
    HIP_TRY(h, dev_realloc(h->dyn_local, (size_t)(P > 0 ? P : 1)));
    HIP_TRY(h, dev_realloc(h->dyn_rot, (size_t)M));
    if (P > 0) HIP_TRY(h, hipMemcpyAsync(h->dyn_local, h->dynamics.pts, sizeof(float2) * (size_t)P, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->dyn_rot, rot.data(), sizeof(float2) * (size_t)M, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));       // `rot` is a local
    HIP_TRY(h, launch_dynamic_boxes(h->dynamics.ctr, h->dynamics.off, h->dyn_local, h->dyn_rot, h->dynamics.pts, M,
                                    h->prm.step_length, 0, h->stream));
    h->dyn_boxes = true;
    return SFM_OK;
}

int sfm_download_dynamic_obstacles(SfmHandle* h, float* cx, float* cy, float* px, float* py) {
    int rc = bind(h);
    if (rc) return rc;
    rc = flush_lazy_geo(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int M = h->dynamics.K, P = h->dynamics.P;
    if (M == 0) return SFM_OK;
    std::vector<float4> c((size_t)M);
    std::vector<float2> p((size_t)(P > 0 ? P : 1));
    HIP_TRY(h, hipMemcpy(c.data(), h->dynamics.ctr, sizeof(float4) * (size_t)M, hipMemcpyDeviceToHost));
    if (P > 0) HIP_TRY(h, hipMemcpy(p.data(), h->dynamics.pts, sizeof(float2) * (size_t)P, hipMemcpyDeviceToHost));
    for (int k = 0; k < M; ++k) { if (cx) cx[k] = c[k].x; if (cy) cy[k] = c[k].y; }
    for (int q = 0; q < P; ++q) { if (px) px[q] = p[q].x; if (py) py[q] = p[q].y; }
    return SFM_OK;
}

// Row length of the slab in float2: one tile more than the n_t * 64 columns, so that the same column of consecutive rows
// (what an epilogue workgroup sums, what a run of list items writes) does not sit at a power-of-two stride -- those all
// land on one HBM channel.
static inline int slab_stride(int n_t) { return n_t * WAVE + WAVE; }

// The packing plan of this handle: one block = plain strip packing, or the gx x gy blocks of sfm_set_partition with their row
// bounds (equal split of the padded row count when none were given) and a strip size per block that makes its tiles square.
static BlockPlan block_plan(const SfmHandle* h, int N, int n_pad) {
    BlockPlan pl{};
    const int G = h->part_gx * h->part_gy;
    if (G <= 1 || G > MAX_BLOCKS) {
        pl.n_blocks = 1; pl.gy = 1; pl.bound[0] = 0; pl.bound[1] = n_pad; pl.strip_rows[0] = h->strip_rows;
        return pl;
    }
    pl.n_blocks = G;
    pl.gy = h->part_gy;
    for (int b = 0; b <= G; ++b)
        pl.bound[b] = (int)h->part_bounds.size() == G + 1 ? h->part_bounds[b] : (int)((long long)n_pad * b / G / WAVE * WAVE);
    pl.bound[0] = 0;
    pl.bound[G] = std::max(pl.bound[G], n_pad);
    // a block covers 1/gx of the extent in x and about 1/gy in y
    const double aspect = std::fmin(std::fmax(h->pack_aspect * h->part_gy / h->part_gx, 1.0 / 64.0), 64.0);
    for (int b = 0; b < G; ++b) {
        const int tiles = std::max(1, (std::min(N, pl.bound[b + 1]) - std::min(N, pl.bound[b]) + WAVE - 1) / WAVE);
        const int n_strips = std::max(1, std::min(tiles, (int)std::lround(std::sqrt((double)tiles * aspect))));
        pl.strip_rows[b] = WAVE * ((tiles + n_strips - 1) / n_strips);
    }
    return pl;
}

int sfm_upload_state(SfmHandle* h, int N, const float* x, const float* y, const float* z, const float* vx,
                     const float* vy, const float* vz, const float* wx, const float* wy,
                     const float* target_speed, const float* radius, const uint8_t* crossing_mask) {
    int rc = bind(h);
    if (rc) return rc;
    if (N < 0) return fail(h, SFM_ERR_INVALID, "N < 0");
    if (N > 0 && (!x || !y || !vx || !vy || !wx || !wy || !target_speed))
        return fail(h, SFM_ERR_INVALID, "a required state array is NULL");
    if ((z == nullptr) != (vz == nullptr)) return fail(h, SFM_ERR_INVALID, "z and vz must be given together");
    if (h->prm.use_ped_radius && N > 0 && !radius) return fail(h, SFM_ERR_INVALID, "use_ped_radius needs radius");
    drop_geo_ahead(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int n_pad = ((N + TILE_J - 1) / TILE_J) * TILE_J;
    const bool z3 = (z != nullptr);
    const bool rad = h->prm.use_ped_radius != 0;
    if (n_pad > h->cap) {
        for (int b = 0; b < 2; ++b) HIP_TRY(h, dev_realloc(h->pk[b], (size_t)n_pad));
        for (int b = 0; b < 2; ++b) HIP_TRY(h, dev_realloc(h->zv[b], (size_t)n_pad));
        HIP_TRY(h, dev_realloc(h->own, (size_t)n_pad));
        HIP_TRY(h, dev_realloc(h->own_alt, 0));      // (the fused tick's twin of own: re-made at the new size when next needed)
        h->own_alt_cap = 0;
        HIP_TRY(h, dev_realloc(h->radius, (size_t)n_pad));
        HIP_TRY(h, dev_realloc(h->crossing, (size_t)n_pad));
        HIP_TRY(h, dev_realloc(h->arrived, (size_t)n_pad));
        HIP_TRY(h, dev_realloc(h->draws, (size_t)n_pad));
        HIP_TRY(h, dev_realloc(h->ids, (size_t)n_pad));
        HIP_TRY(h, dev_realloc(h->rec, (size_t)n_pad * 18));
        HIP_TRY(h, dev_realloc(h->geo, (size_t)n_pad * 6 * GEO_SLICES_MAX));
        h->cap = n_pad;
    }
    h->N = N; h->N_pad = n_pad; h->z3 = z3; h->rad = rad;
    h->i_begin = 0; h->i_end = N; h->cur = 0; h->rec_valid = false; h->timing_valid = false;
    h->fsm_on = false;                     // a new crowd: the caller sets the FSM again if it wants it
    if (N == 0) return SFM_OK;
    // The rows are assembled in one pinned host block [pk | own | zv | radius | crossing], copied asynchronously in one piece
    // and spread over the device arrays by one small kernel (a host-in-the-loop caller uploads every tick: no pageable
    // staging, no allocation, no wait at the end; the stream was drained above, so the block is free to overwrite).
    const size_t np = (size_t)n_pad;
    const size_t b_pk = 0, b_own = b_pk + sizeof(float4) * np, b_zv = b_own + sizeof(float4) * np,
                 b_rr = b_zv + sizeof(float2) * np, b_cm = b_rr + sizeof(float) * np, b_end = b_cm + np;
    if (b_end > h->up_stage_cap) {
        if (h->up_stage) { hipHostFree(h->up_stage); h->up_stage = nullptr; }
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->up_stage), b_end + b_end / 2, 0));
        h->up_stage_cap = b_end + b_end / 2;
    }
    memset(h->up_stage, 0, b_end);
    float4* pk = reinterpret_cast<float4*>(h->up_stage + b_pk);
    float4* own = reinterpret_cast<float4*>(h->up_stage + b_own);
    float2* zv = reinterpret_cast<float2*>(h->up_stage + b_zv);
    float* rr = reinterpret_cast<float*>(h->up_stage + b_rr);
    uint8_t* cm = reinterpret_cast<uint8_t*>(h->up_stage + b_cm);
    // padding rows are ghost pedestrians parked far away at distinct positions: every interaction with them
    // underflows to exactly 0 (exp2 of ~ -1e16), so the symmetric kernel needs no masking
    for (int i = N; i < n_pad; ++i) pk[i] = make_float4(3.0e15f + 1.0e12f * (float)(i - N + 1), 3.0e15f, 0.f, 0.f);
    // spatial order (sfm_reorder.hip): sort by x, cut into strips of strip_rows rows, sort each strip by y, so every
    // 64-row tile is the content of one axis-aligned rectangle.  Stable sorts on order-preserving integer keys: the
    // order is a pure function of the uploaded state, identical on every rank.
    h->reordered = (h->reorder_mode == 1 || (h->reorder_mode < 0 && N >= 2048));
    h->perm.resize((size_t)N);
    std::iota(h->perm.begin(), h->perm.end(), 0u);
    if (h->reordered) {
        auto float_key = [](float v) { uint32_t b; memcpy(&b, &v, sizeof(b)); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); };
        // strips: about sqrt(#tiles) of them, corrected for the aspect of the crowd's extent so that tiles come out square
        float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
        for (int i = 0; i < N; ++i) {
            if (std::fabs(x[i]) < 1.0e12f && std::fabs(y[i]) < 1.0e12f) {
                x0 = std::fmin(x0, x[i]); x1 = std::fmax(x1, x[i]); y0 = std::fmin(y0, y[i]); y1 = std::fmax(y1, y[i]);
            }
        }
        const int n_tiles = (N + WAVE - 1) / WAVE;
        double aspect = (x1 > x0 && y1 > y0) ? (double)(x1 - x0) / (double)(y1 - y0) : 1.0;
        aspect = std::fmin(std::fmax(aspect, 1.0 / 64.0), 64.0);
        h->pack_aspect = aspect;
        const int n_strips = std::max(1, std::min(n_tiles, (int)std::lround(std::sqrt((double)n_tiles * aspect))));
        h->strip_rows = WAVE * ((n_tiles + n_strips - 1) / n_strips);
        std::vector<uint32_t> kx((size_t)N), ky((size_t)N);
        for (int i = 0; i < N; ++i) { kx[i] = float_key(x[i]); ky[i] = float_key(y[i]); }
        auto by_x = [&](uint32_t a_, uint32_t b_) { return kx[a_] < kx[b_]; };
        auto by_y = [&](uint32_t a_, uint32_t b_) { return ky[a_] < ky[b_]; };
        auto strip_pack = [&](int r0, int r1, int strip_rows) {         // rows [r0, r1) of perm: sort by x, strips, each by y
            std::stable_sort(h->perm.begin() + r0, h->perm.begin() + r1, by_x);
            for (int q = r0; q < r1; q += strip_rows) std::stable_sort(h->perm.begin() + q, h->perm.begin() + std::min(r1, q + strip_rows), by_y);
        };
        const BlockPlan pl = block_plan(h, N, n_pad);
        if (pl.n_blocks <= 1) {
            strip_pack(0, N, h->strip_rows);
        } else {                                                          // block-major (sfm_set_partition): the device re-pack's four sorts
            std::stable_sort(h->perm.begin(), h->perm.end(), by_x);
            const int gx = pl.n_blocks / pl.gy;
            for (int c = 0; c < gx; ++c) {
                const int c0 = std::min(N, pl.bound[c * pl.gy]), c1 = std::min(N, pl.bound[(c + 1) * pl.gy]);
                std::stable_sort(h->perm.begin() + c0, h->perm.begin() + c1, by_y);
            }
            for (int b = 0; b < pl.n_blocks; ++b) strip_pack(std::min(N, pl.bound[b]), std::min(N, pl.bound[b + 1]), pl.strip_rows[b]);
        }
    }
    for (int s_ = 0; s_ < N; ++s_) {
        const int i = (int)h->perm[s_];
        pk[s_] = make_float4(x[i], y[i], vx[i], vy[i]);
        own[s_] = make_float4(wx[i], wy[i], target_speed[i], radius ? radius[i] : 0.f);
        if (z3) zv[s_] = make_float2(z[i], vz[i]);
        if (radius) rr[s_] = radius[i];
        if (crossing_mask) cm[s_] = crossing_mask[i] ? 1 : 0;
    }
    if (b_end <= ((size_t)256 << 10)) {
        // a host-in-the-loop crowd (round 4): the unpack kernel reads the pinned block over the bus by itself -- one launch instead of
        // a copy and a launch in front of every tick (the block is not overwritten before the next upload's stream synchronisation)
        DevGeo& g = h->dynamics;
        HIP_TRY(h, launch_unpack_rows(h->up_stage, b_own, b_zv, b_rr, b_cm, n_pad, h->pk[0], h->pk[1], h->own, h->zv[0], h->zv[1],
                                      h->radius, h->crossing, h->draws, g.lazy ? g.stage : nullptr, g.ctr, g.K, g.pts, g.P, g.off,
                                      g.lazy_off, h->stream));
        g.lazy = g.lazy_off = false;
    } else {
        HIP_TRY(h, dev_reserve(h->up_block, h->up_block_cap, b_end));
        HIP_TRY(h, hipMemcpyAsync(h->up_block, h->up_stage, b_end, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, launch_unpack_rows(h->up_block, b_own, b_zv, b_rr, b_cm, n_pad, h->pk[0], h->pk[1], h->own, h->zv[0], h->zv[1],
                                      h->radius, h->crossing, h->draws, nullptr, nullptr, 0, nullptr, 0, nullptr, false, h->stream));
    }
    if (h->reordered) {
        HIP_TRY(h, hipMemcpyAsync(h->ids, h->perm.data(), sizeof(uint32_t) * (size_t)N, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));          // perm is pageable and may be resized by the next upload
        if (n_pad > h->sort_cap) {
            HIP_TRY(h, dev_realloc(h->own2, (size_t)n_pad)); HIP_TRY(h, dev_realloc(h->radius2, (size_t)n_pad));
            HIP_TRY(h, dev_realloc(h->crossing2, (size_t)n_pad)); HIP_TRY(h, dev_realloc(h->draws2, (size_t)n_pad));
            HIP_TRY(h, dev_realloc(h->ids2, (size_t)n_pad)); HIP_TRY(h, dev_realloc(h->sort_buf, (size_t)n_pad * 8));
            h->sort_temp_bytes = reorder_temp_bytes(n_pad);
            if (h->sort_temp) { hipFree(h->sort_temp); h->sort_temp = nullptr; }
            HIP_TRY(h, hipMalloc(&h->sort_temp, h->sort_temp_bytes > 0 ? h->sort_temp_bytes : 16));
            h->sort_cap = n_pad;
        }
        HIP_TRY(h, hipMemsetAsync(h->own2, 0, sizeof(float4) * np, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->radius2, 0, sizeof(float) * np, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->crossing2, 0, np, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->draws2, 0, sizeof(uint32_t) * np, h->stream));
    }
    h->ticks_since_sort = 0;
    h->perm_stale = false;
    // (the symmetric path's partial forces -- dense slab without the list cutoff, row pool with it -- are reserved by the first tick
    //  that needs them: sym_reserve)
    h->n_t = (N + WAVE - 1) / WAVE;
    // cutoff bookkeeping: per-tile box / max speed, and the work list of the symmetric kernel
    h->r_max = 0.f;
    if (rad) for (int i = 0; i < N; ++i) h->r_max = std::fmax(h->r_max, radius[i]);
    if ((size_t)h->n_t > h->box_cap) {
        const size_t want = (size_t)h->n_t + (size_t)h->n_t / 2 + 4;
        HIP_TRY(h, dev_realloc(h->tile_box, want * 2));
        HIP_TRY(h, dev_realloc(h->tile_vmax, want * 2));
        HIP_TRY(h, dev_realloc(h->strip_box, want));
        HIP_TRY(h, dev_realloc(h->strip_vmax, want));
        h->box_cap = want;
    }
    h->box_cur = 0;
    h->boxes_valid = false;
    {
        const size_t items = (size_t)h->n_t * (size_t)(h->n_t / 2 + 1);
        if (h->n_t < 65536 && items > h->work_cap) { HIP_TRY(h, dev_realloc(h->work, items)); h->work_cap = items; }
        if (!h->work_count) HIP_TRY(h, dev_realloc(h->work_count, (size_t)4));    // [0] list, [1] own-own list of a split tick, [2] copy of [0] left by a zeroing epilogue
        h->begin_done = false;
    }
    return SFM_OK;
}

int sfm_set_mode_fsm(SfmHandle* h, int N, const uint8_t* mode, const float* target_speed, const float* initial_speed,
                     const float* crossing_speed, const float* safety_margin, const float* next_mode_time,
                     const int32_t* wp_offsets, const float* wp_x, const float* wp_y, const uint8_t* wp_crossing,
                     int despawn_on_arrival, float sim_time0, const float* first_vehicle_extent) {
    int rc = bind(h);
    if (rc) return rc;
    drop_geo_ahead(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (N == 0) { h->fsm_on = false; return SFM_OK; }
    if (N != h->N) return fail(h, SFM_ERR_STATE, "sfm_set_mode_fsm: N differs from the uploaded state");
    if (!mode || !target_speed || !initial_speed || !crossing_speed || !safety_margin || !next_mode_time || !wp_offsets)
        return fail(h, SFM_ERR_INVALID, "a required FSM array is NULL");
    if (wp_offsets[0] != 0) return fail(h, SFM_ERR_INVALID, "wp_offsets[0] must be 0");
    for (int i = 0; i < N; ++i) {
        if (wp_offsets[i + 1] < wp_offsets[i]) return fail(h, SFM_ERR_INVALID, "wp_offsets must be non-decreasing");
        if (mode[i] > 4) return fail(h, SFM_ERR_INVALID, "mode must be a PedMode value 0..4");
    }
    const int W = wp_offsets[N];
    if (W > 0 && (!wp_x || !wp_y || !wp_crossing)) return fail(h, SFM_ERR_INVALID, "waypoint arrays are NULL");
    std::vector<float2> xy((size_t)(W > 0 ? W : 1));
    for (int e = 0; e < W; ++e) xy[e] = make_float2(wp_x[e], wp_y[e]);
    const size_t n = (size_t)N;
    HIP_TRY(h, dev_realloc(h->f_mode, n)); HIP_TRY(h, dev_realloc(h->f_target, n)); HIP_TRY(h, dev_realloc(h->f_initial, n));
    HIP_TRY(h, dev_realloc(h->f_crossing, n)); HIP_TRY(h, dev_realloc(h->f_margin, n)); HIP_TRY(h, dev_realloc(h->f_next, n));
    HIP_TRY(h, dev_realloc(h->f_off, n + 1)); HIP_TRY(h, dev_realloc(h->f_cursor, n));
    HIP_TRY(h, dev_realloc(h->f_xy, xy.size())); HIP_TRY(h, dev_realloc(h->f_cross, xy.size()));
    HIP_TRY(h, hipMemcpy(h->f_mode, mode, n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->f_target, target_speed, 4 * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->f_initial, initial_speed, 4 * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->f_crossing, crossing_speed, 4 * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->f_margin, safety_margin, 4 * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->f_next, next_mode_time, 4 * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->f_off, wp_offsets, 4 * (n + 1), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemset(h->f_cursor, 0, 4 * n));
    if (W > 0) {
        HIP_TRY(h, hipMemcpy(h->f_xy, xy.data(), sizeof(float2) * (size_t)W, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->f_cross, wp_crossing, (size_t)W, hipMemcpyHostToDevice));
    }
    h->fsm_n = N;
    h->despawn = despawn_on_arrival ? 1 : 0;
    h->sim_time = sim_time0;
    h->veh_ext[0] = first_vehicle_extent ? first_vehicle_extent[0] : 0.f;
    h->veh_ext[1] = first_vehicle_extent ? first_vehicle_extent[1] : 0.f;
    h->fsm_on = true;
    return SFM_OK;
}

int sfm_download_modes(SfmHandle* h, uint8_t* mode, float* target_speed, int32_t* cursor) {
    int rc = bind(h);
    if (rc) return rc;
    if (!h->fsm_on) return fail(h, SFM_ERR_STATE, "sfm_set_mode_fsm has not been called");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->fsm_n;
    if (mode) HIP_TRY(h, hipMemcpy(mode, h->f_mode, n, hipMemcpyDeviceToHost));
    if (target_speed) HIP_TRY(h, hipMemcpy(target_speed, h->f_target, 4 * n, hipMemcpyDeviceToHost));
    if (cursor) HIP_TRY(h, hipMemcpy(cursor, h->f_cursor, 4 * n, hipMemcpyDeviceToHost));
    return SFM_OK;
}

int sfm_set_shard(SfmHandle* h, int i_begin, int i_end) {
    if (!h) return SFM_ERR_INVALID;
    if (i_begin < 0 || i_end < i_begin || i_end > h->N) return fail(h, SFM_ERR_INVALID, "shard out of range");
    drop_geo_ahead(h);
    h->carry_ok = false;
    h->i_begin = i_begin;
    h->i_end = i_end;
    return SFM_OK;
}

int sfm_set_partition(SfmHandle* h, int gx, int gy, const int32_t* bounds) {
    if (!h) return SFM_ERR_INVALID;
    h->carry_ok = false;
    if (gx < 0 || gy < 0 || (gx == 0) != (gy == 0) || gx * gy > MAX_BLOCKS) return fail(h, SFM_ERR_INVALID, "partition must be gx x gy blocks, at most 16");
    const int G = gx * gy;
    if (bounds) {
        if (bounds[0] != 0) return fail(h, SFM_ERR_INVALID, "bounds[0] must be 0");
        for (int b = 0; b < G; ++b)
            if (bounds[b + 1] < bounds[b] || bounds[b] % WAVE) return fail(h, SFM_ERR_INVALID, "bounds must be non-decreasing multiples of 64");
    }
    h->part_gx = gx;
    h->part_gy = gy;
    h->part_bounds.assign(bounds ? bounds : nullptr, bounds ? bounds + G + 1 : nullptr);
    return SFM_OK;
}

int sfm_set_waypoint_stream(SfmHandle* h, uint32_t seed, float world_side, float arrive_threshold) {
    if (!h) return SFM_ERR_INVALID;
    h->carry_ok = false;
    h->seed = seed;
    h->world_side = world_side;
    h->arrive_thr = arrive_threshold;
    return SFM_OK;
}

// Launch shape: rows per wave (IPW) and waves sharing a row set (TEAM).  The pair body is one long dependent
// chain, so a SIMD needs ~16 independent chains (waves x IPW) to issue every cycle; small shards get there by
// splitting the j range over the 4 waves of a workgroup (TEAM 4), large ones by more rows per wave.
static void pick_shape(const SfmHandle* h, int n_local, int* ipw, int* team) {
    int t = (n_local < 16384) ? 4 : 1;
    if (h->team_override == 1 || h->team_override == 4) t = h->team_override;
    int w;
    if (t == 4) w = (n_local >= 2048) ? 4 : (n_local >= 1024 ? 2 : 1);
    else w = (n_local >= 4 * 2048) ? 4 : (n_local >= 2 * 2048 ? 2 : 1);
    if (h->ipw_override == 1 || h->ipw_override == 2 || h->ipw_override == 4 || h->ipw_override == 8)
        w = h->ipw_override;
    *ipw = w;
    *team = t;
}

// Above this many pedestrians the tile-pair list cutoff is on by default (and with it the two- / three-launch tick); up to it a
// device-resident run is one launch per tick on the fused kernel, whose 512 workgroups of 16 waves are exactly one residency round
// at N = 4096 -- one more group and a second round starts.  Measured (tools/threshold_probe.py, 0.25 ped/m2, us per tick, fused /
// list cutoff): N = 4096 16.3 / 24.1, 5120 29.2 / 27.6, 6144 37.1 / 31.6, 8000 56.5 / 40.7; with border / obstacle forces 21.5 / 25.5,
// 31.3 / 25.5, 41.3 / 27.9, 59.9 / 28.2.  (Rounds 1-2 switched at 8192.)
constexpr int AUTO_CUTOFF_N = 4096;

static void fill_args(SfmHandle* h, TickArgs& a, uint32_t flags) {
    const SfmParams& p = h->prm;
    memset(&a, 0, sizeof(a));
    a.pk_cur = h->pk[h->cur];
    a.pk_next = h->pk[h->cur ^ 1];
    a.zv_cur = h->zv[h->cur];
    a.zv_next = h->zv[h->cur ^ 1];
    a.own = h->own;
    a.radius = h->radius;
    a.crossing = h->crossing;
    a.draws = h->draws;
    a.ids = h->reordered ? h->ids : nullptr;
    a.rec = (flags & SFM_TICK_RECORD_FORCES) ? h->rec : nullptr;
    a.host_pk = h->hosted ? reinterpret_cast<float4*>(h->down_stage) : nullptr;
    a.host_zv = h->hosted ? reinterpret_cast<float2*>(h->down_stage + sizeof(float4) * (size_t)h->N_pad) : nullptr;
    a.N = h->N; a.N_pad = h->N_pad; a.i_begin = h->i_begin; a.i_end = h->i_end;
    const bool any_geo = (p.enabled[SFM_FORCE_BORDER] && h->borders.K > 0) || (p.enabled[SFM_FORCE_STATIC_OBSTACLE] && h->statics.K > 0) ||
                         (p.enabled[SFM_FORCE_DYNAMIC_OBSTACLE] && h->dynamics.K > 0);
    a.geo = any_geo ? h->geo : nullptr;
    {   // a handful of tiles cannot fill 256 CUs: split each tile's polylines over up to 8 workgroups
        const int tiles = std::max(1, (h->i_end + WAVE - 1) / WAVE - h->i_begin / WAVE);
        // (round 2: with the straight-border shortcut a tile's scan is short, and every slice repeats the start-up chain; c1, one
        //  tile, with the polylines of all kinds dealt evenly over the slices: 18.3 us per tick at 1 slice, 15.8 at 2, 14.3 at 3,
        //  13.9 at 4, 14.4 at 6, 14.5 at 8)
        a.geo_slices = tiles <= 64 ? 4 : tiles <= 128 ? 2 : 1;
        if (h->geo_slices_override > 0) a.geo_slices = h->geo_slices_override;
    }
    // device-side vehicles move on at the end of an integrating tick: extra workgroups of the tick's last kernel
    a.adv = DynAdvance{h->dynamics.ctr, h->dynamics.off, h->dyn_local, h->dyn_rot, h->dynamics.pts,
                       (h->dyn_boxes && (flags & SFM_TICK_INTEGRATE)) ? h->dynamics.K : 0, h->prm.step_length, 0};
    // (up to 4096 tiles the ordered kernel can use the boxes too; beyond, only the symmetric path's list does -- items carry 15-bit shifts)
    const bool cut = p.enabled[SFM_FORCE_PEDESTRIAN] && h->tile_box && (h->N <= 64 * 64 * WAVE || (h->dpp_dir == 1 && h->sym_mode != 0 && h->n_t < 32768)) &&
                     (h->cut_mode == 1 || (h->cut_mode < 0 && h->N > AUTO_CUTOFF_N)) && p.pedestrian.gamma > 0.f && p.pedestrian.lambda >= 0.f;
    a.tile_box = cut ? h->tile_box + (size_t)h->box_cur * h->n_t : nullptr;
    a.tile_vmax = cut ? h->tile_vmax + (size_t)h->box_cur * h->n_t : nullptr;
    // whole crowd on the symmetric path: the epilogue leaves the next tick's boxes and a zeroed list counter, so the next tick starts
    // with its list kernel instead of a bounds kernel and a memset (the boxes do not depend on the radii -- cut_pad carries 2 r_max
    // into every test -- and a 3-D crowd's largest speed includes v_z)
    // (a cutoff for small crowds -- workgroups testing their own tile pair, a cost-balanced deal of the items -- was built and
    //  measured in round 2: on c2 it cost as much in boxes, dealer and re-packs as it saved in steps.  Removed in round 3)
    const bool carry = cut && h->dpp_dir == 1 && h->i_begin == 0 && h->i_end == h->N && h->sym_mode != 0 && h->carry_mode != 0 && !h->fsm_on;
    a.tile_box_out = carry ? h->tile_box + (size_t)(h->box_cur ^ 1) * h->n_t : nullptr;
    a.tile_vmax_out = carry ? h->tile_vmax + (size_t)(h->box_cur ^ 1) * h->n_t : nullptr;
    a.cut_scale = (float)((double)p.pedestrian.gamma * 41.0 * 0.6931471805599453 * 1.001);
    a.cut_pad = h->rad ? 2.0f * h->r_max * 1.001f : 0.f;
    if (h->fsm_on)
        a.fsm = FsmArgs{h->f_mode, h->f_target, h->f_initial, h->f_crossing, h->f_margin, h->f_next, h->f_off, h->f_xy, h->f_cross,
                        h->f_cursor, h->sim_time, h->veh_ext[0], h->veh_ext[1], h->despawn};
    a.flags = flags;
    a.en_acc = p.enabled[SFM_FORCE_ACCELERATION];
    a.en_ped = p.enabled[SFM_FORCE_PEDESTRIAN];
    a.en_border = p.enabled[SFM_FORCE_BORDER];
    a.en_static = p.enabled[SFM_FORCE_STATIC_OBSTACLE];
    a.en_dynamic = p.enabled[SFM_FORCE_DYNAMIC_OBSTACLE];
    a.ped = fold(p.pedestrian);
    a.stat = fold(p.static_obstacle);
    a.dyn = fold(p.dynamic_obstacle);
    a.border_a = p.border_a;
    a.border_nlb = (float)(-1.4426950408889634 / (double)p.border_b);
    a.border_skip = (p.border_b > 0.f && p.border_a != 0.f) ? (float)(40.0 * 0.6931471805599453 * (double)p.border_b) : 0.f;
    a.inv_tau = (float)(1.0 / (double)p.tau);
    a.dt = p.step_length;
    a.max_speed_factor = p.max_speed_factor;
    a.seed = h->seed;
    a.world_side = h->world_side;
    a.arrive_thr2 = (float)((double)h->arrive_thr * (double)h->arrive_thr);
    a.geo_stamps = h->geo_stamps;
    a.borders = Geo{h->borders.off, h->borders.pts, h->borders.ctr, h->borders.seg, h->borders.K};
    a.statics = Geo{h->statics.off, h->statics.pts, h->statics.ctr, nullptr, h->statics.K};
    a.dynamics = Geo{h->dynamics.off, h->dynamics.pts, h->dynamics.ctr, nullptr, h->dynamics.K};
}

// Re-packs the rows spatially (strips in x, sorted by y: sfm_reorder.hip).  Whole-crowd handles only: a
// shard's per-row data (waypoints, draw counters) of rows it does not own is not kept current.
static int resort_rows(SfmHandle* h) {
    drop_geo_ahead(h);
    const int N = h->N, np_ = h->N_pad;
    uint32_t* w = h->sort_buf;                     // 8 N_pad words: two 64-bit key arrays, two row arrays, two 32-bit key arrays
    ReorderBufs b{reinterpret_cast<unsigned long long*>(w), reinterpret_cast<unsigned long long*>(w + 2 * (size_t)np_),
                  w + 4 * (size_t)np_, w + 5 * (size_t)np_, w + 6 * (size_t)np_, w + 7 * (size_t)np_, h->sort_temp, h->sort_temp_bytes};
    const BlockPlan pl = block_plan(h, N, np_);
    if (pl.n_blocks > 1) HIP_TRY(h, launch_resort_blocks(h->pk[h->cur], N, pl, b, h->stream));
    else HIP_TRY(h, launch_resort(h->pk[h->cur], N, h->strip_rows, b, h->stream));
    HIP_TRY(h, launch_gather(b.row_b, N, h->pk[h->cur], h->pk[h->cur ^ 1], h->z3 ? h->zv[h->cur] : nullptr, h->zv[h->cur ^ 1],
                             h->own, h->own2, h->radius, h->radius2, h->crossing, h->crossing2, h->draws, h->draws2, h->ids,
                             h->ids2, h->stream));
    h->cur ^= 1;
    std::swap(h->own, h->own2); std::swap(h->radius, h->radius2); std::swap(h->crossing, h->crossing2);
    std::swap(h->draws, h->draws2); std::swap(h->ids, h->ids2);
    h->perm_stale = true;
    h->ticks_since_sort = 0;
    h->boxes_valid = false;
    return SFM_OK;
}

// downloads translate rows to the caller's index: fetch the row -> id map if a device re-sort changed it
static int sync_perm(SfmHandle* h) {
    if (!h->perm_stale) return SFM_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(h->perm.data(), h->ids, sizeof(uint32_t) * (size_t)h->N, hipMemcpyDeviceToHost));
    h->perm_stale = false;
    return SFM_OK;
}

// two-level tile-pair list: runs of tps tiles = the x-strips of the spatial packing; worth its extra launch from ~2000 tiles on
static void strip_shape(const SfmHandle* h, int* tps, int* n_strips) {
    *tps = std::max(1, h->strip_rows / WAVE);
    bool on = h->reordered && h->n_t >= 2048 && *tps >= 8;
    if (h->strips_override >= 0) on = h->strips_override != 0;       // SFM_STRIPS: tests force either way
    *n_strips = on ? (h->n_t + *tps - 1) / *tps : 0;
}

// Memory of the symmetric path's partial forces for the shape the next tick has.  Without the tile-pair list (grid mode: crowds up to
// ~8000 pedestrians, or SFM_CUTOFF=0) the dense slab n_t x (n_t * 64) float2, up to 16 GiB.  With the list (round 4) a pool of row pairs:
// 80 per own tile of a whole crowd (c5 keeps 43), 144 per own tile of a shard (pairs with other ranks' tiles are listed one-sided by
// both ranks), never more than every pair there is; the split tick's own-own list gets as many again behind them.  c5: 0.34 GB instead
// of 8.6.  More kept pairs than that (tiles that overlap: a crowd uploaded in random order with SFM_REORDER=0) spill into the overflow
// sums -- slower, never wrong.  false: does not fit (the caller falls back to the ordered kernel).
static bool sym_reserve(SfmHandle* h, bool list, bool split) {
    // The pool costs the epilogue one more dependent load per partner tile (the row's index) and the list kernels a scattered store per
    // item: c3 +5 %, c4 +2.5 %, c5 +2 % per tick against the dense slab (profiles/r04_row_pool_ab.txt).  So the dense slab stays while it
    // is small -- up to 1 GiB: crowds of up to ~90 000 pedestrians, c3 (0.27 GB) and c4 (0.54 GB) among them -- and the pool takes over
    // where the slab is what limits the crowd (c5: 8.6 GB -> 0.34 GB; N = 524 288: 34 GB -> 0.7 GB).  SFM_POOL=0 / 1 forces either.
    static const int pool_ov = getenv("SFM_POOL") ? atoi(getenv("SFM_POOL")) : -1;
    const size_t dense = (size_t)h->n_t * (size_t)slab_stride(h->n_t);
    h->pooled = list && (pool_ov >= 0 ? pool_ov != 0 : dense * sizeof(float2) > ((size_t)1 << 30));
    if (!h->pooled) {
        const size_t need = dense;
        if (need * sizeof(float2) > ((size_t)16 << 30)) return false;
        if (need > h->slab_cap) { if (dev_realloc(h->slab, need) != hipSuccess) return false; h->slab_cap = need; }
        if (h->z3 && need > h->slabz_cap) { if (dev_realloc(h->slabz, need) != hipSuccess) return false; h->slabz_cap = need; }
        return true;
    }
    const size_t own = (size_t)((h->i_end + WAVE - 1) / WAVE - h->i_begin / WAVE), n_t = (size_t)h->n_t;
    const bool whole = h->i_begin == 0 && h->i_end == h->N;
    const size_t every = whole ? n_t * (n_t / 2 + 1) : own * n_t;
    static const long per_ov = getenv("SFM_POOL_PER_TILE") ? atol(getenv("SFM_POOL_PER_TILE")) : 0;   // tests: a small pool makes pairs spill
    const size_t per_tile = per_ov > 0 ? (size_t)per_ov : (whole ? 80 : 144);
    const size_t main_pairs = std::min(every, std::max<size_t>(per_ov > 0 ? 1 : 4096, own * per_tile));
    const size_t pairs = main_pairs * (split ? 2 : 1);
    if (pairs * 2 * WAVE * sizeof(float2) > ((size_t)16 << 30)) return false;
    if (pairs > h->pool_pairs) { if (dev_realloc(h->pool, pairs * 2 * WAVE) != hipSuccess) return false; h->pool_pairs = pairs; }
    if (h->z3 && pairs > h->poolz_pairs) { if (dev_realloc(h->poolz, pairs * 2 * WAVE) != hipSuccess) return false; h->poolz_pairs = pairs; }
    h->pool_main = (uint32_t)main_pairs;
    if (own * n_t > h->pair_idx_cap) { if (dev_realloc(h->pair_idx, own * n_t) != hipSuccess) return false; h->pair_idx_cap = own * n_t; }
    const size_t no = (size_t)4 * (size_t)h->N_pad;
    if (no > h->ovf_cap) {
        if (dev_realloc(h->ovf, no) != hipSuccess) return false;
        h->ovf_cap = no;
        if (hipMemsetAsync(h->ovf, 0, sizeof(long long) * no, h->stream) != hipSuccess) return false;
    }
    if (!h->spill && dev_realloc(h->spill, (size_t)1) != hipSuccess) return false;
    if (h->spill_ovf != h->ovf || h->spill_n_pad != h->N_pad) {        // (once per upload that changes them; a pageable 24-byte copy)
        const SpillArgs sp{h->ovf, h->N_pad, 0};
        if (hipMemcpyAsync(h->spill, &sp, sizeof(sp), hipMemcpyHostToDevice, h->stream) != hipSuccess) return false;
        if (hipStreamSynchronize(h->stream) != hipSuccess) return false;
        h->spill_ovf = h->ovf; h->spill_n_pad = h->N_pad;
    }
    return true;
}

// the symmetric path's view of one tick: slab or row pool, tile-pair list (cutoff on), strips, own tile range
static SymArgs make_sym_args(const SfmHandle* h, const TickArgs& a, int tps, int n_strips, int debug_steps, unsigned long long* stamps) {
    const bool list = a.tile_box != nullptr;
    SymArgs sa{h->slab, h->z3 ? h->slabz : nullptr, h->n_t, slab_stride(h->n_t), debug_steps,
               list ? h->work : nullptr, list ? h->work_count : nullptr, list ? a.tile_vmax : nullptr,
               a.cut_scale, a.cut_pad, stamps, h->strip_box, h->strip_vmax, tps, n_strips,
               h->i_begin / WAVE, (h->i_end + WAVE - 1) / WAVE,
               (list && a.tile_box_out) ? 1 : 0,      // (shards: run_ticks sets it, it knows whether the tick integrates)
               nullptr, 0u, 0u, nullptr, h->tick_serial};
    if (list && h->pooled) {
        sa.slab = h->pool; sa.slabz = h->z3 ? h->poolz : nullptr; sa.idx = h->pair_idx;
        sa.row_base = 0u; sa.cap_pairs = h->pool_main;
        sa.spill = h->spill;
    }
    return sa;
}

constexpr int PHASE_FULL = 0, PHASE_BEGIN = 1, PHASE_END = 2;
constexpr int LIST_ALL = 0, LIST_OWN = 1, LIST_REMOTE = 2;        // = PARTNERS_* of sfm_kernels.hip

// Fused tick (sfm_fused_tick_kernel): one launch -- mode 0 evaluates the stored state's pairs, mode 1 integrates the
// state by one tick, stores it and evaluates the new state's pairs.  *sl = which half of fslab this launch writes.
static int fused_launch(SfmHandle* h, uint32_t flags, int mode, int* sl) {
    const int n_g = (h->n_t + 1) / 2;
    const size_t rows = (size_t)n_g * (size_t)h->N_pad;
    TickArgs a;
    fill_args(h, a, flags);
    const bool geo = a.geo != nullptr;
    int slices = 0, nw = h->fused_waves;
    const int n_pair = fused_pair_workgroups(n_g);
    if (geo) {
        // a geometry workgroup's time is the longest chain of kept polylines on one of its waves (each a dependent load -> scan), so
        // more slices = more waves per tile pay as long as every workgroup of the launch still gets a CU of its own: up to 8 from
        // 6 tiles on (all forces, N = 512 / 1024: 11.4 / 12.5 -> 10.9 / 11.3 us; N = 64 / 256: no change; 8 at N = 2048: 12.1 -> 15.1)
        slices = h->n_t <= 64 ? 4 : 2;
        if (h->n_t >= 6) slices = std::max(slices, std::min(FUSED_GEO_SLICES_MAX, (256 - n_pair) / h->n_t));
        if (h->fused_geo_slices > 0) slices = std::min(FUSED_GEO_SLICES_MAX, h->fused_geo_slices);
        // two 16-wave workgroups fill a CU: when pair + geometry workgroups do not fit in 512 such slots the launch runs 8-wave
        // workgroups (four per CU; the pair phase is within a few per cent at 4 waves per SIMD)
        if (n_pair + h->n_t * slices > 512) nw = 8;
    }
    const size_t grow = (size_t)FUSED_GEO_SLICES_MAX * (size_t)h->N_pad;
    // device-side vehicles: this launch's geometry workgroups read the vehicles at the time of the state they evaluate; its vehicle
    // workgroups write the next tick's into the other half.  Launch in front (mode 0): reads the stored ones, writes the alternate.
    // An integrating launch evaluates the NEXT state: reads the alternate, overwrites the stored half -- and then they swap.
    if (geo && a.adv.M > 0) {
        float4* c_in = mode ? h->dyn_ctr_alt : h->dynamics.ctr;
        float2* p_in = mode ? h->dyn_pts_alt : h->dynamics.pts;
        a.dynamics.ctr = c_in; a.dynamics.pts = p_in;
        a.adv.ctr = c_in; a.adv.pts = p_in;
        a.adv.ctr_out = mode ? h->dynamics.ctr : h->dyn_ctr_alt;
        a.adv.pts_out = mode ? h->dynamics.pts : h->dyn_pts_alt;
    }
    const FusedArgs f{h->fslab + (size_t)(*sl ^ 1) * rows, h->fslab + (size_t)*sl * rows,
                      h->z3 ? h->fslabz + (size_t)(*sl ^ 1) * rows : nullptr, h->z3 ? h->fslabz + (size_t)*sl * rows : nullptr,
                      h->own, h->own_alt, n_g, h->n_t, (h->fused_blocked != 0 && n_g % 8 == 0) ? 1 : 0,
                      geo ? h->fgeo + (size_t)(*sl ^ 1) * grow : nullptr, geo ? h->fgeo + (size_t)*sl * grow : nullptr, slices,
                      geo ? h->n_t * slices : 0, n_pair, h->fused_stamps, mode};
    HIP_TRY(h, launch_fused_tick(h->rad, a, f, h->stream, nw));
    if (mode != 0) {
        h->cur ^= 1;
        std::swap(h->own, h->own_alt);
        if (geo && a.adv.M > 0) {
            std::swap(h->dynamics.ctr, h->dyn_ctr_alt); std::swap(h->dynamics.ctr_cap, h->dyn_ctr_alt_cap);
            std::swap(h->dynamics.pts, h->dyn_pts_alt); std::swap(h->dynamics.pts_cap, h->dyn_pts_alt_cap);
        }
    }
    *sl ^= 1;
    return SFM_OK;
}

static int fused_reserve(SfmHandle* h) {
    const size_t need = (size_t)2 * (size_t)((h->n_t + 1) / 2) * (size_t)h->N_pad;
    if (need > h->fslab_cap) { HIP_TRY(h, dev_realloc(h->fslab, need)); h->fslab_cap = need; }
    if (h->z3 && need > h->fslabz_cap) { HIP_TRY(h, dev_realloc(h->fslabz, need)); h->fslabz_cap = need; }
    if (h->own_alt_cap < h->cap) { HIP_TRY(h, dev_realloc(h->own_alt, (size_t)h->cap)); h->own_alt_cap = h->cap; }   // same size as own: they swap
    const size_t gneed = (size_t)2 * FUSED_GEO_SLICES_MAX * (size_t)h->N_pad;
    if (gneed > h->fgeo_cap) { HIP_TRY(h, dev_realloc(h->fgeo, gneed)); h->fgeo_cap = gneed; }
    if (h->dyn_boxes && h->dynamics.K > 0) {             // the vehicles' other half: same capacities as the stored one (they swap)
        if (h->dyn_ctr_alt_cap != h->dynamics.ctr_cap) { HIP_TRY(h, dev_realloc(h->dyn_ctr_alt, std::max<size_t>(1, h->dynamics.ctr_cap))); h->dyn_ctr_alt_cap = h->dynamics.ctr_cap; }
        if (h->dyn_pts_alt_cap != h->dynamics.pts_cap) { HIP_TRY(h, dev_realloc(h->dyn_pts_alt, std::max<size_t>(1, h->dynamics.pts_cap))); h->dyn_pts_alt_cap = h->dynamics.pts_cap; }
    }
    return SFM_OK;
}

// `ticks` ticks of a whole planar crowd below the list cutoff, one launch each.  `carry`: the previous call on this handle was such
// a run too, its last launch left the partial forces of the current state in fslab (fgeo); otherwise one launch in front
// evaluates them.  Either way the run ends with state AND partial forces current.  With border / obstacle forces the rows are
// re-packed every resort_every ticks (compact tiles are what the geometry workgroups' culls live on), each re-pack followed by a
// launch in front again (the partial forces are per row).
static int run_fused(SfmHandle* h, int ticks, uint32_t flags, bool carry, bool geo) {
    int rc = fused_reserve(h);
    if (rc) return rc;
    snprintf(h->variant, sizeof(h->variant), geo ? "sfm_fused_tick_kernel(geo)" : "sfm_fused_tick_kernel");
    if (h->timing_on) HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    int sl = carry ? h->carry_sl : 0;
    bool front = !carry;
    int launches = 0;
    for (int t = 0; t < ticks && rc == SFM_OK; ++t) {
        // (only the border / obstacle culls care how compact a tile is here -- no tile-pair list below the cutoff -- and they degrade
        //  slowly: all forces at N = 2048 / 4096, 2000 ticks: 11.3 / 19.5 us per tick re-packing every 64 ticks, 11.0 / 18.8 every 128,
        //  10.8 / 18.5 every 256, 10.7 / 18.5 every 512 -- so this path re-packs every 256 ticks unless SFM_RESORT_EVERY says otherwise)
        const int period = h->resort_every_set ? h->resort_every : 4 * h->resort_every;
        if (geo && h->reordered && period > 0 && h->ticks_since_sort >= period) {
            rc = resort_rows(h);
            if (rc) return rc;
            launches += 5;
            front = true;
        }
        if (front) { rc = fused_launch(h, flags, 0, &sl); ++launches; front = false; }
        if (rc == SFM_OK) { rc = fused_launch(h, flags, 1, &sl); ++launches; }
        ++h->ticks_since_sort;
    }
    if (rc) return rc;
    if (h->timing_on) HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    h->used_fused = true;
    h->carry_ok = true;
    h->carry_seq = h->api_seq;
    h->carry_sl = sl;
    h->last_list = h->last_split = false;
    h->boxes_valid = false;
    h->count_zeroed = false;
    h->timed_ticks = ticks;
    h->timed_launches = launches;
    h->timing_valid = h->timing_on;
    h->rec_valid = false;
    return SFM_OK;
}

// geometry workgroups inside a pair launch are 4 waves each; per tile: 16 of them up to 32 tiles, 8 up to 63, 4 up to 1023, one
// from 1024 tiles on (where they are spread over the grid).  Measured, all forces (tools/mid_crowd_probe.py, c3): N = 512: 16.6 /
// 14.3 / 14.0 us per tick at 4 / 8 / 16; N = 2048: 21.0 / 16.4 / 16.4; N = 4096: 27.7 / 23.2 / 24.8 / 26.7 at 2 / 4 / 8 / 16;
// c3 (256 tiles): 47.3 / 35.2 / 34.9 / 44 at 1 / 2 / 4 / 8; c5: 777 / 781 / 797 at 1 / 2 / 4.
static int merged_geo_slices(int tiles, int slices) {
    static const int ov = exp_env("SFM_PG_SLICES") ? atoi(exp_env("SFM_PG_SLICES")) : 0;      // A/B only
    (void)slices;
    if (ov > 0) return std::min(GEO_SLICES_MAX, ov);
    // (a shard of a large crowd -- 512 own tiles of 4096: c5, one rank of 8 -- 150.3 us per tick at 4, 146.3 at 2, tools/shard_rank_time.py)
    return tiles >= 1024 ? 1 : tiles >= 512 ? 2 : tiles >= 64 ? 4 : tiles > 32 ? 8 : 16;
}

// ---- one call of sfm_tick / sfm_run / sfm_tick_begin / sfm_tick_end -------------------------------------------------------------
// run_ticks decides ONCE what kind of ticks the call runs (TickPlan) and hands over to the arrangement that ships for it:
//   run_fused            a device-resident run of a whole crowd below the list cutoff: one launch per tick (above)
//   shard_begin          first half of a shard's split tick: own tile boxes, own-own list, own-own pairs (+ geometry workgroups)
//   tick_loop            everything else, tick by tick:  tick_boxes -> tick_geometry -> tick_symmetric | tick_ordered -> tick_carry
//                        (whole crowd under the list cutoff with carried boxes, two-launch tick below it, a shard's plain tick or the
//                         second half of its split tick, the ordered kernel)
struct TickPlan {
    int n_local, ipw, team, tps, n_strips;
    bool order_pays;      // compact tiles matter (tile cutoff or border / obstacle forces): device re-packs are worth their launches
    bool list_cut;        // the tile-pair list is on
    bool plain;           // no border / obstacle forces, no list, no device-side vehicles
    bool fused_geo;       // border / obstacle forces without the list: the fused tick's geometry role can take them
    bool whole;           // the handle owns every row
    bool sym;             // the symmetric path runs (else the ordered kernel)
    bool fusable;         // ... as the one-launch tick
    bool carry;           // the call before this one was a fused run and nothing came between
};

static int plan_ticks(SfmHandle* h, uint32_t flags, int phase, bool device_run, bool carry, TickPlan& p) {
    p.carry = carry;
    p.n_local = h->i_end - h->i_begin;
    p.ipw = 1; p.team = 1;
    pick_shape(h, p.n_local, &p.ipw, &p.team);
    h->ipw_last = p.ipw;
    p.tps = 1; p.n_strips = 0;
    strip_shape(h, &p.tps, &p.n_strips);
    {
        TickArgs probe;
        fill_args(h, probe, flags);
        p.order_pays = probe.geo != nullptr || probe.tile_box != nullptr;
        p.list_cut = probe.tile_box != nullptr;
        p.plain = probe.geo == nullptr && probe.tile_box == nullptr && probe.adv.M == 0;
        p.fused_geo = probe.geo != nullptr && probe.tile_box == nullptr && h->fused_geo_mode != 0;     // (SFM_FUSED=0 switches both off)
    }
    // symmetric path: the whole crowd on this handle, or -- tile-pair list on -- a shard of whole tiles: pairs with a tile of another
    // rank are then evaluated one-sided by both ranks
    p.whole = h->i_begin == 0 && h->i_end == h->N;
    const bool tile_shard = p.n_local > 0 && (h->i_begin % WAVE) == 0 && ((h->i_end % WAVE) == 0 || h->i_end == h->N) && p.list_cut &&
                            h->n_t < 32768;
    const bool sym_wanted = (p.whole || tile_shard) && h->dpp_dir == 1 && h->sym_mode != 0 && h->prm.enabled[SFM_FORCE_PEDESTRIAN] &&
                            (h->sym_mode == 1 || h->N >= 256 || device_run || carry);
    const bool sym_any_size = sym_wanted && sym_reserve(h, p.list_cut, !p.whole && h->split_mode != 0);
    // ---- a device-resident run of a whole crowd below the list cutoff: one launch per tick (sfm_fused_tick_kernel).
    //      Every sfm_run / sfm_run_recorded takes it, whatever its length -- what a device-resident run computes must not depend on
    //      how the caller cuts it into calls (a lone sfm_run(1) pays a launch in front like the two-launch tick pays its epilogue) --
    //      and a single sfm_tick when it carries on from such a run.
    p.fusable = p.whole && (p.plain || p.fused_geo) && phase == PHASE_FULL && (device_run || carry) && h->fused_mode != 0 &&
                (flags & SFM_TICK_INTEGRATE) && !(flags & SFM_TICK_RECORD_FORCES) && !h->fsm_on && h->debug_steps < 0 && !h->stamps &&
                !h->geo_stamps && h->N >= 2;
    // Auto mode keeps host-in-the-loop ticks of crowds under 256 pedestrians on the ordered kernel (one launch against the symmetric
    // path's two); their device-resident runs are one launch per tick on the fused kernel like everybody else's (round 3: c1).
    const bool small_run = h->sym_mode < 0 && h->N < 256 && p.fusable;
    p.sym = sym_any_size && (h->sym_mode == 1 || (h->sym_mode < 0 && h->N >= 256) || small_run);
    h->used_sym = p.sym;
    if (p.sym) snprintf(h->variant, sizeof(h->variant), "sfm_pair_sym_kernel+sfm_sym_epilogue_kernel");
    else snprintf(h->variant, sizeof(h->variant), "sfm_tick_kernel<%d,%s,%s,%d>", p.ipw, h->z3 ? "true" : "false",
                  h->rad ? "true" : "false", p.team);
    h->used_fused = false;
    return SFM_OK;
}

// the geometry kernel on the side stream, joined later through ev_join
static int fork_geometry(SfmHandle* h, const TickArgs& a) {
    HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->aux, h->ev_fork, 0));
    HIP_TRY(h, launch_geometry(h->rad, a, h->aux));
    HIP_TRY(h, hipEventRecord(h->ev_join, h->aux));
    return SFM_OK;
}

// ---- split tick of a shard, first half (sfm_tick_begin): what needs only this rank's rows -- own tile boxes, the own-own list, the
//      own-own pairs with the border / obstacle workgroups in the same launch (sfm_pair_geo_kernel; SFM_PAIR_GEO=0: the geometry kernel
//      on the side stream).  Anything that cannot be split: nothing happens here and sfm_tick_end runs the whole tick.
static int shard_begin(SfmHandle* h, const TickPlan& p, uint32_t flags) {
    h->begin_done = false;
    if (!(p.sym && !p.whole && p.list_cut && !h->fsm_on && (flags & SFM_TICK_INTEGRATE) && p.n_local > 0 && h->split_mode != 0)) return SFM_OK;
    const size_t items = (size_t)h->n_t * (size_t)(h->n_t / 2 + 1);
    if (items > h->work2_cap) { HIP_TRY(h, dev_realloc(h->work2, items)); h->work2_cap = items; }
    if (h->timing_on) HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    ++h->ticks_since_sort;
    ++h->tick_serial;                              // (the two halves of a split tick share it: overflow sums of both lists)
    TickArgs a;
    fill_args(h, a, flags);
    const int t_lo = h->i_begin / WAVE, t_hi = (h->i_end + WAVE - 1) / WAVE;
    HIP_TRY(h, launch_tile_bounds(a.pk_cur, h->z3 ? a.zv_cur : nullptr, h->N, const_cast<float4*>(a.tile_box), const_cast<float*>(a.tile_vmax), h->stream,
                                  t_lo, t_hi));
    const bool ahead = h->geo_ahead && a.geo;
    h->geo_ahead = false;
    const bool merged = a.geo && !ahead && a.en_ped && h->pair_geo_mode != 0 && h->debug_steps < 0 && !h->geo_stamps;
    h->begin_forked = a.geo != nullptr && !merged;
    h->begin_geo_slices = 0;
    if (a.geo && !ahead && !merged) { const int rc = fork_geometry(h, a); if (rc) return rc; }
    SymArgs sa = make_sym_args(h, a, p.tps, 0, h->debug_steps, nullptr);
    sa.work = h->work2;
    sa.work_count = h->work_count + 1;
    sa.row_base = h->pool_main;                       // the own-own list's row pairs sit behind the main list's
    HIP_TRY(h, launch_sym_list(a, sa, h->stream, LIST_OWN, h->count_zeroed));
    if (merged) {
        a.geo_slices = merged_geo_slices(t_hi - t_lo, a.geo_slices);
        h->begin_geo_slices = a.geo_slices;
        HIP_TRY(h, launch_sym_pair_geo(h->rad, a, sa, h->stream));
    } else {
        HIP_TRY(h, launch_sym_pair(h->rad, a, sa, h->stream));
    }
    h->begin_done = true;
    h->begin_flags = flags;
    h->timed_launches = 3 + (a.geo && !ahead && !merged ? 1 : 0);
    return SFM_OK;
}

// what one tick of tick_loop has settled about its launches
struct TickShape {
    bool carried;         // the previous epilogue left this tick's tile boxes (whole crowd under the list cutoff)
    bool merged_bounds;   // tile and strip boxes came out of one launch
    bool geo_in_pair;     // the border / obstacle workgroups ride in the pair kernel's launch (sfm_pair_geo_kernel)
    bool fork;            // the geometry kernel runs on the side stream and is joined before the epilogue
    bool begun_merged;    // second half of a split tick whose first half's pair launch held the geometry workgroups
    bool list_in_geo;     // the flat tile-pair list was built by extra workgroups of the geometry launch
    bool shard_zeroed;    // the epilogue left a shard's list counters at zero
};

// boxes / largest speeds of this tick's input state (all tiles) unless the previous epilogue carried them over; with the two-level
// list the strips' boxes come out of the same launch
static int tick_boxes(SfmHandle* h, const TickPlan& p, const TickArgs& a, TickShape& s, int* launches) {
    s.carried = a.tile_box_out != nullptr && h->boxes_valid;
    s.merged_bounds = a.tile_box && !s.carried && p.sym && p.list_cut && p.n_strips > 0 && !h->z3;
    if (s.merged_bounds) {
        HIP_TRY(h, launch_tile_strip_bounds(a.pk_cur, h->N, h->n_t, p.tps, p.n_strips, const_cast<float4*>(a.tile_box),
                                            const_cast<float*>(a.tile_vmax), h->strip_box, h->strip_vmax, h->stream));
        ++*launches;
    } else if (a.tile_box && !s.carried) {
        HIP_TRY(h, launch_tile_bounds(a.pk_cur, h->z3 ? a.zv_cur : nullptr, h->N, const_cast<float4*>(a.tile_box), const_cast<float*>(a.tile_vmax), h->stream));
        ++*launches;
    }
    return SFM_OK;
}

// Where the border / obstacle forces of the tick run.  They only need the tick's input state.
//  * in the pair kernel's launch (sfm_pair_geo_kernel), the default with the symmetric path: whole crowd under the list cutoff --
//    mid-sized: four 4-wave workgroups per tile, first in the grid, c3 41.5 -> 35.1 us; two-level list: one per tile spread evenly over
//    the grid, c5 792 us with the geometry kernel on the side stream, 778 us this way (812 us with 8 192 of them in front of the pair
//    workgroups: they hold every slot for four rounds) -- and whole crowds below the cutoff (the pair kernel's 2-D grid): all forces at
//    N = 512 / 2048 / 4096: 38.9 / 37.9 / 43.6 us with the geometry kernel on the side stream (round 1's default), 19.3 / 22.2 / 31.0 us
//    with it in line on the main stream, 13.9 / 16.3 / 23.2 us in the pair kernel's launch (tools/mid_crowd_probe.py);
//  * on the side stream, joined before the epilogue: shards without the merged launch (beside the exchange and the list) and large whole
//    crowds whose boxes are not carried (c5 825 us in line, 799 us forked); a mid-sized whole crowd keeps it in line -- with carried
//    boxes nothing small runs in front of the list / pair kernels any more, their resident grid takes every wave slot before the side
//    stream's workgroups get in and the two end up back to back behind a cross-stream wait: c3 65 us forked against 45 us in line;
//  * in line on the main stream otherwise (the ordered kernel consumes them itself); a whole crowd with a flat list, carried boxes and
//    a zeroed counter then has the LIST built by extra workgroups of that launch (c3: 4 -> 3 launches).
static int tick_geometry(SfmHandle* h, const TickPlan& p, TickArgs& a, bool finishing, TickShape& s, int* launches) {
    const bool ahead = (h->geo_ahead && a.geo && p.n_local > 0 && p.sym) ||   // launched at the end of the previous tick ...
                       (finishing && h->begin_forked);                         // ... or by sfm_tick_begin: join only
    h->geo_ahead = false;
    static const int fork_ov = exp_env("SFM_FORK") ? atoi(exp_env("SFM_FORK")) : -1;      // A/B only: 0 / 1 = in line / side stream with carried boxes
    const bool fork_carried = fork_ov >= 0 ? fork_ov == 1 : h->n_t >= 1024;
    const bool plain_grid = !a.tile_box;                           // no cutoff: the pair kernel runs its 2-D grid
    s.geo_in_pair = !ahead && a.geo && p.n_local > 0 && p.sym && !finishing && a.en_ped && h->N > 1 && h->debug_steps < 0 &&
                    !h->stamps && !h->geo_stamps && h->pair_geo_mode != 0 && (p.list_cut || plain_grid);
    s.fork = !s.geo_in_pair && a.geo && p.n_local > 0 && p.sym && ((h->overlap_geo && (!p.whole || fork_carried)) || finishing);
    s.list_in_geo = false;
    if (s.geo_in_pair) a.geo_slices = merged_geo_slices((h->i_end + WAVE - 1) / WAVE - h->i_begin / WAVE, a.geo_slices);
    if (finishing && h->begin_geo_slices > 0) a.geo_slices = h->begin_geo_slices;      // sfm_tick_begin's pair launch held the geometry workgroups
    s.begun_merged = finishing && h->begin_geo_slices > 0;        // the geometry forces of this tick are already there
    if (s.geo_in_pair || s.begun_merged || ahead) return SFM_OK;
    if (s.fork) {
        const int rc = fork_geometry(h, a);
        if (rc) return rc;
        ++*launches;
    } else if (a.geo && p.n_local > 0) {
        s.list_in_geo = p.sym && p.whole && s.carried && h->count_zeroed && p.list_cut && p.n_strips == 0 && !finishing && a.en_ped &&
                        h->N > 1 && h->list_merge_mode != 0;
        if (s.list_in_geo) { a.list_work = h->work; a.list_count = h->work_count; a.list_n_t = h->n_t; a.list_idx = h->pooled ? h->pair_idx : nullptr; a.list_cap = h->pool_main; }
        HIP_TRY(h, launch_geometry(h->rad, a, h->stream));
        ++*launches;
    }
    return SFM_OK;
}

// the symmetric path's launches of one tick: [strip boxes] -> [tile-pair list] -> pair kernel (+ geometry workgroups) -> epilogue
static int tick_symmetric(SfmHandle* h, const TickPlan& p, const TickArgs& a, bool finishing, TickShape& s, int* launches) {
    const SymArgs sa = make_sym_args(h, a, p.tps, p.n_strips, h->debug_steps, h->stamps);
    h->last_list = sa.work != nullptr;
    if (sa.work && p.n_strips > 0 && !s.merged_bounds) {
        HIP_TRY(h, launch_strip_bounds(a.tile_box, a.tile_vmax, h->n_t, p.tps, p.n_strips, h->strip_box, h->strip_vmax, h->stream));
        ++*launches;
    }
    // a shard's epilogue leaves the list counter(s) at zero as well (its boxes cannot be carried -- the other ranks' rows
    // arrive in between -- but the memset can go)
    const bool shard_zero = !p.whole && p.list_cut && h->carry_mode != 0;
    if (sa.work && !s.list_in_geo) {
        HIP_TRY(h, launch_sym_list(a, sa, h->stream, finishing ? LIST_REMOTE : LIST_ALL, (s.carried || !p.whole) && h->count_zeroed));
        ++*launches;
    }
    if (s.geo_in_pair) HIP_TRY(h, launch_sym_pair_geo(h->rad, a, sa, h->stream));
    else HIP_TRY(h, launch_sym_pair(h->rad, a, sa, h->stream));
    if (s.fork && !s.begun_merged) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
    SymArgs se = sa;
    if (shard_zero) se.zero_count = 2;
    HIP_TRY(h, launch_sym_epilogue(h->rad, a, se, h->stream));
    s.shard_zeroed = shard_zero;
    *launches += 2;
    return SFM_OK;
}

// what a tick leaves for the next one: a shard's next geometry forces started beside the exchange, carried boxes, the clock, vehicles
static int tick_carry(SfmHandle* h, const TickPlan& p, const TickArgs& a, uint32_t flags, bool last, const TickShape& s, int* launches) {
    // a shard's next geometry forces only need its own new rows: start them now, beside the exchange the caller issues next
    //  (not when the geometry workgroups ride in the pair launches: then the next sfm_tick_begin / sfm_tick hosts them)
    if (p.sym && s.fork && !s.begun_merged && !p.whole && (flags & SFM_TICK_INTEGRATE) && !h->fsm_on && h->geo_ahead_mode != 0 && last &&
        !(flags & SFM_TICK_RECORD_FORCES)) {
        TickArgs nx;
        fill_args(h, nx, flags);
        const int rc = fork_geometry(h, nx);
        if (rc) return rc;
        h->geo_ahead = true;
        ++*launches;
    }
    // the epilogue left the next tick's boxes in the other buffer (valid only if this tick moved the crowd)
    if (a.tile_box_out && p.sym) {
        h->box_cur ^= 1;
        h->boxes_valid = (flags & SFM_TICK_INTEGRATE) != 0;
        h->count_zeroed = true;
    } else {
        h->boxes_valid = false;
        h->count_zeroed = s.shard_zeroed;
    }
    if (h->fsm_on) h->sim_time += h->prm.step_length;
    // CARLA-free runs: the vehicles move between ticks (run_simulation.py:77-95)
    if (a.adv.M > 0 && p.n_local <= 0) {          // a rank without rows still has to move its copy of the vehicles
        HIP_TRY(h, launch_dynamic_boxes(h->dynamics.ctr, h->dynamics.off, h->dyn_local, h->dyn_rot, h->dynamics.pts,
                                        h->dynamics.K, h->prm.step_length, 1, h->stream));
        ++*launches;
    }
    return SFM_OK;
}

// ---- ticks one by one: host-in-the-loop sfm_tick, sfm_run above the list cutoff, a shard's plain tick, and (finishing) the second
//      half of its split tick -- the own-own pairs of that tick are already in the slab / pool
static int tick_loop(SfmHandle* h, const TickPlan& p, int ticks, uint32_t flags, bool finishing) {
    int rc;
    if (finishing) snprintf(h->variant, sizeof(h->variant), "sfm_pair_sym_kernel(own|remote)+sfm_sym_epilogue_kernel");
    h->begin_done = false;
    h->last_split = finishing;
    if (!finishing && h->timing_on) HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    int launches = finishing ? h->timed_launches : 0;
    for (int t = 0; t < ticks; ++t) {
        if (h->reordered && h->resort_every > 0 && (flags & SFM_TICK_INTEGRATE) && p.whole && h->ticks_since_sort >= h->resort_every &&
            p.order_pays) {
            rc = resort_rows(h);
            if (rc) return rc;
            launches += 5;
        }
        if (!finishing) { ++h->ticks_since_sort; ++h->tick_serial; }
        TickArgs a;
        fill_args(h, a, flags);
        if (h->fsm_on && p.n_local > 0) {           // modes first: target speeds and the border mask feed the forces
            HIP_TRY(h, launch_modes(a, h->stream));
            ++launches;
        }
        TickShape s{};
        if ((rc = tick_boxes(h, p, a, s, &launches)) != SFM_OK) return rc;
        if ((rc = tick_geometry(h, p, a, finishing, s, &launches)) != SFM_OK) return rc;
        if (p.sym) {
            if ((rc = tick_symmetric(h, p, a, finishing, s, &launches)) != SFM_OK) return rc;
        } else if (p.n_local > 0) {
            HIP_TRY(h, launch_tick(p.ipw, p.team, h->z3, h->rad, a, h->stream));      // the ordered kernel: the whole tick in one launch
            ++launches;
        }
        h->cur ^= 1;
        if ((rc = tick_carry(h, p, a, flags, t + 1 == ticks, s, &launches)) != SFM_OK) return rc;
    }
    if (h->timing_on) HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    h->timed_ticks = ticks;
    h->timed_launches = launches;
    h->timing_valid = h->timing_on;
    h->rec_valid = (flags & SFM_TICK_RECORD_FORCES) != 0;
    return SFM_OK;
}

static int run_ticks(SfmHandle* h, int ticks, uint32_t flags, int phase = PHASE_FULL, bool device_run = false) {
    int rc = bind(h);
    if (rc) return rc;
    const bool carry = h->carry_ok && h->api_seq == h->carry_seq + 1;     // the call before this one was a fused run, nothing in between
    h->carry_ok = false;
    if (ticks < 0) return fail(h, SFM_ERR_INVALID, "ticks < 0");
    if (phase == PHASE_END && h->begin_done) flags = h->begin_flags;
    if (phase != PHASE_END) h->timing_valid = false;
    if (h->N == 0 || ticks == 0) return SFM_OK;        // tick() early-out (pedestrian_simulation.py:60-61)
    if (!h->pk[0]) return fail(h, SFM_ERR_STATE, "sfm_upload_state has not been called");
    rc = flush_lazy_geo(h);
    if (rc) return rc;
    TickPlan p;
    rc = plan_ticks(h, flags, phase, device_run, carry, p);
    if (rc) return rc;
    if (p.sym && p.fusable) return run_fused(h, ticks, flags, carry, p.fused_geo);
    if (phase == PHASE_BEGIN) return shard_begin(h, p, flags);
    return tick_loop(h, p, ticks, flags, phase == PHASE_END && h->begin_done);
}

int sfm_profile_dominant_kernel(SfmHandle* h, int reps, float* avg_us) {
    int rc = bind(h);
    if (rc) return rc;
    if (reps <= 0 || !avg_us) return fail(h, SFM_ERR_INVALID, "reps <= 0 or avg_us is NULL");
    if (h->N == 0 || !h->pk[0]) return fail(h, SFM_ERR_STATE, "no state uploaded");
    rc = flush_lazy_geo(h);
    if (rc) return rc;
    if (h->used_fused) {
        // the last sfm_run took the fused tick: its launches integrate, so they are timed for real on a saved state -- `reps`
        // mid-run launches (integrate + pairs) between the events -- and the state is put back afterwards
        const size_t np_ = (size_t)h->N_pad;
        float4 *pk_keep = nullptr, *own_keep = nullptr;
        uint32_t* draws_keep = nullptr;
        struct Keep {                                  // freed on every way out (HIP_TRY returns early)
            float4 *&a, *&b; uint32_t*& c;
            ~Keep() { if (a) hipFree(a); if (b) hipFree(b); if (c) hipFree(c); }
        } keep{pk_keep, own_keep, draws_keep};
        HIP_TRY(h, dev_realloc(pk_keep, np_)); HIP_TRY(h, dev_realloc(own_keep, np_)); HIP_TRY(h, dev_realloc(draws_keep, np_));
        // vehicles that move on the device move in these launches too: their centres and rings are put back as well
        const bool veh = h->dyn_boxes && h->dynamics.K > 0;
        float4* ctr_keep = nullptr;
        float2* pts_keep = nullptr;
        struct KeepV { float4*& a; float2*& b; ~KeepV() { if (a) hipFree(a); if (b) hipFree(b); } } keepv{ctr_keep, pts_keep};
        if (veh) {
            HIP_TRY(h, dev_realloc(ctr_keep, (size_t)h->dynamics.K)); HIP_TRY(h, dev_realloc(pts_keep, (size_t)std::max(1, h->dynamics.P)));
            HIP_TRY(h, hipMemcpyAsync(ctr_keep, h->dynamics.ctr, sizeof(float4) * (size_t)h->dynamics.K, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(pts_keep, h->dynamics.pts, sizeof(float2) * (size_t)h->dynamics.P, hipMemcpyDeviceToDevice, h->stream));
        }
        // a 3-D crowd's {z, vz} rows are integrated by these launches as well (round-3 advisor finding: they were left reps + 1 ticks ahead)
        float2* zv_keep = nullptr;
        struct KeepZ { float2*& a; ~KeepZ() { if (a) hipFree(a); } } keepz{zv_keep};
        if (h->z3) {
            HIP_TRY(h, dev_realloc(zv_keep, np_));
            HIP_TRY(h, hipMemcpyAsync(zv_keep, h->zv[h->cur], sizeof(float2) * np_, hipMemcpyDeviceToDevice, h->stream));
        }
        HIP_TRY(h, hipMemcpyAsync(pk_keep, h->pk[h->cur], sizeof(float4) * np_, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(own_keep, h->own, sizeof(float4) * np_, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(draws_keep, h->draws, sizeof(uint32_t) * np_, hipMemcpyDeviceToDevice, h->stream));
        rc = fused_reserve(h);
        const uint32_t fl = SFM_TICK_INTEGRATE | SFM_TICK_REDRAW_WAYPOINTS;
        int sl = 0;
        if (!rc) rc = fused_launch(h, fl, 0, &sl);
        if (!rc) rc = fused_launch(h, fl, 1, &sl);
        if (rc) return rc;
        HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
        for (int r = 0; r < reps && rc == SFM_OK; ++r) rc = fused_launch(h, fl, 1, &sl);
        if (rc) return rc;
        HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->pk[h->cur], pk_keep, sizeof(float4) * np_, hipMemcpyDeviceToDevice, h->stream));
        if (h->z3) HIP_TRY(h, hipMemcpyAsync(h->zv[h->cur], zv_keep, sizeof(float2) * np_, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->own, own_keep, sizeof(float4) * np_, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->draws, draws_keep, sizeof(uint32_t) * np_, hipMemcpyDeviceToDevice, h->stream));
        if (veh) {
            HIP_TRY(h, hipMemcpyAsync(h->dynamics.ctr, ctr_keep, sizeof(float4) * (size_t)h->dynamics.K, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->dynamics.pts, pts_keep, sizeof(float2) * (size_t)h->dynamics.P, hipMemcpyDeviceToDevice, h->stream));
        }
        HIP_TRY(h, hipEventSynchronize(h->ev1));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        *avg_us = ms * 1000.0f / (float)reps;
        h->carry_ok = false;
        h->timing_valid = false;
        return SFM_OK;
    }
    rc = run_ticks(h, 1, 0);                       // settles the launch shape (and the cutoff work list) ...
    if (rc) return rc;
    h->cur ^= 1;                                   // ... and steps back: the probe must not advance the state
    const int n_local = h->i_end - h->i_begin;
    int ipw = 1, team = 1;
    pick_shape(h, n_local, &ipw, &team);
    int tps = 1, n_strips = 0;
    strip_shape(h, &tps, &n_strips);
    TickArgs a;
    fill_args(h, a, 0);
    if (h->used_sym) a.geo = nullptr;
    const SymArgs sa = make_sym_args(h, a, tps, n_strips, -1, nullptr);
    if (a.tile_box) HIP_TRY(h, launch_tile_bounds(a.pk_cur, h->z3 ? a.zv_cur : nullptr, h->N, const_cast<float4*>(a.tile_box), const_cast<float*>(a.tile_vmax), h->stream));
    if (sa.work && n_strips > 0)
        HIP_TRY(h, launch_strip_bounds(a.tile_box, a.tile_vmax, h->n_t, tps, n_strips, h->strip_box, h->strip_vmax, h->stream));
    if (h->used_sym) HIP_TRY(h, launch_sym_list(a, sa, h->stream));      // the list is built once, outside the timed launches
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < reps; ++r) {
        if (h->used_sym) HIP_TRY(h, launch_sym_pair(h->rad, a, sa, h->stream));
        else if (n_local > 0) HIP_TRY(h, launch_tick(ipw, team, h->z3, h->rad, a, h->stream));
    }
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_us = ms * 1000.0f / (float)reps;
    h->timing_valid = false;
    return SFM_OK;
}

int sfm_tick(SfmHandle* h, uint32_t flags) { return run_ticks(h, 1, flags); }

int sfm_tick_begin(SfmHandle* h, uint32_t flags) { return run_ticks(h, 1, flags | SFM_TICK_INTEGRATE, PHASE_BEGIN); }

int sfm_tick_end(SfmHandle* h, uint32_t flags) { return run_ticks(h, 1, flags | SFM_TICK_INTEGRATE, PHASE_END); }

int sfm_run(SfmHandle* h, int ticks, uint32_t flags) { return run_ticks(h, ticks, flags | SFM_TICK_INTEGRATE, PHASE_FULL, true); }

int sfm_run_recorded(SfmHandle* h, int ticks, uint32_t flags, int stride, float* frames, int max_frames, int* n_frames) {
    int rc = bind(h);
    if (rc) return rc;
    if (ticks < 0 || stride <= 0 || max_frames < 0 || (!frames && max_frames > 0) || !n_frames)
        return fail(h, SFM_ERR_INVALID, "bad recording arguments");
    *n_frames = 0;
    if (h->N == 0 || ticks == 0) return SFM_OK;
    if (!h->pk[0]) return fail(h, SFM_ERR_STATE, "sfm_upload_state has not been called");
    const int want = std::min(max_frames, (ticks + stride - 1) / stride);
    const size_t frame_recs = (size_t)h->N;
    float4* stage = nullptr;                                   // pinned, rows in the library's order ...
    uint32_t* stage_ids = nullptr;                             // ... which a device re-sort may change between frames
    if (want > 0) HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&stage), sizeof(float4) * frame_recs * (size_t)want, 0));
    if (want > 0 && h->reordered)
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&stage_ids), sizeof(uint32_t) * frame_recs * (size_t)want, 0));
    int done = 0, f = 0;
    while (done < ticks && rc == SFM_OK) {
        if (f < want) {
            hipError_t e = hipMemcpyAsync(stage + frame_recs * (size_t)f, h->pk[h->cur], sizeof(float4) * frame_recs,
                                          hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess && stage_ids)
                e = hipMemcpyAsync(stage_ids + frame_recs * (size_t)f, h->ids, sizeof(uint32_t) * frame_recs, hipMemcpyDeviceToHost,
                                   h->stream);
            if (e != hipSuccess) { h->err = std::string("hipMemcpyAsync: ") + hipGetErrorString(e); rc = SFM_ERR_HIP; break; }
            ++f;
        }
        const int chunk = std::min(stride, ticks - done);
        rc = run_ticks(h, chunk, flags | SFM_TICK_INTEGRATE, PHASE_FULL, true);
        done += chunk;
    }
    if (rc == SFM_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, SFM_ERR_HIP, "hipStreamSynchronize failed");
    if (rc == SFM_OK) {
        for (int k = 0; k < f; ++k)
            for (int s_ = 0; s_ < h->N; ++s_) {
                const float4 v = stage[frame_recs * (size_t)k + s_];
                const uint32_t id = stage_ids ? stage_ids[frame_recs * (size_t)k + s_] : h->perm[s_];
                float* dst = frames + ((size_t)k * h->N + id) * 4;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
        *n_frames = f;
    }
    if (stage) hipHostFree(stage);
    if (stage_ids) hipHostFree(stage_ids);
    h->timing_valid = false;
    return rc;
}

// ---- downloads ------------------------------------------------------------------------------------

static int fetch_packed(SfmHandle* h, std::vector<float4>& pk, std::vector<float2>& zv, bool want_z) {
    const int n = h->i_end - h->i_begin;
    pk.resize((size_t)n);
    HIP_TRY(h, hipMemcpyAsync(pk.data(), h->pk[h->cur] + h->i_begin, sizeof(float4) * (size_t)n,
                              hipMemcpyDeviceToHost, h->stream));
    if (want_z && h->z3) {
        zv.resize((size_t)n);
        HIP_TRY(h, hipMemcpyAsync(zv.data(), h->zv[h->cur] + h->i_begin, sizeof(float2) * (size_t)n,
                                  hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SFM_OK;
}

int sfm_download_velocities(SfmHandle* h, float* vx, float* vy, float* vz) {
    int rc = bind(h);
    if (rc) return rc;
    rc = sync_perm(h);
    if (rc) return rc;
    if (h->N == 0) return SFM_OK;
    if (!vx || !vy) return fail(h, SFM_ERR_INVALID, "vx / vy is NULL");
    std::vector<float4> pk;
    std::vector<float2> zv;
    rc = fetch_packed(h, pk, zv, vz != nullptr);
    if (rc) return rc;
    for (int s_ = h->i_begin; s_ < h->i_end; ++s_) {
        const int i = (int)h->perm[s_];
        vx[i] = pk[s_ - h->i_begin].z;
        vy[i] = pk[s_ - h->i_begin].w;
        if (vz) vz[i] = h->z3 ? zv[s_ - h->i_begin].y : 0.f;
    }
    return SFM_OK;
}

// One host-in-the-loop tick in ONE call (round 4): what the drop-in facade does per PedestrianSimulation.tick
// (pedestrian_simulation.py:57-83 as run_simulation.py:102-114 drives it) -- numeric columns up, one tick, v' down -- without three
// trips through the binding and nine column conversions on the caller's side.
//   rows  [N][9]  {x, y, vx, vy, waypoint x, waypoint y, target_speed, radius, border-force-off flag (0 / 1)}
//   zvz   [N][2]  {z, vz}, or NULL for a planar crowd
//   v_out [N][3]  {vx', vy', vz'} (vz' = 0 for a planar crowd)
// The block is taken apart into the arrays sfm_upload_state consumes (crowds that go through here are host-in-the-loop crowds of a
// few thousand pedestrians at most: a pass over 36 N bytes), so everything an upload settles -- packing, padding rows, capacities --
// is settled by the same code; v' comes back through a pinned block.
// sfm_step_packed / sfm_step_records behind their column fill: h->step_cols holds {x, y, vx, vy, wx, wy, ts, rr, z, vz} [N] each and
// h->step_mask the border-force-off flags; upload, one tick with its last kernel writing v' into pinned host memory, v' out.
static int step_core(SfmHandle* h, int N, bool z3, uint32_t flags, float* v_out) {
    const size_t n = (size_t)N;
    float* c = h->step_cols.data();
    float *x = c, *y = x + n, *vx = y + n, *vy = vx + n, *wx = vy + n, *wy = wx + n, *ts = wy + n, *rr = ts + n, *z = rr + n, *vz = z + n;
    int rc = sfm_upload_state(h, N, x, y, z3 ? z : nullptr, vx, vy, z3 ? vz : nullptr, wx, wy, ts, rr, h->step_mask.data());
    if (rc || N == 0) return rc;
    // v' of all rows (the upload reset the shard): the tick's last kernel writes the new rows into a pinned host block as well, so
    // after the stream has drained they are simply there -- no device-to-host copy on the way back
    const size_t np_ = (size_t)h->N_pad;
    const size_t need = (sizeof(float4) + sizeof(float2)) * np_;
    if (need > h->down_stage_cap) {
        if (h->down_stage) { hipHostFree(h->down_stage); h->down_stage = nullptr; h->down_stage_cap = 0; }
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->down_stage), need + need / 2, 0));
        h->down_stage_cap = need + need / 2;
    }
    h->hosted = true;
    rc = run_ticks(h, 1, flags);
    h->hosted = false;
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const float4* pk = reinterpret_cast<const float4*>(h->down_stage);
    const float2* zv = reinterpret_cast<const float2*>(h->down_stage + sizeof(float4) * np_);
    for (size_t s_ = 0; s_ < n; ++s_) {
        const size_t i = h->perm[s_];
        v_out[3 * i] = pk[s_].z; v_out[3 * i + 1] = pk[s_].w; v_out[3 * i + 2] = h->z3 ? zv[s_].y : 0.f;
    }
    return SFM_OK;
}

int sfm_step_packed(SfmHandle* h, int N, const float* rows, const float* zvz, uint32_t flags, float* v_out) {
    if (!h) return SFM_ERR_INVALID;
    if (N < 0 || (N > 0 && (!rows || !v_out))) { int rc = bind(h); return rc ? rc : fail(h, SFM_ERR_INVALID, "N < 0, or rows / v_out is NULL"); }
    std::vector<float>& c = h->step_cols;
    const size_t n = (size_t)N;
    c.resize(10 * n + 1);
    float *x = c.data(), *y = x + n, *vx = y + n, *vy = vx + n, *wx = vy + n, *wy = wx + n, *ts = wy + n, *rr = ts + n, *z = rr + n, *vz = z + n;
    h->step_mask.resize(n + 1);
    uint8_t* cm = h->step_mask.data();
    for (size_t i = 0; i < n; ++i) {
        const float* r = rows + 9 * i;
        x[i] = r[0]; y[i] = r[1]; vx[i] = r[2]; vy[i] = r[3]; wx[i] = r[4]; wy[i] = r[5]; ts[i] = r[6]; rr[i] = r[7];
        cm[i] = r[8] != 0.0f ? 1 : 0;
        if (zvz) { z[i] = zvz[2 * i]; vz[i] = zvz[2 * i + 1]; }
    }
    return step_core(h, N, zvz != nullptr, flags, v_out);
}

// The same straight from the caller's RECORDS (ABI 5): what PedestrianState keeps per pedestrian is one packed structured-array row
// (pedestrian_state.py:17-23: name, id, loc[3], vel[3], next_waypoint[3], mode object, radius, target_speed -- 132 bytes, float64 fields
// at unaligned offsets), and pedestrian_simulation.py:57-83 reads five of its fields every tick.  Gathering them here instead of in six
// strided NumPy assignments takes 5-10 us off a drop-in tick.
int sfm_step_records(SfmHandle* h, int N, const void* records, int64_t stride, const int32_t* field_offsets, const uint8_t* border_off,
                     float planar_tolerance, uint32_t flags, float* v_out, int32_t* was_planar) {
    if (!h) return SFM_ERR_INVALID;
    if (N < 0 || (N > 0 && (!records || !field_offsets || !v_out)) || stride < 0) {
        int rc = bind(h);
        return rc ? rc : fail(h, SFM_ERR_INVALID, "N < 0, negative stride, or records / field_offsets / v_out is NULL");
    }
    std::vector<float>& c = h->step_cols;
    const size_t n = (size_t)N;
    c.resize(10 * n + 1);
    float *x = c.data(), *y = x + n, *vx = y + n, *vy = vx + n, *wx = vy + n, *wy = wx + n, *ts = wy + n, *rr = ts + n, *z = rr + n, *vz = z + n;
    h->step_mask.resize(n + 1);
    uint8_t* cm = h->step_mask.data();
    const char* base = static_cast<const char*>(records);
    const int o_loc = field_offsets[0], o_vel = field_offsets[1], o_wp = field_offsets[2], o_rad = field_offsets[3], o_ts = field_offsets[4];
    double z0 = 0.0, z_lo = 0.0, z_hi = 0.0, vz_max = 0.0;
    bool flat = true;
    for (size_t i = 0; i < n; ++i) {
        const char* r = base + (size_t)stride * i;
        double l[3], v[3], w[2], rad, t;                   // (memcpy: the float64 fields of a packed record are not 8-byte aligned)
        memcpy(l, r + o_loc, sizeof(l)); memcpy(v, r + o_vel, sizeof(v)); memcpy(w, r + o_wp, sizeof(w));
        memcpy(&rad, r + o_rad, sizeof(rad)); memcpy(&t, r + o_ts, sizeof(t));
        x[i] = (float)l[0]; y[i] = (float)l[1]; z[i] = (float)l[2];
        vx[i] = (float)v[0]; vy[i] = (float)v[1]; vz[i] = (float)v[2];
        wx[i] = (float)w[0]; wy[i] = (float)w[1]; ts[i] = (float)t; rr[i] = (float)rad;
        cm[i] = (border_off && border_off[i]) ? 1 : 0;
        if (i == 0) { z0 = z_lo = z_hi = l[2]; }
        flat = flat && l[2] == z0 && v[2] == 0.0;
        z_lo = std::fmin(z_lo, l[2]); z_hi = std::fmax(z_hi, l[2]); vz_max = std::fmax(vz_max, std::fabs(v[2]));
    }
    // the 2-D kernels iff all z are equal and no pedestrian has a v_z (then the 3-component formulas of forces.py:74-117 and
    // stateutils.py:18-23 reduce to them exactly) -- or, planar_tolerance >= 0, nearly so: every |z - median z| and |v_z| within it
    // (the facade's documented deviation for walkers on almost level ground)
    if (!flat && planar_tolerance >= 0.0f && N > 0 && vz_max <= (double)planar_tolerance && z_hi - z_lo <= 2.0 * (double)planar_tolerance) {
        std::vector<double> zs(n);
        for (size_t i = 0; i < n; ++i) { double l2; memcpy(&l2, base + (size_t)stride * i + o_loc + 2 * sizeof(double), sizeof(l2)); zs[i] = l2; }
        std::sort(zs.begin(), zs.end());
        const double med = (n & 1) ? zs[n / 2] : 0.5 * (zs[n / 2 - 1] + zs[n / 2]);        // np.median
        flat = std::fmax(z_hi - med, med - z_lo) <= (double)planar_tolerance;
    }
    if (was_planar) *was_planar = flat ? 1 : 0;
    return step_core(h, N, !flat, flags, v_out);
}

int sfm_download_state(SfmHandle* h, float* x, float* y, float* z, float* vx, float* vy, float* vz, float* wx,
                       float* wy) {
    int rc = bind(h);
    if (rc) return rc;
    rc = sync_perm(h);
    if (rc) return rc;
    if (h->N == 0) return SFM_OK;
    std::vector<float4> pk, own;
    std::vector<float2> zv;
    rc = fetch_packed(h, pk, zv, z || vz);
    if (rc) return rc;
    const int n = h->i_end - h->i_begin;
    if (wx || wy) {
        own.resize((size_t)n);
        HIP_TRY(h, hipMemcpy(own.data(), h->own + h->i_begin, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost));
    }
    for (int s_ = h->i_begin; s_ < h->i_end; ++s_) {
        const int k = s_ - h->i_begin;
        const int i = (int)h->perm[s_];
        if (x) x[i] = pk[k].x;
        if (y) y[i] = pk[k].y;
        if (vx) vx[i] = pk[k].z;
        if (vy) vy[i] = pk[k].w;
        if (z) z[i] = h->z3 ? zv[k].x : 0.f;
        if (vz) vz[i] = h->z3 ? zv[k].y : 0.f;
        if (wx) wx[i] = own[k].x;
        if (wy) wy[i] = own[k].y;
    }
    return SFM_OK;
}

int sfm_download_forces(SfmHandle* h, int which, float* fx, float* fy, float* fz) {
    int rc = bind(h);
    if (rc) return rc;
    rc = sync_perm(h);
    if (rc) return rc;
    if (which < 0 || which > SFM_FORCE_TOTAL) return fail(h, SFM_ERR_INVALID, "unknown force index");
    if (h->N == 0) return SFM_OK;
    if (!h->rec_valid) return fail(h, SFM_ERR_STATE, "last tick did not run with SFM_TICK_RECORD_FORCES");
    if (!fx || !fy) return fail(h, SFM_ERR_INVALID, "fx / fy is NULL");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->N;
    const int cnt = h->i_end - h->i_begin;
    float* dst[3] = {fx, fy, fz};
    std::vector<float> tmp((size_t)cnt);
    for (int c = 0; c < 3; ++c) {
        if (!dst[c]) continue;
        HIP_TRY(h, hipMemcpy(tmp.data(), h->rec + ((size_t)which * 3 + c) * n + h->i_begin,
                             sizeof(float) * (size_t)cnt, hipMemcpyDeviceToHost));
        for (int k = 0; k < cnt; ++k) dst[c][h->perm[(size_t)h->i_begin + k]] = tmp[k];
    }
    return SFM_OK;
}

int sfm_get_arrived(SfmHandle* h, float threshold, uint8_t* mask) {
    int rc = bind(h);
    if (rc) return rc;
    rc = sync_perm(h);
    if (rc) return rc;
    if (h->N == 0) return SFM_OK;
    if (!mask) return fail(h, SFM_ERR_INVALID, "mask is NULL");
    const float thr2 = (float)((double)threshold * (double)threshold);
    HIP_TRY(h, launch_arrived(h->pk[h->cur], h->own, h->N, thr2, h->arrived, h->stream));
    std::vector<uint8_t> tmp((size_t)h->N);
    HIP_TRY(h, hipMemcpyAsync(tmp.data(), h->arrived, (size_t)h->N, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int s_ = 0; s_ < h->N; ++s_) mask[h->perm[s_]] = tmp[s_];
    return SFM_OK;
}

int sfm_download_draw_counts(SfmHandle* h, uint32_t* counts) {
    int rc = bind(h);
    if (rc) return rc;
    rc = sync_perm(h);
    if (rc) return rc;
    if (h->N == 0) return SFM_OK;
    if (!counts) return fail(h, SFM_ERR_INVALID, "counts is NULL");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int cnt = h->i_end - h->i_begin;
    std::vector<uint32_t> tmp((size_t)cnt);
    HIP_TRY(h, hipMemcpy(tmp.data(), h->draws + h->i_begin, sizeof(uint32_t) * (size_t)cnt, hipMemcpyDeviceToHost));
    for (int k = 0; k < cnt; ++k) counts[h->perm[(size_t)h->i_begin + k]] = tmp[k];
    return SFM_OK;
}

// A device pointer handed out: the caller may write rows through it, so what was derived from the stored state does not survive:
// the fused tick's partial forces and the tile boxes the last epilogue left for the next tick's list.  On a WHOLE-crowd handle
// also geometry forces launched ahead.  A shard's protocol is that the caller writes the OTHER ranks' rows only (the exchange):
// its geometry forces launched ahead and a begun split tick (sfm_tick_begin) read nothing but own rows and stay.
static void caller_may_write(SfmHandle* h) {
    h->carry_ok = false;
    h->boxes_valid = false;
    if (h->i_begin == 0 && h->i_end == h->N) drop_geo_ahead(h);
}

void* sfm_packed_state_ptr(SfmHandle* h, int* n_pad) {
    if (!h) return nullptr;
    caller_may_write(h);
    if (n_pad) *n_pad = h->N_pad;
    return h->pk[h->cur];
}

void* sfm_packed_z_ptr(SfmHandle* h) {
    if (!h || !h->z3) return nullptr;
    caller_may_write(h);
    return h->zv[h->cur];
}

void* sfm_row_data_ptr(SfmHandle* h, int which, int* bytes_per_row) {
    if (!h || which < 0 || which > 1) return nullptr;
    caller_may_write(h);
    if (bytes_per_row) *bytes_per_row = which == 0 ? (int)sizeof(float4) : (int)sizeof(uint32_t);
    return which == 0 ? (void*)h->own : (void*)h->draws;
}

int sfm_resort(SfmHandle* h) {
    int rc = bind(h);
    if (rc) return rc;
    if (!h->pk[0]) return fail(h, SFM_ERR_STATE, "sfm_upload_state has not been called");
    if (!h->reordered || h->N == 0) return SFM_OK;
    if (h->fsm_on && !(h->i_begin == 0 && h->i_end == h->N))
        return fail(h, SFM_ERR_STATE, "sfm_resort on a shard with the device-side mode state machine");
    return resort_rows(h);
}

const char* sfm_last_error(const SfmHandle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int sfm_set_timing(SfmHandle* h, int enable) {
    if (!h) return SFM_ERR_INVALID;
    h->carry_ok = false;
    h->timing_on = enable != 0;
    if (!h->timing_on) h->timing_valid = false;
    return SFM_OK;
}

int sfm_get_timing(SfmHandle* h, float* elapsed_ms, int* ticks, int* launches) {
    int rc = bind(h);
    if (rc) return rc;
    if (!h->timing_valid) return fail(h, SFM_ERR_STATE, h->timing_on ? "no timed run yet" : "timing is switched off (sfm_set_timing)");
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    if (ticks) *ticks = h->timed_ticks;
    if (launches) *launches = h->timed_launches;
    return SFM_OK;
}

const char* sfm_kernel_variant(const SfmHandle* h) { return h ? h->variant : "none"; }

int sfm_get_pair_work(SfmHandle* h, long long* tile_pair_items, long long* pair_terms) {
    int rc = bind(h);
    if (rc) return rc;
    if (!h->pk[0]) return fail(h, SFM_ERR_STATE, "sfm_upload_state has not been called");
    if (!h->used_sym) return fail(h, SFM_ERR_STATE, "the last tick did not run the symmetric pair kernel");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const long long t_own = (h->i_end + WAVE - 1) / WAVE - h->i_begin / WAVE;
    long long items, diag_items = (t_own + 1) / 2;
    if (h->last_list) {
        int cnt[4] = {0, 0, 0, 0};
        HIP_TRY(h, hipMemcpy(cnt, h->work_count, sizeof(int) * 4, hipMemcpyDeviceToHost));
        // (a zeroing epilogue kept copies in [2], [3]; split tick: the own-own items were listed separately, in [1])
        items = (h->count_zeroed ? cnt[2] : cnt[0]) + (h->last_split ? (h->count_zeroed ? cnt[3] : cnt[1]) : 0);
    } else {                                       // 2-D grid: every unordered tile pair once + the diagonal items
        const long long n_t = h->n_t;
        items = n_t * (n_t - 1) / 2 + diag_items;
    }
    if (tile_pair_items) *tile_pair_items = items;
    // a tile-pair item meets 64 x 64 pairs; a diagonal item holds two diagonal tiles of 64 x 32 evaluations each
    if (pair_terms) *pair_terms = (items - diag_items) * (long long)(WAVE * WAVE) + t_own * (long long)(WAVE * WAVE / 2);
    return SFM_OK;
}

}  // extern "C"
