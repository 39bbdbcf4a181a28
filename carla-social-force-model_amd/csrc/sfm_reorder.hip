// sfm_reorder.hip -- periodic spatial re-sort of the pedestrian rows on the device.
//
// The tile cutoff (tiles_negligible) and the geometry kernel's tile-level culling only pay while every 64-row tile
// is a compact patch.  Row order is "sort-tile-recursive" packing: sort by x, cut the ranking into strips of
// `strip_rows` rows (a multiple of 64, about sqrt(#tiles) tiles each), sort every strip by y.  Each tile is then
// exactly the 64 pedestrians of one axis-aligned rectangle (strip width x a run in y): no L-shaped or split tiles,
// whatever the density, and no padding rows.  (A space-filling-curve order gives tiles that straddle two or three
// curve blocks as soon as the counts stop lining up with powers of two: 2-3x larger boxes on an evolved crowd.)
// sfm_upload_state packs once on the host; pedestrians then walk ~0.06 m per tick, so a device-resident run
// re-packs every few dozen ticks: two rocPRIM radix sorts + a gather of every per-row array.  A pure function of
// the state (stable sorts, ties by row), so it is deterministic; despawned pedestrians (parked ~3e15 m away) sort
// into the last strip.
#include "sfm_device.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace sfm {

// order-preserving map float -> uint32 (NaN after +inf).  Same routine on the host in sfm_capi.hip.
__host__ __device__ inline uint32_t float_key(float v) {
    uint32_t b;
    memcpy(&b, &v, sizeof(b));
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void sfm_x_keys_kernel(const float4* __restrict__ pk, int N, uint32_t* __restrict__ key, uint32_t* __restrict__ row) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    key[s] = float_key(pk[s].x);
    row[s] = (uint32_t)s;
}

// rank r of the x order -> key (strip, y)
__global__ void sfm_strip_keys_kernel(const float4* __restrict__ pk, int N, int strip_rows, const uint32_t* __restrict__ row,
                                      unsigned long long* __restrict__ key) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    key[r] = ((unsigned long long)(uint32_t)(r / strip_rows) << 32) | float_key(pk[row[r]].y);
}

__global__ void sfm_gather_rows_kernel(const uint32_t* __restrict__ src, int N, const float4* __restrict__ pk_in,
                                       float4* __restrict__ pk_out, const float2* __restrict__ zv_in, float2* __restrict__ zv_out,
                                       const float4* __restrict__ own_in, float4* __restrict__ own_out,
                                       const float* __restrict__ rad_in, float* __restrict__ rad_out,
                                       const uint8_t* __restrict__ cr_in, uint8_t* __restrict__ cr_out,
                                       const uint32_t* __restrict__ dr_in, uint32_t* __restrict__ dr_out,
                                       const uint32_t* __restrict__ id_in, uint32_t* __restrict__ id_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const uint32_t o = src[s];
    pk_out[s] = pk_in[o];
    if (zv_in) zv_out[s] = zv_in[o];
    own_out[s] = own_in[o];
    rad_out[s] = rad_in[o];
    cr_out[s] = cr_in[o];
    dr_out[s] = dr_in[o];
    id_out[s] = id_in[o];
}

// a small polyline set staged as [ctr | pts | off] words in a pinned host block (launch_unpack_geo below), and where its words go
struct GeoWords {
    const uint32_t* block;
    uint32_t *ctr, *pts, *off;       // off null: the device already holds these offsets
    int w_ctr, w_pts, w_off;
};

__device__ __forceinline__ void unpack_geo_words(const GeoWords& g, int first, int step) {
    const int total = g.w_ctr + g.w_pts + g.w_off;
    for (int q = first; q < total; q += step) {
        const uint32_t v = g.block[q];
        if (q < g.w_ctr) g.ctr[q] = v;
        else if (q < g.w_ctr + g.w_pts) g.pts[q - g.w_ctr] = v;
        else if (g.off) g.off[q - g.w_ctr - g.w_pts] = v;
    }
}

// sfm_upload_state: the rows arrive as one block [pk | own | zv | radius | crossing]; one thread per row spreads it over the
// device arrays (both halves of the ping-pong buffers) and zeroes the draw counter.  A vehicle report staged since the last tick
// (sfm_set_dynamic_obstacles_packed) rides along in one more workgroup: a host-in-the-loop tick is launch-bound, and this was one.
__global__ void sfm_unpack_rows_kernel(const char* __restrict__ block, size_t b_own, size_t b_zv, size_t b_rr, size_t b_cm,
                                       int n_pad, float4* __restrict__ pk0, float4* __restrict__ pk1, float4* __restrict__ own,
                                       float2* __restrict__ zv0, float2* __restrict__ zv1, float* __restrict__ radius,
                                       uint8_t* __restrict__ crossing, uint32_t* __restrict__ draws, int row_blocks, const GeoWords geo) {
    if ((int)blockIdx.x >= row_blocks) {
        unpack_geo_words(geo, (int)threadIdx.x, (int)blockDim.x);
        return;
    }
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_pad) return;
    const float4 p = reinterpret_cast<const float4*>(block)[s];
    pk0[s] = p; pk1[s] = p;
    own[s] = reinterpret_cast<const float4*>(block + b_own)[s];
    const float2 z = reinterpret_cast<const float2*>(block + b_zv)[s];
    zv0[s] = z; zv1[s] = z;
    radius[s] = reinterpret_cast<const float*>(block + b_rr)[s];
    crossing[s] = reinterpret_cast<const uint8_t*>(block + b_cm)[s];
    draws[s] = 0u;
}

static GeoWords geo_words(const char* block, float4* ctr, int K, float2* pts, int P, int* off, bool with_off) {
    return GeoWords{reinterpret_cast<const uint32_t*>(block), reinterpret_cast<uint32_t*>(ctr), reinterpret_cast<uint32_t*>(pts),
                    with_off ? reinterpret_cast<uint32_t*>(off) : nullptr, 4 * K, 2 * P, K + 1};
}

// geo_block null: rows only
hipError_t launch_unpack_rows(const char* block, size_t b_own, size_t b_zv, size_t b_rr, size_t b_cm, int n_pad, float4* pk0,
                              float4* pk1, float4* own, float2* zv0, float2* zv1, float* radius, uint8_t* crossing, uint32_t* draws,
                              const char* geo_block, float4* ctr, int K, float2* pts, int P, int* off, bool with_off, hipStream_t st) {
    const int row_blocks = (n_pad + 255) / 256;
    const GeoWords g = geo_block ? geo_words(geo_block, ctr, K, pts, P, off, with_off) : GeoWords{nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    hipLaunchKernelGGL(sfm_unpack_rows_kernel, dim3(row_blocks + (geo_block ? 1 : 0)), dim3(256), 0, st, block, b_own, b_zv, b_rr, b_cm,
                       n_pad, pk0, pk1, own, zv0, zv1, radius, crossing, draws, row_blocks, g);
    return hipGetLastError();
}

// set_geo for a small polyline set that arrives every tick (the simulator's vehicles, obstacles.py:297-329): ONE launch reads the
// pinned host block [ctr | pts | off] over the bus and spreads it over the three device arrays, instead of three small copies
// (round 4: ~8 us each in front of every host-in-the-loop tick).  Word copies; a few KiB.
__global__ void sfm_unpack_geo_kernel(const GeoWords g) {
    unpack_geo_words(g, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

hipError_t launch_unpack_geo(const char* block, float4* ctr, int K, float2* pts, int P, int* off, bool with_off, hipStream_t st) {
    const GeoWords g = geo_words(block, ctr, K, pts, P, off, with_off);
    const int total = g.w_ctr + g.w_pts + g.w_off;
    hipLaunchKernelGGL(sfm_unpack_geo_kernel, dim3(std::min(64, (total + 255) / 256)), dim3(256), 0, st, g);
    return hipGetLastError();
}

__device__ __forceinline__ int plan_block_of(const BlockPlan& pl, int r) {       // block holding rank r of the block order
    int b = 0;
    while (b + 1 < pl.n_blocks && r >= pl.bound[b + 1]) ++b;
    return b;
}

// rank r of the x order -> key (column, y): the columns take their row counts from the block bounds
__global__ void sfm_column_keys_kernel(const float4* __restrict__ pk, int N, const BlockPlan pl, const uint32_t* __restrict__ row,
                                       unsigned long long* __restrict__ key) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const int col = plan_block_of(pl, r) / pl.gy;
    key[r] = ((unsigned long long)(uint32_t)col << 32) | float_key(pk[row[r]].y);
}

// rank r of the (column, y) order -> key (block, x)
__global__ void sfm_block_keys_kernel(const float4* __restrict__ pk, int N, const BlockPlan pl, const uint32_t* __restrict__ row,
                                      unsigned long long* __restrict__ key) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    key[r] = ((unsigned long long)(uint32_t)plan_block_of(pl, r) << 32) | float_key(pk[row[r]].x);
}

// rank r of the (block, x) order -> key (block, strip inside the block, y)
__global__ void sfm_block_strip_keys_kernel(const float4* __restrict__ pk, int N, const BlockPlan pl, const uint32_t* __restrict__ row,
                                            unsigned long long* __restrict__ key) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const int b = plan_block_of(pl, r);
    const uint32_t strip = (uint32_t)((r - pl.bound[b]) / pl.strip_rows[b]);
    key[r] = ((unsigned long long)(((uint32_t)b << 20) | strip) << 32) | float_key(pk[row[r]].y);
}

struct ReorderBufs {
    unsigned long long *key64_in, *key64_out;
    uint32_t *row_a, *row_b, *key32_in, *key32_out;   // the final order is in row_b: row_b[s] = old row that moves to row s
    void* temp;
    size_t temp_bytes;
};

size_t reorder_temp_bytes(int N) {
    size_t b32 = 0, b64 = 0;
    uint32_t* p = nullptr;
    unsigned long long* q = nullptr;
    rocprim::radix_sort_pairs(nullptr, b32, p, p, p, p, (size_t)N, 0, 32, nullptr);
    rocprim::radix_sort_pairs(nullptr, b64, q, q, p, p, (size_t)N, 0, 64, nullptr);
    return b32 > b64 ? b32 : b64;
}

// x sort, then (column, y), (block, x), (block, strip, y): four radix sorts; the final order is in row_b
hipError_t launch_resort_blocks(const float4* pk, int N, const BlockPlan& pl, const ReorderBufs& b, hipStream_t st) {
    const dim3 grid((N + 255) / 256), block(256);
    hipLaunchKernelGGL(sfm_x_keys_kernel, grid, block, 0, st, pk, N, b.key32_in, b.row_b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t bytes = b.temp_bytes;
    e = rocprim::radix_sort_pairs(b.temp, bytes, b.key32_in, b.key32_out, b.row_b, b.row_a, (size_t)N, 0, 32, st);      // -> row_a
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sfm_column_keys_kernel, grid, block, 0, st, pk, N, pl, b.row_a, b.key64_in);
    bytes = b.temp_bytes;
    e = rocprim::radix_sort_pairs(b.temp, bytes, b.key64_in, b.key64_out, b.row_a, b.row_b, (size_t)N, 0, 40, st);        // -> row_b
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sfm_block_keys_kernel, grid, block, 0, st, pk, N, pl, b.row_b, b.key64_in);
    bytes = b.temp_bytes;
    e = rocprim::radix_sort_pairs(b.temp, bytes, b.key64_in, b.key64_out, b.row_b, b.row_a, (size_t)N, 0, 40, st);        // -> row_a
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sfm_block_strip_keys_kernel, grid, block, 0, st, pk, N, pl, b.row_a, b.key64_in);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    bytes = b.temp_bytes;
    return rocprim::radix_sort_pairs(b.temp, bytes, b.key64_in, b.key64_out, b.row_a, b.row_b, (size_t)N, 0, 64, st);     // -> row_b
}

hipError_t launch_resort(const float4* pk, int N, int strip_rows, const ReorderBufs& b, hipStream_t st) {
    const dim3 grid((N + 255) / 256), block(256);
    hipLaunchKernelGGL(sfm_x_keys_kernel, grid, block, 0, st, pk, N, b.key32_in, b.row_b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t bytes = b.temp_bytes;
    e = rocprim::radix_sort_pairs(b.temp, bytes, b.key32_in, b.key32_out, b.row_b, b.row_a, (size_t)N, 0, 32, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sfm_strip_keys_kernel, grid, block, 0, st, pk, N, strip_rows, b.row_a, b.key64_in);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    int strip_bits = 1;
    while (((N + strip_rows - 1) / strip_rows) >> strip_bits) ++strip_bits;
    bytes = b.temp_bytes;
    return rocprim::radix_sort_pairs(b.temp, bytes, b.key64_in, b.key64_out, b.row_a, b.row_b, (size_t)N, 0, 32 + strip_bits, st);
}

hipError_t launch_gather(const uint32_t* src, int N, const float4* pk_in, float4* pk_out, const float2* zv_in, float2* zv_out,
                         const float4* own_in, float4* own_out, const float* rad_in, float* rad_out, const uint8_t* cr_in,
                         uint8_t* cr_out, const uint32_t* dr_in, uint32_t* dr_out, const uint32_t* id_in, uint32_t* id_out,
                         hipStream_t st) {
    hipLaunchKernelGGL(sfm_gather_rows_kernel, dim3((N + 255) / 256), dim3(256), 0, st, src, N, pk_in, pk_out, zv_in, zv_out, own_in,
                       own_out, rad_in, rad_out, cr_in, cr_out, dr_in, dr_out, id_in, id_out);
    return hipGetLastError();
}

}  // namespace sfm
