// sfm_reorder.hip -- periodic spatial re-sort of the pedestrian rows on the device.
//
// The cutoff of sfm_kernels.hip (tiles_negligible) only pays while 64-row tiles are spatially compact.
// sfm_upload_state puts the rows in Hilbert order once; pedestrians then walk ~0.06 m per tick, so a
// device-resident run re-sorts every few dozen ticks: keys (Hilbert-curve index of the 1 m cell) -> stable radix sort of
// (key, row) with rocPRIM -> gather of every per-row array.  A pure function of the state, so it is
// deterministic; despawned pedestrians (parked far away) sort to the end.
#include "sfm_device.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace sfm {

// Position along the Hilbert curve of the 65536 x 65536 grid of 1 m cells.  Unlike the Z-order curve it has no
// jumps: any 64 consecutive pedestrians along it occupy one compact patch, so fixed 64-row tiles have small
// bounding boxes wherever the tile boundaries fall.  (Same routine on the host in sfm_capi.hip.)
__host__ __device__ inline uint32_t hilbert_key(uint32_t x, uint32_t y) {
    uint32_t d = 0;
    for (uint32_t s = 32768u; s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) { x = 65535u - x; y = 65535u - y; }
            const uint32_t t = x; x = y; y = t;
        }
    }
    return d;
}

__global__ void sfm_cell_keys_kernel(const float4* __restrict__ pk, int N, float x0, float y0, uint32_t* __restrict__ key,
                                       uint32_t* __restrict__ row) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const float4 p = pk[s];
    const float fx = p.x - x0, fy = p.y - y0;
    const uint32_t cx = (fx >= 0.f && fx < 65535.f) ? (uint32_t)fx : (fx < 0.f ? 0u : 65535u);
    const uint32_t cy = (fy >= 0.f && fy < 65535.f) ? (uint32_t)fy : (fy < 0.f ? 0u : 65535u);
    key[s] = (fabsf(p.x) < 1.0e14f) ? hilbert_key(cx, cy) : 0xffffffffu;   // parked ghosts last
    row[s] = (uint32_t)s;
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

struct ReorderBufs {
    uint32_t *key_in, *key_out, *row_in, *row_out;
    void* temp;
    size_t temp_bytes;
};

size_t reorder_temp_bytes(int N) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, (size_t)N, 0, 32, nullptr);
    return bytes;
}

// keys -> sort -> row_out[s] = old row that moves to row s
hipError_t launch_resort(const float4* pk, int N, float x0, float y0, const ReorderBufs& b, hipStream_t st) {
    hipLaunchKernelGGL(sfm_cell_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, st, pk, N, x0, y0, b.key_in, b.row_in);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t bytes = b.temp_bytes;
    return rocprim::radix_sort_pairs(b.temp, bytes, b.key_in, b.key_out, b.row_in, b.row_out, (size_t)N, 0, 32, st);
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
