// sfm_kernels.hip -- the fused Social-Force-Model tick for gfx950 (MI355X), hand-written HIP.
//
// One launch = one tick of PedestrianSimulation.tick's numeric part (pedestrian_simulation.py:81-83):
//   F_i = acceleration + pedestrian (N x N Moussaid) + border + static + dynamic obstacle forces,
//   v'  = cap(v + dt*F, 1.3*v_target),  optionally x' = x + dt*v' and arrival -> next waypoint.
//
// Mapping (wave = 64 lanes, 4 waves per workgroup):
//   * a WAVE owns IPW consecutive pedestrians i; their {x,y,vx,vy} live in SGPRs (wave-uniform);
//   * the 64 LANES span j: the workgroup stages tiles of 256 packed records {x,y,vx,vy} (16 B, one
//     coalesced float4 per thread) from HBM/L2 into LDS, double-buffered, one barrier per tile; every wave
//     reads its lane's record with one conflict-free ds_read_b128 per 64*IPW pairs;
//   * per-lane partial force sums live in VGPRs and are reduced across the wave with shuffles once per tick
//     (deterministic order, no atomics);  the same wave then runs the border / obstacle culls and
//     nearest-point scans (lanes over polylines, then over points) and the O(1) epilogue for its
//     pedestrians, so F never touches HBM.
// No MFMA: the pair body is ~60 dependent VALU ops with 5 transcendentals, not a contraction.
#include "sfm_device.h"

namespace sfm {

__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float uniform(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// atan2(s, c) for (s, c) not both zero; minimax odd polynomial of degree 15 on [0,1]
// (fit error 8.9e-8), octant fix-up by selects.  GUARD handles (0,0) -> 0 like np.arctan2.
template <bool GUARD>
__device__ __forceinline__ float atan2_poly(float s, float c) {
    const float ax = fabsf(c), ay = fabsf(s);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float r = mn * rcp(mx);
    if (GUARD) r = (mx == 0.0f) ? 0.0f : r;
    const float z = r * r;
    float p = -0.00478021258342167f;
    p = fmaf(p, z, 0.02455628334061523f);
    p = fmaf(p, z, -0.0599035729461461f);
    p = fmaf(p, z, 0.09942682328546971f);
    p = fmaf(p, z, -0.1402939336508994f);
    p = fmaf(p, z, 0.1997137085401642f);
    p = fmaf(p, z, -0.33332093252900674f);
    p = fmaf(p, z, 0.9999999113883665f);
    float phi = p * r;
    phi = (ay > ax) ? (1.57079632679489662f - phi) : phi;
    phi = (c < 0.0f) ? (3.14159265358979324f - phi) : phi;
    return copysignf(phi, s);
}

// One Moussaid interaction (forces.py:85-115 / :241-270) without the common factor -A:
//   (dx,dy,dz) = other - self, (dvx,dvy,dvz) = v_self - v_other, rsum = radii to subtract.
// Adds e1*t + g*n to (gx,gy,gz), n = (-t_y, t_x, 0).
// EXACT reproduces the reference's zero-vector conventions (stateutils.normalize's divide-by-1,
// np.arctan2(0,0) = 0, -0/0 = NaN); the fast form differs from it only for coincident pairs, which the
// caller detects through `rinv_out` and recomputes.
template <bool Z3, bool RAD, bool EXACT>
__device__ __forceinline__ void moussaid(const IxConst& c, float dx, float dy, float dz, float dvx, float dvy,
                                         float dvz, float rsum, float& gx, float& gy, float& gz,
                                         float& rinv_out) {
    const float d2 = fmaf(dx, dx, fmaf(dy, dy, Z3 ? fmaf(dz, dz, TINY) : TINY));
    const float rinv = rsq(d2);
    rinv_out = rinv;
    float d = d2 * rinv;
    const float ex = dx * rinv, ey = dy * rinv;
    const float ez = Z3 ? dz * rinv : 0.0f;
    const float Dx = fmaf(c.lam, dvx, ex), Dy = fmaf(c.lam, dvy, ey);
    const float Dz = Z3 ? fmaf(c.lam, dvz, ez) : 0.0f;
    const float D2 = fmaf(Dx, Dx, fmaf(Dy, Dy, Z3 ? fmaf(Dz, Dz, TINY) : TINY));
    const float rD = rsq(D2);
    float Dn = D2 * rD;                        // |D|
    const float tx = Dx * rD, ty = Dy * rD;
    const float tz = Z3 ? Dz * rD : 0.0f;
    float sn, cs, aL;
    if (EXACT) {
        const bool e_flat = (dx == 0.0f) & (dy == 0.0f);             // xy part of e is the zero vector
        const bool t_flat = (Dx == 0.0f) & (Dy == 0.0f);
        const bool coincident = e_flat & (Z3 ? (dz == 0.0f) : true);
        const bool d_zero = t_flat & (Z3 ? (Dz == 0.0f) : true);
        const float exa = e_flat ? 1.0f : ex;                         // arctan2(0,0) = 0 = angle of (1,0)
        const float txa = t_flat ? 1.0f : tx;
        sn = fmaf(txa, ey, -(ty * exa));
        cs = fmaf(txa, exa, ty * ey);
        d = coincident ? 0.0f : d;
        Dn = d_zero ? 0.0f : Dn;
        const float deff = RAD ? d - rsum : d;
        aL = deff * c.c1 * (1.0f / Dn);                               // -d/B*log2e; B = 0 -> -inf or NaN
    } else {
        sn = fmaf(tx, ey, -(ty * ex));                                // sin / cos of angle(e) - angle(t)
        cs = fmaf(tx, ex, ty * ey);
        const float deff = RAD ? d - rsum : d;
        aL = deff * (rD * c.c1);
    }
    const float ang = atan2_poly<EXACT>(sn, cs);                      // == wrapped atan2 difference (stateutils.py:104-112)
    const float theta = fmaf(-c.eg, Dn, ang);                         // forces.py:101
    const float q = Dn * theta;
    const float q2 = q * q;
    const float e1 = ex2(fmaf(q2, c.k1, aL));                         // exp(-d/B - (n' B theta)^2)
    const float e2 = ex2(fmaf(q2, c.k2, aL));                         // exp(-d/B - (n  B theta)^2)
    float g = copysignf(e2, theta);                                   // sign(theta) * e2
    if (EXACT) g = (theta == 0.0f) ? 0.0f : g;                        // np.sign(0) = 0 (forces.py:108)
    gx = fmaf(e1, tx, gx);
    gx = fmaf(-g, ty, gx);
    gy = fmaf(e1, ty, gy);
    gy = fmaf(g, tx, gy);
    if (Z3) gz = fmaf(e1, tz, gz);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// First nearest sampled point of polyline [o0,o1) to (x,y): np.argmin's first-minimum rule
// (forces.py:154,228).  Wave-cooperative; returns the (uniform) point index.
__device__ __forceinline__ int wave_nearest(const float2* __restrict__ pts, int o0, int o1, float x, float y,
                                            int lane) {
    float bd = __builtin_inff();
    int bi = 0x7fffffff;
    for (int p = o0 + lane; p < o1; p += WAVE) {
        const float2 q = pts[p];
        const float ddx = x - q.x, ddy = y - q.y;
        const float dd = fmaf(ddx, ddx, ddy * ddy);
        if (dd < bd) { bd = dd; bi = p; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float od = __shfl_xor(bd, m);
        const int oi = __shfl_xor(bi, m);
        const bool take = (od < bd) | ((od == bd) & (oi < bi));
        bd = take ? od : bd;
        bi = take ? oi : bi;
    }
    return uniform(bi);
}

__device__ __forceinline__ uint32_t mix32(uint32_t a) {   // lowbias32
    a ^= a >> 16; a *= 0x7FEB352Du; a ^= a >> 15; a *= 0x846CA68Bu; a ^= a >> 16;
    return a;
}
__device__ __forceinline__ float waypoint_coord(uint32_t seed, uint32_t ped, uint32_t draw, uint32_t c, float side) {
    const uint32_t h = mix32(seed ^ mix32(2u * ped + c + 0x9E3779B9u * draw));
    return (float)(h >> 8) * 5.9604644775390625e-08f * side;   // 2^-24
}

// BorderForce._get_force for one pedestrian (forces.py:145-167), wave-cooperative.
template <bool RAD>
__device__ __forceinline__ void border_force(const TickArgs& a, float xi, float yi, float ri, int lane,
                                             float& fx, float& fy) {
    const Geo& g = a.borders;
    float ax = 0.0f, ay = 0.0f, spx = 0.0f, spy = 0.0f;
    int cnt = 0;
    auto flush = [&]() {
        if (lane < cnt) {
            const float ddx = xi - spx, ddy = yi - spy;
            const float d = sqrtf(fmaf(ddx, ddx, ddy * ddy));
            const float inv = (d == 0.0f) ? 1.0f : 1.0f / d;               // stateutils.normalize zero guard
            const float dist = RAD ? d - ri : d;                            // forces.py:160-161
            const float mag = a.border_a * ex2(dist * a.border_nlb);        // a*exp(-dist/b), :163
            ax = fmaf(ddx * inv, mag, ax);
            ay = fmaf(ddy * inv, mag, ay);
        }
        cnt = 0;
    };
    for (int kb = 0; kb < g.K; kb += WAVE) {
        const int k = kb + lane;
        bool keep = false;
        if (k < g.K) {
            const float4 c = g.ctr[k];
            const float ddx = xi - c.x, ddy = yi - c.y;
            keep = fmaf(ddx, ddx, ddy * ddy) < c.z;                         // |x - center| < section_length, strict (:149-150)
        }
        unsigned long long m = __ballot(keep);
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int kk = kb + b;
            const int o0 = g.off[kk], o1 = g.off[kk + 1];
            if (o1 <= o0) continue;
            const int bi = wave_nearest(g.pts, o0, o1, xi, yi, lane);
            const float2 p = g.pts[bi];
            if (lane == cnt) { spx = p.x; spy = p.y; }
            if (++cnt == WAVE) flush();
        }
    }
    flush();
    fx = wave_sum(ax);
    fy = wave_sum(ay);
}

// ObstacleForce._get_force for one pedestrian (forces.py:217-275), wave-cooperative.
template <bool RAD>
__device__ __forceinline__ void obstacle_force(const Geo& g, const IxConst& c, bool moving, float xi, float yi,
                                               float vxi, float vyi, float ri, int lane, float& fx, float& fy) {
    float gx = 0.0f, gy = 0.0f, gz = 0.0f, spx = 0.0f, spy = 0.0f, svx = 0.0f, svy = 0.0f;
    int cnt = 0;
    auto flush = [&]() {
        if (lane < cnt) {
            float unused;
            moussaid<false, RAD, true>(c, spx - xi, spy - yi, 0.0f, vxi - svx, vyi - svy, 0.0f, ri, gx, gy, gz,
                                       unused);
        }
        cnt = 0;
    };
    for (int kb = 0; kb < g.K; kb += WAVE) {
        const int k = kb + lane;
        bool keep = false;
        float4 ck = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < g.K) {
            ck = g.ctr[k];
            const float ddx = xi - ck.x, ddy = yi - ck.y;
            keep = fmaf(ddx, ddx, ddy * ddy) < c.thr2;                      // |x - c_k| < perception_threshold (:222-223)
        }
        unsigned long long m = __ballot(keep);
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int kk = kb + b;
            const int o0 = g.off[kk], o1 = g.off[kk + 1];
            if (o1 <= o0) continue;
            const int bi = wave_nearest(g.pts, o0, o1, xi, yi, lane);
            const float2 p = g.pts[bi];
            const float ovx = moving ? __shfl(ck.z, b) : 0.0f;              // static obstacles: v = 0 (:212-213)
            const float ovy = moving ? __shfl(ck.w, b) : 0.0f;
            if (lane == cnt) { spx = p.x; spy = p.y; svx = ovx; svy = ovy; }
            if (++cnt == WAVE) flush();
        }
    }
    flush();
    fx = c.negA * wave_sum(gx);
    fy = c.negA * wave_sum(gy);
}

// ------------------------------------------------------------------------------------------------------
// the fused tick
// ------------------------------------------------------------------------------------------------------
template <int IPW, bool Z3, bool RAD>
__global__ __launch_bounds__(BLOCK) void sfm_tick_kernel(const TickArgs a) {
    __shared__ __attribute__((aligned(16))) float4 s_pk[2][TILE_J];
    __shared__ __attribute__((aligned(16))) float2 s_zv[Z3 ? 2 : 1][Z3 ? TILE_J : 1];
    __shared__ float s_rad[RAD ? 2 : 1][RAD ? TILE_J : 1];

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = uniform(tid >> 6);
    const int ibase = a.i_begin + (blockIdx.x * WAVES_PER_BLOCK + wave) * IPW;   // uniform
    const int N = a.N;

    // own rows -> SGPRs (clamped so inactive tail slots read a valid row; they never store)
    float xi[IPW], yi[IPW], zi[IPW], vxi[IPW], vyi[IPW], vzi[IPW], ri[IPW];
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
        const int i = min(ibase + k, a.i_end - 1);
        const float4 s = a.pk_cur[i];
        xi[k] = uniform(s.x); yi[k] = uniform(s.y); vxi[k] = uniform(s.z); vyi[k] = uniform(s.w);
        zi[k] = 0.0f; vzi[k] = 0.0f; ri[k] = 0.0f;
        if (Z3) { const float2 zz = a.zv_cur[i]; zi[k] = uniform(zz.x); vzi[k] = uniform(zz.y); }
        if (RAD) ri[k] = uniform(a.radius[i]);
    }

    float gx[IPW], gy[IPW], gz[IPW];
#pragma unroll
    for (int k = 0; k < IPW; ++k) { gx[k] = 0.0f; gy[k] = 0.0f; gz[k] = 0.0f; }

    if (a.en_ped && N > 1) {
        float flag = 0.0f;                       // max rsq(d2) over valid pairs: coincidence detector
        const int ntiles = (N + TILE_J - 1) / TILE_J;
        float4 r_pk = a.pk_cur[tid];
        float2 r_zv = make_float2(0.f, 0.f);
        float r_rad = 0.0f;
        if (Z3) r_zv = a.zv_cur[tid];
        if (RAD) r_rad = a.radius[tid];
        s_pk[0][tid] = r_pk;
        if (Z3) s_zv[0][tid] = r_zv;
        if (RAD) s_rad[0][tid] = r_rad;
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            const int buf = t & 1;
            if (t + 1 < ntiles) {                // prefetch the next tile while this one is consumed
                r_pk = a.pk_cur[(t + 1) * TILE_J + tid];
                if (Z3) r_zv = a.zv_cur[(t + 1) * TILE_J + tid];
                if (RAD) r_rad = a.radius[(t + 1) * TILE_J + tid];
            }
#pragma unroll 1
            for (int s = 0; s < TILE_J / WAVE; ++s) {
                const int j0 = t * TILE_J + s * WAVE;                    // uniform
                if (j0 >= N) break;
                const float4 pj = s_pk[buf][s * WAVE + lane];
                float zj = 0.0f, vzj = 0.0f, rj = 0.0f;
                if (Z3) { const float2 zz = s_zv[buf][s * WAVE + lane]; zj = zz.x; vzj = zz.y; }
                if (RAD) rj = s_rad[buf][s * WAVE + lane];
                const bool clean = (j0 + WAVE <= N) && (ibase + IPW <= j0 || ibase >= j0 + WAVE);   // uniform
                if (clean) {
#pragma unroll
                    for (int k = 0; k < IPW; ++k) {
                        float rinv;
                        moussaid<Z3, RAD, Z3>(a.ped, pj.x - xi[k], pj.y - yi[k], zj - zi[k], vxi[k] - pj.z,
                                              vyi[k] - pj.w, vzi[k] - vzj, ri[k] + rj, gx[k], gy[k], gz[k], rinv);
                        flag = fmaxf(flag, rinv);
                    }
                } else {                                                  // tile holds the diagonal or the tail
                    const int j = j0 + lane;
#pragma unroll
                    for (int k = 0; k < IPW; ++k) {
                        float cx = 0.0f, cy = 0.0f, cz = 0.0f, rinv;
                        moussaid<Z3, RAD, Z3>(a.ped, pj.x - xi[k], pj.y - yi[k], zj - zi[k], vxi[k] - pj.z,
                                              vyi[k] - pj.w, vzi[k] - vzj, ri[k] + rj, cx, cy, cz, rinv);
                        const bool valid = (j < N) & (j != ibase + k);   // all_diffs drops j == i (stateutils.py:41-49)
                        gx[k] += valid ? cx : 0.0f;                       // select, so a NaN on the diagonal never leaks
                        gy[k] += valid ? cy : 0.0f;
                        if (Z3) gz[k] += valid ? cz : 0.0f;
                        flag = fmaxf(flag, valid ? rinv : 0.0f);
                    }
                }
            }
            if (t + 1 < ntiles) {
                s_pk[buf ^ 1][tid] = r_pk;
                if (Z3) s_zv[buf ^ 1][tid] = r_zv;
                if (RAD) s_rad[buf ^ 1][tid] = r_rad;
            }
            __syncthreads();
        }
        // 2-D fast path: a coincident valid pair needs the reference's zero-vector conventions -> redo this
        // wave's rows with the exact body, straight from global memory (rare; no workgroup barrier inside).
        if (!Z3) {
            flag = wave_max(flag);
            if (flag >= COINCIDENT_RINV) {
#pragma unroll
                for (int k = 0; k < IPW; ++k) { gx[k] = 0.0f; gy[k] = 0.0f; }
                for (int j0 = 0; j0 < N; j0 += WAVE) {
                    const int j = j0 + lane;
                    const float4 pj = a.pk_cur[min(j, N - 1)];
                    const float rj = RAD ? a.radius[min(j, N - 1)] : 0.0f;
#pragma unroll
                    for (int k = 0; k < IPW; ++k) {
                        float cx = 0.0f, cy = 0.0f, cz = 0.0f, rinv;
                        moussaid<false, RAD, true>(a.ped, pj.x - xi[k], pj.y - yi[k], 0.0f, vxi[k] - pj.z,
                                                   vyi[k] - pj.w, 0.0f, ri[k] + rj, cx, cy, cz, rinv);
                        const bool valid = (j < N) & (j != ibase + k);
                        gx[k] += valid ? cx : 0.0f;
                        gy[k] += valid ? cy : 0.0f;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < IPW; ++k) {
            gx[k] = a.ped.negA * wave_sum(gx[k]);
            gy[k] = a.ped.negA * wave_sum(gy[k]);
            if (Z3) gz[k] = a.ped.negA * wave_sum(gz[k]);
        }
    }

    // ---- per-pedestrian part: geometry forces + epilogue (wave-uniform values, lane 0 stores) ----------
#pragma unroll 1
    for (int k = 0; k < IPW; ++k) {
        const int i = ibase + k;
        if (i >= a.i_end) break;
        // static indexing of the register arrays: pick row k with a chain of selects
        float x = xi[0], y = yi[0], z = zi[0], vx = vxi[0], vy = vyi[0], vz = vzi[0], r = ri[0];
        float fpx = gx[0], fpy = gy[0], fpz = gz[0];
#pragma unroll
        for (int kk = 1; kk < IPW; ++kk) {
            if (k == kk) {
                x = xi[kk]; y = yi[kk]; z = zi[kk]; vx = vxi[kk]; vy = vyi[kk]; vz = vzi[kk]; r = ri[kk];
                fpx = gx[kk]; fpy = gy[kk]; fpz = gz[kk];
            }
        }
        const float4 o = a.own[i];
        float wx = uniform(o.x), wy = uniform(o.y);
        const float ts = uniform(o.z);
        if (!RAD) r = uniform(o.w);

        float fbx = 0.f, fby = 0.f, fsx = 0.f, fsy = 0.f, fdx = 0.f, fdy = 0.f;
        if (a.en_border && a.borders.K > 0 && !(a.crossing && a.crossing[i]))    // forces.py:140-141,176-177
            border_force<RAD>(a, x, y, r, lane, fbx, fby);
        if (a.en_static && a.statics.K > 0)
            obstacle_force<RAD>(a.statics, a.stat, false, x, y, vx, vy, r, lane, fsx, fsy);
        if (a.en_dynamic && a.dynamics.K > 0)
            obstacle_force<RAD>(a.dynamics, a.dyn, true, x, y, vx, vy, r, lane, fdx, fdy);

        // AccelerationForce (forces.py:46-53, stateutils.py:7-15)
        float fax = 0.f, fay = 0.f, faz = 0.f;
        if (a.en_acc) {
            const float tx_ = wx - x, ty_ = wy - y;
            const float nrm = sqrtf(fmaf(tx_, tx_, ty_ * ty_));
            const float inv = (nrm == 0.0f) ? 1.0f : 1.0f / nrm;
            fax = (ts * (tx_ * inv) - vx) * a.inv_tau;
            fay = (ts * (ty_ * inv) - vy) * a.inv_tau;
            faz = (0.0f - vz) * a.inv_tau;
        }
        if (!a.en_ped) { fpx = 0.f; fpy = 0.f; fpz = 0.f; }
        // sum in the dict order acceleration, pedestrian, border, static, dynamic (pedestrian_simulation.py:37-48,81)
        const float Fx = (((fax + fpx) + fbx) + fsx) + fdx;
        const float Fy = (((fay + fpy) + fby) + fsy) + fdy;
        const float Fz = faz + fpz;

        // calculate_new_velocities + cap_velocity (pedestrian_simulation.py:117-124, stateutils.py:18-23)
        float nvx = fmaf(a.dt, Fx, vx), nvy = fmaf(a.dt, Fy, vy), nvz = fmaf(a.dt, Fz, vz);
        float sp = sqrtf(fmaf(nvx, nvx, fmaf(nvy, nvy, nvz * nvz)));
        sp = (sp == 0.0f) ? 1.0f : sp;
        const float fac = fminf(1.0f, (ts * a.max_speed_factor) / sp);
        nvx *= fac; nvy *= fac; nvz *= fac;

        // arrival on the pre-move position -> next waypoint (pedestrian_simulation.py:92-95, run_simulation.py:118-126)
        uint32_t nd = 0;
        bool redraw = false;
        if (a.flags & 2u) {
            const float ax_ = wx - x, ay_ = wy - y;
            if (fmaf(ax_, ax_, ay_ * ay_) < a.arrive_thr2) {
                redraw = true;
                nd = a.draws[i] + 1u;
                wx = waypoint_coord(a.seed, (uint32_t)i, nd, 0u, a.world_side);
                wy = waypoint_coord(a.seed, (uint32_t)i, nd, 1u, a.world_side);
            }
        }
        float nx = x, ny = y, nz = z;
        if (a.flags & 1u) { nx = fmaf(a.dt, nvx, x); ny = fmaf(a.dt, nvy, y); nz = fmaf(a.dt, nvz, z); }

        if (lane == 0) {
            a.pk_next[i] = make_float4(nx, ny, nvx, nvy);
            if (Z3) a.zv_next[i] = make_float2(nz, nvz);
            if (redraw) { a.own[i] = make_float4(wx, wy, o.z, o.w); a.draws[i] = nd; }
            if (a.rec) {
                float* rc = a.rec;
                const size_t n = (size_t)N;
                rc[(0 * 3 + 0) * n + i] = fax; rc[(0 * 3 + 1) * n + i] = fay; rc[(0 * 3 + 2) * n + i] = faz;
                rc[(1 * 3 + 0) * n + i] = fpx; rc[(1 * 3 + 1) * n + i] = fpy; rc[(1 * 3 + 2) * n + i] = fpz;
                rc[(2 * 3 + 0) * n + i] = fbx; rc[(2 * 3 + 1) * n + i] = fby; rc[(2 * 3 + 2) * n + i] = 0.f;
                rc[(3 * 3 + 0) * n + i] = fsx; rc[(3 * 3 + 1) * n + i] = fsy; rc[(3 * 3 + 2) * n + i] = 0.f;
                rc[(4 * 3 + 0) * n + i] = fdx; rc[(4 * 3 + 1) * n + i] = fdy; rc[(4 * 3 + 2) * n + i] = 0.f;
                rc[(5 * 3 + 0) * n + i] = Fx;  rc[(5 * 3 + 1) * n + i] = Fy;  rc[(5 * 3 + 2) * n + i] = Fz;
            }
        }
    }
}

// get_arrived_peds (pedestrian_simulation.py:88-97) on the current device state.
__global__ void sfm_arrived_kernel(const float4* __restrict__ pk, const float4* __restrict__ own, int N, float thr2,
                                   uint8_t* __restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float4 s = pk[i];
    const float4 o = own[i];
    const float ddx = o.x - s.x, ddy = o.y - s.y;
    mask[i] = fmaf(ddx, ddx, ddy * ddy) < thr2 ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------
// launch helpers (host)
// ------------------------------------------------------------------------------------------------------
template <int IPW, bool Z3, bool RAD>
static hipError_t launch_one(const TickArgs& a, hipStream_t st) {
    const int n_local = a.i_end - a.i_begin;
    if (n_local <= 0) return hipSuccess;
    const int per_block = IPW * WAVES_PER_BLOCK;
    const int grid = (n_local + per_block - 1) / per_block;
    hipLaunchKernelGGL((sfm_tick_kernel<IPW, Z3, RAD>), dim3(grid), dim3(BLOCK), 0, st, a);
    return hipGetLastError();
}

template <bool Z3, bool RAD>
static hipError_t launch_ipw(int ipw, const TickArgs& a, hipStream_t st) {
    switch (ipw) {
        case 1: return launch_one<1, Z3, RAD>(a, st);
        case 2: return launch_one<2, Z3, RAD>(a, st);
        case 4: return launch_one<4, Z3, RAD>(a, st);
        default: return launch_one<8, Z3, RAD>(a, st);
    }
}

hipError_t launch_tick(int ipw, bool z3, bool rad, const TickArgs& a, hipStream_t st) {
    if (z3) return rad ? launch_ipw<true, true>(ipw, a, st) : launch_ipw<true, false>(ipw, a, st);
    return rad ? launch_ipw<false, true>(ipw, a, st) : launch_ipw<false, false>(ipw, a, st);
}

hipError_t launch_arrived(const float4* pk, const float4* own, int N, float thr2, uint8_t* mask, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(sfm_arrived_kernel, dim3((N + 255) / 256), dim3(256), 0, st, pk, own, N, thr2, mask);
    return hipGetLastError();
}

}  // namespace sfm
