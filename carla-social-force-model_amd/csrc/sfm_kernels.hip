// sfm_kernels.hip -- the Social-Force-Model tick for gfx950 (MI355X), hand-written HIP.  DESIGN.md section 3 is the map.
//
// A tick is the numeric part of PedestrianSimulation.tick (pedestrian_simulation.py:81-83):
//   F_i = acceleration + pedestrian (N x N Moussaid) + border + static + dynamic obstacle forces,
//   v'  = cap(v + dt*F, 1.3*v_target),  optionally x' = x + dt*v' and arrival -> next waypoint.
//
// Kernels, in file order:
//   sfm_geometry_kernel        border / obstacle forces: a workgroup per tile of 64 pedestrians (x slices), find then scan
//   sfm_mode_kernel            pedestrian modes, waypoint queues, gap acceptance (device-resident runs)
//   sfm_tick_kernel            ORDERED tick, fully fused: a wave owns IPW pedestrians (state in SGPRs), the lanes span j; 3-D
//                              crowds, crowds under 256 pedestrians, ragged shards
//   sfm_tile_bounds / strip_bounds / tile_strip_bounds, sfm_pair_list / list2
//                              boxes and the tile-pair list of the "provably < 2^-40 A" cutoff
//   sfm_pair_sym_kernel        SYMMETRIC pair kernel: every unordered pair once, systolic over the wavefront (DPP rotate)
//   sfm_pair_geo_kernel        the same with the tick's geometry workgroups in front, in one launch
//   sfm_sym_epilogue_kernel    column sums of the slab + integration for the symmetric path
//   sfm_fused_tick_kernel      one launch per tick: the epilogue of tick t as the prologue of tick t+1's pair kernel
// No MFMA: the pair body is ~60 dependent VALU ops with 5 transcendentals, not a contraction.  No atomics on forces, fixed
// summation orders: every result is deterministic run to run.
#include "sfm_device.h"

#ifndef SFM_X2
#define SFM_X2 1                       // 1: the fused tick's planar step (no use_ped_radius) evaluates TWO pairs per lane on packed fp32 (A/B, round 4)
#endif
#ifndef SFM_PK
#define SFM_PK 1                       // 1: the fused tick's planar systolic step on packed fp32 instructions (A/B, round 4)
#endif

#include <algorithm>
#include <type_traits>
#include <cstdlib>

namespace sfm {

// launch-shape A/B knobs exist only in a build with -DSFM_EXPERIMENTS (sfm_capi.hip: exp_env)
static inline const char* exp_env(const char* name) {
#ifdef SFM_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

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

// atan2(s, c) for a UNIT vector (s, c) (2-D fast path: both directions are normalised), half-angle form:
//   tan(theta/2) = s / (1 + c)  ->  theta = 2 atan(s / (1 + |c|)) for c >= 0, sign(s)*pi - that for c < 0.
// The ratio is in [-1, 1] for every quadrant, so there is no octant swap, and theta = r * P(r^2) keeps
// full RELATIVE precision for small angles (head-on encounters, where the force is largest).
// 8-coefficient minimax fit of 2*atan(r)/r on [0,1], relative error 8.9e-8 (below the 2^-22 rad resolution of
// the angle between two fp32-rounded directions).  (0,0) -> 0.
__device__ __forceinline__ float atan2_unit(float s, float c) {
    const float r = s * rcp(1.0f + fabsf(c));
    const float z = r * r;
    float p = -0.0095607885413262813f;
    p = fmaf(p, z, 0.049113825228842972f);
    p = fmaf(p, z, -0.11980885478692463f);
    p = fmaf(p, z, 0.1988547939908939f);
    p = fmaf(p, z, -0.28058826128196529f);
    p = fmaf(p, z, 0.39942748114880167f);
    p = fmaf(p, z, -0.66664186893326649f);
    p = fmaf(p, z, 1.9999998228145017f);
    const float a = p * r;
    return (c < 0.0f) ? (copysignf(3.14159265358979324f, s) - a) : a;
}

// One Moussaid interaction (forces.py:85-115 / :241-270) without the common factor -A:
//   (dx,dy,dz) = other - self, (dvx,dvy,dvz) = v_self - v_other, rsum = radii to subtract.
// Adds e1*t + g*n to (gx,gy,gz), n = (-t_y, t_x, 0).
// EXACT reproduces the reference's zero-vector conventions (stateutils.normalize's divide-by-1,
// np.arctan2(0,0) = 0, -0/0 = NaN); the fast form differs from it only for coincident pairs, which the
// caller detects through `rinv_out` and recomputes.
template <bool Z3, bool RAD, bool EXACT>
__device__ __forceinline__ void moussaid(const IxConst& c, float dx, float dy, float dz, float dvx,
                                         float dvy, float dvz, float rsum, float& gx, float& gy, float& gz,
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
    // == wrapped atan2 difference (stateutils.py:104-112); planar fast path: (sn, cs) is a unit vector
    const float ang = (EXACT || Z3) ? atan2_poly<EXACT>(sn, cs) : atan2_unit(sn, cs);
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

// The wrapped angle between two planar directions, biased (stateutils.py:104-112, forces.py:94,101), from S = m sin, C = m cos of
// the angle, m > 0 the common scale:  theta = atan2(S, C) - eg Dn.
// Half-angle form: tan(angle / 2) = S / (m + C); with |C| in the denominator the ratio stays in [-1, 1] for every quadrant and
//   angle = 2 atan(r) for C >= 0,  sign(S) pi - 2 atan(r) for C < 0,  r = S / (m + |C|),
// 2 atan(r) = r P(r^2): 8-coefficient minimax fit on [0, 1], relative error 9.9e-8 (below the 2^-22 rad that the angle between two
// fp32-rounded directions resolves), full relative precision for small angles (head-on encounters, where the force is largest).
// Round 4: the C < 0 branch is sign arithmetic instead of v_cmp + v_cndmask (issue cost of that pair on MI355X: 16 cycles, as much as
// 6.7 fmas -- tools/valu_microbench.hip): with s = copysign(1, C) the denominator s (m + |C|) = s m + C hands the ratio the sign that
// the branch would have given 2 atan(r), and what is left of the branch is the constant k = (1 - s) copysign(pi / 2, S) = 0 or
// sign(S) pi, added by the fma that finishes the polynomial.  Same values as the branch (C >= 0: p r + 0; C < 0: sign(S) pi - p |r|..)
// up to the rounding of the denominator; a NaN in m (coincident pair) still comes out as NaN.
__device__ __forceinline__ float half_angle_theta(float S, float C, float m, float eg, float Dn) {
#ifndef SFM_ATAN_TERMS
#define SFM_ATAN_TERMS 8
#endif
#if SFM_FIXUP_BRANCH
    const float r = S * rcp(m + fabsf(C));
#else
    const float sc = copysignf(1.0f, C);
    const float r = S * rcp(fmaf(sc, m, C));
#endif
    const float z = r * r;
#if SFM_ATAN_TERMS == 8
    float p = -0.0095607885413262813f;
    p = fmaf(p, z, 0.049113825228842972f);
    p = fmaf(p, z, -0.11980885478692463f);
    p = fmaf(p, z, 0.1988547939908939f);
    p = fmaf(p, z, -0.28058826128196529f);
    p = fmaf(p, z, 0.39942748114880167f);
    p = fmaf(p, z, -0.66664186893326649f);
    p = fmaf(p, z, 1.9999998228145017f);
#else                                         // 7 coefficients: relative error 6.5e-7 (A/B only)
    float p = 0.015726754441857338f;
    p = fmaf(p, z, -0.07402600347995758f);
    p = fmaf(p, z, 0.16774238646030426f);
    p = fmaf(p, z, -0.26974382996559143f);
    p = fmaf(p, z, 0.39762964844703674f);
    p = fmaf(p, z, -0.6665303111076355f);
    p = fmaf(p, z, 1.999998688697815f);
#endif
#if SFM_FIXUP_BRANCH
    const float a = p * r;
    const float ang = (C < 0.0f) ? (copysignf(3.14159265358979324f, S) - a) : a;
    return fmaf(-eg, Dn, ang);
#else
    const float h = copysignf(1.57079632679489662f, S);
    const float k = fmaf(-eg, Dn, fmaf(-sc, h, h));                    // 0 or sign(S) pi (pi / 2 doubles exactly), minus the bias eps B
    return fmaf(p, r, k);
#endif
}

// The same interaction for the symmetric kernel's planar fast path, arranged for the fewest issued instructions:
//   (dx,dy) = other - self and d2 = dx^2 + dy^2 come from the caller (it has already tested d2 against the reach);
//   (wx,wy) = lambda (v_self - v_other) -- both velocities are pre-multiplied by lambda once per tile, so D = w + e is one
//   fma per component and e itself is never formed: sin / cos of the angle come out scaled by d, S = t x (dx,dy) = d sin,
//   C = t . (dx,dy) = d cos, and the half-angle ratio is S / (d + |C|).
// d2 is NOT padded: a coincident pair gives rsq(0) = inf -> d = NaN -> a NaN term, which the epilogue takes as the
// signal to recompute the tile with the exact body (the reference's conventions for zero vectors, forces.py:97,105).
// CUT: once -d/B is known (two of the five transcendentals in), a step whose 64 exponents are ALL below -41 is dropped: both
// exponentials of every lane are then < 2^-41, the term < 2^-40 A (the same policy as the tile and the reach tests, now with the
// pair's actual |D| instead of a speed bound).  A NaN exponent (coincident pair) never counts as small.  Returns false if dropped.
template <bool RAD, bool CUT>
__device__ __forceinline__ bool moussaid_planar(const IxConst& c, float dx, float dy, float d2, float wx, float wy, float rsum,
                                                float& cx, float& cy) {
    const float rinv = rsq(d2);
    const float d = d2 * rinv;
    const float Dx = fmaf(dx, rinv, wx), Dy = fmaf(dy, rinv, wy);
    const float D2 = fmaf(Dx, Dx, fmaf(Dy, Dy, TINY));
    const float rD = rsq(D2);
    const float deff = RAD ? d - rsum : d;
    const float aL = deff * (rD * c.c1);
    if (CUT && !__any(!(aL <= -41.0f))) return false;
    const float Dn = D2 * rD;                                          // |D|
    const float tx = Dx * rD, ty = Dy * rD;
    const float S = fmaf(tx, dy, -(ty * dx));                          // d sin(angle(e) - angle(t))
    const float C = fmaf(tx, dx, ty * dy);                             // d cos
    const float theta = half_angle_theta(S, C, d, c.eg, Dn);          // forces.py:94,101
    const float q = Dn * theta;
    const float q2 = q * q;
    const float e1 = ex2(fmaf(q2, c.k1, aL));
    const float e2 = ex2(fmaf(q2, c.k2, aL));
    const float g = copysignf(e2, theta);
    cx = fmaf(e1, tx, -(g * ty));
    cy = fmaf(e1, ty, g * tx);
    return true;
}

// The planar body on PACKED fp32 instructions (round 4).  v_pk_add / v_pk_mul / v_pk_fma_f32 issue in 4.3 cycles per wave on MI355X
// against 2 x 2.4 for the two scalar instructions they replace (tools/valu_microbench.hip), and the interaction is full of x / y pairs:
// d = p_j - p_i, w = lambda (v_i - v_j), D = d / |d| + w, t = D / |D|, (S, C) = (t x d, t . d), the two exponents, the term (c_x, c_y)
// and the resident sums.  Ten packed instructions take the place of twenty; swizzles and broadcasts are op_sel bits.  The two
// products whose halves differ in SIGN -- (S, C) and (-g t_y, g t_x) -- are inline asm: the compiler does not fold a half-negated
// operand into neg_lo and would build the vector with two more instructions.
// Same operations on the same operands as moussaid_planar<RAD, false>, hence the same bits.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f bcast(float v) { v2f r; r.x = v; r.y = v; return r; }
// CUT (the list-cutoff workloads' pair kernel): the two wave-uniform tests of moussaid_planar<RAD, true> and its caller -- all 64 pairs
// beyond reach2, all 64 exponents below -41 -- return false with the term not evaluated.
template <bool RAD, bool CUT>
__device__ __forceinline__ bool moussaid_planar_pk(const IxConst& c, v2f pj, v2f uj, v2f Ti, v2f Ui, float rsum, float reach2, v2f& term) {
    const v2f dd = pj - Ti;                                            // (dx, dy) = other - self
    const float d2 = fmaf(dd.x, dd.x, dd.y * dd.y);
    if (CUT && !__any(!(d2 > reach2))) return false;
    v2f sq, rr;                                                        // (d2, D2) and (1/d, 1/|D|) as register pairs: their product is (d, |D|)
    sq.x = d2;
    rr.x = rsq(d2);
    const v2f Dv = __builtin_elementwise_fma(dd, __builtin_shufflevector(rr, rr, 0, 0), Ui - uj);   // D = e + lambda (v_i - v_j)
    const float D2 = fmaf(Dv.x, Dv.x, fmaf(Dv.y, Dv.y, TINY));
    sq.y = D2;
    rr.y = rsq(D2);
    const float rD = rr.y;
    float aL = 0.f;
    if (CUT) {                                                         // the exponent is needed now: -d / B log2 e
        const float d_ = d2 * rr.x;
        aL = (RAD ? d_ - rsum : d_) * (rD * c.c1);
        if (!__any(!(aL <= -41.0f))) return false;
    }
    v2f t;                                                             // t = D / |D|  (asm: the compiler would copy 1/|D| out of its pair first.
    // The s_nop is the wait state a VALU instruction needs before it reads a transcendental's result on gfx950: the compiler puts it
    // in front of its own instructions and does not look inside an asm statement -- without it t was computed from a stale 1/|D|)
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "v"(Dv), "v"(rr));
    const v2f A = bcast(t.x) * __builtin_shufflevector(dd, dd, 1, 0);  // (tx dy, tx dx)
    v2f SC;                                                            // (S, C) = (tx dy - ty dx, tx dx + ty dy)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0]" : "=v"(SC) : "v"(t), "v"(dd), "v"(A));
    const v2f dDn = sq * rr;                                           // (d, |D|): neither is needed before here
    const float theta = half_angle_theta(SC.x, SC.y, dDn.x, c.eg, dDn.y);   // forces.py:94,101
    float q;
    if (CUT) {
        q = dDn.y * theta;
    } else if (RAD) {
        aL = (dDn.x - rsum) * (rD * c.c1);
        q = dDn.y * theta;
    } else {
        v2f X;                                                         // (-d / B log2 e, |D| theta) = (d, |D|) * (c1 / |D|, theta)
        X.x = rD * c.c1; X.y = theta;
        const v2f aq = dDn * X;
        aL = aq.x; q = aq.y;
    }
    const float q2 = q * q;
    v2f kk; kk.x = c.k1; kk.y = c.k2;
    const v2f arg = __builtin_elementwise_fma(bcast(q2), kk, bcast(aL));
    const float e1 = ex2(arg.x);
    const float e2 = ex2(arg.y);
    const float g = copysignf(e2, theta);
    v2f gv; gv.x = g;                                                  // (the high half is never read: op_sel picks the low one twice)
    v2f B;                                                             // (-g ty, g tx)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]" : "=v"(B) : "v"(gv), "v"(t));
    term = __builtin_elementwise_fma(bcast(e1), t, B);                 // (e1 tx - g ty, e1 ty + g tx)
    return true;
}

// TWO pairs per lane (round 4, the fused tick's planar step without use_ped_radius): the lane's resident pedestrian against two
// travelling ones at once, every quantity a register pair {pair A, pair B}, so that ALL of the body's full-rate arithmetic issues as
// packed instructions -- 41 v_pk_*_f32, 6 v_bfi, 10 transcendentals and 4 v_add_f32_dpp per TWO pairs instead of 2 x 41 instructions
// (the x / y packing above only reaches the 20 operations that come in x / y pairs).  Same operations on the same operands per pair
// as moussaid_planar<false, false>; only the order in which a pedestrian's terms are summed changes (two travelling chains per wave).
__device__ __forceinline__ v2f rsq2(v2f v) { v2f r; r.x = rsq(v.x); r.y = rsq(v.y); return r; }
__device__ __forceinline__ v2f rcp2(v2f v) { v2f r; r.x = rcp(v.x); r.y = rcp(v.y); return r; }
__device__ __forceinline__ v2f ex22(v2f v) { v2f r; r.x = ex2(v.x); r.y = ex2(v.y); return r; }
__device__ __forceinline__ v2f copysign2(v2f m, v2f sg) { v2f r; r.x = copysignf(m.x, sg.x); r.y = copysignf(m.y, sg.y); return r; }
__device__ __forceinline__ v2f fma2(v2f a_, v2f b_, v2f c_) { return __builtin_elementwise_fma(a_, b_, c_); }
template <bool RAD = false>
__device__ __forceinline__ void moussaid_planar_x2(const IxConst& c, float pjx, float pjy, float ujx, float ujy, v2f X, v2f Y, v2f U, v2f V,
                                                   v2f& cx, v2f& cy, v2f rsum = v2f{0.f, 0.f}) {
    const v2f dx = bcast(pjx) - X, dy = bcast(pjy) - Y;                // other - self, pairs A and B
    const v2f d2 = fma2(dx, dx, dy * dy);
    const v2f rinv = rsq2(d2);
    const v2f d = d2 * rinv;
    const v2f Dx = fma2(dx, rinv, U - bcast(ujx)), Dy = fma2(dy, rinv, V - bcast(ujy));     // D = e + lambda (v_i - v_j)
    const v2f D2 = fma2(Dx, Dx, fma2(Dy, Dy, bcast(TINY)));
    const v2f rD = rsq2(D2);
    const v2f aL = (RAD ? d - rsum : d) * (rD * bcast(c.c1));      // forces.py:80-81: the radii come off the distance
    const v2f Dn = D2 * rD;                                            // |D|
    const v2f tx = Dx * rD, ty = Dy * rD;
    const v2f S = fma2(tx, dy, -(ty * dx)), C = fma2(tx, dx, ty * dy); // d sin, d cos of angle(e) - angle(t)
    // half_angle_theta, both pairs (forces.py:94,101)
    const v2f sc = copysign2(bcast(1.0f), C);
    const v2f r = S * rcp2(fma2(sc, d, C));
    const v2f z = r * r;
    v2f p = bcast(-0.0095607885413262813f);
    p = fma2(p, z, bcast(0.049113825228842972f));
    p = fma2(p, z, bcast(-0.11980885478692463f));
    p = fma2(p, z, bcast(0.1988547939908939f));
    p = fma2(p, z, bcast(-0.28058826128196529f));
    p = fma2(p, z, bcast(0.39942748114880167f));
    p = fma2(p, z, bcast(-0.66664186893326649f));
    p = fma2(p, z, bcast(1.9999998228145017f));
    const v2f h = copysign2(bcast(1.57079632679489662f), S);
    const v2f theta = fma2(p, r, fma2(bcast(-c.eg), Dn, fma2(-sc, h, h)));
    const v2f q = Dn * theta;
    const v2f q2 = q * q;
    const v2f e1 = ex22(fma2(q2, bcast(c.k1), aL)), e2 = ex22(fma2(q2, bcast(c.k2), aL));
    const v2f g = copysign2(e2, theta);
    cx = fma2(e1, tx, -(g * ty));
    cy = fma2(e1, ty, g * tx);
}

// The same for a 3-D crowd (round 3; pedestrian_state.py:17-19 keeps 3-component positions and velocities and forces.py:75-117
// takes 3-component norms): e, D and t are 3-vectors, the force is f_v t + f_theta n with n = (-t_y, t_x, 0), and the angle is the
// one between the xy-projections of e and t (stateutils.angle_diff_2d uses components 0 and 1 only) -- neither projection is a unit
// vector, so the half-angle ratio is S / (h + |C|) with h = sqrt(S^2 + C^2) (one more rsq).  d2 is the 3-D squared distance.
// Two pedestrians above one another (e_xy = 0) or a vertical D give h = 0 -> rsq(0) = inf -> NaN, the signal for the exact body
// (np.arctan2(0, 0) = 0 there), like a coincident pair in the planar body.
template <bool RAD, bool CUT>
__device__ __forceinline__ bool moussaid_spatial(const IxConst& c, float dx, float dy, float dz, float d2, float wx, float wy, float wz,
                                                 float rsum, float& cx, float& cy, float& cz) {
    const float rinv = rsq(d2);
    const float d = d2 * rinv;
    const float Dx = fmaf(dx, rinv, wx), Dy = fmaf(dy, rinv, wy), Dz = fmaf(dz, rinv, wz);
    const float D2 = fmaf(Dx, Dx, fmaf(Dy, Dy, fmaf(Dz, Dz, TINY)));
    const float rD = rsq(D2);
    const float deff = RAD ? d - rsum : d;
    const float aL = deff * (rD * c.c1);
    if (CUT && !__any(!(aL <= -41.0f))) return false;
    const float Dn = D2 * rD;                                          // |D|
    const float tx = Dx * rD, ty = Dy * rD, tz = Dz * rD;
    const float S = fmaf(tx, dy, -(ty * dx));                          // |t_xy| |d_xy| sin(angle(e_xy) - angle(t_xy))
    const float C = fmaf(tx, dx, ty * dy);                             // ... cos
    const float h2 = fmaf(S, S, C * C);
    const float h = h2 * rsq(h2);
    const float theta = half_angle_theta(S, C, h, c.eg, Dn);          // forces.py:94,101
    const float q = Dn * theta;
    const float q2 = q * q;
    const float e1 = ex2(fmaf(q2, c.k1, aL));
    const float e2 = ex2(fmaf(q2, c.k2, aL));
    const float g = copysignf(e2, theta);
    cx = fmaf(e1, tx, -(g * ty));
    cy = fmaf(e1, ty, g * tx);
    cz = e1 * tz;
    return true;
}

// Two pairs per lane for a 3-D crowd (round 4, late; the fused tick's 8-wave form -- all forces at N = 4096: 24.0 -> 22.5 us): moussaid_spatial's
// operations on register pairs {pair A, pair B}, the same operations on the same operands per pair.
template <bool RAD = false>
__device__ __forceinline__ void moussaid_spatial_x2(const IxConst& c, float pjx, float pjy, float pjz, float ujx, float ujy, float ujz,
                                                    v2f X, v2f Y, v2f Z, v2f U, v2f V, v2f W, v2f& cx, v2f& cy, v2f& cz, v2f rsum = v2f{0.f, 0.f}) {
    const v2f dx = bcast(pjx) - X, dy = bcast(pjy) - Y, dz = bcast(pjz) - Z;          // other - self, pairs A and B
    const v2f d2 = fma2(dx, dx, fma2(dy, dy, dz * dz));
    const v2f rinv = rsq2(d2);
    const v2f d = d2 * rinv;
    const v2f Dx = fma2(dx, rinv, U - bcast(ujx)), Dy = fma2(dy, rinv, V - bcast(ujy)), Dz = fma2(dz, rinv, W - bcast(ujz));
    const v2f D2 = fma2(Dx, Dx, fma2(Dy, Dy, fma2(Dz, Dz, bcast(TINY))));
    const v2f rD = rsq2(D2);
    const v2f aL = (RAD ? d - rsum : d) * (rD * bcast(c.c1));
    const v2f Dn = D2 * rD;                                            // |D|
    const v2f tx = Dx * rD, ty = Dy * rD, tz = Dz * rD;
    const v2f S = fma2(tx, dy, -(ty * dx)), C = fma2(tx, dx, ty * dy); // the xy projections' sin / cos, scaled by |t_xy| |d_xy|
    const v2f h2 = fma2(S, S, C * C);
    const v2f hm = h2 * rsq2(h2);
    // half_angle_theta with m = h, both pairs (forces.py:94,101)
    const v2f sc = copysign2(bcast(1.0f), C);
    const v2f r = S * rcp2(fma2(sc, hm, C));
    const v2f z = r * r;
    v2f p = bcast(-0.0095607885413262813f);
    p = fma2(p, z, bcast(0.049113825228842972f));
    p = fma2(p, z, bcast(-0.11980885478692463f));
    p = fma2(p, z, bcast(0.1988547939908939f));
    p = fma2(p, z, bcast(-0.28058826128196529f));
    p = fma2(p, z, bcast(0.39942748114880167f));
    p = fma2(p, z, bcast(-0.66664186893326649f));
    p = fma2(p, z, bcast(1.9999998228145017f));
    const v2f h = copysign2(bcast(1.57079632679489662f), S);
    const v2f theta = fma2(p, r, fma2(bcast(-c.eg), Dn, fma2(-sc, h, h)));
    const v2f q = Dn * theta;
    const v2f q2 = q * q;
    const v2f e1 = ex22(fma2(q2, bcast(c.k1), aL)), e2 = ex22(fma2(q2, bcast(c.k2), aL));
    const v2f g = copysign2(e2, theta);
    cx = fma2(e1, tx, -(g * ty));
    cy = fma2(e1, ty, g * tx);
    cz = e1 * tz;
}

// The 3-D body with its x / y pairs on packed fp32 instructions (round 4; the z components stay scalar): the drop-in CARLA path is a
// 3-D crowd (pedestrian_state.py:17-19 -- walkers never have exactly equal z), so this is the step real callers run.  Same operations
// on the same operands as moussaid_spatial, hence the same bits.  d2 comes out of here (3-D squared distance), reach2 as in the planar form.
template <bool RAD, bool CUT>
__device__ __forceinline__ bool moussaid_spatial_pk(const IxConst& c, v2f pj, float zj, v2f uj, float ujz, v2f Ti, float zi, v2f Ui, float uiz,
                                                    float rsum, float reach2, v2f& term, float& cz) {
    const v2f dd = pj - Ti;                                            // (dx, dy) = other - self
    const float dz = zj - zi;
    const float d2 = fmaf(dd.x, dd.x, fmaf(dd.y, dd.y, dz * dz));
    if (CUT && !__any(!(d2 > reach2))) return false;
    const float rinv = rsq(d2);
    const float d = d2 * rinv;
    const v2f Dv = __builtin_elementwise_fma(dd, bcast(rinv), Ui - uj);  // D = e + lambda (v_i - v_j)
    const float Dz = fmaf(dz, rinv, uiz - ujz);
    const float D2 = fmaf(Dv.x, Dv.x, fmaf(Dv.y, Dv.y, fmaf(Dz, Dz, TINY)));
    const float rD = rsq(D2);
    const float deff = RAD ? d - rsum : d;
    const float aL = deff * (rD * c.c1);
    if (CUT && !__any(!(aL <= -41.0f))) return false;
    const float Dn = D2 * rD;                                          // |D|
    const v2f t = Dv * bcast(rD);
    const float tz = Dz * rD;
    const v2f A = bcast(t.x) * __builtin_shufflevector(dd, dd, 1, 0);  // (tx dy, tx dx)
    v2f SC;                                                            // (S, C) = (tx dy - ty dx, tx dx + ty dy): |t_xy| |d_xy| (sin, cos)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0]" : "=v"(SC) : "v"(t), "v"(dd), "v"(A));
    const float h2 = fmaf(SC.x, SC.x, SC.y * SC.y);
    const float h = h2 * rsq(h2);
    const float theta = half_angle_theta(SC.x, SC.y, h, c.eg, Dn);    // forces.py:94,101
    const float q = Dn * theta;
    const float q2 = q * q;
    v2f kk; kk.x = c.k1; kk.y = c.k2;
    const v2f arg = __builtin_elementwise_fma(bcast(q2), kk, bcast(aL));
    const float e1 = ex2(arg.x);
    const float e2 = ex2(arg.y);
    const float g = copysignf(e2, theta);
    v2f gv; gv.x = g;
    v2f B;                                                             // (-g ty, g tx)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]" : "=v"(B) : "v"(gv), "v"(t));
    term = __builtin_elementwise_fma(bcast(e1), t, B);                 // (e1 tx - g ty, e1 ty + g tx)
    cz = e1 * tz;
    return true;
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

// One wave moves vehicle k by dt * v and regenerates its ring p = c + R(yaw) u (obstacles.py:297-329 without the simulator).
__device__ __forceinline__ void advance_vehicle(const DynAdvance& d, int k, int lane, bool advance) {
    float4 c = d.ctr[k];
    if (advance) {
        c.x = fmaf(d.dt, c.z, c.x);
        c.y = fmaf(d.dt, c.w, c.y);
        if (lane == 0) (d.ctr_out ? d.ctr_out : d.ctr)[k] = c;
    }
    const float2 r = d.rot[k];                     // {cos yaw, sin yaw}
    const int o1 = d.off[k + 1];
    float2* pts = d.pts_out ? d.pts_out : d.pts;
    for (int p = d.off[k] + lane; p < o1; p += WAVE) {
        const float2 u = d.local[p];
        pts[p] = make_float2(fmaf(r.x, u.x, fmaf(-r.y, u.y, c.x)), fmaf(r.y, u.x, fmaf(r.x, u.y, c.y)));
    }
}

// ------------------------------------------------------------------------------------------------------
// provably negligible tile pairs
// ------------------------------------------------------------------------------------------------------
// |f_ij| <= A (e1 + e2) <= 2 A exp(-d_eff / B),  B = gamma |D|,  |D| = |lambda (v_i - v_j) + e| <= lambda (|v_i| + |v_j|) + 1.
// So if the bounding boxes of two 64-pedestrian tiles are farther apart than
//     gamma (lambda (vmax_a + vmax_b) + 1) * 41 ln 2   (+ 2 r_max with use_ped_radius)
// every one of their 4096 terms is below 2^-40 A and the tile pair is not evaluated.  The bound uses the
// actual speeds of THIS tick (sfm_tile_bounds_kernel), so it holds for any state the caller uploads; a
// crowd in random index order simply has overlapping boxes and nothing is skipped.  The test is symmetric in
// its two tiles bit for bit, so the pair kernel and the epilogue always agree on it.
__device__ __forceinline__ bool tiles_negligible(const float4 ba, const float va, const float4 bb, const float vb,
                                                 const float lam, const float cut_scale, const float cut_pad) {
    const float gx = fmaxf(0.0f, fmaxf(ba.x - bb.z, bb.x - ba.z));
    const float gy = fmaxf(0.0f, fmaxf(ba.y - bb.w, bb.y - ba.w));
    const float reach = fmaf(cut_scale, fmaf(lam, va + vb, 1.0f), cut_pad);
    return fmaf(gx, gx, gy * gy) > reach * reach;
}

// ---- nearest sampled point of a polyline: np.argmin's first-minimum rule (forces.py:154,228) -------------
// One LANE per pedestrian, the polyline [o0,o1) is wave-uniform.  A trip brings 64 points in with one coalesced
// load (the next trip's load is in flight meanwhile), parks them in the wave's LDS row, and every lane walks them
// through broadcast ds_read_b128 (two points each).  The running minimum is kept per GROUP of 8 points (4 ops
// per distance + one min3 tree + one compare per group instead of a compare-select pair per point); at the end
// of the trip a lane whose minimum moved re-reads its winning group from the row and takes the first slot
// whose distance -- recomputed with the same operations, so bit-identical -- equals the minimum.
// Groups in ascending order with a strict `<`, first equal slot inside the group: np.argmin's first-minimum rule.
// Slots past the end hold copies of the last point (they can tie with it, never beat it, and come later).
// LDS operations of one wave execute in order, so the row needs no barrier.
// An empty polyline yields a far-away sentinel whose force term underflows to exactly 0.
__device__ __forceinline__ float dist2(float x, float y, float px, float py) {
    const float ax = x - px, ay = y - py;
    return fmaf(ax, ax, ay * ay);
}
__device__ __forceinline__ float2 lane_nearest(const float2* __restrict__ pts, int o0, int o1, float x, float y,
                                               float2* __restrict__ row, int lane) {
    float bd = __builtin_inff();
    float2 sp = make_float2(3.0e15f, 3.0e15f);
    if (o1 <= o0) return sp;
    const int last = o1 - 1;
    float2 nxt = pts[min(o0 + lane, last)];
    for (int p = o0; p < o1; p += WAVE) {
        row[lane] = nxt;
        if (p + WAVE < o1) nxt = pts[min(p + WAVE + lane, last)];
        __builtin_amdgcn_wave_barrier();
        const int cnt = min(WAVE, o1 - p);
        int bl = -1;
        auto group = [&](int u8) {
            const float4* q = reinterpret_cast<const float4*>(row + u8);
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const float d0 = dist2(x, y, q0.x, q0.y), d1 = dist2(x, y, q0.z, q0.w);
            const float d2 = dist2(x, y, q1.x, q1.y), d3 = dist2(x, y, q1.z, q1.w);
            const float d4 = dist2(x, y, q2.x, q2.y), d5 = dist2(x, y, q2.z, q2.w);
            const float d6 = dist2(x, y, q3.x, q3.y), d7 = dist2(x, y, q3.z, q3.w);
            const float m = fminf(fminf(fminf(fminf(d0, d1), d2), fminf(fminf(d3, d4), d5)), fminf(d6, d7));
            const bool take = m < bd;
            bd = take ? m : bd;
            bl = take ? u8 : bl;
        };
#pragma unroll
        for (int u8 = 0; u8 < WAVE; u8 += 8)
            if (u8 < cnt) group(u8);                                   // uniform
        if (bl >= 0) {                                                 // this lane's minimum moved: which slot?
            const float4* q = reinterpret_cast<const float4*>(row + bl);
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            sp = make_float2(q3.z, q3.w);
            if (dist2(x, y, q3.x, q3.y) == bd) sp = make_float2(q3.x, q3.y);
            if (dist2(x, y, q2.z, q2.w) == bd) sp = make_float2(q2.z, q2.w);
            if (dist2(x, y, q2.x, q2.y) == bd) sp = make_float2(q2.x, q2.y);
            if (dist2(x, y, q1.z, q1.w) == bd) sp = make_float2(q1.z, q1.w);
            if (dist2(x, y, q1.x, q1.y) == bd) sp = make_float2(q1.x, q1.y);
            if (dist2(x, y, q0.z, q0.w) == bd) sp = make_float2(q0.z, q0.w);
            if (dist2(x, y, q0.x, q0.y) == bd) sp = make_float2(q0.x, q0.y);
        }
        __builtin_amdgcn_wave_barrier();
    }
    return sp;
}

__device__ __forceinline__ uint32_t mix32(uint32_t a) {   // lowbias32
    a ^= a >> 16; a *= 0x7FEB352Du; a ^= a >> 15; a *= 0x846CA68Bu; a ^= a >> 16;
    return a;
}
__device__ __forceinline__ float waypoint_coord(uint32_t seed, uint32_t ped, uint32_t draw, uint32_t c, float side) {
    const uint32_t h = mix32(seed ^ mix32(2u * ped + c + 0x9E3779B9u * draw));
    return (float)(h >> 8) * 5.9604644775390625e-08f * side;   // 2^-24
}

// ------------------------------------------------------------------------------------------------------
// geometry forces: border + static + dynamic obstacles, one workgroup per tile of 64 pedestrians
// ------------------------------------------------------------------------------------------------------
// One lane per pedestrian, polylines wave-uniform.  Two phases:
//  1. find.  The polylines are dealt to the 16 waves by index (k mod 16).  A wave tests 64 of its polylines at
//     once, one per lane (all loads of the trip in flight together), against the TILE's bounding box -- a superset
//     of every pedestrian's own test -- and for each survivor, its parameters read out of the owning lane with
//     v_readlane (no memory access), runs the reference's exact per-pedestrian tests lane-parallel (strict `<`,
//     forces.py:149-150,222-223).  A polyline kept by at least one lane becomes a 64-byte item in the wave's LDS
//     list.
//  2. scan.  The items of all waves, in (wave, index) order, are dealt round-robin, so the expensive part balances
//     whatever the find phase produced.  A wave re-derives the keep mask, scans the polyline once for all 64
//     pedestrians (lane_nearest) and evaluates the force term in the lanes that kept it.
// Per-wave partial sums meet in LDS in wave order; every assignment above is a function of the data only, so the
// result is deterministic.  Compact tiles (a crowd in spatial index order) make the survivors few and the lanes
// agree; a scrambled crowd keeps every polyline (list overflow: the finder scans on the spot) and only costs time.
// Depends on the tick's input state only: runs beside the pair kernel on a side stream and leaves
// {fbx,fby,fsx,fsy,fdx,fdy} in geo[6][N_pad].
constexpr int GEO_WAVES = 16;
constexpr int GEO_BLOCK = GEO_WAVES * WAVE;
constexpr int GEO_ITEMS = 32;                  // list capacity per wave

struct GeoItem {
    int o0, o1, kind, pad;                     // kind 0 border, 1 static, 2 dynamic obstacle
    float4 c;                                  // borders {cx, cy, section_length^2, 0}; obstacles {cx, cy, vx, vy}
    float4 s0, s1;                             // borders: chord {ax, ay, abx, aby}, {1/|ab|^2, max deviation, 0, 0}
};

struct GeoLane {                               // one pedestrian
    float x, y, vx, vy, r, skip;
    bool live, walk;
};

__device__ __forceinline__ float readlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float4 readlane(const float4 v, int l) {
    return make_float4(readlane(v.x, l), readlane(v.y, l), readlane(v.z, l), readlane(v.w, l));
}

// the reference's own per-pedestrian cull for one polyline (plus, for borders, the 2^-40 chord bound)
__device__ __forceinline__ bool geo_keep(const TickArgs& a, const GeoLane& me, int kind, const float4 c, const float4 s0,
                                         const float4 s1) {
    const float cdx = me.x - c.x, cdy = me.y - c.y;
    const float d2 = fmaf(cdx, cdx, cdy * cdy);
    if (kind == 0) {
        // A border whose every point is provably farther than 40 ln2 * b (+ radius) contributes less than
        // 2^-40 a: the bound is the distance to the chord first-last point minus the polyline's largest
        // deviation from that chord (exact for the straight config borders of obstacles.py:344-355).
        const float px = me.x - s0.x, py = me.y - s0.y;
        const float tt = fminf(fmaxf(fmaf(px, s0.z, py * s0.w) * s1.x, 0.0f), 1.0f);
        const float qx = fmaf(-tt, s0.z, px), qy = fmaf(-tt, s0.w, py);
        return me.walk & (d2 < c.z) &                                 // |x - center| < section_length, strict (:149-150)
               !(sqrtf(fmaf(qx, qx, qy * qy)) - s1.y > me.skip);
    }
    return me.live & (d2 < (kind == 1 ? a.stat.thr2 : a.dyn.thr2));  // |x - c_k| < perception_threshold (:222-223)
}

// BorderForce._get_force (forces.py:145-167) / ObstacleForce._get_force (forces.py:217-275) for one polyline
template <bool RAD>
__device__ __forceinline__ void geo_item(const TickArgs& a, const GeoLane& me, const GeoItem& it, bool keep, float2* row,
                                         int lane, float (&f)[6]) {
    const Geo& g = it.kind == 0 ? a.borders : it.kind == 1 ? a.statics : a.dynamics;
    float2 sp;
    if (it.kind == 0 && it.s1.z > 0.0f) {
        // A border that sfm_set_borders verified to be a straight, uniformly sampled line (the config-defined borders of
        // obstacles.py:344-355 are np.linspace between two points): the nearest SAMPLED point is the sample next to the foot of
        // the perpendicular, so np.argmin's scan over all P points (forces.py:154) collapses to the five samples around it --
        // evaluated with the same dist2, in ascending order with a strict `<` (first-minimum rule).  The samples either side of
        // the optimum are > spacing^2 farther in d^2, far above the fp32 rounding of d^2 for any pedestrian that keeps the border.
        sp = make_float2(3.0e15f, 3.0e15f);
        if (keep) {
            const float n_seg = it.s1.z;                                           // P - 1
            const float px = me.x - it.s0.x, py = me.y - it.s0.y;
            const float tt = fminf(fmaxf(fmaf(px, it.s0.z, py * it.s0.w) * it.s1.x, 0.0f), 1.0f);
            const int k0 = (int)rintf(tt * n_seg);
            const int lo = max(k0 - 2, 0), hi = min(k0 + 2, it.o1 - it.o0 - 1);
            float bd = __builtin_inff();
            for (int k = lo; k <= hi; ++k) {
                const float2 q = g.pts[it.o0 + k];
                const float d2 = dist2(me.x, me.y, q.x, q.y);
                const bool take = d2 < bd;
                bd = take ? d2 : bd;
                sp = take ? q : sp;
            }
        }
    } else {
        sp = lane_nearest(g.pts, it.o0, it.o1, me.x, me.y, row, lane);
    }
    if (!keep) return;
    if (it.kind == 0) {
        const float ddx = me.x - sp.x, ddy = me.y - sp.y;
        const float d = sqrtf(fmaf(ddx, ddx, ddy * ddy));
        const float inv = (d == 0.0f) ? 1.0f : 1.0f / d;                       // stateutils.normalize zero guard
        const float dist = RAD ? d - me.r : d;                                  // forces.py:160-161
        const float mag = a.border_a * ex2(dist * a.border_nlb);                // a*exp(-dist/b), :163
        f[0] = fmaf(ddx * inv, mag, f[0]);
        f[1] = fmaf(ddy * inv, mag, f[1]);
    } else {
        const bool moving = it.kind == 2;                                       // static obstacles: v = 0 (:212-213)
        const float ovx = moving ? it.c.z : 0.0f, ovy = moving ? it.c.w : 0.0f;
        float gx = 0.f, gy = 0.f, gz = 0.f, unused;
        moussaid<false, RAD, true>(moving ? a.dyn : a.stat, sp.x - me.x, sp.y - me.y, 0.0f, me.vx - ovx, me.vy - ovy, 0.0f,
                                   me.r, gx, gy, gz, unused);
        if (moving) { f[4] += gx; f[5] += gy; } else { f[2] += gx; f[3] += gy; }
    }
}

// The tile's view of the geometry: bounding box of its pedestrians, centre / half diagonal, largest chord skip.
struct GeoTile { float x0, y0, x1, y1, cx, cy, half_diag, skip_max; };

// One pass of a wave over its polylines (dealt round-robin to the n_gwaves = waves x slices of the tile, borders first, the obstacles carrying on where
// the borders ended).  64 at a time, one per lane, they are tested
// against the tile; each survivor's parameters are read out of its lane with v_readlane and the reference's exact
// per-pedestrian test runs lane-parallel.  SCAN = false: the first GEO_ITEMS kept polylines go to the wave's list;
// returns how many were kept.  SCAN = true (list overflow): the kept polylines beyond GEO_ITEMS are scanned on the spot.
template <bool RAD, bool SCAN>
__device__ __forceinline__ int geo_find(const TickArgs& a, const GeoLane& me, const GeoTile& tb, GeoItem* list, float2* row,
                                        int lane, int gwave, int n_gwaves, float (&f)[6], int cap = GEO_ITEMS) {
    int n_found = 0;                                                   // uniform
    int dealt = 0;                                                     // polylines of the kinds before: the deal carries on where they ended
#pragma unroll
    for (int kind = 0; kind < 3; ++kind) {
        const Geo& g = kind == 0 ? a.borders : kind == 1 ? a.statics : a.dynamics;
        if (!(kind == 0 ? a.en_border : kind == 1 ? a.en_static : a.en_dynamic)) continue;
        const float thr2 = kind == 1 ? a.stat.thr2 : a.dyn.thr2;
        // (every kind dealt from wave 0 put all 16 obstacles and all 4 vehicles of c1 -- rings, scanned point by point -- into the
        //  first of the tile's two workgroups: scan 7.2 us there against 2.8 us in the other)
        const int gw = (gwave + n_gwaves - dealt % n_gwaves) % n_gwaves;
        dealt += g.K;
        // Round 4: with many polylines per wave they are dealt in runs of up to 8 neighbours instead of one by one.  A lane's records
        // (16 B centre, 2 x 16 B chord, two offsets) then share cache lines with its neighbours' -- dealt singly, at a stride of
        // n_gwaves records, every one of a trip's five loads touched 64 lines -- while the scan work still spreads over the waves
        // (>= 8 runs per wave before runs are used at all; small crowds keep the one-by-one deal).
        const int run_log = g.K >= 64 * n_gwaves ? 3 : g.K >= 32 * n_gwaves ? 2 : g.K >= 16 * n_gwaves ? 1 : 0;   // (no division: this is on every geometry workgroup's chain)
        for (int base = 0; base < g.K; base += WAVE * n_gwaves) {
            const int k = base + ((((lane >> run_log) * n_gwaves + gw) << run_log) | (lane & ((1 << run_log) - 1)));
            bool near = false;
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f), s0 = c, s1 = c;
            int o0 = 0, o1 = 0;
            if (k < g.K) {
                c = g.ctr[k];
                o0 = g.off[k]; o1 = g.off[k + 1];
                // the box point nearest to the centre is at least as near as any pedestrian of the tile
                const float gx = fmaxf(0.0f, fmaxf(tb.x0 - c.x, c.x - tb.x1)), gy = fmaxf(0.0f, fmaxf(tb.y0 - c.y, c.y - tb.y1));
                near = fmaf(gx, gx, gy * gy) < (kind == 0 ? c.z : thr2);
                if (kind == 0) {
                    s0 = g.seg[2 * k]; s1 = g.seg[2 * k + 1];
                    // no pedestrian is nearer to the chord than the box centre is, minus the half diagonal
                    const float px = tb.cx - s0.x, py = tb.cy - s0.y;
                    const float tt = fminf(fmaxf(fmaf(px, s0.z, py * s0.w) * s1.x, 0.0f), 1.0f);
                    const float qx = fmaf(-tt, s0.z, px), qy = fmaf(-tt, s0.w, py);
                    near &= !(sqrtf(fmaf(qx, qx, qy * qy)) - tb.half_diag - s1.y > tb.skip_max);
                }
            }
            unsigned long long m = __ballot(near);
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                GeoItem it;
                it.o0 = __builtin_amdgcn_readlane(o0, b); it.o1 = __builtin_amdgcn_readlane(o1, b);
                it.kind = kind; it.pad = 0;
                it.c = readlane(c, b);
                it.s0 = kind == 0 ? readlane(s0, b) : make_float4(0.f, 0.f, 0.f, 0.f);
                it.s1 = kind == 0 ? readlane(s1, b) : make_float4(0.f, 0.f, 0.f, 0.f);
                const bool keep = geo_keep(a, me, kind, it.c, it.s0, it.s1);
                if (!__any(keep)) continue;
                if (SCAN) {
                    if (n_found >= cap) geo_item<RAD>(a, me, it, keep, row, lane, f);
                } else if (n_found < GEO_ITEMS && lane == 0) {
                    list[n_found] = it;
                }
                ++n_found;
            }
        }
    }
    return n_found;
}

// Compacts the tile-pair items the symmetric kernel has to evaluate: one candidate per thread, (bx, shift) with
// bx an own tile and tb = bx + shift (mod n_t).  Own-own pairs are kept once (shift <= n_t / 2, as in the kernel's 2-D
// grid); a pair with a tile of another rank is kept by both ranks, each evaluating its own side only (bit 31).
// Diagonal items (shift 0) are always kept and carry two diagonal tiles each.
constexpr uint32_t WORK_ONE_SIDED = 0x80000000u;
// `partners`: PARTNERS_ALL, or only the items whose two tiles are both own (PARTNERS_OWN: what a shard can evaluate before the
// other ranks' rows have arrived), or only the others (PARTNERS_REMOTE).
constexpr int PARTNERS_ALL [[maybe_unused]] = 0, PARTNERS_OWN = 1, PARTNERS_REMOTE = 2;
// Returns whether candidate (bx, shift) is an item, and the item.
__device__ __forceinline__ bool list_candidate(const float4* __restrict__ box, const float* __restrict__ vmax, int n_t, int t_lo, int t_hi,
                                               float lam, float cut_scale, float cut_pad, int partners, int bx, int shift, uint32_t& item) {
    if (bx >= t_hi) return false;
    bool keep;
    uint32_t one = 0u;
    if (shift == 0) {
        keep = (bx - t_lo) < ((t_hi - t_lo + 1) >> 1) && partners != PARTNERS_REMOTE;
    } else {
        int tb = bx + shift;
        if (tb >= n_t) tb -= n_t;
        const bool own = tb >= t_lo && tb < t_hi;
        if ((partners == PARTNERS_OWN && !own) || (partners == PARTNERS_REMOTE && own)) return false;
        if (own) keep = shift <= (n_t >> 1) && !(!(n_t & 1) && shift == (n_t >> 1) && bx >= (n_t >> 1));
        else { keep = true; one = WORK_ONE_SIDED; }
        if (keep) keep = !tiles_negligible(box[bx], vmax[bx], box[tb], vmax[tb], lam, cut_scale, cut_pad);
    }
    item = (uint32_t)bx | ((uint32_t)shift << 16) | one;
    return keep;
}

// Appends the kept items of a whole workgroup to the list with ONE atomic: the counter is a single word, a device-scope atomic
// on it takes ~11 ns whatever else goes on, and a wave-level append (what `work[atomicAdd(count, 1)]` compiles to) made those
// atomics the flat list kernel's whole duration (c4: ~800 waves with an item, 10 us).  Every thread of the workgroup must call it.
// s_n: [waves + 1] ints of LDS.
// where the list kernels leave an item's row pair for the epilogue (SymArgs::idx): item = bx | shift << 16 [| one-sided]
// (position `pos` of the launch's list, `cap` row pairs per list; beyond: 0xffffffff = the pair spills into the overflow sums)
__device__ __forceinline__ void list_index(uint32_t* __restrict__ idx, int n_t, int t_lo, uint32_t item, uint32_t row_base, uint32_t pos, uint32_t cap) {
    if (idx) idx[(size_t)((int)(item & 0xffffu) - t_lo) * (size_t)n_t + ((item >> 16) & 0x7fffu)] = pos < cap ? row_base + pos : 0xffffffffu;
}

__device__ __forceinline__ void list_emit(bool keep, uint32_t item, uint32_t* __restrict__ work, int* __restrict__ count, int* s_n,
                                          uint32_t* __restrict__ idx = nullptr, int n_t = 0, int t_lo = 0, uint32_t row_base = 0u, uint32_t cap = 0u) {
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6, waves = (blockDim.x + WAVE - 1) >> 6;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_n[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < waves; ++w) { const int n = s_n[w]; s_n[w] = total; total += n; }   // exclusive prefix
        const int base = total > 0 ? atomicAdd(count, total) : 0;
        for (int w = 0; w < waves; ++w) s_n[w] += base;
    }
    __syncthreads();
    if (keep) {
        const int pos = s_n[wave] + __popcll(m & ((1ull << lane) - 1ull));
        work[pos] = item;
        list_index(idx, n_t, t_lo, item, row_base, (uint32_t)pos, cap);
    }
}

#ifndef GEO_DIRECT_PER_WAVE
#define GEO_DIRECT_PER_WAVE 64
#endif
template <int GW>
struct GeoShared {                              // LDS of one geometry workgroup: ~4 KiB per wave
    float2 row[GW][WAVE];
    float acc[GW][6][WAVE];
    GeoItem item[GW][GEO_ITEMS];
    int count[GW];
};

// The border / obstacle forces on the 64 pedestrians `me` (one per lane, the same in every wave of the workgroup), slice `slice` of
// n_slices of the polylines: find, then scan.  Leaves every wave's partial sums in sh.acc[wave][6][lane]
// ({border x, y, static x, y, dynamic x, y}, the obstacle terms still without their factor -A) behind a workgroup barrier.
// DIRECT: the on-the-spot form below is compiled in (the fused tick's geometry workgroups.  In the stand-alone geometry kernel and the
// pair + geometry launch of the list-cutoff workloads the third inlined copy of the scan spills registers, their crowds have hundreds
// of polylines per wave, and a variant build with it measured the same on c3 / c5 and on the host-in-the-loop tick).
template <bool RAD, int GW, bool DIRECT = false>
__device__ __forceinline__ void geometry_forces(const TickArgs& a, GeoShared<GW>& sh, GeoLane& me, int slice, int n_slices, int tid,
                                                unsigned long long* st1) {
    const int lane = tid & (WAVE - 1);
    const int wave = uniform((int)(tid >> 6));
    const float inf = __builtin_inff();
    me.skip = a.border_skip > 0.0f ? a.border_skip + (RAD ? fmaxf(me.r, 0.0f) : 0.0f) : inf;
    // the tile's bounding box; parked (despawned) pedestrians sit ~3e15 m away, no cull can keep them
    const bool inbox = me.live && fabsf(me.x) < 1.0e12f && fabsf(me.y) < 1.0e12f;
    float x0 = inbox ? me.x : inf, y0 = inbox ? me.y : inf, x1 = inbox ? me.x : -inf, y1 = inbox ? me.y : -inf;
    float skip_max = me.live ? me.skip : 0.0f;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, m)); y0 = fminf(y0, __shfl_xor(y0, m));
        x1 = fmaxf(x1, __shfl_xor(x1, m)); y1 = fmaxf(y1, __shfl_xor(y1, m));
        skip_max = fmaxf(skip_max, __shfl_xor(skip_max, m));
    }
    GeoTile tb;
    tb.x0 = x0; tb.y0 = y0; tb.x1 = x1; tb.y1 = y1;
    tb.cx = 0.5f * (x0 + x1); tb.cy = 0.5f * (y0 + y1);
    const float hx = 0.5f * (x1 - x0), hy = 0.5f * (y1 - y0);
    tb.half_diag = sqrtf(fmaf(hx, hx, hy * hy)) * 1.0001f + 1.0e-3f;
    tb.skip_max = skip_max;
    float2* row = sh.row[wave];
    float f[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- phase 1: find
    const int gwave = slice * GW + wave, n_gwaves = GW * n_slices;       // small crowds: the tile's polylines are split over n_slices workgroups
    // Up to GEO_DIRECT_PER_WAVE polylines per wave (one trip of find per kind): every wave scans what it keeps on the spot -- no list,
    // no barrier, no second deal.  Their fixed cost is most of a small crowd's geometry workgroup, and up to 26 polylines per wave
    // the on-the-spot form measured 10-35 % better on the whole tick (tools/direct_scan_probe.py; more per wave: not measured)
    const int k_all = (a.en_border ? a.borders.K : 0) + (a.en_static ? a.statics.K : 0) + (a.en_dynamic ? a.dynamics.K : 0);
    if (DIRECT && k_all <= GEO_DIRECT_PER_WAVE * n_gwaves) {
        geo_find<RAD, true>(a, me, tb, sh.item[wave], row, lane, gwave, n_gwaves, f, 0);
#pragma unroll
        for (int q = 0; q < 6; ++q) sh.acc[wave][q][lane] = f[q];
        __syncthreads();
        return;
    }
    const int n_found = geo_find<RAD, false>(a, me, tb, sh.item[wave], row, lane, gwave, n_gwaves, f);
    if (lane == 0) sh.count[wave] = min(n_found, GEO_ITEMS);
    __syncthreads();
    if (st1) *st1 = __builtin_amdgcn_s_memrealtime();

    // ---- phase 2: scan, items dealt round-robin in (wave, index) order
    {
        int w = 0, first = 0, cnt = sh.count[0];                       // items [first, first + cnt) belong to list w
        int total = 0;
#pragma unroll
        for (int q = 0; q < GW; ++q) total += sh.count[q];
        total = uniform(total);
        for (int j = wave; j < total; j += GW) {
            while (j >= first + cnt) { first += cnt; ++w; cnt = sh.count[w]; }
            const GeoItem it = sh.item[uniform(w)][uniform(j - first)];
            GeoItem u = it;
            u.o0 = uniform(it.o0); u.o1 = uniform(it.o1); u.kind = uniform(it.kind);
            const bool keep = geo_keep(a, me, u.kind, u.c, u.s0, u.s1);
            geo_item<RAD>(a, me, u, keep, row, lane, f);
        }
    }
    // ---- list overflow (a tile whose pedestrians are spread over the whole map): this wave walks its polylines
    // again and scans, on the spot, the kept ones that did not fit
    if (n_found > GEO_ITEMS) geo_find<RAD, true>(a, me, tb, sh.item[wave], row, lane, gwave, n_gwaves, f);
#pragma unroll
    for (int q = 0; q < 6; ++q) sh.acc[wave][q][lane] = f[q];
    __syncthreads();
}

// The border / obstacle forces of tile (a.i_begin / 64 + bx), slice slice_y of n_slices: the body of sfm_geometry_kernel, callable
// from another kernel's workgroups as well (sfm_pair_geo_kernel).  n_bx only places the diagnostic stamps.
template <bool RAD, int GW>
__device__ __forceinline__ void geometry_block(const TickArgs& a, GeoShared<GW>& sh, int bx, int slice_y, int n_slices, int n_bx, int tid) {
    const int lane = tid & (WAVE - 1);
    const int wave = uniform((int)(tid >> 6));
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (a.geo_stamps) st0 = __builtin_amdgcn_s_memrealtime();
    const int t = (a.i_begin >> 6) + bx;                      // tile index
    const int p0 = max(a.i_begin, t * WAVE), p1 = min(a.i_end, (t + 1) * WAVE);
    const int i = t * WAVE + lane;
    GeoLane me;
    me.live = i >= p0 && i < p1;
    me.x = 3.0e15f; me.y = 3.0e15f; me.vx = 0.f; me.vy = 0.f; me.r = 0.f;
    if (me.live) {
        const float4 s = a.pk_cur[i];
        me.x = s.x; me.y = s.y; me.vx = s.z; me.vy = s.w;
        me.r = a.own[i].w;
    }
    me.walk = me.live && !(a.crossing && a.crossing[i]);              // forces.py:140-141,176-177
    const int slice = slice_y;
    geometry_forces<RAD, GW>(a, sh, me, slice, n_slices, tid, a.geo_stamps ? &st1 : nullptr);
    if (a.geo_stamps) st2 = __builtin_amdgcn_s_memrealtime();
    for (int q = wave; q < 6 && me.live; q += GW) {                    // wave w finishes component w (and w + GW)
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < GW; ++w) v += sh.acc[w][q][lane];
        if (q >= 4) v *= a.dyn.negA;
        else if (q >= 2) v *= a.stat.negA;
        a.geo[((size_t)slice * 6 + q) * a.N_pad + i] = v;
    }
    if (a.geo_stamps && tid == 0) {
        unsigned long long* o = a.geo_stamps + 4 * ((size_t)slice_y * n_bx + bx);
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// GW waves per tile: 16 for up to 1023 tiles (two workgroups per CU: 64 VGPRs, 64 KB of LDS each); 8 from 1024 tiles on --
// a workgroup's life is a chain of memory round trips, not work, so four thinner workgroups per CU beat two fat ones there.
template <bool RAD, int GW>
__global__ __launch_bounds__(GW * WAVE, 8) void sfm_geometry_kernel(const TickArgs a) {
    __shared__ GeoShared<GW> sh;
    if (a.list_work && (int)blockIdx.x >= a.list_block0) {             // the extra workgroups: the next pair kernel's tile-pair list
        if (blockIdx.y == 0) {
            __shared__ int s_n[GW + 1];
            const int idx = ((int)blockIdx.x - a.list_block0) * (GW * WAVE) + (int)threadIdx.x;     // shift-major, bx fastest
            const int shift = idx / a.list_n_t;
            uint32_t item = 0u;
            const bool keep = shift <= (a.list_n_t >> 1) &&
                              list_candidate(a.tile_box, a.tile_vmax, a.list_n_t, 0, a.list_n_t, a.ped.lam, a.cut_scale, a.cut_pad,
                                             PARTNERS_ALL, idx - shift * a.list_n_t, shift, item);
            list_emit(keep, item, a.list_work, a.list_count, s_n, a.list_idx, a.list_n_t, 0, 0u, a.list_cap);
        }
        return;
    }
    geometry_block<RAD, GW>(a, sh, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, (int)gridDim.x, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------------------
// pedestrian modes, waypoint queues and gap acceptance on the device
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 park_position(uint32_t pid) {      // where a despawned pedestrian waits: a ghost
    return make_float2(3.0e15f + 1.0e12f * (float)(pid + 1u), -3.0e15f);
}

// PedModeManager._activate_mode (ped_mode_manager.py:49-70): entry action of `m`; returns the new target speed.
__device__ __forceinline__ float fsm_enter(uint8_t m, float target, float initial, float crossing) {
    if (m == MODE_IDLE || m == MODE_CHECKING) return 0.0f;
    if (m == MODE_WALKING) return initial;
    if (m == MODE_CROSSING) return crossing;
    return target;                                                    // ROAD_TO_SIDEWALK keeps the speed
}
// PedModeManager.set_mode (ped_mode_manager.py:37-47): two requests are diverted through an intermediate mode.
__device__ __forceinline__ uint8_t fsm_request(uint8_t cur, uint8_t want) {
    if (cur == MODE_WALKING && want == MODE_CROSSING) return MODE_CHECKING;
    if (cur == MODE_CROSSING && want == MODE_WALKING) return MODE_ROAD_TO_SIDEWALK;
    return want;
}

// Distance from p to the segment a-b (shapely LineString.distance(Point)).
__device__ __forceinline__ float seg_dist(float2 a, float2 b, float2 p) {
    const float abx = b.x - a.x, aby = b.y - a.y;
    const float ab2 = fmaf(abx, abx, aby * aby);
    float t = ab2 > 0.0f ? (fmaf(p.x - a.x, abx, (p.y - a.y) * aby) / ab2) : 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float qx = fmaf(t, abx, a.x) - p.x, qy = fmaf(t, aby, a.y) - p.y;
    return sqrtf(fmaf(qx, qx, qy * qy));
}

// check_traffic (check_traffic.py:7-61) in closed form: does any vehicle's extrapolated path cross the
// pedestrian's path while the pedestrian would be there?  true = safe to cross.
__device__ bool gap_accepted(const TickArgs& a, float2 loc, float2 goal, float ped_speed, float margin) {
    if (margin < 0.0f) return true;                                   // crosses without looking (:24)
    if (!(ped_speed > 0.0f)) return true;
    const float rx = goal.x - loc.x, ry = goal.y - loc.y;
    const float time_ped = sqrtf(fmaf(rx, rx, ry * ry)) / ped_speed;
    const float rr = fmaf(rx, rx, ry * ry);
    for (int k = 0; k < a.dynamics.K; ++k) {
        const float4 c = a.dynamics.ctr[k];                           // {cx, cy, vx, vy}
        const float sp = sqrtf(fmaf(c.z, c.z, c.w * c.w));
        const float inv = sp == 0.0f ? 1.0f : 1.0f / sp;
        // (sic) every vehicle is offset by the FIRST vehicle's extent, element-wise (check_traffic.py:35-36)
        const float ox = c.z * inv * a.fsm.veh_ext_x, oy = c.w * inv * a.fsm.veh_ext_y;
        const float2 front = make_float2(c.x + ox, c.y + oy), back = make_float2(c.x - ox, c.y - oy);
        const float T = time_ped + margin;
        const float2 vgoal = make_float2(fmaf(c.z, T, front.x), fmaf(c.w, T, front.y));
        // segment loc-goal (p0 + t r) against back-vgoal (q0 + u s)
        const float sx = vgoal.x - back.x, sy = vgoal.y - back.y;
        const float qpx = back.x - loc.x, qpy = back.y - loc.y;
        const float rxs = rx * sy - ry * sx, qpxr = qpx * ry - qpy * rx;
        bool hit = false, is_seg = false;
        float2 h0 = make_float2(0.f, 0.f), h1 = h0;
        if (rxs != 0.0f) {
            const float t = (qpx * sy - qpy * sx) / rxs, u = qpxr / rxs;
            if (t >= 0.0f && t <= 1.0f && u >= 0.0f && u <= 1.0f) { hit = true; h0 = make_float2(fmaf(t, rx, loc.x), fmaf(t, ry, loc.y)); }
        } else if (qpxr == 0.0f && rr > 0.0f) {                       // collinear: overlap interval on the pedestrian's segment
            const float t0 = fmaf(qpx, rx, qpy * ry) / rr, t1 = t0 + fmaf(sx, rx, sy * ry) / rr;
            const float lo = fmaxf(0.0f, fminf(t0, t1)), hi = fminf(1.0f, fmaxf(t0, t1));
            if (lo <= hi) {
                hit = true; is_seg = lo < hi;
                h0 = make_float2(fmaf(lo, rx, loc.x), fmaf(lo, ry, loc.y));
                h1 = make_float2(fmaf(hi, rx, loc.x), fmaf(hi, ry, loc.y));
            }
        }
        if (!hit || sp == 0.0f) continue;
        if (!is_seg) h1 = h0;
        const float tti_ped = seg_dist(h0, h1, loc) / ped_speed;
        const float tti_front = seg_dist(h0, h1, front) / sp, tti_back = seg_dist(h0, h1, back) / sp;
        if (tti_front - margin < tti_ped && tti_ped < tti_back + margin) return false;
    }
    return true;
}

// Start of PedestrianSimulation.tick (pedestrian_simulation.py:63-73): apply_current_mode, the FSM tick, gap
// acceptance for the pedestrians waiting at the kerb.  One thread per row; FSM arrays are in the caller's index.
__global__ void sfm_mode_kernel(const TickArgs a) {
    const int i = a.i_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.i_end) return;
    const uint32_t pid = a.ids ? a.ids[i] : (uint32_t)i;
    uint8_t m = a.fsm.mode[pid];
    float4 o = a.own[i];
    if (m == MODE_DESPAWNED) { o.z = 0.0f; a.own[i] = o; return; }
    float tgt = a.fsm.target[pid];
    o.z = tgt;                                                        // apply_current_mode (pedestrian_state.py:94-95)
    if (m == MODE_IDLE && a.fsm.next_mode_time[pid] <= a.fsm.sim_time) {   // ped_mode_manager.py:30-35
        m = MODE_WALKING;
        tgt = a.fsm.initial_speed[pid];
    }
    if (m == MODE_CHECKING) {                                         // pedestrian_simulation.py:67-73
        const float4 s = a.pk_cur[i];
        const bool ready = a.dynamics.K == 0 ||
                           gap_accepted(a, make_float2(s.x, s.y), make_float2(o.x, o.y), a.fsm.crossing_speed[pid],
                                        a.fsm.safety_margin[pid]);
        if (ready) { m = MODE_CROSSING; tgt = a.fsm.crossing_speed[pid]; }
    }
    a.fsm.mode[pid] = m;
    a.fsm.target[pid] = tgt;
    a.own[i] = o;
    a.crossing[i] = (m == MODE_CROSSING || m == MODE_ROAD_TO_SIDEWALK) ? 1 : 0;   // forces.py:176-177
}

// run_simulation.py:118-132 for one arrived pedestrian: next waypoint from its list (and the mode request that
// comes with it, pedestrian_state.py:83-92), or despawn when the list is exhausted.
__device__ __forceinline__ void fsm_arrived(const TickArgs& a, uint32_t pid, float& wx, float& wy, bool& wp_changed,
                                            bool& despawn) {
    const int c = a.fsm.cursor[pid];
    if (c < a.fsm.wp_off[pid + 1] - a.fsm.wp_off[pid]) {
        const int e = a.fsm.wp_off[pid] + c;
        const float2 w = a.fsm.wp_xy[e];
        wx = w.x; wy = w.y; wp_changed = true;
        a.fsm.cursor[pid] = c + 1;
        const uint8_t cur = a.fsm.mode[pid];
        const uint8_t nm = fsm_request(cur, a.fsm.wp_cross[e] ? MODE_CROSSING : MODE_WALKING);
        a.fsm.target[pid] = fsm_enter(nm, a.fsm.target[pid], a.fsm.initial_speed[pid], a.fsm.crossing_speed[pid]);
        a.fsm.mode[pid] = nm;
    } else if (a.fsm.despawn_on_arrival) {
        despawn = true;
        a.fsm.mode[pid] = MODE_DESPAWNED;
        a.fsm.target[pid] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------------
// the fused tick
// ------------------------------------------------------------------------------------------------------
// TEAM = number of waves that share the same IPW pedestrians:
//   TEAM 1: every wave has its own rows; the workgroup stages j-tiles through LDS (large shards);
//   TEAM 4: the 4 waves of the workgroup share the rows and split the j range (wave w takes records
//           64*(4s + w) ...), each streaming its own coalesced float4 per lane straight from L2 with a
//           register prefetch; partial sums meet in LDS.  This multiplies the number of independent pair
//           chains per SIMD by 4 -- what a small crowd (N = 4096: only 4 rows per SIMD) needs to hide the
//           latency of the dependent rsq -> rsq -> rcp -> exp chain.
template <int IPW, bool Z3, bool RAD, int TEAM>
__global__ __launch_bounds__(BLOCK) void sfm_tick_kernel(const TickArgs a) {
    constexpr bool STAGED = (TEAM == 1);
    __shared__ __attribute__((aligned(16))) float4 s_pk[STAGED ? 2 : 1][STAGED ? TILE_J : 1];
    __shared__ __attribute__((aligned(16))) float2 s_zv[(STAGED && Z3) ? 2 : 1][(STAGED && Z3) ? TILE_J : 1];
    __shared__ float s_rad[(STAGED && RAD) ? 2 : 1][(STAGED && RAD) ? TILE_J : 1];
    __shared__ float s_part[WAVES_PER_BLOCK][IPW][3];      // TEAM 4: per-wave partial pedestrian forces

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = uniform(tid >> 6);
    if (a.adv.M > 0 && (int)blockIdx.x >= a.adv.block0) {            // the extra workgroups: vehicles move on (nobody in this
        const int k = ((int)blockIdx.x - a.adv.block0) * WAVES_PER_BLOCK + wave;   // launch reads their rings)
        if (k < a.adv.M) advance_vehicle(a.adv, k, lane, true);
        return;
    }
    const int ibase = a.i_begin + (TEAM == 1 ? (blockIdx.x * WAVES_PER_BLOCK + wave) : blockIdx.x) * IPW;   // uniform
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

        // cutoff: bit t of lane l says whether partner tile 64 t + l has to be evaluated against this wave's
        // rows (they all lie in one 64-tile); see tiles_negligible()
        const bool cut = a.tile_box != nullptr && N <= 64 * 64 * WAVE;
        unsigned int keep_lo = 0xffffffffu, keep_hi = 0xffffffffu;
        if (cut) {
            // the wave's rows lie in one 64-tile when the shard starts on a multiple of IPW; a ragged shard start can put
            // them in two, and then a partner tile is kept if it is not negligible for either
            const int n_t = (N + WAVE - 1) / WAVE, ti = min(ibase, a.i_end - 1) >> 6, ti2 = min(ibase + IPW - 1, a.i_end - 1) >> 6;
            const float4 bt = a.tile_box[ti], bt2 = a.tile_box[ti2];
            const float vt = a.tile_vmax[ti], vt2 = a.tile_vmax[ti2];
            unsigned long long km = 0ull;
            for (int t = 0; t * WAVE < n_t; ++t) {
                const int u = t * WAVE + lane;
                const float4 bu = a.tile_box[min(u, n_t - 1)];
                const float vu = a.tile_vmax[min(u, n_t - 1)];
                bool keep = (u < n_t) && !tiles_negligible(bt, vt, bu, vu, a.ped.lam, a.cut_scale, a.cut_pad);
                if (ti2 != ti) keep |= (u < n_t) && !tiles_negligible(bt2, vt2, bu, vu, a.ped.lam, a.cut_scale, a.cut_pad);   // uniform
                km |= (unsigned long long)keep << t;
            }
            keep_lo = (unsigned int)km;
            keep_hi = (unsigned int)(km >> 32);
        }

        // 64 pairs x IPW rows: lane's record pj against the wave's rows
        auto step = [&](int j0, const float4 pj, float zj, float vzj, float rj) {
            const bool clean = (j0 + WAVE <= N) && (ibase + IPW <= j0 || ibase >= j0 + WAVE);   // uniform
            if (clean) {
#pragma unroll
                for (int k = 0; k < IPW; ++k) {
                    float rinv;
                    moussaid<Z3, RAD, Z3>(a.ped, pj.x - xi[k], pj.y - yi[k], zj - zi[k], vxi[k] - pj.z,
                                          vyi[k] - pj.w, vzi[k] - vzj, ri[k] + rj, gx[k], gy[k], gz[k], rinv);
                    flag = fmaxf(flag, rinv);
                }
            } else {                                                  // step holds the diagonal or the tail
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
        };

        if (cut) {
            // Only the kept partner tiles are visited, in ascending order, each as one coalesced 1-KiB load per
            // wave straight from L2 (the next kept tile's load is in flight while this one is evaluated).  TEAM 4:
            // the waves take the kept tiles round-robin.
            const int n_t = (N + WAVE - 1) / WAVE;
            int ord = 0;                                               // uniform
            bool have = false;
            int c_j0 = 0;
            float4 c_pk = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 c_zv = make_float2(0.f, 0.f);
            float c_rad = 0.0f;
            for (int t = 0; t * WAVE < n_t; ++t) {
                const unsigned int kw = t < 32 ? keep_lo : keep_hi;
                unsigned long long m = __ballot((kw >> (t & 31)) & 1u);
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if (TEAM > 1 && (ord++ & (TEAM - 1)) != wave) continue;
                    const int j0 = (t * WAVE + b) * WAVE;            // rows up to N_pad are allocated
                    const float4 n_pk = a.pk_cur[j0 + lane];
                    float2 n_zv = make_float2(0.f, 0.f);
                    float n_rad = 0.0f;
                    if (Z3) n_zv = a.zv_cur[j0 + lane];
                    if (RAD) n_rad = a.radius[j0 + lane];
                    if (have) step(c_j0, c_pk, c_zv.x, c_zv.y, c_rad);
                    c_j0 = j0; c_pk = n_pk; c_zv = n_zv; c_rad = n_rad;
                    have = true;
                }
            }
            if (have) step(c_j0, c_pk, c_zv.x, c_zv.y, c_rad);
        } else if (STAGED) {
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
                    step(j0, pj, zj, vzj, rj);
                }
                if (t + 1 < ntiles) {
                    s_pk[buf ^ 1][tid] = r_pk;
                    if (Z3) s_zv[buf ^ 1][tid] = r_zv;
                    if (RAD) s_rad[buf ^ 1][tid] = r_rad;
                }
                __syncthreads();
            }
        } else {
            // TEAM 4: this wave streams records 64*(4s + wave) + lane, one step ahead in registers
            int j0 = wave * WAVE;
            float4 n_pk = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 n_zv = make_float2(0.f, 0.f);
            float n_rad = 0.0f;
            if (j0 < N) {
                n_pk = a.pk_cur[j0 + lane];
                if (Z3) n_zv = a.zv_cur[j0 + lane];
                if (RAD) n_rad = a.radius[j0 + lane];
            }
#pragma unroll 1
            for (; j0 < N; j0 += WAVE * TEAM) {
                const float4 pj = n_pk;
                const float2 zz = n_zv;
                const float rj = n_rad;
                const int jn = j0 + WAVE * TEAM;
                if (jn < N) {                                         // rows up to N_pad are allocated
                    n_pk = a.pk_cur[jn + lane];
                    if (Z3) n_zv = a.zv_cur[jn + lane];
                    if (RAD) n_rad = a.radius[jn + lane];
                }
                step(j0, pj, zz.x, zz.y, rj);
            }
        }
        // 2-D fast path: a coincident valid pair needs the reference's zero-vector conventions -> redo this
        // wave's part with the exact body, straight from global memory (rare; no workgroup barrier inside).
        if (!Z3) {
            flag = wave_max(flag);
            if (flag >= COINCIDENT_RINV) {
#pragma unroll
                for (int k = 0; k < IPW; ++k) { gx[k] = 0.0f; gy[k] = 0.0f; }
                for (int j0 = (TEAM == 1 ? 0 : wave * WAVE); j0 < N; j0 += WAVE * TEAM) {
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

    if (TEAM > 1) {                                  // partial sums of the team meet in LDS, fixed order
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < IPW; ++k) { s_part[wave][k][0] = gx[k]; s_part[wave][k][1] = gy[k]; s_part[wave][k][2] = gz[k]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < IPW; ++k) {
            gx[k] = ((s_part[0][k][0] + s_part[1][k][0]) + s_part[2][k][0]) + s_part[3][k][0];
            gy[k] = ((s_part[0][k][1] + s_part[1][k][1]) + s_part[2][k][1]) + s_part[3][k][1];
            gz[k] = ((s_part[0][k][2] + s_part[1][k][2]) + s_part[2][k][2]) + s_part[3][k][2];
        }
    }

    // ---- per-pedestrian part: geometry forces + epilogue (wave-uniform values, lane 0 stores) ----------
    // TEAM 1: the wave walks its own rows; TEAM 4: row k is finished by wave k mod 4.
#pragma unroll 1
    for (int k = (TEAM == 1 ? 0 : wave); k < IPW; k += TEAM) {
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
        (void)r;

        float fbx = 0.f, fby = 0.f, fsx = 0.f, fsy = 0.f, fdx = 0.f, fdy = 0.f;
        if (a.geo) {                                // border / obstacle forces from sfm_geometry_kernel
            const size_t np_ = (size_t)a.N_pad;
            for (int sl = 0; sl < a.geo_slices; ++sl) {
                const float* gsl = a.geo + (size_t)sl * 6 * np_ + i;
                fbx += gsl[0 * np_]; fby += gsl[1 * np_]; fsx += gsl[2 * np_];
                fsy += gsl[3 * np_]; fdx += gsl[4 * np_]; fdy += gsl[5 * np_];
            }
        }

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
        bool redraw = false, despawn = false;
        const uint32_t pid = a.ids ? a.ids[i] : (uint32_t)i;             // queues / streams are keyed by the caller's index
        const bool gone = a.fsm.mode && a.fsm.mode[pid] == MODE_DESPAWNED;
        {
            const float ax_ = wx - x, ay_ = wy - y;
            const bool here = fmaf(ax_, ax_, ay_ * ay_) < a.arrive_thr2;
            if (a.fsm.mode) {
                if (here && !gone && lane == 0) fsm_arrived(a, pid, wx, wy, redraw, despawn);
                wx = uniform(wx); wy = uniform(wy);
                redraw = __builtin_amdgcn_readfirstlane((int)redraw) != 0;
                despawn = __builtin_amdgcn_readfirstlane((int)despawn) != 0;
            } else if ((a.flags & 2u) && here) {
                redraw = true;
                nd = a.draws[i] + 1u;
                wx = waypoint_coord(a.seed, pid, nd, 0u, a.world_side);
                wy = waypoint_coord(a.seed, pid, nd, 1u, a.world_side);
            }
        }
        float nx = x, ny = y, nz = z;
        if (a.flags & 1u) { nx = fmaf(a.dt, nvx, x); ny = fmaf(a.dt, nvy, y); nz = fmaf(a.dt, nvz, z); }
        if (despawn || gone) {                                           // destroy_pedestrian: parked as a ghost
            const float2 pp = park_position(pid);
            nx = pp.x; ny = pp.y; nvx = 0.f; nvy = 0.f; nvz = 0.f;
        }

        if (lane == 0) {
            a.pk_next[i] = make_float4(nx, ny, nvx, nvy);
            if (Z3) a.zv_next[i] = make_float2(nz, nvz);
            if (a.host_pk) { a.host_pk[i] = make_float4(nx, ny, nvx, nvy); if (Z3) a.host_zv[i] = make_float2(nz, nvz); }
            if (redraw) { a.own[i] = make_float4(wx, wy, o.z, o.w); if (!a.fsm.mode) a.draws[i] = nd; }
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

__global__ __launch_bounds__(WAVE) void sfm_tile_bounds_kernel(const float4* __restrict__ pk, const float2* __restrict__ zv, int N,
                                                               float4* __restrict__ box, float* __restrict__ vmax, int t0) {
    const int t = t0 + blockIdx.x, lane = threadIdx.x;
    const int i = t * WAVE + lane;
    const float inf = __builtin_inff();
    float x0 = inf, y0 = inf, x1 = -inf, y1 = -inf, v = 0.0f;
    if (i < N) {
        const float4 s = pk[i];
        if (fabsf(s.x) < 1.0e14f) {                             // despawned pedestrians are parked far away: not in the box
            x0 = x1 = s.x; y0 = y1 = s.y;
            const float vz = zv ? zv[i].y : 0.0f;               // 3-D crowds: |D| is a 3-component norm (forces.py:85-86), so is the speed
            v = sqrtf(fmaf(s.z, s.z, fmaf(s.w, s.w, vz * vz))) * 1.000001f;  // rounded up: the bound must stay a bound
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, m)); y0 = fminf(y0, __shfl_xor(y0, m));
        x1 = fmaxf(x1, __shfl_xor(x1, m)); y1 = fmaxf(y1, __shfl_xor(y1, m));
        v = fmaxf(v, __shfl_xor(v, m));
    }
    if (lane == 0) { box[t] = make_float4(x0, y0, x1, y1); vmax[t] = v; }
}

// Box and largest speed of each run of `tps` consecutive tiles (one wave per run).  A run is an x-strip of the spatial
// packing, so its box is a slab of the map; any order is fine for correctness (the box is just the union).
__global__ __launch_bounds__(WAVE) void sfm_strip_bounds_kernel(const float4* __restrict__ box, const float* __restrict__ vmax,
                                                                int n_t, int tps, float4* __restrict__ sbox,
                                                                float* __restrict__ svmax) {
    const int s = blockIdx.x, lane = threadIdx.x;
    const float inf = __builtin_inff();
    float x0 = inf, y0 = inf, x1 = -inf, y1 = -inf, v = 0.0f;
    for (int q = lane; q < tps; q += WAVE) {
        const int t = s * tps + q;
        if (t < n_t) {
            const float4 b = box[t];
            x0 = fminf(x0, b.x); y0 = fminf(y0, b.y); x1 = fmaxf(x1, b.z); y1 = fmaxf(y1, b.w);
            v = fmaxf(v, vmax[t]);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, m)); y0 = fminf(y0, __shfl_xor(y0, m));
        x1 = fmaxf(x1, __shfl_xor(x1, m)); y1 = fmaxf(y1, __shfl_xor(y1, m));
        v = fmaxf(v, __shfl_xor(v, m));
    }
    if (lane == 0) { sbox[s] = make_float4(x0, y0, x1, y1); svmax[s] = v; }
}

// Both in one launch (large crowds with the two-level list): one 16-wave workgroup per strip, wave w takes tiles w, w+16, ... of
// the strip, leaves each tile's box / largest speed and the workgroup combines the strip's.
constexpr int TSB_WAVES = 16;
__global__ __launch_bounds__(TSB_WAVES * WAVE) void sfm_tile_strip_bounds_kernel(const float4* __restrict__ pk, int N, int n_t, int tps,
                                                                                 float4* __restrict__ box, float* __restrict__ vmax,
                                                                                 float4* __restrict__ sbox, float* __restrict__ svmax) {
    __shared__ float s_part[TSB_WAVES][5];
    const int s = blockIdx.x, lane = threadIdx.x & (WAVE - 1);
    const int wave = uniform((int)(threadIdx.x >> 6));
    const float inf = __builtin_inff();
    float sx0 = inf, sy0 = inf, sx1 = -inf, sy1 = -inf, sv = 0.0f;            // this wave's share of the strip (uniform)
    // (round 4: four tiles' rows in flight per wave -- a strip of 64 tiles used to be four dependent round trips per wave, and on a
    //  shard of a large crowd this launch is a fixed cost of every tick)
    constexpr int TSB_AHEAD = 4;
  for (int q0 = wave; q0 < tps; q0 += TSB_WAVES * TSB_AHEAD) {
    float4 pre[TSB_AHEAD];
#pragma unroll
    for (int u = 0; u < TSB_AHEAD; ++u) {
        const int t = s * tps + q0 + u * TSB_WAVES, i = t * WAVE + lane;
        pre[u] = (q0 + u * TSB_WAVES < tps && t < n_t && i < N) ? pk[i] : make_float4(3.0e15f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < TSB_AHEAD; ++u) {
        const int q = q0 + u * TSB_WAVES;
        const int t = s * tps + q;
        if (q >= tps || t >= n_t) break;
        float x0 = inf, y0 = inf, x1 = -inf, y1 = -inf, v = 0.0f;
        {
            const float4 p = pre[u];
            if (fabsf(p.x) < 1.0e14f) {
                x0 = x1 = p.x; y0 = y1 = p.y;
                v = sqrtf(fmaf(p.z, p.z, p.w * p.w)) * 1.000001f;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            x0 = fminf(x0, __shfl_xor(x0, m)); y0 = fminf(y0, __shfl_xor(y0, m));
            x1 = fmaxf(x1, __shfl_xor(x1, m)); y1 = fmaxf(y1, __shfl_xor(y1, m));
            v = fmaxf(v, __shfl_xor(v, m));
        }
        if (lane == 0) { box[t] = make_float4(x0, y0, x1, y1); vmax[t] = v; }
        sx0 = fminf(sx0, x0); sy0 = fminf(sy0, y0); sx1 = fmaxf(sx1, x1); sy1 = fmaxf(sy1, y1); sv = fmaxf(sv, v);
    }
  }
    if (lane == 0) { s_part[wave][0] = sx0; s_part[wave][1] = sy0; s_part[wave][2] = sx1; s_part[wave][3] = sy1; s_part[wave][4] = sv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TSB_WAVES; ++w) {
            sx0 = fminf(sx0, s_part[w][0]); sy0 = fminf(sy0, s_part[w][1]); sx1 = fmaxf(sx1, s_part[w][2]); sy1 = fmaxf(sy1, s_part[w][3]);
            sv = fmaxf(sv, s_part[w][4]);
        }
        sbox[s] = make_float4(sx0, sy0, sx1, sy1);
        svmax[s] = sv;
    }
}

__global__ __launch_bounds__(256) void sfm_pair_list_kernel(const float4* __restrict__ box, const float* __restrict__ vmax, int n_t, int t_lo,
                                                            int t_hi, float lam, float cut_scale, float cut_pad,
                                                            uint32_t* __restrict__ work, int* __restrict__ count, int partners,
                                                            uint32_t* __restrict__ idx, uint32_t row_base, uint32_t cap) {
    __shared__ int s_n[5];
    uint32_t item = 0u;
    const bool keep = list_candidate(box, vmax, n_t, t_lo, t_hi, lam, cut_scale, cut_pad, partners,
                                     t_lo + (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)blockIdx.y, item);
    list_emit(keep, item, work, count, s_n, idx, n_t, t_lo, row_base, cap);
}

// The same list for large crowds, two levels: one wave per own tile tests the strips first, 64 at a time (a strip's box
// contains its tiles' boxes and its speed bounds theirs, and tiles_negligible is monotone in both, so a negligible strip
// holds only negligible tiles -- the result is exactly the flat kernel's), then the tiles of the surviving strips.
constexpr int LIST2_WAVES = 16;          // tiles per workgroup = atomics saved
__global__ __launch_bounds__(LIST2_WAVES * WAVE) void sfm_pair_list2_kernel(const float4* __restrict__ box, const float* __restrict__ vmax,
                                                               const SymArgs sa, float lam, uint32_t* __restrict__ work,
                                                               int* __restrict__ count, int partners) {
    constexpr int BUF = 256;                              // a wave collects its items in LDS: one atomic per flush
    __shared__ uint32_t s_item[LIST2_WAVES][BUF];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = uniform((int)(threadIdx.x >> 6));
    // one wave per own tile.  The list's single counter is what this kernel waits for (one device-scope atomic costs ~11 ns on one
    // word: 4096 of them were 46 of the kernel's 54 us at c5), so a wave only appends on its own when its LDS buffer overflows;
    // at the end the 16 waves of a workgroup reserve their space with ONE atomic.
    __shared__ int s_n[LIST2_WAVES], s_base;
    const int bx = sa.t_lo + blockIdx.x * LIST2_WAVES + wave;
    const bool live = bx < sa.t_hi;
    uint32_t* buf = s_item[wave];
    int n_buf = 0;                                         // uniform
    auto flush = [&]() {
        if (n_buf == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(count, n_buf);
        base = __builtin_amdgcn_readfirstlane(base);
        for (int q = lane; q < n_buf; q += WAVE) { work[base + q] = buf[q]; list_index(sa.idx, sa.n_t, sa.t_lo, buf[q], sa.row_base, (uint32_t)(base + q), sa.cap_pairs); }
        n_buf = 0;
    };
    const int n_t = sa.n_t;
    const float4 bt = box[live ? bx : sa.t_lo];
    const float vt = vmax[live ? bx : sa.t_lo];
    for (int sb = 0; live && sb < sa.n_strips; sb += WAVE) {
        const int s = sb + lane;
        const bool hit = s < sa.n_strips && !tiles_negligible(bt, vt, sa.sbox[min(s, sa.n_strips - 1)], sa.svmax[min(s, sa.n_strips - 1)],
                                                              lam, sa.cut_scale, sa.cut_pad);
        unsigned long long ms = __ballot(hit);
        // the surviving strips in trips of 64 tiles, FOUR trips at a time with their boxes in flight together (one trip at a time,
        // the next one prefetched, left a chain of ~20 dependent round trips per tile: 18 us at c5)
        int s0 = -1, q0 = 0;                                   // generator state: trip = tiles s0 + q0 + lane; s0 = -1: none left
        auto advance = [&]() {
            if (s0 >= 0 && q0 + WAVE < sa.tps) { q0 += WAVE; return; }
            if (!ms) { s0 = -1; return; }
            s0 = (sb + __ffsll((long long)ms) - 1) * sa.tps;
            ms &= ms - 1;
            q0 = 0;
        };
        for (;;) {
            int tb[4];
            float4 fb[4];
            float fv[4];
            bool ok[4];
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                advance();
                any = any || s0 >= 0;
                tb[j] = s0 + q0 + lane;
                ok[j] = s0 >= 0 && (q0 + lane) < sa.tps && tb[j] < n_t;
                const int tc = ok[j] ? tb[j] : bx;
                fb[j] = box[tc];
                fv[j] = vmax[tc];
            }
            if (!any) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bool keep = ok[j];
                uint32_t item = 0u;
                if (keep) {
                    int shift = tb[j] - bx;
                    if (shift < 0) shift += n_t;
                    if (shift == 0) {
                        keep = (bx - sa.t_lo) < ((sa.t_hi - sa.t_lo + 1) >> 1) && partners != PARTNERS_REMOTE;
                    } else {
                        const bool own = tb[j] >= sa.t_lo && tb[j] < sa.t_hi;
                        if ((partners == PARTNERS_OWN && !own) || (partners == PARTNERS_REMOTE && own)) keep = false;
                        else if (own) keep = shift <= (n_t >> 1) && !(!(n_t & 1) && shift == (n_t >> 1) && bx >= (n_t >> 1));
                        if (keep) keep = !tiles_negligible(bt, vt, fb[j], fv[j], lam, sa.cut_scale, sa.cut_pad);
                        item = own ? 0u : WORK_ONE_SIDED;
                    }
                    item |= (uint32_t)bx | ((uint32_t)shift << 16);
                }
                const unsigned long long mk = __ballot(keep);
                if (mk) {
                    if (n_buf + WAVE > BUF) flush();
                    if (keep) buf[n_buf + __popcll(mk & ((1ull << lane) - 1ull))] = item;
                    n_buf += __popcll(mk);
                }
            }
            if (s0 < 0) break;
        }
    }
    if (lane == 0) s_n[wave] = n_buf;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < LIST2_WAVES; ++w) total += s_n[w];
        s_base = total ? atomicAdd(count, total) : 0;
    }
    __syncthreads();
    int base = s_base;
    for (int w = 0; w < wave; ++w) base += s_n[w];
    for (int q = lane; q < n_buf; q += WAVE) { work[base + q] = buf[q]; list_index(sa.idx, sa.n_t, sa.t_lo, buf[q], sa.row_base, (uint32_t)(base + q), sa.cap_pairs); }
}

// ------------------------------------------------------------------------------------------------------
// symmetric pedestrian force: every unordered pair once (F_ji = -F_ij), systolic over the wavefront
// ------------------------------------------------------------------------------------------------------
// Pedestrians are cut into tiles of 64.  A workgroup owns one unordered tile pair (ta, tb): each lane keeps
// ONE pedestrian j of tile tb fixed in registers, while the 64 pedestrians i of tile ta -- together with
// their force accumulators -- rotate through the lanes one step at a time (6 DPP wavefront-rotate moves per
// 64 pairs).  Each step evaluates the Moussaid term once and adds it to the travelling F_i and subtracts it
// from the resident F_j, so nothing is reduced across lanes and nothing is atomically added.  The 64 steps
// of a tile pair are split over the 4 waves; partial sums meet in LDS in a fixed order and are written, one
// coalesced 512-B row per side, to slab[partner tile][pedestrian] -- every slab entry is written exactly
// once per tick, and the epilogue kernel sums a pedestrian's n_t entries in a fixed order (deterministic).
// Padding rows are "ghost" pedestrians parked ~3e15 m away at distinct positions: their terms underflow to
// exactly 0, so no masking is needed.  Planar crowds without use_ped_radius only (whole crowd, or a shard of whole
// tiles under the tile-pair list, where a pair with another rank's tile is evaluated one-sided); anything else uses
// the ordered kernel above.
// wave_rol:1 of a value: lane l receives lane l + 1's (probed at init).  With nothing kept of the old value the move folds into the
// VALU instruction that consumes it (v_add_f32_dpp)
__device__ __forceinline__ float rot_in(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x134 /* wave_rol:1 */, 0xf, 0xf, true));
}

__global__ void sfm_dpp_probe_kernel(int* out) {       // which way does wave_rol:1 move data?
    const int lane = threadIdx.x & 63;
    out[lane] = __builtin_amdgcn_update_dpp(0, lane, 0x134, 0xf, 0xf, false);
}

struct PairShared {                             // LDS of one pair workgroup
    float2 fi[WAVES_PER_BLOCK][WAVE];
    float2 fj[WAVES_PER_BLOCK][WAVE];
    float4 trav[2][2 * WAVE];                   // the travelling tile(s) {x, y, lambda vx, lambda vy}, each twice back to back (two: a diagonal item)
    float radt[2][2 * WAVE];                    // ... and their radii (use_ped_radius)
    float2 travz[2][2 * WAVE];                  // 3-D crowds: {z, lambda vz} of the travelling tile(s)
    float fiz[WAVES_PER_BLOCK][WAVE];           // ... and the z components of the two sums
    float fjz[WAVES_PER_BLOCK][WAVE];
};

// The body of sfm_pair_sym_kernel for workgroup (bid_x, bid_y) of a grid grid_x wide, callable from another kernel's workgroups as
// well (sfm_pair_geo_kernel).
template <bool RAD, bool CUT, bool Z3>
__device__ __forceinline__ void pair_block(const float4* __restrict__ pk, const float2* __restrict__ zv, const float* __restrict__ radius, const IxConst& c,
                                           const SymArgs& sa, PairShared& sh, int bid_x, int bid_y, int grid_x, int tid) {
    const int lane = tid & (WAVE - 1);
    const int wave = uniform(tid >> 6);
    const int n_t = sa.n_t;
    const int half_up = (sa.t_hi - sa.t_lo + 1) >> 1;      // diagonal items pair own tile bx with bx + half_up
    unsigned long long t_start = 0;
    if (sa.stamps) t_start = __builtin_amdgcn_s_memrealtime();
    // work items: the (bx, shift) of the 2-D grid, or -- cutoff on -- entries of the compacted list, strided
    const int n_items = sa.work ? *sa.work_count : 1;
    // list mode: a workgroup takes a contiguous run of the list (consecutive items share a tile: its rows stay hot in this
    // CU's cache and no two workgroups hammer the same lines at the same moment)
    const int run = sa.work ? (n_items + grid_x - 1) / grid_x : 1;
    const int item0 = sa.work ? bid_x * run : 0;
  for (int item = item0; item < min(n_items, item0 + run); ++item) {
    int shift = bid_y, bx = bid_x;      // shift = tile distance: 0 = diagonal items
    bool one_sided = false;                       // tb belongs to another rank: only tile ta's side is kept
    if (sa.work) {
        const uint32_t w = sa.work[item];
        bx = (int)(w & 0xffffu); shift = (int)((w >> 16) & 0x7fffu); one_sided = (w & WORK_ONE_SIDED) != 0u;
    }
    int ta, tb, sig0, nsteps;
    bool diag = false;
    if (shift == 0) {
        // diagonal tiles are half the work (32 steps): two of them share a workgroup, two waves each
        if (!sa.work && bx - sa.t_lo >= half_up) return;   // grid mode only (the list holds each diagonal item once)
        ta = (wave < 2) ? bx : bx + half_up;
        if (ta >= sa.t_hi) { ta = -1; }
        tb = ta;
        diag = true;
        sig0 = 1 + 16 * (wave & 1);               // sigma 1..16 / 17..32 (sigma = 32 is one-sided)
        nsteps = 16;
    } else {
        // Antipodal pairs (even n_t, shift n_t/2) appear twice in the 2-D grid: the upper half leaves.  The list
        // builders have already dropped that duplicate for own-own pairs, and a ONE-SIDED antipodal item (partner tile
        // on another rank) must be evaluated by whichever own tile it names -- so this is grid mode only.
        if (!sa.work && !(n_t & 1) && shift == (n_t >> 1) && bx >= (n_t >> 1)) return;
        ta = bx;
        tb = bx + shift;
        if (tb >= n_t) tb -= n_t;
        sig0 = 16 * wave;
        nsteps = 16;
    }
    if (sa.debug_steps >= 0) nsteps = min(sa.debug_steps, 16);   // (the doubled LDS image holds slots lane + sig0 + s <= 127 only for s < 16)

    float fxi = 0.f, fyi = 0.f, fxj = 0.f, fyj = 0.f;
    float fzi = 0.f, fzj = 0.f;                   // 3-D crowds (Z3): the z components (forces.py:112-117 keeps 3-component forces)
    int i_end_loc = lane;
    // The travelling tile goes through LDS (round 3): one wave stages it, twice back to back and with lambda v folded in, and
    // step s of a lane is one ds_read_b128 at slot lane + sig0 + s -- no operand moves between lanes on the VALU; only the two
    // travelling sums still rotate, inside the add that takes the step's term (v_add_f32_dpp).  54 instead of 60 VALU per step.
    const int tsel = diag ? (wave >> 1) : 0;
    if (ta >= 0 && (wave == 0 || (diag && wave == 2))) {
        const float4 q = pk[ta * WAVE + lane];
        const float4 t = make_float4(q.x, q.y, c.lam * q.z, c.lam * q.w);
        sh.trav[tsel][lane] = t; sh.trav[tsel][lane + WAVE] = t;
        if (RAD) { const float r_ = radius[ta * WAVE + lane]; sh.radt[tsel][lane] = r_; sh.radt[tsel][lane + WAVE] = r_; }
        if (Z3) { const float2 qz = zv[ta * WAVE + lane]; const float2 tz = make_float2(qz.x, c.lam * qz.y); sh.travz[tsel][lane] = tz; sh.travz[tsel][lane + WAVE] = tz; }
    }
    __syncthreads();
    if (ta >= 0) {
        const float4 pj = pk[tb * WAVE + lane];
        float rj = 0.f;
        if (RAD) rj = radius[tb * WAVE + lane];
        float zj = 0.f, ujz = 0.f;
        if (Z3) { const float2 qz = zv[tb * WAVE + lane]; zj = qz.x; ujz = c.lam * qz.y; }
        float reach2 = __builtin_inff();             // CUT: squared distance beyond which a term is < 2^-40 A (uniform)
        if (CUT && shift != 0) {
            const float reach = fmaf(sa.cut_scale, fmaf(c.lam, sa.vmax[ta] + sa.vmax[tb], 1.0f), sa.cut_pad);
            reach2 = reach * reach;
        }
      {
        // lambda v travels with i / stays with j: D = lambda (v_i - v_j) + e is then one fma per component
        const float ujx = c.lam * pj.z, ujy = c.lam * pj.w;
        const float4* trav = &sh.trav[tsel][lane + sig0];
        const float* radt = &sh.radt[tsel][lane + sig0];
        const float2* travz = &sh.travz[tsel][lane + sig0];
        // sigma = 32 on a diagonal tile meets every unordered pair {l, l+32} in BOTH lanes: one-sided there
        const int one_sided_from = one_sided ? 0 : ((diag && (sig0 + nsteps - 1 == 32)) ? nsteps - 1 : nsteps);   // uniform
        float4 T = trav[0];
        float ri = RAD ? radt[0] : 0.f;
        float2 Tz = make_float2(0.f, 0.f);
        if (Z3) Tz = travz[0];
        v2f fjv; fjv.x = 0.f; fjv.y = 0.f;       // planar: the resident sums as a register pair (v_pk_add_f32)
        // (steps are not interleaved: measured on MI355X, round 2 -- eight waves per SIMD already hide a step's dependent chain;
        //  unrolled by four only so that the slot offsets are immediates of the LDS reads)
        auto step = [&](int s_) __attribute__((always_inline)) {
            const float4 Tn = trav[s_ + 1];       // the next step's operand is in flight during this one (slot <= 127: inside the image)
            float rin = 0.f;
            if (RAD) rin = radt[s_ + 1];
            float2 Tzn = make_float2(0.f, 0.f);
            if (Z3) Tzn = travz[s_ + 1];
            __builtin_amdgcn_sched_barrier(0);
            bool done = false;
#if SFM_PK
            {                                     // the step on packed fp32 instructions (round 4): x / y pairs packed, z scalar
                v2f pjv, ujv, Tv, Uv, cv;
                float cz = 0.f;
                pjv.x = pj.x; pjv.y = pj.y; ujv.x = ujx; ujv.y = ujy; Tv.x = T.x; Tv.y = T.y; Uv.x = T.z; Uv.y = T.w;
                const bool kept = Z3 ? moussaid_spatial_pk<RAD, CUT>(c, pjv, zj, ujv, ujz, Tv, Tz.x, Uv, Tz.y, RAD ? ri + rj : 0.f, reach2, cv, cz)
                                     : moussaid_planar_pk<RAD, CUT>(c, pjv, ujv, Tv, Uv, RAD ? ri + rj : 0.f, reach2, cv);
                if (kept) {
                    fxi = rot_in(fxi) + cv.x;
                    fyi = rot_in(fyi) + cv.y;
                    if (Z3) fzi = rot_in(fzi) + cz;
                    if (s_ < one_sided_from) { fjv -= cv; if (Z3) fzj -= cz; }
                    done = true;
                }
            }
#else
            {
            const float dx = pj.x - T.x, dy = pj.y - T.y;
            const float dz = Z3 ? zj - Tz.x : 0.f;
            const float d2 = Z3 ? fmaf(dx, dx, fmaf(dy, dy, dz * dz)) : fmaf(dx, dx, dy * dy);
            if (!CUT || __any(!(d2 > reach2))) {
                float cx, cy, cz = 0.f;
                const bool kept = Z3 ? moussaid_spatial<RAD, CUT>(c, dx, dy, dz, d2, T.z - ujx, T.w - ujy, Tz.y - ujz, RAD ? ri + rj : 0.f, cx, cy, cz)
                                     : moussaid_planar<RAD, CUT>(c, dx, dy, d2, T.z - ujx, T.w - ujy, RAD ? ri + rj : 0.f, cx, cy);
                if (kept) {
                    // the sums of the pedestrian this lane has just met were in lane + 1 a step ago: rotation and add in one
                    fxi = rot_in(fxi) + cx;
                    fyi = rot_in(fyi) + cy;
                    if (Z3) fzi = rot_in(fzi) + cz;
                    if (s_ < one_sided_from) { fxj -= cx; fyj -= cy; if (Z3) fzj -= cz; }
                    done = true;
                }
            }
            }
#endif
            if (CUT && !done) { fxi = rot_in(fxi); fyi = rot_in(fyi); if (Z3) fzi = rot_in(fzi); }
            T = Tn;
            ri = rin;
            Tz = Tzn;
            __builtin_amdgcn_sched_barrier(0);
        };
        if (nsteps == 16) {                      // the normal case: fixed trip count
#pragma unroll 4
            for (int s_ = 0; s_ < 16; ++s_) step(s_);
        } else {                                 // timing probe (SymArgs::debug_steps)
#pragma unroll 1
            for (int s_ = 0; s_ < nsteps; ++s_) step(s_);
        }
        i_end_loc = (lane + sig0 + nsteps - 1) & (WAVE - 1);
        if (SFM_PK) { fxj = fjv.x; fyj = fjv.y; }
      }
    }
    sh.fi[wave][i_end_loc] = make_float2(fxi, fyi);
    sh.fj[wave][lane] = make_float2(fxj, fyj);
    if (Z3) { sh.fiz[wave][i_end_loc] = fzi; sh.fjz[wave][lane] = fzj; }
    __syncthreads();
    // Where the two rows of this item go: the dense slab (grid mode), or -- list mode, round 4 -- row pair q of the pool; a pair
    // beyond the pool's capacity adds its sums to the overflow accumulators instead (integer atomics: order-independent).
    const bool pooled = sa.idx != nullptr;
    const bool spill = pooled && (uint32_t)item >= sa.cap_pairs;
    auto put = [&](int side, int t_of, int t_from, int p, float vx_, float vy_, float vz_) {   // force on pedestrian p of tile t_of from tile t_from
        if (spill) {
            SpillArgs* sp = sa.spill;
            const size_t np_ = (size_t)sp->n_pad;
            const size_t i_ = (size_t)t_of * WAVE + p;
            constexpr float S = 68719476736.0f;                        // 2^36
            const bool fin = fabsf(vx_) < __builtin_inff() && fabsf(vy_) < __builtin_inff() && (!Z3 || fabsf(vz_) < __builtin_inff());
            unsigned long long* ov = reinterpret_cast<unsigned long long*>(sp->ovf);
            if (fin) {
                atomicAdd(ov + i_, (unsigned long long)__float2ll_rn(vx_ * S));
                atomicAdd(ov + np_ + i_, (unsigned long long)__float2ll_rn(vy_ * S));
                if (Z3) atomicAdd(ov + 2 * np_ + i_, (unsigned long long)__float2ll_rn(vz_ * S));
            } else {
                atomicAdd(ov + 3 * np_ + i_, 1ull);                    // a coincident pair's NaN: the epilogue recomputes the tile exactly
            }
            sp->tick = sa.tick_serial;
        } else {
            // row pair q = row_base + item of the pool, or the dense slab's (partner tile, pedestrian) entry
            const size_t r = pooled ? ((size_t)2 * (sa.row_base + (uint32_t)item) + (size_t)side) * WAVE + p
                                    : (size_t)t_from * sa.stride + t_of * WAVE + p;
            sa.slab[r] = make_float2(vx_, vy_);
            if (Z3) sa.slabz[r] = vz_;
        }
    };
    if (shift == 0) {
        // waves 0,1 -> tile bx ; waves 2,3 -> tile bx + half_up
        if (tid < 2 * WAVE) {
            const int g = tid >> 6;                       // 0 or 1
            const int t = (g == 0) ? bx : bx + half_up;
            if (t < sa.t_hi) {
                const int p = tid & (WAVE - 1);
                const float2 a0 = sh.fi[2 * g][p], a1 = sh.fi[2 * g + 1][p], b0 = sh.fj[2 * g][p], b1 = sh.fj[2 * g + 1][p];
                put(g, t, t, p, ((a0.x + a1.x) + b0.x) + b1.x, ((a0.y + a1.y) + b0.y) + b1.y,
                    Z3 ? ((sh.fiz[2 * g][p] + sh.fiz[2 * g + 1][p]) + sh.fjz[2 * g][p]) + sh.fjz[2 * g + 1][p] : 0.f);
            }
        }
    } else if (tid < 2 * WAVE) {
        const int p = tid & (WAVE - 1);
        if (tid < WAVE) {     // force on tile ta's pedestrians from tile tb
            const float2 a0 = sh.fi[0][p], a1 = sh.fi[1][p], a2 = sh.fi[2][p], a3 = sh.fi[3][p];
            put(0, ta, tb, p, ((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y,
                Z3 ? ((sh.fiz[0][p] + sh.fiz[1][p]) + sh.fiz[2][p]) + sh.fiz[3][p] : 0.f);
        } else if (!one_sided) {   // force on tile tb's pedestrians from tile ta
            const float2 a0 = sh.fj[0][p], a1 = sh.fj[1][p], a2 = sh.fj[2][p], a3 = sh.fj[3][p];
            put(1, tb, ta, p, ((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y,
                Z3 ? ((sh.fjz[0][p] + sh.fjz[1][p]) + sh.fjz[2][p]) + sh.fjz[3][p] : 0.f);
        }
    }
    if (sa.work) __syncthreads();                 // LDS is reused by the next item
  }
    if (sa.stamps && tid == 0 && (size_t)bid_y * grid_x + bid_x < (size_t)PAIR_STAMP_WGS) {
        const size_t b = (size_t)bid_y * grid_x + bid_x;
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        sa.stamps[3 * b] = t_start;
        sa.stamps[3 * b + 1] = __builtin_amdgcn_s_memrealtime();
        sa.stamps[3 * b + 2] = ((unsigned long long)xcc << 32) | hw;
    }
}

// CUT: the provably negligible part of a tile pair is not evaluated.  reach = the distance beyond which a
// term is below 2^-40 A given the largest speeds of the two tiles (tiles_negligible's bound, per pair instead of per box):
// a systolic step whose 64 pairs are ALL farther apart than that costs two subtractions, two fmas, a compare and the
// rotation.  The test is on the pair's own distance, so it needs no agreement with anything else.
template <bool RAD, bool CUT, bool Z3>
__global__ __launch_bounds__(BLOCK) void sfm_pair_sym_kernel(const float4* __restrict__ pk, const float2* __restrict__ zv, const float* __restrict__ radius,
                                                             const IxConst c, const SymArgs sa) {
    __shared__ PairShared sh;
    pair_block<RAD, CUT, Z3>(pk, zv, radius, c, sa, sh, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)threadIdx.x);
}

// Geometry workgroups in front of the pair kernel's workgroups in ONE launch (whole mid-sized crowd, list cutoff): the border /
// obstacle forces only need the tick's input state and their workgroups are chains of memory round trips, the pair workgroups
// are VALU work -- side by side on the CUs they overlap, which two launches on two streams do not at this size.
// The geometry part runs as 4-wave workgroups, geo_slices of them per tile (the same 16 waves per tile as sfm_geometry_kernel),
// so that both kinds share the block size; they come first in the grid and are therefore dispatched first.
template <bool RAD, bool CUT, bool Z3>
__global__ __launch_bounds__(BLOCK) void sfm_pair_geo_kernel(const TickArgs a, const SymArgs sa, int geo_tiles, int geo_stride) {
    constexpr size_t LDS = sizeof(GeoShared<WAVES_PER_BLOCK>) > sizeof(PairShared) ? sizeof(GeoShared<WAVES_PER_BLOCK>) : sizeof(PairShared);
    __shared__ __attribute__((aligned(16))) char smem[LDS];
    const int n_geo = geo_tiles * a.geo_slices;
    // geo_stride 1: the geometry workgroups come first (they take half the wave slots of a mid-sized crowd's first round);
    // geo_stride s > 1: every s-th workgroup of the grid is one of them -- a large crowd has several rounds of them, and in front
    // of the pair workgroups they would hold every slot for those rounds
    const int bid = (int)blockIdx.x;
    const int g = bid / geo_stride;
    const bool is_geo = g < n_geo && bid == g * geo_stride;
    if (is_geo) {
        geometry_block<RAD, WAVES_PER_BLOCK>(a, *reinterpret_cast<GeoShared<WAVES_PER_BLOCK>*>(smem), g % geo_tiles, g / geo_tiles, a.geo_slices,
                                             geo_tiles, (int)threadIdx.x);
    } else {
        const int pb = bid - min(n_geo, g + 1);   // index among the pair workgroups: a run of the list, or (bx, shift) of the 2-D grid
        if (sa.work) pair_block<RAD, CUT, Z3>(a.pk_cur, a.zv_cur, a.radius, a.ped, sa, *reinterpret_cast<PairShared*>(smem), pb, 0, (int)gridDim.x - n_geo, (int)threadIdx.x);
        else pair_block<RAD, CUT, Z3>(a.pk_cur, a.zv_cur, a.radius, a.ped, sa, *reinterpret_cast<PairShared*>(smem), pb % sa.n_t, pb / sa.n_t, sa.n_t, (int)threadIdx.x);
    }
}

// List mode (round 4): the pool row that holds the force on tile t's pedestrians from tile u -- the same canonical form as the list
// kernels' items (list_candidate): an own-own pair is listed once, by the tile from which the other is at most n_t / 2 ahead (the
// antipodal tie goes to the lower half); a pair with another rank's tile by the own tile; a tile's own pairs by the diagonal item
// it shares with tile + half_up.  An entry of 0xffffffff: the pair went to the overflow accumulators.
__device__ __forceinline__ size_t pool_slot(const SymArgs& sa, int t, int u, uint32_t& side) {   // where the pair's entry of idx[] is
    const int n_t = sa.n_t;
    int owner = t, shift = 0;
    side = 0u;
    if (u == t) {
        const int half_up = (sa.t_hi - sa.t_lo + 1) >> 1;
        if (t - sa.t_lo >= half_up) { owner = t - half_up; side = 1u; }
    } else {
        int s_ = u - t;
        if (s_ < 0) s_ += n_t;
        shift = s_;
        if (u >= sa.t_lo && u < sa.t_hi) {
            const int s2 = n_t - s_;
            if (s2 < s_ || (s2 == s_ && t >= (n_t >> 1))) { owner = u; shift = s2; side = 1u; }
        }
    }
    return (size_t)(owner - sa.t_lo) * (size_t)n_t + shift;
}

constexpr int EPI_WAVES = 16;
#ifndef EPI_INFLIGHT
#define EPI_INFLIGHT 8                     // slab-row loads a wave keeps in flight under a cutoff (a multiple of 4; the sum's association follows it)
#endif

template <bool RAD, int EW, bool Z3>
__global__ __launch_bounds__(EW * WAVE) void sfm_sym_epilogue_kernel(const TickArgs a, const SymArgs sa) {
    __shared__ float2 s_sum[EW][WAVE];
    __shared__ float2 s_exact[WAVE];
    __shared__ float s_sumz[Z3 ? EW : 1][WAVE];     // 3-D crowds: the z components
    __shared__ float s_exactz[WAVE];
    __shared__ int s_bad[EW];
    __shared__ uint16_t s_own[EW][4096 / EW];   // two-level cutoff: the strips a wave sums (n_strips <= n_t <= 4096 under a cutoff)
    __shared__ uint32_t s_kept[EW][4 * WAVE];   // ... and the kept tiles of four of them (list mode with the row pool: their rows)
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = uniform(tid >> 6);
    if (a.adv.M > 0 && (int)blockIdx.x >= a.adv.block0) {            // the extra workgroups: vehicles move on
        const int k = ((int)blockIdx.x - a.adv.block0) * EW + wave;
        if (k < a.adv.M) advance_vehicle(a.adv, k, lane, true);
        return;
    }
    if (sa.zero_count && blockIdx.x == 0 && tid == 0) {     // this tick's pair kernels have read them: keep copies for sfm_get_pair_work
        int* wc = const_cast<int*>(sa.work_count);
        wc[2] = wc[0];
        wc[0] = 0;
        if (sa.zero_count > 1) { wc[3] = wc[1]; wc[1] = 0; }   // split tick: the own-own list's counter too
    }
    const int t = sa.t_lo + blockIdx.x;
    const int N = a.N;
    const int i_end = a.i_end;                             // rows of this handle end here (whole crowd: N)
    const int i = t * WAVE + lane;

    // own state first: independent of the slab, so its latency overlaps the column sum
    float4 st = make_float4(0.f, 0.f, 0.f, 0.f), o = st;
    float2 stz = make_float2(0.f, 0.f);
    uint32_t nd0 = 0;
    if (wave == 0 && i < i_end) { st = a.pk_cur[i]; o = a.own[i]; if (a.flags & 2u) nd0 = a.draws[i]; if (Z3) stz = a.zv_cur[i]; }

    // 1. slab column sums
    float2 acc = make_float2(0.f, 0.f);
    float accz = 0.f;
    const float* colz = Z3 ? sa.slabz + i : nullptr;
    if (a.en_ped && a.tile_box && sa.n_strips > 0) {
        // cutoff on, large crowd (sfm_pair_list2_kernel's two levels: a negligible strip holds only negligible tiles).  Three
        // passes, each with all of its loads in flight together instead of a round trip per strip (this kernel is a latency
        // chain: 58 -> 36 us at c5): (1) lanes over the strips -- the survivors are dealt round-robin to the waves; (2) the tiles
        // of four own strips at a time, lanes over tiles -- the kept ones go to a list in LDS; (3) their slab rows, EPI_INFLIGHT
        // at a time.  Only the rows of evaluated tile pairs are read, in a fixed order (deterministic).
        const float2* col = sa.slab + i;
        const float4 bt = a.tile_box[t];
        const float vt = a.tile_vmax[t];
        uint16_t* own = s_own[wave];
        uint32_t* kept = s_kept[wave];
        const bool pooled = sa.idx != nullptr;
        int n_alive = 0;
        for (int s0 = 0; s0 < sa.n_strips; s0 += WAVE) {
            const int s = s0 + lane;
            const int sc = min(s, sa.n_strips - 1);
            const bool alive = s < sa.n_strips && !tiles_negligible(bt, vt, sa.sbox[sc], sa.svmax[sc], a.ped.lam, a.cut_scale, a.cut_pad);
            const unsigned long long ms = __ballot(alive);
            const int rank = n_alive + __popcll(ms & ((1ull << lane) - 1ull));
            if (alive && (rank % EW) == wave) own[rank / EW] = (uint16_t)s;
            n_alive += __popcll(ms);
        }
        const int n_own = n_alive > wave ? (n_alive - wave - 1) / EW + 1 : 0;   // survivors of rank wave, wave + EW, ...
        const int cps = (sa.tps + WAVE - 1) / WAVE;                             // 64-tile chunks per strip
        const int n_units = n_own * cps;
        for (int k = 0; k < n_units; k += 4) {
            float4 tb[4];
            float tv[4];
            int uu[4];
            bool ok[4];
            uint32_t qv[4] = {0u, 0u, 0u, 0u}, sd[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = min(k + j, n_units - 1);
                const int q0 = (e % cps) * WAVE;
                const int u = own[e / cps] * sa.tps + q0 + lane;
                ok[j] = k + j < n_units && q0 + lane < sa.tps && u < sa.n_t;
                uu[j] = ok[j] ? u : t;
                tb[j] = a.tile_box[uu[j]];
                tv[j] = a.tile_vmax[uu[j]];
                // (the row's index is fetched WITH the box, for every candidate -- a lookup after the test would be one more dependent
                //  round trip in a kernel that is a chain of them; a candidate that fails the test has a stale entry, never used)
                if (pooled) qv[j] = sa.idx[pool_slot(sa, t, uu[j], sd[j])];
            }
            int n_kept = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bool keep = ok[j] && !tiles_negligible(bt, vt, tb[j], tv[j], a.ped.lam, a.cut_scale, a.cut_pad);
                uint32_t where = (uint32_t)uu[j];
                if (keep && pooled) { where = 2u * qv[j] + sd[j]; keep = qv[j] != 0xffffffffu; }   // (spilled pairs are in the overflow sums)
                const unsigned long long m = __ballot(keep);
                if (keep) kept[n_kept + __popcll(m & ((1ull << lane) - 1ull))] = where;
                n_kept += __popcll(m);
            }
            for (int r = 0; r < n_kept; r += EPI_INFLIGHT) {
                float2 v[EPI_INFLIGHT];
                float vz[EPI_INFLIGHT];
#pragma unroll
                for (int q = 0; q < EPI_INFLIGHT; ++q) {
                    const size_t at = r + q < n_kept ? (pooled ? (size_t)kept[r + q] * WAVE + lane : (size_t)kept[r + q] * sa.stride) : 0;
                    v[q] = (r + q < n_kept) ? (pooled ? sa.slab[at] : col[at]) : make_float2(0.f, 0.f);
                    vz[q] = (Z3 && r + q < n_kept) ? (pooled ? sa.slabz[at] : colz[at]) : 0.f;
                }
#pragma unroll
                for (int q = 0; q < EPI_INFLIGHT; q += 4) {
                    acc.x += (v[q].x + v[q + 1].x) + (v[q + 2].x + v[q + 3].x);
                    acc.y += (v[q].y + v[q + 1].y) + (v[q + 2].y + v[q + 3].y);
                    if (Z3) accz += (vz[q] + vz[q + 1]) + (vz[q + 2] + vz[q + 3]);
                }
            }
        }
    } else if (a.en_ped && a.tile_box) {
        // cutoff on, flat: wave w owns partner tiles w, w + EW, ...; its lanes test 64 of them at a time, four such chunks with
        // their boxes in flight together, the kept tiles go to a list in LDS and their rows are read EPI_INFLIGHT at a time
        // (only the rows of evaluated tile pairs, ascending order within a chunk: deterministic)
        const float2* col = sa.slab + i;
        const float4 bt = a.tile_box[t];
        const float vt = a.tile_vmax[t];
        uint32_t* kept = s_kept[wave];
        const bool pooled = sa.idx != nullptr;
        for (int base = wave; base < sa.n_t; base += 4 * EW * WAVE) {
            float4 tb[4];
            float tv[4];
            int uu[4];
            bool ok[4];
            uint32_t qv[4] = {0u, 0u, 0u, 0u}, sd[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = base + j * EW * WAVE + EW * lane;
                ok[j] = u < sa.n_t;
                uu[j] = ok[j] ? u : t;
                tb[j] = a.tile_box[uu[j]];
                tv[j] = a.tile_vmax[uu[j]];
                if (pooled) qv[j] = sa.idx[pool_slot(sa, t, uu[j], sd[j])];
            }
            int n_kept = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bool keep = ok[j] && !tiles_negligible(bt, vt, tb[j], tv[j], a.ped.lam, a.cut_scale, a.cut_pad);
                uint32_t where = (uint32_t)uu[j];
                if (keep && pooled) { where = 2u * qv[j] + sd[j]; keep = qv[j] != 0xffffffffu; }   // (spilled pairs are in the overflow sums)
                const unsigned long long m = __ballot(keep);
                if (keep) kept[n_kept + __popcll(m & ((1ull << lane) - 1ull))] = where;
                n_kept += __popcll(m);
            }
            for (int r = 0; r < n_kept; r += EPI_INFLIGHT) {
                float2 v[EPI_INFLIGHT];
                float vz[EPI_INFLIGHT];
#pragma unroll
                for (int q = 0; q < EPI_INFLIGHT; ++q) {
                    const size_t at = r + q < n_kept ? (pooled ? (size_t)kept[r + q] * WAVE + lane : (size_t)kept[r + q] * sa.stride) : 0;
                    v[q] = (r + q < n_kept) ? (pooled ? sa.slab[at] : col[at]) : make_float2(0.f, 0.f);
                    vz[q] = (Z3 && r + q < n_kept) ? (pooled ? sa.slabz[at] : colz[at]) : 0.f;
                }
#pragma unroll
                for (int q = 0; q < EPI_INFLIGHT; q += 4) {
                    acc.x += (v[q].x + v[q + 1].x) + (v[q + 2].x + v[q + 3].x);
                    acc.y += (v[q].y + v[q + 1].y) + (v[q + 2].y + v[q + 3].y);
                    if (Z3) accz += (vz[q] + vz[q + 1]) + (vz[q + 2] + vz[q + 3]);
                }
            }
        }
    } else if (a.en_ped) {                                   // wave w takes partner tiles w, w+16, ...
        const float2* col = sa.slab + i;
        int u = wave;
        for (; u + 3 * EW < sa.n_t; u += 4 * EW) {          // 4 independent loads in flight
            const float2 v0 = col[(size_t)u * sa.stride], v1 = col[(size_t)(u + EW) * sa.stride];
            const float2 v2 = col[(size_t)(u + 2 * EW) * sa.stride], v3 = col[(size_t)(u + 3 * EW) * sa.stride];
            acc.x += (v0.x + v1.x) + (v2.x + v3.x);
            acc.y += (v0.y + v1.y) + (v2.y + v3.y);
            if (Z3) accz += (colz[(size_t)u * sa.stride] + colz[(size_t)(u + EW) * sa.stride]) + (colz[(size_t)(u + 2 * EW) * sa.stride] + colz[(size_t)(u + 3 * EW) * sa.stride]);
        }
        for (; u < sa.n_t; u += EW) {
            const float2 v = col[(size_t)u * sa.stride];
            acc.x += v.x;
            acc.y += v.y;
            if (Z3) accz += colz[(size_t)u * sa.stride];
        }
    }
    s_sum[wave][lane] = acc;
    if (Z3) s_sumz[Z3 ? wave : 0][lane] = accz;
    // A coincident pair leaves a NaN (moussaid_planar) in the sums of both pedestrians: any non-finite partial sum sends the
    // whole tile through the exact body below.
    // (list mode with the row pool: pairs beyond its capacity left their sums in ovf[] -- 2^-36 fixed point, this tick's serial in
    //  ovf_tick -- and a non-finite one a count in the fourth plane; wave 0 takes them in and clears them for the next tick)
    const bool spilled = sa.idx != nullptr && a.en_ped && sa.spill->tick == sa.tick_serial;
    long long ov[4] = {0, 0, 0, 0};
    if (spilled && wave == 0 && i < a.N_pad) {
        long long* ovf = sa.spill->ovf;
#pragma unroll
        for (int k = 0; k < 4; ++k) { ov[k] = ovf[(size_t)k * a.N_pad + i]; ovf[(size_t)k * a.N_pad + i] = 0; }
    }
    {
        const bool bad = !(fabsf(acc.x) < __builtin_inff()) || !(fabsf(acc.y) < __builtin_inff()) || (Z3 && !(fabsf(accz) < __builtin_inff())) || ov[3] != 0;
        if (lane == 0) s_bad[wave] = 0;
        if (bad) s_bad[wave] = 1;                                  // same wave, LDS operations in order
    }
    __syncthreads();
    bool exact = false;                                            // uniform per workgroup
    if (a.en_ped) {
        int any_bad = 0;
#pragma unroll
        for (int w = 0; w < EW; ++w) any_bad |= s_bad[w];
        exact = uniform(any_bad) != 0;
    }

    // 2. coincident pairs in this tile: recompute its rows with the exact ordered body (rare)
    if (exact) {
        for (int p = wave; p < WAVE; p += EW) {
            const int ip = t * WAVE + p;
            if (ip >= i_end) break;
            const float4 si = a.pk_cur[ip];
            const float xi = uniform(si.x), yi = uniform(si.y), vxi = uniform(si.z), vyi = uniform(si.w);
            float zi = 0.f, vzi = 0.f;
            if (Z3) { const float2 sz = a.zv_cur[ip]; zi = uniform(sz.x); vzi = uniform(sz.y); }
            float gx = 0.f, gy = 0.f, gz = 0.f;
            for (int j0 = 0; j0 < N; j0 += WAVE) {
                const int j = j0 + lane;
                const float4 pj = a.pk_cur[min(j, N - 1)];
                float2 pz = make_float2(0.f, 0.f);
                if (Z3) pz = a.zv_cur[min(j, N - 1)];
                float cx = 0.f, cy = 0.f, cz = 0.f, rinv;
                moussaid<Z3, RAD, true>(a.ped, pj.x - xi, pj.y - yi, pz.x - zi, vxi - pj.z, vyi - pj.w, vzi - pz.y,
                                        RAD ? a.radius[ip] + a.radius[min(j, N - 1)] : 0.f, cx, cy, cz, rinv);
                const bool valid = (j < N) & (j != ip);
                gx += valid ? cx : 0.f;
                gy += valid ? cy : 0.f;
                if (Z3) gz += valid ? cz : 0.f;
            }
            gx = wave_sum(gx);
            gy = wave_sum(gy);
            if (Z3) gz = wave_sum(gz);
            if (lane == 0) { s_exact[p] = make_float2(gx, gy); s_exactz[p] = gz; }
        }
    }
    if (exact) __syncthreads();
    if (wave != 0) return;
    const bool live = i < i_end;                           // padding rows ride along (wave reductions below) but never store

    // 4. lane-parallel epilogue for the 64 pedestrians of the tile
    float2 g = s_sum[0][lane];
#pragma unroll
    for (int w = 1; w < EW; ++w) { const float2 b = s_sum[w][lane]; g.x += b.x; g.y += b.y; }
    if (spilled) { g.x = (float)((double)g.x + (double)ov[0] * 0x1p-36); g.y = (float)((double)g.y + (double)ov[1] * 0x1p-36); }
    if (exact) g = s_exact[lane];
    float gz = 0.f;
    if (Z3) {
        gz = s_sumz[0][lane];
#pragma unroll
        for (int w = 1; w < (Z3 ? EW : 1); ++w) gz += s_sumz[w][lane];
        if (spilled) gz = (float)((double)gz + (double)ov[2] * 0x1p-36);
        if (exact) gz = s_exactz[lane];
    }
    const float fpx = a.en_ped ? a.ped.negA * g.x : 0.f, fpy = a.en_ped ? a.ped.negA * g.y : 0.f;
    const float fpz = (Z3 && a.en_ped) ? a.ped.negA * gz : 0.f;
    float fbx = 0.f, fby = 0.f, fsx = 0.f, fsy = 0.f, fdx = 0.f, fdy = 0.f;
    if (a.geo) {
        const size_t np_ = (size_t)a.N_pad;
        for (int sl = 0; sl < a.geo_slices; ++sl) {
            const float* gsl = a.geo + (size_t)sl * 6 * np_ + i;
            fbx += gsl[0 * np_]; fby += gsl[1 * np_]; fsx += gsl[2 * np_];
            fsy += gsl[3 * np_]; fdx += gsl[4 * np_]; fdy += gsl[5 * np_];
        }
    }
    const float x = st.x, y = st.y, vx = st.z, vy = st.w, ts = o.z;
    const float z = stz.x, vz = stz.y;
    float wx = o.x, wy = o.y;
    float fax = 0.f, fay = 0.f, faz = 0.f;
    if (a.en_acc) {
        const float tx_ = wx - x, ty_ = wy - y;
        const float nrm = sqrtf(fmaf(tx_, tx_, ty_ * ty_));
        const float inv = (nrm == 0.0f) ? 1.0f : 1.0f / nrm;
        fax = (ts * (tx_ * inv) - vx) * a.inv_tau;
        fay = (ts * (ty_ * inv) - vy) * a.inv_tau;
        if (Z3) faz = (0.0f - vz) * a.inv_tau;                     // the desired direction has no z (stateutils.py:12-13)
    }
    const float Fx = (((fax + fpx) + fbx) + fsx) + fdx;
    const float Fy = (((fay + fpy) + fby) + fsy) + fdy;
    const float Fz = faz + fpz;
    float nvx = fmaf(a.dt, Fx, vx), nvy = fmaf(a.dt, Fy, vy), nvz = Z3 ? fmaf(a.dt, Fz, vz) : 0.f;
    float sp = Z3 ? sqrtf(fmaf(nvx, nvx, fmaf(nvy, nvy, nvz * nvz))) : sqrtf(fmaf(nvx, nvx, nvy * nvy));   // cap on the 3-D speed (stateutils.py:18-23)
    sp = (sp == 0.0f) ? 1.0f : sp;
    const float fac = fminf(1.0f, (ts * a.max_speed_factor) / sp);
    nvx *= fac; nvy *= fac; nvz *= fac;
    const uint32_t pid = live ? (a.ids ? a.ids[i] : (uint32_t)i) : 0u;
    bool despawn = false;
    const bool gone = live && a.fsm.mode && a.fsm.mode[pid] == MODE_DESPAWNED;
    {
        const float ax_ = wx - x, ay_ = wy - y;
        const bool here = fmaf(ax_, ax_, ay_ * ay_) < a.arrive_thr2;
        if (a.fsm.mode) {
            bool changed = false;
            if (here && !gone && live) fsm_arrived(a, pid, wx, wy, changed, despawn);
            if (changed) a.own[i] = make_float4(wx, wy, o.z, o.w);
        } else if ((a.flags & 2u) && here && live) {
            const uint32_t nd = nd0 + 1u;
            wx = waypoint_coord(a.seed, pid, nd, 0u, a.world_side);
            wy = waypoint_coord(a.seed, pid, nd, 1u, a.world_side);
            a.own[i] = make_float4(wx, wy, o.z, o.w);
            a.draws[i] = nd;
        }
    }
    float nx = x, ny = y, nz = z;
    if (a.flags & 1u) { nx = fmaf(a.dt, nvx, x); ny = fmaf(a.dt, nvy, y); if (Z3) nz = fmaf(a.dt, nvz, z); }
    if (despawn || gone) { const float2 pp = park_position(pid); nx = pp.x; ny = pp.y; nvx = 0.f; nvy = 0.f; nvz = 0.f; }
    if (live) a.pk_next[i] = make_float4(nx, ny, nvx, nvy);
    if (Z3 && live) a.zv_next[i] = make_float2(nz, nvz);
    if (a.host_pk && live) { a.host_pk[i] = make_float4(nx, ny, nvx, nvy); if (Z3) a.host_zv[i] = make_float2(nz, nvz); }
    if (a.tile_box_out) {                             // whole crowd under the list cutoff: box and largest speed of the tile in the NEXT state
        const float inf = __builtin_inff();
        const bool in_box = live && fabsf(nx) < 1.0e14f;
        float x0 = in_box ? nx : inf, y0 = in_box ? ny : inf, x1 = in_box ? nx : -inf, y1 = in_box ? ny : -inf;
        float v = in_box ? sqrtf(fmaf(nvx, nvx, fmaf(nvy, nvy, nvz * nvz))) * 1.000001f : 0.0f;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            x0 = fminf(x0, __shfl_xor(x0, m)); y0 = fminf(y0, __shfl_xor(y0, m));
            x1 = fmaxf(x1, __shfl_xor(x1, m)); y1 = fmaxf(y1, __shfl_xor(y1, m));
            v = fmaxf(v, __shfl_xor(v, m));
        }
        if (lane == 0) { a.tile_box_out[t] = make_float4(x0, y0, x1, y1); a.tile_vmax_out[t] = v; }
    }
    if (a.rec && live) {
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

// ------------------------------------------------------------------------------------------------------
// fused tick: one launch per tick for a whole planar crowd with the acceleration and pedestrian forces only
// ------------------------------------------------------------------------------------------------------
// The two-kernel tick above spends 4.5 of its 19.5 us at N = 4096 on the epilogue launch and the two kernel boundaries around
// it.  Here the epilogue of tick t is the PROLOGUE of tick t+1's pair kernel: tiles go in groups of two, a workgroup (16 waves)
// owns one unordered pair of groups (GX, GY) = four tile pairs, or two diagonal groups.  It first integrates its own 256
// pedestrians from the previous launch's partial forces (a column sum over n_g slab rows, split over the eight 128-thread
// parts of the workgroup), keeps the new state in LDS, and then runs the systolic steps of sfm_pair_sym_kernel on it: every
// wave 16 steps.
// A tile is integrated by every workgroup that needs it -- the same arithmetic on the same operands, hence the same bits --
// and stored by the workgroup of its group's diagonal item.  No flag, no atomic, no fence: the only synchronisation is the
// kernel boundary, and state / waypoints / partial forces ping-pong across it.  Grouping tiles by two halves the slab rows a
// pedestrian's sum runs over (n_g = n_t / 2), which is what makes reading them in every workgroup affordable.
// A coincident pair leaves a NaN in the sums of its two pedestrians, as in the two-kernel path: those rows are recomputed with
// the exact body (by every workgroup that holds them).
constexpr int GROUP = 2 * WAVE;                  // pedestrians per group of two tiles

// The travelling tile reaches the lanes from LDS (round 3; round 2 rotated it through the wavefront in registers, six v_mov_b32_dpp
// per step): the tile sits in LDS twice back to back, {x, y, lambda vx, lambda vy} per pedestrian, and step s of a lane is ONE
// ds_read_b128 at an immediate offset 16 s from the lane's base address -- no operand moves between lanes on the VALU.  Only the
// sums of the travelling side still rotate, and their rotation is folded into the add (v_add_f32_dpp wave_rol:1: the sum arrives
// from the neighbouring lane and takes this step's term in one instruction).  60 -> 54 issued VALU instructions per step: c2 17.55
// -> 16.54 us.  Needs wave_rol:1 to hand lane l the value of lane l+1 (probed at init; otherwise the symmetric path is off).
// (Measured and dropped: the sums in LDS as well, by ds_add_f32 -- LDS float atomics run at ~2 cycles per LANE: 88.8 against 17.5 us.)
// Z3: a 3-D crowd -- {z, lambda vz} travel beside (one more ds_read_b64), the body is moussaid_spatial, the sums have a z component
// that goes through slabz rows, and the integration is the 3-D one of sfm_sym_epilogue_kernel.
// GEO (round 3): the crowd also feels border / obstacle forces.  Their workgroups are a second ROLE of the same launch -- blocks
// [0, n_geo_wg): one workgroup per (tile, slice of the polylines); it integrates its tile's group exactly like a pair workgroup
// does (same code, same bits, nothing stored) and then runs the geometry kernel's body on the new state, leaving ONE float2 per
// pedestrian and slice in geo_next; every integrating prologue adds the previous launch's geo_prev rows to the force.  Vehicles that
// move on the device are a third role (blocks behind the pair workgroups, a wave per vehicle): they write the NEXT launch's centres and
// rings into the other half of a ping-pong while this launch's geometry workgroups read the current one.
template <bool RAD, int NW, bool Z3>
struct FusedShared {                             // LDS of one pair-role workgroup
    float4 st[2 * GROUP];                        // the workgroup's pedestrians in the state the pairs are evaluated on: GX then GY
    float rad[2 * GROUP];
    float2 q[Z3 ? 1 : NW / 2][2 * GROUP];        // partial column sums (3-D crowds: q / qz lie over fi / fj / fiz / fjz -- the prologue is done with
                                                 // them two barriers before the first sum of a wave is written -- which pays for the second chain's
                                                 // sums below: the 8-wave 3-D form stays at three workgroups per CU)
    float2 part[2 * GROUP];                      // exact rows
    int badrow[2 * GROUP];
    int any;
    float2 fi[NW][WAVE];
    float2 fj[NW][WAVE];
    float fiz[Z3 ? NW : 1][WAVE];                // (3-D: z components; these four arrays are one block of (NW / 2) * 2 * GROUP * 12 bytes = q + qz)
    float fjz[Z3 ? NW : 1][WAVE];
    static constexpr bool X2 = SFM_X2;           // two pairs per lane: every form of the pair role (round 4, late: also with radii, also 3-D)
    float2 fi2[X2 ? NW : 1][WAVE];               // ... the sums of a wave's second travelling chain
    float fi2z[(X2 && Z3) ? NW : 1][WAVE];
    float4 trav[4][2 * WAVE];                    // the four tiles as travelling operands, each twice back to back (two pairs per lane: as four
                                                 // planes x, y, lambda vx, lambda vy of [4][2 * WAVE] floats -- a lane reads its two pedestrians' x as one ds_read2_b32)
    float radt[RAD ? 4 : 1][2 * WAVE];
    float2 stz[Z3 ? 2 * GROUP : 1];              // 3-D crowds: {z, vz} of the same pedestrians ...
    float2 travz[Z3 ? 4 : 1][2 * WAVE];          // ... {z, lambda vz} as travelling operands
    float partz[Z3 ? 2 * GROUP : 1];
    __device__ __forceinline__ float2 (*qv())[2 * GROUP] { return Z3 ? reinterpret_cast<float2 (*)[2 * GROUP]>(&fi[0][0]) : q; }
    __device__ __forceinline__ float (*qzv())[2 * GROUP] {       // 3-D only: behind the (NW / 2) x 2 GROUP float2 of q
        return reinterpret_cast<float (*)[2 * GROUP]>(reinterpret_cast<char*>(&fi[0][0]) + sizeof(float2) * (NW / 2) * 2 * GROUP);
    }
};

#ifdef SFM_EXPERIMENTS
// per workgroup FUSED_STAMP_STRIDE words: [0, 5) thread 0 at entry, column sums in, state in LDS, its steps done, its row stored; then per
// WAVE [5 + w] the end of its last systolic step and [5 + 16 + w] the end of its share of the row store (round 4: is "steps done ->
// rows stored" of thread 0 a store tail or the skew between the workgroup's waves?)
#define FUSED_STAMP(k) do { if (f.stamps && threadIdx.x == 0 && blockIdx.x < FUSED_STAMP_WGS) f.stamps[FUSED_STAMP_STRIDE * (size_t)blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define FUSED_WAVE_STAMP(k) do { if (f.stamps && (threadIdx.x & 63) == 0 && blockIdx.x < FUSED_STAMP_WGS) f.stamps[FUSED_STAMP_STRIDE * (size_t)blockIdx.x + 5 + (k) * 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FUSED_STAMP(k) do { } while (0)
#define FUSED_WAVE_STAMP(k) do { } while (0)
#endif

template <bool RAD, int NW, bool Z3, bool GEO>   // NW waves per workgroup (8 or 16): 256 / NW systolic steps each
__global__ __launch_bounds__(NW * WAVE, (NW == 8 && Z3) ? 6 : 8) void sfm_fused_tick_kernel(const TickArgs a, const FusedArgs f) {   // 8 waves per SIMD: two 16-wave (four 8-wave) workgroups per CU (3-D, 8 waves: LDS allows three)
    constexpr int PARTS = NW / 2;                // the slab rows are split over this many 128-thread parts
    constexpr int SPW = 4 * WAVE / NW;           // systolic steps per wave
    constexpr int D = NW / 8;                    // waves per diagonal tile
    constexpr bool X2 = FusedShared<RAD, NW, Z3>::X2;   // two pairs per lane (moussaid_planar_x2 / moussaid_spatial_x2)
    using Sh = FusedShared<RAD, NW, Z3>;
    constexpr size_t LDS = (GEO && sizeof(GeoShared<NW>) > sizeof(Sh)) ? sizeof(GeoShared<NW>) : sizeof(Sh);
    __shared__ __attribute__((aligned(16))) char smem[LDS];
    Sh& sh = *reinterpret_cast<Sh*>(smem);
    static_assert(!Z3 || (offsetof(Sh, fjz) + sizeof(sh.fjz) - offsetof(Sh, fi) == (sizeof(float2) + sizeof(float)) * (NW / 2) * 2 * GROUP &&
                          offsetof(Sh, fj) == offsetof(Sh, fi) + sizeof(sh.fi) && offsetof(Sh, fiz) == offsetof(Sh, fj) + sizeof(sh.fj) &&
                          offsetof(Sh, fjz) == offsetof(Sh, fiz) + sizeof(sh.fiz)),
                  "3-D: q / qz are laid over fi | fj | fiz | fjz, which must be one block of exactly that size");
    FUSED_STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = uniform(tid >> 6);
    const int n_g = f.n_g, n_t = f.n_t;
    const int half_up = (n_g + 1) >> 1;          // diagonal items pair group bx with group bx + half_up
    // roles: [0, n_geo_wg) border / obstacle forces of one (tile, slice); then the pair workgroups; then the vehicles
    const bool geo_role = GEO && (int)blockIdx.x < f.n_geo_wg;
    const int bid = (int)blockIdx.x - (GEO ? f.n_geo_wg : 0);
    if (GEO && bid >= f.n_pair_wg) {                                   // the vehicles move on: V(in) -> V(out), a wave each
        const int k = (bid - f.n_pair_wg) * NW + wave;
        if (k < a.adv.M) advance_vehicle(a.adv, k, lane, true);
        return;
    }
    const int geo_tile = geo_role ? (int)blockIdx.x % n_t : 0, geo_slice = geo_role ? (int)blockIdx.x / n_t : 0;
    // Work items, one per workgroup and no idle workgroup in the grid (the dispatcher deals workgroup w to CU w mod 256 whatever
    // it holds, so a grid with holes leaves some CUs a whole workgroup more than others).
    int GX, GY;
    bool diag_item;
    if (geo_role) {
        // a geometry workgroup integrates the GROUP of its tile as a pair workgroup would (GY absent, nothing stored)
        GX = geo_tile >> 1; GY = n_g;
        diag_item = false;
    } else if (f.blocked) {
        // n_g a multiple of 8: the groups go in four bands of s, and the workgroups that land on one XCD (w mod 8, observed on
        // MI355X; nothing but speed depends on it) take their group pairs from one pair of bands -- XCDs 0-5 the six pairs of
        // different bands (s x s workgroups each), XCDs 6 and 7 two bands' inner pairs each (s/2 diagonal items of neighbouring
        // groups + s (s-1) / 2 pairs per band = s x s for two bands).  An XCD's L2 then fetches the column sums of 2 s groups
        // instead of all 4 s: half the traffic across the fabric at the start of the launch.
        const int s = n_g >> 2;
        const int x = bid & 7;
        int k = bid >> 3;
        if (x < 6) {
            const int I = x < 3 ? 0 : (x < 5 ? 1 : 2), J = x < 3 ? x + 1 : (x < 5 ? x - 1 : 3);
            GX = I * s + k / s;
            GY = J * s + k % s;
            diag_item = false;
        } else {
            const int per_band = (s * s) >> 1;
            const int I = 2 * (x - 6) + (k >= per_band ? 1 : 0);
            if (k >= per_band) k -= per_band;
            if (k < (s >> 1)) {
                GX = I * s + 2 * k; GY = GX + 1;
                diag_item = true;
            } else {
                int t = k - (s >> 1), u = 0;
                while (t >= s - 1 - u) { t -= s - 1 - u; ++u; }      // triangular index -> (u, v), u < v < s
                GX = I * s + u; GY = I * s + u + 1 + t;
                diag_item = false;
            }
        }
    } else {
        // first the diagonal items (group bx with group bx + half_up), then the group pairs (bx, bx + shift) shift by shift; the
        // last shift of an even n_g (antipodal pairs) has only its lower half
        int shift = 0, bx = bid;
        if (bx >= half_up) {
            const int id = bx - half_up;
            shift = 1 + id / n_g;
            bx = id - (shift - 1) * n_g;
        }
        GX = bx;
        GY = bx + (shift == 0 ? half_up : shift);   // a diagonal item's second group may not exist (odd n_g): its waves idle
        if (shift != 0 && GY >= n_g) GY -= n_g;
        diag_item = shift == 0;
    }

    // ---- 1. the state of this launch: integrate the workgroup's 256 pedestrians from the previous launch's partial forces.
    //      Every load of the prologue is issued before the first use of any (they come from other XCDs' writes: one round
    //      trip, not several): own state first, then the column sums -- a thread takes two neighbouring pedestrians (16-B
    //      loads) and 1/PARTS of the slab rows.
    // (everything that decides WHICH rows a wave touches is wave-uniform and kept in SGPRs -- round 4: the twelve upper waves of a
    //  workgroup branch around the own-row loads and the integration instead of running them with an empty exec mask)
    const bool integrate = f.mode != 0;
    const int quarter = wave & (2 * GROUP / WAVE - 1);          // which 64 of the workgroup's 256 pedestrian slots this wave maps to
    const int p = quarter * WAVE + lane;         // pedestrian slot = tid mod 256 (the first 256 threads integrate)
    const bool lower = wave < 2 * GROUP / WAVE;
    const int G = (quarter < 2) ? GX : GY;
    const bool present = G < n_g;
    const int i = G * GROUP + (quarter & 1) * WAVE + lane;      // < N_pad whenever the group exists (N_pad is a multiple of four tiles)
    const bool live = present && i < a.N;
    float4 st = make_float4(0.f, 0.f, 0.f, 0.f), o = st;
    float2 stz = make_float2(0.f, 0.f);          // Z3: {z, vz}
    float2 geo_f = make_float2(0.f, 0.f);        // GEO: border + obstacle forces on the stored state, left by the previous launch
    uint32_t nd0 = 0, pid = 0;
    if (present && lower) {
        st = a.pk_cur[i];
        if (Z3) stz = a.zv_cur[i];
        if (live && integrate) {
            o = f.own_cur[i];
            if (GEO) {
                float2 gv[FUSED_GEO_SLICES_MAX];
#pragma unroll
                for (int k = 0; k < FUSED_GEO_SLICES_MAX; ++k) gv[k] = k < f.geo_slices ? f.geo_prev[(size_t)k * a.N_pad + i] : make_float2(0.f, 0.f);
                geo_f = make_float2(((gv[0].x + gv[1].x) + (gv[2].x + gv[3].x)) + ((gv[4].x + gv[5].x) + (gv[6].x + gv[7].x)),
                                    ((gv[0].y + gv[1].y) + (gv[2].y + gv[3].y)) + ((gv[4].y + gv[5].y) + (gv[6].y + gv[7].y)));
            }
            if ((a.flags & 2u) && diag_item) { nd0 = a.draws[i]; pid = a.ids ? a.ids[i] : (uint32_t)i; }
        }
        if (RAD) {
            const float r_ = a.radius[i];
            sh.rad[p] = r_;
            sh.radt[RAD ? (p >> 6) : 0][p & (WAVE - 1)] = r_; sh.radt[RAD ? (p >> 6) : 0][(p & (WAVE - 1)) + WAVE] = r_;
        }
    }
    if (tid == 0) sh.any = 0;
    {
        const int pp = (wave & 1) * WAVE + lane; // = tid mod 128: pedestrians 2 pp, 2 pp + 1 of the workgroup's 256
        const int part = wave >> 1;
        const int Gq = (wave & 1) ? GY : GX;     // (2 pp < GROUP for the even waves)
        const int i2 = Gq * GROUP + 2 * lane;
        float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 accz = make_float2(0.f, 0.f);
        if (integrate && a.en_ped && Gq < n_g && i2 < a.N) {
            const float4* col = reinterpret_cast<const float4*>(f.slab_prev + i2);
            const float2* colz = Z3 ? reinterpret_cast<const float2*>(f.slabz_prev + i2) : nullptr;
            const size_t stride4 = (size_t)a.N_pad / 2;
            const int per = (n_g + PARTS - 1) / PARTS;
            const int r1 = min(n_g, (part + 1) * per);
            const int r00 = part * per;
            if (!Z3 && r1 - r00 == 4) {
                // c2's shape (32 groups over 8 parts): exactly four rows, no predication and no zero fill -- the same association as
                // the general form below with its absent rows at 0, hence the same bits, in 16 adds instead of 32 + 26 moves (round 4:
                // every wave of every workgroup runs this before its first systolic step)
                const float4 v0 = col[(size_t)r00 * stride4], v1 = col[(size_t)(r00 + 1) * stride4];
                const float4 v2 = col[(size_t)(r00 + 2) * stride4], v3 = col[(size_t)(r00 + 3) * stride4];
                acc4 = make_float4((v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y),
                                   (v0.z + v1.z) + (v2.z + v3.z), (v0.w + v1.w) + (v2.w + v3.w));
            } else if (!Z3) {
                for (int r0 = r00; r0 < r1; r0 += 8) {
                    float4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (r0 + k < r1) ? col[(size_t)(r0 + k) * stride4] : make_float4(0.f, 0.f, 0.f, 0.f);
                    acc4.x += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
                    acc4.y += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
                    acc4.z += ((v[0].z + v[1].z) + (v[2].z + v[3].z)) + ((v[4].z + v[5].z) + (v[6].z + v[7].z));
                    acc4.w += ((v[0].w + v[1].w) + (v[2].w + v[3].w)) + ((v[4].w + v[5].w) + (v[6].w + v[7].w));
                }
            } else {
                // (four rows in flight instead of eight: the z rows ride along and the registers are the same 64 per lane)
                for (int r0 = part * per; r0 < r1; r0 += 4) {
                    float4 v[4];
                    float2 vz[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = (r0 + k < r1) ? col[(size_t)(r0 + k) * stride4] : make_float4(0.f, 0.f, 0.f, 0.f);
                        vz[k] = (r0 + k < r1) ? colz[(size_t)(r0 + k) * stride4] : make_float2(0.f, 0.f);
                    }
                    acc4.x += (v[0].x + v[1].x) + (v[2].x + v[3].x);
                    acc4.y += (v[0].y + v[1].y) + (v[2].y + v[3].y);
                    acc4.z += (v[0].z + v[1].z) + (v[2].z + v[3].z);
                    acc4.w += (v[0].w + v[1].w) + (v[2].w + v[3].w);
                    accz.x += (vz[0].x + vz[1].x) + (vz[2].x + vz[3].x);
                    accz.y += (vz[0].y + vz[1].y) + (vz[2].y + vz[3].y);
                }
            }
        }
        sh.qv()[part][2 * pp] = make_float2(acc4.x, acc4.y);
        sh.qv()[part][2 * pp + 1] = make_float2(acc4.z, acc4.w);
        if (Z3) { sh.qzv()[part][2 * pp] = accz.x; sh.qzv()[part][2 * pp + 1] = accz.y; }
    }
    FUSED_STAMP(1);
    __syncthreads();
    // the arithmetic of sfm_sym_epilogue_kernel without border / obstacle forces (pedestrian_simulation.py:57-83,
    // forces.py:40-52): acceleration towards the waypoint, capped velocity, position, arrival -> next waypoint
    auto put_state = [&](const float4 ns, const float2 nsz) {   // pedestrian slot p of the workgroup in the state the pairs are evaluated on
        sh.st[p] = ns;
        const float4 t = make_float4(ns.x, ns.y, a.ped.lam * ns.z, a.ped.lam * ns.w);
        if (X2) {
            float* pl = reinterpret_cast<float*>(sh.trav) + (p >> 6) * 2 * WAVE + (p & (WAVE - 1));      // plane c at + c * 4 * 2 * WAVE
            pl[0] = t.x; pl[WAVE] = t.x;
            pl[8 * WAVE] = t.y; pl[9 * WAVE] = t.y;
            pl[16 * WAVE] = t.z; pl[17 * WAVE] = t.z;
            pl[24 * WAVE] = t.w; pl[25 * WAVE] = t.w;
        } else {
        sh.trav[p >> 6][p & (WAVE - 1)] = t;
        sh.trav[p >> 6][(p & (WAVE - 1)) + WAVE] = t;
        }
        if (Z3) {
            sh.stz[Z3 ? p : 0] = nsz;
            const float2 tz = make_float2(nsz.x, a.ped.lam * nsz.y);
            if (X2) {
                float* plz = reinterpret_cast<float*>(sh.travz) + (p >> 6) * 2 * WAVE + (p & (WAVE - 1));   // planes z | lambda vz of [4][2 * WAVE] floats
                plz[0] = tz.x; plz[WAVE] = tz.x;
                plz[8 * WAVE] = tz.y; plz[9 * WAVE] = tz.y;
            } else {
            sh.travz[Z3 ? (p >> 6) : 0][p & (WAVE - 1)] = tz;
            sh.travz[Z3 ? (p >> 6) : 0][(p & (WAVE - 1)) + WAVE] = tz;
            }
        }
    };
    auto finish = [&](const float2 g, const float gz) {
        const float fpx = a.en_ped ? a.ped.negA * g.x : 0.f, fpy = a.en_ped ? a.ped.negA * g.y : 0.f;
        const float fpz = (Z3 && a.en_ped) ? a.ped.negA * gz : 0.f;
        const float x = st.x, y = st.y, vx = st.z, vy = st.w, ts = o.z;
        const float z = stz.x, vz = stz.y;
        float wx = o.x, wy = o.y;
        float fax = 0.f, fay = 0.f, faz = 0.f;
        // (round 4: 1 / |.| by v_rsq_f32 -- 1 ulp -- instead of IEEE sqrt + divide, ~35 instructions less in front of every
        //  workgroup's first systolic step; the zero vector still normalises to zero, stateutils.py:85-92, and a pedestrian at rest
        //  with target speed 0 still stays at rest, stateutils.py:20-23)
        if (a.en_acc) {
            const float tx_ = wx - x, ty_ = wy - y;
            const float n2 = fmaf(tx_, tx_, ty_ * ty_);
            const float inv = (n2 > 0.0f) ? rsq(n2) : 1.0f;
            fax = (ts * (tx_ * inv) - vx) * a.inv_tau;
            fay = (ts * (ty_ * inv) - vy) * a.inv_tau;
            if (Z3) faz = (0.0f - vz) * a.inv_tau;
        }
        const float Fx = (fax + fpx) + geo_f.x, Fy = (fay + fpy) + geo_f.y;   // forces.py order: acceleration, pedestrian, border + obstacles
        const float Fz = faz + fpz;
        float nvx = fmaf(a.dt, Fx, vx), nvy = fmaf(a.dt, Fy, vy), nvz = Z3 ? fmaf(a.dt, Fz, vz) : 0.f;
        const float s2 = Z3 ? fmaf(nvx, nvx, fmaf(nvy, nvy, nvz * nvz)) : fmaf(nvx, nvx, nvy * nvy);
        const float fac = (s2 > 0.0f) ? fminf(1.0f, (ts * a.max_speed_factor) * rsq(s2)) : 0.0f;   // (speed 0: the capped velocity is 0 whatever the factor)
        nvx *= fac; nvy *= fac; nvz *= fac;
        float nx = x, ny = y, nz = z;
        if (a.flags & 1u) { nx = fmaf(a.dt, nvx, x); ny = fmaf(a.dt, nvy, y); if (Z3) nz = fmaf(a.dt, nvz, z); }
        const float4 ns = make_float4(nx, ny, nvx, nvy);
        const float2 nsz = make_float2(nz, nvz);
        if (diag_item) {                                             // this workgroup stores the group
            if (a.flags & 2u) {
                const float ax_ = wx - x, ay_ = wy - y;
                if (fmaf(ax_, ax_, ay_ * ay_) < a.arrive_thr2) {
                    const uint32_t nd = nd0 + 1u;
                    wx = waypoint_coord(a.seed, pid, nd, 0u, a.world_side);
                    wy = waypoint_coord(a.seed, pid, nd, 1u, a.world_side);
                    a.draws[i] = nd;
                }
            }
            f.own_next[i] = make_float4(wx, wy, o.z, o.w);
            a.pk_next[i] = ns;
            if (Z3) a.zv_next[i] = nsz;
        }
        put_state(ns, nsz);
    };
    bool bad = false;
    if (lower && present) {
        if (integrate && live) {
            float2 g = sh.qv()[0][p];
            float gz = Z3 ? sh.qzv()[0][p] : 0.f;
#pragma unroll
            for (int k = 1; k < PARTS; ++k) { const float2 q = sh.qv()[k][p]; g.x += q.x; g.y += q.y; if (Z3) gz += sh.qzv()[k][p]; }
            bad = a.en_ped && (!(fabsf(g.x) < __builtin_inff()) || !(fabsf(g.y) < __builtin_inff()) || (Z3 && !(fabsf(gz) < __builtin_inff())));
            if (bad) sh.any = 1; else finish(g, gz);
        } else {
            put_state(st, stz);                                      // ghosts, and the first launch of a run: as stored
        }
    }
    if (lower) sh.badrow[p] = bad ? 1 : 0;
    __syncthreads();
    if (sh.any) {
        // coincident pairs: the rows that caught a NaN are recomputed with the exact ordered body (rare; every workgroup that
        // holds such a row does it, and they all get the same bits)
        const int N = a.N;
        for (int q = wave; q < 2 * GROUP; q += NW) {
            if (!sh.badrow[q]) continue;                                  // uniform
            const int Gq = (q < GROUP) ? GX : GY;
            const int ip = Gq * GROUP + (q & (GROUP - 1));
            const float4 si = a.pk_cur[ip];
            const float xi = uniform(si.x), yi = uniform(si.y), vxi = uniform(si.z), vyi = uniform(si.w);
            float zi = 0.f, vzi = 0.f;
            if (Z3) { const float2 sz = a.zv_cur[ip]; zi = uniform(sz.x); vzi = uniform(sz.y); }
            float gx = 0.f, gy = 0.f, gz = 0.f;
            for (int j0 = 0; j0 < N; j0 += WAVE) {
                const int j = j0 + lane;
                const float4 pj = a.pk_cur[min(j, N - 1)];
                float2 pz = make_float2(0.f, 0.f);
                if (Z3) pz = a.zv_cur[min(j, N - 1)];
                float cx = 0.f, cy = 0.f, cz = 0.f, rinv;
                moussaid<Z3, RAD, true>(a.ped, pj.x - xi, pj.y - yi, pz.x - zi, vxi - pj.z, vyi - pj.w, vzi - pz.y,
                                        RAD ? a.radius[ip] + a.radius[min(j, N - 1)] : 0.f, cx, cy, cz, rinv);
                const bool valid = (j < N) & (j != ip);
                gx += valid ? cx : 0.f;
                gy += valid ? cy : 0.f;
                if (Z3) gz += valid ? cz : 0.f;
            }
            gx = wave_sum(gx);
            gy = wave_sum(gy);
            if (Z3) gz = wave_sum(gz);
            if (lane == 0) { sh.part[q] = make_float2(gx, gy); if (Z3) sh.partz[Z3 ? q : 0] = gz; }
        }
        __syncthreads();
        if (bad) finish(sh.part[p], Z3 ? sh.partz[Z3 ? p : 0] : 0.f);
        __syncthreads();
    }

    if (GEO && geo_role) {
        // ---- 2g. border / obstacle forces on the new state of tile geo_tile (the geometry kernel's body), one float2 per pedestrian
        const int ig = geo_tile * WAVE + lane;
        const float4 sg = sh.st[(geo_tile & 1) * WAVE + lane];
        GeoLane me;
        me.live = ig < a.N;
        me.x = 3.0e15f; me.y = 3.0e15f; me.vx = 0.f; me.vy = 0.f; me.r = 0.f;
        if (me.live) { me.x = sg.x; me.y = sg.y; me.vx = sg.z; me.vy = sg.w; me.r = f.own_cur[ig].w; }
        me.walk = me.live && !(a.crossing && a.crossing[ig]);          // forces.py:140-141,176-177
        __syncthreads();                                               // the geometry body's LDS lies over the prologue's
        GeoShared<NW>& gs = *reinterpret_cast<GeoShared<NW>*>(smem);
        geometry_forces<RAD, NW, true>(a, gs, me, geo_slice, f.geo_slices, tid, nullptr);
        if (wave < 2 && me.live) {                                     // wave 0: x, wave 1: y -- border, static, dynamic in that order
            float vb = 0.f, vs = 0.f, vd = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { vb += gs.acc[w][wave][lane]; vs += gs.acc[w][2 + wave][lane]; vd += gs.acc[w][4 + wave][lane]; }
            reinterpret_cast<float*>(f.geo_next + (size_t)geo_slice * a.N_pad + ig)[wave] = (vb + a.stat.negA * vs) + a.dyn.negA * vd;
        }
        return;
    }

    FUSED_STAMP(2);
    // ---- 2. this workgroup's tile pairs on the new state: every wave SPW systolic steps (sfm_pair_sym_kernel's step)
    int ia, ib, sig0;                             // LDS slots of the travelling / resident tile's lane 0, first rotation
    bool diag = false, work = true;
    if (diag_item) {
        const int sel = wave / (NW / 2);          // lower half of the waves: group GX, upper half: group GY
        const int lw = wave % (NW / 2);
        const int base = sel * GROUP;
        const int Gd = sel ? GY : GX;
        if (lw < 2 * D) {                         // the two diagonal tiles: sigma 1..32, D waves each
            const int tl = lw / D;
            ia = ib = base + tl * WAVE;
            sig0 = 1 + (lw % D) * SPW;
            diag = true;
            work = Gd < n_g && 2 * Gd + tl < n_t;
        } else {                                  // tile 0 travelling past tile 1: sigma 0..63
            ia = base; ib = base + WAVE;
            sig0 = (lw - 2 * D) * SPW;
            work = Gd < n_g && 2 * Gd + 1 < n_t;
        }
    } else {
        const int q = wave / (NW / 4);            // (tile of GX, tile of GY): 00 01 10 11
        ia = (q >> 1) * WAVE;
        ib = GROUP + (q & 1) * WAVE;
        sig0 = (wave % (NW / 4)) * SPW;
        work = 2 * GX + (q >> 1) < n_t && 2 * GY + (q & 1) < n_t;
    }
    float fxi = 0.f, fyi = 0.f, fxj = 0.f, fyj = 0.f, fzi = 0.f, fzj = 0.f;
    int i_end_loc = lane;
    float fxb = 0.f, fyb = 0.f, fzb = 0.f;        // X2: the sums of the second travelling chain, and where they end up
    int i_end_b = lane;
    if (X2 && work) {
        // Two pairs per lane: the wave's SPW steps are SPW / 2 double steps; chain A meets sigma0 .. sigma0 + SPW/2 - 1, chain B the same
        // XB further on -- half a tile pair (32) away, on a diagonal tile half of ITS 32 rotations (16) -- so the waves of a tile pair
        // still cover every rotation once.  A lane's two travelling pedestrians are XB slots apart in the doubled image: one
        // ds_read2_b32 per plane brings both into a register pair.
        const IxConst& c = a.ped;
        const float4 pj = sh.st[ib + lane];
        const float ujx = c.lam * pj.z, ujy = c.lam * pj.w;
        float zj = 0.f, ujz = 0.f;
        if (Z3) { const float2 qz = sh.stz[Z3 ? ib + lane : 0]; zj = qz.x; ujz = c.lam * qz.y; }
        float rj = 0.f;
        if (RAD) rj = sh.rad[ib + lane];
        constexpr int HALF = SPW / 2;
        const int s0 = diag ? 1 + ((sig0 - 1) >> 1) : sig0 >> 1;      // (sig0 was laid out for SPW steps per wave: 1 + j SPW / j SPW)
        const float* pl = reinterpret_cast<const float*>(sh.trav) + (ia >> 6) * 2 * WAVE + lane + s0;
        const float* plz = reinterpret_cast<const float*>(sh.travz) + (Z3 ? (ia >> 6) * 2 * WAVE + lane + s0 : 0);
        const float* plr = &sh.radt[RAD ? (ia >> 6) : 0][RAD ? lane + s0 : 0];
        v2f fjx = bcast(0.f), fjy = bcast(0.f), fjzv = bcast(0.f);
        auto chain = [&](auto xb_tag) __attribute__((always_inline)) {
            constexpr int XB = decltype(xb_tag)::value;
            const bool tail_one_sided = diag && (s0 + XB + HALF - 1 == 32);       // sigma = 32 on a diagonal tile: one-sided (uniform)
            v2f X, Y, U, V;
            X.x = pl[0]; X.y = pl[XB]; Y.x = pl[8 * WAVE]; Y.y = pl[8 * WAVE + XB];
            U.x = pl[16 * WAVE]; U.y = pl[16 * WAVE + XB]; V.x = pl[24 * WAVE]; V.y = pl[24 * WAVE + XB];
            v2f Zp = bcast(0.f), Wp = bcast(0.f), Rp = bcast(0.f);
            if (Z3) { Zp.x = plz[0]; Zp.y = plz[XB]; Wp.x = plz[8 * WAVE]; Wp.y = plz[8 * WAVE + XB]; }
            if (RAD) { Rp.x = plr[0]; Rp.y = plr[XB]; }
#pragma unroll
            for (int s_ = 0; s_ < HALF; ++s_) {
                constexpr bool PRE = !(Z3 && NW == 16);             // (the 16-wave 3-D form has 64 VGPRs: no operands in flight across a double step)
                if (!PRE && s_ > 0) {
                    X.x = pl[s_]; X.y = pl[s_ + XB]; Y.x = pl[8 * WAVE + s_]; Y.y = pl[8 * WAVE + s_ + XB];
                    U.x = pl[16 * WAVE + s_]; U.y = pl[16 * WAVE + s_ + XB]; V.x = pl[24 * WAVE + s_]; V.y = pl[24 * WAVE + s_ + XB];
                    if (Z3) { Zp.x = plz[s_]; Zp.y = plz[s_ + XB]; Wp.x = plz[8 * WAVE + s_]; Wp.y = plz[8 * WAVE + s_ + XB]; }
                    if (RAD) { Rp.x = plr[s_]; Rp.y = plr[s_ + XB]; }
                }
                v2f Xn = X, Yn = Y, Un = U, Vn = V, Zn = Zp, Wn = Wp, Rn = Rp;
                if (PRE && s_ + 1 < HALF) {                                  // the next double step's operands are in flight during this one
                    Xn.x = pl[s_ + 1]; Xn.y = pl[s_ + 1 + XB]; Yn.x = pl[8 * WAVE + s_ + 1]; Yn.y = pl[8 * WAVE + s_ + 1 + XB];
                    Un.x = pl[16 * WAVE + s_ + 1]; Un.y = pl[16 * WAVE + s_ + 1 + XB]; Vn.x = pl[24 * WAVE + s_ + 1]; Vn.y = pl[24 * WAVE + s_ + 1 + XB];
                    if (Z3) { Zn.x = plz[s_ + 1]; Zn.y = plz[s_ + 1 + XB]; Wn.x = plz[8 * WAVE + s_ + 1]; Wn.y = plz[8 * WAVE + s_ + 1 + XB]; }
                    if (RAD) { Rn.x = plr[s_ + 1]; Rn.y = plr[s_ + 1 + XB]; }
                }
                __builtin_amdgcn_sched_barrier(0);
                v2f cx, cy, cz = bcast(0.f);
                if (Z3) moussaid_spatial_x2<RAD>(c, pj.x, pj.y, zj, ujx, ujy, ujz, X, Y, Zp, U, V, Wp, cx, cy, cz, Rp + bcast(rj));
                else moussaid_planar_x2<RAD>(c, pj.x, pj.y, ujx, ujy, X, Y, U, V, cx, cy, Rp + bcast(rj));
                fxi = rot_in(fxi) + cx.x; fyi = rot_in(fyi) + cy.x;
                fxb = rot_in(fxb) + cx.y; fyb = rot_in(fyb) + cy.y;
                if (Z3) { fzi = rot_in(fzi) + cz.x; fzb = rot_in(fzb) + cz.y; }
                if (s_ + 1 == HALF && tail_one_sided) { fjx.x -= cx.x; fjy.x -= cy.x; if (Z3) fjzv.x -= cz.x; }      // (chain B's last pair: the travelling side only)
                else { fjx -= cx; fjy -= cy; if (Z3) fjzv -= cz; }
                X = Xn; Y = Yn; U = Un; V = Vn; Zp = Zn; Wp = Wn; Rp = Rn;
                __builtin_amdgcn_sched_barrier(0);
            }
            i_end_loc = (lane + s0 + HALF - 1) & (WAVE - 1);
            i_end_b = (lane + s0 + XB + HALF - 1) & (WAVE - 1);
        };
        if (diag) chain(std::integral_constant<int, 16>{});
        else chain(std::integral_constant<int, 32>{});
        fxj = fjx.x + fjx.y;
        fyj = fjy.x + fjy.y;
        if (Z3) fzj = fjzv.x + fjzv.y;
    } else
    if (work) {
        const IxConst& c = a.ped;
        const float4 pj = sh.st[ib + lane];
        const float ujx = c.lam * pj.z, ujy = c.lam * pj.w;
        float rj = 0.f;
        if (RAD) rj = sh.rad[ib + lane];
        float zj = 0.f, ujz = 0.f;
        if (Z3) { const float2 qz = sh.stz[Z3 ? ib + lane : 0]; zj = qz.x; ujz = c.lam * qz.y; }
        // step s meets pedestrian (lane + sig0 + s) mod 64 of the travelling tile: slot lane + sig0 + s of the doubled image
        const float4* trav = &sh.trav[ia >> 6][lane + sig0];
        const float* radt = &sh.radt[RAD ? (ia >> 6) : 0][lane + sig0];
        const float2* travz = &sh.travz[Z3 ? (ia >> 6) : 0][lane + sig0];
        // sigma = 32 on a diagonal tile meets every unordered pair {l, l+32} in BOTH lanes: one-sided there
        const bool tail_one_sided = diag && (sig0 + SPW - 1 == 32);  // uniform
        float4 T = trav[0];
        float ri = RAD ? radt[0] : 0.f;
        float2 Tz = make_float2(0.f, 0.f);
        if (Z3) Tz = travz[0];
#if SFM_PK
        v2f fjv; fjv.x = 0.f; fjv.y = 0.f;       // the resident sums as a register pair (v_pk_add_f32)
#endif
#pragma unroll
        for (int s_ = 0; s_ < SPW; ++s_) {
            float4 Tn = T;
            float rin = ri;
            float2 Tzn = Tz;
            if (s_ + 1 < SPW) { Tn = trav[s_ + 1]; if (RAD) rin = radt[s_ + 1]; if (Z3) Tzn = travz[s_ + 1]; }   // the next step's operand is in flight during this one
            __builtin_amdgcn_sched_barrier(0);       // (... so its read is issued here, not where the scheduler would sink it to)
            const float dx = pj.x - T.x, dy = pj.y - T.y;
            float cx, cy, cz = 0.f;
            if (Z3 && SFM_PK) {
                v2f pjv, ujv, Tv, Uv, cv;
                pjv.x = pj.x; pjv.y = pj.y; ujv.x = ujx; ujv.y = ujy; Tv.x = T.x; Tv.y = T.y; Uv.x = T.z; Uv.y = T.w;
                moussaid_spatial_pk<RAD, false>(c, pjv, zj, ujv, ujz, Tv, Tz.x, Uv, Tz.y, RAD ? ri + rj : 0.f, 0.f, cv, cz);
                cx = cv.x; cy = cv.y;
            } else if (Z3) {
                const float dz = zj - Tz.x;
                moussaid_spatial<RAD, false>(c, dx, dy, dz, fmaf(dx, dx, fmaf(dy, dy, dz * dz)), T.z - ujx, T.w - ujy, Tz.y - ujz, RAD ? ri + rj : 0.f, cx, cy, cz);
            } else {
#if SFM_PK
                v2f pjv, ujv, Tv, Uv;
                pjv.x = pj.x; pjv.y = pj.y; ujv.x = ujx; ujv.y = ujy; Tv.x = T.x; Tv.y = T.y; Uv.x = T.z; Uv.y = T.w;
                v2f cv;
                moussaid_planar_pk<RAD, false>(c, pjv, ujv, Tv, Uv, RAD ? ri + rj : 0.f, 0.f, cv);
                cx = cv.x; cy = cv.y;
                if (s_ + 1 < SPW || !tail_one_sided) fjv -= cv;
#else
                moussaid_planar<RAD, false>(c, dx, dy, fmaf(dx, dx, dy * dy), T.z - ujx, T.w - ujy, RAD ? ri + rj : 0.f, cx, cy);
#endif
            }
            // the sums of the pedestrian this lane has just met were in lane + 1 a step ago: rotation and add in one instruction
            fxi = rot_in(fxi) + cx;
            fyi = rot_in(fyi) + cy;
            if (Z3) fzi = rot_in(fzi) + cz;
            if (SFM_PK && !Z3) { fxj = fjv.x; fyj = fjv.y; }
            else if (s_ + 1 < SPW || !tail_one_sided) { fxj -= cx; fyj -= cy; if (Z3) fzj -= cz; }
            T = Tn;
            ri = rin;
            Tz = Tzn;
            // steps are not interleaved (measured again in round 3, variant builds: without this barrier 16.44, without either 16.60,
            // pairs of steps free to interleave 16.51, as here 16.15-16.30 us per c2 tick)
            __builtin_amdgcn_sched_barrier(0);
        }
        i_end_loc = (lane + sig0 + SPW - 1) & (WAVE - 1);
    }
    FUSED_WAVE_STAMP(0);
    sh.fi[wave][i_end_loc] = make_float2(fxi, fyi);
    if (X2) sh.fi2[X2 ? wave : 0][i_end_b] = make_float2(fxb, fyb);
    if (X2 && Z3) sh.fi2z[(X2 && Z3) ? wave : 0][i_end_b] = fzb;
    sh.fj[wave][lane] = make_float2(fxj, fyj);
    if (Z3) { sh.fiz[Z3 ? wave : 0][i_end_loc] = fzi; sh.fjz[Z3 ? wave : 0][lane] = fzj; }
    FUSED_STAMP(3);
    __syncthreads();
    FUSED_WAVE_STAMP(1);
    if (!lower || !present) return;
    const int tl = (p >> 6) & 1, l = lane;       // tile of the group, pedestrian of the tile
    // sums of wave w for pedestrian l of its travelling / resident tile, z in the third component
    auto fi = [&](int w) {
        float2 v = sh.fi[w][l];
        if (X2) { const float2 v2 = sh.fi2[X2 ? w : 0][l]; v.x += v2.x; v.y += v2.y; }       // (both travelling chains of the wave)
        float vz_ = Z3 ? sh.fiz[Z3 ? w : 0][l] : 0.f;
        if (X2 && Z3) vz_ += sh.fi2z[(X2 && Z3) ? w : 0][l];
        return make_float3(v.x, v.y, vz_);
    };
    auto fj = [&](int w) { const float2 v = sh.fj[w][l]; return make_float3(v.x, v.y, Z3 ? sh.fjz[Z3 ? w : 0][l] : 0.f); };
    float3 r = make_float3(0.f, 0.f, 0.f);
    int row;                                     // partner group = slab row
    if (diag_item) {
        const int w0 = (p < GROUP) ? 0 : NW / 2;
#pragma unroll
        for (int k = 0; k < D; ++k) {            // the tile's own diagonal item: both sides are this tile
            const float3 u = fi(w0 + tl * D + k), v = fj(w0 + tl * D + k);
            r.x += u.x + v.x; r.y += u.y + v.y; r.z += u.z + v.z;
        }
#pragma unroll
        for (int k = 0; k < 2 * D; ++k) {        // the other tile of the group
            const float3 u = tl ? fj(w0 + 2 * D + k) : fi(w0 + 2 * D + k);
            r.x += u.x; r.y += u.y; r.z += u.z;
        }
        row = G;
    } else if (p < GROUP) {                      // force on GX's pedestrians from GY: the travelling sides
#pragma unroll
        for (int k = 0; k < NW / 2; ++k) { const float3 u = fi(tl * (NW / 2) + k); r.x += u.x; r.y += u.y; r.z += u.z; }
        row = GY;
    } else {                                     // force on GY's pedestrians from GX: the resident sides
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < NW / 4; ++k) { const float3 u = fj((2 * h + tl) * (NW / 4) + k); r.x += u.x; r.y += u.y; r.z += u.z; }
        row = GX;
    }
    f.slab_next[(size_t)row * a.N_pad + i] = make_float2(r.x, r.y);
    if (Z3) f.slabz_next[(size_t)row * a.N_pad + i] = r.z;
    FUSED_STAMP(4);
}

// Dynamic obstacles on the device (obstacles.py:297-329 without the simulator): one wave per vehicle moves the centre
// by dt*v (advance != 0) and regenerates its ring p = c + R(yaw) u, the lanes over the ring points.
__global__ __launch_bounds__(BLOCK) void sfm_dynamic_boxes_kernel(float4* __restrict__ ctr, const int* __restrict__ off,
                                                                  const float2* __restrict__ local, const float2* __restrict__ rot,
                                                                  float2* __restrict__ pts, int M, float dt, int advance) {
    const int k = blockIdx.x * WAVES_PER_BLOCK + (int)(threadIdx.x >> 6);
    if (k >= M) return;
    advance_vehicle(DynAdvance{ctr, off, local, rot, pts, M, dt, 0, nullptr, nullptr}, k, threadIdx.x & (WAVE - 1), advance != 0);
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
template <int IPW, bool Z3, bool RAD, int TEAM>
static hipError_t launch_one(const TickArgs& a, hipStream_t st) {
    const int n_local = a.i_end - a.i_begin;
    if (n_local <= 0) return hipSuccess;
    const int per_block = IPW * (WAVES_PER_BLOCK / TEAM);
    const int grid = (n_local + per_block - 1) / per_block;
    TickArgs b = a;
    b.adv.block0 = grid;
    const int extra = a.adv.M > 0 ? (a.adv.M + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK : 0;
    hipLaunchKernelGGL((sfm_tick_kernel<IPW, Z3, RAD, TEAM>), dim3(grid + extra), dim3(BLOCK), 0, st, b);
    return hipGetLastError();
}

template <bool Z3, bool RAD, int TEAM>
static hipError_t launch_ipw(int ipw, const TickArgs& a, hipStream_t st) {
    switch (ipw) {
        case 1: return launch_one<1, Z3, RAD, TEAM>(a, st);
        case 2: return launch_one<2, Z3, RAD, TEAM>(a, st);
        case 4: return launch_one<4, Z3, RAD, TEAM>(a, st);
        default: return launch_one<8, Z3, RAD, TEAM>(a, st);
    }
}

template <int TEAM>
static hipError_t launch_team(int ipw, bool z3, bool rad, const TickArgs& a, hipStream_t st) {
    if (z3) return rad ? launch_ipw<true, true, TEAM>(ipw, a, st) : launch_ipw<true, false, TEAM>(ipw, a, st);
    return rad ? launch_ipw<false, true, TEAM>(ipw, a, st) : launch_ipw<false, false, TEAM>(ipw, a, st);
}

hipError_t launch_tick(int ipw, int team, bool z3, bool rad, const TickArgs& a, hipStream_t st) {
    return team == 4 ? launch_team<4>(ipw, z3, rad, a, st) : launch_team<1>(ipw, z3, rad, a, st);
}

hipError_t launch_modes(const TickArgs& a, hipStream_t st) {
    const int n_local = a.i_end - a.i_begin;
    if (n_local <= 0) return hipSuccess;
    hipLaunchKernelGGL(sfm_mode_kernel, dim3((n_local + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_geometry(bool rad, const TickArgs& a, hipStream_t st) {
    const int n_local = a.i_end - a.i_begin;
    if (n_local <= 0) return hipSuccess;
    const int grid = ((a.i_end + WAVE - 1) >> 6) - (a.i_begin >> 6);      // tiles overlapping the shard
    static const int gw_ov = exp_env("SFM_GEO_WAVES") ? atoi(exp_env("SFM_GEO_WAVES")) : 0;        // A/B only
    const bool thin = gw_ov ? gw_ov == 8 : grid >= 1024;
    TickArgs b = a;
    int extra = 0;                                  // workgroups that build the flat tile-pair list (TickArgs::list_work)
    if (a.list_work) {
        const int threads = thin ? 8 * WAVE : GEO_BLOCK;
        extra = (a.list_n_t * (a.list_n_t / 2 + 1) + threads - 1) / threads;
        b.list_block0 = grid;
    }
    if (thin) {
        if (rad) hipLaunchKernelGGL((sfm_geometry_kernel<true, 8>), dim3(grid + extra, a.geo_slices), dim3(8 * WAVE), 0, st, b);
        else hipLaunchKernelGGL((sfm_geometry_kernel<false, 8>), dim3(grid + extra, a.geo_slices), dim3(8 * WAVE), 0, st, b);
    } else {
        if (rad) hipLaunchKernelGGL((sfm_geometry_kernel<true, GEO_WAVES>), dim3(grid + extra, a.geo_slices), dim3(GEO_BLOCK), 0, st, b);
        else hipLaunchKernelGGL((sfm_geometry_kernel<false, GEO_WAVES>), dim3(grid + extra, a.geo_slices), dim3(GEO_BLOCK), 0, st, b);
    }
    return hipGetLastError();
}

// cutoff on: compact the tile pairs that have to be evaluated (the work list of the pair kernel)
hipError_t launch_sym_list(const TickArgs& a, const SymArgs& sa, hipStream_t st, int partners = PARTNERS_ALL, bool count_is_zero = false) {
    if (a.N <= 1 || !a.en_ped || !sa.work) return hipSuccess;
    hipError_t e = count_is_zero ? hipSuccess : hipMemsetAsync(const_cast<int*>(sa.work_count), 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    const bool whole = sa.t_lo == 0 && sa.t_hi == sa.n_t;      // a shard also looks at partners behind it
    // own partners only (PARTNERS_OWN): the strips' boxes need every tile's box, which a shard does not have yet -- flat kernel
    if (sa.n_strips > 0 && partners != PARTNERS_OWN)
        hipLaunchKernelGGL(sfm_pair_list2_kernel, dim3((sa.t_hi - sa.t_lo + LIST2_WAVES - 1) / LIST2_WAVES), dim3(LIST2_WAVES * WAVE), 0, st,
                           a.tile_box, a.tile_vmax, sa, a.ped.lam, const_cast<uint32_t*>(sa.work), const_cast<int*>(sa.work_count),
                           partners);
    else
        hipLaunchKernelGGL(sfm_pair_list_kernel, dim3((sa.t_hi - sa.t_lo + 255) / 256, whole ? sa.n_t / 2 + 1 : sa.n_t), dim3(256), 0, st,
                           a.tile_box, a.tile_vmax, sa.n_t, sa.t_lo, sa.t_hi, a.ped.lam, a.cut_scale, a.cut_pad,
                           const_cast<uint32_t*>(sa.work), const_cast<int*>(sa.work_count), partners, sa.idx, sa.row_base, sa.cap_pairs);
    return hipGetLastError();
}

template <bool RAD, bool CUT>
static void launch_sym_pair_t(dim3 grid, const TickArgs& a, const SymArgs& sa, hipStream_t st) {
    static const int pad_lds = exp_env("SFM_PAIR_LDS") ? atoi(exp_env("SFM_PAIR_LDS")) : 0;   // experiment: limits the resident workgroups per CU
    if (sa.slabz) hipLaunchKernelGGL((sfm_pair_sym_kernel<RAD, CUT, true>), grid, dim3(BLOCK), (size_t)pad_lds, st, a.pk_cur, a.zv_cur, a.radius, a.ped, sa);
    else hipLaunchKernelGGL((sfm_pair_sym_kernel<RAD, CUT, false>), grid, dim3(BLOCK), (size_t)pad_lds, st, a.pk_cur, a.zv_cur, a.radius, a.ped, sa);
}

template <bool RAD, bool CUT>
static void launch_sym_pair_geo_t(dim3 grid, const TickArgs& a, const SymArgs& sa, int tiles, int stride, hipStream_t st) {
    if (sa.slabz) hipLaunchKernelGGL((sfm_pair_geo_kernel<RAD, CUT, true>), grid, dim3(BLOCK), 0, st, a, sa, tiles, stride);
    else hipLaunchKernelGGL((sfm_pair_geo_kernel<RAD, CUT, false>), grid, dim3(BLOCK), 0, st, a, sa, tiles, stride);
}

// List mode: the grid is a few times what fits at once, each workgroup a contiguous run of the list.  The list holds the items of the
// OWN tiles (a shard's list is an eighth of the whole crowd's: with the whole crowd's 16 rounds its workgroups got 0.8 items each and
// a third of the launch was workgroups that found nothing -- c5, one rank of 8: 151.7 us per tick at 16 rounds, 148.2 at 8,
// tools/shard_rank_time.py), so the rounds follow the own tile count: 2 up to 128 tiles ... 16 from 1024 on.
static int list_rounds(const SymArgs& sa) { return std::min(16, std::max(2, (sa.t_hi - sa.t_lo) / 64)); }

hipError_t launch_sym_pair(bool rad, const TickArgs& a, const SymArgs& sa, hipStream_t st) {
    if (a.N <= 1 || !a.en_ped) return hipSuccess;
    dim3 grid(sa.n_t, sa.n_t / 2 + 1);
    if (sa.work) {
        // a resident grid takes contiguous runs of the list: a few times more workgroups than fit at once (8 per CU), short runs
        // interleave better with the geometry kernel's workgroups and even out the tail; measured best 4x at 256 tiles, 16x from
        // 1024 tiles on
        static const int rounds_ov = exp_env("SFM_ROUNDS") ? atoi(exp_env("SFM_ROUNDS")) : 0;      // A/B only
        const int rounds = rounds_ov > 0 ? rounds_ov : list_rounds(sa);
        grid = dim3(256 * 8 * rounds);
    }
    const bool cut = sa.vmax != nullptr;           // list cutoff: the per-step reach and exponent tests are on as well
    if (rad) { if (cut) launch_sym_pair_t<true, true>(grid, a, sa, st); else launch_sym_pair_t<true, false>(grid, a, sa, st); }
    else { if (cut) launch_sym_pair_t<false, true>(grid, a, sa, st); else launch_sym_pair_t<false, false>(grid, a, sa, st); }
    return hipGetLastError();
}

// the pair kernel of a list-cutoff tick with the geometry workgroups of the same tick in front (sfm_pair_geo_kernel)
hipError_t launch_sym_pair_geo(bool rad, const TickArgs& a, const SymArgs& sa, hipStream_t st) {
    const int tiles = ((a.i_end + WAVE - 1) >> 6) - (a.i_begin >> 6);
    static const int rounds_ov = exp_env("SFM_ROUNDS") ? atoi(exp_env("SFM_ROUNDS")) : 0;      // A/B only
    const int rounds = rounds_ov > 0 ? rounds_ov : list_rounds(sa);
    const int n_geo = tiles * a.geo_slices, n_pair = sa.work ? 256 * 8 * rounds : sa.n_t * (sa.n_t / 2 + 1);
    const dim3 grid(n_geo + n_pair);
    // more geometry workgroups than the CUs hold in one round beside the pair workgroups: spread them evenly over the grid
    static const int stride_ov = exp_env("SFM_PG_STRIDE") ? atoi(exp_env("SFM_PG_STRIDE")) : 0;      // A/B only
    int stride = n_geo > 256 * 4 ? std::max(1, (n_geo + n_pair) / n_geo) : 1;
    if (stride_ov > 0) stride = std::min(stride_ov, std::max(1, (n_geo + n_pair) / n_geo));
    if (stride > 1 && !(stride & 1)) --stride;      // odd: workgroup w runs on CU w mod 256, an even stride would put every geometry workgroup on a few CUs
    const bool cut = sa.vmax != nullptr;           // as launch_sym_pair: list cutoff -> the per-step tests are on
    if (rad) { if (cut) launch_sym_pair_geo_t<true, true>(grid, a, sa, tiles, stride, st); else launch_sym_pair_geo_t<true, false>(grid, a, sa, tiles, stride, st); }
    else { if (cut) launch_sym_pair_geo_t<false, true>(grid, a, sa, tiles, stride, st); else launch_sym_pair_geo_t<false, false>(grid, a, sa, tiles, stride, st); }
    return hipGetLastError();
}

hipError_t launch_tile_bounds(const float4* pk, const float2* zv, int N, float4* box, float* vmax, hipStream_t st, int t_lo = 0,
                              int t_hi = -1) {
    if (N <= 0) return hipSuccess;
    if (t_hi < 0) t_hi = (N + WAVE - 1) / WAVE;                 // default: every tile
    if (t_hi <= t_lo) return hipSuccess;
    hipLaunchKernelGGL(sfm_tile_bounds_kernel, dim3(t_hi - t_lo), dim3(WAVE), 0, st, pk, zv, N, box, vmax, t_lo);
    return hipGetLastError();
}

hipError_t launch_tile_strip_bounds(const float4* pk, int N, int n_t, int tps, int n_strips, float4* box, float* vmax, float4* sbox,
                                    float* svmax, hipStream_t st) {
    if (N <= 0 || n_strips <= 0) return hipSuccess;
    hipLaunchKernelGGL(sfm_tile_strip_bounds_kernel, dim3(n_strips), dim3(TSB_WAVES * WAVE), 0, st, pk, N, n_t, tps, box, vmax, sbox, svmax);
    return hipGetLastError();
}

hipError_t launch_strip_bounds(const float4* box, const float* vmax, int n_t, int tps, int n_strips, float4* sbox, float* svmax,
                               hipStream_t st) {
    if (n_strips <= 0) return hipSuccess;
    hipLaunchKernelGGL(sfm_strip_bounds_kernel, dim3(n_strips), dim3(WAVE), 0, st, box, vmax, n_t, tps, sbox, svmax);
    return hipGetLastError();
}

template <bool RAD, int EW>
static void launch_sym_epilogue_t(const TickArgs& a, const SymArgs& sa, hipStream_t st) {
    TickArgs b = a;
    SymArgs sb = sa;
    b.adv.block0 = sa.t_hi - sa.t_lo;
    const int extra = a.adv.M > 0 ? (a.adv.M + EW - 1) / EW : 0;
    const dim3 grid(sa.t_hi - sa.t_lo + extra);
    if (sa.slabz) hipLaunchKernelGGL((sfm_sym_epilogue_kernel<RAD, EW, true>), grid, dim3(EW * WAVE), 0, st, b, sb);
    else hipLaunchKernelGGL((sfm_sym_epilogue_kernel<RAD, EW, false>), grid, dim3(EW * WAVE), 0, st, b, sb);
}

// 16 waves per tile for small and mid-sized crowds (one dependent round of slab-row loads); from 1024 tiles on there are
// plenty of workgroups and 4 waves per tile measured 2 % better on the c5 tick.
hipError_t launch_sym_epilogue(bool rad, const TickArgs& a, const SymArgs& sa, hipStream_t st) {
    if (a.N <= 0) return hipSuccess;
    static const int ew_ov = exp_env("SFM_EPI_WAVES") ? atoi(exp_env("SFM_EPI_WAVES")) : 0;      // A/B only: 4 / 16
    const bool thin = ew_ov ? ew_ov == 4 : sa.n_t >= 1024;
    if (rad) { if (thin) launch_sym_epilogue_t<true, 4>(a, sa, st); else launch_sym_epilogue_t<true, EPI_WAVES>(a, sa, st); }
    else { if (thin) launch_sym_epilogue_t<false, 4>(a, sa, st); else launch_sym_epilogue_t<false, EPI_WAVES>(a, sa, st); }
    return hipGetLastError();
}

// one launch of the fused tick (FusedArgs::mode)
// number of pair workgroups of the fused tick: diagonal items, full shifts, the half shift of an even n_g
int fused_pair_workgroups(int n_g) { return (n_g + 1) / 2 + n_g * ((n_g - 1) / 2) + ((n_g & 1) ? 0 : n_g / 2); }

template <bool RAD, int NW>
static void launch_fused_t(dim3 grid, bool geo, bool z3, const TickArgs& a, const FusedArgs& f, hipStream_t st) {
    if (z3) {
        if (geo) hipLaunchKernelGGL((sfm_fused_tick_kernel<RAD, NW, true, true>), grid, dim3(NW * WAVE), 0, st, a, f);
        else hipLaunchKernelGGL((sfm_fused_tick_kernel<RAD, NW, true, false>), grid, dim3(NW * WAVE), 0, st, a, f);
    } else {
        if (geo) hipLaunchKernelGGL((sfm_fused_tick_kernel<RAD, NW, false, true>), grid, dim3(NW * WAVE), 0, st, a, f);
        else hipLaunchKernelGGL((sfm_fused_tick_kernel<RAD, NW, false, false>), grid, dim3(NW * WAVE), 0, st, a, f);
    }
}

hipError_t launch_fused_tick(bool rad, const TickArgs& a, const FusedArgs& f, hipStream_t st, int nw) {
    if (a.N <= 1 || f.n_g < 1) return hipErrorInvalidValue;
    const bool geo = f.n_geo_wg > 0, z3 = f.slabz_next != nullptr;
    const int n_adv = (geo && a.adv.M > 0) ? (a.adv.M + nw - 1) / nw : 0;
    const dim3 grid(f.n_geo_wg + f.n_pair_wg + n_adv);
    if (nw == 8) { if (rad) launch_fused_t<true, 8>(grid, geo, z3, a, f, st); else launch_fused_t<false, 8>(grid, geo, z3, a, f, st); }
    else { if (rad) launch_fused_t<true, 16>(grid, geo, z3, a, f, st); else launch_fused_t<false, 16>(grid, geo, z3, a, f, st); }
    return hipGetLastError();
}

// +1 if wave_rol:1 makes lane l receive lane l+1's value, -1 if lane l-1's, 0 on failure
int probe_dpp_direction(hipStream_t st) {
    int* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), 64 * sizeof(int)) != hipSuccess) return 0;
    hipLaunchKernelGGL(sfm_dpp_probe_kernel, dim3(1), dim3(64), 0, st, d);
    int h[64];
    hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return 0;
    bool up = true, down = true;
    for (int l = 0; l < 64; ++l) { up &= (h[l] == ((l + 1) & 63)); down &= (h[l] == ((l + 63) & 63)); }
    return up ? 1 : (down ? -1 : 0);
}

hipError_t launch_dynamic_boxes(float4* ctr, const int* off, const float2* local, const float2* rot, float2* pts, int M,
                                float dt, int advance, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(sfm_dynamic_boxes_kernel, dim3((M + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(BLOCK), 0, st, ctr, off, local,
                       rot, pts, M, dt, advance);
    return hipGetLastError();
}

hipError_t launch_arrived(const float4* pk, const float4* own, int N, float thr2, uint8_t* mask, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(sfm_arrived_kernel, dim3((N + 255) / 256), dim3(256), 0, st, pk, own, N, thr2, mask);
    return hipGetLastError();
}

}  // namespace sfm
