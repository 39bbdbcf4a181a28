// Measures per-SIMD issue rates of the VALU ops the pair body is made of (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(float* out, int iters, float seed, unsigned long long* stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a3}, p5 = {a0, a2}, p6 = {a5, a7}, p7 = {a4, a6};
    const float c = 1.0001f, d = 0.5f;
    const float2v c2 = {c, c}, d2 = {d, d};
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {  // v_fma_f32 x8
            a0 = fmaf(a0, c, d); a1 = fmaf(a1, c, d); a2 = fmaf(a2, c, d); a3 = fmaf(a3, c, d);
            a4 = fmaf(a4, c, d); a5 = fmaf(a5, c, d); a6 = fmaf(a6, c, d); a7 = fmaf(a7, c, d);
        } else if (OP == 1) {  // v_pk_fma_f32 x8
            p0 = __builtin_elementwise_fma(p0, c2, d2); p1 = __builtin_elementwise_fma(p1, c2, d2);
            p2 = __builtin_elementwise_fma(p2, c2, d2); p3 = __builtin_elementwise_fma(p3, c2, d2);
            p4 = __builtin_elementwise_fma(p4, c2, d2); p5 = __builtin_elementwise_fma(p5, c2, d2);
            p6 = __builtin_elementwise_fma(p6, c2, d2); p7 = __builtin_elementwise_fma(p7, c2, d2);
        } else if (OP == 2) {  // v_exp_f32 x8
            a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
        } else if (OP == 3) {  // v_rsq_f32 x8
            a0 = __builtin_amdgcn_rsqf(a0); a1 = __builtin_amdgcn_rsqf(a1); a2 = __builtin_amdgcn_rsqf(a2); a3 = __builtin_amdgcn_rsqf(a3);
            a4 = __builtin_amdgcn_rsqf(a4); a5 = __builtin_amdgcn_rsqf(a5); a6 = __builtin_amdgcn_rsqf(a6); a7 = __builtin_amdgcn_rsqf(a7);
        } else if (OP == 4) {  // v_pk_mul_f32 x8
            p0 = p0 * c2; p1 = p1 * c2; p2 = p2 * c2; p3 = p3 * c2; p4 = p4 * c2; p5 = p5 * c2; p6 = p6 * c2; p7 = p7 * c2;
        } else if (OP == 5) {  // v_cmp + v_cndmask (select) x8
            a0 = a0 > d ? a1 : a0; a1 = a1 > d ? a2 : a1; a2 = a2 > d ? a3 : a2; a3 = a3 > d ? a4 : a3;
            a4 = a4 > d ? a5 : a4; a5 = a5 > d ? a6 : a5; a6 = a6 > d ? a7 : a6; a7 = a7 > d ? a0 : a7;
        } else if (OP == 7 || OP == 8 || OP == 9) {  // DPP moves: wave_rol:1 / row_ror:1 / quad_perm
            constexpr int ctrl = OP == 7 ? 0x134 : (OP == 8 ? 0x121 : 0xB1);
#define ROT(v) v = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xf, 0xf, false))
            ROT(a0); ROT(a1); ROT(a2); ROT(a3); ROT(a4); ROT(a5); ROT(a6); ROT(a7);
#undef ROT
        } else if (OP == 10) {  // ds_bpermute (LDS crossbar) x8
            const int idx = ((threadIdx.x + 1) & 63) << 2;
            a0 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(a0))); a1 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(a1)));
            a2 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(a2))); a3 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(a3)));
            a4 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(a4))); a5 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(a5)));
            a6 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(a6))); a7 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(a7)));
        } else if (OP == 11) {  // v_fmaak (literal constant) x8
            a0 = fmaf(a0, a1, 0.12345f); a1 = fmaf(a1, a2, 0.22345f); a2 = fmaf(a2, a3, 0.32345f); a3 = fmaf(a3, a4, 0.42345f);
            a4 = fmaf(a4, a5, 0.52345f); a5 = fmaf(a5, a6, 0.62345f); a6 = fmaf(a6, a7, 0.72345f); a7 = fmaf(a7, a0, 0.82345f);
        } else if (OP >= 30 && OP <= 47) {  // operand-form sweep of v_fma / v_mul / v_add (which forms run at full rate?)
            const float sc = seed * 3.0f;   // lands in an SGPR
#define X8(stmt) { float& a = a0; stmt } { float& a = a1; stmt } { float& a = a2; stmt } { float& a = a3; stmt } \
                 { float& a = a4; stmt } { float& a = a5; stmt } { float& a = a6; stmt } { float& a = a7; stmt }
            if (OP == 30) { X8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));) }
            if (OP == 31) { X8(asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a) : "s"(sc), "v"(d));) }
            if (OP == 32) { X8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "s"(sc), "v"(d));) }
            if (OP == 33) { X8(asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(a) : "v"(c));) }
            if (OP == 34) { X8(asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(a) : "s"(sc));) }
            if (OP == 35) { X8(asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a) : "s"(sc));) }
            if (OP == 36) { X8(asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a) : "v"(c), "v"(d));) }
            if (OP == 37) { X8(asm volatile("v_add_f32_e64 %0, |%0|, 1.0" : "+v"(a));) }
            if (OP == 38) { X8(asm volatile("v_mul_f32_e64 %0, %0, -%1" : "+v"(a) : "v"(c));) }
            if (OP == 39) { X8(asm volatile("v_fmamk_f32 %0, %0, 0x3dfcd35b, %1" : "+v"(a) : "v"(c));) }
            if (OP == 40) { X8(asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a) : "s"(sc));) }
            if (OP == 41) { X8(asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(a));) }
            if (OP == 42) { X8(asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(c));) }
            if (OP == 43) { X8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "s"(sc));) }
            if (OP == 44) { X8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c) : );) }
            if (OP == 46) { X8(asm volatile("v_add_f32_dpp %0, %0, %1 wave_rol:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(d));) }
            if (OP == 47) { X8(asm volatile("v_add_f32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(d));) }
            if (OP == 45) { X8(asm volatile("v_mul_f32_e32 %0, 0.5, %0" : "+v"(a));) }
#undef X8
        } else if (OP >= 48 && OP <= 55) {  // bitwise forms a copysign can be made of (round 4: is any of them full rate?)
            const float sc = seed * 3.0f;
#define X8(stmt) { float& a = a0; stmt } { float& a = a1; stmt } { float& a = a2; stmt } { float& a = a3; stmt } \
                 { float& a = a4; stmt } { float& a = a5; stmt } { float& a = a6; stmt } { float& a = a7; stmt }
            if (OP == 48) { X8(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "s"(0x80000000), "v"(c));) }
            if (OP == 49) { X8(asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a) : "s"(0xbfffffff));) }
            if (OP == 50) { X8(asm volatile("v_or_b32_e32 %0, %1, %0" : "+v"(a) : "v"(c));) }
            if (OP == 51) { X8(asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a) : "v"(c));) }
            if (OP == 52) { X8(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(d), "v"(c));) }
            if (OP == 53) { X8(asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(a) : "v"(c));) }
            if (OP == 54) { X8(asm volatile("v_lshl_or_b32 %0, %0, 0, %1" : "+v"(a) : "v"(c));) }
            if (OP == 55) { X8(asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a) : "v"(c));) }
            (void)sc;
#undef X8
        } else if (OP >= 12 && OP <= 21) {  // single instructions, forced encodings, 8 independent accumulators
            const float sc = seed * 3.0f;   // lands in an SGPR
#define X8(stmt) { float& a = a0; stmt } { float& a = a1; stmt } { float& a = a2; stmt } { float& a = a3; stmt } \
                 { float& a = a4; stmt } { float& a = a5; stmt } { float& a = a6; stmt } { float& a = a7; stmt }
            if (OP == 12) { X8(asm volatile("v_fmaak_f32 %0, %0, %1, 0x3dfcd35b" : "+v"(a) : "v"(c));) }
            if (OP == 13) { X8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "s"(sc));) }
            if (OP == 14) { X8(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));) }
            if (OP == 15) { X8(asm volatile("v_mul_f32 %0, 0x3f800347, %0" : "+v"(a));) }
            if (OP == 16) { X8(asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "v"(c));) }
            if (OP == 17) { X8(asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "s"(0x7fffffff), "v"(c));) }
            if (OP == 18) { X8(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(c));) }
            if (OP == 19) { X8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c));) }
            if (OP == 20) { X8(asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a), "v"(c) : "vcc");) }
            if (OP == 21) { X8(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(c));) }
#undef X8
        } else if (OP == 6) {  // mixed: 4 fma + 4 exp interleaved
            a0 = fmaf(a0, c, d); a1 = __builtin_amdgcn_exp2f(a1); a2 = fmaf(a2, c, d); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = fmaf(a4, c, d); a5 = __builtin_amdgcn_exp2f(a5); a6 = fmaf(a6, c, d); a7 = __builtin_amdgcn_exp2f(a7);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}
template <int OP>
int run(const char* name, int waves_per_simd, float lanes_per_inst) {
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves -> one wave per SIMD per block per CU
    float* out; CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
    const int iters = 100000;
    unsigned long long* stamps; CHECK(hipMalloc(&stamps, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, 0.001f, stamps);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.001f, stamps);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hs[2]; CHECK(hipMemcpy(hs, stamps, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)hs[0] / (double)hs[1] * 0.1;   // s_memtime ticks per 100 MHz realtime tick
    const double insts_per_simd = (double)iters * 8 * waves_per_simd;   // wave-instructions issued on one SIMD
    const double cyc = ms * 1e-3 * ghz * 1e9;
    printf("%-14s waves/SIMD=%d  %.3f ms  clock %.2f GHz  %.2f cycles/wave-inst  -> %.1f lane-ops/clk/SIMD\n", name, waves_per_simd, ms, ghz,
           cyc / insts_per_simd, 64.0 * lanes_per_inst * insts_per_simd / cyc);
    CHECK(hipFree(out));
    return 0;
}
int main() {
    for (int w : {4, 8}) {
        run<0>("v_fma_f32", w, 1); run<1>("v_pk_fma_f32", w, 2); run<4>("v_pk_mul_f32", w, 2); run<2>("v_exp_f32", w, 1);
        run<3>("v_rsq_f32", w, 1); run<5>("cmp+cndmask", w, 1); run<6>("fma+exp mix", w, 1);
        run<7>("dpp wave_rol:1", w, 1); run<8>("dpp row_ror:1", w, 1); run<9>("dpp quad_perm", w, 1); run<10>("ds_bpermute", w, 1); run<11>("v_fmaak lit", w, 1);
        run<12>("v_fmaak asm", w, 1); run<13>("v_fma sgpr", w, 1); run<14>("v_fmac", w, 1); run<15>("v_mul literal", w, 1);
        run<16>("v_mul", w, 1); run<17>("v_bfi", w, 1); run<18>("v_max", w, 1); run<19>("v_cndmask", w, 1); run<20>("v_cmp", w, 1);
        run<21>("v_sub", w, 1);
        run<30>("fma v,v,v", w, 1); run<31>("fma s,v,v", w, 1); run<32>("fma v,s,v", w, 1); run<33>("fma v,v,0.5", w, 1);
        run<34>("fma v,s,0.5", w, 1); run<35>("mul_e32 s,v", w, 1); run<36>("fma -v,v,v", w, 1); run<37>("add |v|,1.0", w, 1);
        run<38>("mul_e64 v,-v", w, 1); run<39>("v_fmamk", w, 1); run<40>("sub_e32 s,v", w, 1); run<41>("add_e32 1.0,v", w, 1);
        run<42>("v_max_i32", w, 1); run<43>("fma v,v,s", w, 1); run<44>("cndmask only", w, 1); run<45>("mul_e32 0.5,v", w, 1);
        run<46>("add_dpp wave_rol", w, 1); run<47>("add_dpp row_ror", w, 1);
        run<48>("v_and_or s,v", w, 1); run<49>("v_and_b32", w, 1); run<50>("v_or_b32", w, 1); run<51>("v_xor_b32", w, 1);
        run<52>("v_and_or v,v", w, 1); run<53>("mul |v|,v", w, 1); run<54>("v_lshl_or", w, 1); run<55>("v_mov", w, 1);
    }
    return 0;
}
