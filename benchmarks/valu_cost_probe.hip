// Probe (experiment): issue cost per wave64 instruction of the VALU instruction forms the hot kernels are made of, on
// gfx950, with 2 / 4 / 8 waves per SIMD (16 independent destination registers, 64 instructions per loop iteration).
// The issue-bound rooflines of bench.py price every VALU instruction at one flat cost; this table shows what each
// form really holds the SIMD's vector pipe for.
// build + run: hipcc -w --offload-arch=gfx950 -O3 -o /tmp/vcp benchmarks/valu_cost_probe.hip && /tmp/vcp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

#define BODY(ASM, ...)                                                                                       \
    for (int i = 0; i < iters; ++i) {                                                                        \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) _Pragma("unroll") for (int q = 0; q < 16; ++q)          \
            asm volatile(ASM : __VA_ARGS__);                                                                 \
    }

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float sx, unsigned su) {
    float x = threadIdx.x * 1e-3f + 0.5f, y = 1.0001f;
    unsigned ux = threadIdx.x * 2654435761u + 12345u, uy = 0x9E3779B9u ^ threadIdx.x;
    float v[16]; unsigned u[16]; f2 w[16];
    for (int q = 0; q < 16; ++q) { v[q] = x + q; u[q] = ux + q * 977u; w[q] = (f2){x + q, x - q}; }
    f2 yy = {y, y};
    if (KIND == 0) BODY("v_add_f32 %0, %0, %1", "+v"(v[q]) : "v"(x))
    if (KIND == 1) BODY("v_mul_f32 %0, %0, %1", "+v"(v[q]) : "v"(y))
    if (KIND == 2) BODY("v_fma_f32 %0, %0, %1, %2", "+v"(v[q]) : "v"(y), "v"(x))
    if (KIND == 3) BODY("v_fma_f32 %0, %0, %1, %2", "+v"(v[q]) : "s"(sx), "v"(x))
    if (KIND == 4) BODY("v_fmac_f32 %0, %1, %2", "+v"(v[q]) : "v"(y), "v"(x))
    if (KIND == 5) BODY("v_fmac_f32 %0, %1, %2", "+v"(v[q]) : "s"(sx), "v"(x))
    if (KIND == 6) BODY("v_fma_f32 %0, %0, %1, 1.0", "+v"(v[q]) : "v"(y))
    if (KIND == 7) BODY("v_exp_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 8) BODY("v_log_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 9) BODY("v_rcp_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 10) BODY("v_sqrt_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 11) BODY("v_sin_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 12) BODY("v_mul_lo_u32 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 13) BODY("v_mul_lo_u32 %0, %0, %1", "+v"(u[q]) : "s"(su))
    if (KIND == 14) BODY("v_mul_u32_u24 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 15) BODY("v_mad_u32_u24 %0, %0, %1, %2", "+v"(u[q]) : "v"(uy), "v"(ux))
    if (KIND == 16) BODY("v_xor_b32 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 17) BODY("v_and_b32 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 18) BODY("v_lshlrev_b32 %0, 3, %0", "+v"(u[q]) : )
    if (KIND == 19) BODY("v_add_u32 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 20) BODY("v_add3_u32 %0, %0, %1, %2", "+v"(u[q]) : "v"(uy), "v"(ux))
    if (KIND == 21) BODY("v_xad_u32 %0, %0, %1, %2", "+v"(u[q]) : "v"(uy), "v"(ux))
    if (KIND == 22) BODY("v_lshl_add_u32 %0, %0, 3, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 23) BODY("v_pk_sub_i16 %0, %0, %1 clamp", "+v"(u[q]) : "v"(uy))
    if (KIND == 24) BODY("v_pk_ashrrev_i16 %0, 15, %0", "+v"(u[q]) : )
    if (KIND == 25) BODY("v_pk_max_i16 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 26) BODY("v_pk_mul_f16 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 27) BODY("v_pk_fma_f16 %0, %0, %1, %2", "+v"(u[q]) : "v"(uy), "v"(ux))
    if (KIND == 28) BODY("v_cvt_f16_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 29) BODY("v_cvt_pk_f16_f32 %0, %0, %1", "+v"(v[q]) : "v"(x))
    if (KIND == 30) BODY("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]", "+v"(v[q]) : "v"(y), "v"(x))
    if (KIND == 31) BODY("v_perm_b32 %0, %0, %1, %2", "+v"(u[q]) : "v"(uy), "v"(ux))
    if (KIND == 32) BODY("v_cndmask_b32 %0, %0, %1, vcc", "+v"(u[q]) : "v"(uy))
    if (KIND == 33) BODY("v_mov_b32 %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 34) BODY("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(u[q]) : "v"(uy))
    if (KIND == 35) BODY("v_floor_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 36) BODY("v_fract_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 37) BODY("v_cvt_i32_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 38) BODY("v_cvt_f32_u32 %0, %0", "+v"(v[q]) : )
    if (KIND == 39) BODY("v_max_f32 %0, %0, %1", "+v"(v[q]) : "v"(x))
    if (KIND == 40) BODY("v_max3_f32 %0, %0, %1, %2", "+v"(v[q]) : "v"(x), "v"(y))
    if (KIND == 41) BODY("v_sub_f32 %0, %1, %0", "+v"(v[q]) : "v"(x))
    if (KIND == 42) BODY("v_pk_fma_f32 %0, %0, %1, %1", "+v"(w[q]) : "v"(yy))
    if (KIND == 43) BODY("v_mul_f32 %0, %0, %1", "+v"(v[q]) : "s"(sx))
    if (KIND == 44) BODY("v_mul_f32 %0, 0x40490fdb, %0", "+v"(v[q]) : )
    if (KIND == 45) BODY("v_alignbit_b32 %0, %0, %1, 7", "+v"(u[q]) : "v"(uy))
    if (KIND == 46) BODY("v_bfe_u32 %0, %0, 3, 9", "+v"(u[q]) : )
    if (KIND == 47) BODY("v_mul_hi_u32 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 48) BODY("v_cmp_lt_f32 vcc, %0, %1", "+v"(v[q]) : "v"(x) : "vcc")
    if (KIND == 49) BODY("v_ldexp_f32 %0, %0, %1", "+v"(v[q]) : "v"(ux))
    if (KIND == 50) { unsigned long long m = 0x5555aaaa5555aaaaull ^ su; BODY("v_cndmask_b32_e64 %0, %0, %1, %2", "+v"(u[q]) : "v"(uy), "s"(m)) }
    if (KIND == 51) BODY("v_ceil_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 52) BODY("v_trunc_f32 %0, %0", "+v"(v[q]) : )
    if (KIND == 53) BODY("v_cmp_ne_u32 vcc, %0, %1", "+v"(u[q]) : "v"(uy) : "vcc")
    if (KIND == 54) BODY("v_or_b32 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 55) BODY("v_sub_u32 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 56) BODY("v_lshrrev_b32 %0, 5, %0", "+v"(u[q]) : )
    if (KIND == 57) BODY("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", "+v"(u[q]) : "v"(uy), "v"(ux))
    if (KIND == 58) BODY("v_add_f32 %0, %0, %1", "+v"(v[q]) : "s"(sx))
    if (KIND == 59) BODY("v_min_u32 %0, %0, %1", "+v"(u[q]) : "v"(uy))
    if (KIND == 60) BODY("v_add_lshl_u32 %0, %0, %1, 3", "+v"(u[q]) : "v"(uy))
    if (KIND == 61) BODY("v_med3_f32 %0, %0, %1, %2", "+v"(v[q]) : "v"(x), "v"(y))
    if (KIND == 62) BODY("v_and_b32 %0, %0, %1", "+v"(u[q]) : "s"(su))
    if (KIND == 63) { asm volatile("s_mov_b64 vcc, 0x5555aaaa" ::: "vcc"); BODY("v_cndmask_b32_e32 %0, %0, %1, vcc", "+v"(u[q]) : "v"(uy)) }
    if (KIND == 64) { asm volatile("s_mov_b64 vcc, 0x5555aaaa" ::: "vcc"); BODY("v_cndmask_b32_e64 %0, %0, %1, vcc", "+v"(u[q]) : "v"(uy)) }
    if (KIND == 65) { asm volatile("s_mov_b64 vcc, 0x5555aaaa" ::: "vcc"); BODY("v_cndmask_b32_e32 %0, %1, %2, vcc", "=v"(u[q]) : "v"(uy), "v"(ux)) }
    if (KIND == 66) BODY("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %1, %2, vcc", "=v"(v[q]) : "v"(x), "v"(y) : "vcc")
    if (KIND == 68) BODY("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %1, %2, vcc\n\tv_cndmask_b32_e32 %0, %2, %1, vcc\n\tv_cndmask_b32_e32 %0, %1, %2, vcc", "=v"(v[q]) : "v"(x), "v"(y) : "vcc")
    if (KIND == 69) { unsigned long long m; BODY("v_cmp_lt_f32_e64 %1, %2, %3\n\tv_cndmask_b32_e64 %0, %2, %3, %1\n\tv_cndmask_b32_e64 %0, %3, %2, %1\n\tv_cndmask_b32_e64 %0, %2, %3, %1", "=v"(v[q]), "=s"(m) : "v"(x), "v"(y)) }
    if (KIND == 70) BODY("v_cmp_lt_f32 vcc, %1, %2\n\tv_add_f32 %0, %1, %2\n\tv_cndmask_b32_e32 %0, %1, %2, vcc\n\tv_mul_f32 %0, %1, %2\n\tv_cndmask_b32_e32 %0, %2, %1, vcc", "=v"(v[q]) : "v"(x), "v"(y) : "vcc")
    if (KIND == 67) { unsigned long long m; BODY("v_cmp_lt_f32_e64 %1, %2, %3\n\tv_cndmask_b32_e64 %0, %2, %3, %1", "=v"(v[q]), "=s"(m) : "v"(x), "v"(y)) }
    float r = 0.f;
    for (int q = 0; q < 16; ++q) r += v[q] + w[q][0] + w[q][1] + (float)u[q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
static void run(const char* name, float* out) {
    const int iters = 8000;
    printf("{\"inst\": \"%s\"", name);
    for (int waves_per_simd : {2, 4, 8}) {
        const int blocks = 256 * waves_per_simd;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0x9E3779B1u);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0x9E3779B1u);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ns_per_inst = (double)ms * 1e6 / ((double)waves_per_simd * iters * 64.0);
        printf(", \"ns_w%d\": %.3f", waves_per_simd, ns_per_inst);
        if (waves_per_simd == 8) printf(", \"rel_to_v_add_f32\": null");
    }
    printf("}\n");
}

#define RUN(K, NAME) run<K>(NAME, out)
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    RUN(0, "v_add_f32 v,v"); RUN(1, "v_mul_f32 v,v"); RUN(41, "v_sub_f32 v,v"); RUN(39, "v_max_f32 v,v"); RUN(43, "v_mul_f32 v,s"); RUN(44, "v_mul_f32 literal,v");
    RUN(2, "v_fma_f32 v,v,v"); RUN(3, "v_fma_f32 v,s,v"); RUN(6, "v_fma_f32 v,v,1.0"); RUN(4, "v_fmac_f32 v,v"); RUN(5, "v_fmac_f32 s,v"); RUN(40, "v_max3_f32 v,v,v"); RUN(42, "v_pk_fma_f32");
    RUN(7, "v_exp_f32"); RUN(8, "v_log_f32"); RUN(9, "v_rcp_f32"); RUN(10, "v_sqrt_f32"); RUN(11, "v_sin_f32");
    RUN(12, "v_mul_lo_u32 v,v"); RUN(13, "v_mul_lo_u32 v,s"); RUN(47, "v_mul_hi_u32 v,v"); RUN(14, "v_mul_u32_u24 v,v"); RUN(15, "v_mad_u32_u24 v,v,v");
    RUN(16, "v_xor_b32"); RUN(17, "v_and_b32"); RUN(18, "v_lshlrev_b32 imm"); RUN(19, "v_add_u32"); RUN(20, "v_add3_u32"); RUN(21, "v_xad_u32"); RUN(22, "v_lshl_add_u32");
    RUN(45, "v_alignbit_b32"); RUN(46, "v_bfe_u32");
    RUN(23, "v_pk_sub_i16 clamp"); RUN(24, "v_pk_ashrrev_i16"); RUN(25, "v_pk_max_i16"); RUN(26, "v_pk_mul_f16"); RUN(27, "v_pk_fma_f16");
    RUN(28, "v_cvt_f16_f32"); RUN(29, "v_cvt_pk_f16_f32"); RUN(30, "v_fma_mix_f32"); RUN(31, "v_perm_b32"); RUN(32, "v_cndmask_b32"); RUN(33, "v_mov_b32");
    RUN(34, "v_mov_b32_dpp row_shr"); RUN(35, "v_floor_f32"); RUN(36, "v_fract_f32"); RUN(37, "v_cvt_i32_f32"); RUN(38, "v_cvt_f32_u32"); RUN(48, "v_cmp_lt_f32"); RUN(49, "v_ldexp_f32");
    RUN(50, "v_cndmask_b32_e64 sgpr mask"); RUN(51, "v_ceil_f32"); RUN(52, "v_trunc_f32"); RUN(53, "v_cmp_ne_u32"); RUN(54, "v_or_b32"); RUN(55, "v_sub_u32");
    RUN(56, "v_lshrrev_b32 imm"); RUN(57, "v_bitop3_b32"); RUN(58, "v_add_f32 v,s"); RUN(59, "v_min_u32"); RUN(60, "v_add_lshl_u32"); RUN(61, "v_med3_f32"); RUN(62, "v_and_b32 v,s");
    RUN(63, "v_cndmask_b32_e32 vcc (set)"); RUN(64, "v_cndmask_b32_e64 vcc (set)"); RUN(65, "v_cndmask_b32_e32 vcc, dst != src"); RUN(66, "v_cmp_lt_f32 vcc + v_cndmask e32 (pair)"); RUN(67, "v_cmp_lt_f32 sgpr + v_cndmask e64 (pair)");
    RUN(68, "v_cmp vcc + 3 x v_cndmask e32 (4 insts)"); RUN(69, "v_cmp sgpr + 3 x v_cndmask e64 (4 insts)"); RUN(70, "v_cmp vcc, add, cndmask e32, mul, cndmask e32 (5 insts)");
    return 0;
}
