#!/usr/bin/env python
"""Generates benchmarks/valu_rate_probe.hip (experiment): the issue cost of the VALU instructions the field
kernels are made of, on gfx950, relative to v_fma_f32 (4 cycles per wave64 instruction).  Each kernel issues 64
copies of one instruction per loop iteration on rotating registers (no back-to-back dependencies), at 1, 2 and 4
waves per SIMD.

    python benchmarks/valu_rate_probe.py && hipcc -w --offload-arch=gfx950 -O3 -o valu_rate_probe benchmarks/valu_rate_probe.hip
"""
import os

INSTS = {
    "v_fma_f32": "v_fma_f32 {d}, {s}, {b}, {c}",
    "v_pk_fma_f32": "v_pk_fma_f32 {D}, {S}, {S}, {S}",
    "v_pk_mul_f32": "v_pk_mul_f32 {D}, {S}, {S}",
    "v_pk_add_f32": "v_pk_add_f32 {D}, {S}, {S}",
    "v_mul_f32": "v_mul_f32 {d}, {s}, {b}",
    "v_max_f32": "v_max_f32 {d}, {s}, {b}",
    "v_med3_f32": "v_med3_f32 {d}, {s}, {b}, {c}",
    "v_max_i32": "v_max_i32 {d}, {s}, {b}",
    "v_and_b32": "v_and_b32 {d}, {s}, {b}",
    "v_add_u32": "v_add_u32 {d}, {s}, {b}",
    "v_lshrrev_b32": "v_lshrrev_b32 {d}, 17, {s}",
    "v_bitop3_b32": "v_bitop3_b32 {d}, {s}, {b}, {c} bitop3:0x96",
    "v_lshl_add_u32": "v_lshl_add_u32 {d}, {s}, 6, {b}",
    "v_alignbit_b32": "v_alignbit_b32 {d}, {s}, {s}, 22",
    "v_cvt_pk_f16_f32": "v_cvt_pk_f16_f32 {d}, {s}, {b}",
    "v_fma_mixlo_f16": "v_fma_mixlo_f16 {d}, {s}, -1.0, {b} op_sel:[0,0,0] op_sel_hi:[1,0,0]",
    "v_fma_mixhi_f16": "v_fma_mixhi_f16 {d}, {s}, -1.0, {b} op_sel:[1,0,0] op_sel_hi:[1,0,0]",
    "v_pk_sub_i16_clamp": "v_pk_sub_i16 {d}, {s}, {b} clamp",
    "v_pk_ashrrev_i16": "v_pk_ashrrev_i16 {d}, 15, {s} op_sel_hi:[0,1]",
    "v_pk_max_f16": "v_pk_max_f16 {d}, {s}, {b}",
    "v_pk_max_i16": "v_pk_max_i16 {d}, {s}, {b}",
    "v_cmp_lt_i32_sdwa+v_cndmask": "v_cmp_lt_i32_sdwa vcc, sext({s}), sext({b}) src0_sel:WORD_1 src1_sel:WORD_0\\nv_cndmask_b32 {d}, 0, {s}, vcc",
    "v_cmp_lt_i32+v_cndmask": "v_cmp_lt_i32 vcc, {s}, {b}\\nv_cndmask_b32 {d}, 0, {s}, vcc",
    "v_mov_b64": "v_mov_b64 {D}, {S}",
    "v_mov_b32": "v_mov_b32 {d}, {s}",
    "v_mul_lo_u32": "v_mul_lo_u32 {d}, {s}, {b}",
    "v_mad_u32_u24": "v_mad_u32_u24 {d}, {s}, {b}, {c}",
    "v_exp_f32": "v_exp_f32 {d}, {s}",
    "v_rcp_f32": "v_rcp_f32 {d}, {s}",
    "v_cvt_i32_f32": "v_cvt_i32_f32 {d}, {s}",
    "v_floor_f32": "v_floor_f32 {d}, {s}",
    "v_permlane32_swap_b32": "v_permlane32_swap_b32 {d}, {s}",
    "v_add_f32_dpp": "v_add_f32_dpp {d}, {s}, {b} row_shr:1 row_mask:0xf bank_mask:0xf",
    # second batch (r2): which instructions belong to the class that issues in ~2 cycles per wave64 at >= 2 waves per SIMD?
    "v_add_f32": "v_add_f32 {d}, {s}, {b}",
    "v_sub_f32": "v_sub_f32 {d}, {s}, {b}",
    "v_add_f32_abs_e64": "v_add_f32_e64 {d}, {s}, |{s}|",
    "v_mul_f32_e64_neg": "v_mul_f32_e64 {d}, -{s}, {b}",
    "v_fmac_f32": "v_fmac_f32 {d}, {s}, {b}",
    "v_min_f32": "v_min_f32 {d}, {s}, {b}",
    "v_or_b32": "v_or_b32 {d}, {s}, {b}",
    "v_xor_b32": "v_xor_b32 {d}, {s}, {b}",
    "v_sub_u32": "v_sub_u32 {d}, {s}, {b}",
    "v_ashrrev_i32": "v_ashrrev_i32 {d}, 31, {s}",
    "v_lshlrev_b32": "v_lshlrev_b32 {d}, 6, {s}",
    "v_cndmask_b32": "v_cndmask_b32 {d}, {s}, {b}, vcc",
    "v_max_u32": "v_max_u32 {d}, {s}, {b}",
    "v_bfe_i32": "v_bfe_i32 {d}, {s}, 3, 1",
    "v_bfi_b32": "v_bfi_b32 {d}, {s}, {b}, {c}",
    "v_and_or_b32": "v_and_or_b32 {d}, {s}, {b}, {c}",
    "v_or3_b32": "v_or3_b32 {d}, {s}, {b}, {c}",
    "v_add3_u32": "v_add3_u32 {d}, {s}, {b}, {c}",
    "v_xad_u32": "v_xad_u32 {d}, {s}, {b}, {c}",
    "v_lshl_or_b32": "v_lshl_or_b32 {d}, {s}, 6, {b}",
    "v_perm_b32": "v_perm_b32 {d}, {s}, {b}, {c}",
    "v_mul_u32_u24": "v_mul_u32_u24 {d}, {s}, {b}",
    "v_cvt_f32_f16": "v_cvt_f32_f16 {d}, {s}",
    "v_cvt_f32_f16_sdwa_hi": "v_cvt_f32_f16_sdwa {d}, {s} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",
    "v_cvt_f16_f32": "v_cvt_f16_f32 {d}, {s}",
    "v_cvt_u32_f32": "v_cvt_u32_f32 {d}, {s}",
    "v_cvt_f32_u32": "v_cvt_f32_u32 {d}, {s}",
    "v_fract_f32": "v_fract_f32 {d}, {s}",
    "v_rndne_f32": "v_rndne_f32 {d}, {s}",
    "v_pk_fma_f16": "v_pk_fma_f16 {d}, {s}, {b}, {c}",
    "v_pk_mul_f16": "v_pk_mul_f16 {d}, {s}, {b}",
    "v_pk_add_u16": "v_pk_add_u16 {d}, {s}, {b}",
    "v_and_b32_sdwa_sext": "v_and_b32_sdwa {d}, {s}, sext({b}) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1",
    "v_mul_f32_sdwa": "v_mul_f32_sdwa {d}, {s}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD",
    "v_cmp_lt_i32": "v_cmp_lt_i32 vcc, {s}, {b}",
    "v_cmp_lt_i32_e64_sgpr": "v_cmp_lt_i32_e64 s[20:21], {s}, {b}",
    "v_log_f32": "v_log_f32 {d}, {s}",
    "v_sqrt_f32": "v_sqrt_f32 {d}, {s}",
    "v_sin_f32": "v_sin_f32 {d}, {s}",
    "v_ldexp_f32": "v_ldexp_f32 {d}, {s}, {b}",
    "v_mad_u64_u32": "v_mad_u64_u32 {D}, vcc, {s}, {b}, {S}",
    "v_mul_hi_u32": "v_mul_hi_u32 {d}, {s}, {b}",
    # mixes: does a fast-class instruction keep its rate next to 4-cycle instructions / MFMAs of the same or another wave?
    "mix:v_add_f32_abs_e64+v_max_i32": "v_add_f32_e64 {d}, {s}, |{s}|\\nv_max_i32 {d2}, {s}, {b}",
    "mix:v_and_b32+v_pk_fma_f32": "v_and_b32 {d}, {s}, {b}\\nv_pk_fma_f32 {D}, {S}, {S}, {S}",
    "mix:v_and_b32+v_alignbit_b32": "v_and_b32 {d}, {s}, {b}\\nv_alignbit_b32 {d2}, {s}, {s}, 22",
    "mix:3x v_and_b32+v_alignbit_b32": "v_and_b32 {d}, {s}, {b}\\nv_xor_b32 {d2}, {s}, {b}\\nv_or_b32 {d}, {s}, {b}\\nv_alignbit_b32 {d2}, {s}, {s}, 22",
}


def main():
    src = ["// GENERATED by benchmarks/valu_rate_probe.py -- issue cost of VALU instructions on gfx950 (experiment)\n"
           "#include <hip/hip_runtime.h>\n#include <stdio.h>\n"]
    names = list(INSTS)
    for i, name in enumerate(names):
        body = []
        for k in range(64):
            body.append(INSTS[name].format(d=f"%{k % 8}", d2=f"%{(k + 5) % 8}", s=f"%{(k + 3) % 8}", b="%12", c="%13",
                                           D=f"%{8 + k % 4}", S=f"%{8 + (k + 1) % 4}"))
        asm = "\\n".join(body)
        src.append(f'''__global__ __launch_bounds__(256) void k{i}(float* out, int iters) {{
    unsigned r0 = threadIdx.x, r1 = r0 * 3u + 1u, r2 = r0 * 5u + 7u, r3 = r0 ^ 0x1234u, r4 = r0 + 99u, r5 = r0 * 7u, r6 = r0 + 5u, r7 = r0 * 11u;
    double q0 = r0, q1 = r1, q2 = r2, q3 = r3;
    unsigned b = 0x3F800100u, c = 0x4CCD4CCDu;
    for (int i = 0; i < iters; ++i)
        asm volatile("{asm}" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7),
                     "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b), "v"(c) : "vcc", "s20", "s21");
    out[blockIdx.x * 256 + threadIdx.x] = (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7) + (float)(q0 + q1 + q2 + q3);
}}
''')
    src.append('''typedef void (*kern_t)(float*, int);
static float run(kern_t k, float* out, int iters, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    const int iters = 20000;
''')
    src.append("    struct { const char* n; kern_t k; int per; } ks[] = {"
               + ", ".join(f'{{"{n}", k{i}, {4 if n.startswith("mix:3x") else (2 if "+" in n else 1)}}}' for i, n in enumerate(names)) + "};\n")
    src.append('''    for (int wps = 1; wps <= 4; wps *= 2) {
        float base = run(k0, out, iters, 256 * wps);
        printf("{\\"waves_per_simd\\": %d, ", wps);
        for (auto& e : ks) printf("\\"%s\\": %.2f, ", e.n, 4.0 * run(e.k, out, iters, 256 * wps) / base / e.per);
        printf("\\"unit\\": \\"cycles per wave64 instruction (v_fma_f32 = 4)\\", \\"ms_v_fma_f32\\": %.3f}\\n", base);
    }
    return 0;
}
''')
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_rate_probe.hip")
    open(out, "w").write("".join(src))


if __name__ == "__main__":
    main()
