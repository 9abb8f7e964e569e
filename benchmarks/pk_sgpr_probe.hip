// Round 5, DESIGN 4.5.73: the SLP vectoriser's packed-fp32 forms that appear ONLY in the non-reproducible fused-blend build
// are the ones with a SCALAR operand: v_pk_mul_f32 v[..], v[..], s[n:n+1] (an SGPR PAIR as a 64-bit source), and inline
// constants with op_sel_hi:[1,0].  This probe runs those forms from inline assembly on every lane of every SIMD, many times,
// next to waves that keep the scalar unit busy, and counts results that differ from the lane-wise fp32 product.
//   hipcc --offload-arch=gfx950 -O2 -o pk_sgpr_probe benchmarks/pk_sgpr_probe.hip && ./pk_sgpr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void probe(const float* in, unsigned long long* bad, int iters, float slo, float shi) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    v2f a = {in[2 * (t & 1023)], in[2 * (t & 1023) + 1]};
    unsigned long long wrong_pair = 0, wrong_rep = 0, wrong_inl = 0, wrong_neg = 0;
    if ((threadIdx.x >> 6) & 1) {   // odd waves: scalar-unit noise (uniform loads + SALU chains) beside the probing waves
        uint32_t s = 0;
        for (int i = 0; i < iters * 4; ++i) {
            const float* p = in + ((s + i) & 1023);
            s += __builtin_amdgcn_readfirstlane(__float_as_uint(*p)) * 2654435761u + 12345u;
        }
        if (s == 0xdeadbeefu) bad[7] = s;
        return;
    }
    // the scalar pair is made by SALU moves of the kernel arguments (uniform): slo / shi distinct
    for (int i = 0; i < iters; ++i) {
        v2f r0, r1, r2;
        asm volatile("s_mov_b32 s40, %3\n\ts_mov_b32 s41, %4\n\ts_nop 4\n\t"
                     "v_pk_mul_f32 %0, %5, s[40:41]\n\t"
                     "v_pk_add_f32 %1, %5, s[40:41] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_mul_f32 %2, %5, 0.5 op_sel_hi:[1,0]\n\t"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "s"(slo), "s"(shi), "v"(a) : "s40", "s41");
        // expectations: (a) the pair is read as 64 bits: hi lane x shi; (b) only the low 32 bits, replicated: hi lane x slo
        if (r0.x != a.x * slo) ++wrong_pair, ++wrong_rep;
        if (r0.y != a.y * shi) ++wrong_pair;
        if (r0.y != a.y * slo) ++wrong_rep;
        if (r1.x != a.x - slo || r1.y != a.y - shi) ++wrong_neg;
        if (r2.x != a.x * 0.5f || r2.y != a.y * 0.5f) ++wrong_inl;
        a.x += 1.0f; a.y -= 0.5f;
    }
    if (wrong_pair) atomicAdd(&bad[0], wrong_pair);
    if (wrong_rep) atomicAdd(&bad[1], wrong_rep);
    if (wrong_neg) atomicAdd(&bad[2], wrong_neg);
    if (wrong_inl) atomicAdd(&bad[3], wrong_inl);
}

int main() {
    float* in; unsigned long long* bad; float h[2048];
    for (int i = 0; i < 2048; ++i) h[i] = 0.37f * (float)(i % 97) - 11.f;
    hipMalloc(&in, sizeof(h)); hipMalloc(&bad, 64); hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice); hipMemset(bad, 0, 64);
    const int iters = 20000, blocks = 4096;
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, in, bad, iters, 3.0f, 7.0f);
    hipDeviceSynchronize();
    unsigned long long r[8]; hipMemcpy(r, bad, 64, hipMemcpyDeviceToHost);
    const double n = 5.0 * blocks * 128 * iters;
    printf("{\"checks_per_form\": %.0f, \"v_pk_mul_f32_sgpr_pair_differs_from_64bit_read\": %llu, \"..._from_low32_replicated\": %llu, "
           "\"v_pk_add_f32_sgpr_pair_neg_differs\": %llu, \"v_pk_mul_f32_inline_const_differs\": %llu}\n", n, r[0], r[1], r[2], r[3]);
    return 0;
}
