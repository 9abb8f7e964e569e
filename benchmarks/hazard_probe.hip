// Probe (experiment): how many wait states does gfx950 need between a VALU instruction that writes a VGPR and a
// v_mfma_f32_32x32x16_f16 that reads it as its B operand?  The compiler keeps its own producers two or more wait states
// away (GCNHazardRecognizer) and does not look inside inline assembly; the split-f16 kernels complete their B operands
// with v_fma_mixhi_f16 in inline assembly (unerf_nerf.hip: mf16_split8).  Here producer, gap and MFMA sit in ONE
// assembly statement, so the gap is exactly what is written; each lane checks its 16 accumulators against the sums it
// computes from the values it wrote.
// build: hipcc -w --offload-arch=gfx950 -O2 -o hazard_probe hazard_probe.hip ; run: ./hazard_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define GAP0 ""
#define GAP1 "s_nop 0\n\t"
#define GAP2 "s_nop 1\n\t"
#define GAP3 "s_nop 2\n\t"

// PRODUCER 0: v_mov_b32 (full-register write)    1: v_fma_mixhi_f16 completing a register whose low half is in place
template <int PRODUCER, int GAP>
__global__ __launch_bounds__(256) void probe(unsigned int* bad_by_lane, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)1.0f;           // every row of A is ones: acc[row][col] = sum_k B[k][col]
    unsigned int bad = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t w[4];
        float expect = 0.f, hi_f[4];
        for (int p = 0; p < 4; ++p) {
            const float lo = (float)((i * 7 + lane * 3 + 2 * p) & 15), hi = (float)((i * 5 + lane + 2 * p + 1) & 15);
            const f16x2 hh = {(_Float16)lo, (_Float16)hi};
            w[p] = __builtin_bit_cast(uint32_t, hh);
            hi_f[p] = hi;
            expect += lo + hi;
        }
        expect += __shfl_xor(expect, 32, 64);                     // the other k-group of this column
        f32x16 acc = {0};
        // stale contents of the B registers: last iteration's values with every half + 1 (a hazard shows as a wrong sum)
        if (PRODUCER == 0) {
#define BODY(G)                                                                                                      \
            asm volatile("v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\t" G        \
                         "v_mfma_f32_32x32x16_f16 %0, %1, v[40:43], %0\n\ts_nop 15\n\ts_nop 7"                           \
                         : "+v"(acc) : "v"(a), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : "v40", "v41", "v42", "v43")
            if (GAP == 0) BODY(GAP0); else if (GAP == 1) BODY(GAP1); else if (GAP == 2) BODY(GAP2); else BODY(GAP3);
#undef BODY
        } else {
            // low halves placed early (and settled), high halves written by v_fma_mixhi_f16 (0 * x + hi) right before the MFMA
            uint32_t lo_only[4];
            for (int p = 0; p < 4; ++p) lo_only[p] = (w[p] & 0xFFFFu) | 0x7E000000u;   // high half: a NaN until it is written
            float h0 = hi_f[0], h1 = hi_f[1], h2 = hi_f[2], h3 = hi_f[3];
#define BODY(G)                                                                                                      \
            asm volatile("v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\ts_nop 7\n\t" \
                         "v_fma_mixhi_f16 v40, 0, 0, %6 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\t"                              \
                         "v_fma_mixhi_f16 v41, 0, 0, %7 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\t"                              \
                         "v_fma_mixhi_f16 v42, 0, 0, %8 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\t"                              \
                         "v_fma_mixhi_f16 v43, 0, 0, %9 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\t" G                            \
                         "v_mfma_f32_32x32x16_f16 %0, %1, v[40:43], %0\n\ts_nop 15\n\ts_nop 7"                           \
                         : "+v"(acc) : "v"(a), "v"(lo_only[0]), "v"(lo_only[1]), "v"(lo_only[2]), "v"(lo_only[3]),      \
                           "v"(h0), "v"(h1), "v"(h2), "v"(h3) : "v40", "v41", "v42", "v43")
            if (GAP == 0) BODY(GAP0); else if (GAP == 1) BODY(GAP1); else if (GAP == 2) BODY(GAP2); else BODY(GAP3);
#undef BODY
        }
        for (int r = 0; r < 16; ++r) bad += (acc[r] != expect) ? 1u : 0u;
    }
    atomicAdd(&bad_by_lane[lane], bad);
}

template <int PRODUCER, int GAP>
static void run(unsigned int* d, const char* name) {
    hipMemset(d, 0, 64 * sizeof(unsigned int));
    hipLaunchKernelGGL((probe<PRODUCER, GAP>), dim3(1024), dim3(256), 0, 0, d, 200);
    unsigned int h[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long rows[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) rows[l >> 4] += h[l];
    printf("{\"producer\": \"%s\", \"wait_states_between\": %d, \"wrong_accumulators_by_16_lane_row\": [%llu, %llu, %llu, %llu], "
           "\"checked\": %llu}\n", name, GAP, rows[0], rows[1], rows[2], rows[3], 1024ull * 256 * 200 * 16);
}

int main() {
    unsigned int* d;
    hipMalloc(&d, 64 * sizeof(unsigned int));
    run<0, 0>(d, "v_mov_b32"); run<0, 1>(d, "v_mov_b32"); run<0, 2>(d, "v_mov_b32"); run<0, 3>(d, "v_mov_b32");
    run<1, 0>(d, "v_fma_mixhi_f16"); run<1, 1>(d, "v_fma_mixhi_f16"); run<1, 2>(d, "v_fma_mixhi_f16"); run<1, 3>(d, "v_fma_mixhi_f16");
    return 0;
}
