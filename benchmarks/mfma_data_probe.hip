// Probe (experiment): does the time of v_mfma_f32_32x32x16_f16 on gfx950 depend on the operand DATA (zero rows,
// tiny values) or on whether successive MFMAs form one dependent accumulator chain?  Motivation: folding the K-pass
// kernel's 16-row trunk-out layer into two MFMAs per k-step (rows 16..31 of A carrying W_lo instead of zeros) cut
// 4 of 42 MFMAs per pass and the kernel got 4 % SLOWER (benchmarks/exp_trunk_fold.sh).
// build: hipcc -w --offload-arch=gfx950 -O3 -o mfma_data_probe mfma_data_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// FILL: 0 dense random-ish A; 1 rows 16..31 of A zero; 2 A all zero; 3 A tiny (f16 subnormals); 4 A and B zero
// CHAINS: 1 = one dependent accumulator chain, 2 = two independent chains
template <int FILL, int CHAINS>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    f32x16 acc0 = {0}, acc1 = {0};
    f16x8 a, b;
    const int row = threadIdx.x & 31;
    for (int e = 0; e < 8; ++e) {
        float av = 0.37f * (float)(((threadIdx.x * 37 + e * 11) % 29) - 14) / 14.f;
        if (FILL == 1 && row >= 16) av = 0.f;
        if (FILL == 2 || FILL == 4) av = 0.f;
        if (FILL == 3) av *= 1e-5f;
        a[e] = (_Float16)av;
        b[e] = (FILL == 4) ? (_Float16)0.f : (_Float16)(0.21f * (float)(((threadIdx.x * 13 + e * 7) % 23) - 11) / 11.f);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (CHAINS == 2 && (k & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        }
        // keep the accumulators bounded without leaving the matrix pipe idle for long
        if ((i & 63) == 63) { acc0 *= 1e-3f; acc1 *= 1e-3f; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc0[9] + acc1[3];
}

template <int FILL, int CHAINS>
static float run(float* out, int iters, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<FILL, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, iters / 4);   // warm the clocks
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<FILL, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 100000;   // 1.6 M MFMAs per wave: ~25 ms per wave per SIMD at 32 cycles and 2 GHz
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256 * wps;
        const double n = (double)iters * 16 * wps;
        float t[5][2];
        t[0][0] = run<0, 1>(out, iters, blocks); t[0][1] = run<0, 2>(out, iters, blocks);
        t[1][0] = run<1, 1>(out, iters, blocks); t[1][1] = run<1, 2>(out, iters, blocks);
        t[2][0] = run<2, 1>(out, iters, blocks); t[2][1] = run<2, 2>(out, iters, blocks);
        t[3][0] = run<3, 1>(out, iters, blocks); t[3][1] = run<3, 2>(out, iters, blocks);
        t[4][0] = run<4, 1>(out, iters, blocks); t[4][1] = run<4, 2>(out, iters, blocks);
        const char* names[5] = {"dense", "half_rows_zero", "a_zero", "a_tiny", "all_zero"};
        printf("{\"waves_per_simd\": %d", wps);
        for (int f = 0; f < 5; ++f)
            printf(", \"%s\": {\"ms_one_chain\": %.2f, \"ms_two_chains\": %.2f, \"ns_per_mfma_one_chain\": %.2f, \"ns_per_mfma_two_chains\": %.2f}",
                   names[f], t[f][0], t[f][1], t[f][0] * 1e6 / n, t[f][1] * 1e6 / n);
        printf("}\n");
    }
    return 0;
}
