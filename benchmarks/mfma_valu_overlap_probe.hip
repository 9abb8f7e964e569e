// Probe (experiment): do v_mfma_f32_32x32x16_f16 and fp32 VALU instructions overlap on gfx950 -- inside one
// wave (interleaved or clustered instruction streams) and between the waves of a SIMD (1, 2, 4 waves per SIMD)?
// The split-f16 field kernels spend 4 cycles per VALU instruction + 32 per MFMA and the two ADD in every profile;
// this measures whether any instruction order or occupancy makes them overlap instead.
// build: hipcc -w --offload-arch=gfx950 -O3 -o mfma_valu_overlap_probe mfma_valu_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MODE 0: MFMA only (NM per iteration, two independent accumulator chains)
// MODE 1: VALU only (NV per iteration, 8 independent fma chains)
// MODE 2: both, clustered: NM MFMAs then NV VALU
// MODE 3: both, interleaved: after every MFMA, NV/NM VALU
// MODE 4: both, DEPENDENT like the field kernel: VALU consumes the accumulator of the MFMAs before it and
//         produces the B operand of the MFMAs after it
template <int MODE, int NM, int NV>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    f32x16 acc0 = {0}, acc1 = {0};
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * e); }
    float y = 1.0001f, x = threadIdx.x * 1e-3f;
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = x + q;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < NM; k += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            }
        }
        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < NV / 8; ++k)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaf(v[q], y, x);
        }
        if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NV / NM; ++q) v[q & 7] = fmaf(v[q & 7], y, x);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 4) {
            // MFMAs -> VALU on their accumulator -> next B operand (a serial chain, as in the pass loop)
#pragma unroll
            for (int k = 0; k < NM; ++k) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < NV; ++q) s = fmaf(acc0[q & 15], y, s);
            b[0] = (_Float16)s;
        }
    }
    float r = acc0[0] + acc0[5] + acc1[3];
    for (int q = 0; q < 8; ++q) r += v[q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE, int NM, int NV>
static float run(float* out, int iters, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, NM, NV>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, NM, NV>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 4000;
    const int NM = 16, NV = 128;   // 16 x 32 = 512 MFMA cycles, 128 x 4 = 512 VALU cycles per iteration and wave
    for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD = blocks per CU (a 256-thread block = 1 wave per SIMD)
        const int blocks = 256 * wps;
        float m = run<0, NM, NV>(out, iters, blocks), v = run<1, NM, NV>(out, iters, blocks);
        float c = run<2, NM, NV>(out, iters, blocks), il = run<3, NM, NV>(out, iters, blocks), dep = run<4, NM, NV>(out, iters, blocks);
        printf("{\"waves_per_simd\": %d, \"ms_mfma_only\": %.3f, \"ms_valu_only\": %.3f, \"ms_clustered\": %.3f, "
               "\"ms_interleaved\": %.3f, \"ms_dependent_chain\": %.3f, \"clk_per_mfma\": %.1f, \"clk_per_valu\": %.2f}\n",
               wps, m, v, c, il, dep, m * 1e-3 * 2.4e9 / (double)(iters * NM * wps), v * 1e-3 * 2.4e9 / (double)(iters * NV * wps));
    }
    return 0;
}
