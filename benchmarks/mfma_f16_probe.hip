// Probe (experiment): v_mfma_f32_32x32x8_f16 on gfx950 -- (1) are f16 subnormal inputs honoured, (2) issue rate
// against v_mfma_f32_32x32x2_f32, (3) does it overlap with fp32 VALU work from another wave?
// build: hipcc -w --offload-arch=gfx950 -O3 -o mfma_f16_probe mfma_f16_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__global__ void denorm_kernel(float* out) {
    // A = all 1.0, B[k][col] = 2^-20 (f16 subnormal): D = 8 * 2^-20 if subnormals are honoured, 0 if flushed
    f16x4 a = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
    _Float16 tiny = (_Float16)9.5367431640625e-07f;  // 2^-20
    f16x4 b = {tiny, tiny, tiny, tiny};
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)tiny; }
}

template <int MODE>  // 0: f32 mfma chain, 1: f16 mfma chain, 2: f16 mfma + VALU in the same wave, 3: VALU only
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    f32x16 acc = {0};
    f16x4 a = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)2.f}, b = a;
    float x = threadIdx.x * 1e-3f, y = 1.0001f, v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, acc, 0, 0, 0);
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int k = 0; k < 64; ++k) { v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[5] + v0 + v1 + v2 + v3;
}

template <int MODE>
static float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<MODE>), dim3(256 * 2), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate_kernel<MODE>), dim3(256 * 2), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 2 * 256 * 4);
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, out);
    float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("{\"f16_subnormal_input\": %g, \"mfma_sum_of_8\": %g, \"expected_if_honoured\": %g}\n", h[1], h[0], 8 * 9.5367431640625e-07);
    const int iters = 2000;
    // per SIMD: 2 blocks/CU * 4 waves / 4 SIMDs = 2 waves per SIMD, each issues iters*16 MFMAs
    float t0 = run<0>(out, iters), t1 = run<1>(out, iters), t2 = run<2>(out, iters), t3 = run<3>(out, iters);
    double mf = 2.0 * iters * 16;  // MFMAs per SIMD
    printf("{\"clk_per_mfma_f32_32x32x2\": %.1f, \"clk_per_mfma_f16_32x32x8\": %.1f, \"ms_f16_mfma\": %.3f, \"ms_valu_only\": %.3f, \"ms_f16_mfma_plus_valu\": %.3f}\n",
           t0 * 1e-3 * 2.4e9 / mf, t1 * 1e-3 * 2.4e9 / mf, t1, t3, t2);
    return 0;
}
