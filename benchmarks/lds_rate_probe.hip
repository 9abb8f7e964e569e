// Probe (experiment): LDS read bandwidth per CU on gfx950 for ds_read_b32 / b64 / b128 (lane-consecutive, conflict-free)
// and for a b128 BROADCAST (all lanes of a half-wave read the same 16 bytes), at 4 and 8 waves per CU.
// Motivation: the K-pass field kernel issues 64 ds_read_b128 per MC-dropout pass and wave, and extra VALU / MFMA
// instructions per pass turned out to be free (benchmarks/exp_issue_model.sh) -- is it LDS-bound?
// build: hipcc -w --offload-arch=gfx950 -O3 -o lds_rate_probe lds_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ float4 lds[2560];   // 40 KiB
    for (int i = threadIdx.x; i < 2560; i += 256) lds[i] = make_float4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float4 acc = make_float4(0, 0, 0, 0);
    const char* base = reinterpret_cast<const char*>(lds);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int slab = ((i + k) & 31) * 1024;
            if (MODE == 0) {
                float4 v = *reinterpret_cast<const float4*>(base + slab + lane * 16);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            } else if (MODE == 1) {
                float2 v = *reinterpret_cast<const float2*>(base + slab + lane * 8);
                acc.x += v.x; acc.y += v.y;
            } else if (MODE == 2) {
                float v = *reinterpret_cast<const float*>(base + slab + lane * 4);
                acc.x += v;
            } else {
                float4 v = *reinterpret_cast<const float4*>(base + slab + (lane >> 5) * 16);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
static float run(float* out, int iters, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    const int iters = 20000;
    const int bytes[4] = {16, 8, 4, 16};
    const char* names[4] = {"ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_read_b128_broadcast"};
    for (int wpc = 4; wpc <= 8; wpc *= 2) {          // waves per CU = blocks per CU * 4
        const int blocks = 256 * (wpc / 4);
        float t[4] = {run<0>(out, iters, blocks), run<1>(out, iters, blocks), run<2>(out, iters, blocks), run<3>(out, iters, blocks)};
        printf("{\"waves_per_cu\": %d", wpc);
        for (int m = 0; m < 4; ++m) {
            const double insts_per_cu = (double)iters * 16 * wpc;
            const double ns_per_inst = t[m] * 1e6 / insts_per_cu;
            printf(", \"%s\": {\"ms\": %.3f, \"ns_per_wave_instruction_per_cu\": %.2f, \"bytes_per_ns_per_cu\": %.1f}", names[m], t[m],
                   ns_per_inst, 64.0 * bytes[m] / ns_per_inst);
        }
        printf("}\n");
    }
    return 0;
}
