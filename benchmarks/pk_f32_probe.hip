// Probe (experiment): issue cost of the packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
// against their scalar forms on gfx950, at 1 / 2 / 4 waves per SIMD.  hipcc -O3 SLP-packs adjacent scalar fp32
// operations into the packed forms on its own (a third of the fp32 arithmetic of the proposal kernels); this tells
// whether a packed instruction is two operations in one 4-cycle issue slot (a lever for the VALU-issue-bound kernels) or
// holds the pipe as long as the two scalar ones.
// build + run: hipcc -w --offload-arch=gfx950 -O3 -o /tmp/pkp benchmarks/pk_f32_probe.hip && /tmp/pkp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

// KIND 0: 64 x v_fma_f32   1: 64 x v_pk_fma_f32   2: 64 x v_mul_f32   3: 64 x v_pk_mul_f32   4: v_add_f32  5: v_pk_add_f32
// (16 independent chains; per iteration 64 instructions of the kind)
template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float v[16];
    f2 w[16];
    for (int q = 0; q < 16; ++q) { v[q] = x + q; w[q] = (f2){x + q, x - q}; }
    f2 yy = {y, y}, xx = {x, x};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(y), "v"(x));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[q]) : "v"(yy), "v"(xx));
                if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[q]) : "v"(y));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[q]) : "v"(yy));
                if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[q]) : "v"(x));
                if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[q]) : "v"(xx));
            }
    }
    float r = 0.f;
    for (int q = 0; q < 16; ++q) r += v[q] + w[q][0] + w[q][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
static void run(const char* name, float* out) {
    const int iters = 20000;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = 256 * waves_per_simd;      // 256 CUs x (waves_per_simd) workgroups of 4 waves = that many per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(256), 0, 0, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // instructions issued per SIMD = waves_per_simd x iters x 64
        const double ns_per_inst = (double)ms * 1e6 / ((double)waves_per_simd * iters * 64.0);
        printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"ns_per_wave_inst_per_simd\": %.3f, \"cycles_at_2.4GHz\": %.2f}\n",
               name, waves_per_simd, ms, ns_per_inst, ns_per_inst * 2.4);
    }
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out);
    run<2>("v_mul_f32", out); run<3>("v_pk_mul_f32", out);
    run<4>("v_add_f32", out); run<5>("v_pk_add_f32", out);
    return 0;
}
