#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// hi = (f16(x0), f16(x1)); lo = (f16(x0 - hi.x), f16(x1 - hi.y)), residuals formed exactly inside the mixed fma
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    f16x2 h = {(_Float16)x0, (_Float16)x1};
    hi = __builtin_bit_cast(uint32_t, h);
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(x1));
    lo = l;
}
__global__ void k(float* out, const float* in) {
    uint32_t hi[4], lo[4];
    for (int p = 0; p < 4; ++p) split2(in[threadIdx.x * 8 + 2 * p], in[threadIdx.x * 8 + 2 * p + 1], hi[p], lo[p]);
    for (int p = 0; p < 4; ++p) {
        f16x2 h = __builtin_bit_cast(f16x2, hi[p]), l = __builtin_bit_cast(f16x2, lo[p]);
        out[threadIdx.x * 16 + 4 * p + 0] = (float)h.x; out[threadIdx.x * 16 + 4 * p + 1] = (float)h.y;
        out[threadIdx.x * 16 + 4 * p + 2] = (float)l.x; out[threadIdx.x * 16 + 4 * p + 3] = (float)l.y;
    }
}
int main() {
    float hin[8] = {1.2345678f, -0.000123456f, 3.14159265f, 1000.123f, 1e-5f, -7.654321f, 0.333333343f, 65000.f};
    float *din, *dout; hipMalloc(&din, 64 * 8 * 4); hipMalloc(&dout, 64 * 16 * 4);
    hipMemset(din, 0, 64 * 8 * 4); hipMemcpy(din, hin, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, din);
    float ho[16]; hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
    for (int p = 0; p < 4; ++p) {
        for (int e = 0; e < 2; ++e) {
            float x = hin[2 * p + e], h = ho[4 * p + e], l = ho[4 * p + 2 + e];
            printf("x=%.9g hi=%.9g lo=%.9g  x-(hi+lo)=%.3g  rel=%.3g\n", x, h, l, (double)x - ((double)h + l), ((double)x - ((double)h + l)) / x);
        }
    }
    return 0;
}
