// Does gfx950 record an f32 -> f16 conversion overflow in the sticky TRAPSTS.EXCP bits (readable with s_getreg_b32 without
// any trap handler)?  If so, the single-product f16 field kernel can detect operand overflow with ONE scalar instruction
// per tile instead of a max-reduction over every converted activation.
//   hipcc --offload-arch=gfx950 -O2 -o benchmarks/build_probe/trapsts_probe benchmarks/trapsts_probe.hip && ./trapsts_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(const float* in, uint32_t* out) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x;
    uint32_t before, after_small, after_big, after_clear, after_exp, after_mfma;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(before));
    // in range
    float a = in[lane], b = in[lane + 64];
    f16x2 h = {(_Float16)a, (_Float16)b};
    uint32_t hv = __builtin_bit_cast(uint32_t, h);
    asm volatile("s_nop 4" ::"v"(hv));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after_small));
    // out of range in ONE lane only
    float c = in[128 + lane], d = in[192 + lane];
    f16x2 g = {(_Float16)c, (_Float16)d};
    uint32_t gv = __builtin_bit_cast(uint32_t, g);
    asm volatile("s_nop 4" ::"v"(gv));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after_big));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after_clear));
    float e = __expf(in[256 + lane]);   // exp(200) -> inf in one lane
    asm volatile("s_nop 4" ::"v"(e));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after_exp));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0");
    // an MFMA whose result overflows fp32: does the matrix pipe raise it?  (it should not matter either way)
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)60000.f; B[i] = (_Float16)60000.f; }
    f32x16 acc = {0};
    for (int i = 0; i < 16; ++i) acc[i] = 3e38f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    asm volatile("s_nop 7" ::"v"(acc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after_mfma));
    if (lane == 0) {
        out[0] = before; out[1] = after_small; out[2] = after_big; out[3] = after_clear; out[4] = after_exp; out[5] = after_mfma;
    }
    out[8 + lane] = hv ^ gv ^ __float_as_uint(e) ^ __float_as_uint(acc[0]);
}

int main() {
    float h[320];
    for (int i = 0; i < 320; ++i) h[i] = 1.5f + i * 0.25f;
    h[128 + 17] = 1.0e6f;     // one lane beyond 65504
    for (int i = 256; i < 320; ++i) h[i] = 1.0f;
    h[256 + 40] = 200.f;      // exp overflow in one lane
    float* din; uint32_t* dout;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 128 * 4);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout);
    uint32_t r[8];
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    printf("{\"trapsts_excp\": {\"cleared\": %u, \"after_in_range_cvt\": %u, \"after_overflowing_cvt\": %u, \"after_clear\": %u, "
           "\"after_exp_overflow\": %u, \"after_mfma_overflow\": %u}, \"bits\": \"0 invalid, 1 input denormal, 2 div0, 3 overflow, "
           "4 underflow, 5 inexact, 6 int div0\"}\n", r[0], r[1], r[2], r[3], r[4], r[5]);
    return 0;
}
