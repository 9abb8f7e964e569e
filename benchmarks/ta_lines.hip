// Micro-benchmark (experiment, not product code): how does the cost of a divergent 8-byte gather
// instruction on gfx950 depend on the number of DISTINCT 128-byte lines its 64 lanes touch?
//   share=1 : every lane its own random row            (64 lines / instruction)
//   share=2 : lanes 2p, 2p+1 read rows r, r^1          (32 lines)
//   share=4 : quads read rows r^0..3                   (16 lines)
//   share=8 : 8 lanes read rows r^0..7                 ( 8 lines)
// Table: 16 levels x 2^19 rows x 8 B (64 MiB, Infinity-Cache resident like the field's hash grid).
// build: hipcc --offload-arch=gfx950 -O3 -o ta_lines ta_lines.hip ; run: ./ta_lines
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ uint32_t h32(uint32_t x) {
    x ^= x >> 16; x *= 0x21F0AAADu; x ^= x >> 15; x *= 0x735A2D97u; x ^= x >> 15;
    return x;
}

template <int SHARE, int WIDE>
__global__ __launch_bounds__(256) void gather(const float2* __restrict__ table, float* out, int iters) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t grp = tid / SHARE, sub = lane % SHARE;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            uint32_t r = h32(grp * 977u + it * 16u + k) & ((1u << 19) - 1u);
            if (WIDE) r &= ~1u;  // 16-byte aligned pair, one load
            r ^= WIDE ? (sub << 1) : sub;
            const char* base = reinterpret_cast<const char*>(table) + ((size_t)k << 22);
            if (WIDE) {
                float4 q = *reinterpret_cast<const float4*>(base + (r << 3));
                v[k] = make_float2(q.x + q.z, q.y + q.w);
            } else {
                v[k] = *reinterpret_cast<const float2*>(base + (r << 3));
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k].x + v[k].y;
    }
    out[tid] = acc;
}

template <int SHARE, int WIDE>
static void run(const float2* table, float* out, const char* name) {
    const int blocks = 256 * 12, iters = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((gather<SHARE, WIDE>), dim3(blocks), dim3(256), 0, 0, table, out, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((gather<SHARE, WIDE>), dim3(blocks), dim3(256), 0, 0, table, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instrs = (double)blocks * 4 * iters * 16;
    const double clk_per_instr_per_cu = ms * 1e-3 * 2.4e9 / (wave_instrs / 256.0);
    printf("{\"variant\": \"%s\", \"ms\": %.3f, \"load_instr_per_cu\": %.0f, \"clk_per_load_instr\": %.1f}\n", name, ms,
           wave_instrs / 256.0, clk_per_instr_per_cu);
}

int main() {
    const size_t rows = (size_t)16 << 19;
    float2* table; float* out;
    hipMalloc(&table, rows * sizeof(float2) + 4096);
    hipMalloc(&out, (size_t)256 * 12 * 256 * sizeof(float));
    std::vector<float2> h(rows);
    for (size_t i = 0; i < rows; ++i) h[i] = make_float2((float)(i & 255) * 1e-3f, 1.f);
    hipMemcpy(table, h.data(), rows * sizeof(float2), hipMemcpyHostToDevice);
    run<1, 0>(table, out, "8B x 64 lines");
    run<2, 0>(table, out, "8B x 32 lines (lane pairs share)");
    run<4, 0>(table, out, "8B x 16 lines (quads share)");
    run<8, 0>(table, out, "8B x  8 lines (octets share)");
    run<1, 1>(table, out, "16B x 64 lines");
    run<2, 1>(table, out, "16B x 32 lines");
    return 0;
}
