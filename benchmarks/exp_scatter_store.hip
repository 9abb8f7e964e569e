// Scattered 4-byte stores on MI355X: time per store as a function of how many consecutive lanes write consecutive dwords
// (run length L) and of how far apart in TIME the writes to one 64-byte line are.  Behind DESIGN.md 4.5.76 (the one-pass tile
// sort's scatter: 20 M ids, every lane of a store instruction on a line of its own).
//   hipcc --offload-arch=gfx950 -O3 -o benchmarks/bin/exp_scatter_store benchmarks/exp_scatter_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

__global__ __launch_bounds__(256) void scatter(const uint32_t* __restrict__ pos, const int32_t* __restrict__ val, int64_t n, int32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) out[pos[i]] = val[i];
}

int main() {
    const int64_t n = 20 * 1000 * 1000;
    uint32_t* d_pos; int32_t *d_val, *d_out;
    hipMalloc(&d_pos, n * 4); hipMalloc(&d_val, n * 4); hipMalloc(&d_out, n * 4);
    std::vector<int32_t> val(n);
    for (int64_t i = 0; i < n; ++i) val[i] = (int32_t)i;
    hipMemcpy(d_val, val.data(), n * 4, hipMemcpyHostToDevice);
    std::mt19937_64 rng(1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: random permutation of L-dword blocks (a line's writes are far apart in time unless L >= 16)
        // mode 1: "tile lists": 8160 lists of equal length; the stream visits the lists in random order, L consecutive slots of
        //         a list per visit; consecutive visits of one list are ~8160 x L stores apart (the tile sort's pattern)
        for (int L : {1, 2, 4, 8, 16, 64}) {
            std::vector<uint32_t> pos(n);
            if (mode == 0) {
                const int64_t nb = n / L;
                std::vector<uint32_t> blk(nb);
                for (int64_t b = 0; b < nb; ++b) blk[b] = (uint32_t)b;
                std::shuffle(blk.begin(), blk.end(), rng);
                for (int64_t i = 0; i < nb * L; ++i) pos[i] = blk[i / L] * L + (uint32_t)(i % L);
                for (int64_t i = nb * L; i < n; ++i) pos[i] = (uint32_t)i;
            } else {
                const int T = 8160; const int64_t per = n / T;
                std::vector<uint32_t> fill(T, 0);
                int64_t i = 0;
                std::vector<int> order(T);
                for (int t = 0; t < T; ++t) order[t] = t;
                while (i < (int64_t)T * per) {
                    std::shuffle(order.begin(), order.end(), rng);
                    for (int t : order) {
                        for (int j = 0; j < L && fill[t] < per; ++j) pos[i++] = (uint32_t)(t * per + fill[t]++);
                    }
                }
                for (; i < n; ++i) pos[i] = (uint32_t)i;
            }
            hipMemcpy(d_pos, pos.data(), n * 4, hipMemcpyHostToDevice);
            for (int grid : {1024, 8192}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(scatter, dim3(grid), dim3(256), 0, 0, d_pos, d_val, n, d_out);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    best = std::min(best, ms);
                }
                printf("{\"mode\": %d, \"run\": %d, \"grid\": %d, \"us\": %.1f, \"Gstores_s\": %.1f}\n", mode, L, grid, best * 1e3, n / (best * 1e-3) * 1e-9);
            }
        }
    }
    return 0;
}
