// Depth sort of N splats (32-bit keys + 32-bit ids): rocprim's default config takes its merge-sort path up to
// 2^20 items (radix_sort_config<>::merge_sort_limit), which is where the 1 M-splat bench scene sits.  This times the
// default against a config that sends everything above 64 K items to Onesweep.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 benchmarks/exp_depth_sort.hip -o /tmp/ds && /tmp/ds
#include <cstring>
#include <cstdio>
#include <vector>
#include <cstdint>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

using cfg_onesweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 65536>;

template <class Cfg>
static float run(const uint32_t* kin, uint32_t* kout, const int32_t* vin, int32_t* vout, int n, int bits, std::vector<int32_t>& host) {
    size_t tmp = 0;
    rocprim::radix_sort_pairs<Cfg>(nullptr, tmp, kin, kout, vin, vout, (size_t)n, 0, bits, 0);
    void* ws; hipMalloc(&ws, tmp);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) rocprim::radix_sort_pairs<Cfg>(ws, tmp, kin, kout, vin, vout, (size_t)n, 0, bits, 0);
    hipEventRecord(a, 0);
    for (int i = 0; i < 50; ++i) rocprim::radix_sort_pairs<Cfg>(ws, tmp, kin, kout, vin, vout, (size_t)n, 0, bits, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    host.resize(n); hipMemcpy(host.data(), vout, n * 4, hipMemcpyDeviceToHost);
    hipFree(ws);
    return ms / 50 * 1000.f;
}

int main() {
    for (int n : {100000, 500000, 1000000, 1048576, 2000000, 4000000}) {
        std::vector<uint32_t> k(n); std::vector<int32_t> v(n);
        uint32_t s = 12345u;
        for (int i = 0; i < n; ++i) {
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            float d = 0.5f + (float)(s >> 8) * (1.f / 16777216.f) * 20.f;      // positive depths, many ties
            d = (float)(int)(d * 4096.f) / 4096.f;
            memcpy(&k[i], &d, 4); v[i] = i;
            if ((s & 15) == 0) k[i] = 0xFFFFFFFFu;                             // culled splats
        }
        uint32_t *kin, *kout; int32_t *vin, *vout;
        hipMalloc(&kin, n * 4); hipMalloc(&kout, n * 4); hipMalloc(&vin, n * 4); hipMalloc(&vout, n * 4);
        hipMemcpy(kin, k.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(vin, v.data(), n * 4, hipMemcpyHostToDevice);
        std::vector<int32_t> r0, r1, r2;
        float t0 = run<rocprim::default_config>(kin, kout, vin, vout, n, 32, r0);
        float t1 = run<cfg_onesweep>(kin, kout, vin, vout, n, 32, r1);
        float t2 = run<cfg_onesweep>(kin, kout, vin, vout, n, 31, r2);
        bool same = (r0 == r1);
        printf("{\"n\": %d, \"default_us\": %.1f, \"onesweep_us\": %.1f, \"onesweep_31bit_us\": %.1f, \"identical_order\": %s}\n", n, t0, t1, t2,
               same ? "true" : "false");
        hipFree(kin); hipFree(kout); hipFree(vin); hipFree(vout);
    }
    return 0;
}
