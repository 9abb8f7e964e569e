// Probe (round 6): what does one SIMD of gfx950 issue per cycle when v_mfma_f32_32x32x16_f16 and VALU
// instructions share a wave's stream?  Hand-written: every variant is ONE asm volatile statement that holds the
// register set-up, the s_memtime / s_memrealtime stamps, the loop and its 16 gaps -- the compiler schedules nothing
// (benchmarks/issue_sweep_probe.isa.txt is the listing of this file as committed).
//
//   gap      = one MFMA followed by NF independent "filler" VALU instructions (NF = 0 ... 12), 16 gaps per loop trip
//   chains   = 1: all 16 MFMAs accumulate into v[64:79];  2: even gaps into v[64:79], odd gaps into v[80:95]
//   filler   = v_fma_f32, v_add_f32, v_pk_fma_f32, v_pk_mul_f32, v_cvt_pk_f16_f32, v_pk_mad_u16, v_pk_sub_i16 clamp,
//              v_pk_max_i16, v_and_b32, v_max_i32, v_exp_f32 (eight destination registers in rotation, so no filler
//              depends on one closer than eight behind it; none touches an MFMA register)
//   waves    = 1 ... 4 per SIMD asked for: 256-thread workgroups (one wave per SIMD) with 160 / 64 / 32 / 16 KiB of dynamic LDS
//              and a grid of 256 x that many.  How many a CU really holds is MEASURED, not assumed: the MFMA-only variant
//              runs 32.3 cycles per MFMA per SIMD whatever the occupancy, so resident = (its cycles per gap per wave) / 32.3; every row carries that figure (`resident_waves_per_simd`) and the
//              cycles re-normalised with it.  (Asking for 3 gave 2 resident workgroups per CU and asking for 4 gave 3, at every
//              LDS size tried from 48 down to 16 KiB per workgroup and with 104 VGPRs per lane: whatever caps it, it is not the
//              LDS; the table therefore has columns for 1, 2, 2 and 3 waves per SIMD.)
//   "valu"   = the same 16 x NF fillers with no MFMA (the VALU-only price at that occupancy)
//
// Units: shader cycles from s_memtime (NOT an assumed clock); the clock itself = d(s_memtime) / d(s_memrealtime) x 100 MHz.
// Reported per variant: cycles per gap per SIMD = median over waves of d(s_memtime) / (iters x 16 x waves per SIMD).
// build: hipcc -w --offload-arch=gfx950 -O3 -o build_probe/issue_sweep_probe issue_sweep_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

// fixed registers: fillers v[32:47] (pairs for the packed-f32 forms), x = v[48:49], y = v[50:51], accumulators
// v[64:79] / v[80:95], A = v[96:99], B = v[100:103]
#define MF0 "v_mfma_f32_32x32x16_f16 v[64:79], v[96:99], v[100:103], v[64:79]\n"
#define MF1 "v_mfma_f32_32x32x16_f16 v[80:95], v[96:99], v[100:103], v[80:95]\n"
#define NOMF ""

#define T_FMA(d, p) "v_fma_f32 v" #d ", v" #d ", v50, v48\n"
#define T_ADD(d, p) "v_add_f32 v" #d ", v" #d ", v48\n"
#define T_PKFMA(d, p) "v_pk_fma_f32 v[" #p "], v[" #p "], v[50:51], v[48:49]\n"
#define T_PKMUL(d, p) "v_pk_mul_f32 v[" #p "], v[" #p "], v[50:51]\n"
#define T_CVT(d, p) "v_cvt_pk_f16_f32 v" #d ", v48, v50\n"
#define T_PKMAD(d, p) "v_pk_mad_u16 v" #d ", v" #d ", v52, v53\n"
#define T_PKSUB(d, p) "v_pk_sub_i16 v" #d ", v" #d ", v52 clamp\n"
#define T_PKMAX(d, p) "v_pk_max_i16 v" #d ", v" #d ", v52\n"
#define T_AND(d, p) "v_and_b32 v" #d ", v" #d ", v53\n"
#define T_MAXI(d, p) "v_max_i32 v" #d ", v" #d ", v52\n"
#define T_EXP(d, p) "v_exp_f32 v" #d ", v" #d "\n"
#define T_ASHR(d, p) "v_pk_ashrrev_i16 v" #d ", 15, v" #d "\n"

// NF fillers of type T: destinations v32, v34, ..., v46, then v33, v35, ... (pairs 32:33, 34:35, ... for packed f32)
#define F0(T) ""
#define F1(T) T(32, 32:33)
#define F2(T) F1(T) T(34, 34:35)
#define F3(T) F2(T) T(36, 36:37)
#define F4(T) F3(T) T(38, 38:39)
#define F5(T) F4(T) T(40, 40:41)
#define F6(T) F5(T) T(42, 42:43)
#define F7(T) F6(T) T(44, 44:45)
#define F8(T) F7(T) T(46, 46:47)
#define F10(T) F8(T) T(33, 32:33) T(35, 34:35)
#define F12(T) F10(T) T(37, 36:37) T(39, 38:39)

#define GAP(M, FILL) M FILL
#define BODY16(MA, MB, FILL) \
    GAP(MA, FILL) GAP(MB, FILL) GAP(MA, FILL) GAP(MB, FILL) GAP(MA, FILL) GAP(MB, FILL) GAP(MA, FILL) GAP(MB, FILL) \
    GAP(MA, FILL) GAP(MB, FILL) GAP(MA, FILL) GAP(MB, FILL) GAP(MA, FILL) GAP(MB, FILL) GAP(MA, FILL) GAP(MB, FILL)

#define CLOBBERS                                                                                                         \
    "memory", "scc", "vcc", "s20", "s21", "s22", "s23", "s24", "s26", "s27", "s28", "s29", "v32", "v33", "v34", "v35", "v36", \
        "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52",  \
        "v53", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78",  \
        "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94",  \
        "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103"

// set-up: fillers and MFMA operands hold small non-zero values (f16 0.0625 / 0.03125 pairs; accumulators start at 0 and
// stay finite for the trip counts used: 32000 MFMAs x 16 x 0.0625 x 0.03125 = 1000)
#define SETUP                                                                                                            \
    "v_cvt_f32_u32 v48, %[lane]\n v_mul_f32 v48, 0x3a83126f, v48\n v_mov_b32 v49, v48\n"                                  \
    "v_mov_b32 v50, 0x3f7fbe77\n v_mov_b32 v51, 0x3f7fbe77\n v_mov_b32 v52, 0x12345\n v_mov_b32 v53, 0x7fff7fff\n"          \
    "v_mov_b32 v32, v48\n v_mov_b32 v33, v48\n v_mov_b32 v34, v48\n v_mov_b32 v35, v48\n v_mov_b32 v36, v48\n"              \
    "v_mov_b32 v37, v48\n v_mov_b32 v38, v48\n v_mov_b32 v39, v48\n v_mov_b32 v40, v48\n v_mov_b32 v41, v48\n"              \
    "v_mov_b32 v42, v48\n v_mov_b32 v43, v48\n v_mov_b32 v44, v48\n v_mov_b32 v45, v48\n v_mov_b32 v46, v48\n"              \
    "v_mov_b32 v47, v48\n"                                                                                                 \
    "v_mov_b32 v96, 0x2c002c00\n v_mov_b32 v97, 0x2c002c00\n v_mov_b32 v98, 0x2c002c00\n v_mov_b32 v99, 0x2c002c00\n"      \
    "v_mov_b32 v100, 0x28002800\n v_mov_b32 v101, 0x28002800\n v_mov_b32 v102, 0x28002800\n v_mov_b32 v103, 0x28002800\n"  \
    "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n"    \
    "v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n v_mov_b32 v72, 0\n v_mov_b32 v73, 0\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0\n"    \
    "v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n v_mov_b32 v78, 0\n v_mov_b32 v79, 0\n v_mov_b32 v80, 0\n v_mov_b32 v81, 0\n"    \
    "v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v86, 0\n v_mov_b32 v87, 0\n"    \
    "v_mov_b32 v88, 0\n v_mov_b32 v89, 0\n v_mov_b32 v90, 0\n v_mov_b32 v91, 0\n v_mov_b32 v92, 0\n v_mov_b32 v93, 0\n"    \
    "v_mov_b32 v94, 0\n v_mov_b32 v95, 0\n s_nop 4\n"

#define DEFK(NAME, BODY)                                                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters) {                                     \
        uint32_t dt, drt, sink;                                                                                 \
        const uint32_t lane = threadIdx.x;                                                                      \
        asm volatile(SETUP                                                                                      \
                     "s_mov_b32 s24, %[iters]\n"                                                                \
                     "s_barrier\n"                                                                              \
                     "s_memtime s[20:21]\n s_memrealtime s[22:23]\n s_waitcnt lgkmcnt(0)\n"                     \
                     "L_loop_%=:\n" BODY                                                                        \
                     "s_sub_u32 s24, s24, 1\n s_cmp_lg_u32 s24, 0\n s_cbranch_scc1 L_loop_%=\n"                  \
                     "s_nop 15\n s_nop 15\n"                                                                    \
                     "v_add_f32 v32, v32, v64\n v_add_f32 v32, v32, v80\n"                                      \
                     "s_nop 0\n v_readfirstlane_b32 s24, v32\n"                                                 \
                     "s_memtime s[26:27]\n s_memrealtime s[28:29]\n s_waitcnt lgkmcnt(0)\n"                     \
                     "s_sub_u32 s26, s26, s20\n s_sub_u32 s28, s28, s22\n"                                      \
                     "v_mov_b32 %[dt], s26\n v_mov_b32 %[drt], s28\n"                                           \
                     "v_add_f32 %[sink], v32, v34\n v_add_f32 %[sink], %[sink], v36\n v_add_f32 %[sink], %[sink], v33\n" \
                     : [dt] "=v"(dt), [drt] "=v"(drt), [sink] "=v"(sink)                                        \
                     : [lane] "v"(lane), [iters] "s"(iters)                                                     \
                     : CLOBBERS);                                                                               \
        const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;                                                 \
        if ((threadIdx.x & 63) == 0) {                                                                          \
            out[3 * wave] = dt;                                                                                 \
            out[3 * wave + 1] = drt;                                                                            \
        }                                                                                                       \
        if (sink == 0x7fc12345u) out[3 * wave + 2] = sink; /* keeps the filler results live */                  \
    }

// the sweep: NF = 0..8, 10, 12 for the types the field kernels' pass loop is made of; 4 and 8 for the rest
#define SWEEP(TAG, T)                                   \
    DEFK(k_##TAG##_1c_0, BODY16(MF0, MF0, F0(T)))       \
    DEFK(k_##TAG##_1c_1, BODY16(MF0, MF0, F1(T)))       \
    DEFK(k_##TAG##_1c_2, BODY16(MF0, MF0, F2(T)))       \
    DEFK(k_##TAG##_1c_3, BODY16(MF0, MF0, F3(T)))       \
    DEFK(k_##TAG##_1c_4, BODY16(MF0, MF0, F4(T)))       \
    DEFK(k_##TAG##_1c_5, BODY16(MF0, MF0, F5(T)))       \
    DEFK(k_##TAG##_1c_6, BODY16(MF0, MF0, F6(T)))       \
    DEFK(k_##TAG##_1c_7, BODY16(MF0, MF0, F7(T)))       \
    DEFK(k_##TAG##_1c_8, BODY16(MF0, MF0, F8(T)))       \
    DEFK(k_##TAG##_1c_10, BODY16(MF0, MF0, F10(T)))     \
    DEFK(k_##TAG##_1c_12, BODY16(MF0, MF0, F12(T)))     \
    DEFK(k_##TAG##_2c_0, BODY16(MF0, MF1, F0(T)))       \
    DEFK(k_##TAG##_2c_2, BODY16(MF0, MF1, F2(T)))       \
    DEFK(k_##TAG##_2c_4, BODY16(MF0, MF1, F4(T)))       \
    DEFK(k_##TAG##_2c_6, BODY16(MF0, MF1, F6(T)))       \
    DEFK(k_##TAG##_2c_8, BODY16(MF0, MF1, F8(T)))       \
    DEFK(k_##TAG##_valu_8, BODY16(NOMF, NOMF, F8(T)))
#define SHORT(TAG, T)                                   \
    DEFK(k_##TAG##_1c_4, BODY16(MF0, MF0, F4(T)))       \
    DEFK(k_##TAG##_1c_8, BODY16(MF0, MF0, F8(T)))       \
    DEFK(k_##TAG##_valu_8, BODY16(NOMF, NOMF, F8(T)))

SWEEP(fma, T_FMA)
SWEEP(pkfma, T_PKFMA)
SWEEP(cvtpk, T_CVT)
SWEEP(pkmad16, T_PKMAD)
SWEEP(and32, T_AND)
SHORT(add, T_ADD)
SHORT(pkmul, T_PKMUL)
SHORT(pksub16, T_PKSUB)
SHORT(pkmax16, T_PKMAX)
SHORT(pkashr16, T_ASHR)
SHORT(maxi32, T_MAXI)
SHORT(exp, T_EXP)

typedef void (*kern_t)(uint32_t*, int);
struct Variant { const char* filler; const char* kind; int nf; kern_t fn; };
#define ROW(TAG, KIND, NF) {#TAG, #KIND, NF, k_##TAG##_##KIND##_##NF}
#define SWEEP_ROWS(TAG)                                                                                                     \
    ROW(TAG, 1c, 0), ROW(TAG, 1c, 1), ROW(TAG, 1c, 2), ROW(TAG, 1c, 3), ROW(TAG, 1c, 4), ROW(TAG, 1c, 5), ROW(TAG, 1c, 6),  \
        ROW(TAG, 1c, 7), ROW(TAG, 1c, 8), ROW(TAG, 1c, 10), ROW(TAG, 1c, 12), ROW(TAG, 2c, 0), ROW(TAG, 2c, 2),             \
        ROW(TAG, 2c, 4), ROW(TAG, 2c, 6), ROW(TAG, 2c, 8), ROW(TAG, valu, 8)
#define SHORT_ROWS(TAG) ROW(TAG, 1c, 4), ROW(TAG, 1c, 8), ROW(TAG, valu, 8)
static const Variant kVariants[] = {SWEEP_ROWS(fma),    SWEEP_ROWS(pkfma),   SWEEP_ROWS(cvtpk),   SWEEP_ROWS(pkmad16),
                                    SWEEP_ROWS(and32),  SHORT_ROWS(add),     SHORT_ROWS(pkmul),   SHORT_ROWS(pksub16),
                                    SHORT_ROWS(pkmax16), SHORT_ROWS(pkashr16), SHORT_ROWS(maxi32), SHORT_ROWS(exp)};

static double median(std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t* out;
    const size_t max_waves = (size_t)cus * 4 * 4;
    hipMalloc(&out, max_waves * 3 * sizeof(uint32_t));
    std::vector<uint32_t> host(max_waves * 3);
    for (const Variant& v : kVariants) hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // keep the chip busy for a moment first so that the clock has settled before the first variant
    for (int warm = 0; warm < 50; ++warm) hipLaunchKernelGGL(k_fma_2c_8, dim3(cus * 2), dim3(256), 64 * 1024, 0, out, iters);
    hipDeviceSynchronize();
    double resident[5] = {0, 1, 2, 3, 4};
    for (const Variant& v : kVariants) {
        for (int wps = 1; wps <= 4; ++wps) {
            const int blocks = cus * wps;
            static const int lds_kib[5] = {0, 160, 64, 32, 16};
            const size_t lds = (size_t)lds_kib[wps] * 1024;
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), lds, 0, out, 20);
            hipMemset(out, 0, max_waves * 3 * sizeof(uint32_t));
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), lds, 0, out, iters);
            if (hipDeviceSynchronize() != hipSuccess) { printf("{\"error\": \"%s\"}\n", hipGetErrorString(hipGetLastError())); return 1; }
            hipMemcpy(host.data(), out, (size_t)blocks * 4 * 3 * sizeof(uint32_t), hipMemcpyDeviceToHost);
            std::vector<double> cyc, clk;
            for (int w = 0; w < blocks * 4; ++w) {
                cyc.push_back((double)host[3 * w] / ((double)iters * 16.0 * wps));
                if (host[3 * w + 1]) clk.push_back((double)host[3 * w] / (double)host[3 * w + 1] * 0.1);
            }
            const double med = median(cyc);
            // the first variant of the table is MFMA-only (fma, 1 chain, 0 fillers): its time per MFMA per SIMD is the pipe's 32.3
            // cycles at every occupancy, so what it reads for `wps` asked waves tells how many were resident
            if (&v == &kVariants[0]) resident[wps] = med * wps / 32.26;
            printf("{\"filler\": \"%s\", \"kind\": \"%s\", \"fillers_per_gap\": %d, \"waves_per_simd_asked\": %d, "
                   "\"resident_waves_per_simd\": %.2f, \"cycles_per_gap_per_simd\": %.2f, \"clock_ghz\": %.3f}\n",
                   v.filler, v.kind, v.nf, wps, resident[wps], med * wps / resident[wps], clk.empty() ? 0.0 : median(clk));
        }
    }
    hipFree(out);
    return 0;
}
