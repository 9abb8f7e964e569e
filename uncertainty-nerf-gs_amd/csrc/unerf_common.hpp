// Shared host/device helpers for libunerf (gfx950 only).
// Built with -ffp-contract=off: every fused multiply-add in here is an explicit fmaf(),
// so the index/blend arithmetic that must be bit-exact against the CPU oracle
// (hash-grid corners + trilinear blend, splat projection, tile boxes, sort keys)
// rounds exactly like numpy/torch fp32 ops do.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "unerf.h"

// ---- host-side error plumbing -------------------------------------------------------
void unerf_set_error(const char* fmt, ...);
int unerf_check_launch(const char* what);

#define UNERF_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            unerf_set_error(__VA_ARGS__);   \
            return UNERF_ERR_ARG;           \
        }                                   \
    } while (0)

// ---- counter-based RNG (twin: oracle/nerf_oracle.py::_hash32 / mc_keep_mask) --------
#define UNERF_GOLDEN 0x9E3779B9u

__host__ __device__ __forceinline__ constexpr uint32_t unerf_hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21F0AAADu;
    x ^= x >> 15;
    x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}
__host__ __device__ __forceinline__ uint32_t unerf_mc_key(uint32_t seed, uint32_t pass) {
    return unerf_hash32(seed + pass * UNERF_GOLDEN);
}
__host__ __device__ __forceinline__ uint32_t unerf_mc_base(uint32_t key, uint32_t sample_idx) {
    return unerf_hash32(unerf_hash32(sample_idx) + key);
}
// MC-dropout mask words (round 5 definition; twin: oracle/nerf_oracle.py::mc_keep_mask).  Unit pair j (units 2j, 2j + 1)
// of stream s (0 = density trunk, 1 = last colour layer, 2 = second colour layer, 3 = the head's inputs) of one sample:
//   b_h    = hash32(hash32(sample) + key + h GOLDEN),  h = bit 1 of j       (unerf_mc_base_h: the full hash -- two
//            quarter-rate multiplies -- is paid ONCE per lane of the matrix kernels: a lane holds the pairs of one h)
//   pass 0 : w = (b_h & 0xFFFFFF) A(s, j & ~2) + (b_h >> 8) B(s, j & ~2)  mod 2^32, A / B odd 24-bit constants
//            (a multilinear hash of the two overlapping 24-bit windows of b_h: v_mul_u32_u24 + v_mad_u32_u24, both
//            full rate, the constants compile-time literals in the unrolled kernels)
//   pass k : each 16-bit half steps as the LCG x -> 25173 x + 13849 mod 2^16 (one v_pk_mad_u16 for the word)
// low half gates unit 2j, high half unit 2j+1: a unit is kept iff its half, read as a SIGNED 16-bit number, is below
// thr_s = round((1-p) 65536) - 32768 (the same event as "unsigned half ^ 0x8000 < round((1-p) 65536)").  The signed form
// lets the f16 kernels build the AND mask of a packed f16 pair in two packed instructions (saturating v_pk_sub_i16,
// v_pk_ashrrev_i16 15).
// Why this form: the K-pass kernel is VALU-issue-bound and rounds 1 - 4 spent 8 instructions per word on pass 0 (a full
// hash32 with two quarter-rate v_mul_lo_u32: ~56 issue cycles x 32 words per lane = one MC pass' worth of time per tile)
// plus a zero test, and two per word and pass on the step (rotate + shift-add); now 2 - 3 and 1.  Only <= K - 1 LCG
// steps are ever taken from a hashed start; tests/test_golden_cpu.py checks keep rate, independence between ALL pairs of
// K = 10 passes (same unit, either half), between the halves, between neighbouring words and between all pairs of units
// of a sample, and the Binomial(8, 1 - p) count of keeps, on 4 M units per pass.
__host__ __device__ __forceinline__ uint32_t unerf_mc_pre(uint32_t key, uint32_t sample_idx) {
    return unerf_hash32(sample_idx) + key;
}
__host__ __device__ __forceinline__ uint32_t unerf_mc_base_h(uint32_t pre, uint32_t h) {
    return unerf_hash32(pre + h * UNERF_GOLDEN);
}
__host__ __device__ __forceinline__ constexpr uint32_t unerf_mask_mul_a(uint32_t stream_id, uint32_t jc) {
    return (unerf_hash32(0xA5A50000u + 64u * stream_id + jc) & 0xFFFFFEu) | 1u;
}
__host__ __device__ __forceinline__ constexpr uint32_t unerf_mask_mul_b(uint32_t stream_id, uint32_t jc) {
    return (unerf_hash32(0x5A5A0000u + 64u * stream_id + jc) & 0xFFFFFEu) | 1u;
}
// b_h must be the base of h = (j >> 1) & 1
__host__ __device__ __forceinline__ uint32_t unerf_mask_word0(uint32_t b_h, uint32_t stream_id, uint32_t j) {
    const uint32_t jc = j & ~2u;
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(b_h, unerf_mask_mul_a(stream_id, jc)) + __umul24(b_h >> 8, unerf_mask_mul_b(stream_id, jc));
#else
    return (b_h & 0xFFFFFFu) * unerf_mask_mul_a(stream_id, jc) + (b_h >> 8) * unerf_mask_mul_b(stream_id, jc);
#endif
}
#define UNERF_MASK_LCG_A 25173u   // 0x6255 = 1 mod 4, increment odd: full period 2^16 per half
#define UNERF_MASK_LCG_C 13849u
__host__ __device__ __forceinline__ uint32_t unerf_mask_step(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short unerf_u16x2 __attribute__((ext_vector_type(2)));
    unerf_u16x2 v = __builtin_bit_cast(unerf_u16x2, x);
    const unerf_u16x2 a = {(unsigned short)UNERF_MASK_LCG_A, (unsigned short)UNERF_MASK_LCG_A};
    const unerf_u16x2 c = {(unsigned short)UNERF_MASK_LCG_C, (unsigned short)UNERF_MASK_LCG_C};
    v = v * a + c;   // v_pk_mad_u16
    return __builtin_bit_cast(uint32_t, v);
#else
    const uint32_t lo = ((x & 0xFFFFu) * UNERF_MASK_LCG_A + UNERF_MASK_LCG_C) & 0xFFFFu;
    const uint32_t hi = ((x >> 16) * UNERF_MASK_LCG_A + UNERF_MASK_LCG_C) & 0xFFFFu;
    return lo | (hi << 16);
#endif
}
// keep tests for one word on a scalar path: thr_hi = thr_s << 16 (a signed 32-bit number with zero low half).
// Written as 16-bit signed compares so that the compiler can select v_cmp_lt_i16 (low half) and its SDWA form
// reading WORD_1 (high half): one compare per unit, no shift / mask in front of it.
__host__ __device__ __forceinline__ bool unerf_keep_lo(uint32_t w, int32_t thr_hi) {
    return (int16_t)(uint16_t)w < (int16_t)(thr_hi >> 16);
}
__host__ __device__ __forceinline__ bool unerf_keep_hi(uint32_t w, int32_t thr_hi) {
    return (int16_t)(uint16_t)(w >> 16) < (int16_t)(thr_hi >> 16);
}

// ---- spacing (UniformLinDispPiecewiseSampler) ---------------------------------------
__host__ __device__ __forceinline__ float unerf_spacing_fn(float x) {
    return x < 1.f ? x / 2.f : 1.f - 1.f / (2.f * x);
}
__host__ __device__ __forceinline__ float unerf_spacing_inv(float x) {
    return x < 0.5f ? 2.f * x : 1.f / (2.f - 2.f * x);
}
// `lin` (uniform over a launch): UNERF_SPACING_UNIFORM -- nerfstudio's UniformSampler as the proposal sampler's initial
// sampler (proposal_initial_sampler="uniform", /root/reference/README.md:153): spacing_fn = its inverse = identity, so
// s_near / s_far are the planes themselves and a bin maps to b far + (1 - b) near.
__host__ __device__ __forceinline__ float unerf_spacing_of(float x, int lin) { return lin ? x : unerf_spacing_fn(x); }
__device__ __forceinline__ float unerf_s2e(float b, float s_near, float s_far, int lin) {
    const float t = b * s_far + (1.f - b) * s_near;
    return lin ? t : unerf_spacing_inv(t);
}

// exp for the sampler and compositing kernels (proposal densities, get_weights): the hardware exponential
// (v_exp_f32 after one multiply by log2 e; ~2 ulp, plus |x| 2^-24 relative from the scaled argument) instead of
// the ~15-instruction libm expf.  These kernels are VALU-issue bound and take two exponentials per sample; the
// arguments are -delta*sigma and -cumsum (results in [0,1], where both forms are limited by the rounding of a number
// near 1) or a density logit.  The exact-fp32 / VALU field kernels keep expf.
__device__ __forceinline__ float unerf_exp(float x) { return __expf(x); }

// torch.nan_to_num: NaN -> 0, +-inf -> +-FLT_MAX.  The clamp is one v_med3_f32 (three instructions in all instead of the
// six of three compare / select pairs; these sit in the per-sample loops of the PDF and composite kernels).
__device__ __forceinline__ float unerf_nan_to_num(float w) {
    const float c = __builtin_amdgcn_fmed3f(w, -FLT_MAX, FLT_MAX);
    return (w != w) ? 0.f : c;
}

// ---- SceneContraction(inf) -> (x+2)/4 -> selector mask ------------------------------
// `box` = NULL-like (use_aabb == 0): the contraction path above.  use_aabb: disable_scene_contraction
// (mcdropout_models.py:60-63 -> spatial_distortion = None): SceneBox.get_normalized_positions,
// (x - aabb_min) / (aabb_max - aabb_min), a division as upstream does it.
struct unerf_norm_box {
    int use_aabb;
    float lo[3], len[3];
};
__device__ __forceinline__ float unerf_normalize_position(float& x, float& y, float& z, const unerf_norm_box& box) {
    if (box.use_aabb) {  // uniform
        x = (x - box.lo[0]) / box.len[0];
        y = (y - box.lo[1]) / box.len[1];
        z = (z - box.lo[2]) / box.len[2];
    } else {
        float mag = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z));
        if (!(mag < 1.f)) {
            float s = 2.f - (1.f / mag);
            x = s * (x / mag);
            y = s * (y / mag);
            z = s * (z / mag);
        }
        x = (x + 2.f) / 4.f;
        y = (y + 2.f) / 4.f;
        z = (z + 2.f) / 4.f;
    }
    float sel = (x > 0.f && x < 1.f && y > 0.f && y < 1.f && z > 0.f && z < 1.f) ? 1.f : 0.f;
    x *= sel;
    y *= sel;
    z *= sel;
    return sel;
}

// ---- one level of the nerfstudio torch HashEncoding ---------------------------------
// corner order ccc,cfc,ffc,fcc,ccf,cff,fff,fcf; every level hashed; primes 1,2654435761,805459861.
// Returns BYTE offsets of the 8 corner rows inside the level (row index * 8, 8-byte fp32x2 rows):
// (hx ^ hy ^ hz) & mask, then << 3, equals ((hx<<3) ^ (hy<<3) ^ (hz<<3)) & (mask<<3), and
// (v * prime) << 3 == v * (prime << 3) mod 2^32, so the shift is folded into the constants and a
// corner costs one v_bitop3; the loads then take a uniform (SGPR) level base + this 32-bit VGPR
// offset, with no 64-bit address arithmetic per corner.  Needs log2T <= 28.
//
// EXACT_CEIL = false (the fused kernels): the "ceil" corner of an axis is taken as floor + 1 always.  It differs from
// ceil only where the scaled coordinate is an exact integer (masked samples sit at 0) -- and there the offset is 0, so
// the ceil-side row is multiplied by exactly 0 in the blend whichever row it is: the feature comes out the same (table
// entries are finite).  Per axis that is one truncating convert + one v_fract_f32 + adds instead of floor, ceil, three
// converts, a subtract, a compare and a select (positions are >= 0 here, so truncation is floor).  All of those are
// 4-cycle-class VALU instructions (profiles/r3_13_probe_valu_cost.jsonl): ~50 issue cycles less per level and sample.
// EXACT_CEIL = true: ceilf itself -- unerf_hashgrid_fwd, whose corner indices are an output and match the reference's.
template <bool WITH_BASE = false, bool EXACT_CEIL = false>
__device__ __forceinline__ void unerf_hash_corners(float px, float py, float pz, float scale, uint32_t mask,
                                                   uint32_t (&off)[8], float& ox, float& oy, float& oz,
                                                   uint32_t base = 0u) {
    float sx = px * scale, sy = py * scale, sz = pz * scale;
    const uint32_t P1 = 2654435761u << 3, P2 = 805459861u << 3, m8 = mask << 3;
    uint32_t hfx, hcx, hfy, hfz, hcy, hcz;
    if (EXACT_CEIL) {
        int cx = (int)ceilf(sx), cy = (int)ceilf(sy), cz = (int)ceilf(sz);
        int fx = (int)floorf(sx), fy = (int)floorf(sy), fz = (int)floorf(sz);
        ox = sx - (float)fx;
        oy = sy - (float)fy;
        oz = sz - (float)fz;
        hfx = (uint32_t)fx << 3; hcx = (uint32_t)cx << 3;
        // ceil = floor + 1 unless the coordinate is an exact integer, so the ceil products are the floor
        // products plus the prime (mod 2^32): two quarter-rate v_mul_lo_u32 per level instead of four
        hfy = (uint32_t)fy * P1; hfz = (uint32_t)fz * P2;
        hcy = hfy + (cy != fy ? P1 : 0u); hcz = hfz + (cz != fz ? P2 : 0u);
    } else {
        ox = __builtin_amdgcn_fractf(sx);     // = sx - floor(sx), exact for these magnitudes
        oy = __builtin_amdgcn_fractf(sy);
        oz = __builtin_amdgcn_fractf(sz);
        hfx = (uint32_t)(int)sx << 3; hcx = hfx + 8u;
        hfy = (uint32_t)(int)sy * P1; hfz = (uint32_t)(int)sz * P2;
        hcy = hfy + P1; hcz = hfz + P2;
    }
    if (!WITH_BASE) {  // 4 pair xors + one v_bitop3 ((t ^ z) & m8) per corner
        off[0] = (hcx ^ hcy ^ hcz) & m8;
        off[1] = (hcx ^ hfy ^ hcz) & m8;
        off[2] = (hfx ^ hfy ^ hcz) & m8;
        off[3] = (hfx ^ hcy ^ hcz) & m8;
        off[4] = (hcx ^ hcy ^ hfz) & m8;
        off[5] = (hcx ^ hfy ^ hfz) & m8;
        off[6] = (hfx ^ hfy ^ hfz) & m8;
        off[7] = (hfx ^ hcy ^ hfz) & m8;
        return;
    }
    // with a level offset to merge in (`base`: bits above the mask), mask the three terms once (6 ops,
    // base rides in on the z term); every corner is then a single three-input xor
    hfx &= m8; hcx &= m8; hfy &= m8; hcy &= m8;
    hfz = (hfz & m8) | base;
    hcz = (hcz & m8) | base;
    off[0] = hcx ^ hcy ^ hcz;
    off[1] = hcx ^ hfy ^ hcz;
    off[2] = hfx ^ hfy ^ hcz;
    off[3] = hfx ^ hcy ^ hcz;
    off[4] = hcx ^ hcy ^ hfz;
    off[5] = hcx ^ hfy ^ hfz;
    off[6] = hfx ^ hfy ^ hfz;
    off[7] = hfx ^ hcy ^ hfz;
}

// Trilinear blend, the reference's lerp order (x, then y, then z; a*o + b*(1-o)).  Both features of a
// row ride one packed fp32 pair (v_pk_mul_f32 / v_pk_add_f32 round each half exactly like the scalar
// op): 21 packed instructions per level.  Left as scalar code the SLP vectoriser paired (f0*ox, f3*mx)
// instead and spent a half-wasted horizontal v_pk_add plus ~25 register moves per level.
// FUSED (the proposal kernels, UNERF_PROP_BLEND_FMA): each lerp as fma(a, o, b * (1 - o)) -- one rounding fewer than torch's
// mul, mul, add, two instructions instead of three: 14 per level.  The proposal kernels are VALU-issue bound and the blend
// was 37 % of their instructions: first proposal pass 6.78 -> 6.49 ms per frame, second 3.12 -> 3.08
// (profiles/r4_exp_blend_fma_*.json).  unerf_hashgrid_fwd (whose outputs are the reference's bits) and the field kernels
// keep torch's order.  UNERF_FIELD_BLEND_FMA=1 builds the field kernels with the fused form too (-1.1 % K-pass, -1.7 %
// ACTIVE) -- NOT shippable: the split-f16 kernels of that build return different values from run to run in the columns
// 16..31 of a few tiles per launch (every other build and kernel is bit-reproducible, tests/test_gpu_repeatability.py);
// the cause was not found (not the inline assembly, not the MFMA register form, not the scheduler strategy: DESIGN.md 4.5).
#ifndef UNERF_PROP_BLEND_FMA
#define UNERF_PROP_BLEND_FMA 1
#endif
#ifndef UNERF_FIELD_BLEND_FMA
#define UNERF_FIELD_BLEND_FMA 0
#endif
typedef float unerf_v2f __attribute__((ext_vector_type(2)));
template <bool FUSED = false>
__device__ __forceinline__ unerf_v2f unerf_lerp2(unerf_v2f a, unerf_v2f b, float o, float m) {
#if UNERF_FIELD_BLEND_FMA == 7
    // experiment: the scalar weights as REAL register pairs, both halves written -- no packed operand
    // whose upper half is an undefined register the allocator may hand to another live value
    unerf_v2f oo = {o, o}, mm = {m, m};
    asm volatile("" : "+v"(oo), "+v"(mm));
    if (FUSED) return __builtin_elementwise_fma(a, oo, b * mm);
    return a * oo + b * mm;
#else
    if (FUSED) return __builtin_elementwise_fma(a, unerf_v2f{o, o}, b * m);
    return a * o + b * m;
#endif
}
template <bool FUSED = false>
__device__ __forceinline__ float unerf_lerp1(float a, float b, float o, float m) {
    if (FUSED) return __builtin_fmaf(a, o, b * m);
    return a * o + b * m;
}
template <bool FUSED = false>
__device__ __forceinline__ float2 unerf_blend8(const float2 (&f)[8], float ox, float oy, float oz) {
    const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
    unerf_v2f v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = unerf_v2f{f[k].x, f[k].y};
    unerf_v2f f03 = unerf_lerp2<FUSED>(v[0], v[3], ox, mx);
    unerf_v2f f12 = unerf_lerp2<FUSED>(v[1], v[2], ox, mx);
    unerf_v2f f56 = unerf_lerp2<FUSED>(v[5], v[6], ox, mx);
    unerf_v2f f47 = unerf_lerp2<FUSED>(v[4], v[7], ox, mx);
    unerf_v2f f0312 = unerf_lerp2<FUSED>(f03, f12, oy, my);
    unerf_v2f f4756 = unerf_lerp2<FUSED>(f47, f56, oy, my);
    unerf_v2f r = unerf_lerp2<FUSED>(f0312, f4756, oz, mz);
    return make_float2(r.x, r.y);
}

// Scalar statement of the same blend (identical roundings) for the MFMA field kernels, which are bound
// by the gather rather than by VALU issue: the packed form's aligned register pairs cost them ~18 VGPRs
// and with that the third wave per SIMD.
template <bool FUSED = false>
__device__ __forceinline__ float2 unerf_blend8_scalar(const float2 (&f)[8], float ox, float oy, float oz) {
    float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
    float2 r;
    {
        float f03 = unerf_lerp1<FUSED>(f[0].x, f[3].x, ox, mx);
        float f12 = unerf_lerp1<FUSED>(f[1].x, f[2].x, ox, mx);
        float f56 = unerf_lerp1<FUSED>(f[5].x, f[6].x, ox, mx);
        float f47 = unerf_lerp1<FUSED>(f[4].x, f[7].x, ox, mx);
        float f0312 = unerf_lerp1<FUSED>(f03, f12, oy, my);
        float f4756 = unerf_lerp1<FUSED>(f47, f56, oy, my);
        r.x = unerf_lerp1<FUSED>(f0312, f4756, oz, mz);
    }
    {
        float f03 = unerf_lerp1<FUSED>(f[0].y, f[3].y, ox, mx);
        float f12 = unerf_lerp1<FUSED>(f[1].y, f[2].y, ox, mx);
        float f56 = unerf_lerp1<FUSED>(f[5].y, f[6].y, ox, mx);
        float f47 = unerf_lerp1<FUSED>(f[4].y, f[7].y, ox, mx);
        float f0312 = unerf_lerp1<FUSED>(f03, f12, oy, my);
        float f4756 = unerf_lerp1<FUSED>(f47, f56, oy, my);
        r.y = unerf_lerp1<FUSED>(f0312, f4756, oz, mz);
    }
    return r;
}

// All 8 corner rows are requested back to back (one s_waitcnt for the batch).  Tried and rejected
// (r1, MI355X): fetching x-neighbour rows (idx, idx^1 for even floor(x)) as one 16-B load plus a
// predicated 8-B load for the odd case -- the divergent second load serialises into four dependent
// round trips per level and ran 1.6-2.1x slower; the neighbour row is an L1 hit anyway.
__device__ __forceinline__ void unerf_fetch_corners(const float2* __restrict__ lvl, const uint32_t (&off)[8],
                                                    float2 (&f)[8]) {
    const char* base = reinterpret_cast<const char*>(lvl);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = *reinterpret_cast<const float2*>(base + off[k]);
}

template <bool EXACT_CEIL = false, bool FUSED = false>
__device__ __forceinline__ float2 unerf_hash_level(const float2* __restrict__ lvl, float px, float py, float pz,
                                                   float scale, uint32_t mask) {
    uint32_t off[8];
    float ox, oy, oz;
    unerf_hash_corners<false, EXACT_CEIL>(px, py, pz, scale, mask, off, ox, oy, oz);
    float2 f[8];
    unerf_fetch_corners(lvl, off, f);
    return unerf_blend8<FUSED>(f, ox, oy, oz);
}

// Dense re-indexed level (see unerf_density_net in include/unerf.h): cell (x,y,z) holds
// { table[hash(x,y,z)], table[hash(x+1,y,z)] }, so the floor-x and ceil-x corners of one (y,z) edge
// arrive in a single 16-byte load.  Same values, same blend order as unerf_hash_level.
template <bool FUSED = false>
__device__ __forceinline__ float2 unerf_dense_level(const float4* __restrict__ cells, int dim, float px, float py,
                                                    float pz, float scale) {
    float sx = px * scale, sy = py * scale, sz = pz * scale;
    // floor + 1 for the "ceil" corners (see unerf_hash_corners): at an exact integer the ceil-side value is weighted 0
    const int fx = (int)sx, fy = (int)sy, fz = (int)sz;
    const int cy = fy + 1, cz = fz + 1;
    const float ox = __builtin_amdgcn_fractf(sx), oy = __builtin_amdgcn_fractf(sy), oz = __builtin_amdgcn_fractf(sz);
    // 32-bit byte offsets off a uniform base.  dim <= 640 (checked by the caller): every OPERAND below stays under
    // 2^24 (v_mad_u32_u24 multiplies the low 24 bits of its operands exactly into 32), and the cell index dim^3 under
    // 2^28, so its byte offset (x 16) fits 32 bits.
    const char* base = reinterpret_cast<const char*>(cells);
    const uint32_t udim = (uint32_t)dim;
    const uint32_t rcc = __umul24((uint32_t)cz, udim) + (uint32_t)cy, rfc = __umul24((uint32_t)cz, udim) + (uint32_t)fy;
    const uint32_t rcf = __umul24((uint32_t)fz, udim) + (uint32_t)cy, rff = __umul24((uint32_t)fz, udim) + (uint32_t)fy;
    const uint32_t ufx = (uint32_t)fx;
    const float4 pcc = *reinterpret_cast<const float4*>(base + ((__umul24(rcc, udim) + ufx) << 4));
    const float4 pfc = *reinterpret_cast<const float4*>(base + ((__umul24(rfc, udim) + ufx) << 4));
    const float4 pcf = *reinterpret_cast<const float4*>(base + ((__umul24(rcf, udim) + ufx) << 4));
    const float4 pff = *reinterpret_cast<const float4*>(base + ((__umul24(rff, udim) + ufx) << 4));
    float2 f[8];
    f[3] = make_float2(pcc.x, pcc.y);
    f[0] = make_float2(pcc.z, pcc.w);
    f[2] = make_float2(pfc.x, pfc.y);
    f[1] = make_float2(pfc.z, pfc.w);
    f[7] = make_float2(pcf.x, pcf.y);
    f[4] = make_float2(pcf.z, pcf.w);
    f[6] = make_float2(pff.x, pff.y);
    f[5] = make_float2(pff.z, pff.w);
    return unerf_blend8<FUSED>(f, ox, oy, oz);
}

// ---- one level of a tiny-cuda-nn HashGrid (include/unerf.h: unerf_tcnn_level) ----------------------
// rows[k] = absolute row (level offset included) of corner k; bit d of k steps +1 along dim d.
__device__ __forceinline__ void unerf_tcnn_corners(const unerf_tcnn_level& lv, float px, float py, float pz,
                                                   uint32_t (&rows)[8], float& wx, float& wy, float& wz) {
    const float fx = fmaf(lv.scale, px, 0.5f), fy = fmaf(lv.scale, py, 0.5f), fz = fmaf(lv.scale, pz, 0.5f);
    const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
    wx = fx - gx;
    wy = fy - gy;
    wz = fz - gz;
    const uint32_t x0 = (uint32_t)(int)gx, y0 = (uint32_t)(int)gy, z0 = (uint32_t)(int)gz;
    if (lv.dense) {
        // (x + y res + z res^2) mod size; the index stays below 2 size (size >= res^3, coordinates <= res).
        // One base index and the +res / +res^2 steps (24-bit multiplies: res < 4096) instead of two
        // quarter-rate 32-bit multiplies per corner; i - size wraps to a huge value when i < size.
        const uint32_t r = lv.res, r2 = __umul24(lv.res, lv.res);
        const uint32_t b00 = x0 + __umul24(y0, r) + __umul24(z0, r2);
        const uint32_t b[4] = {b00, b00 + r, b00 + r2, b00 + r + r2};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t i = b[k >> 1] + (uint32_t)(k & 1);
            i = min(i, i - lv.size);
            rows[k] = lv.offset + i;
        }
    } else {
        // hashed levels have size = 2^log2_hashmap_size
        const uint32_t m = lv.size - 1u;
        const uint32_t hy0 = y0 * 2654435761u, hz0 = z0 * 805459861u;
        const uint32_t hy1 = hy0 + 2654435761u, hz1 = hz0 + 805459861u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t hx = x0 + (k & 1), hy = (k & 2) ? hy1 : hy0, hz = (k & 4) ? hz1 : hz0;
            rows[k] = lv.offset + ((hx ^ hy ^ hz) & m);
        }
    }
}
// weight of corner k = (wx_k * wy_k) * wz_k (the product order of tcnn's loop over dims, starting from 1), the
// sum accumulated corner by corner with one fma per feature: four xy products shared by the two z layers, both
// features of a row on one packed fma.
__device__ __forceinline__ float2 unerf_tcnn_blend(const float2 (&f)[8], float wx, float wy, float wz) {
    const float mx = 1.f - wx, my = 1.f - wy, mz = 1.f - wz;
    const float wxy[4] = {mx * my, wx * my, mx * wy, wx * wy};
    unerf_v2f r = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float w = wxy[k & 3] * ((k & 4) ? wz : mz);
        r = __builtin_elementwise_fma(unerf_v2f{w, w}, unerf_v2f{f[k].x, f[k].y}, r);
    }
    return make_float2(r.x, r.y);
}
// tcnn's OWN arithmetic (kernel_grid of tiny-cuda-nn's encodings/grid.h with T = __half, the precision tcnn is built with
// on the GPUs the reference targets; twin: oracle tcnn_hash_encode_half): the table is the HALF copy of the fp32 master
// parameters, one 4-byte half2 row per corner; per corner the fp32 weight product above is rounded to half and
// result = fma((T)weight, row, result) runs as a half-precision fused multiply-add per feature (__hfma2 ->
// v_pk_fma_f16), result starting at zero, corners in index order.  -> the level's two features as one packed half2.
typedef _Float16 unerf_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t unerf_tcnn_blend_half(const uint32_t (&rows)[8], float wx, float wy, float wz) {
    const float mx = 1.f - wx, my = 1.f - wy, mz = 1.f - wz;
    const float wxy[4] = {mx * my, wx * my, mx * wy, wx * wy};
    unerf_h2 r = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        // two corner weights per v_cvt_pk_f16_f32 (round to nearest even, as __float2half_rn); each packed fma then
        // takes its weight from one half of the pair
        const unerf_v2f w2 = {wxy[k & 3] * ((k & 4) ? wz : mz), wxy[(k + 1) & 3] * ((k & 4) ? wz : mz)};
        const unerf_h2 wh = __builtin_convertvector(w2, unerf_h2);
        r = __builtin_elementwise_fma(unerf_h2{wh.x, wh.x}, __builtin_bit_cast(unerf_h2, rows[k]), r);
        r = __builtin_elementwise_fma(unerf_h2{wh.y, wh.y}, __builtin_bit_cast(unerf_h2, rows[k + 1]), r);
    }
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float2 unerf_h2_to_float2(uint32_t packed) {
    const unerf_h2 h = __builtin_bit_cast(unerf_h2, packed);
    return make_float2((float)h.x, (float)h.y);
}
// Corner BYTE offsets (level offset included) off the table base for the persistent field kernels:
// a uniform (SGPR) base + 32-bit lane offset per load instead of 64-bit address arithmetic per corner; the level
// record arrives as five scalars (staged in LDS by the caller).  Same rows as unerf_tcnn_corners.  Needs the
// table below 2^32 bytes and, on dense levels, res^2 * (res + 1) < 2^24.  RS = log2 of the row size in bytes:
// 3 (fp32 rows of two features), 2 (half2 rows); off_b = the level's byte offset (lv.offset << RS).
template <int RS = 3>
__device__ __forceinline__ void unerf_tcnn_offsets(float scale, uint32_t res, uint32_t off_b, uint32_t size,
                                                   uint32_t dense, float px, float py, float pz, uint32_t (&off)[8],
                                                   float& wx, float& wy, float& wz) {
    const float fx = fmaf(scale, px, 0.5f), fy = fmaf(scale, py, 0.5f), fz = fmaf(scale, pz, 0.5f);
    const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
    wx = fx - gx;
    wy = fy - gy;
    wz = fz - gz;
    const uint32_t x0 = (uint32_t)(int)gx, y0 = (uint32_t)(int)gy, z0 = (uint32_t)(int)gz;
    constexpr uint32_t ROW = 1u << RS;
    if (dense) {
        const uint32_t r2 = __umul24(res, res);
        const uint32_t b00 = x0 + __umul24(y0, res) + __umul24(z0, r2);
        if (__any(b00 + res + r2 + 1u >= size)) {  // some lane's cell straddles the wrap at `size` (grid boundary only)
            const uint32_t b[4] = {b00, b00 + res, b00 + r2, b00 + res + r2};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t i = b[k >> 1] + (uint32_t)(k & 1);
                i = min(i, i - size);   // i >= size ? i - size : i  (the difference wraps to a huge value when i < size)
                off[k] = off_b + (i << RS);
            }
        } else {  // one base offset + uniform steps (+ROW rides in the load's immediate offset)
            const uint32_t o00 = off_b + (b00 << RS), r8 = res << RS, r28 = r2 << RS;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                off[k] = o00 + ((k & 1) ? ROW : 0u) + ((k & 2) ? r8 : 0u) + ((k & 4) ? r28 : 0u);
        }
    } else {
        const uint32_t m8 = (size - 1u) << RS, P1 = 2654435761u << RS, P2 = 805459861u << RS;
        const uint32_t hx0 = x0 << RS, hx1 = hx0 + ROW;
        const uint32_t hy0 = y0 * P1, hz0 = z0 * P2, hy1 = hy0 + P1, hz1 = hz0 + P2;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t hx = (k & 1) ? hx1 : hx0, hy = (k & 2) ? hy1 : hy0, hz = (k & 4) ? hz1 : hz0;
            off[k] = off_b + ((hx ^ hy ^ hz) & m8);
        }
    }
}
__device__ __forceinline__ float2 unerf_tcnn_level_feat(const float2* __restrict__ params, const unerf_tcnn_level& lv,
                                                        float px, float py, float pz) {
    // uniform base + 32-bit byte offsets (the table is below 2^32 bytes): no 64-bit address arithmetic per corner
    const char* base = reinterpret_cast<const char*>(params);
    float2 f[8];
    float wx, wy, wz;
    if (lv.dense) {  // uniform here (every lane evaluates the same level)
        const float fx = fmaf(lv.scale, px, 0.5f), fy = fmaf(lv.scale, py, 0.5f), fz = fmaf(lv.scale, pz, 0.5f);
        const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
        const uint32_t x0 = (uint32_t)(int)gx, y0 = (uint32_t)(int)gy, z0 = (uint32_t)(int)gz;
        const uint32_t r2 = __umul24(lv.res, lv.res);
        const uint32_t b00 = x0 + __umul24(y0, lv.res) + __umul24(z0, r2);
        if (!__any(b00 + lv.res + r2 + 1u >= lv.size)) {
            // a dense tcnn level keeps x-neighbours in adjacent rows: the (x0, x0+1) corners of a (y, z) edge are 16
            // contiguous bytes (8-byte aligned) -- four loads per level instead of eight
            struct __attribute__((packed, aligned(8))) Pair { float a, b, c, d; };
            const uint32_t o00 = (lv.offset + b00) << 3, r8 = lv.res << 3, r28 = r2 << 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const Pair v = *reinterpret_cast<const Pair*>(base + (o00 + ((e & 1) ? r8 : 0u) + ((e & 2) ? r28 : 0u)));
                f[2 * e] = make_float2(v.a, v.b);
                f[2 * e + 1] = make_float2(v.c, v.d);
            }
            return unerf_tcnn_blend(f, fx - gx, fy - gy, fz - gz);
        }
    }
    uint32_t off[8];
    unerf_tcnn_offsets<3>(lv.scale, lv.res, lv.offset << 3, lv.size, lv.dense, px, py, pz, off, wx, wy, wz);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = *reinterpret_cast<const float2*>(base + off[k]);
    return unerf_tcnn_blend(f, wx, wy, wz);
}
// the same level of a HALF table (4-byte half2 rows), tcnn's half arithmetic -> the two features as floats holding
// the half values.  Dense levels away from the wrap: the x-neighbours of an edge are one 8-byte load (4-byte aligned).
__device__ __forceinline__ float2 unerf_tcnn_level_feat_half(const void* __restrict__ params, const unerf_tcnn_level& lv,
                                                             float px, float py, float pz) {
    const char* base = reinterpret_cast<const char*>(params);
    uint32_t rows[8];
    float wx, wy, wz;
    if (lv.dense) {
        const float fx = fmaf(lv.scale, px, 0.5f), fy = fmaf(lv.scale, py, 0.5f), fz = fmaf(lv.scale, pz, 0.5f);
        const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
        const uint32_t x0 = (uint32_t)(int)gx, y0 = (uint32_t)(int)gy, z0 = (uint32_t)(int)gz;
        const uint32_t r2 = __umul24(lv.res, lv.res);
        const uint32_t b00 = x0 + __umul24(y0, lv.res) + __umul24(z0, r2);
        if (!__any(b00 + lv.res + r2 + 1u >= lv.size)) {
            struct __attribute__((packed, aligned(4))) Pair { uint32_t a, b; };
            const uint32_t o00 = (lv.offset + b00) << 2, r4 = lv.res << 2, r24 = r2 << 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const Pair v = *reinterpret_cast<const Pair*>(base + (o00 + ((e & 1) ? r4 : 0u) + ((e & 2) ? r24 : 0u)));
                rows[2 * e] = v.a;
                rows[2 * e + 1] = v.b;
            }
            return unerf_h2_to_float2(unerf_tcnn_blend_half(rows, fx - gx, fy - gy, fz - gz));
        }
    }
    uint32_t off[8];
    unerf_tcnn_offsets<2>(lv.scale, lv.res, lv.offset << 2, lv.size, lv.dense, px, py, pz, off, wx, wy, wz);
#pragma unroll
    for (int k = 0; k < 8; ++k) rows[k] = *reinterpret_cast<const uint32_t*>(base + off[k]);
    return unerf_h2_to_float2(unerf_tcnn_blend_half(rows, wx, wy, wz));
}

// ---- real SH, 4 levels (components_from_spherical_harmonics) -------------------------
__device__ __forceinline__ void unerf_sh16(float x, float y, float z, float (&c)[16]) {
    float xx = x * x, yy = y * y, zz = z * z;
    c[0] = 0.28209479177387814f;
    c[1] = 0.4886025119029199f * y;
    c[2] = 0.4886025119029199f * z;
    c[3] = 0.4886025119029199f * x;
    c[4] = 1.0925484305920792f * x * y;
    c[5] = 1.0925484305920792f * y * z;
    c[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
    c[7] = 1.0925484305920792f * x * z;
    c[8] = 0.5462742152960396f * (xx - yy);
    c[9] = 0.5900435899266435f * y * (3.f * xx - yy);
    c[10] = 2.890611442640554f * x * y * z;
    c[11] = 0.4570457994644658f * y * (5.f * zz - 1.f);
    c[12] = 0.3731763325901154f * z * (5.f * zz - 3.f);
    c[13] = 0.4570457994644658f * x * (5.f * zz - 1.f);
    c[14] = 1.445305721320277f * z * (xx - yy);
    c[15] = 0.5900435899266435f * x * (xx - 3.f * yy);
}

__device__ __forceinline__ float unerf_softplus(float x) {
    // torch.nn.Softplus(beta=1, threshold=20)
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float unerf_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
