// NeRF half of libunerf: ray generation, hash grid, proposal density, weights+PDF resampling,
// fused main field (active / mc-dropout / laplace heads), composite with variance, moments.
// gfx950 only; wave = 64.  See include/unerf.h for the contract of each entry point.
#include "unerf_common.hpp"

#include <mutex>
#include <unordered_map>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

// ======================================================================================
// host plumbing
// ======================================================================================
static thread_local char g_err[512] = "";

void unerf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int unerf_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        unerf_set_error("%s: %s", what, hipGetErrorString(e));
        return UNERF_ERR_HIP;
    }
    return UNERF_OK;
}
// ---- build switches measured against each other on one box (benchmarks/exp_kpass_variants.sh,
// profiles/r2_exp_kpass_variants.json); the defaults are what ships.  unerf_build_flags() reports them, and the host
// packs the operands to match (ops.pack_field_mfma16: fold_trunk).
#ifndef UNERF_TRUNK_FOLD
#define UNERF_TRUNK_FOLD 1       // K-pass kernel, 16-row trunk-out layer: two MFMAs per k-step (-1 % kernel time)
#endif
#ifndef UNERF_LAP_EXP2
#define UNERF_LAP_EXP2 1         // LAPLACE: lap16_blob rows pre-scaled by +-log2(e), bare exp2 in the epilogue
#endif
#ifndef UNERF_RGB_SCALAR
#define UNERF_RGB_SCALAR 2   // the split-f16 kernels' fp32 colour layer as scalar fmas instead of v_pk_fma_f32: 1 every mode, 2 ACTIVE only
#endif
#ifndef UNERF_LAP_SCALAR_MOMENTS
#define UNERF_LAP_SCALAR_MOMENTS 1   // 1: no packed-fp32 fma in the Laplace heads' moment sums (-1.4 % Laplace field kernel, same box)
#endif
#ifndef UNERF_LAP_NO_FENCE
#define UNERF_LAP_NO_FENCE 0         // 1 (with scalar moments only: no inline assembly left in the heads): no scheduling fences
#endif
#if UNERF_LAP_NO_FENCE && UNERF_LAP_SCALAR_MOMENTS
#define LAP_FENCE() ((void)0)
#else
#define LAP_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef UNERF_KPASS_FILL
#define UNERF_KPASS_FILL 0   // 1: "f16" K-pass kernel with the mask arithmetic placed behind the MFMAs of a pass (round 6)
#endif
#ifndef UNERF_FIELD_BLEND_SCALAR
#define UNERF_FIELD_BLEND_SCALAR 0   // 1: no packed-fp32 instruction in the matrix kernels' grid blend (same roundings)
#endif
#ifndef UNERF_TRUNK_RESIDENT
#define UNERF_TRUNK_RESIDENT 1   // ... and its operands kept in registers across the passes (-3.5 %)
#endif

extern "C" const char* unerf_last_error(void) { return g_err; }
extern "C" int unerf_build_flags(void) {
    return (UNERF_TRUNK_FOLD ? UNERF_BUILD_TRUNK_FOLD : 0) | (UNERF_LAP_EXP2 ? UNERF_BUILD_LAP_EXP2 : 0);
}
// 11xx: round-2 ABI (drop_sites, sample_major planes, aabb, ...); 1101: ray_box_bins / ray_planes_bins; 1102: build flags,
// folded trunk-out slabs; 12xx: round-3 ABI -- `spacing` (UNERF_SPACING_*) behind every near / far pair, `background`
// (UNERF_BG_*) on the composite / GGN entry points
extern "C" int unerf_version(void) { return UNERF_ABI_VERSION; }
extern "C" int unerf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

struct TcnnLevels {
    unerf_tcnn_level v[32];
};

static inline unerf_norm_box make_norm_box(int use_aabb, const float* aabb) {
    unerf_norm_box b;
    b.use_aabb = use_aabb ? 1 : 0;
    for (int c = 0; c < 3; ++c) {
        b.lo[c] = use_aabb ? aabb[c] : 0.f;
        b.len[c] = use_aabb ? aabb[3 + c] - aabb[c] : 1.f;
    }
    return b;
}
// RGBRenderer(background_color=...) at eval [UPSTREAM nerfstudio 1.1.0 renderers.RGBRenderer.combine_rgb / forward]:
// "last_sample": comp + rgb[..., -1, :] (1 - acc); "random": comp as it is (no blending at eval, "as if the background
// were black"); "white" / "black": comp + colour (1 - acc); then clamp to [0, 1].
template <typename Args>
static inline int unerf_set_background(Args& a, int background, const float* rgb_host, const char* what) {
    if (background != UNERF_BG_LAST_SAMPLE && background != UNERF_BG_NONE && background != UNERF_BG_COLOR) {
        unerf_set_error("%s: background=%d (expected UNERF_BG_LAST_SAMPLE / _NONE / _COLOR)", what, background);
        return UNERF_ERR_ARG;
    }
    if (background == UNERF_BG_COLOR && !rgb_host) {
        unerf_set_error("%s: UNERF_BG_COLOR needs background_rgb (3 host floats)", what);
        return UNERF_ERR_ARG;
    }
    a.bg_mode = background;
    for (int c = 0; c < 3; ++c) a.bg[c] = (background == UNERF_BG_COLOR) ? rgb_host[c] : 0.f;
    return UNERF_OK;
}

#define UNERF_REQUIRE_SPACING(sp) \
    UNERF_REQUIRE((sp) == UNERF_SPACING_PIECEWISE || (sp) == UNERF_SPACING_UNIFORM, "spacing=%d (expected UNERF_SPACING_*)", (int)(sp))
static inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// Division of an index < 2^31 by a launch-invariant divisor (samples per ray): one multiply-high and a
// shift instead of the ~30 issue slots of the generic 32-bit division (Granlund-Montgomery, N = 31:
// m = floor(2^(31+l) / d) + 1 with l = ceil(log2 d) satisfies 2^(31+l) < m d <= 2^(31+l) + 2^l).
struct FastDiv {
    uint32_t magic;  // 0: d == 1
    uint32_t shift;
    uint32_t d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{0u, 0u, d};
    if (d > 1u) {
        uint32_t l = 0;
        while ((1ull << l) < d) ++l;
        f.magic = (uint32_t)((1ull << (31 + l)) / d + 1ull);
        f.shift = l - 1;
    }
    return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t x, const FastDiv& f) {
    return f.magic ? (__umulhi(x, f.magic) >> f.shift) : x;
}

// ======================================================================================
// wave / group helpers
// ======================================================================================
// Cross-lane traffic on DPP (data-parallel primitives: the lane permutation rides on the consuming VALU
// instruction) instead of __shfl_* (which compile to ds_bpermute_b32 through the LDS crossbar plus address,
// compare and select instructions: ~5 VALU + 1 LDS op per scan step).  The PDF, composite and depth-draw
// kernels are VALU-bound and spend a fifth of their instructions in scans.  gfx9 DPP controls:
// row_shr:n = 0x110+n (within a 16-lane row, zero fill), row_bcast:15 = 0x142, row_bcast:31 = 0x143,
// wave_shr:1 = 0x138, quad_perm = 0x00-0xFF, row_half_mirror = 0x141, row_mirror = 0x140.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {  // lanes without a source (or masked-off rows) read 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_f_or(float v, float fill) {  // lanes without a source lane read `fill`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}

// sum over a WIDTH-lane group (16: one DPP row; 64: the wave), result in every lane
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(WIDTH == 16 || WIDTH == 64, "group width");
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);   // row_half_mirror
    v += dpp_f<0x140>(v);   // row_mirror: every lane of a row holds the row sum
    if (WIDTH == 64) {
        v += dpp_f<0x142, 0xA>(v);   // rows 1,3 += row sums of rows 0,2
        v += dpp_f<0x143, 0xC>(v);   // rows 2,3 += sum of rows 0,1  -> lane 63 holds the total
        v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    }
    return v;
}
template <int WIDTH>
__device__ __forceinline__ int group_sum_i(int v) {
    static_assert(WIDTH == 16 || WIDTH == 64, "group width");
    v += dpp_i<0xB1>(v);
    v += dpp_i<0x4E>(v);
    v += dpp_i<0x141>(v);
    v += dpp_i<0x140>(v);
    if (WIDTH == 64) {
        v += dpp_i<0x142, 0xA>(v);
        v += dpp_i<0x143, 0xC>(v);
        v = __builtin_amdgcn_readlane(v, 63);
    }
    return v;
}
// inclusive scan over a WIDTH-lane group (Hillis-Steele inside the 16-lane rows, then the row totals)
template <int WIDTH>
__device__ __forceinline__ float group_incl_scan(float v, int /*lane_in_group*/) {
    static_assert(WIDTH == 16 || WIDTH == 64, "group width");
    v += dpp_f<0x111>(v);
    v += dpp_f<0x112>(v);
    v += dpp_f<0x114>(v);
    v += dpp_f<0x118>(v);
    if (WIDTH == 64) {
        v += dpp_f<0x142, 0xA>(v);
        v += dpp_f<0x143, 0xC>(v);
    }
    return v;
}

// exclusive scan: the inclusive result shifted by one lane.  (NOT "inclusive - own": that loses
// the low bits of small prefixes behind a large element and turns inf - inf into NaN.)
template <int WIDTH>
__device__ __forceinline__ float group_excl_scan(float v, int lane_in_group) {
    const float incl = group_incl_scan<WIDTH>(v, lane_in_group);
    return WIDTH == 16 ? dpp_f<0x111>(incl) : dpp_f<0x138>(incl);   // row_shr:1 / wave_shr:1, zero fill
}

// ======================================================================================
// 1. rays
// ======================================================================================
struct RayGenArgs {
    float R[9];
    float T[3];
    float fx, fy, cx, cy;
    float k1, k2, k3, k4, p1, p2;   // OPENCV lens parameters, nerfstudio's order
    int distorted;                  // any of the six != 0 (and the camera type takes lens parameters)
    int camera_type;                // UNERF_CAMERA_*
    int H, W;
    int64_t start, count;
    float* o;
    float* d;
    float* pa;
};

// camera_utils.radial_and_tangential_undistort [UPSTREAM nerfstudio 1.1.0, after MultiNeRF]: Newton iterations from the
// distorted point on  f(x, y) = (x d + 2 p1 x y + p2 (r + 2 x^2) - xd,  y d + 2 p2 x y + p1 (r + 2 y^2) - yd),
// r = x^2 + y^2, d = 1 + r (k1 + r (k2 + r (k3 + r k4))); every product and sum in the order the torch expression
// evaluates it (fp32, nothing contracted), a step only where |det J| > eps.
__device__ __forceinline__ void raygen_undistort(const RayGenArgs& a, float& u, float& v) {
    const float xd = u, yd = v;
    float x = xd, y = yd;
#pragma unroll 1
    for (int it = 0; it < UNERF_UNDISTORT_ITERATIONS; ++it) {
        const float r = x * x + y * y;
        const float d = 1.0f + r * (a.k1 + r * (a.k2 + r * (a.k3 + r * a.k4)));
        const float fx = ((d * x + ((2.f * a.p1) * x) * y) + a.p2 * (r + (2.f * x) * x)) - xd;
        const float fy = ((d * y + ((2.f * a.p2) * x) * y) + a.p1 * (r + (2.f * y) * y)) - yd;
        const float d_r = a.k1 + r * (2.0f * a.k2 + r * (3.0f * a.k3 + (r * 4.0f) * a.k4));
        const float d_x = (2.0f * x) * d_r;
        const float d_y = (2.0f * y) * d_r;
        const float fx_x = ((d + d_x * x) + (2.0f * a.p1) * y) + (6.0f * a.p2) * x;
        const float fx_y = (d_y * x + (2.0f * a.p1) * x) + (2.0f * a.p2) * y;
        const float fy_x = (d_x * y + (2.0f * a.p2) * y) + (2.0f * a.p1) * x;
        const float fy_y = ((d + d_y * y) + (2.0f * a.p2) * x) + (6.0f * a.p1) * y;
        const float den = fy_x * fx_y - fx_x * fy_y;
        const float xn = fx * fy_y - fy * fx_y;
        const float yn = fy * fx_x - fx * fy_x;
        const bool ok = fabsf(den) > UNERF_UNDISTORT_EPS;
        x = x + (ok ? xn / den : 0.f);
        y = y + (ok ? yn / den : 0.f);
    }
    u = x;
    v = y;
}

__device__ __forceinline__ void raygen_dir(const RayGenArgs& a, float u, float v, float& dx, float& dy, float& dz) {
    if (a.distorted) raygen_undistort(a, u, v);
    // camera-frame direction (include/unerf.h: camera_type), uniform branch; PERSPECTIVE keeps the exact (u, v, -1)
    float cxd = u, cyd = v, czd = -1.f;
    if (a.camera_type == UNERF_CAMERA_FISHEYE) {
        float theta = sqrtf(u * u + v * v);
        theta = fminf(fmaxf(theta, 0.f), 3.14159265358979323846f);
        const float st = sinf(theta);
        cxd = u * st / theta;     // 0 / 0 at the exact principal point, as upstream
        cyd = v * st / theta;
        czd = -cosf(theta);
    } else if (a.camera_type == UNERF_CAMERA_EQUIRECTANGULAR) {
        const float theta = -3.14159265358979323846f * u, phi = 3.14159265358979323846f * (0.5f - v);
        cxd = -sinf(theta) * sinf(phi);
        cyd = cosf(phi);
        czd = -cosf(theta) * sinf(phi);
    } else if (a.camera_type == UNERF_CAMERA_ORTHOPHOTO) {
        cxd = 0.f;
        cyd = 0.f;
    }
    // sum_c dir[c] * R[r][c]; torch.sum over 3 elements left to right
    float x = (cxd * a.R[0] + cyd * a.R[1]) + czd * a.R[2];
    float y = (cxd * a.R[3] + cyd * a.R[4]) + czd * a.R[5];
    float z = (cxd * a.R[6] + cyd * a.R[7]) + czd * a.R[8];
    float n = fmaxf(sqrtf((x * x + y * y) + z * z), 1e-7f);
    dx = x / n;
    dy = y / n;
    dz = z / n;
}

__global__ __launch_bounds__(256) void raygen_kernel(RayGenArgs a) {
    int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= a.count) return;
    int64_t g = a.start + n;
    int i = (int)(g / a.W), j = (int)(g % a.W);
    float y = (float)i + 0.5f, x = (float)j + 0.5f;
    float u0 = (x - a.cx) / a.fx, v0 = -(y - a.cy) / a.fy;
    float u1 = (x - a.cx + 1.f) / a.fx;
    float v2 = -(y - a.cy + 1.f) / a.fy;
    float d0x, d0y, d0z, d1x, d1y, d1z, d2x, d2y, d2z;
    raygen_dir(a, u0, v0, d0x, d0y, d0z);
    if (a.camera_type == UNERF_CAMERA_ORTHOPHOTO) {   // the origin moves over the image plane: c2w (u, v, 0, 1)
        a.o[n * 3 + 0] = (u0 * a.R[0] + v0 * a.R[1]) + a.T[0];
        a.o[n * 3 + 1] = (u0 * a.R[3] + v0 * a.R[4]) + a.T[1];
        a.o[n * 3 + 2] = (u0 * a.R[6] + v0 * a.R[7]) + a.T[2];
    } else {
        a.o[n * 3 + 0] = a.T[0];
        a.o[n * 3 + 1] = a.T[1];
        a.o[n * 3 + 2] = a.T[2];
    }
    a.d[n * 3 + 0] = d0x;
    a.d[n * 3 + 1] = d0y;
    a.d[n * 3 + 2] = d0z;
    if (a.pa) {
        raygen_dir(a, u1, v0, d1x, d1y, d1z);
        raygen_dir(a, u0, v2, d2x, d2y, d2z);
        float ex = d0x - d1x, ey = d0y - d1y, ez = d0z - d1z;
        float fx_ = d0x - d2x, fy_ = d0y - d2y, fz_ = d0z - d2z;
        float dxn = sqrtf((ex * ex + ey * ey) + ez * ez);
        float dyn = sqrtf((fx_ * fx_ + fy_ * fy_) + fz_ * fz_);
        a.pa[n] = dxn * dyn;
    }
}

extern "C" int unerf_generate_rays(const float* c2w, float fx, float fy, float cx, float cy, const float* distortion,
                                   int camera_type, int H, int W, int64_t ray_start, int64_t count, float* origins,
                                   float* directions, float* pixel_area, void* stream) {
    UNERF_REQUIRE(c2w && (count == 0 || (origins && directions)), "generate_rays: null pointer");
    UNERF_REQUIRE(camera_type == UNERF_CAMERA_PERSPECTIVE || camera_type == UNERF_CAMERA_FISHEYE ||
                      camera_type == UNERF_CAMERA_EQUIRECTANGULAR || camera_type == UNERF_CAMERA_ORTHOPHOTO,
                  "generate_rays: camera_type %d is not built (PERSPECTIVE 1, FISHEYE 2, EQUIRECTANGULAR 3, ORTHOPHOTO 8)", camera_type);
    UNERF_REQUIRE(H > 0 && W > 0 && ray_start >= 0 && count >= 0 && ray_start + count <= (int64_t)H * W,
                  "generate_rays: ray range [%lld,+%lld) outside %dx%d", (long long)ray_start, (long long)count, H, W);
    if (count == 0) return UNERF_OK;
    RayGenArgs a;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) a.R[r * 3 + c] = c2w[r * 4 + c];
        a.T[r] = c2w[r * 4 + 3];
    }
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.H = H; a.W = W;
    a.k1 = a.k2 = a.k3 = a.k4 = a.p1 = a.p2 = 0.f;
    a.distorted = 0;
    a.camera_type = camera_type;
    if (distortion && camera_type != UNERF_CAMERA_EQUIRECTANGULAR) {   // upstream: "do not apply distortion for equirectangular images"
        for (int i = 0; i < 6; ++i) {
            UNERF_REQUIRE(std::isfinite(distortion[i]), "generate_rays: distortion[%d] is not finite", i);
            a.distorted |= distortion[i] != 0.f;
        }
        a.k1 = distortion[0]; a.k2 = distortion[1]; a.k3 = distortion[2]; a.k4 = distortion[3];
        a.p1 = distortion[4]; a.p2 = distortion[5];
    }
    a.start = ray_start; a.count = count; a.o = origins; a.d = directions; a.pa = pixel_area;
    hipLaunchKernelGGL(raygen_kernel, dim3(blocks_for(count, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return unerf_check_launch("generate_rays");
}

// ---- oriented crop box: per-ray planes folded into the first level's spacing bins ----
// Cameras.generate_rays(..., obb_box=box) sets RayBundle.nears / fars from the ray / box slab test and the collider then
// keeps them; the sampler maps its [0,1] spacing bins through each ray's own planes.  Every kernel after this one takes
// ONE pair of planes (near0, far0) per launch, so the per-ray planes are folded into the bins instead:
//   b' = (b s(far_r) + (1 - b) s(near_r) - s(near0)) / (s(far0) - s(near0))
// gives the same euclidean edges under (near0, far0), and the pdf resampling is affine in the bins, so every later
// level stays consistent without knowing about the box.
struct BoxBinsArgs {
    const float* o; const float* d; const float* row;   // rays, shared [n+1] spacing row
    float w2b[12];                                       // inverse([R|T]) as 3x4 row-major
    float half[3];
    float near0, far0;
    int lin;                                             // UNERF_SPACING_*
    int64_t R; int n;
    float* bins; float* nears; float* fars;              // [R,n+1], [R], [R] (planes may be null)
    const float* in_nears; const float* in_fars;         // given: the bundle's own planes, no box test
};

__global__ __launch_bounds__(256) void box_bins_kernel(BoxBinsArgs a) {
    int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const int lane = threadIdx.x & 63;
    float tmin = -INFINITY, tmax = INFINITY;
    if (a.in_nears) {   // the bundle's own planes: a.o / a.d are null here
        tmin = a.in_nears[r];
        tmax = a.in_fars[r];
    } else {
    const float ox = a.o[r * 3], oy = a.o[r * 3 + 1], oz = a.o[r * 3 + 2];
    const float dx = a.d[r * 3], dy = a.d[r * 3 + 1], dz = a.d[r * 3 + 2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* m = a.w2b + c * 4;
        float bo = ((m[0] * ox + m[1] * oy) + m[2] * oz) + m[3];
        float bd = (m[0] * dx + m[1] * dy) + m[2] * dz;
        float t0 = (-a.half[c] - bo) / bd, t1 = (a.half[c] - bo) / bd;
        tmin = fmaxf(tmin, fminf(t0, t1));
        tmax = fminf(tmax, fmaxf(t0, t1));
    }
    tmin = fminf(fmaxf(tmin, 0.f), 1e10f);
    tmax = fminf(fmaxf(tmax, 0.f), 1e10f);
    if (!(tmax > tmin)) tmin = tmax = 1e10f;             // a miss: both planes at the invalid value
    }
    if (lane == 0) {
        if (a.nears) a.nears[r] = tmin;
        if (a.fars) a.fars[r] = tmax;
    }
    // a miss upstream puts every sample at s_inv(s(1e10)) = s_inv(1.f) = inf, i.e. undefined pixels; here its samples
    // collapse onto the far plane (zero-length intervals -> zero weights, accumulation 0), which keeps everything finite
    const float sn = unerf_spacing_of(fminf(tmin, a.far0), a.lin), sf = unerf_spacing_of(fminf(tmax, a.far0), a.lin);
    const float sn0 = unerf_spacing_of(a.near0, a.lin), inv = 1.f / (unerf_spacing_of(a.far0, a.lin) - sn0);
    for (int i = lane; i <= a.n; i += 64) {
        float b = a.row[i];
        // sf == sn (a miss, or an empty interval): every edge the same float, so all later lerps b0 + t (b1 - b0) and
        // interval lengths are exactly degenerate
        a.bins[r * (a.n + 1) + i] = ((sf == sn ? sn : (b * sf + (1.f - b) * sn)) - sn0) * inv;
    }
}

extern "C" int unerf_ray_box_bins(const float* origins, const float* directions, int64_t R, const float* world_to_box,
                                  const float* half_extent, float near, float far, int spacing, const float* sbins_row,
                                  int n, float* sbins, float* nears, float* fars, void* stream) {
    UNERF_REQUIRE(world_to_box && half_extent && (R == 0 || (origins && directions && sbins_row && sbins)), "ray_box_bins: null pointer");
    UNERF_REQUIRE(R >= 0 && n >= 1 && far > near && near >= 0.f, "ray_box_bins: R=%lld n=%d near=%g far=%g",
                  (long long)R, n, (double)near, (double)far);
    if (R == 0) return UNERF_OK;
    BoxBinsArgs a;
    for (int i = 0; i < 12; ++i) a.w2b[i] = world_to_box[i];
    for (int i = 0; i < 3; ++i) a.half[i] = half_extent[i];
    a.o = origins; a.d = directions; a.row = sbins_row; a.near0 = near; a.far0 = far; UNERF_REQUIRE_SPACING(spacing); a.lin = spacing; a.R = R; a.n = n;
    a.bins = sbins; a.nears = nears; a.fars = fars; a.in_nears = a.in_fars = nullptr;
    hipLaunchKernelGGL(box_bins_kernel, dim3(blocks_for(R, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return unerf_check_launch("ray_box_bins");
}

extern "C" int unerf_ray_planes_bins(const float* nears, const float* fars, int64_t R, float near, float far,
                                     int spacing, const float* sbins_row, int n, float* sbins, void* stream) {
    UNERF_REQUIRE(R == 0 || (nears && fars && sbins_row && sbins), "ray_planes_bins: null pointer");
    UNERF_REQUIRE(R >= 0 && n >= 1 && far > near && near >= 0.f, "ray_planes_bins: R=%lld n=%d near=%g far=%g",
                  (long long)R, n, (double)near, (double)far);
    if (R == 0) return UNERF_OK;
    BoxBinsArgs a = {};
    a.row = sbins_row; a.near0 = near; a.far0 = far; UNERF_REQUIRE_SPACING(spacing); a.lin = spacing; a.R = R; a.n = n;
    a.bins = sbins; a.in_nears = nears; a.in_fars = fars;
    hipLaunchKernelGGL(box_bins_kernel, dim3(blocks_for(R, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return unerf_check_launch("ray_planes_bins");
}

// ======================================================================================
// 2. stand-alone hash grid (parity surface for the index bookkeeping)
// ======================================================================================
__global__ __launch_bounds__(256) void hashgrid_kernel(const float* __restrict__ xyz, const float* __restrict__ table,
                                                       const float* __restrict__ scalings, int64_t N, int L, int log2T,
                                                       float* __restrict__ out, int32_t* __restrict__ out_idx) {
    int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float px = xyz[n * 3 + 0], py = xyz[n * 3 + 1], pz = xyz[n * 3 + 2];
    const uint32_t mask = (1u << log2T) - 1u;
    for (int l = 0; l < L; ++l) {
        const float2* lvl = reinterpret_cast<const float2*>(table) + ((size_t)l << log2T);
        float2 f = unerf_hash_level<true>(lvl, px, py, pz, scalings[l], mask);   // the reference's ceil / floor, any sign
        out[n * (2 * L) + 2 * l + 0] = f.x;
        out[n * (2 * L) + 2 * l + 1] = f.y;
        if (out_idx) {
            uint32_t idx[8];
            float ox, oy, oz;
            unerf_hash_corners<false, true>(px, py, pz, scalings[l], mask, idx, ox, oy, oz);
#pragma unroll
            for (int k = 0; k < 8; ++k) out_idx[(n * L + l) * 8 + k] = (int32_t)((idx[k] >> 3) + ((uint32_t)l << log2T));
        }
    }
}

extern "C" int unerf_hashgrid_fwd(const float* xyz, const float* table, const float* scalings, int64_t N, int L,
                                  int log2T, float* out, int32_t* out_idx, void* stream) {
    UNERF_REQUIRE(L >= 1 && L <= 32 && log2T >= 1 && log2T <= 24 && N >= 0, "hashgrid_fwd: bad L=%d log2T=%d", L, log2T);
    if (N == 0) return UNERF_OK;  // empty input: nothing to read or write (pointers may be null)
    UNERF_REQUIRE(xyz && table && scalings && out, "hashgrid_fwd: null pointer");
    hipLaunchKernelGGL(hashgrid_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, xyz, table,
                       scalings, N, L, log2T, out, out_idx);
    return unerf_check_launch("hashgrid_fwd");
}

__global__ __launch_bounds__(256) void hashgrid_tcnn_kernel(const float* __restrict__ xyz,
                                                            const float* __restrict__ params, TcnnLevels lv, int64_t N,
                                                            int L, float* __restrict__ out, int32_t* __restrict__ out_idx) {
    int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float px = xyz[n * 3 + 0], py = xyz[n * 3 + 1], pz = xyz[n * 3 + 2];
    for (int l = 0; l < L; ++l) {
        uint32_t rows[8];
        float wx, wy, wz;
        unerf_tcnn_corners(lv.v[l], px, py, pz, rows, wx, wy, wz);
        float2 f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = reinterpret_cast<const float2*>(params)[rows[k]];
        float2 r = unerf_tcnn_blend(f, wx, wy, wz);
        out[n * (2 * L) + 2 * l + 0] = r.x;
        out[n * (2 * L) + 2 * l + 1] = r.y;
        if (out_idx) {
#pragma unroll
            for (int k = 0; k < 8; ++k) out_idx[(n * L + l) * 8 + k] = (int32_t)rows[k];
        }
    }
}

static int unerf_tcnn_levels_from_host(TcnnLevels& lv, const unerf_tcnn_level* levels_host, int L, const char* what) {
    for (int l = 0; l < L; ++l) {
        lv.v[l] = levels_host[l];
        UNERF_REQUIRE(lv.v[l].res >= 2 && lv.v[l].size >= 8 && (lv.v[l].dense || (lv.v[l].size & (lv.v[l].size - 1)) == 0),
                      "%s: level %d: res=%u size=%u (hashed levels need a power-of-two size)", what, l,
                      lv.v[l].res, lv.v[l].size);
    }
    return UNERF_OK;
}

extern "C" int unerf_hashgrid_fwd_tcnn(const float* xyz, const float* params, const unerf_tcnn_level* levels_host,
                                       int64_t N, int L, float* out, int32_t* out_idx, void* stream) {
    UNERF_REQUIRE(L >= 1 && L <= 32 && N >= 0, "hashgrid_fwd_tcnn: bad L=%d", L);
    if (N == 0) return UNERF_OK;
    UNERF_REQUIRE(xyz && params && levels_host && out, "hashgrid_fwd_tcnn: null pointer");
    TcnnLevels lv;
    if (int rc = unerf_tcnn_levels_from_host(lv, levels_host, L, "hashgrid_fwd_tcnn")) return rc;
    hipLaunchKernelGGL(hashgrid_tcnn_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, xyz, params,
                       lv, N, L, out, out_idx);
    return unerf_check_launch("hashgrid_fwd_tcnn");
}

// the same lookup in tcnn's own half arithmetic: params_half = the half copy of the parameter vector, [rows] half2
__global__ __launch_bounds__(256) void hashgrid_tcnn_half_kernel(const float* __restrict__ xyz,
                                                                 const void* __restrict__ params_half, TcnnLevels lv, int64_t N,
                                                                 int L, float* __restrict__ out) {
    int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float px = xyz[n * 3 + 0], py = xyz[n * 3 + 1], pz = xyz[n * 3 + 2];
    for (int l = 0; l < L; ++l) {
        uint32_t rows[8], c8[8];
        float wx, wy, wz;
        unerf_tcnn_corners(lv.v[l], px, py, pz, rows, wx, wy, wz);
#pragma unroll
        for (int k = 0; k < 8; ++k) c8[k] = reinterpret_cast<const uint32_t*>(params_half)[rows[k]];
        const float2 r = unerf_h2_to_float2(unerf_tcnn_blend_half(c8, wx, wy, wz));
        out[n * (2 * L) + 2 * l + 0] = r.x;
        out[n * (2 * L) + 2 * l + 1] = r.y;
    }
}

extern "C" int unerf_hashgrid_fwd_tcnn_half(const float* xyz, const void* params_half, const unerf_tcnn_level* levels_host,
                                            int64_t N, int L, float* out, void* stream) {
    UNERF_REQUIRE(L >= 1 && L <= 32 && N >= 0, "hashgrid_fwd_tcnn_half: bad L=%d", L);
    if (N == 0) return UNERF_OK;
    UNERF_REQUIRE(xyz && params_half && levels_host && out, "hashgrid_fwd_tcnn_half: null pointer");
    TcnnLevels lv;
    if (int rc = unerf_tcnn_levels_from_host(lv, levels_host, L, "hashgrid_fwd_tcnn_half")) return rc;
    hipLaunchKernelGGL(hashgrid_tcnn_half_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, xyz,
                       params_half, lv, N, L, out);
    return unerf_check_launch("hashgrid_fwd_tcnn_half");
}

// ======================================================================================
// 3. proposal density: positions -> contraction -> hash grid (L levels) -> MLP(2L->HID->1)
// ======================================================================================
struct PropArgs {
    const float* origins;
    const float* dirs;
    const float* sbins;
    int64_t sstride;
    int64_t R;
    int n;
    float s_near, s_far;
    int lin;   // UNERF_SPACING_*: 1 = identity spacing (UniformSampler)
    unerf_density_net net;
    float avg;
    float* out;
    FastDiv fd;  // by n; used when R * n < 2^31
    int small;
    // Pixel-patch schedule (image_width hint): a workgroup = an 8x8 pixel patch x 4 consecutive sample indices,
    // a wave = the 64 pixels of the patch at ONE sample index.  Neighbouring pixels at equal depth sit in the
    // same one or two grid cells, so a gather instruction touches a handful of cache lines instead of the
    // ~20 of 64 consecutive samples along one ray (the proposal kernels saturate the texture-address unit).
    uint32_t img_w, pcols, nsg, nblocks;   // img_w = 0: linear schedule; nsg = ceil(n / 4) sample groups; nblocks: workgroups of the patch schedule
    FastDiv div_nsg, div_pcols;
    int64_t g0, first_row;
    int vec4;   // prop_patch_kernel: n % 4 == 0 and density_out 16-byte aligned
    unerf_norm_box box;
};

template <int L, int HID>
__global__ __launch_bounds__(256) void prop_density_kernel(PropArgs a) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r;
    int i;
    if (a.img_w) {  // uniform
        const uint32_t patch = fastdiv(blockIdx.x, a.div_nsg), sg = blockIdx.x - patch * a.nsg;
        const uint32_t band = fastdiv(patch, a.div_pcols), pc = patch - band * a.pcols;
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t x = pc * 8u + (lane & 7u);
        const int64_t y = a.first_row + (int64_t)band * 8 + (lane >> 3);
        r = y * (int64_t)a.img_w + x - a.g0;
        i = (int)(sg * 4u + (threadIdx.x >> 6));
        if (x >= a.img_w || r < 0 || r >= a.R || i >= a.n) return;
        idx = r * a.n + i;
    } else if (idx >= a.R * a.n) {
        return;
    } else if (a.small) {  // uniform
        const uint32_t r32 = fastdiv((uint32_t)idx, a.fd);
        r = r32;
        i = (int)((uint32_t)idx - r32 * (uint32_t)a.n);
    } else {
        r = idx / a.n;
        i = (int)(idx - r * a.n);
    }
    const float* sb = a.sbins + r * a.sstride;
    float e0 = unerf_s2e(sb[i], a.s_near, a.s_far, a.lin);
    float e1 = unerf_s2e(sb[i + 1], a.s_near, a.s_far, a.lin);
    float t = e0 + e1;
    float px = a.origins[r * 3 + 0] + a.dirs[r * 3 + 0] * t / 2.f;
    float py = a.origins[r * 3 + 1] + a.dirs[r * 3 + 1] * t / 2.f;
    float pz = a.origins[r * 3 + 2] + a.dirs[r * 3 + 2] * t / 2.f;
    float sel = unerf_normalize_position(px, py, pz, a.box);
    const uint32_t mask = (1u << a.net.log2T) - 1u;
    float feat[2 * L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float2 f;
        if (a.net.tcnn_levels && a.net.grid_half) {  // uniform: tcnn layout, half2 rows, tcnn's half arithmetic
            f = unerf_tcnn_level_feat_half(a.net.table, a.net.tcnn_levels[l], px, py, pz);
        } else if (a.net.tcnn_levels) {  // uniform: tcnn-layout grid
            f = unerf_tcnn_level_feat(reinterpret_cast<const float2*>(a.net.table), a.net.tcnn_levels[l], px, py, pz);
        } else if (l < a.net.n_dense) {  // wave-uniform: coarse level with a dense, x-paired copy
            f = unerf_dense_level<(UNERF_PROP_BLEND_FMA != 0)>(reinterpret_cast<const float4*>(a.net.dense) + a.net.dense_off[l], a.net.dense_dim[l],
                                  px, py, pz, a.net.scalings[l]);
        } else {
            const float2* lvl = reinterpret_cast<const float2*>(a.net.table) + ((size_t)l << a.net.log2T);
            f = unerf_hash_level<false, (UNERF_PROP_BLEND_FMA != 0)>(lvl, px, py, pz, a.net.scalings[l], mask);
        }
        feat[2 * l] = f.x;
        feat[2 * l + 1] = f.y;
    }
    const float* __restrict__ w0t = a.net.w0t;
    const float* __restrict__ b0 = a.net.b0;
    const float* __restrict__ w1t = a.net.w1t;
    float o = a.net.b1[0];
    if constexpr (HID == 0) {
        // use_linear=True (HashMLPDensityField: `self.linear = nn.Linear(encoding.get_out_dim(), 1)` straight on the grid
        // features, no hidden layer): w1t = that layer's 2L weights
#pragma unroll
        for (int k = 0; k < 2 * L; ++k) o = fmaf(feat[k], w1t[k], o);
    } else {
    // two hidden units per v_pk_fma_f32 (each half is an IEEE fma, same bits as fmaf): the kernel is
    // VALU-issue-bound, and the MLP was a third of its instructions.  Weights arrive as SGPR pairs.
    static_assert(HID % 2 == 0, "hidden width must be even");
    const unerf_v2f* __restrict__ w0p = reinterpret_cast<const unerf_v2f*>(w0t);
    const unerf_v2f* __restrict__ b0p = reinterpret_cast<const unerf_v2f*>(b0);
    unerf_v2f h2[HID / 2 + 1];
#pragma unroll
    for (int j = 0; j < HID / 2; ++j) h2[j] = b0p[j];
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) {
        const unerf_v2f fk = {feat[k], feat[k]};
#pragma unroll
        for (int j = 0; j < HID / 2; ++j) h2[j] = __builtin_elementwise_fma(fk, w0p[k * (HID / 2) + j], h2[j]);
    }
#pragma unroll
    for (int j = 0; j < HID / 2; ++j) {  // ReLU as an integer max on the bit pattern (one op, see mf_relu)
        o = fmaf(__int_as_float(max(__float_as_int(h2[j].x), 0)), w1t[2 * j], o);
        o = fmaf(__int_as_float(max(__float_as_int(h2[j].y), 0)), w1t[2 * j + 1], o);
    }
    }
    a.out[idx] = a.avg * unerf_exp(o) * sel;
}

// Patch-schedule form of prop_density_kernel with the per-RAY traffic staged through LDS.  A workgroup is an 8x8
// pixel patch x 4 consecutive sample indices, so its four waves need the same 64 origins / directions and five
// consecutive bin edges of the same 64 rays, and produce 4 adjacent densities per ray.  Ray-indexed accesses cost
// the texture path one cache line per LANE (64 rays = 64 rows of sbins / density_out): loading them once per
// workgroup and storing one 16-byte vector per ray removes about half of the cache-line accesses of the 96-sample
// pass (own sbins row per ray) and a quarter of the 256-sample pass (shared sbins).  Same arithmetic on
// the same values: bit-identical to prop_density_kernel.
#ifndef UNERF_PROP_XCD
#define UNERF_PROP_XCD 1
#endif
template <int L, int HID>
__global__ __launch_bounds__(256) void prop_patch_kernel(PropArgs a) {
    __shared__ float s_od[6][64];
    __shared__ float s_e[5][64];
    __shared__ float s_out[64][4];
    // XCD-aware order (UNERF_PROP_XCD): workgroups go to the 8 XCDs round-robin; every XCD takes a CONTIGUOUS
    // eighth of the (patch, sample group) list -- a band of the image -- instead of every eighth workgroup, so the grid lines
    // neighbouring patches share are fetched into one L2 instead of all eight.  Pure scheduling: same values per sample.
#if UNERF_PROP_XCD
    const uint32_t per = (gridDim.x + 7u) >> 3, bid = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (bid >= a.nblocks) return;
#else
    const uint32_t bid = blockIdx.x;
#endif
    const uint32_t patch = fastdiv(bid, a.div_nsg), sg = bid - patch * a.nsg;
    const uint32_t band = fastdiv(patch, a.div_pcols), pc = patch - band * a.pcols;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t x = pc * 8u + (lane & 7u);
    const int64_t y = a.first_row + (int64_t)band * 8 + (lane >> 3);
    int64_t r = y * (int64_t)a.img_w + x - a.g0;
    const bool ray_ok = x < a.img_w && r >= 0 && r < a.R;
    if (!ray_ok) r = 0;
    const int i_base = (int)(sg * 4u);
    {   // stage: wave 0 the ray, wave 1 its five raw bin edges (one 16-byte + one 4-byte load when the row allows)
        if (wv == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                s_od[c][lane] = a.origins[r * 3 + c];
                s_od[3 + c][lane] = a.dirs[r * 3 + c];
            }
        } else if (wv == 1 && a.sstride == 0) {
            // one shared row of bin edges (the 256 initial bins): the Euclidean edges are the same for every ray, so
            // five lanes convert them ONCE for the workgroup (one conversion stream instead of two per wave; a
            // conversion is an IEEE division and the kernel is VALU-issue bound) and every lane reads them back
            const float* sb = a.sbins + i_base;
            const float e = unerf_s2e(sb[min((int)(lane < 5u ? lane : 4u), a.n - i_base)], a.s_near, a.s_far, a.lin);
            if (lane < 5u) s_e[lane][0] = e;
        } else if (wv == 1) {
            const float* sb = a.sbins + r * a.sstride + i_base;
            if (i_base + 4 <= a.n) {  // uniform
                struct __attribute__((packed, aligned(4))) Q { float x, y, z, w; };
                const Q q = *reinterpret_cast<const Q*>(sb);
                s_e[0][lane] = q.x; s_e[1][lane] = q.y; s_e[2][lane] = q.z; s_e[3][lane] = q.w;
                s_e[4][lane] = sb[4];
            } else {
#pragma unroll
                for (int e = 0; e < 5; ++e) s_e[e][lane] = sb[min(e, a.n - i_base)];
            }
        }
    }
    __syncthreads();
    const int i = i_base + (int)wv;
    float dens = 0.f;
    if (ray_ok && i < a.n) {
        const float t = a.sstride == 0 ? s_e[wv][0] + s_e[wv + 1][0]   // uniform: already Euclidean (same bits)
                                       : unerf_s2e(s_e[wv][lane], a.s_near, a.s_far, a.lin) + unerf_s2e(s_e[wv + 1][lane], a.s_near, a.s_far, a.lin);
        float px = s_od[0][lane] + s_od[3][lane] * t / 2.f;
        float py = s_od[1][lane] + s_od[4][lane] * t / 2.f;
        float pz = s_od[2][lane] + s_od[5][lane] * t / 2.f;
        const float sel = unerf_normalize_position(px, py, pz, a.box);
        const uint32_t mask = (1u << a.net.log2T) - 1u;
        float feat[2 * L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float2 f;
            if (a.net.tcnn_levels && a.net.grid_half) {  // uniform: tcnn layout, half2 rows, tcnn's half arithmetic
                f = unerf_tcnn_level_feat_half(a.net.table, a.net.tcnn_levels[l], px, py, pz);
            } else if (a.net.tcnn_levels) {  // uniform: tcnn-layout grid
                f = unerf_tcnn_level_feat(reinterpret_cast<const float2*>(a.net.table), a.net.tcnn_levels[l], px, py, pz);
            } else if (l < a.net.n_dense) {  // wave-uniform: coarse level with a dense, x-paired copy
                f = unerf_dense_level<(UNERF_PROP_BLEND_FMA != 0)>(reinterpret_cast<const float4*>(a.net.dense) + a.net.dense_off[l], a.net.dense_dim[l],
                                      px, py, pz, a.net.scalings[l]);
            } else {
                const float2* lvl = reinterpret_cast<const float2*>(a.net.table) + ((size_t)l << a.net.log2T);
                f = unerf_hash_level<false, (UNERF_PROP_BLEND_FMA != 0)>(lvl, px, py, pz, a.net.scalings[l], mask);
            }
            feat[2 * l] = f.x;
            feat[2 * l + 1] = f.y;
        }
        const float* __restrict__ w1t = a.net.w1t;
        float o = a.net.b1[0];
        if constexpr (HID == 0) {   // use_linear=True: one Linear on the grid features (see prop_density_kernel)
#pragma unroll
            for (int k = 0; k < 2 * L; ++k) o = fmaf(feat[k], w1t[k], o);
        } else {
        static_assert(HID % 2 == 0, "hidden width must be even");
        const unerf_v2f* __restrict__ w0p = reinterpret_cast<const unerf_v2f*>(a.net.w0t);
        const unerf_v2f* __restrict__ b0p = reinterpret_cast<const unerf_v2f*>(a.net.b0);
        unerf_v2f h2[HID / 2 + 1];
#pragma unroll
        for (int j = 0; j < HID / 2; ++j) h2[j] = b0p[j];
#pragma unroll
        for (int k = 0; k < 2 * L; ++k) {
            const unerf_v2f fk = {feat[k], feat[k]};
#pragma unroll
            for (int j = 0; j < HID / 2; ++j) h2[j] = __builtin_elementwise_fma(fk, w0p[k * (HID / 2) + j], h2[j]);
        }
        // [probe:prop-mlp-out begin]  (benchmarks/probe_source.py rewrites the marked span in COPIES of this file)
#pragma unroll
        for (int j = 0; j < HID / 2; ++j) {
            o = fmaf(__int_as_float(max(__float_as_int(h2[j].x), 0)), w1t[2 * j], o);
            o = fmaf(__int_as_float(max(__float_as_int(h2[j].y), 0)), w1t[2 * j + 1], o);
        }
        // [probe:prop-mlp-out end]
        }
        dens = a.avg * unerf_exp(o) * sel;
    }
    if (a.vec4) {  // uniform: n % 4 == 0 and a 16-byte aligned output: the ray's 4 densities leave as one store
        s_out[lane][wv] = dens;
        __syncthreads();
        if (wv == 0 && ray_ok && i_base < a.n)
            *reinterpret_cast<float4*>(a.out + r * a.n + i_base) = *reinterpret_cast<const float4*>(&s_out[lane][0]);
    } else if (ray_ok && i < a.n) {
        a.out[r * a.n + i] = dens;
    }
}

extern "C" int unerf_proposal_density(const float* origins, const float* directions, const float* sbins,
                                      int64_t sbins_stride, int64_t R, int n, float near_plane, float far_plane, int spacing,
                                      const unerf_density_net* net, float average_init_density, float* density_out,
                                      int64_t ray_offset, int image_width, void* stream) {
    UNERF_REQUIRE(net && (R == 0 || (origins && directions && sbins && density_out)), "proposal_density: null pointer");
    UNERF_REQUIRE(net->table && (net->scalings || net->tcnn_levels) && (net->hidden == 0 || (net->w0t && net->b0)) && net->w1t && net->b1,
                  "proposal_density: null pointer inside unerf_density_net");
    UNERF_REQUIRE(R >= 0 && n >= 1, "proposal_density: bad R/n");
    UNERF_REQUIRE(sbins_stride == 0 || sbins_stride >= n + 1, "proposal_density: sbins_stride %lld < n+1",
                  (long long)sbins_stride);
    UNERF_REQUIRE(net->tcnn_levels || (net->log2T >= 1 && net->log2T <= 24), "proposal_density: bad log2T");
    UNERF_REQUIRE(net->n_dense >= 0 && net->n_dense <= 8 && net->n_dense <= net->L && (net->n_dense == 0 || net->dense),
                  "proposal_density: bad dense level description");
    for (int l = 0; l < net->n_dense; ++l)  // a level is addressed with 32-bit byte offsets
        UNERF_REQUIRE(net->dense_dim[l] >= 2 && net->dense_dim[l] <= 640, "proposal_density: dense_dim[%d]=%d outside [2,640]",
                      l, net->dense_dim[l]);
    if (R == 0) return UNERF_OK;
    PropArgs a;
    a.origins = origins; a.dirs = directions; a.sbins = sbins; a.sstride = sbins_stride; a.R = R; a.n = n;
    UNERF_REQUIRE_SPACING(spacing); a.lin = spacing; a.s_near = unerf_spacing_of(near_plane, spacing); a.s_far = unerf_spacing_of(far_plane, spacing);
    a.net = *net; a.avg = average_init_density; a.out = density_out;
    a.box = make_norm_box(net->use_aabb, net->aabb);
    a.fd = make_fastdiv((uint32_t)n); a.small = (R * (int64_t)n < (1ll << 31)) ? 1 : 0;
    a.img_w = 0; a.pcols = 0; a.nsg = 0; a.nblocks = 0; a.div_nsg = a.div_pcols = make_fastdiv(1); a.g0 = 0; a.first_row = 0;
    a.vec4 = (n % 4 == 0 && ((uintptr_t)density_out & 15u) == 0) ? 1 : 0;
    dim3 grid(blocks_for(R * (int64_t)n, 256)), block(256);
    if (image_width >= 8 && R >= 8 * (int64_t)image_width && ray_offset >= 0) {  // at least one full 8-row band
        const int64_t band0 = (ray_offset / image_width) / 8, band1 = ((ray_offset + R - 1) / image_width) / 8;
        const int64_t pcols = (image_width + 7) / 8, nsg = (n + 3) / 4;
        const int64_t blocks = (band1 - band0 + 1) * pcols * nsg;
        if (blocks < (1ll << 31)) {
            a.img_w = (uint32_t)image_width; a.pcols = (uint32_t)pcols; a.nsg = (uint32_t)nsg;
            a.div_nsg = make_fastdiv((uint32_t)nsg); a.div_pcols = make_fastdiv((uint32_t)pcols);
            a.g0 = ray_offset; a.first_row = band0 * 8;
            a.nblocks = (uint32_t)blocks;
            grid = dim3((unsigned)(UNERF_PROP_XCD ? ((blocks + 7) / 8) * 8 : blocks));
        }
    }
    hipStream_t st = (hipStream_t)stream;
    // image-ordered rays: the LDS-staged patch kernel; anything else: one thread per sample
    const bool patch = a.img_w != 0 && !getenv("UNERF_NO_PATCH_KERNEL");
#define UNERF_PROP_LAUNCH(LL, HH)                                                                          \
    if (patch) hipLaunchKernelGGL((prop_patch_kernel<LL, HH>), grid, block, 0, st, a);                    \
    else hipLaunchKernelGGL((prop_density_kernel<LL, HH>), grid, block, 0, st, a)
    if (net->L == 5 && net->hidden == 16) { UNERF_PROP_LAUNCH(5, 16); }
    else if (net->L == 5 && net->hidden == 64) { UNERF_PROP_LAUNCH(5, 64); }
    else if (net->L == 8 && net->hidden == 64) { UNERF_PROP_LAUNCH(8, 64); }
    else if (net->L == 8 && net->hidden == 16) { UNERF_PROP_LAUNCH(8, 16); }
    else if (net->L == 5 && net->hidden == 0) { UNERF_PROP_LAUNCH(5, 0); }   // use_linear=True
    else if (net->L == 8 && net->hidden == 0) { UNERF_PROP_LAUNCH(8, 0); }
    else {
        unerf_set_error("proposal_density: unsupported (L=%d, hidden=%d); built: (5,16) (5,64) (8,16) (8,64) (5,0) (8,0)", net->L,
                        net->hidden);
        return UNERF_ERR_ARG;
    }
#undef UNERF_PROP_LAUNCH
    return unerf_check_launch("proposal_density");
}

// ======================================================================================
// 4. weights + PDF resampling: one wave per ray, lane owns EPL consecutive samples
// ======================================================================================
struct PdfArgs {
    const float* density;
    const float* sbins;
    int64_t sstride;
    int64_t R;
    int n;
    float s_near, s_far;
    int lin;   // UNERF_SPACING_*: 1 = identity spacing (UniformSampler)
    const float* u;
    int m;
    float pad, eps;
    float* sbins_out;
    float* prop_depth;
    float* weights_out;
    float* clip;
    int64_t ray_offset, chunk_rays;
};

#define PDF_MAXN 256

#define PDF_RAYS_PER_BLOCK 32

// Each wave of the pdf kernel works on its own LDS rows, and a wave's DS operations complete in issue
// order, so a compiler-level fence is all the "barrier" it needs; the four waves of a block are then
// free to drift apart and hide one another's latency (s_barrier kept them in lock-step).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Thousands of rays share one chunk word: peek first (L2-coherent relaxed load) and only send the atomic
// when it can still win -- a stale peek merely costs a redundant atomic, never a wrong result.
__device__ __forceinline__ void pdf_clip_commit(float* clip, int64_t chunk, unsigned int lo, unsigned int hi) {
    unsigned int* c = reinterpret_cast<unsigned int*>(clip) + chunk * 2;
    if (lo < __hip_atomic_load(c + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(c + 0, lo);
    if (hi > __hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(c + 1, hi);
}

template <int EPL>
__global__ __launch_bounds__(256) void pdf_kernel(PdfArgs a) {
    __shared__ float s_sb[4][PDF_MAXN + 4];
    __shared__ float s_cdf[4][PDF_MAXN + 4];
    __shared__ float s_nb[4][PDF_MAXN + 4];
    __shared__ unsigned int s_clip[4][2];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = a.n, nb = a.m + 1;
    const int top_step = 1 << (31 - __builtin_clz((unsigned)n));   // largest power of two <= n
    // A block walks PDF_RAYS_PER_BLOCK consecutive rays, four (one per wave) at a time, and keeps the
    // running min/max of the first/last sample positions in registers: one peek + atomic per block and
    // chunk instead of one per ray (2 M same-address L2 reads per frame made this kernel 3.5 ms).
    int64_t cur_chunk = -1;
    unsigned int cmin = 0x7F800000u, cmax = 0u;
    // One shared row of bin edges (sstride == 0: the 256 initial bins): its Euclidean edges are the same for every ray,
    // so each wave converts them ONCE per block into its own LDS row (one conversion = an IEEE division, ~15
    // instructions; per ray that was EPL + 1 of them per lane) and the raw row is staged once as well.  Same values.
    __shared__ float s_eu[4][PDF_MAXN + 4];
    const bool shared_row = a.sstride == 0;
    if (shared_row) {
        for (int k = lane; k <= n; k += 64) {
            const float b = a.sbins[k];
            s_sb[wv][k] = b;
            s_eu[wv][k] = unerf_s2e(b, a.s_near, a.s_far, a.lin);
        }
    }
    for (int it = 0; it < PDF_RAYS_PER_BLOCK / 4; ++it) {
    int64_t r = (int64_t)blockIdx.x * PDF_RAYS_PER_BLOCK + it * 4 + wv;
    const bool ray_ok = r < a.R;
    if (!ray_ok) r = a.R - 1;  // keep the wave alive for the block barrier at the end
    wave_lds_sync();           // previous ray's LDS rows are free again
    if (!shared_row) {
        const float* sb = a.sbins + r * a.sstride;
        for (int k = lane; k <= n; k += 64) s_sb[wv][k] = sb[k];
    }
    wave_lds_sync();

    float eu[EPL + 1], w[EPL], dd[EPL];
    const int k0 = lane * EPL;
    if (shared_row) {
#pragma unroll
        for (int e = 0; e <= EPL; ++e) eu[e] = s_eu[wv][min(k0 + e, n)];
    } else {
#pragma unroll
        for (int e = 0; e <= EPL; ++e) eu[e] = unerf_s2e(s_sb[wv][min(k0 + e, n)], a.s_near, a.s_far, a.lin);
    }
    float lsum = 0.f, lexcl[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int k = k0 + e;
        float dens = (k < n) ? a.density[r * n + k] : 0.f;
        dd[e] = (k < n) ? (eu[e + 1] - eu[e]) * dens : 0.f;
        lexcl[e] = lsum;
        lsum += dd[e];
    }
    float carry = group_excl_scan<64>(lsum, lane);
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        float alpha = 1.f - unerf_exp(-dd[e]);
        float T = unerf_exp(-(carry + lexcl[e]));
        w[e] = (k0 + e < n) ? unerf_nan_to_num(alpha * T) : 0.f;
        if (a.weights_out && ray_ok && k0 + e < n) a.weights_out[r * n + k0 + e] = w[e];
    }
    // median depth of this level (prop_depth_i)
    if (a.prop_depth) {
        float ls = 0.f, lc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            ls += w[e];
            lc[e] = ls;
        }
        float cbase = group_excl_scan<64>(ls, lane);
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) cnt += (k0 + e < n && (cbase + lc[e]) < 0.5f) ? 1 : 0;
        cnt = group_sum_i<64>(cnt);
        int idx = min(cnt, n - 1);
        int owner = idx / EPL, slot = idx - owner * EPL;
        float val = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            float st = (eu[e] + eu[e + 1]) / 2.f;
            float got = __shfl(st, owner, 64);
            if (e == slot) val = got;
        }
        if (lane == 0 && ray_ok) a.prop_depth[r] = val;
    }
    // histogram -> pdf -> cdf
    float wp[EPL], lw = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        wp[e] = (k0 + e < n) ? w[e] + a.pad : 0.f;
        lw += wp[e];
    }
    float wsum = group_sum<64>(lw);
    float padding = fmaxf(a.eps - wsum, 0.f);
    float padn = padding / (float)n;
    wsum += padding;
    float lp = 0.f, lcdf[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        float pdf = (k0 + e < n) ? (wp[e] + padn) / wsum : 0.f;
        lp += pdf;
        lcdf[e] = lp;
    }
    float cb = group_excl_scan<64>(lp, lane);
    if (lane == 0) s_cdf[wv][0] = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        if (k0 + e < n) s_cdf[wv][k0 + e + 1] = fminf(1.f, cb + lcdf[e]);
    wave_lds_sync();

    for (int j = lane; j < nb; j += 64) {
        float u = a.u[j];
        // searchsorted(cdf[0..n], u, side="right") = 1 + the last index with cdf <= u (cdf[0] = 0 <= u): descend
        // by powers of two without a data-dependent loop -- 5 instructions per step against ~9 for the
        // lo / hi / mid form, and the kernel is VALU-issue bound.  A probe past the end is clamped to n, whose
        // entry is an ordinary candidate, so the result is the same index.
        int pos = 0;
        for (int step = top_step; step > 0; step >>= 1) {   // uniform trip count
            const int probe = min(pos + step, n);
            pos = (s_cdf[wv][probe] <= u) ? probe : pos;
        }
        const int lo = pos + 1;
        int below = min(max(lo - 1, 0), n), above = min(max(lo, 0), n);
        float g0 = s_cdf[wv][below], g1 = s_cdf[wv][above];
        float b0 = s_sb[wv][below], b1 = s_sb[wv][above];
        float t = (u - g0) / (g1 - g0);
        t = unerf_nan_to_num(t);
        t = fminf(fmaxf(t, 0.f), 1.f);
        float v = b0 + t * (b1 - b0);
        s_nb[wv][j] = v;
        if (ray_ok) a.sbins_out[r * nb + j] = v;
    }
    if (a.clip) {
        wave_lds_sync();
        if (ray_ok) {  // wave-uniform
            // the four edges (first two, last two) converted by lanes 0..3 at once, one conversion stream instead of four
            const int ei = (lane & 2) ? nb - 2 + (lane & 1) : (lane & 1);
            const float ev = unerf_s2e(s_nb[wv][ei], a.s_near, a.s_far, a.lin);
            const float f0 = __shfl(ev, 0, 64), f1 = __shfl(ev, 1, 64), l0 = __shfl(ev, 2, 64), l1 = __shfl(ev, 3, 64);
            float first = (f0 + f1) / 2.f, last = (l0 + l1) / 2.f;
            const int64_t chunk = (a.ray_offset + r) / a.chunk_rays;
            if (chunk != cur_chunk) {
                if (cur_chunk >= 0 && lane == 0) pdf_clip_commit(a.clip, cur_chunk, cmin, cmax);
                cur_chunk = chunk;
                cmin = 0x7F800000u;
                cmax = 0u;
            }
            // positive floats order like their bit patterns
            cmin = min(cmin, __float_as_uint(first));
            cmax = max(cmax, __float_as_uint(last));
        }
    }
    }  // rays of this block
    if (a.clip) {
        // the four waves normally sit in one chunk: merge them through LDS and send one update
        const int64_t blk_chunk = (a.ray_offset + (int64_t)blockIdx.x * PDF_RAYS_PER_BLOCK) / a.chunk_rays;
        const bool mergeable = cur_chunk == blk_chunk;  // this wave never left the block's first chunk
        if (lane == 0) {
            s_clip[wv][0] = mergeable ? cmin : 0x7F800000u;
            s_clip[wv][1] = mergeable ? cmax : 0u;
            if (!mergeable && cur_chunk >= 0) pdf_clip_commit(a.clip, cur_chunk, cmin, cmax);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int lo = min(min(s_clip[0][0], s_clip[1][0]), min(s_clip[2][0], s_clip[3][0]));
            unsigned int hi = max(max(s_clip[0][1], s_clip[1][1]), max(s_clip[2][1], s_clip[3][1]));
            if (lo != 0x7F800000u || hi != 0u) pdf_clip_commit(a.clip, blk_chunk, lo, hi);
        }
    }
}

extern "C" int unerf_weights_pdf_resample(const float* density, const float* sbins, int64_t sbins_stride, int64_t R,
                                          int n, float near_plane, float far_plane, int spacing, const float* u, int m,
                                          float histogram_padding, float eps, float* sbins_out, float* prop_depth_out,
                                          float* weights_out, float* clip_minmax, int64_t ray_offset,
                                          int64_t chunk_rays, void* stream) {
    UNERF_REQUIRE(R <= 0 || (density && sbins && u && sbins_out), "weights_pdf_resample: null pointer");
    UNERF_REQUIRE(n >= 1 && n <= PDF_MAXN, "weights_pdf_resample: n=%d outside [1,%d]", n, PDF_MAXN);
    UNERF_REQUIRE(m >= 1 && m + 1 <= PDF_MAXN, "weights_pdf_resample: m=%d outside [1,%d]", m, PDF_MAXN - 1);
    UNERF_REQUIRE(sbins_stride == 0 || sbins_stride >= n + 1, "weights_pdf_resample: sbins_stride < n+1");
    UNERF_REQUIRE(!clip_minmax || chunk_rays > 0, "weights_pdf_resample: chunk_rays must be > 0 with clip_minmax");
    UNERF_REQUIRE(near_plane > 0.f, "weights_pdf_resample: near plane must be > 0");
    if (R <= 0) return UNERF_OK;
    PdfArgs a;
    a.density = density; a.sbins = sbins; a.sstride = sbins_stride; a.R = R; a.n = n;
    UNERF_REQUIRE_SPACING(spacing); a.lin = spacing; a.s_near = unerf_spacing_of(near_plane, spacing); a.s_far = unerf_spacing_of(far_plane, spacing);
    a.u = u; a.m = m; a.pad = histogram_padding; a.eps = eps; a.sbins_out = sbins_out; a.prop_depth = prop_depth_out;
    a.weights_out = weights_out; a.clip = clip_minmax; a.ray_offset = ray_offset; a.chunk_rays = chunk_rays;
    dim3 grid(blocks_for(R, PDF_RAYS_PER_BLOCK)), block(256);
    hipStream_t st = (hipStream_t)stream;
    int epl = (n + 63) / 64;
    switch (epl) {
        case 1: hipLaunchKernelGGL((pdf_kernel<1>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((pdf_kernel<2>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((pdf_kernel<3>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((pdf_kernel<4>), grid, block, 0, st, a); break;
    }
    return unerf_check_launch("weights_pdf_resample");
}

// ======================================================================================
// 5. main field.  One wave per block; lane = one sample; activations live in LDS as
//    [feature][lane] (bank-conflict free), weights stream through the scalar cache
//    (uniform addresses -> s_load), accumulators are static registers.
// ======================================================================================
// Which 32 rays form the columns of a matrix-kernel tile.  Default: 32 consecutive rays.  When the caller says
// that its rays are consecutive pixels of a row-major image (unerf_field_params.image_width), a tile is an
// 8 x 4 PIXEL PATCH instead: neighbouring samples then sit in fewer distinct grid cells on the fine levels, so a
// gather instruction touches fewer cache lines (the texture-address unit is what binds the split-f16 ACTIVE
// kernel).  Pure scheduling: every (ray, sample) is computed exactly as before.
struct TileMap {
    uint32_t img_w;      // 0: consecutive rays
    uint32_t pcols;      // 8-pixel patch columns per 4-row band
    FastDiv div_pcols;
    int64_t g0;          // index of local ray 0 in the image
    int64_t first_row;   // first image row of the first band touched by this launch
};

struct FieldArgs {
    const float* origins;
    const float* dirs;
    const float* sbins;
    int64_t R;
    int S;
    float s_near, s_far;
    int lin;   // UNERF_SPACING_*: 1 = identity spacing (UniformSampler)
    int64_t ray_offset;
    unerf_field_params p;
    float* density;
    float* rgb;
    float* aux;
    float* aux2;
    int32_t keep_hi;     // MC-dropout keep threshold thr_s << 16 (unerf_keep_lo / unerf_keep_hi)
    uint32_t keep_pk;    // thr_s in both 16-bit halves (packed-f16 masks of the split-f16 kernels)
    int drop_on;         // K > 0 and p_drop > 0: masks are generated (p_drop == 0 keeps every unit)
    int drop_sites;      // UNERF_DROP_* bits actually in use (0 when !drop_on)
    float drop_scale;
    const float* features;  // optional [16][N][2] level-major planes from unerf_field_gather (MFMA kernel)
    TileMap tm;
    unerf_norm_box box;
    FastDiv div_chunk;      // LAPLACE per-chunk sample sets: division by p.lap_chunk_rays
    uint32_t chunk0;        // ... and ray_offset as a 32-bit ray index (the sample counter bound keeps it below 2^31)
};

// Bin edge -> Euclidean distance.  unerf_field_fwd(near_plane < 0) sets s_near = -1: sbins then already holds
// Euclidean edges (RaySamples.frustums.starts / ends of a caller-made sampler) and passes through untouched.
__device__ __forceinline__ float field_bin_edge(const FieldArgs& a, float b) {
    return a.s_near < 0.f ? b : unerf_s2e(b, a.s_near, a.s_far, a.lin);
}

// ray of column j of ray-block rb (a tile is (rb, sample index)); invalid columns are clamped by the caller
__device__ __forceinline__ void tile_ray(const FieldArgs& a, uint32_t rb, int j, int64_t& r, bool& valid) {
    if (a.tm.img_w == 0) {  // uniform
        r = (int64_t)rb * 32 + j;
        valid = r < a.R;
    } else {
        const uint32_t band = fastdiv(rb, a.tm.div_pcols), pc = rb - band * a.tm.pcols;
        const uint32_t x = pc * 8u + (uint32_t)(j & 7);
        const int64_t y = a.tm.first_row + (int64_t)band * 4 + (j >> 3);
        r = y * (int64_t)a.tm.img_w + x - a.tm.g0;
        valid = x < a.tm.img_w && r >= 0 && r < a.R;
    }
}

// One column of a matrix-kernel tile: which (ray, sample) it is and where the sample sits (mid-point of its
// spacing bin, euclidean, before contraction).  Shared by the four field_kernel_mfma* kernels.
struct TileSample {
    int64_t n;        // r * S + s
    int64_t r;        // ray (clamped to a valid one when !valid)
    int s;            // sample slot
    bool valid;       // column maps to a ray of this launch (else clamped to the last ray, results dropped)
    float dx, dy, dz;
    float px, py, pz;
};
__device__ __forceinline__ TileSample tile_sample(const FieldArgs& a, uint32_t tile, const FastDiv& div_s, int j) {
    TileSample t;
    const uint32_t rb = fastdiv(tile, div_s);
    const int s = (int)(tile - rb * (uint32_t)a.S);
    int64_t r;
    tile_ray(a, rb, j, r, t.valid);
    if (!t.valid) r = a.R - 1;
    t.n = r * a.S + s;
    t.r = r;
    t.s = s;
    const float* sb = a.sbins + r * (a.S + 1);
    const float e0 = field_bin_edge(a, sb[s]), e1 = field_bin_edge(a, sb[s + 1]);
    const float t01 = e0 + e1;
    t.dx = a.dirs[r * 3 + 0];
    t.dy = a.dirs[r * 3 + 1];
    t.dz = a.dirs[r * 3 + 2];
    t.px = a.origins[r * 3 + 0] + t.dx * t01 / 2.f;
    t.py = a.origins[r * 3 + 1] + t.dy * t01 / 2.f;
    t.pz = a.origins[r * 3 + 2] + t.dz * t01 / 2.f;
    return t;
}

// Where one (pass k, ray r, sample s) lands in the outputs.  Default (sample_major = 0): density [B,R,S], rgb [B,R,S,3],
// aux [R,S] -- the RaySamples layout the Field-level API returns.  sample_major = 1: planes density [B,S,R],
// rgb [B,S,3,R], aux [S,R].  A tile's 32 columns are 32 rays at ONE sample slot, so in the ray-major layout each of
// its stores is a lone 4-byte write into its own cache line (2.9 x the algorithmic bytes reached the fabric,
// profiles/r1_11_active_pmc_summary.json); in the plane layout the same store covers whole 32-byte sectors (8
// consecutive rays of a pixel patch, 32 of a 1-D tile), and the composite kernel reads the planes with a lane per ray.
struct OutIndex {
    int64_t dens, rgb, rgb_stride, aux;
};
__device__ __forceinline__ OutIndex out_index(const FieldArgs& a, int k, const TileSample& ts) {
    OutIndex o;
    if (a.p.sample_major) {  // uniform
        const int64_t plane = (int64_t)k * a.S + ts.s;
        o.dens = plane * a.R + ts.r;
        o.rgb = plane * 3 * a.R + ts.r;
        o.rgb_stride = a.R;
        o.aux = (int64_t)ts.s * a.R + ts.r;
    } else {
        o.dens = (int64_t)k * (a.R * (int64_t)a.S) + ts.n;
        o.rgb = o.dens * 3;
        o.rgb_stride = 1;
        o.aux = ts.n;
    }
    return o;
}
// packed_out: ONE 16-byte row (sigma, r, g, b) per (pass, ray, sample) in `rgb` [B,R,S,4] instead of a dword into
// `density` and three into `rgb`.  A tile's columns are 32 rays at one sample slot, so every store of the ray-major
// layouts is a lone write into its own cache line: four 4-byte stores per sample reached the fabric as 1.84 x (K-pass)
// to 3.4 x (ACTIVE) the algorithmic bytes at 2^20-ray launch groups (profiles/traffic_*.json, round 3).
__device__ __forceinline__ void store_packed(const FieldArgs& a, int k, int64_t n, float sigma, float r, float g, float b) {
    reinterpret_cast<float4*>(a.rgb)[(int64_t)k * (a.R * (int64_t)a.S) + n] = make_float4(sigma, r, g, b);
}

// LAPLACE with per-chunk sample sets (unerf_field_params.lap_chunk_rays): the blob of the set a tile's rays belong to.
// Tiles are 1-D there (32 consecutive rays, host: make_tiles without an image width) and chunk / launch boundaries are
// multiples of 32, so the set is uniform over the tile; `tile` is wave-uniform, so this is scalar arithmetic.
__device__ __forceinline__ const float* lap_set_blob(const FieldArgs& a, const float* blob, uint32_t tile, const FastDiv& div_s) {
    if (a.p.lap_chunk_rays == 0) return blob;
    const uint32_t set = fastdiv(a.chunk0 + fastdiv(tile, div_s) * 32u, a.div_chunk);
    return blob + (size_t)set * UNERF_LAP_BLOB_FLOATS;
}

// host side: fills a.tm and returns the number of tiles
static int64_t make_tiles(FieldArgs& a, int image_width) {
    a.tm.img_w = 0; a.tm.pcols = 0; a.tm.div_pcols = make_fastdiv(1); a.tm.g0 = 0; a.tm.first_row = 0;
    if (image_width >= 8 && a.R >= 4 * (int64_t)image_width) {   // at least one full band, else 1-D tiles
        const int64_t g0 = a.ray_offset, g1 = a.ray_offset + a.R - 1;
        const int64_t band0 = (g0 / image_width) / 4, band1 = (g1 / image_width) / 4;
        const int64_t pcols = (image_width + 7) / 8;
        const int64_t tiles = (band1 - band0 + 1) * pcols * (int64_t)a.S;
        if (tiles < (1ll << 28)) {
            a.tm.img_w = (uint32_t)image_width; a.tm.pcols = (uint32_t)pcols; a.tm.div_pcols = make_fastdiv((uint32_t)pcols);
            a.tm.g0 = g0; a.tm.first_row = band0 * 4;
            return tiles;
        }
    }
    return ((a.R + 31) / 32) * (int64_t)a.S;
}

// acc[o] = b[o] + sum_i act[i] * Wt[i][o]   (sequential over i, fused multiply-add)
template <int IN, int OUT>
__device__ __forceinline__ void dense_lds(const float* __restrict__ Wt, const float* __restrict__ b,
                                          const float* act, int lane, float (&acc)[OUT]) {
#pragma unroll
    for (int o = 0; o < OUT; ++o) acc[o] = b[o];
#pragma unroll 2
    for (int i = 0; i < IN; ++i) {
        float x = act[i * 64 + lane];
#pragma unroll
        for (int o = 0; o < OUT; ++o) acc[o] = fmaf(x, Wt[i * OUT + o], acc[o]);
    }
}

// same, with an inverted-dropout mask on the IN inputs (mask words: unerf_mask_word0 / unerf_mask_step; unit 2j is the
// low half of word j, unit 2j + 1 the high half)
template <int OUT, int IN = 64>
__device__ __forceinline__ void dense_lds_dropout(const float* __restrict__ Wt, const float* __restrict__ b,
                                                  const float* act, int lane, uint32_t pre, int pass,
                                                  uint32_t stream_id, int32_t thr_hi, float scale, float (&acc)[OUT]) {
#pragma unroll
    for (int o = 0; o < OUT; ++o) acc[o] = b[o];
    const uint32_t bh[2] = {unerf_mc_base_h(pre, 0u), unerf_mc_base_h(pre, 1u)};
    for (int j = 0; j < (IN + 1) / 2; ++j) {
        uint32_t rnd = unerf_mask_word0(bh[(j >> 1) & 1], stream_id, (uint32_t)j);
        for (int q = 0; q < pass; ++q) rnd = unerf_mask_step(rnd);
        float x0 = act[(2 * j) * 64 + lane];
        x0 = unerf_keep_lo(rnd, thr_hi) ? x0 * scale : 0.f;
#pragma unroll
        for (int o = 0; o < OUT; ++o) acc[o] = fmaf(x0, Wt[(2 * j) * OUT + o], acc[o]);
        if (2 * j + 1 < IN) {
            float x1 = act[(2 * j + 1) * 64 + lane];
            x1 = unerf_keep_hi(rnd, thr_hi) ? x1 * scale : 0.f;
#pragma unroll
            for (int o = 0; o < OUT; ++o) acc[o] = fmaf(x1, Wt[(2 * j + 1) * OUT + o], acc[o]);
        }
    }
}

template <int N>
__device__ __forceinline__ void store_act(float* act, int lane, const float (&v)[N], int row0, bool relu) {
#pragma unroll
    for (int o = 0; o < N; ++o) act[(row0 + o) * 64 + lane] = relu ? fmaxf(v[o], 0.f) : v[o];
}

template <int MODE>
__global__ __launch_bounds__(64) void field_kernel(FieldArgs a) {
    extern __shared__ float lds[];
    float* A = lds;                 // [64][64]
    float* Bf = lds + 64 * 64;      // [64][64] (MCDROPOUT only)
    const int lane = threadIdx.x;
    const int64_t N = a.R * (int64_t)a.S;
    int64_t n = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = n < N;
    if (!valid) n = N - 1;
    const int64_t r = n / a.S;
    const int s = (int)(n - r * a.S);

    const float* sb = a.sbins + r * (a.S + 1);
    float e0 = field_bin_edge(a, sb[s]), e1 = field_bin_edge(a, sb[s + 1]);
    float t = e0 + e1;
    float dxr = a.dirs[r * 3 + 0], dyr = a.dirs[r * 3 + 1], dzr = a.dirs[r * 3 + 2];
    float px = a.origins[r * 3 + 0] + dxr * t / 2.f;
    float py = a.origins[r * 3 + 1] + dyr * t / 2.f;
    float pz = a.origins[r * 3 + 2] + dzr * t / 2.f;
    const float sel = unerf_normalize_position(px, py, pz, a.box);

    const uint32_t mask = (1u << a.p.log2T) - 1u;
#pragma unroll 4
    for (int l = 0; l < 16; ++l) {
        float2 f;
        if (a.p.tcnn_levels && a.p.grid_half) {   // uniform
            f = unerf_tcnn_level_feat_half(a.p.table, a.p.tcnn_levels[l], px, py, pz);
        } else if (a.p.tcnn_levels) {
            f = unerf_tcnn_level_feat(reinterpret_cast<const float2*>(a.p.table), a.p.tcnn_levels[l], px, py, pz);
        } else {
            const float2* lvl = reinterpret_cast<const float2*>(a.p.table) + ((size_t)l << a.p.log2T);
            f = unerf_hash_level(lvl, px, py, pz, a.p.scalings[l], mask);
        }
        A[(2 * l) * 64 + lane] = f.x;
        A[(2 * l + 1) * 64 + lane] = f.y;
    }

    // direction encoding inputs: get_normalized_directions(d) = (d+1)/2 (torch SHEncoding uses it as is)
    float sh[16];
    {
        float ux = (dxr + 1.f) / 2.f, uy = (dyr + 1.f) / 2.f, uz = (dzr + 1.f) / 2.f;
        if (a.p.sh_remap) {
            ux = ux * 2.f - 1.f;
            uy = uy * 2.f - 1.f;
            uz = uz * 2.f - 1.f;
        }
        unerf_sh16(ux, uy, uz, sh);
    }

    float acc[64];
    dense_lds<32, 64>(a.p.w0t, a.p.b0, A, lane, acc);

    if constexpr (MODE == UNERF_FIELD_ACTIVE) {
        store_act<64>(A, lane, acc, 0, true);
        float o1[17];
        dense_lds<64, 17>(a.p.w1t, a.p.b1, A, lane, o1);
        float density = a.p.average_init_density * expf(o1[0]) * sel;
        float beta = unerf_softplus(o1[16]) + a.p.beta_min;
        store_act<16>(A, lane, sh, 0, false);
#pragma unroll
        for (int g = 0; g < 15; ++g) A[(16 + g) * 64 + lane] = o1[1 + g];
        dense_lds<31, 64>(a.p.h0t, a.p.hb0, A, lane, acc);
        store_act<64>(A, lane, acc, 0, true);
        dense_lds<64, 64>(a.p.h1t, a.p.hb1, A, lane, acc);
        store_act<64>(A, lane, acc, 0, true);
        float c[3];
        dense_lds<64, 3>(a.p.h2t, a.p.hb2, A, lane, c);
        if (valid) {
            a.aux[n] = beta;
            if (a.p.packed_out) {
                store_packed(a, 0, n, density, unerf_sigmoid(c[0]), unerf_sigmoid(c[1]), unerf_sigmoid(c[2]));
            } else {
                a.density[n] = density;
                a.rgb[n * 3 + 0] = unerf_sigmoid(c[0]);
                a.rgb[n * 3 + 1] = unerf_sigmoid(c[1]);
                a.rgb[n * 3 + 2] = unerf_sigmoid(c[2]);
            }
        }
    } else if constexpr (MODE == UNERF_FIELD_MCDROPOUT) {
        store_act<64>(A, lane, acc, 0, true);  // hidden h stays in A for every pass
        const int passes = a.p.K > 0 ? a.p.K : 1;
        const uint32_t sidx = (uint32_t)((uint64_t)a.ray_offset * (uint64_t)a.S + (uint64_t)n);
        for (int k = 0; k < passes; ++k) {
            const uint32_t base = unerf_mc_pre(unerf_mc_key(a.p.seed, 0u), sidx);   // dense_lds_dropout derives both half bases
            float o1[16];
            if (a.drop_sites & UNERF_DROP_TRUNK) dense_lds_dropout<16>(a.p.w1t, a.p.b1, A, lane, base, k, 0u, a.keep_hi, a.drop_scale, o1);
            else dense_lds<64, 16>(a.p.w1t, a.p.b1, A, lane, o1);
            float density = a.p.average_init_density * expf(o1[0]) * sel;
            store_act<16>(Bf, lane, sh, 0, false);
#pragma unroll
            for (int g = 0; g < 15; ++g) Bf[(16 + g) * 64 + lane] = o1[1 + g];
            if (a.drop_sites & UNERF_DROP_HEADIN) {   // rgb_dropout_layers contains 0: Dropout on the head's 63 inputs
                for (int e = 0; e < 32; ++e) Bf[(31 + e) * 64 + lane] = a.p.app_embed[e];
                dense_lds_dropout<64, 63>(a.p.h0_full_t, a.p.hb0_raw, Bf, lane, base, k, 3u, a.keep_hi, a.drop_scale, acc);
            } else {
                dense_lds<31, 64>(a.p.h0t, a.p.hb0, Bf, lane, acc);
            }
            store_act<64>(Bf, lane, acc, 0, true);
            if (a.drop_sites & UNERF_DROP_HEAD0) {
                float acc2[64];
                dense_lds_dropout<64>(a.p.h1t, a.p.hb1, Bf, lane, base, k, 2u, a.keep_hi, a.drop_scale, acc2);
                store_act<64>(Bf, lane, acc2, 0, true);
            } else {
                dense_lds<64, 64>(a.p.h1t, a.p.hb1, Bf, lane, acc);
                store_act<64>(Bf, lane, acc, 0, true);
            }
            float c[3];
            if (a.drop_sites & UNERF_DROP_HEAD1) dense_lds_dropout<3>(a.p.h2t, a.p.hb2, Bf, lane, base, k, 1u, a.keep_hi, a.drop_scale, c);
            else dense_lds<64, 3>(a.p.h2t, a.p.hb2, Bf, lane, c);
            if (valid && a.p.packed_out) {
                store_packed(a, k, n, density, unerf_sigmoid(c[0]), unerf_sigmoid(c[1]), unerf_sigmoid(c[2]));
            } else if (valid) {
                int64_t q = (int64_t)k * N + n;
                a.density[q] = density;
                a.rgb[q * 3 + 0] = unerf_sigmoid(c[0]);
                a.rgb[q * 3 + 1] = unerf_sigmoid(c[1]);
                a.rgb[q * 3 + 2] = unerf_sigmoid(c[2]);
            }
        }
    } else {  // LAPLACE: bare Linear base (no ReLU), sampled last layers
        store_act<64>(A, lane, acc, 0, false);
        float geo[15];
        dense_lds<64, 15>(a.p.w1t, a.p.b1, A, lane, geo);
        const int nl = a.p.n_lap, nr = a.p.n_lap_rgb;
        // per-chunk sample sets (unerf_field_params.lap_chunk_rays): a wave of 64 consecutive samples may straddle two
        const size_t set = a.p.lap_chunk_rays ? (size_t)fastdiv(a.chunk0 + (uint32_t)r, a.div_chunk) : 0;
        float mu = 0.f, mu2 = 0.f;
        for (int q = 0; q < nl; ++q) {
            const float* __restrict__ w = a.p.ws_density + (set * nl + (size_t)q) * 65;
            float pre = 0.f;
#pragma unroll
            for (int i = 0; i < 64; ++i) pre = fmaf(acc[i], w[i], pre);
            pre += w[64];
            float pred = a.p.lap_softplus ? unerf_softplus(pre) : expf(pre);
            mu += pred;
            mu2 += pred * pred;
        }
        mu /= (float)nl;
        mu2 /= (float)nl;
        float var_d = a.p.lap_mask_density ? 0.f : mu2 - mu * mu;
        store_act<16>(A, lane, sh, 0, false);
        store_act<15>(A, lane, geo, 16, false);
        dense_lds<31, 64>(a.p.h0t, a.p.hb0, A, lane, acc);
        store_act<64>(A, lane, acc, 0, true);
        dense_lds<64, 64>(a.p.h1t, a.p.hb1, A, lane, acc);
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i] = fmaxf(acc[i], 0.f);
        float m1[3] = {0.f, 0.f, 0.f}, m2[3] = {0.f, 0.f, 0.f};
        for (int q = 0; q < nr; ++q) {
            const float* __restrict__ w = a.p.ws_rgb + (set * nr + (size_t)q) * 195;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float pre = 0.f;
#pragma unroll
                for (int i = 0; i < 64; ++i) pre = fmaf(acc[i], w[c * 64 + i], pre);
                pre += w[192 + c];
                float pred = unerf_sigmoid(pre);
                m1[c] += pred;
                m2[c] += pred * pred;
            }
        }
        float vsum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            m1[c] /= (float)nr;
            m2[c] /= (float)nr;
            vsum += fmaxf(m2[c] - m1[c] * m1[c], 0.f);
        }
        if (valid) {
            a.density[n] = a.p.lap_mask_density ? mu * sel : mu;  // inference: NOT selector-masked (laplace_field.py:356-362)
            a.aux[n] = var_d;
            a.aux2[n] = vsum / 3.f;
            a.rgb[n * 3 + 0] = m1[0];
            a.rgb[n * 3 + 1] = m1[1];
            a.rgb[n * 3 + 2] = m1[2];
        }
    }
}

// --------------------------------------------------------------------------------------
// 5a'. ANY-WIDTH field (the slow path).  The reference forwards hidden_dim, hidden_dim_color, features_per_level and
// appearance_embed_dim from the model config to the field (activenerfacto_model.py:63-77, mcdropout_models.py:66-80,
// laplace_model.py:169-186) and the field class takes geo_feat_dim; the matrix kernels and field_kernel above are
// built for nerfacto's 64 / 64 / 2 / 15.  This kernel takes every width at run time (unerf_field_params.hidden, ...):
// one lane per sample, activations in LDS as [unit][lane], weights through the scalar cache, eight outputs per sweep
// of a layer's inputs.  Same arithmetic definitions (mask words, SH, activations) as field_kernel -- roughly 10-20 x the
// time of the matrix kernels: a configuration that is kept CORRECT, and says so once (ops.py warns).
// --------------------------------------------------------------------------------------
__device__ __forceinline__ void dense_any(const float* __restrict__ Wt, const float* __restrict__ b, const float* in,
                                          float* out, int IN, int OUT, int lane, bool relu) {
    for (int o0 = 0; o0 < OUT; o0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = o0 + j < OUT ? b[o0 + j] : 0.f;
        for (int i = 0; i < IN; ++i) {
            const float x = in[i * 64 + lane];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (o0 + j < OUT) acc[j] = fmaf(x, Wt[i * OUT + o0 + j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (o0 + j < OUT) out[(o0 + j) * 64 + lane] = relu ? fmaxf(acc[j], 0.f) : acc[j];
    }
}
// inverted dropout of `n` units: dst = keep ? src * scale : 0 (dst may be src)
__device__ __forceinline__ void mask_any(const float* src, float* dst, int n, int lane, uint32_t pre, int pass,
                                         uint32_t stream_id, int32_t thr_hi, float scale) {
    const uint32_t bh[2] = {unerf_mc_base_h(pre, 0u), unerf_mc_base_h(pre, 1u)};
    for (int j = 0; 2 * j < n; ++j) {
        uint32_t rnd = unerf_mask_word0(bh[(j >> 1) & 1], stream_id, (uint32_t)j);
        for (int q = 0; q < pass; ++q) rnd = unerf_mask_step(rnd);
        dst[(2 * j) * 64 + lane] = unerf_keep_lo(rnd, thr_hi) ? src[(2 * j) * 64 + lane] * scale : 0.f;
        if (2 * j + 1 < n) dst[(2 * j + 1) * 64 + lane] = unerf_keep_hi(rnd, thr_hi) ? src[(2 * j + 1) * 64 + lane] * scale : 0.f;
    }
}
// one level of the torch-layout grid with FOUR features per row (16-byte rows): the same corners and lerp order
__device__ __forceinline__ void unerf_hash_level4(const float4* __restrict__ lvl, float px, float py, float pz, float scale,
                                                  uint32_t mask, float (&f4)[4]) {
    uint32_t off[8];
    float ox, oy, oz;
    unerf_hash_corners<false, false>(px, py, pz, scale, mask, off, ox, oy, oz);
    float2 lo[8], hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4 v = lvl[off[k] >> 3];
        lo[k] = make_float2(v.x, v.y);
        hi[k] = make_float2(v.z, v.w);
    }
    const float2 a = unerf_blend8(lo, ox, oy, oz), c = unerf_blend8(hi, ox, oy, oz);
    f4[0] = a.x; f4[1] = a.y; f4[2] = c.x; f4[3] = c.y;
}

template <int MODE>
__global__ __launch_bounds__(64) void field_kernel_generic(FieldArgs a, int rows) {
    extern __shared__ float lds[];
    float* A = lds;                       // hidden trunk units (pass-invariant)
    float* B = lds + (size_t)rows * 64;   // ping
    float* Cb = B + (size_t)rows * 64;    // head input [SH16 | geo | (appearance)]
    float* D = Cb + (size_t)rows * 64;    // pong / trunk output
    const int lane = threadIdx.x;
    const int H = a.p.hidden, HC = a.p.hidden_color, G = a.p.geo_dim, F = a.p.feat_per_level, L = a.p.L, AD = a.p.app_dim;
    const int IN0 = L * F, OUT1 = a.p.out1, INC = 16 + G;
    const int64_t N = a.R * (int64_t)a.S;
    int64_t n = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = n < N;
    if (!valid) n = N - 1;
    const int64_t r = n / a.S;
    const int s = (int)(n - r * a.S);
    const float* sb = a.sbins + r * (a.S + 1);
    const float t = field_bin_edge(a, sb[s]) + field_bin_edge(a, sb[s + 1]);
    const float dxr = a.dirs[r * 3 + 0], dyr = a.dirs[r * 3 + 1], dzr = a.dirs[r * 3 + 2];
    float px = a.origins[r * 3 + 0] + dxr * t / 2.f;
    float py = a.origins[r * 3 + 1] + dyr * t / 2.f;
    float pz = a.origins[r * 3 + 2] + dzr * t / 2.f;
    const float sel = unerf_normalize_position(px, py, pz, a.box);
    const uint32_t mask = (1u << a.p.log2T) - 1u;
    for (int l = 0; l < L; ++l) {   // grid features -> B rows 0..IN0-1
        if (F == 4) {
            float f4[4];
            unerf_hash_level4(reinterpret_cast<const float4*>(a.p.table) + ((size_t)l << a.p.log2T), px, py, pz, a.p.scalings[l], mask, f4);
#pragma unroll
            for (int e = 0; e < 4; ++e) B[(4 * l + e) * 64 + lane] = f4[e];
        } else {
            float2 f;
            if (a.p.tcnn_levels && a.p.grid_half) f = unerf_tcnn_level_feat_half(a.p.table, a.p.tcnn_levels[l], px, py, pz);
            else if (a.p.tcnn_levels) f = unerf_tcnn_level_feat(reinterpret_cast<const float2*>(a.p.table), a.p.tcnn_levels[l], px, py, pz);
            else f = unerf_hash_level(reinterpret_cast<const float2*>(a.p.table) + ((size_t)l << a.p.log2T), px, py, pz, a.p.scalings[l], mask);
            B[(2 * l) * 64 + lane] = f.x;
            B[(2 * l + 1) * 64 + lane] = f.y;
        }
    }
    float sh[16];
    {
        float ux = (dxr + 1.f) / 2.f, uy = (dyr + 1.f) / 2.f, uz = (dzr + 1.f) / 2.f;
        if (a.p.sh_remap) {
            ux = ux * 2.f - 1.f;
            uy = uy * 2.f - 1.f;
            uz = uz * 2.f - 1.f;
        }
        unerf_sh16(ux, uy, uz, sh);
    }
    dense_any(a.p.w0t, a.p.b0, B, A, IN0, H, lane, MODE != UNERF_FIELD_LAPLACE);   // LAPLACE: bare Linear (utils.py:22-23)
    if constexpr (MODE == UNERF_FIELD_LAPLACE) {
        dense_any(a.p.w1t, a.p.b1, A, D, H, G, lane, false);                       // mlp_hidden: the geo features
        const int nl = a.p.n_lap, nr = a.p.n_lap_rgb;
        const size_t set = a.p.lap_chunk_rays ? (size_t)fastdiv(a.chunk0 + (uint32_t)r, a.div_chunk) : 0;
        float mu = 0.f, mu2 = 0.f;
        for (int q = 0; q < nl; ++q) {
            const float* __restrict__ w = a.p.ws_density + (set * nl + (size_t)q) * (size_t)(H + 1);
            float pre = 0.f;
            for (int i = 0; i < H; ++i) pre = fmaf(A[i * 64 + lane], w[i], pre);
            pre += w[H];
            const float pred = a.p.lap_softplus ? unerf_softplus(pre) : expf(pre);
            mu += pred;
            mu2 += pred * pred;
        }
        mu /= (float)nl;
        mu2 /= (float)nl;
        const float var_d = a.p.lap_mask_density ? 0.f : mu2 - mu * mu;
#pragma unroll
        for (int e = 0; e < 16; ++e) Cb[e * 64 + lane] = sh[e];
        for (int g = 0; g < G; ++g) Cb[(16 + g) * 64 + lane] = D[g * 64 + lane];
        dense_any(a.p.h0t, a.p.hb0, Cb, B, INC, HC, lane, true);
        dense_any(a.p.h1t, a.p.hb1, B, D, HC, HC, lane, true);
        float m1[3] = {0.f, 0.f, 0.f}, m2[3] = {0.f, 0.f, 0.f};
        for (int q = 0; q < nr; ++q) {
            const float* __restrict__ w = a.p.ws_rgb + (set * nr + (size_t)q) * (size_t)(3 * HC + 3);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float pre = 0.f;
                for (int i = 0; i < HC; ++i) pre = fmaf(D[i * 64 + lane], w[c * HC + i], pre);
                pre += w[3 * HC + c];
                const float pred = unerf_sigmoid(pre);
                m1[c] += pred;
                m2[c] += pred * pred;
            }
        }
        float vsum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            m1[c] /= (float)nr;
            m2[c] /= (float)nr;
            vsum += fmaxf(m2[c] - m1[c] * m1[c], 0.f);
        }
        if (valid) {
            a.density[n] = a.p.lap_mask_density ? mu * sel : mu;
            a.aux[n] = var_d;
            a.aux2[n] = vsum / 3.f;
            a.rgb[n * 3 + 0] = m1[0];
            a.rgb[n * 3 + 1] = m1[1];
            a.rgb[n * 3 + 2] = m1[2];
        }
        return;
    } else {
        const int passes = (MODE == UNERF_FIELD_MCDROPOUT && a.p.K > 0) ? a.p.K : 1;
        const uint32_t sidx = (uint32_t)((uint64_t)a.ray_offset * (uint64_t)a.S + (uint64_t)n);
        const uint32_t pre = unerf_mc_pre(unerf_mc_key(a.p.seed, 0u), sidx);
        for (int k = 0; k < passes; ++k) {
            const float* src = A;
            if (a.drop_sites & UNERF_DROP_TRUNK) {
                mask_any(A, B, H, lane, pre, k, 0u, a.keep_hi, a.drop_scale);
                src = B;
            }
            dense_any(a.p.w1t, a.p.b1, src, D, H, OUT1, lane, false);
            const float density = a.p.average_init_density * expf(D[lane]) * sel;
            const float beta = MODE == UNERF_FIELD_ACTIVE ? unerf_softplus(D[(G + 1) * 64 + lane]) + a.p.beta_min : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) Cb[e * 64 + lane] = sh[e];
            for (int g = 0; g < G; ++g) Cb[(16 + g) * 64 + lane] = D[(1 + g) * 64 + lane];
            if (a.drop_sites & UNERF_DROP_HEADIN) {   // Dropout on the head's inputs: the appearance block unfolded
                for (int e = 0; e < AD; ++e) Cb[(INC + e) * 64 + lane] = a.p.app_embed[e];
                mask_any(Cb, Cb, INC + AD, lane, pre, k, 3u, a.keep_hi, a.drop_scale);
                dense_any(a.p.h0_full_t, a.p.hb0_raw, Cb, B, INC + AD, HC, lane, true);
            } else {
                dense_any(a.p.h0t, a.p.hb0, Cb, B, INC, HC, lane, true);
            }
            if (a.drop_sites & UNERF_DROP_HEAD0) mask_any(B, B, HC, lane, pre, k, 2u, a.keep_hi, a.drop_scale);
            dense_any(a.p.h1t, a.p.hb1, B, D, HC, HC, lane, true);
            if (a.drop_sites & UNERF_DROP_HEAD1) mask_any(D, D, HC, lane, pre, k, 1u, a.keep_hi, a.drop_scale);
            dense_any(a.p.h2t, a.p.hb2, D, B, HC, 3, lane, false);
            if (valid) {
                const int64_t q = (int64_t)k * N + n;
                a.density[q] = density;
                a.rgb[q * 3 + 0] = unerf_sigmoid(B[lane]);
                a.rgb[q * 3 + 1] = unerf_sigmoid(B[64 + lane]);
                a.rgb[q * 3 + 2] = unerf_sigmoid(B[128 + lane]);
                if (MODE == UNERF_FIELD_ACTIVE) a.aux[n] = beta;
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// 5b. MFMA variant (ACTIVE, MCDROPOUT).  v_mfma_f32_32x32x2_f32 = exact fp32 FMA chains at the
// fp32 vector rate; what it buys is operand delivery: one 256-B A fragment (weights, LDS
// resident for the lifetime of a persistent workgroup) feeds 2048 MACs, where the VALU kernel
// needs the scalar cache to deliver a weight pair for every 128 MACs (56 KB of weights do not
// fit the scalar cache, so that kernel runs at the L2 scalar-fetch rate, rocprof r1_01).
//
// Mapping: a wave owns a tile of 32 samples.  Samples sit on the MFMA columns (lane & 31), layer
// units on the rows, so D of one layer is directly the B operand of the next (no LDS, no
// shuffles).  The two lane halves hold the two k-slices of every step; for the hash grid that
// means half h looks up levels 8h..8h+7 of the same 32 samples.  Fragment/bias layout:
// ops.py::pack_field_mfma (emulated bit for bit in tests/test_mfma_pack_cpu.py).
// --------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF_BIAS_OFF (160 * 64)
#define MF_H2_OFF (MF_BIAS_OFF + 7 * 32)

__device__ __forceinline__ f32x16 mf_bias(const float* lds, int k, int h) {
    const float4* b = reinterpret_cast<const float4*>(lds + MF_BIAS_OFF + (k * 2 + h) * 16);
    float4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
    f32x16 v = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
    return v;
}
// ReLU as a signed-integer max on the bit pattern: one v_max_i32 per element.  fmaxf() costs two
// VALU ops here (a canonicalising v_max before the real one, IEEE maxnum), and with ~10 VALU per
// MFMA the K-pass kernel is issue-bound (rocprof r1_04: MFMA busy 61 % + VALU busy 38 %).
// Negative floats (incl. -0.0) have negative bit patterns -> 0; non-negative ones pass unchanged.
__device__ __forceinline__ f32x16 mf_relu(f32x16 v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = __int_as_float(max(__float_as_int(v[r]), 0));
    return v;
}
// acc += W_frag(frag0 + r) x src[r], r = 0..15 (one 16-step K slab)
__device__ __forceinline__ f32x16 mf_slab(const float* lds, int frag0, int lane, const f32x16& src, f32x16 acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[(frag0 + r) * 64 + lane], src[r], acc, 0, 0, 0);
    return acc;
}
// inverted dropout on one accumulator block (units 32*blk + row(r,h)).  Register pairs (r, r+1) are
// units (u, u+1) = one mask word; the lane keeps its 8 words per block in `st` across the passes.
// (base_h: unerf_mc_base_h of THIS lane half h -- bit 1 of the pair index u >> 1 is h, and the word constants are taken at
// that bit cleared, so they are compile-time literals)
__device__ __forceinline__ void mf_mask_init(uint32_t (&st)[8], int blk, int h, uint32_t base_h, uint32_t stream_id) {
    (void)h;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = 2 * q;
        const uint32_t u0 = 32u * blk + (r & 3) + 8 * (r >> 2);   // the unit at h = 0; + 4 h: bit 1 of the pair index
        st[q] = unerf_mask_word0(base_h, stream_id, u0 >> 1);
    }
}
__device__ __forceinline__ void mf_mask_step(uint32_t (&st)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) st[q] = unerf_mask_step(st[q]);
}
// the eight words of one block at pass k, recomputed from the sample's base hash (k chained steps): used only by the
// non-default dropout site UNERF_DROP_HEAD0, which therefore costs the default configuration no registers
__device__ __forceinline__ void mf_mask_words_at(uint32_t (&st)[8], int blk, int h, uint32_t base_h, uint32_t stream_id, int k) {
    mf_mask_init(st, blk, h, base_h, stream_id);
    for (int q = 0; q < k; ++q) mf_mask_step(st);
}
__device__ __forceinline__ f32x16 mf_dropout(f32x16 v, const uint32_t (&st)[8], int32_t thr_hi, float scale) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        v[2 * q] = unerf_keep_lo(st[q], thr_hi) ? v[2 * q] * scale : 0.f;
        v[2 * q + 1] = unerf_keep_hi(st[q], thr_hi) ? v[2 * q + 1] * scale : 0.f;
    }
    return v;
}

// This half's 8 levels of the main grid for one sample -> the 16 k-steps of layer 0.
// Two batches of 4 levels: all 32 corner rows of a batch are requested back to back (uniform table
// base + 32-bit byte offset per lane), THEN blended.  Left to itself the compiler interleaves
// address math, 4-load groups and waits (about 20 dependent round trips per tile in the r1 ISA).
// PACKED picks the fp32x2 blend (fewer VALU issues, ~18 more VGPRs): right for the K-pass and Laplace
// kernels, which sit at 2 waves/SIMD anyway; the ACTIVE kernel keeps the scalar blend and its third wave
// (packed + 168-VGPR cap: 20 B of scratch, 22.2 vs 21.8 ms/frame).
// The 16 level records of a tcnn-layout grid, staged once per workgroup as five 16-entry LDS rows (scale, res,
// byte offset, size, dense): read per lane from the device array they were 40 vector loads per tile (the level
// index differs between the wave's halves, and loads after stores in the tile loop cannot be scalar).
#define MF_TL_WORDS 80
template <int RS = 3>   // log2 of the row size in bytes: 3 = fp32 rows, 2 = half2 rows (grid_half)
__device__ __forceinline__ void mf_stage_tcnn_levels(const FieldArgs& a, uint32_t* tl) {
    if (threadIdx.x < 16) {
        const unerf_tcnn_level lv = a.p.tcnn_levels[threadIdx.x];
        tl[threadIdx.x] = __float_as_uint(lv.scale);
        tl[16 + threadIdx.x] = lv.res;
        tl[32 + threadIdx.x] = lv.offset << RS;
        tl[48 + threadIdx.x] = lv.size;
        tl[64 + threadIdx.x] = lv.dense;
    }
}

// TCNN: 0 = nerfstudio's torch HashEncoding, 1 = tcnn layout on fp32 rows, 2 = tcnn layout in tcnn's own half arithmetic
// (unerf_field_params.grid_half: half2 rows, unerf_tcnn_blend_half).  TCNN == 2 also hands back the features as they
// come out of the blend -- `packed`[4 hb + q] = level 8 h + 4 hb + q as one half2 = operand quad [4 st .. 4 st + 3] of
// layer 0's k-step st: the f16 matrix kernels take them as they are (the floats returned hold the same values).
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
template <bool PACKED, int TCNN>
__device__ __forceinline__ f32x16 mf_gather_feats(const FieldArgs& a, float px, float py, float pz, int h, uint32_t mask,
                                                  const uint32_t* tl = nullptr, u32x8* packed = nullptr) {
    f32x16 feat;
    if (TCNN == 2) {
        const char* tbase = reinterpret_cast<const char*>(a.p.table);
        u32x8 pk;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            uint32_t cd[32];
            float wf[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lev = 8 * h + 4 * hb + q;
                uint32_t off[8];
                unerf_tcnn_offsets<2>(__uint_as_float(tl[lev]), tl[16 + lev], tl[32 + lev], tl[48 + lev], tl[64 + lev], px, py, pz,
                                      off, wf[3 * q], wf[3 * q + 1], wf[3 * q + 2]);
#pragma unroll
                for (int k = 0; k < 8; ++k) cd[8 * q + k] = *reinterpret_cast<const uint32_t*>(tbase + off[k]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t c8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) c8[k] = cd[8 * q + k];
                const uint32_t f = unerf_tcnn_blend_half(c8, wf[3 * q], wf[3 * q + 1], wf[3 * q + 2]);
                pk[4 * hb + q] = f;
                const float2 ff = unerf_h2_to_float2(f);
                feat[2 * (4 * hb + q)] = ff.x;
                feat[2 * (4 * hb + q) + 1] = ff.y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (packed) *packed = pk;
        return feat;
    }
    if (TCNN) {  // tcnn-layout grid: same batching (4 levels = 32 corner rows in flight), tcnn indexing + blend
        const char* tbase = reinterpret_cast<const char*>(a.p.table);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            float2 cd[32];
            float wf[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lev = 8 * h + 4 * hb + q;
                uint32_t off[8];
                unerf_tcnn_offsets<3>(__uint_as_float(tl[lev]), tl[16 + lev], tl[32 + lev], tl[48 + lev], tl[64 + lev], px, py, pz,
                                      off, wf[3 * q], wf[3 * q + 1], wf[3 * q + 2]);
#pragma unroll
                for (int k = 0; k < 8; ++k) cd[8 * q + k] = *reinterpret_cast<const float2*>(tbase + off[k]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float2 c8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) c8[k] = cd[8 * q + k];
                float2 f = unerf_tcnn_blend(c8, wf[3 * q], wf[3 * q + 1], wf[3 * q + 2]);
                feat[2 * (4 * hb + q)] = f.x;
                feat[2 * (4 * hb + q) + 1] = f.y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return feat;
    }
    const char* tbase = reinterpret_cast<const char*>(a.p.table);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        float2 cd[32];
        float of[12];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lev = 8 * h + 4 * hb + q;
            uint32_t off[8];
            // lev differs between the wave's halves: the table base stays uniform (SGPR) and the level
            // goes into the 32-bit lane offset (whole table < 2^32 bytes: L * 2^(log2T+3))
            unerf_hash_corners<true>(px, py, pz, a.p.scalings[lev], mask, off, of[3 * q], of[3 * q + 1], of[3 * q + 2],
                                     (uint32_t)lev << (a.p.log2T + 3));
#pragma unroll
            for (int k = 0; k < 8; ++k) cd[8 * q + k] = *reinterpret_cast<const float2*>(tbase + off[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float2 c8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) c8[k] = cd[8 * q + k];
            // UNERF_FIELD_BLEND_FMA (experiments; 0 ships): 1 = fused lerps, 2 = fused lerps in the SCALAR form
            // in every kernel (no v_pk_fma_f32 in the blend), 3 = as 1 with two wait states behind every level's blend
            float2 f = (PACKED && UNERF_FIELD_BLEND_FMA != 2 && !UNERF_FIELD_BLEND_SCALAR) ? unerf_blend8<(UNERF_FIELD_BLEND_FMA != 0)>(c8, of[3 * q], of[3 * q + 1], of[3 * q + 2])
                              : unerf_blend8_scalar<(UNERF_FIELD_BLEND_FMA != 0)>(c8, of[3 * q], of[3 * q + 1], of[3 * q + 2]);
#if UNERF_FIELD_BLEND_FMA == 3
            asm volatile("s_nop 1" : "+v"(f.x), "+v"(f.y));
#elif UNERF_FIELD_BLEND_FMA == 4
            asm volatile("s_nop 0" : "+v"(f.x), "+v"(f.y));
#elif UNERF_FIELD_BLEND_FMA == 5
            asm volatile("" : "+v"(f.x), "+v"(f.y));
#endif
            feat[2 * (4 * hb + q)] = f.x;
            feat[2 * (4 * hb + q) + 1] = f.y;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return feat;
}

// The same lookup as a ROLLING stream (torch layout, packed blend): four batches of two levels (16 corner rows each); the
// loads of batch b + 1 are issued BEFORE batch b is blended, and `filler` -- work that does not depend on the grid (the
// direction encoding and its two MFMAs) -- runs behind the first two batches' loads.  mf_gather_feats asks for 32 rows,
// blends them all, and only then asks for the next 32: between the two bursts the wave has nothing in flight, and at
// the start of a tile it sits through a full memory round trip with the SH arithmetic already behind it.  Same 64
// data registers (two 16-row buffers), same values: every level is blended exactly as before.
// MEASURED AND NOT ADOPTED (profiles/r5_exp_gather_pipe_ab.txt): K-pass "f16" field 24.9 -> 25.5 ms per frame,
// ACTIVE "f16" 9.82 -> 10.0, ACTIVE split 12.2 -> 12.9 (its 168-register budget spills 40 B) -- the K-pass kernel issues on
// 0.93 of its cycles (nothing to hide latency under), and the ACTIVE kernels pay for the second address set with their third
// wave's registers.  UNERF_GATHER_PIPE=1 builds it.
#ifndef UNERF_GATHER_PIPE
#define UNERF_GATHER_PIPE 0
#endif
template <typename Filler>
__device__ __forceinline__ f32x16 mf_gather_feats_pipe(const FieldArgs& a, float px, float py, float pz, int h, uint32_t mask,
                                                       Filler&& filler) {
    f32x16 feat;
    const char* tbase = reinterpret_cast<const char*>(a.p.table);
    float2 cd[2][16];
    float of[2][6];
    auto issue = [&](int b, int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int lev = 8 * h + 2 * b + q;
            uint32_t off[8];
            unerf_hash_corners<true>(px, py, pz, a.p.scalings[lev], mask, off, of[buf][3 * q], of[buf][3 * q + 1], of[buf][3 * q + 2],
                                     (uint32_t)lev << (a.p.log2T + 3));
#pragma unroll
            for (int k = 0; k < 8; ++k) cd[buf][8 * q + k] = *reinterpret_cast<const float2*>(tbase + off[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto blend = [&](int b, int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float2 c8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) c8[k] = cd[buf][8 * q + k];
            const float2 f = unerf_blend8<(UNERF_FIELD_BLEND_FMA != 0)>(c8, of[buf][3 * q], of[buf][3 * q + 1], of[buf][3 * q + 2]);
            feat[2 * (2 * b + q)] = f.x;
            feat[2 * (2 * b + q) + 1] = f.y;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    issue(0, 0);
    issue(1, 1);
    filler();
    __builtin_amdgcn_sched_barrier(0);
    blend(0, 0);
    issue(2, 0);
    blend(1, 1);
    issue(3, 1);
    blend(2, 0);
    blend(3, 1);
    return feat;
}

template <int MODE, bool FEAT_IN, int TCNN = 0>
// ACTIVE is bound by the gather (texture-address unit): three waves per SIMD (<= 168 VGPRs) hide more of
// its latency than two (measured 21.7 vs 23.7 ms/frame when a 176-VGPR build lost the third wave); the
// K-pass mode needs the registers instead.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MODE == UNERF_FIELD_ACTIVE && TCNN != 1) ? 3 : 2)))
void field_kernel_mfma(FieldArgs a, uint32_t num_tiles, FastDiv div_s) {
    extern __shared__ float lds[];
    {
        const float4* src = reinterpret_cast<const float4*>(a.p.mfma_blob);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < UNERF_MFMA_BLOB_FLOATS / 4; i += 256) dst[i] = src[i];
    }
    __shared__ uint32_t s_tl[TCNN ? MF_TL_WORDS : 1];
    if (TCNN) mf_stage_tcnn_levels<(TCNN == 2 ? 2 : 3)>(a, s_tl);
    __syncthreads();
    // wave index as a scalar: the tile walk and its divisions then run on the scalar unit
    const int lane_c = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane_c & 31, h = lane_c >> 5;
    const int64_t N = a.R * (int64_t)a.S;
    const uint32_t mask = (1u << a.p.log2T) - 1u;
    // XCD-aware persistent walk: blocks b and b+8 share an XCD (L2); give each XCD one contiguous
    // eighth of the tiles so neighbouring rays (same coarse hash cells) meet in the same L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const uint32_t tpx = (num_tiles + 7u) / 8u;  // num_tiles < 2^28 (the RNG counter bound in unerf_field_fwd)
    const uint32_t tile_end = (xcd + 1) * tpx < num_tiles ? (xcd + 1) * tpx : num_tiles;
    for (uint32_t tile = xcd * tpx + (uint32_t)slot * 4u + (uint32_t)wv; tile < tile_end; tile += (uint32_t)bpx * 4u) {
        // The LDS fragment reads are invariant across tiles; left alone, LICM hoists all 160 of
        // them into registers (490 VGPR+AGPR, scratch spills).  An opaque copy of the lane index
        // keeps them inside the iteration, where each read is consumed by the next MFMA.
        int lane = lane_c;
        asm volatile("" : "+v"(lane));
        // tile -> (block of 32 neighbouring rays, sample index s): the 32 columns of a tile are the SAME
        // sample slot of 32 adjacent pixels, which sit in the same or neighbouring grid cells, so a gather
        // instruction presents few distinct lines to the texture-address unit (it retires ~1 divergent
        // lane per clock: TA_BUSY 74 % with 32 consecutive samples of one ray per tile, rocprof r1_04)
        const TileSample ts = tile_sample(a, tile, div_s, j);
        const bool valid = ts.valid;
        const int64_t n = ts.n;
        const float dxr = ts.dx, dyr = ts.dy, dzr = ts.dz;
        float px = ts.px, py = ts.py, pz = ts.pz;
        const float sel = unerf_normalize_position(px, py, pz, a.box);

        // hash grid: this half's 8 levels -> 16 features = the 16 k-steps of layer 0
        f32x16 feat;
        if (FEAT_IN) {  // features were gathered level-major by field_gather_kernel: coalesced 8-B reads
            const float2* fp = reinterpret_cast<const float2*>(a.features);
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                float2 f = fp[(int64_t)(8 * h + l) * N + n];
                feat[2 * l] = f.x;
                feat[2 * l + 1] = f.y;
            }
        } else {
            feat = mf_gather_feats<(MODE != UNERF_FIELD_ACTIVE), TCNN>(a, px, py, pz, h, mask, s_tl);
        }
        // Colour layer 0 sees [geo(15) | SH(16)]; the SH half does not depend on the MC pass, so its
        // 16 MFMAs (+ bias) are done once per tile and every pass starts from that partial sum.
        f32x16 csh0 = mf_bias(lds, 3, h), csh1 = mf_bias(lds, 4, h);
        {
            float sh[16];
            float ux = (dxr + 1.f) / 2.f, uy = (dyr + 1.f) / 2.f, uz = (dzr + 1.f) / 2.f;
            if (a.p.sh_remap) {
                ux = ux * 2.f - 1.f;
                uy = uy * 2.f - 1.f;
                uz = uz * 2.f - 1.f;
            }
            unerf_sh16(ux, uy, uz, sh);
            // this half feeds components 8h..8h+7.  Bitwise per-half select: a plain
            // `h ? sh[8+k] : sh[k]` is rewritten into a lane-indexed load from a scratch copy of sh[]
            const uint32_t hm = 0u - (uint32_t)h;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v = __uint_as_float((__float_as_uint(sh[8 + q]) & hm) | (__float_as_uint(sh[q]) & ~hm));
                csh0 = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[(64 + 8 + q) * 64 + lane], v, csh0, 0, 0, 0);
                csh1 = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[(80 + 8 + q) * 64 + lane], v, csh1, 0, 0, 0);
            }
        }

        // layer 0: 32 -> 64, ReLU
        f32x16 hid0 = mf_relu(mf_slab(lds, 0, lane, feat, mf_bias(lds, 0, h)));
        f32x16 hid1 = mf_relu(mf_slab(lds, 16, lane, feat, mf_bias(lds, 1, h)));

        const int passes = (MODE == UNERF_FIELD_MCDROPOUT && a.p.K > 0) ? a.p.K : 1;
        const bool drop = (MODE == UNERF_FIELD_MCDROPOUT) && a.drop_on;
        const uint32_t sidx = (uint32_t)((uint64_t)a.ray_offset * (uint64_t)a.S + (uint64_t)n);
        uint32_t mk0[8], mk1[8], mk2[8], mk3[8];  // this lane's mask words: trunk blk 0/1, head blk 0/1
        const uint32_t base0 = drop ? unerf_mc_base_h(unerf_mc_pre(unerf_mc_key(a.p.seed, 0u), sidx), (uint32_t)h) : 0u;   // this lane half's
        if (drop) {
            mf_mask_init(mk0, 0, h, base0, 0u);
            mf_mask_init(mk1, 1, h, base0, 0u);
            mf_mask_init(mk2, 0, h, base0, 1u);
            mf_mask_init(mk3, 1, h, base0, 1u);
        }
        for (int k = 0; k < passes; ++k) {
            asm volatile("" : "+v"(lane));  // same reason: keep the fragment reads inside the pass
            f32x16 m0 = hid0, m1 = hid1;
            if (drop) {
                if (k > 0) {
                    mf_mask_step(mk0);
                    mf_mask_step(mk1);
                    mf_mask_step(mk2);
                    mf_mask_step(mk3);
                }
                if (a.drop_sites & UNERF_DROP_TRUNK) {
                    m0 = mf_dropout(hid0, mk0, a.keep_hi, a.drop_scale);
                    m1 = mf_dropout(hid1, mk1, a.keep_hi, a.drop_scale);
                }
            }
            // trunk out: 64 -> out1 (rows >= out1 are zero-padded): row 0 density, 1..15 geo, 16 beta
            f32x16 t = mf_bias(lds, 2, h);
            t = mf_slab(lds, 32, lane, m0, t);
            t = mf_slab(lds, 48, lane, m1, t);
            // colour 0: the geo rows of t (regs 0..7) on top of the per-tile SH partial sum, ReLU
            f32x16 c0 = csh0, c1 = csh1;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[(64 + q) * 64 + lane], t[q], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[(80 + q) * 64 + lane], t[q], c1, 0, 0, 0);
            }
            c0 = mf_relu(c0);
            c1 = mf_relu(c1);
            if (drop && (a.drop_sites & UNERF_DROP_HEAD0)) {   // rgb_dropout_layers contains 1: Dropout in front of Linear 1
                uint32_t mw[8];
                mf_mask_words_at(mw, 0, h, base0, 2u, k);
                c0 = mf_dropout(c0, mw, a.keep_hi, a.drop_scale);
                mf_mask_words_at(mw, 1, h, base0, 2u, k);
                c1 = mf_dropout(c1, mw, a.keep_hi, a.drop_scale);
            }
            // colour 1: 64 -> 64, ReLU
            f32x16 d0 = mf_bias(lds, 5, h), d1 = mf_bias(lds, 6, h);
            d0 = mf_slab(lds, 96, lane, c0, d0);
            d0 = mf_slab(lds, 112, lane, c1, d0);
            d1 = mf_slab(lds, 128, lane, c0, d1);
            d1 = mf_slab(lds, 144, lane, c1, d1);
            d0 = mf_relu(d0);
            d1 = mf_relu(d1);
            if (drop && (a.drop_sites & UNERF_DROP_HEAD1)) {
                d0 = mf_dropout(d0, mk2, a.keep_hi, a.drop_scale);
                d1 = mf_dropout(d1, mk3, a.keep_hi, a.drop_scale);
            }
            // colour 2: 64 -> 3 on the VALU: each half sums its 32 units, halves meet by one shuffle
            float rgbv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* w0p = lds + MF_H2_OFF + ((0 * 2 + h) * 3 + c) * 16;
                const float* w1p = lds + MF_H2_OFF + ((1 * 2 + h) * 3 + c) * 16;
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc = fmaf(d0[q], w0p[q], acc);
#pragma unroll
                for (int q = 0; q < 16; ++q) acc = fmaf(d1[q], w1p[q], acc);
                acc += __shfl_xor(acc, 32, 64);
                rgbv[c] = unerf_sigmoid(acc + lds[MF_H2_OFF + 192 + c]);
            }
            if (valid && h == 0) {
                const OutIndex q = out_index(a, k, ts);
                const float sigma = a.p.average_init_density * expf(t[0]) * sel;
                if (a.p.packed_out) {   // uniform
                    store_packed(a, k, ts.n, sigma, rgbv[0], rgbv[1], rgbv[2]);
                } else {
                    a.density[q.dens] = sigma;
                    a.rgb[q.rgb] = rgbv[0];
                    a.rgb[q.rgb + q.rgb_stride] = rgbv[1];
                    a.rgb[q.rgb + 2 * q.rgb_stride] = rgbv[2];
                }
                if (MODE == UNERF_FIELD_ACTIVE) a.aux[q.aux] = unerf_softplus(t[8]) + a.p.beta_min;
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// 5b''. Split-f16 variant of 5b (ACTIVE, MCDROPOUT): the same network, same tile mapping, same gathers and
// epilogues, but the dense layers run on v_mfma_f32_32x32x16_f16 with every fp32 operand carried as two
// halves (hi = f16(x), lo = f16(x - hi): 22 mantissa bits) and hi*hi + hi*lo + lo*hi accumulated in fp32.
// Why: the fp32-input MFMA runs at the fp32 VECTOR rate and does not overlap with VALU work (rocprof
// r1_05: MFMA-busy 58 % + VALU-busy 35 % = the elapsed cycles), so the exact-fp32 kernel spends 10.2 k of
// its 17.4 k cycles per tile in 160 MFMAs.  The f16 matrix pipe is 16x faster per k-step; three products
// per step leave 60 MFMAs x ~36 cycles = 2.2 k cycles, plus ~3 VALU per activation element for the split.
// Accuracy: dropped lo*lo term and the rounding of lo are each <= 2^-22 relative per product, i.e. the
// result is fp32-equivalent (|d rgb| ~ 1e-7; tests/test_gpu_nerf_kernels.py bounds it against the exact
// kernel, the e2e PSNR / AUSE gates are unchanged).  f16 subnormal inputs are honoured by the MFMA
// (benchmarks/mfma_f16_probe.hip); operands must stay below 65504 in magnitude (checked for the weights at
// pack time; activations of a trained nerfacto field are orders of magnitude smaller).
// Operand order: accumulator registers 8s..8s+7 of a lane in half g are layer units
// 16s + 4g + (e&3) + 8(e>>2) -- the k order of the next layer's B operand, hence the order
// ops.pack_field_mfma16 packs the A slabs in.  LDS blob: 20 slabs x (hi|lo) x 64 lanes x 16 B = 40 KiB, then
// the same bias rows / rgb layer as the fp32 blob (same offsets, same total size).
// --------------------------------------------------------------------------------------
// MC-dropout on PACKED f16 operands.  The inverted-dropout scale 1/(1-p) is folded into the weights of the layer
// that follows (ops.pack_field_mfma16(drop_scale=...)), so a dropped unit is just a zeroed operand, and an operand
// pair (units 2j, 2j+1 = the two halves of one register of the hi and of the lo quad) is zeroed by ONE and with a
// mask whose halves are 0xFFFF / 0.  That mask costs two packed instructions per RNG word: the saturating signed
// difference half - thr_s is negative exactly when the unit is kept (v_pk_sub_i16 clamp), and an arithmetic shift
// by 15 spreads the sign over the half (v_pk_ashrrev_i16).  Per unit: 1 (mask) + 1 (two ands) instructions, against
// compare + select on the fp32 accumulator (2) in round 1 -- and the trunk layer's input is pass-invariant, so its
// operand split (1.5 per unit) moves out of the K loop altogether.
__device__ __forceinline__ f32x16 mf_dropout_keep(f32x16 v, const uint32_t (&st)[8], int32_t thr_hi) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        v[2 * q] = unerf_keep_lo(st[q], thr_hi) ? v[2 * q] : 0.f;
        v[2 * q + 1] = unerf_keep_hi(st[q], thr_hi) ? v[2 * q + 1] : 0.f;
    }
    return v;
}
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mf16_keep_mask(uint32_t word, uint32_t thr_pk) {
    const i16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(i16x2, word), __builtin_bit_cast(i16x2, thr_pk));
    return __builtin_bit_cast(uint32_t, d >> (short)15);
}
// sigmoid on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1e-7 relative each) instead of the ~25-instruction
// exact expf + IEEE division
__device__ __forceinline__ float mf_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
// softplus on the hardware exp / log (density_activation = "softplus", laplace_model.py:151): log1p by its series
// where 1 + e would round e away
__device__ __forceinline__ float mf_softplus_fast(float x) {
    const float e = __expf(x);
    const float small = e * (1.f - e * (0.5f - e * (1.f / 3.f)));
    return x > 20.f ? x : (e < 1e-3f ? small : __logf(1.f + e));
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Two elements per step: hi = v_cvt_pk_f16_f32(x0, x1) (round to nearest even), then each residual
// x - float(hi) is formed and rounded to f16 inside ONE mixed-precision fma (v_fma_mixlo/mixhi_f16: f16 and f32
// sources, exact internal product and sum): 1.5 VALU per element instead of the ~3 of convert-back / subtract /
// convert (the kernels are VALU-issue-bound once the matrix work is on the f16 pipe).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// F1 (every helper below, and the kernels): the REFERENCE-PRECISION form, `precision = "f16"` / unerf_field_params.f16_single.
// One f16 product per MAC with fp32 accumulation: operands rounded to f16 once (v_cvt_pk_f16_f32 for the activations,
// the hi halves of the packed weights), no lo halves anywhere -- the arithmetic of torch.autocast(float16) Linear layers
// (forced at eval by mcdropout_models.py:86-92) and at least that of tiny-cuda-nn's FullyFusedMLP (fp16 weights,
// activations AND accumulators; the reference's default implementation="tcnn", activenerfacto_field.py:89).  A third of
// the split form's MFMAs and none of its 1.5-instruction-per-element residual splits.
template <bool F1 = false>
__device__ __forceinline__ void mf16_split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
    u32x4 hv, lv;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f16x2 hh = {(_Float16)x[2 * p], (_Float16)x[2 * p + 1]};
        hv[p] = lv[p] = __builtin_bit_cast(uint32_t, hh);   // F1: lo is never read in this form
    }
    if (!F1) {
        // The eight residuals of a k-step operand in ONE assembly statement that ends in two wait states.  An MFMA must not
        // read a VGPR within two wait states of a VALU write to it (and a VALU instruction not within one of a half-register
        // write); the compiler places those for the instructions it knows, and does not look inside inline assembly: written
        // as one statement per instruction (rounds 2 - 4) the listing had MFMAs ONE instruction behind the v_fma_mixhi_f16
        // that completes their B operand.  No run of the test suite ever differed because of it -- the other wave of the
        // SIMD usually separates the two -- but nothing guaranteed that.
        uint32_t l0, l1, l2, l3;
        asm("v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "s_nop 1"
            : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
            : "v"(hv[0]), "v"(hv[1]), "v"(hv[2]), "v"(hv[3]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]),
              "v"(x[6]), "v"(x[7]));
        lv[0] = l0; lv[1] = l1; lv[2] = l2; lv[3] = l3;
    }
    hi = __builtin_bit_cast(f16x8, hv);
    lo = __builtin_bit_cast(f16x8, lv);
}
template <bool F1 = false>
__device__ __forceinline__ void mf16_split(const f32x16& v, int s, f16x8& hi, f16x8& lo) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = v[8 * s + e];
    mf16_split8<F1>(x, hi, lo);
}
// F1: ReLU AFTER the conversion, on the packed halves: v_pk_max_i16(bits, 0) maps every negative f16 (sign bit = negative
// int16, -0 included) to +0 and leaves the others alone -- relu(cvt(x)) = cvt(relu(x)) exactly, at one instruction per
// TWO units instead of one v_max_i32 per unit on the fp32 accumulators.
__device__ __forceinline__ void mf16_split_relu(const f32x16& v, int s, f16x8& hi) {
    u32x4 hv;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        // (a VECTOR fptrunc: from two scalar conversions behind an integer max the compiler converts the halves one by one
        // and packs them with a v_perm -- three instructions where v_cvt_pk_f16_f32 is one.  Not inline asm: the
        // compiler has to see this read of MFMA results to place the wait states between the two.)
        const unerf_v2f pr = {v[8 * s + 2 * p], v[8 * s + 2 * p + 1]};
        const uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, f16x2));
        const i16x2 z = {0, 0};
        hv[p] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, w), z));
    }
    hi = __builtin_bit_cast(f16x8, hv);
}
// acc += W(slab) x B: small terms first
template <bool F1 = false>
__device__ __forceinline__ f32x16 mf16_mac(const float* lds, int slab, int lane, const f16x8& bhi, const f16x8& blo,
                                           f32x16 acc) {
    const f16x8 ahi = *reinterpret_cast<const f16x8*>(lds + slab * 512 + lane * 4);
    if (F1) return __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
    const f16x8 alo = *reinterpret_cast<const f16x8*>(lds + slab * 512 + 256 + lane * 4);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
    return acc;
}
// two row blocks against one B operand, the two accumulator chains interleaved.  (Round 2 read benchmarks/mfma_data_probe.hip
// as "an MFMA whose SrcC is the result of the MFMA right before it issues 16 cycles late"; round 6's hand-written sweep,
// benchmarks/issue_sweep_probe.hip, measures 32.3 cycles per MFMA on ONE chain as on two at every occupancy.  The interleave
// costs nothing and stays.)
// LOZ: the activations are half values already (tcnn's half grid features): their lo residual is zero and the W_hi x lo
// product is dropped
template <bool F1 = false, bool LOZ = false>
__device__ __forceinline__ void mf16_mac2(const float* lds, int slab_a, int slab_b, int lane, const f16x8& bhi, const f16x8& blo,
                                          f32x16& o0, f32x16& o1) {
    const f16x8 ahi0 = *reinterpret_cast<const f16x8*>(lds + slab_a * 512 + lane * 4);
    const f16x8 ahi1 = *reinterpret_cast<const f16x8*>(lds + slab_b * 512 + lane * 4);
    if (F1) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi0, bhi, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi1, bhi, o1, 0, 0, 0);
        return;
    }
    const f16x8 alo0 = *reinterpret_cast<const f16x8*>(lds + slab_a * 512 + 256 + lane * 4);
    const f16x8 alo1 = *reinterpret_cast<const f16x8*>(lds + slab_b * 512 + 256 + lane * 4);
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo0, bhi, o0, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo1, bhi, o1, 0, 0, 0);
    if (!LOZ) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi0, blo, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi1, blo, o1, 0, 0, 0);
    }
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi0, bhi, o0, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi1, bhi, o1, 0, 0, 0);
}
// layer 0's operand quad of k-step st: split (or, F1, rounded) from the fp32 features, or -- TCNN == 2 -- the packed half
// features of mf_gather_feats as they are (no conversion, no residual: mf16_mac2<F1, true>)
template <int TCNN, bool F1>
__device__ __forceinline__ void mf16_feat_operand(const f32x16& feat, const u32x8& pk, int st, f16x8& bhi, f16x8& blo) {
    if (TCNN == 2) {
        const u32x4 q = {pk[4 * st], pk[4 * st + 1], pk[4 * st + 2], pk[4 * st + 3]};
        bhi = blo = __builtin_bit_cast(f16x8, q);
    } else {
        mf16_split<F1>(feat, st, bhi, blo);
    }
}
// Folded slab (ops.pack_field_mfma16: fold_trunk) of a layer with <= 16 output rows: the slab's second operand holds
// rows 0..15 = W_hi and rows 16..31 = W_lo, so one MFMA against the hi halves of the activations produces W_hi a_hi in
// accumulator registers 0..7 and W_lo a_hi in registers 8..15 of the same lane (row(r + 8) = row(r) + 16); the first
// operand (W_hi, rows 16..31 zero) takes the lo halves.  Two MFMAs per k-step instead of three; the caller adds
// registers r + 8 onto r once per layer (mf16_fold_rows).
__device__ __forceinline__ f32x16 mf16_mac_fold_ops(const f16x8& ahi, const f16x8& amix, const f16x8& bhi, const f16x8& blo,
                                                    f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(amix, bhi, acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x16 mf16_fold_rows(f32x16 acc) {
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] += acc[r + 8];
    return acc;
}
// a 64-wide layer input held as two accumulator blocks (units 0..31 in v0, 32..63 in v1) against the
// 4 k-steps x NB row blocks of slabs starting at `slab0` (slab = slab0 + NB*step + block)
template <int NB, bool F1 = false, bool RELU_PACKED = false>
__device__ __forceinline__ void mf16_layer64(const float* lds, int slab0, int lane, const f32x16& v0, const f32x16& v1,
                                             f32x16& o0, f32x16& o1) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f16x8 bhi, blo;
        if (RELU_PACKED) {   // (F1 only) the input's ReLU rides on the converted operands
            mf16_split_relu(s < 2 ? v0 : v1, s & 1, bhi);
            blo = bhi;
        } else {
            mf16_split<F1>(s < 2 ? v0 : v1, s & 1, bhi, blo);
        }
        if (NB == 2) mf16_mac2<F1>(lds, slab0 + NB * s, slab0 + NB * s + 1, lane, bhi, blo, o0, o1);
        else o0 = mf16_mac<F1>(lds, slab0 + NB * s, lane, bhi, blo, o0);
    }
}

// split-f16 blob (ops.pack_field_mfma16): 20 operand slabs, then the bias rows and the fp32 rgb layer at the
// offsets of the fp32 blob (same total size)
#define mf16_bias mf_bias
// and both operand quads of k-step `st` (accumulator registers 8 st .. 8 st + 7 = mask words 4 st .. 4 st + 3 of
// the block's eight) with their keep masks
template <bool F1 = false>
__device__ __forceinline__ void mf16_apply_masks(f16x8& hi, f16x8& lo, const uint32_t (&words)[8], int st, uint32_t thr_pk) {
    u32x4 m;
#pragma unroll
    for (int p = 0; p < 4; ++p) m[p] = mf16_keep_mask(words[4 * st + p], thr_pk);
    hi = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4, hi) & m);
    if (!F1) lo = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4, lo) & m);
}

// UNERF_KPASS_FILL: the keep test of eight words in its two halves, so that the halves can sit in different MFMA shadows
__device__ __forceinline__ void mf16_keep_sub(const uint32_t (&w)[8], uint32_t thr_pk, uint32_t (&d)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
        d[q] = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(i16x2, w[q]), __builtin_bit_cast(i16x2, thr_pk)));
}
__device__ __forceinline__ void mf16_keep_sign(uint32_t (&d)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, d[q]) >> (short)15);
}
// the eight values exist HERE: an empty volatile statement the optimiser can neither sink into the loop latch (where it put the
// mask steps, whose results only the next pass reads) nor hoist
__device__ __forceinline__ void mf_pin8(uint32_t (&x)[8]) {
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
// scheduling fence that LDS reads and scalar instructions may cross (the next layer's operand reads go up, nothing else moves)
#define MF_FENCE() __builtin_amdgcn_sched_barrier(0x0104)

// SITES = false: the reference's default Dropout placement (trunk + last head layer), every site test a compile-time
// constant.  SITES = true: any other unerf_field_params.drop_sites (run-time site tests, and the words of the
// UNERF_DROP_HEAD0 site recomputed per pass).  A separate instantiation: as run-time branches of the default kernel
// they cost that kernel 50 VGPRs and 84 bytes of scratch (K = 8 field kernel 50 -> 55.6 ms).
// DROP: masks are generated (MCDROPOUT with K > 0 and p > 0).  A compile-time flag: as a run-time (uniform) flag every
// k-step of the masked layers carried a branch and the operand quads were copied to merge the two paths.
template <int MODE, int TCNN, bool SITES = false, bool DROP = false, bool F1 = false>
// (the single-product K-pass kernel at 3 waves per SIMD -- 168 VGPRs, 96 B of scratch, trunk operands re-read from LDS --
// was measured and lost: 4.84 vs 3.89 ms per launch, same box, profiles/r3_exp_f16_single_occ3.json)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MODE == UNERF_FIELD_ACTIVE && TCNN != 1) ? 3 : 2)))
void field_kernel_mfma16(FieldArgs a, uint32_t num_tiles, FastDiv div_s) {
    extern __shared__ float lds[];
    {
        const float4* src = reinterpret_cast<const float4*>(a.p.mfma16_blob);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < (F1 ? UNERF_MFMA16_BLOB_FLOATS : UNERF_MFMA_BLOB_FLOATS) / 4; i += 256) dst[i] = src[i];
    }
    __shared__ uint32_t s_tl[TCNN ? MF_TL_WORDS : 1];
    if (TCNN) mf_stage_tcnn_levels<(TCNN == 2 ? 2 : 3)>(a, s_tl);
    __syncthreads();
    const int lane_c = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane_c & 31, h = lane_c >> 5;
    const int64_t N = a.R * (int64_t)a.S;
    const uint32_t mask = (1u << a.p.log2T) - 1u;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const uint32_t tpx = (num_tiles + 7u) / 8u;
    const uint32_t tile_end = (xcd + 1) * tpx < num_tiles ? (xcd + 1) * tpx : num_tiles;
    bool f1_bad = false;   // an f16 operand overflowed: F1 -- an output pre-activation of this lane was inf / NaN (see the
                           // epilogue); split form -- a colour layer's pre-activations were NaN (see colour 0 below)
    for (uint32_t tile = xcd * tpx + (uint32_t)slot * 4u + (uint32_t)wv; tile < tile_end; tile += (uint32_t)bpx * 4u) {
        int lane = lane_c;  // opaque per iteration: keeps the (tile-invariant) LDS operand reads inside the loop
        asm volatile("" : "+v"(lane));
        const TileSample ts = tile_sample(a, tile, div_s, j);
        const bool valid = ts.valid;
        const int64_t n = ts.n;
        const float dxr = ts.dx, dyr = ts.dy, dzr = ts.dz;
        float px = ts.px, py = ts.py, pz = ts.pz;
        const float sel = unerf_normalize_position(px, py, pz, a.box);
        // packed fp32x2 blend: this kernel has the registers for it (123 VGPRs without) in every mode
        // colour layer 0, SH half (pass-invariant): components 8h..8h+7 of this lane half, one k-step -- as a closure, so
        // that the torch-layout kernels can run it behind the first grid loads of the tile (mf_gather_feats_pipe)
        f32x16 csh0 = mf16_bias(lds, 3, h), csh1 = mf16_bias(lds, 4, h);
        auto sh_layer = [&]() {
            float sh[16];
            float ux = (dxr + 1.f) / 2.f, uy = (dyr + 1.f) / 2.f, uz = (dzr + 1.f) / 2.f;
            if (a.p.sh_remap) {
                ux = ux * 2.f - 1.f;
                uy = uy * 2.f - 1.f;
                uz = uz * 2.f - 1.f;
            }
            unerf_sh16(ux, uy, uz, sh);
            const uint32_t hm = 0u - (uint32_t)h;
            float mine[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                mine[q] = __uint_as_float((__float_as_uint(sh[8 + q]) & hm) | (__float_as_uint(sh[q]) & ~hm));
            f16x8 bhi, blo;
            mf16_split8<F1>(mine, bhi, blo);
            mf16_mac2<F1>(lds, 10, 11, lane, bhi, blo, csh0, csh1);
        };
        u32x8 feat_pk;
        f32x16 feat;
        if constexpr (TCNN == 0 && UNERF_GATHER_PIPE != 0) {
            feat = mf_gather_feats_pipe(a, px, py, pz, h, mask, sh_layer);
        } else {
            feat = mf_gather_feats<true, TCNN>(a, px, py, pz, h, mask, s_tl, &feat_pk);
            sh_layer();
        }

        // layer 0: 32 -> 64 (this half's 16 features = two k-steps), ReLU; the 64 hidden units are the trunk
        // layer's four k-steps and do not depend on the MC pass: they are split into f16 operand quads ONCE
        f16x8 hhi[4], hlo[4];
        {
            f32x16 hid0 = mf16_bias(lds, 0, h), hid1 = mf16_bias(lds, 1, h);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f16x8 bhi, blo;
                mf16_feat_operand<TCNN, F1>(feat, feat_pk, st, bhi, blo);
                mf16_mac2<F1, TCNN == 2>(lds, 2 * st, 2 * st + 1, lane, bhi, blo, hid0, hid1);
            }
            if constexpr (F1) {   // ReLU on the packed halves (relu(cvt(x)) = cvt(relu(x)): mf16_split_relu), half the instructions
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    mf16_split_relu(st < 2 ? hid0 : hid1, st & 1, hhi[st]);
                    hlo[st] = hhi[st];
                }
            } else {
                hid0 = mf_relu(hid0);
                hid1 = mf_relu(hid1);
#pragma unroll
                for (int st = 0; st < 4; ++st) mf16_split<F1>(st < 2 ? hid0 : hid1, st & 1, hhi[st], hlo[st]);
            }
        }

        const int passes = (MODE == UNERF_FIELD_MCDROPOUT && a.p.K > 0) ? a.p.K : 1;
        // variants: the trunk-out operands (4 k-steps x 2 quads = 32 VGPRs) kept in registers across the passes; the
        // 16-row trunk-out layer of MCDROPOUT folded into two MFMAs per k-step
        constexpr bool TRUNK_RESIDENT = UNERF_TRUNK_RESIDENT && MODE == UNERF_FIELD_MCDROPOUT && DROP && !SITES;
        // F1: the first operand of a (folded or plain) trunk slab is W_hi, which is all the single-product form reads
        constexpr bool FOLD = UNERF_TRUNK_FOLD && MODE == UNERF_FIELD_MCDROPOUT && !F1;
        f16x8 ta0[4], ta1[4];   // first / second operand of trunk slab 4 + st
        if (TRUNK_RESIDENT) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                ta0[st] = *reinterpret_cast<const f16x8*>(lds + (4 + st) * 512 + lane * 4);
                if (!F1) ta1[st] = *reinterpret_cast<const f16x8*>(lds + (4 + st) * 512 + 256 + lane * 4);
            }
        }
        constexpr bool drop = DROP;   // host: a.drop_on
        const uint32_t sidx = (uint32_t)((uint64_t)a.ray_offset * (uint64_t)a.S + (uint64_t)n);
        uint32_t mk0[8], mk1[8], mk2[8], mk3[8];
        uint32_t base0_h0 = 0u;   // SITES only: the sample's base hash stays live across the passes
        const bool drop_trunk = SITES ? (drop && (a.drop_sites & UNERF_DROP_TRUNK)) : drop;
        const bool drop_head1 = SITES ? (drop && (a.drop_sites & UNERF_DROP_HEAD1)) : drop;
        if (drop) {
            const uint32_t base0 = unerf_mc_base_h(unerf_mc_pre(unerf_mc_key(a.p.seed, 0u), sidx), (uint32_t)h);   // this lane half's
            if (SITES) base0_h0 = base0;
            mf_mask_init(mk0, 0, h, base0, 0u);
            mf_mask_init(mk1, 1, h, base0, 0u);
            mf_mask_init(mk2, 0, h, base0, 1u);
            mf_mask_init(mk3, 1, h, base0, 1u);
        }
        // UNERF_KPASS_FILL (F1, default sites): AND masks of the trunk for the pass about to run (am_t*), of the head for the
        // running pass (am_h*).  The words are stepped and tested BEHIND the matrix instructions of a pass instead of in front
        // of them: an MFMA holds the SIMD's issue for ~10 of its 32 cycles, five or six independent VALU instructions ride in
        // its shadow for nothing (benchmarks/issue_sweep_probe.hip), and the mask arithmetic -- 96 of a pass' 216 VALU
        // instructions -- depends on nothing the pass computes.  Same words, same tests, same bits.
        constexpr bool FILL = UNERF_KPASS_FILL && F1 && DROP && !SITES && MODE == UNERF_FIELD_MCDROPOUT;
        uint32_t am_t0[8], am_t1[8], am_h0[8], am_h1[8];
        if (FILL) {
            mf16_keep_sub(mk0, a.keep_pk, am_t0);
            mf16_keep_sub(mk1, a.keep_pk, am_t1);
            mf16_keep_sign(am_t0);
            mf16_keep_sign(am_t1);
        }
        for (int k = 0; k < passes; ++k) {
            asm volatile("" : "+v"(lane));
            if (!FILL && drop && k > 0) {
                mf_mask_step(mk0);
                mf_mask_step(mk1);
                mf_mask_step(mk2);
                mf_mask_step(mk3);
            }
            // [probe:kpass-pass-start]
            // Wave priority: low through the matrix layers of a pass, high from the rgb layer to the end of the pass --
            // and, after the last pass, through the next tile's gathers.  The tail (packed-fma chains, the half exchange,
            // exp / rcp, stores) and the gather prologue are short instruction streams that wait on latencies; letting
            // them go first when both waves of a SIMD are ready takes 4.2 % off the K = 8 kernel (same box, six
            // placements tried: benchmarks/multi_ab.sh, profiles/r2_exp_setprio.json); high priority for the matrix
            // layers instead gives 1.3 %, for the prologue alone nothing.
            __builtin_amdgcn_s_setprio(0);
            // trunk out: 64 -> out1 rows (row 0 density, 1..15 geo, 16 beta) from the (masked) hidden operands
            f32x16 t = mf16_bias(lds, 2, h);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16x8 bhi = hhi[st], blo = hlo[st];
                if (FILL) {
                    const uint32_t (&am)[8] = st < 2 ? am_t0 : am_t1;
                    const u32x4 m = {am[4 * (st & 1)], am[4 * (st & 1) + 1], am[4 * (st & 1) + 2], am[4 * (st & 1) + 3]};
                    bhi = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4, bhi) & m);
                } else if (drop_trunk) mf16_apply_masks<F1>(bhi, blo, st < 2 ? mk0 : mk1, st & 1, a.keep_pk);
                f16x8 a0, a1;
                if (TRUNK_RESIDENT) {
                    a0 = ta0[st];
                    a1 = F1 ? ta0[st] : ta1[st];
                } else {
                    a0 = *reinterpret_cast<const f16x8*>(lds + (4 + st) * 512 + lane * 4);
                    a1 = F1 ? a0 : *reinterpret_cast<const f16x8*>(lds + (4 + st) * 512 + 256 + lane * 4);
                }
                if (F1) {
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bhi, t, 0, 0, 0);
                } else if (FOLD) {
                    t = mf16_mac_fold_ops(a0, a1, bhi, blo, t);
                } else {   // a0 = W_hi, a1 = W_lo: small terms first
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bhi, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, blo, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bhi, t, 0, 0, 0);
                }
            }
            if (FOLD) t = mf16_fold_rows(t);
            if (FILL) {   // behind the trunk's four MFMAs: this pass' head words tested (first half), then stepped for the next pass
                MF_FENCE();
                mf16_keep_sub(mk2, a.keep_pk, am_h0);
                mf16_keep_sub(mk3, a.keep_pk, am_h1);
                mf_mask_step(mk2);
                mf_mask_step(mk3);
                mf_pin8(am_h0); mf_pin8(am_h1); mf_pin8(mk2); mf_pin8(mk3);
                MF_FENCE();
            }
            // colour 0: geo rows of t (registers 0..7 = one k-step) on top of the SH partial sum, ReLU
            f32x16 c0 = csh0, c1 = csh1;
            {
                f16x8 bhi, blo;
                mf16_split<F1>(t, 0, bhi, blo);
                mf16_mac2<F1>(lds, 8, 9, lane, bhi, blo, c0, c1);
            }
            if (!F1) {   // F1: ReLU on the packed f16 operands instead (mf16_split_relu: half the instructions)
                // Split form: an activation beyond 65504 is carried as hi = +inf, lo = -inf, and EVERY unit of the layer it
                // feeds becomes inf - inf = NaN.  Trunk overflows reach the density logit as NaN and are caught by the
                // composite kernels; behind a ReLU they would not be -- the integer maximum below maps a NaN whose sign bit
                // is set to 0, and the layers after it then see a plausible all-zero hidden vector.  So one accumulator of
                // each colour layer is tested before its ReLU (all 64 are NaN or none): two compares per pass.
                f1_bad |= c0[0] != c0[0];
                c0 = mf_relu(c0);
                c1 = mf_relu(c1);
            }
            if (FILL) {   // behind colour 0's two MFMAs: the head masks' second half
                MF_FENCE();
                mf16_keep_sign(am_h0);
                mf16_keep_sign(am_h1);
                mf_pin8(am_h0); mf_pin8(am_h1);
                MF_FENCE();
            }
            // colour 1: 64 -> 64, ReLU
            f32x16 d0 = mf16_bias(lds, 5, h), d1 = mf16_bias(lds, 6, h);
            if (SITES && drop && (a.drop_sites & UNERF_DROP_HEAD0)) {   // rgb_dropout_layers contains 1 (non-default): masks on c
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    f16x8 bhi, blo;
                    if (F1) mf16_split_relu(st < 2 ? c0 : c1, st & 1, bhi);
                    else mf16_split<F1>(st < 2 ? c0 : c1, st & 1, bhi, blo);
                    uint32_t mw[8];
                    mf_mask_words_at(mw, st >> 1, h, base0_h0, 2u, k);
                    mf16_apply_masks<F1>(bhi, blo, mw, st & 1, a.keep_pk);
                    mf16_mac2<F1>(lds, 12 + 2 * st, 12 + 2 * st + 1, lane, bhi, blo, d0, d1);
                }
            } else {
                mf16_layer64<2, F1, F1>(lds, 12, lane, c0, c1, d0, d1);
            }
            // (F1 with this layer as four more k-steps on the matrix pipe -- 16 conversions + 16 packed ReLUs + 48 packed-mask
            // instructions + 4 MFMAs instead of the 154 instructions below -- was built and measured: 3.68 vs 3.89 ms per
            // launch, but the f16 rounding of the last layer's operands moved one MC-dropout AUSE figure past its 1e-3
            // gate; profiles/r3_exp_f16_rgb_on_mfma.json, DESIGN.md 4.5.  The colour layer stays fp32 in every form.)
            float o[3];
            if constexpr (F1) {
                // colour 2: 64 -> 3 as four more k-steps on the matrix pipe (rows 0..2 of one 32-row block; slabs behind the
                // fp32 tail of the blob), its operands = the hidden units rounded to f16 -- which is what the reference's
                // Linear computes under its forced autocast (mcdropout_models.py:86-92: f16 inputs and weights, fp32
                // accumulate) and what tcnn's FullyFusedMLP does.  ReLU and the dropout masks ride on the packed halves like
                // the trunk's: convert 0.5 + ReLU 0.5 + mask 1.5 instructions per unit instead of ReLU 1 + compare 1 +
                // select 1 on the fp32 accumulators, and 4 MFMAs (128 issue cycles) instead of 48 packed FMAs + the half
                // exchange.  (Round 3 measured this form at -5 % and dropped it over ONE AUSE figure at 1.04e-3 -- a gate
                // that the reference's own two arithmetics miss by more on that target)
                if (FILL) {   // behind colour 1's last MFMAs: the trunk words stepped for the next pass
                    MF_FENCE();
                    mf_mask_step(mk0);
                    mf_mask_step(mk1);
                    mf_pin8(mk0); mf_pin8(mk1);
                    MF_FENCE();
                }
                __builtin_amdgcn_s_setprio(1);
                f32x16 o4;
#pragma unroll
                for (int r = 0; r < 16; ++r) o4[r] = 0.f;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    f16x8 bhi, blo;
                    mf16_split_relu(st < 2 ? d0 : d1, st & 1, bhi);
                    blo = bhi;
                    if (FILL) {
                        const uint32_t (&am)[8] = st < 2 ? am_h0 : am_h1;
                        const u32x4 m = {am[4 * (st & 1)], am[4 * (st & 1) + 1], am[4 * (st & 1) + 2], am[4 * (st & 1) + 3]};
                        bhi = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4, bhi) & m);
                    } else if (drop_head1) mf16_apply_masks<true>(bhi, blo, st < 2 ? mk2 : mk3, st & 1, a.keep_pk);
                    const f16x8 aw = *reinterpret_cast<const f16x8*>(lds + UNERF_MFMA_BLOB_FLOATS + st * 256 + lane * 4);
                    o4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, bhi, o4, 0, 0, 0);
                }
                if (FILL) {   // behind colour 2's four MFMAs: the next pass' trunk masks
                    MF_FENCE();
                    mf16_keep_sub(mk0, a.keep_pk, am_t0);
                    mf16_keep_sub(mk1, a.keep_pk, am_t1);
                    mf16_keep_sign(am_t0);
                    mf16_keep_sign(am_t1);
                    mf_pin8(am_t0); mf_pin8(am_t1);
                    MF_FENCE();
                }
                // rows 0..2 = registers 0..2 of the h = 0 half: one v_permlane32_swap each hands them to both halves
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o4[c]), __float_as_uint(o4[c]), false, false);
                    o[c] = __uint_as_float(sw[0]) + lds[MF_H2_OFF + 192 + c];
                }
            } else {
            if (!F1) f1_bad |= d0[0] != d0[0];
            d0 = mf_relu(d0);
            d1 = mf_relu(d1);
            if (drop_head1) {   // masks on the fp32 accumulators: one half-word compare + one select per unit
                d0 = mf_dropout_keep(d0, mk2, a.keep_hi);
                d1 = mf_dropout_keep(d1, mk3, a.keep_hi);
            }
            __builtin_amdgcn_s_setprio(1);
            // colour 2: 64 -> 3 on the VALU in fp32 (weights pre-scaled by the dropout scale when masks are on).
            // A SIMD has an issue lane (4 cycles per VALU instruction, ~10 per f16 MFMA) beside its matrix lane (32 per MFMA:
            // benchmarks/issue_sweep_probe.hip), and these kernels are bound by the issue lane, so a layer belongs
            // where it costs fewer ISSUE cycles: as four more k-steps on the matrix pipe this one took 12 MFMAs + 48 split
            // instructions (~310 issue cycles, 29 of 32 output rows wasted), as packed fp32 FMAs it takes 48 + the half-to-half
            // exchange (~220).  (Rounds 2 - 5 argued the same choice from "MFMA and VALU never overlap, 32 + 4 cycles".)
            {
                const float4* wq = reinterpret_cast<const float4*>(lds + MF_H2_OFF + h * 48);
                // Per 32-unit block the 12 weight quads (3 channels x 4) are read into an array FIRST and the 24 packed FMAs
                // follow.  Written as one load per use the compiler issued each ds_read_b128 directly in front of its two FMAs
                // and waited for it: 24 exposed LDS round trips per pass (`D1 W1 v1 n0 v1` 24 times in the listing), in which
                // both waves of a SIMD tended to sit at once -- the K-pass "f16" kernel went 16.2 -> 15.0 ms per launch with the
                // reads grouped (profiles/r4_exp_rgb_weight_reads_*.json).  Forcing the grouping further with
                // sched_group_barrier was slower, and issuing the three channels' chains interleaved (no `s_nop` between
                // dependent packed FMAs, 439 instead of 476 instructions per pass) changed nothing: the other wave fills
                // those slots (profiles/r4_exp_rgb_interleave_*.json).
                // (the split form has no registers for 12 quads in flight -- 28 B of scratch with them -- and groups 4)
                constexpr int CG = F1 ? 3 : 1;   // channels whose weights are read together
                unerf_v2f acc2[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const f32x16& dv = blk ? d1 : d0;
#pragma unroll
                    for (int c0g = 0; c0g < 3; c0g += CG) {
                        float4 wv[CG][4];
#pragma unroll
                        for (int c = 0; c < CG; ++c)
#pragma unroll
                            for (int q4 = 0; q4 < 4; ++q4) wv[c][q4] = wq[blk * 24 + (c0g + c) * 4 + q4];
#pragma unroll
                        for (int c = 0; c < CG; ++c) {
#pragma unroll
                            for (int q4 = 0; q4 < 4; ++q4) {
                                // (round 6) UNERF_RGB_SCALAR: the same four fused multiply-adds on scalar registers (1: every
                                // mode, 2: ACTIVE only -- measured -1.8 % there and +0.7 % in the K-pass kernel, same box)
                                if constexpr (UNERF_RGB_SCALAR == 1 || (UNERF_RGB_SCALAR == 2 && MODE == UNERF_FIELD_ACTIVE)) {
                                    acc2[c0g + c].x = __builtin_fmaf(dv[4 * q4], wv[c][q4].x, acc2[c0g + c].x);
                                    acc2[c0g + c].y = __builtin_fmaf(dv[4 * q4 + 1], wv[c][q4].y, acc2[c0g + c].y);
                                    acc2[c0g + c].x = __builtin_fmaf(dv[4 * q4 + 2], wv[c][q4].z, acc2[c0g + c].x);
                                    acc2[c0g + c].y = __builtin_fmaf(dv[4 * q4 + 3], wv[c][q4].w, acc2[c0g + c].y);
                                } else {
                                    acc2[c0g + c] = __builtin_elementwise_fma(unerf_v2f{dv[4 * q4], dv[4 * q4 + 1]}, unerf_v2f{wv[c][q4].x, wv[c][q4].y}, acc2[c0g + c]);
                                    acc2[c0g + c] = __builtin_elementwise_fma(unerf_v2f{dv[4 * q4 + 2], dv[4 * q4 + 3]}, unerf_v2f{wv[c][q4].z, wv[c][q4].w}, acc2[c0g + c]);
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float half_sum = acc2[c].x + acc2[c].y;
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(half_sum), __float_as_uint(half_sum), false, false);
                    o[c] = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) + lds[MF_H2_OFF + 192 + c];
                }
            }
            }
            // Epilogue split over the two lane halves (both hold the three colour sums after the exchange; the density
            // logit, row 0, lives in the h = 0 half): h = 0 finishes (density, red), h = 1 (green, blue) -- two
            // exponentials, two reciprocals and two stores per lane instead of four, four and four on half the lanes.
            if (valid) {
                const OutIndex q = out_index(a, k, ts);
                const float x = h ? o[1] : o[0];
                const float y = h ? -o[2] : t[0];                 // h = 1: exp(-blue) for its sigmoid; h = 0: exp(logit)
                // F1 has no lo halves to turn an operand overflow into NaN: an activation beyond 65504 becomes an f16 inf,
                // which reaches every unit of the next layer (inf w, or inf - inf = NaN) and from there the density logit
                // (trunk units) or the three colour sums (head units) -- but sigmoid / exp map +-inf to 0, 1, inf: plausible
                // pixels.  So the four PRE-activation values are tested here (two per lane): two compares per pass.
                if (F1) f1_bad |= !(fabsf(x) < INFINITY) | !(fabsf(y) < INFINITY);
                const float ey = __expf(y);
                const float vy = h ? __builtin_amdgcn_rcpf(1.f + ey) : a.p.average_init_density * ey * sel;
                const float vx = mf_sigmoid_fast(x);
                if (a.p.packed_out) {   // uniform.  (vx, vy) = (red, sigma) in the h = 0 half, (green, blue) in the h = 1 half:
                    // one v_permlane32_swap each brings the upper half's pair down, and the lower half stores 16 bytes
                    // (a column's two lanes are the same sample: `valid` is the same in both)
                    const auto gx = __builtin_amdgcn_permlane32_swap(__float_as_uint(vx), __float_as_uint(vx), false, false);
                    const auto gy = __builtin_amdgcn_permlane32_swap(__float_as_uint(vy), __float_as_uint(vy), false, false);
                    if (h == 0) store_packed(a, k, ts.n, vy, vx, __uint_as_float(gx[1]), __uint_as_float(gy[1]));
                } else {
                    a.rgb[q.rgb + (h ? q.rgb_stride : 0)] = vx;
                    float* py = h ? a.rgb + (q.rgb + 2 * q.rgb_stride) : a.density + q.dens;
                    *py = vy;
                }
                if (MODE == UNERF_FIELD_ACTIVE && h == 0) a.aux[q.aux] = unerf_softplus(t[8]) + a.p.beta_min;
            }
        }
    }
    if (a.p.overflow_flag) {   // one atomic per offending wave and launch
        const uint64_t m = __builtin_amdgcn_ballot_w64(f1_bad);
        if (m != 0 && lane_c == (int)__builtin_ctzll(m)) atomicOr(a.p.overflow_flag, 1);
    }
}

// --------------------------------------------------------------------------------------
// 5b'. LAPLACE on the matrix cores.  The reference evaluates the two sampled last layers in a
// 100-iteration Python loop of GEMVs (laplace_field.py:553-560); as a matrix product the weight
// SAMPLES are the output rows: density head 128(100) x 64, colour head 3 x 128(100) x 64 per
// 32-sample tile, followed by exp / sigmoid and the running sums of p and p^2 on the accumulator
// registers (rows = registers + lane half).  Padded rows carry bias -1e30 -> contribute exactly 0.
// The 512 sampled-head fragments (128 KB) do not fit LDS next to the base network; they stream
// from L2 with coalesced 256-B loads (one per MFMA), the base network stays LDS-resident.
// --------------------------------------------------------------------------------------
#define LAP_BLOCKS 4
#define LAP_BIAS_OFF (4 * LAP_BLOCKS * 32 * 64)

// One sampled head (q) over its LAP_BLOCKS row blocks.  The A fragments come from L2 (one coalesced
// 256-B load per MFMA); the 32 fragments of block b+1 are requested BEFORE the 32 MFMAs of block b
// issue, so ~2000 cycles of matrix work cover their latency (without this every MFMA waited on its
// own load: 254 ms/frame instead of the ~70 ms the MFMA count predicts).
template <bool SIGMOID>
__device__ __forceinline__ void mf_lap_head(const float* __restrict__ lap, int q, int lane, int h, const f32x16& s0,
                                            const f32x16& s1, float& sum1, float& sum2, int softplus = 0) {
    sum1 = 0.f;
    sum2 = 0.f;
    float cur[32], nxt[32];
    {
        const float* f = lap + (size_t)((q * LAP_BLOCKS + 0) * 32) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 32; ++r) cur[r] = f[r * 64];
    }
#pragma unroll
    for (int b = 0; b < LAP_BLOCKS; ++b) {
        if (b + 1 < LAP_BLOCKS) {
            const float* f = lap + (size_t)((q * LAP_BLOCKS + b + 1) * 32) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 32; ++r) nxt[r] = f[r * 64];
        }
        const float4* bp = reinterpret_cast<const float4*>(lap + LAP_BIAS_OFF + ((q * LAP_BLOCKS + b) * 2 + h) * 16);
        float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        f32x16 acc = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[r], s0[r], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[16 + r], s1[r], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1e-6 relative) instead of the ~25-instruction exact
            // expf / division: 16 activations follow every 32 MFMAs here and would otherwise take as long as
            // the matrix work.  The results only enter means / variances over the n_lap samples.
            float p = SIGMOID ? __builtin_amdgcn_rcpf(1.f + __expf(-acc[r])) : (softplus ? mf_softplus_fast(acc[r]) : __expf(acc[r]));
            sum1 += p;
            sum2 += p * p;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (b + 1 < LAP_BLOCKS) {
#pragma unroll
            for (int r = 0; r < 32; ++r) cur[r] = nxt[r];
        }
    }
    sum1 += __shfl_xor(sum1, 32, 64);
    sum2 += __shfl_xor(sum2, 32, 64);
}

// store one accumulator block (rows = units 32*blk + (r&3) + 8*(r>>2) + 4h of sample n) to a [N][64] plane
__device__ __forceinline__ void mf_store_units(float* plane, int64_t n, int blk, int h, const f32x16& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(plane + n * 64 + 32 * blk + 8 * q + 4 * h) =
            make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
// this lane's half of a 64-wide dot product with a weight row in global memory (mean last layer)
__device__ __forceinline__ float mf_half_dot(const float* __restrict__ w, int h, const f32x16& v0, const f32x16& v1) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = fmaf(v0[r], w[(r & 3) + 8 * (r >> 2) + 4 * h], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = fmaf(v1[r], w[32 + (r & 3) + 8 * (r >> 2) + 4 * h], acc);
    return acc + __shfl_xor(acc, 32, 64);
}

// CAPTURE = the deterministic (is_inference=False) forward for GGN fitting: ws_density / ws_rgb hold the MEAN
// last layers, density = exp(.) * selector (laplace_field.py:317-345), rgb = sigmoid(.), and the inputs of the
// two last layers (base_mlp output, colour hidden) are written to [N][64] planes a.aux / a.aux2.
template <bool CAPTURE, int TCNN = 0>
__global__ __launch_bounds__(256) void field_kernel_mfma_laplace(FieldArgs a, uint32_t num_tiles, FastDiv div_s) {
    extern __shared__ float lds[];
    {
        const float4* src = reinterpret_cast<const float4*>(a.p.mfma_blob);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < UNERF_MFMA_BLOB_FLOATS / 4; i += 256) dst[i] = src[i];
    }
    __shared__ uint32_t s_tl[TCNN ? MF_TL_WORDS : 1];
    if (TCNN) mf_stage_tcnn_levels<(TCNN == 2 ? 2 : 3)>(a, s_tl);
    __syncthreads();
    // wave index as a scalar: the tile walk and its divisions then run on the scalar unit
    const int lane_c = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane_c & 31, h = lane_c >> 5;
    const int64_t N = a.R * (int64_t)a.S;
    const uint32_t mask = (1u << a.p.log2T) - 1u;
    const float inv_n = 1.f / (float)a.p.n_lap, inv_nr = 1.f / (float)a.p.n_lap_rgb;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const uint32_t tpx = (num_tiles + 7u) / 8u;  // num_tiles < 2^28 (the RNG counter bound in unerf_field_fwd)
    const uint32_t tile_end = (xcd + 1) * tpx < num_tiles ? (xcd + 1) * tpx : num_tiles;
    for (uint32_t tile = xcd * tpx + (uint32_t)slot * 4u + (uint32_t)wv; tile < tile_end; tile += (uint32_t)bpx * 4u) {
        int lane = lane_c;  // opaque per iteration: keeps the (tile-invariant) fragment reads in the loop
        asm volatile("" : "+v"(lane));
        // tile -> (block of 32 neighbouring rays, sample index s): the 32 columns of a tile are the SAME
        // sample slot of 32 adjacent pixels, which sit in the same or neighbouring grid cells, so a gather
        // instruction presents few distinct lines to the texture-address unit (it retires ~1 divergent
        // lane per clock: TA_BUSY 74 % with 32 consecutive samples of one ray per tile, rocprof r1_04)
        const TileSample ts = tile_sample(a, tile, div_s, j);
        const bool valid = ts.valid;
        const int64_t n = ts.n;
        const float dxr = ts.dx, dyr = ts.dy, dzr = ts.dz;
        float px = ts.px, py = ts.py, pz = ts.pz;
        // inference: the returned mu_d is NOT selector-masked (laplace_field.py:356-362)
        const float sel = unerf_normalize_position(px, py, pz, a.box);
        f32x16 feat = mf_gather_feats<true, TCNN>(a, px, py, pz, h, mask, s_tl);

        // base_mlp is a bare Linear: no ReLU (utils.py:22-23)
        f32x16 hb0 = mf_slab(lds, 0, lane, feat, mf_bias(lds, 0, h));
        f32x16 hb1 = mf_slab(lds, 16, lane, feat, mf_bias(lds, 1, h));
        // geo = mlp_hidden(hb): rows 0..14
        f32x16 t = mf_bias(lds, 2, h);
        t = mf_slab(lds, 32, lane, hb0, t);
        t = mf_slab(lds, 48, lane, hb1, t);
        // density head: mean / variance of exp(w_s . hb + b_s) over the n_lap sampled rows
        float mu_d, mu2_d = 0.f;
        if (CAPTURE) {
            const float pre_d = mf_half_dot(a.p.ws_density, h, hb0, hb1) + a.p.ws_density[64];
            mu_d = (a.p.lap_softplus ? unerf_softplus(pre_d) : expf(pre_d)) * sel;
            if (valid) {
                mf_store_units(a.aux, n, 0, h, hb0);
                mf_store_units(a.aux, n, 1, h, hb1);
            }
        } else {
            float d1, d2;
            mf_lap_head<false>(lap_set_blob(a, a.p.lap_blob, tile, div_s), 0, lane, h, hb0, hb1, d1, d2, a.p.lap_softplus);
            mu_d = d1 * inv_n;
            mu2_d = d2 * inv_n;
            if (a.p.lap_mask_density) {  // use_deterministic_density: selector-masked mean, no variance
                mu_d *= sel;
                mu2_d = mu_d * mu_d;
            }
        }

        // colour trunk
        f32x16 c0 = mf_bias(lds, 3, h), c1 = mf_bias(lds, 4, h);
        {
            float sh[16];
            float ux = (dxr + 1.f) / 2.f, uy = (dyr + 1.f) / 2.f, uz = (dzr + 1.f) / 2.f;
            if (a.p.sh_remap) {
                ux = ux * 2.f - 1.f;
                uy = uy * 2.f - 1.f;
                uz = uz * 2.f - 1.f;
            }
            unerf_sh16(ux, uy, uz, sh);
            const uint32_t hm = 0u - (uint32_t)h;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[(64 + q) * 64 + lane], t[q], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[(80 + q) * 64 + lane], t[q], c1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v = __uint_as_float((__float_as_uint(sh[8 + q]) & hm) | (__float_as_uint(sh[q]) & ~hm));
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[(64 + 8 + q) * 64 + lane], v, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[(80 + 8 + q) * 64 + lane], v, c1, 0, 0, 0);
            }
        }
        c0 = mf_relu(c0);
        c1 = mf_relu(c1);
        f32x16 x0 = mf_bias(lds, 5, h), x1 = mf_bias(lds, 6, h);
        x0 = mf_slab(lds, 96, lane, c0, x0);
        x0 = mf_slab(lds, 112, lane, c1, x0);
        x1 = mf_slab(lds, 128, lane, c0, x1);
        x1 = mf_slab(lds, 144, lane, c1, x1);
        x0 = mf_relu(x0);
        x1 = mf_relu(x1);
        // colour head: per channel mean / variance of sigmoid(w_s . x + b_s)
        float mu_c[3], vsum = 0.f;
        if (CAPTURE) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                mu_c[c] = unerf_sigmoid(mf_half_dot(a.p.ws_rgb + c * 64, h, x0, x1) + a.p.ws_rgb[192 + c]);
            if (valid) {
                mf_store_units(a.aux2, n, 0, h, x0);
                mf_store_units(a.aux2, n, 1, h, x1);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float c1s, c2s;
                mf_lap_head<true>(lap_set_blob(a, a.p.lap_blob, tile, div_s), 1 + c, lane, h, x0, x1, c1s, c2s);
                mu_c[c] = c1s * inv_nr;
                vsum += fmaxf(c2s * inv_nr - mu_c[c] * mu_c[c], 0.f);
            }
        }
        if (valid && h == 0) {
            a.density[n] = mu_d;
            if (!CAPTURE) {
                a.aux[n] = mu2_d - mu_d * mu_d;
                a.aux2[n] = vsum / 3.f;
            }
            a.rgb[n * 3 + 0] = mu_c[0];
            a.rgb[n * 3 + 1] = mu_c[1];
            a.rgb[n * 3 + 2] = mu_c[2];
        }
    }
}

// --------------------------------------------------------------------------------------
// 5b-3. Split-f16 variant of the LAPLACE kernel (inference): base network as in field_kernel_mfma16, the 4 x 128
// sampled last-layer rows as 16 (head, block) groups of 4 k-steps x 3 products = 12 MFMAs (instead of 32),
// operands streamed from L2 as 16-byte quads (ops.pack_laplace_heads16), one block prefetched ahead.  With the
// matrix work cut 5x the exp / rcp epilogues (quarter-rate transcendentals) are the larger half, so register
// quads that only hold padding rows (rows >= n_lap) are skipped.
// --------------------------------------------------------------------------------------
// ACT: 0 exp, 1 sigmoid, 2 softplus.  With UNERF_LAP_EXP2 the rows of lap16_blob carry the base change
// (ops.pack_laplace_heads16: density rows x log2 e, colour rows x -log2 e), so exp / sigmoid are the bare v_exp_f32
// (+ v_rcp_f32); softplus rows are unscaled.  The running sums of p and p^2 are one packed fma per activation,
// (s1, s2) += (p, p) * (1, p), instead of a multiply and a packed add.
template <int ACT, bool F1 = false>
__device__ __forceinline__ void mf16_lap_head(const float* __restrict__ lap, int q, int n_lap, int lane,
                                              const f16x8 (&bhi)[4], const f16x8 (&blo)[4], int h, float& sum1,
                                              float& sum2) {
    unerf_v2f s12 = {0.f, 0.f}, pr = {1.f, 0.f};
    f16x8 cur[8], nxt[8];
    {
        const float* f = lap + (size_t)((q * LAP_BLOCKS + 0) * 8) * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < 8; i += (F1 ? 2 : 1)) cur[i] = *reinterpret_cast<const f16x8*>(f + i * 256);   // F1: hi quads only
    }
#pragma unroll
    for (int b = 0; b < LAP_BLOCKS; ++b) {
        if (b + 1 < LAP_BLOCKS) {
            const float* f = lap + (size_t)((q * LAP_BLOCKS + b + 1) * 8) * 256 + lane * 4;
#pragma unroll
            for (int i = 0; i < 8; i += (F1 ? 2 : 1)) nxt[i] = *reinterpret_cast<const f16x8*>(f + i * 256);
        }
        const float4* bp = reinterpret_cast<const float4*>(lap + LAP_BIAS_OFF + ((q * LAP_BLOCKS + b) * 2 + h) * 16);
        float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        f32x16 acc = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
        // the fences keep each row block's MFMAs and its activations (with the inline-asm packed fma, whose hazards
        // the compiler does not model) as separate scheduling regions: without them the kernel is 1.3 % faster and wrong
        LAP_FENCE();
#pragma unroll
        for (int st = 0; st < 4; ++st) {  // cur[2 st] = hi, cur[2 st + 1] = lo of k-step st
            if (!F1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[2 * st + 1], bhi[st], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[2 * st], blo[st], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[2 * st], bhi[st], acc, 0, 0, 0);
        }
        // register quad qd holds rows 8 qd + 4 h + (0..3) of this block: skip quads that are padding in both halves
        const int rows = n_lap - 32 * b;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            if (8 * qd < rows) {  // uniform
#pragma unroll
                for (int r = 4 * qd; r < 4 * qd + 4; ++r) {
                    float p;
                    if (ACT == 2) {
                        p = mf_softplus_fast(acc[r]);
                    } else {
                        const float e = UNERF_LAP_EXP2 ? __builtin_amdgcn_exp2f(acc[r]) : __expf(ACT == 1 ? -acc[r] : acc[r]);
                        p = ACT == 1 ? __builtin_amdgcn_rcpf(1.f + e) : e;
                    }
                    // (s1, s2) += (p, p) * (1, p): both factors read from the pair P = (1, p) through op_sel (src0 takes
                    // the high dword for both lanes, src1 low / high).  Spelled with vector types the compiler builds the
                    // two factor pairs with a v_mov each.  s_nop: a transcendental's result (p comes from v_exp / v_rcp)
                    // needs one wait state before a VALU instruction the compiler cannot see may read it.
#if UNERF_LAP_SCALAR_MOMENTS
                    // round 6: the same two sums as one add and one fma on their own registers (the same roundings as the
                    // packed fma's two halves).  A packed-fp32 instruction cannot run beside an MFMA -- one behind an MFMA
                    // waits for it and costs 18 cycles, the next MFMA waits for the packed instruction in turn
                    // (benchmarks/issue_sweep_probe.hip) -- and at two or three waves per SIMD the other wave's head MFMAs
                    // are always in flight; v_add_f32 / v_fmac_f32 issue in their shadow.
                    s12.x += p;
                    s12.y = __builtin_fmaf(p, p, s12.y);
#else
                    pr.y = p;
                    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(s12) : "v"(pr));
#endif
                }
            }
        }
        LAP_FENCE();
        if (b + 1 < LAP_BLOCKS) {
#pragma unroll
            for (int i = 0; i < 8; i += (F1 ? 2 : 1)) cur[i] = nxt[i];
        }
    }
    sum1 = s12.x + __shfl_xor(s12.x, 32, 64);
    sum2 = s12.y + __shfl_xor(s12.y, 32, 64);
}

// UNERF_LAP_PREFETCH >= 2: the same head evaluation as ONE operand stream over NHEADS consecutive heads with the loads
// running TWO row blocks ahead of the matrix work (three rotating register buffers instead of two).  The kernel was
// latency-bound on exactly this stream -- 8 KB per wave and block from L2, one block (~0.9 us of two waves' work) of
// lead against ~1.4 us of loaded L2 latency: issue-busy 0.575, SQ_WAIT_ANY 0.34 (profiles/issue_laplace.json, round
// 3) -- and every head started cold; the three colour heads now run as one 12-block stream.
#ifndef UNERF_LAP_PREFETCH
#define UNERF_LAP_PREFETCH 2
#endif
#ifndef UNERF_LAP_BIAS_LDS
#define UNERF_LAP_BIAS_LDS 1     // the 512 bias words of a sample set staged in LDS per wave and tile (see the kernel)
#endif
template <int ACT, bool F1, int NHEADS>
__device__ __forceinline__ void mf16_lap_stream(const float* __restrict__ lap, const float* __restrict__ lds_bias, int q0,
                                                int n_lap, int lane, const f16x8 (&bhi)[4], const f16x8 (&blo)[4], int h,
                                                float (&sum1)[NHEADS], float (&sum2)[NHEADS]) {
    constexpr int NB = NHEADS * LAP_BLOCKS;
    f16x8 buf[3][8];
    const float* base = lap + (size_t)(q0 * LAP_BLOCKS * 8) * 256 + lane * 4;
#pragma unroll
    for (int i = 0; i < 8; i += (F1 ? 2 : 1)) buf[0][i] = *reinterpret_cast<const f16x8*>(base + i * 256);
    if (NB > 1) {
#pragma unroll
        for (int i = 0; i < 8; i += (F1 ? 2 : 1)) buf[1][i] = *reinterpret_cast<const f16x8*>(base + (8 + i) * 256);
    }
    unerf_v2f s12 = {0.f, 0.f}, pr = {1.f, 0.f};
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const int b = blk % LAP_BLOCKS, hd = blk / LAP_BLOCKS;
        if (blk + 2 < NB) {
#pragma unroll
            for (int i = 0; i < 8; i += (F1 ? 2 : 1))
                buf[(blk + 2) % 3][i] = *reinterpret_cast<const f16x8*>(base + ((blk + 2) * 8 + i) * 256);
        }
        const f16x8 (&cur)[8] = buf[blk % 3];
        const float4* bp = reinterpret_cast<const float4*>((UNERF_LAP_BIAS_LDS ? lds_bias : lap + LAP_BIAS_OFF) +
                                                           (((q0 + hd) * LAP_BLOCKS + b) * 2 + h) * 16);
        float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        f32x16 acc = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
        if (b == 0) s12 = unerf_v2f{0.f, 0.f};
        LAP_FENCE();   // (see mf16_lap_head: the inline-asm packed fma and its operands stay in one region)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (!F1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[2 * st + 1], bhi[st], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[2 * st], blo[st], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[2 * st], bhi[st], acc, 0, 0, 0);
        }
        const int rows = n_lap - 32 * b;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            if (8 * qd < rows) {  // uniform
#pragma unroll
                for (int r = 4 * qd; r < 4 * qd + 4; ++r) {
                    float p;
                    if (ACT == 2) {
                        p = mf_softplus_fast(acc[r]);
                    } else {
                        const float e = UNERF_LAP_EXP2 ? __builtin_amdgcn_exp2f(acc[r]) : __expf(ACT == 1 ? -acc[r] : acc[r]);
                        p = ACT == 1 ? __builtin_amdgcn_rcpf(1.f + e) : e;
                    }
#if UNERF_LAP_SCALAR_MOMENTS
                    // round 6: the same two sums as one add and one fma on their own registers (the same roundings as the
                    // packed fma's two halves).  A packed-fp32 instruction cannot run beside an MFMA -- one behind an MFMA
                    // waits for it and costs 18 cycles, the next MFMA waits for the packed instruction in turn
                    // (benchmarks/issue_sweep_probe.hip) -- and at two or three waves per SIMD the other wave's head MFMAs
                    // are always in flight; v_add_f32 / v_fmac_f32 issue in their shadow.
                    s12.x += p;
                    s12.y = __builtin_fmaf(p, p, s12.y);
#else
                    pr.y = p;
                    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(s12) : "v"(pr));
#endif
                }
            }
        }
        LAP_FENCE();
        if (b == LAP_BLOCKS - 1) {
            sum1[hd] = s12.x + __shfl_xor(s12.x, 32, 64);
            sum2[hd] = s12.y + __shfl_xor(s12.y, 32, 64);
        }
    }
}

template <int TCNN, bool F1 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((F1 && !TCNN) ? 3 : 2))) void field_kernel_mfma16_laplace(FieldArgs a, uint32_t num_tiles, FastDiv div_s) {
    extern __shared__ float lds[];
    {
        const float4* src = reinterpret_cast<const float4*>(a.p.mfma16_blob);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < UNERF_MFMA_BLOB_FLOATS / 4; i += 256) dst[i] = src[i];
    }
    __shared__ uint32_t s_tl[TCNN ? MF_TL_WORDS : 1];
    // The bias rows of the sampled heads (16 per lane half and row block: the accumulators' initial values).  As global
    // loads in front of every block's first MFMA they were the one load of the head loop whose latency nothing hid --
    // 16 round trips to L2 per tile.  Each wave brings its tile's 512 words in with two 16-byte loads per lane while the
    // hash grid is gathered, parks them in LDS, and the blocks read them back with ds_read_b128.
    __shared__ float s_lbias[UNERF_LAP_BIAS_LDS && UNERF_LAP_PREFETCH >= 2 ? 4 : 1][UNERF_LAP_BIAS_LDS && UNERF_LAP_PREFETCH >= 2 ? 512 : 4];
    if (TCNN) mf_stage_tcnn_levels<(TCNN == 2 ? 2 : 3)>(a, s_tl);
    __syncthreads();
    const int lane_c = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane_c & 31, h = lane_c >> 5;
    const uint32_t mask = (1u << a.p.log2T) - 1u;
    const float inv_n = 1.f / (float)a.p.n_lap, inv_nr = 1.f / (float)a.p.n_lap_rgb;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const uint32_t tpx = (num_tiles + 7u) / 8u;
    const uint32_t tile_end = (xcd + 1) * tpx < num_tiles ? (xcd + 1) * tpx : num_tiles;
    bool f1_bad = false;   // F1: a sampled-head mean of this lane came out inf / NaN
    for (uint32_t tile = xcd * tpx + (uint32_t)slot * 4u + (uint32_t)wv; tile < tile_end; tile += (uint32_t)bpx * 4u) {
        int lane = lane_c;
        asm volatile("" : "+v"(lane));
        const TileSample ts = tile_sample(a, tile, div_s, j);
        const bool valid = ts.valid;
        const int64_t n = ts.n;
        const float dxr = ts.dx, dyr = ts.dy, dzr = ts.dz;
        float px = ts.px, py = ts.py, pz = ts.pz;
        // inference: the returned mu_d is NOT selector-masked (laplace_field.py:356-362) unless lap_mask_density
        const float sel = unerf_normalize_position(px, py, pz, a.box);
        constexpr bool BIAS_LDS = UNERF_LAP_BIAS_LDS && UNERF_LAP_PREFETCH >= 2 && !(F1 && !TCNN);
        float4 lb0, lb1;
        if (BIAS_LDS) {
            const float4* lbsrc = reinterpret_cast<const float4*>(lap_set_blob(a, a.p.lap16_blob, tile, div_s) + LAP_BIAS_OFF) + lane_c * 2;
            lb0 = lbsrc[0];
            lb1 = lbsrc[1];
        }
        u32x8 feat_pk;
        const f32x16 feat = mf_gather_feats<true, TCNN>(a, px, py, pz, h, mask, s_tl, &feat_pk);
        if (BIAS_LDS) {   // (the previous tile's heads are done with the buffer: a wave's LDS operations execute in order)
            float4* dst = reinterpret_cast<float4*>(s_lbias[wv]) + lane_c * 2;
            dst[0] = lb0;
            dst[1] = lb1;
        }

        const float* lap16 = lap_set_blob(a, a.p.lap16_blob, tile, div_s);
        // (the single-product kernel at three waves per SIMD has no registers for a third operand buffer: it keeps the
        // one-block-ahead heads)
        constexpr bool STREAM = UNERF_LAP_PREFETCH >= 2 && !(F1 && !TCNN);
        // base_mlp: bare Linear 32 -> 64 (no ReLU, utils.py:22-23)
        f32x16 hb0 = mf16_bias(lds, 0, h), hb1 = mf16_bias(lds, 1, h);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f16x8 bhi, blo;
            mf16_feat_operand<TCNN, F1>(feat, feat_pk, st, bhi, blo);
            mf16_mac2<F1, TCNN == 2>(lds, 2 * st, 2 * st + 1, lane, bhi, blo, hb0, hb1);
        }
        // the 64 base outputs feed both mlp_hidden (geo) and the sampled density rows: split them once
        f16x8 xhi[4], xlo[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) mf16_split<F1>(st < 2 ? hb0 : hb1, st & 1, xhi[st], xlo[st]);
        f32x16 t = mf16_bias(lds, 2, h);
#pragma unroll
        for (int st = 0; st < 4; ++st) t = mf16_mac<F1>(lds, 4 + st, lane, xhi[st], xlo[st], t);
        float d1, d2;
        if constexpr (STREAM) {
            float ds1[1], ds2[1];
            if (a.p.lap_softplus) mf16_lap_stream<2, F1, 1>(lap16, s_lbias[wv], 0, a.p.n_lap, lane, xhi, xlo, h, ds1, ds2);   // uniform
            else mf16_lap_stream<0, F1, 1>(lap16, s_lbias[wv], 0, a.p.n_lap, lane, xhi, xlo, h, ds1, ds2);
            d1 = ds1[0];
            d2 = ds2[0];
        } else {
            if (a.p.lap_softplus) mf16_lap_head<2, F1>(lap16, 0, a.p.n_lap, lane, xhi, xlo, h, d1, d2);   // uniform
            else mf16_lap_head<0, F1>(lap16, 0, a.p.n_lap, lane, xhi, xlo, h, d1, d2);
        }
        float mu_d = d1 * inv_n, mu2_d = d2 * inv_n;
        if (a.p.lap_mask_density) {  // use_deterministic_density: selector-masked mean, no variance
            mu_d *= sel;
            mu2_d = mu_d * mu_d;
        }

        // colour trunk: [geo15 | SH16] -> 64 -> 64
        f32x16 c0 = mf16_bias(lds, 3, h), c1 = mf16_bias(lds, 4, h);
        {
            f16x8 bhi, blo;
            mf16_split<F1>(t, 0, bhi, blo);
            mf16_mac2<F1>(lds, 8, 9, lane, bhi, blo, c0, c1);
            float sh[16];
            float ux = (dxr + 1.f) / 2.f, uy = (dyr + 1.f) / 2.f, uz = (dzr + 1.f) / 2.f;
            if (a.p.sh_remap) {
                ux = ux * 2.f - 1.f;
                uy = uy * 2.f - 1.f;
                uz = uz * 2.f - 1.f;
            }
            unerf_sh16(ux, uy, uz, sh);
            const uint32_t hm = 0u - (uint32_t)h;
            float mine[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                mine[q] = __uint_as_float((__float_as_uint(sh[8 + q]) & hm) | (__float_as_uint(sh[q]) & ~hm));
            mf16_split8<F1>(mine, bhi, blo);
            mf16_mac2<F1>(lds, 10, 11, lane, bhi, blo, c0, c1);
        }
        // (an overflowed operand of the split form makes every unit of the next layer NaN, which the integer-maximum ReLU
        // may turn into 0: one accumulator per colour layer is tested first -- see field_kernel_mfma16)
        if (!F1) f1_bad |= c0[0] != c0[0];
        c0 = mf_relu(c0);
        c1 = mf_relu(c1);
        f32x16 x0 = mf16_bias(lds, 5, h), x1 = mf16_bias(lds, 6, h);
        mf16_layer64<2, F1>(lds, 12, lane, c0, c1, x0, x1);
        if (!F1) f1_bad |= x0[0] != x0[0];
        x0 = mf_relu(x0);
        x1 = mf_relu(x1);
#pragma unroll
        for (int st = 0; st < 4; ++st) mf16_split<F1>(st < 2 ? x0 : x1, st & 1, xhi[st], xlo[st]);
        float mu_c[3], vsum = 0.f;
        if constexpr (STREAM) {
            float cs1[3], cs2[3];
            mf16_lap_stream<1, F1, 3>(lap16, s_lbias[wv], 1, a.p.n_lap_rgb, lane, xhi, xlo, h, cs1, cs2);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                mu_c[c] = cs1[c] * inv_nr;
                vsum += fmaxf(cs2[c] * inv_nr - mu_c[c] * mu_c[c], 0.f);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float c1s, c2s;
                mf16_lap_head<1, F1>(lap16, 1 + c, a.p.n_lap_rgb, lane, xhi, xlo, h, c1s, c2s);
                mu_c[c] = c1s * inv_nr;
                vsum += fmaxf(c2s * inv_nr - mu_c[c] * mu_c[c], 0.f);
            }
        }
        // F1: an f16 operand beyond 65504 turns the sampled rows into +-inf / NaN; a density mean of +inf from a FINITE
        // logit is not possible below e^88, so non-finite means are treated as operand overflow (see field_kernel_mfma16)
        if (F1 && valid) f1_bad |= !(fabsf(mu_d) < INFINITY) | !(fabsf(mu_c[0] + mu_c[1] + mu_c[2]) < INFINITY);
        if (valid && h == 0) {
            a.density[n] = mu_d;
            a.aux[n] = mu2_d - mu_d * mu_d;
            a.aux2[n] = vsum / 3.f;
            a.rgb[n * 3 + 0] = mu_c[0];
            a.rgb[n * 3 + 1] = mu_c[1];
            a.rgb[n * 3 + 2] = mu_c[2];
        }
    }
    if (a.p.overflow_flag) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(f1_bad);
        if (m != 0 && lane_c == (int)__builtin_ctzll(m)) atomicOr(a.p.overflow_flag, 1);
    }
}

// --------------------------------------------------------------------------------------
// 5c. level-major hash-grid gather.  One level of the main grid is 2^19 x 8 B = 4 MiB -- exactly
// one XCD's L2.  Sample-major lookup (all 16 levels per sample) keeps 64 MiB live and runs at the
// Infinity-Cache random-64-B-request rate (40 ms/frame inside the fused kernel, 5.05 ms per 2^18
// rays stand-alone); walking the levels in the SLOW grid dimension makes every XCD sweep one
// 4-MiB table at a time out of its own L2: 2.3 ms per 2^18 rays (benchmarks/exp_level_major.py).
// Features go to level-major planes [16][N] float2 (coalesced 8-B stores here, coalesced 8-B
// loads in field_kernel_mfma<.., FEAT_IN>), and the gather runs on its own stream underneath the
// previous launch group's matrix work (render.py).
// --------------------------------------------------------------------------------------
struct GatherArgs {
    const float* origins;
    const float* dirs;
    const float* sbins;
    int64_t R;
    int S;
    float s_near, s_far;
    int lin;   // UNERF_SPACING_*: 1 = identity spacing (UniformSampler)
    const float* table;
    const float* scalings;
    int L, log2T;
    float* planes;
    unerf_norm_box box;   // contraction path only (unerf_field_gather has no aabb argument)
};

__global__ __launch_bounds__(256) void field_gather_kernel(GatherArgs a) {
    const int64_t N = a.R * (int64_t)a.S;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int lev = blockIdx.y;
    const int64_t r = n / a.S;
    const int s = (int)(n - r * a.S);
    const float* sb = a.sbins + r * (a.S + 1);
    float e0 = unerf_s2e(sb[s], a.s_near, a.s_far, a.lin), e1 = unerf_s2e(sb[s + 1], a.s_near, a.s_far, a.lin);
    float t01 = e0 + e1;
    float px = a.origins[r * 3 + 0] + a.dirs[r * 3 + 0] * t01 / 2.f;
    float py = a.origins[r * 3 + 1] + a.dirs[r * 3 + 1] * t01 / 2.f;
    float pz = a.origins[r * 3 + 2] + a.dirs[r * 3 + 2] * t01 / 2.f;
    (void)unerf_normalize_position(px, py, pz, a.box);
    const float2* lvl = reinterpret_cast<const float2*>(a.table) + ((size_t)lev << a.log2T);
    float2 f = unerf_hash_level(lvl, px, py, pz, a.scalings[lev], (1u << a.log2T) - 1u);
    reinterpret_cast<float2*>(a.planes)[(int64_t)lev * N + n] = f;
}

extern "C" int unerf_field_gather(const float* origins, const float* directions, const float* sbins, int64_t R, int S,
                                  float near_plane, float far_plane, int spacing, const float* table, const float* scalings, int L,
                                  int log2T, float* feature_planes, void* stream) {
    UNERF_REQUIRE(table && scalings && (R == 0 || (origins && directions && sbins && feature_planes)), "field_gather: null pointer");
    UNERF_REQUIRE(L >= 1 && L <= 32 && log2T >= 1 && log2T <= 24 && R >= 0 && S >= 1, "field_gather: bad L/log2T/R/S");
    if (R == 0) return UNERF_OK;
    GatherArgs a;
    a.origins = origins; a.dirs = directions; a.sbins = sbins; a.R = R; a.S = S;
    UNERF_REQUIRE_SPACING(spacing); a.lin = spacing; a.s_near = unerf_spacing_of(near_plane, spacing); a.s_far = unerf_spacing_of(far_plane, spacing);
    a.table = table; a.scalings = scalings; a.L = L; a.log2T = log2T; a.planes = feature_planes;
    a.box = make_norm_box(0, nullptr);
    // blockIdx.x runs fastest in dispatch order, so all workgroups of level l are issued before
    // level l+1: the chip works on (at most) two adjacent level tables at any time
    dim3 grid(blocks_for(R * (int64_t)S, 256), L), block(256);
    hipLaunchKernelGGL(field_gather_kernel, grid, block, 0, (hipStream_t)stream, a);
    return unerf_check_launch("field_gather");
}

// Persistent grid of the matrix kernels: exactly as many workgroups as are co-resident (occupancy x CUs, a
// multiple of the 8 XCDs).  The ACTIVE kernels fit 3 per CU (42.6 KB LDS, <= 168 VGPRs), the K-pass and Laplace
// kernels 2; launching 3 per CU for those left a third of the tiles to a half-empty second round.
template <typename Kern>
static int mfma_grid_for(Kern kernel, int64_t num_tiles, size_t lds_bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, int> cache;
    int cap = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find(reinterpret_cast<const void*>(kernel));
        if (it != cache.end()) cap = it->second;
    }
    if (cap == 0) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds_bytes) != hipSuccess || per_cu < 1)
            per_cu = 2;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipGetLastError();
        cap = per_cu * cus;
        if (getenv("UNERF_DEBUG_GRID")) fprintf(stderr, "[unerf] persistent grid: %d workgroups/CU x %d CUs\n", per_cu, cus);
        std::lock_guard<std::mutex> lock(mu);
        cache[reinterpret_cast<const void*>(kernel)] = cap;
    }
    int64_t blocks = (num_tiles + 3) / 4;
    if (blocks > cap) blocks = cap;
    return (int)((blocks + 7) / 8 * 8);
}
// one persistent matrix-kernel launch: tile map from the image_width hint, LDS = the operand blob
#define MF_LDS_FP32 ((size_t)UNERF_MFMA_BLOB_FLOATS * 4)
#define MF_LDS_F16 MF_LDS_FP32
#define MF_LDS_F16S ((size_t)UNERF_MFMA16_BLOB_FLOATS * 4)   // the "f16" form of field_kernel_mfma16: + the colour-2 slabs
template <typename Kern>
static void launch_matrix_kernel(Kern kernel, size_t lds_bytes, FieldArgs& a, hipStream_t st) {
    const int64_t tiles = make_tiles(a, a.p.image_width);
    hipLaunchKernelGGL(kernel, dim3(mfma_grid_for(kernel, tiles, lds_bytes)), dim3(256), lds_bytes, st, a, (uint32_t)tiles,
                       make_fastdiv((uint32_t)a.S));
}

extern "C" int unerf_field_fwd(const float* origins, const float* directions, const float* sbins, int64_t R, int S,
                               float near_plane, float far_plane, int spacing, int64_t ray_offset, const unerf_field_params* p,
                               const float* features, float* density, float* rgb, float* aux, float* aux2,
                               void* stream) {
    UNERF_REQUIRE(p && (R == 0 || (origins && directions && sbins && (density || p->packed_out) && rgb)), "field_fwd: null pointer");
    UNERF_REQUIRE(!p->packed_out || (p->mode != UNERF_FIELD_LAPLACE && !p->sample_major),
                  "field_fwd: packed_out rows are written by the ACTIVE / MCDROPOUT kernels in the ray-major layout only");
    UNERF_REQUIRE(p->table && (p->scalings || p->tcnn_levels) && p->w0t && p->b0 && p->w1t && p->b1 && p->h0t &&
                      p->hb0 && p->h1t && p->hb1 && p->h2t && p->hb2,
                  "field_fwd: null weight pointer");
    // any-width slow path (field_kernel_generic): widths given and different from nerfacto's 64 / 64 / 15 / 2 (or L != 16)
    const int gH = p->hidden ? p->hidden : 64, gHC = p->hidden_color ? p->hidden_color : 64, gG = p->geo_dim ? p->geo_dim : 15;
    const int gF = p->feat_per_level ? p->feat_per_level : 2, gAD = p->app_dim ? p->app_dim : 32;
    const bool headin_site = p->mode == UNERF_FIELD_MCDROPOUT && (p->drop_sites & UNERF_DROP_HEADIN);
    const bool generic = gH != 64 || gHC != 64 || gG != 15 || gF != 2 || p->L != 16 || (gAD != 32 && headin_site);
    UNERF_REQUIRE(generic || p->L == 16, "field_fwd: L=%d", p->L);
    if (generic) {
        UNERF_REQUIRE(p->L >= 1 && p->L <= 32 && (gF == 2 || gF == 4) && gH >= 1 && gH <= 256 && gHC >= 1 && gHC <= 256 && gG >= 0 && gG <= 64,
                      "field_fwd: widths outside the any-width kernel's range (L=%d F=%d hidden=%d hidden_color=%d geo=%d)", p->L, gF, gH, gHC, gG);
        UNERF_REQUIRE(gF == 2 || !p->tcnn_levels, "field_fwd: features_per_level = 4 is built for the torch-layout grid");
        UNERF_REQUIRE(!features && !p->sample_major && !p->packed_out, "field_fwd: the any-width kernel writes the plain ray-major layout only");
        UNERF_REQUIRE(p->mode != UNERF_FIELD_MCDROPOUT || (gH <= 128 && gHC <= 128 && (!headin_site || 16 + gG + gAD <= 128)),
                      "field_fwd: a dropout site has at most 128 units (mask stream layout)");
    }
    UNERF_REQUIRE(!features || (p->mfma_blob && p->mode != UNERF_FIELD_LAPLACE),
                  "field_fwd: pre-gathered features are consumed by the MFMA kernel only (ACTIVE/MCDROPOUT with mfma_blob)");
    UNERF_REQUIRE(p->tcnn_levels || (p->log2T >= 1 && p->log2T <= 24), "field_fwd: bad log2T=%d", p->log2T);
    UNERF_REQUIRE(!(p->tcnn_levels && features), "field_fwd: pre-gathered feature planes are built for the torch-layout grid only");
    UNERF_REQUIRE(!p->grid_half || p->tcnn_levels, "field_fwd: grid_half (half2 rows, tcnn's half arithmetic) needs a tcnn-layout grid");
    UNERF_REQUIRE(R >= 0 && S >= 1, "field_fwd: bad R/S");
    UNERF_REQUIRE(!(near_plane < 0.f && features), "field_fwd: Euclidean bins (near_plane < 0) cannot be combined with pre-gathered features");
    UNERF_REQUIRE((uint64_t)(ray_offset + R) * (uint64_t)S < (1ull << 32),
                  "field_fwd: sample index exceeds 32 bits (RNG counter)");
    UNERF_REQUIRE(!p->sample_major || (p->mode != UNERF_FIELD_LAPLACE && (p->mfma16_blob || p->mfma_blob)),
                  "field_fwd: sample_major planes are written by the ACTIVE / MCDROPOUT matrix kernels only");
    if (R == 0) return UNERF_OK;
    FieldArgs a;
    a.origins = origins; a.dirs = directions; a.sbins = sbins; a.R = R; a.S = S;
    UNERF_REQUIRE_SPACING(spacing); a.lin = spacing;
    a.s_near = near_plane < 0.f ? -1.f : unerf_spacing_of(near_plane, spacing);   // < 0: sbins are Euclidean edges
    a.s_far = unerf_spacing_of(far_plane, spacing); a.ray_offset = ray_offset;
    a.p = *p; a.density = density; a.rgb = rgb; a.aux = aux; a.aux2 = aux2;
    if (a.p.n_lap_rgb <= 0) a.p.n_lap_rgb = a.p.n_lap;
    a.div_chunk = make_fastdiv(p->lap_chunk_rays > 0 ? (uint32_t)p->lap_chunk_rays : 1u);
    a.chunk0 = (uint32_t)ray_offset;
    a.features = features;
    a.box = make_norm_box(p->use_aabb, p->aabb);
    UNERF_REQUIRE(!(features && p->use_aabb), "field_fwd: pre-gathered feature planes are built for the contraction path only");
    {   // keep iff (signed 16-bit half) < thr_s = round((1-p) 65536) - 32768; p = 0 keeps everything (no masks)
        const long thr = lrint((1.0 - (double)p->p_drop) * 65536.0);
        a.drop_on = (p->mode == UNERF_FIELD_MCDROPOUT && p->K > 0 && thr < 65536) ? 1 : 0;
        a.drop_sites = a.drop_on ? (p->drop_sites ? p->drop_sites : (UNERF_DROP_TRUNK | UNERF_DROP_HEAD1)) : 0;
        const int32_t thr_s = (int32_t)(thr < 65536 ? thr : 65535) - 32768;
        a.keep_hi = (int32_t)((uint32_t)thr_s << 16);
        a.keep_pk = ((uint32_t)thr_s & 0xFFFFu) * 0x10001u;
    }
    a.drop_scale = 1.f / (1.f - p->p_drop);
    dim3 grid(blocks_for(R * (int64_t)S, 64)), block(64);
    hipStream_t st = (hipStream_t)stream;
    const int tc = p->tcnn_levels ? (p->grid_half ? 2 : 1) : 0;   // the TCNN template argument of the matrix kernels
    const bool f1 = p->f16_single != 0;
    UNERF_REQUIRE(!f1 || (p->mfma16_blob && !features && (p->mode != UNERF_FIELD_LAPLACE || (p->lap16_blob && p->n_lap <= 32 * LAP_BLOCKS))),
                  "field_fwd: f16_single needs mfma16_blob (and lap16_blob with n_lap <= 128 for LAPLACE), without pre-gathered features");
    if (generic) {
        a.p.hidden = gH; a.p.hidden_color = gHC; a.p.geo_dim = gG; a.p.feat_per_level = gF; a.p.app_dim = gAD;
        int rows = p->L * gF;
        for (int v : {gH, gHC, 16 + gG + gAD, 1 + gG + 1}) rows = v > rows ? v : rows;
        const size_t lds_bytes = (size_t)rows * 64 * 4 * 4;
        UNERF_REQUIRE(lds_bytes <= 160 * 1024, "field_fwd: %zu bytes of LDS for the any-width kernel", lds_bytes);
        const int want1 = p->mode == UNERF_FIELD_ACTIVE ? gG + 2 : (p->mode == UNERF_FIELD_MCDROPOUT ? gG + 1 : gG);
        UNERF_REQUIRE(p->out1 == want1, "field_fwd: out1=%d, expected %d for geo_dim=%d in this mode", p->out1, want1, gG);
        if (a.drop_sites & UNERF_DROP_HEADIN)
            UNERF_REQUIRE(p->h0_full_t && p->hb0_raw && p->app_embed, "field_fwd MCDROPOUT: UNERF_DROP_HEADIN needs h0_full_t / hb0_raw / app_embed");
#define UNERF_GENERIC_LAUNCH(MODE_)                                                                                      \
    do {                                                                                                                 \
        if (lds_bytes > 64 * 1024)                                                                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(field_kernel_generic<MODE_>),                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                       \
        hipLaunchKernelGGL((field_kernel_generic<MODE_>), grid, block, lds_bytes, st, a, rows);                          \
    } while (0)
        switch (p->mode) {
            case UNERF_FIELD_ACTIVE:
                UNERF_REQUIRE(aux, "field_fwd ACTIVE: aux (beta) must be non-null");
                UNERF_GENERIC_LAUNCH(UNERF_FIELD_ACTIVE);
                break;
            case UNERF_FIELD_MCDROPOUT:
                UNERF_REQUIRE(p->K >= 0 && p->p_drop >= 0.f && p->p_drop < 1.f, "field_fwd MCDROPOUT: bad K/p_drop");
                UNERF_GENERIC_LAUNCH(UNERF_FIELD_MCDROPOUT);
                break;
            case UNERF_FIELD_LAPLACE:
                UNERF_REQUIRE(aux && aux2 && p->ws_density && p->ws_rgb && p->n_lap >= 1, "field_fwd LAPLACE: need aux, aux2, ws_density, ws_rgb, n_lap>=1");
                UNERF_REQUIRE(p->lap_chunk_rays == 0 || (ray_offset + R - 1) / p->lap_chunk_rays < p->lap_sets, "field_fwd LAPLACE: rays reach past the sample sets");
                UNERF_GENERIC_LAUNCH(UNERF_FIELD_LAPLACE);
                break;
            default:
                unerf_set_error("field_fwd: unknown mode %d", p->mode);
                return UNERF_ERR_ARG;
        }
#undef UNERF_GENERIC_LAUNCH
        return unerf_check_launch("field_fwd (any-width kernel)");
    }
    switch (p->mode) {
        case UNERF_FIELD_ACTIVE:
            UNERF_REQUIRE(p->out1 == 17 && aux, "field_fwd ACTIVE: out1 must be 17 and aux (beta) non-null");
            if (p->mfma16_blob && !features && f1) {
                if (tc == 2) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_ACTIVE, 2, false, false, true>, MF_LDS_F16S, a, st);
                else if (tc) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_ACTIVE, 1, false, false, true>, MF_LDS_F16S, a, st);
                else launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_ACTIVE, false, false, false, true>, MF_LDS_F16S, a, st);
            } else if (p->mfma16_blob && !features) {
                if (tc == 2) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_ACTIVE, 2>, MF_LDS_F16, a, st);
                else if (tc) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_ACTIVE, 1>, MF_LDS_F16, a, st);
                else launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_ACTIVE, false>, MF_LDS_F16, a, st);
            } else if (p->mfma_blob) {
                if (features) launch_matrix_kernel(field_kernel_mfma<UNERF_FIELD_ACTIVE, true>, MF_LDS_FP32, a, st);
                else if (tc == 2) launch_matrix_kernel(field_kernel_mfma<UNERF_FIELD_ACTIVE, false, 2>, MF_LDS_FP32, a, st);
                else if (tc) launch_matrix_kernel(field_kernel_mfma<UNERF_FIELD_ACTIVE, false, 1>, MF_LDS_FP32, a, st);
                else launch_matrix_kernel(field_kernel_mfma<UNERF_FIELD_ACTIVE, false>, MF_LDS_FP32, a, st);
            } else {
                hipLaunchKernelGGL((field_kernel<UNERF_FIELD_ACTIVE>), grid, block, 64 * 64 * 4, st, a);
            }
            break;
        case UNERF_FIELD_MCDROPOUT:
            UNERF_REQUIRE(p->out1 == 16, "field_fwd MCDROPOUT: out1 must be 16");
            UNERF_REQUIRE(p->K >= 0 && p->p_drop >= 0.f && p->p_drop < 1.f, "field_fwd MCDROPOUT: bad K/p_drop");
            UNERF_REQUIRE((p->drop_sites & ~(UNERF_DROP_TRUNK | UNERF_DROP_HEAD0 | UNERF_DROP_HEAD1 | UNERF_DROP_HEADIN)) == 0,
                          "field_fwd MCDROPOUT: unknown bits in drop_sites=%d", p->drop_sites);
            if (a.drop_sites & UNERF_DROP_HEADIN) {   // dropout on the head's inputs: the VALU kernel (include/unerf.h)
                UNERF_REQUIRE(p->h0_full_t && p->hb0_raw && p->app_embed,
                              "field_fwd MCDROPOUT: UNERF_DROP_HEADIN needs h0_full_t / hb0_raw / app_embed");
                UNERF_REQUIRE(!features && !p->sample_major,
                              "field_fwd MCDROPOUT: UNERF_DROP_HEADIN has no feature-plane / sample-major form");
                hipLaunchKernelGGL((field_kernel<UNERF_FIELD_MCDROPOUT>), grid, block, 2 * 64 * 64 * 4, st, a);
            } else if (p->mfma16_blob && !features && f1) {
                const bool head0 = a.drop_on && a.drop_sites != (UNERF_DROP_TRUNK | UNERF_DROP_HEAD1);   // non-default sites
                if (tc == 2 && head0) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 2, true, true, true>, MF_LDS_F16S, a, st);
                else if (tc && head0) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 1, true, true, true>, MF_LDS_F16S, a, st);
                else if (head0) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, false, true, true, true>, MF_LDS_F16S, a, st);
                else if (tc == 2 && a.drop_on) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 2, false, true, true>, MF_LDS_F16S, a, st);
                else if (tc && a.drop_on) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 1, false, true, true>, MF_LDS_F16S, a, st);
                else if (a.drop_on) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, false, false, true, true>, MF_LDS_F16S, a, st);
                else if (tc == 2) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 2, false, false, true>, MF_LDS_F16S, a, st);
                else if (tc) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 1, false, false, true>, MF_LDS_F16S, a, st);
                else launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, false, false, false, true>, MF_LDS_F16S, a, st);
            } else if (p->mfma16_blob && !features) {
                const bool head0 = a.drop_on && a.drop_sites != (UNERF_DROP_TRUNK | UNERF_DROP_HEAD1);   // non-default sites
                if (tc == 2 && head0) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 2, true, true>, MF_LDS_F16S, a, st);
                else if (tc && head0) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 1, true, true>, MF_LDS_F16S, a, st);
                else if (head0) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, false, true, true>, MF_LDS_F16S, a, st);
                else if (tc == 2 && a.drop_on) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 2, false, true>, MF_LDS_F16S, a, st);
                else if (tc && a.drop_on) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 1, false, true>, MF_LDS_F16S, a, st);
                else if (a.drop_on) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, false, false, true>, MF_LDS_F16S, a, st);
                else if (tc == 2) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 2>, MF_LDS_F16, a, st);
                else if (tc) launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, 1>, MF_LDS_F16, a, st);
                else launch_matrix_kernel(field_kernel_mfma16<UNERF_FIELD_MCDROPOUT, false>, MF_LDS_F16, a, st);
            } else if (p->mfma_blob) {
                if (features) launch_matrix_kernel(field_kernel_mfma<UNERF_FIELD_MCDROPOUT, true>, MF_LDS_FP32, a, st);
                else if (tc == 2) launch_matrix_kernel(field_kernel_mfma<UNERF_FIELD_MCDROPOUT, false, 2>, MF_LDS_FP32, a, st);
                else if (tc) launch_matrix_kernel(field_kernel_mfma<UNERF_FIELD_MCDROPOUT, false, 1>, MF_LDS_FP32, a, st);
                else launch_matrix_kernel(field_kernel_mfma<UNERF_FIELD_MCDROPOUT, false>, MF_LDS_FP32, a, st);
            } else {
                hipLaunchKernelGGL((field_kernel<UNERF_FIELD_MCDROPOUT>), grid, block, 2 * 64 * 64 * 4, st, a);
            }
            break;
        case UNERF_FIELD_LAPLACE:
            UNERF_REQUIRE(p->out1 == 15 && aux && aux2 && p->ws_density && p->ws_rgb && p->n_lap >= 1,
                          "field_fwd LAPLACE: need out1=15, aux, aux2, ws_density, ws_rgb, n_lap>=1");
            UNERF_REQUIRE(!(p->lap_blob || p->lap16_blob) || a.p.n_lap_rgb <= 32 * LAP_BLOCKS || p->n_lap > 32 * LAP_BLOCKS,
                          "field_fwd LAPLACE: n_lap_rgb=%d rows do not fit the head blobs (at most %d); pass them as ws_* only",
                          a.p.n_lap_rgb, 32 * LAP_BLOCKS);
            if (p->lap_chunk_rays != 0) {   // per-chunk sample sets: 1-D tiles that never straddle two sets
                UNERF_REQUIRE(p->lap_chunk_rays > 0 && p->lap_chunk_rays % 32 == 0 && ray_offset % 32 == 0 && p->lap_sets >= 1,
                              "field_fwd LAPLACE: lap_chunk_rays=%d and ray_offset=%lld must be multiples of 32, lap_sets=%d >= 1",
                              p->lap_chunk_rays, (long long)ray_offset, p->lap_sets);
                UNERF_REQUIRE((ray_offset + R - 1) / p->lap_chunk_rays < p->lap_sets,
                              "field_fwd LAPLACE: rays [%lld, +%lld) reach past the %d sample sets of %d rays",
                              (long long)ray_offset, (long long)R, p->lap_sets, p->lap_chunk_rays);
                a.p.image_width = 0;
            }
            if (p->mfma16_blob && p->lap16_blob && p->n_lap <= 32 * LAP_BLOCKS && f1) {
                if (tc == 2) launch_matrix_kernel(field_kernel_mfma16_laplace<2, true>, MF_LDS_F16, a, st);
                else if (tc) launch_matrix_kernel(field_kernel_mfma16_laplace<1, true>, MF_LDS_F16, a, st);
                else launch_matrix_kernel(field_kernel_mfma16_laplace<false, true>, MF_LDS_F16, a, st);
            } else if (p->mfma16_blob && p->lap16_blob && p->n_lap <= 32 * LAP_BLOCKS) {
                if (tc == 2) launch_matrix_kernel(field_kernel_mfma16_laplace<2>, MF_LDS_F16, a, st);
                else if (tc) launch_matrix_kernel(field_kernel_mfma16_laplace<1>, MF_LDS_F16, a, st);
                else launch_matrix_kernel(field_kernel_mfma16_laplace<false>, MF_LDS_F16, a, st);
            } else if (p->mfma_blob && p->lap_blob && p->n_lap <= 32 * LAP_BLOCKS) {
                if (tc == 2) launch_matrix_kernel(field_kernel_mfma_laplace<false, 2>, MF_LDS_FP32, a, st);
                else if (tc) launch_matrix_kernel(field_kernel_mfma_laplace<false, 1>, MF_LDS_FP32, a, st);
                else launch_matrix_kernel(field_kernel_mfma_laplace<false>, MF_LDS_FP32, a, st);
            } else {
                hipLaunchKernelGGL((field_kernel<UNERF_FIELD_LAPLACE>), grid, block, 64 * 64 * 4, st, a);
            }
            break;
        default:
            unerf_set_error("field_fwd: unknown mode %d", p->mode);
            return UNERF_ERR_ARG;
    }
    return unerf_check_launch("field_fwd");
}

// --------------------------------------------------------------------------------------
// 5d. Laplace GGN fitting (NerfactoLaplaceModel.compute_hessian_naive, laplace_model.py:343-400).
// The reference multiplies the GGN onto each of the 260 unit vectors (one double-backward per
// parameter per batch).  The loss is the summed MSE, so its Hessian w.r.t. a rendered pixel is 2 I
// and the diagonal is 2 * sum_{ray,channel} J^2 with the Jacobian of the rendered colour
//   C_c = sum_i w_i c_ic + (1 - sum_i w_i) c_{S-1,c}          (background = last sample)
// in closed form.  Last colour layer, row c, input unit k (h = colour hidden, s' = c(1-c)):
//   dC_c/dW_ck = sum_i (w_i + [i = S-1] T_end) s'_ic h_ik
// density layer, unit k (x = base_mlp output, sigma = exp(.) * selector so dsigma/dpre = sigma; with the softplus
// activation dsigma/dpre = 1 - exp(-sigma) takes the place of the last factor sigma_i):
//   dC_c/dw_k = sum_i delta_i [ (1 - alpha_i) T_i c_ic - sum_{j>i} w_j c_jc - T_end c_{S-1,c} ] sigma_i x_ik
// (bias: x = h = 1).  One wave per ray: lane i prepares sample i's scalars with wave scans, then lane k
// accumulates unit k over the samples.  Per-wave partial sums, reduced in a fixed order (deterministic).
// --------------------------------------------------------------------------------------
#define GGN_PARAMS 260
struct GgnArgs {
    const float* sbins;
    int64_t R;
    int S;
    float s_near, s_far;
    int lin;   // UNERF_SPACING_*: 1 = identity spacing (UniformSampler)
    const float* sigma;  // [R,S]
    int softplus;        // density activation: 0 trunc_exp (dsigma/dpre = sigma), 1 softplus (dsigma/dpre = 1 - exp(-sigma))
    const float* rgb;    // [R,S,3]
    const float* X;      // [R,S,64] base_mlp output
    const float* Hc;     // [R,S,64] colour hidden (input of mlp_rgb_ll)
    float* partials;     // [waves][260]
    int bg_mode;         // UNERF_BG_*: the renderer's background (LAST_SAMPLE: the formulas above; COLOR / NONE: a constant
    float bg[3];         // colour (zero for NONE) in place of c_{S-1}, which then takes no gradient from the T_end term)
};

__global__ __launch_bounds__(256) void laplace_ggn_kernel(GgnArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wv, nw = (int64_t)gridDim.x * 4;
    const int S = a.S;
    const bool in = lane < S;
    float acc_d = 0.f, acc_db = 0.f, acc_r[3] = {0.f, 0.f, 0.f}, acc_rb[3] = {0.f, 0.f, 0.f};
    for (int64_t r = gw; r < a.R; r += nw) {
        const float* sb = a.sbins + r * (S + 1);
        const int i = in ? lane : S - 1;
        const float e0 = unerf_s2e(sb[i], a.s_near, a.s_far, a.lin), e1 = unerf_s2e(sb[i + 1], a.s_near, a.s_far, a.lin);
        const float delta = e1 - e0;
        const float sig = in ? a.sigma[r * S + i] : 0.f;
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = unerf_nan_to_num(a.rgb[(r * S + i) * 3 + k]);
        // dsigma / d(pre-activation): exp -> sigma itself; softplus -> sigmoid(pre) = 1 - exp(-softplus(pre)), and the
        // selector (0 or 1) that multiplies sigma carries over (sigma = 0 -> derivative 0)
        const float dsig = a.softplus ? -expm1f(-sig) : sig;
        const float dd = in ? delta * sig : 0.f;
        const float em = expf(-dd);  // 1 - alpha
        const float T = expf(-group_excl_scan<64>(dd, lane));
        const float w = in ? unerf_nan_to_num((1.f - em) * T) : 0.f;
        const float Tf = 1.f - group_sum<64>(w);
        float g[3], q[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool last_bg = a.bg_mode == UNERF_BG_LAST_SAMPLE;
            const float bg = last_bg ? __shfl(c[k], S - 1, 64) : a.bg[k];
            const float wc = w * c[k];
            const float incl = group_incl_scan<64>(wc, lane);
            const float tot = __shfl(incl, 63, 64);
            const float pred = tot + Tf * bg;
            // eval-mode renderer clamps to [0,1]: the gradient passes only inside (torch.clamp backward)
            const float live = (pred >= 0.f && pred <= 1.f && in) ? 1.f : 0.f;
            g[k] = live * delta * (em * T * c[k] - (tot - incl) - Tf * bg) * dsig;
            const float wt = w + ((last_bg && lane == S - 1) ? Tf : 0.f);
            q[k] = live * wt * c[k] * (1.f - c[k]);
        }
        float M[3] = {0.f, 0.f, 0.f}, N[3] = {0.f, 0.f, 0.f};
        const float* xr = a.X + r * S * 64 + lane;
        const float* hr = a.Hc + r * S * 64 + lane;
        for (int s = 0; s < S; ++s) {
            const float x = xr[s * 64], hh = hr[s * 64];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                M[k] = fmaf(__shfl(g[k], s, 64), x, M[k]);
                N[k] = fmaf(__shfl(q[k], s, 64), hh, N[k]);
            }
        }
        acc_d += 2.f * (M[0] * M[0] + M[1] * M[1] + M[2] * M[2]);
        float mb2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            acc_r[k] += 2.f * N[k] * N[k];
            const float mb = group_sum<64>(g[k]), nb = group_sum<64>(q[k]);
            mb2 += mb * mb;
            acc_rb[k] += 2.f * nb * nb;
        }
        acc_db += 2.f * mb2;
    }
    float* out = a.partials + gw * GGN_PARAMS;
    out[lane] = acc_d;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[65 + k * 64 + lane] = acc_r[k];
    if (lane == 0) {
        out[64] = acc_db;
#pragma unroll
        for (int k = 0; k < 3; ++k) out[65 + 192 + k] = acc_rb[k];
    }
}

__global__ void laplace_ggn_reduce_kernel(const float* partials, int waves, float* ggn_density, float* ggn_rgb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= GGN_PARAMS) return;
    float sum = 0.f;
    for (int w = 0; w < waves; ++w) sum += partials[(size_t)w * GGN_PARAMS + t];
    if (t < 65) ggn_density[t] += sum;
    else ggn_rgb[t - 65] += sum;
}

static int ggn_blocks(int64_t R) {
    int64_t b = (R + 3) / 4;
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" size_t unerf_laplace_ggn_workspace_bytes(int64_t R, int S) {
    if (R <= 0 || S <= 0) return 0;
    const size_t N = (size_t)R * (size_t)S;
    return (N * (64 + 64 + 1 + 3) + (size_t)ggn_blocks(R) * 4 * GGN_PARAMS) * sizeof(float);
}

extern "C" int unerf_laplace_ggn_diag(const float* origins, const float* directions, const float* sbins, int64_t R,
                                      int S, float near_plane, float far_plane, int spacing, const unerf_field_params* p,
                                      int background, const float* background_rgb, void* workspace, size_t workspace_bytes,
                                      float* ggn_density, float* ggn_rgb, void* stream) {
    UNERF_REQUIRE(R >= 0 && S >= 1 && S <= 64, "laplace_ggn_diag: S=%d outside [1,64]", S);
    if (R == 0) return UNERF_OK;
    UNERF_REQUIRE(origins && directions && sbins && p && workspace && ggn_density && ggn_rgb,
                  "laplace_ggn_diag: null pointer");
    UNERF_REQUIRE(p->mode == UNERF_FIELD_LAPLACE && p->out1 == 15 && p->L == 16 && p->mfma_blob && p->ws_density &&
                      p->ws_rgb && p->table && (p->scalings || p->tcnn_levels),
                  "laplace_ggn_diag: needs a LAPLACE field with mfma_blob and the mean last layers in ws_density[65] / "
                  "ws_rgb[195]");
    UNERF_REQUIRE(p->tcnn_levels || (p->log2T >= 1 && p->log2T <= 24), "laplace_ggn_diag: bad log2T=%d", p->log2T);
    UNERF_REQUIRE((uint64_t)R * (uint64_t)S < (1ull << 32), "laplace_ggn_diag: R*S exceeds 32 bits, split the batch");
    UNERF_REQUIRE(workspace_bytes >= unerf_laplace_ggn_workspace_bytes(R, S),
                  "laplace_ggn_diag: workspace %zu < %zu bytes", workspace_bytes,
                  unerf_laplace_ggn_workspace_bytes(R, S));
    const size_t N = (size_t)R * (size_t)S;
    float* X = static_cast<float*>(workspace);
    float* Hc = X + N * 64;
    float* sigma = Hc + N * 64;
    float* col = sigma + N;
    float* partials = col + N * 3;
    hipStream_t st = (hipStream_t)stream;
    FieldArgs a;
    a.origins = origins; a.dirs = directions; a.sbins = sbins; a.R = R; a.S = S;
    UNERF_REQUIRE_SPACING(spacing); a.lin = spacing; a.s_near = unerf_spacing_of(near_plane, spacing); a.s_far = unerf_spacing_of(far_plane, spacing); a.ray_offset = 0;
    a.p = *p; a.density = sigma; a.rgb = col; a.aux = X; a.aux2 = Hc; a.features = nullptr;
    a.keep_hi = 0; a.keep_pk = 0; a.drop_on = 0; a.drop_sites = 0; a.drop_scale = 1.f;
    a.box = make_norm_box(p->use_aabb, p->aabb);
    a.p.image_width = 0;   // 1-D tiles
    if (p->tcnn_levels && p->grid_half) launch_matrix_kernel(field_kernel_mfma_laplace<true, 2>, MF_LDS_FP32, a, st);
    else if (p->tcnn_levels) launch_matrix_kernel(field_kernel_mfma_laplace<true, 1>, MF_LDS_FP32, a, st);
    else launch_matrix_kernel(field_kernel_mfma_laplace<true>, MF_LDS_FP32, a, st);
    GgnArgs g;
    g.sbins = sbins; g.R = R; g.S = S; g.s_near = a.s_near; g.s_far = a.s_far; g.lin = a.lin;
    g.sigma = sigma; g.softplus = p->lap_softplus; g.rgb = col; g.X = X; g.Hc = Hc; g.partials = partials;
    if (int rc = unerf_set_background(g, background, background_rgb, "laplace_ggn_diag")) return rc;
    const int blocks = ggn_blocks(R);
    hipLaunchKernelGGL(laplace_ggn_kernel, dim3(blocks), dim3(256), 0, st, g);
    hipLaunchKernelGGL(laplace_ggn_reduce_kernel, dim3(2), dim3(256), 0, st, partials, blocks * 4, ggn_density, ggn_rgb);
    return unerf_check_launch("laplace_ggn_diag");
}

// ======================================================================================
// 6. per-ray groups of 16 lanes x SPL samples: get_weights, composite, laplace depth draws
// ======================================================================================
template <int SPL>
__device__ __forceinline__ void group_weights(const float (&dens)[SPL], const float (&delta)[SPL], int l16,
                                              float (&w)[SPL]) {
    float dd[SPL], lex[SPL], ls = 0.f;
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        dd[e] = delta[e] * dens[e];
        lex[e] = ls;
        ls += dd[e];
    }
    float carry = group_excl_scan<16>(ls, l16);
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        float alpha = 1.f - unerf_exp(-dd[e]);
        float T = unerf_exp(-(carry + lex[e]));
        w[e] = unerf_nan_to_num(alpha * T);
    }
}

struct CompArgs {
    const float* density;
    const float* rgb;
    const float* beta;
    const float* walt;
    const float* sbins;
    int B;
    int64_t R;
    int S;
    float s_near, s_far;
    int lin;   // UNERF_SPACING_*: 1 = identity spacing (UniformSampler)
    const float* clip;
    int64_t ray_offset, chunk_rays;
    float* out;
    int bg_mode;      // UNERF_BG_*: what RGBRenderer blends behind the samples
    float bg[3];      // UNERF_BG_COLOR
    int32_t* flag;    // nonfinite_flag (may be null): |= 1 when a density or colour read here is NaN
};

// one (pass b, ray r) composite, executed by a 16-lane group; g = b*R + r.  Results are replicated on all 16 lanes.
// RAGGED: S is not a multiple of 16 (SPL = ceil(S / 16)); the slots k >= S of the last lanes are masked: their bin
// edges collapse onto edge S (delta = 0), density and weight are 0, and they are excluded from the median count
// and from the "last sample" background colour.  The aligned instantiations are unchanged.
// The pass-independent part of a ray's composite: this lane's bin widths and mid-points (SPL + 1 spacing->Euclidean
// conversions, one IEEE division each).  The K-pass kernel computes it once per ray, not once per pass.
template <int SPL>
struct CompGeom {
    float delta[SPL], steps[SPL];
};
template <int SPL, bool RAGGED>
__device__ __forceinline__ CompGeom<SPL> composite_geom(const CompArgs& a, int64_t r, int l16) {
    const int S = a.S, k0 = l16 * SPL;
    const float* sb = a.sbins + r * (S + 1);
    float eu[SPL + 1];
    CompGeom<SPL> gm;
#pragma unroll
    for (int e = 0; e <= SPL; ++e) eu[e] = unerf_s2e(sb[RAGGED ? min(k0 + e, S) : k0 + e], a.s_near, a.s_far, a.lin);
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        gm.delta[e] = eu[e + 1] - eu[e];
        gm.steps[e] = (eu[e] + eu[e + 1]) / 2.f;
    }
    return gm;
}

template <int SPL, bool RAGGED = false, bool PACKED = false>
__device__ __forceinline__ void composite_one(const CompArgs& a, int64_t g, int64_t r, int l16, float (&o8)[8],
                                              const CompGeom<SPL>& gm) {
    const int S = a.S, k0 = l16 * SPL;
    float delta[SPL], steps[SPL], dens[SPL], w[SPL];
    struct Rgb { float r, g, b; };
    Rgb col[SPL];
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        delta[e] = gm.delta[e];
        steps[e] = gm.steps[e];
    }
    if (PACKED) {   // packed rows (sigma, r, g, b) [B,R,S,4] (unerf_field_params.packed_out): one 16-byte load per sample
        // (a compile-time switch: with both layouts in one kernel the K-pass form needs 72 registers instead of 59)
#pragma unroll
        for (int e = 0; e < SPL; ++e) {
            const float4 v = (!RAGGED || k0 + e < S) ? reinterpret_cast<const float4*>(a.rgb)[g * S + k0 + e]
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            dens[e] = v.x;
            col[e] = Rgb{v.y, v.z, v.w};
        }
    } else {
#pragma unroll
        for (int e = 0; e < SPL; ++e) dens[e] = (!RAGGED || k0 + e < S) ? a.density[g * S + k0 + e] : 0.f;
        // colours requested with the densities, one 12-byte load per sample (the three channels as separate dword
        // loads tripled the texture-unit work of this kernel), and consumed after the scan
#pragma unroll
        for (int e = 0; e < SPL; ++e)
            col[e] = (!RAGGED || k0 + e < S) ? reinterpret_cast<const Rgb*>(a.rgb)[g * S + k0 + e] : Rgb{0.f, 0.f, 0.f};
    }
    if (a.flag) {   // uniform.  NaN inputs: the signature of an f16 operand overflow in the field kernel (|x| >= 65504 ->
        // hi = inf, lo = -inf -> NaN), which the nan_to_num calls below would otherwise turn into a plausible pixel.
        // One ballot per wave, one atomic per OFFENDING wave.
        bool bad = false;
#pragma unroll
        for (int e = 0; e < SPL; ++e)
            bad |= (dens[e] != dens[e]) | (col[e].r != col[e].r) | (col[e].g != col[e].g) | (col[e].b != col[e].b);
        const uint64_t m = __builtin_amdgcn_ballot_w64(bad);
        if (m != 0 && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicOr(a.flag, 1);
    }
    group_weights<SPL>(dens, delta, l16, w);

    float cr = 0.f, cg = 0.f, cb = 0.f, accw = 0.f, uvar = 0.f, lr = 0.f, lg = 0.f, lb = 0.f;
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        if (RAGGED && k0 + e >= S) continue;
        float r0 = unerf_nan_to_num(col[e].r), g0 = unerf_nan_to_num(col[e].g), b0 = unerf_nan_to_num(col[e].b);
        cr += w[e] * r0;
        cg += w[e] * g0;
        cb += w[e] * b0;
        accw += w[e];
        if (a.beta) uvar += (w[e] * w[e]) * a.beta[r * S + k0 + e];
        lr = r0; lg = g0; lb = b0;  // after the loop: this lane's last (real) sample
    }
    cr = group_sum<16>(cr);
    cg = group_sum<16>(cg);
    cb = group_sum<16>(cb);
    accw = group_sum<16>(accw);
    uvar = group_sum<16>(uvar);
    if (a.bg_mode == UNERF_BG_LAST_SAMPLE) {   // uniform; background_color = "last_sample", the nerfacto default
        const int last_lane = RAGGED ? (S - 1) / SPL : 15;   // the lane that holds sample S-1
        float bgr = __shfl(lr, last_lane, 16), bgg = __shfl(lg, last_lane, 16), bgb = __shfl(lb, last_lane, 16);
        cr = cr + bgr * (1.f - accw);
        cg = cg + bgg * (1.f - accw);
        cb = cb + bgb * (1.f - accw);
    } else if (a.bg_mode == UNERF_BG_COLOR) {  // "white" / "black"
        cr = cr + a.bg[0] * (1.f - accw);
        cg = cg + a.bg[1] * (1.f - accw);
        cb = cb + a.bg[2] * (1.f - accw);
    }                                          // UNERF_BG_NONE ("random" at eval): no blending
    cr = fminf(fmaxf(cr, 0.f), 1.f);
    cg = fminf(fmaxf(cg, 0.f), 1.f);
    cb = fminf(fmaxf(cb, 0.f), 1.f);

    // depth-side weights: the laplace mean sampled weights when given
    float wd[SPL];
#pragma unroll
    for (int e = 0; e < SPL; ++e)
        wd[e] = (RAGGED && k0 + e >= S) ? 0.f : (a.walt ? a.walt[r * S + k0 + e] : w[e]);
    float ls = 0.f, lc[SPL], wt = 0.f;
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        ls += wd[e];
        lc[e] = ls;
        wt += wd[e] * steps[e];
    }
    float tot = group_incl_scan<16>(ls, l16);
    float cbase = dpp_f<0x111>(tot);   // row_shr:1, lane 0 of the group reads 0
    float acc = __shfl(tot, 15, 16);
    wt = group_sum<16>(wt);
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < SPL; ++e) cnt += ((!RAGGED || k0 + e < S) && (cbase + lc[e]) < 0.5f) ? 1 : 0;
    cnt = group_sum_i<16>(cnt);
    int idx = min(cnt, S - 1);
    int owner = idx / SPL, slot = idx - owner * SPL;
    float depth = 0.f;
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        float got = __shfl(steps[e], owner, 16);
        if (e == slot) depth = got;
    }
    float dv = 0.f;
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        float df = steps[e] - depth;
        dv += wd[e] * (df * df);
    }
    dv = group_sum<16>(dv) + 1e-5f;
    float ed = wt / (acc + 1e-10f);
    if (a.clip) {
        int64_t chunk = (a.ray_offset + r) / a.chunk_rays;
        ed = fminf(fmaxf(ed, a.clip[chunk * 2 + 0]), a.clip[chunk * 2 + 1]);
    }
    o8[0] = cr; o8[1] = cg; o8[2] = cb; o8[3] = acc;
    o8[4] = depth; o8[5] = ed; o8[6] = uvar; o8[7] = dv;
}

template <int SPL, bool RAGGED, bool PACKED>
__device__ __forceinline__ void composite_kernel_body(const CompArgs& a) {
    const int l16 = threadIdx.x & 15;
    int64_t g = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t G = (int64_t)a.B * a.R;
    const bool ok = g < G;
    if (!ok) g = G - 1;
    float o8[8];
    const int64_t r = g % a.R;
    composite_one<SPL, RAGGED, PACKED>(a, g, r, l16, o8, composite_geom<SPL, RAGGED>(a, r, l16));
    if (ok && l16 == 0) {
        float4* o = reinterpret_cast<float4*>(a.out + g * 8);
        o[0] = make_float4(o8[0], o8[1], o8[2], o8[3]);
        o[1] = make_float4(o8[4], o8[5], o8[6], o8[7]);
    }
}
template <int SPL, bool RAGGED = false>
__global__ __launch_bounds__(256) void composite_kernel(CompArgs a) {
    composite_kernel_body<SPL, RAGGED, false>(a);
}
template <int SPL, bool RAGGED = false>
__global__ __launch_bounds__(256) void composite_kernel_packed(CompArgs a) {
    composite_kernel_body<SPL, RAGGED, true>(a);
}

// K-pass form: one 16-lane group walks the B <= 16 passes of a ray, lane b keeps pass b's eight
// outputs, and the per-pixel mean and unbiased variance over the passes (two-pass, like
// torch.stack(...).mean(0) / .var(0), mcdropout_models.py:121-126) come out of two group reductions:
// the [B,R,8] per-pass images never touch HBM.
template <int SPL, bool RAGGED, bool PACKED>
__device__ __forceinline__ void composite_moments_body(const CompArgs& a, float* __restrict__ mean_out,
                                                       float* __restrict__ var_out) {
    const int l16 = threadIdx.x & 15;
    int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = r < a.R;
    if (!ok) r = a.R - 1;
    float mine[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) mine[c] = 0.f;
    const CompGeom<SPL> gm = composite_geom<SPL, RAGGED>(a, r, l16);   // bin edges are the same in every pass
    for (int b = 0; b < a.B; ++b) {
        float o8[8];
        composite_one<SPL, RAGGED, PACKED>(a, (int64_t)b * a.R + r, r, l16, o8, gm);
        if (l16 == b) {
#pragma unroll
            for (int c = 0; c < 8; ++c) mine[c] = o8[c];
        }
    }
    const float invB = 1.f / (float)a.B, invB1 = 1.f / (float)(a.B - 1);
    float m8[8], v8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        m8[c] = group_sum<16>(mine[c]) * invB;
        float d = (l16 < a.B) ? mine[c] - m8[c] : 0.f;
        v8[c] = group_sum<16>(d * d) * invB1;
    }
    if (ok && l16 == 0) {
        float4* mo = reinterpret_cast<float4*>(mean_out + r * 8);
        float4* vo = reinterpret_cast<float4*>(var_out + r * 8);
        mo[0] = make_float4(m8[0], m8[1], m8[2], m8[3]);
        mo[1] = make_float4(m8[4], m8[5], m8[6], m8[7]);
        vo[0] = make_float4(v8[0], v8[1], v8[2], v8[3]);
        vo[1] = make_float4(v8[4], v8[5], v8[6], v8[7]);
    }
}
template <int SPL, bool RAGGED = false>
__global__ __launch_bounds__(256) void composite_moments_kernel(CompArgs a, float* __restrict__ mean_out,
                                                                float* __restrict__ var_out) {
    composite_moments_body<SPL, RAGGED, false>(a, mean_out, var_out);
}
template <int SPL, bool RAGGED = false>
__global__ __launch_bounds__(256) void composite_moments_kernel_packed(CompArgs a, float* __restrict__ mean_out,
                                                                       float* __restrict__ var_out) {
    composite_moments_body<SPL, RAGGED, true>(a, mean_out, var_out);
}

// Samples per lane: S = 16 * SPL for SPL in {1,2,3,4,6,8,16} (every nerfacto sample count) takes the aligned
// kernels; any other 1 <= S <= 256 takes the RAGGED instantiation of the next SPL up, whose slots k >= S are masked.
static inline int unerf_spl_for(int S) {
    const int want = (S + 15) / 16;
    for (int spl : {1, 2, 3, 4, 6, 8, 16})
        if (spl >= want) return spl;
    return 0;
}
#define UNERF_SPL_CASE(N, KERNEL, ...)                                                                              \
    case N:                                                                                                         \
        if (ragged_) hipLaunchKernelGGL((KERNEL<N, true>), __VA_ARGS__);                                            \
        else hipLaunchKernelGGL((KERNEL<N, false>), __VA_ARGS__);                                                   \
        break;
#define UNERF_DISPATCH_SPL(S, KERNEL, ...)                                                                          \
    {                                                                                                               \
        const int spl_ = unerf_spl_for(S);                                                                          \
        const bool ragged_ = spl_ * 16 != (S);                                                                      \
        switch (spl_) {                                                                                             \
            UNERF_SPL_CASE(1, KERNEL, __VA_ARGS__)                                                                  \
            UNERF_SPL_CASE(2, KERNEL, __VA_ARGS__)                                                                  \
            UNERF_SPL_CASE(3, KERNEL, __VA_ARGS__)                                                                  \
            UNERF_SPL_CASE(4, KERNEL, __VA_ARGS__)                                                                  \
            UNERF_SPL_CASE(6, KERNEL, __VA_ARGS__)                                                                  \
            UNERF_SPL_CASE(8, KERNEL, __VA_ARGS__)                                                                  \
            UNERF_SPL_CASE(16, KERNEL, __VA_ARGS__)                                                                 \
            default:                                                                                                \
                unerf_set_error("samples per ray S=%d outside [1,256]", (S));                                       \
                return UNERF_ERR_ARG;                                                                               \
        }                                                                                                           \
    }

extern "C" int unerf_composite_var(const float* density, const float* rgb, const float* beta, const float* weights_alt,
                                   const float* sbins, int B, int64_t R, int S, float near_plane, float far_plane, int spacing,
                                   const float* clip_minmax, int64_t ray_offset, int64_t chunk_rays, int background,
                                   const float* background_rgb, int32_t* nonfinite_flag, float* out, void* stream) {
    UNERF_REQUIRE(R == 0 || (rgb && sbins && out), "composite_var: null pointer");   // density NULL: rgb holds packed rows
    UNERF_REQUIRE(B >= 1 && R >= 0, "composite_var: bad B/R");
    UNERF_REQUIRE(S >= 1 && S <= 256, "composite_var: S=%d outside [1,256]", S);
    UNERF_REQUIRE(!clip_minmax || chunk_rays > 0, "composite_var: chunk_rays must be > 0 with clip_minmax");
    if (R == 0) return UNERF_OK;
    CompArgs a;
    a.density = density; a.rgb = rgb; a.beta = beta; a.walt = weights_alt; a.sbins = sbins; a.B = B; a.R = R; a.S = S;
    UNERF_REQUIRE_SPACING(spacing); a.lin = spacing; a.s_near = unerf_spacing_of(near_plane, spacing); a.s_far = unerf_spacing_of(far_plane, spacing);
    a.clip = clip_minmax; a.ray_offset = ray_offset; a.chunk_rays = chunk_rays; a.out = out;
    if (int rc = unerf_set_background(a, background, background_rgb, "composite_var")) return rc;
    a.flag = nonfinite_flag;
    dim3 grid(blocks_for((int64_t)B * R, 16)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (density) {
        UNERF_DISPATCH_SPL(S, composite_kernel, grid, block, 0, st, a);
    } else {
        UNERF_DISPATCH_SPL(S, composite_kernel_packed, grid, block, 0, st, a);
    }
    return unerf_check_launch("composite_var");
}

extern "C" int unerf_composite_moments(const float* density, const float* rgb, const float* sbins, int B, int64_t R, int S,
                                       float near_plane, float far_plane, int spacing, const float* clip_minmax, int64_t ray_offset,
                                       int64_t chunk_rays, int background, const float* background_rgb,
                                       int32_t* nonfinite_flag, float* mean_out, float* var_out, void* stream) {
    UNERF_REQUIRE(R == 0 || (rgb && sbins && mean_out && var_out), "composite_moments: null pointer");   // density NULL: packed rows
    UNERF_REQUIRE(B >= 1 && B <= 16 && R >= 0, "composite_moments: B=%d outside [1,16] (use composite_var + moments)", B);
    UNERF_REQUIRE(S >= 1 && S <= 256, "composite_moments: S=%d outside [1,256]", S);
    UNERF_REQUIRE(!clip_minmax || chunk_rays > 0, "composite_moments: chunk_rays must be > 0 with clip_minmax");
    if (R == 0) return UNERF_OK;
    CompArgs a;
    a.density = density; a.rgb = rgb; a.beta = nullptr; a.walt = nullptr; a.sbins = sbins; a.B = B; a.R = R; a.S = S;
    UNERF_REQUIRE_SPACING(spacing); a.lin = spacing; a.s_near = unerf_spacing_of(near_plane, spacing); a.s_far = unerf_spacing_of(far_plane, spacing);
    a.clip = clip_minmax; a.ray_offset = ray_offset; a.chunk_rays = chunk_rays; a.out = nullptr;
    if (int rc = unerf_set_background(a, background, background_rgb, "composite_moments")) return rc;
    a.flag = nonfinite_flag;
    dim3 grid(blocks_for(R, 16)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (density) {
        UNERF_DISPATCH_SPL(S, composite_moments_kernel, grid, block, 0, st, a, mean_out, var_out);
    } else {
        UNERF_DISPATCH_SPL(S, composite_moments_kernel_packed, grid, block, 0, st, a, mean_out, var_out);
    }
    return unerf_check_launch("composite_moments");
}

// ---- composite over sample-major planes: one lane per ray, front to back --------------------------------------
// Input = the planes unerf_field_fwd writes with sample_major = 1: density [B,S,R], rgb [B,S,3,R], beta [S,R].  A lane
// walks its ray's samples in order (every load is a coalesced row of 64 consecutive rays), so get_weights is the
// plain sequential recurrence -- no cross-lane scan -- and the K passes of a ray are reduced to mean / unbiased
// variance in the same thread (MOMENTS).  Passes are walked four at a time so that the bin edges (one IEEE
// division each) are converted once per group, not once per pass.
//   depth_var = sum_i w_i (t_i - d_median)^2 needs d_median, which is only known after the walk; instead of a second
//   walk (two more exp per sample) the three moments sum w, sum w t, sum w t^2 are kept in fp64 (products of fp32
//   numbers are exact there) and combined at the end: sum w t^2 - 2 d sum w t + d^2 sum w.
//   mean / variance over the passes: sums of (x - x_0) and (x - x_0)^2 around the first pass's value x_0 -- the
//   two-pass result of torch.stack(...).var(0) up to fp32 rounding, without keeping the B per-pass values.
struct CompSmArgs {
    const float* density;
    const float* rgb;
    const float* beta;
    const float* sbins;
    int B;
    int64_t R;
    int S;
    float s_near, s_far;
    int lin;   // UNERF_SPACING_*: 1 = identity spacing (UniformSampler)
    const float* clip;
    int64_t ray_offset, chunk_rays;
    float* out;       // [B,R,8] (MOMENTS = false)
    float* mean_out;  // [R,8]
    float* var_out;   // [R,8]
    int bg_mode;      // as CompArgs
    float bg[3];
    int32_t* flag;
};

#define CSM_G 4   // passes per walk

template <bool MOMENTS>
__global__ __launch_bounds__(256) void composite_sm_kernel(CompSmArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.R) return;
    const int S = a.S;
    const int64_t R = a.R;
    const float* sb = a.sbins + r * (S + 1);
    float clip_lo = 0.f, clip_hi = 0.f;
    if (a.clip) {
        const int64_t chunk = (a.ray_offset + r) / a.chunk_rays;
        clip_lo = a.clip[chunk * 2 + 0];
        clip_hi = a.clip[chunk * 2 + 1];
    }
    bool saw_nan = false;         // -> a.flag (see CompArgs)
    float x0[8], sd[8], sd2[8];   // moments over the passes, around pass 0
#pragma unroll
    for (int c = 0; c < 8; ++c) x0[c] = sd[c] = sd2[c] = 0.f;
    for (int b0 = 0; b0 < a.B; b0 += CSM_G) {
        const int nb = min(CSM_G, a.B - b0);
        float cum[CSM_G], cr[CSM_G], cg[CSM_G], cb[CSM_G], accw[CSM_G], wt[CSM_G], uvar[CSM_G], depth[CSM_G];
        float lr[CSM_G], lg[CSM_G], lb[CSM_G];
        double m0[CSM_G], m1[CSM_G], m2[CSM_G];
        bool found[CSM_G];
#pragma unroll
        for (int j = 0; j < CSM_G; ++j) {
            cum[j] = cr[j] = cg[j] = cb[j] = accw[j] = wt[j] = uvar[j] = depth[j] = 0.f;
            lr[j] = lg[j] = lb[j] = 0.f;
            m0[j] = m1[j] = m2[j] = 0.0;
            found[j] = false;
        }
        float e0 = unerf_s2e(sb[0], a.s_near, a.s_far, a.lin);
        float step = 0.f;
        for (int s = 0; s < S; ++s) {
            const float e1 = unerf_s2e(sb[s + 1], a.s_near, a.s_far, a.lin);
            const float delta = e1 - e0;
            step = (e0 + e1) / 2.f;
            e0 = e1;
            const double t64 = (double)step, t264 = t64 * t64;
            const float bt = a.beta ? a.beta[(int64_t)s * R + r] : 0.f;
#pragma unroll
            for (int j = 0; j < CSM_G; ++j) {
                if (j < nb) {  // uniform
                    const int64_t plane = (int64_t)(b0 + j) * S + s;
                    const float dens = a.density[plane * R + r];
                    const float* cp = a.rgb + plane * 3 * R + r;
                    const float r0 = unerf_nan_to_num(cp[0]), g0 = unerf_nan_to_num(cp[R]), bl0 = unerf_nan_to_num(cp[2 * R]);
                    saw_nan |= (dens != dens) | (cp[0] != cp[0]) | (cp[R] != cp[R]) | (cp[2 * R] != cp[2 * R]);
                    const float dd = delta * dens;
                    const float alpha = 1.f - unerf_exp(-dd);
                    const float T = unerf_exp(-cum[j]);
                    cum[j] += dd;
                    const float w = unerf_nan_to_num(alpha * T);
                    cr[j] += w * r0;
                    cg[j] += w * g0;
                    cb[j] += w * bl0;
                    accw[j] += w;
                    wt[j] += w * step;
                    uvar[j] += (w * w) * bt;
                    if (!found[j] && !(accw[j] < 0.5f)) {   // first index whose inclusive cumsum reaches 0.5
                        found[j] = true;
                        depth[j] = step;
                    }
                    const double w64 = (double)w;
                    m0[j] += w64;
                    m1[j] = fma(w64, t64, m1[j]);
                    m2[j] = fma(w64, t264, m2[j]);
                    lr[j] = r0; lg[j] = g0; lb[j] = bl0;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CSM_G; ++j) {
            if (j < nb) {
                if (!found[j]) depth[j] = step;   // clamp(searchsorted, 0, S-1): the last sample's mid-point
                float o8[8];
                // RGBRenderer background (see unerf_set_background): last sample | constant colour | none
                const bool ls = a.bg_mode == UNERF_BG_LAST_SAMPLE, none = a.bg_mode == UNERF_BG_NONE;
                const float br = ls ? lr[j] : a.bg[0], bgn = ls ? lg[j] : a.bg[1], bb = ls ? lb[j] : a.bg[2];
                o8[0] = fminf(fmaxf(none ? cr[j] : cr[j] + br * (1.f - accw[j]), 0.f), 1.f);
                o8[1] = fminf(fmaxf(none ? cg[j] : cg[j] + bgn * (1.f - accw[j]), 0.f), 1.f);
                o8[2] = fminf(fmaxf(none ? cb[j] : cb[j] + bb * (1.f - accw[j]), 0.f), 1.f);
                o8[3] = accw[j];
                o8[4] = depth[j];
                float ed = wt[j] / (accw[j] + 1e-10f);
                if (a.clip) ed = fminf(fmaxf(ed, clip_lo), clip_hi);
                o8[5] = ed;
                o8[6] = uvar[j];
                const double d64 = (double)depth[j];
                o8[7] = (float)(m2[j] - 2.0 * d64 * m1[j] + d64 * d64 * m0[j]) + 1e-5f;
                if (!MOMENTS) {
                    float4* o = reinterpret_cast<float4*>(a.out + ((int64_t)(b0 + j) * R + r) * 8);
                    o[0] = make_float4(o8[0], o8[1], o8[2], o8[3]);
                    o[1] = make_float4(o8[4], o8[5], o8[6], o8[7]);
                } else if (b0 + j == 0) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) x0[c] = o8[c];
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float d = o8[c] - x0[c];
                        sd[c] += d;
                        sd2[c] += d * d;
                    }
                }
            }
        }
    }
    if (a.flag && saw_nan) atomicOr(a.flag, 1);
    if (MOMENTS) {
        const float invB = 1.f / (float)a.B, invB1 = 1.f / (float)(a.B - 1);
        float m8[8], v8[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            m8[c] = x0[c] + sd[c] * invB;
            v8[c] = fmaxf(sd2[c] - sd[c] * sd[c] * invB, 0.f) * invB1;
        }
        float4* mo = reinterpret_cast<float4*>(a.mean_out + r * 8);
        float4* vo = reinterpret_cast<float4*>(a.var_out + r * 8);
        mo[0] = make_float4(m8[0], m8[1], m8[2], m8[3]);
        mo[1] = make_float4(m8[4], m8[5], m8[6], m8[7]);
        vo[0] = make_float4(v8[0], v8[1], v8[2], v8[3]);
        vo[1] = make_float4(v8[4], v8[5], v8[6], v8[7]);
    }
}

static int composite_planes_launch(const float* density, const float* rgb, const float* beta, const float* sbins, int B,
                                   int64_t R, int S, float near_plane, float far_plane, int spacing, const float* clip_minmax,
                                   int64_t ray_offset, int64_t chunk_rays, int background, const float* background_rgb,
                                   int32_t* nonfinite_flag, float* out, float* mean_out, float* var_out, void* stream,
                                   const char* what) {
    UNERF_REQUIRE(R == 0 || (density && rgb && sbins), "%s: null pointer", what);
    UNERF_REQUIRE(B >= 1 && R >= 0 && S >= 1, "%s: bad B/R/S", what);
    UNERF_REQUIRE(!clip_minmax || chunk_rays > 0, "%s: chunk_rays must be > 0 with clip_minmax", what);
    if (R == 0) return UNERF_OK;
    CompSmArgs a;
    a.density = density; a.rgb = rgb; a.beta = beta; a.sbins = sbins; a.B = B; a.R = R; a.S = S;
    UNERF_REQUIRE_SPACING(spacing); a.lin = spacing; a.s_near = unerf_spacing_of(near_plane, spacing); a.s_far = unerf_spacing_of(far_plane, spacing);
    a.clip = clip_minmax; a.ray_offset = ray_offset; a.chunk_rays = chunk_rays;
    a.out = out; a.mean_out = mean_out; a.var_out = var_out;
    if (int rc = unerf_set_background(a, background, background_rgb, what)) return rc;
    a.flag = nonfinite_flag;
    dim3 grid(blocks_for(R, 256)), block(256);
    if (out) hipLaunchKernelGGL(composite_sm_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(composite_sm_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
    return unerf_check_launch(what);
}

extern "C" int unerf_composite_var_planes(const float* density, const float* rgb, const float* beta, const float* sbins,
                                          int B, int64_t R, int S, float near_plane, float far_plane, int spacing,
                                          const float* clip_minmax, int64_t ray_offset, int64_t chunk_rays, int background,
                                          const float* background_rgb, int32_t* nonfinite_flag, float* out, void* stream) {
    UNERF_REQUIRE(R == 0 || out, "composite_var_planes: null pointer");
    return composite_planes_launch(density, rgb, beta, sbins, B, R, S, near_plane, far_plane, spacing, clip_minmax, ray_offset,
                                   chunk_rays, background, background_rgb, nonfinite_flag, out, nullptr, nullptr, stream,
                                   "composite_var_planes");
}

extern "C" int unerf_composite_moments_planes(const float* density, const float* rgb, const float* sbins, int B,
                                              int64_t R, int S, float near_plane, float far_plane, int spacing,
                                              const float* clip_minmax, int64_t ray_offset, int64_t chunk_rays,
                                              int background, const float* background_rgb, int32_t* nonfinite_flag,
                                              float* mean_out, float* var_out, void* stream) {
    UNERF_REQUIRE(R == 0 || (mean_out && var_out), "composite_moments_planes: null pointer");
    UNERF_REQUIRE(B >= 2, "composite_moments_planes: B=%d (the unbiased variance needs at least two passes)", B);
    return composite_planes_launch(density, rgb, nullptr, sbins, B, R, S, near_plane, far_plane, spacing, clip_minmax, ray_offset,
                                   chunk_rays, background, background_rgb, nonfinite_flag, nullptr, mean_out, var_out, stream,
                                   "composite_moments_planes");
}

// ---- laplace depth draws --------------------------------------------------------------
struct LapDepthArgs {
    const float* mu;
    const float* var;
    const float* sbins;
    int64_t R;
    int S;
    float s_near, s_far;
    int lin;   // UNERF_SPACING_*: 1 = identity spacing (UniformSampler)
    const float* noise;
    int D;
    uint32_t seed;
    int64_t ray_offset;
    float* out;
};

// Built-in generator of the depth draws (noise == NULL; twin: oracle normal_noise).  The round-2 form spent two full
// 32-bit hashes (four quarter-rate integer multiplies) per draw PAIR and sample -- more issue slots than the Box-Muller
// transform they feed.  Now every (ray, sample) owns ONE xorshift32 stream, seeded by the counter hash of its global sample
// index (so launch grouping still cannot change a draw) and stepped once per draw pair (six full-rate shift / xor
// instructions); the two 16-bit halves of the state are the two uniforms of a Box-Muller pair, BOTH of whose outputs are
// used (cos for the even draw, sin for the odd one).  v_log_f32 is log2, v_sin / v_cos take revolutions: no argument
// scaling.  16-bit uniforms bound |z| by sqrt(34 ln 2) = 4.85 and quantise the radius to 65536 levels -- far below the
// Monte-Carlo error of a 100-draw mean (tests: moments, independence between samples, draws and halves).
__device__ __forceinline__ uint32_t unerf_xorshift32(uint32_t x) {
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}
__device__ __forceinline__ uint32_t unerf_depth_stream_seed(uint32_t seed, uint32_t sample_idx) {
    const uint32_t x = unerf_mc_base(unerf_mc_key(seed, 0u), sample_idx);
    return x ? x : UNERF_GOLDEN;   // 0 is xorshift's fixed point
}
__device__ __forceinline__ void unerf_normal_pair_from_state(uint32_t x, float& z0, float& z1) {
    const float u1 = fmaf((float)(x >> 16), 1.0f / 65536.0f, 0.5f / 65536.0f);
    const float u2 = fmaf((float)(x & 0xFFFFu), 1.0f / 65536.0f, 0.5f / 65536.0f);
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
    z0 = rad * __builtin_amdgcn_cosf(u2);
    z1 = rad * __builtin_amdgcn_sinf(u2);
}

// get_weights for the Monte-Carlo depth draws.  With e_i = exp(-delta_i sigma_i) the weights are
//   w_i = (1 - e_i) prod_{j<i} e_j = P_i - P_{i+1},   P_i = prod_{j<i} e_j,
// the transmittance as a running product (a 16-lane multiplicative scan of the lanes' own products) instead of a second
// exp of the running sum.  Per draw and sample: the exponent min(a_i z + b_i, 0) with a_i = -delta_i log2(e) sd_i and
// b_i = -delta_i log2(e) mu_i formed once per sample (= -delta log2(e) relu(mu + sd z): -delta <= 0 turns the relu into a
// min), one v_exp_f32, one multiply for the running product, one subtract, and one fma that adds carry * (P_i - P_{i+1})
// to the sample's sum over the draws -- 7 issue slots + 5 DPP multiplies per lane and draw (round 2: 12 + 5).
// NaN (a NaN mean or variance reaches every draw alike) poisons the sums it would zero in get_weights' nan_to_num: this
// sample, the later ones of the lane and, through the carry, of the ray; the caller maps the final NaN to 0.
template <int SPL>
__device__ __forceinline__ void group_weights_accumulate(const float (&z)[SPL], const float (&a)[SPL], const float (&b)[SPL],
                                                         float (&wsum)[SPL]) {
    float dl[SPL], lp = 1.f;
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        const float em = __builtin_amdgcn_exp2f(fminf(fmaf(a[e], z[e], b[e]), 0.f));
        const float nx = lp * em;
        dl[e] = lp - nx;     // this lane's share of w_e: (product before) - (product after)
        lp = nx;
    }
    // exclusive multiplicative scan over the 16 lanes of the ray
    // (DPP row shifts; lanes without a source lane keep the `old` operand = 1)
    float incl = lp;
    incl *= dpp_f_or<0x111>(incl, 1.f);
    incl *= dpp_f_or<0x112>(incl, 1.f);
    incl *= dpp_f_or<0x114>(incl, 1.f);
    incl *= dpp_f_or<0x118>(incl, 1.f);
    const float carry = dpp_f_or<0x111>(incl, 1.f);
#pragma unroll
    for (int e = 0; e < SPL; ++e) wsum[e] = fmaf(carry, dl[e], wsum[e]);
}

template <int SPL, bool RAGGED = false>
__global__ __launch_bounds__(256) void lap_depth_kernel(LapDepthArgs a) {
    const int l16 = threadIdx.x & 15;
    int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = r < a.R;
    if (!ok) r = a.R - 1;
    const int S = a.S, k0 = l16 * SPL;
    const float* sb = a.sbins + r * (S + 1);
    float eu[SPL + 1], ca[SPL], cb[SPL], wsum[SPL];
    uint32_t st[SPL];
#pragma unroll
    for (int e = 0; e <= SPL; ++e) eu[e] = unerf_s2e(sb[RAGGED ? min(k0 + e, S) : k0 + e], a.s_near, a.s_far, a.lin);
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
        const float nd2 = -(eu[e + 1] - eu[e]) * 1.4426950408889634f;
        const bool live = !RAGGED || k0 + e < S;   // masked slots: mu = sd = 0 -> exponent 0, e = 1, weight 0
        const float mu = live ? a.mu[r * S + k0 + e] : 0.f;
        const float s = live ? sqrtf(a.var[r * S + k0 + e]) : 0.f;
        const float sd = !live ? 0.f : (s != s) ? 1e-10f : fmaxf(s, 1e-10f);
        ca[e] = nd2 * sd;
        cb[e] = nd2 * mu;
        wsum[e] = 0.f;
        st[e] = a.noise ? 0u : unerf_depth_stream_seed(a.seed, (uint32_t)((a.ray_offset + r) * S + k0 + e));
    }
    for (int d0 = 0; d0 < a.D; d0 += 2) {
        float z[2][SPL];
        if (a.noise) {
#pragma unroll
            for (int e = 0; e < SPL; ++e) {
                const bool live = !RAGGED || k0 + e < S;
                z[0][e] = live ? a.noise[((int64_t)d0 * a.R + r) * S + k0 + e] : 0.f;
                z[1][e] = (live && d0 + 1 < a.D) ? a.noise[((int64_t)(d0 + 1) * a.R + r) * S + k0 + e] : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < SPL; ++e) {
                unerf_normal_pair_from_state(st[e], z[0][e], z[1][e]);
                st[e] = unerf_xorshift32(st[e]);
            }
        }
        group_weights_accumulate<SPL>(z[0], ca, cb, wsum);
        if (d0 + 1 < a.D) group_weights_accumulate<SPL>(z[1], ca, cb, wsum);
    }
    if (ok) {
#pragma unroll
        for (int e = 0; e < SPL; ++e) {
            const float w = wsum[e] / (float)a.D;
            if (!RAGGED || k0 + e < S) a.out[r * S + k0 + e] = (w != w) ? 0.f : w;
        }
    }
}

extern "C" int unerf_laplace_depth_weights(const float* density_mu, const float* density_var, const float* sbins,
                                           int64_t R, int S, float near_plane, float far_plane, int spacing, const float* noise,
                                           int D, uint32_t seed, int64_t ray_offset, float* weights_out, void* stream) {
    UNERF_REQUIRE(R == 0 || (density_mu && density_var && sbins && weights_out), "laplace_depth_weights: null pointer");
    UNERF_REQUIRE(D >= 1 && R >= 0 && S >= 1 && S <= 256, "laplace_depth_weights: bad D/R/S");
    if (R == 0) return UNERF_OK;
    LapDepthArgs a;
    a.mu = density_mu; a.var = density_var; a.sbins = sbins; a.R = R; a.S = S;
    UNERF_REQUIRE_SPACING(spacing); a.lin = spacing; a.s_near = unerf_spacing_of(near_plane, spacing); a.s_far = unerf_spacing_of(far_plane, spacing);
    a.noise = noise; a.D = D; a.seed = seed; a.ray_offset = ray_offset; a.out = weights_out;
    dim3 grid(blocks_for(R, 16)), block(256);
    hipStream_t st = (hipStream_t)stream;
    UNERF_DISPATCH_SPL(S, lap_depth_kernel, grid, block, 0, st, a);
    return unerf_check_launch("laplace_depth_weights");
}

// ======================================================================================
// 7. moments over the leading (pass / member) dimension
// ======================================================================================
__global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ x, int K, int64_t NC,
                                                      float* __restrict__ mean, float* __restrict__ var) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= NC) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += x[(int64_t)k * NC + i];
    float m = s / (float)K;
    mean[i] = m;
    if (var) {
        float q = 0.f;
        for (int k = 0; k < K; ++k) {
            float d = x[(int64_t)k * NC + i] - m;
            q += d * d;
        }
        var[i] = q / (float)(K - 1);  // K==1 -> NaN, as torch.var(unbiased) does
    }
}

extern "C" int unerf_moments(const float* x, int K, int64_t N, int C, float* mean, float* var, void* stream) {
    UNERF_REQUIRE(N == 0 || (x && mean), "moments: null pointer");
    UNERF_REQUIRE(K >= 1 && N >= 0 && C >= 1, "moments: bad K/N/C");
    if (N == 0) return UNERF_OK;
    hipLaunchKernelGGL(moments_kernel, dim3(blocks_for(N * C, 256)), dim3(256), 0, (hipStream_t)stream, x, K, N * C,
                       mean, var);
    return unerf_check_launch("moments");
}
