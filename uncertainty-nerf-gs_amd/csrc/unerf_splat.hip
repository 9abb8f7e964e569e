// Gaussian-splat half of libunerf (gsplat 0.1.11 semantics, see include/unerf.h):
// EWA projection, SH colours + per-splat beta, one bin-and-sort per frame, tile rasteriser with
// C interleaved channels, alpha normalisation and the per-splat depth-difference gather.
// Built with -ffp-contract=off: projection / tile boxes / sort keys are bit-exact against the
// numpy oracle (oracle/splat_oracle.py) which performs the same fp32 operations in the same order.
#include "unerf_common.hpp"
#include <type_traits>

#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <cstring>

static inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// ======================================================================================
// projection
// ======================================================================================
struct ProjArgs {
    const float* means;
    const float* scales;
    float glob_scale;
    const float* quats;
    float V[12];
    float fx, fy, cx, cy;
    int H, W, bw;
    float clip;
    int64_t N;
    float* xys;
    float* depths;
    int32_t* radii;
    float* conics;
    float* comp;
    int32_t* tiles;
    float* cov3d;
    // tight binning (unerf_splat_project_raw with opacity logits): the activated opacity leaves this kernel and
    // num_tiles_hit counts only the tiles the splat's alpha >= 1/255 ellipse can reach
    const float* opl;
    float* opac_out;
    int antialiased;
};

__device__ __forceinline__ void tile_bbox(float cx, float cy, float radius, int bw, int tbx, int tby, int& x0, int& y0,
                                          int& x1, int& y1) {
    float tcx = cx / (float)bw, tcy = cy / (float)bw, tr = radius / (float)bw;
    x0 = min(max(0, (int)(tcx - tr)), tbx);
    x1 = min(max(0, (int)(tcx + tr + 1.f)), tbx);
    y0 = min(max(0, (int)(tcy - tr)), tby);
    y1 = min(max(0, (int)(tcy + tr + 1.f)), tby);
}

// ---- tight tile lists ------------------------------------------------------------------------------------------------
// gsplat bins a splat into every tile of the square [xy -+ radius] with radius = ceil(3 sqrt(lambda_max)).  The blend
// loop then skips each (pixel, splat) pair with alpha = min(0.999, o exp(-sigma)) < 1/255, i.e. everything outside the
// ellipse sigma <= ln(255 o): for an anisotropic or faint splat most tiles of the square hold no such pixel (49 % of
// the 37 M pairs of the 1 M-splat bench frame), yet each costs a sort entry and a staging slot in the rasteriser.
// TightSplat describes that ellipse (padded far above fp32 rounding, as raster_quad_mask); tight_row gives, for one
// tile row, the range of tiles whose pixel centres it can reach -- always a subset of gsplat's box, so the lists
// lose only pairs the blend loop would have skipped: images, transmittances and orders of the blended terms are
// bit-identical (tests/test_gpu_splat.py::test_tight_tile_lists_*).
struct TightSplat {
    float x, y, b;          // centre, conic b
    float k, det, inv_a, hy, ry;   // k = 2 tau a; ry = -(b / c) hx: dy of the ellipse's rightmost point (leftmost: -ry)
    float pad;
    int valid;              // 0: keep gsplat's box (not an ellipse / non-finite numbers); -1: never visible
};
// (hardware sqrt / rcp, ~1 ulp: the pad is orders of magnitude above that, and both kernels that walk a splat's rows --
// the count in project_kernel, the emission in map_intersects_kernel -- run this same code on the same stored numbers)

__device__ __forceinline__ TightSplat tight_splat(float x, float y, float op, float a, float b, float c) {
    TightSplat t;
    t.x = x; t.y = y; t.b = b;
    t.valid = 0; t.k = 0.f; t.det = 0.f; t.inv_a = 0.f; t.hy = 0.f; t.ry = 0.f; t.pad = 0.f;
    if (!(op >= 0.0039f)) {            // alpha <= opacity < 1/255 (0.00392...) at every pixel
        t.valid = (op != op) ? 0 : -1;
        return t;
    }
    const float det = a * c - b * b;
    if (!(det > 0.f) || !(a > 0.f) || !(c > 0.f)) return t;
    const float two_tau = 2.f * fmaf(__logf(255.f * op), 1.01f, 0.01f);
    const float inv_det = __builtin_amdgcn_rcpf(det);
    const float hx = __builtin_amdgcn_sqrtf(two_tau * c * inv_det);
    t.k = two_tau * a;
    t.det = det;
    t.inv_a = __builtin_amdgcn_rcpf(a);
    t.hy = __builtin_amdgcn_sqrtf(t.k * inv_det);
    t.ry = -(b * __builtin_amdgcn_rcpf(c)) * hx;
    t.pad = fmaf(fmaxf(hx, t.hy), 0.01f, 0.05f);
    const bool fin = fabsf(hx) < INFINITY && fabsf(t.hy) < INFINITY && fabsf(t.ry) < INFINITY && fabsf(t.inv_a) < INFINITY;   // false for NaN too
    t.valid = fin ? 1 : 0;
    return t;
}

// tiles [tx0, tx1) of tile row ty (inside gsplat's [x0, x1)) holding a pixel centre with sigma <= tau
__device__ __forceinline__ void tight_row(const TightSplat& t, int ty, int bw, int x0, int x1, int& tx0, int& tx1) {
    tx0 = x0; tx1 = x1;
    if (t.valid == 0) return;
    if (t.valid < 0) { tx1 = x0; return; }
    const float fbw = (float)bw, inv_bw = 1.f / fbw;   // (bw is uniform: one scalar division; exact for 16)
    // pixel-centre rows of the tile row, relative to the splat, widened by the pad
    float d0 = ((float)ty * fbw + 0.5f) - t.y - t.pad, d1 = ((float)ty * fbw + (fbw - 0.5f)) - t.y + t.pad;
    if (d0 > t.hy || d1 < -t.hy) { tx1 = x0; return; }
    d0 = fmaxf(d0, -t.hy); d1 = fminf(d1, t.hy);
    // right end (-b dy + sqrt(2 tau a - det dy^2)) / a is concave in dy: its maximum over the rows is at the rightmost
    // point's dy clamped into them; the left end is the mirror image
    const float dr = fminf(fmaxf(t.ry, d0), d1), dl = fminf(fmaxf(-t.ry, d0), d1);
    const float sr = __builtin_amdgcn_sqrtf(fmaxf(t.k - t.det * dr * dr, 0.f));
    const float sl = __builtin_amdgcn_sqrtf(fmaxf(t.k - t.det * dl * dl, 0.f));
    const float xr = (-t.b * dr + sr) * t.inv_a + t.pad, xl = (-t.b * dl - sl) * t.inv_a - t.pad;
    if (!(xr >= xl)) return;   // NaN: keep the box
    // a pixel centre c = j + 0.5 lies in tile floor((c - 0.5) / bw); the pad covers the rounding of the product
    const float fl = floorf((t.x + xl - 0.5f) * inv_bw), fr = floorf((t.x + xr - 0.5f) * inv_bw);
    tx0 = max(x0, (int)fmaxf(fl, -1e6f));
    tx1 = min(x1, (int)fminf(fr, 1e6f) + 1);
    if (tx1 < tx0) tx1 = tx0;
}

__device__ __forceinline__ int tight_count(const TightSplat& t, int bw, int x0, int y0, int x1, int y1) {
    int n = 0;
    for (int ty = y0; ty < y1; ++ty) {
        int a0, a1;
        tight_row(t, ty, bw, x0, x1, a0, a1);
        n += a1 - a0;
    }
    return n;
}

// RAW: `scales` / `quats` are the model's parameters as stored -- log-scales and unnormalised quaternions -- and the
// reference's per-frame torch prologue (activesplatfacto_model.py:221-223: torch.exp(scales_crop),
// quats_crop / quats_crop.norm(dim=-1, keepdim=True)) happens here instead of in three elementwise launches.
template <bool RAW>
__global__ __launch_bounds__(256) void project_kernel(ProjArgs a) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N) return;
    // gsplat allocates every output zero-filled and a culled splat keeps the zeros of whatever it had not reached yet.  Every
    // output is held in a register (zero until computed) and stored ONCE at the end -- zero-filling the 15 dwords first and
    // overwriting them for visible splats wrote the arrays twice (WRITE_SIZE 119 MB for 60 MB of outputs).
    float o_xy[2] = {0.f, 0.f}, o_depth = 0.f, o_comp = 0.f, o_conic[3] = {0.f, 0.f, 0.f}, o_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int o_radius = 0, o_tiles = 0;
    float opac = 0.f, o_opac = 0.f;
    if (a.opl) {   // uniform.  sigmoid(opacities) (:256); "antialiased" multiplies the compensation in below (:252-254)
        opac = unerf_sigmoid(a.opl[i]);
        o_opac = a.antialiased ? 0.f : opac;
    }
    auto per_splat = [&]() {
    const float p0 = a.means[i * 3], p1 = a.means[i * 3 + 1], p2 = a.means[i * 3 + 2];
    const float* V = a.V;
    float tx = ((V[0] * p0 + V[1] * p1) + V[2] * p2) + V[3];
    float ty = ((V[4] * p0 + V[5] * p1) + V[6] * p2) + V[7];
    float tz = ((V[8] * p0 + V[9] * p1) + V[10] * p2) + V[11];
    if (tz <= a.clip) return;      // (leaves the lambda: the stores are below)

    // scale_rot_to_cov3d
    float qw = a.quats[i * 4], qx = a.quats[i * 4 + 1], qy = a.quats[i * 4 + 2], qz = a.quats[i * 4 + 3];
    if (RAW) {   // the model's normalisation (a true division, as torch's), then gsplat's own below
        const float qn = sqrtf(((qw * qw + qx * qx) + qy * qy) + qz * qz);
        qw = qw / qn; qx = qx / qn; qy = qy / qn; qz = qz / qn;
    }
    float qs = 1.f / sqrtf(((qw * qw + qx * qx) + qy * qy) + qz * qz);
    float w = qw * qs, x = qx * qs, y = qy * qs, z = qz * qs;
    float Rm[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z),       2.f * (x * z + w * y),
                   2.f * (x * y + w * z),       1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x),
                   2.f * (x * z - w * y),       2.f * (y * z + w * x),       1.f - 2.f * (x * x + y * y)};
    float sc0 = a.scales[i * 3], sc1 = a.scales[i * 3 + 1], sc2 = a.scales[i * 3 + 2];
    if (RAW) { sc0 = expf(sc0); sc1 = expf(sc1); sc2 = expf(sc2); }
    float s0 = a.glob_scale * sc0, s1 = a.glob_scale * sc1, s2 = a.glob_scale * sc2;
    float M[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        M[r * 3 + 0] = Rm[r * 3 + 0] * s0;
        M[r * 3 + 1] = Rm[r * 3 + 1] * s1;
        M[r * 3 + 2] = Rm[r * 3 + 2] * s2;
    }
    float Sg[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            Sg[r * 3 + c] = (M[r * 3] * M[c * 3] + M[r * 3 + 1] * M[c * 3 + 1]) + M[r * 3 + 2] * M[c * 3 + 2];
    o_cov[0] = Sg[0]; o_cov[1] = Sg[1]; o_cov[2] = Sg[2]; o_cov[3] = Sg[4]; o_cov[4] = Sg[5]; o_cov[5] = Sg[8];
    // symmetric V from the 6 stored entries (as gsplat rebuilds it)
    float C3[9] = {Sg[0], Sg[1], Sg[2], Sg[1], Sg[4], Sg[5], Sg[2], Sg[5], Sg[8]};

    // project_cov3d_ewa
    float tan_fovx = 0.5f * (float)a.W / a.fx, tan_fovy = 0.5f * (float)a.H / a.fy;
    float lim_x = 1.3f * tan_fovx, lim_y = 1.3f * tan_fovy;
    float ex = tz * fminf(lim_x, fmaxf(-lim_x, tx / tz));
    float ey = tz * fminf(lim_y, fmaxf(-lim_y, ty / tz));
    float rz = 1.f / tz, rz2 = rz * rz;
    float J00 = a.fx * rz, J02 = (-a.fx * ex) * rz2, J11 = a.fy * rz, J12 = (-a.fy * ey) * rz2;
    float T0[3], T1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        T0[c] = J00 * V[0 * 4 + c] + J02 * V[2 * 4 + c];
        T1[c] = J11 * V[1 * 4 + c] + J12 * V[2 * 4 + c];
    }
    float TV0[3], TV1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        TV0[c] = (T0[0] * C3[0 * 3 + c] + T0[1] * C3[1 * 3 + c]) + T0[2] * C3[2 * 3 + c];
        TV1[c] = (T1[0] * C3[0 * 3 + c] + T1[1] * C3[1 * 3 + c]) + T1[2] * C3[2 * 3 + c];
    }
    float c00 = (TV0[0] * T0[0] + TV0[1] * T0[1]) + TV0[2] * T0[2];
    float c01 = (TV0[0] * T1[0] + TV0[1] * T1[1]) + TV0[2] * T1[2];
    float c11 = (TV1[0] * T1[0] + TV1[1] * T1[1]) + TV1[2] * T1[2];
    float det_orig = c00 * c11 - c01 * c01;
    float ca = c00 + 0.3f, cb = c01, cc = c11 + 0.3f;
    float det = ca * cc - cb * cb;
    float comp = sqrtf(fmaxf(0.f, det_orig / det));
    // compute_cov2d_bounds
    if (det == 0.f) return;
    float inv_det = 1.f / det;
    o_conic[0] = cc * inv_det;
    o_conic[1] = -cb * inv_det;
    o_conic[2] = ca * inv_det;
    float bh = 0.5f * (ca + cc);
    float sq = sqrtf(fmaxf(0.1f, bh * bh - det));
    float v1 = bh + sq, v2 = bh - sq;
    float radius = ceilf(3.f * sqrtf(fmaxf(v1, v2)));
    // project_pix
    float rw = 1.f / (tz + 1e-6f);
    float u = (tx * rw) * a.fx + a.cx, v = (ty * rw) * a.fy + a.cy;
    int tbx = (a.W + a.bw - 1) / a.bw, tby = (a.H + a.bw - 1) / a.bw;
    int x0, y0, x1, y1;
    tile_bbox(u, v, radius, a.bw, tbx, tby, x0, y0, x1, y1);
    int area = (x1 - x0) * (y1 - y0);
    if (area <= 0) return;
    if (a.opl) {
        if (a.antialiased) {
            opac = opac * comp;
            o_opac = opac;
        }
        area = tight_count(tight_splat(u, v, opac, cc * inv_det, -cb * inv_det, ca * inv_det), a.bw, x0, y0, x1, y1);
    }
    o_tiles = area;
    o_depth = tz;
    o_radius = (int)radius;
    o_xy[0] = u;
    o_xy[1] = v;
    o_comp = comp;
    };
    per_splat();
    a.radii[i] = o_radius;
    a.tiles[i] = o_tiles;
    a.xys[i * 2] = o_xy[0];
    a.xys[i * 2 + 1] = o_xy[1];
    a.depths[i] = o_depth;
    a.comp[i] = o_comp;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.conics[i * 3 + c] = o_conic[c];
#pragma unroll
    for (int c = 0; c < 6; ++c) a.cov3d[i * 6 + c] = o_cov[c];
    if (a.opl) a.opac_out[i] = o_opac;
}

static int splat_project_impl(bool raw, const float* opacity_logits, int antialiased, float* opacities_out,
                              const float* means3d, const float* scales, float glob_scale, const float* quats,
                              const float* viewmat, float fx, float fy, float cx, float cy, int H, int W,
                              int block_width, float clip_thresh, int64_t N, float* xys, float* depths,
                              int32_t* radii, float* conics, float* compensation, int32_t* num_tiles_hit,
                              float* cov3d, void* stream) {
    UNERF_REQUIRE(viewmat && (N == 0 || (means3d && scales && quats && xys && depths && radii && conics && compensation &&
                                       num_tiles_hit && cov3d)),
                  "splat_project: null pointer");
    UNERF_REQUIRE(H > 0 && W > 0 && block_width > 0 && block_width <= 16 && N >= 0, "splat_project: bad H/W/block/N");
    if (N == 0) return UNERF_OK;
    ProjArgs a;
    a.means = means3d; a.scales = scales; a.glob_scale = glob_scale; a.quats = quats;
    for (int k = 0; k < 12; ++k) a.V[k] = viewmat[k];
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.H = H; a.W = W; a.bw = block_width; a.clip = clip_thresh; a.N = N;
    a.xys = xys; a.depths = depths; a.radii = radii; a.conics = conics; a.comp = compensation;
    a.tiles = num_tiles_hit; a.cov3d = cov3d;
    a.opl = opacity_logits; a.opac_out = opacities_out; a.antialiased = antialiased;
    if (raw) hipLaunchKernelGGL(project_kernel<true>, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(project_kernel<false>, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return unerf_check_launch("splat_project");
}

extern "C" int unerf_splat_project(const float* means3d, const float* scales, float glob_scale, const float* quats,
                                   const float* viewmat, float fx, float fy, float cx, float cy, int H, int W,
                                   int block_width, float clip_thresh, int64_t N, float* xys, float* depths,
                                   int32_t* radii, float* conics, float* compensation, int32_t* num_tiles_hit,
                                   float* cov3d, void* stream) {
    return splat_project_impl(false, nullptr, 0, nullptr, means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, H, W, block_width,
                              clip_thresh, N, xys, depths, radii, conics, compensation, num_tiles_hit, cov3d, stream);
}

extern "C" int unerf_splat_project_raw(const float* means3d, const float* log_scales, float glob_scale,
                                       const float* raw_quats, const float* viewmat, float fx, float fy, float cx,
                                       float cy, int H, int W, int block_width, float clip_thresh, int64_t N,
                                       const float* opacity_logits, int antialiased, float* opacities_out, float* xys,
                                       float* depths, int32_t* radii, float* conics, float* compensation,
                                       int32_t* num_tiles_hit, float* cov3d, void* stream) {
    UNERF_REQUIRE(!opacity_logits || opacities_out || N <= 0, "splat_project_raw: opacity_logits without opacities_out");
    return splat_project_impl(true, opacity_logits, antialiased, opacities_out, means3d, log_scales, glob_scale, raw_quats, viewmat, fx, fy, cx, cy, H, W, block_width,
                              clip_thresh, N, xys, depths, radii, conics, compensation, num_tiles_hit, cov3d, stream);
}

// ======================================================================================
// SH colours (+0.5, clamp>=0) and beta = softplus(log_unc) + beta_min
// ======================================================================================
// SPLIT: the coefficients arrive as the model stores them -- features_dc [N,3] and features_rest [N,15,3]
// (`coeffs` = dc, `rest` = the 45-float rows) -- which saves the per-frame torch.cat of 192 B per splat that the
// reference performs (activesplatfacto_model.py:242-243) and this kernel would only read back once.
// PACK (unerf_splat_shade_inputs): everything the rasteriser reads per splat leaves this kernel in its final layout --
// one interleaved row [rgb, (beta), depth] per splat and the activated opacity -- instead of through the reference's
// torch.cat / torch.sigmoid launches (activesplatfacto_model.py:256, 286-305).
struct ShadePack {
    const float* opacity_logits;   // [N]
    const float* compensation;     // [N] or NULL (rasterize_mode == "antialiased": opacities * comp, :252-254)
    const float* depths;           // [N]
    float* opacities;              // [N] out
    int C;                         // row length: 4 = rgb + depth, 5 = rgb + beta + depth
};

// STAGE (SPLIT rows, degree 3, 16-byte aligned features_rest): the workgroup's 256 rows of 180 B are one contiguous 46-KB
// block -- it is read with coalesced 16-byte loads into LDS and every thread takes its 45 floats from there (stride 45
// words: conflict-free), instead of 12 loads per thread that touch 64 cache lines each (87 us for 208 MB: 2.4 TB/s).
template <bool SPLIT, bool PACK, bool STAGE = false>
__global__ __launch_bounds__(256) void sh_colors_kernel(int degree, const float* __restrict__ means, float cxp,
                                                        float cyp, float czp, const float* __restrict__ coeffs,
                                                        const float* __restrict__ rest,
                                                        const float* __restrict__ log_unc, float beta_min, int64_t N,
                                                        float* __restrict__ colors, float* __restrict__ beta,
                                                        ShadePack pk) {
    __shared__ float s_rest[STAGE ? 256 * 45 : 1];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (STAGE) {
        const int64_t b0 = (int64_t)blockIdx.x * 256;
        const int nb = (int)((N - b0 < 256) ? N - b0 : 256);
        const float4* src = reinterpret_cast<const float4*>(rest + b0 * 45);     // (b0 * 180 B is a multiple of 16)
        const int nq4 = nb * 45 / 4;
#pragma unroll 4
        for (int q = threadIdx.x; q < nq4; q += 256) reinterpret_cast<float4*>(s_rest)[q] = src[q];
        for (int r = nq4 * 4 + (int)threadIdx.x; r < nb * 45; r += 256) s_rest[r] = rest[b0 * 45 + r];
        __syncthreads();
    }
    if (i >= N) return;
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                         0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    // the splat's coefficient row (192 B, 16-byte aligned) as 16-byte loads: a thread-per-splat kernel reads with a
    // 192-B stride, so every load instruction touches 64 cache lines -- 12 of them instead of 46 dword loads
    float k[48];
    if (SPLIT && STAGE) {
        const float* dc = coeffs + i * 3;
        k[0] = dc[0]; k[1] = dc[1]; k[2] = dc[2];
#pragma unroll
        for (int j = 0; j < 45; ++j) k[3 + j] = s_rest[threadIdx.x * 45 + j];
    } else if (SPLIT) {
        const float* dc = coeffs + i * 3;
        k[0] = dc[0]; k[1] = dc[1]; k[2] = dc[2];
        // 180-byte rows are only 4-byte aligned: 11 x 16-byte + 1 x 4-byte unaligned-capable global loads
        struct __attribute__((packed, aligned(4))) Q { float x, y, z, w; };
        const Q* r4 = reinterpret_cast<const Q*>(rest + i * 45);
        const int nq = degree <= 0 ? 0 : degree == 1 ? 3 : degree == 2 ? 6 : 11;       // 9 / 24 / 45 floats used
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            Q v = {0.f, 0.f, 0.f, 0.f};
            if (q < nq) v = r4[q];
            k[3 + 4 * q] = v.x; k[4 + 4 * q] = v.y; k[5 + 4 * q] = v.z; k[6 + 4 * q] = v.w;
        }
        k[47] = (degree >= 3) ? rest[i * 45 + 44] : 0.f;
    } else {
        const float4* k4 = reinterpret_cast<const float4*>(coeffs + i * 48);
        const int nq = degree <= 0 ? 1 : degree == 1 ? 3 : degree == 2 ? 7 : 12;   // 3 / 12 / 27 / 48 floats used
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const float4 v = (q < nq) ? k4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
        }
    }
    float col[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) col[c] = C0 * k[c];
    if (degree >= 1) {
        float vx = means[i * 3] - cxp, vy = means[i * 3 + 1] - cyp, vz = means[i * 3 + 2] - czp;
        float nrm = sqrtf((vx * vx + vy * vy) + vz * vz);
        float x = vx / nrm, y = vy / nrm, z = vz / nrm;
        float xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z, zz = z * z;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            col[c] += C1 * (-y * k[1 * 3 + c] + z * k[2 * 3 + c] - x * k[3 * 3 + c]);
            if (degree >= 2) {
                col[c] += (C2[0] * xy * k[4 * 3 + c] + C2[1] * yz * k[5 * 3 + c] +
                           C2[2] * (2.f * zz - xx - yy) * k[6 * 3 + c] + C2[3] * xz * k[7 * 3 + c] +
                           C2[4] * (xx - yy) * k[8 * 3 + c]);
            }
            if (degree >= 3) {
                col[c] += (C3[0] * y * (3.f * xx - yy) * k[9 * 3 + c] + C3[1] * xy * z * k[10 * 3 + c] +
                           C3[2] * y * (4.f * zz - xx - yy) * k[11 * 3 + c] +
                           C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * k[12 * 3 + c] +
                           C3[4] * x * (4.f * zz - xx - yy) * k[13 * 3 + c] + C3[5] * z * (xx - yy) * k[14 * 3 + c] +
                           C3[6] * x * (xx - 3.f * yy) * k[15 * 3 + c]);
            }
        }
    }
    const int64_t row = PACK ? i * pk.C : i * 3;
    if (degree < 0) {  // config.sh_degree == 0: rgbs = sigmoid(features_dc) (activesplatfacto_model.py:247-248)
#pragma unroll
        for (int c = 0; c < 3; ++c) colors[row + c] = unerf_sigmoid(k[c]);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) colors[row + c] = fmaxf(col[c] + 0.5f, 0.f);
    }
    if (PACK) {
        if (pk.C == 5) colors[row + 3] = unerf_softplus(log_unc[i]) + beta_min;
        colors[row + pk.C - 1] = pk.depths[i];
        if (pk.opacity_logits) {   // uniform
            const float o = unerf_sigmoid(pk.opacity_logits[i]);
            pk.opacities[i] = pk.compensation ? o * pk.compensation[i] : o;
        }
    } else if (beta) {
        beta[i] = unerf_softplus(log_unc[i]) + beta_min;
    }
}

extern "C" int unerf_splat_sh_colors(int degree, const float* means3d, const float* cam_pos, const float* sh_coeffs,
                                     const float* log_unc, float beta_min, int64_t N, float* colors_out,
                                     float* beta_out, void* stream) {
    UNERF_REQUIRE(cam_pos && (N <= 0 || (means3d && sh_coeffs && colors_out)), "splat_sh_colors: null pointer");
    UNERF_REQUIRE(degree >= -1 && degree <= 3, "splat_sh_colors: degree %d outside [-1,3]", degree);
    UNERF_REQUIRE(!beta_out || log_unc || N <= 0, "splat_sh_colors: beta_out without log_unc");
    UNERF_REQUIRE(((uintptr_t)sh_coeffs & 15u) == 0, "splat_sh_colors: sh_coeffs must be 16-byte aligned");
    if (N <= 0) return UNERF_OK;
    hipLaunchKernelGGL((sh_colors_kernel<false, false>), dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       degree, means3d, cam_pos[0], cam_pos[1], cam_pos[2], sh_coeffs, nullptr, log_unc, beta_min, N,
                       colors_out, beta_out, ShadePack{});
    return unerf_check_launch("splat_sh_colors");
}

extern "C" int unerf_splat_sh_colors_split(int degree, const float* means3d, const float* cam_pos, const float* features_dc,
                                           const float* features_rest, const float* log_unc, float beta_min, int64_t N,
                                           float* colors_out, float* beta_out, void* stream) {
    UNERF_REQUIRE(cam_pos && (N <= 0 || (means3d && features_dc && colors_out)), "splat_sh_colors_split: null pointer");
    UNERF_REQUIRE(degree >= -1 && degree <= 3, "splat_sh_colors_split: degree %d outside [-1,3]", degree);
    UNERF_REQUIRE(degree <= 0 || N <= 0 || features_rest, "splat_sh_colors_split: degree %d needs features_rest", degree);
    UNERF_REQUIRE(!beta_out || log_unc || N <= 0, "splat_sh_colors_split: beta_out without log_unc");
    if (N <= 0) return UNERF_OK;
    hipLaunchKernelGGL((sh_colors_kernel<true, false>), dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       degree, means3d, cam_pos[0], cam_pos[1], cam_pos[2], features_dc, features_rest, log_unc, beta_min,
                       N, colors_out, beta_out, ShadePack{});
    return unerf_check_launch("splat_sh_colors_split");
}

extern "C" int unerf_splat_shade_inputs(int degree, const float* means3d, const float* cam_pos, const float* features_dc,
                                        const float* features_rest, const float* log_unc, float beta_min,
                                        const float* opacity_logits, const float* compensation, const float* depths,
                                        int64_t N, int C, float* rows_out, float* opacities_out, void* stream) {
    UNERF_REQUIRE(cam_pos && (N <= 0 || (means3d && features_dc && depths && rows_out)), "splat_shade_inputs: null pointer");
    UNERF_REQUIRE(!opacity_logits || opacities_out || N <= 0, "splat_shade_inputs: opacity_logits without opacities_out");
    UNERF_REQUIRE(degree >= -1 && degree <= 3, "splat_shade_inputs: degree %d outside [-1,3]", degree);
    UNERF_REQUIRE(degree <= 0 || N <= 0 || features_rest, "splat_shade_inputs: degree %d needs features_rest", degree);
    UNERF_REQUIRE(C == 4 || C == 5, "splat_shade_inputs: C=%d, rows are [rgb, depth] (4) or [rgb, beta, depth] (5)", C);
    UNERF_REQUIRE(C == 4 || log_unc || N <= 0, "splat_shade_inputs: C=5 needs log_unc");
    if (N <= 0) return UNERF_OK;
    ShadePack pk;
    pk.opacity_logits = opacity_logits; pk.compensation = compensation; pk.depths = depths; pk.opacities = opacities_out;
    pk.C = C;
    if (degree >= 3 && ((uintptr_t)features_rest & 15u) == 0) {      // rows staged through LDS (see sh_colors_kernel)
        hipLaunchKernelGGL((sh_colors_kernel<true, true, true>), dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, degree,
                           means3d, cam_pos[0], cam_pos[1], cam_pos[2], features_dc, features_rest, log_unc, beta_min, N,
                           rows_out, nullptr, pk);
    } else {
        hipLaunchKernelGGL((sh_colors_kernel<true, true>), dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, degree,
                           means3d, cam_pos[0], cam_pos[1], cam_pos[2], features_dc, features_rest, log_unc, beta_min, N,
                           rows_out, nullptr, pk);
    }
    return unerf_check_launch("splat_shade_inputs");
}

// ======================================================================================
// bin and sort (once per frame)
// ======================================================================================
static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
#define SCAN_BLOCK 1024    // own_inclusive_scan: elements per workgroup
#define SCAN_DIRECT_BLOCKS 4096   // up to this many blocks (4 M elements) every block re-sums the earlier blocks' sums itself
#define TS_SEG 32          // one-pass tile sort (below): the prefix over chunks runs in 32 independent row segments
#define TS_MAX_T1 12000    // tiles + 1 sentinel: the whole-key counters + 16 waves' digit counters must fit 64 KB of LDS (beyond: rocprim)
struct TileSortPlan {
    int chunk, nblk, rows, T1;
};
static TileSortPlan tile_sort_plan(int64_t I, int tiles) {
    TileSortPlan p;
    p.T1 = tiles + 1;
    // one single-wave workgroup per chunk, four of them per CU (their 32-KB tables): ~1024 chunks fill the chip
    int64_t chunk = (I + 1023) / 1024;
    chunk = ((chunk + 255) / 256) * 256;
    if (chunk < 2048) chunk = 2048;
    p.chunk = (int)chunk;
    p.nblk = (int)((I + chunk - 1) / chunk);
    if (p.nblk < 1) p.nblk = 1;
    p.rows = ((p.nblk + TS_SEG - 1) / TS_SEG) * TS_SEG;
    return p;
}

// ---- staged LSD radix passes (rs_* kernels below): the plan ---------------------------------------------------------------
// One pass = per-chunk digit histogram -> prefix over chunks (per digit) -> scatter; a chunk is the M pairs ONE wave ranks,
// stages in LDS in digit order and writes out as runs of consecutive slots (see rs_scatter_kernel for why).
#define RS_M_TILE 2048      // pairs per wave, tile sort (u16 / u32 keys, <= 128 digits: runs of >= 16 slots per digit)
#define RS_M_DEPTH 1024     // pairs per wave, depth sort (u32 keys, 256 digits: runs of 4)
#define RS_CPW 1            // chunks per wave in the histogram kernels; the whole-key histogram's workgroup = 16 waves = 16 chunks
struct RsTilePlan {
    int nchunk, b0, B0, B1, nhw;      // digit 0 = key & (B0 - 1), digit 1 = key >> b0 (< B1; B0, B1 powers of two <= 128)
};
static RsTilePlan rs_tile_plan(int64_t I, int T1) {
    RsTilePlan p;
    int bits = 1;
    while ((1 << bits) < T1) ++bits;         // keys 0 .. T1 - 1 (the sentinel tile is T1 - 1)
    p.b0 = bits <= 7 ? 0 : bits / 2;         // <= 128 keys: one pass on the whole key
    p.B0 = 1 << p.b0;
    p.B1 = 1 << (bits - p.b0);
    p.nchunk = (int)((I + RS_M_TILE - 1) / RS_M_TILE);
    if (p.nchunk < 1) p.nchunk = 1;
    p.nhw = (p.nchunk + 15) / 16;
    return p;
}

static int tile_bits(int H, int W, int bw) {   // bits of the largest key: tiles - 1, and `tiles` itself (the sentinel)
    int64_t tiles = (int64_t)((W + bw - 1) / bw) * ((H + bw - 1) / bw);
    int b = 1;
    while (((int64_t)1 << b) <= tiles) ++b;
    return b;
}

// Workspace of unerf_splat_count_intersects / unerf_splat_bin_sort.  bin_sort orders the N splats by depth
// first (32-bit keys, small), emits the intersections in that order and then needs only a STABLE sort by
// tile id (13 bits at 1080p, 16-bit keys): two 6-byte-per-entry radix passes over the I intersections
// instead of six 12-byte passes over 64-bit (tile | depth) keys.  Same final order, bit for bit: ties in
// depth keep the splat-index order in both schemes.
// (rocprim's default dispatch for the depth sort -- kept behind UNERF_SPLAT_DEPTH_SORT=rocprim -- is a merge sort up to 2^20
// items, where a 1 M-splat scene sits; forcing its Onesweep there was 137 us against 162 for the sort alone
// (benchmarks/exp_depth_sort.hip), net zero inside the frame.  The default is the four staged LSD passes of rs_scatter_kernel;
// making the keys inside the first pass instead of in depth_keys_kernel was measured too: +8 us, its histogram loses the
// 16-byte key loads.)
static hipError_t depth_sort_pairs(void* tmp, size_t& tmp_bytes, const uint32_t* kin, uint32_t* kout, const int32_t* vin,
                                   int32_t* vout, int64_t n, hipStream_t st) {
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, vin, vout, (int)n, 0, 32, st);
}

struct SortWs {
    int64_t tmp, dkey_in, dkey_out, id_in, order, counts, cum, tkey_in, tkey_out, val_in, val_mid, ts_table, ts_segsum, ts_start, ds_table, ds_total, total;
};
static SortWs sort_ws_layout(int64_t N, int64_t I) {
    size_t scan_tmp = 0, sortN_tmp = 0, sortI_tmp = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_tmp, (const int32_t*)nullptr, (int32_t*)nullptr, (int)N);
    (void)depth_sort_pairs(nullptr, sortN_tmp, nullptr, nullptr, nullptr, nullptr, N, 0);
    if (I > 0)
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sortI_tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                                 (const int32_t*)nullptr, (int32_t*)nullptr, (int)I, 0, 32);
    (void)hipGetLastError();
    size_t tmp = scan_tmp > sortN_tmp ? scan_tmp : sortN_tmp;
    tmp = tmp > sortI_tmp ? tmp : sortI_tmp;
    const size_t own_scan = (size_t)((N + SCAN_BLOCK - 1) / SCAN_BLOCK + 1) * 4;      // own_inclusive_scan's block sums
    tmp = tmp > own_scan ? tmp : own_scan;
    SortWs w;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off += align256(bytes); return o; };
    w.tmp = take((int64_t)tmp);
    w.dkey_in = take(N * 4); w.dkey_out = take(N * 4); w.id_in = take(N * 4); w.order = take(N * 4);
    w.counts = take(N * 4); w.cum = take(N * 4);
    w.tkey_in = take(I * 4); w.tkey_out = take(I * 4); w.val_in = take(I * 4); w.val_mid = take(I * 4);
    // the one-pass tile sort's tables, sized for the largest tile count it serves (the image size is not known here)
    const TileSortPlan tp = tile_sort_plan(I, TS_MAX_T1 - 1);
    const RsTilePlan lp = rs_tile_plan(I, TS_MAX_T1);     // the two-pass sort's tables: 2 x [128][nchunk] + [nhw][T1]
    const int64_t onepass_bytes = (int64_t)tp.rows * TS_MAX_T1 * 4, lsd_bytes = ((int64_t)256 * lp.nchunk + (int64_t)lp.nhw * TS_MAX_T1) * 4;
    w.ts_table = take(onepass_bytes > lsd_bytes ? onepass_bytes : lsd_bytes);
    w.ts_segsum = take((int64_t)TS_SEG * TS_MAX_T1 * 4);
    w.ts_start = take((int64_t)(2 * TS_MAX_T1 + 2) * 4);     // start [T1 + 1] + total [T1]
    w.ds_table = take((int64_t)256 * ((N + RS_M_DEPTH - 1) / RS_M_DEPTH) * 4);     // the depth sort's [256][chunks] counters
    w.ds_total = take(256 * 4);
    w.total = off + 1024;
    return w;
}

extern "C" int64_t unerf_splat_sort_workspace_bytes(int64_t N, int64_t I) {
    if (N < 1) N = 1;
    if (I < 0) I = 0;
    return sort_ws_layout(N, I).total;
}

// ---- inclusive scan of N int32 in two launches -------------------------------------------------------------------------------
// (rocprim's look-back scan is three launches of 4 - 8 us for the 4 MB the frame scans twice: the tile counts in splat order and
// again in depth order.)  Blocks of 1,024 elements: (1) block sums; (2) every block adds up the sums of the blocks before it
// (at most N / 1024 values) and scans its own elements behind that.  Integer sums: the same numbers whatever the order.
__global__ __launch_bounds__(256) void scan_sums_kernel(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ bsum) {
    __shared__ int32_t s_w[4];
    const int64_t e0 = (int64_t)blockIdx.x * SCAN_BLOCK + 4 * threadIdx.x;
    int32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) v += (e0 + i < n) ? in[e0 + i] : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// Middle step for MANY blocks (own_inclusive_scan: nb > SCAN_DIRECT_BLOCKS): the block sums become their own exclusive prefix,
// in place, by one workgroup -- each thread sums a contiguous run, the 1,024 run totals are scanned through LDS, each thread
// rewrites its run.  Linear in nb: scan_apply_kernel's re-summing of every earlier block is nb^2 / 2 reads in total, fine at
// the 1 M splats of the bench frame (0.5 M reads) and 50 M reads at 10 M splats.
__global__ __launch_bounds__(1024) void scan_bsum_kernel(int32_t* __restrict__ bsum, int nb) {
    __shared__ int32_t s_run[1024];
    const int per = (nb + 1023) / 1024, b0 = threadIdx.x * per, b1 = min(b0 + per, nb);
    int32_t tot = 0;
    for (int b = b0; b < b1; ++b) tot += bsum[b];
    s_run[threadIdx.x] = tot;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int32_t v = threadIdx.x >= d ? s_run[threadIdx.x - d] : 0;
        __syncthreads();
        s_run[threadIdx.x] += v;
        __syncthreads();
    }
    int32_t run = s_run[threadIdx.x] - tot;
    for (int b = b0; b < b1; ++b) {
        const int32_t v = bsum[b];
        bsum[b] = run;
        run += v;
    }
}

// PRE: bsum holds exclusive prefixes already (scan_bsum_kernel)
template <bool PRE>
__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t* __restrict__ in, int64_t n, const int32_t* __restrict__ bsum,
                                                         int32_t* __restrict__ out) {
    __shared__ int32_t s_w[4], s_p[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int32_t before = 0;      // sum of the earlier blocks' sums
    if (PRE) {
        before = (threadIdx.x & 63) == 0 && wv == 0 ? bsum[blockIdx.x] : 0;     // one lane carries it into the sum below
    } else {
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += bsum[b];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) before += __shfl_xor(before, m, 64);
    const int64_t e0 = (int64_t)blockIdx.x * SCAN_BLOCK + 4 * threadIdx.x;
    int32_t x[4], mine = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = (e0 + i < n) ? in[e0 + i] : 0;
        mine += x[i];
    }
    int32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) s_w[wv] = incl;
    if (lane == 0) s_p[wv] = before;
    __syncthreads();
    int32_t run = s_p[0] + s_p[1] + s_p[2] + s_p[3] + incl - mine;
#pragma unroll
    for (int w = 0; w < 4; ++w) run += w < wv ? s_w[w] : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        run += x[i];
        if (e0 + i < n) out[e0 + i] = run;
    }
}

// bsum: (n + 1023) / 1024 int32 of scratch
static void own_inclusive_scan(const int32_t* in, int32_t* out, int64_t n, int32_t* bsum, hipStream_t st) {
    const unsigned nb = blocks_for(n, SCAN_BLOCK);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(nb), dim3(256), 0, st, in, n, bsum);
    if (nb > SCAN_DIRECT_BLOCKS) {
        hipLaunchKernelGGL(scan_bsum_kernel, dim3(1), dim3(1024), 0, st, bsum, (int)nb);
        hipLaunchKernelGGL(scan_apply_kernel<true>, dim3(nb), dim3(256), 0, st, in, n, bsum, out);
    } else {
        hipLaunchKernelGGL(scan_apply_kernel<false>, dim3(nb), dim3(256), 0, st, in, n, bsum, out);
    }
}
static bool use_rocprim_scan() {
    const char* env = getenv("UNERF_SPLAT_SCAN");
    return env && strcmp(env, "rocprim") == 0;
}

extern "C" int unerf_splat_count_intersects(const int32_t* num_tiles_hit, int64_t N, int32_t* cum_tiles_hit,
                                            void* workspace, int64_t workspace_bytes, void* stream) {
    UNERF_REQUIRE(num_tiles_hit && cum_tiles_hit && workspace, "splat_count_intersects: null pointer");
    UNERF_REQUIRE(N >= 1 && N < (1ll << 31), "splat_count_intersects: bad N");
    UNERF_REQUIRE(workspace_bytes >= (int64_t)blocks_for(N, SCAN_BLOCK) * 4, "splat_count_intersects: workspace %lld bytes too small",
                  (long long)workspace_bytes);
    if (!use_rocprim_scan()) {      // (UNERF_SPLAT_SCAN=rocprim: hipcub's scan, for A/B runs)
        own_inclusive_scan(num_tiles_hit, cum_tiles_hit, N, reinterpret_cast<int32_t*>(workspace), (hipStream_t)stream);
        return unerf_check_launch("splat_count_intersects");
    }
    size_t tmp = (size_t)workspace_bytes;
    hipError_t e = hipcub::DeviceScan::InclusiveSum(workspace, tmp, num_tiles_hit, cum_tiles_hit, (int)N,
                                                    (hipStream_t)stream);
    if (e != hipSuccess) {
        unerf_set_error("splat_count_intersects: %s", hipGetErrorString(e));
        return UNERF_ERR_HIP;
    }
    return unerf_check_launch("splat_count_intersects");
}

// depth key of every splat (culled ones last) + identity payload
__global__ __launch_bounds__(256) void depth_keys_kernel(const float* __restrict__ depths,
                                                         const int32_t* __restrict__ radii, int64_t N,
                                                         uint32_t* __restrict__ keys, int32_t* __restrict__ ids) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    keys[i] = radii[i] > 0 ? (uint32_t)__float_as_int(depths[i]) : 0xFFFFFFFFu;
    ids[i] = (int32_t)i;
}

// tiles hit by the j-th splat in depth order (num_tiles_hit recovered from its inclusive scan)
__global__ __launch_bounds__(256) void sorted_counts_kernel(const int32_t* __restrict__ order,
                                                            const int32_t* __restrict__ radii,
                                                            const int32_t* __restrict__ cum, int64_t N,
                                                            int32_t* __restrict__ counts) {
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const int32_t i = order[j];
    counts[j] = radii[i] > 0 ? cum[i] - (i == 0 ? 0 : cum[i - 1]) : 0;
}

// LPS lanes per splat (8: a tight list is ~20 tiles in ~5 rows on the 1 M-splat bench frame, and the per-splat set-up -- the
// ellipse, a row range per lane, the scan -- is the same instruction stream whether 4 or 8 splats share the wave; 16 lanes per
// splat spent 338 VALU instructions per wave on 4 splats: VALU-bound, 137 of 148 us, rocprofv3 r5).
// Lane t writes entries t, t+LPS, ... of the splat's row-major tile box, so a splat's run of
// (tile, id) pairs leaves as 32-B / 64-B segments instead of one thread trickling out 2-B and 4-B stores (an average
// splat of the 1 M-splat bench scene covers 37 tiles).  Entry index = first + (ty - y0) * w + (tx - x0): the order a
// serial row-major walk produces, which is gsplat's.
template <typename TKey, int LPS>
__global__ __launch_bounds__(256) void map_intersects_kernel(const float* __restrict__ xys,
                                                             const int32_t* __restrict__ radii,
                                                             const int32_t* __restrict__ order,
                                                             const int32_t* __restrict__ cum_sorted, int64_t N, int bw,
                                                             int tbx, int tby, const float* __restrict__ conics,
                                                             const float* __restrict__ opac, TKey* __restrict__ tkeys,
                                                             int32_t* __restrict__ vals) {
    const int l16 = threadIdx.x & (LPS - 1);
    const int64_t j = (int64_t)blockIdx.x * (256 / LPS) + threadIdx.x / LPS;
    if (j >= N) return;
    const int32_t i = order[j];
    if (radii[i] <= 0) return;
    int x0, y0, x1, y1;
    const float sx = xys[(int64_t)i * 2], sy = xys[(int64_t)i * 2 + 1];
    tile_bbox(sx, sy, (float)radii[i], bw, tbx, tby, x0, y0, x1, y1);
    const int w = x1 - x0, count = w * (y1 - y0);
    if (count <= 0) return;
    const int64_t first = (j == 0) ? 0 : cum_sorted[j - 1];
    if (conics) {   // uniform: tight lists -- the rows project_kernel counted, in the same order
        const TightSplat t = tight_splat(sx, sy, opac[i], conics[(int64_t)i * 3], conics[(int64_t)i * 3 + 1],
                                         conics[(int64_t)i * 3 + 2]);
        // The slot [first, end) was sized by project_kernel's tight_count -- a second inlining of tight_row on the same
        // stored numbers.  Should the two ever disagree (another build behind UNERF_LIB, a compiler that contracts one of
        // them), the emission stays inside its slot and pads what is left with the sentinel tile `tbx * tby`, which the
        // sort puts behind every real tile and tile_bins never covers: a wrong count costs pairs, never memory safety.
        const int64_t end = cum_sorted[j];
        const TKey sentinel = (TKey)(tbx * tby);
        // LPS tile rows at a time, one per lane of the splat's lane group; their widths are scanned inside the group and
        // the round's entries go out striped over the lanes like the box form below (a row is only ~5 tiles wide: a
        // lane-per-row or row-by-row emission leaves most lanes of every store idle -- 2.5 x the kernel time)
        const int nrows = y1 - y0;
        int64_t at = first;
        for (int rb = 0; rb < nrows; rb += LPS) {
            int a0 = 0, a1 = 0;
            if (rb + l16 < nrows) tight_row(t, y0 + rb + l16, bw, x0, x1, a0, a1);
            const int wd = a1 - a0;
            int incl = wd;
#pragma unroll
            for (int d = 1; d < LPS; d <<= 1) {
                const int v = __shfl_up(incl, d, LPS);
                if (l16 >= d) incl += v;
            }
            const int total = __shfl(incl, LPS - 1, LPS), excl = incl - wd;
            for (int e0 = 0; e0 < total; e0 += LPS) {          // (uniform inside the group: every lane takes every shuffle)
                const int e = e0 + l16;
                int r = 0;                                    // row of entry e: first lane whose inclusive sum exceeds e
#pragma unroll
                for (int step = LPS / 2; step >= 1; step >>= 1) {
                    const int probe = __shfl(incl, r + step - 1, LPS);
                    if (probe <= e) r += step;
                }
                r = min(r, LPS - 1);
                const int rex = __shfl(excl, r, LPS), ra0 = __shfl(a0, r, LPS);
                if (e < total && at + e < end) {
                    tkeys[at + e] = (TKey)((y0 + rb + r) * tbx + ra0 + (e - rex));
                    vals[at + e] = i;
                }
            }
            at += total;
        }
        for (int64_t e = at + l16; e < end; e += LPS) {        // never taken when the two counts agree
            tkeys[e] = sentinel;
            vals[e] = i;
        }
        return;
    }
    // row = floor(t / w) through the reciprocal, then made exact by one step either way (the estimate is within 1
    // for any box that fits an image)
    const float rw = 1.f / (float)w;
    for (int t = l16; t < count; t += LPS) {
        int row = (int)(((float)t + 0.5f) * rw);
        row -= (row * w > t) ? 1 : 0;
        row += ((row + 1) * w <= t) ? 1 : 0;
        const int col = t - row * w;
        tkeys[first + t] = (TKey)((y0 + row) * tbx + (x0 + col));
        vals[first + t] = i;
    }
}

// tile ranges + the gsplat-style 64-bit ids (tile << 32 | depth bits) of the sorted intersections (rocprim path; entries
// with the sentinel tile `tiles` sort behind every real tile and belong to no range)
template <typename TKey>
__global__ __launch_bounds__(256) void tile_edges_kernel(const TKey* __restrict__ tkeys,
                                                         const int32_t* __restrict__ gids,
                                                         const float* __restrict__ depths, int64_t I, int32_t tiles,
                                                         int32_t* __restrict__ bins, int64_t* __restrict__ isect_ids) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= I) return;
    const int32_t cur = (int32_t)tkeys[i];
    if (isect_ids) isect_ids[i] = ((int64_t)cur << 32) | (int64_t)(uint32_t)__float_as_int(depths[gids[i]]);
    const int32_t prev = i > 0 ? (int32_t)tkeys[i - 1] : -1;
    if (cur >= tiles) {
        if (prev >= 0 && prev < tiles) bins[prev * 2 + 1] = (int32_t)i;
        return;
    }
    if (i == 0) bins[cur * 2] = 0;
    if (i == I - 1) bins[cur * 2 + 1] = (int32_t)I;
    if (i > 0 && prev != cur) {
        bins[prev * 2 + 1] = (int32_t)i;
        bins[cur * 2] = (int32_t)i;
    }
}

// ---- stable ONE-PASS tile sort ---------------------------------------------------------------------------------------
// The emission above leaves (tile, splat) pairs in depth order; what remains is a STABLE sort by tile -- at 1080p 8,160
// tiles, a 13-bit key.  rocprim's radix sort takes 8 bits per pass, i.e. two read-and-scatter sweeps over keys and
// values plus a histogram sweep (0.28 ms for the 20 M pairs of the bench frame).  A key range this small fits a
// workgroup's LDS as ONE digit: 8,161 32-bit counters are 32 KB of the CU's 160 KB.  So: one histogram per chunk of the
// pair stream (tile_hist_kernel), a prefix over chunks and tiles that turns the counts into write offsets -- which are
// also the tile_bins the rasteriser wants, so no edge-detection sweep either -- and ONE scatter sweep in which a single
// wave owns a chunk and its offset table (tile_scatter_kernel).  Stability inside a wave costs nothing here: the pairs
// of one splat name distinct tiles, so a wave hands out slots splat segment by splat segment with plain LDS atomics,
// which the LDS executes in program order.  The sorted tile keys are never written: the bins say where each tile's ids
// sit.  Same lists as the radix sort, bit for bit (tests/test_gpu_splat.py).
template <typename TKey>
__global__ __launch_bounds__(256) void tile_hist_kernel(const TKey* __restrict__ keys, int64_t I, int chunk, int T1,
                                                        uint32_t* __restrict__ table) {
    extern __shared__ uint32_t s_hist[];
    for (int t = threadIdx.x; t < T1; t += 256) s_hist[t] = 0u;
    __syncthreads();
    const int64_t k0 = (int64_t)blockIdx.x * chunk, k1 = (k0 + chunk < I) ? k0 + chunk : I;   // rows past the last chunk: zeros
    // 16 bytes of keys per lane and load (chunk starts are multiples of 256 keys: aligned); the tail key by key
    constexpr int PER = 16 / (int)sizeof(TKey);
    const int64_t nvec = k1 > k0 ? (k1 - k0) / PER : 0;
    const uint4* kv = reinterpret_cast<const uint4*>(keys + k0);
    for (int64_t v = threadIdx.x; v < nvec; v += 256) {
        const uint4 q = kv[v];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (sizeof(TKey) == 2) {
                const uint32_t a = w[i] & 0xFFFFu, b = w[i] >> 16;
                atomicAdd(&s_hist[a < (uint32_t)T1 ? a : (uint32_t)(T1 - 1)], 1u);
                atomicAdd(&s_hist[b < (uint32_t)T1 ? b : (uint32_t)(T1 - 1)], 1u);
            } else {
                atomicAdd(&s_hist[w[i] < (uint32_t)T1 ? w[i] : (uint32_t)(T1 - 1)], 1u);
            }
        }
    }
    for (int64_t k = k0 + nvec * PER + threadIdx.x; k < k1; k += 256) {
        const uint32_t key = (uint32_t)keys[k];
        atomicAdd(&s_hist[key < (uint32_t)T1 ? key : (uint32_t)(T1 - 1)], 1u);
    }
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * T1;
    for (int t = threadIdx.x; t < T1; t += 256) row[t] = s_hist[t];
}

// column sums of each row segment: segsum[seg][t] = sum over the segment's rows of table[row][t]
__global__ __launch_bounds__(256) void tile_colsum_kernel(const uint32_t* __restrict__ table, int rows_per_seg, int T1,
                                                          uint32_t* __restrict__ segsum) {
    const int t = blockIdx.x * 256 + threadIdx.x, seg = blockIdx.y;
    if (t >= T1) return;
    const uint32_t* col = table + (size_t)seg * rows_per_seg * T1 + t;
    uint32_t sum = 0u;
#pragma unroll 8
    for (int r = 0; r < rows_per_seg; ++r) sum += col[(size_t)r * T1];
    segsum[(size_t)seg * T1 + t] = sum;
}

// per tile: the exclusive prefix of its segment sums over the segments (in place) and the tile's total
__global__ __launch_bounds__(256) void tile_segscan_kernel(uint32_t* __restrict__ segsum, int T1, uint32_t* __restrict__ total) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T1) return;
    uint32_t x[TS_SEG];
#pragma unroll
    for (int seg = 0; seg < TS_SEG; ++seg) x[seg] = segsum[(size_t)seg * T1 + t];
    uint32_t run = 0u;
#pragma unroll
    for (int seg = 0; seg < TS_SEG; ++seg) {
        segsum[(size_t)seg * T1 + t] = run;
        run += x[seg];
    }
    total[t] = run;
}

// one workgroup: the exclusive prefix of the tile totals over the tiles -> start[t] (start[T1] = all pairs), and the
// tile_bins the rasteriser reads.  A tile's total is the sum of NSEG partial rows total[s][t] (1: already summed).
template <int NSEG>
__global__ __launch_bounds__(1024) void tile_scan_kernel(const uint32_t* __restrict__ total, int T1, int tiles,
                                                         uint32_t* __restrict__ start, int32_t* __restrict__ bins) {
    extern __shared__ uint32_t s_tot[];      // [T1] tile totals, then [16] wave sums
    uint32_t* s_w = s_tot + T1;
    for (int t = threadIdx.x; t < T1; t += 1024) {      // (coalesced over the tiles, segment by segment)
        uint32_t n = 0u;
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) n += total[(size_t)sg * T1 + t];
        s_tot[t] = n;
    }
    __syncthreads();
    const int per = (T1 + 1023) / 1024;
    const int t0 = threadIdx.x * per, t1 = (t0 + per < T1) ? t0 + per : T1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t mine = 0u;
    for (int t = t0; t < t1; ++t) mine += s_tot[t];
    uint32_t incl = mine;      // inclusive scan of the 1,024 partial sums: inside each wave, then over the 16 waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    uint32_t before = 0u;
#pragma unroll
    for (int w = 0; w < 16; ++w) before += w < wv ? s_w[w] : 0u;
    uint32_t run = before + incl - mine;
    for (int t = t0; t < t1; ++t) {
        const uint32_t n = s_tot[t];
        start[t] = run;
        if (t < tiles) {      // an empty tile keeps the (0, 0) of gsplat's zero-filled tile_bins
            bins[t * 2] = n ? (int32_t)run : 0;
            bins[t * 2 + 1] = n ? (int32_t)(run + n) : 0;
        }
        run += n;
    }
    if (threadIdx.x == 1023) start[T1] = run;
}

// counts -> write offsets, in place: table[row][t] = start[t] + segsum[seg][t] + sum of the segment's earlier rows
__global__ __launch_bounds__(256) void tile_apply_kernel(uint32_t* __restrict__ table, int rows_per_seg, int T1,
                                                         const uint32_t* __restrict__ segsum,
                                                         const uint32_t* __restrict__ start) {
    const int t = blockIdx.x * 256 + threadIdx.x, seg = blockIdx.y;
    if (t >= T1) return;
    uint32_t* col = table + (size_t)seg * rows_per_seg * T1 + t;
    uint32_t run = start[t] + segsum[(size_t)seg * T1 + t];
#pragma unroll 8
    for (int r = 0; r < rows_per_seg; ++r) {
        const uint32_t c = col[(size_t)r * T1];
        col[(size_t)r * T1] = run;
        run += c;
    }
}

// One wave per chunk.  Pairs arrive in depth order; within a 64-pair vector the pairs of ONE splat (a run of equal ids)
// name distinct tiles, so all of them can take their slots with one conflict-free LDS atomic; runs are served in order
// (the LDS executes a wave's operations in program order), which makes the scatter stable.  A wave is alone with its
// memory latency (four waves per CU: the 32-KB tables), so it works on four vectors per round and asks for the next
// round's pairs before it ranks this round's.
#define TS_VEC 4
#ifndef UNERF_SPLAT_XCD
#define UNERF_SPLAT_XCD 0
#endif
template <typename TKey>
__global__ __launch_bounds__(64) void tile_scatter_kernel(const TKey* __restrict__ keys, const int32_t* __restrict__ vals,
                                                          int64_t I, int chunk, int T1, const uint32_t* __restrict__ table,
                                                          int32_t* __restrict__ out) {
    extern __shared__ uint32_t s_off[];
    const int lane = threadIdx.x;
    // XCD-aware chunk order (UNERF_SPLAT_XCD): workgroups go to the 8 XCDs round-robin, and chunk c writes, for every tile,
    // the slots right behind chunk c - 1's.  With chunk = blockIdx the eight chunks that fill one 64-byte line of a tile's
    // list run on eight different XCDs, each of which holds the line partially written in its own L2 (WRITE_SIZE 7.6 x the
    // payload, rocprofv3 r4_09).  Giving every XCD a CONTIGUOUS eighth of the chunks makes it the only writer of (almost)
    // every line it touches, and its ~128 co-resident waves are exactly consecutive chunks: the line fills while it is
    // still in that L2.  Same slots, same lists.
#if UNERF_SPLAT_XCD
    const int cpx = (int)(gridDim.x >> 3);      // the launcher rounds the grid up to 8 x ceil(nblk / 8) workgroups
    const int cid = (int)(blockIdx.x & 7u) * cpx + (int)(blockIdx.x >> 3);
    if ((int64_t)cid * chunk >= I) return;      // (the whole single-wave workgroup: a slot past the last chunk)
#else
    const int cid = (int)blockIdx.x;
#endif
    {
        const uint32_t* row = table + (size_t)cid * T1;
#pragma unroll 8
        for (int t = lane; t < T1; t += 64) s_off[t] = row[t];
    }
    __syncthreads();
    const int64_t k0 = (int64_t)cid * chunk, k1 = (k0 + chunk < I) ? k0 + chunk : I;
    uint32_t key[TS_VEC], nkey[TS_VEC];
    int32_t id[TS_VEC], nid[TS_VEC];
    auto fetch = [&](int64_t k, uint32_t (&kk)[TS_VEC], int32_t (&ii)[TS_VEC]) {
#pragma unroll
        for (int v = 0; v < TS_VEC; ++v) {
            const int64_t e = k + 64 * v + lane;
            const bool ok = e < k1;
            kk[v] = ok ? (uint32_t)keys[e] : 0u;
            ii[v] = ok ? vals[e] : -1;
        }
    };
    fetch(k0, nkey, nid);
    for (int64_t k = k0; k < k1; k += 64 * TS_VEC) {
#pragma unroll
        for (int v = 0; v < TS_VEC; ++v) {
            key[v] = nkey[v];
            id[v] = nid[v];
        }
        if (k + 64 * TS_VEC < k1) fetch(k + 64 * TS_VEC, nkey, nid);
#pragma unroll
        for (int v = 0; v < TS_VEC; ++v) {
            const bool valid = k + 64 * v + lane < k1;
            const uint32_t kq = key[v] < (uint32_t)T1 ? key[v] : (uint32_t)(T1 - 1);
            const int32_t prev = __shfl_up(id[v], 1, 64);
            uint64_t starts = __builtin_amdgcn_ballot_w64(valid && (lane == 0 || id[v] != prev));
            uint32_t pos = 0u;
            while (starts) {      // uniform: one round per splat run of the vector
                const int lo = __builtin_ctzll(starts);
                starts &= starts - 1;
                const int hi = starts ? __builtin_ctzll(starts) : 64;
                if (valid && lane >= lo && lane < hi) pos = atomicAdd(&s_off[kq], 1u);
            }
            if (valid) out[pos] = id[v];      // (sentinel pairs land behind the last tile's range)
        }
    }
}

// ---- staged LSD radix passes ---------------------------------------------------------------------------------------------
// MI355X retires about 81 G isolated 4-byte stores per second, whatever the occupancy and wherever they land (246 us for
// the 20 M ids of the bench frame, benchmarks/exp_scatter_store.hip; 132 / 77 / 53 / 38 us when 2 / 4 / 8 / 16 consecutive
// lanes write consecutive dwords) -- which is where the one-pass tile sort's scatter sits (279 us: every lane of every
// store on a cache line of its own, because a chunk of the depth-ordered stream holds ~1 pair per tile).  A stable sort by a
// SMALL digit does not have that problem: with <= 128 digit values a wave's 2,048 pairs hold >= 16 per value, so the wave
// ranks them, stages ids and keys in LDS in digit order and writes each digit's pairs as ONE run of consecutive slots.  Two
// such passes (low digit, then high digit: LSD) sort by the 13-bit tile; four of them with 8-bit digits sort the 32-bit
// depth keys.  Ranking: the lanes of a 64-pair vector that share a digit find each other with one ballot per digit bit (a
// lane keeps the lanes that agree with it on every bit); the lowest takes the slots for all with one LDS atomic, the others
// add their position among the peers -- stream order, i.e. stable.  Same lists as the one-pass sort, rocprim's radix sort
// (tile) and rocprim's stable sort (depth), bit for bit (tests/test_gpu_splat.py).
// One wave talks to itself through LDS in these kernels (lane A's atomic, lane B's read of the same word).  The LDS executes a
// wave's instructions in program order, so no wait is needed -- this only tells the COMPILER that the accesses on either side
// may not change places (wavefront-scope fences emit no instruction).
__device__ __forceinline__ void rs_wave_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename TKey, bool FULL>
__global__ __launch_bounds__(1024) void rs_hist_kernel(const TKey* __restrict__ keys, int64_t n, int chunk, int nchunk, uint32_t kmax,
                                                      int shift, int B, int cpw, uint32_t* __restrict__ table, uint32_t* __restrict__ full) {
    extern __shared__ uint32_t s_rs[];       // [waves][256] digit counters, then FULL: [kmax + 1] whole-key counters of the workgroup
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = (int)(blockDim.x >> 6);
    uint32_t* dig = s_rs + wv * 256;
    uint32_t* fh = s_rs + wpb * 256;
    if (FULL) {
        for (uint32_t t = threadIdx.x; t <= kmax; t += blockDim.x) fh[t] = 0u;
        __syncthreads();
    }
    const uint32_t dmask = (uint32_t)B - 1u;
    constexpr int PER = 16 / (int)sizeof(TKey);
    for (int cc = 0; cc < cpw; ++cc) {
        const int c = ((int)blockIdx.x * wpb + wv) * cpw + cc;
        if (c >= nchunk) break;                   // (uniform per wave)
#pragma unroll
        for (int i = 0; i < 4; ++i) dig[lane + 64 * i] = 0u;
        const int64_t k0 = (int64_t)c * chunk, k1 = (k0 + chunk < n) ? k0 + chunk : n;
        const int64_t nvec = (k1 - k0) / PER;
        const uint4* kv = reinterpret_cast<const uint4*>(keys + k0);      // chunks are multiples of 256 keys: aligned
        auto count = [&](uint32_t key) {
            const uint32_t kq = key < kmax ? key : kmax;
            atomicAdd(&dig[(kq >> shift) & dmask], 1u);
            if (FULL) atomicAdd(&fh[kq], 1u);
        };
        for (int64_t v0 = 0; v0 < nvec; v0 += 256) {      // four 16-byte loads per lane in flight
            uint4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t v = v0 + 64 * u + lane;
                q[u] = kv[v < nvec ? v : nvec - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (v0 + 64 * u + lane < nvec) {
                    const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (sizeof(TKey) == 2) {
                            count(w[i] & 0xFFFFu);
                            count(w[i] >> 16);
                        } else {
                            count(w[i]);
                        }
                    }
                }
            }
        }
        for (int64_t k = k0 + nvec * PER + lane; k < k1; k += 64) count((uint32_t)keys[k]);
        rs_wave_order();      // (the reads below see every atomic above)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < B) table[(size_t)(lane + 64 * i) * nchunk + c] = dig[lane + 64 * i];
    }
    if (FULL) {
        __syncthreads();
        uint32_t* row = full + (size_t)blockIdx.x * (kmax + 1);
        for (uint32_t t = threadIdx.x; t <= kmax; t += blockDim.x) row[t] = fh[t];
    }
}

// column sums of the whole-key histogram rows, by row segment: seg[s][t] = sum of rows [s rps, (s + 1) rps) of column t
__global__ __launch_bounds__(256) void rs_colsum_kernel(const uint32_t* __restrict__ full, int rows, int rps, int T1,
                                                        uint32_t* __restrict__ seg) {
    const int t = blockIdx.x * 256 + threadIdx.x, sg = blockIdx.y;
    if (t >= T1) return;
    const int r0 = sg * rps, r1 = (r0 + rps < rows) ? r0 + rps : rows;
    uint32_t sum = 0u;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) sum += full[(size_t)r * T1 + t];
    seg[(size_t)sg * T1 + t] = sum;
}

// one workgroup per digit: exclusive prefix of its row over the chunks (in place) and the digit's total
__global__ __launch_bounds__(1024) void rs_rowscan_kernel(uint32_t* __restrict__ table, int nchunk, uint32_t* __restrict__ dtotal) {
    __shared__ uint32_t s_w[16];
    uint32_t* row = table + (size_t)blockIdx.x * nchunk;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t carry = 0u;
    for (int base = 0; base < nchunk; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const uint32_t x = i < nchunk ? row[i] : 0u;
        uint32_t incl = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        uint32_t before = 0u, all = 0u;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t t = s_w[w];
            before += w < wv ? t : 0u;
            all += t;
        }
        if (i < nchunk) row[i] = carry + before + incl - x;
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) dtotal[blockIdx.x] = carry;
}

// one wave per chunk of M pairs: rank, stage in LDS in digit order, write out digit run by digit run.  DMAX = digit values the
// LDS tables are sized for (128: tile digits, 256: depth digits).
// Ranking a 64-pair vector: every lane ORs its lane bit into its digit's 64-bit word (one LDS instruction), reads the word back
// -- the lanes that share its digit -- and clears it; its slot is the digit's running position (read by all, advanced by the
// first of the peers) plus the number of peers in lower lanes.  The LDS executes a wave's instructions in program order, so the
// five of them need no wait on one another, and four vectors are in flight at a time.
template <int M, int DMAX, typename TKey>
struct RsLds {
    static constexpr int WAVE_WORDS = 4 * DMAX + M + (M * (int)sizeof(TKey) + 3) / 4;     // cur | delta | peer words | vals | keys
};
// COUNTS (the depth sort's last pass): also writes, in output order, how many tiles each splat hits -- num_tiles_hit recovered from
// its inclusive scan, 0 for culled splats -- the gather a separate kernel used to make (21 us, waiting on three random reads per splat).
template <typename TKey, int M, int DMAX, bool KEYS_OUT, bool COUNTS = false>
__global__ __launch_bounds__(256) void rs_scatter_kernel(const TKey* __restrict__ keys, const int32_t* __restrict__ vals, int64_t n,
                                                         int nchunk, uint32_t kmax, int shift, int B,
                                                         const uint32_t* __restrict__ table, const uint32_t* __restrict__ dtotal,
                                                         TKey* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                                                         const int32_t* __restrict__ radii = nullptr, const int32_t* __restrict__ cum = nullptr,
                                                         int32_t* __restrict__ counts_out = nullptr) {
    extern __shared__ uint32_t s_rs[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = (int)blockIdx.x * (int)(blockDim.x >> 6) + wv;      // (1 - 4 waves per workgroup: the launcher's choice)
    if (c >= nchunk) return;
    uint32_t* cur = s_rs + (size_t)wv * RsLds<M, DMAX, TKey>::WAVE_WORDS;
    uint32_t* delta = cur + DMAX;
    uint32_t* pw = delta + DMAX;             // [DMAX][2]
    int32_t* sval = reinterpret_cast<int32_t*>(pw + 2 * DMAX);
    TKey* skey = reinterpret_cast<TKey*>(sval + M);
    const uint32_t dmask = (uint32_t)B - 1u;
    constexpr int DPL = DMAX / 64;           // digits per lane in the prologue
    {   // lane l owns digits DPL l .. DPL l + DPL - 1.  For digit d: this chunk's pairs start at slot ls[d] of the staged order
        // (prefix of the chunk's counts over the digits) and at gs[d] = (all pairs with a smaller digit) + (pairs with digit d in
        // earlier chunks) of the output; the chunk's count is the difference of two neighbours of the prefix row
        uint32_t tot[DPL], pre[DPL], cnt[DPL], tsum = 0u, csum = 0u;
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int d = DPL * lane + i;
            tot[i] = pre[i] = cnt[i] = 0u;
            if (d < B) {
                tot[i] = dtotal[d];
                pre[i] = table[(size_t)d * nchunk + c];
                cnt[i] = (c + 1 < nchunk ? table[(size_t)d * nchunk + c + 1] : tot[i]) - pre[i];
            }
            tsum += tot[i];
            csum += cnt[i];
        }
        uint32_t ti = tsum, ci = csum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v0 = __shfl_up(ti, d, 64), v1 = __shfl_up(ci, d, 64);
            if (lane >= d) { ti += v0; ci += v1; }
        }
        uint32_t gb = ti - tsum, lb = ci - csum;
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int d = DPL * lane + i;
            cur[d] = lb;
            delta[d] = gb + pre[i] - lb;
            pw[2 * d] = 0u;
            pw[2 * d + 1] = 0u;
            gb += tot[i];
            lb += cnt[i];
        }
        rs_wave_order();
    }
    const int64_t k0 = (int64_t)c * M;
    const int m = (int)((n - k0 < M) ? n - k0 : M);       // pairs of this chunk
    // every pair of the chunk is requested before the first is ranked (2 registers per vector: the ranking below is a chain
    // of LDS round trips, a global load inside it would cost its whole latency once per vector)
    constexpr int NV = M / 64;
    uint32_t key[NV];
    int32_t val[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int e = 64 * v + lane;
        const int64_t ec = k0 + (e < m ? e : m - 1);       // clamped, not predicated
        key[v] = (uint32_t)keys[ec];
        val[v] = vals[ec];
    }
    const uint32_t mybit = 1u << (lane & 31), half = (uint32_t)lane >> 5;
#pragma unroll
    for (int g = 0; g < NV; g += 4) {
        uint32_t kq[4], d[4], base[4], rank[4];
        uint2 peers[4];
        bool valid[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            valid[i] = 64 * (g + i) + lane < m;
            kq[i] = key[g + i] < kmax ? key[g + i] : kmax;
            d[i] = (kq[i] >> shift) & dmask;
            atomicOr(&pw[2 * d[i] + half], valid[i] ? mybit : 0u);
            rs_wave_order();
            peers[i] = *reinterpret_cast<const uint2*>(&pw[2 * d[i]]);
            *reinterpret_cast<uint2*>(&pw[2 * d[i]]) = make_uint2(0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rank[i] = __builtin_amdgcn_mbcnt_hi(peers[i].y, __builtin_amdgcn_mbcnt_lo(peers[i].x, 0u));
            const uint32_t cntp = (uint32_t)(__builtin_popcount(peers[i].x) + __builtin_popcount(peers[i].y));
            base[i] = cur[d[i]];
            atomicAdd(&cur[d[i]], (valid[i] && rank[i] == 0u) ? cntp : 0u);      // the first of the peers advances it for all
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (valid[i]) {
                sval[base[i] + rank[i]] = val[g + i];
                skey[base[i] + rank[i]] = (TKey)kq[i];
            }
        }
    }
    rs_wave_order();      // the staged order is complete
#pragma unroll 4
    for (int j = 0; j < M / 64; ++j) {
        const int slot = 64 * j + lane;
        if (slot < m) {
            const uint32_t kq = (uint32_t)skey[slot];
            const uint32_t dst = delta[(kq >> shift) & dmask] + (uint32_t)slot;
            if (KEYS_OUT) keys_out[dst] = (TKey)kq;
            const int32_t i = sval[slot];
            vals_out[dst] = i;
            if (COUNTS) counts_out[dst] = radii[i] > 0 ? cum[i] - (i == 0 ? 0 : cum[i - 1]) : 0;
        }
    }
}

// the gsplat-style 64-bit ids of the sorted lists (on request only): the tile of entry i is found in the tile offsets
__global__ __launch_bounds__(256) void tile_isect_ids_kernel(const uint32_t* __restrict__ start, int T1,
                                                             const int32_t* __restrict__ gids,
                                                             const float* __restrict__ depths, int64_t n,
                                                             int64_t* __restrict__ isect_ids) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = T1;                       // last tile t with start[t] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= (uint32_t)i) lo = mid; else hi = mid;
    }
    isect_ids[i] = ((int64_t)lo << 32) | (int64_t)(uint32_t)__float_as_int(depths[gids[i]]);
}

template <typename TKey>
static int bin_sort_impl(const float* xys, const float* depths, const int32_t* radii, const int32_t* order,
                         const int32_t* cum_sorted, const float* conics, const float* opac, int64_t N, int64_t I, int bw,
                         int tbx, int tby, int bits, int own_sort,
                         int64_t* isect_ids_sorted, int32_t* gaussian_ids_sorted, int32_t* tile_bins, char* ws,
                         const SortWs& L, size_t tmp_bytes, hipStream_t st) {
    TKey* tk_in = reinterpret_cast<TKey*>(ws + L.tkey_in);
    TKey* tk_out = reinterpret_cast<TKey*>(ws + L.tkey_out);
    int32_t* v_in = reinterpret_cast<int32_t*>(ws + L.val_in);
    // tight lists: 8 lanes per splat; gsplat's boxes (37 tiles per splat on the bench frame): 16
    if (conics)
        hipLaunchKernelGGL((map_intersects_kernel<TKey, 8>), dim3(blocks_for(N, 32)), dim3(256), 0, st, xys, radii, order,
                           cum_sorted, N, bw, tbx, tby, conics, opac, tk_in, v_in);
    else
    hipLaunchKernelGGL((map_intersects_kernel<TKey, 16>), dim3(blocks_for(N, 16)), dim3(256), 0, st, xys, radii, order,
                       cum_sorted, N, bw, tbx, tby, conics, opac, tk_in, v_in);
    int rc = unerf_check_launch("splat_bin_sort map");
    if (rc) return rc;
    const int tiles = tbx * tby;
    if (sizeof(TKey) != 2) own_sort = 0;      // (the own sorts serve <= TS_MAX_T1 tiles: always 16-bit keys; their LDS images are sized for them)
    if (own_sort == 2) {   // two staged LSD passes (rs_* kernels above)
        const int T1 = tiles + 1;
        const RsTilePlan rp = rs_tile_plan(I, T1);
        uint32_t* base = reinterpret_cast<uint32_t*>(ws + L.ts_table);
        uint32_t* table0 = base;                                        // [B0][nchunk]
        uint32_t* table1 = table0 + (size_t)128 * rp.nchunk;            // [B1][nchunk]
        uint32_t* full = table1 + (size_t)128 * rp.nchunk;              // [nhw][T1]
        uint32_t* dtotal = reinterpret_cast<uint32_t*>(ws + L.ts_segsum);      // [128] + [128]
        uint32_t* total = dtotal + 256;                                 // [T1]
        uint32_t* start = reinterpret_cast<uint32_t*>(ws + L.ts_start);
        TKey* tk_mid = tk_out;
        int32_t* v_mid = reinterpret_cast<int32_t*>(ws + L.val_mid);
        const int swpb = 1, sgrid = (rp.nchunk + swpb - 1) / swpb;      // single-wave workgroups (1, 2 or 4 waves measured alike: 4.5.77)
        const uint32_t kmax = (uint32_t)tiles;
        const size_t lds_full = (16 * 256 + (size_t)T1) * sizeof(uint32_t), lds_dig = 1024 * sizeof(uint32_t);
        const int hgrid = (rp.nchunk + 3) / 4;      // digit-only histogram: one chunk per wave, four waves per workgroup
        const size_t lds_sc = swpb * (size_t)RsLds<RS_M_TILE, 128, TKey>::WAVE_WORDS * sizeof(uint32_t);
        if (rp.b0 == 0) {      // <= 128 keys: one pass
            hipLaunchKernelGGL((rs_hist_kernel<TKey, true>), dim3(rp.nhw), dim3(1024), lds_full, st, tk_in, I, RS_M_TILE, rp.nchunk, kmax, 0,
                               rp.B1, 1, table1, full);
            hipLaunchKernelGGL(rs_rowscan_kernel, dim3(rp.B1), dim3(1024), 0, st, table1, rp.nchunk, dtotal + 128);
            hipLaunchKernelGGL((rs_scatter_kernel<TKey, RS_M_TILE, 128, false>), dim3(sgrid), dim3(64 * swpb), lds_sc, st, tk_in, v_in, I, rp.nchunk, kmax,
                               0, rp.B1, table1, dtotal + 128, (TKey*)nullptr, gaussian_ids_sorted);
        } else {
            hipLaunchKernelGGL((rs_hist_kernel<TKey, true>), dim3(rp.nhw), dim3(1024), lds_full, st, tk_in, I, RS_M_TILE, rp.nchunk, kmax, 0,
                               rp.B0, 1, table0, full);
            hipLaunchKernelGGL(rs_rowscan_kernel, dim3(rp.B0), dim3(1024), 0, st, table0, rp.nchunk, dtotal);
            hipLaunchKernelGGL((rs_scatter_kernel<TKey, RS_M_TILE, 128, true>), dim3(sgrid), dim3(64 * swpb), lds_sc, st, tk_in, v_in, I, rp.nchunk, kmax,
                               0, rp.B0, table0, dtotal, tk_mid, v_mid);
            hipLaunchKernelGGL((rs_hist_kernel<TKey, false>), dim3(hgrid), dim3(256), lds_dig, st, tk_mid, I, RS_M_TILE, rp.nchunk, kmax,
                               rp.b0, rp.B1, 1, table1, (uint32_t*)nullptr);
            hipLaunchKernelGGL(rs_rowscan_kernel, dim3(rp.B1), dim3(1024), 0, st, table1, rp.nchunk, dtotal + 128);
            hipLaunchKernelGGL((rs_scatter_kernel<TKey, RS_M_TILE, 128, false>), dim3(sgrid), dim3(64 * swpb), lds_sc, st, tk_mid, v_mid, I, rp.nchunk,
                               kmax, rp.b0, rp.B1, table1, dtotal + 128, (TKey*)nullptr, gaussian_ids_sorted);
        }
        // tile totals (column sums of the whole-key histograms) -> tile starts and the tile_bins
        {   // (16 row segments summed in parallel; tile_scan_kernel adds them up per tile)
            const int rps = (rp.nhw + 15) / 16;
            hipLaunchKernelGGL(rs_colsum_kernel, dim3(blocks_for(T1, 256), 16), dim3(256), 0, st, full, rp.nhw, rps, T1, total);
        }
        hipLaunchKernelGGL(tile_scan_kernel<16>, dim3(1), dim3(1024), ((size_t)T1 + 16) * sizeof(uint32_t), st, total, T1, tiles, start, tile_bins);
        if (isect_ids_sorted)
            hipLaunchKernelGGL(tile_isect_ids_kernel, dim3(blocks_for(I, 256)), dim3(256), 0, st, start, T1,
                               gaussian_ids_sorted, depths, I, isect_ids_sorted);
        return unerf_check_launch("splat_bin_sort two-pass tile sort");
    }
    if (own_sort == 1) {   // the one-pass LDS-digit sort (above)
        const TileSortPlan tp = tile_sort_plan(I, tiles);
        uint32_t* table = reinterpret_cast<uint32_t*>(ws + L.ts_table);
        uint32_t* segsum = reinterpret_cast<uint32_t*>(ws + L.ts_segsum);
        uint32_t* start = reinterpret_cast<uint32_t*>(ws + L.ts_start);
        const size_t lds = (size_t)tp.T1 * sizeof(uint32_t);
        const int rps = tp.rows / TS_SEG;
        hipLaunchKernelGGL((tile_hist_kernel<TKey>), dim3(tp.rows), dim3(256), lds, st, tk_in, I, tp.chunk, tp.T1, table);
        hipLaunchKernelGGL(tile_colsum_kernel, dim3(blocks_for(tp.T1, 256), TS_SEG), dim3(256), 0, st, table, rps, tp.T1, segsum);
        uint32_t* total = start + tp.T1 + 1;
        hipLaunchKernelGGL(tile_segscan_kernel, dim3(blocks_for(tp.T1, 256)), dim3(256), 0, st, segsum, tp.T1, total);
        hipLaunchKernelGGL(tile_scan_kernel<1>, dim3(1), dim3(1024), ((size_t)tp.T1 + 16) * sizeof(uint32_t), st, total, tp.T1, tiles, start, tile_bins);
        hipLaunchKernelGGL(tile_apply_kernel, dim3(blocks_for(tp.T1, 256), TS_SEG), dim3(256), 0, st, table, rps, tp.T1, segsum,
                           start);
        hipLaunchKernelGGL((tile_scatter_kernel<TKey>), dim3(UNERF_SPLAT_XCD ? ((tp.nblk + 7) / 8) * 8 : tp.nblk), dim3(64), lds, st,
                           tk_in, v_in, I, tp.chunk, tp.T1, table, gaussian_ids_sorted);
        if (isect_ids_sorted)
            hipLaunchKernelGGL(tile_isect_ids_kernel, dim3(blocks_for(I, 256)), dim3(256), 0, st, start, tp.T1,
                               gaussian_ids_sorted, depths, I, isect_ids_sorted);
        return unerf_check_launch("splat_bin_sort tile sort");
    }
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(ws + L.tmp, tmp_bytes, tk_in, tk_out, v_in, gaussian_ids_sorted,
                                                      (int)I, 0, bits, st);
    if (e != hipSuccess) {
        unerf_set_error("splat_bin_sort: tile sort: %s", hipGetErrorString(e));
        return UNERF_ERR_HIP;
    }
    hipLaunchKernelGGL((tile_edges_kernel<TKey>), dim3(blocks_for(I, 256)), dim3(256), 0, st, tk_out,
                       gaussian_ids_sorted, depths, I, tiles, tile_bins, isect_ids_sorted);
    return unerf_check_launch("splat_bin_sort edges");
}

extern "C" int unerf_splat_bin_sort(const float* xys, const float* depths, const int32_t* radii,
                                    const int32_t* cum_tiles_hit, int64_t N, int64_t I, int H, int W, int block_width,
                                    const float* tight_conics, const float* tight_opacities,
                                    int64_t* isect_ids_sorted, int32_t* gaussian_ids_sorted, int32_t* tile_bins,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
    UNERF_REQUIRE(xys && depths && radii && cum_tiles_hit && tile_bins && workspace, "splat_bin_sort: null pointer");
    UNERF_REQUIRE((tight_conics == nullptr) == (tight_opacities == nullptr),
                  "splat_bin_sort: tight lists need both the conics and the opacities the tile counts were made with");
    UNERF_REQUIRE(N >= 1 && N < (1ll << 31) && I >= 0 && I < (1ll << 31), "splat_bin_sort: bad N/I");
    hipStream_t st = (hipStream_t)stream;
    int tbx = (W + block_width - 1) / block_width, tby = (H + block_width - 1) / block_width;
    if (hipMemsetAsync(tile_bins, 0, (size_t)tbx * tby * 2 * sizeof(int32_t), st) != hipSuccess)
        return unerf_check_launch("splat_bin_sort memset");
    if (I == 0) return UNERF_OK;
    UNERF_REQUIRE(gaussian_ids_sorted, "splat_bin_sort: null output");
    const SortWs L = sort_ws_layout(N, I);
    UNERF_REQUIRE(workspace_bytes >= L.total, "splat_bin_sort: workspace %lld < %lld bytes (unerf_splat_sort_workspace_bytes)",
                  (long long)workspace_bytes, (long long)L.total);
    char* ws = (char*)workspace;
    size_t tmp_bytes = (size_t)(L.dkey_in - L.tmp);
    uint32_t* dk_in = reinterpret_cast<uint32_t*>(ws + L.dkey_in);
    uint32_t* dk_out = reinterpret_cast<uint32_t*>(ws + L.dkey_out);
    int32_t* id_in = reinterpret_cast<int32_t*>(ws + L.id_in);
    int32_t* order = reinterpret_cast<int32_t*>(ws + L.order);
    int32_t* counts = reinterpret_cast<int32_t*>(ws + L.counts);
    int32_t* cum_sorted = reinterpret_cast<int32_t*>(ws + L.cum);
    // 1. splats in depth order (stable: equal depths keep their index order)
    hipError_t e = hipSuccess;
    const char* denv = getenv("UNERF_SPLAT_DEPTH_SORT");
    if (denv && strcmp(denv, "rocprim") == 0) {      // rocprim's stable sort (kept for A/B runs and the identity test)
        hipLaunchKernelGGL(depth_keys_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, st, depths, radii, N, dk_in, id_in);
        e = depth_sort_pairs(ws + L.tmp, tmp_bytes, dk_in, dk_out, id_in, order, N, st);
        if (e != hipSuccess) {
            unerf_set_error("splat_bin_sort: depth sort: %s", hipGetErrorString(e));
            return UNERF_ERR_HIP;
        }
    } else {      // four staged 8-bit LSD passes (rs_* kernels); the last one writes `order`
        const int nchunk = (int)((N + RS_M_DEPTH - 1) / RS_M_DEPTH), grid = (nchunk + 3) / 4, hgrid = grid;
        uint32_t* table = reinterpret_cast<uint32_t*>(ws + L.ds_table);
        uint32_t* dtotal = reinterpret_cast<uint32_t*>(ws + L.ds_total);
        const size_t lds_sc = 4 * (size_t)RsLds<RS_M_DEPTH, 256, uint32_t>::WAVE_WORDS * sizeof(uint32_t), lds_dig = 1024 * sizeof(uint32_t);
        hipLaunchKernelGGL(depth_keys_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, st, depths, radii, N, dk_out, order);
        uint32_t* kin = dk_out; uint32_t* kout = dk_in;
        int32_t* vin = order; int32_t* vout = id_in;
        for (int p = 0; p < 4; ++p) {      // (dk_out, order) -> (dk_in, id_in) -> (dk_out, order) -> (dk_in, id_in) -> order
            hipLaunchKernelGGL((rs_hist_kernel<uint32_t, false>), dim3(hgrid), dim3(256), lds_dig, st, kin, N, RS_M_DEPTH, nchunk, 0xFFFFFFFFu,
                               8 * p, 256, 1, table, (uint32_t*)nullptr);
            hipLaunchKernelGGL(rs_rowscan_kernel, dim3(256), dim3(1024), 0, st, table, nchunk, dtotal);
            if (p < 3)
                hipLaunchKernelGGL((rs_scatter_kernel<uint32_t, RS_M_DEPTH, 256, true>), dim3(grid), dim3(256), lds_sc, st, kin, vin, N, nchunk,
                                   0xFFFFFFFFu, 8 * p, 256, table, dtotal, kout, vout);
            else
                hipLaunchKernelGGL((rs_scatter_kernel<uint32_t, RS_M_DEPTH, 256, false, true>), dim3(grid), dim3(256), lds_sc, st, kin, vin, N, nchunk,
                                   0xFFFFFFFFu, 8 * p, 256, table, dtotal, (uint32_t*)nullptr, order, radii, cum_tiles_hit, counts);
            uint32_t* tk = kin; kin = kout; kout = tk;
            int32_t* tv = vin; vin = vout; vout = tv;
        }
        int rc = unerf_check_launch("splat_bin_sort depth sort");
        if (rc) return rc;
    }
    // 2. where each depth-ordered splat's intersections start
    if (denv && strcmp(denv, "rocprim") == 0)      // (the staged depth sort's last pass has written them)
        hipLaunchKernelGGL(sorted_counts_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, st, order, radii, cum_tiles_hit, N,
                           counts);
    tmp_bytes = (size_t)(L.dkey_in - L.tmp);
    if (!use_rocprim_scan()) {
        own_inclusive_scan(counts, cum_sorted, N, reinterpret_cast<int32_t*>(ws + L.tmp), st);
    } else {
        e = hipcub::DeviceScan::InclusiveSum(ws + L.tmp, tmp_bytes, counts, cum_sorted, (int)N, st);
        if (e != hipSuccess) {
            unerf_set_error("splat_bin_sort: scan: %s", hipGetErrorString(e));
            return UNERF_ERR_HIP;
        }
    }
    // 3. emit in depth order, stable sort by tile (two staged LSD passes when the image has <= TS_MAX_T1 - 1 tiles; rocprim's
    // radix sort beyond that or when UNERF_SPLAT_TILE_SORT=radix asks for it, the round-4 one-pass sort with =onepass: A/B
    // timing and the identity tests), tile ranges + ids
    const int bits = tile_bits(H, W, block_width);      // (the sentinel tile `tbx * tby` included)
    const char* env = getenv("UNERF_SPLAT_TILE_SORT");
    int own_sort = 0;      // 0: rocprim radix sort, 1: one-pass LDS-digit sort, 2: two-pass LSD sort (default)
    if (tbx * tby + 1 <= TS_MAX_T1 && !(env && strcmp(env, "radix") == 0)) own_sort = (env && strcmp(env, "onepass") == 0) ? 1 : 2;
    tmp_bytes = (size_t)(L.dkey_in - L.tmp);
    if (bits <= 16)
        return bin_sort_impl<uint16_t>(xys, depths, radii, order, cum_sorted, tight_conics, tight_opacities, N, I, block_width, tbx, tby, bits,
                                       own_sort, isect_ids_sorted, gaussian_ids_sorted, tile_bins, ws, L, tmp_bytes, st);
    return bin_sort_impl<uint32_t>(xys, depths, radii, order, cum_sorted, tight_conics, tight_opacities, N, I, block_width, tbx, tby, bits,
                                   own_sort, isect_ids_sorted, gaussian_ids_sorted, tile_bins, ws, L, tmp_bytes, st);
}

// ======================================================================================
// rasteriser: one 16x16 tile per workgroup, 256-splat LDS batches, C channels at once
// ======================================================================================
// gsplat's schedule (a tile per workgroup, a pixel per thread, the tile's depth-sorted splats staged through shared
// memory in batches) with two MI355X-side changes that leave every blended term untouched:
//  * WAVE-LEVEL CULLING.  A wave64 of a 16 x 16 tile is one 8 x 8 quadrant of it.  A splat contributes to a pixel
//    only where alpha = min(0.999, o e^-sigma) >= 1/255, i.e. inside the ellipse sigma <= ln(255 o) -- usually a good
//    deal smaller than the 3-sigma square the tile lists are built from (and empty when o < 1/255).  The thread that
//    stages a splat tests the ellipse's bounding box against the tile's four quadrants (conservatively: a margin far
//    above the rounding of the in-loop test) and leaves a 4-bit mask; every wave then compacts the batch into its own
//    index list with ballots and walks only that.  A culled (pixel, splat) pair is one the loop body would have skipped with
//    `continue`, so sums, transmittances and final indices are bit-identical to the uncull loop (test_gpu_splat.py).
//  * BOUNDED second pass.  The depth-variance pass blends with the same alphas as the first, so each pixel stops at the
//    final index the first pass recorded instead of re-deriving it from the transmittance (stop_idx).
struct RasterArgs {
    const int32_t* ids;
    const int32_t* bins;
    const float* xys;
    const float* conics;
    const float* colors;
    const float* opac;
    const float* bg;
    int H, W, bw;
    float* out;
    float* finalT;
    int32_t* final_idx;
    const int32_t* stop_idx;   // BOUNDED: per-pixel last index to visit (a previous pass's final_idx)
    int cull;                  // 0: walk every staged splat (reference schedule; kept for the identity test)
    unsigned int* chan_max;    // NULL, or the running max (float bits, values >= 0) of out[..., max_ch] over the image
    int max_ch;
};

// which of the tile's four 8 x 8 quadrants can see the splat?  Conservative: returns 0xF when in doubt.  (A wave64 of the
// 16 x 16 tile is one quadrant -- bit q = 2 (row half) + (column half) -- instead of a 4-row strip: the square has the
// smallest perimeter a 64-pixel footprint can have, so an r = 2..4 px ellipse reaches fewer of them.)
__device__ __forceinline__ uint32_t raster_quad_mask(float x, float y, float op, float ca, float cb, float cc, float tile_x0,
                                                     float tile_y0) {
    if (!(op >= 0.0039f)) return (op != op) ? 0xFu : 0u;      // alpha <= opacity < 1/255 (0.00392...) everywhere
    const float det = ca * cc - cb * cb;
    if (!(det > 0.f) || !(ca > 0.f) || !(cc > 0.f)) return 0xFu;   // not an ellipse (or NaN): no culling
    // sigma <= tau bounds |dx| <= sqrt(2 tau cc / det), |dy| <= sqrt(2 tau ca / det); tau = ln(255 o) padded by 1 % + 0.01,
    // the box by another 1 % + 0.05 px: orders of magnitude above the fp32 rounding of sigma and of __expf
    const float tau = fmaf(__logf(255.f * op), 1.01f, 0.01f);
    const float inv = 2.f * tau / det;
    const float hx = fmaf(sqrtf(inv * cc), 1.01f, 0.05f), hy = fmaf(sqrtf(inv * ca), 1.01f, 0.05f);
    if (!(hx == hx) || !(hy == hy)) return 0xFu;
    // pixel centres of quadrant (qx, qy): x in [x0 + 8 qx + 0.5, x0 + 8 qx + 7.5], y likewise
    uint32_t mx = 0u, my = 0u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float xlo = tile_x0 + 8.f * (float)q + 0.5f, ylo = tile_y0 + 8.f * (float)q + 0.5f;
        if (!(x + hx < xlo || x - hx > xlo + 7.f)) mx |= 1u << q;
        if (!(y + hy < ylo || y - hy > ylo + 7.f)) my |= 1u << q;
    }
    // bit 2 qy + qx
    return ((my & 1u) ? mx : 0u) | ((my & 2u) ? (mx << 2) : 0u);
}

template <int C, bool BOUNDED>
__global__ __launch_bounds__(256) void raster_kernel(RasterArgs a) {
    // one 16-byte + one 8-byte broadcast read per splat instead of six 4-byte ones
    // one record per staged splat, REC4 16-byte words: [x, y, opacity, conic a | conic b, c, colour 0, 1 | colour 2 ...]
    constexpr int REC4 = (6 + C + 3) / 4;
    __shared__ float4 s_rec[256 * REC4];
    __shared__ uint8_t s_mask[256];
    __shared__ uint16_t s_list[4][256];
    const int bw = a.bw;
    const int tbx = (a.W + bw - 1) / bw;
    // XCD-aware tile order (UNERF_SPLAT_XCD): every XCD rasterises a contiguous eighth of the row-major tile list -- a band of
    // the image -- so the splats its tiles gather (neighbouring tiles share most of theirs) are looked up in ONE L2 instead
    // of all eight.  Pure scheduling: a tile is computed exactly as before.
#if UNERF_SPLAT_XCD
    const int ntile = tbx * (int)gridDim.y;
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x), tpx = (ntile + 7) >> 3;
    const int tile = (lin & 7) * tpx + (lin >> 3);
    if ((lin >> 3) >= tpx || tile >= ntile) return;      // the launcher pads the grid to at least 8 x ceil(ntile / 8) workgroups
    const int bx = tile % tbx, by = tile / tbx;
#else
    const int bx = blockIdx.x, by = blockIdx.y;
    const int tile = by * tbx + bx;
#endif
    const int tr = threadIdx.x;
    // 16-wide tiles: a wave owns one 8 x 8 quadrant (wave w: quadrant column w & 1, row w >> 1; lane l: pixel (l & 7,
    // l >> 3) of it); narrower tiles keep the row-major mapping (no culling there)
    const int ly = bw == 16 ? 8 * (tr >> 7) + ((tr & 63) >> 3) : tr / bw;
    const int lx = bw == 16 ? 8 * ((tr >> 6) & 1) + (tr & 7) : tr - (tr / bw) * bw;
    const int i = by * bw + ly, j = bx * bw + lx;
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const bool inside = (ly < bw) && (i < a.H) && (j < a.W);
    const int64_t p = (int64_t)i * a.W + j;
    bool done = !inside;
    const int r0 = a.bins[tile * 2], r1 = a.bins[tile * 2 + 1];
    const int nbatch = (r1 - r0 + 255) / 256;
    const int stop = (BOUNDED && inside) ? a.stop_idx[p] : 0;
    const bool cull = a.cull && bw == 16;                       // uniform: the quadrant geometry is the 16-wide tile's
    const int wv = __builtin_amdgcn_readfirstlane(tr >> 6), lane = tr & 63;
    const float tile_x0 = (float)(bx * bw), tile_y0 = (float)(by * bw);
    float T = 1.f;
    int cur_idx = 0;
    float pix[C];
#pragma unroll
    for (int c = 0; c < C; ++c) pix[c] = 0.f;
    for (int b = 0; b < nbatch; ++b) {
        const int start = r0 + 256 * b;
        if (BOUNDED) done = done || start > stop;
        if (__syncthreads_count(done ? 1 : 0) >= 256) break;
        const int idx = start + tr;
        if (idx < r1) {
            int g = a.ids[idx];
            const float x = a.xys[g * 2], y = a.xys[g * 2 + 1], op = a.opac[g];
            const float ca = a.conics[g * 3], cb = a.conics[g * 3 + 1], cc = a.conics[g * 3 + 2];
            struct __attribute__((packed, aligned(4))) Q4 { float x, y, z, w; };
            const float* crow = a.colors + (int64_t)g * C;
            float rec[4 * REC4];
            rec[0] = x; rec[1] = y; rec[2] = op; rec[3] = ca; rec[4] = cb; rec[5] = cc;
#pragma unroll
            for (int c4 = 0; c4 + 4 <= C; c4 += 4) {
                const Q4 q = *reinterpret_cast<const Q4*>(crow + c4);
                rec[6 + c4] = q.x; rec[7 + c4] = q.y; rec[8 + c4] = q.z; rec[9 + c4] = q.w;
            }
#pragma unroll
            for (int c = C & ~3; c < C; ++c) rec[6 + c] = crow[c];
#pragma unroll
            for (int c = 6 + C; c < 4 * REC4; ++c) rec[c] = 0.f;
#pragma unroll
            for (int q = 0; q < REC4; ++q) s_rec[tr * REC4 + q] = make_float4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
            if (cull) s_mask[tr] = (uint8_t)raster_quad_mask(x, y, op, ca, cb, cc, tile_x0, tile_y0);
        }
        __syncthreads();
        const int bsz = min(256, r1 - start);
        int n_mine = bsz;
        if (cull) {   // this wave's (order-preserving) list of the staged splats its quadrant can see
            n_mine = 0;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const int t = ch * 64 + lane;
                const bool need = t < bsz && ((s_mask[t] >> wv) & 1u);
                const uint64_t m = __builtin_amdgcn_ballot_w64(need);
                if (need) s_list[wv][n_mine + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint16_t)t;
                n_mine += __builtin_popcountll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the list is read back by this wave only
            __builtin_amdgcn_wave_barrier();
        }
        // Hand-pipelined walk, two splats per trip: the record of splat k + 1 and the list entries of k + 2, k + 3 are requested
        // before splat k is blended, so no blend waits for an LDS round trip (the chain list entry -> record -> colours was three
        // of them per visited splat, on the critical path of the tile's slowest quadrant).
        struct Rec { float v[4 * REC4]; };
        auto load_rec = [&](int t, Rec& r) {
#pragma unroll
            for (int q = 0; q < REC4; ++q) {
                const float4 w = s_rec[t * REC4 + q];
                r.v[4 * q] = w.x; r.v[4 * q + 1] = w.y; r.v[4 * q + 2] = w.z; r.v[4 * q + 3] = w.w;
            }
        };
        auto entry = [&](int k) { const int kk = k < n_mine ? k : n_mine - 1; return cull ? (int)s_list[wv][kk] : kk; };
        auto blend = [&](const Rec& r, int t) {
            if (!done) {
                if (BOUNDED && start + t > stop) {   // past this pixel's last blended splat of the first pass
                    done = true;
                } else {
                    float dx = r.v[0] - px, dy = r.v[1] - py, op = r.v[2];
                    float ca = r.v[3], cb = r.v[4], cc = r.v[5];
                    float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                    float alpha = fminf(0.999f, op * __expf(-sigma));
                    if (!(sigma < 0.f || alpha < 1.f / 255.f)) {
                        float nT = T * (1.f - alpha);
                        if (!BOUNDED && nT <= 1e-4f) {
                            done = true;
                        } else {
                            float vis = alpha * T;
#pragma unroll
                            for (int c = 0; c < C; ++c) pix[c] += r.v[6 + c] * vis;
                            T = nT;
                            cur_idx = start + t;
                        }
                    }
                }
            }
        };
        if (n_mine > 0) {
            Rec ra, rb;
            int ta = entry(0), tb = entry(1);
            load_rec(ta, ra);
            for (int k = 0; k < n_mine; k += 2) {
                if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;     // uniform: every pixel of this wave has finished
                load_rec(tb, rb);
                const int ta2 = entry(k + 2), tb2 = entry(k + 3);
                blend(ra, ta);
                load_rec(ta2, ra);
                if (k + 1 < n_mine) blend(rb, tb);
                ta = ta2;
                tb = tb2;
            }
        }
    }
    float vmax = 0.f;
    if (inside) {
        a.finalT[p] = T;
        if (a.final_idx) a.final_idx[p] = cur_idx;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float v = pix[c] + T * (a.bg ? a.bg[c] : 0.f);
            a.out[p * C + c] = v;
            if (c == a.max_ch) vmax = v;
        }
    }
    // The `img.max()` that the alpha normalisation of this channel needs (depth_im.detach().max(), :319 / :356) is
    // taken here, where the values are in registers: one compare per wave, and an atomic only from a wave that raises
    // the maximum (a separate reduction kernel cost 28 us per pass).
    if (a.chan_max) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, m, 64));
        if (lane == 0) {
            const unsigned int bits = __float_as_uint(fmaxf(vmax, 0.f));   // >= 0: bit patterns order like the values
            if (bits > __hip_atomic_load(a.chan_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.chan_max, bits);
        }
    }
}

extern "C" int unerf_splat_rasterize(const int32_t* gaussian_ids_sorted, const int32_t* tile_bins, const float* xys,
                                     const float* conics, const float* colors, const float* opacities,
                                     const float* background, int C, int H, int W, int block_width,
                                     const int32_t* stop_idx, int flags, int max_channel, float* chan_max, float* out_img,
                                     float* final_T, int32_t* final_idx, void* stream) {
    UNERF_REQUIRE(tile_bins && xys && conics && colors && opacities && out_img && final_T,
                  "splat_rasterize: null pointer");
    UNERF_REQUIRE(C >= 1 && C <= 8, "splat_rasterize: C=%d outside [1,8]", C);
    UNERF_REQUIRE(block_width >= 1 && block_width <= 16 && H > 0 && W > 0, "splat_rasterize: bad block_width/H/W");
    UNERF_REQUIRE((flags & ~UNERF_RASTER_NO_CULL) == 0, "splat_rasterize: unknown flags %d", flags);
    UNERF_REQUIRE(!chan_max || (max_channel >= 0 && max_channel < C), "splat_rasterize: max_channel %d outside [0,%d)",
                  max_channel, C);
    RasterArgs a;
    a.ids = gaussian_ids_sorted; a.bins = tile_bins; a.xys = xys; a.conics = conics; a.colors = colors;
    a.opac = opacities; a.bg = background; a.H = H; a.W = W; a.bw = block_width; a.out = out_img; a.finalT = final_T;
    a.final_idx = final_idx; a.stop_idx = stop_idx; a.cull = (flags & UNERF_RASTER_NO_CULL) ? 0 : 1;
    a.chan_max = reinterpret_cast<unsigned int*>(chan_max); a.max_ch = chan_max ? max_channel : -1;
    dim3 grid((W + block_width - 1) / block_width, (H + block_width - 1) / block_width), block(256);
#if UNERF_SPLAT_XCD
    {   // linear ids 0 .. 8 ceil(ntile / 8) - 1 must exist: one more grid column covers the padding (tby >= 1 workgroups more)
        const unsigned ntile = grid.x * grid.y, need = ((ntile + 7u) / 8u) * 8u;
        if (grid.x * grid.y < need) grid.x += 1;
    }
#endif
    hipStream_t st = (hipStream_t)stream;
#define UNERF_RASTER_CASE(N)                                                                       \
    case N:                                                                                        \
        if (stop_idx) hipLaunchKernelGGL((raster_kernel<N, true>), grid, block, 0, st, a);         \
        else hipLaunchKernelGGL((raster_kernel<N, false>), grid, block, 0, st, a);                 \
        break;
    switch (C) {
        UNERF_RASTER_CASE(1)
        UNERF_RASTER_CASE(2)
        UNERF_RASTER_CASE(3)
        UNERF_RASTER_CASE(4)
        UNERF_RASTER_CASE(5)
        UNERF_RASTER_CASE(6)
        UNERF_RASTER_CASE(7)
        default:
            if (stop_idx) hipLaunchKernelGGL((raster_kernel<8, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((raster_kernel<8, false>), grid, block, 0, st, a);
            break;
    }
#undef UNERF_RASTER_CASE
    return unerf_check_launch("splat_rasterize");
}

// ======================================================================================
// alpha normalisation and per-splat depth difference
// ======================================================================================
// grid-stride max with one atomic per workgroup (one per wave cost 370 us on a 1080p image: 32 K atomics
// serialise on a single L2 word)
__global__ __launch_bounds__(256) void chan_max_kernel(const float* __restrict__ img, int stride, int ch, int64_t HW,
                                                       unsigned int* __restrict__ mx) {
    __shared__ float s_max[4];
    float v = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256)
        v = fmaxf(v, img[i * stride + ch]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        const unsigned int bits = __float_as_uint(v);  // v >= 0: bit patterns order like the values
        if (bits > __hip_atomic_load(mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(mx, bits);
    }
}

__global__ __launch_bounds__(256) void alpha_norm_kernel(float* __restrict__ img, int stride, int ch,
                                                         const float* __restrict__ finalT, int64_t HW,
                                                         const float* __restrict__ mx) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    float alpha = 1.f - finalT[i];
    float v = img[i * stride + ch];
    img[i * stride + ch] = (alpha > 0.f) ? v / alpha : mx[0];
}

extern "C" int unerf_splat_alpha_normalize(float* img, int stride, int ch, const float* final_T, int64_t HW,
                                           float* scratch_max, int max_ready, void* stream) {
    UNERF_REQUIRE(img && final_T && scratch_max, "splat_alpha_normalize: null pointer");
    UNERF_REQUIRE(stride >= 1 && ch >= 0 && ch < stride && HW >= 0, "splat_alpha_normalize: bad stride/ch");
    if (HW == 0) return UNERF_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!max_ready) {   // otherwise the rasteriser that produced img left the channel's maximum in scratch_max
        if (hipMemsetAsync(scratch_max, 0, sizeof(float), st) != hipSuccess)
            return unerf_check_launch("splat_alpha_normalize memset");
        const unsigned max_blocks = 2048;  // 8 workgroups per CU
        const unsigned nblk = blocks_for(HW, 256) < max_blocks ? blocks_for(HW, 256) : max_blocks;
        hipLaunchKernelGGL(chan_max_kernel, dim3(nblk), dim3(256), 0, st, img, stride, ch, HW,
                           reinterpret_cast<unsigned int*>(scratch_max));
    }
    hipLaunchKernelGGL(alpha_norm_kernel, dim3(blocks_for(HW, 256)), dim3(256), 0, st, img, stride, ch, final_T, HW,
                       scratch_max);
    return unerf_check_launch("splat_alpha_normalize");
}

// the frame's per-pixel epilogue in one pass (unerf_splat_normalize_outputs): the alpha normalisation of channel ch as above
// plus the elementwise outputs the reference forms with torch calls around it
__global__ __launch_bounds__(256) void norm_outputs_kernel(float* __restrict__ img, int stride, int ch,
                                                           const float* __restrict__ finalT, int64_t HW,
                                                           const float* __restrict__ mx, float* __restrict__ rgb_out,
                                                           float* __restrict__ acc_out, int sq_ch, float* __restrict__ sq_out,
                                                           float* __restrict__ sqrt_out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float alpha = 1.f - finalT[i];
    const float v = img[i * stride + ch];
    const float nv = (alpha > 0.f) ? v / alpha : mx[0];
    img[i * stride + ch] = nv;
    if (rgb_out) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = img[i * stride + c];
            rgb_out[i * 3 + c] = x > 1.f ? 1.f : x;      // torch.clamp(max=1): a NaN stays a NaN
        }
    }
    if (acc_out) acc_out[i] = alpha;
    if (sq_out) {
        const float u = img[i * stride + sq_ch];
        sq_out[i] = u * u;
    }
    if (sqrt_out) sqrt_out[i] = sqrtf(nv);
}

extern "C" int unerf_splat_normalize_outputs(float* img, int stride, int ch, const float* final_T, int64_t HW,
                                             const float* scratch_max, float* rgb_out, float* acc_out, int sq_ch,
                                             float* sq_out, float* sqrt_out, void* stream) {
    UNERF_REQUIRE(img && final_T && scratch_max, "splat_normalize_outputs: null pointer");
    UNERF_REQUIRE(stride >= 1 && ch >= 0 && ch < stride && HW >= 0, "splat_normalize_outputs: bad stride/ch");
    UNERF_REQUIRE(!rgb_out || (stride >= 3 && ch >= 3), "splat_normalize_outputs: rgb_out needs channels 0..2 beside channel ch");
    UNERF_REQUIRE(!sq_out || (sq_ch >= 0 && sq_ch < stride && sq_ch != ch), "splat_normalize_outputs: bad sq_ch");
    if (HW == 0) return UNERF_OK;
    hipLaunchKernelGGL(norm_outputs_kernel, dim3(blocks_for(HW, 256)), dim3(256), 0, (hipStream_t)stream, img, stride, ch, final_T, HW,
                       scratch_max, rgb_out, acc_out, sq_ch, sq_out, sqrt_out);
    return unerf_check_launch("splat_normalize_outputs");
}

__global__ __launch_bounds__(256) void depth_sqdiff_kernel(const float* __restrict__ xys,
                                                           const float* __restrict__ depths,
                                                           const float* __restrict__ dimg, int stride, int ch, int H,
                                                           int W, int64_t N, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float fxp = floorf(xys[i * 2]), fyp = floorf(xys[i * 2 + 1]);
    float z = depths[i];
    // reference uses strict ">0" on both axes (activesplatfacto_model.py:327-332)
    bool valid = fxp > 0.f && fxp < (float)W && fyp > 0.f && fyp < (float)H;
    float d = z;
    if (valid) d = z - dimg[((int64_t)fyp * W + (int64_t)fxp) * stride + ch];
    out[i] = d * d;
}

extern "C" int unerf_splat_depth_sqdiff(const float* xys, const float* depths, const float* depth_img, int stride,
                                        int ch, int H, int W, int64_t N, float* sq_diff_out, void* stream) {
    UNERF_REQUIRE(N <= 0 || (xys && depths && depth_img && sq_diff_out), "splat_depth_sqdiff: null pointer");
    UNERF_REQUIRE(stride >= 1 && ch >= 0 && ch < stride, "splat_depth_sqdiff: bad stride/ch");
    if (N <= 0) return UNERF_OK;
    hipLaunchKernelGGL(depth_sqdiff_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, xys, depths,
                       depth_img, stride, ch, H, W, N, sq_diff_out);
    return unerf_check_launch("splat_depth_sqdiff");
}
