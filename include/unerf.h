/*
 * libunerf -- C ABI of the MI355X (gfx950) uncertainty-rendering hot path.
 *
 * The reference (AaltoML/uncertainty-nerf-gs) is pure Python and has no FFI of its
 * own: every number on its hot path is produced by nerfstudio 1.1.0 torch ops,
 * tiny-cuda-nn and gsplat 0.1.11 CUDA kernels that it *calls*.  Each entry point
 * below therefore replaces a call site of the reference (cited per function,
 * paths relative to /root/reference/nerfuncertainty) and is what a maintainer
 * would bind from the reference's Model/Field classes (INTEGRATION.md shows the
 * ctypes stubs).
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer unless it says "host";
 *     the library never allocates, frees or retains caller memory
 *     (exception: unerf_splat_sort_workspace_bytes + caller-provided workspace);
 *   - `stream` is the caller's hipStream_t (NULL = default stream); no hidden sync;
 *   - all functions return 0 on success, <0 on error; unerf_last_error() gives
 *     a thread-local message;
 *   - fp32 everywhere; row-major; rays are in image row-major order;
 *   - a count of zero (R, N, count = 0: an empty chunk, a rank without views, a crop that keeps nothing) is a
 *     successful no-op and the per-element pointers may then be NULL; parameter structs and host pointers must
 *     still be valid.  Exceptions: unerf_splat_count_intersects / unerf_splat_bin_sort need N >= 1;
 *   - there is NO CPU implementation behind these symbols.
 */
#ifndef UNERF_H
#define UNERF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNERF_OK 0
#define UNERF_ERR_ARG -1     /* bad argument / unsupported shape */
#define UNERF_ERR_HIP -2     /* HIP runtime error (launch, no device) */

const char* unerf_last_error(void);
/* Library/ABI version (major*1000+minor).  A binding built against this header must find exactly UNERF_ABI_VERSION
 * (struct layouts and argument lists change with it; uncertainty-nerf-gs_amd/lib.py::load checks). */
#define UNERF_ABI_VERSION 1420
int unerf_version(void);

/* Spacing function of the proposal sampler's initial sampler, passed behind every (near_plane, far_plane) pair:
 *   UNERF_SPACING_PIECEWISE  UniformLinDispPiecewiseSampler, nerfacto's default (proposal_initial_sampler="piecewise"):
 *                            s(x) = x/2 (x < 1) else 1 - 1/(2x); spacing bin b -> s^-1(b s(far) + (1 - b) s(near))
 *   UNERF_SPACING_UNIFORM    UniformSampler (proposal_initial_sampler="uniform", the reference's few-view configuration,
 *                            /root/reference/README.md:153): identity spacing, b -> b far + (1 - b) near
 * [UPSTREAM nerfstudio 1.1.0 NerfactoModel.populate_modules / ray_samplers.SpacedSampler]. */
#define UNERF_SPACING_PIECEWISE 0
#define UNERF_SPACING_UNIFORM 1

/* RGBRenderer(background_color=config.background_color) at eval (activenerfacto_model.py:98 -> nerfstudio 1.1.0
 * renderers.RGBRenderer.forward / combine_rgb), for the composite and GGN entry points:
 *   UNERF_BG_LAST_SAMPLE  "last_sample" (nerfacto default): comp + rgb[..., -1, :] (1 - accumulation)
 *   UNERF_BG_NONE         "random": at eval the composited colour is returned unblended
 *   UNERF_BG_COLOR        "white" / "black" (or any constant): comp + background_rgb (1 - accumulation);
 *                         background_rgb = 3 HOST floats
 * followed in every case by the eval-mode clamp to [0, 1]. */
#define UNERF_BG_LAST_SAMPLE 0
#define UNERF_BG_NONE 1
#define UNERF_BG_COLOR 2
/* Build switches that change what the operand blobs must look like (bit mask).  UNERF_BUILD_TRUNK_FOLD: the
 * MCDROPOUT split-f16 kernels expect the four trunk-out slabs of mfma16_blob folded (see unerf_field_params). */
#define UNERF_BUILD_TRUNK_FOLD 1
#define UNERF_BUILD_LAP_EXP2 2   /* lap16_blob rows carry their activation's base change (see unerf_field_params) */
int unerf_build_flags(void);
/* Number of visible HIP devices (<=0: none -> every other call fails with UNERF_ERR_HIP). */
int unerf_device_count(void);

/* ------------------------------------------------------------------ rays --
 * Replaces Cameras.generate_rays(camera_indices=0, keep_shape=True) + row-major
 * slicing used by get_outputs_for_camera (scripts/eval_uncertainty.py:1097,1127;
 * models/laplace/laplace_model.py:269-297, 403-415).
 * c2w: HOST pointer to 12 floats (3x4 row-major).  Rays [ray_start, ray_start+count)
 * of the H*W row-major image.  pixel_area may be NULL.
 * distortion: HOST pointer to the camera's 6 OPENCV lens parameters in nerfstudio's order (k1, k2, k3, k4, p1, p2 --
 * camera_utils.get_distortion_params, as the reference's dataparsers hand them to `Cameras`:
 * dataparsers/sparse_mipnerf360/sparse_mipnerf360_dataparser.py:113-125, 248-274; every `ns-process-data images` scene
 * of /root/reference/README.md:52-56 carries them), or NULL.  Non-zero parameters bend the rays the way
 * Cameras._generate_rays_from_coords does for a perspective camera: the image-plane coordinate ((x - cx) / fx,
 * -(y - cy) / fy) of the pixel centre AND of its +1-pixel x / y neighbours (which feed pixel_area) goes through
 * camera_utils.radial_and_tangential_undistort -- UNERF_UNDISTORT_ITERATIONS Newton steps on the 2x2 Jacobian of the
 * forward model, a step taken only where |det| > UNERF_UNDISTORT_EPS -- before it is rotated into the world.  NULL or
 * six zeros: no iteration runs (upstream's `(distortion_params != 0).any()` guard) and the rays are bit-for-bit those
 * of the distortion-free camera. */
#define UNERF_UNDISTORT_ITERATIONS 10
#define UNERF_UNDISTORT_EPS 1e-3f
/* camera_type: nerfstudio's CameraType value of the camera (cameras.py: PERSPECTIVE = 1, FISHEYE = 2, EQUIRECTANGULAR = 3,
 * ORTHOPHOTO = 8), as the reference's dataparsers hand it to `Cameras` (CAMERA_MODEL_TO_TYPE[meta["camera_model"]]:
 * dataparsers/sparse_mipnerf360/sparse_mipnerf360_dataparser.py:237-239, sparse/sparse_nerfstudio_dataparser.py:277-279,
 * nerfonthego, ood_mipnerf360, robustnerf alike).  Camera-frame direction of an (undistorted) image-plane coordinate (u, v)
 * [UPSTREAM-RECALL nerfstudio 1.1.0 Cameras._generate_rays_from_coords]:
 *   PERSPECTIVE      (u, v, -1)
 *   FISHEYE          theta = clip(|(u, v)|, 0, pi):  (u sin(theta) / theta, v sin(theta) / theta, -cos(theta))
 *                    (OPENCV_FISHEYE's k1..k4 act on (u, v) through the same undistortion as the perspective lens)
 *   EQUIRECTANGULAR  theta = -pi u, phi = pi (0.5 - v):  (-sin(theta) sin(phi), cos(phi), -cos(theta) sin(phi));
 *                    distortion parameters are ignored for this type, as upstream does
 *   ORTHOPHOTO       (0, 0, -1) for every pixel, the ORIGIN moves instead: c2w (u, v, 0, 1)
 * then rotated by c2w[:3,:3] and normalised (norm floored at 1e-7); pixel_area from the +1-pixel x / y neighbours as for
 * the perspective camera.  The other CameraType values (omnidirectional stereo, VR180, FISHEYE624) are refused. */
#define UNERF_CAMERA_PERSPECTIVE 1
#define UNERF_CAMERA_FISHEYE 2
#define UNERF_CAMERA_EQUIRECTANGULAR 3
#define UNERF_CAMERA_ORTHOPHOTO 8
int unerf_generate_rays(const float* c2w_host, float fx, float fy, float cx, float cy, const float* distortion_host,
                        int camera_type, int H, int W, int64_t ray_start, int64_t count, float* origins, float* directions,
                        float* pixel_area, void* stream);

/* Oriented crop box (`obb_box` of Model.get_outputs_for_camera / get_outputs_for_camera_unc,
 * models/laplace/laplace_model.py:403-415, models/ensemble/ensemble_pipeline.py:144-157): what
 * Cameras.generate_rays(..., obb_box=box) does to the bundle -- nears / fars from the ray / box slab test
 * (nerfstudio.utils.math.intersect_obb: t clamped to [0,1e10], a miss = 1e10 for both) -- expressed as per-ray
 * first-level spacing bins under the launch-wide planes (near, far):
 *   sbins[r,i] = (b_i s(far_r) + (1-b_i) s(near_r) - s(near)) / (s(far) - s(near)),  b = sbins_row [n+1].
 * Feed sbins [R,n+1] (stride n+1) to unerf_proposal_density / unerf_weights_pdf_resample; every later stage is
 * unchanged.  world_to_box: HOST 12 floats, inverse([R|T]) 3x4 row-major; half_extent: HOST 3 floats (S/2).
 * nears / fars [R] may be NULL.  Rays that miss: planes 1e10 as upstream; their samples collapse onto `far`
 * (upstream they are at infinity, pixels undefined), giving zero accumulation. */
int unerf_ray_box_bins(const float* origins, const float* directions, int64_t R, const float* world_to_box_host,
                       const float* half_extent_host, float near, float far, int spacing, const float* sbins_row, int n,
                       float* sbins, float* nears, float* fars, void* stream);
/* The same fold for a bundle that already carries planes (RayBundle.nears / fars [R], e.g. made by
 * camera.generate_rays(obb_box=...) on the nerfstudio side; SceneCollider.forward keeps planes that are set). */
int unerf_ray_planes_bins(const float* nears, const float* fars, int64_t R, float near, float far, int spacing,
                          const float* sbins_row, int n, float* sbins, void* stream);

/* ------------------------------------------------------------- hash grid --
 * Replaces HashEncoding(implementation="torch").forward called at
 * models/activenerfacto/activenerfacto_field.py:140-147,
 * models/mcdropout/mcdropout_fields.py:115-122, models/laplace/laplace_field.py:129-136.
 * xyz [N,3] in [0,1]; table [L<<log2T, 2]; scalings [L]; out [N, 2L].
 * out_idx (may be NULL) receives the 8 table-row indices per level, [N,L,8] int32,
 * corner order ccc,cfc,ffc,fcc,ccf,cff,fff,fcf (bit-exact bookkeeping check). */
int unerf_hashgrid_fwd(const float* xyz, const float* table, const float* scalings, int64_t N, int L,
                       int log2T, float* out, int32_t* out_idx, void* stream);

/* tiny-cuda-nn `HashGrid` layout, as nerfstudio's HashEncoding(implementation="tcnn") configures it (the
 * reference's default, models/activenerfacto/activenerfacto_field.py:89,146): `table` is then the flat fp32
 * `tcnn_encoding.params` vector viewed as rows of 2 features, and level l is described by one record
 * (host helper: uncertainty-nerf-gs_amd/ops.py::tcnn_grid_levels):
 *   scale  = exp2f(l * log2f(per_level_scale)) * base_res - 1     res = ceilf(scale) + 1
 *   size   = min(next_multiple(res^3, 8), 2^log2_hashmap_size)    rows of this level
 *   offset = sum of the sizes of the levels below                  dense = (res^3 <= size)
 * Lookup: pos = x*scale + 0.5, cell = floor(pos), w = pos - cell; corner k (bit d of k = +1 along dim d) has
 * row (x + y res + z res^2) mod size when dense, else (x ^ y*2654435761 ^ z*805459861) mod size, and weight
 * prod_d (bit ? w_d : 1 - w_d); the 8 corners are accumulated in corner order.  fp32 throughout (tcnn itself
 * interpolates fp16 copies of the same fp32 master parameters: documented divergence). */
typedef struct {
    float scale;
    uint32_t res, offset, size, dense;
} unerf_tcnn_level;

/* xyz [N,3] in [0,1]; params = tcnn_encoding.params (fp32, [rows][2]); levels_host [L]; out [N,2L];
 * out_idx (may be NULL) [N,L,8] int32 = absolute row indices in corner order k = 0..7. */
int unerf_hashgrid_fwd_tcnn(const float* xyz, const float* params, const unerf_tcnn_level* levels_host, int64_t N,
                            int L, float* out, int32_t* out_idx, void* stream);
/* The same lookup the way tiny-cuda-nn computes it when built with TCNN_HALF_PRECISION (its default on every GPU the
 * reference targets; kernel_grid in include/tiny-cuda-nn/encodings/grid.h with T = __half) -- what the reference's
 * HashEncoding(implementation="tcnn") call sites above return: params_half = the parameter vector cast to half
 * ([rows] half2, round to nearest even); positions and the three interpolation weights in fp32 as above; per corner
 * k = 0..7 the fp32 weight product ((w_x w_y) w_z) is rounded to half and result = fma(half weight, row, result) is a
 * HALF fused multiply-add per feature (__hfma2), result starting at 0.  out [N,2L] fp32 holding the half values (rows are
 * those of unerf_hashgrid_fwd_tcnn's out_idx). */
int unerf_hashgrid_fwd_tcnn_half(const float* xyz, const void* params_half, const unerf_tcnn_level* levels_host, int64_t N,
                                 int L, float* out, void* stream);

/* hash grid + small MLP (torch nn.Linear weights pre-transposed to [in][out]). */
typedef struct {
    const float* table;      /* [L<<log2T][2] */
    const float* scalings;   /* [L] */
    int L, log2T;
    const float* w0t;        /* [2L][hidden] */
    const float* b0;         /* [hidden] */
    const float* w1t;        /* [hidden][1] */
    const float* b1;         /* [1] */
    int hidden;              /* 16 (nerfacto proposal nets) or 64; 0 = use_linear=True (HashMLPDensityField's single
                                Linear on the grid features: w1t [2L] and b1 are that layer, w0t / b0 unused) */
    /* Optional dense re-indexing of the first n_dense (coarse) levels, n_dense <= 8 (0 = none).  Level l
       then has (dense_dim[l])^3 cells, dense_dim = scalings[l]+1, stored x-fastest from cell offset
       dense_off[l] in `dense` as float4 = { table[hash(x,y,z)], table[hash(x+1,y,z)] }: the same values the
       hashed lookup returns, but both x-neighbours of a cell edge arrive in ONE 16-byte load (4 instead of 8
       gather instructions per level; the texture-address unit bounds these kernels).  2 <= dense_dim <= 640 (32-bit
       byte offsets); ops.DENSE_LEVEL_BYTES makes copies of the levels up to resolution 128. */
    const float* dense;
    int n_dense;
    int dense_off[8];
    int dense_dim[8];
    /* NULL: nerfstudio torch HashEncoding (above).  Else DEVICE array [L] of level records: `table` is a
       tcnn-layout parameter vector (`scalings`, `log2T`, `dense*` are ignored). */
    const unerf_tcnn_level* tcnn_levels;
    /* use_aabb = 1: spatial_distortion = None (disable_scene_contraction, mcdropout_models.py:60-63): positions are
       normalised with SceneBox.get_normalized_positions, (x - aabb[0..2]) / (aabb[3..5] - aabb[0..2]), instead of
       SceneContraction(inf) followed by (x + 2) / 4.  aabb = {min xyz, max xyz}. */
    int use_aabb;
    float aabb[6];
    /* tcnn_levels only: 1 = `table` points at the HALF copy of the parameter vector ([rows] half2, 4 bytes per row) and
       the level is interpolated in tiny-cuda-nn's own half arithmetic (see unerf_hashgrid_fwd_tcnn_half); the features
       then enter the fp32 MLP as the half values they are.  0: fp32 rows, fp32 blend. */
    int grid_half;
} unerf_density_net;

/* -------------------------------------------------- proposal density --
 * Replaces HashMLPDensityField.density_fn as invoked by ProposalNetworkSampler
 * (call site models/activenerfacto/activenerfacto_model.py:89,
 * models/laplace/laplace_model.py:210,459).  Sample i of ray r sits at the mid-point of
 * spacing bins [i, i+1] (converted to euclidean with the spacing fn `spacing` = UNERF_SPACING_*
 * and near/far).  sbins: [R, n+1] with row stride `sbins_stride`
 * (0 = one shared row, the initial uniform bins).  density_out [R,n]. */
int unerf_proposal_density(const float* origins, const float* directions, const float* sbins,
                           int64_t sbins_stride, int64_t R, int n, float near_plane, float far_plane, int spacing,
                           const unerf_density_net* net /* host struct of device ptrs */,
                           float average_init_density, float* density_out,
                           int64_t ray_offset, int image_width /* scheduling hint, 0 = none: rays [ray_offset,
                           ray_offset + R) are consecutive pixels of a row-major image this wide; a wave then
                           evaluates an 8x8 pixel patch at one sample index.  Same results either way. */,
                           void* stream);

/* ------------------------------------------- weights + PDF resampling --
 * Replaces RaySamples.get_weights + PDFSampler.generate_ray_samples (eval branch) inside
 * ProposalNetworkSampler, and DepthRenderer("median") for prop_depth_i
 * (models/activenerfacto/activenerfacto_model.py:150-151).
 * u: [m+1] = linspace(0,1-1/(m+1),m+1)+1/(2(m+1)) (host computes it the torch way).
 * sbins_out [R,m+1]; prop_depth_out [R] (may be NULL); weights_out [R,n] (may be NULL).
 * clip_minmax (may be NULL): [ceil(R_total/chunk_rays)][2] floats, pre-set to {+inf,0};
 * receives per-chunk min/max of the NEW samples' mid-points (the bounds
 * DepthRenderer("expected") clips to); ray_offset = index of ray 0 inside the frame. */
int unerf_weights_pdf_resample(const float* density, const float* sbins, int64_t sbins_stride, int64_t R,
                               int n, float near_plane, float far_plane, int spacing, const float* u, int m,
                               float histogram_padding, float eps, float* sbins_out, float* prop_depth_out,
                               float* weights_out, float* clip_minmax, int64_t ray_offset,
                               int64_t chunk_rays, void* stream);

/* ------------------------------------------------------- main field --
 * Replaces, per mode:
 *  ACTIVE    ActiveNerfactoField.get_density + NerfactoField.get_outputs
 *            (models/activenerfacto/activenerfacto_field.py:162-215)
 *  MCDROPOUT NerfactoMCDropoutField.get_density + create_mlp heads, K stochastic passes
 *            (models/mcdropout/mcdropout_fields.py:110-174, mcdropout_models.py:116-119,
 *            utils.py:6-43); K=0 means dropout off, one pass.
 *  LAPLACE   NerfactoLaplaceField.forward_unc with is_inference=True
 *            (models/laplace/laplace_field.py:279-362, 365-485, 528-568): ws_* are the
 *            n_lap sampled last-layer parameter rows mu+randn*std ([out,in] row-major, bias).
 */
#define UNERF_FIELD_ACTIVE 0
#define UNERF_FIELD_MCDROPOUT 1
#define UNERF_FIELD_LAPLACE 2

typedef struct {
    int mode;
    /* hash grid (L=16 levels, F=2) */
    const float* table; const float* scalings; int L, log2T;
    /* trunk: w0t [32][64], b0[64]; w1t [64][out1], b1[out1]
       out1 = 17 ACTIVE (density, geo15, beta) | 16 MCDROPOUT | 15 LAPLACE (mlp_hidden) */
    const float* w0t; const float* b0; const float* w1t; const float* b1; int out1;
    /* colour head with the constant eval appearance embedding folded into the first bias:
       h0t [31][64] (rows: SH16 then geo15), hb0[64]; h1t [64][64], hb1[64]; h2t [64][3], hb2[3] */
    const float* h0t; const float* hb0; const float* h1t; const float* hb1; const float* h2t; const float* hb2;
    float average_init_density, beta_min;
    int sh_remap;            /* 0: SH on (d+1)/2 as the torch SHEncoding does; 1: tcnn ([-1,1]) */
    /* MCDROPOUT */
    int K; uint32_t seed; float p_drop;
    /* LAPLACE */
    const float* ws_density;  /* [n_lap][65]  */
    const float* ws_rgb;      /* [n_lap_rgb (n_lap when that is 0)][195] */
    int n_lap;
    int lap_mask_density;     /* 1: multiply the density mean by the box selector (the use_deterministic_density=True
                                 path, laplace_field.py:501-506 / :317-345, fed with n_lap copies of the mean row) */
    /* Optional (ACTIVE / MCDROPOUT): the same MLP weights pre-arranged as fp32-MFMA A-operand
       fragments (layout: uncertainty-nerf-gs_amd/ops.py::pack_field_mfma, UNERF_MFMA_BLOB_FLOATS
       floats).  When non-NULL the fused kernel runs its dense layers on v_mfma_f32_32x32x2_f32
       (exact fp32) with the weights resident in LDS; NULL selects the VALU kernel. */
    const float* mfma_blob;
    /* Optional (LAPLACE, with mfma_blob): the n_lap <= 128 sampled last-layer rows of both heads as MFMA A
       fragments (ops.py::pack_laplace_heads, UNERF_LAP_BLOB_FLOATS floats; streamed from L2, not LDS). */
    const float* lap_blob;
    /* NULL: nerfstudio torch HashEncoding.  Else DEVICE array [16] of tcnn level records: `table` is the flat
       tcnn-layout parameter vector of the main grid (`scalings` / `log2T` ignored). */
    const unerf_tcnn_level* tcnn_levels;
    /* Optional, preferred when present (all modes): the dense layers as SPLIT-F16 matrix operands
       (ops.py::pack_field_mfma16, same size as mfma_blob; LAPLACE additionally lap16_blob,
       ops.py::pack_laplace_heads16, UNERF_LAP_BLOB_FLOATS floats).  Every fp32 weight and activation is carried
       as hi = f16(x), lo = f16(x - hi) and hi*hi + hi*lo + lo*hi is accumulated in fp32 on
       v_mfma_f32_32x32x16_f16: fp32-equivalent results (relative deviation ~1e-7 from the exact kernels) at a
       third of the matrix-pipe time of the fp32-input MFMA, which runs at the vector rate.  NULL selects the
       exact-fp32 kernels above.  With UNERF_BUILD_TRUNK_FOLD (unerf_build_flags) MCDROPOUT expects the four
       trunk-out slabs folded: their second operand holds rows 0..15 = W_hi and rows 16..31 = W_lo, so the 16-row
        layer takes two MFMAs per k-step instead of three (ops.pack_field_mfma16(fold_trunk=True)).  With
       UNERF_BUILD_LAP_EXP2 lap16_blob's rows (weights and bias) carry the base change of their activation: density
       rows x log2(e) (unscaled when lap_softplus = 1), colour rows x -log2(e), and the kernel applies the bare hardware
       exp2 (padded rows: bias -1e30 for the density head, +1e30 for the colour heads -> they contribute 0). */
    const float* mfma16_blob;
    const float* lap16_blob;
    /* Scheduling hint, 0 = none: the rays [ray_offset, ray_offset + R) of this call are consecutive pixels of a
       row-major image `image_width` wide.  The matrix kernels then take 8x4 pixel patches (instead of 32
       consecutive pixels) as the 32 columns of a tile: fewer distinct cache lines per gather instruction.
       Results are identical with and without the hint. */
    int image_width;
    /* Output layout of the ACTIVE / MCDROPOUT matrix kernels.  0: density [B,R,S], rgb [B,R,S,3], aux [R,S] (the
       RaySamples layout).  1: sample-major planes density [B,S,R], rgb [B,S,3,R], aux [S,R]: a kernel tile is 32 rays
       at one sample slot, so its stores then fill whole 32-byte sectors instead of 4 bytes per cache line; consumed
       by unerf_composite_var_planes / unerf_composite_moments_planes. */
    int sample_major;
    /* MCDROPOUT: where the Dropout modules sit (mcdropout_fields.py:112-144 via create_mlp, utils.py:6-43).  0 = the
       reference default UNERF_DROP_TRUNK | UNERF_DROP_HEAD1 (density_dropout_layers=True, rgb_dropout_layers=[-1]).
       UNERF_DROP_TRUNK: after the trunk's hidden ReLU (density_dropout_layers); UNERF_DROP_HEAD0: in front of the
       colour head's Linear 1 (rgb_dropout_layers contains 1); UNERF_DROP_HEAD1: in front of its last Linear (contains
       2 or -1).  Mask streams 0 / 2 / 1 of the counter RNG.  With mfma16_blob the scale 1/(1-p) is expected folded
       into the layer behind each active site (ops.pack_field_mfma16).
       UNERF_DROP_HEADIN: in front of the colour head's Linear 0 (rgb_dropout_layers contains 0), i.e. on its 63 INPUTS
       [SH16 | geo15 | appearance32] -- mask stream 3.  The appearance block can then no longer ride in the bias, so this
       site needs h0_full_t / hb0_raw / app_embed below, and it is served by the VALU kernel only (any precision setting;
       roughly 10 x the time of the matrix kernels -- a rarely used configuration kept correct rather than fast). */
    int drop_sites;
    /* LAPLACE: 1 = density_activation "softplus" (laplace_model.py:151, laplace_field.py:323) instead of trunc_exp on
       the (sampled) density head (unerf_laplace_ggn_diag: dsigma/dpre = 1 - exp(-sigma) instead of sigma). */
    int lap_softplus;
    /* as unerf_density_net: 1 = normalise positions with the scene box instead of the contraction */
    int use_aabb;
    float aabb[6];
    /* 1 (with mfma16_blob / lap16_blob): REFERENCE-PRECISION dense layers -- one f16 product per MAC, fp32 accumulation:
       every operand is rounded to f16 once (the hi halves of the blobs, v_cvt_pk_f16_f32 of the activations) and the lo
       halves are not used.  That is the arithmetic of the Linear layers under torch.autocast(float16), which the
       reference forces at eval (models/mcdropout/mcdropout_models.py:86-92), and no narrower than tiny-cuda-nn's
       FullyFusedMLP (fp16 weights, activations and accumulators; the reference's default implementation="tcnn",
       models/activenerfacto/activenerfacto_field.py:89).  0: the split form above (fp32-equivalent).  Biases, the
       64 -> 3 colour layer and every activation function stay fp32 in both forms. */
    int f16_single;
    /* With mfma16_blob, DEVICE int32 (may be NULL): |= 1 when an f16 OPERAND of the dense layers overflowed (an activation
       beyond 65504).  f16_single: an output pre-activation (density logit, colour sums; LAPLACE: a sampled-head mean)
       comes out inf / NaN -- the trace such an activation leaves in this form, which has no lo halves to turn it into a
       NaN sample.  Split form: the pre-activations of a colour layer are NaN (hi = inf, lo = -inf make every unit of the
       next layer NaN, which the ReLU behind it could otherwise zero; trunk overflows reach the density as NaN and are
       flagged by the composite entry points).  Same word and same host protocol as the composite entry points'
       nonfinite_flag (render.OverflowGuard: fp32 re-render of the launch group). */
    int32_t* overflow_flag;
    /* UNERF_DROP_HEADIN only (else may be NULL): the colour head's first layer unfolded -- weights transposed
       [63][64] over the inputs [SH16 | geo15 | appearance32], its bias [64] WITHOUT the appearance term, and the eval
       appearance embedding [32] (zeros or the mean embedding, [UPSTREAM NerfactoField.get_outputs]). */
    const float* h0_full_t;
    const float* hb0_raw;
    const float* app_embed;
    /* LAPLACE: the reference draws a fresh set of last-layer samples in EVERY eval chunk of a frame -- sample_laplace
       runs inside get_outputs_unc, which get_outputs_for_camera_ray_bundle_unc calls per 32,768-ray chunk
       (models/laplace/laplace_model.py:432-443 -> laplace_field.py:331-339, 468-476, 545).  lap_chunk_rays > 0:
       ws_density [lap_sets][n_lap][65], ws_rgb [lap_sets][n_lap][195], lap_blob / lap16_blob
       [lap_sets][UNERF_LAP_BLOB_FLOATS] are STACKS of sets and ray g = ray_offset + r is evaluated with set
       g / lap_chunk_rays (every ray of the call must map below lap_sets).  lap_chunk_rays and ray_offset must be
       multiples of 32: the matrix kernels then take 32 consecutive rays as a tile (image_width is not used), so no tile
       straddles two sets.  0: one set for every ray (lap_sets ignored). */
    int lap_chunk_rays;
    int lap_sets;
    /* LAPLACE: rows of ws_rgb (and of the colour heads in the blobs) when it differs from n_lap; <= 0: n_lap.  The
       reference's forward_unc does not pass n_samples on to the colour head (laplace_field.py:516-520), which therefore
       always draws its default 100 whatever the density head was asked for. */
    int n_lap_rgb;
    /* ACTIVE / MCDROPOUT, ray-major layout: 1 = the outputs leave as ONE 16-byte row (sigma, r, g, b) per (pass, ray,
       sample): `rgb` is [B,R,S,4] and `density` is not written (may be NULL).  A kernel tile is 32 rays at one sample
       slot, so every store of the [B,R,S] + [B,R,S,3] layout is a lone 4-byte write into its own cache line; the packed
       row is one dwordx4 store, and unerf_composite_var / unerf_composite_moments read it back with one 16-byte load
       (pass density = NULL and the packed rows as rgb).  Same values either way. */
    int packed_out;
    /* tcnn_levels only: 1 = `table` points at the HALF copy of the main grid's parameter vector ([rows] half2) and the
       lookup runs in tiny-cuda-nn's own half arithmetic (unerf_hashgrid_fwd_tcnn_half) -- what
       HashEncoding(implementation="tcnn"), the reference's default (models/activenerfacto/activenerfacto_field.py:89,
       140-147; mcdropout_fields.py:78, 115-122; laplace_field.py:91, 129-136), hands to the MLP.  The f16 matrix kernels
       take the packed half features as their layer-0 operands without a conversion; 4 bytes per gathered corner
       instead of 8.  0: fp32 rows and blend (a tcnn built without TCNN_HALF_PRECISION). */
    int grid_half;
    /* Network widths, 0 = nerfacto's (hidden 64, hidden_color 64, geo_dim 15, feat_per_level 2, app_dim 32).  The reference
       forwards hidden_dim, hidden_dim_color, features_per_level and appearance_embed_dim from its model configs to the
       field (models/activenerfacto/activenerfacto_model.py:63-77, models/mcdropout/mcdropout_models.py:66-80,
       models/laplace/laplace_model.py:169-186); the field classes also take geo_feat_dim.  Any other combination (and
       L != 16) runs the ANY-WIDTH kernel: one lane per sample, every width a run-time loop -- correct, 10-20 x slower than
       the matrix kernels, plain ray-major outputs only (no packed_out / sample_major / feature planes), features_per_level
       4 on the torch-layout grid only (table rows of 4 floats).  Shapes then: w0t [L F][hidden], w1t [hidden][out1] with
       out1 = geo_dim + 2 (ACTIVE) / + 1 (MCDROPOUT) / + 0 (LAPLACE), h0t [16 + geo_dim][hidden_color], h1t
       [hidden_color][hidden_color], h2t [hidden_color][3], ws_density [..][hidden + 1], ws_rgb [..][3 hidden_color + 3],
       h0_full_t [16 + geo_dim + app_dim][hidden_color], app_embed [app_dim].  A dropout site has at most 128 units. */
    int hidden, hidden_color, geo_dim, feat_per_level, app_dim;
} unerf_field_params;
#define UNERF_DROP_TRUNK 1
#define UNERF_DROP_HEAD0 2
#define UNERF_DROP_HEAD1 4
#define UNERF_DROP_HEADIN 8
#define UNERF_MFMA_BLOB_FLOATS 10660
#define UNERF_MFMA16_BLOB_FLOATS 11684   /* mfma16_blob: + the 64 -> 3 colour layer as four f16 operand slabs (f16_single) */
#define UNERF_LAP_BLOB_FLOATS 33280

/* outputs: B = max(K,1) passes
 *   density [B,R,S]; rgb [B,R,S,3];
 *   aux     ACTIVE: beta [R,S] | LAPLACE: density_var [R,S] | else NULL
 *   aux2    LAPLACE: rgb_var [R,S] (relu, channel mean) | else NULL
 * sbins [R,S+1] spacing-domain bins of the final samples; ray_offset keys the dropout RNG.
 * near_plane < 0: sbins holds EUCLIDEAN bin edges instead (the starts / last end of a RaySamples made by the
 * caller's own sampler -- Field.forward(ray_samples)); far_plane is ignored then.  Not with `features`. */
int unerf_field_fwd(const float* origins, const float* directions, const float* sbins, int64_t R, int S,
                    float near_plane, float far_plane, int spacing, int64_t ray_offset,
                    const unerf_field_params* p /* host struct */,
                    const float* features /* NULL, or the planes written by unerf_field_gather */,
                    float* density, float* rgb, float* aux, float* aux2, void* stream);

/* Level-major form of the same HashEncoding lookup for the main field: writes the features of the
 * R*S final samples as planes [L][R*S][2] (level, sample, feature).  Each level table is 4 MiB --
 * one XCD's L2 -- so sweeping level by level runs the gathers out of L2 instead of the Infinity
 * Cache; unerf_field_fwd(features=...) then skips its own lookup.  Same values bit for bit. */
int unerf_field_gather(const float* origins, const float* directions, const float* sbins, int64_t R, int S,
                       float near_plane, float far_plane, int spacing, const float* table, const float* scalings, int L,
                       int log2T, float* feature_planes, void* stream);

/* Laplace depth path: models/laplace/laplace_model.py:486-507.  mean over D draws of
 * get_weights(relu(mu + max(sqrt(var),1e-10) * eps)).  noise [D,R,S] or NULL (then the
 * built-in counter RNG with `seed`).  weights_out [R,S]. */
int unerf_laplace_depth_weights(const float* density_mu, const float* density_var, const float* sbins,
                                int64_t R, int S, float near_plane, float far_plane, int spacing, const float* noise, int D,
                                uint32_t seed, int64_t ray_offset, float* weights_out, void* stream);

/* Laplace GGN fitting: NerfactoLaplaceModel.compute_hessian_naive (models/laplace/laplace_model.py:343-400),
 * one batch of rays.  Adds (+=) the batch's diagonal generalised Gauss-Newton of the summed-MSE loss w.r.t.
 * mlp_density (ggn_density[65]: weight[64], bias) and mlp_rgb_ll (ggn_rgb[195]: weight[3][64] row-major,
 * bias[3]) -- the layout of field.mlp_density_ggn / field.mlp_rgb_ggn (laplace_field.py:231-238).
 * Forward = the deterministic is_inference=False path (laplace_field.py:317-345, 462-465) rendered with the
 * eval-mode RGB renderer.  p: mode LAPLACE with mfma_blob; p->ws_density[65] / p->ws_rgb[195] hold the MEAN
 * last layers ([out,in] row-major, then bias); n_lap / lap_blob are ignored.  sbins [R,S+1] as for
 * unerf_field_fwd, S <= 64.  workspace: unerf_laplace_ggn_workspace_bytes(R, S) bytes of device scratch. */
size_t unerf_laplace_ggn_workspace_bytes(int64_t R, int S);
int unerf_laplace_ggn_diag(const float* origins, const float* directions, const float* sbins, int64_t R, int S,
                           float near_plane, float far_plane, int spacing, const unerf_field_params* p /* host struct */,
                           int background, const float* background_rgb_host /* UNERF_BG_*: the renderer the loss sees */,
                           void* workspace, size_t workspace_bytes, float* ggn_density, float* ggn_rgb,
                           void* stream);

/* ------------------------------------------------- composite + variance --
 * Replaces RaySamples.get_weights + RGB/Accumulation/Depth(median,expected)/Uncertainty
 * renderers + the depth-variance sum at models/activenerfacto/activenerfacto_model.py:94-112
 * and models/laplace/laplace_model.py:471-521.
 * density [B,R,S], rgb [B,R,S,3], beta [B? no: R,S] (NULL -> rgb_var = 0);
 * weights_alt [R,S] (NULL or the laplace mean sampled weights: then accumulation, depth,
 * expected depth and depth_var use it while rgb and rgb_var use get_weights(density)).
 * clip_minmax as produced by unerf_weights_pdf_resample.  1 <= S <= 256, any value.
 * background / background_rgb_host: UNERF_BG_* (above).
 * nonfinite_flag (DEVICE int32, may be NULL): |= 1 (atomically, once per offending wave) when a density or colour
 * read by this call is NaN.  The renderers turn NaN into 0 (nan_to_num, as upstream does), which would hide the one
 * failure mode of the f16 matrix kernels -- an operand at or beyond 65504 (hi = inf, lo = -inf -> NaN); the host
 * checks the word once per frame and re-renders the launch group with the exact-fp32 kernels (render.OverflowGuard).
 * out [B,R,8] = rgb(3), accumulation, depth(median), expected_depth, rgb_var, depth_var(+1e-5). */
int unerf_composite_var(const float* density, const float* rgb, const float* beta, const float* weights_alt,
                        const float* sbins, int B, int64_t R, int S, float near_plane, float far_plane, int spacing,
                        const float* clip_minmax, int64_t ray_offset, int64_t chunk_rays, int background,
                        const float* background_rgb_host, int32_t* nonfinite_flag, float* out, void* stream);

/* Fused K-pass form of the two calls above/below for MC-dropout: composites the B <= 16 passes of every
 * ray and reduces them in registers.  mean_out / var_out [R,8] over the passes of
 * rgb(3), accumulation, depth, expected_depth, rgb_var, depth_var (var unbiased, B-1). */
int unerf_composite_moments(const float* density, const float* rgb, const float* sbins, int B, int64_t R, int S,
                            float near_plane, float far_plane, int spacing, const float* clip_minmax, int64_t ray_offset,
                            int64_t chunk_rays, int background, const float* background_rgb_host, int32_t* nonfinite_flag,
                            float* mean_out, float* var_out, void* stream);

/* The same two reductions over the sample-major planes unerf_field_fwd writes with sample_major = 1
 * (density [B,S,R], rgb [B,S,3,R], beta [S,R] or NULL): one lane per ray walks the samples front to back, so
 * get_weights is the sequential recurrence of activenerfacto_model.py:94 / RaySamples.get_weights, and the K passes
 * of mcdropout_models.py:116-126 are reduced to mean / unbiased variance in the same thread.
 * out [B,R,8]; mean_out / var_out [R,8] (B >= 2); channel order as unerf_composite_var. */
int unerf_composite_var_planes(const float* density, const float* rgb, const float* beta, const float* sbins, int B,
                               int64_t R, int S, float near_plane, float far_plane, int spacing, const float* clip_minmax,
                               int64_t ray_offset, int64_t chunk_rays, int background, const float* background_rgb_host,
                               int32_t* nonfinite_flag, float* out, void* stream);
int unerf_composite_moments_planes(const float* density, const float* rgb, const float* sbins, int B, int64_t R, int S,
                                   float near_plane, float far_plane, int spacing, const float* clip_minmax,
                                   int64_t ray_offset, int64_t chunk_rays, int background,
                                   const float* background_rgb_host, int32_t* nonfinite_flag, float* mean_out,
                                   float* var_out, void* stream);

/* ------------------------------------------------------ moments over K --
 * Replaces torch.stack(...).mean(0) / .std(0) / .var(0) over MC passes
 * (models/mcdropout/mcdropout_models.py:121-126) and over ensemble members
 * (models/ensemble/ensemble_pipeline.py:159-189).  x [K,N,C] -> mean [N,C],
 * var [N,C] (unbiased, K-1; NULL ok).  Two-pass in fp32 like torch. */
int unerf_moments(const float* x, int K, int64_t N, int C, float* mean, float* var, void* stream);

/* ================================================================ splats ==
 * gsplat 0.1.11 call sites in models/activesplatfacto/activesplatfacto_model.py. */

/* :221-234 project_gaussians(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, H, W, 16).
 * viewmat_host: 12 or 16 floats row-major (first 3 rows used).  Outputs as gsplat:
 * xys[N,2], depths[N], radii[N] i32, conics[N,3], compensation[N], num_tiles_hit[N] i32, cov3d[N,6]. */
int unerf_splat_project(const float* means3d, const float* scales, float glob_scale, const float* quats,
                        const float* viewmat_host, float fx, float fy, float cx, float cy, int H, int W,
                        int block_width, float clip_thresh, int64_t N, float* xys, float* depths, int32_t* radii,
                        float* conics, float* compensation, int32_t* num_tiles_hit, float* cov3d, void* stream);

/* The same projection fed with the model's parameters as stored: log_scales [N,3] (gauss_params.scales) and raw_quats
 * [N,4] (gauss_params.quats, unnormalised).  Replaces torch.exp(scales_crop) and quats_crop / quats_crop.norm(dim=-1,
 * keepdim=True) (:221-223) + project_gaussians: three elementwise launches less per frame.
 * opacity_logits [N] (may be NULL): gauss_params.opacities.  When given, opacities_out [N] receives
 * sigmoid(opacity_logits) (:256), times the compensation when antialiased != 0 (:252-254), and num_tiles_hit becomes
 * TIGHT: it counts only the tiles of gsplat's box in which some pixel centre can have alpha >= 1/255 for that opacity
 * (the ellipse sigma <= ln(255 o); about half of the box for the bench scene).  The pairs left out are pairs the blend
 * loop skips by itself, so every rasterised output is bit-identical -- with shorter lists to sort and stage.  A tight
 * count must be binned by unerf_splat_bin_sort with tight_conics / tight_opacities = these conics / opacities_out.
 * radii, xys, ... stay gsplat's (radii > 0 is still "visible"). */
int unerf_splat_project_raw(const float* means3d, const float* log_scales, float glob_scale, const float* raw_quats,
                            const float* viewmat_host, float fx, float fy, float cx, float cy, int H, int W,
                            int block_width, float clip_thresh, int64_t N, const float* opacity_logits, int antialiased,
                            float* opacities_out, float* xys, float* depths, int32_t* radii, float* conics,
                            float* compensation, int32_t* num_tiles_hit, float* cov3d, void* stream);

/* :245-246 spherical_harmonics(degree, viewdirs, coeffs[N,16,3]) then clamp(+0.5, min 0);
 * and :286 softplus(log_unc)+beta_min.  cam_pos_host: 3 floats.  colors_out [N,3], beta_out [N].
 * degree = -1: the config.sh_degree == 0 branch (:247-248), colors = sigmoid(DC coefficients), no SH, no +0.5.
 * sh_coeffs must be 16-byte aligned (rows of 48 floats are read as 16-byte loads). */
int unerf_splat_sh_colors(int degree, const float* means3d, const float* cam_pos_host, const float* sh_coeffs,
                          const float* log_unc, float beta_min, int64_t N, float* colors_out, float* beta_out,
                          void* stream);

/* The same with the coefficients as the model stores them -- gauss_params.features_dc [N,3] and
 * gauss_params.features_rest [N,15,3] -- i.e. without the torch.cat of activesplatfacto_model.py:242-243
 * (192 B per splat written and read back every frame).  features_rest may be NULL for degree 0. */
int unerf_splat_sh_colors_split(int degree, const float* means3d, const float* cam_pos_host, const float* features_dc,
                                const float* features_rest, const float* log_unc, float beta_min, int64_t N,
                                float* colors_out, float* beta_out, void* stream);

/* Everything the rasteriser reads per splat, in one launch and in its final layout: SH colours (as
 * unerf_splat_sh_colors_split), beta = softplus(log_unc) + beta_min (:286), the depth channel, and
 * opacities = sigmoid(opacity_logits) [* compensation] (:252-256).  Replaces the torch.cat([rgbs, beta, depths]) and
 * torch.sigmoid launches of the frame.  rows_out [N,C]: C = 5 -> [r, g, b, beta, depth]; C = 4 -> [r, g, b, depth]
 * (plain splatfacto; log_unc may be NULL).  compensation may be NULL ("classic").  opacities_out [N].
 * opacity_logits may be NULL (the opacities were made by unerf_splat_project_raw): opacities_out is then not written. */
int unerf_splat_shade_inputs(int degree, const float* means3d, const float* cam_pos_host, const float* features_dc,
                             const float* features_rest, const float* log_unc, float beta_min,
                             const float* opacity_logits, const float* compensation, const float* depths, int64_t N,
                             int C, float* rows_out, float* opacities_out, void* stream);

/* bin-and-sort done ONCE per frame (the reference repeats it inside each of its four
 * rasterize_gaussians calls :260,:289,:306,:343).  cum_tiles_hit [N] i32 (inclusive scan,
 * output), isect_ids [I] i64 + gaussian_ids [I] i32 sorted outputs, tile_bins [tiles,2] i32.
 * Two-step use: call unerf_splat_count_intersects (async; writes the scan and *num_intersects
 * on device), read it back, size the buffers, then unerf_splat_bin_sort.
 * The sorted order is that of gsplat's stable sort of (tile << 32 | depth bits) keys; it is produced by
 * ordering the N splats by depth first and then sorting the intersections by tile id only (stable).
 * isect_ids_sorted may be NULL (the rasteriser needs only gaussian_ids_sorted and tile_bins).
 * tight_conics [N,3] + tight_opacities [N] (both or neither): cum_tiles_hit is the scan of a TIGHT num_tiles_hit
 * (unerf_splat_project_raw with opacity logits) -- each splat then emits exactly the tiles that count was made of.
 * NULL, NULL: gsplat's lists (every tile of the radius box). */
int64_t unerf_splat_sort_workspace_bytes(int64_t N, int64_t num_intersects);
int unerf_splat_count_intersects(const int32_t* num_tiles_hit, int64_t N, int32_t* cum_tiles_hit,
                                 void* workspace, int64_t workspace_bytes, void* stream);
int unerf_splat_bin_sort(const float* xys, const float* depths, const int32_t* radii,
                         const int32_t* cum_tiles_hit, int64_t N, int64_t num_intersects, int H, int W,
                         int block_width, const float* tight_conics, const float* tight_opacities,
                         int64_t* isect_ids_sorted, int32_t* gaussian_ids_sorted,
                         int32_t* tile_bins, void* workspace, int64_t workspace_bytes, void* stream);

/* rasterize_forward / nd_rasterize_forward: C channels blended in one pass.
 * colors [N,C] (C<=8), opacities [N], background [C] DEVICE (NULL = zeros),
 * out_img [H,W,C], final_T [H,W], final_idx [H,W] i32 (may be NULL): index (into gaussian_ids_sorted) of the last splat
 * blended into the pixel, 0 when there is none.
 * stop_idx [H,W] i32 (may be NULL): the final_idx of an earlier pass over the SAME ids / bins / xys / conics / opacities
 * (the reference's depth-variance pass, :343-356, after its rgb pass): each pixel then stops behind that index instead
 * of re-deriving its end from the transmittance -- same blended terms, same result.
 * flags: 0, or UNERF_RASTER_NO_CULL to walk every staged splat in every wave (gsplat's schedule).  By default a wave (a
 * 4-row strip of the 16 x 16 tile) walks only the splats whose alpha >= 1/255 ellipse can reach its strip; the skipped
 * (pixel, splat) pairs are pairs the blend loop would have skipped itself, so both settings give identical bits.
 * chan_max (may be NULL): 1 DEVICE float the caller zeroed; on return it holds max(chan_max, max over the image of
 * out_img[..., max_channel]) -- the `img.max()` that unerf_splat_alpha_normalize(max_ready = 1) then needs not
 * compute.  Values of that channel must be >= 0 (depths, squared differences). */
#define UNERF_RASTER_NO_CULL 1
int unerf_splat_rasterize(const int32_t* gaussian_ids_sorted, const int32_t* tile_bins, const float* xys,
                          const float* conics, const float* colors, const float* opacities,
                          const float* background, int C, int H, int W, int block_width, const int32_t* stop_idx,
                          int flags, int max_channel, float* chan_max, float* out_img, float* final_T,
                          int32_t* final_idx, void* stream);

/* :319 / :356  img = where(alpha>0, img/alpha, max(img)) with alpha = 1-final_T, applied IN PLACE
 * to channel `ch` of an interleaved image [HW, stride]; scratch_max = 1 device float.
 * max_ready = 0: max(img[..., ch]) is computed here into scratch_max;  1: scratch_max already holds it (the chan_max
 * of the unerf_splat_rasterize call that produced img). */
int unerf_splat_alpha_normalize(float* img, int stride, int ch, const float* final_T, int64_t HW,
                                float* scratch_max, int max_ready, void* stream);

/* The frame's per-pixel epilogue in ONE pass over the image (activesplatfacto_model.py:275 rgb clamp, :319 / :356 depth
 * normalisation, :359-367 accumulation / rgb_var / depth_std -- torch calls in the reference).  Channel `ch` of img [HW, stride]
 * is normalised IN PLACE exactly as unerf_splat_alpha_normalize(max_ready = 1) does (scratch_max: the 1 device float that holds
 * max(img[..., ch]), left by the unerf_splat_rasterize call that produced img), and any of these is written (NULL: skipped):
 *   rgb_out  [HW,3] = min(img[..., 0:3], 1)   torch.clamp(max=1.0): NaN stays NaN; needs ch >= 3
 *   acc_out  [HW]   = 1 - final_T
 *   sq_out   [HW]   = img[..., sq_ch]^2       (rgb_var = uncertainty^2; sq_ch != ch)
 *   sqrt_out [HW]   = sqrt(the normalised channel ch)   (depth_std = sqrt(depth_var)) */
int unerf_splat_normalize_outputs(float* img, int stride, int ch, const float* final_T, int64_t HW,
                                  const float* scratch_max, float* rgb_out, float* acc_out, int sq_ch, float* sq_out,
                                  float* sqrt_out, void* stream);
/* :325-341 per-splat (z_i - depth[floor(xy_i)])^2, bounds test with the reference's strict ">0";
 * depth image = channel `ch` of [H,W,stride].  sq_diff_out [N]. */
int unerf_splat_depth_sqdiff(const float* xys, const float* depths, const float* depth_img, int stride, int ch,
                             int H, int W, int64_t N, float* sq_diff_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNERF_H */
