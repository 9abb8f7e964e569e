"""ctypes binding of libunerf (include/unerf.h).

The library is built in-tree by ``build_library()`` (called from ``__graft_entry__.build()``)
with ``hipcc --offload-arch=gfx950``.  There is no Python/CPU fallback: if the shared
object is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
# UNERF_LIB: another build of the same ABI, for A/B timing on one box (benchmarks/ab_bench.sh); unset in normal use
LIB_PATH = os.environ.get("UNERF_LIB") or os.path.join(CSRC, "libunerf.so")
SOURCES = ["unerf_nerf.hip", "unerf_splat.hip"]
# -amdgpu-mfma-vgpr-form: gfx950 has one unified register file; let the MFMAs write their accumulators to
# ordinary VGPRs so ReLU / dropout / the next layer's B operand read them without v_accvgpr_read copies
# (97 copies per tile in the K-pass kernel, which is VALU-issue-bound; rocprof r1_04).
# -fno-slp-vectorize: the one KNOWN trigger of run-to-run differences in the split-f16 field kernels is code the SLP
# vectoriser makes (packed v_pk_mul_f32 forms of the position x scale products in front of the hash): the fused-blend
# experiment build (UNERF_FIELD_BLEND_FMA=1) differs from launch to launch with it and repeats bit for bit without it
# (profiles/r5_exp_blend_defect.jsonl).  The shipped sources never showed the defect, but the flag costs
# nothing (same frame times, same bits: every hot loop is written on explicit 2-vectors) and removes the trigger class.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
               "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]
# The compiler the kernels' hazard placement (hand-placed wait states inside inline assembly, the internal
# -amdgpu-mfma-vgpr-form switch) was validated with: tests/test_gpu_repeatability.py on MI355X.  Another version builds,
# with a warning -- rerun that test before trusting it.
VALIDATED_HIPCC = "HIP version: 7.2"


class UnerfError(RuntimeError):
    pass


def _source_digest() -> str:
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for f in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "unerf_common.hpp"),
                                                          os.path.join(INCLUDE, "unerf.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libunerf.so (cross-compiles without a GPU).
    Up-to-date-ness is a content hash of sources + flags (mtimes do not survive being copied to the GPU
    box); concurrent callers (one process per GPU) serialise on a lock file and the library is
    replaced atomically, so nobody dlopens a half-written file."""
    import fcntl
    if os.environ.get("UNERF_LIB"):
        # an explicitly chosen build (A/B timing): never rebuilt from the in-tree sources, used as it is
        if not os.path.exists(LIB_PATH):
            raise UnerfError(f"UNERF_LIB={LIB_PATH} does not exist")
        return LIB_PATH
    digest, stamp = _source_digest(), LIB_PATH + ".sha256"

    def fresh():
        return os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == digest

    if not force and fresh():
        return LIB_PATH
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():   # another rank built it while we waited
                return LIB_PATH
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
            if VALIDATED_HIPCC not in ver:
                import warnings
                warnings.warn(f"libunerf: building with {ver.splitlines()[0] if ver else hipcc!r}; the kernels were validated with "
                              f"'{VALIDATED_HIPCC}*' (run tests/test_gpu_repeatability.py on the GPU before trusting this build)")
            tmp = f"{LIB_PATH}.tmp.{os.getpid()}"
            cmd = [hipcc] + HIPCC_FLAGS + ["-I", INCLUDE, "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise UnerfError("hipcc failed:\n" + res.stdout + res.stderr)
            # the listing of what was just built: no MFMA may read a VGPR within two wait states of the VALU instruction that
            # writes it (the compiler guarantees that for its own instructions, nobody does for inline assembly: isa_check.py)
            if not os.environ.get("UNERF_SKIP_ISA_CHECK"):
                from . import isa_check
                try:
                    isa_check.check_library(tmp, verbose=verbose)
                except isa_check.IsaHazard as e:
                    os.remove(tmp)
                    raise UnerfError(f"libunerf: refused by the ISA check -- {e}")
                except (FileNotFoundError, subprocess.CalledProcessError) as e:
                    import warnings
                    warnings.warn(f"libunerf: ISA check skipped ({e})")
            os.replace(tmp, LIB_PATH)
            with open(stamp, "w") as f:
                f.write(digest)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


class TcnnLevel(C.Structure):
    """include/unerf.h: unerf_tcnn_level"""
    _fields_ = [("scale", C.c_float), ("res", C.c_uint32), ("offset", C.c_uint32), ("size", C.c_uint32),
                ("dense", C.c_uint32)]


class DensityNet(C.Structure):
    _fields_ = [
        ("table", C.c_void_p), ("scalings", C.c_void_p), ("L", C.c_int), ("log2T", C.c_int),
        ("w0t", C.c_void_p), ("b0", C.c_void_p), ("w1t", C.c_void_p), ("b1", C.c_void_p), ("hidden", C.c_int),
        ("dense", C.c_void_p), ("n_dense", C.c_int), ("dense_off", C.c_int * 8), ("dense_dim", C.c_int * 8),
        ("tcnn_levels", C.c_void_p), ("use_aabb", C.c_int), ("aabb", C.c_float * 6), ("grid_half", C.c_int),
    ]


class FieldParams(C.Structure):
    _fields_ = [
        ("mode", C.c_int),
        ("table", C.c_void_p), ("scalings", C.c_void_p), ("L", C.c_int), ("log2T", C.c_int),
        ("w0t", C.c_void_p), ("b0", C.c_void_p), ("w1t", C.c_void_p), ("b1", C.c_void_p), ("out1", C.c_int),
        ("h0t", C.c_void_p), ("hb0", C.c_void_p), ("h1t", C.c_void_p), ("hb1", C.c_void_p),
        ("h2t", C.c_void_p), ("hb2", C.c_void_p),
        ("average_init_density", C.c_float), ("beta_min", C.c_float), ("sh_remap", C.c_int),
        ("K", C.c_int), ("seed", C.c_uint32), ("p_drop", C.c_float),
        ("ws_density", C.c_void_p), ("ws_rgb", C.c_void_p), ("n_lap", C.c_int), ("lap_mask_density", C.c_int),
        ("mfma_blob", C.c_void_p), ("lap_blob", C.c_void_p),
        ("tcnn_levels", C.c_void_p),
        ("mfma16_blob", C.c_void_p), ("lap16_blob", C.c_void_p),
        ("image_width", C.c_int), ("sample_major", C.c_int), ("drop_sites", C.c_int), ("lap_softplus", C.c_int), ("use_aabb", C.c_int),
        ("aabb", C.c_float * 6), ("f16_single", C.c_int), ("overflow_flag", C.c_void_p),
        ("h0_full_t", C.c_void_p), ("hb0_raw", C.c_void_p), ("app_embed", C.c_void_p),
        ("lap_chunk_rays", C.c_int), ("lap_sets", C.c_int), ("n_lap_rgb", C.c_int), ("packed_out", C.c_int),
        ("grid_half", C.c_int),
        ("hidden", C.c_int), ("hidden_color", C.c_int), ("geo_dim", C.c_int), ("feat_per_level", C.c_int), ("app_dim", C.c_int),
    ]


ABI_VERSION = 1420                                # include/unerf.h: UNERF_ABI_VERSION (struct layouts / argument lists)
FIELD_ACTIVE, FIELD_MCDROPOUT, FIELD_LAPLACE = 0, 1, 2
SPACING_PIECEWISE, SPACING_UNIFORM = 0, 1         # include/unerf.h: UNERF_SPACING_*
BG_LAST_SAMPLE, BG_NONE, BG_COLOR = 0, 1, 2       # include/unerf.h: UNERF_BG_*
# include/unerf.h: UNERF_CAMERA_* = nerfstudio's CameraType values of the camera models the ray kernel restates
CAMERA_PERSPECTIVE, CAMERA_FISHEYE, CAMERA_EQUIRECTANGULAR, CAMERA_ORTHOPHOTO = 1, 2, 3, 8
CAMERA_TYPES = (CAMERA_PERSPECTIVE, CAMERA_FISHEYE, CAMERA_EQUIRECTANGULAR, CAMERA_ORTHOPHOTO)
RASTER_NO_CULL = 1                                # include/unerf.h: UNERF_RASTER_NO_CULL
BUILD_TRUNK_FOLD, BUILD_LAP_EXP2 = 1, 2   # include/unerf.h: UNERF_BUILD_*
DROP_TRUNK, DROP_HEAD0, DROP_HEAD1, DROP_HEADIN = 1, 2, 4, 8     # include/unerf.h: UNERF_DROP_*

_vp, _i, _i64, _f, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol include/unerf.h declares
SIGNATURES = {
    "unerf_last_error": (C.c_char_p, []),
    "unerf_version": (_i, []),
    "unerf_build_flags": (_i, []),
    "unerf_device_count": (_i, []),
    "unerf_generate_rays": (_i, [_fp, _f, _f, _f, _f, _fp, _i, _i, _i, _i64, _i64, _vp, _vp, _vp, _vp]),
    "unerf_ray_box_bins": (_i, [_vp, _vp, _i64, _fp, _fp, _f, _f, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "unerf_ray_planes_bins": (_i, [_vp, _vp, _i64, _f, _f, _i, _vp, _i, _vp, _vp]),
    "unerf_hashgrid_fwd": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "unerf_hashgrid_fwd_tcnn": (_i, [_vp, _vp, C.POINTER(TcnnLevel), _i64, _i, _vp, _vp, _vp]),
    "unerf_hashgrid_fwd_tcnn_half": (_i, [_vp, _vp, C.POINTER(TcnnLevel), _i64, _i, _vp, _vp]),
    "unerf_proposal_density": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _f, _f, _i, C.POINTER(DensityNet), _f, _vp, _i64, _i, _vp]),
    "unerf_weights_pdf_resample": (_i, [_vp, _vp, _i64, _i64, _i, _f, _f, _i, _vp, _i, _f, _f, _vp, _vp, _vp, _vp,
                                        _i64, _i64, _vp]),
    "unerf_field_fwd": (_i, [_vp, _vp, _vp, _i64, _i, _f, _f, _i, _i64, C.POINTER(FieldParams), _vp, _vp, _vp, _vp, _vp,
                             _vp]),
    "unerf_field_gather": (_i, [_vp, _vp, _vp, _i64, _i, _f, _f, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "unerf_laplace_depth_weights": (_i, [_vp, _vp, _vp, _i64, _i, _f, _f, _i, _vp, _i, _u32, _i64, _vp, _vp]),
    "unerf_laplace_ggn_workspace_bytes": (C.c_size_t, [_i64, _i]),
    "unerf_laplace_ggn_diag": (_i, [_vp, _vp, _vp, _i64, _i, _f, _f, _i, C.POINTER(FieldParams), _i, _fp, _vp, C.c_size_t,
                                    _vp, _vp, _vp]),
    "unerf_composite_var": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _f, _f, _i, _vp, _i64, _i64, _i, _fp, _vp, _vp, _vp]),
    "unerf_composite_moments": (_i, [_vp, _vp, _vp, _i, _i64, _i, _f, _f, _i, _vp, _i64, _i64, _i, _fp, _vp, _vp, _vp, _vp]),
    "unerf_composite_var_planes": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _f, _f, _i, _vp, _i64, _i64, _i, _fp, _vp, _vp, _vp]),
    "unerf_composite_moments_planes": (_i, [_vp, _vp, _vp, _i, _i64, _i, _f, _f, _i, _vp, _i64, _i64, _i, _fp, _vp, _vp,
                                            _vp, _vp]),
    "unerf_moments": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp]),
    "unerf_splat_project": (_i, [_vp, _vp, _f, _vp, _fp, _f, _f, _f, _f, _i, _i, _i, _f, _i64, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp, _vp]),
    "unerf_splat_project_raw": (_i, [_vp, _vp, _f, _vp, _fp, _f, _f, _f, _f, _i, _i, _i, _f, _i64, _vp, _i, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _vp, _vp]),
    "unerf_splat_sh_colors": (_i, [_i, _vp, _fp, _vp, _vp, _f, _i64, _vp, _vp, _vp]),
    "unerf_splat_sh_colors_split": (_i, [_i, _vp, _fp, _vp, _vp, _vp, _f, _i64, _vp, _vp, _vp]),
    "unerf_splat_shade_inputs": (_i, [_i, _vp, _fp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "unerf_splat_sort_workspace_bytes": (_i64, [_i64, _i64]),
    "unerf_splat_count_intersects": (_i, [_vp, _i64, _vp, _vp, _i64, _vp]),
    "unerf_splat_bin_sort": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "unerf_splat_rasterize": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp,
                                   _vp]),
    "unerf_splat_alpha_normalize": (_i, [_vp, _i, _i, _vp, _i64, _vp, _i, _vp]),
    "unerf_splat_normalize_outputs": (_i, [_vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "unerf_splat_depth_sqdiff": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i64, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen csrc/libunerf.so and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the one that
    # initialises the device, so import torch before dlopen-ing libunerf (same SONAME -> shared).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise UnerfError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # the ctypes Structures and argument lists above are one ABI: a library of another one (UNERF_LIB pointing at a
    # stale A/B build) would be driven with mismatched layouts and answer with silent garbage
    got = lib.unerf_version()
    if got != ABI_VERSION:
        raise UnerfError(f"{LIB_PATH}: ABI version {got}, this binding is written for {ABI_VERSION} "
                         "(rebuild: python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().unerf_last_error().decode(errors="replace")
        raise UnerfError(f"{what or 'libunerf'} failed (rc={rc}): {msg}")


def require_gpu() -> None:
    if load().unerf_device_count() <= 0:
        raise UnerfError("no HIP device visible: libunerf has no CPU path")
