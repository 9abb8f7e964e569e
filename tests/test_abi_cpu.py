"""C-ABI surface: the library builds, loads, exports every symbol include/unerf.h declares, and
validates arguments before touching the GPU.  No compute is launched here."""
import threading
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "unerf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(unerf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_bound_and_exported(lib):
    declared = _declared_symbols()
    assert len(declared) >= 19
    assert sorted(lib.SIGNATURES) == declared, "lib.SIGNATURES must list exactly the symbols of include/unerf.h"
    handle = lib.load()
    for name in declared:
        assert getattr(handle, name) is not None


def test_version_and_error_string(lib):
    h = lib.load()
    assert h.unerf_version() == lib.ABI_VERSION == 1420
    assert h.unerf_build_flags() & lib.BUILD_TRUNK_FOLD          # the shipped build folds the K-pass trunk-out slabs
    assert isinstance(h.unerf_last_error(), bytes)


def test_null_pointers_are_rejected_before_launch(lib):
    h = lib.load()
    rc = h.unerf_hashgrid_fwd(None, None, None, 10, 16, 19, None, None, None)
    assert rc == -1
    assert b"null pointer" in h.unerf_last_error()
    rc = h.unerf_composite_var(1, 1, None, None, 1, 1, 4, 300, 0.05, 1000.0, 0, None, 0, 32768, 0, None, None, 1, None)
    assert rc == -1 and b"outside [1,256]" in h.unerf_last_error()
    # unknown background mode / a constant colour without its 3 floats
    rc = h.unerf_composite_var(1, 1, None, None, 1, 1, 4, 48, 0.05, 1000.0, 0, None, 0, 32768, 7, None, None, 1, None)
    assert rc == -1 and b"background=7" in h.unerf_last_error()
    rc = h.unerf_composite_var(1, 1, None, None, 1, 1, 4, 48, 0.05, 1000.0, 0, None, 0, 32768, lib.BG_COLOR, None, None, 1, None)
    assert rc == -1 and b"background_rgb" in h.unerf_last_error()
    rc = h.unerf_moments(None, 8, 4, 3, None, None, None)
    assert rc == -1


def test_bad_shapes_are_rejected(lib):
    h = lib.load()
    c2w = (C.c_float * 12)(*([0.0] * 12))
    rc = h.unerf_generate_rays(c2w, 1.0, 1.0, 0.0, 0.0, None, 1, 4, 4, 10, 10, 1, 1, None, None)
    assert rc == -1 and b"outside" in h.unerf_last_error()
    lens = (C.c_float * 6)(-0.05, float("nan"), 0.0, 0.0, 0.0, 0.0)
    rc = h.unerf_generate_rays(c2w, 1.0, 1.0, 0.0, 0.0, lens, 1, 4, 4, 0, 16, 1, 1, None, None)
    assert rc == -1 and b"distortion[1]" in h.unerf_last_error()
    for bad_type in (0, 4, 5, 6, 7, 9):   # omnidirectional stereo, VR180, FISHEYE624: not built, refused before any launch
        rc = h.unerf_generate_rays(c2w, 1.0, 1.0, 0.0, 0.0, None, bad_type, 4, 4, 0, 16, 1, 1, None, None)
        assert rc == -1 and b"camera_type" in h.unerf_last_error()
    net = lib.DensityNet(1, 1, 7, 17, 1, 1, 1, 1, 16, None, 0)
    rc = h.unerf_proposal_density(1, 1, 1, 0, 4, 256, 0.05, 1000.0, 0, C.byref(net), 0.01, 1, 0, 0, None)
    assert rc == -1 and b"unsupported" in h.unerf_last_error()
    rc = h.unerf_weights_pdf_resample(1, 1, 0, 4, 300, 0.05, 1000.0, 0, 1, 96, 0.01, 1e-5, 1, None, None, None, 0, 32768,
                                      None)
    assert rc == -1 and b"outside" in h.unerf_last_error()
    rc = h.unerf_splat_rasterize(None, 1, 1, 1, 1, 1, None, 9, 16, 16, 16, None, 0, -1, None, 1, 1, None, None)
    assert rc == -1 and b"C=9" in h.unerf_last_error()
    rc = h.unerf_splat_rasterize(None, 1, 1, 1, 1, 1, None, 5, 16, 16, 16, None, 0, 5, 1, 1, 1, None, None)
    assert rc == -1 and b"max_channel 5" in h.unerf_last_error()
    cam = (C.c_float * 3)(0.0, 0.0, 0.0)
    rc = h.unerf_splat_shade_inputs(3, 1, cam, 1, 1, None, 0.01, 1, None, 1, 4, 5, 1, 1, None)
    assert rc == -1 and b"needs log_unc" in h.unerf_last_error()
    rc = h.unerf_splat_shade_inputs(3, 1, cam, 1, 1, 1, 0.01, 1, None, 1, 4, 3, 1, 1, None)
    assert rc == -1 and b"C=3" in h.unerf_last_error()


def test_zero_counts_are_no_ops_and_oversized_inputs_are_refused(lib):
    """include/unerf.h conventions: R / N / count = 0 returns UNERF_OK without touching the per-element pointers (no
    launch, so this runs without a GPU); sample counters beyond 32 bits (the RNG counter / flat index) are refused."""
    h = lib.load()
    c2w = (C.c_float * 12)(*([0.0] * 12))
    assert h.unerf_generate_rays(c2w, 1.0, 1.0, 0.0, 0.0, None, 1, 4, 4, 16, 0, None, None, None, None) == 0
    assert h.unerf_moments(None, 8, 0, 3, None, None, None) == 0
    assert h.unerf_composite_var(None, None, None, None, None, 2, 0, 48, 0.05, 1000.0, 0, None, 0, 32768, 0, None, None, None, None) == 0
    assert h.unerf_composite_moments(None, None, None, 8, 0, 48, 0.05, 1000.0, 0, None, 0, 32768, 0, None, None, None, None, None) == 0
    assert h.unerf_weights_pdf_resample(None, None, 0, 0, 256, 0.05, 1000.0, 1, None, 96, 0.01, 1e-5, None, None, None, None, 0,
                                        32768, None) == 0
    assert h.unerf_laplace_depth_weights(None, None, None, 0, 48, 0.05, 1000.0, 0, None, 100, 0, 0, None, None) == 0
    w2b, half = (C.c_float * 12)(*([0.0] * 12)), (C.c_float * 3)(1.0, 1.0, 1.0)
    assert h.unerf_ray_box_bins(None, None, 0, w2b, half, 0.05, 1000.0, 0, None, 256, None, None, None, None) == 0
    assert h.unerf_ray_planes_bins(None, None, 0, 0.05, 1000.0, 0, None, 256, None, None) == 0
    vm = (C.c_float * 12)(*([0.0] * 12))
    assert h.unerf_splat_project(None, None, 1.0, None, vm, 1.0, 1.0, 0.0, 0.0, 16, 16, 16, 0.01, 0, None, None, None, None,
                                 None, None, None, None) == 0
    assert h.unerf_splat_project_raw(None, None, 1.0, None, vm, 1.0, 1.0, 0.0, 0.0, 16, 16, 16, 0.01, 0, None, 0, None, None,
                                     None, None, None, None, None, None, None) == 0
    # tight tile counts need somewhere to put the opacities they were made with; tight binning needs both arrays
    assert h.unerf_splat_project_raw(1, 1, 1.0, 1, vm, 1.0, 1.0, 0.0, 0.0, 16, 16, 16, 0.01, 4, 1, 0, None, 1, 1, 1, 1, 1, 1,
                                     1, None) == -1
    assert b"opacities_out" in h.unerf_last_error()
    assert h.unerf_splat_bin_sort(1, 1, 1, 1, 4, 4, 16, 16, 16, 1, None, None, 1, 1, 1, 1024, None) == -1
    assert b"tight lists need both" in h.unerf_last_error()
    assert h.unerf_splat_shade_inputs(3, None, (C.c_float * 3)(0.0, 0.0, 0.0), None, None, None, 0.01, None, None, None, 0, 5,
                                      None, None, None) == 0
    assert h.unerf_splat_depth_sqdiff(None, None, None, 1, 0, 16, 16, 0, None, None) == 0
    # ... but a non-zero count still needs its pointers
    assert h.unerf_ray_planes_bins(None, None, 5, 0.05, 1000.0, 0, None, 256, None, None) == -1
    assert b"null pointer" in h.unerf_last_error()
    # near must lie in front of far; bins need at least one interval
    assert h.unerf_ray_planes_bins(1, 1, 5, 10.0, 1.0, 0, 1, 256, 1, None) == -1
    fp = lib.FieldParams()
    for name in ("table", "scalings", "w0t", "b0", "w1t", "b1", "h0t", "hb0", "h1t", "hb1", "h2t", "hb2"):
        setattr(fp, name, 1)
    fp.L, fp.log2T, fp.mode, fp.out1 = 16, 19, 0, 17
    rc = h.unerf_field_fwd(1, 1, 1, 1 << 27, 48, 0.05, 1000.0, 0, 0, C.byref(fp), None, 1, 1, 1, None, None)
    assert rc == -1 and b"32 bits" in h.unerf_last_error()


def test_ops_refuse_cpu_tensors(lib):
    import torch
    from uncertainty_nerf_gs_amd import ops
    x = torch.rand(8, 3)
    with pytest.raises(lib.UnerfError, match="no CPU path"):
        ops.hashgrid_fwd(x, torch.rand(16 << 4, 2), torch.ones(16), 4)


def test_library_of_another_abi_version_is_refused(lib, monkeypatch):
    """UNERF_LIB may point at a build that is never rebuilt; a library whose unerf_version() is not the one
    the ctypes layouts were written for must not be driven"""
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "ABI_VERSION", lib.ABI_VERSION + 1)
    with pytest.raises(lib.UnerfError, match="ABI version"):
        lib.load()


def test_missing_library_fails_loudly(lib, monkeypatch, tmp_path):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.UnerfError, match="no CPU fallback"):
        lib.load()


def test_product_package_does_not_import_oracle():
    """The product path must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "uncertainty-nerf-gs_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_ctypes_structs_match_the_header_layout(lib, tmp_path):
    """sizeof / offsetof of every parameter struct, as a C compiler sees include/unerf.h, against the ctypes
    mirrors in lib.py (a silent mismatch would hand the kernels garbage pointers)."""
    import os
    import subprocess
    structs = {"unerf_density_net": lib.DensityNet, "unerf_field_params": lib.FieldParams, "unerf_tcnn_level": lib.TcnnLevel}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "unerf.h"', 'int main(void) {']
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} SIZEOF %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-I", inc, "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for row in out.strip().splitlines():
        cname, what, val = row.split()
        ct = structs[cname]
        if what == "SIZEOF":
            assert C.sizeof(ct) == int(val), (cname, C.sizeof(ct), val)
        else:
            assert getattr(ct, what).offset == int(val), (cname, what, getattr(ct, what).offset, val)
        seen += 1
    assert seen == sum(len(ct._fields_) + 1 for ct in structs.values())


def test_tight_tile_counts_are_only_offered_with_the_raw_projection(lib):
    """ops.splat_project: the tight counts depend on the activated opacity, which only the raw-parameter entry point makes"""
    import torch
    from uncertainty_nerf_gs_amd import ops
    z = torch.zeros(4, 3)
    with pytest.raises(ValueError, match="raw=True"):
        ops.splat_project(z, z, 1.0, torch.zeros(4, 4), torch.eye(4)[:3], 1.0, 1.0, 0.0, 0.0, 16, 16, opacity_logits=torch.zeros(4))


def test_error_string_is_per_thread(lib):
    """unerf_last_error() is thread-local: a call that fails in one thread leaves another thread's (empty or older) text
    alone -- two threads fail DIFFERENT argument checks at the same time and each reads back its own message"""
    h = lib.load()
    seen, gate = {}, threading.Barrier(2)

    def a():
        gate.wait(timeout=30)
        for _ in range(200):
            rc = h.unerf_hashgrid_fwd(None, None, None, 10, 16, 19, None, None, None)
            seen.setdefault("a", set()).add((rc, h.unerf_last_error()))

    def b():
        gate.wait(timeout=30)
        for _ in range(200):
            rc = h.unerf_composite_var(1, 1, None, None, 1, 1, 4, 48, 0.05, 1000.0, 0, None, 0, 32768, 7, None, None, 1, None)
            seen.setdefault("b", set()).add((rc, h.unerf_last_error()))
    ths = [threading.Thread(target=f) for f in (a, b)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert len(seen["a"]) == 1 and len(seen["b"]) == 1
    (rca, ea), = seen["a"]
    (rcb, eb), = seen["b"]
    assert rca == -1 and b"null pointer" in ea and rcb == -1 and b"background=7" in eb
