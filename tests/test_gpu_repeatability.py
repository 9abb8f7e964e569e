"""Bit-reproducibility of the frame path at the BASELINE size: the same 1080p frame rendered several times by the same
scene must come out with the same bits, for every method and every arithmetic.  No kernel of the path has a data-dependent
order of floating-point operations (the only atomics are integer min / max of the per-chunk clip bounds and flag ORs), so
any difference is a defect -- a missed dependency, a hardware hazard the compiler does not insert wait states for, a read of
uninitialised memory.  (Round 4: a build of the split-f16 field kernels with a fused-multiply-add blend returned different
values in columns 16..31 of a few tiles per launch; only a full-size repeat shows that -- the sampled-ray parity tests
look at 4,096 of 2 million rays.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,precisions,grid", [("active", ("f16", "f16x2", "fp32"), "torch"), ("mcdropout", ("f16", "f16x2", "fp32"), "torch"),
                                                  ("laplace", ("f16x2", "fp32"), "torch"),
                                                  # round 5: the TCNN = 2 kernel instances (tcnn layout, half2 rows, half blend)
                                                  ("active", ("f16", "f16x2"), "tcnn-half"), ("mcdropout", ("f16", "f16x2"), "tcnn-half"),
                                                  ("laplace", ("f16x2",), "tcnn-half")])
def test_full_size_frames_repeat_bit_for_bit(dev, kind, precisions, grid):
    from uncertainty_nerf_gs_amd import render, synthetic
    t = synthetic.make_scene_tensors(seed=0, kind=kind, grid="tcnn" if grid == "tcnn-half" else "torch")
    if grid == "tcnn-half":
        t["grid_precision"] = "f16"
    kw = dict(K=8, seed=1234, p_drop=0.2) if kind == "mcdropout" else {}
    if kind == "laplace":
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=100)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
    sd = synthetic.scene_to_device(t, dev, **kw)
    cam, c2w = dict(synthetic.CAMERA_1080P), synthetic.orbit_c2w(0.7)
    shade = dict(keep_density=True) if kind == "active" else {}
    for precision in precisions:
        sd.field.precision = precision
        ref = None
        for rep in range(4 if precision != "fp32" else 2):
            out = render.render_camera(sd, c2w, depth_seed=7, **cam, **shade)
            if ref is None:
                ref = {k: v.clone() for k, v in out.items()}
                continue
            for k in ref:
                n = int((ref[k] != out[k]).sum())
                assert n == 0, f"{kind} {precision} frame {rep}: {n} values of `{k}` differ from the first render"
    assert sd.overflow_rerenders == 0


def test_full_size_splat_frames_repeat_bit_for_bit(dev):
    """active-splatfacto, 1 M splats at 1080p: the staged depth and tile sorts (LDS atomics, peer words, run-wise stores) and both
    rasteriser passes give the same bits every time."""
    import math
    from uncertainty_nerf_gs_amd import splat, synthetic
    gp = {k: v.to(dev) for k, v in synthetic.make_splat_tensors(seed=7, N=1000000).items()}
    H, W = 1080, 1920
    cam = dict(fx=1111.0, fy=1111.0, cx=W / 2, cy=H / 2, H=H, W=W)
    pose = synthetic.orbit_c2w(2 * math.pi * 5 / 24, radius=2.5, height=0.5).to(dev)
    bg = torch.zeros(3, device=dev)
    ref = None
    for rep in range(4):
        out = splat.active_splatfacto_outputs(gp, pose, background=bg, **cam)
        if ref is None:
            ref = {k: v.clone() for k, v in out.items()}
            continue
        for k in ref:
            n = int((ref[k] != out[k]).sum())
            assert n == 0, f"splat frame {rep}: {n} values of `{k}` differ from the first render"
