"""Empty inputs through every ray-/sample-/splat-shaped entry point of the C ABI: zero rays (an image chunk can be
empty after cropping, a rank can own no views) must return empty outputs, not a launch error; a whole render of zero
rays gives a dict of [0, C] tensors with the usual keys."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(dev, kind, **kw):
    from uncertainty_nerf_gs_amd import synthetic
    t = synthetic.make_scene_tensors(seed=1, kind=kind, log2T=12, prop_log2T=10)
    if kind == "laplace":
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=1, n_samples=20)
        kw.update(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
    return synthetic.scene_to_device(t, dev, **kw)


@pytest.mark.parametrize("kind,kw", [("active", {}), ("mcdropout", dict(K=8, seed=1, p_drop=0.2)), ("mcdropout", dict(K=0)),
                                     ("laplace", {})])
def test_zero_rays_through_the_nerf_entry_points(dev, kind, kw):
    from uncertainty_nerf_gs_amd import ops, render
    sd = _scene(dev, kind, **kw)
    z3 = torch.zeros(0, 3, device=dev)
    row = sd.const("bins", 256)
    dens = ops.proposal_density(z3, z3, row, sd.props[0], sd.near, sd.far, 0.01)
    assert dens.shape == (0, 256)
    sb, pd, w = ops.weights_pdf_resample(dens, row, sd.const("u", 96), sd.near, sd.far, want_weights=True)
    assert sb.shape == (0, 97) and pd.shape == (0, 1) and w.shape == (0, 256)
    fb = torch.zeros(0, 49, device=dev)
    density, rgb, aux, aux2 = ops.field_fwd(z3, z3, fb, sd.field, sd.near, sd.far)
    assert density.shape[-2:] == (0, 48) and rgb.shape[-3:] == (0, 48, 3)
    bins, planes = ops.ray_box_bins(z3, z3, torch.eye(4)[:3], torch.ones(3), sd.near, sd.far, row, want_planes=True)
    assert bins.shape == (0, 257) and planes[0].shape == (0, 1)
    assert ops.ray_planes_bins(torch.zeros(0, device=dev), torch.zeros(0, device=dev), sd.near, sd.far, row).shape == (0, 257)
    out = render.render_rays(sd, z3, z3)
    want = {"rgb", "accumulation", "depth", "expected_depth"}
    if kw.get("K", 1) != 0:            # K = 0 is the plain nerfacto render: no *_std keys
        want |= {"rgb_std", "depth_std"}
    assert want <= set(out)
    for k, v in out.items():
        assert v.shape[0] == 0, k
    o, d, pa = ops.generate_rays(torch.eye(4)[:3], 10.0, 10.0, 4.0, 3.0, 6, 8, dev, ray_start=48, count=0, pixel_area=True)
    assert o.shape == (0, 3) and d.shape == (0, 3) and pa.shape == (0, 1)


def test_zero_rays_through_composite_and_moments(dev):
    from uncertainty_nerf_gs_amd import ops
    S = 48
    fb = torch.zeros(0, S + 1, device=dev)
    out = ops.composite_var(torch.zeros(2, 0, S, device=dev), torch.zeros(2, 0, S, 3, device=dev), fb, 0.05, 1000.0)
    assert out.shape == (2, 0, 8)
    mean, var = ops.composite_moments(torch.zeros(4, 0, S, device=dev), torch.zeros(4, 0, S, 3, device=dev), fb, 0.05, 1000.0)
    assert mean.shape == (0, 8) and var.shape == (0, 8)
    m, v = ops.moments(torch.zeros(5, 0, 3, device=dev))
    assert m.shape == (0, 3) and v.shape == (0, 3)
    w = ops.laplace_depth_weights(torch.zeros(0, S, device=dev), torch.zeros(0, S, device=dev), fb, 0.05, 1000.0, None, 10, 0, 0)
    assert w.shape == (0, S)
    assert ops.hashgrid_fwd(torch.zeros(0, 3, device=dev), torch.zeros(4 << 10, 2, device=dev), torch.tensor([15.0, 31.0, 63.0, 127.0], device=dev), 10).shape == (0, 8)


def test_zero_splats_render_the_background(dev):
    from uncertainty_nerf_gs_amd import splat, synthetic
    gp = {k: v[:0].contiguous().to(dev) for k, v in synthetic.make_splat_tensors(seed=1, N=16).items()}
    H, W = 32, 48
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev)
    out = splat.active_splatfacto_outputs(gp, synthetic.orbit_c2w(0.3, radius=2.5, height=0.5), background=bg,
                                          fx=40.0, fy=40.0, cx=W / 2, cy=H / 2, H=H, W=W)
    assert out["rgb"].shape == (H, W, 3) and torch.allclose(out["rgb"], bg.expand(H, W, 3))
    assert out["accumulation"].abs().max() == 0
