"""Gaussian-splat path through the C ABI against the numpy oracle.  Projection, tile boxes, sort
keys, sorted ids and bin edges are BIT-EXACT (same fp32 op order, -ffp-contract=off); the blended
images agree within the stated tolerances (exp implementation differs by ulps)."""
import numpy as np
import pytest
import torch

from oracle import splat_oracle as SO

pytestmark = pytest.mark.gpu


def _scene(N, seed=7, scale_shift=1.5):
    from uncertainty_nerf_gs_amd import synthetic
    gp = synthetic.make_splat_tensors(seed, N)
    gp["scales"] = gp["scales"] + scale_shift
    return gp


def _camera(theta=0.5, radius=2.5):
    from uncertainty_nerf_gs_amd import synthetic
    return synthetic.orbit_c2w(theta, radius=radius, height=0.5)


def _project_both(gp, c2w, fx, fy, cx, cy, H, W, dev):
    from uncertainty_nerf_gs_amd import ops, splat
    V = splat.viewmat_from_c2w(c2w)
    scales = torch.exp(gp["scales"])
    quats = gp["quats"] / gp["quats"].norm(dim=-1, keepdim=True)
    ref = SO.project_gaussians(gp["means"].numpy(), scales.numpy(), 1.0, quats.numpy(), V[:3].numpy(), fx, fy, cx, cy, H, W)
    got = ops.splat_project(gp["means"].to(dev), scales.to(dev), 1.0, quats.to(dev), V[:3], fx, fy, cx, cy, H, W)
    return ref, got, V


@pytest.mark.parametrize("N,H,W", [(20000, 48, 64), (5000, 200, 200), (1, 16, 16)])
def test_project_bit_exact(dev, N, H, W):
    gp = _scene(N)
    ref, got, _ = _project_both(gp, _camera(), 0.9 * W, 0.9 * W, W / 2, H / 2, H, W, dev)
    names = ("xys", "depths", "radii", "conics", "compensation", "num_tiles_hit", "cov3d")
    for name, g in zip(names, got):
        r = ref[name]
        g = g.cpu().numpy()
        same = (g == r) | (np.isnan(g) & np.isnan(r))
        assert same.all(), f"{name}: {np.count_nonzero(~same)} of {same.size} entries differ (bit-exact required)"
    assert (ref["radii"] > 0).sum() > 0 or N == 1


def test_project_culls_behind_camera_and_keeps_zeros(dev):
    from uncertainty_nerf_gs_amd import ops
    gp = _scene(4096)
    gp["means"][:100] *= 50.0       # far outside / behind
    gp["scales"][100:110] = -30.0   # degenerate tiny splats
    ref, got, _ = _project_both(gp, _camera(), 60.0, 60.0, 32.0, 24.0, 48, 64, dev)
    radii = got[2].cpu().numpy()
    assert np.array_equal(radii, ref["radii"])
    dead = radii == 0
    assert dead.any()
    assert np.all(got[0].cpu().numpy()[dead] == 0) and np.all(got[5].cpu().numpy()[dead] == 0)


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_sh_colors_and_beta(dev, degree):
    from uncertainty_nerf_gs_amd import ops
    gp = _scene(3000)
    c2w = _camera()
    coeffs = torch.cat((gp["features_dc"][:, None, :], gp["features_rest"]), dim=1).contiguous()
    col, beta = ops.splat_sh_colors(degree, gp["means"].to(dev), c2w[:3, 3], coeffs.to(dev),
                                    gp["log_uncertainties"].reshape(-1).to(dev), 0.01)
    ref = np.maximum(SO.spherical_harmonics(degree, (gp["means"] - c2w[:3, 3]).numpy(), coeffs.numpy()) + np.float32(0.5), 0)
    np.testing.assert_allclose(col.cpu().numpy(), ref, rtol=0, atol=2e-6)
    np.testing.assert_allclose(beta.cpu().numpy(), SO.softplus(gp["log_uncertainties"].numpy()).reshape(-1) + np.float32(0.01),
                               rtol=2e-6, atol=0)
    # the in-place form (features_dc / features_rest as the model stores them, no concatenation): the same bits
    col2, beta2 = ops.splat_sh_colors_split(degree, gp["means"].to(dev), c2w[:3, 3], gp["features_dc"].to(dev),
                                            gp["features_rest"].contiguous().to(dev),
                                            gp["log_uncertainties"].reshape(-1).to(dev), 0.01)
    assert torch.equal(col2, col) and torch.equal(beta2, beta)


@pytest.mark.parametrize("N,H,W", [(6000, 48, 64), (2000, 100, 37)])
def test_bin_sort_bit_exact(dev, N, H, W):
    from uncertainty_nerf_gs_amd import ops
    gp = _scene(N)
    ref, got, _ = _project_both(gp, _camera(1.3), 0.9 * W, 0.9 * W, W / 2, H / 2, H, W, dev)
    xys, depths, radii, conics, comp, tiles, _ = got
    I, cum, keys, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W)
    Ir, cumr, keysr, gidsr, binsr = SO.bin_and_sort(ref["xys"], ref["depths"], ref["radii"], ref["num_tiles_hit"], H, W)
    assert I == Ir and I > 0
    assert np.array_equal(cum.cpu().numpy(), cumr)
    assert np.array_equal(keys.cpu().numpy(), keysr), "sorted (tile<<32|depth) keys must be bit-exact"
    assert np.array_equal(gids.cpu().numpy(), gidsr), "sorted gaussian ids must match (stable radix order)"
    assert np.array_equal(bins.cpu().numpy(), binsr), "tile bin edges must be bit-exact"
    k = keys.cpu().numpy()
    assert np.all(k[1:] >= k[:-1])


def test_bin_sort_no_intersections(dev):
    from uncertainty_nerf_gs_amd import ops
    N, H, W = 128, 32, 32
    z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
    I, cum, keys, gids, bins = ops.splat_bin_sort(z(N, 2), z(N), z(N, dt=torch.int32), z(N, dt=torch.int32), H, W)
    assert I == 0 and keys.numel() == 0 and int(bins.abs().sum()) == 0
    img, fT, _ = ops.splat_rasterize(gids, bins, z(N, 2), z(N, 3), z(N, 3), z(N), H, W,
                                     torch.tensor([0.25, 0.5, 0.75], device=dev))
    assert torch.all(fT == 1) and torch.allclose(img[3, 5], torch.tensor([0.25, 0.5, 0.75], device=dev))


@pytest.mark.parametrize("C", [1, 3, 5])
def test_rasterize_matches_oracle(dev, C):
    from uncertainty_nerf_gs_amd import ops
    N, H, W = 4000, 50, 70   # ragged: not a multiple of the 16-pixel tile
    gp = _scene(N)
    ref, got, _ = _project_both(gp, _camera(2.2), 60.0, 60.0, W / 2, H / 2, H, W, dev)
    xys, depths, radii, conics, comp, tiles, _ = got
    I, cum, keys, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W)
    g = torch.Generator().manual_seed(C)
    colors = torch.rand(N, C, generator=g)
    opac = torch.sigmoid(gp["opacities"]).reshape(-1)
    bg = torch.rand(C, generator=g)
    img, fT, fidx = ops.splat_rasterize(gids, bins, xys, conics, colors.to(dev), opac.to(dev), H, W, bg.to(dev),
                                        want_final_idx=True)
    img_r, fT_r, fidx_r = SO.rasterize(gids.cpu().numpy(), bins.cpu().numpy(), ref["xys"], ref["conics"], colors.numpy(),
                                       opac.numpy(), H, W, bg.numpy())
    d = np.abs(img.cpu().numpy() - img_r)
    # a splat whose alpha grazes 1/255 (or T grazes 1e-4) may flip with a 1-ulp exp difference
    assert (d > 2e-5).mean() <= 2e-3, f"{(d > 2e-5).mean():.2e} of values off, worst {d.max():.2e}"
    assert np.abs(fT.cpu().numpy() - fT_r).max() <= 5e-3 and (np.abs(fT.cpu().numpy() - fT_r) > 2e-6).mean() <= 2e-3
    assert (fidx.cpu().numpy() != fidx_r).mean() <= 2e-3


def test_active_splatfacto_full_pipeline(dev):
    """ActiveSplatfactoModel.get_outputs: one sort + 5-channel pass + depth-variance pass vs the
    oracle's restatement of the reference's four-pass formulation."""
    from uncertainty_nerf_gs_amd import splat
    N, H, W = 6000, 60, 80
    gp = _scene(N)
    c2w = _camera(0.9)
    bg = torch.tensor([0.1, 0.2, 0.3])
    out = splat.active_splatfacto_outputs({k: v.to(dev) for k, v in gp.items()}, c2w, 70.0, 70.0, W / 2, H / 2, H, W,
                                          bg.to(dev))
    ref = SO.active_splatfacto_outputs({k: v.numpy() for k, v in gp.items()}, c2w.numpy(), 70.0, 70.0, W / 2, H / 2, H, W,
                                       bg.numpy())
    for k, atol, rtol in (("rgb", 3e-5, 0), ("accumulation", 3e-5, 0), ("uncertainty", 3e-5, 1e-5),
                          ("rgb_var", 3e-5, 1e-4), ("depth", 0, 2e-4), ("depth_var", 1e-6, 2e-3), ("depth_std", 1e-5, 2e-3)):
        g, r = out[k].cpu().numpy().astype(np.float64), ref[k].astype(np.float64)
        bad = np.abs(g - r) > atol + rtol * np.abs(r)
        assert bad.mean() <= 5e-3, f"{k}: {bad.mean():.2e} of pixels off, worst {np.abs(g - r).max():.2e}"
    assert out["rgb"].max() <= 1.0


def _fixture_model(dev, background_color="random", sh_degree=3, rasterize_mode="classic", step=30000):
    """ActiveSplatfactoModel holding the 400 splats of tests/golden/splat_get_outputs.npz (loaded like a checkpoint)"""
    from conftest import golden
    from uncertainty_nerf_gs_amd import models
    g = golden("splat_get_outputs.npz")
    cfg = models.ActiveSplatfactoModelConfig(background_color=background_color, sh_degree=sh_degree,
                                             rasterize_mode=rasterize_mode)
    m = models.ActiveSplatfactoModel(cfg, num_points=7)
    m.load_state_dict({f"gauss_params.{k[3:]}": torch.from_numpy(g[k]) for k in g.files if k.startswith("gp_")})
    m.step = step
    fx, fy, cx, cy, H, W = g["intr"]
    cam = models.Camera(torch.from_numpy(g["c2w"]), fx, fy, cx, cy, int(H), int(W))
    return m.to(dev).eval(), cam, g


@pytest.mark.parametrize("tag,kw", [("default", {}), ("white", dict(background_color="white")), ("sh0", dict(background_color="black", sh_degree=0)),
                                    ("early", dict(step=1500)), ("aa", dict(rasterize_mode="antialiased"))])
def test_splat_model_against_the_references_get_outputs(dev, tag, kw):
    """The Model mirror (config defaults included: Viser-grey "random" background, named colours, sh_degree 0, the
    SH-degree schedule, antialiased opacities) against the dict the REFERENCE's ActiveSplatfactoModel.get_outputs
    produced for the same splats and camera (tests/golden/splat_get_outputs.npz)."""
    m, cam, g = _fixture_model(dev, **kw)
    out = m.get_outputs(cam)
    keys = {k[len(tag) + 5:] for k in g.files if k.startswith(f"{tag}_out_")}
    assert set(out) == keys
    assert torch.equal(out["background"].cpu(), torch.from_numpy(g[f"{tag}_out_background"]))
    for k, atol, rtol in (("rgb", 3e-5, 0), ("accumulation", 3e-5, 0), ("uncertainty", 3e-5, 1e-5), ("rgb_var", 3e-5, 1e-4),
                          ("rgb_std", 3e-5, 1e-5), ("depth", 0, 2e-4), ("depth_var", 1e-6, 2e-3), ("depth_std", 1e-5, 2e-3)):
        got, ref = out[k].cpu().numpy().astype(np.float64), g[f"{tag}_out_{k}"].astype(np.float64)
        bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
        assert bad.mean() <= 5e-3, f"{tag}:{k}: {bad.mean():.2e} of pixels off, worst {np.abs(got - ref).max():.2e}"


class _Box:
    """stand-in for a nerfstudio OrientedBox: axis-aligned, `within(points) -> bool [N,1]`"""

    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    def within(self, pts):
        lo, hi = torch.tensor(self.lo, device=pts.device), torch.tensor(self.hi, device=pts.device)
        return ((pts > lo) & (pts < hi)).all(dim=-1, keepdim=True)


def test_splat_model_crop_box_and_empty_outputs(dev):
    """crop_box keeps only the splats inside it (activesplatfacto_model.py:174-180, 202-217); an empty crop or a
    camera that sees nothing returns get_empty_outputs (:176-177, :239-240): background image, depth 10, alpha 0."""
    m, cam, g = _fixture_model(dev)
    gp = {k[3:]: g[k] for k in g.files if k.startswith("gp_")}
    fx, fy, cx, cy, H, W = g["intr"]
    box = _Box([-0.5, -0.5, -0.5], [0.6, 0.6, 0.6])
    out = m.get_outputs_for_camera(cam, obb_box=box)
    ids = box.within(torch.from_numpy(gp["means"])).squeeze().numpy()
    assert 10 < ids.sum() < ids.size - 10
    ref = SO.active_splatfacto_outputs(gp, g["c2w"], fx, fy, cx, cy, int(H), int(W), np.array(splat_bg := [0.1490, 0.1647, 0.2157], np.float32),
                                       crop_ids=ids)
    for k, atol, rtol in (("rgb", 3e-5, 0), ("accumulation", 3e-5, 0), ("uncertainty", 3e-5, 1e-5), ("depth", 0, 2e-4)):
        got, r = out[k].cpu().numpy().astype(np.float64), ref[k].astype(np.float64)
        bad = np.abs(got - r) > atol + rtol * np.abs(r)
        assert bad.mean() <= 5e-3, f"crop:{k}: {bad.mean():.2e} off, worst {np.abs(got - r).max():.2e}"
    full = m.get_outputs_for_camera(cam)           # obb_box=None clears the crop again
    assert not torch.equal(full["rgb"], out["rgb"]) and m.crop_box is None
    for empty in (m.get_outputs_for_camera(cam, obb_box=_Box([5.0, 5.0, 5.0], [6.0, 6.0, 6.0])),):
        assert set(empty) == {"rgb", "depth", "accumulation", "background"}
        assert empty["rgb"].shape == (int(H), int(W), 3) and torch.equal(empty["rgb"][3, 4].cpu(), torch.tensor(splat_bg))
        assert torch.all(empty["depth"] == 10) and torch.all(empty["accumulation"] == 0)
    # camera looking away from every splat: all radii are zero
    from uncertainty_nerf_gs_amd import models
    away = torch.from_numpy(g["c2w"]).clone()
    away[:3, 3] = torch.tensor([50.0, 50.0, 50.0])
    away[:3, :3] = torch.eye(3)                    # looks down -z from far outside: splats are behind / off-screen
    cam2 = models.Camera(away, fx, fy, cx, cy, int(H), int(W))
    e2 = m.get_outputs(cam2)
    assert set(e2) == {"rgb", "depth", "accumulation", "background"} and torch.all(e2["accumulation"] == 0)


def test_full_size_splat_frame_properties(dev):
    """BASELINE size (N = 1 M splats, 1920x1080): size-independent properties of the bookkeeping --
    sorted keys, bins that tile the intersection list exactly, every listed splat really overlaps its
    tile, alpha in [0,1], finite images -- and idempotence (two renders are bit-identical)."""
    from uncertainty_nerf_gs_amd import ops, splat, synthetic
    N, H, W = 1_000_000, 1080, 1920
    gp = {k: v.to(dev) for k, v in synthetic.make_splat_tensors(seed=7, N=N).items()}
    c2w = synthetic.orbit_c2w(0.3, radius=2.5, height=0.5)
    fx = fy = 1111.0
    V = splat.viewmat_from_c2w(c2w)
    quats = (gp["quats"] / gp["quats"].norm(dim=-1, keepdim=True)).contiguous()
    xys, depths, radii, conics, comp, tiles, _ = ops.splat_project(gp["means"], torch.exp(gp["scales"]), 1.0, quats, V[:3],
                                                                  fx, fy, W / 2, H / 2, H, W)
    I, cum, keys, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W)
    assert I == int(tiles.sum().item()) and I > 0 and int(cum[-1]) == I
    assert torch.all(keys[1:] >= keys[:-1]), "intersection keys must be sorted (tile, then depth)"
    tbx, tby = (W + 15) // 16, (H + 15) // 16
    tile_of = (keys >> 32)
    assert int(tile_of.max()) < tbx * tby
    # bins: [start,end) per tile; non-empty bins are disjoint, ordered and cover [0,I)
    b = bins.long()
    nonempty = b[:, 1] > b[:, 0]
    assert int((b[nonempty, 1] - b[nonempty, 0]).sum()) == I
    starts = b[nonempty, 0]
    assert torch.all(starts[1:] >= b[nonempty, 1][:-1])
    # every entry of a bin belongs to that tile
    tid = torch.arange(tbx * tby, device=dev)[nonempty]
    assert torch.equal(tile_of[starts], tid) and torch.equal(tile_of[b[nonempty, 1] - 1], tid)
    # depth bits inside a tile are ascending (front to back)
    d_sorted = depths[gids.long()]
    same_tile = tile_of[1:] == tile_of[:-1]
    assert torch.all(d_sorted[1:][same_tile] >= d_sorted[:-1][same_tile])
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev)
    out1 = splat.active_splatfacto_outputs(gp, c2w, fx, fy, W / 2, H / 2, H, W, bg)
    out2 = splat.active_splatfacto_outputs(gp, c2w, fx, fy, W / 2, H / 2, H, W, bg)
    for k in ("rgb", "depth", "accumulation", "uncertainty", "depth_var"):
        assert torch.isfinite(out1[k]).all(), k
        assert torch.equal(out1[k], out2[k]), f"{k}: not deterministic"
    assert out1["accumulation"].min() >= 0 and out1["accumulation"].max() <= 1
    assert out1["rgb"].max() <= 1 and out1["rgb"].min() >= 0
    assert (out1["uncertainty"] >= 0).all() and (out1["depth_var"] >= 0).all()
    empty = out1["accumulation"][..., 0] == 0
    if empty.any():   # untouched pixels show the background exactly
        assert torch.equal(out1["rgb"][empty], bg.expand_as(out1["rgb"][empty]))
    # the frame above ran on the tight tile lists (the default); on gsplat's own lists -- every tile of each splat's radius
    # box, the bookkeeping checked in the first half of this test -- every output has the same bits, at full size too
    out3 = splat.active_splatfacto_outputs(gp, c2w, fx, fy, W / 2, H / 2, H, W, bg, tight=False)
    for k in out3:
        assert torch.equal(out1[k], out3[k]), f"{k}: tight and gsplat tile lists disagree"
    logits = gp["opacities"].reshape(-1).contiguous()
    tight = ops.splat_project(gp["means"], gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(), V[:3], fx, fy, W / 2, H / 2,
                              H, W, raw=True, opacity_logits=logits)
    assert bool((tight[5] <= tiles).all()) and 0.4 * I < int(tight[5].sum()) < 0.7 * I


@pytest.mark.parametrize("N,H,W,tight", [(6000, 48, 64, False), (6000, 48, 64, True), (300000, 600, 800, True),
                                         (1_000_000, 1080, 1920, True), (1_000_000, 1080, 1920, False), (50, 1080, 1920, True),
                                         (20000, 176, 176, False),       # 121 tiles: the single-pass form with ~1 M pairs
                                         (20000, 192, 192, True),        # 144 tiles: two passes with a 1-bit high digit
                                         (3000, 2000, 3000, True)])      # 23,500 tiles: beyond the own sorts (rocprim)
def test_one_pass_tile_sort_gives_the_radix_sorts_lists(dev, N, H, W, tight, monkeypatch):
    """unerf_splat_bin_sort's own tile sorts -- the default two-pass LSD sort (two digits of <= 7 bits, <= 128 write fronts
    per wave; one pass when the image has <= 127 tiles: the 48 x 64 cases) and the one-pass LDS-digit sort it replaced
    (UNERF_SPLAT_TILE_SORT=onepass: 8,160 tile counters in a wave's LDS) -- against rocprim's radix sort of the same
    (tile, splat) pairs (UNERF_SPLAT_TILE_SORT=radix): gaussian_ids_sorted, tile_bins and the 64-bit isect ids are
    bit-identical -- at toy sizes, with box and tight lists, at the BASELINE size (37 M / 20 M pairs) and with a handful of
    pairs."""
    from uncertainty_nerf_gs_amd import ops, splat, synthetic
    gp = {k: v.to(dev) for k, v in synthetic.make_splat_tensors(seed=11, N=N).items()}
    if N <= 6000:
        gp["scales"] = gp["scales"] + 1.5
    c2w = synthetic.orbit_c2w(0.7, radius=2.5, height=0.5)
    V = splat.viewmat_from_c2w(c2w)
    f = 1111.0 * W / 1920
    logits = gp["opacities"].reshape(-1).contiguous()
    pr = ops.splat_project(gp["means"], gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(), V[:3], f, f, W / 2, H / 2, H, W,
                           raw=True, opacity_logits=logits if tight else None)
    xys, depths, radii, conics, comp, tiles = pr[:6]
    tl = (conics, pr[7]) if tight else None
    monkeypatch.setenv("UNERF_SPLAT_TILE_SORT", "radix")
    I2, _, keys2, gids2, bins2 = ops.splat_bin_sort(xys, depths, radii, tiles, H, W, tight=tl)
    for which in (None, "onepass"):
        if which is None:
            monkeypatch.delenv("UNERF_SPLAT_TILE_SORT", raising=False)
        else:
            monkeypatch.setenv("UNERF_SPLAT_TILE_SORT", which)
        I, _, keys, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W, tight=tl)
        assert I == I2 and I > 0
        assert torch.equal(bins, bins2), f"{which}: tile ranges"
        assert torch.equal(gids, gids2), f"{which}: depth-ordered splat ids of every tile"
        assert torch.equal(keys, keys2), f"{which}: (tile << 32 | depth bits) ids"


def test_4k_frame_on_the_rocprim_tile_sort_repeats_and_is_timed(dev):
    """3840 x 2160 = 32,400 tiles: beyond TS_MAX_T1 (the own tile sorts' whole-key counters are sized for 12,000 tiles in 64 KB of
    LDS), so unerf_splat_bin_sort takes rocprim's radix sort for the (tile, splat) pairs -- SURVEY.md allows the library sort
    there.  VERDICT r5 weak #12: that path had one parity case (3000 x 2000, above) and no timing.  Here the BASELINE's 1 M
    splats at 4K: the frame renders, repeats bit for bit, agrees with the 1080p frame of the same camera where the two sample
    the same directions (every second pixel centre of the 4K grid is NOT a 1080p pixel centre, so only image statistics are
    compared), and its time goes to the parity report next to the 1080p frame's."""
    import math
    from uncertainty_nerf_gs_amd import splat, synthetic
    from test_gpu_nerf_e2e import _report
    gp = {k: v.to(dev) for k, v in synthetic.make_splat_tensors(seed=7, N=1_000_000).items()}
    pose = synthetic.orbit_c2w(2 * math.pi * 5 / 24, radius=2.5, height=0.5).to(dev)
    bg = torch.zeros(3, device=dev)
    rec = {}
    outs = {}
    for tag, (H, W) in (("1080p", (1080, 1920)), ("4k", (2160, 3840))):
        cam = dict(fx=1111.0 * W / 1920, fy=1111.0 * W / 1920, cx=W / 2, cy=H / 2, H=H, W=W)
        first = splat.active_splatfacto_outputs(gp, pose, background=bg, **cam)
        first = {k: v.clone() for k, v in first.items() if torch.is_tensor(v)}
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = splat.active_splatfacto_outputs(gp, pose, background=bg, **cam)
        e1.record()
        e1.synchronize()
        rec[f"ms_per_frame_{tag}"] = e0.elapsed_time(e1) / 5
        for k, v in first.items():
            assert torch.equal(out[k], v), f"{tag}: `{k}` differs between two renders"
            assert torch.isfinite(v).all(), (tag, k)
        outs[tag] = first
    a, b = outs["1080p"], outs["4k"]
    assert b["rgb"].shape == (2160, 3840, 3)
    for k in ("rgb", "accumulation"):
        assert abs(float(a[k].mean()) - float(b[k].mean())) < 5e-3, k
    rec["tiles_4k"] = (2160 // 16) * (3840 // 16)
    rec["ms_per_megapixel_1080p"] = rec["ms_per_frame_1080p"] / 2.0736
    rec["ms_per_megapixel_4k"] = rec["ms_per_frame_4k"] / 8.2944
    _report("splat-4k-rocprim-tile-sort", rec)


@pytest.mark.parametrize("N,H,W", [(5000, 96, 128), (1_000_000, 1080, 1920), (1025, 64, 64), (3, 64, 64)])
def test_depth_sort_is_rocprims(dev, N, H, W, monkeypatch):
    """unerf_splat_bin_sort's own depth sort (four 8-bit LSD passes, one wave per 1,024 splats) against rocprim's stable
    sort of the same (depth bits, index) pairs (UNERF_SPLAT_DEPTH_SORT=rocprim): the lists, the ranges and the 64-bit ids
    are bit-identical -- with EQUAL depths in the set (a block of splats duplicated at the same position: stability decides
    their order inside a tile) and culled splats (key 0xFFFFFFFF) in between."""
    from uncertainty_nerf_gs_amd import ops, splat, synthetic
    gp = {k: v.to(dev) for k, v in synthetic.make_splat_tensors(seed=5, N=N).items()}
    if N >= 2000:
        for k in ("means", "scales", "quats"):
            gp[k][1000:2000] = gp[k][:1000]              # same centres, same shapes: equal depths, same tiles
    if N <= 5000:
        gp["scales"] = gp["scales"] + 1.0
    c2w = synthetic.orbit_c2w(0.3, radius=2.5, height=0.4)
    V = splat.viewmat_from_c2w(c2w)
    f = 1111.0 * W / 1920
    pr = ops.splat_project(gp["means"], gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(), V[:3], f, f, W / 2, H / 2, H, W, raw=True)
    xys, depths, radii, conics, comp, tiles = pr[:6]
    if N >= 2000:
        assert torch.equal(depths[1000:2000], depths[:1000])
    monkeypatch.setenv("UNERF_SPLAT_DEPTH_SORT", "rocprim")
    I2, _, keys2, gids2, bins2 = ops.splat_bin_sort(xys, depths, radii, tiles, H, W)
    monkeypatch.delenv("UNERF_SPLAT_DEPTH_SORT")
    I, _, keys, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W)
    assert I == I2
    assert torch.equal(bins, bins2) and torch.equal(gids, gids2) and torch.equal(keys, keys2)


@pytest.mark.parametrize("N", [1, 5, 1023, 1024, 1025, 4097, 1_000_003, 4096 * 1024, 4096 * 1024 + 1, 6_300_017])
def test_count_intersects_scan(dev, N, monkeypatch):
    """unerf_splat_count_intersects' own inclusive scan (block sums, then every block behind the sum of the blocks before it;
    above 4,096 blocks = 4 M splats a one-workgroup middle step turns the block sums into their prefixes first, so the
    work stays linear) against torch.cumsum and against hipcub's scan (UNERF_SPLAT_SCAN=rocprim), ragged sizes included"""
    from uncertainty_nerf_gs_amd import ops
    g = torch.Generator().manual_seed(N)
    tiles = torch.randint(0, 60 if N < 2_000_000 else 30, (N,), generator=g, dtype=torch.int32).to(dev)
    want = torch.cumsum(tiles.long(), 0).to(torch.int32)
    c = ops.SplatCount(tiles)
    assert c.wait() == int(want[-1]) and torch.equal(c.cum, want)
    monkeypatch.setenv("UNERF_SPLAT_SCAN", "rocprim")
    c2 = ops.SplatCount(tiles)
    assert c2.wait() == int(want[-1]) and torch.equal(c2.cum, want)


def test_normalize_outputs_equals_the_torch_calls_it_replaces(dev):
    """unerf_splat_normalize_outputs (one pass: depth normalisation + rgb clamp + accumulation + uncertainty^2, and for the
    second pass sqrt of the normalised channel) against unerf_splat_alpha_normalize + torch.clamp / 1 - T / ** 2 / sqrt
    (activesplatfacto_model.py:275, 319, 356, 359-367): same bits, NaN, inf, values above 1 and alpha = 0 pixels included."""
    from uncertainty_nerf_gs_amd import ops
    H, W = 37, 53
    g = torch.Generator().manual_seed(3)
    img = (torch.rand(H, W, 5, generator=g) * 1.6).to(dev)
    img[0, 0, 1] = float("nan"); img[0, 1, 2] = float("inf"); img[1, 0, 0] = -0.25; img[2, 2, 3] = float("nan")
    fT = torch.rand(H, W, generator=g).to(dev)
    fT[3, :] = 1.0                                     # alpha = 0: the channel maximum goes in
    mx = img[..., 4].max().reshape(1).clone()
    want_img = img.clone()
    ops.splat_alpha_normalize(want_img, 4, fT, max_ready=mx)
    want = (torch.clamp(img[..., 0:3], max=1.0), (1.0 - fT)[..., None], img[..., 3:4] ** 2, want_img[..., 4:5].sqrt())
    got_img = img.clone()
    rgb, acc, sq, _ = ops.splat_normalize_outputs(got_img, 4, fT, mx, rgb=True, acc=True, sq_ch=3)
    assert torch.equal(got_img.nan_to_num(-7.0), want_img.nan_to_num(-7.0))
    for a, b, name in ((rgb, want[0], "rgb"), (acc, want[1], "accumulation"), (sq, want[2], "rgb_var")):
        assert a.shape == b.shape and torch.equal(a.nan_to_num(-7.0), b.nan_to_num(-7.0)), name
    assert bool(torch.isnan(rgb[0, 0, 1])) and float(rgb[0, 1, 2]) == 1.0 and float(rgb[1, 0, 0]) == -0.25
    one = img[..., 4:5].clone().contiguous()           # second pass: a one-channel image, sqrt of the normalised value
    want1 = one.clone()
    mx1 = one.max().reshape(1).clone()
    ops.splat_alpha_normalize(want1, 0, fT, max_ready=mx1)
    _, _, _, rt = ops.splat_normalize_outputs(one, 0, fT, mx1, sqrt=True)
    assert torch.equal(one, want1) and torch.equal(rt, want1.sqrt())


@pytest.mark.parametrize("C", [1, 5])
def test_wave_level_culling_and_bounded_pass_change_no_bit(dev, C):
    """the default schedule (a wave walks only the splats whose alpha >= 1/255 ellipse reaches its 8 x 8 quadrant of the tile) against
    gsplat's (every wave walks every staged splat): identical images, transmittances and final indices; and a second
    pass bounded by the first pass's final indices (the depth-variance pass) equals the unbounded one.  Splats with
    tiny opacity and huge / needle-shaped / degenerate / indefinite conics are in the set."""
    from uncertainty_nerf_gs_amd import ops
    N, H, W = 6000, 100, 150
    gp = _scene(N)
    gp["scales"][:50] += 2.0                     # a few very large splats
    gp["scales"][50:100, 0] -= 3.0               # needles
    gp["opacities"][100:400] = -7.0              # sigmoid -> 9e-4 < 1/255: never visible
    ref, got, _ = _project_both(gp, _camera(2.2), 60.0, 60.0, W / 2, H / 2, H, W, dev)
    xys, depths, radii, conics, comp, tiles, _ = got
    # (xys / radii / tiles stay as projected: the binning derives its intersection counts from them)
    conics = conics.clone()
    conics[450:460] = 0.0                                    # det = 0: no ellipse -> must not be culled wrongly
    conics[460:470, 1] = 5.0                                 # indefinite form (sigma can be negative)
    I, cum, keys, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W)
    g = torch.Generator().manual_seed(C)
    colors = torch.rand(N, C, generator=g).to(dev)
    opac = torch.sigmoid(gp["opacities"]).reshape(-1).to(dev)
    bg = torch.rand(C, generator=g).to(dev)
    a = ops.splat_rasterize(gids, bins, xys, conics, colors, opac, H, W, bg, want_final_idx=True)
    b = ops.splat_rasterize(gids, bins, xys, conics, colors, opac, H, W, bg, want_final_idx=True, cull=False)
    for x, y, what in zip(a, b, ("image", "final_T", "final_idx")):
        assert torch.equal(x, y), what
    assert float((a[1] < 1).float().mean()) > 0.5            # the scene does cover the image
    c = ops.splat_rasterize(gids, bins, xys, conics, colors, opac, H, W, bg, stop_idx=a[2])
    assert torch.equal(c[0], a[0])
    # final_T of a bounded pass: the transmittance behind the last BLENDED splat -- the unbounded pass may have multiplied
    # one more (1 - alpha) in before its T <= 1e-4 stop... it does not: the stopping splat is not applied either
    assert torch.equal(c[1], a[1])


def test_frame_prologue_fused_into_the_kernels(dev):
    """the per-frame torch launches of the reference's get_outputs that the frame now leaves to its kernels:
    exp(scales) and quats / quats.norm() inside the projection (unerf_splat_project_raw); the SH colours, beta, the
    [rgb, beta, depth] concatenation and sigmoid(opacities) [* compensation] in one launch (unerf_splat_shade_inputs)"""
    from uncertainty_nerf_gs_amd import ops, splat
    N, H, W = 20000, 120, 160
    gp = {k: v.to(dev) for k, v in _scene(N).items()}
    gp["quats"] = gp["quats"] * (0.25 + 3.0 * torch.rand(N, 1, device=dev))      # far from unit length
    c2w = _camera()
    V = splat.viewmat_from_c2w(c2w)
    K = (0.9 * W, 0.9 * W, W / 2, H / 2, H, W)
    want = ops.splat_project(gp["means"], torch.exp(gp["scales"]), 1.0,
                             (gp["quats"] / gp["quats"].norm(dim=-1, keepdim=True)).contiguous(), V[:3], *K)
    got = ops.splat_project(gp["means"], gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(), V[:3], *K, raw=True)
    # the same operations; only the summation order inside torch's norm() is not pinned, so: equal up to an ulp of the
    # quaternion, which can move a radius across an integer for a handful of splats at most
    same_r = (got[2] == want[2])
    assert float(same_r.float().mean()) > 0.999
    for name, g, w_, tol in (("xys", got[0], want[0], 1e-3), ("depths", got[1], want[1], 1e-6), ("conics", got[3], want[3], 1e-4),
                             ("compensation", got[4], want[4], 1e-5)):
        g, w_ = g[same_r], w_[same_r]
        assert float((g - w_).abs().max()) <= tol * max(1.0, float(w_.abs().max())), name
    assert int((got[5][same_r] != want[5][same_r]).sum()) == 0
    assert int((want[2] > 0).sum()) > N // 4

    xys, depths, radii, conics, comp, tiles, _ = want
    lu = gp["log_uncertainties"].reshape(-1).contiguous()
    for degree in (3, 1, -1):
        col, beta = ops.splat_sh_colors_split(degree, gp["means"], c2w[:3, 3], gp["features_dc"].contiguous(),
                                              gp["features_rest"].contiguous(), lu, 0.01)
        for compensation in (None, comp):
            rows, opac = ops.splat_shade_inputs(degree, gp["means"], c2w[:3, 3], gp["features_dc"].contiguous(),
                                                gp["features_rest"].contiguous(), lu, 0.01,
                                                gp["opacities"].reshape(-1).contiguous(), compensation, depths)
            assert torch.equal(rows, torch.cat([col, beta[:, None], depths[:, None]], dim=1))
            o = torch.sigmoid(gp["opacities"]).reshape(-1)
            o = o * comp if compensation is not None else o
            torch.testing.assert_close(opac, o, rtol=3e-7, atol=0)
        rows4, _ = ops.splat_shade_inputs(degree, gp["means"], c2w[:3, 3], gp["features_dc"].contiguous(),
                                          gp["features_rest"].contiguous(), None, 0.01,
                                          gp["opacities"].reshape(-1).contiguous(), None, depths)
        assert torch.equal(rows4, torch.cat([col, depths[:, None]], dim=1))


@pytest.mark.parametrize("C,ch", [(5, 4), (4, 3), (1, 0)])
def test_channel_maximum_taken_inside_the_rasteriser(dev, C, ch):
    """`depth_im.max()` of the alpha normalisation (:319, :356) comes out of the raster pass: the same float as a
    reduction over the written image, and the normalised image is the one the stand-alone path gives"""
    from uncertainty_nerf_gs_amd import ops
    N, H, W = 6000, 100, 150
    gp = _scene(N)
    ref, got, _ = _project_both(gp, _camera(2.2), 60.0, 60.0, W / 2, H / 2, H, W, dev)
    xys, depths, radii, conics, comp, tiles, _ = got
    I, cum, keys, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W)
    g = torch.Generator().manual_seed(C)
    colors = (torch.rand(N, C, generator=g) * 7.0).to(dev)
    opac = torch.sigmoid(gp["opacities"]).reshape(-1).to(dev)
    bg = torch.zeros(C, device=dev)
    mx = torch.zeros(1, device=dev)
    img, fT, fidx = ops.splat_rasterize(gids, bins, xys, conics, colors, opac, H, W, bg, want_final_idx=True, chan_max=(ch, mx))
    plain, fT0, _ = ops.splat_rasterize(gids, bins, xys, conics, colors, opac, H, W, bg)
    assert torch.equal(img, plain) and torch.equal(fT, fT0)
    assert float(mx) == float(img[..., ch].max()) > 0
    a, b = img.clone(), img.clone()
    ops.splat_alpha_normalize(a, ch, fT, max_ready=mx)
    ops.splat_alpha_normalize(b, ch, fT)
    assert torch.equal(a, b)
    # a bounded pass reports its maximum as well
    mx2 = torch.zeros(1, device=dev)
    img2, _, _ = ops.splat_rasterize(gids, bins, xys, conics, colors, opac, H, W, bg, stop_idx=fidx, chan_max=(ch, mx2))
    assert torch.equal(img2, img) and float(mx2) == float(mx)


def _hard_scene(N, dev, seed=11):
    """needles, pancakes, huge and faint splats, opacities on both sides of 1/255"""
    gp = _scene(N, seed=seed)
    gp["scales"][:N // 20] += 2.0
    gp["scales"][N // 20:N // 8, 0] -= 3.0
    gp["scales"][N // 8:N // 5, 1] += 1.5
    gp["opacities"][N // 5:N // 3] = -5.6 + 0.4 * torch.randn(N // 3 - N // 5, 1)     # sigmoid ~ 1/255 +- a factor
    gp["opacities"][N // 3:N // 2] = 6.0                                             # nearly opaque: ellipse beyond 3 sigma
    return {k: v.to(dev) for k, v in gp.items()}


@pytest.mark.parametrize("mode", ["classic", "antialiased"])
def test_tight_tile_lists_change_no_output_bit(dev, mode):
    """binning a splat only into the tiles its alpha >= 1/255 ellipse reaches (instead of gsplat's radius box) removes only
    pairs the blend loop skips: every output of the frame keeps its bits, with fewer intersections to sort"""
    from uncertainty_nerf_gs_amd import ops, splat
    N, H, W = 30000, 200, 296
    gp = _hard_scene(N, dev)
    c2w = _camera(0.9, 2.3)
    K = (0.8 * W, 0.8 * W, W / 2, H / 2, H, W)
    bg = torch.tensor([0.2, 0.5, 0.7])
    a = splat.active_splatfacto_outputs(gp, c2w, *K, bg, rasterize_mode=mode, tight=True)
    b = splat.active_splatfacto_outputs(gp, c2w, *K, bg, rasterize_mode=mode, tight=False)
    for k in b:
        assert torch.equal(a[k], b[k]), k
    assert float(b["accumulation"].mean()) > 0.3
    plain = {k: v for k, v in gp.items() if k != "log_uncertainties"}
    a = splat.active_splatfacto_outputs(plain, c2w, *K, bg, rasterize_mode=mode, tight=True)
    b = splat.active_splatfacto_outputs(plain, c2w, *K, bg, rasterize_mode=mode, tight=False)
    for k in b:
        assert torch.equal(a[k], b[k]), k
    # how much shorter the lists are
    V = splat.viewmat_from_c2w(c2w)
    logits = gp["opacities"].reshape(-1).contiguous()
    full = ops.splat_project(gp["means"], gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(), V[:3], *K, raw=True)
    tight = ops.splat_project(gp["means"], gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(), V[:3], *K, raw=True,
                              opacity_logits=logits, antialiased=mode == "antialiased")
    for x, y in zip(full[:5], tight[:5]):
        assert torch.equal(x, y)                       # xys, depths, radii, conics, compensation: gsplat's, untouched
    assert bool((tight[5] <= full[5]).all()) and int(tight[5].sum()) < 0.75 * int(full[5].sum())
    o = torch.sigmoid(logits) * (full[4] if mode == "antialiased" else 1.0)
    vis = full[2] > 0
    torch.testing.assert_close(tight[7][vis], o[vis], rtol=3e-7, atol=0)


def test_tight_tile_lists_are_conservative(dev):
    """brute force: every (tile, splat) pair with a pixel that the blend loop would NOT skip (sigma >= 0 and
    alpha = min(0.999, o exp(-sigma)) >= 1/255) is in the tight lists, the tight lists are a subset of gsplat's, and
    both keep the depth order"""
    from uncertainty_nerf_gs_amd import ops, splat
    N, H, W = 4000, 112, 176
    gp = _hard_scene(N, dev, seed=5)
    c2w = _camera(0.3, 2.4)
    V = splat.viewmat_from_c2w(c2w)
    K = (0.8 * W, 0.8 * W, W / 2, H / 2, H, W)
    logits = gp["opacities"].reshape(-1).contiguous()
    args = (gp["means"], gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(), V[:3]) + K
    xys, depths, radii, conics, comp, tiles_t, _, opac = ops.splat_project(*args, raw=True, opacity_logits=logits)
    tiles_f = ops.splat_project(*args, raw=True)[5]
    It, _, _, gids_t, bins_t = ops.splat_bin_sort(xys, depths, radii, tiles_t, H, W, want_isect_ids=False, tight=(conics, opac))
    If, _, _, gids_f, bins_f = ops.splat_bin_sort(xys, depths, radii, tiles_f, H, W, want_isect_ids=False)
    assert 0 < It < If
    tbx, tby = (W + 15) // 16, (H + 15) // 16

    def member(gids, bins):
        m = torch.zeros(tbx * tby, N, dtype=torch.bool, device=dev)
        b = bins.cpu()
        for t in range(tbx * tby):
            ids = gids[int(b[t, 0]):int(b[t, 1])].long()
            assert bool((depths[ids][1:] >= depths[ids][:-1]).all())           # depth order inside a tile
            assert ids.unique().numel() == ids.numel()
            m[t, ids] = True
        return m

    mt, mf = member(gids_t, bins_t), member(gids_f, bins_f)
    assert not bool((mt & ~mf).any())                                          # subset of gsplat's lists
    # pixels that blend: per splat chunk, alpha at every pixel centre
    py, px = torch.meshgrid(torch.arange(H, device=dev) + 0.5, torch.arange(W, device=dev) + 0.5, indexing="ij")
    tile_of = ((py - 0.5).long() // 16) * tbx + (px - 0.5).long() // 16       # [H,W]
    need = torch.zeros_like(mt)
    vis = torch.nonzero(radii > 0).reshape(-1)
    for chunk in vis.split(256):
        dx = xys[chunk, 0, None, None] - px[None]
        dy = xys[chunk, 1, None, None] - py[None]
        ca, cb, cc = conics[chunk, 0, None, None], conics[chunk, 1, None, None], conics[chunk, 2, None, None]
        sigma = 0.5 * (ca * dx * dx + cc * dy * dy) + cb * dx * dy
        alpha = torch.clamp(opac[chunk, None, None] * torch.exp(-sigma), max=0.999)
        hit = (sigma >= 0) & (alpha >= (1.0 / 255.0) * (1 - 1e-4))             # a hair more than the loop keeps
        for k, g in enumerate(chunk.tolist()):
            need[tile_of[hit[k]].unique(), g] = True
    need &= mf                                                                 # gsplat never blends outside its box
    missing = need & ~mt
    assert not bool(missing.any()), f"{int(missing.sum())} blending (tile, splat) pairs are not in the tight lists"
    assert int(need.sum()) > 0.5 * int(mt.sum())                               # and the tight lists are tight


def test_tight_lists_all_faint_frame_is_the_references_not_the_empty_one(dev):
    """radii.sum() > 0 with every opacity under 1/255: the reference rasterises (nothing blends: depth 0, rgb =
    background), it does not return get_empty_outputs (depth 10)"""
    from uncertainty_nerf_gs_amd import splat
    gp = {k: v.to(dev) for k, v in _scene(2000).items()}
    gp["opacities"][:] = -9.0
    bg = torch.tensor([0.1, 0.2, 0.3])
    K = (50.0, 50.0, 32.0, 24.0, 48, 64)
    a = splat.active_splatfacto_outputs(gp, _camera(), *K, bg, tight=True)
    b = splat.active_splatfacto_outputs(gp, _camera(), *K, bg, tight=False)
    for k in b:
        assert torch.equal(a[k], b[k]), k
    assert float(a["depth"].abs().max()) == 0.0 and float(a["accumulation"].max()) == 0.0
