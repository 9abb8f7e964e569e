#!/usr/bin/env python
"""Pin the [UPSTREAM-RECALL] half of the oracle against a REAL install, wherever one exists.  (test infrastructure)

    python tests/tools/diff_upstream.py [--json out.json] [--cuda]

nerfstudio 1.1.0, gsplat 0.1.11 and tiny-cuda-nn are absent from the build container, so oracle/nerf_oracle.py and
oracle/splat_oracle.py restate their L0 primitives from the published sources (SURVEY.md 8a "L0") and those
restatements are pinned only through the reference's own Field / Model code that calls them.  This script is the missing
hook (VERDICT r5 "missing" 2): on a machine where `import nerfstudio` / `import gsplat` work it runs every oracle L0
function and the real upstream function on the same seeded inputs and prints, per function, the largest difference (and for
index work whether the integers are equal).  Where a package is missing the function is reported as "upstream absent" and
the script exits 0 -- nothing here is imported by the product or by the test suite's pass/fail path
(tests/test_tools_cpu.py only checks that it runs and skips cleanly).

Upstream entry points compared (module paths as of nerfstudio 1.1.0 / gsplat 0.1.11):
  L0.1 HashEncoding(implementation="torch").pytorch_fwd, its hash_fn and scalings   nerfstudio.field_components.encodings
  L0.2 MLP(implementation="torch")                                                  nerfstudio.field_components.mlp
  L0.3 SHEncoding(levels=4, implementation="torch")                                 nerfstudio.field_components.encodings
  L0.4 SceneContraction(order=inf)                                                  nerfstudio.field_components.spatial_distortions
  L0.5 trunc_exp                                                                    nerfstudio.field_components.activations
  L0.6 UniformLinDispPiecewiseSampler spacing, PDFSampler (eval)                    nerfstudio.model_components.ray_samplers
  L0.7 RaySamples.get_weights                                                       nerfstudio.cameras.rays
  L0.8 RGBRenderer / AccumulationRenderer / DepthRenderer / UncertaintyRenderer     nerfstudio.model_components.renderers
  L0.9 Cameras.generate_rays (perspective, with and without distortion)             nerfstudio.cameras.cameras
  L0.10 project_gaussians / spherical_harmonics / rasterize_gaussians (CUDA only)   gsplat
Each comparison is wrapped on its own: an API that moved reports the exception text instead of hiding the other rows.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _try_import(name):
    try:
        return __import__(name, fromlist=["_"]), None
    except Exception as e:  # noqa: BLE001 -- absence is the normal case here
        return None, f"{type(e).__name__}: {e}"


ROWS = []


def row(name, needs):
    """decorator: register one comparison; `needs` = top-level packages it imports"""
    def deco(fn):
        ROWS.append((name, needs, fn))
        return fn
    return deco


def _maxdiff(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    fin = torch.isfinite(a) & torch.isfinite(b)
    same_nonfinite = bool(((a == b) | (torch.isnan(a) & torch.isnan(b)))[~fin].all()) if (~fin).any() else True
    return {"max_abs": float((a - b)[fin].abs().max()) if fin.any() else 0.0, "nonfinite_agree": same_nonfinite,
            "scale": float(b[fin].abs().max()) if fin.any() else 0.0}


@row("L0.1 hash_scalings / hash_fn / HashEncoding.pytorch_fwd", ["nerfstudio"])
def _hash():
    from nerfstudio.field_components.encodings import HashEncoding
    from oracle import nerf_oracle as O
    out = {}
    for (L, lo, hi, log2T) in ((16, 16, 2048, 19), (5, 16, 128, 17), (5, 16, 256, 17), (4, 16, 4096, 15)):
        enc = HashEncoding(num_levels=L, min_res=lo, max_res=hi, log2_hashmap_size=log2T, features_per_level=2, implementation="torch")
        g = torch.Generator().manual_seed(L * 131 + log2T)
        x = torch.rand(4096, 3, generator=g)
        x[:64] = torch.randint(0, 17, (64, 3), generator=g).float() / 16.0     # exact lattice points: ceil == floor
        sc = O.hash_scalings(L, lo, hi)
        out[f"scalings_equal_L{L}_{lo}_{hi}"] = bool(torch.equal(sc, enc.scalings.float().reshape(-1)))
        table = enc.hash_table.detach().float()
        got = O.hash_encode(x, table, sc, log2T)
        want = enc.pytorch_fwd(x).detach()
        out[f"features_L{L}_{lo}_{hi}_T{log2T}"] = _maxdiff(got, want)
        idx, _ = O.hash_indices(x, sc, log2T)
        scaled = x[..., None, :] * enc.scalings.view(-1, 1)
        up = enc.hash_fn(torch.ceil(scaled).type(torch.int32))            # the 'ccc' corner
        out[f"hash_fn_ccc_equal_L{L}_T{log2T}"] = bool(torch.equal(idx[..., 0], up.long()))
    return out


@row("L0.2 MLP(implementation='torch')", ["nerfstudio"])
def _mlp():
    from nerfstudio.field_components.mlp import MLP
    from torch import nn
    from oracle import nerf_oracle as O
    out = {}
    for (i, h, nl, o, act) in ((32, 64, 2, 16, None), (63, 64, 3, 3, "sigmoid"), (10, 16, 2, 1, None)):
        m = MLP(in_dim=i, num_layers=nl, layer_width=h, out_dim=o, activation=nn.ReLU(),
                out_activation=nn.Sigmoid() if act else None, implementation="torch")
        x = torch.randn(777, i, generator=torch.Generator().manual_seed(i))
        ws = [l.weight.detach() for l in m.layers]
        bs = [l.bias.detach() for l in m.layers]
        out[f"{i}-{h}x{nl - 1}-{o}"] = _maxdiff(O.mlp_forward(x, ws, bs, out_activation=act), m(x).detach())
    return out


@row("L0.3 SHEncoding(levels=4, torch)", ["nerfstudio"])
def _sh():
    from nerfstudio.field_components.encodings import SHEncoding
    from oracle import nerf_oracle as O
    d = torch.nn.functional.normalize(torch.randn(5000, 3, generator=torch.Generator().manual_seed(3)), dim=-1)
    enc = SHEncoding(levels=4, implementation="torch")
    return {"unit_directions": _maxdiff(O.sh16(d), enc(d).detach()),
            "as_the_torch_fields_feed_it_(d+1)/2": _maxdiff(O.sh16((d + 1) / 2), enc((d + 1) / 2).detach())}


@row("L0.4 SceneContraction(order=inf)", ["nerfstudio"])
def _contract():
    from nerfstudio.field_components.spatial_distortions import SceneContraction
    from oracle import nerf_oracle as O
    g = torch.Generator().manual_seed(4)
    x = torch.cat([torch.randn(4000, 3, generator=g) * 3, torch.rand(1000, 3, generator=g) * 2 - 1,
                   torch.tensor([[1.0, 0, 0], [0, -1.0, 0.5], [0, 0, 0], [1e6, -3, 2]])])
    return _maxdiff(O.contract_inf(x), SceneContraction(order=float("inf"))(x))


@row("L0.5 trunc_exp", ["nerfstudio"])
def _texp():
    from nerfstudio.field_components.activations import trunc_exp
    x = torch.linspace(-30, 20, 2001)
    return _maxdiff(torch.exp(x), trunc_exp(x))


@row("L0.6 spacing functions, initial bins, PDFSampler (eval branch)", ["nerfstudio"])
def _samplers():
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.model_components.ray_samplers import PDFSampler, UniformLinDispPiecewiseSampler
    from oracle import nerf_oracle as O
    out = {}
    R, n0, n1 = 512, 256, 96
    g = torch.Generator().manual_seed(6)
    o = torch.randn(R, 3, generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    near, far = 0.05, 1000.0
    bundle = RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1), nears=torch.full((R, 1), near), fars=torch.full((R, 1), far))
    s0 = UniformLinDispPiecewiseSampler(num_samples=n0, single_jitter=False, train_stratified=False)
    s0.eval()
    rs = s0(bundle)
    sb = O.initial_spacing_bins(n0).expand(R, n0 + 1)
    eb = O.spacing_to_euclidean(sb, near, far)
    out["initial_spacing_bins"] = _maxdiff(sb, torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], -1))
    out["initial_euclidean_edges"] = _maxdiff(eb, torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[:, -1:, 0]], -1))
    w = torch.rand(R, n0, generator=g) ** 8
    w[:7] = 0.0                                # empty rays: the padding branch
    w = w / w.sum(-1, keepdim=True).clamp_min(1e-9) * torch.rand(R, 1, generator=g)
    pdf = PDFSampler(num_samples=n1, single_jitter=False, train_stratified=False, include_original=False)
    pdf.eval()
    rs1 = pdf(bundle, rs, w[..., None])
    nb = O.pdf_resample(w, sb, n1)
    out["pdf_spacing_bins"] = _maxdiff(nb, torch.cat([rs1.spacing_starts[..., 0], rs1.spacing_ends[:, -1:, 0]], -1))
    out["pdf_euclidean_edges"] = _maxdiff(O.spacing_to_euclidean(nb, near, far),
                                          torch.cat([rs1.frustums.starts[..., 0], rs1.frustums.ends[:, -1:, 0]], -1))
    return out


@row("L0.7 RaySamples.get_weights", ["nerfstudio"])
def _weights():
    from nerfstudio.cameras.rays import Frustums, RaySamples
    from oracle import nerf_oracle as O
    g = torch.Generator().manual_seed(7)
    R, S = 300, 48
    edges = torch.cumsum(torch.rand(R, S + 1, generator=g) * 0.2, -1)
    dens = torch.exp(torch.randn(R, S, generator=g) * 3)
    dens[0, 5] = float("inf")
    fr = Frustums(origins=torch.zeros(R, S, 3), directions=torch.ones(R, S, 3), starts=edges[:, :-1, None], ends=edges[:, 1:, None],
                  pixel_area=torch.ones(R, S, 1))
    rs = RaySamples(frustums=fr, deltas=(edges[:, 1:] - edges[:, :-1])[..., None])
    return _maxdiff(O.get_weights(dens, edges[:, 1:] - edges[:, :-1]), rs.get_weights(dens[..., None])[..., 0])


@row("L0.8 renderers (RGB x 4 backgrounds, accumulation, median / expected depth, uncertainty)", ["nerfstudio"])
def _renderers():
    from nerfstudio.cameras.rays import Frustums, RaySamples
    from nerfstudio.model_components import renderers as RN
    from oracle import nerf_oracle as O
    g = torch.Generator().manual_seed(8)
    R, S = 400, 48
    edges = torch.cumsum(torch.rand(R, S + 1, generator=g) * 0.2, -1)
    w = torch.rand(R, S, generator=g) ** 6
    w = w / w.sum(-1, keepdim=True) * torch.rand(R, 1, generator=g)
    rgb = torch.rand(R, S, 3, generator=g)
    rgb[3, 4, 1] = float("nan")
    beta = torch.rand(R, S, generator=g)
    fr = Frustums(origins=torch.zeros(R, S, 3), directions=torch.ones(R, S, 3), starts=edges[:, :-1, None], ends=edges[:, 1:, None],
                  pixel_area=torch.ones(R, S, 1))
    rs = RaySamples(frustums=fr)
    steps = (edges[:, :-1] + edges[:, 1:]) / 2
    out = {}
    for bg in ("last_sample", "random", "white", "black"):
        r = RN.RGBRenderer(background_color=bg)
        r.eval()
        out[f"rgb_{bg}"] = _maxdiff(O.render_rgb(rgb, w, bg), r(rgb=rgb, weights=w[..., None]))
    out["accumulation"] = _maxdiff(O.render_accumulation(w), RN.AccumulationRenderer()(weights=w[..., None]))
    out["depth_median"] = _maxdiff(O.render_depth_median(w, steps), RN.DepthRenderer(method="median")(weights=w[..., None], ray_samples=rs))
    out["depth_expected"] = _maxdiff(O.render_depth_expected(w, steps), RN.DepthRenderer(method="expected")(weights=w[..., None], ray_samples=rs))
    out["uncertainty"] = _maxdiff(O.render_uncertainty(beta, w), RN.UncertaintyRenderer()(betas=beta[..., None], weights=w[..., None]))
    return out


@row("L0.9 Cameras.generate_rays (perspective plain / OPENCV-distorted; fisheye, equirectangular, orthophoto)", ["nerfstudio"])
def _rays():
    from nerfstudio.cameras.cameras import Cameras, CameraType
    from oracle import nerf_oracle as O
    from uncertainty_nerf_gs_amd import synthetic
    out = {}
    H, W = 60, 80
    c2w = synthetic.orbit_c2w(0.7)
    for name, dist in (("plain", None), ("opencv", [0.08, -0.03, 0.004, 0.0, 0.001, -0.002])):
        cam = Cameras(camera_to_worlds=c2w[:3, :4][None], fx=70.0, fy=68.0, cx=W / 2 - 0.7, cy=H / 2 + 0.4, width=W, height=H,
                      distortion_params=torch.tensor(dist)[None] if dist else None, camera_type=CameraType.PERSPECTIVE)
        rb = cam.generate_rays(camera_indices=0, keep_shape=True)
        o, d, pa = O.generate_rays(c2w, 70.0, 68.0, W / 2 - 0.7, H / 2 + 0.4, H, W, distortion=dist)
        out[f"{name}_origins"] = _maxdiff(o, rb.origins)
        out[f"{name}_directions"] = _maxdiff(d, rb.directions)
        out[f"{name}_pixel_area"] = _maxdiff(pa, rb.pixel_area)
    # the other camera models the ray kernel restates (round 6): FISHEYE with OPENCV_FISHEYE's k1..k4, EQUIRECTANGULAR
    # (fx = fy = H = W / 2), ORTHOPHOTO
    for name, ct, dist, (h, w, f) in (("fisheye", CameraType.FISHEYE, [0.05, -0.01, 0.002, -0.0004, 0.0, 0.0], (48, 64, 25.0)),
                                      ("equirectangular", CameraType.EQUIRECTANGULAR, None, (32, 64, 32.0)),
                                      ("orthophoto", CameraType.ORTHOPHOTO, None, (48, 64, 40.0))):
        cam = Cameras(camera_to_worlds=c2w[:3, :4][None], fx=f, fy=f, cx=w / 2, cy=h / 2, width=w, height=h,
                      distortion_params=torch.tensor(dist)[None] if dist else None, camera_type=ct)
        rb = cam.generate_rays(camera_indices=0, keep_shape=True)
        o, d, pa = O.generate_rays(c2w, f, f, w / 2, h / 2, h, w, distortion=dist, camera_type=int(ct.value))
        out[f"{name}_origins"] = _maxdiff(o, rb.origins)
        out[f"{name}_directions"] = _maxdiff(d, rb.directions)
        out[f"{name}_pixel_area"] = _maxdiff(pa, rb.pixel_area)
    return out


def _splat_inputs(n=4000, seed=7):
    from uncertainty_nerf_gs_amd import synthetic
    gp = synthetic.make_splat_tensors(seed, n)
    c2w = synthetic.orbit_c2w(0.9, radius=2.5, height=0.5)
    return gp, c2w


@row("L0.10 gsplat project_gaussians / spherical_harmonics / rasterize_gaussians", ["gsplat"])
def _gsplat():
    if not torch.cuda.is_available():
        return {"skipped": "gsplat's ops are CUDA-only and no device is visible"}
    import gsplat
    from gsplat.sh import spherical_harmonics
    from oracle import splat_oracle as SO
    dev = torch.device("cuda")
    gp, c2w = _splat_inputs()
    H, W, fx, fy = 96, 128, 110.0, 108.0
    cx, cy = W / 2, H / 2
    vm = torch.from_numpy(SO.viewmat_from_c2w(c2w.numpy())).float()
    means, scales, quats = gp["means"], torch.exp(gp["scales"]), gp["quats"] / gp["quats"].norm(dim=-1, keepdim=True)
    res = gsplat.project_gaussians(means.to(dev), scales.to(dev), 1.0, quats.to(dev), vm[:3].to(dev), fx, fy, cx, cy, H, W, 16)
    xys, depths, radii, conics, comp, tiles, _ = res
    ref = SO.project_gaussians(means.numpy(), scales.numpy(), 1.0, quats.numpy(), vm.numpy(), fx, fy, cx, cy, H, W, 16)
    out = {"radii_equal": bool(np.array_equal(ref["radii"], radii.cpu().numpy())),
           "num_tiles_hit_equal": bool(np.array_equal(ref["num_tiles_hit"], tiles.cpu().numpy()))}
    vis = ref["radii"] > 0
    for k, t in (("xys", xys), ("depths", depths), ("conics", conics), ("compensation", comp)):
        out[k] = _maxdiff(ref[k][vis], t.cpu().numpy()[vis])
    dirs = torch.nn.functional.normalize(means - c2w[:3, 3], dim=-1)
    sh = gp["sh"] if "sh" in gp else torch.cat([gp["features_dc"][:, None], gp["features_rest"]], 1)
    for deg in (0, 1, 2, 3):
        out[f"sh_degree_{deg}"] = _maxdiff(SO.spherical_harmonics(deg, dirs.numpy(), sh.numpy()),
                                           spherical_harmonics(deg, dirs.to(dev), sh.to(dev)).cpu())
    colors = torch.rand(means.shape[0], 3, generator=torch.Generator().manual_seed(1))
    opac = torch.sigmoid(gp["opacities"]).reshape(-1, 1)
    bg = torch.tensor([0.1, 0.2, 0.3])
    img, alpha = gsplat.rasterize_gaussians(xys, depths, radii, conics, tiles, colors.to(dev), opac.to(dev), H, W, 16,
                                            background=bg.to(dev), return_alpha=True)
    n_isect, _, _, gids, bins = SO.bin_and_sort(ref["xys"], ref["depths"], ref["radii"], ref["num_tiles_hit"], H, W)
    ro = SO.rasterize(gids, bins, ref["xys"], ref["conics"], colors.numpy(), opac.numpy()[:, 0], H, W, background=bg.numpy())
    out["raster_rgb"] = _maxdiff(ro[0], img.cpu())
    out["raster_alpha"] = _maxdiff(1.0 - ro[1], alpha.cpu())
    out["intersections"] = int(n_isect)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None, help="write the report here as well")
    args = ap.parse_args()
    have = {}
    for pkg in ("nerfstudio", "gsplat", "tinycudann"):
        mod, why = _try_import(pkg)
        have[pkg] = {"present": mod is not None, "version": getattr(mod, "__version__", None) if mod else None, "why": why}
    report = {"packages": have, "rows": {}}
    for name, needs, fn in ROWS:
        missing = [p for p in needs if not have[p]["present"]]
        if missing:
            report["rows"][name] = {"status": "upstream absent", "missing": missing}
            continue
        try:
            with torch.no_grad():
                report["rows"][name] = {"status": "compared", "result": fn()}
        except Exception as e:  # noqa: BLE001 -- one moved API must not hide the other rows
            report["rows"][name] = {"status": "error", "error": f"{type(e).__name__}: {e}"}
    for name, r in report["rows"].items():
        print(f"{name}: {r['status']}" + (f" ({', '.join(r['missing'])})" if r["status"] == "upstream absent" else ""))
        if r["status"] == "compared":
            for k, v in r["result"].items():
                print(f"    {k}: {v}")
        elif r["status"] == "error":
            print(f"    {r['error']}")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
