"""Fixtures computed by the CPU ORACLE (oracle/nerf_oracle.py), not by the reference: cases where the oracle needs
minutes on the host, so the `-m gpu` tests compare against stored oracle outputs instead of re-running it.
These are self-consistent vectors (the oracle's restatement of the nerfstudio path, see its header), unlike the
reference-generated files of make_golden.py.  Needs neither /root/reference nor a GPU:

    python tests/golden/make_oracle_fixtures.py

lego200_mcdropout.npz -- BASELINE.json configs[0]: Blender-lego-shaped 200x200 single view (fx = fy = 277.78,
camera on a radius-4.03 orbit in scene units scaled by 0.33), nerfacto-mcdropout with the torch-layout field,
K = 8 dropout passes (the BASELINE K; seed 7, p = 0.2), reference chunking 32768 rays.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LEGO = dict(seed=1, log2T=15, prop_log2T=13, theta=0.9, radius=4.03 * 0.33, height=0.4, K=8, mc_seed=7, p_drop=0.2)


def lego200():
    import conftest  # noqa: F401  (registers the package alias)
    from oracle import nerf_oracle as O
    from uncertainty_nerf_gs_amd import synthetic
    c = LEGO
    t = synthetic.make_scene_tensors(seed=c["seed"], kind="mcdropout", log2T=c["log2T"], prop_log2T=c["prop_log2T"])
    sc = O.scene_from_tensors(t)
    cam = dict(synthetic.CAMERA_LEGO200)
    c2w = synthetic.orbit_c2w(c["theta"], radius=c["radius"], height=c["height"])
    o, d, _ = O.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["H"], cam["W"])
    diag = {}
    ref = O.render_camera(lambda oo, dd, off: O.mcdropout_outputs(sc, oo, dd, c["K"], c["mc_seed"], c["p_drop"], ray_offset=off,
                                                                  diagnostics=diag), o, d, chunk=32768)
    # test diagnostic, not an output: the smallest |cdf - 0.5| of the K passes per ray (oracle.median_margin) -- the depth
    # gate treats rays whose median differs as ties only when this is within rounding
    ref["median_margin"] = torch.cat(diag["median_margin"]).view(cam["H"], cam["W"], 1)
    np.savez_compressed(os.path.join(HERE, "lego200_mcdropout.npz"), **{k: v.numpy().astype(np.float32) for k, v in ref.items()})
    print({k: tuple(v.shape) for k, v in ref.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    lego200()
