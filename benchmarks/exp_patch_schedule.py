"""Experiment: linear vs pixel-patch schedule of the proposal kernel and the field kernel on one 1080p launch
group (2^18 rays = 136.5 image rows).  Run on the GPU box: python benchmarks/exp_patch_schedule.py"""
import sys, time, torch
sys.path.insert(0, ".")
from uncertainty_nerf_gs_amd import ops, render, synthetic

dev = torch.device("cuda:0")
t = synthetic.make_scene_tensors(seed=0, kind="active")
sd = synthetic.scene_to_device(t, dev)
cam = synthetic.CAMERA_1080P
W = cam["W"]
R, start = 1 << 18, 3 << 18
o, d, _ = ops.generate_rays(synthetic.orbit_c2w(0.3), cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["H"], W, dev, start, R)
sb0 = sd.const("bins", 256)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


dens0 = ops.proposal_density(o, d, sb0, sd.props[0], sd.near, sd.far, 0.01)
sb1, _, _ = ops.weights_pdf_resample(dens0, sb0, sd.const("u", 96), sd.near, sd.far)
dens1 = ops.proposal_density(o, d, sb1, sd.props[1], sd.near, sd.far, 0.01)
sb2, _, _ = ops.weights_pdf_resample(dens1, sb1, sd.const("u", 48), sd.near, sd.far)
for name, fn in (
    ("prop_256 linear", lambda: ops.proposal_density(o, d, sb0, sd.props[0], sd.near, sd.far, 0.01)),
    ("prop_256 patch ", lambda: ops.proposal_density(o, d, sb0, sd.props[0], sd.near, sd.far, 0.01, ray_offset=start, image_width=W)),
    ("prop_96  linear", lambda: ops.proposal_density(o, d, sb1, sd.props[1], sd.near, sd.far, 0.01)),
    ("prop_96  patch ", lambda: ops.proposal_density(o, d, sb1, sd.props[1], sd.near, sd.far, 0.01, ray_offset=start, image_width=W)),
    ("field    linear", lambda: ops.field_fwd(o, d, sb2, sd.field, sd.near, sd.far, start)),
    ("field    patch ", lambda: ops.field_fwd(o, d, sb2, sd.field, sd.near, sd.far, start, image_width=W)),
):
    print(name, "%.3f ms per 2^18 rays" % timeit(fn))
