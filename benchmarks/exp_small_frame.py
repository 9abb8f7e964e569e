"""Experiment: fixed cost of the persistent matrix field kernel on small ray counts (200x200 lego config)."""
import sys, time, torch
sys.path.insert(0, ".")
from uncertainty_nerf_gs_amd import ops, render, synthetic
dev = torch.device("cuda:0")
t = synthetic.make_scene_tensors(seed=0, kind="active")
sd = synthetic.scene_to_device(t, dev)
cam = synthetic.CAMERA_1080P
o, d, _ = ops.generate_rays(synthetic.orbit_c2w(0.3), cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["H"], cam["W"], dev, 900000, 1 << 17)
sb, _ = render.sample_rays(sd, o, d, None, want_prop_depth=False)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for R in (256, 1024, 4096, 16384, 40000, 65536, 131072):
    a = (o[:R].contiguous(), d[:R].contiguous(), sb[:R].contiguous())
    print("R=%6d field_fwd %.3f ms  (%.1f ns/ray)" % (R, timeit(lambda: ops.field_fwd(*a, sd.field, sd.near, sd.far)), 0))
