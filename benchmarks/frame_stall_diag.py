"""Which host call blocks when a frame stalls?  Wraps every callable of ops / torch.empty / torch.zeros with a host timer."""
import math, os, sys, time, socket, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa
from uncertainty_nerf_gs_amd import ops, render, synthetic, lib

dev = torch.device("cuda:0")
t = synthetic.make_scene_tensors(seed=0, kind="mcdropout")
scene = synthetic.scene_to_device(t, dev, K=8, seed=1234, p_drop=0.2)
scene.field.precision = "f16"
cam = dict(synthetic.CAMERA_1080P)
poses = [synthetic.orbit_c2w(2 * math.pi * i / 24) for i in range(24)]
log = []
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); dt = time.perf_counter() - t0
        log.append((name, dt)); return r
    setattr(mod, name, g)
for n in dir(ops):
    if not n.startswith("_") and callable(getattr(ops, n)) and not isinstance(getattr(ops, n), type):
        wrap(ops, n)
for n in ("empty", "zeros", "empty_like", "zeros_like", "cat", "stack"):
    wrap(torch, n)
rows = []
for i in range(12):
    log.clear()
    t0 = time.perf_counter()
    out = render.render_camera(scene, poses[i % 24], depth_seed=7, **cam)
    dt = (time.perf_counter() - t0) * 1e3
    top = sorted(log, key=lambda x: -x[1])[:3]
    rows.append((round(dt, 1), [(n, round(d * 1e3, 1)) for n, d in top], round(torch.cuda.memory_reserved() / 2**30, 2)))
torch.cuda.synchronize()
print(socket.gethostname())
for r in rows: print(r)
