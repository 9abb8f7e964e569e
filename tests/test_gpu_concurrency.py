"""include/unerf.h promises that the entry points may be called concurrently from several host threads on different
streams (no hidden synchronisation, no global mutable state except a mutex-guarded attribute cache, a thread-local error
string; SURVEY.md:353).  Here two -- and four -- host threads render DIFFERENT scenes on their own HIP streams at the same
time, every kernel of both frame paths in flight against the other thread's; each thread's frames must equal, bit for bit,
the frames the same scene renders alone.  (VERDICT r5 "weak" 11 / "next" 8.)"""
import math
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _nerf_job(dev, kind, precision, seed, H, W, angle):
    from uncertainty_nerf_gs_amd import render, synthetic
    t = synthetic.make_scene_tensors(seed=seed, kind=kind, log2T=15, prop_log2T=13)
    kw = dict(K=8, seed=1234 + seed, p_drop=0.2) if kind == "mcdropout" else {}
    if kind == "laplace":
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=32)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
    sd = synthetic.scene_to_device(t, dev, **kw)
    sd.field.precision = precision
    cam = dict(fx=0.9 * W, fy=0.9 * W, cx=W / 2, cy=H / 2, H=H, W=W)
    c2w = synthetic.orbit_c2w(angle)

    def run():
        # several launch groups per frame, so that the two threads' kernels interleave for the whole frame
        out = render.render_camera(sd, c2w, rays_per_launch=32768, depth_seed=7, **cam)
        return {k: v.clone() for k, v in out.items()}
    return run


def _splat_job(dev, seed, n, H, W, angle):
    from uncertainty_nerf_gs_amd import splat, synthetic
    gp = {k: v.to(dev) for k, v in synthetic.make_splat_tensors(seed=seed, N=n).items()}
    pose = synthetic.orbit_c2w(angle, radius=2.5, height=0.5).to(dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    cam = dict(fx=0.6 * W, fy=0.6 * W, cx=W / 2, cy=H / 2, H=H, W=W)

    def run():
        out = splat.active_splatfacto_outputs(gp, pose, background=bg, **cam)
        return {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
    return run


def _run_concurrently(dev, jobs, frames):
    """every job renders `frames` frames on a stream of its own, all threads released together"""
    results, errors = [None] * len(jobs), []
    gate = threading.Barrier(len(jobs))

    def worker(i, job):
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                gate.wait(timeout=60)
                outs = [job() for _ in range(frames)]
                st.synchronize()
            results[i] = outs
        except BaseException as e:   # noqa: BLE001 -- reported below, in the test's thread
            errors.append((i, repr(e)))
            try:
                gate.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(i, j)) for i, j in enumerate(jobs)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors
    assert all(r is not None for r in results)
    return results


def _assert_equal(name, alone, together):
    for f, frame in enumerate(together):
        assert set(frame) == set(alone), (name, set(frame) ^ set(alone))
        for k in alone:
            n = int((alone[k] != frame[k]).sum()) if alone[k].dtype.is_floating_point else int((alone[k] != frame[k]).sum())
            # NaN-free outputs: != counts real differences
            assert n == 0, f"{name}, concurrent frame {f}: {n} values of `{k}` differ from the scene rendered alone"


def test_two_threads_two_streams_render_the_frames_they_render_alone(dev):
    jobs = [_nerf_job(dev, "mcdropout", "f16", 1, 120, 160, 0.7), _nerf_job(dev, "active", "f16x2", 2, 96, 200, 2.1)]
    alone = [j() for j in jobs]
    torch.cuda.synchronize()
    together = _run_concurrently(dev, jobs, frames=6)
    for name, a, t in zip(("mcdropout f16", "active f16x2"), alone, together):
        _assert_equal(name, a, t)


def test_four_threads_nerf_and_splat_paths_at_once(dev):
    """all three NeRF methods and the splat frame (its count read-back goes through a ring of pinned words and one side
    stream per device: ops.SplatCount, slot allocation under a lock) in four threads"""
    jobs = [_nerf_job(dev, "mcdropout", "f16x2", 3, 64, 96, 0.3), _nerf_job(dev, "laplace", "f16x2", 4, 64, 80, 1.3),
            _splat_job(dev, 7, 60000, 200, 320, 2 * math.pi * 5 / 24), _splat_job(dev, 8, 45000, 180, 240, 0.9)]
    names = ("mcdropout f16x2", "laplace f16x2", "splat A", "splat B")
    alone = [j() for j in jobs]
    torch.cuda.synchronize()
    together = _run_concurrently(dev, jobs, frames=5)
    for name, a, t in zip(names, alone, together):
        _assert_equal(name, a, t)

