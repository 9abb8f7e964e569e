#!/usr/bin/env python
"""bench.py -- throughput of the uncertainty-rendering hot path on MI355X.

A "step" is one full 1920x1080 frame (2,073,600 rays, Mip-NeRF360-garden-shaped synthetic
camera orbit) through the whole path: ray generation -> proposal sampling (256 -> 96 -> 48
samples, two hash-MLP density nets + PDF resampling) -> main field (16-level hash grid + fused
MLPs with the method's uncertainty head) -> front-to-back composite with variance
[-> per-pixel mean/std over the K MC passes].  Inputs (weights, tables, camera) are resident in
HBM before the timed region; synthetic seeded data, random-init weights of the nerfacto shape.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--method active|mcdropout|laplace|splat|ensemble] [--mc-samples 8]

Without --method the headline is the north-star target config (BASELINE.json configs[2]): nerfacto-mcdropout with
K = 8 fused passes; active-nerfacto, nerfacto-laplace and active-splatfacto follow as `sub_records` of the same line.

N>1 runs one rank per GPU under torch.distributed.run: either the caller launches it that way, or plain
`python bench.py --gpus N` starts `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` itself as
a child process (before this process has touched the GPU) and relays rank 0's JSON line and the child's exit code.
Rays are independent, so ranks render different cameras of the orbit with replicated weights and no data-path
collective ("weak" scaling); the barrier + max-over-ranks timing uses RCCL.  The default run also carries
`sub_records.ensemble`: the 8-member nerfacto ensemble of BASELINE.json configs[3], members sharded over the ranks
(8 / N per GPU), per-pixel moments through ONE all_to_all + ONE all_gather over RCCL -- strong scaling, the north-star's
">= 6x at 8 GPUs" figure is value(N=8) / value(N=1) of that record.
Prints ONE JSON line on rank 0 (contract in the task statement), carrying `roofline` for the
dominant kernel and `cpu_baseline` (the CPU oracle timed on a bounded sample of the same rays).
"""
import argparse
import gc
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3       # fp32 vector = fp32-input MFMA peak (the exact-fp32 field kernels, --exact-fp32)
F16_PEAK_TFLOPS = 2500.0       # dense f16/bf16 MFMA peak (the split-f16 field kernels issue 3 f16 products per fp32 MAC)

# algorithmic work per ray (SURVEY.md 8d / DESIGN.md "Kernels"): bytes the algorithm must touch
def _alg(kind, K, row_bytes=8):
    S0, S1, S = 256, 96, 48
    corner = 8 * row_bytes  # 8 corners x one table row: 2 x fp32, or half2 (tcnn grids in tcnn's half arithmetic)
    passes = max(K, 1)
    mlp_shared = 2 * 32 * 64
    mlp_tail = {"active": 2 * 64 * 17, "mcdropout": 2 * 64 * 16, "laplace": 2 * 64 * 15}[kind] + 2 * 63 * 64 + 2 * 64 * 64
    last = 2 * 64 * 3 if kind != "laplace" else 100 * (2 * 64 * 1 + 2 * 64 * 3)
    return {
        "proposal_density_256": {"bytes": S0 * 5 * corner + 24 + S0 * 4, "flops": S0 * (2 * 10 * 16 + 2 * 16)},
        "proposal_density_96": {"bytes": S1 * 5 * corner + 24 + (S1 + 1) * 4 + S1 * 4, "flops": S1 * (2 * 10 * 16 + 2 * 16)},
        "weights_pdf_resample_256": {"bytes": S0 * 4 + (S1 + 1) * 4 + 4, "flops": 0},
        "weights_pdf_resample_96": {"bytes": S1 * 4 + (S1 + 1) * 4 + (S + 1) * 4 + 4, "flops": 0},
        "field_fwd": {"bytes": S * 16 * corner + 24 + (S + 1) * 4 + passes * S * 16 + S * 8,
                      "flops": S * (mlp_shared + passes * (mlp_tail + last))},
        "composite_var": {"bytes": passes * (S * 16 + 32) + (S + 1) * 4 + S * 4, "flops": 0},
        "moments": {"bytes": passes * 24 + 48, "flops": 0},
        "laplace_depth_weights": {"bytes": S * 12 + (S + 1) * 4, "flops": 0},
        "generate_rays": {"bytes": 24, "flops": 0},
        "field_gather": {"bytes": S * 16 * corner + 24 + (S + 1) * 4 + S * 128, "flops": 0},
    }


# The arithmetic the REFERENCE runs each method's Linear layers in at eval: MC-dropout forces torch.autocast(float16)
# (mcdropout_models.py:86-92); active-nerfacto's MLPs are tiny-cuda-nn FullyFusedMLPs, fp16 throughout
# (activenerfacto_field.py:89, upstream's default implementation="tcnn"); the Laplace field calls `.float()` in front of
# every Linear (laplace_field.py:305, :460).  "f16" = one f16 product per MAC, fp32 accumulate -- no narrower than either.
REFERENCE_PRECISION = {"mcdropout": "f16", "active": "f16", "laplace": "f16x2"}
DTYPE_OF = {"f16": "f16 operands, f32 accumulate", "f16x2": "f32 (split-f16 operands, f32-equivalent)", "fp32": "f32"}
STALE_PROFILES = []            # issue / traffic profiles under profiles/ that were taken from other kernel sources
# shader cycles per wave64 instruction on one SIMD, measured by benchmarks/issue_sweep_probe.hip with s_memtime
# (profiles/r6_issue_sweep.jsonl): a VALU instruction of the packed / converting / VOP3 kind 4.0 - 4.2 at 1 - 3 waves per
# SIMD (plain VOP2 add / mul / and on registers: 2.1 at two waves, 1.4 at three); a v_mfma_f32_32x32x16_f16 holds the issue
# for ~10 cycles and the matrix pipe for 32, and up to six VALU instructions ride in the remaining 22 for 0.25 - 0.5 each
ISSUE_CYC_VALU, ISSUE_CYC_MFMA, MATRIX_CYC_MFMA = 4.0, 10.0, 32.0
ISSUE_PEAK_GCYC = 1024 * 2.4   # 256 CUs x 4 SIMDs x 2.4 GHz peak engine clock: issue cycles per nanosecond x 1e9


def _issue_profile(method, K):
    """Instruction-issue cycles of the field kernel per launch, from the committed rocprofv3 --pmc pass of this
    command (profiles/issue_<method>.json, written by benchmarks/summarize_pmc.py): the exact instruction counts
    SQ_INSTS_VALU / SQ_INSTS_MFMA.  They are priced in run_nerf with the constants benchmarks/issue_sweep_probe.hip measured
    (ISSUE_CYC_VALU, ISSUE_CYC_MFMA, MATRIX_CYC_MFMA): the issue lane and the matrix lane of a SIMD run side by side, the
    launch needs the larger of the two -- a property of the code, not of the clock it ran at."""
    f = os.path.join(ROOT, "profiles", f"issue_{method}.json")
    if not os.path.exists(f):
        return None
    j = json.load(open(f))
    if j.get("K", 0) != K:
        return None
    # the instruction counts describe ONE build of the kernels: a profile taken from other sources is not used (the
    # roofline then falls back to the matrix-pipe / algorithmic-gather figures and says so)
    from uncertainty_nerf_gs_amd import lib
    if j.get("kernel_source_digest") not in (None, lib._source_digest()):
        STALE_PROFILES.append(os.path.basename(f))
        return None
    return j


# _timed_region_gc: every timed loop below is bracketed by gc.freeze() / gc.unfreeze().  CPython's full (generation-2)
# collection walks every tracked object of the process -- after `import torch` that is 15 - 120 ms during which the host
# issues nothing -- and WHEN it runs depends on allocation counts: frame 42 of a 400-frame soak took 60 - 91 ms instead of 45
# on every run, frame 1 of the five default steps 163 ms on a box that had parsed more JSON before (step_wall_ms of those runs;
# with the long-lived objects frozen, or the collector off, no frame of 2 x 400 exceeded 47.1 ms).  freeze() moves what exists
# after the warm-up into the permanent generation: the collector keeps running, on the young objects only.  No work is skipped.
def run_nerf(args, method, K, steps, warmup, rank, world, dev, dist, exact_check=True, want_cpu=True, precision=None,
             grid=None, grid_precision=None):
    """One NeRF method through the whole frame path; returns the record (headline fields + roofline + cpu_baseline).
    grid / grid_precision (default: args): "tcnn" = the tiny-cuda-nn table layout the reference's default
    implementation trains with, "f16" = in tcnn's own half arithmetic (half2 rows)."""
    from uncertainty_nerf_gs_amd import ops, render, synthetic
    grid = grid or args.grid
    grid_precision = grid_precision or (args.grid_precision or ("f16" if grid == "tcnn" else "f32"))
    t = synthetic.make_scene_tensors(seed=0, kind=method, grid=grid)   # full nerfacto shape: 16x2^19x2 + 2 x 5x2^17x2
    if grid == "tcnn":
        t["grid_precision"] = grid_precision
    row_bytes = 4 if (grid == "tcnn" and grid_precision == "f16") else 8
    grid_tag = "" if grid == "torch" else ("_tcnn" if grid_precision == "f16" else "_tcnn32")
    kw = {}
    if method == "mcdropout":
        kw = dict(K=K, seed=1234, p_drop=0.2)
    if method == "laplace":
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=100)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
    scene = synthetic.scene_to_device(t, dev, **kw)
    scene.field.precision = precision or ("fp32" if (args.exact_fp32 or args.split_gather) else
                                          (args.precision or REFERENCE_PRECISION[method]))
    scene.split_gather = args.split_gather
    H, W = args.height, args.width
    cam = dict(synthetic.CAMERA_1080P)
    cam.update(H=H, W=W, cx=W / 2, cy=H / 2)
    n_views = 24
    poses = [synthetic.orbit_c2w(2 * math.pi * i / n_views) for i in range(n_views)]
    # the reference's active-nerfacto output dict carries the raw density [H,W,48] too (activenerfacto_model.py:115,122)
    shade_kw = dict(keep_density=True) if method == "active" else {}

    def frame(i):
        c2w = poses[(rank + i * world) % n_views]   # view-batch partition across ranks
        return render.render_camera(scene, c2w, rays_per_launch=args.rays_per_launch, overlap=args.overlap,
                                    depth_seed=7, **cam, **shade_kw)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        out = frame(i)
    sync_all()
    ops.TIMER = ops.KernelTimer(prealloc=64 * steps)
    gc.freeze()   # see _timed_region_gc
    t0 = time.perf_counter()
    marks = []
    for i in range(steps):
        out = frame(warmup + i)
        marks.append(time.perf_counter())   # host time when frame i had been ISSUED (a frame ends with the overflow-flag read)
    sync_all()
    elapsed = time.perf_counter() - t0
    gc.unfreeze()
    step_wall_ms = [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks[:-1], marks)]
    timer = ops.TIMER
    ops.TIMER = None
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out["rgb"]).all()
    mrays = H * W * steps * world / elapsed / 1e6

    # Reported next to the headline, outside its timed region: the same frames with the dense layers in the next WIDER
    # arithmetic (f16 -> split-f16, split-f16 -> exact fp32-input MFMA), and how far the headline image is from that one.
    exact = None
    head_prec = scene.field.precision
    if head_prec in ("f16x2", "f16") and exact_check:
        wider = "fp32" if head_prec == "f16x2" else "f16x2"
        scene.field.precision = wider
        n_alt = max(1, min(steps, 3))
        ref = frame(warmup + steps - 1)          # also warms those kernels up
        sync_all()
        gc.freeze()
        t1 = time.perf_counter()
        for i in range(n_alt):
            frame(warmup + i)
        sync_all()
        alt = time.perf_counter() - t1
        gc.unfreeze()
        if dist is not None:
            tt = torch.tensor([alt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            alt = float(tt.item())
        scene.field.precision = head_prec
        exact = {"precision": wider, "value": H * W * n_alt * world / alt / 1e6, "unit": "Mrays/s", "ms_per_step": alt / n_alt * 1e3,
                 "steps": n_alt, "max_abs_rgb_diff_vs_headline": float((ref["rgb"] - out["rgb"]).abs().max()),
                 "mean_abs_rgb_diff_vs_headline": float((ref["rgb"] - out["rgb"]).abs().mean()),
                 "max_abs_rgb_std_diff_vs_headline": float((ref["rgb_std"] - out["rgb_std"]).abs().max())}
        del ref

    rec = {"value": mrays, "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup, "wider_arithmetic": exact,
           "step_wall_ms": step_wall_ms}
    split = scene.field.precision == "f16x2"
    single = scene.field.precision == "f16"
    rec["workload"] = (f"{method}-nerfacto {W}x{H} render with variance" + (f", K={K} MC-dropout passes" if K else "")
                       + (", 100 last-layer Laplace samples" if method == "laplace" else "")
                       + (", density [H,W,48] kept" if method == "active" else "")
                       + ("" if grid == "torch" else (", tcnn-layout tables in tcnn's half arithmetic (half2 rows)" if row_bytes == 4
                                                      else ", tcnn-layout tables, fp32 rows")))
    rec["hash_grid"] = ("16x2^19x2 fp32 (nerfstudio torch layout)" if grid == "torch" else
                        f"tcnn layout, 16 levels up to 2^19 rows, {'half2' if row_bytes == 4 else 'fp32x2'} rows")
    rec["dense_layers"] = ("fp32 operands split into two f16 halves, 3 products on v_mfma_f32_32x32x16_f16, "
                           "fp32 accumulate (fp32-equivalent, DESIGN.md 4.2)" if split else
                           "f16 operands (one product per MAC on v_mfma_f32_32x32x16_f16), fp32 accumulate: the reference's eval "
                           "precision (forced autocast fp16, mcdropout_models.py:86-92; tcnn FullyFusedMLP)" if single
                           else "exact fp32 (v_mfma_f32_32x32x2_f32)")
    rec["precision"] = scene.field.precision
    if rank != 0:
        return rec
    ksum = timer.summary()
    alg = _alg(method, K, row_bytes)
    dom = max(ksum, key=lambda k: ksum[k]["total_ms"])
    rays_per_launch = H * W * steps / ksum[dom]["launches"]
    avg_s = ksum[dom]["avg_ms"] * 1e-3
    a = alg.get(dom, {"bytes": 0, "flops": 0})
    # Side figures for the dominant kernel: algorithmic gather rate and matrix-pipe rate.  Neither binds: the tables
    # are cache resident (the gather rate can exceed the DRAM peak) and the f16 matrix pipe is a third busy.
    gather_gbs = a["bytes"] * rays_per_launch / avg_s / 1e9
    mfma_ach = a["flops"] * rays_per_launch / avg_s / 1e12
    mfma_peak = F16_PEAK_TFLOPS if (split or single) else FP32_PEAK_TFLOPS
    mfma_issued = 3.0 * mfma_ach if split else mfma_ach
    roof = {"kernel": dom, "avg_launch_ms": ksum[dom]["avg_ms"], "launches": ksum[dom]["launches"],
            "rays_per_launch": rays_per_launch}
    # The contract's roofline: ALGORITHMIC bytes (or flops) of the dominant kernel per launch over its live
    # launch duration, against the HBM peak (or the dense MFMA peak of the operand type).  The hash tables are cache
    # resident, so for the lighter kernels the algorithmic gather rate exceeds what HBM could deliver; a fraction above 1
    # says "not the bound", and the matrix-pipe form is reported instead.
    frac_m, frac_h = mfma_issued / mfma_peak, gather_gbs / HBM_PEAK_GBS
    if frac_h > 1.0:
        roof.update({"bound": "mfma", "achieved": mfma_issued, "peak": mfma_peak, "unit": "TFLOP/s", "frac": frac_m,
                     "note": "matrix-pipe rate on issued flops (3 products per MAC in the split-f16 form, 1 otherwise); the "
                             "algorithmic gather rate of this kernel is above the HBM peak (cache-resident tables, other_roofs), "
                             "so HBM is not its roof.  What binds is instruction issue: issue_roofline"})
    else:
        roof.update({"bound": "hbm", "achieved": gather_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac_h,
                     "note": "algorithmic bytes of the dominant kernel (8 B per hash-grid corner + its streams, "
                             "algorithmic_bytes_per_ray x rays_per_launch) over the live launch duration; `traffic` = the bytes "
                             "that actually cross the fabric (PMC).  The tables are cache resident and the kernel is bound by "
                             "instruction issue, not by this roof: issue_roofline"})
    prof = _issue_profile(method + ("_f16" if single else "") + grid_tag, K) if dom == "field_fwd" and (split or single) else None
    roof["issue_roofline"] = None
    if prof is not None:
        # the roof that binds: instruction issue.  achieved = cycles the launch's instruction stream needs on the busier of
        # a SIMD's two lanes (PMC instruction counts per launch of prof["rays_per_launch"] rays, probe prices) / live launch
        # duration; peak = every SIMD busy every cycle at the peak engine clock.
        sc = rays_per_launch / prof["rays_per_launch"]
        lane_issue = (ISSUE_CYC_VALU * prof["valu_insts_per_launch"] + ISSUE_CYC_MFMA * prof["mfma_insts_per_launch"]) * sc
        lane_matrix = MATRIX_CYC_MFMA * prof["mfma_insts_per_launch"] * sc
        cyc = max(lane_issue, lane_matrix)
        ach = cyc / avg_s / 1e9
        clock = prof.get("engine_clock_GHz_under_profiler")
        roof["issue_roofline"] = {
            "bound": "valu-issue" if lane_issue >= lane_matrix else "matrix-pipe", "achieved": ach, "peak": ISSUE_PEAK_GCYC,
            "unit": "Gcycle/s", "frac": ach / ISSUE_PEAK_GCYC,
            "frac_at_measured_clock": (ach / (1024 * clock)) if clock else None, "engine_clock_GHz_under_profiler": clock,
            "issue_lane_Gcycles_per_launch": lane_issue / 1e9, "matrix_lane_Gcycles_per_launch": lane_matrix / 1e9,
            "prices": {"valu": ISSUE_CYC_VALU, "mfma_issue": ISSUE_CYC_MFMA, "mfma_pipe": MATRIX_CYC_MFMA,
                       "source": "benchmarks/issue_sweep_probe.hip, profiles/r6_issue_sweep.jsonl (s_memtime cycles, 1 - 3 waves per SIMD)"},
            "note": "a SIMD has an issue lane (4 cycles per wave64 VALU instruction, ~10 per v_mfma_f32_32x32x16_f16) and a "
                    "matrix lane (32 per MFMA) that run side by side: up to six VALU instructions per MFMA ride in its shadow. "
                    "achieved = max(lane) of one launch, exact instruction counts from the committed PMC pass, over the live "
                    "launch time; peak = 1024 SIMDs x 2.4 GHz; frac_at_measured_clock divides by the clock the kernel held "
                    "under the profiler instead.  The VALU price is the packed / converting / VOP3 class' (what the pass loop is "
                    "made of); plain VOP2 add / mul / and / xor on registers cost 2.1 at two waves per SIMD, so for their "
                    "share of the stream (~25 % of the K-pass kernel's VALU count) the lane is over-estimated.  Rounds 2 - 5 "
                    "priced 4 x VALU + 32 x MFMA as one lane ('no overlap', 0.79 for this kernel); that probe had 42 issue "
                    "cycles of fillers in every 32-cycle gap.  What the kernel leaves of the roof is stall: the pass is one "
                    "dependent chain (MFMA -> convert -> MFMA) and at two waves per SIMD both waves wait at once for part "
                    "of it (DESIGN.md 4.4; placing the independent mask arithmetic in the MFMA shadows was built and "
                    "measured: +0.6 %, docs/experiments.md 6.2)",
            "issue_source": prof["source"], "valu_insts_per_ray": prof["valu_insts_per_launch"] / prof["rays_per_launch"],
            "mfma_insts_per_ray": prof["mfma_insts_per_launch"] / prof["rays_per_launch"],
            "simd_busy_frac_under_profiler": prof.get("busy_frac")}
    roof["other_roofs"] = {"algorithmic_gather_GBps": gather_gbs,
                           "algorithmic_gather_note": "8 B per hash-grid corner; tables are L2 / Infinity-Cache resident, so "
                                                      "this is a cache-gather rate and may exceed the DRAM peak",
                           "algorithmic_TFLOPs": mfma_ach, "matrix_pipe": "f16 (3 products per MAC)" if split else ("f16 (1 product per MAC)" if single else "fp32"),
                           "issued_matrix_TFLOPs": mfma_issued, "matrix_frac": mfma_issued / mfma_peak}
    roof["algorithmic_bytes_per_ray"] = a["bytes"]
    roof["algorithmic_flops_per_ray"] = a["flops"]
    roof["traffic"] = None
    tfile = os.path.join(ROOT, "profiles", f"traffic_{method}{'_f16' if single else ''}{grid_tag}.json")
    if os.path.exists(tfile):  # HBM-side bytes from committed rocprofv3 --pmc passes of this command
        tj = json.load(open(tfile))
        tk = tj.get("kernels", {}).get(dom)
        from uncertainty_nerf_gs_amd import lib
        if tj.get("kernel_source_digest") not in (None, lib._source_digest()):
            STALE_PROFILES.append(os.path.basename(tfile))
            tk = None
        if tk and K == tj.get("K", 0):
            scale = rays_per_launch / tj["rays_per_launch"]
            roof["traffic"] = (tk["fetch_bytes"] + tk["write_bytes"]) * scale
            roof["traffic_source"] = tj["source"]
            roof["other_roofs"]["hbm_traffic_GBps"] = roof["traffic"] / avg_s / 1e9
            roof["other_roofs"]["hbm_frac"] = roof["traffic"] / avg_s / 1e9 / HBM_PEAK_GBS
    if STALE_PROFILES:
        roof["stale_profiles_ignored"] = sorted(set(STALE_PROFILES))
    roof["per_kernel_ms_per_frame"] = {k: round(v["total_ms"] / steps, 3) for k, v in sorted(ksum.items())}
    path_bytes = sum(alg[k]["bytes"] for k in ksum if k in alg)
    roof["path_bytes_per_ray"] = path_bytes
    rec["roofline"] = roof
    rec["cpu_baseline"] = None
    rec["parity_at_bench_size"] = None
    if want_cpu and not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only (bench contract)
        # the oracle's outputs for its strided chunks are KEPT and compared with the same rays of a GPU frame of the same
        # pose (rendered here, outside every timed region)
        prec = scene.field.precision
        rec["cpu_baseline"], ids, ref = cpu_baseline(t, args, method, poses[0], cam, K)
        refs = {"fp32": ref}
        if prec == "f16":   # the arithmetic the reference computes in (forced autocast fp16 / tcnn): the oracle's emulation of it
            refs["autocast16"] = cpu_baseline(t, args, method, poses[0], cam, K, autocast=torch.float16, ids_only=ids)[2]
        got = render.render_camera(scene, poses[0], rays_per_launch=args.rays_per_launch, depth_seed=7, **cam, **shade_kw)
        par = {"random_init": parity_record(got, ids, refs, prec)}
        if method in ("mcdropout", "active") and not args.no_trained_like_parity:
            # ... and on the TRAINED-LIKE scene (density logits +-12, colour-head activations ~1e3), where f16 is hard: one
            # more frame and two more oracle passes, outside every timed region
            ts = synthetic.make_scene_tensors(seed=0, kind=method, sharp=True)
            sc2 = synthetic.scene_to_device(ts, dev, **kw)
            sc2.field.precision = prec
            got2 = render.render_camera(sc2, poses[0], rays_per_launch=args.rays_per_launch, depth_seed=7, **cam, **shade_kw)
            refs2 = {"fp32": cpu_baseline(ts, args, method, poses[0], cam, K, ids_only=ids)[2]}
            if prec == "f16":
                refs2["autocast16"] = cpu_baseline(ts, args, method, poses[0], cam, K, autocast=torch.float16, ids_only=ids)[2]
            par["trained_like"] = parity_record(got2, ids, refs2, prec)
            par["trained_like"]["overflow_rerenders"] = int(sc2.overflow_rerenders)
            del sc2, got2
        par["overflow_rerenders"] = int(scene.overflow_rerenders)
        par["gates"] = ("|dPSNR| <= 1e-4 dB, |dAUSE| <= 1e-3 on the informative synthetic target (oracle/targets.py), for every "
                        "scene and every reference listed; plain-target numbers are recorded next to the gap of the reference's "
                        "own two arithmetics on that target (profiles/r5_exp_ause_oracle_gap.json)")
        par["inside_gates"] = all(r["inside_gates"] for k, r in par.items() if isinstance(r, dict) and "inside_gates" in r)
        rec["parity_at_bench_size"] = par
    return rec


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: run the same command line under torch.distributed.run (one rank
    per GPU, rendezvous on 127.0.0.1) as a CHILD process -- never an exec of this one, and before anything here has
    initialised the GPU -- stream its output through, and return its exit code.  Rank 0 prints the JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    # --standalone: torchrun's own c10d store on a port IT binds (localhost:0) -- no port probed here and taken by someone
    # else before the store binds it; --local-addr: the ranks' MASTER_ADDR (the container hostname may not resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n_gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--method", default=None, choices=["active", "mcdropout", "laplace", "splat", "ensemble"],
                    help="default: nerfacto-mcdropout K=8 (the north-star target config) as the headline, with active / "
                         "laplace / splat sub-records in the same line")
    ap.add_argument("--members", type=int, default=8, help="ensemble size M (members are sharded over the ranks)")
    ap.add_argument("--splats", type=int, default=1_000_000)
    ap.add_argument("--mc-samples", type=int, default=8)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--rays-per-launch", type=int, default=1 << 20)
    ap.add_argument("--overlap", action="store_true", help="sampling / shading stages on two HIP streams (experiment)")
    ap.add_argument("--split-gather", action="store_true", help="level-major gather kernel + feature planes (experiment)")
    ap.add_argument("--precision", default=None, choices=["f16x2", "f16", "fp32"],
                    help="dense layers: f16 = one f16 product per MAC, fp32 accumulate (the reference's autocast / tcnn "
                         "precision); f16x2 = split-f16 (fp32-equivalent); fp32 = exact fp32-input MFMA.  Default: what the "
                         "reference computes the method in -- f16 for mcdropout (forced autocast) and active (tcnn), f16x2 for "
                         "laplace (`.float()` Linears)")
    ap.add_argument("--grid", default="torch", choices=["torch", "tcnn"],
                    help="table layout: nerfstudio's torch HashEncoding (the BASELINE's synthetic recipe) or tiny-cuda-nn's "
                         "(implementation=\"tcnn\", the reference's default)")
    ap.add_argument("--grid-precision", default=None, choices=["f32", "f16"],
                    help="tcnn layout only: f16 (default) = half2 rows + tcnn's half interpolation, f32 = fp32 rows")
    ap.add_argument("--exact-fp32", action="store_true",
                    help="dense layers on the exact fp32-input MFMA kernels instead of the split-f16 ones")
    ap.add_argument("--no-exact-check", action="store_true",
                    help="skip the extra exact-fp32 frames rendered after the timed region")
    ap.add_argument("--no-sub-records", action="store_true", help="headline only (default run: skip active / laplace / splat)")
    ap.add_argument("--no-trained-like-parity", action="store_true",
                    help="skip the trained-like-scene leg of parity_at_bench_size (two more oracle passes, ~25 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="cpu_baseline: time cap of the oracle leg")
    ap.add_argument("--cpu-chunks", type=int, default=16, help="cpu_baseline: 1024-ray chunks of the frame the oracle renders "
                    "(the same sample every run, so parity_at_bench_size is comparable between runs; ~12 s on 16 host threads)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))      # nothing has touched the GPU in this process
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus})", file=sys.stderr)
        sys.exit(2)

    # The contract is ONE JSON line on stdout.  Libraries write there too -- RCCL prints a version banner through C stdio when a
    # communicator is made, behind Python's own buffer -- so from here on file descriptor 1 IS stderr, and the line goes to a
    # private duplicate of the real stdout at the very end (emit).
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    def emit(line):
        print(json.dumps(line), file=real_stdout, flush=True)

    from uncertainty_nerf_gs_amd import lib
    lib.build_library()
    lib.require_gpu()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    comm = {"world_size": 1, "backend": None}
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else "?"
        comm = {"world_size": dist.get_world_size(), "backend": f"{dist.get_backend()} (RCCL {ver}, torch.distributed, "
                                                               "one process per GPU)"}

    default_run = args.method is None
    method = args.method or "mcdropout"
    if method == "splat":
        line = bench_splat(args, rank, world, dev, dist, args.steps, args.warmup)
        if rank == 0:
            emit(line)
    elif method == "ensemble":
        line = bench_ensemble(args, rank, world, dev, dist, args.steps, args.warmup)
        if rank == 0:
            line.update(comm)
            emit(line)
    else:
        K = args.mc_samples if method == "mcdropout" else 0
        rec = run_nerf(args, method, K, args.steps, args.warmup, rank, world, dev, dist, exact_check=not args.no_exact_check)
        subs = None
        if default_run and not args.no_sub_records and args.members % world == 0:
            # BASELINE.json configs[3] at every N: the M-member nerfacto ensemble sharded over the ranks (strong scaling)
            ens = bench_ensemble(args, rank, world, dev, dist, 3, 1)
            torch.cuda.synchronize()
            if rank == 0:
                subs = {"ensemble": {k: ens[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "scaling", "config", "exchange")}}
        if default_run and not args.no_sub_records and world == 1:
            # the other single-GPU configs of BASELINE.json, each with its own per-kernel times (fewer steps: the
            # default run must stay within minutes); the headline above is the north-star target config
            # the headline's workload with the dense layers at the REFERENCE's own eval precision (opt-in, --precision f16):
            # one f16 product per MAC, fp32 accumulate = torch.autocast(float16), forced by mcdropout_models.py:86-92
            # (mcdropout_f32eq: the headline's workload in the fp32-equivalent split-f16 form, round 3's headline)
            other = "f16x2" if rec["precision"] == "f16" else "f16"
            # mcdropout_tcnn: the headline's workload on the table layout and in the grid arithmetic the reference's DEFAULT
            # implementation="tcnn" computes (half2 rows: 4 B per gathered corner, its own algorithmic_bytes_per_ray)
            for name, m, kk, prec, gr in ((f"mcdropout_{'f32eq' if other == 'f16x2' else 'f16'}", "mcdropout", K, other, None),
                                          ("mcdropout_tcnn", "mcdropout", K, "f16", "tcnn"),
                                          ("active", "active", 0, None, None), ("laplace", "laplace", 0, None, None)):
                r = run_nerf(args, m, kk, 3, 2, rank, world, dev, dist, exact_check=False, want_cpu=False, precision=prec, grid=gr)
                subs[name] = {"value": r["value"], "unit": "Mrays/s", "ms_per_step": r["ms_per_step"], "steps": 3, "warmup": 2,
                              "workload": r["workload"], "per_kernel_ms_per_frame": r["roofline"]["per_kernel_ms_per_frame"],
                              "dtype": DTYPE_OF[r["precision"]], "dense_layers": r["dense_layers"], "hash_grid": r["hash_grid"],
                              "roofline": {k: r["roofline"].get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac",
                                                                             "avg_launch_ms", "traffic", "issue_roofline",
                                                                             "algorithmic_bytes_per_ray")}}
            sp = bench_splat(args, rank, world, dev, dist, 5, 2)
            subs["splat"] = {"value": sp["value"], "unit": sp["unit"], "ms_per_step": sp["ms_per_step"], "steps": 5, "warmup": 2,
                             "workload": sp["config"]["workload"],
                             "per_kernel_ms_per_frame": sp["roofline"]["per_kernel_ms_per_frame"],
                             "roofline": {k: sp["roofline"].get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac",
                                                                             "avg_launch_ms", "traffic", "raster_roofline")}}
        if rank == 0:
            H, W = args.height, args.width
            line = {
                "metric": "Mrays/s (+var), Mip-NeRF360-garden-shaped 1080p", "value": rec["value"], "unit": "Mrays/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_OF[rec["precision"]],
                "data": "synthetic",
                "config": {"workload": rec["workload"], "rays_per_step": H * W, "samples_per_ray": [256, 96, 48],
                           "hash_grid": rec["hash_grid"], "dense_layers": rec["dense_layers"],
                           "parallelism": f"views x{world}" if world > 1 else "single"},
                "roofline": rec["roofline"], "cpu_baseline": rec["cpu_baseline"],
                "parity_at_bench_size": rec["parity_at_bench_size"], "wider_arithmetic": rec["wider_arithmetic"],
                "step_wall_ms": rec["step_wall_ms"],
            }
            line.update(comm)
            if subs is not None:
                line["sub_records"] = subs
            emit(line)
    if dist is not None:
        dist.destroy_process_group()


def nerfacto_member_state_dict(t):
    """synthetic weights -> a plain `nerfacto` pipeline checkpoint in nerfstudio 1.1.0's key layout
    (MLPWithHashEncoding: mlp_base.encoder / mlp_base.mlp; `_model.` prefix), what an ensemble member's
    step-*.ckpt holds (ensemble_utils.py:71-72, :149-150)"""
    f = t["field"]
    sd = {"field.mlp_base.encoder.hash_table": f["table"]}
    for i, (w, b) in enumerate(((f["w0"], f["b0"]), (f["w1"], f["b1"]))):
        sd[f"field.mlp_base.mlp.layers.{i}.weight"], sd[f"field.mlp_base.mlp.layers.{i}.bias"] = w, b
    for i in range(3):
        sd[f"field.mlp_head.layers.{i}.weight"], sd[f"field.mlp_head.layers.{i}.bias"] = f["head_w"][i], f["head_b"][i]
    sd["field.embedding_appearance.embedding.weight"] = f["appearance"][None]
    for i, p in enumerate(t["props"]):
        sd[f"proposal_networks.{i}.mlp_base.encoder.hash_table"] = p["table"]
        for j, (w, b) in enumerate(((p["w0"], p["b0"]), (p["w1"], p["b1"]))):
            sd[f"proposal_networks.{i}.mlp_base.mlp.layers.{j}.weight"] = w
            sd[f"proposal_networks.{i}.mlp_base.mlp.layers.{j}.bias"] = b
    return {"_model." + k: v for k, v in sd.items()}


def bench_ensemble(args, rank, world, dev, dist, steps, warmup):
    """config 3: M-member nerfacto ensemble (ensemble_pipeline.py:144-191), members sharded over the ranks
    (M/N per GPU; 1 per GPU at N = M = 8).  Members are plain `nerfacto` models (ensemble_utils.py:149-150) loaded from
    checkpoints in upstream's key layout.  Every rank renders the SAME camera with its own members, then the per-pixel
    moments are formed exactly (two-pass) on pixel slices: one all_to_all + one all_gather over RCCL.  Strong scaling:
    the frame's total work (M renders) is fixed, so value = frame rays / time grows with N."""
    from uncertainty_nerf_gs_amd import ensemble, models, plugin, synthetic
    M = args.members
    assert M % world == 0, "members must divide evenly over the ranks"
    mine = [rank * (M // world) + i for i in range(M // world)]
    members = []
    for m in mine:   # seeds 100..100+M-1
        model = plugin.build_model("nerfacto", num_train_data=1)
        t = synthetic.make_scene_tensors(seed=100 + m, kind="mcdropout")
        t["field"]["b1"][0] += 4.6      # nerfacto's field carries average_init_density = 0.01: keep the scene as opaque
        model.load_state_dict(nerfacto_member_state_dict(t), strict=True)
        model.rays_per_launch = args.rays_per_launch
        members.append(model)
    pipe = ensemble.EnsemblePipeline(members)
    H, W = args.height, args.width
    cams = [models.Camera(synthetic.orbit_c2w(2 * math.pi * i / 24), 1111.0 * W / 1920, 1111.0 * W / 1920, W / 2, H / 2, H, W)
            for i in range(24)]

    def frame(i):
        return pipe.get_ensemble_outputs_for_camera_ray_bundle(cams[i % 24])

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.device(dev):
        for i in range(warmup):
            out = frame(i)
        sync_all()
        gc.freeze()   # _timed_region_gc
        t0 = time.perf_counter()
        for i in range(steps):
            out = frame(warmup + i)
        sync_all()
        elapsed = time.perf_counter() - t0
        gc.unfreeze()
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out["rgb"]).all() and "rgb_std" in out
    exchange = ensemble_exchange_record(pipe, cams[0], M, world, dev, dist, elapsed / steps * 1e3)
    del pipe, members
    torch.cuda.empty_cache()
    return {
        "exchange": exchange,
        "metric": "Mrays/s (+var), Mip-NeRF360-shaped 1080p, M-member ensemble", "value": H * W * steps / elapsed / 1e6,
        "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{M}-member nerfacto ensemble {W}x{H}, per-pixel mean/std over members",
                   "members": M, "members_per_gpu": M // world, "member_render_Mrays_per_s": M * H * W * steps / elapsed / 1e6,
                   "parallelism": (f"ensemble members x{world} (all_to_all of pixel slices + exact two-pass moments + all_gather, RCCL)"
                                   if world > 1 else "single GPU: members rendered in sequence, moments on the device")},
        "roofline": None, "cpu_baseline": None,
    }


def ensemble_exchange_record(pipe, cam, M, world, dev, dist, frame_ms):
    """The distributed aggregation's stages timed one by one at THIS world size: pack -> all_to_all of
    pixel slices -> exact two-pass moments -> all_gather of the reduced slices -> unpack (ensemble.aggregate_distributed,
    HIP events on the stream, outside the timed region).  At world size 1 the two collectives are RCCL's self-copies, so
    what this measures there is everything EXCEPT the fabric; the bytes each rank would send / receive at N = 8 and a
    projection from them are written next to it and labelled as what they are -- no 2 - 8 GPU run exists."""
    from uncertainty_nerf_gs_amd import ensemble
    import torch.distributed as tdist
    own_group = False
    try:
        if dist is None:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            # an explicit master store of our own with a short timeout: under a launcher that sets TORCHELASTIC_USE_AGENT_STORE
            # (torchrun at --nproc-per-node 1) an init_method URL makes rank 0 a CLIENT of a store nobody hosts -- ten minutes
            from datetime import timedelta
            store = tdist.TCPStore("127.0.0.1", port, 1, is_master=True, timeout=timedelta(seconds=60))
            tdist.init_process_group("nccl", store=store, rank=0, world_size=1, device_id=dev, timeout=timedelta(seconds=120))
            own_group = True
        outs = [m.get_outputs_for_camera(cam) for m in pipe.models]
        st = {}
        ensemble.aggregate_distributed(outs, stage_ms={})          # warm-up (communicator set-up, allocator)
        for _ in range(3):
            ensemble.aggregate_distributed(outs, stage_ms=st)
        calls = st.pop("calls")
        stages = {k: st[k] / calls for k in ("pack", "all_to_all", "moments", "all_gather", "unpack")}
        img = st["packed_image_bytes_per_member"]
        # N = 8, one member per GPU: every rank sends 7/8 of its packed image (1/8 to each peer, one xGMI link per peer) and
        # receives 7 reduced slices of (mean, var) = 2 x 1/8 image each
        n8_send, n8_recv = img * 7 // 8, 2 * img * 7 // 8
        link = 64e9 * 0.7       # one direction of one xGMI link (153 GB/s both ways, MI355X_MICROARCH), 70 % assumed achievable
        proj_ms = (n8_send / 7 / link + n8_recv / 7 / link) * 1e3
        member_ms = frame_ms / max(len(pipe.models), 1)
        ml = max(len(pipe.models), 1)
        # a rank packs its own members and reduces P x (members per rank) pixel-members at every world size: scale both to 1
        local = stages["pack"] / ml + stages["moments"] / ml + stages["unpack"]
        return {"world_size": world, "members_on_this_rank": len(pipe.models), "stage_ms": stages,
                "stage_ms_sum": sum(stages.values()), "packed_image_bytes_per_member": img,
                "bytes_all_to_all_sent_per_rank": st["bytes_all_to_all_sent_per_rank"],
                "bytes_all_gather_received_per_rank": st["bytes_all_gather_received_per_rank"],
                "projection_8_gpus_UNMEASURED": {
                    "assumes": "one member per GPU; each peer pair on its own xGMI link at 0.7 x 64 GB/s per direction; "
                               "pack and moments scaled from this rank's members to one, unpack as measured",
                    "all_to_all_bytes_sent_per_rank": n8_send, "all_gather_bytes_received_per_rank": n8_recv,
                    "fabric_ms": proj_ms, "member_render_ms": member_ms,
                    "frame_ms": member_ms + local + proj_ms,
                    "speedup_over_one_gpu": frame_ms * (M / max(len(pipe.models), 1)) / (member_ms + local + proj_ms)}}
    except Exception as e:  # noqa: BLE001 -- a record, not a gate: the ensemble line itself is already measured
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        if own_group:
            tdist.destroy_process_group()


def bench_splat(args, rank, world, dev, dist, steps, warmup):
    """config 5: active-splatfacto, N splats (SURVEY.md 8d synthetic set), 1080p, per-splat variance.
    A step = one frame: project + SH/beta + ONE bin-and-sort + 5-channel raster + depth-variance raster.
    Ranks render different views of the orbit with replicated splats (view-batch DP, no collective)."""
    from uncertainty_nerf_gs_amd import ops, splat, synthetic
    gp = {k: v.to(dev) for k, v in synthetic.make_splat_tensors(seed=7, N=args.splats).items()}
    H, W = args.height, args.width
    cam = dict(fx=1111.0 * W / 1920, fy=1111.0 * W / 1920, cx=W / 2, cy=H / 2, H=H, W=W)
    n_views = 24
    # device-resident poses, as a nerfstudio camera hands them to Model.get_outputs: the frame's timing covers the pose
    # read-back the model path performs (once, before anything is queued)
    poses = [synthetic.orbit_c2w(2 * math.pi * i / n_views, radius=2.5, height=0.5).to(dev) for i in range(n_views)]
    bg = torch.zeros(3, device=dev)

    def frame(i):
        return splat.active_splatfacto_outputs(gp, poses[(rank + i * world) % n_views], background=bg, **cam)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        out = frame(i)
    sync_all()
    ops.TIMER = ops.KernelTimer(prealloc=32 * steps)
    gc.freeze()   # _timed_region_gc
    t0 = time.perf_counter()
    for i in range(steps):
        out = frame(warmup + i)
    sync_all()
    elapsed = time.perf_counter() - t0
    gc.unfreeze()
    timer, ops.TIMER = ops.TIMER, None
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out["rgb"]).all()
    if rank == 0:
        ksum = timer.summary()
        dom = max(ksum, key=lambda k: ksum[k]["total_ms"])
        # algorithmic bytes of one frame: splat parameters in, projection record out and back in,
        # the sort's key/payload traffic as the reference's 64-bit-key radix sort would move it (I x 96 B),
        # the rasteriser's per-intersection reads (twice: 5-channel pass + depth-variance pass) and the images.
        from uncertainty_nerf_gs_amd.splat import viewmat_from_c2w
        q = gp["quats"] / gp["quats"].norm(dim=-1, keepdim=True)
        pr = ops.splat_project(gp["means"].contiguous(), torch.exp(gp["scales"]), 1.0, q.contiguous(),
                               viewmat_from_c2w(poses[0])[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, 16)
        n_isect = int(pr[5].sum().item())
        # what the frame actually sorts: the tight lists (tiles each splat's alpha >= 1/255 ellipse reaches)
        pt = ops.splat_project(gp["means"].contiguous(), gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(),
                               viewmat_from_c2w(poses[0])[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, 16, raw=True,
                               opacity_logits=gp["opacities"].reshape(-1).contiguous())
        n_isect_tight = int(pt[5].sum().item())
        N = args.splats
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        # Bytes the bin-and-sort call REALLY moves, kernel by kernel (algorithmic: every array read or written once per
        # kernel that touches it), for the lists the frame actually sorts -- Is = num_intersects_sorted tight pairs:
        Is = n_isect_tight
        # (unerf_splat.hip: four staged 8-bit LSD passes over the N depth keys, emission, two staged LSD passes over the pairs)
        T1 = tiles + 1
        nchunk_d, nchunk_t = -(-N // 1024), -(-Is // 2048)
        nhw = -(-nchunk_t // 16)
        sort_parts = {
            "depth_keys (N x 16 B)": N * 16,
            "depth sort: 4 x (histogram N x 4 B read; scatter N x 8 B read + N x 8 B written; 256 x chunks table written, scanned, read)":
                4 * (N * 20 + 256 * nchunk_d * 4 * 4),
            "depth-ordered counts (gather N x 12 B, N x 4 B written, inside the last depth pass) + their scan (N x 8 B)": N * 24,
            "emission: 40 B per splat read, 6 B per pair written": N * 40 + Is * 6,
            "tile sort pass 1 (low digit): histogram 2 B per pair read + whole-key rows written; 6 B per pair read, 6 B written":
                Is * 14 + nhw * T1 * 4 + 64 * nchunk_t * 4 * 4,
            "tile sort pass 2 (high digit): histogram 2 B per pair read; 6 B per pair read, 4 B written":
                Is * 12 + 128 * nchunk_t * 4 * 4,
            "tile totals -> tile_bins: whole-key rows read, 16 partial rows written and read, bins written": nhw * T1 * 4 + 32 * T1 * 4 + T1 * 12,
        }
        sort_bytes = sum(sort_parts.values())
        # the frame's algorithmic bytes with the sort counted the same way (SURVEY 8d counted gsplat's 64-bit-key sort of
        # gsplat's box lists, I x 96 B: kept below as a labelled side figure, not as work this code performs)
        frame_bytes = N * 240 + 2 * N * 36 + sort_bytes + 2 * Is * 36 + H * W * 7 * 4
        kbytes = {"splat_bin_sort": sort_bytes, "splat_rasterize_c5": Is * 36 + H * W * 7 * 4,
                  "splat_rasterize_c1": Is * 20 + H * W * 3 * 4}
        ach = kbytes.get(dom, 0) / (ksum[dom]["avg_ms"] * 1e-3) / 1e9 if dom in kbytes else None
        # The rasteriser is bound by VALU issue per (pixel, splat) pair, not by bytes (its splat lists are staged through
        # LDS once per tile): with a committed SQ counter pass of this command the roof is instruction issue, as for the
        # field kernels -- achieved = 4 x VALU instructions of one 5-channel raster launch / its live duration.
        iss_r = _issue_profile("splat", 0)                       # raster_kernel<5>: issue profile of the 5-channel pass
        iss = iss_r if dom == "splat_rasterize_c5" else None
        raster = None
        if iss_r is not None and "splat_rasterize_c5" in ksum:
            g = iss_r["issue_cycles_per_launch"] / (ksum["splat_rasterize_c5"]["avg_ms"] * 1e-3) / 1e9
            raster = {"kernel": "splat_rasterize_c5", "bound": "valu-issue", "achieved": g, "peak": ISSUE_PEAK_GCYC, "unit": "Gcycle/s",
                      "frac": g / ISSUE_PEAK_GCYC, "avg_launch_ms": ksum["splat_rasterize_c5"]["avg_ms"],
                      "valu_insts_per_launch": iss_r["valu_insts_per_launch"], "issue_source": iss_r["source"]}
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_splat.json")
        if os.path.exists(tfile):
            from uncertainty_nerf_gs_amd import lib
            tj = json.load(open(tfile))
            if tj.get("kernel_source_digest") in (None, lib._source_digest()) and "field_fwd" in tj.get("kernels", {}):
                tk = tj["kernels"]["field_fwd"]     # (summarize_pmc's slot name for the profile's raster_kernel<5>)
                if raster is not None:
                    raster["traffic"] = tk["fetch_bytes"] + tk["write_bytes"]
                if dom == "splat_rasterize_c5":
                    traffic = tk["fetch_bytes"] + tk["write_bytes"]
                ts = tj["kernels"].get("splat_bin_sort")     # FETCH_SIZE + WRITE_SIZE summed over the call's kernels
                if dom == "splat_bin_sort" and ts:
                    traffic = ts["fetch_bytes"] + ts["write_bytes"]
        line = {
            "metric": "Mrays/s (+var) [pixels of an active-splatfacto frame], 1080p", "value": H * W * steps * world / elapsed / 1e6,
            "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"active-splatfacto {W}x{H}, N={args.splats} splats, rgb+beta+depth+depth_var",
                       "parallelism": f"views x{world}" if world > 1 else "single"},
            "roofline": {"kernel": dom,
                         **({"bound": "valu-issue", "unit": "Gcycle/s", "peak": ISSUE_PEAK_GCYC,
                             "achieved": iss["issue_cycles_per_launch"] / (ksum[dom]["avg_ms"] * 1e-3) / 1e9,
                             "frac": iss["issue_cycles_per_launch"] / (ksum[dom]["avg_ms"] * 1e-3) / 1e9 / ISSUE_PEAK_GCYC,
                             "issue_source": iss["source"], "issue_kernel": iss["kernel_name"],
                             "note": "issue cycles of the 5-channel raster launch (4 x VALU instructions, committed PMC pass) "
                                     "over the mean live duration of the frame's raster launches"}
                            if iss is not None else
                            {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": None if ach is None else ach / HBM_PEAK_GBS,
                             "note": ("bin-and-sort = depth sort of the N splats (four staged 8-bit LSD passes), emission of the "
                                      "num_intersects_sorted (tile, id) pairs in depth order, and the stable tile sort (two staged LSD "
                                      "passes: per-chunk digit histogram, prefix over chunks, rank + LDS staging + run-wise scatter): "
                                      "achieved = the bytes those kernels read and write (sort_bytes_by_kernel) over this call's "
                                      "time.  Launch-latency, VALU (emission) and store-request bound, not HBM bound (DESIGN.md 4.3)"
                                      if dom == "splat_bin_sort" else "algorithmic bytes of the dominant kernel")}),
                         "raster_roofline": raster,
                         "traffic": traffic, "avg_launch_ms": ksum[dom]["avg_ms"], "num_intersects": n_isect,
                         "num_intersects_sorted": n_isect_tight,
                         "sort_bytes_by_kernel": sort_parts,
                         "gsplat_style_sort_bytes_not_performed": n_isect * 96,
                         "frame_algorithmic_bytes": frame_bytes,
                         "frame_frac_of_hbm_peak": frame_bytes * steps * world / elapsed / 1e9 / HBM_PEAK_GBS,
                         "per_kernel_ms_per_frame": {k: round(v["total_ms"] / steps, 3) for k, v in sorted(ksum.items())}},
            "cpu_baseline": None,
        }
        return line
    return None


def cpu_baseline(t, args, method, c2w, cam, K, autocast=None, ids_only=None):
    """The CPU oracle (a port: the reference's own stack is not installable here) on a bounded,
    strided sample of the same frame's rays, all host cores.  -> (record, ray ids [n], oracle outputs {key: [n, C]})
    autocast: torch.float16 = the oracle's emulation of the autocast the reference forces at eval (parity legs only; the
    timed baseline is the fp32 pass).  ids_only: exactly these rays (the parity legs reuse the timed pass' sample)."""
    import numpy as np
    from oracle import nerf_oracle as O
    from oracle import sampled_frame as SF
    # torch-CPU oversubscribes badly on many-core hosts for these small tensors (256 threads were
    # 40x slower than 16 on the MI355X host): cap the pool and report the threads actually used.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    sc = O.scene_from_tensors(t)
    o, d, _ = O.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["H"], cam["W"])
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    chunk = 1024
    stride = max(1, o.shape[0] // 64 // chunk) * chunk
    kw = {}
    if method == "mcdropout":
        kw = dict(K=K, mc_seed=1234, p_drop=0.2)
    if method == "laplace":
        from uncertainty_nerf_gs_amd import synthetic
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=100)
        kw = dict(ws_density=wsd, ws_rgb=wsr, depth_seed=7, depth_draws=100)   # the frame's own depth draws (depth_seed=7)
    ids_all = (np.arange(o.shape[0] // stride, dtype=np.int64)[:, None] * stride + np.arange(chunk, dtype=np.int64)[None]).reshape(-1)
    if ids_only is not None:
        ids_all = np.asarray(ids_only, dtype=np.int64)
    ids, lists, t0 = [], {}, time.perf_counter()
    for part, out in SF.reference_chunks(method, sc, o, d, ids_all, step=chunk, autocast=autocast, **kw):
        ids.append(part)
        for k, v in out.items():
            lists.setdefault(k, []).append(v)
        if ids_only is None and (len(ids) >= args.cpu_chunks or time.perf_counter() - t0 >= args.cpu_seconds):
            break
    dt = time.perf_counter() - t0
    ids = np.concatenate(ids)
    done = len(ids)
    rec = {"value": done / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
           "sample": f"{done} rays ({done // chunk} strided 1024-ray chunks of the same 1080p camera), {dt:.1f} s, "
                     f"torch-CPU fp32 oracle, {cores} threads"}
    return rec, ids, {k: torch.cat(v) for k, v in lists.items()}


def parity_record(got, ids, refs, precision):
    """One scene of `parity_at_bench_size`: the GPU frame (full tables, 1080p, the bench's launch groups) against the oracle
    outputs for `ids` -- refs = {"fp32": ..., "autocast16": ... (precision "f16" only)} -- through the north-star gate
    quantities on the tests' targets (oracle/targets.py, the same functions tests/test_gpu_nerf_e2e._gates calls): the
    informative target carries the absolute gates; the plain one is gated over 8 noise seeds RELATIVE to what the two oracles
    differ by on the same seeds (`plain_gate`, `reference_arithmetics_gap`: no implementation of the reference's f16
    arithmetic can be closer to its fp32 semantics than the reference itself)."""
    from oracle import targets
    sel = torch.from_numpy(ids).to(got["rgb"].device)
    pick = lambda k: got[k].reshape(-1, got[k].shape[-1])[sel].cpu()
    rgb, std = pick("rgb"), pick("rgb_std")

    def against(r_rgb, r_std, o_rgb, o_std):
        g = targets.gate_deltas(o_rgb, o_std, r_rgb, r_std, targets.gt_image_informative(r_rgb, r_std), err_types=("mse",))
        loose = [targets.gate_deltas(o_rgb, o_std, r_rgb, r_std, targets.gt_image_plain(r_rgb, 123 + i), err_types=("mse",))
                 for i in range(8)]
        return {"max_abs_rgb": float((o_rgb - r_rgb).abs().max()), "max_abs_rgb_std": float((o_std - r_std).abs().max()),
                "psnr_vs_target": g["psnr_ref"], "d_psnr": g["d_psnr"], "ause_mse_oracle": g["ause_mse_ref"], "d_ause_mse": g["d_ause_mse"],
                "plain_target": {"ause_mse_oracle": loose[0]["ause_mse_ref"], "d_psnr_seed123": loose[0]["d_psnr"],
                                 "d_ause_mse_seed123": loose[0]["d_ause_mse"],
                                 "d_ause_mse_mean": float(sum(x["d_ause_mse"] for x in loose) / len(loose)),
                                 "d_ause_mse_max": float(max(x["d_ause_mse"] for x in loose)), "seeds": len(loose)},
                "inside_gates": bool(g["d_psnr"] <= 1e-4 and g["d_ause_mse"] <= 1e-3)}

    rec = {"rays": int(len(ids)), "precision": precision,
           "oracle": "torch-CPU (oracle/sampled_frame.py), same pose, same mask / depth-draw counters", "vs": {}}
    for name, ref in refs.items():
        r = against(ref["rgb"], ref["rgb_std"], rgb, std)
        # the plain target, gated relative to what it resolves (oracle/targets.plain_gate): against the oracle of the build's
        # own arithmetic the floor is the two oracles' gap (2.5 x), otherwise noise of the build's RMS difference (3 x)
        other = None
        if name == "autocast16" and "fp32" in refs:
            other = (refs["fp32"]["rgb"], refs["fp32"]["rgb_std"])
        r["plain_gate"] = targets.plain_gate(rgb, std, ref["rgb"], ref["rgb_std"], other=other, err_types=("mse",))
        r["inside_gates"] = bool(r["inside_gates"] and r["plain_gate"]["ok"])
        r["max_abs_accumulation"] = float((pick("accumulation") - ref["accumulation"]).abs().max())
        dd = (pick("depth") - ref["depth"]).abs() > 1e-3 * ref["depth"].abs()
        r["median_depth_pixels_off_1e-3"] = float(dd.double().mean())
        rec["vs"][name] = r
    if "autocast16" in refs:
        a, b = refs["fp32"], refs["autocast16"]
        gap = against(a["rgb"], a["rgb_std"], b["rgb"], b["rgb_std"])
        gap.pop("inside_gates")
        rec["reference_arithmetics_gap"] = gap
    rec["inside_gates"] = all(r["inside_gates"] for r in rec["vs"].values())
    return rec


if __name__ == "__main__":
    main()
