"""bench.py's output contract, checked on the committed line of the latest profiled run (profiles/*_default_bench.json
is the stdout line of `python bench.py` on MI355X) and on the script's command line -- no GPU needed."""
import glob
import json
import os
import subprocess
import sys

from conftest import ROOT


def _latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_default_bench.json")))
    assert files, "no committed default bench line under profiles/"
    return json.load(open(files[-1])), files[-1]


def test_committed_bench_line_has_every_contract_field():
    d, path = _latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_at_bench_size"):
        assert k in d, (path, k)
    assert d["unit"] == "Mrays/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["n_gpus"] == 1
    assert d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    # the headline computes in the reference's own eval arithmetic: f16 operands (forced autocast, mcdropout_models.py:86-92)
    assert d["data"] == "synthetic" and d["dtype"] == "f16 operands, f32 accumulate"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) < 1e-6 * d["value"]
    # the driver-timed headline is the north-star target config (BASELINE.json configs[2]): nerfacto-mcdropout, K = 8
    assert "mcdropout" in d["config"]["workload"] and "K=8" in d["config"]["workload"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert k in r, k
    # the contract's roofline: algorithmic bytes (or flops) of the dominant kernel per launch / its live duration
    assert (r["bound"], r["unit"], r["peak"]) in (("hbm", "GB/s", 8000.0), ("mfma", "TFLOP/s", 2500.0))
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.0
    if r["bound"] == "hbm":
        assert abs(r["achieved"] - r["algorithmic_bytes_per_ray"] * r["rays_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is not None and r["traffic"] > 0 and 0 < r["other_roofs"]["hbm_frac"] < 1
    # instruction issue is the roof that binds: VALU + MFMA cycles over SIMD cycles at the peak
    # clock -- carried next to the contract's figure, reproducible from the committed PMC pass: issue cycles per launch /
    # live launch time
    ir = r["issue_roofline"]
    assert ir["bound"] == "valu-issue" and ir["unit"] == "Gcycle/s" and ir["peak"] == 1024 * 2.4
    assert abs(ir["frac"] - ir["achieved"] / ir["peak"]) < 1e-9 and 0.0 < ir["frac"] <= 1.0
    iss = json.load(open(os.path.join(ROOT, "profiles", "issue_mcdropout_f16.json")))
    want = iss["issue_cycles_per_launch"] * r["rays_per_launch"] / iss["rays_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
    assert abs(want - ir["achieved"]) < 1e-6 * want
    assert os.path.exists(os.path.join(ROOT, iss["source"].split(" ")[0]))
    # oracle parity AT THE BENCH SIZE: the cpu_baseline leg's oracle outputs against the same rays of a GPU frame
    pb = d["parity_at_bench_size"]
    if "random_init" in pb:    # round 5 layout: per scene, per reference oracle, the targets of oracle/targets.py
        assert pb["inside_gates"] is True and set(pb) >= {"random_init", "trained_like", "overflow_rerenders", "gates"}
        assert pb["overflow_rerenders"] == 0 and pb["trained_like"]["overflow_rerenders"] == 0
        for scene in ("random_init", "trained_like"):
            sc = pb[scene]
            assert sc["rays"] == 16384 and sc["precision"] == "f16" and set(sc["vs"]) == {"fp32", "autocast16"}
            for name, r in sc["vs"].items():     # gated: the informative target (0 < AUSE well below the ~0.6 of a random ranking)
                assert r["inside_gates"] is True and r["d_psnr"] <= 1e-4 and r["d_ause_mse"] <= 1e-3, (scene, name)
                assert 0.1 < r["ause_mse_oracle"] < 0.5 and r["plain_target"]["seeds"] == 8 and r["plain_target"]["ause_mse_oracle"] > 0.5
            # the plain target is a tie-order lottery: the reference's two arithmetics differ on it by about as much as
            # the build differs from either (and by more than the gate on the trained-like scene)
            gap = sc["reference_arithmetics_gap"]
            assert gap["d_ause_mse"] <= 1e-3 and gap["plain_target"]["d_ause_mse_max"] > gap["d_ause_mse"]
        assert pb["random_init"]["vs"]["fp32"]["max_abs_rgb"] < 2e-4
    else:                      # rounds 3 - 4
        assert pb["rays"] == 16384 and pb["inside_gates"] is True and pb["d_psnr"] <= 1e-4 and pb["d_ause_mse"] <= 1e-3
        assert pb["max_abs_rgb"] < 1e-4 and pb["precision"] == "f16"
        assert 0.1 < pb["ause_mse_oracle"] < 0.5 and "informative" in pb["target"]
        assert pb["d_ause_mse_uninformative_target"]["seeds"] == 8 and pb["d_ause_mse_uninformative_target"]["max"] < 5e-3
    assert len(d["step_wall_ms"]) == d["steps"] and abs(sum(d["step_wall_ms"]) / d["steps"] - d["ms_per_step"]) < 0.05 * d["ms_per_step"]
    subs = d["sub_records"]
    assert set(subs) - {"mcdropout_tcnn"} == {"ensemble", "mcdropout_f32eq", "active", "laplace", "splat"}
    if "mcdropout_tcnn" in subs:   # round 5: the headline's workload on tcnn-layout HALF tables (4 B per gathered corner)
        tc = subs["mcdropout_tcnn"]
        assert "half2" in tc["hash_grid"] and "K=8" in tc["workload"] and tc["dtype"].startswith("f16")
        assert tc["roofline"]["algorithmic_bytes_per_ray"] == d["roofline"]["algorithmic_bytes_per_ray"] - 48 * 16 * 8 * 4
    for k, v in subs.items():
        assert v["value"] > 0 and v["ms_per_step"] > 0, k
        if k != "ensemble":
            assert v["per_kernel_ms_per_frame"], k
    assert "density [H,W,48] kept" in subs["active"]["workload"]
    # the fp32-equivalent (split-f16) form of the headline's workload, with its arithmetic spelled out
    eq = subs["mcdropout_f32eq"]
    assert "K=8" in eq["workload"] and eq["dtype"].startswith("f32") and eq["value"] < d["value"]
    assert abs(d["wider_arithmetic"]["max_abs_rgb_diff_vs_headline"]) < 1e-3 and d["wider_arithmetic"]["precision"] == "f16x2"
    # every NeRF sub-record carries the same two definitions as the headline (committed PMC passes of this round)
    for k in ("mcdropout_f32eq", "active", "laplace"):
        rr = subs[k]["roofline"]
        assert rr["bound"] in ("hbm", "mfma") and 0 < rr["frac"] <= 1 and rr["traffic"] > 0, k
        assert rr["issue_roofline"]["bound"] == "valu-issue" and 0 < rr["issue_roofline"]["frac"] <= 1, k
    # the splat sort's roofline counts the bytes its kernels really move (and the PMC traffic of those kernels)
    sp = subs["splat"]["roofline"]
    assert sp["kernel"] == "splat_bin_sort" and 0 < sp["frac"] < 1 and sp["traffic"] > 0
    # BASELINE.json configs[3]: the 8-member ensemble, strong scaling over --gpus N
    ens = subs["ensemble"]
    assert ens["scaling"] == "strong" and ens["config"]["members"] == 8 and ens["config"]["members_per_gpu"] == 8 // d["n_gpus"]
    assert d["world_size"] == d["n_gpus"] and (d["backend"] is None) == (d["n_gpus"] == 1)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"]


def test_bench_launches_its_own_ranks_for_multi_gpu():
    """The driver's command is `python bench.py --gpus N`: without a launcher around it the script starts
    torch.distributed.run itself (a child process, one rank per GPU).  On this GPU-less box the RANKS then fail at
    require_gpu(); the parent neither refuses (rc 2) nor touches the GPU, and hands the child's exit code on."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node=2" in p.stderr, p.stderr[-2000:]
    import torch
    if not torch.cuda.is_available():
        assert p.returncode not in (0, 2), (p.returncode, p.stderr[-2000:])
        assert "no HIP device visible" in p.stderr, p.stderr[-3000:]       # raised inside the ranks
        assert p.stdout.strip() == ""


def test_a_rank_count_that_contradicts_gpus_is_refused():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert p.returncode == 2 and "WORLD_SIZE=4" in p.stderr


def test_parity_record_is_zero_for_identical_frames_and_well_conditioned():
    """bench.parity_record on CPU tensors: identical frames give zero deltas; a perturbation of rgb_std at the level the f16
    kernels differ from the oracle moves the gated AUSE (target error follows the uncertainty) far less than the AUSE against
    the tests' uninformative target -- the reason the gate reads the first (tests/tools/ause_conditioning.py)."""
    import importlib.util
    import numpy as np
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    g = torch.Generator().manual_seed(0)
    n = 8192
    ref = {"rgb": torch.rand(n, 3, generator=g) * 0.5 + 0.2, "rgb_std": 0.002 + 0.007 * torch.rand(n, 1, generator=g) ** 3,
           "accumulation": torch.ones(n, 1), "depth": 1.0 + torch.rand(n, 1, generator=g)}
    ids = np.arange(n, dtype=np.int64)
    same = bench.parity_record({k: v.clone() for k, v in ref.items()}, ids, {"fp32": ref}, "f16")
    r0 = same["vs"]["fp32"]
    assert same["rays"] == n and r0["d_psnr"] == 0 and r0["d_ause_mse"] == 0 and same["inside_gates"] is True
    assert r0["plain_target"]["d_ause_mse_max"] == 0 and r0["median_depth_pixels_off_1e-3"] == 0
    got = {k: v.clone() for k, v in ref.items()}
    got["rgb_std"] = (ref["rgb_std"] + (torch.rand(n, 1, generator=g) - 0.5) * 1.2e-5).clamp(min=0)
    got["rgb"] = ref["rgb"] + (torch.rand(n, 3, generator=g) - 0.5) * 2e-5
    rec = bench.parity_record(got, ids, {"fp32": ref, "autocast16": got}, "f16")
    r1 = rec["vs"]["fp32"]
    assert rec["inside_gates"] is True and r1["d_psnr"] < 1e-4 and rec["vs"]["autocast16"]["d_ause_mse"] == 0
    assert r1["d_ause_mse"] < r1["plain_target"]["d_ause_mse_mean"]
    assert 0 < r1["ause_mse_oracle"] < 0.6 < r1["plain_target"]["ause_mse_oracle"] + 0.1
    assert abs(rec["reference_arithmetics_gap"]["d_ause_mse"] - r1["d_ause_mse"]) < 1e-12     # same pair of frames here
    # the plain target is GATED, relative to its floor: the two oracles' gap where the build is held against the oracle of
    # its own arithmetic (2.5 x), noise of the build's RMS difference elsewhere (3 x)
    pa, pf = rec["vs"]["autocast16"]["plain_gate"], r1["plain_gate"]
    assert pa["floor"] == "reference arithmetics gap" and pa["ok"] and pa["d_ause_mse_mean"] == 0
    assert pf["floor"].startswith("noise") and pf["ok"] and pf["d_ause_mse_bound"] >= 1e-3
    # an image that is systematically off fails both targets (the plain target cannot see a wrong RANKING by itself -- on it
    # every ranking is a random order, oracle/targets.py -- which is why the informative target carries the absolute gate)
    bad = {k: v.clone() for k, v in got.items()}
    bad["rgb"] = ref["rgb"] + 2e-3
    rb = bench.parity_record(bad, ids, {"fp32": ref, "autocast16": got}, "f16")
    assert rb["inside_gates"] is False and rb["vs"]["autocast16"]["plain_gate"]["ok"] is False
    worse = {k: v.clone() for k, v in got.items()}
    worse["rgb_std"] = ref["rgb_std"].flip(0).clone()
    assert bench.parity_record(worse, ids, {"fp32": ref}, "f16")["inside_gates"] is False
