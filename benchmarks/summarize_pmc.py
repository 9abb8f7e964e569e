"""Condense rocprofv3 --pmc counter_collection CSVs (experiment tooling, not product code).

  reduce  <counter_collection.csv> <out.csv>   per (kernel, counter): launches, mean value, mean duration
  summary <dir> <tag>                          merge the reduced CSVs into <tag>_<method>_pmc_summary.json and
                                               traffic_<method>.json (FETCH_SIZE / WRITE_SIZE are in KB)
Only kernels of libunerf (names without `at::native` / `rocclr`) are kept.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def reduce(src, dst):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    with open(src, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if "at::native" in k or "rocclr" in k or "hipcub" in k.lower() and False:
                continue
            a = acc[(short(k), row["Counter_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            a[2] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "launches", "mean_value", "mean_dur_us"])
        for (k, c), (n, v, d) in sorted(acc.items()):
            w.writerow([k, c, n, f"{v / n:.6f}", f"{d / n:.3f}"])


# the dominant field kernel of a method = the first kernel name with one of these prefixes (template arguments after the
# prefix -- tcnn / sites / drop flags -- vary with the round)
FIELD_KERNELS = {"active": ("field_kernel_mfma16<0, 0, false, false, false", "field_kernel_mfma16<0, false, false, false, false", "field_kernel_mfma<0, false"),
                 "active_f16": ("field_kernel_mfma16<0, 0, false, false, true", "field_kernel_mfma16<0, false, false, false, true"),
                 "mcdropout": ("field_kernel_mfma16<1, 0, false, true, false", "field_kernel_mfma16<1, false, false, true, false",
                               "field_kernel_mfma16<1, false", "field_kernel_mfma<1, false"),
                 "mcdropout_f16": ("field_kernel_mfma16<1, 0, false, true, true", "field_kernel_mfma16<1, false, false, true, true"),
                 "mcdropout_f16_tcnn": ("field_kernel_mfma16<1, 2, false, true, true",),      # tcnn layout, half2 rows
                 "mcdropout_f16_tcnn32": ("field_kernel_mfma16<1, 1, false, true, true",),    # tcnn layout, fp32 rows
                 "laplace": ("field_kernel_mfma16_laplace<0, false", "field_kernel_mfma16_laplace<false, false", "field_kernel_mfma16_laplace<false"),
                 "splat": ("raster_kernel<5",)}
# the kernels of one unerf_splat_bin_sort call (substrings of the short names)
SORT_KERNELS = ("depth_keys_kernel", "sorted_counts_kernel", "map_intersects_kernel", "tile_hist_kernel", "tile_colsum_kernel",
                "tile_segscan_kernel", "tile_scan_kernel", "tile_apply_kernel", "tile_scatter_kernel", "tile_edges_kernel",
                "rs_hist_kernel", "rs_rowscan_kernel", "rs_scatter_kernel", "rs_colsum_kernel", "scan_sums_kernel", "scan_apply_kernel",
                "merge_sort", "radix_sort", "onesweep", "scan_config", "lookback_scan")
# further kernels of a profile that get their own issue_<name>.json (same definition)
EXTRA_ISSUE = {"laplace": {"lap_depth": ("lap_depth_kernel<3",)}, "splat": {"splat_raster1": ("raster_kernel<1",)}}
K_OF = {"mcdropout": 8, "mcdropout_f16": 8, "mcdropout_f16_tcnn": 8, "mcdropout_f16_tcnn32": 8}
FRAME_RAYS = 1920 * 1080      # bench.py's frame
PMC_FRAMES = 2                # collect_profiles.sh pmc(): --steps 1 --warmup 1


def _rays_per_launch(method, launches):
    """mean rays of one launch of a per-launch-group kernel (the last group of a frame is a partial one); the splat
    kernels see the whole frame"""
    if method == "splat" or not launches:
        return FRAME_RAYS
    return FRAME_RAYS * PMC_FRAMES / launches


def _source_digest():
    """sha256 of the kernel sources the counters were taken from (uncertainty_nerf_gs_amd.lib._source_digest): bench.py
    uses an issue profile only while the sources it describes are the ones that are built"""
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in _sys.path:
        _sys.path.insert(0, root)
    from uncertainty_nerf_gs_amd import lib
    return lib._source_digest()


def summary(d, tag):
    digest = _source_digest()
    for method in ("active", "active_f16", "mcdropout", "mcdropout_f16", "mcdropout_f16_tcnn", "mcdropout_f16_tcnn32", "laplace", "splat"):
        kernels = defaultdict(dict)
        for fn in sorted(os.listdir(d)):
            m = re.match(rf"{tag}_{method}_pmc_(fetch|write|sq|lds|ta|tcc)\.csv$", fn)
            if not m:
                continue
            with open(os.path.join(d, fn), newline="") as f:
                for row in csv.DictReader(f):
                    kernels[row["kernel"]][row["counter"]] = float(row["mean_value"])
                    kernels[row["kernel"]].setdefault("avg_dur_us", float(row["mean_dur_us"]))
                    kernels[row["kernel"]].setdefault("launches", int(row["launches"]))
        if not kernels:
            continue
        out = {"command": "benchmarks/collect_profiles.sh (one rocprofv3 --pmc pass per counter set, --kernel-trace only)",
               "units": "per-launch means; FETCH_SIZE/WRITE_SIZE in KB; SQ_* busy/wait counters in quad-cycles; "
                        "GRBM_GUI_ACTIVE summed over 8 XCDs",
               "kernels": kernels}
        with open(os.path.join(d, f"{tag}_{method}_pmc_summary.json"), "w") as f:
            json.dump(out, f, indent=1)
        fk = next((k for pre in FIELD_KERNELS[method] for k in sorted(kernels) if k.startswith(pre)), None)
        if fk and "FETCH_SIZE" in kernels[fk] and "WRITE_SIZE" in kernels[fk]:
            t = {"method": method, "K": K_OF.get(method, 0), "rays_per_launch": _rays_per_launch(method, kernels[fk].get("launches")), "kernel_source_digest": digest,
                 "source": f"profiles/{tag}_{method}_pmc_fetch.csv + {tag}_{method}_pmc_write.csv "
                           "(separate rocprofv3 --pmc passes)",
                 "kernels": {"field_fwd": {
                     "kernel_name": fk,
                     "fetch_bytes": kernels[fk]["FETCH_SIZE"] * 1024.0,
                     "write_bytes": kernels[fk]["WRITE_SIZE"] * 1024.0,
                     "correction": "none: the reads are 8-byte gathers (one 64-B fabric request each), not the wide "
                                   "coalesced streams for which MI355X_MICROARCH.md prescribes x2; writes read exact"}}}
            if method == "splat":
                # the bin-and-sort CALL is a dozen kernels (depth keys, rocprim merge sort + scans, emission, the one-pass
                # tile sort): its fabric traffic = the per-frame sum of their FETCH_SIZE + WRITE_SIZE
                sort_names = [k for k in kernels if any(p in k for p in SORT_KERNELS) and "FETCH_SIZE" in kernels[k]
                              and "WRITE_SIZE" in kernels[k]]
                per_frame = lambda k, c: kernels[k][c] * 1024.0 * kernels[k].get("launches", 0) / PMC_FRAMES
                t["kernels"]["splat_bin_sort"] = {
                    "kernel_names": sorted(sort_names),
                    "fetch_bytes": sum(per_frame(k, "FETCH_SIZE") for k in sort_names),
                    "write_bytes": sum(per_frame(k, "WRITE_SIZE") for k in sort_names),
                    "per_kernel_bytes": {k: {"fetch": per_frame(k, "FETCH_SIZE"), "write": per_frame(k, "WRITE_SIZE"),
                                             "launches_per_frame": kernels[k].get("launches", 0) / PMC_FRAMES} for k in sorted(sort_names)},
                    "correction": "none (see field_fwd)"}
            with open(os.path.join(d, f"traffic_{method}.json"), "w") as f:
                json.dump(t, f, indent=1)
        need = ("SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE")
        targets = [(method, fk)]
        for name, prefixes in EXTRA_ISSUE.get(method, {}).items():
            targets.append((name, next((k for pre in prefixes for k in sorted(kernels) if k.startswith(pre)), None)))
        for issue_name, fk in targets:
          if fk and all(c in kernels[fk] for c in need):
            kk = kernels[fk]
            # Cycles a launch NEEDS on each of the two resources a SIMD has, priced with the constants
            # benchmarks/issue_sweep_probe.hip measured in shader cycles (profiles/r6_issue_sweep.jsonl, round 6):
            #   issue lane : 4 per wave64 VALU instruction (4.0 - 4.2 at 1, 2 and 3 waves per SIMD for every packed, converting,
            #                VOP3 or transcendental-free form tried; plain VOP2 add / mul / and / xor on registers drop to 2.1 at
            #                two waves and 1.4 at three -- for the share of the stream made of those this is an over-estimate)
            #                + 10 per v_mfma_f32_32x32x16_f16 (what an MFMA holds the issue for: gap = 10.3 + 4 n once n > 6);
            #   matrix lane: 32 per MFMA (back-to-back rate of the pipe; one chain or two, no difference).
            # The lanes run side by side (up to six 4-cycle VALU instructions per MFMA cost 0.25 - 0.5 cycle each), so a launch
            # needs max(issue lane, matrix lane).  Rounds 2 - 5 priced 4 VALU + 32 MFMA as ONE lane ("no overlap"): that
            # probe put 8 fillers = 42 issue cycles into every 32-cycle gap and assumed a 2.4 GHz clock.
            # SQ_INSTS_VALU (which includes the MFMAs) and SQ_INSTS_MFMA are exact instruction counts, so this is a property
            # of the instruction stream.  The busy counters are kept next to it: 4 x SQ_ACTIVE_INST_VALU (quad-cycles)
            # + SQ_VALU_MFMA_BUSY_CYCLES adds the matrix lane ON TOP of the issue lane and is not a utilisation.
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs -> elapsed cycles of one XCD x 1024 SIMDs = available cycles.
            valu_n = kk["SQ_INSTS_VALU"] - kk["SQ_INSTS_MFMA"]
            issue_lane = 4.0 * valu_n + 10.0 * kk["SQ_INSTS_MFMA"]
            matrix_lane = 32.0 * kk["SQ_INSTS_MFMA"]
            issue = max(issue_lane, matrix_lane)
            counters = 4.0 * kk["SQ_ACTIVE_INST_VALU"] + kk["SQ_VALU_MFMA_BUSY_CYCLES"]
            simd_cycles = kk["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            j = {"method": issue_name, "K": K_OF.get(method, 0), "rays_per_launch": _rays_per_launch(method, kk.get("launches")),
                 "kernel_name": fk,
                 "kernel_source_digest": digest, "launches_per_frame": (kk.get("launches") or 0) / PMC_FRAMES,
                 "wait_any_frac_of_wave_cycles": (kk["SQ_WAIT_ANY"] / kk["SQ_WAVE_CYCLES"]) if kk.get("SQ_WAVE_CYCLES") else None,
                 "source": f"profiles/{tag}_{method}_pmc_sq.csv (rocprofv3 --pmc, own pass)",
                 "issue_cycles_per_launch": issue,
                 "definition": "max(4 x (SQ_INSTS_VALU - SQ_INSTS_MFMA) + 10 x SQ_INSTS_MFMA, 32 x SQ_INSTS_MFMA): issue lane vs matrix "
                               "lane, constants of benchmarks/issue_sweep_probe.hip",
                 "issue_lane_cycles_per_launch": issue_lane, "matrix_lane_cycles_per_launch": matrix_lane,
                 "valu_insts_per_launch": valu_n, "mfma_insts_per_launch": kk["SQ_INSTS_MFMA"],
                 "valu_active_quad_cycles": kk["SQ_ACTIVE_INST_VALU"], "mfma_busy_cycles": kk["SQ_VALU_MFMA_BUSY_CYCLES"],
                 "simd_cycles_per_launch": simd_cycles, "busy_frac": issue / simd_cycles,
                 "busy_frac_from_busy_counters": counters / simd_cycles,
                 "engine_clock_GHz_under_profiler": kk["GRBM_GUI_ACTIVE"] / 8.0 / (kk["avg_dur_us"] * 1e3)}
            with open(os.path.join(d, f"issue_{issue_name}.json"), "w") as f:
                json.dump(j, f, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "reduce":
        reduce(sys.argv[2], sys.argv[3])
    else:
        summary(sys.argv[2], sys.argv[3])
