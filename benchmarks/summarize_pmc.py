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


FIELD_KERNELS = {"active": ("field_kernel_mfma16<0, false, false>", "field_kernel_mfma16<0, false>", "field_kernel_mfma<0, false, false>",
                            "field_kernel_mfma<0, false>"),
                 "mcdropout": ("field_kernel_mfma16<1, false, false, true>", "field_kernel_mfma16<1, false, false>", "field_kernel_mfma16<1, false>", "field_kernel_mfma<1, false, false>",
                               "field_kernel_mfma<1, false>")}


def summary(d, tag):
    for method in ("active", "mcdropout"):
        kernels = defaultdict(dict)
        for fn in sorted(os.listdir(d)):
            m = re.match(rf"{tag}_{method}_pmc_(\w+)\.csv$", fn)
            if not m:
                continue
            with open(os.path.join(d, fn), newline="") as f:
                for row in csv.DictReader(f):
                    kernels[row["kernel"]][row["counter"]] = float(row["mean_value"])
                    kernels[row["kernel"]].setdefault("avg_dur_us", float(row["mean_dur_us"]))
        if not kernels:
            continue
        out = {"command": "benchmarks/collect_profiles.sh (one rocprofv3 --pmc pass per counter set, --kernel-trace only)",
               "units": "per-launch means; FETCH_SIZE/WRITE_SIZE in KB; SQ_* busy/wait counters in quad-cycles; "
                        "GRBM_GUI_ACTIVE summed over 8 XCDs",
               "kernels": kernels}
        with open(os.path.join(d, f"{tag}_{method}_pmc_summary.json"), "w") as f:
            json.dump(out, f, indent=1)
        fk = next((k for k in FIELD_KERNELS[method] if k in kernels), None)
        if fk and "FETCH_SIZE" in kernels[fk] and "WRITE_SIZE" in kernels[fk]:
            t = {"method": method, "K": 8 if method == "mcdropout" else 0, "rays_per_launch": 262144,
                 "source": f"profiles/{tag}_{method}_pmc_fetch.csv + {tag}_{method}_pmc_write.csv "
                           "(separate rocprofv3 --pmc passes)",
                 "kernels": {"field_fwd": {
                     "kernel_name": fk,
                     "fetch_bytes": kernels[fk]["FETCH_SIZE"] * 1024.0,
                     "write_bytes": kernels[fk]["WRITE_SIZE"] * 1024.0,
                     "correction": "none: the reads are 8-byte gathers (one 64-B fabric request each), not the wide "
                                   "coalesced streams for which MI355X_MICROARCH.md prescribes x2; writes read exact"}}}
            with open(os.path.join(d, f"traffic_{method}.json"), "w") as f:
                json.dump(t, f, indent=1)
        need = ("SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE")
        if fk and all(c in kernels[fk] for c in need):
            kk = kernels[fk]
            # SQ_ACTIVE_INST_VALU counts quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (32 per v_mfma_f32_32x32x16_f16);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs -> elapsed cycles of one XCD x 1024 SIMDs = available cycles
            issue = 4.0 * kk["SQ_ACTIVE_INST_VALU"] + kk["SQ_VALU_MFMA_BUSY_CYCLES"]
            simd_cycles = kk["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            j = {"method": method, "K": 8 if method == "mcdropout" else 0, "rays_per_launch": 262144, "kernel_name": fk,
                 "source": f"profiles/{tag}_{method}_pmc_sq.csv (rocprofv3 --pmc, own pass)",
                 "issue_cycles_per_launch": issue,
                 "valu_active_quad_cycles": kk["SQ_ACTIVE_INST_VALU"], "mfma_busy_cycles": kk["SQ_VALU_MFMA_BUSY_CYCLES"],
                 "valu_insts_per_launch": kk["SQ_INSTS_VALU"] - kk["SQ_INSTS_MFMA"], "mfma_insts_per_launch": kk["SQ_INSTS_MFMA"],
                 "simd_cycles_per_launch": simd_cycles, "busy_frac": issue / simd_cycles,
                 "engine_clock_GHz_under_profiler": kk["GRBM_GUI_ACTIVE"] / 8.0 / (kk["avg_dur_us"] * 1e3)}
            with open(os.path.join(d, f"issue_{method}.json"), "w") as f:
                json.dump(j, f, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "reduce":
        reduce(sys.argv[2], sys.argv[3])
    else:
        summary(sys.argv[2], sys.argv[3])
