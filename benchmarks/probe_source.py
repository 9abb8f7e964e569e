#!/usr/bin/env python
"""Experiment builds of csrc/unerf_nerf.hip WITHOUT experiment code in the product translation unit.

The product source carries only marker comments (`// [probe:<name>]`); this script writes a COPY of it with a snippet
inserted at (or substituted for the span of) a marker, for the measurement scripts that used to pass -DUNERF_PROBE_*:

    python benchmarks/probe_source.py --extra-valu 32 -o /tmp/unerf_nerf_valu32.hip      # benchmarks/exp_issue_model.sh
    python benchmarks/probe_source.py --extra-mfma 8 -o /tmp/unerf_nerf_mfma8.hip
    python benchmarks/probe_source.py --no-prop-mlp -o /tmp/unerf_nerf_nomlp.hip         # benchmarks/exp_prop_mlp_bound.sh
    python benchmarks/probe_source.py --no-prop-level0 -o /tmp/unerf_nerf_nol0.hip       # benchmarks/exp_prop_level0.sh

The copies produce WRONG or slower results by design; nothing in the package builds or loads them."""
import argparse
import os
import re

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uncertainty-nerf-gs_amd", "csrc", "unerf_nerf.hip")


def extra_valu(n):
    return f"""            {{   // probe: {n} independent VALU instructions per MC-dropout pass
                uint32_t dummy = (uint32_t)k;
#pragma unroll
                for (int q = 0; q < {n}; ++q) asm volatile("v_alignbit_b32 %0, %0, %0, 5" : "+v"(dummy));
                asm volatile("" ::"v"(dummy));
            }}
"""


def extra_mfma(n):
    return f"""            {{   // probe: {n} independent MFMAs per MC-dropout pass
                f32x16 junk = {{0}};
#pragma unroll
                for (int q = 0; q < {n}; ++q) junk = __builtin_amdgcn_mfma_f32_32x32x16_f16(hhi[0], hhi[1], junk, 0, 0, 0);
                asm volatile("" ::"v"(junk));
            }}
"""


NO_PROP_MLP = """        (void)h2; (void)w1t;   // probe: the proposal MLP's output layer replaced by ten adds (wrong results)
#pragma unroll
        for (int k = 0; k < 2 * L; ++k) o += feat[k];
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--extra-valu", type=int, default=0)
    ap.add_argument("--extra-mfma", type=int, default=0)
    ap.add_argument("--no-prop-mlp", action="store_true")
    ap.add_argument("--no-prop-level0", action="store_true",
                    help="proposal kernels: level 0 of the grid without its gathers (upper bound of what LDS staging could save)")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    s = open(SRC).read()
    if a.extra_valu or a.extra_mfma:
        marker = "            // [probe:kpass-pass-start]\n"
        assert s.count(marker) == 1
        s = s.replace(marker, marker + (extra_valu(a.extra_valu) if a.extra_valu else "") + (extra_mfma(a.extra_mfma) if a.extra_mfma else ""))
    if a.no_prop_mlp:
        s, n = re.subn(r"        // \[probe:prop-mlp-out begin\].*?// \[probe:prop-mlp-out end\]\n", NO_PROP_MLP, s, flags=re.S)
        assert n == 1
    if a.no_prop_level0:
        pat = re.compile(r"^( *)\} else if \(l < a\.net\.n_dense\) \{  // wave-uniform: coarse level with a dense, x-paired copy\n", re.M)
        s, n = pat.subn(lambda m: f"{m.group(1)}}} else if (l == 0) {{  // probe: no gathers for level 0 (wrong results)\n"
                                  f"{m.group(1)}    f = make_float2(px * py, pz);\n" + m.group(0), s)
        assert n == 2      # prop_density_kernel and prop_patch_kernel
    # the copy lives outside csrc/: point its includes back at the product headers
    s = s.replace('#include "unerf_common.hpp"', f'#include "{os.path.join(os.path.dirname(SRC), "unerf_common.hpp")}"')
    with open(a.out, "w") as f:
        f.write(s)


if __name__ == "__main__":
    main()
