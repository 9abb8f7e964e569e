#!/bin/bash
# Upper bound on what staging the coarsest proposal level (17^3 cells) in LDS could gain: the proposal
# kernels with level 0 computed WITHOUT its four gathers, its index arithmetic and its blend (benchmarks/probe_source.py
# --no-prop-level0; wrong results) against the shipped ones, same box.  An LDS-staged level would still pay the index
# arithmetic, the blend and four ds_read_b128 per sample, plus 39-78 KB of staging per workgroup -- so it can gain at most
# what this probe gains.
#   here:        bash benchmarks/exp_prop_level0.sh build
#   on the box:  bash benchmarks/exp_prop_level0.sh run     -> gpurun_out/multi_ab.json
cd "$(dirname "$0")/.."
B=benchmarks/build_probe
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -I include"
SRC="uncertainty-nerf-gs_amd/csrc/unerf_nerf.hip uncertainty-nerf-gs_amd/csrc/unerf_splat.hip"
if [ "$1" = build ]; then
    mkdir -p $B
    /opt/rocm/bin/hipcc $FLAGS -o $B/libunerf_base.so $SRC &
    python benchmarks/probe_source.py --no-prop-level0 -o $B/unerf_nerf_nol0.hip   # a patched COPY of the product source
    /opt/rocm/bin/hipcc $FLAGS -o $B/libunerf_nol0.so $B/unerf_nerf_nol0.hip uncertainty-nerf-gs_amd/csrc/unerf_splat.hip &
    wait
    exit 0
fi
bash benchmarks/multi_ab.sh active base nol0
