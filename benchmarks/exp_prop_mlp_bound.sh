#!/bin/bash
# Upper bound on what moving the proposal MLP (10 -> 16 -> 1, 112 of ~540 VALU instructions per sample) to the matrix
# pipe could gain: the proposal kernel with the MLP REMOVED (benchmarks/probe_source.py --no-prop-mlp: ten adds instead; wrong
# results) against the shipped one, same box.
#   here:        bash benchmarks/exp_prop_mlp_bound.sh build
#   on the box:  bash benchmarks/exp_prop_mlp_bound.sh run     -> gpurun_out/multi_ab.json
cd "$(dirname "$0")/.."
B=benchmarks/build_probe
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -I include"
SRC="uncertainty-nerf-gs_amd/csrc/unerf_nerf.hip uncertainty-nerf-gs_amd/csrc/unerf_splat.hip"
if [ "$1" = build ]; then
    mkdir -p $B
    /opt/rocm/bin/hipcc $FLAGS -o $B/libunerf_base.so $SRC &
    python benchmarks/probe_source.py --no-prop-mlp -o $B/unerf_nerf_nomlp.hip   # a patched COPY of the product source
    /opt/rocm/bin/hipcc $FLAGS -o $B/libunerf_nomlp.so $B/unerf_nerf_nomlp.hip uncertainty-nerf-gs_amd/csrc/unerf_splat.hip &
    wait
    exit 0
fi
bash benchmarks/multi_ab.sh active base nomlp
