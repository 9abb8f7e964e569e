#!/bin/bash
# A/B/C... of K-pass field-kernel variants on ONE box (box-to-box spread is +-4 %).  Each variant is its own build of
# the library (-DUNERF_<X>=0/1) used through UNERF_LIB; the host asks unerf_build_flags() how to pack the operands.
# (The run recorded in profiles/r2_exp_kpass_variants.json also had a ReLU-as-fma(|x|, 1, x) variant "relu2", since
# removed: +1.9 %.)
#   here:        bash benchmarks/exp_kpass_variants.sh build
#   on the box:  bash benchmarks/exp_kpass_variants.sh run [method]   -> gpurun_out/exp_kpass_variants_<method>.json
cd "$(dirname "$0")/.."
B=benchmarks/build_probe
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -I include"
SRC="uncertainty-nerf-gs_amd/csrc/unerf_nerf.hip uncertainty-nerf-gs_amd/csrc/unerf_splat.hip"
VARIANTS="shipped nofold noresident neither"
defs() {
    case $1 in
        shipped) echo "";;
        nofold) echo "-DUNERF_TRUNK_FOLD=0";;
        noresident) echo "-DUNERF_TRUNK_RESIDENT=0";;
        neither) echo "-DUNERF_TRUNK_FOLD=0 -DUNERF_TRUNK_RESIDENT=0";;
    esac
}
if [ "$1" = build ]; then
    mkdir -p $B
    for v in $VARIANTS; do /opt/rocm/bin/hipcc $FLAGS $(defs $v) -o $B/libunerf_kp_$v.so $SRC & done
    wait
    exit 0
fi
METHOD=${2:-mcdropout}
mkdir -p gpurun_out
for rep in 1 2 3; do
    for v in $VARIANTS; do
        UNERF_LIB=$PWD/$B/libunerf_kp_$v.so python bench.py --method $METHOD --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/_kp_${v}_$rep.json
    done
done
python - "$METHOD" $VARIANTS <<'PY'
import json, sys
method, variants = sys.argv[1], sys.argv[2:]
out = {}
for v in variants:
    rs = [json.load(open(f"gpurun_out/_kp_{v}_{i}.json")) for i in (1, 2, 3)]
    out[v] = {"field_launch_ms": [round(r["roofline"]["avg_launch_ms"], 4) for r in rs], "value": [round(r["value"], 2) for r in rs],
              "max_abs_rgb_diff_vs_exact_fp32": rs[0].get("exact_fp32_kernels", {}).get("max_abs_rgb_diff_vs_split_f16")}
base = sum(out["neither"]["field_launch_ms"]) / 3
for v in out:
    out[v]["delta_pct"] = round(100 * (sum(out[v]["field_launch_ms"]) / 3 / base - 1), 2)
json.dump(out, open(f"gpurun_out/exp_kpass_variants_{method}.json", "w"), indent=1)
print(json.dumps(out))
PY
