#!/bin/bash
# Test of the issue model "kernel time = 4 x VALU + 32 x MFMA cycles" on the K-pass field kernel: the same kernel with
# N extra independent VALU instructions, or N extra MFMAs, per MC-dropout pass (inserted into copies of unerf_nerf.hip by benchmarks/probe_source.py).
#   here:        bash benchmarks/exp_issue_model.sh build
#   on the box:  bash benchmarks/exp_issue_model.sh run    -> gpurun_out/exp_issue_model.json
#                bash benchmarks/exp_issue_model.sh clocks -> gpurun_out/exp_issue_model_clocks.json (GRBM_GUI_ACTIVE per
#                launch of the field kernel for each variant: do the cycles grow while the time stays?)
cd "$(dirname "$0")/.."
B=benchmarks/build_probe
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -I include"
SRC="uncertainty-nerf-gs_amd/csrc/unerf_nerf.hip uncertainty-nerf-gs_amd/csrc/unerf_splat.hip"
if [ "$1" = build ]; then
    mkdir -p $B
    # the probe code lives in benchmarks/probe_source.py, which writes patched COPIES of the product source
    SPLAT=uncertainty-nerf-gs_amd/csrc/unerf_splat.hip
    /opt/rocm/bin/hipcc $FLAGS -o $B/libunerf_v0.so $SRC &
    for v in valu32:--extra-valu=32 valu64:--extra-valu=64 mfma4:--extra-mfma=4 mfma8:--extra-mfma=8; do
        python benchmarks/probe_source.py ${v#*:} -o $B/unerf_nerf_${v%%:*}.hip
    done
    /opt/rocm/bin/hipcc $FLAGS -o $B/libunerf_valu32.so $B/unerf_nerf_valu32.hip $SPLAT &
    /opt/rocm/bin/hipcc $FLAGS -o $B/libunerf_valu64.so $B/unerf_nerf_valu64.hip $SPLAT &
    wait
    /opt/rocm/bin/hipcc $FLAGS -o $B/libunerf_mfma4.so $B/unerf_nerf_mfma4.hip $SPLAT &
    /opt/rocm/bin/hipcc $FLAGS -o $B/libunerf_mfma8.so $B/unerf_nerf_mfma8.hip $SPLAT &
    wait
    exit 0
fi
mkdir -p gpurun_out
if [ "$1" = clocks ]; then
    ROOT=$PWD
    export TMPDIR=/tmp
    cd /tmp
    for v in v0 valu64 mfma8; do
        export UNERF_LIB=$ROOT/$B/libunerf_$v.so
        rm -rf $ROOT/gpurun_out/prof_im_$v
        rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_im_$v -- \
            python3 $ROOT/bench.py --method mcdropout --steps 1 --warmup 1 --no-cpu-baseline --no-exact-check > $ROOT/gpurun_out/prof_im_$v.log 2>&1
        f=$(find $ROOT/gpurun_out/prof_im_$v -name '*counter_collection.csv' | head -1)
        python3 $ROOT/benchmarks/summarize_pmc.py reduce "$f" $ROOT/gpurun_out/_im_pmc_$v.csv
    done
    cd $ROOT
    python - <<'PY'
import csv, json
out = {}
for v in ("v0", "valu64", "mfma8"):
    rec = {}
    for r in csv.DictReader(open(f"gpurun_out/_im_pmc_{v}.csv")):
        if r["kernel"].startswith("field_kernel_mfma16"):
            rec[r["counter"]] = float(r["mean_value"])
            rec["mean_dur_us"] = float(r["mean_dur_us"])
    rec["clock_GHz"] = round(rec["GRBM_GUI_ACTIVE"] / 8 / (rec["mean_dur_us"] * 1e-6) / 1e9, 3)
    out[v] = rec
json.dump(out, open("gpurun_out/exp_issue_model_clocks.json", "w"), indent=1)
print(json.dumps(out))
PY
    exit 0
fi
for rep in 1 2; do
    for v in v0 valu32 valu64 mfma4 mfma8; do
        UNERF_LIB=$PWD/$B/libunerf_$v.so python bench.py --method mcdropout --steps 4 --warmup 2 --no-cpu-baseline --no-exact-check 2>/dev/null | tail -1 > gpurun_out/_im_${v}_$rep.json
    done
done
python - <<'PY'
import json
out = {}
for v in ("v0", "valu32", "valu64", "mfma4", "mfma8"):
    rs = [json.load(open(f"gpurun_out/_im_{v}_{i}.json")) for i in (1, 2)]
    out[v] = {"field_launch_ms": [round(r["roofline"]["avg_launch_ms"], 4) for r in rs]}
base = sum(out["v0"]["field_launch_ms"]) / 2
for v in out:
    out[v]["delta_ms_per_launch"] = round(sum(out[v]["field_launch_ms"]) / 2 - base, 4)
json.dump(out, open("gpurun_out/exp_issue_model.json", "w"), indent=1)
print(json.dumps(out))
PY
