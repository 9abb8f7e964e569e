#!/bin/bash
# One SQ counter pass over the MC-dropout K=8 bench (experiment tooling): bash benchmarks/pmc_sq_mc.sh <tag>
set -u
TAG=${1:-r2_xx}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$DST"
export TMPDIR=/tmp
cd /tmp
pmc() {  # method, set name, counters...
    local m=$1 name=$2; shift 2
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/${m}_pmc_$name" -- \
        python3 "$ROOT/bench.py" --method "$m" --steps 1 --warmup 1 --no-cpu-baseline --no-exact-check > "$OUT/${m}_pmc_$name.log" 2>&1
    local f
    f=$(find "$OUT/${m}_pmc_$name" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 "$ROOT/benchmarks/summarize_pmc.py" reduce "$f" "$DST/${TAG}_${m}_pmc_$name.csv"
}
pmc mcdropout sq SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc mcdropout lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE
grep -h "field_kernel" "$DST"/*.csv
