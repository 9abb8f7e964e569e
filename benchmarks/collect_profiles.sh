#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun):
#   bash benchmarks/collect_profiles.sh <tag>          e.g. r1_05
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and the condensed, tracked-size files under
# gpurun_out/profiles_<tag>/ (copy those into profiles/ and commit).  Kernel-trace/--stats and every
# --pmc counter set run as SEPARATE passes (never combined with other trace domains); the profiled
# program is `python3 bench.py ...` directly after `--`.
set -u
TAG=${1:-r1_xx}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$DST"
export TMPDIR=/tmp
cd /tmp

margs() {  # profile id -> bench.py arguments
    case $1 in
        # profile ids name the arithmetic: <method> = the split-f16 (fp32-equivalent) kernels, <method>_f16 = one f16
        # product per MAC (bench.py's default for mcdropout and active: the reference's own eval precision)
        mcdropout) echo "--method mcdropout --precision f16x2" ;;
        mcdropout_f16) echo "--method mcdropout --precision f16" ;;
        active) echo "--method active --precision f16x2" ;;
        active_f16) echo "--method active --precision f16" ;;
        # the headline's workload on tcnn-layout tables: half2 rows + tcnn's half interpolation (the reference's default
        # implementation), and fp32 rows (FETCH_SIZE before / after)
        mcdropout_f16_tcnn) echo "--method mcdropout --precision f16 --grid tcnn --grid-precision f16" ;;
        mcdropout_f16_tcnn32) echo "--method mcdropout --precision f16 --grid tcnn --grid-precision f32" ;;
        *) echo "--method $1" ;;
    esac
}

stats() {  # profile id
    local m=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${m}_stats" -- \
        python3 "$ROOT/bench.py" $(margs $m) --steps 4 --warmup 1 --no-cpu-baseline --no-exact-check "$@" > "$OUT/${m}_stats.log" 2>&1
    local f
    f=$(find "$OUT/${m}_stats" -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp "$f" "$DST/${TAG}_${m}_kernel_stats.csv"
    grep -a "^{\"metric\"" "$OUT/${m}_stats.log" | tail -1 > "$DST/${TAG}_${m}_bench_under_rocprof.json"
}

pmc() {  # method, set name, counters...
    local m=$1 name=$2; shift 2
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/${m}_pmc_$name" -- \
        python3 "$ROOT/bench.py" $(margs $m) --steps 1 --warmup 1 --no-cpu-baseline --no-exact-check > "$OUT/${m}_pmc_$name.log" 2>&1
    local f
    f=$(find "$OUT/${m}_pmc_$name" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 "$ROOT/benchmarks/summarize_pmc.py" reduce "$f" "$DST/${TAG}_${m}_pmc_$name.csv"
}

# ONLY="id id ..." restricts the collection (a supplementary run for profile ids added later)
ALL=${ONLY:-"active active_f16 mcdropout mcdropout_f16 mcdropout_f16_tcnn mcdropout_f16_tcnn32 laplace splat"}
for m in $ALL; do stats $m; done
for m in $ALL; do
    pmc $m fetch FETCH_SIZE
    pmc $m write WRITE_SIZE
done
for m in $ALL; do
    pmc $m sq SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
done
if [ -z "${ONLY:-}" ]; then
pmc splat lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES
pmc active ta TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pmc active tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
fi
python3 "$ROOT/benchmarks/summarize_pmc.py" summary "$DST" "$TAG"
# the default line (what the driver runs), outside the profiler -- with THIS run's traffic_* / issue_* profiles in place, so that
# its roofline block reads counters of the sources it times (bench.py ignores profiles of another source digest)
cp "$DST"/traffic_*.json "$DST"/issue_*.json "$ROOT/profiles/" 2>/dev/null
cd "$ROOT" && python3 bench.py > "$DST/${TAG}_default_bench.json" 2> "$OUT/default_bench.err"
ls -la "$DST"
