#!/bin/bash
# Experiment: HBM write traffic and timings of the sample-major plane path (UNERF_SAMPLE_MAJOR=1) vs the default
# ray-major outputs.  bash benchmarks/exp_planes_traffic.sh <tag>  ->  gpurun_out/profiles_<tag>/<tag>_planes_*.csv|json
set -u
TAG=${1:-r2_xx}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$DST"
export TMPDIR=/tmp
export UNERF_SAMPLE_MAJOR=1
cd /tmp
for m in active mcdropout; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/planes_${m}_$c" -- \
        python3 "$ROOT/bench.py" --method "$m" --steps 1 --warmup 1 --no-cpu-baseline --no-exact-check > "$OUT/planes_${m}_$c.log" 2>&1
    f=$(find "$OUT/planes_${m}_$c" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 "$ROOT/benchmarks/summarize_pmc.py" reduce "$f" "$DST/${TAG}_planes_${m}_pmc_$c.csv"
  done
  cd "$ROOT" && python3 bench.py --method "$m" --steps 5 --warmup 2 --no-cpu-baseline --no-exact-check > "$DST/${TAG}_planes_${m}_bench.json" 2>/dev/null; cd /tmp
done
grep -h "field_kernel\|composite" "$DST"/${TAG}_planes_*_pmc_*.csv
