#!/bin/bash
# rocprofv3 kernel stats of the splat frame, sort kernels only (run through gpurun): bash benchmarks/splat_stats.sh <tag>
TAG=${1:-x}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -- python3 $ROOT/bench.py --method splat --steps 5 --warmup 1 --no-cpu-baseline --no-exact-check > /dev/null 2>&1
f=$(find $ROOT/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv,sys
tot = 0.0
for r in list(csv.DictReader(open(sys.argv[1]))):
    if any(k in r["Name"] for k in ("rs_","sorted_counts","depth_keys","tile_","map_inter","rocprim","project_kernel","scan_")):
        per_frame = float(r["AverageNs"]) * int(r["Calls"]) / 6 / 1e3
        tot += per_frame
        print(r["Name"][:80].ljust(80), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us; per frame", round(per_frame, 1))
print("bin-and-sort kernels per frame:", round(tot, 1), "us")
PY
