#!/bin/bash
# A/B of two builds of libunerf on ONE box (box-to-box spread is +-4 %, larger than most kernel changes):
#   here:        build the variant to compare against into benchmarks/build_probe/libunerf_base.so
#                (e.g. `git stash; hipcc ... -o benchmarks/build_probe/libunerf_base.so csrc/*.hip; git stash pop`)
#   on the box:  bash benchmarks/ab_bench.sh [method] [tag]   -> gpurun_out/ab_<tag>.json
# alternates the "new" library (benchmarks/build_probe/libunerf_new.so when it exists, else the in-tree one -- which
# bench.py rebuilds from the sources if they changed since it was built) and the base library three times each.
cd "$(dirname "$0")/.."
METHOD=${1:-mcdropout}
TAG=${2:-$METHOD}
ALT=$PWD/benchmarks/build_probe/libunerf_base.so
NEW=$PWD/benchmarks/build_probe/libunerf_new.so   # optional: an explicit "new" build (e.g. with -D switches); else the in-tree library
[ -f "$NEW" ] && NEWENV="UNERF_LIB=$NEW" || NEWENV="UNERF_NOP=1"
mkdir -p gpurun_out
for rep in 1 2 3; do
    env $NEWENV python bench.py --method $METHOD --steps 5 --warmup 2 --no-cpu-baseline --no-exact-check 2>/dev/null | tail -1 > gpurun_out/_new_$rep.json
    UNERF_LIB=$ALT python bench.py --method $METHOD --steps 5 --warmup 2 --no-cpu-baseline --no-exact-check 2>/dev/null | tail -1 > gpurun_out/_base_$rep.json
done
python - "$TAG" <<'PY'
import json, sys
out = {}
for tag in ("new", "base"):
    rs = [json.load(open(f"gpurun_out/_{tag}_{i}.json")) for i in (1, 2, 3)]
    out[tag] = {"value": [round(r["value"], 3) for r in rs], "unit": rs[0]["unit"],
                "dominant_kernel_launch_ms": [round(r["roofline"]["avg_launch_ms"], 4) for r in rs],
                "per_kernel_ms_per_frame": rs[-1].get("per_kernel_ms_per_frame")}
json.dump(out, open(f"gpurun_out/ab_{sys.argv[1]}.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_kernel_ms_per_frame"} for k, v in out.items()}))
PY
