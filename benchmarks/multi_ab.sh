#!/bin/bash
# Same-box comparison of several explicit builds of libunerf (UNERF_LIB), three alternating repetitions each.
# usage: [BENCH_ARGS="--precision f16x2"] multi_ab.sh method lib1 lib2 ... (names in benchmarks/build_probe/libunerf_<name>.so)
cd "$(dirname "$0")/.."
M=$1; shift
mkdir -p gpurun_out
for rep in 1 2 3; do for v in "$@"; do
  UNERF_LIB=$PWD/benchmarks/build_probe/libunerf_$v.so python bench.py --method $M --steps 4 --warmup 2 --no-cpu-baseline --no-exact-check ${BENCH_ARGS:-} 2>/dev/null | tail -1 > gpurun_out/_m_${v}_$rep.json
done; done
python - "$@" <<'PY'
import json, sys
out={}
for v in sys.argv[1:]:
    rs=[json.load(open(f"gpurun_out/_m_{v}_{i}.json")) for i in (1,2,3)]
    out[v]={"dominant_launch_ms": [round(r["roofline"]["avg_launch_ms"],4) for r in rs], "frame_ms": [round(r["ms_per_step"],3) for r in rs],
            "per_kernel_ms_per_frame": {k: [round(r["roofline"]["per_kernel_ms_per_frame"][k],3) for r in rs] for k in rs[0]["roofline"]["per_kernel_ms_per_frame"]}}
print(json.dumps(out))
json.dump(out, open("gpurun_out/multi_ab.json","w"))
PY
