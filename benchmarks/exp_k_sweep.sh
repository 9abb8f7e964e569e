#!/bin/bash
# Field-kernel time per launch as a function of the number of MC-dropout passes K (fit: per-tile prologue + K x pass).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for K in 1 2 4 8 12 16; do
    python bench.py --method mcdropout --mc-samples $K --steps 3 --warmup 1 --no-cpu-baseline --no-exact-check 2>/dev/null | tail -1 > gpurun_out/_ks_$K.json
done
python - <<'PY'
import json
out = {}
for K in (1, 2, 4, 8, 12, 16):
    r = json.load(open(f"gpurun_out/_ks_{K}.json"))
    out[K] = {"field_launch_ms": round(r["roofline"]["avg_launch_ms"], 4), "frame_ms": round(r["ms_per_step"], 3)}
ks = sorted(out)
import numpy as np
A = np.array([[1, k] for k in ks], dtype=float)
y = np.array([out[k]["field_launch_ms"] for k in ks])
(a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
res = {"per_K": out, "fit_prologue_ms": round(float(a), 4), "fit_ms_per_pass": round(float(b), 4)}
json.dump(res, open("gpurun_out/exp_k_sweep.json", "w"), indent=1)
print(json.dumps(res))
PY
