#!/bin/bash
# A/B of one environment switch on one box, alternating runs (clock / thermal drift cancels):
#   bash benchmarks/ab_env.sh <VAR> <valueA> <valueB> <repeats> <out.jsonl> -- <bench.py arguments>
# Appends one JSON object per run: {"var", "value", "ms_per_step", "value_mrays", "per_kernel_ms_per_frame"}.
VAR=$1; A=$2; B=$3; REP=$4; OUT=$5; shift 6
for i in $(seq 1 "$REP"); do
  for v in "$A" "$B"; do
    env "$VAR=$v" python3 bench.py "$@" --no-cpu-baseline --no-exact-check --no-sub-records 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'var': '$VAR', 'value': '$v', 'ms_per_step': d['ms_per_step'], 'value_mrays': d['value'],
                  'per_kernel_ms_per_frame': d['roofline']['per_kernel_ms_per_frame']}))" >> "$OUT"
  done
done
