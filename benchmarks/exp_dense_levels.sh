#!/bin/bash
# Proposal kernels vs the size cap of the dense x-paired level copies (ops.DENSE_LEVEL_BYTES), same box, same library.
cd "$(dirname "$0")/.."
for rep in 1 2; do for cap in ${CAPS:-6291456 8388608 50331648 314572800}; do
  UNERF_DENSE_LEVEL_BYTES=$cap python bench.py --method ${METHOD:-active} --steps 4 --warmup 2 --no-cpu-baseline --no-exact-check 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_ms_per_frame']; print($cap, round(d['ms_per_step'],2), k['proposal_density_256'], k['proposal_density_96'])"
done; done
