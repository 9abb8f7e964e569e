#!/bin/bash
# PMC passes over the splat frame for the tile-sort kernels (run through gpurun): bash benchmarks/pmc_splat_sort.sh <tag> [env...]
# Separate rocprofv3 --pmc passes (never combined with other trace domains); prints per-kernel means for the sort kernels.
TAG=${1:-x}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_sort_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for e in "$@"; do export "$e"; done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --method splat --steps 1 --warmup 1 --no-cpu-baseline --no-exact-check > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if any(k in n for k in ("tile_scatter", "rs_scatter", "rs_hist", "map_intersects", "sorted_counts", "raster_kernel")):
        acc[n.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    print(n, {k: round(sum(v) / len(v)) for k, v in c.items()})
PY
done
