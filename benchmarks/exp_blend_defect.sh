#!/bin/bash
# DESIGN 4.5, the fused-blend defect: poison experiments (benchmarks/exp_blend_defect.py) on the bad build
# (libunerf_bf.so = -DUNERF_FIELD_BLEND_FMA=1, built next to the default library) and on the default build.
OUT=gpurun_out/r5_exp_blend_defect.jsonl
BF=$PWD/uncertainty-nerf-gs_amd/csrc/libunerf_bf.so
for kind in active mcdropout; do
  for p in none nan zero ones none; do
    UNERF_LIB=$BF timeout 300 python3 benchmarks/exp_blend_defect.py --poison $p --kind $kind >> $OUT 2>> gpurun_out/r5_exp_blend_defect.err
  done
done
for p in none nan; do
  timeout 300 python3 benchmarks/exp_blend_defect.py --poison $p --kind active >> $OUT 2>> gpurun_out/r5_exp_blend_defect.err
done
cat $OUT
