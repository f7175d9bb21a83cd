#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6z; mkdir -p $O
for wl in c2 cnr30 c5; do
  for v in "" "BVGPU_EXP_MID_BLOCKS=768" "BVGPU_EXP_MID_BLOCKS=512" "BVGPU_EXP_MID_BLOCKS=2048" "BVGPU_LEVEL_BINS=0" ""; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
