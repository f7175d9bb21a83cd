#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6ay; mkdir -p $O
for wl in c5 c2 cnr30; do
  for v in "" "BVGPU_STREAM_PRIO=5" "BVGPU_STREAM_PRIO=4" "BVGPU_STREAM_PRIO=7" ""; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
