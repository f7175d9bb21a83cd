#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6al; mkdir -p $O
for wl in c2; do
  for v in "" "BVGPU_COOP_MIN=1536" "BVGPU_COOP_MIN=3072" "BVGPU_COOP_MIN=4096" "BVGPU_GIANT_MIN=32768" "BVGPU_COOP_WAVES=3072" "BVGPU_COOP_WAVES=2048" "BVGPU_LEVEL_BLOCKS=32768" "BVGPU_WAIT_GIANTS=0" ""; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
