#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6f; mkdir -p $O
for v in "" "BVGPU_COOP_MIN=1024" "BVGPU_COOP_MIN=3072" "BVGPU_COOP_MIN=4096" "BVGPU_GIANT_MIN=32768" "BVGPU_GIANT_MIN=131072" "BVGPU_COOP_WAVES=3072" "BVGPU_COOP_WAVES=8192" "BVGPU_LEVEL_BLOCKS=8192" "BVGPU_LEVEL_BLOCKS=32768"; do env AB_NO_PROFILE=1 $v python scripts/ab_time.py c2 20 2>/dev/null | tail -1 | cut -c1-140; done | tee $O/ab.txt
