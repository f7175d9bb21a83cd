#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6k; mkdir -p $O
for wl in c2 c5; do
for v in "" "BVGPU_GIANT_MIN=32768" "BVGPU_GIANT_MIN=16384" "BVGPU_GIANT_GROUPS=512" "BVGPU_GIANT_MIN=32768 BVGPU_GIANT_GROUPS=512" "BVGPU_COOP_MIN=4096 BVGPU_GIANT_MIN=32768"; do env AB_NO_PROFILE=1 $v python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-140; done
done | tee $O/ab.txt
