#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6ak; mkdir -p $O
V=$PWD/webgraph_amd/variants
for wl in cnr30 c2 c5; do
  for v in "" "BVGPU_LIB=$V/libbvgpu_w8.so" "" "BVGPU_LIB=$V/libbvgpu_w8.so"; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | sed "s#$V/##" | cut -c1-150
  done
done | tee $O/ab.txt
