#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6aq; mkdir -p $O
V=$PWD/webgraph_amd/variants
for i in 1 2 3; do
  for v in "BVGPU_LIB=$V/libbvgpu_head.so" ""; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py cnr30 20 2>/dev/null | tail -1 | sed "s#$V/##" | cut -c1-150
  done
done | tee $O/ab.txt
