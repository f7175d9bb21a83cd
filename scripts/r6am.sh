#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6am; mkdir -p $O
V=$PWD/webgraph_amd/variants
timeout 900 env BVGPU_LIB=$V/libbvgpu_wide.so python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py tests/test_gpu_malformed.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.txt
for wl in cnr30 c2 c5; do
  for v in "" "BVGPU_LIB=$V/libbvgpu_wide.so" "" "BVGPU_LIB=$V/libbvgpu_wide.so"; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | sed "s#$V/##" | cut -c1-150
  done
done | tee $O/ab.txt
