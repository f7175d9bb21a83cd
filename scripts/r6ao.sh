#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6ao; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py tests/test_gpu_malformed.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for wl in cnr30 c2 c5; do
  for v in "BVGPU_LIST_REFS=0" "" "BVGPU_LIST_REFS=0" ""; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
