#!/bin/bash
# round 6: the tile kernel through the wave's loop once more, with a ring of four intervals and a register bound (six / seven blocks per CU)
cd "$(dirname "$0")/.."
O=gpurun_out/r6as; mkdir -p $O
V=$PWD/webgraph_amd/variants
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py tests/test_gpu_malformed.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for i in 1 2 3; do
  for v in "BVGPU_LIB=$V/libbvgpu_head.so" "" "BVGPU_LIB=$V/libbvgpu_t7.so"; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py cnr30 20 2>/dev/null | tail -1 | sed "s#$V/##" | cut -c1-150
  done
done | tee $O/ab.txt
for v in "BVGPU_LIB=$V/libbvgpu_head.so" "" "BVGPU_LIB=$V/libbvgpu_t7.so"; do env $v timeout 600 python scripts/ab_time.py cnr30 10 2>/dev/null | tail -1 | cut -c100-330; done | tee -a $O/ab.txt
