#!/bin/bash
# round 6: the wave class in two kernels (head: header/blocks/intervals, tail: residuals + ids) -- parity, then A/B with smaller LDS footprints
cd "$(dirname "$0")/.."
O=gpurun_out/r6r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py tests/test_gpu_malformed.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.txt
V=$PWD/webgraph_amd/variants
for wl in c2 c5 cnr30; do
  for v in "BVGPU_WAVE_SPLIT=0" "" "BVGPU_LIB=$V/libbvgpu_ck12.so BVGPU_WAVE_SPLIT=0" "BVGPU_LIB=$V/libbvgpu_ck12.so" "BVGPU_LIB=$V/libbvgpu_ck12b256.so" "BVGPU_LIB=$V/libbvgpu_ck12iv256.so" "BVGPU_LIB=$V/libbvgpu_ck10b256iv256.so" "BVGPU_LIB=$V/libbvgpu_ck10b256iv256.so BVGPU_WAVE_SPLIT=0"; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | sed "s#$V/##" | cut -c1-150
  done
done | tee $O/ab.txt
env timeout 600 python scripts/ab_time.py c2 10 2>/dev/null | tail -40 > $O/serial_c2.txt
grep -E "k_parse_wave_head|k_wave_res|k_parse_big|k_parse_list" $O/serial_c2.txt
