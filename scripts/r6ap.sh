#!/bin/bash
# round 6: the tile kernel's records through the wave's loop (parse_node_lwc over the tile's image)
cd "$(dirname "$0")/.."
O=gpurun_out/r6ap; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py tests/test_gpu_malformed.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
for wl in cnr30; do
  for v in "BVGPU_TILE_LOOP=0" "" "BVGPU_TILE_LOOP=0" ""; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
for v in "BVGPU_TILE_LOOP=0" ""; do env $v timeout 600 python scripts/ab_time.py cnr30 10 2>/dev/null | tail -1 | cut -c100-400; done | tee -a $O/ab.txt
for v in "BVGPU_TILE=1 BVGPU_TILE_LOOP=0" "BVGPU_TILE=1"; do env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py c2 10 2>/dev/null | tail -1 | cut -c1-150; done | tee -a $O/ab.txt
