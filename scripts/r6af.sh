#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6af; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for v in "" "BVGPU_LEVEL_BINS=0" "BVGPU_COPY_LOOP=0 BVGPU_LEVEL_BINS=0" "" "BVGPU_LEVEL_BINS=0"; do env $v python scripts/c4_time.py 20 2>/dev/null | tail -1; done | tee $O/ab.txt
