#!/bin/bash
# timing experiments (a -DBV_EXP_TIMING tuning build: scripts/variants.sh -s bv_kernels.hip timing "-DBV_EXP_TIMING", BVGPU_LIB=...; results are garbage with any switch set): what each step of the wave class's residual phase costs -- BVGPU_DBG bits: 0x800 stage only, 0x1000 no run-in,
# 0x2000 no rounds, 128 no value pass, 0x4000 no residual phase, 0x8000 no interval expansion
cd "$(dirname "$0")/.."
O=gpurun_out/r6g; mkdir -p $O
for d in 0 2048 4096 8192 128 12416 16384 32768 49152; do
	env BVGPU_DBG=$d python scripts/ab_time.py c2 10 2>/dev/null | tail -1 | sed 's/.*| scan/scan/' | cut -c1-200 | sed "s/^/dbg=$d /"
done | tee $O/ab.txt
