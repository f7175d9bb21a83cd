#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6d; mkdir -p $O
for d in 0 32 64 96 128 224 512 1024 1536; do
	env BVGPU_DBG=$d python scripts/ab_time.py c2 10 2>/dev/null | tail -1 | cut -c1-300
done | tee $O/ab.txt
