#!/bin/bash
cd "$(dirname "$0")/.."
V=$PWD/webgraph_amd/variants
for dbg in 0 1048576 2097152 3145728; do env BVGPU_LIB=$V/libbvgpu_timing.so BVGPU_DBG=$dbg timeout 600 python scripts/ab_time.py cnr30 10 2>/dev/null | tail -1 | cut -c1-30,100-330; done
