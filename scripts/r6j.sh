#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6j; mkdir -p $O
for lib in "" d3 d4 d3s g3; do
	for wl in c2 c5; do
		env ${lib:+BVGPU_LIB=$PWD/webgraph_amd/variants/libbvgpu_$lib.so} python scripts/ab_time.py $wl 15 2>/dev/null | tail -1 | sed 's/.*thr/thr/' | cut -c1-260 | sed "s/^/${lib:-default} $wl /"
	done
done | tee $O/ab.txt
