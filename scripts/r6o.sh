#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6o; mkdir -p $O
for lib in "" b384 b768 b1024 ck12 ck32; do
	for wl in c2 c5; do
		env ${lib:+BVGPU_LIB=$PWD/webgraph_amd/variants/libbvgpu_$lib.so} python scripts/ab_time.py $wl 15 2>/dev/null | tail -1 | sed 's/.*thr/thr/' | cut -c1-260 | sed "s/^/${lib:-default} $wl /"
	done
done | tee $O/ab.txt
