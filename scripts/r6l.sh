#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py tests/test_gpu_malformed.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for lib in "" w1old; do
	for wl in c2 c5 cnr30; do
		env ${lib:+BVGPU_LIB=$PWD/webgraph_amd/variants/libbvgpu_$lib.so} python scripts/ab_time.py $wl 15 2>/dev/null | tail -1 | sed 's/.*thr/thr/' | cut -c1-260 | sed "s/^/${lib:-straight} $wl /"
	done
done | tee $O/ab.txt
