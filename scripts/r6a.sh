#!/bin/bash
# GPU box, round 6: parity of the new one-lane loop + copy tables, then A/B timings against round 4's loop.
cd "$(dirname "$0")/.."
O=gpurun_out/r6a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_malformed.py tests/test_gpu_random.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for wl in c2 c5 cnr30; do
	for v in "" "BVGPU_LANE_LOOP=0 BVGPU_COPY_TABLES=0" "BVGPU_COPY_TABLES=0"; do
		env $v python scripts/ab_time.py $wl 10 2>/dev/null | tail -1
	done
done | tee $O/ab.txt
bash scripts/kstats.sh r6a_c2 c2 > /dev/null 2>&1; cp gpurun_out/kstats_r6a_c2.txt $O/ 2>/dev/null; head -30 $O/kstats_r6a_c2.txt | cut -c1-150
