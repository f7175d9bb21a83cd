#!/bin/bash
# GPU box, round 6: the set-up's order (tiled top scan + giants behind the parse list), the wave class's ticks by phase
cd "$(dirname "$0")/.."
O=gpurun_out/r6b; mkdir -p $O
for wl in c2 c5 cnr30; do
	for v in "" "BVGPU_SCAN_TOP_TILED_MIN=1" "BVGPU_SCAN_TOP_TILED_MIN=1 BVGPU_GIANTS_AFTER_LIST=0" "BVGPU_GIANTS_AFTER_LIST=0"; do
		env AB_NO_PROFILE=1 $v python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
	done
done | tee $O/ab.txt
BVGPU_STATS=1 python scripts/tune.py --reps 3 2>/dev/null | tail -9 | tee $O/tune_c2.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/prof_tl
for v in "" "BVGPU_SCAN_TOP_TILED_MIN=1"; do
	rm -rf /tmp/prof_tl; env $v rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py c2 3 > /dev/null 2>&1
	python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_c2_${v:-default}.txt --back 3 > /dev/null; head -36 $R/$O/timeline_c2_${v:-default}.txt
done
