#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py tests/test_gpu_boundary.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for wl in c2 c5 cnr30; do
	for v in "" "BVGPU_KEYS_IN_HEADERS=0" "BVGPU_SCAN_TOP_TILED_MIN=1"; do env AB_NO_PROFILE=1 $v python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-140; done
done | tee $O/ab.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py c2 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_c2.txt --back 3 > /dev/null; head -22 $R/$O/timeline_c2.txt
