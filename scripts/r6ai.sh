#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6ai; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.txt
for wl in cnr30; do
  for v in "BVGPU_WAVES_ON_B=0" "" "BVGPU_WAVES_ON_B=0" "" "BVGPU_LEVEL_LISTS_EARLY=0"; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf /tmp/prof_tl; rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py cnr30 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_cnr30.txt --back 3 > /dev/null; sed -n 2,40p $R/$O/timeline_cnr30.txt | cut -c1-100
