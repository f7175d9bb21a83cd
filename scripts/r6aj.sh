#!/bin/bash
# round 6: the copy pass's lane class on rows staged in LDS -- parity, A/B, timelines
cd "$(dirname "$0")/.."
O=gpurun_out/r6aj; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py tests/test_gpu_malformed.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
for wl in cnr30 c2 c5; do
  for v in "BVGPU_COPY_STAGE=0" "" "BVGPU_COPY_STAGE=0" ""; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
for wl in cnr30 c2; do
rm -rf /tmp/prof_tl; rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py $wl 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_$wl.txt --back 3 > /dev/null; grep -E "k_copy_[lmb]" $R/$O/timeline_$wl.txt | cut -c1-100
done
