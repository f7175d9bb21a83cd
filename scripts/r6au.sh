#!/bin/bash
# round 6: k_copy_mid builds its tables from the one-lane parse's CopyTab for the rows the pre-walk left
cd "$(dirname "$0")/.."
O=gpurun_out/r6au; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py tests/test_gpu_malformed.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for wl in c2 c5 cnr30; do
  for v in "BVGPU_MID_TABLES=0" "" "BVGPU_MID_TABLES=0" ""; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf /tmp/prof_tl; rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py c2 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_c2.txt --back 3 > /dev/null; grep -E "k_copy_[lmb]" $R/$O/timeline_c2.txt | cut -c1-100
