#!/bin/bash
# round 6: the copy pass's class boundary (lane / wave per row) -- sweep and timelines
cd "$(dirname "$0")/.."
O=gpurun_out/r6s; mkdir -p $O
for wl in cnr30 c2 c5; do
  for v in "" "BVGPU_COPY_MID_MIN=32" "BVGPU_COPY_MID_MIN=64" "BVGPU_COPY_MID_MIN=256" "BVGPU_COPY_MID_MIN=512"; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
for mm in 32 64; do
rm -rf /tmp/prof_tl; env BVGPU_COPY_MID_MIN=$mm rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py cnr30 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_cnr30_mid$mm.txt --back 3 > /dev/null; grep -E "k_copy|k_parse" $R/$O/timeline_cnr30_mid$mm.txt | cut -c1-100
done
