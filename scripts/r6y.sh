#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6y; mkdir -p $O
for wl in cnr30; do
  for v in "" "BVGPU_LEVEL_LISTS_EARLY=0" "BVGPU_LEVEL_LISTS_EARLY=0 BVGPU_LISTS_ON_B=3" "BVGPU_LEVEL_LISTS_EARLY=0 BVGPU_LISTS_ON_B=2" "BVGPU_PREWALK_LONG=0" "BVGPU_PREWALK_LONG=2"; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
for wl in cnr30; do
rm -rf /tmp/prof_tl; env BVGPU_LEVEL_LISTS_EARLY=0 rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py $wl 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_$wl.txt --back 3 > /dev/null; sed -n 2,40p $R/$O/timeline_$wl.txt | cut -c1-100
done
