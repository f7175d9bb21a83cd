#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6q; mkdir -p $O
for wl in c2 c5; do
for v in "" "BVGPU_LEVEL_LISTS_EARLY=2" "BVGPU_LEVEL_LISTS_EARLY=2 BVGPU_PREWALK_LONG=2"; do env AB_NO_PROFILE=1 $v python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-140; done
done | tee $O/ab.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf /tmp/prof_tl; env BVGPU_LEVEL_LISTS_EARLY=2 rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py c2 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_c2_early.txt --back 3 > /dev/null; sed -n 2,36p $R/$O/timeline_c2_early.txt
