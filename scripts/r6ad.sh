#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6ad; mkdir -p $O
export TMPDIR=/tmp; R=$PWD; cd /tmp
for cl in 0 1; do
rm -rf /tmp/prof_tl; env BVGPU_LEVEL_BINS=0 BVGPU_COPY_LOOP=$cl rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/bench.py --mode random --steps 3 --warmup 2 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_c4_loop$cl.txt --back 3 > /dev/null; echo "== loop $cl"; sed -n 1,70p $R/$O/timeline_c4_loop$cl.txt | cut -c1-100
done
