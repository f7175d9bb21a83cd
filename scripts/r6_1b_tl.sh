#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6_1b; mkdir -p $O
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf /tmp/prof_tl; rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/bench.py --steps 3 --warmup 2 --nodes 50000000 --arcs 1000000000 --no-extras --no-pmc --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_1b.txt --back 3 > /dev/null; sed -n 1,60p $R/$O/timeline_1b.txt | cut -c1-110
