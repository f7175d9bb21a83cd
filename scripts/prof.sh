#!/bin/bash
# GPU box: rocprofv3 kernel stats of the tuning scan.  usage: scripts/prof.sh <tag> [env assignments...]
# writes gpurun_out/prof_<tag>.txt (per-kernel stats) and gpurun_out/timeline_<tag>.txt (last scan, kernel by kernel)
# streams are serialised (BVGPU_OVERLAP=0) unless BVGPU_OVERLAP=1 is passed
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_$tag
env BVGPU_OVERLAP=0 TUNE_NO_PROFILE=1 "$@" rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o res -- python $R/scripts/tune.py --reps 3 > /tmp/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $R/scripts/rocprof_summary.py $db $R/gpurun_out/prof_$tag.txt
python $R/scripts/timeline.py $db $R/gpurun_out/timeline_$tag.txt
head -16 $R/gpurun_out/prof_$tag.txt | cut -c1-130
