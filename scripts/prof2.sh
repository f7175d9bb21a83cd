#!/bin/bash
# GPU box: rocprofv3 kernel stats of scripts/ab_time.py.  usage: scripts/prof2.sh <tag> <workload> [env assignments...]
# kernels run one after the other (BVGPU_OVERLAP=0) unless BVGPU_OVERLAP=1 is passed; writes gpurun_out/prof_<tag>.txt
tag=$1; wl=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
env BVGPU_OVERLAP=0 "$@" rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o res -- python $R/scripts/ab_time.py $wl 3 > /tmp/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $R/scripts/rocprof_summary.py $db $R/gpurun_out/prof_$tag.txt
head -${LINES_SHOWN:-14} $R/gpurun_out/prof_$tag.txt | cut -c1-150
