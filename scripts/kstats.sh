#!/bin/bash
# GPU box: per-kernel time of one workload, kernels one after the other (BVGPU_OVERLAP=0).  usage: scripts/kstats.sh <tag> <c2|c5|cnr30> [env...]
tag=$1; wl=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ks_$tag
env BVGPU_OVERLAP=0 "$@" rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o res -- python $R/scripts/ab_time.py $wl 5 > /tmp/ks_$tag.log 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/ks_$tag -name "*.db" | head -1) $R/gpurun_out/kstats_$tag.txt
head -${LINES_SHOWN:-24} $R/gpurun_out/kstats_$tag.txt | cut -c1-170
