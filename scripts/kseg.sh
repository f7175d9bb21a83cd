#!/bin/bash
# GPU box: per-kernel times of the parse kernels of one workload, serial.  usage: scripts/kseg.sh <tag> <workload> [env...]
tag=$1; wl=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ks_$tag
env BVGPU_OVERLAP=0 "$@" rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o res -- python $R/scripts/ab_time.py $wl 5 > /tmp/ks_$tag.log 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/ks_$tag -name "*.db" | head -1) $R/gpurun_out/kstats_$tag.txt
echo "== $tag: $(grep '| scan' /tmp/ks_$tag.log | tail -1 | cut -c1-120)"
grep -E "k_seg|k_sg_|k_parse" $R/gpurun_out/kstats_$tag.txt | cut -c1-60,75-125
