#!/bin/bash
# GPU box: per-kernel time of the EFGraph scan of the C2 graph.  usage: scripts/ef_prof.sh <tag>
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
python $R/scripts/ef_time.py > /tmp/efp_warm.log 2>&1
rm -rf /tmp/efp_$tag
rocprofv3 --kernel-trace --stats -d /tmp/efp_$tag -o res -- python $R/scripts/ef_time.py > /tmp/efp_$tag.log 2>&1
grep -E "BVGraph|EFGraph" /tmp/efp_$tag.log
python $R/scripts/rocprof_summary.py $(find /tmp/efp_$tag -name "*.db" | head -1) $R/gpurun_out/ef_kstats_$tag.txt
grep -E "k_ef|k_scan" $R/gpurun_out/ef_kstats_$tag.txt | cut -c1-170
