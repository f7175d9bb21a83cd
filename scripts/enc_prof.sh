#!/bin/bash
# GPU box: per-kernel time of the device compressor on a graph of the C2 recipe.  usage: scripts/enc_prof.sh <tag> [nodes] [arcs]
tag=$1; n=${2:-10000000}; m=${3:-200000000}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
python $R/scripts/enc_time.py $n $m 1 > /tmp/encp_warm.log 2>&1   # generates and caches the graph outside the profile
rm -rf /tmp/encp_$tag
rocprofv3 --kernel-trace --stats -d /tmp/encp_$tag -o res -- python $R/scripts/enc_time.py $n $m 2 > /tmp/encp_$tag.log 2>&1
tail -4 /tmp/encp_$tag.log
python $R/scripts/rocprof_summary.py $(find /tmp/encp_$tag -name "*.db" | head -1) $R/gpurun_out/enc_kstats_$tag.txt
grep -E "k_enc|k_scan|Name|name" $R/gpurun_out/enc_kstats_$tag.txt | head -30 | cut -c1-200
