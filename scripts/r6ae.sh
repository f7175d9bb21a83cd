#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6ae; mkdir -p $O
V=$PWD/webgraph_amd/variants
export TMPDIR=/tmp; R=$PWD; cd /tmp
for dbg in 0 131072; do
rm -rf /tmp/prof_tl; env BVGPU_LIB=$V/libbvgpu_timing.so BVGPU_DBG=$dbg BVGPU_LEVEL_BINS=0 rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/c4_time.py 3 > /tmp/log_$dbg.txt 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_c4_dbg$dbg.txt --back 3 > /dev/null; echo "== dbg $dbg"; grep -E "k_copy_[lmb]" $R/$O/timeline_c4_dbg$dbg.txt | cut -c1-100; grep "^c4" /tmp/log_$dbg.txt | cut -c1-200
done
