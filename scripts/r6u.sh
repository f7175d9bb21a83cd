#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py tests/test_gpu_malformed.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.txt
V=$PWD/webgraph_amd/variants
export TMPDIR=/tmp; R=$PWD; cd /tmp
for dbg in 0 131072; do for wl in cnr30 c2; do
rm -rf /tmp/prof_tl; env BVGPU_LIB=$V/libbvgpu_timing.so BVGPU_DBG=$dbg rocprofv3 --kernel-trace -d /tmp/prof_tl -o res -- python $R/scripts/ab_time.py $wl 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/prof_tl -name "*.db" | head -1) $R/$O/timeline_${wl}_$dbg.txt --back 3 > /dev/null; echo "== $wl dbg $dbg"; grep -E "k_copy_[lmb]" $R/$O/timeline_${wl}_$dbg.txt | cut -c1-100
done; done
