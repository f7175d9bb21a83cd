#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for wl in c2 c5 cnr30; do env $V python scripts/ab_time.py $wl 10 2>/dev/null | tail -1 | cut -c1-330; done | tee $O/ab.txt
