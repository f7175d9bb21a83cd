#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6av; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
timeout 600 env BVGPU_PREWALK=0 BVGPU_COPY_MID_MIN=16 python scripts/fuzz_params.py 300 69 > $O/params_69.log 2>&1; echo "fuzz_params seed 69 (PREWALK=0 COPY_MID_MIN=16) rc=$? $(tail -1 $O/params_69.log | cut -c1-200)"
timeout 600 python scripts/fuzz_params.py 300 70 > $O/params_70.log 2>&1; echo "fuzz_params seed 70 rc=$? $(tail -1 $O/params_70.log | cut -c1-200)"
timeout 600 env BVGPU_PREWALK=0 BVGPU_COPY_MID_MIN=16 python scripts/fuzz_corrupt.py 80 69 > $O/corrupt_69.log 2>&1; echo "fuzz_corrupt seed 69 (PREWALK=0 COPY_MID_MIN=16) rc=$? $(tail -1 $O/corrupt_69.log | cut -c1-200)"
for f in tests/test_gpu_scan.py tests/test_gpu_malformed.py tests/test_gpu_random.py; do
	b=$(basename $f .py)
	GUARD_MAX_BYTES=$((1<<44)) BVGPU_EXACT_ALLOC=1 timeout 1200 bash scripts/guard_run.sh python -u -m pytest $f -m gpu -x -v -p no:cacheprovider > $O/guard_$b.log 2>&1
	echo "guard $b rc=$? $(grep -c PASSED $O/guard_$b.log) passed; $(grep -a 'Memory access fault' $O/guard_$b.log | head -1)"
done
GUARD_MAX_BYTES=$((1<<44)) BVGPU_EXACT_ALLOC=1 BVGPU_PREWALK=0 BVGPU_COPY_MID_MIN=16 timeout 1200 bash scripts/guard_run.sh python -u -m pytest tests/test_gpu_scan.py -m gpu -x -v -p no:cacheprovider > $O/guard_scan_mid.log 2>&1; echo "guard test_gpu_scan PREWALK=0 COPY_MID_MIN=16 rc=$? $(grep -c PASSED $O/guard_scan_mid.log) passed; $(grep -a 'Memory access fault' $O/guard_scan_mid.log | head -1)"
