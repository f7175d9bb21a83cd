#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6at; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
timeout 600 python scripts/fuzz_params.py 300 67 > $O/params_67.log 2>&1; echo "fuzz_params seed 67 rc=$? $(tail -1 $O/params_67.log | cut -c1-200)"
timeout 600 env BVGPU_TILE=1 python scripts/fuzz_params.py 300 68 > $O/params_68.log 2>&1; echo "fuzz_params seed 68 (BVGPU_TILE=1) rc=$? $(tail -1 $O/params_68.log | cut -c1-200)"
timeout 600 env BVGPU_TILE=1 python scripts/fuzz_corrupt.py 80 67 > $O/corrupt_67.log 2>&1; echo "fuzz_corrupt seed 67 (BVGPU_TILE=1) rc=$? $(tail -1 $O/corrupt_67.log | cut -c1-200)"
for f in tests/test_gpu_scan.py tests/test_gpu_malformed.py tests/test_gpu_random.py tests/test_gpu_boundary.py; do
	b=$(basename $f .py)
	GUARD_MAX_BYTES=$((1<<44)) BVGPU_EXACT_ALLOC=1 timeout 1200 bash scripts/guard_run.sh python -u -m pytest $f -m gpu -x -v -p no:cacheprovider > $O/guard_$b.log 2>&1
	echo "guard $b rc=$? $(grep -c PASSED $O/guard_$b.log) passed; $(grep -a 'Memory access fault' $O/guard_$b.log | head -1)"
done
GUARD_MAX_BYTES=$((1<<44)) BVGPU_EXACT_ALLOC=1 BVGPU_TILE=1 timeout 1200 bash scripts/guard_run.sh python -u -m pytest tests/test_gpu_scan.py -m gpu -x -v -p no:cacheprovider > $O/guard_scan_tile1.log 2>&1; echo "guard test_gpu_scan BVGPU_TILE=1 rc=$? $(grep -c PASSED $O/guard_scan_tile1.log) passed; $(grep -a 'Memory access fault' $O/guard_scan_tile1.log | head -1)"
