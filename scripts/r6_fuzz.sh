#!/bin/bash
# GPU box, round 6: the fuzzers and the guard allocator behind the new lane loop, the copy tables, the keys from k_headers and the lighter giants
cd "$(dirname "$0")/.."
O=gpurun_out/r6_fuzz; mkdir -p $O
SEEDS="${FUZZ_SEEDS:-61 62 63}"
for sd in $SEEDS; do
	timeout 600 python scripts/fuzz_params.py 300 $sd > $O/params_$sd.log 2>&1; echo "fuzz_params seed $sd rc=$? $(tail -1 $O/params_$sd.log | cut -c1-200)"
done
for sd in $SEEDS; do
	timeout 600 python scripts/fuzz_corrupt.py 80 $sd > $O/corrupt_$sd.log 2>&1; echo "fuzz_corrupt seed $sd rc=$? $(tail -1 $O/corrupt_$sd.log | cut -c1-200)"
done
timeout 600 python scripts/fuzz_store.py 100 61 > $O/store.log 2>&1; echo "fuzz_store rc=$? $(tail -1 $O/store.log | cut -c1-200)"
timeout 600 python scripts/fuzz_consumers.py 80 61 > $O/consumers.log 2>&1; echo "fuzz_consumers rc=$? $(tail -1 $O/consumers.log | cut -c1-200)"
timeout 600 python scripts/cross_check.py > $O/cross.log 2>&1; echo "cross_check rc=$? $(tail -1 $O/cross.log | cut -c1-200)"
GUARD_MAX_BYTES=$((1<<44)) GUARD_FILE_TIMEOUT=1200 BVGPU_EXACT_ALLOC=1 bash scripts/guard_suite.sh 2>&1 | tee $O/guard_suite.txt
