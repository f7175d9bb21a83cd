#!/bin/bash
# every GPU test file under the guard allocator, one process per file (a fault ends only that file's run)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/guard
for f in tests/test_gpu_*.py; do
	b=$(basename $f .py)
	timeout ${GUARD_FILE_TIMEOUT:-900} bash scripts/guard_run.sh python -u -m pytest $f -m gpu -x -v -p no:cacheprovider > gpurun_out/guard/$b.log 2>&1
	echo "$b rc=$? $(grep -c PASSED gpurun_out/guard/$b.log) passed; $(grep -a 'Memory access fault' gpurun_out/guard/$b.log | head -1)"
done
