#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6i; mkdir -p $O
( time timeout 3000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_all.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$? $(tail -1 $O/pytest_all.log) $(grep real $O/pytest.time)"
grep -n "Fatal\|FAILED\|Error" $O/pytest_all.log | head -5
