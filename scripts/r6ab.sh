#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6ab; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for wl in c2 cnr30 c5; do
  for v in "BVGPU_PICK_ASIDE=0" "" "BVGPU_PICK_ASIDE=0" ""; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-150
  done
done | tee $O/ab.txt
for v in "BVGPU_PICK_ASIDE=0" "" "BVGPU_PICK_ASIDE=0" ""; do
  env $v timeout 900 python bench.py --steps 20 --warmup 5 --nodes 50000000 --arcs 1000000000 --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 > $O/line.json
  python - "$v" <<'PY'
import json,sys
j=json.load(open('gpurun_out/r6ab/line.json'))
print("1B %-25s %.3f ms  %.2f G edges/s" % (sys.argv[1] or "(defaults)", j["ms_per_step"], j["value"]/1e9))
PY
done | tee -a $O/ab.txt
