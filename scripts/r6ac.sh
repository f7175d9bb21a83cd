#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6ac; mkdir -p $O
for v in "BVGPU_LEVEL_BINS=0 BVGPU_COPY_LOOP=0" "BVGPU_LEVEL_BINS=0 BVGPU_LIST_REFS=0" "BVGPU_LEVEL_BINS=0 BVGPU_PICK_ASIDE=0" "BVGPU_COPY_LOOP=0 BVGPU_LIST_REFS=0" "BVGPU_LIST_REFS=0 BVGPU_PICK_ASIDE=0" "BVGPU_LEVEL_BINS=0 BVGPU_COPY_LOOP=0 BVGPU_LIST_REFS=0" "BVGPU_LEVEL_BINS=0 BVGPU_COPY_LOOP=0 BVGPU_PICK_ASIDE=0"; do
  env $v timeout 600 python bench.py --mode random 2>/dev/null | tail -1 > $O/line.json
  python - "$v" <<'PY'
import json,sys
j=json.load(open('gpurun_out/r6ac/line.json'))
print("C4 %-90s %.3f ms  %.2f G lists/s" % (sys.argv[1] or "(defaults)", j["ms_per_step"], j["value"]/1e9))
PY
done | tee $O/ab.txt
