#!/bin/bash
# round 6: the 1 B-arc scan (north star's size) under the copy pass's knobs
cd "$(dirname "$0")/.."
O=gpurun_out/r6_1b; mkdir -p $O
for v in "BVGPU_LIB=$PWD/webgraph_amd/variants/libbvgpu_head.so" "" "BVGPU_LIB=$PWD/webgraph_amd/variants/libbvgpu_head.so" ""; do
  env $v timeout 900 python bench.py --steps 20 --warmup 5 --nodes 50000000 --arcs 1000000000 --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 > $O/line.json
  python - "$v" <<'PY'
import json,sys
j=json.load(open('gpurun_out/r6_1b/line.json'))
print("%-20s %.3f ms  %.2f G edges/s  scan_frac %.4f" % (("HEAD" if "head" in sys.argv[1] else "new"), j["ms_per_step"], j["value"]/1e9, j["roofline"]["scan_frac"]))
PY
done | tee $O/ab.txt
