#!/bin/bash
# round 6: quarter-octave work bins (the lanes of a wave differ by < 1.19x instead of < 1.41x)
cd "$(dirname "$0")/.."
O=gpurun_out/r6ah; mkdir -p $O
V=$PWD/webgraph_amd/variants
timeout 900 env BVGPU_LIB=$V/libbvgpu_q4.so python -m pytest tests/test_gpu_scan.py tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.txt
for wl in c2 c5 cnr30; do
  for v in "" "BVGPU_LIB=$V/libbvgpu_q4.so" "" "BVGPU_LIB=$V/libbvgpu_q4.so"; do
    env AB_NO_PROFILE=1 $v timeout 600 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | sed "s#$V/##" | cut -c1-150
  done
done | tee $O/ab.txt
for v in "" "BVGPU_LIB=$V/libbvgpu_q4.so" "" "BVGPU_LIB=$V/libbvgpu_q4.so"; do
  env $v timeout 900 python bench.py --steps 20 --warmup 5 --nodes 50000000 --arcs 1000000000 --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 > $O/line.json
  python - "$v" <<'PY'
import json,sys
j=json.load(open('gpurun_out/r6ah/line.json'))
print("1B %-40s %.3f ms  %.2f G edges/s" % (sys.argv[1].split('/')[-1] or "(defaults)", j["ms_per_step"], j["value"]/1e9))
PY
done | tee -a $O/ab.txt
for v in "" "BVGPU_LIB=$V/libbvgpu_q4.so"; do env $v timeout 600 python scripts/ab_time.py c2 10 2>/dev/null | tail -1 | cut -c100-400; done
