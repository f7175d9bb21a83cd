#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6an; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for i in 1 2; do
  timeout 900 python bench.py --steps 20 --warmup 5 --nodes 50000000 --arcs 1000000000 --no-extras --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 > $O/line.json
  python - <<'PY'
import json
j=json.load(open('gpurun_out/r6an/line.json'))
print("1B %.3f ms  %.2f G edges/s" % (j["ms_per_step"], j["value"]/1e9))
PY
done | tee $O/ab.txt
python scripts/c4_time.py 20 2>/dev/null | tail -1
for wl in cnr30 c2 c5; do env AB_NO_PROFILE=1 python scripts/ab_time.py $wl 20 2>/dev/null | tail -1 | cut -c1-140; done
