#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6h; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$? $(tail -3 $O/bench.time | tr '\n' ' ')"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6h/bench.json"))
print("value %.2f G edges/s  ms %.3f  frac %.4f scan_frac %.4f bound %s" % (d["value"]/1e9, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["scan_frac"], d["roofline"]["bound"]))
print("issue", {k:v for k,v in (d["roofline"].get("issue") or {}).items() if k not in ("note","per_kernel_valu_winst")})
for k,v in d["roofline"]["kernels"].items(): print("  ", k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!="note"})
for k,v in d.get("extras",{}).items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms","ms_per_step","edges_per_s","lists_per_s","scan_frac","frac","error","skipped","parity","child_wall_s","scan_checksum_ms","scan_stats_ms","equal_range_ms","hyperball_step_ms")})
PY
( time timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_all.log 2>&1 ) 2> $O/pytest.time; echo "pytest rc=$? $(tail -1 $O/pytest_all.log) $(grep real $O/pytest.time)"
