cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 200 python scripts/host_path_time.py 2>&1 | tail -2
timeout 300 python scripts/ab_time.py c5 5 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python - <<'PY' 2>&1 | tail -3
import sys, time
sys.path.insert(0, "/root/repo")
from scripts.ab_time import workload
from webgraph_amd.bvgraph import BVGraph
from oracle import oracle as O
base = workload("c5")
g = BVGraph.load(base)
for i in range(3):
    t0 = time.perf_counter(); h = g.scan_checksum(); dt = time.perf_counter() - t0
    print("c5 scan_checksum %.2f ms" % (dt * 1e3), h)
og = O.OracleGraph.load(base)
print("oracle hash", og.hashcode_mt(threads=64) if hasattr(og, "hashcode_mt") else None)
PY
