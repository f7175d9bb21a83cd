import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BVGPU_STATS"] = "1"
os.environ.setdefault("BVGPU_TILE", "2")
import torch
from scripts.ab_time import workload
from webgraph_amd.bvgraph import BVGraph
base = workload(sys.argv[1] if len(sys.argv) > 1 else "c2")
g = BVGraph.load(base)
n, m = g.numNodes(), g.numArcs()
dev = torch.device("cuda", 0)
rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
g.debug_stats(reset=True)
g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
st = g.debug_stats()
names = ["stage", "H", "S", "R1", "R2", "R3", "R4", "X"]
tiles = max(int(st[8]), 1)
tot = sum(int(st[i]) for i in range(8))
print("tiles with work %d, jobs/tile %.1f, segments/tile %.1f, R2 rounds/tile %.2f" % (tiles, st[10] / tiles, st[11] / tiles, st[9] / tiles))
for i, nm in enumerate(names):
    print("  %-6s %10.0f ticks/tile  %5.1f%%" % (nm, st[i] / tiles, 100.0 * st[i] / max(tot, 1)))
