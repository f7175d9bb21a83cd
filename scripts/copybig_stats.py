import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BVGPU_STATS"] = "1"
os.environ["BVGPU_DBG"] = os.environ.get("BVGPU_DBG", "144")  # 16: phase ticks of k_copy_big, 128: longest row per level
import torch
from scripts.ab_time import workload
from webgraph_amd.bvgraph import BVGraph
base = workload(sys.argv[1] if len(sys.argv) > 1 else "c5")
g = BVGraph.load(base)
n, m = g.numNodes(), g.numArcs()
dev = torch.device("cuda", 0)
rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
g.debug_stats(reset=True)
g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
st = g.debug_stats()
print("k_copy_big rows %d, kept blocks %d (max %d), ids %d" % (st[8], st[9], st[15], st[4]))
names = ["walk", "gather", "rank", "move", "scatter", "g:gather", "g:splits", "g:tiles"]
tot = sum(int(st[24 + i]) for i in range(8))
for i, nm in enumerate(names):
    print("  %-8s %14d ticks %5.1f%%" % (nm, st[24 + i], 100.0 * st[24 + i] / max(tot, 1)))
print("head %d; seek+3codes %d; serial walks %d rows %d codes %d ticks; coop walks %d rows %d codes %d ticks" % (st[47], st[40], st[43], st[45], st[41], st[44], st[46], st[42]))
print("longest row, ticks: level 1 %d, level 2 %d, level 3 %d; longest walk %d" % (st[0], st[1], st[2], st[3]))
