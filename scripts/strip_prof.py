"""Per-phase tick counters of the strip kernel (BVGPU_STATS=1): average microseconds per strip and phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BVGPU_STATS"] = "1"
os.environ.setdefault("BVGPU_OVERLAP", "0")
import numpy as np, torch, bench
import __graft_entry__ as ge
ge.build()
from webgraph_amd.bvgraph import BVGraph
which = sys.argv[1] if len(sys.argv) > 1 else "C2"
if which in bench.WORKLOADS:
    wl = bench.WORKLOADS[which]
    base = bench.prepare_graph(wl["n"], wl["m"], wl["seed"], wl["p_copy"], "/tmp/bvgpu_cache", os.cpu_count() or 1, p_same=wl["p_same"], p_keep=wl["p_keep"])[0]
else:
    base = which
g = BVGraph.load(base)
n = g.numNodes()
dev = torch.device("cuda", 0)
rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
m = g.decode_range_device(0, n, rowptr.data_ptr(), None, 0)
succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
g.debug_stats(True)
R = 3
for _ in range(R):
    g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
st = g.debug_stats(True).astype(np.float64)
strips = st[32 + 15] / R
names = ["loads+staging", "S structure", "segments + A anchors", "B chain", "R residuals", "X intervals"]
print("strips %d  nodes/strip %.1f  intervals/strip %.1f  segs/strip %.1f  longsegs/strip %.1f" % (strips, st[32 + 14] / R / strips, st[32 + 11] / R / strips, st[32 + 13] / R / strips, st[32 + 12] / R / strips))
tot = 0
for k, nm in enumerate(names):
    us = st[32 + k] / R / strips / 100.0  # 100 MHz clock
    tot += us
    print("  %-20s %7.2f us" % (nm, us))
print("  %-20s %7.2f us per strip (1 in 64 sampled); x strips / (256 CUs x 20 waves) = %.3f ms" % ("total", tot, tot * strips * 64 / 5120 / 1e3))
g.set_profile(True)
g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
print("phases", {k: round(v, 3) for k, v in g.get_profile().items()})
