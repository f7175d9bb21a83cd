#!/usr/bin/env python3
"""GPU box: full scans of the cnr-2000 fixture (325 557 nodes, 3.2 M arcs), device outputs: time per call (small-job floor)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from webgraph_amd.bvgraph import BVGraph
g = BVGraph.load(os.path.join(ROOT, "tests", "golden", "cnr-2000"))
n, m = g.numNodes(), g.numArcs()
dev = torch.device("cuda", 0)
d_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
d_succ = torch.empty(m, dtype=torch.int32, device=dev)
for lo, hi in ((0, n), (100000, 125000)):
    ts = []
    for rep in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.decode_range_device(lo, hi, d_rowptr.data_ptr(), d_succ.data_ptr(), m)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("cnr-2000 [%d, %d): best %.3f ms, median %.3f ms" % (lo, hi, min(ts[2:]), sorted(ts[2:])[len(ts[2:]) // 2]))
g.decode_range_device(0, n, d_rowptr.data_ptr(), d_succ.data_ptr(), m)
g.close()
