#!/usr/bin/env python3
"""GPU box: time of bvg_csr_hashcode (ImmutableGraph.hashCode of a decoded CSR) on the C2 graph."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from webgraph_amd.bvgraph import BVGraph
from oracle import oracle as O
n, m = 10_000_000, 200_000_000
base, meta = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
g = BVGraph.load(base)
dev = torch.device("cuda", 0)
d_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
d_succ = torch.empty(m, dtype=torch.int32, device=dev)
g.decode_range_device(0, n, d_rowptr.data_ptr(), d_succ.data_ptr(), m)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = g.csr_hashcode(0, n, d_rowptr.data_ptr(), d_succ.data_ptr(), -1)
    dt = time.perf_counter() - t0
    print("hashCode of %d nodes / %d arcs: %d in %.2f ms" % (n, m, h, dt * 1e3))
og = O.OracleGraph.load(base)
print("oracle:", og.hashcode())
g.close()
