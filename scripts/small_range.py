#!/usr/bin/env python3
"""GPU box: a few scans of a 100 000-node range in the middle of the C2 graph (for kernel timelines of small calls)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from webgraph_amd.bvgraph import BVGraph
n, m = 10_000_000, 200_000_000
base, meta = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
g = BVGraph.load(base)
dev = torch.device("cuda", 0)
d_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
d_succ = torch.empty(m, dtype=torch.int32, device=dev)
lo = int(os.environ.get("LO", "5000000")); cnt = int(os.environ.get("CNT", "100000"))
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.decode_range_device(lo, lo + cnt, d_rowptr.data_ptr(), d_succ.data_ptr(), m)
    print("call %d: %.3f ms" % (rep, (time.perf_counter() - t0) * 1e3))
g.close()
g = BVGraph.load(base)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.decode_range_device(lo, lo + cnt, d_rowptr.data_ptr(), d_succ.data_ptr(), m, asynchronous=True)
    t1 = time.perf_counter()
    g.sync()
    t2 = time.perf_counter()
    print("async call %d: host returns after %.3f ms, done after %.3f ms" % (rep, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
g.close()
