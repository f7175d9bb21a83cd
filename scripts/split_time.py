#!/usr/bin/env python3
"""GPU box experiment: a full scan as K independent sub-range jobs on K flyweight handles (own streams and scratch each),
enqueued back to back and running side by side, against the single job.  usage: split_time.py [c2|c5|cnr30] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scripts.ab_time import workload
from webgraph_amd.bvgraph import BVGraph

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = BVGraph.load(workload(name))
n, m = g.numNodes(), g.numArcs()
dev = torch.device("cuda", 0)
rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
succ = torch.empty(m, dtype=torch.int32, device=dev)
g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m)
ref = succ.clone()
b = g.shard_bounds(K)
rp = rowptr.cpu().numpy()
hs = [g.copy() for _ in range(K)]
rps = [torch.empty(int(b[k + 1] - b[k]) + 1, dtype=torch.int64, device=dev) for k in range(K)]
succ.zero_()

def run():
    for k in range(K):
        lo, hi = int(b[k]), int(b[k + 1])
        hs[k].decode_range_device(lo, hi, rps[k].data_ptr(), succ.data_ptr() + 4 * int(rp[lo]), int(rp[hi] - rp[lo]), asynchronous=True)
    for k in range(K):
        hs[k].sync()

for _ in range(3):
    run()
torch.cuda.synchronize()
print("equal:", bool(torch.equal(succ, ref)))
t0 = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
for _ in range(10):
    g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m, asynchronous=True)
g.sync()
d1 = (time.perf_counter() - t0) / 10
print("%s: single job %.3f ms | %d jobs side by side %.3f ms" % (name, d1 * 1e3, K, dt * 1e3))
