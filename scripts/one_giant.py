#!/usr/bin/env python3
"""GPU box: decode single long records of the C2 graph on their own (latency of the cooperative decoders)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from webgraph_amd.bvgraph import BVGraph
n, m = 10_000_000, 200_000_000
base, meta = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
g = BVGraph.load(base)
dev = torch.device("cuda", 0)
d_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
d_succ = torch.empty(m, dtype=torch.int32, device=dev)
outd = g.outdegrees(0, n)
off = np.fromfile(base + ".offsets", dtype=np.uint8)  # only for sizes: decode on the host
from webgraph_amd.bvgraph import decode_offsets_host
offs = decode_offsets_host(off, n)
order = np.argsort(outd)[::-1]
picks = [int(order[0]), int(order[50]), int(order[300]), int(order[2000]), int(order[20000])]
g.set_profile(True)
for x in picks:
    for rep in range(3):
        g.decode_range_device(x, x + 1, d_rowptr.data_ptr(), d_succ.data_ptr(), m)
    ph = g.get_profile()
    if os.environ.get("BVGPU_STATS"):
        st = g.debug_stats(reset=True)
        f = lambda a: [int(v) // 3000 for v in a]
        print("   kilo-ticks per call: wave A,I,R,X = %s group A,I,R,X = %s | residual tiles: stage,spec,ivstage,values,last,sync,X,top = %s | spec: %s" % (f(st[16:20]), f(st[20:24]), f(st[24:32]), f(st[8:16])))
    print("node %d: outdegree %d, %d bits: phases(ms) %s" % (x, outd[x], offs[x + 1] - offs[x], {k: round(v, 3) for k, v in ph.items() if v > 0.02}))
g.close()
