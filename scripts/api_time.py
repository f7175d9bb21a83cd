#!/usr/bin/env python3
"""GPU box: a few entry points beside the scan, timed on a cached workload (looking for calls that take many times a scan).  usage: api_time.py [c2|cnr30]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.ab_time import workload


def main():
    import numpy as np
    import torch
    from webgraph_amd.bvgraph import BVGraph
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    g = BVGraph.load(workload(name))
    n, m = g.numNodes(), g.numArcs()
    dev = torch.device("cuda", 0)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)

    def timed(f, reps=3):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    res = {}
    res["scan"] = timed(lambda: g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel()))
    res["csr_hashcode"] = timed(lambda: g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1))
    res["outdegrees(host)"] = timed(lambda: g.outdegrees(0, n))
    res["count only"] = timed(lambda: g.decode_range_device(0, n, rowptr.data_ptr(), 0, 0))
    g2 = g.copy()
    res["equals(copy)"] = timed(lambda: g.equals(g2), 1)
    q = np.random.default_rng(3).integers(0, n, size=100_000).astype(np.int32)
    res["batch 100k ids (host)"] = timed(lambda: g.successors_batch(q))
    res["hashCode()"] = timed(lambda: g.hashCode())
    print(name, " | ".join("%s %.2f ms" % kv for kv in res.items()))
    g.close()


if __name__ == "__main__":
    main()
