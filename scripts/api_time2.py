#!/usr/bin/env python3
"""GPU box: more entry points beside the scan on a cached workload: the second format (store / recompress / consumers on an EFGraph handle), a full breadth-first visit,
a node iterator drained through the mirror, arc-label lists.  usage: api_time2.py [c2|cnr30]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.ab_time import workload


def main():
    import numpy as np
    import torch
    from webgraph_amd.bvgraph import BVGraph, EFGraph
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    base = workload(name)
    g = BVGraph.load(base)
    n, m = g.numNodes(), g.numArcs()
    tmp = tempfile.mkdtemp(prefix="api2")

    def timed(f, reps=2):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    res = {}
    res["store_ef (files)"] = timed(lambda: g.store_ef(os.path.join(tmp, "ef")), 1)
    res["store (BV files, recompress)"] = timed(lambda: g.store(os.path.join(tmp, "bv")), 1)
    h = EFGraph.load(os.path.join(tmp, "ef"))
    res["EF scan_stats"] = timed(lambda: h.scan_stats(0, n))
    res["EF scan_checksum"] = timed(lambda: h.scan_checksum(0, n, -1))
    res["EF equal_range vs BV"] = timed(lambda: g.equal_range(h, 0, n))
    t0 = time.perf_counter()
    q, cut, _ = g.bfs(0)
    res["bfs(0): %d nodes, %d rounds" % (len(q), len(cut) - 1)] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    it = g.nodeIterator(0)
    tot = 0
    for _ in range(min(n, 200000)):
        it.nextInt()
        tot += it.outdegree()
    res["nodeIterator, 200 k nodes"] = (time.perf_counter() - t0) * 1e3
    print(name, " | ".join("%s %.2f ms" % kv for kv in res.items()))
    h.close()
    g.close()


if __name__ == "__main__":
    main()
