#!/usr/bin/env python3
"""GPU box: how much of a scan's time is latency that a second, independent scan could fill?  The graph's node range is cut into P parts
(equal compressed size), each decoded by a handle of its own (own streams), all enqueued at once; compared with one scan of the whole.
usage: concurrent_parts.py [c2|c5|cnr30] [parts ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    import torch
    import numpy as np
    from ab_time import workload
    from webgraph_amd.bvgraph import BVGraph
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    parts_list = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
    base = workload(name)
    g0 = BVGraph.load(base)
    n, m = g0.numNodes(), g0.numArcs()
    dev = torch.device("cuda", 0)
    for P in parts_list:
        bounds = g0.shard_bounds(P)
        hs = [g0.copy() for _ in range(P)]
        bufs = []
        for p in range(P):
            lo, hi = int(bounds[p]), int(bounds[p + 1])
            rp = torch.empty(hi - lo + 1, dtype=torch.int64, device=dev)
            sc = torch.empty(int(m * 1.0 / P * 1.5) + (1 << 20), dtype=torch.int32, device=dev)
            bufs.append((lo, hi, rp, sc))
        def run(async_all=True):
            for h, (lo, hi, rp, sc) in zip(hs, bufs):
                h.decode_range_device(lo, hi, rp.data_ptr(), sc.data_ptr(), sc.numel(), asynchronous=True)
                if not async_all:
                    h.sync()
            tot = 0
            for h in hs:
                tot += h.sync()
            return tot
        for _ in range(3):
            tot = run()
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        dt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            run(False)
        ds = (time.perf_counter() - t0) / reps
        print("%s parts %d: arcs %d | all enqueued at once %.3f ms = %.1f G edges/s | one after the other %.3f ms" % (name, P, tot, dt * 1e3, tot / dt / 1e9, ds * 1e3))
        for h in hs:
            h.close()
    g0.close()


if __name__ == "__main__":
    main()
