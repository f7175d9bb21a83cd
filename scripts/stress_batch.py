#!/usr/bin/env python3
"""GPU box: random access to the LONGEST rows of the C2 graph (worst case of the batch path), timed and checked."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import bench
    from webgraph_amd.bvgraph import BVGraph
    from oracle import oracle as O
    base, _ = bench.prepare_graph(10_000_000, 200_000_000, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    g = BVGraph.load(base)
    d = g.outdegrees()
    top = np.argsort(d)[::-1][:2000].astype(np.int32)
    og = O.OracleGraph.load(base)
    for k in (10, 200, 2000):
        q = np.ascontiguousarray(top[:k])
        g.successors_batch(q)
        t0 = time.perf_counter()
        rp, sc = g.successors_batch(q)
        dt = time.perf_counter() - t0
        orp, osc = og.successors_batch(q)
        print("top-%d rows: %d arcs in %.2f ms (%.2f G edges/s), bit-exact %s" % (k, rp[-1], dt * 1e3, rp[-1] / dt / 1e9, np.array_equal(rp, orp) and np.array_equal(sc, osc)))


if __name__ == "__main__":
    main()
