#!/usr/bin/env python3
"""GPU box: randomized cross-check at C2 size.  Random sub-ranges of the scan and random batches (sparse and dense
strategy, with repeats) against the full decode of the same graph (itself checked against the oracle by bench.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import bench
    from webgraph_amd.bvgraph import BVGraph
    n, m = 10_000_000, 200_000_000
    base, _ = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    g = BVGraph.load(base)
    rowptr, succ = g.decode_range()
    assert rowptr[-1] == m
    rng = np.random.default_rng(2026)
    bad = 0
    for t in range(24):
        size = int(10 ** rng.uniform(1, 6.8))
        lo = 0 if t % 4 == 0 else int(rng.integers(0, n - size))  # prefixes of the graph take the path without a halo
        if t % 8 == 0:
            size = int(rng.integers(1 << 20, n))  # ... and large ones the fused key path of k_headers
        rp, sc = g.decode_range(lo, lo + size)
        ok = np.array_equal(rp, rowptr[lo:lo + size + 1] - rowptr[lo]) and np.array_equal(sc, succ[rowptr[lo]:rowptr[lo + size]])
        bad += not ok
        print("range [%d, %d): %d arcs %s" % (lo, lo + size, rp[-1], "ok" if ok else "MISMATCH"))
    for t in range(16):
        q = int(10 ** rng.uniform(0, 6.5))
        span = int(10 ** rng.uniform(np.log10(max(q, 10)), 7))
        lo = int(rng.integers(0, n - span + 1))
        nodes = rng.integers(lo, lo + span, size=q).astype(np.int32)
        rp, sc = g.successors_batch(nodes)
        deg = (rowptr[nodes.astype(np.int64) + 1] - rowptr[nodes]).astype(np.int64)
        erp = np.concatenate([[0], np.cumsum(deg)])
        ok = np.array_equal(rp, erp)
        if ok:
            idx = np.repeat(rowptr[nodes] - erp[:-1], deg) + np.arange(erp[-1])
            ok = np.array_equal(sc, succ[idx])
        bad += not ok
        print("batch q=%d in a span of %d nodes (%s): %d arcs %s" % (q, span, "dense" if q * 32 >= n else "slots", rp[-1], "ok" if ok else "MISMATCH"))
    g.close()
    print("cross-check:", "all ok" if not bad else "%d MISMATCHES" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(main())
