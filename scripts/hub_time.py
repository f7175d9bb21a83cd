#!/usr/bin/env python3
"""GPU box: a power-law graph with a few hub rows of millions of successors each (social-graph shape): the longest records are where a
record-per-group decoder has its tail.  usage: hub_time.py [hub sizes ...]   (BVGPU_* knobs from the environment)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def build(sizes, n=10_000_000, m=100_000_000, cache="/tmp/bvgpu_cache"):
    from webgraph_amd import tools as T
    os.makedirs(cache, exist_ok=True)
    base = os.path.join(cache, "hubs_%d_%s" % (n, "_".join(str(s) for s in sizes)))
    if not os.path.exists(base + ".graph"):
        rowptr, succ = T.generate(n, m, seed=0x5EEDB5E70009, p_copy=0.5)
        rng = np.random.Generator(np.random.PCG64(17))
        rows = {}
        for i, d in enumerate(sizes):
            rows[1000 + 7919 * i] = np.unique(rng.integers(0, n, size=int(d * 1.3)))[:d].astype(np.int32)
        deg = np.diff(rowptr)
        for x, r in rows.items():
            deg[x] = r.size
        rp = np.zeros(n + 1, dtype=np.int64)
        rp[1:] = np.cumsum(deg)
        out = np.empty(rp[-1], dtype=np.int32)
        prev = 0
        for x in sorted(rows):  # copy the stretches between the hubs
            out[rp[prev]:rp[x]] = succ[rowptr[prev]:rowptr[x]]
            out[rp[x]:rp[x + 1]] = rows[x]
            prev = x + 1
        out[rp[prev]:] = succ[rowptr[prev]:]
        T.store(base, rp, out, window=7, max_ref_count=3, min_interval=4, zeta_k=3, threads=os.cpu_count())
    return base


def main():
    import torch
    from webgraph_amd.bvgraph import BVGraph
    from oracle import oracle as O
    sizes = [int(a) for a in sys.argv[1:]] or [8_000_000, 4_000_000, 2_000_000]
    base = build(sizes)
    g = BVGraph.load(base)
    n, m = g.numNodes(), g.numArcs()
    dev = torch.device("cuda", 0)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    arcs = g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
    h = g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1)
    want = O.OracleGraph.load(base).hashcode_mt() if os.environ.get("HUB_CHECK", "1") == "1" else h
    for _ in range(3):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel(), asynchronous=True)
    g.sync()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel(), asynchronous=True)
    g.sync()
    dt = (time.perf_counter() - t0) / reps
    knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("BVGPU_") and k not in ("BVGPU_CACHE", "BVGPU_LIB"))
    print("hubs %s | %-40s arcs %d hash %s | scan %.3f ms = %.1f G edges/s" % (sizes, knobs or "(defaults)", arcs, "ok" if h == want else "MISMATCH %d vs %d" % (h, want), dt * 1e3, arcs / dt / 1e9))
    g.close()


if __name__ == "__main__":
    main()
