#!/usr/bin/env python3
"""GPU box: a web-graph-shaped workload at scale -- the cnr-2000 fixture tiled K times (node ids shifted per copy),
recompressed with this repository's writer (W=7, maxRefCount=3, minIntervalLength=3, as the fixture), scanned on the
GPU, checked against the CPU oracle and timed.  cnr-2000: 66 % copied / 11 % interval / 23 % residual arcs, 47 % of the
non-empty nodes at chain depth 3 -- a very different mix from the synthetic C2 workload."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import BVGraph
    from oracle import oracle as O
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    og = O.OracleGraph.load(os.path.join(ROOT, "tests", "golden", "cnr-2000"))
    rp, sc, _ = og.scan()
    n0, m0 = og.n, sc.size
    rowptr = np.concatenate([[0], (rp[1:][None, :] + (np.arange(K, dtype=np.int64) * m0)[:, None]).ravel()])
    succ = (sc[None, :].astype(np.int64) + (np.arange(K, dtype=np.int64) * n0)[:, None]).astype(np.int32).ravel()
    base = "/tmp/bvgpu_cache/cnr_x%d" % K
    os.makedirs("/tmp/bvgpu_cache", exist_ok=True)
    st = T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=3, threads=os.cpu_count())
    n, m = n0 * K, m0 * K
    print("tiled cnr-2000 x%d: %d nodes, %d arcs, %.2f bits/link, copied %.0f%% intervals %.0f%% residuals %.0f%%" % (
        K, n, m, st["written_bits"] / m, 100 * st["copied_arcs"] / m, 100 * st["intervalised_arcs"] / m, 100 * st["residual_arcs"] / m))
    g = BVGraph.load(base)
    dev = torch.device("cuda", 0)
    d_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    d_succ = torch.empty(m, dtype=torch.int32, device=dev)
    g.decode_range_device(0, n, d_rowptr.data_ptr(), d_succ.data_ptr(), m)
    ok = np.array_equal(d_rowptr.cpu().numpy(), rowptr) and np.array_equal(d_succ.cpu().numpy(), succ)
    t0 = time.perf_counter()
    for _ in range(10):
        g.decode_range_device(0, n, d_rowptr.data_ptr(), d_succ.data_ptr(), m, asynchronous=True)
    g.sync()
    dt = (time.perf_counter() - t0) / 10
    g.set_profile(True)
    g.decode_range_device(0, n, d_rowptr.data_ptr(), d_succ.data_ptr(), m)
    ph = {k: round(v, 3) for k, v in g.get_profile().items()}
    o2 = O.OracleGraph.load(base)
    t0 = time.perf_counter()
    o2.scan(0, n // 10)
    cdt = time.perf_counter() - t0
    print("bit-exact %s | scan %.3f ms = %.1f G edges/s | serial phases %s | CPU oracle 1 thread %.1f M edges/s" % (ok, dt * 1e3, m / dt / 1e9, ph, rowptr[n // 10] / cdt / 1e6))


if __name__ == "__main__":
    main()
