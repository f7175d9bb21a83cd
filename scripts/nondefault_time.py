#!/usr/bin/env python3
"""GPU box: the C2 graph stored with non-default parameters (zeta_k != 3 / delta residuals): scan time and parity."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import BVGraph, flags_from_string
    n, m = 10_000_000, 200_000_000
    os.makedirs("/tmp/bvgpu_cache", exist_ok=True)
    rowptr, succ = T.generate(n, m, seed=bench.SEED, p_copy=0.5, threads=os.cpu_count())
    dev = torch.device("cuda", 0)
    d_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    d_succ = torch.empty(m, dtype=torch.int32, device=dev)
    for name, kw in (("zeta2", dict(zeta_k=2)), ("zeta4", dict(zeta_k=4)), ("delta residuals", dict(flags=flags_from_string("RESIDUALS_DELTA")))):
        base = "/tmp/bvgpu_cache/nd_" + name.replace(" ", "_")
        T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=4, threads=os.cpu_count(), **kw)
        g = BVGraph.load(base)
        g.decode_range_device(0, n, d_rowptr.data_ptr(), d_succ.data_ptr(), m)
        ok = np.array_equal(d_rowptr.cpu().numpy(), rowptr) and np.array_equal(d_succ.cpu().numpy(), succ)
        t0 = time.perf_counter()
        for _ in range(3):
            g.decode_range_device(0, n, d_rowptr.data_ptr(), d_succ.data_ptr(), m, asynchronous=True)
        g.sync()
        dt = (time.perf_counter() - t0) / 3
        print("%s: scan %.2f ms = %.1f G edges/s, bit-exact %s" % (name, dt * 1e3, m / dt / 1e9, ok))
        g.close()


if __name__ == "__main__":
    main()
