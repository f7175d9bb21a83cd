#!/usr/bin/env python3
"""GPU box: a .graph file of more than 2 GiB (BVGraph.java:1562-1568 loads such a file in several byte arrays): rows without locality -- 16 successors each,
gaps of ~10^6 -- so that every arc costs ~28 bits.  Stored by the CPU writer, loaded (the file goes to HBM in pieces), scanned, hashCode and sampled rows against the oracle,
a batch of random ids near the end of the file against the scan.  usage: big_file.py [nodes] [successors per node]"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 42_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    base = "/tmp/bvgpu_cache/bigfile_%d_%d" % (n, d)
    os.makedirs("/tmp/bvgpu_cache", exist_ok=True)
    if not os.path.exists(base + ".properties"):
        t0 = time.time()
        rng = np.random.default_rng(2024)
        step = n // (d + 1)
        succ = np.empty((n, d), dtype=np.int32)
        CH = 2_000_000
        for a in range(0, n, CH):  # strictly increasing rows: a start below `step`, then d - 1 gaps in [1, step]
            b = min(a + CH, n)
            g = rng.integers(1, step + 1, size=(b - a, d), dtype=np.int64)
            g[:, 0] = rng.integers(0, step, size=b - a)
            succ[a:b] = np.cumsum(g, axis=1).astype(np.int32)
        assert int(succ.max()) < n
        rowptr = np.arange(n + 1, dtype=np.int64) * d
        T.store(base, rowptr, succ.reshape(-1), window=7, max_ref_count=3, min_interval=4, zeta_k=3, threads=os.cpu_count())
        print("generated and stored in %.0f s" % (time.time() - t0), flush=True)
        del succ, rowptr
    size = os.path.getsize(base + ".graph")
    print(".graph %d bytes = %.3f GiB (%s 2 GiB)" % (size, size / 2**30, "MORE than" if size > 2**31 else "NOT more than"), flush=True)
    t0 = time.time()
    g = BVGraph.load(base)
    print("load %.2f s" % (time.time() - t0), flush=True)
    n, m = g.numNodes(), g.numArcs()
    rowptr = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    succ = torch.empty(m, dtype=torch.int32, device="cuda")
    arcs = g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m)
    h = g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1)
    hs = g.hashCode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m, asynchronous=True)
    g.sync()
    dt = (time.perf_counter() - t0) / 3
    og = O.OracleGraph.load(base)
    want = og.hashcode_mt()
    ok = arcs == m and h == want and hs == want
    # rows at the far end of the file, one by one from the oracle, and the same ids as a batch
    ids = np.concatenate([np.arange(n - 2000, n), np.random.default_rng(1).integers(n - n // 50, n, 3000)]).astype(np.int32)
    rp, sc = g.successors_batch(ids)
    srp = rowptr.cpu().numpy()
    for k, x in enumerate(ids[:2500]):
        row = og.successors(int(x))
        a = int(srp[x])
        ok = ok and np.array_equal(sc[rp[k]:rp[k + 1]], row) and np.array_equal(succ[a:a + row.size].cpu().numpy(), row)
    print("big file: n %d m %d | scan %.2f ms = %.1f G edges/s | hashCode scan/fold %s, far rows (scan and batch) vs oracle: %s" % (n, m, dt * 1e3, m / dt / 1e9, "ok" if h == want and hs == want else "MISMATCH", "ok" if ok else "MISMATCH"))
    og.close()
    g.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
