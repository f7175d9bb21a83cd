#!/usr/bin/env python3
"""GPU box: the shape of the reference's own slow test (slow/it/unimi/dsi/webgraph/BVGraphSlowTest.java:30-101, BigGraph(Integer.MAX_VALUE, 1 << 30, 4)): n = 2^31 - 1 nodes, nodes
0 and 1 with 2^30 successors each (i * step), every other node x with the successors {x - 2, x - 1} -- 6.4 G arcs, more than 2^32.  (step = 2 here: with the test's 4 the ids
i * step overflow a Java int from i = 2^29 on.)  Stored by the CPU writer, loaded, scanned in ONE call into 26 GB of successors; hashCode by scan and by fold, the two long rows
and rows at both ends against the CPU oracle.  usage: slow_test_shape.py [nodes [log2 of the long rows' outdegree]]"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2**31 - 1
    lg = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    D = 1 << lg
    step = max(1, min(4, (n - 1) // D))
    base = "/tmp/bvgpu_cache/slowshape_%d_%d" % (n, lg)
    os.makedirs("/tmp/bvgpu_cache", exist_ok=True)
    if not os.path.exists(base + ".properties"):
        t0 = time.time()
        rowptr = np.empty(n + 1, dtype=np.int64)
        rowptr[0], rowptr[1] = 0, D
        rowptr[2:] = 2 * D + 2 * np.arange(0, n - 1, dtype=np.int64)
        succ = np.empty(int(rowptr[-1]), dtype=np.int32)
        succ[:D] = (np.arange(D, dtype=np.int64) * step).astype(np.int32)
        succ[D:2 * D] = succ[:D]
        succ[2 * D::2] = np.arange(0, n - 2, dtype=np.int32)
        succ[2 * D + 1::2] = np.arange(1, n - 1, dtype=np.int32)
        print("generated in %.0f s: %d arcs" % (time.time() - t0, succ.size), flush=True)
        t0 = time.time()
        T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=4, zeta_k=3, threads=os.cpu_count())
        print("stored in %.0f s" % (time.time() - t0), flush=True)
        del succ, rowptr
    print(".graph %.2f GB .offsets %.2f GB" % (os.path.getsize(base + ".graph") / 1e9, os.path.getsize(base + ".offsets") / 1e9), flush=True)
    t0 = time.time()
    g = BVGraph.load(base)
    print("load %.2f s" % (time.time() - t0), flush=True)
    n, m = g.numNodes(), g.numArcs()
    rowptr = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    succ = torch.empty(m, dtype=torch.int32, device="cuda")
    t0 = time.perf_counter()
    arcs = g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m)
    print("first scan %.1f ms, %d arcs" % ((time.perf_counter() - t0) * 1e3, arcs), flush=True)
    h = g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1)
    hs = g.hashCode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m)
    dt = time.perf_counter() - t0
    og = O.OracleGraph.load(base)
    want = og.hashcode_mt()
    ok = arcs == m and h == want and hs == want
    # the long rows: arithmetic progressions; rows at both ends from the oracle
    for x in (0, 1):
        a = int(rowptr[x].item())
        row = succ[a:a + D]
        ok = ok and int(rowptr[x + 1].item()) - a == D and bool(torch.equal(row.to(torch.int64), torch.arange(D, device="cuda", dtype=torch.int64) * step))
    ids = np.concatenate([np.arange(2, 200), np.arange(n - 200, n)]).astype(np.int32)
    rp, sc = g.successors_batch(ids)
    for k, x in enumerate(ids):
        ok = ok and np.array_equal(sc[rp[k]:rp[k + 1]], og.successors(int(x)))
    print("slow-test shape: n %d m %d (> 2^32: %s) | scan %.1f ms = %.1f G edges/s | hashCode scan/fold %s, the long rows and rows at both ends vs oracle: %s" % (
        n, m, m > 2**32, dt * 1e3, m / dt / 1e9, "ok" if h == want and hs == want else "MISMATCH (%d %d want %d)" % (h, hs, want), "ok" if ok else "MISMATCH"))
    og.close()
    g.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
