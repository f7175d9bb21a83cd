#!/usr/bin/env python3
"""GPU box: the largest graph the format allows by NODES (n <= 2^31 - 1, ids are Java ints; BVGraph.java:1537): n = 2 147 483 647 nodes, one row in 64 non-empty
(two successors each, the last node with a loop on the largest id).  Stored by the CPU writer, loaded, scanned in one call, hashCode against the oracle, the last rows and a batch.
Every node index, row start and grid dimension of the library meets its 32-bit limit here.  usage: max_nodes.py [nodes]"""
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
    base = "/tmp/bvgpu_cache/maxnodes_%d" % n
    os.makedirs("/tmp/bvgpu_cache", exist_ok=True)
    if not os.path.exists(base + ".properties"):
        t0 = time.time()
        rows = np.arange(0, n, 64, dtype=np.int64)
        deg = np.zeros(n, dtype=np.int8)
        deg[rows] = 2
        deg[n - 1] = 1 if (n - 1) % 64 else 2
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(deg, out=rowptr[1:])
        del deg
        succ = np.empty(int(rowptr[-1]), dtype=np.int32)
        a = rowptr[rows]
        succ[a] = np.minimum(rows + 1, n - 2).astype(np.int32)
        succ[a + 1] = np.minimum(rows + 1000003, n - 1).astype(np.int32)
        if (n - 1) % 64:
            succ[-1] = n - 1
        bad = succ[a] >= succ[a + 1]
        succ[a[bad]] = succ[a[bad] + 1] - 1
        T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=4, zeta_k=3, threads=os.cpu_count())
        print("generated and stored in %.0f s: %d arcs" % (time.time() - t0, succ.size), flush=True)
        del succ, rowptr, rows, a
    print(".graph %.2f GB .offsets %.2f GB" % (os.path.getsize(base + ".graph") / 1e9, os.path.getsize(base + ".offsets") / 1e9), flush=True)
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
    ids = np.concatenate([np.arange(n - 300, n), np.arange(0, 300), (np.random.default_rng(1).integers(0, n // 64, 2000) * 64)]).astype(np.int32)
    rp, sc = g.successors_batch(ids)
    for k, x in enumerate(ids):
        ok = ok and np.array_equal(sc[rp[k]:rp[k + 1]], og.successors(int(x)))
    tail = g.decode_range(n - 1000, n)
    orp, osc, _ = og.scan(n - 1000, n)
    ok = ok and np.array_equal(tail[0], orp) and np.array_equal(tail[1], osc)
    od = g.outdegrees(n - 130, n)
    ok = ok and np.array_equal(od, np.diff(orp)[-130:])
    st = g.scan_stats(0, n)  # (Stats.java:111-160 on the device: every node and arc counted once)
    ok = ok and int(st["nodes"]) == n and int(st["arcs"]) == m and int(st["max_outdegree"]) == 2 and int(st["dangling"]) == n - (n + 63) // 64 - (1 if (n - 1) % 64 else 0)
    print("max nodes: n %d m %d | scan %.2f ms = %.1f G nodes/s | hashCode scan/fold %s, rows at both ends, a batch, a sub-range at the end vs oracle: %s" % (
        n, m, dt * 1e3, n / dt / 1e9, "ok" if h == want and hs == want else "MISMATCH (%d %d want %d)" % (h, hs, want), "ok" if ok else "MISMATCH"))
    og.close()
    if os.environ.get("MAXN_EF"):  # the same lists as an EFGraph image in HBM (bvg_cache_as_efgraph: decode, re-encode, switch): the second format's kernels at the same limit
        t0 = time.perf_counter()
        g.cache_as_efgraph()
        t1 = time.perf_counter()
        arcs2 = g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m)
        h2 = g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1)
        hs2 = g.hashCode()
        ok2 = arcs2 == m and h2 == want and hs2 == want
        print("as an EFGraph image (%.1f s to re-encode): hashCode scan/fold %s" % (t1 - t0, "ok" if ok2 else "MISMATCH (%d %d want %d)" % (h2, hs2, want)))
        ok = ok and ok2
    g.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
