#!/usr/bin/env python3
"""GPU box: the parity-test configurations of BASELINE.md timed once (not bench lines): C1 scan of the cnr-2000
fixture, C4 random access (10 M uniform ids on the C2 graph).  Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import numpy as np
    import torch
    import bench
    import __graft_entry__ as ge
    ge.build()
    from webgraph_amd.bvgraph import BVGraph
    from oracle import oracle as O
    dev = torch.device("cuda", 0)
    out = {}
    # ---- C1
    base = os.path.join(ROOT, "tests", "golden", "cnr-2000")
    g = BVGraph.load(base)
    n, m = g.numNodes(), g.numArcs()
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    succ = torch.empty(m, dtype=torch.int32, device=dev)
    g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m)
    assert g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1) == 1711395807
    for _ in range(3):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m, asynchronous=True)
    g.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m, asynchronous=True)
    g.sync()
    dt = (time.perf_counter() - t0) / 10
    og = O.OracleGraph.load(base)
    t0 = time.perf_counter()
    og.scan()
    cdt = time.perf_counter() - t0
    out["C1"] = {"graph": "cnr-2000", "nodes": n, "arcs": m, "gpu_ms": dt * 1e3, "gpu_edges_per_s": m / dt, "hashCode": 1711395807,
                 "cpu_oracle_edges_per_s": m / cdt, "cpu_cores": 1}
    g.close()
    # ---- C4
    base, _ = bench.prepare_graph(10_000_000, 200_000_000, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    g = BVGraph.load(base)
    n = g.numNodes()
    rng = np.random.Generator(np.random.PCG64(0x5EEDB5E70004))
    q = rng.integers(0, n, size=10_000_000, dtype=np.int64).astype(np.int32)
    g.successors_batch(q)  # first call: staging buffers and arenas are allocated
    t0 = time.perf_counter()
    rp, sc = g.successors_batch(q)
    dt = time.perf_counter() - t0
    og = O.OracleGraph.load(base)
    k = 200_000
    orp, osc = og.successors_batch(q[:k])
    assert np.array_equal(rp[:k + 1], orp) and np.array_equal(sc[:int(orp[-1])], osc), "random access differs from the oracle"
    t0 = time.perf_counter()
    og.successors_batch(q[:k])
    cdt = time.perf_counter() - t0
    out["C4"] = {"graph": "C2 synthetic", "queries": int(q.size), "arcs_out": int(rp[-1]), "gpu_wall_ms_host_buffers": dt * 1e3,
                 "gpu_queries_per_s": q.size / dt, "gpu_edges_per_s": float(rp[-1]) / dt, "parity": "first %d queries bit-exact vs oracle" % k,
                 "cpu_oracle_queries_per_s": k / cdt, "cpu_cores": 1}
    g.close()
    import c4_time
    out["C4_device_resident"] = c4_time.run(10_000_000, out=sys.stderr)  # ids and outputs in HBM, one call
    print(json.dumps(out))


if __name__ == "__main__":
    main()
