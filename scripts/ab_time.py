#!/usr/bin/env python3
"""GPU box: time of a full scan (device-resident outputs) of a cached workload, overlapped and per phase, for A/B runs of the
library's knobs (environment variables, one process per variant).  usage: ab_time.py [c2|c5|cnr30] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workload(name):
    import numpy as np
    import bench
    from webgraph_amd import tools as T
    if name == "c2":
        return bench.prepare_graph(10_000_000, 200_000_000, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())[0]
    if name == "c5":
        return bench.prepare_graph(12_500_000, 250_000_000, 0x5EEDB5E70005, 0.85, "/tmp/bvgpu_cache", os.cpu_count(), p_same=0.95, p_keep=0.95)[0]
    if name.startswith("cnr"):
        K = int(name[3:] or 30)
        base = "/tmp/bvgpu_cache/cnr_x%d" % K
        if not os.path.exists(base + ".graph"):
            from oracle import oracle as O
            og = O.OracleGraph.load(os.path.join(ROOT, "tests", "golden", "cnr-2000"))
            rp, sc, _ = og.scan()
            n0, m0 = og.n, sc.size
            rowptr = np.concatenate([[0], (rp[1:][None, :] + (np.arange(K, dtype=np.int64) * m0)[:, None]).ravel()])
            succ = (sc[None, :].astype(np.int64) + (np.arange(K, dtype=np.int64) * n0)[:, None]).astype(np.int32).ravel()
            os.makedirs("/tmp/bvgpu_cache", exist_ok=True)
            T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=3, threads=os.cpu_count())
        return base
    return name


def main():
    import torch
    from webgraph_amd.bvgraph import BVGraph
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    base = workload(name)
    g = BVGraph.load(base)
    n, m = g.numNodes(), g.numArcs()
    dev = torch.device("cuda", 0)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    arcs = g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
    h = g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1)
    for _ in range(3):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel(), asynchronous=True)
    g.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel(), asynchronous=True)
    g.sync()
    dt = (time.perf_counter() - t0) / reps
    g.set_profile(not os.environ.get("AB_NO_PROFILE"))
    ph = {}
    for _ in range(0 if os.environ.get("AB_NO_PROFILE") else 3):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
        for k, v in g.get_profile().items():
            ph[k] = ph.get(k, 0.0) + v / 3
    g.set_profile(False)
    knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("BVGPU_") and k != "BVGPU_CACHE")
    print("%-6s %-40s thr %s arcs %d hash %d | scan %.3f ms = %.1f G edges/s | serial %s sum %.3f" % (
        name, knobs or "(defaults)", "%d/%d" % g.last_thresholds(), arcs, h, dt * 1e3, arcs / dt / 1e9, " ".join("%s %.3f" % (k, v) for k, v in ph.items()), sum(ph.values())))
    g.close()


if __name__ == "__main__":
    main()
