#!/usr/bin/env python3
"""GPU box: C4 (10 M uniform random successors(x) queries on the C2 graph) with ids and outputs resident in HBM:
time of one bvg_successors_batch call, bit-exact check of a sample against the oracle."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(nq=None, out=sys.stdout):
    import numpy as np
    import torch
    import bench
    from webgraph_amd import bvgraph as B
    from oracle import oracle as O
    n, m = 10_000_000, 200_000_000
    nq = nq or int(os.environ.get("C4_QUERIES", "10000000"))
    base, meta = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    g = B.BVGraph.load(base)
    rng = np.random.Generator(np.random.PCG64(0x5EEDB5E70004))
    q = rng.integers(0, n, size=nq, dtype=np.int64).astype(np.int32)
    if os.environ.get("C4_TOP"):  # the K longest rows instead of uniform ids: few queries, many arcs
        q = np.ascontiguousarray(np.argsort(g.outdegrees())[::-1][:int(os.environ["C4_TOP"])].astype(np.int32))
        nq = q.size
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(q).to(dev)
    d_rowptr = torch.empty(nq + 1, dtype=torch.int64, device=dev)
    arcs = C.c_uint64(0)
    lib = B.lib()
    fl = B.BVG_OUT_DEVICE
    rc = lib.bvg_successors_batch(g._h, d_q.data_ptr(), nq, d_rowptr.data_ptr(), None, 0, C.byref(arcs), fl)
    assert rc == 0, rc
    d_succ = torch.empty(max(arcs.value, 1), dtype=torch.int32, device=dev)
    times = []
    for it in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.bvg_successors_batch(g._h, d_q.data_ptr(), nq, d_rowptr.data_ptr(), d_succ.data_ptr(), d_succ.numel(), C.byref(arcs), fl)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        assert rc == 0, rc
    dt = min(times[1:])
    k = min(100_000, nq)
    og = O.OracleGraph.load(base)
    orp, osc = og.successors_batch(q[:k])
    rp = d_rowptr[:k + 1].cpu().numpy()
    sc = d_succ[:int(orp[-1])].cpu().numpy()
    ok = np.array_equal(rp, orp) and np.array_equal(sc, osc)
    # the last k queries too (a long batch's successors pass 2^31 on the way), and the total
    orp, osc = og.successors_batch(q[nq - k:])
    rp = d_rowptr[nq - k:].cpu().numpy()
    sc = d_succ[int(rp[0]):int(rp[-1])].cpu().numpy()
    ok = ok and np.array_equal(rp - rp[0], orp) and np.array_equal(sc, osc) and int(rp[-1]) == arcs.value
    print("C4 device-resident: %d queries, %d arcs: %.2f ms = %.1f M queries/s, %.2f G edges/s (all runs ms: %s), first and last %d bit-exact: %s"
          % (nq, arcs.value, dt * 1e3, nq / dt / 1e6, arcs.value / dt / 1e9, " ".join("%.1f" % (t * 1e3) for t in times), k, ok), file=out)
    g.close()
    return {"queries": nq, "arcs_out": int(arcs.value), "gpu_ms_device_resident": dt * 1e3, "gpu_queries_per_s": nq / dt, "gpu_edges_per_s": arcs.value / dt,
            "parity": "first and last %d queries bit-exact vs oracle: %s" % (k, ok)}


def main():
    run()


if __name__ == "__main__":
    main()
