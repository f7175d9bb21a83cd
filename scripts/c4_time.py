#!/usr/bin/env python3
"""GPU box: time of BASELINE configs[3] (10 M random ids on the C2 graph through bvg_successors_batch), without bench.py's parity gate -- for timing builds whose
results are garbage on purpose (BVGPU_DBG switches) and for rocprofv3 timelines.  usage: c4_time.py [reps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    base = bench.prepare_graph(10_000_000, 200_000_000, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())[0]
    g = B.BVGraph.load(base)
    dev = torch.device("cuda", 0)
    q = T.random_nodes(g.numNodes(), 10_000_000, seed=0x5EEDB5E70004)
    d_q = torch.from_numpy(q).to(dev)
    d_rowptr = torch.empty(q.size + 1, dtype=torch.int64, device=dev)
    arcs = C.c_uint64(0)
    lib = B.lib()

    def batch(succ_t):
        rc = lib.bvg_successors_batch(g._h, d_q.data_ptr(), q.size, d_rowptr.data_ptr(), succ_t.data_ptr() if succ_t is not None else None,
                                      succ_t.numel() if succ_t is not None else 0, C.byref(arcs), B.BVG_OUT_DEVICE)
        assert rc == 0, rc
    batch(None)
    d_succ = torch.empty(max(arcs.value, 1), dtype=torch.int32, device=dev)
    for _ in range(3):
        batch(d_succ)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        batch(d_succ)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("BVGPU_") and k != "BVGPU_LIB")
    print("c4     %-40s arcs %d | batch %.3f ms = %.2f G lists/s" % (knobs or "(defaults)", arcs.value, dt * 1e3, q.size / dt / 1e9), flush=True)
    g.close()


if __name__ == "__main__":
    main()
