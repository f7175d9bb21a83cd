#!/usr/bin/env python3
"""GPU box: times BVGraph.store on the device on a graph of the C2 recipe: decode the stored graph into HBM,
compress it there (bvg_compress, device pointers), compare the streams with the files the CPU writer produced, and time the CPU
writer on a slice.  usage: enc_time.py [nodes] [arcs] [reps]   (BVGPU_ENC_TRACE=1 prints the phases of every run)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 20 * n
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    base, _ = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    g = B.BVGraph.load(base)
    n, m = g.numNodes(), g.numArcs()
    rp = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    sc = torch.empty(m, dtype=torch.int32, device="cuda")
    g.decode_range_device(0, n, rp.data_ptr(), sc.data_ptr(), m)
    torch.cuda.synchronize()
    L = B.lib()
    best = None
    for i in range(reps + 1):
        c = B.BvgCompressed()
        err = C.create_string_buffer(512)
        t0 = time.perf_counter()
        rc = L.bvg_compress(0, n, rp.data_ptr(), sc.data_ptr(), B.BVG_OUT_DEVICE, 7, 3, 4, 3, 0, os.cpu_count(), C.byref(c), err, 512)
        dt = time.perf_counter() - t0
        assert rc == 0, err.value
        if i == 0:  # the CPU writer wrote the files with os.cpu_count() threads: same parts here
            graph = np.empty((c.graph_bits + 7) // 8, dtype=np.uint8)
            offs = np.empty((c.offsets_bits + 7) // 8, dtype=np.uint8)
            assert L.bvg_compressed_copy(C.byref(c), n, graph.ctypes.data, offs.ctypes.data, None) == 0
            same = graph.tobytes() == open(base + ".graph", "rb").read() and offs.tobytes() == open(base + ".offsets", "rb").read()
            print("streams equal to the CPU writer's files: %s (%d bits, %d selection rounds)" % (same, c.graph_bits, c.stats.selection_rounds))
            assert same
        else:
            best = dt if best is None else min(best, dt)
        L.bvg_compressed_free(C.byref(c))
    print("GPU bvg_compress (CSR in HBM -> streams in HBM): %.1f ms = %.2f G arcs/s" % (best * 1e3, m / best / 1e9))
    # CPU writer, one thread, on a slice of the same graph
    k = min(n, 500_000)
    rph = rp[:k + 1].cpu().numpy()
    sch = sc[:int(rph[-1])].cpu().numpy()
    t0 = time.perf_counter()
    T.store("/tmp/bvgpu_cache/enc_slice", rph, sch, threads=1)
    dt1 = time.perf_counter() - t0
    print("CPU writer, 1 thread, first %d nodes (%d arcs): %.2f s = %.2f M arcs/s" % (k, rph[-1], dt1, rph[-1] / dt1 / 1e6))
    g.close()


if __name__ == "__main__":
    main()
