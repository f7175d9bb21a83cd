#!/usr/bin/env python3
"""GPU box: the C2 graph stored as an EFGraph (second format), scanned on the GPU: time, edges/s and the share of
the HBM peak for the algorithmic bytes of SURVEY 8(d) (file + offsets read, successors + rowptr written); the BVGraph scan of the
same lists beside it.  usage: ef_time.py [nodes] [arcs]"""
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
    base, _ = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    g = B.BVGraph.load(base)
    n, m = g.numNodes(), g.numArcs()
    rp = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    sc = torch.empty(m, dtype=torch.int32, device="cuda")

    def scan(h, reps=10):
        h.decode_range_device(0, n, rp.data_ptr(), sc.data_ptr(), m)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            h.decode_range_device(0, n, rp.data_ptr(), sc.data_ptr(), m)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best
    tbv = scan(g)
    want_rp, want_sc = rp.clone(), sc.clone()
    ef = base + "_ef"
    if not os.path.exists(ef + ".properties"):
        t0 = time.perf_counter()
        T.store_ef(ef, want_rp.cpu().numpy(), want_sc.cpu().numpy())
        print("EFGraph written by the CPU writer in %.1f s" % (time.perf_counter() - t0))
    h = B.EFGraph.load(ef)
    tef = scan(h)
    assert torch.equal(rp, want_rp) and torch.equal(sc, want_sc), "EFGraph scan differs from the BVGraph scan"
    for name, t, bytes_ in (("BVGraph", tbv, g.info.graph_bytes), ("EFGraph", tef, h.info.graph_bytes)):
        alg = bytes_ + 8 * (n + 1) + 4 * m + 8 * (n + 1)
        print("%s: %.1f MB file (%.2f bits/link), scan %.3f ms = %.1f G edges/s; algorithmic %.0f MB -> %.0f GB/s = %.1f %% of 8 TB/s" % (
            name, bytes_ / 1e6, 8 * bytes_ / m, t * 1e3, m / t / 1e9, alg / 1e6, alg / t / 1e9, 100 * alg / t / 8e12))
    q = torch.randint(0, n, (10_000_000,), dtype=torch.int32, device="cuda")
    brp = torch.empty(q.numel() + 1, dtype=torch.int64, device="cuda")
    bsc = torch.empty(int(m * 1.2) + 1024, dtype=torch.int32, device="cuda")
    import ctypes as C
    for name, hh in (("BVGraph", g), ("EFGraph", h)):
        arcs = C.c_uint64(0)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            rc = B.lib().bvg_successors_batch(hh._h, q.data_ptr(), q.numel(), brp.data_ptr(), bsc.data_ptr(), bsc.numel(), C.byref(arcs), B.BVG_OUT_DEVICE)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
            assert rc == 0, rc
        print("%s: 10 M random lists (%d arcs) in %.2f ms = %.1f G edges/s" % (name, arcs.value, best * 1e3, arcs.value / best / 1e9))
    # the in-HBM cache: a BVGraph handle re-encodes its lists as an EFGraph image and answers from it
    g2 = B.BVGraph.load(base)
    t0 = time.perf_counter()
    g2.cache_as_efgraph()
    tc = time.perf_counter() - t0
    tcs = scan(g2)
    assert torch.equal(rp, want_rp) and torch.equal(sc, want_sc)
    print("bvg_cache_as_efgraph on the BVGraph handle: %.1f ms once (decode + re-encode in HBM, %.1f MB image), then scan %.3f ms = %.1f G edges/s (was %.3f ms)" % (
        tc * 1e3, g2.info.graph_bytes / 1e6, tcs * 1e3, m / tcs / 1e9, tbv * 1e3))
    g2.close()
    # hashCode(): the BVGraph scan decodes into scratch and folds; the EFGraph scan folds inside its decode kernels (nothing is written)
    for name, hh in (("BVGraph", g), ("EFGraph", h)):
        hv = hh.hashCode()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            assert hh.hashCode() == hv
            best = min(best, time.perf_counter() - t0)
        print("%s: hashCode() = %d by a checksum scan in %.3f ms = %.1f G edges/s" % (name, hv, best * 1e3, m / best / 1e9))
    # EFGraph.store on the device from the decoded CSR (BVGPU_ENC_TRACE=1 prints the device time; the call also writes the three files)
    t0 = time.perf_counter()
    B.store_ef(want_rp, want_sc, "/tmp/bvgpu_cache/ef_dev")
    print("bvg_store_ef from HBM incl. copying back and writing 443 MB of files: %.2f s" % (time.perf_counter() - t0))
    import filecmp
    print("device writer's .graph equal to the CPU writer's:", filecmp.cmp("/tmp/bvgpu_cache/ef_dev.graph", ef + ".graph", shallow=False))
    g.close(); h.close()


if __name__ == "__main__":
    main()
