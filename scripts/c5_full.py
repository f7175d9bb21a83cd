#!/usr/bin/env python3
"""GPU box: C5 at its FULL size (100 M nodes / 2 G arcs, deep reference chains; BASELINE.json configs[4], quoted for 8 GPUs) on ONE GPU: the eight bits-balanced slices that
`bench.py --gpus 8 --workload C5` gives its eight ranks (bvg_open_shard), opened and scanned one after the other on this GPU -- the ranks' own code path, no
collective on the data path -- with the whole graph's hashCode folded from the slices' maps against the CPU oracle's; then the whole graph as one scan.  What eight GPUs would take is
the longest slice (the ranks share nothing but the files).  usage: c5_full.py [nodes arcs [slices [C5|C2]]]   (105900000 3740000000 8 C2: a synthetic graph of the size of C3, uk-2007-05 -- more than 2^31 arcs in one scan)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from oracle import oracle as O
    from webgraph_amd import parallel as P
    from webgraph_amd.bvgraph import BVGraph
    n = int(sys.argv[1]) if len(sys.argv) > 2 else 100_000_000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000_000
    parts = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    wl = bench.WORKLOADS[sys.argv[4] if len(sys.argv) > 4 else "C5"]  # (the recipe: C5's deep chains, or C2's)
    t0 = time.time()
    base, meta = bench.prepare_graph(n, m, wl["seed"], wl["p_copy"], "/tmp/bvgpu_cache", os.cpu_count(), p_same=wl["p_same"], p_keep=wl["p_keep"])
    print("graph ready in %.0f s: .graph %.2f GB, %.3f bits/link" % (time.time() - t0, os.path.getsize(base + ".graph") / 1e9, os.path.getsize(base + ".graph") * 8 / m), flush=True)
    og = O.OracleGraph.load(base)
    t0 = time.time()
    want = og.hashcode_mt()
    print("oracle hashCode %d (%.1f s, %d threads)" % (want, time.time() - t0, os.cpu_count()), flush=True)
    og.close()
    dev = torch.device("cuda", 0)
    maps, times, arcs_all = [], [], 0
    for k in range(parts):
        t0 = time.perf_counter()
        g = BVGraph.load_shard(base, k, parts, device=0)
        t_open = time.perf_counter() - t0
        lo, hi = int(g.info.shard_from), int(g.info.shard_to)
        rowptr = torch.empty(hi - lo + 1, dtype=torch.int64, device=dev)
        arcs = g.decode_range_device(lo, hi, rowptr.data_ptr(), None, 0)
        succ = torch.empty(max(arcs, 1), dtype=torch.int32, device=dev)
        assert g.decode_range_device(lo, hi, rowptr.data_ptr(), succ.data_ptr(), succ.numel()) == arcs
        h0 = g.csr_hashcode(lo, hi, rowptr.data_ptr(), succ.data_ptr(), 0)
        h1 = g.csr_hashcode(lo, hi, rowptr.data_ptr(), succ.data_ptr(), 1)
        maps.append(P.affine_from_two_hashes(h0, h1))
        for _ in range(2):
            g.decode_range_device(lo, hi, rowptr.data_ptr(), succ.data_ptr(), succ.numel(), asynchronous=True)
        g.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            g.decode_range_device(lo, hi, rowptr.data_ptr(), succ.data_ptr(), succ.numel(), asynchronous=True)
        g.sync()
        dt = (time.perf_counter() - t0) / 5
        times.append(dt)
        arcs_all += arcs
        print("slice %d/%d: nodes [%d, %d) %d arcs | open %.0f ms | scan %.3f ms = %.1f G edges/s" % (k, parts, lo, hi, arcs, t_open * 1e3, dt * 1e3, arcs / dt / 1e9), flush=True)
        g.close()
        del rowptr, succ
    got = P.fold_affine(maps)
    print("slices: %d arcs, folded hashCode %d %s the oracle's | longest slice %.3f ms -> %d GPUs, one slice each: %.1f G edges/s (sum of the slices on this one GPU: %.2f ms)" % (
        arcs_all, got, "==" if got == want and arcs_all == m else "!=", max(times) * 1e3, parts, m / max(times) / 1e9, sum(times) * 1e3), flush=True)
    ok = got == want and arcs_all == m
    # the whole graph as ONE scan on this GPU
    g = BVGraph.load(base)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    succ = torch.empty(m, dtype=torch.int32, device=dev)
    arcs = g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m)
    h = g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1)
    hs = g.hashCode()  # (bvg_scan_checksum: the fold inside the scan, piece by piece)
    st = g.scan_stats(0, n)
    for _ in range(2):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m, asynchronous=True)
    g.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m, asynchronous=True)
    g.sync()
    dt = (time.perf_counter() - t0) / 5
    stats_ok = int(st["arcs"]) == m
    print("whole graph, one scan on one GPU: %d arcs, hashCode %s (folded inside the scan: %s; bvg_scan_stats counts %d arcs) | %.2f ms = %.1f G edges/s" % (
        arcs, "ok" if h == want else "MISMATCH", "ok" if hs == want else "MISMATCH", int(st["arcs"]), dt * 1e3, m / dt / 1e9), flush=True)
    g.close()
    return 0 if ok and h == want and hs == want and stats_ok and arcs == m else 1


if __name__ == "__main__":
    sys.exit(main())
