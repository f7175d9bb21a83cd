#!/usr/bin/env python3
"""Tuning helper (GPU box): times the phases of a full scan and prints the cooperative decoder's counters.

usage: BVGPU_STATS=1 python scripts/tune.py [--nodes N --arcs M] [--graph basename] [--reps R]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--arcs", type=int, default=200_000_000)
    ap.add_argument("--graph", default=None)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    import __graft_entry__ as ge
    ge.build()
    from webgraph_amd.bvgraph import BVGraph
    if args.graph:
        base = args.graph
    else:
        base, _ = bench.prepare_graph(args.nodes, args.arcs, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    g = BVGraph.load(base)
    n, m = g.numNodes(), g.numArcs()
    dev = torch.device("cuda", 0)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    if os.environ.get("TUNE_TORCH_STREAM"):
        st_ = torch.cuda.Stream(device=dev)
        g.set_stream(st_.cuda_stream)
    g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
    if os.environ.get("BVGPU_STATS"):
        g.debug_stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel(), asynchronous=True)
    g.sync()
    wall = (time.perf_counter() - t0) / args.reps
    acc = {}
    if not os.environ.get("TUNE_NO_PROFILE"):
        g.set_profile(True)
    for _ in range(0 if os.environ.get("TUNE_NO_PROFILE") else args.reps):
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
        for k, v in g.get_profile().items():
            acc[k] = acc.get(k, 0) + v / args.reps
    g.set_profile(False)
    print("scan %.3f ms = %.2f Gedges/s | serial phases(ms): %s sum %.3f" % (wall * 1e3, m / wall / 1e9, {k: round(v, 3) for k, v in acc.items()}, sum(acc.values())))
    if os.environ.get("BVGPU_STATS"):
        st = g.debug_stats() // args.reps
        print("res tiles %d rounds/tile %.2f | int tiles %d rounds/tile %.2f | big nodes %d ticks/node %.0f max ticks %d (x reps)" % (
            st[0], st[1] / max(st[0], 1), st[2], st[3] / max(st[2], 1), st[5], st[6] / max(st[5], 1), st[7] * args.reps))
        print("copy_big rows %d sum bc %d max bc %d sum d %d max d %d fallback rows %d (x reps; max raw)" % (st[8], st[9], st[15] * args.reps, st[12], st[14] * args.reps, st[13]))
        print("giant residual tiles %d: first parse %.1f M ticks, local rounds %.1f M ticks, exchange phase %.1f M ticks (%.2f exchanges/tile)" % (st[11], st[12] / 1e6, st[13] / 1e6, st[14] / 1e6, st[10] / max(st[11], 1)))
        print("residual tile rounds histogram (<=2,<=4,<=8,<=16,<=32,<=64):", [int(v) for v in st[8:14]])
        print("ticks (M) NW=1: A %.1f I %.1f R+X %.1f | NW=16: A %.1f I %.1f R+X %.1f" % tuple(float(v) / 1e6 for v in (st[16], st[17], st[18], st[20], st[21], st[22])))
        print("giants residual ticks (M): stage %.0f rounds %.0f values %.0f rank %.0f flush %.0f fence %.0f expand %.0f loop %.0f" % tuple(float(v) / 1e6 for v in st[24:32]))
        print("wave class (NW=1) ticks (M): A %.0f I %.0f R %.0f X %.0f | residual steps: stage %.0f run-in %.0f first parse %.0f rounds %.0f scans+iv staging %.0f search+values %.0f tail %.0f | records %d tiles %d extra rounds %d avg B %.0f codes %d" % (
            tuple(float(v) / 1e6 for v in (st[16], st[17], st[18], st[19])) + tuple(float(v) / 1e6 for v in st[48:55]) + (int(st[57]), int(st[55]), int(st[56]), float(st[58]) / max(int(st[57]), 1), int(st[59]))))
        ns = max(int(st[13]), 1)
        print("slow tiles: avg B %.0f avg remaining codes %.0f avg codes in tile %.0f" % (st[15] / ns, st[4] / ns, st[14] / ns))
    if args.check:
        from oracle import oracle as O
        og = O.OracleGraph.load(base)
        import numpy as np
        rp, sc, _ = og.scan()
        print("parity:", np.array_equal(rp, rowptr.cpu().numpy()) and np.array_equal(sc, succ.cpu().numpy()))


if __name__ == "__main__":
    main()
