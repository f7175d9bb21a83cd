#!/usr/bin/env python3
"""GPU box: bvg_scan_stats (Stats.run's scan, rows kept on the device) against the scan that hands the rows to a caller and against bvg_scan_checksum, on a cached
workload; BVGPU_SCAN_PIECE sets the arcs per piece of the consumers' scans.  usage: stats_time.py [c2|c5|cnr30] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.ab_time import workload


def main():
    import torch
    from webgraph_amd.bvgraph import BVGraph
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    g = BVGraph.load(workload(name))
    n, m = g.numNodes(), g.numArcs()
    dev = torch.device("cuda", 0)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)

    def timed(f):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    t_scan = timed(lambda: g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel()))
    t_stats = timed(lambda: g.scan_stats(0, n))
    t_sum = timed(lambda: g.scan_checksum(0, n, 0))
    st = g.scan_stats(0, n)
    print("%-6s piece %s | scan to caller %.3f ms | scan_stats %.3f ms | scan_checksum %.3f ms | arcs %d loops %d" % (
        name, os.environ.get("BVGPU_SCAN_PIECE", "(default)"), t_scan, t_stats, t_sum, st["arcs"], st["loops"]))
    g.close()


if __name__ == "__main__":
    main()
