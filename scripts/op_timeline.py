#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace: one call of an entry point on a cached workload, three times; scripts/op_dump.py prints the kernels of the last call.
usage: op_timeline.py <c2|c5|cnr30|hubs> <scan|range|checksum|sparse|stats>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.ab_time import workload


def main():
    import numpy as np
    import torch
    from webgraph_amd.bvgraph import BVGraph
    name, op = sys.argv[1], sys.argv[2]
    if name == "hubs":  # scripts/hub_time.py's graph: three rows of 8 M / 4 M / 2 M successors
        from scripts.hub_time import build
        g = BVGraph.load(build([8_000_000, 4_000_000, 2_000_000]))
    else:
        g = BVGraph.load(workload(name))
    n, m = g.numNodes(), g.numArcs()
    dev = torch.device("cuda", 0)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    q = np.random.default_rng(3).integers(0, n, size=100_000).astype(np.int32)
    for _ in range(3):
        torch.cuda.synchronize()
        marker = torch.zeros(1 << 20, device=dev)  # (a fill kernel in front of every call: op_dump.py cuts there)
        torch.cuda.synchronize()
        if op == "range":
            g.decode_range_device(n // 3, n // 3 + n // 4, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
        elif op == "scan":
            g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
        elif op == "checksum":
            g.scan_checksum(0, n, -1)
        elif op == "sparse":
            g.successors_batch(q)
        elif op == "stats":
            g.scan_stats(0, n)
        torch.cuda.synchronize()
    g.close()


if __name__ == "__main__":
    main()
