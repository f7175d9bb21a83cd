#!/usr/bin/env python3
"""GPU box: what a caller waits for before the first successor arrives: BVGraph.load (files -> HBM, .offsets decoded on the device), the first scan
(scratch is allocated inside it), a later scan, copy() and the copy's first scan.  usage: load_time.py [c2|c5|cnr30|1b|basename]"""
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
    base = workload(name)
    torch.zeros(1, device="cuda")
    sizes = {e: os.path.getsize(base + e) for e in (".graph", ".offsets")}
    for e in sizes:  # the files in the page cache: what is timed is the library, not the disk
        with open(base + e, "rb") as f:
            while f.read(1 << 26):
                pass

    def ms(f):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) * 1e3

    g, t_load = ms(lambda: BVGraph.load(base))
    n, m = g.numNodes(), g.numArcs()
    rowptr = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    succ = torch.empty(max(m, 1), dtype=torch.int32, device="cuda")
    scan = lambda h: h.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
    _, t_first = ms(lambda: scan(g))
    _, t_second = ms(lambda: scan(g))
    _, t_third = ms(lambda: scan(g))
    g2, t_copy = ms(lambda: g.copy())
    _, t_cfirst = ms(lambda: scan(g2))
    _, t_csecond = ms(lambda: scan(g2))
    g3, t_load2 = ms(lambda: BVGraph.load(base))
    print("%s: n %d m %d .graph %.1f MB .offsets %.1f MB | load %.1f ms (again %.1f) | scans %.2f / %.2f / %.2f ms | copy() %.2f ms, its scans %.2f / %.2f ms" % (
        name, n, m, sizes[".graph"] / 1e6, sizes[".offsets"] / 1e6, t_load, t_load2, t_first, t_second, t_third, t_copy, t_cfirst, t_csecond))
    if os.environ.get("BVGPU_TRACE_HOST"):
        pass
    g3.close(); g2.close(); g.close()


if __name__ == "__main__":
    main()
