#!/usr/bin/env python3
"""GPU box: one HyperBall iteration (every counter modified) and one round of the breadth-first visit from a frontier of every 16th node, on a cached workload.
usage: consumers_time.py [c2|c5|cnr30] [log2m]"""
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
    log2m = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    g = BVGraph.load(workload(name))
    n, m = g.numNodes(), 1 << log2m
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    regs_in = torch.randint(0, 20, (n, m), dtype=torch.uint8, device=dev, generator=gen)
    regs_out = torch.empty_like(regs_in)
    mod_out = torch.empty(n, dtype=torch.uint8, device=dev)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ch = g.hyperball_step(log2m, regs_in.data_ptr(), regs_out.data_ptr(), None, mod_out.data_ptr())
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    frontier = torch.arange(0, n, 16, dtype=torch.int32, device=dev)
    marker = torch.full((n,), -1, dtype=torch.int32, device=dev)
    marker[frontier.long()] = 0
    out = torch.empty(n, dtype=torch.int32, device=dev)
    tb = []
    for _ in range(3):
        mk = marker.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nxt = g.bfs_expand(frontier.data_ptr(), frontier.numel(), mk.data_ptr(), 1, False, out.data_ptr(), out.numel())
        torch.cuda.synchronize()
        tb.append((time.perf_counter() - t0) * 1e3)
    print("%-6s hyperball step (m = %d, %d counters changed) %.2f ms | bfs round (frontier %d -> %d) %.2f ms" % (name, m, ch, min(ts), frontier.numel(), nxt, min(tb)))
    g.close()


if __name__ == "__main__":
    main()
