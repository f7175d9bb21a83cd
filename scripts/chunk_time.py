#!/usr/bin/env python3
"""GPU box: the C2 scan done in chunks of nodes (what a NodeIterator that pulls batches does): time per full pass."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from webgraph_amd.bvgraph import BVGraph
    n, m = 10_000_000, 200_000_000
    base, meta = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    g = BVGraph.load(base)
    dev = torch.device("cuda", 0)
    d_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    d_succ = torch.empty(m, dtype=torch.int32, device=dev)
    for chunk in (n, 2_500_000, 1_000_000, 250_000, 100_000, 25_000):
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for lo in range(0, n, chunk):
                hi = min(n, lo + chunk)
                g.decode_range_device(lo, hi, d_rowptr.data_ptr(), d_succ.data_ptr(), m)
            best = min(best, time.perf_counter() - t0)
        calls = (n + chunk - 1) // chunk
        print("chunk %8d nodes: %7.2f ms per pass over the graph, %d calls, %.3f ms per call" % (chunk, best * 1e3, calls, best * 1e3 / calls))
    g.close()


if __name__ == "__main__":
    main()


def pipelined():
    """The same chunks through K clones of the handle (each has its own streams and scratch), calls enqueued with
    BVG_ASYNC round-robin and collected in order: what a prefetching NodeIterator does."""
    import torch
    import bench
    from webgraph_amd.bvgraph import BVGraph
    n, m = 10_000_000, 200_000_000
    base, meta = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    g = BVGraph.load(base)
    dev = torch.device("cuda", 0)
    for K in (2, 4):
        clones = [g.copy() for _ in range(K)]
        rps = [torch.empty(n + 1, dtype=torch.int64, device=dev) for _ in range(K)]
        d_succ = torch.empty(m, dtype=torch.int32, device=dev)
        off = g.shard_bounds(1)  # noqa: F841
        for chunk in (2_500_000, 1_000_000, 250_000, 100_000):
            starts = list(range(0, n, chunk))
            # row offsets of the chunks in the shared output: from a count-only pass
            import numpy as np
            outd = g.outdegrees(0, n)
            cum = np.concatenate([[0], np.cumsum(outd, dtype=np.int64)])
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i, lo in enumerate(starts):
                    c = clones[i % K]
                    if i >= K:
                        c.sync()
                    hi = min(n, lo + chunk)
                    c.decode_range_device(lo, hi, rps[i % K].data_ptr(), d_succ.data_ptr() + 4 * int(cum[lo]), int(cum[hi] - cum[lo]), asynchronous=True)
                for c in clones:
                    c.sync()
                best = min(best, time.perf_counter() - t0)
            print("%d clones, chunk %8d nodes: %7.2f ms per pass, %.3f ms per call" % (K, chunk, best * 1e3, best * 1e3 / len(starts)))
        for c in clones:
            c.close()
    g.close()


if __name__ == "__main__" and os.environ.get("PIPELINED"):
    pipelined()
