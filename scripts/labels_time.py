#!/usr/bin/env python3
"""GPU box: times the arc-label decode on the C2 graph with synthetic gamma / 13-bit labels."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import ArcLabelledBVGraph
    from oracle import oracle as O
    base, _ = bench.prepare_graph(10_000_000, 200_000_000, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    og = O.OracleGraph.load(base)
    rowptr, _, m = og.scan(0, og.n, want_succ=False)
    n = og.n
    rng = np.random.Generator(np.random.PCG64(11))
    dev = torch.device("cuda", 0)
    out = torch.empty(m, dtype=torch.int32, device=dev)
    for kind, width in (("gamma", 0), ("fixed", 13)):
        labels = (rng.pareto(1.0, size=m) * 3).astype(np.int64).clip(0, 8000).astype(np.int32) if kind == "gamma" else rng.integers(0, 8192, size=m, dtype=np.int32)
        lbase = os.path.join(os.path.dirname(base), "lab_" + kind)
        T.store_labels(lbase, os.path.basename(base), rowptr, labels, kind=kind, width=width)
        g = ArcLabelledBVGraph.load(lbase)
        g.decode_labels_device(0, n, m, out.data_ptr())
        assert np.array_equal(out.cpu().numpy(), labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.decode_labels_device(0, n, m, out.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        d = np.diff(rowptr).astype(np.int32)
        k = 2_000_000
        t0 = time.perf_counter()
        O.labels_decode(lbase, n, d[:k], 0, k)
        cdt = time.perf_counter() - t0
        print("%s labels: %.1f MB stream, GPU %.2f ms = %.1f G labels/s (bit-exact); CPU oracle 1 thread %.1f M labels/s" % (
            kind if kind == "gamma" else "fixed-%d" % width, g.info.labels_bytes / 1e6, dt * 1e3, m / dt / 1e9, rowptr[k] / cdt / 1e6))
        g.close()


if __name__ == "__main__":
    main()
