#!/usr/bin/env python3
"""GPU box: bvg_scan_checksum (the hash folded inside the scan) against the scan that materialises every row and the old decode-then-fold path.
usage: checksum_time.py [c2|c5|cnr30] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    import torch
    from ab_time import workload
    from webgraph_amd.bvgraph import BVGraph
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    base = workload(name)
    g = BVGraph.load(base)
    n, m = g.numNodes(), g.numArcs()
    dev = torch.device("cuda", 0)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
    want = g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1)

    def t(f):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, r
    only = os.environ.get("CK_ONLY")  # counter passes: one of the two paths alone (fold | materialise), marked by a k_totals-free region: every kernel of the run counts
    if only:
        g.set_option("hash_materialise", 1 if only == "materialise" else 0)
        for _ in range(reps):
            assert g.scan_checksum() == (want, m)
        print("CK_ONLY=%s: %d checksum scans (+ 1 plain scan and 1 fold from memory at start-up)" % (only, reps))
        g.close()
        return
    scan_ms, _ = t(lambda: g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel()))
    g.set_option("hash_materialise", 0)
    fold_ms, r = t(lambda: g.scan_checksum())
    assert r == (want, m), (r, want, m)
    g.set_option("hash_materialise", 1)
    mat_ms, r = t(lambda: g.scan_checksum())
    assert r == (want, m), (r, want, m)
    print("%-6s arcs %d hash %d | scan (rows to the caller) %.3f ms | checksum folded in the scan %.3f ms | checksum decode-then-fold %.3f ms" % (name, m, want, scan_ms, fold_ms, mat_ms))
    g.close()


if __name__ == "__main__":
    main()
