#!/usr/bin/env python3
"""GPU box: what a host-side caller (the JNI binding) sees for a full C2 scan: bvg_decode_range_view (one call, pinned
results owned by the handle), BVG_OUT_HOST into a pinned and into a pageable buffer of the caller.  usage: host_path_time.py [nodes arcs]   (the C2 recipe at that size)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import bench
    from webgraph_amd import bvgraph as B
    n = int(sys.argv[1]) if len(sys.argv) > 2 else 10_000_000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000_000
    base, _ = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
    g = B.BVGraph.load(base)
    from oracle import oracle as O
    og = O.OracleGraph.load(base)
    tail = og.scan(n - 1000, n)  # the last rows, wherever the host path's pieces end
    og.close()

    def check_tail(rp, sc):
        a = int(rp[n - 1000])
        assert int(rp[n]) == m and np.array_equal(np.asarray(rp[n - 1000:n + 1]) - a, tail[0]) and np.array_equal(np.asarray(sc[a:m]), tail[1]), "the last rows differ from the oracle's"

    def best(fn, reps=4):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts[1:]) * 1e3, ts[0] * 1e3

    ref = {}

    def view():
        rp, sc = g.decode_range_view(0, n)
        ref["sum"] = int(sc[::4097].astype(np.int64).sum())
        check_tail(rp, sc)
    t, first = best(view)
    print("bvg_decode_range_view       : %.1f ms (first call %.1f ms) = %.1f G edges/s at the host" % (t, first, m / t / 1e6))
    rp = np.empty(n + 1, dtype=np.int64)
    p = C.c_void_p()
    assert B.lib().bvg_host_alloc(4 * m, C.byref(p)) == 0
    sc = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(m,))
    t, first = best(lambda: g.decode_range_into(0, n, rp, sc))
    assert int(sc[::4097].astype(np.int64).sum()) == ref["sum"]
    check_tail(rp, sc)
    print("BVG_OUT_HOST, pinned succ   : %.1f ms (first %.1f) = %.1f G edges/s" % (t, first, m / t / 1e6))
    sc2 = np.empty(m, dtype=np.int32)
    t, first = best(lambda: g.decode_range_into(0, n, rp, sc2))
    assert int(sc2[::4097].astype(np.int64).sum()) == ref["sum"]
    check_tail(rp, sc2)
    print("BVG_OUT_HOST, pageable succ : %.1f ms (first %.1f) = %.1f G edges/s" % (t, first, m / t / 1e6))
    t, first = best(lambda: g.scan_checksum())
    print("bvg_scan_checksum           : %.1f ms (first %.1f) = %.1f G edges/s, hash %d" % (t, first, m / t / 1e6, g.scan_checksum()[0]))
    g.close()


if __name__ == "__main__":
    main()
