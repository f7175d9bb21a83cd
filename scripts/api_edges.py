#!/usr/bin/env python3
"""GPU box: the degenerate calls of every entry point on cnr-2000 -- empty ranges at 0, in the middle and at n, an empty batch, a batch of one id many times, ranges of one node,
the last node, device and host outputs -- each against the oracle or the obvious answer.  Nothing may crash, hang or answer with a stale result."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from webgraph_amd import bvgraph as B
    from oracle import oracle as O
    base = os.path.join(ROOT, "tests", "golden", "cnr-2000")
    g = B.BVGraph.load(base)
    og = O.OracleGraph.load(base)
    n = g.numNodes()
    orp, osc, _ = og.scan()
    bad = []

    def chk(name, cond):
        if not cond:
            bad.append(name)
    for lo in (0, 1234, n):
        rp, sc = g.decode_range(lo, lo)
        chk("empty range at %d" % lo, rp.size == 1 and rp[0] == 0 and sc.size <= 1)
        h, arcs = g.scan_checksum(lo, lo, 77)
        chk("checksum of an empty range at %d" % lo, h == 77 and arcs == 0)
        st = g.scan_stats(lo, lo)
        chk("stats of an empty range", int(st["nodes"]) == 0 and int(st["arcs"]) == 0)
        chk("outdegrees of an empty range", g.outdegrees(lo, lo).size == 0)
        chk("equal_range empty", g.equal_range(g, lo, lo) is True)
    for x in (0, n - 1, 1234):
        rp, sc = g.decode_range(x, x + 1)
        chk("range of node %d" % x, np.array_equal(sc[:rp[1]], osc[orp[x]:orp[x + 1]]))
        chk("successorArray(%d)" % x, np.array_equal(g.successorArray(x), osc[orp[x]:orp[x + 1]]))
        chk("outdegree(%d)" % x, g.outdegree(x) == orp[x + 1] - orp[x])
    rp, sc = g.successors_batch(np.empty(0, dtype=np.int32))
    chk("empty batch", rp.size == 1 and rp[0] == 0)
    big = int(np.argmax(np.diff(orp)))
    for x in (big, 0, n - 1):
        q = np.full(5000, x, dtype=np.int32)
        rp, sc = g.successors_batch(q)
        row = osc[orp[x]:orp[x + 1]]
        chk("one id 5000 times (%d)" % x, np.array_equal(np.diff(rp), np.full(5000, row.size)) and np.array_equal(sc.reshape(5000, -1) if row.size else sc[:0], np.tile(row, (5000, 1)) if row.size else sc[:0]))
    q = np.arange(n - 1, -1, -1, dtype=np.int32)  # every node, backwards
    rp, sc = g.successors_batch(q)
    chk("all nodes backwards", np.array_equal(np.diff(rp), np.diff(orp)[::-1]) and np.array_equal(sc[rp[5]:rp[6]], osc[orp[n - 6]:orp[n - 5]]) and int(rp[-1]) == osc.size)
    # the same empty calls with device outputs
    d_rp = torch.empty(4, dtype=torch.int64, device="cuda")
    d_sc = torch.empty(4, dtype=torch.int32, device="cuda")
    for lo in (0, n):
        chk("device empty range", g.decode_range_device(lo, lo, d_rp.data_ptr(), d_sc.data_ptr(), 4) == 0 and int(d_rp[0].item()) == 0)
    for call, args in (("decode_range", (5, 4)), ("decode_range", (0, n + 1)), ("decode_range", (-1, 3)), ("outdegrees", (0, n + 1)), ("scan_checksum", (3, 2, -1)), ("successorArray", (n,)), ("successorArray", (-1,))):
        try:
            getattr(g, call)(*args)
            bad.append("%s%s did not raise" % (call, args))
        except (ValueError, B.BvgError):
            pass
    chk("hashCode after all that", g.hashCode() == 1711395807)
    print("api edges: %s" % ("ok" if not bad else "BAD: " + "; ".join(bad)))
    g.close(); og.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
