"""First contact of the strip kernel with the GPU: cnr-2000 whole + sub-range vs the oracle, with BVGPU_STRIP on and off, timing of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
ge.build()
from webgraph_amd.bvgraph import BVGraph
from oracle import oracle as O
base = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/cnr-2000"
og = O.OracleGraph.load(base)
orp, osc, oarcs = og.scan_mt() if og.n > 1000000 else og.scan()
for strip in ("1", "0"):
    os.environ["BVGPU_STRIP"] = strip
    g = BVGraph.load(base)
    n = g.numNodes()
    dev = torch.device("cuda", 0)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    m = g.decode_range_device(0, n, rowptr.data_ptr(), None, 0)
    succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    try:
        got = g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
    except Exception as e:
        print("strip", strip, "FAILED:", e); g.close(); continue
    ok = np.array_equal(rowptr.cpu().numpy(), orp) and np.array_equal(succ[:m].cpu().numpy(), osc)
    if not ok:
        a = succ[:m].cpu().numpy(); bad = np.nonzero(a != osc)[0]
        rows = np.searchsorted(orp, bad[:2000], side="right") - 1
        print("strip", strip, "MISMATCH: %d of %d ids differ; first rows %s" % (bad.size, m, sorted(set(rows.tolist()))[:20]))
        r0 = int(rows[0]); print(" row", r0, "d", int(orp[r0+1]-orp[r0]), "got", a[orp[r0]:orp[r0+1]][:16], "want", osc[orp[r0]:orp[r0+1]][:16])
    ts = []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("strip", strip, "ok" if ok else "WRONG", "arcs", m, "best %.3f ms  median %.3f ms" % (min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3), "thresholds", g.last_thresholds())
    if ok and n > 30000:
        lo, hi = 12345, min(n, 23456 + 50000)
        rp, sc = g.decode_range(lo, hi)
        print("  subrange", np.array_equal(sc, osc[orp[lo]:orp[hi]]))
    g.set_profile(True)
    g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
    print("  phases", {k: round(v, 3) for k, v in g.get_profile().items()})
    g.close()
