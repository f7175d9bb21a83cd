import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench
from webgraph_amd.bvgraph import BVGraph
from oracle import oracle as O
base, meta = bench.prepare_graph(10_000_000, 200_000_000, 0x5EEDB5E70005, 0.85, "/tmp/bvgpu_cache", os.cpu_count())
g = BVGraph.load(base); n, m = g.numNodes(), g.numArcs()
dev = torch.device("cuda", 0)
rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev); succ = torch.empty(m, dtype=torch.int32, device=dev)
g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m)
og = O.OracleGraph.load(base); rp, sc, _ = og.scan()
print("parity", np.array_equal(rp, rowptr.cpu().numpy()) and np.array_equal(sc, succ.cpu().numpy()), "stats", {k: meta["stats"][k] for k in ("copied_arcs", "intervalised_arcs", "residual_arcs", "max_ref_chain")}, "bits/link", meta["stats"]["written_bits"] / m)
t0 = time.perf_counter()
for _ in range(10): g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m, asynchronous=True)
g.sync(); dt = (time.perf_counter() - t0) / 10
g.set_profile(True); g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), m); print("scan %.3f ms = %.1f Gedges/s" % (dt * 1e3, m / dt / 1e9), g.get_profile())
