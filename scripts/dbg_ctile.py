import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from webgraph_amd import tools as T
from webgraph_amd.bvgraph import BVGraph
n, m = int(sys.argv[1]), int(sys.argv[2])
rowptr, succ = T.generate(n, m, seed=11, p_copy=0.6)
base = "/tmp/dbg_ct"
T.store(base, rowptr, succ, threads=16)
g = BVGraph.load(base)
import torch
dev = torch.device("cuda", 0)
rp = torch.empty(n + 1, dtype=torch.int64, device=dev)
sc = torch.empty(m, dtype=torch.int32, device=dev)
a = g.decode_range_device(0, n, rp.data_ptr(), sc.data_ptr(), m)
torch.cuda.synchronize()
ok = np.array_equal(sc.cpu().numpy(), succ)
print("n", n, "m", m, "arcs", a, "equal", ok, flush=True)
if not ok:
    bad = np.nonzero(sc.cpu().numpy() != succ)[0]
    x = np.searchsorted(rowptr, bad[0], side="right") - 1
    print("first bad element", bad[0], "row", x, "d", rowptr[x + 1] - rowptr[x], "nbad", bad.size)
