#!/usr/bin/env python3
"""GPU box: windows and reference chains far beyond the defaults (the reference takes any windowSize / maxRefCount; the decoder never looks at maxrefcount): the copy model
with W up to 5 000 and unlimited chains (depths of hundreds), and PERIODIC graphs whose cheapest reference is hundreds or tens of thousands of nodes back (row x = row x - P).  Whole scan,
hashCode(), the second half as a sub-range (its halo), a batch, against the CPU writer's input and the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from webgraph_amd import tools as T
from webgraph_amd.bvgraph import BVGraph, flags_from_string
from oracle import oracle as O
bad = 0


def check(name, n, rowptr, succ, W, mr, threads, flags=0):
    global bad
    base = "/tmp/wide_%s" % name
    T.store(base, rowptr, succ, window=W, max_ref_count=mr, min_interval=4, zeta_k=3, flags=flags, threads=threads)
    og = O.OracleGraph.load(base)
    refs = og.references().astype(np.int64)
    depth = np.zeros(n, dtype=np.int32)
    for x in range(n):
        if refs[x]:
            depth[x] = depth[x - refs[x]] + 1
    g = BVGraph.load(base)
    t0 = time.perf_counter()
    rp, sc = g.decode_range()
    dt = time.perf_counter() - t0
    ok = np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    ok = ok and g.hashCode() == og.hashcode_mt()
    for lo in (n // 2, n - 1000):
        rp2, sc2 = g.decode_range(lo, n)
        ok = ok and np.array_equal(rp2, rowptr[lo:] - rowptr[lo]) and np.array_equal(sc2, succ[rowptr[lo]:])
    q = np.random.default_rng(1).integers(0, n, 5000).astype(np.int32)
    brp, bsc = g.successors_batch(q)
    for k, x in enumerate(q[:500]):
        ok = ok and np.array_equal(bsc[brp[k]:brp[k + 1]], succ[rowptr[x]:rowptr[x + 1]])
    print("%s: n %d W %d maxref %d: largest reference %d, deepest chain %d | scan %.1f ms | %s" % (name, n, W, mr, int(refs.max()), int(depth.max()), dt * 1e3, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
    g.close(); og.close()


for (P, W, mr, n) in [(777, 1000, 1000000, 100000), (777, 1000, 3, 100000), (20000, 25000, 1000000, 45000)]:  # (one compression thread: a thread's range starts with an empty window)
    rng = np.random.default_rng(P)
    proto = [np.unique(rng.integers(0, n, int(rng.integers(3, 30)))).astype(np.int32) for _ in range(P)]
    rows = []
    for x in range(n):
        r = proto[x % P]
        if x >= P and rng.random() < 0.3:  # a few rows differ a little from the one a period back: copy blocks, not just a whole copy
            r = np.unique(np.concatenate([r[::2], rng.integers(0, n, 2).astype(np.int32)]))
        rows.append(r)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum([r.size for r in rows])
    # (references in gamma: in unary -- the default -- a reference 777 nodes back costs 778 bits and never wins)
    check("periodic%d" % P + ("_r%d" % mr), n, rowptr, np.concatenate(rows), W, mr, 1, flags=flags_from_string("REFERENCES_GAMMA"))
for (n, m, W, mr, p) in [(200000, 4000000, 200, 1000000, 0.95), (300000, 3000000, 1000, 1000000, 0.99), (100000, 3000000, 5000, 50, 0.9), (2000000, 40000000, 64, 100000, 0.97)]:
    rowptr, succ = T.generate(n, m, seed=n + W, p_copy=p, p_same=0.9, p_keep=0.95)
    check("copymodel_%d_%d" % (n, W), n, rowptr, succ, W, mr, 1)
sys.exit(bad)
