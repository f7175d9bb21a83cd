#!/usr/bin/env python3
"""GPU box: zeta_k far from the default (8 .. 31: the run-time-k variant of the kernels up to 16, the generic readers beyond), alone and with other non-default codings: scan, hashCode(), a
batch against the input and the oracle.  (Golomb residuals only with k = 3: the reference writes `zetak` into .properties for zeta residuals only (BVGraph.java:2566) and reads a Golomb graph back
with the default modulus 3 (:1543, :469) -- a Golomb graph stored with another modulus is unreadable by the reference itself, and this library follows the files.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from webgraph_amd import tools as T
from webgraph_amd.bvgraph import BVGraph, flags_from_string
from oracle import oracle as O
bad = 0
for k in (3, 8, 12, 16, 17, 24, 31):
    for fl in ("", "RESIDUALS_GOLOMB", "OUTDEGREES_DELTA | RESIDUALS_ZETA"):
        if fl == "RESIDUALS_GOLOMB" and k != 3: continue
        rowptr, succ = T.generate(300000, 6000000, seed=k, p_copy=0.6)
        base = "/tmp/zk_%d_%d" % (k, len(fl))
        T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=4, zeta_k=k, flags=flags_from_string(fl) if fl else 0, threads=8)
        g = BVGraph.load(base); og = O.OracleGraph.load(base)
        rp, sc = g.decode_range()
        ok = np.array_equal(rp, rowptr) and np.array_equal(sc, succ) and g.hashCode() == og.hashcode_mt()
        q = np.random.default_rng(k).integers(0, 300000, 3000).astype(np.int32)
        brp, bsc = g.successors_batch(q)
        for i, x in enumerate(q[:300]):
            ok = ok and np.array_equal(bsc[brp[i]:brp[i + 1]], succ[rowptr[x]:rowptr[x + 1]])
        print("k=%d flags=[%s]: %s" % (k, fl, "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
        g.close(); og.close()
sys.exit(bad)
