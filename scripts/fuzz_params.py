#!/usr/bin/env python3
"""GPU box: randomized parity sweep.  Random small graphs stored with random parameters (window, maxrefcount,
minintervallength, zeta_k, coding flags), decoded by the library (scan, sub-range, both batch strategies, small
cooperative thresholds now and then) and compared with the CPU oracle.  usage: fuzz_params.py [cases] [seed]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import BVGraph, flags_from_string
    from oracle import oracle as O
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    flagsets = ["", "", "", "", "RESIDUALS_DELTA", "RESIDUALS_GAMMA", "BLOCKS_DELTA | BLOCK_COUNT_DELTA", "OUTDEGREES_DELTA", "REFERENCES_GAMMA", "RESIDUALS_NIBBLE",
                "OUTDEGREES_DELTA | BLOCKS_DELTA | RESIDUALS_DELTA | REFERENCES_DELTA | BLOCK_COUNT_DELTA"]
    bad = 0
    wide = os.environ.get("FUZZ_WIDE") == "1"
    tmp = tempfile.mkdtemp(prefix="bvfuzz")
    for c in range(cases):
        n = int(10 ** rng.uniform(1.5, 4.7))
        m = min(int(n * 10 ** rng.uniform(0.3, 1.6)), n * min(n // 4, 100000) // 2)  # (bvt_generate: outdegrees are capped at n/4, mean at half the cap)
        p_copy = float(rng.uniform(0, 0.97))
        W = int(rng.choice([0, 1, 2, 3, 7, 7, 7, 12, 31]))
        mr = int(rng.choice([1, 2, 3, 3, 5, 10, 40])) if W else 0
        mi = int(rng.choice([0, 2, 3, 4, 4, 8]))
        k = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 7]))
        fl = str(rng.choice(flagsets))
        if wide:  # FUZZ_WIDE=1: windows, chains, interval lengths and zeta_k far from the defaults, the rarer codings
            W = int(rng.choice([0, 1, 5, 40, 100, 300]))
            mr = int(rng.choice([1, 4, 100, 1000000])) if W else 0
            mi = int(rng.choice([0, 1, 2, 16, 100]))
            k = int(rng.choice([1, 6, 8, 11, 16, 17, 30]))
            fl = str(rng.choice(flagsets + ["BLOCKS_DELTA | BLOCK_COUNT_UNARY", "REFERENCES_DELTA | OFFSETS_DELTA", "RESIDUALS_GOLOMB", "OUTDEGREES_DELTA | RESIDUALS_NIBBLE | BLOCKS_DELTA"]))
            if "GOLOMB" in fl:
                k = 3  # (the only modulus a Golomb graph can be read back with: .properties carries zetak for zeta residuals only, BVGraph.java:2566)
        env = {}
        if rng.random() < 0.3:
            env = {"BVGPU_COOP_MIN": str(int(rng.choice([8, 64, 300]))), "BVGPU_GIANT_MIN": str(int(rng.choice([300, 1000, 4000])))}
        dense = str(rng.choice(["0", "32", "1000000000"]))
        env["BVGPU_BATCH_DENSE"] = dense
        desc = "n=%d m=%d p=%.2f W=%d mr=%d mi=%d k=%d flags=[%s] env=%s" % (n, m, p_copy, W, mr, mi, k, fl, env)
        if os.environ.get("FUZZ_VERBOSE"):
            print("case", c, desc, file=sys.stderr, flush=True)
        try:
            rowptr, succ = T.generate(n, m, seed=int(rng.integers(1, 1 << 30)), p_copy=p_copy)
            base = os.path.join(tmp, "g%d" % c)
            T.store(base, rowptr, succ, window=W, max_ref_count=mr, min_interval=mi, zeta_k=k, flags=flags_from_string(fl) if fl else 0, threads=4)
            for kk, vv in env.items():
                os.environ[kk] = vv
            g = BVGraph.load(base)
            for kk in env:
                del os.environ[kk]
            og = O.OracleGraph.load(base)
            orp, osc, _ = og.scan()
            which = []
            ok = np.array_equal(orp, rowptr) and np.array_equal(osc, succ)
            if not ok: which.append("oracle")
            rp, sc = g.decode_range()
            if not (np.array_equal(rp, rowptr) and np.array_equal(sc, succ)): which.append("scan")
            lo = int(rng.integers(0, n)); hi = int(rng.integers(lo, n + 1))
            rp, sc = g.decode_range(lo, hi)
            if not (np.array_equal(rp, rowptr[lo:hi + 1] - rowptr[lo]) and np.array_equal(sc, succ[rowptr[lo]:rowptr[hi]])): which.append("range[%d,%d)" % (lo, hi))
            q = rng.integers(0, n, size=int(10 ** rng.uniform(0, 4))).astype(np.int32)
            rp, sc = g.successors_batch(q)
            brp, bsc = og.successors_batch(q)
            if not (np.array_equal(rp, brp) and np.array_equal(sc, bsc)): which.append("batch(%d)" % q.size)
            if g.hashCode() != og.hashcode(): which.append("hashCode")
            ok = not which
            if which: desc += " FAILED: " + ", ".join(which)
            g.close()
            for ext in (".graph", ".offsets", ".properties"):
                os.remove(base + ext)
        except Exception as ex:  # noqa: BLE001
            ok = False
            desc += " EXCEPTION %r" % (ex,)
        if not ok:
            bad += 1
            print("MISMATCH", desc, flush=True)
        elif c % 20 == 0:
            print("ok", c, desc, flush=True)
    print("fuzz: %d cases, %d bad" % (cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
