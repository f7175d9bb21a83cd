#!/usr/bin/env python3
"""GPU box: corrupted bit streams at a size where every kernel class runs (long rows, deep chains).  Every case flips
a few bits of a valid .graph file; the library must answer with an error or a well-formed CSR -- never hang or crash.
usage: fuzz_corrupt.py [cases] [seed]   (run it under `timeout`)"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import BVGraph, BvgError
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    errs = (ValueError, IOError, RuntimeError, MemoryError, OSError, NotImplementedError, BvgError)
    tmp = tempfile.mkdtemp(prefix="bvcorrupt")
    n, m = 200_000, 8_000_000
    rowptr, succ = T.generate(n, m, seed=99, p_copy=0.8)
    variants = []
    for name, kw in (("z3", dict(zeta_k=3)), ("z5", dict(zeta_k=5)), ("w1", dict(zeta_k=3, window=1, max_ref_count=30))):
        base = os.path.join(tmp, name)
        args = dict(window=7, max_ref_count=3, min_interval=3)
        args.update(kw)
        T.store(base, rowptr, succ, threads=8, **args)
        variants.append((base, open(base + ".graph", "rb").read()))
    print("max outdegree %d" % int(np.diff(rowptr).max()), flush=True)
    outcomes = {"error": 0, "csr": 0}
    for c in range(cases):
        base, good = variants[c % len(variants)]
        raw = bytearray(good)
        for _ in range(int(rng.integers(1, 9))):
            pos = int(rng.integers(0, len(raw) // int(rng.choice([1, 1, 4, 64]))))
            raw[pos] ^= 1 << int(rng.integers(0, 8))
        open(base + ".graph", "wb").write(bytes(raw))
        only = os.environ.get("FUZZ_ONLY")
        skip = only is not None and c != int(only)
        os.environ["BVGPU_BATCH_DENSE"] = str(rng.choice(["0", "32", "1000000000"]))
        if rng.random() < 0.4:
            os.environ["BVGPU_COOP_MIN"] = "64"; os.environ["BVGPU_GIANT_MIN"] = "2000"
        if os.environ.get("FUZZ_VERBOSE") and c >= int(os.environ["FUZZ_VERBOSE"]):
            print("case %d env dense=%s coop=%s" % (c, os.environ.get("BVGPU_BATCH_DENSE"), os.environ.get("BVGPU_COOP_MIN")), flush=True)
        g = None if skip else BVGraph.load(base)
        if only is not None and not skip and os.environ.get("FUZZ_KEEP"):
            import shutil
            for ext in (".graph", ".offsets", ".properties"):
                shutil.copy(base + ext, os.path.join(os.environ["FUZZ_KEEP"], "case" + ext))
        for k in ("BVGPU_BATCH_DENSE", "BVGPU_COOP_MIN", "BVGPU_GIANT_MIN"):
            os.environ.pop(k, None)
        import time as _t
        for what in range(3):
            _t0 = _t.perf_counter()
            if os.environ.get("FUZZ_VERBOSE") and c >= int(os.environ["FUZZ_VERBOSE"]):
                print("case %d op %d start (%s)" % (c, what, base), flush=True)
            try:
                if what == 0:
                    if skip: continue
                    rp, sc = g.decode_range()
                elif what == 1:
                    lo = int(rng.integers(0, n - 1000)); hi = lo + int(rng.integers(1, 50000)) if lo + 50000 < n else n
                    if skip: continue
                    rp, sc = g.decode_range(lo, hi)
                else:
                    qq = rng.integers(0, n, size=int(10 ** rng.uniform(0, 4.7))).astype(np.int32)
                    if skip: continue
                    rp, sc = g.successors_batch(qq)
                assert rp[0] == 0 and np.all(np.diff(rp) >= 0) and rp[-1] == sc.size, "malformed CSR"
                outcomes["csr"] += 1
            except errs:
                outcomes["error"] += 1
            if _t.perf_counter() - _t0 > 1.0:
                print("slow request: case %d op %d took %.1f s (BATCH_DENSE/COOP env of this case: see seed)" % (c, what, _t.perf_counter() - _t0), flush=True)
        if g is not None:
            g.close()
        if c % 10 == 0:
            print("case %d ok %s" % (c, outcomes), flush=True)
    print("corrupt fuzz: %d cases, outcomes %s" % (cases, outcomes))


if __name__ == "__main__":
    main()
