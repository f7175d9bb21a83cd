#!/usr/bin/env python3
"""GPU box: random graphs stored on the device with random parameters (window, maxRefCount, minIntervalLength, zeta_k, coding flags, parts,
and the internal thresholds of the wave walk and of the segment cutting) -- every case byte-compared with the CPU writer's three files, with
BVGPU_ENC_VERIFY pricing the waves' pairs lane by lane as well; the same lists through the EFGraph writer and reader.
usage: fuzz_store.py [cases] [seed]"""
import filecmp
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DELTA, GAMMA, GOLOMB, UNARY, ZETA, NIBBLE = 1, 2, 3, 5, 6, 7


def random_graph(rng, n):
    import numpy as np
    rows = []
    for x in range(n):
        kind = rng.integers(0, 10)
        prev = rows[x - int(rng.integers(1, 8))] if x >= 8 else []
        row = set()
        if kind == 0:
            pass
        elif kind <= 3 and len(prev):
            keep = rng.random(len(prev)) < rng.choice([0.2, 0.6, 0.9, 1.0])
            row = set(np.asarray(prev)[keep].tolist())
        if kind in (2, 3, 4, 5):
            base = int(rng.integers(0, n))
            for _ in range(int(rng.integers(0, 4))):
                s = base + int(rng.integers(0, 300))
                row |= set(range(s, min(n, s + int(rng.integers(1, 200)))))
        if kind >= 4:
            row |= set(int(v) for v in rng.integers(0, n, size=int(rng.pareto(1.2) * 8) % (n // 2 + 1)))
        rows.append(sorted(v for v in row if 0 <= v < n))
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    succ = np.array([v for r in rows for v in r], dtype=np.int32)
    return rowptr, succ


def main():
    import numpy as np
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.Generator(np.random.PCG64(seed))
    os.environ["BVGPU_ENC_VERIFY"] = "1"
    tmp = tempfile.mkdtemp(prefix="fuzz_store_")
    for c in range(cases):
        n = int(rng.choice([50, 700, 3000, 9000]))
        rowptr, succ = random_graph(rng, n)
        W = int(rng.choice([0, 1, 3, 7, 7, 7, 16, 40]))
        R = int(rng.choice([0, 1, 3, 3, 8, 1000]))
        I = int(rng.choice([0, 1, 2, 3, 4, 4, 9, 70]))
        K = int(rng.choice([1, 2, 3, 3, 5]))
        flags = 0
        if rng.random() < 0.4:
            flags = (int(rng.choice([0, DELTA])) << 0) | (int(rng.choice([0, DELTA, UNARY])) << 4) | (int(rng.choice([0, GAMMA, DELTA, NIBBLE, GOLOMB])) << 8) | \
                    (int(rng.choice([0, GAMMA, DELTA])) << 12) | (int(rng.choice([0, DELTA, UNARY])) << 16) | (int(rng.choice([0, DELTA])) << 20)
        parts = int(rng.choice([1, 1, 2, 5]))
        env = {"BVGPU_ENC_BIGBIN": str(int(rng.choice([4, 6, 8, 8, 32]))), "BVGPU_ENC_SEGBIN": str(int(rng.choice([4, 7, 9, 16]))), "BVGPU_ENC_SEGELEMS": str(int(rng.choice([9, 64, 500, 8192])))}
        os.environ.update(env)
        cpu, gpu = os.path.join(tmp, "cpu"), os.path.join(tmp, "gpu")
        st = T.store(cpu, rowptr, succ, window=W, max_ref_count=R, min_interval=I, zeta_k=K, flags=flags, threads=parts)
        try:
            sg = B.store(rowptr, succ, gpu, windowSize=W, maxRefCount=R, minIntervalLength=I, zetaK=K, flags=flags, numberOfThreads=parts)
        except Exception as e:  # noqa: BLE001
            print("case %d FAILED to store: %r  n=%d W=%d R=%d I=%d K=%d flags=%#x parts=%d env=%s seed=%d" % (c, e, n, W, R, I, K, flags, parts, env, seed))
            sys.exit(1)
        for ext in (".graph", ".offsets", ".properties"):
            if not filecmp.cmp(cpu + ext, gpu + ext, shallow=False):
                print("case %d MISMATCH in %s  n=%d W=%d R=%d I=%d K=%d flags=%#x parts=%d env=%s seed=%d" % (c, ext, n, W, R, I, K, flags, parts, env, seed))
                np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_store_case.npz"), rowptr=rowptr, succ=succ)
                sys.exit(1)
        assert all(st[k] == sg[k] for k in st)
        # the same lists as an EFGraph: device writer == CPU writer, device reader == the lists
        lq, big = int(rng.integers(0, 9)), bool(rng.integers(0, 2))
        ub = None if rng.random() < 0.7 else n + int(rng.integers(0, 1000))
        T.store_ef(cpu + "e", rowptr, succ, upper_bound=ub, log2_quantum=lq, big_endian=big)
        B.store_ef(rowptr, succ, gpu + "e", upperBound=ub, log2Quantum=lq, bigEndian=big)
        for ext in (".graph", ".offsets", ".properties"):
            if not filecmp.cmp(cpu + "e" + ext, gpu + "e" + ext, shallow=False):
                print("case %d EF MISMATCH in %s  n=%d lq=%d big=%s ub=%s seed=%d" % (c, ext, n, lq, big, ub, seed))
                sys.exit(1)
        h = B.EFGraph.load(gpu + "e")
        rp, sc = h.decode_range()
        assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ), "EF decode differs in case %d" % c
        h.close()
        if c % 20 == 19:
            print("%d cases equal" % (c + 1), flush=True)
    print("fuzz_store: %d cases, every file equal to the CPU writer's (seed %d)" % (cases, seed))


if __name__ == "__main__":
    main()
