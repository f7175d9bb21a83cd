#!/usr/bin/env python3
"""GPU box: the consumers that keep the rows on the device, on random graphs (power-law lengths, runs of long rows now and then), random ranges, random piece
sizes: bvg_scan_stats (with indegrees), bvg_hyperball_step (random counters, with and without `modified` flags, m = 16 / 64 / 256), bvg_bfs_expand and bvg_equal_range
(the graph against a re-encoding of itself, and against one with a single id changed), each against the numpy restatements of tests/test_gpu_consumers.py.
usage: fuzz_consumers.py [cases] [seed]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import BVGraph
    from test_gpu_consumers import stats_restated, hyperball_restated
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    tmp = tempfile.mkdtemp(prefix="bvfuzzc")
    bad = 0
    for c in range(cases):
        n = int(10 ** rng.uniform(2.0, 4.9))
        m = min(int(n * 10 ** rng.uniform(0.3, 1.7)), n * min(n // 4, 100000) // 2)
        deep = rng.random() < 0.3
        kw = dict(p_copy=0.85, p_same=0.95, p_keep=0.95) if deep else dict(p_copy=float(rng.uniform(0, 0.9)))
        piece = str(int(rng.choice([0, 0, 2000, 50000, 1000000])))
        log2m = int(rng.choice([4, 6, 8]))
        desc = "n=%d m=%d %s piece=%s log2m=%d" % (n, m, "deep" if deep else "p=%.2f" % kw["p_copy"], piece, log2m)
        which = []
        try:
            rowptr, succ = T.generate(n, m, seed=int(rng.integers(1, 1 << 30)), **kw)
            base = os.path.join(tmp, "g%d" % c)
            T.store(base, rowptr, succ)
            T.store(base + "b", rowptr, succ, window=3, max_ref_count=2, min_interval=2)
            succ2 = succ.copy()
            row = int(np.nonzero(np.diff(rowptr))[0][-1])
            changed = succ2[rowptr[row + 1] - 1] < n - 1
            if changed:
                succ2[rowptr[row + 1] - 1] += 1
            T.store(base + "c", rowptr, succ2)
            g, gb, gc = BVGraph.load(base), BVGraph.load(base + "b"), BVGraph.load(base + "c")
            g.set_option("scan_piece", piece)
            lo = int(rng.integers(0, n)); hi = int(rng.integers(lo, n + 1))
            if rng.random() < 0.5:
                lo, hi = 0, n
            indeg = torch.zeros(n, dtype=torch.int32, device="cuda")
            if g.scan_stats(lo, hi, indeg.data_ptr()) != stats_restated(rowptr, succ, lo, hi): which.append("stats[%d,%d)" % (lo, hi))
            if not np.array_equal(indeg.cpu().numpy(), np.bincount(succ[rowptr[lo]:rowptr[hi]], minlength=n).astype(np.int32)): which.append("indegrees")
            mm = 1 << log2m
            regs = (rng.integers(0, 64, size=(n, mm)) * (rng.random((n, mm)) < 0.15)).astype(np.uint8)
            mod = (rng.random(n) < 0.5).astype(np.uint8) if rng.random() < 0.5 else None
            d_in = torch.from_numpy(regs).cuda()
            d_out = d_in.clone()
            d_mo = torch.zeros(n, dtype=torch.uint8, device="cuda")
            d_mi = torch.from_numpy(mod).cuda() if mod is not None else None
            ch = g.hyperball_step(log2m, d_in.data_ptr(), d_out.data_ptr(), d_mi.data_ptr() if d_mi is not None else None, d_mo.data_ptr(), lo, hi)
            want, wmod = hyperball_restated(rowptr, succ, regs, mod, lo, hi)
            if not (np.array_equal(d_out.cpu().numpy(), want) and np.array_equal(d_mo.cpu().numpy()[lo:hi], wmod[lo:hi]) and ch == int(wmod.sum())): which.append("hyperball[%d,%d)" % (lo, hi))
            frontier = np.unique(rng.integers(0, n, size=max(1, n // 9))).astype(np.int32)
            marker = np.full(n, -1, dtype=np.int32)
            marker[frontier] = 0
            d_marker, d_front = torch.from_numpy(marker).cuda(), torch.from_numpy(frontier).cuda()
            d_q = torch.empty(n, dtype=torch.int32, device="cuda")
            cnt = g.bfs_expand(d_front.data_ptr(), frontier.size, d_marker.data_ptr(), 1, False, d_q.data_ptr(), n)
            reach = np.unique(np.concatenate([succ[rowptr[x]:rowptr[x + 1]] for x in frontier] + [np.empty(0, dtype=np.int32)]))
            reach = reach[marker[reach] == -1]
            if not (cnt == reach.size and np.array_equal(np.sort(d_q[:cnt].cpu().numpy()), reach)): which.append("bfs")
            if not (g.equal_range(gb, lo, hi) and g.equals(gb) and gb.equals(g)): which.append("equals(same)")
            if changed and (g.equals(gc) or g.equal_range(gc, 0, n) or (row >= lo and row < hi) == g.equal_range(gc, lo, hi)): which.append("equals(changed)")
            for h in (g, gb, gc):
                h.close()
            for sfx in ("", "b", "c"):
                for ext in (".graph", ".offsets", ".properties"):
                    os.remove(base + sfx + ext)
        except Exception as ex:  # noqa: BLE001
            which.append("EXCEPTION %r" % (ex,))
        if which:
            bad += 1
            print("MISMATCH", desc, "FAILED:", ", ".join(which), flush=True)
        elif c % 10 == 0:
            print("ok", c, desc, flush=True)
    print("fuzz_consumers: %d cases, %d bad" % (cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
