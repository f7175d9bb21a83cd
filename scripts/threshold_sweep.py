#!/usr/bin/env python3
"""GPU box: does the adaptive choice of the two class thresholds (wave class: counted by k_pick_coop; group class: by job size) hold on shapes it was not tuned on?
For each workload: the adaptive scan, then fixed pairs (BVGPU_COOP_MIN / BVGPU_GIANT_MIN pin both), one process each.  VERDICT r4 item 6.
usage: threshold_sweep.py <workload> [pairs...]     workloads: a08 | a16 (the C2 generator with outdegree density ~ d^-1.8 / d^-2.6), cnr100hubs, c2, c5, cnr30"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
PAIRS = [(128, 8192), (256, 8192), (512, 8192), (1024, 32768), (2048, 32768), (4096, 65536), (8192, 131072)]


def build(name):
    import numpy as np
    import bench
    from webgraph_amd import tools as T
    if name in ("a08", "a16"):
        os.environ["BVT_DEGREE_ALPHA"] = {"a08": "0.8", "a16": "1.6"}[name]
        base = bench.prepare_graph(10_000_000, 200_000_000, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())[0]
        del os.environ["BVT_DEGREE_ALPHA"]
        return base
    if name == "cnr100hubs":
        base = "/tmp/bvgpu_cache/cnr_x100_hubs"
        if not os.path.exists(base + ".graph"):
            from oracle import oracle as O
            og = O.OracleGraph.load(os.path.join(ROOT, "tests", "golden", "cnr-2000"))
            rp, sc, _ = og.scan()
            K, n0, m0 = 100, og.n, sc.size
            deg = np.tile(np.diff(rp), K).astype(np.int64)
            n = n0 * K
            rng = np.random.Generator(np.random.PCG64(77))
            hubs = rng.choice(n, size=n // 1000, replace=False)  # 0.1 % of the rows become hubs of 10^4 .. 10^6 successors (log-uniform)
            hub_deg = np.exp(rng.uniform(np.log(1e4), np.log(1e6), size=hubs.size)).astype(np.int64)
            hub_deg[:3] = [1_000_000, 600_000, 300_000]
            deg2 = deg.copy()
            deg2[hubs] = hub_deg
            rowptr = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(deg2, out=rowptr[1:])
            succ = np.empty(rowptr[-1], dtype=np.int32)
            src_rowptr = np.concatenate([[0], np.cumsum(deg)])
            is_hub = np.zeros(n, dtype=bool)
            is_hub[hubs] = True
            # the ordinary rows: cnr-2000's, shifted by the copy's first node (vectorised: positions of the non-hub rows are contiguous runs between hubs)
            base_ids = (sc[None, :].astype(np.int64) + (np.arange(K, dtype=np.int64) * n0)[:, None]).astype(np.int32).ravel()
            order = np.sort(hubs)
            prev = 0
            for h in list(order) + [n]:
                if h > prev:
                    succ[rowptr[prev]:rowptr[h]] = base_ids[src_rowptr[prev]:src_rowptr[h]]
                if h < n:
                    succ[rowptr[h]:rowptr[h + 1]] = np.sort(rng.choice(n, size=int(deg2[h]), replace=False)).astype(np.int32)
                prev = h + 1
            T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=3, threads=os.cpu_count())
        return base
    from ab_time import workload
    return workload(name)


def main():
    name = sys.argv[1]
    pairs = [tuple(int(x) for x in a.split("/")) for a in sys.argv[2:]] or PAIRS
    base = build(name)
    env0 = {k: v for k, v in os.environ.items() if k not in ("BVGPU_COOP_MIN", "BVGPU_GIANT_MIN")}
    env0["AB_NO_PROFILE"] = "1"

    def run(env):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ab_time.py"), base, "10"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        line = [l for l in out.splitlines() if "| scan" in l][-1]
        return float(line.split("| scan")[1].split("ms")[0]), line.split("thr ")[1].split(" ")[0]
    res = []
    ms, thr = run(env0)
    print("%-10s adaptive   thr %-14s %.3f ms" % (name, thr, ms), flush=True)
    for c, gm in pairs:
        e = dict(env0, BVGPU_COOP_MIN=str(c), BVGPU_GIANT_MIN=str(gm))
        m2, _ = run(e)
        res.append((m2, c, gm))
        print("%-10s fixed      thr %-14s %.3f ms" % (name, "%d/%d" % (c, gm), m2), flush=True)
    best = min(res)
    print("%-10s adaptive %.3f ms (%s) vs best fixed %.3f ms (%d/%d): %+.1f %%" % (name, ms, thr, best[0], best[1], best[2], (ms / best[0] - 1) * 100), flush=True)


if __name__ == "__main__":
    main()
