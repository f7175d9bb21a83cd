#!/usr/bin/env python3
"""bench.py -- decoded edges/s of a full sequential BVGraph scan on MI355X (BASELINE.json metric).

One "step" = one full sequential scan of the workload graph: every node's successor list materialised as
int32 in CSR order (rowptr int64[n+1] + succ int32[m]) by libbvgpu, with the .graph bit stream and the
offset table already resident in HBM.  Protocol as the reference's SpeedTest (3 warm-up + 10 timed scans,
src/it/unimi/dsi/webgraph/test/SpeedTest.java:45-46, :167-182).

Workload (config C2 of BASELINE.json / SURVEY.md section 8(d)): synthetic power-law graph, 10M nodes / 200M arcs,
zeta_3 residuals, window 7, maxRefCount 3, minIntervalLength 4, seed 0x5EEDB5E70001 (generator + writer:
webgraph_amd/csrc/host/bvg_tools.cpp).  With --gpus N every rank scans its own C2-sized shard (seed + rank):
node ranges are independent, there is no data-path collective, only a reduction of (arcs, time) -- weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling is ~6290 GB/s
SEED = 0x5EEDB5E70001


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def prepare_graph(n, m, seed, p_copy, cache_dir, threads, p_same=0.0, p_keep=0.7):
    """Generates + compresses the synthetic workload unless it is already in the cache directory."""
    from webgraph_amd import tools as T
    os.makedirs(cache_dir, exist_ok=True)
    base = os.path.join(cache_dir, "syn_n%d_m%d_s%x_p%02d" % (n, m, seed, int(round(p_copy * 100))))
    if p_same:
        base += "_r%02d_k%02d" % (int(round(p_same * 100)), int(round(p_keep * 100)))
    done = base + ".done"
    if not os.path.exists(done):
        t0 = time.time()
        rowptr, succ = T.generate(n, m, seed=seed, p_copy=p_copy, threads=threads, p_same=p_same, p_keep=p_keep)
        t1 = time.time()
        st = T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=4, zeta_k=3, flags=0, threads=threads)
        t2 = time.time()
        with open(done, "w") as f:
            json.dump({"gen_s": t1 - t0, "store_s": t2 - t1, "stats": st}, f)
        del rowptr, succ
        log("[bench] generated %s: gen %.1fs store %.1fs bits/link %.3f" % (base, t1 - t0, t2 - t1, st["written_bits"] / max(m, 1)))
    with open(done) as f:
        meta = json.load(f)
    return base, meta


def cpu_baseline(base, budget_s=20.0):
    """Times the CPU oracle (single thread) on a bounded prefix of the same graph; returns (dict, full_hash_or_None)."""
    from oracle import oracle as O
    g = O.OracleGraph.load(base)
    n = g.n
    # calibrate on a small prefix, then size the sample to ~budget_s of CPU work
    probe = min(n, 200_000)
    t0 = time.perf_counter()
    _, _, arcs = g.scan(0, probe, want_succ=False)
    dt = max(time.perf_counter() - t0, 1e-6)
    rate = arcs / dt
    total_arcs = g.arcs or 0
    want_nodes = n if total_arcs / max(rate, 1) <= budget_s else max(probe, int(n * (budget_s * rate) / max(total_arcs, 1)))
    want_nodes = min(n, want_nodes)
    # the timed sample materialises successors, like the GPU path (and SpeedTest's successorArray())
    import numpy as np
    t0 = time.perf_counter()
    rp, sc, arcs, h = g.scan(0, want_nodes, want_succ=True, want_hash=True, cap=int(total_arcs) if want_nodes == n else None)
    dt = time.perf_counter() - t0
    out = {"value": arcs / dt, "unit": "edges/s", "cores": 1, "kind": "port",
           "sample": "oracle/bvg_oracle.c sequential scan of nodes [0,%d) = %d arcs in %.2fs, successors materialised, 1 thread" % (want_nodes, arcs, dt)}
    # for information: the same restatement on every host core, split like ImmutableGraph.splitNodeIterators
    # (ImmutableGraph.java:379-409: ceil(n/T) contiguous nodes per thread; ctypes releases the GIL during the scan)
    try:
        import concurrent.futures as cf
        T = max(1, min(os.cpu_count() or 1, 256))
        if T > 1:
            per = -(-want_nodes // T)
            rngs = [(k * per, min(want_nodes, (k + 1) * per)) for k in range(T) if k * per < want_nodes]
            t0 = time.perf_counter()
            with cf.ThreadPoolExecutor(max_workers=len(rngs)) as ex:
                tot = sum(r[2] for r in ex.map(lambda ab: g.scan(ab[0], ab[1], want_succ=True, cap=int(rp[ab[1]] - rp[ab[0]])), rngs))
            dtm = time.perf_counter() - t0
            out["all_cores"] = {"value": tot / dtm, "unit": "edges/s", "cores": len(rngs),
                                "sample": "the same %d nodes split into %d contiguous ranges, one thread each, %.2fs" % (want_nodes, len(rngs), dtm)}
    except Exception as e:  # the single-thread figure is the baseline; this one is a courtesy
        out["all_cores"] = {"error": str(e)}
    return out, (h if want_nodes == n else None), (rp, sc, want_nodes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--arcs", type=int, default=200_000_000)
    ap.add_argument("--p-copy", type=float, default=0.5)
    ap.add_argument("--cache", default=os.environ.get("BVGPU_CACHE", "/tmp/bvgpu_cache"))
    ap.add_argument("--graph", default=None, help="basename of an existing BVGraph to scan instead of the synthetic workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        log("[bench] WORLD_SIZE=%d but --gpus %d: using WORLD_SIZE" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from webgraph_amd.bvgraph import BVGraph

    threads = max(1, (os.cpu_count() or 1) // max(world, 1))
    if args.graph:
        base, meta = args.graph, {}
        workload = "graph:" + os.path.basename(args.graph)
    else:
        base, meta = prepare_graph(args.nodes, args.arcs, SEED + rank, args.p_copy, args.cache, threads)
        workload = "C2 synthetic power-law %dM nodes / %dM arcs, zeta3, w=7, maxref=3, minint=4" % (args.nodes // 1_000_000, args.arcs // 1_000_000) \
            if (args.nodes, args.arcs) == (10_000_000, 200_000_000) else "synthetic power-law n=%d m=%d zeta3 w=7" % (args.nodes, args.arcs)

    g = BVGraph.load(base, device=local_rank)
    n, m = g.numNodes(), g.numArcs()
    stream = torch.cuda.Stream(device=dev)
    g.set_stream(stream.cuda_stream)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)

    def scan(asynchronous=True):
        return g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel(), asynchronous=asynchronous)

    # ---- parity gate: nothing is timed before the output is checked
    with torch.cuda.stream(stream):
        arcs = scan(asynchronous=False)
        assert arcs == m, "arc count %d != properties arcs %d" % (arcs, m)
        gpu_hash = g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1)
        d = rowptr[1:] - rowptr[:-1]
        assert int(d.min()) >= 0 and int(rowptr[-1]) == m
        if m > 1:
            # rows strictly increasing: the only non-increasing adjacent pairs sit on row boundaries
            bad = (succ[1:m] <= succ[:m - 1]).nonzero().flatten() + 1
            is_start = torch.zeros(m + 1, dtype=torch.bool, device=dev)
            is_start[rowptr.clamp(max=m)] = True
            assert bool(is_start[bad].all()), "a decoded successor list is not strictly increasing"
            del bad, is_start
        del d
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, full_hash, (orp, osc, on) = cpu_baseline(base, args.cpu_budget)
        import numpy as np
        assert np.array_equal(rowptr[:on + 1].cpu().numpy(), orp), "rowptr differs from the CPU oracle"
        assert np.array_equal(succ[:int(orp[-1])].cpu().numpy(), osc), "successors differ from the CPU oracle"
        if full_hash is not None:
            assert full_hash == gpu_hash, "hashCode mismatch: oracle %d vs GPU %d" % (full_hash, gpu_hash)
        del orp, osc
    parity = "bit-exact vs oracle" if cpu is not None else "arcs + sortedness + hash only"

    # ---- timed region
    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            scan()
        g.sync()
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(args.steps):
            scan()
        e1.record(stream)
        g.sync()
        fence()
        wall = time.perf_counter() - t0
        dev_ms = e0.elapsed_time(e1)

        # ---- per-kernel timing of the same scan with HIP events between the phases (outside the timed region;
        # with profiling on the library runs its normally overlapping parse kernels one after the other)
        g.set_profile(True)
        phases = {}
        reps = max(3, min(args.steps, 10))
        for _ in range(reps):
            scan()
            g.sync()
            for k, v in g.get_profile().items():
                phases[k] = phases.get(k, 0.0) + v / reps
        g.set_profile(False)

    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    cnt = torch.tensor([float(m)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    wall = float(t.item())
    total_m = float(cnt.item())
    info = g.info
    graph_bytes = int(info.graph_bytes)
    b_alg = graph_bytes + 8 * (n + 1) + 4 * m + 8 * (n + 1)
    kernel_of = {"headers": "k_headers", "scan": "k_scan_*", "lists": "k_depth_keys+k_scatter_keys", "parse_giant": "k_parse_big<8>", "parse_big": "k_parse_big<1>",
                 "parse_short": "k_parse_list", "copy": "k_copy_list+k_copy_mid+k_copy_big", "tail": "k_rebase"}
    # The dominant kernel is priced on the units IT processes (bits/8 + 4 B per successor
    # + 16 B per node of offsets and rowptr), not on the whole scan: the two parse kernels split the records by
    # outdegree at the library's BVGPU_COOP_MIN threshold (default 2048).
    coop_min = int(os.environ.get("BVGPU_COOP_MIN", "2048"))
    giant_min = max(coop_min, int(os.environ.get("BVGPU_GIANT_MIN", "32768")))
    with open(base + ".offsets", "rb") as f:
        from webgraph_amd.bvgraph import decode_offsets_host
        offs = decode_offsets_host(f.read(), n, 2 if "OFFSETS_DELTA" not in open(base + ".properties").read() else 1)
    import numpy as np
    deg = (rowptr[1:] - rowptr[:-1]).cpu().numpy()
    bits = np.diff(np.asarray(offs, dtype=np.int64))
    masks = {"parse_giant": deg >= giant_min, "parse_big": (deg >= coop_min) & (deg < giant_min), "parse_short": (deg < coop_min) & (deg > 0)}
    def alg_bytes(mask):
        return float(bits[mask].sum()) / 8.0 + 4.0 * float(deg[mask].sum()) + 16.0 * float(mask.sum())
    units = {k: alg_bytes(mk) for k, mk in masks.items()}
    dom_phase = max(masks, key=lambda k: phases.get(k, 0.0))
    dom = kernel_of[dom_phase]
    dom_ms = phases[dom_phase]
    dom_bytes = units[dom_phase]
    scan_ms = sum(phases.values())
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    out = {
        "metric": "decoded edges/sec, full sequential BVGraph scan",
        "value": total_m * args.steps / wall,
        "unit": "edges/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic" if not args.graph else "file",
        "config": {"workload": workload, "nodes_per_gpu": n, "arcs_per_gpu": m, "bits_per_link": graph_bytes * 8.0 / max(m, 1),
                   "parallelism": "node-range shards, one process per GPU, no collectives" if world > 1 else "1 GPU",
                   "parity": parity, "device_ms_per_step": dev_ms / args.steps,
                   "phase_ms": {k: round(v, 4) for k, v in phases.items()}},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel_algorithmic_bytes": dom_bytes, "kernel_ms": dom_ms,
                     "kernel_units": "%s: %d records, %d successors" % ({"parse_giant": "outdegree >= %d" % giant_min, "parse_big": "%d <= outdegree < %d" % (coop_min, giant_min),
                                                                          "parse_short": "0 < outdegree < %d" % coop_min}[dom_phase],
                                                                         int(masks[dom_phase].sum()), int(deg[masks[dom_phase]].sum())),
                     "algorithmic_bytes_per_scan": b_alg, "bytes_per_edge": b_alg / max(m, 1), "serial_phase_ms_sum": scan_ms,
                     "scan_achieved": b_alg / (dev_ms / args.steps * 1e-3) / 1e9,
                     "scan_frac": b_alg / (dev_ms / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(out), flush=True)
    g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
