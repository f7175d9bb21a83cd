"""Scan time of one workload for a list of environment settings (each in a fresh process: knobs are read at bvg_open)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("SWEEP_CHILD"):
    import time, numpy as np, torch, bench
    from webgraph_amd.bvgraph import BVGraph
    which = sys.argv[1]
    if which in bench.WORKLOADS:
        wl = bench.WORKLOADS[which]
        base = bench.prepare_graph(wl["n"], wl["m"], wl["seed"], wl["p_copy"], "/tmp/bvgpu_cache", os.cpu_count() or 1, p_same=wl["p_same"], p_keep=wl["p_keep"])[0]
    elif which.startswith("cnr"):
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import ab_time
        base = ab_time.workload(which)
    else:
        base = which
    g = BVGraph.load(base)
    n = g.numNodes()
    dev = torch.device("cuda", 0)
    rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    m = g.decode_range_device(0, n, rowptr.data_ptr(), None, 0)
    succ = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
    h = g.csr_hashcode(0, n, rowptr.data_ptr(), succ.data_ptr(), -1)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    g.set_profile(True)
    g.decode_range_device(0, n, rowptr.data_ptr(), succ.data_ptr(), succ.numel())
    ph = {k: round(v, 3) for k, v in g.get_profile().items()}
    print(json.dumps({"best_ms": round(min(ts) * 1e3, 3), "median_ms": round(sorted(ts)[5] * 1e3, 3), "hash": h, "G_edges_s": round(m / min(ts) / 1e9, 1), "phases": ph}))
    sys.exit(0)
which = sys.argv[1]
for setting in sys.argv[2:]:
    env = dict(os.environ, SWEEP_CHILD="1")
    for kv in setting.split(","):
        if kv and kv != "default":
            k, v = kv.split("=")
            env[k] = v
    p = subprocess.run([sys.executable, __file__, which], env=env, capture_output=True, text=True)
    out = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print("%-44s %s" % (setting, out[-1] if out else "FAILED: " + p.stderr[-400:]), flush=True)
