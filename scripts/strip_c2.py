"""strip_try.py on the bench workloads (C2 / C5 shard / cnr x30 are generated into the cache first)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "C2"
wl = bench.WORKLOADS[which]
base = bench.prepare_graph(wl["n"], wl["m"], wl["seed"], wl["p_copy"], "/tmp/bvgpu_cache", os.cpu_count() or 1, p_same=wl["p_same"], p_keep=wl["p_keep"])[0]
os.execv(sys.executable, [sys.executable, os.path.join(ROOT, "scripts", "strip_try.py"), base])
