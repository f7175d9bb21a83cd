import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench
from webgraph_amd.bvgraph import decode_offsets_device, decode_offsets_host, BVGraph
base, _ = bench.prepare_graph(10_000_000, 200_000_000, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())
raw = open(base + ".offsets", "rb").read()
n = 10_000_000
decode_offsets_device(raw, n)
t0 = time.perf_counter(); d = decode_offsets_device(raw, n); t1 = time.perf_counter(); h = decode_offsets_host(raw, n); t2 = time.perf_counter()
print("offsets file %.1f MB: device (incl. H2D of the file, D2H of 80 MB, scratch malloc) %.1f ms, host %.1f ms, equal %s" % (len(raw) / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, np.array_equal(d, h)))
t0 = time.perf_counter(); g = BVGraph.load(base); t1 = time.perf_counter(); g.close()
os.environ["BVGPU_OFFSETS"] = "host"
t2 = time.perf_counter(); g = BVGraph.load(base); t3 = time.perf_counter(); g.close()
print("bvg_open: %.1f ms with device offsets, %.1f ms with host offsets" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3))
