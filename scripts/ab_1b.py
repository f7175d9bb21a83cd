#!/usr/bin/env python3
"""GPU box: generates (once per box) the north star's graph (50 M nodes / 1 B arcs, the C2 recipe) and prints its basename for scripts/ab_time.py."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
print(bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())[0])
