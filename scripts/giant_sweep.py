#!/usr/bin/env python3
"""GPU box: the group-class threshold alone (BVGPU_GIANT_MIN; the wave class stays counted) over a workload.  usage: giant_sweep.py <workload|n,m> [G ...]"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    name = sys.argv[1]
    Gs = [int(a) for a in sys.argv[2:]] or [8192, 16384, 32768, 65536, 131072, 262144, 524288]
    if "," in name:
        import bench
        n, m = (int(x) for x in name.split(","))
        base = bench.prepare_graph(n, m, bench.SEED, 0.5, "/tmp/bvgpu_cache", os.cpu_count())[0]
    else:
        from threshold_sweep import build
        base = build(name)
    env0 = {k: v for k, v in os.environ.items() if k not in ("BVGPU_COOP_MIN", "BVGPU_GIANT_MIN")}
    env0["AB_NO_PROFILE"] = "1"

    def run(env):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ab_time.py"), base, "10"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        line = [l for l in p.stdout.decode().splitlines() if "| scan" in l][-1]
        return float(line.split("| scan")[1].split("ms")[0]), line.split("thr ")[1].split(" ")[0], p.stderr.decode()
    ms, thr, err = run(dict(env0, BVGPU_TRACE_HIST="1"))
    for l in err.splitlines():
        if "outdegree >=" in l:
            print(l)
    print("%-12s adaptive thr %-14s %.3f ms" % (name, thr, ms), flush=True)
    for G in Gs:
        m2, thr2, _ = run(dict(env0, BVGPU_GIANT_MIN=str(G)))
        print("%-12s giant %-8d thr %-14s %.3f ms" % (name, G, thr2, m2), flush=True)


if __name__ == "__main__":
    main()
