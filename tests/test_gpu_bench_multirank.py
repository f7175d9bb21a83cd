"""The N > 1 path of bench.py (bvg_open_shard per rank + reduce_scan of (arcs, hash map) + the hashCode gate against the CPU
oracle) under test on ONE GPU: two ranks launched exactly as the driver launches them (python -m torch.distributed.run,
one process per rank, rendezvous on 127.0.0.1), both on cuda:0 with the gloo backend for the host-side reduction.  What
a real multi-GPU run adds is one device per rank and RCCL for the same three-integer all_gather; the sharding
(ImmutableGraph.splitNodeIterators, ImmutableGraph.java:379-409), the slice staging and the parity
gate are what runs here."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(workload, ranks, nodes, arcs, tmp_path):
    env = dict(os.environ)
    env["BVGPU_CACHE"] = str(tmp_path / "cache")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--one-device",
           "--workload", workload, "--nodes", str(nodes), "--arcs", str(arcs), "--steps", "3", "--warmup", "1", "--no-pmc", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, "bench.py failed:\n" + p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line, got %d" % len(lines)
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("workload,nodes,arcs", [("C2", 1_000_000, 20_000_000), ("C5", 600_000, 12_000_000)])
def test_two_rank_bench_line(tmp_path, workload, nodes, arcs):
    out = _run(workload, 2, nodes, arcs, tmp_path)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["unit"] == "edges/s"
    cfg = out["config"]
    assert cfg["arcs_total"] == 2 * arcs and cfg["nodes_total"] == 2 * nodes
    assert cfg["parity"].endswith("== CPU oracle's")
    assert out["value"] > 0 and abs(out["value"] - cfg["arcs_total"] / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]
    assert out.get("cpu_baseline") is None or isinstance(out["cpu_baseline"], dict)


def _bench(args, tmp_path, timeout=900):
    env = dict(os.environ)
    env["BVGPU_CACHE"] = str(tmp_path / "cache")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, "bench.py failed:\n" + p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_single_gpu_bench_line_prices_every_phase(tmp_path):
    """The N = 1 line at a small size: whole CSR against the oracle before anything is timed, `roofline` with the copy pass priced
    (4 B x ids of the rows with a reference and of their referents: SURVEY.md section 8(d)), `cpu_baseline` present."""
    out = _bench(["--nodes", "400000", "--arcs", "8000000", "--steps", "3", "--warmup", "1", "--no-pmc", "--cpu-budget", "2"], tmp_path)
    assert out["n_gpus"] == 1 and out["unit"] == "edges/s" and out["config"]["parity"].startswith("whole CSR bit-exact vs CPU oracle")
    r = out["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    copy = [v for k, v in r["kernels"].items() if k.startswith("k_copy_list")][0]
    assert copy["alg_bytes"] and copy["alg_bytes"] > 0 and copy["GBps"] > 0
    assert any(v["alg_bytes"] == r["kernel_algorithmic_bytes"] for v in r["kernels"].values())
    assert r["kernel"].startswith("k_parse") and 0 < r["copy_pass_frac"] < 1  # the dominant kernel is ONE kernel, one launch per scan; the copy pass is reported beside it
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["value"] > 0
    # the side measurements on the same lists: the compressor on the device, the EFGraph scan -- each checked before it is timed
    ex = out["extras"]
    assert "error" not in ex["compress"] and "error" not in ex["efgraph_scan"]
    assert ex["compress"]["parity"].startswith("streams byte-equal") and ex["compress"]["ms"] > 0
    assert ex["efgraph_scan"]["parity"].startswith("rowptr and successors equal") and 0 < ex["efgraph_scan"]["frac_of_hbm_peak"] < 1
    assert ex["load"]["ok"] and ex["load"]["ms"] > 0


@pytest.mark.timeout(1200)
def test_random_access_bench_with_first(tmp_path):
    """SpeedTest's random leg with --first (SpeedTest.java:107-108): the first successor of every list, checked against the oracle inside bench.py."""
    out = _bench(["--mode", "random", "--first", "--nodes", "300000", "--arcs", "6000000", "--queries", "200000", "--steps", "2", "--warmup", "1"], tmp_path)
    assert out["unit"] == "lists/s" and out["config"]["first"] is True and "first successor of" in out["config"]["parity"]
    assert 0 < out["roofline"]["frac"] < 1 and out["roofline"]["algorithmic_bytes_per_step"] > 4 * 200000
