"""GPU parity tests of the random-access path (BVGraph.successors(x), BVG:897-904 / :1032-1133 with window == null)."""
import numpy as np
import pytest

from conftest import CNR, make_graph

pytestmark = pytest.mark.gpu


# the two random-access strategies of the library: per-query chains of slots / a masked scan of the graph plus a gather
BATCH_MODES = {"slots": "0", "dense": "1000000000"}


@pytest.fixture(scope="module", params=list(BATCH_MODES))
def cnr_gpu(request):
    import os
    from webgraph_amd.bvgraph import BVGraph
    os.environ["BVGPU_BATCH_DENSE"] = BATCH_MODES[request.param]  # read when the handle is made
    try:
        g = BVGraph.load(CNR)
    finally:
        del os.environ["BVGPU_BATCH_DENSE"]
    yield g
    g.close()


def test_every_node_random_access(cnr_gpu, cnr_oracle):
    """BVGraphTest.testLarge second half: outdegree(i) / successors(i) for every node, incl. the -1 terminator."""
    _, rowptr, succ = cnr_oracle
    n = cnr_gpu.numNodes()
    rp, sc = cnr_gpu.successors_batch(np.arange(n, dtype=np.int32))
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    it = cnr_gpu.successors(7)
    assert [it.nextInt() for _ in range(7)] == [6, 18, 218, 285, 296, -1, -1]
    assert cnr_gpu.successors(5).nextInt() == -1  # empty list


def test_random_batch_with_repeats(cnr_gpu, cnr_oracle):
    og, rowptr, succ = cnr_oracle
    rng = np.random.default_rng(0x5EED)
    q = rng.integers(0, cnr_gpu.numNodes(), 50000).astype(np.int32)
    q[:5] = [46918, 46918, 112690, 0, 325556]  # max outdegree twice, longest record, first, last
    rp, sc = cnr_gpu.successors_batch(q)
    orp, osc = og.successors_batch(q)
    assert np.array_equal(rp, orp) and np.array_equal(sc, osc)


def test_empty_batch_and_errors(cnr_gpu):
    rp, sc = cnr_gpu.successors_batch(np.empty(0, dtype=np.int32))
    assert list(rp) == [0] and sc.size == 0
    with pytest.raises(ValueError):
        cnr_gpu.successors_batch(np.array([1, 325557], dtype=np.int32))
    with pytest.raises(ValueError):
        cnr_gpu.successors_batch(np.array([-1], dtype=np.int32))
    with pytest.raises(ValueError):
        cnr_gpu.successorArray(325557)


@pytest.mark.parametrize("mode", list(BATCH_MODES))
def test_deep_chains_random_access(tmp_path_factory, monkeypatch, mode):
    from webgraph_amd.bvgraph import BVGraph
    from oracle import oracle as O
    monkeypatch.setenv("BVGPU_BATCH_DENSE", BATCH_MODES[mode])
    base, rowptr, succ = make_graph(tmp_path_factory, "deepra", 20000, 300000, 31, 0.95, window=7, max_ref_count=50, min_interval=2)
    g = BVGraph.load(base)
    q = np.random.default_rng(3).integers(0, 20000, 5000).astype(np.int32)
    rp, sc = g.successors_batch(q)
    for j in range(0, len(q), 7):
        assert np.array_equal(sc[rp[j]:rp[j + 1]], succ[rowptr[q[j]]:rowptr[q[j] + 1]])
    og = O.OracleGraph.load(base)
    orp, osc = og.successors_batch(q)
    assert np.array_equal(rp, orp) and np.array_equal(sc, osc)
    g.close()


def test_cpp_host_mirror_runs(tmp_path):
    """webgraph_amd/host/bvgraph.hpp driven like WebGraphTestCase.assertGraph, on the GPU."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "host_mirror_test")
    pkg = os.path.join(ROOT, "webgraph_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"),
                           "-L" + pkg, "-lbvgpu", "-Wl,-rpath," + pkg, "-L/opt/rocm/lib", "-lamdhip64"])
    p = subprocess.run([exe, CNR, "1711395807", "3216152", str(tmp_path / "out"), "3"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "host mirror ok" in p.stdout


@pytest.mark.parametrize("shift", [1, 2, 3])
def test_device_buffers_need_not_be_16_byte_aligned(cnr_gpu, cnr_oracle, shift):
    """Device-pointer form of the batch call with an output buffer that starts `shift` ints past an aligned address."""
    import ctypes as C
    import torch
    from webgraph_amd import bvgraph as B
    og, rowptr, succ = cnr_oracle
    q = np.random.default_rng(shift).integers(0, cnr_gpu.numNodes(), 40000).astype(np.int32)
    q[0] = 46918
    orp, osc = og.successors_batch(q)
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(q).to(dev)
    d_rp = torch.empty(q.size + 1, dtype=torch.int64, device=dev)
    d_sc = torch.full((osc.size + 8,), -7, dtype=torch.int32, device=dev)
    arcs = C.c_uint64(0)
    rc = B.lib().bvg_successors_batch(cnr_gpu._h, d_q.data_ptr(), q.size, d_rp.data_ptr(), d_sc.data_ptr() + 4 * shift, osc.size, C.byref(arcs), B.BVG_OUT_DEVICE)
    assert rc == 0 and arcs.value == osc.size
    out = d_sc.cpu().numpy()
    assert np.array_equal(d_rp.cpu().numpy(), orp) and np.array_equal(out[shift:shift + osc.size], osc)
    assert (out[:shift] == -7).all() and (out[shift + osc.size:] == -7).all()  # nothing written outside the rows


def test_few_queries_for_the_longest_rows(cnr_gpu, cnr_oracle):
    """A batch of few ids whose rows hold a good part of the arcs (the masked-scan strategy is chosen by arcs then,
    and the gather shares the long rows among several blocks), with repeats and in descending order."""
    og, rowptr, succ = cnr_oracle
    d = np.diff(rowptr)
    q = np.argsort(d)[::-1][:3000].astype(np.int32)
    q = np.concatenate([q, q[:17], q[::-1][:5]]).astype(np.int32)
    rp, sc = cnr_gpu.successors_batch(q)
    orp, osc = og.successors_batch(q)
    assert np.array_equal(rp, orp) and np.array_equal(sc, osc)


def test_successor_buffer_too_small_is_reported_not_overrun(cnr_gpu, cnr_oracle):
    """BVG_ECAP (include/bvgpu.h): a caller buffer smaller than the result is an error, arcs_out still reports the
    need, and nothing is written past the capacity given -- for the scan and for both batch strategies."""
    import ctypes as C
    import torch
    from webgraph_amd import bvgraph as B
    og, rowptr, succ = cnr_oracle
    dev = torch.device("cuda", 0)
    lib = B.lib()
    lo, hi = 1000, 200000
    need = int(rowptr[hi] - rowptr[lo])
    cap = need - 5
    d_rp = torch.empty(hi - lo + 1, dtype=torch.int64, device=dev)
    d_sc = torch.full((need + 64,), -7, dtype=torch.int32, device=dev)
    arcs = C.c_uint64(0)
    rc = lib.bvg_decode_range(cnr_gpu._h, lo, hi, d_rp.data_ptr(), d_sc.data_ptr(), cap, C.byref(arcs), B.BVG_OUT_DEVICE)
    assert rc == B.BVG_ECAP and arcs.value == need
    assert (d_sc[cap:].cpu().numpy() == -7).all()
    rc = lib.bvg_decode_range(cnr_gpu._h, lo, hi, d_rp.data_ptr(), d_sc.data_ptr(), need, C.byref(arcs), B.BVG_OUT_DEVICE)
    assert rc == 0 and np.array_equal(d_sc[:need].cpu().numpy(), succ[rowptr[lo]:rowptr[hi]])  # the handle is usable afterwards
    q = np.random.default_rng(11).integers(0, cnr_gpu.numNodes(), 30000).astype(np.int32)
    orp, osc = og.successors_batch(q)
    d_q = torch.from_numpy(q).to(dev)
    d_rp = torch.empty(q.size + 1, dtype=torch.int64, device=dev)
    d_sc = torch.full((osc.size + 64,), -7, dtype=torch.int32, device=dev)
    rc = lib.bvg_successors_batch(cnr_gpu._h, d_q.data_ptr(), q.size, d_rp.data_ptr(), d_sc.data_ptr(), osc.size - 1, C.byref(arcs), B.BVG_OUT_DEVICE)
    assert rc == B.BVG_ECAP and arcs.value == osc.size
    assert (d_sc.cpu().numpy() == -7).all()  # the batch call checks before it decodes
    rc = lib.bvg_successors_batch(cnr_gpu._h, d_q.data_ptr(), q.size, d_rp.data_ptr(), d_sc.data_ptr(), osc.size, C.byref(arcs), B.BVG_OUT_DEVICE)
    assert rc == 0 and np.array_equal(d_sc[:osc.size].cpu().numpy(), osc)
