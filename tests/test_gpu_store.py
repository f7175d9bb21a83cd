"""BVGraph.store on the GPU (include/bvgpu.h bvg_compress / bvg_store) against the reference's own
cnr-2000 files and against the CPU writer, byte for byte; then the decoder reads back what the compressor wrote."""
import filecmp
import os

import numpy as np
import pytest

from conftest import CNR
from test_encode_model_cpu import DELTA, FLAG, GAMMA, GOLOMB, NIBBLE, UNARY

pytestmark = pytest.mark.gpu


def test_store_reproduces_reference_bytes(tmp_path, cnr_oracle):
    """The pin: cnr-2000 recompressed with its own parameters gives the reference-produced .graph / .offsets back."""
    from webgraph_amd import bvgraph as B
    _, rowptr, succ = cnr_oracle
    base = str(tmp_path / "cnr")
    st = B.store(rowptr, succ, base, windowSize=7, maxRefCount=3, minIntervalLength=3, zetaK=3)
    assert filecmp.cmp(base + ".graph", CNR + ".graph", shallow=False)
    assert filecmp.cmp(base + ".offsets", CNR + ".offsets", shallow=False)
    bits = st["bits_outdegrees"] + st["bits_references"] + st["bits_blocks"] + st["bits_intervals"] + st["bits_residuals"]
    assert bits == st["written_bits"] and os.path.getsize(base + ".graph") == (bits + 7) // 8
    assert (st["copied_arcs"], st["intervalised_arcs"], st["residual_arcs"], st["max_ref_chain"]) == (2130833, 361894, 723425, 3)
    # the properties load, and the graph decodes to what went in
    g = B.BVGraph.load(base)
    rp, sc = g.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    g.close()


def test_recompress_on_the_device(tmp_path, cnr_oracle):
    """decode -> compress without leaving HBM: the CSR bvg_decode_range wrote is the compressor's input."""
    import torch
    from webgraph_amd import bvgraph as B
    g = B.BVGraph.load(CNR)
    n, m = g.numNodes(), g.numArcs()
    rp = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    sc = torch.empty(m, dtype=torch.int32, device="cuda")
    g.decode_range_device(0, n, rp.data_ptr(), sc.data_ptr(), m)
    torch.cuda.synchronize()
    graph, offs, bitoff, st = B.compress(rp, sc, windowSize=7, maxRefCount=3, minIntervalLength=3, zetaK=3)
    assert graph == open(CNR + ".graph", "rb").read() and offs == open(CNR + ".offsets", "rb").read()
    from oracle import oracle as O
    assert np.array_equal(bitoff, O.decode_offsets(open(CNR + ".offsets", "rb").read(), n))
    g.close()


@pytest.mark.parametrize("W,R,I,K,flags,threads", [
    (7, 3, 4, 3, 0, 1), (7, 3, 4, 3, 0, 3), (0, 0, 0, 3, 0, 1), (1, 1, 2, 3, 0, 1), (3, 100, 1, 2, 0, 2), (7, 3, 0, 3, 0, 1), (16, 2, 3, 5, 0, 1),
    (7, 3, 4, 3, (DELTA << FLAG["outd"]) | (DELTA << FLAG["blk"]) | (DELTA << FLAG["res"]) | (GAMMA << FLAG["ref"]) | (DELTA << FLAG["bc"]) | (DELTA << FLAG["off"]), 1),
    (7, 3, 4, 3, (UNARY << FLAG["blk"]) | (NIBBLE << FLAG["res"]) | (DELTA << FLAG["ref"]) | (UNARY << FLAG["bc"]), 1),
    (7, 3, 4, 5, (GOLOMB << FLAG["res"]), 1)])
def test_store_matches_cpu_writer(tmp_path, W, R, I, K, flags, threads):
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    rowptr, succ = T.generate(60000, 1200000, seed=23 + W, p_copy=0.7)
    cpu, gpu = str(tmp_path / "cpu"), str(tmp_path / "gpu")
    st_cpu = T.store(cpu, rowptr, succ, window=W, max_ref_count=R, min_interval=I, zeta_k=K, flags=flags, threads=threads)
    st_gpu = B.store(rowptr, succ, gpu, windowSize=W, maxRefCount=R, minIntervalLength=I, zetaK=K, flags=flags, numberOfThreads=threads)
    for ext in (".graph", ".offsets", ".properties"):
        assert filecmp.cmp(cpu + ext, gpu + ext, shallow=False), ext
    for k in st_cpu:
        assert st_cpu[k] == st_gpu[k], k
    # the gap histograms the call hands back are the ones .properties carries (BVGraph.java:2592-2632; the CPU writer's file is the same, byte for byte)
    props = dict(l.strip().split("=", 1) for l in open(gpu + ".properties") if "=" in l and not l.startswith("#"))
    for name in ("successor", "residual"):
        bins = st_gpu[name + "_gap_bins"]
        while bins and bins[-1] == 0:
            bins = bins[:-1]
        assert props[name + "expstats"] == ",".join(str(b) for b in bins), name
    lists_first = succ[rowptr[:-1][np.diff(rowptr) > 0]].astype(np.int64) - np.nonzero(np.diff(rowptr) > 0)[0]
    inner = np.diff(succ.astype(np.int64))
    inner = np.delete(inner, rowptr[1:-1][(rowptr[1:-1] > 0) & (rowptr[1:-1] < len(succ))] - 1)
    nat = np.where(lists_first >= 0, 2 * lists_first, -2 * lists_first - 1)
    want = np.bincount(np.floor(np.log2(np.concatenate([inner[inner > 0], nat[nat > 0]]).astype(np.float64))).astype(np.int64), minlength=32)
    assert [int(v) for v in want] == st_gpu["successor_gap_bins"]


def test_store_c2_shape_round_trip(tmp_path):
    """A 1 M-node slice of the C2 recipe (giant rows included): compress on the GPU, decode on the GPU, compare with the input;
    and the CPU writer produces the same bytes."""
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    rowptr, succ = T.generate(1000000, 20000000, seed=0x5EEDB5E70001, p_copy=0.5)
    gpu, cpu = str(tmp_path / "gpu"), str(tmp_path / "cpu")
    B.store(rowptr, succ, gpu)
    T.store(cpu, rowptr, succ, threads=1)
    assert filecmp.cmp(cpu + ".graph", gpu + ".graph", shallow=False) and filecmp.cmp(cpu + ".offsets", gpu + ".offsets", shallow=False)
    g = B.BVGraph.load(gpu)
    rp, sc = g.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    g.close()


def test_store_errors(tmp_path):
    from webgraph_amd import bvgraph as B
    rowptr = np.array([0, 3, 5], dtype=np.int64)
    with pytest.raises(ValueError):  # a row that is not strictly increasing
        B.store(rowptr, np.array([1, 1, 2, 0, 4], dtype=np.int32), str(tmp_path / "x"))
    with pytest.raises(ValueError):  # a negative id; the id that the wave path pads its tiles with
        B.store(rowptr, np.array([-1, 1, 2, 0, 4], dtype=np.int32), str(tmp_path / "x"))
    with pytest.raises(ValueError):
        B.store(rowptr, np.array([0, 1, 2, 3, 2**31 - 1], dtype=np.int32), str(tmp_path / "x"))
    with pytest.raises(ValueError):
        B.store(rowptr, np.array([0, 1, 2, 3, 4], dtype=np.int32), str(tmp_path / "x"), windowSize=-1)
    with pytest.raises(NotImplementedError):  # windows above the device compressor's limit
        B.store(rowptr, np.array([0, 1, 2, 3, 4], dtype=np.int32), str(tmp_path / "x"), windowSize=64)
    with pytest.raises(NotImplementedError):  # a coding the reference's writer rejects too (BVGraph.java:1846)
        B.store(rowptr, np.array([0, 1, 2, 3, 4], dtype=np.int32), str(tmp_path / "x"), flags=NIBBLE << FLAG["outd"])
    import torch
    bad_rp = torch.tensor([0, 4, 2, 5], dtype=torch.int64, device="cuda")  # a CSR in device memory is checked too
    with pytest.raises(ValueError):
        B.store(bad_rp, torch.arange(5, dtype=torch.int32, device="cuda"), str(tmp_path / "x"))
    st = B.store(np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), str(tmp_path / "empty"))
    assert st["written_bits"] == 0 and open(str(tmp_path / "empty") + ".offsets", "rb").read() == b"\x80"
    # rows may start lower than the previous row ended: only the order inside a row matters
    st = B.store(rowptr, np.array([5, 6, 7, 0, 1], dtype=np.int32), str(tmp_path / "ok"))
    assert st["copied_arcs"] + st["intervalised_arcs"] + st["residual_arcs"] == 5


def _shaped_graph(rng, n=3000):
    """Rows built to stress the wave walk: long runs of consecutive ids (intervals longer than a tile of 64), rows that copy a
    long predecessor keeping 2 of every 3 / dropping stretches / adding a few, empty rows in between, one giant row."""
    rows = []
    for x in range(n):
        k = rng.integers(0, 12)
        prev = rows[x - int(rng.integers(1, 4))] if x >= 3 else []
        if k == 0 or x < 3:
            base = int(rng.integers(0, 200000))
            row = set(range(base, base + int(rng.integers(1, 700))))                      # one long interval
            row |= set(int(v) for v in rng.integers(0, 300000, size=int(rng.integers(0, 90))))
        elif k == 1:
            row = set()
        elif k <= 5 and len(prev):
            keep = rng.random(len(prev)) < rng.choice([0.3, 0.66, 0.95])
            row = set(np.asarray(prev)[keep].tolist())
            row |= set(int(v) for v in rng.integers(0, 300000, size=int(rng.integers(0, 40))))
        elif k <= 8 and len(prev):
            a, b = sorted(rng.integers(0, len(prev) + 1, size=2))
            row = set(prev[:a]) | set(prev[b:])                                         # a stretch dropped
            row |= set(range(int(prev[0]) + 5, int(prev[0]) + 5 + int(rng.integers(0, 150))))
        else:
            row = set(int(v) for v in rng.integers(0, 300000, size=int(rng.integers(1, 400))))
            start = int(rng.integers(0, 250000))
            for i in range(int(rng.integers(0, 5))):                                    # runs of 2..70 consecutive ids
                row |= set(range(start + 200 * i, start + 200 * i + int(rng.integers(2, 71))))
        rows.append(sorted(row))
    rows[n // 2] = sorted(set(int(v) for v in rng.integers(0, 300000, size=60000)) | set(range(100000, 109000)))
    rows[n // 2 + 1] = sorted(set(rows[n // 2][::2]) | set(range(150000, 150400)))
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    succ = np.array([v for r in rows for v in r], dtype=np.int32)
    return rowptr, succ


@pytest.mark.parametrize("seg", [None, (8, 100), (8, 37)])
@pytest.mark.parametrize("W,R,I", [(7, 3, 4), (7, 3, 2), (3, 2, 100), (7, 3, 0), (7, 8, 65), (1, 1, 3)])
def test_store_shapes_that_stress_the_wave_walk(tmp_path, monkeypatch, W, R, I, seg):
    """Pairs of 128 elements or more are priced and written by whole waves (bv_encode_wave.hpp): runs and blocks that span tiles,
    minIntervalLength below / at / above the tile size; and with every such pair cut into short segments that are priced independently and
    stitched (what the library does to pairs of 2^15 elements or more).  BVGPU_ENC_VERIFY makes the library price those pairs lane by lane too."""
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    monkeypatch.setenv("BVGPU_ENC_VERIFY", "1")
    if seg:  # every pair the waves take is cut into segments of ~100 / ~37 elements (the default cuts pairs of 2^15 elements into 8192s)
        monkeypatch.setenv("BVGPU_ENC_SEGBIN", str(seg[0]))
        monkeypatch.setenv("BVGPU_ENC_SEGELEMS", str(seg[1]))
    rowptr, succ = _shaped_graph(np.random.Generator(np.random.PCG64(100 + W + I)))
    cpu, gpu = str(tmp_path / "cpu"), str(tmp_path / "gpu")
    st_cpu = T.store(cpu, rowptr, succ, window=W, max_ref_count=R, min_interval=I, threads=1)
    st_gpu = B.store(rowptr, succ, gpu, windowSize=W, maxRefCount=R, minIntervalLength=I)
    for ext in (".graph", ".offsets", ".properties"):
        assert filecmp.cmp(cpu + ext, gpu + ext, shallow=False), ext
    assert all(st_cpu[k] == st_gpu[k] for k in st_cpu)


def test_store_chains_that_never_forget(tmp_path):
    """Identical rows: the chain lengths repeat with a period that does not divide the chunk size, so every guess at a chunk
    boundary is wrong and the selection has to carry the truth across the whole graph (spans doubling batch after batch)."""
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    n = 300000
    rowptr = np.arange(n + 1, dtype=np.int64) * 10
    succ = np.tile(np.arange(5, 105, 10, dtype=np.int32), n)
    cpu, gpu = str(tmp_path / "cpu"), str(tmp_path / "gpu")
    T.store(cpu, rowptr, succ, window=7, max_ref_count=3, min_interval=4, threads=1)
    st = B.store(rowptr, succ, gpu, windowSize=7, maxRefCount=3, minIntervalLength=4)
    assert filecmp.cmp(cpu + ".graph", gpu + ".graph", shallow=False) and filecmp.cmp(cpu + ".offsets", gpu + ".offsets", shallow=False)
    assert st["selection_rounds"] > 8  # more than one batch was needed


def test_recompress_from_a_handle(tmp_path, cnr_oracle):
    """BVGraph.store(graph, ...) with the graph itself on the GPU: same parameters -> the reference's bytes back; other
    parameters -> what the CPU writer makes of the same lists; a shard handle is refused."""
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    _, rowptr, succ = cnr_oracle
    g = B.BVGraph.load(CNR)
    same = str(tmp_path / "same")
    g.store(same, windowSize=7, maxRefCount=3, minIntervalLength=3, zetaK=3)
    assert filecmp.cmp(same + ".graph", CNR + ".graph", shallow=False) and filecmp.cmp(same + ".offsets", CNR + ".offsets", shallow=False)
    other, cpu = str(tmp_path / "other"), str(tmp_path / "cpu")
    g.store(other, windowSize=3, maxRefCount=10, minIntervalLength=2, zetaK=4)
    T.store(cpu, rowptr, succ, window=3, max_ref_count=10, min_interval=2, zeta_k=4, threads=1)
    for ext in (".graph", ".offsets", ".properties"):
        assert filecmp.cmp(cpu + ext, other + ext, shallow=False), ext
    g.close()
    h = B.BVGraph.load(other)
    rp, sc = h.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    h.close()
    s = B.BVGraph.load_shard(CNR, 1, 4)
    with pytest.raises(NotImplementedError):
        s.store(str(tmp_path / "shard"))
    s.close()
