"""EFGraph decoded on the GPU (bv_ef.hip behind the same C ABI) against the CPU oracle's reader and
against what was stored: scans, sub-ranges, random access, the consumers, both byte orders, an upper bound above n, shards."""
import numpy as np
import pytest

from test_efgraph_cpu import KAT_ROWS, _csr

pytestmark = pytest.mark.gpu


def _store(tmp_path, n, m, seed, **kw):
    from webgraph_amd import tools as T
    rowptr, succ = T.generate(n, m, seed=seed, p_copy=0.5)
    base = str(tmp_path / "ef")
    T.store_ef(base, rowptr, succ, **kw)
    return base, rowptr, succ


def test_hand_made_record(tmp_path):
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    rowptr, succ = _csr(KAT_ROWS)
    T.store_ef(str(tmp_path / "kat"), rowptr, succ)
    g = B.EFGraph.load(str(tmp_path / "kat"))
    assert (g.numNodes(), g.numArcs(), g.info.format, g.upperBound()) == (4, 7, B.BVG_FORMAT_EF, 4)
    rp, sc = g.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    assert [g.successorArray(x).tolist() for x in range(4)] == KAT_ROWS and [g.outdegree(x) for x in range(4)] == [2, 0, 4, 1]
    g.close()


@pytest.mark.parametrize("n,m,lq,big,ub", [(200000, 4000000, 8, False, None), (50000, 1000000, 2, True, None), (3000, 100000, 0, False, 5000)])
def test_scan_and_random_access_match_the_oracle(tmp_path, n, m, lq, big, ub):
    from webgraph_amd import bvgraph as B
    from oracle import oracle as O
    base, rowptr, succ = _store(tmp_path, n, m, 7 + lq, upper_bound=ub, log2_quantum=lq, big_endian=big)
    og = O.OracleEFGraph.load(base)
    g = B.BVGraph.load(base)  # the graphclass property decides
    assert g.info.format == B.BVG_FORMAT_EF and g.numArcs() == m
    rp, sc = g.decode_range()
    orp, osc, _ = og.scan()
    assert np.array_equal(rp, orp) and np.array_equal(sc, osc) and np.array_equal(sc, succ)
    for lo, hi in [(0, 1), (n // 3, n // 3 + 999), (n - 7, n), (5, 5)]:
        rp, sc = g.decode_range(lo, hi)
        orp, osc, _ = og.scan(lo, hi)
        assert np.array_equal(rp, orp) and np.array_equal(sc, osc)
    assert np.array_equal(g.outdegrees(0, n), np.diff(rowptr))
    q = np.random.Generator(np.random.PCG64(3)).integers(0, n, size=20000).astype(np.int32)
    brp, bsc = g.successors_batch(q)
    want = np.concatenate([succ[rowptr[x]:rowptr[x + 1]] for x in q]) if q.size else np.empty(0, np.int32)
    assert np.array_equal(np.diff(brp), np.diff(rowptr)[q]) and np.array_equal(bsc, want)
    with pytest.raises(ValueError):
        g.successors_batch(np.array([0, n], dtype=np.int32))
    # the consumers run on the decoded rows whatever the format: same hashCode() as the BVGraph of the same lists (itself checked
    # against the oracle's in test_gpu_scan), and equals() between the two
    from webgraph_amd import tools as T
    T.store(str(tmp_path / "bv"), rowptr, succ)
    h = B.BVGraph.load(str(tmp_path / "bv"))
    assert g.hashCode() == h.hashCode() and g.equals(h) and h.equals(g)
    h.close()
    g.close()


def test_long_lists_go_to_the_waves(tmp_path):
    """Lists of 256 successors or more are decoded by a wave each: dense (l = 0) and sparse ones, one of 200 000."""
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    rng = np.random.Generator(np.random.PCG64(9))
    n = 300000
    rows = [[] for _ in range(40)]
    rows[3] = sorted(set(rng.integers(0, n, size=200000).tolist()))
    rows[4] = list(range(1000, 260000))                    # denser than one value per slot: l = 0
    rows[7] = sorted(set(rng.integers(0, n, size=256).tolist()))
    rows[8] = sorted(set(rng.integers(0, n, size=255).tolist()))
    rows[20] = list(range(0, n, 2))
    rows[39] = [n - 1]
    rows += [[] for _ in range(n - len(rows))]
    rowptr, succ = _csr(rows)
    T.store_ef(str(tmp_path / "long"), rowptr, succ, log2_quantum=4)
    g = B.EFGraph.load(str(tmp_path / "long"))
    rp, sc = g.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    brp, bsc = g.successors_batch(np.array([20, 3, 39, 4, 0], dtype=np.int32))
    assert np.array_equal(bsc, np.array(rows[20] + rows[3] + rows[39] + rows[4], dtype=np.int32))
    # the same giant lists asked for many times: more rounds than the scratch for them holds
    q = np.array([3, 4, 20] * 400, dtype=np.int32)
    brp, bsc = g.successors_batch(q)
    one = np.array(rows[3] + rows[4] + rows[20], dtype=np.int32)
    assert bsc.size == 400 * one.size and np.array_equal(bsc.reshape(400, -1), np.tile(one, (400, 1)))
    g.close()


def test_recompress_an_efgraph_as_bvgraph(tmp_path):
    """EFGraph -> BVGraph without leaving the device (bvg_recompress), equal to the CPU writer's files for the same lists."""
    import filecmp
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    base, rowptr, succ = _store(tmp_path, 60000, 1200000, 11)
    g = B.EFGraph.load(base)
    g.store(str(tmp_path / "bv"))
    g.close()
    T.store(str(tmp_path / "cpu"), rowptr, succ, threads=1)
    assert filecmp.cmp(str(tmp_path / "bv.graph"), str(tmp_path / "cpu.graph"), shallow=False)
    h = B.BVGraph.load(str(tmp_path / "bv"))
    assert h.info.format == B.BVG_FORMAT_BV
    rp, sc = h.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    h.close()


def test_shards_and_errors(tmp_path):
    from webgraph_amd import bvgraph as B
    base, rowptr, succ = _store(tmp_path, 100000, 2000000, 13)
    parts, chunks = 4, []
    for k in range(parts):
        s = B.BVGraph.load_shard(base, k, parts)
        lo, hi = s.info.shard_from, s.info.shard_to
        rp, sc = s.decode_range(lo, hi)
        assert np.array_equal(rp, rowptr[lo:hi + 1] - rowptr[lo])
        chunks.append(sc)
        s.close()
    assert np.array_equal(np.concatenate(chunks), succ)
    with pytest.raises(IOError):
        B.EFGraph.load(str(tmp_path / "missing"))
    # a BVGraph is not an EFGraph
    from conftest import CNR
    with pytest.raises(IOError):
        B.EFGraph.load(CNR)
    # truncated stream: an error, not a crash
    raw = open(base + ".graph", "rb").read()
    open(base + ".graph", "wb").write(raw[:len(raw) // 2])
    try:
        g = B.BVGraph.load(base)
    except (IOError, B.BvgError):
        return
    with pytest.raises((B.BvgError, ValueError, IOError)):
        g.decode_range()
    g.close()


@pytest.mark.parametrize("n,m,lq,big,ub", [(200000, 4000000, 8, False, None), (50000, 1000000, 2, True, None), (3000, 100000, 0, False, 5000), (4, 0, 8, False, None)])
def test_device_writer_reproduces_the_cpu_writer(tmp_path, n, m, lq, big, ub):
    """EFGraph.store on the device (bvg_store_ef): .graph, .offsets and .properties byte-equal to the CPU writer's, which is itself
    checked against the hand-made record; small quanta give every long list forward pointers."""
    import filecmp
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    if m:
        rowptr, succ = T.generate(n, m, seed=5 + lq, p_copy=0.5)
    else:
        rowptr, succ = _csr(KAT_ROWS)
    cpu, gpu = str(tmp_path / "cpu"), str(tmp_path / "gpu")
    T.store_ef(cpu, rowptr, succ, upper_bound=ub, log2_quantum=lq, big_endian=big)
    B.store_ef(rowptr, succ, gpu, upperBound=ub, log2Quantum=lq, bigEndian=big)
    for ext in (".graph", ".offsets", ".properties"):
        assert filecmp.cmp(cpu + ext, gpu + ext, shallow=False), ext
    if m and lq <= 2:  # the forward pointers the device wrote, read the way the reference's skipTo reads them (EFGraph.java:1147-1215; oracle efo_skip_to)
        from oracle import oracle as O
        og = O.OracleEFGraph.load(gpu)
        rng = np.random.default_rng(11)
        nodes = rng.choice(np.nonzero(np.diff(rowptr) >= 8)[0], 4000).astype(np.int32)
        bounds = rng.integers(0, ub or n, nodes.size).astype(np.int32)
        got, used = og.skip_to(nodes, bounds)
        for x, b, v in zip(nodes, bounds, got):
            row = succ[rowptr[x]:rowptr[x + 1]]
            k = np.searchsorted(row, b)
            assert v == (row[k] if k < row.size else -1)
        assert used.sum() > 500
    with pytest.raises(ValueError):
        B.store_ef(np.array([0, 2], dtype=np.int64), np.array([1, 1], dtype=np.int32), gpu)       # not strictly increasing
    with pytest.raises(ValueError):
        B.store_ef(np.array([0, 1], dtype=np.int64), np.array([7], dtype=np.int32), gpu)          # not below the bound


def test_bvgraph_to_efgraph_on_the_device(tmp_path, cnr_oracle):
    """bvg_recompress_ef: cnr-2000 (a BVGraph) re-encoded as an EFGraph without leaving HBM; the result equals the CPU writer's files
    for the same lists and decodes back to them."""
    import filecmp
    from conftest import CNR
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    _, rowptr, succ = cnr_oracle
    g = B.BVGraph.load(CNR)
    g.store_ef(str(tmp_path / "ef"), log2Quantum=4)
    g.close()
    T.store_ef(str(tmp_path / "cpu"), rowptr, succ, log2_quantum=4)
    for ext in (".graph", ".offsets", ".properties"):
        assert filecmp.cmp(str(tmp_path / "ef") + ext, str(tmp_path / "cpu") + ext, shallow=False), ext
    h = B.EFGraph.load(str(tmp_path / "ef"))
    rp, sc = h.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    h.close()


def test_checksum_without_writing_the_lists(tmp_path, monkeypatch):
    """bvg_scan_checksum on an EFGraph folds hashCode() inside the decode kernels (no successor is written); it must equal the
    decode-then-fold path and the BVGraph's hashCode of the same lists; ranges compose; long and giant lists included."""
    from webgraph_amd import bvgraph as B
    from webgraph_amd import tools as T
    rng = np.random.Generator(np.random.PCG64(21))
    rowptr, succ = T.generate(120000, 2400000, seed=31, p_copy=0.5)
    n = rowptr.size - 1
    rows = [succ[rowptr[x]:rowptr[x + 1]].tolist() for x in range(n)]
    rows[5] = sorted(set(rng.integers(0, n, size=40000).tolist()))      # a giant list, a long one, and neighbours without successors
    rows[6] = []
    rows[7] = list(range(100, 1100))
    rowptr, succ = _csr(rows)
    T.store_ef(str(tmp_path / "ef"), rowptr, succ)
    T.store(str(tmp_path / "bv"), rowptr, succ)
    g, h = B.EFGraph.load(str(tmp_path / "ef")), B.BVGraph.load(str(tmp_path / "bv"))
    want = h.hashCode()
    assert g.hashCode() == want
    a, arcs_a = g.scan_checksum(0, 777, -1)
    b, arcs_b = g.scan_checksum(777, n, a)
    assert b == want and arcs_a + arcs_b == succ.size
    g.set_option("ef_hash_materialise", 1)
    assert g.hashCode() == want
    g.close(); h.close()


def test_corrupt_streams_fail_cleanly(tmp_path):
    """Bit flips anywhere in an EFGraph's stream: every call comes back (an error, or lists that differ), nothing is written past the
    caller's buffers (a guard band behind them stays intact)."""
    import ctypes as C
    from webgraph_amd import bvgraph as B
    base, rowptr, succ = _store(tmp_path, 20000, 400000, 17, log2_quantum=3)
    raw = bytearray(open(base + ".graph", "rb").read())
    rng = np.random.Generator(np.random.PCG64(5))
    n, m = rowptr.size - 1, succ.size
    guard = 4096
    outcomes = set()
    for trial in range(40):
        bad = bytearray(raw)
        for _ in range(int(rng.integers(1, 40))):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        open(base + ".graph", "wb").write(bytes(bad))
        g = B.BVGraph.load(base)
        rp = np.full(n + 1 + guard, -7, dtype=np.int64)
        sc = np.full(2 * m + guard, -7, dtype=np.int32)
        arcs = C.c_uint64(0)
        rc = B.lib().bvg_decode_range(g._h, 0, n, rp.ctypes.data, sc.ctypes.data, 2 * m, C.byref(arcs), B.BVG_OUT_HOST)
        outcomes.add(rc)
        assert rc in (0, B.BVG_EFORMAT, B.BVG_ECAP, B.BVG_ENOMEM), rc
        assert np.all(rp[n + 1:] == -7) and np.all(sc[2 * m:] == -7)
        if rc == 0:
            assert arcs.value <= 2 * m
        q = rng.integers(0, n, size=500).astype(np.int32)
        brp = np.empty(501, dtype=np.int64)
        rc = B.lib().bvg_successors_batch(g._h, q.ctypes.data, 500, brp.ctypes.data, sc.ctypes.data, 2 * m, C.byref(arcs), B.BVG_OUT_HOST)
        assert rc in (0, B.BVG_EFORMAT, B.BVG_ECAP, B.BVG_ENOMEM), rc
        assert np.all(sc[2 * m:] == -7)
        hh = C.c_int32(-1)
        rc = B.lib().bvg_scan_checksum(g._h, 0, n, C.byref(hh), C.byref(arcs))
        assert rc in (0, B.BVG_EFORMAT, B.BVG_ENOMEM), rc
        g.close()
    open(base + ".graph", "wb").write(bytes(raw))
    assert len(outcomes) >= 1


def test_cache_as_efgraph(cnr_oracle):
    """bvg_cache_as_efgraph: the handle re-encodes its lists in HBM and answers from there -- the same lists, the same hashCode; a clone
    made before keeps the BVGraph image."""
    from conftest import CNR
    from webgraph_amd import bvgraph as B
    _, rowptr, succ = cnr_oracle
    g = B.BVGraph.load(CNR)
    c = g.copy()
    h0 = g.hashCode()
    g.cache_as_efgraph()
    assert g.info.format == B.BVG_FORMAT_EF and c.info.format == B.BVG_FORMAT_BV and g.numArcs() == succ.size
    rp, sc = g.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ) and g.hashCode() == h0 == 1711395807
    q = np.array([7, 100000, 46918, 325556, 7], dtype=np.int32)
    brp, bsc = g.successors_batch(q)
    crp, csc = c.successors_batch(q)
    assert np.array_equal(brp, crp) and np.array_equal(bsc, csc)
    g.cache_as_efgraph()  # a second call is a no-op
    rp, sc = c.decode_range(1000, 2000)
    assert np.array_equal(sc, succ[rowptr[1000]:rowptr[2000]])
    g.close(); c.close()
