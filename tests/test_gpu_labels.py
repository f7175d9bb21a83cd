"""GPU decode of arc labels against the CPU oracle and the labels that were stored:
BitStreamArcLabelledImmutableGraph with GammaCodedIntLabel / FixedWidthIntLabel, labels in the CSR order of the scan."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,width,n,m", [("gamma", 0, 20000, 400000), ("fixed", 13, 20000, 400000), ("fixed", 32, 500, 4000),
                                             ("fixed", 0, 500, 4000), ("gamma", 0, 300000, 9000000)])
def test_labels_match_oracle(tmp_path, kind, width, n, m):
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import ArcLabelledBVGraph
    from oracle import oracle as O
    rowptr, succ = T.generate(n, m, seed=31 + width, p_copy=0.5)
    T.store(str(tmp_path / "g"), rowptr, succ, window=7, max_ref_count=3, min_interval=4)
    rng = np.random.Generator(np.random.PCG64(5))
    if kind == "gamma":
        labels = (rng.pareto(0.8, size=m) * 2).astype(np.int64).clip(0, 2**31 - 2).astype(np.int32)  # mostly short codes, some > 2^16
    else:
        labels = (rng.integers(0, 2**width, size=m, dtype=np.int64) if width else np.zeros(m, dtype=np.int64)).astype(np.uint32).view(np.int32)
    lbase = str(tmp_path / "lab")
    T.store_labels(lbase, "g", rowptr, labels, kind=kind, width=width)
    g = ArcLabelledBVGraph.load(lbase)
    assert (g.info.kind, g.info.nodes) == (1 if kind == "gamma" else 2, n)
    rp, sc, lb = g.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ) and np.array_equal(lb, labels)
    d = np.diff(rowptr).astype(np.int32)
    for lo, hi in [(0, 1), (n // 3, min(n, n // 3 + 777)), (n - 5, n), (7, 7)]:
        rp, sc, lb = g.decode_range(lo, hi)
        assert np.array_equal(lb, O.labels_decode(lbase, n, d[lo:hi], lo, hi))
        assert np.array_equal(lb, labels[rowptr[lo]:rowptr[hi]])
    g.close()


def test_labels_wrong_arc_count_is_an_error(tmp_path):
    import ctypes as C
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import ArcLabelledBVGraph, lib
    rowptr, succ = T.generate(2000, 30000, seed=2, p_copy=0.5)
    T.store(str(tmp_path / "g"), rowptr, succ, window=7, max_ref_count=3, min_interval=4)
    T.store_labels(str(tmp_path / "lab"), "g", rowptr, np.arange(succ.size, dtype=np.int32) % 100, kind="gamma")
    g = ArcLabelledBVGraph.load(str(tmp_path / "lab"))
    out = np.empty(succ.size + 10, dtype=np.int32)
    assert lib().bvg_labels_decode_range(g._h, 0, 2000, succ.size + 3, out.ctypes.data, 0) != 0  # more labels asked than stored
    assert lib().bvg_labels_decode_range(g._h, 0, 2000, succ.size - 3, out.ctypes.data, 0) != 0
    assert lib().bvg_labels_decode_range(g._h, 0, 2001, succ.size, out.ctypes.data, 0) != 0      # node range out of bounds
    assert lib().bvg_labels_decode_range(g._h, 0, 2000, succ.size, out.ctypes.data, 0) == 0
    g.close()


@pytest.mark.parametrize("width,n,m", [(9, 20000, 400000), (32, 500, 4000), (0, 500, 4000), (1, 100000, 2000000)])
def test_label_lists_match_oracle(tmp_path, width, n, m):
    """FixedWidthIntListLabel: a list of ints per arc, decoded as a CSR over the arcs (bvg_labels_decode_lists)."""
    import ctypes as C
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import ArcLabelledBVGraph, lib
    from oracle import oracle as O
    from test_labels_cpu import _lists
    rowptr, succ = T.generate(n, m, seed=77 + width, p_copy=0.5)
    T.store(str(tmp_path / "g"), rowptr, succ, window=7, max_ref_count=3, min_interval=4)
    listptr, values = _lists(np.random.Generator(np.random.PCG64(width)), m, width)
    lbase = str(tmp_path / "lab")
    T.store_label_lists(lbase, "g", rowptr, listptr, values, width)
    g = ArcLabelledBVGraph.load(lbase)
    assert (g.info.kind, g.info.width, g.info.nodes) == (3, width, n)
    rp, sc, lp, vals = g.decode_label_lists()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ) and np.array_equal(lp, listptr) and np.array_equal(vals, values)
    d = np.diff(rowptr).astype(np.int32)
    for lo, hi in [(0, 1), (n // 3, min(n, n // 3 + 777)), (n - 5, n), (7, 7)]:
        rp, sc, lp, vals = g.decode_label_lists(lo, hi)
        olp, ovals = O.label_lists_decode(lbase, n, d[lo:hi], lo, hi)
        assert np.array_equal(lp, olp) and np.array_equal(vals, ovals)
    # errors: the one-int-per-arc entry point refuses this class; a wrong arc count is a format error; a short buffer reports the need
    out = np.empty(m + 8, dtype=np.int32)
    assert lib().bvg_labels_decode_range(g._h, 0, n, m, out.ctypes.data, 0) == -3
    lpb = np.empty(m + 8, dtype=np.int64)
    nv = C.c_uint64(0)
    assert lib().bvg_labels_decode_lists(g._h, 0, n, m + 1, lpb.ctypes.data, None, 0, C.byref(nv), 0) == -7
    assert lib().bvg_labels_decode_lists(g._h, 0, n, m, lpb.ctypes.data, None, 0, C.byref(nv), 0) == (-8 if values.size else 0)
    assert nv.value == values.size
    g.close()


@pytest.mark.parametrize("value", [0, 5, 1000])
def test_equal_labels_do_not_resynchronise(tmp_path, value):
    """Every arc with the SAME gamma-coded label: 00110 00110 ... is a stream on which a chain of codes that starts one bit late never meets the true one, so the device decoder's
    rounds advance one chunk each and it hands the stretch to the host walk (bvh::decode_gammas) -- which must give the labels, not an error."""
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import ArcLabelledBVGraph
    rowptr, succ = T.generate(60000, 2000000, seed=9, p_copy=0.5)
    T.store(str(tmp_path / "g"), rowptr, succ, window=7, max_ref_count=3, min_interval=4)
    lab = np.full(succ.size, value, dtype=np.int32)
    T.store_labels(str(tmp_path / "lab"), "g", rowptr, lab, kind="gamma")
    g = ArcLabelledBVGraph.load(str(tmp_path / "lab"))
    rp, sc, got = g.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ) and np.array_equal(got[:succ.size], lab)
    lo, hi = 20000, 41000
    rp, sc, got = g.decode_range(lo, hi)
    assert np.array_equal(got[:rowptr[hi] - rowptr[lo]], lab[rowptr[lo]:rowptr[hi]])
    g.close()
