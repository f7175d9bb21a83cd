"""The drop-in boundary beyond the two hot entry points: one-call host results (bvg_decode_range_view, BVG_OUT_HOST into
pageable and pinned buffers), the checksum scan (bvg_scan_checksum) and split / copied node iterators drained by
concurrent threads, each through a flyweight handle of its own (NodeIterator.copy, BVGraph.java:1253-1260, used by
ImmutableGraph.splitNodeIterators, ImmutableGraph.java:379-409, and by BVGraph.java:2471-2477)."""
import ctypes as C
import threading

import os

import numpy as np
import pytest

from conftest import CNR, make_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cnr_gpu():
    from webgraph_amd.bvgraph import BVGraph
    g = BVGraph.load(CNR)
    yield g
    g.close()


@pytest.mark.parametrize("lo,hi", [(0, 325557), (1000, 21000), (5, 6), (325557, 325557), (0, 0), (100000, 300000)])
def test_view_is_one_call_and_bit_exact(cnr_gpu, cnr_oracle, lo, hi):
    og, rp, sc = cnr_oracle
    rowptr, succ = cnr_gpu.decode_range_view(lo, hi)
    assert np.array_equal(rowptr, rp[lo:hi + 1] - rp[lo])
    assert np.array_equal(succ, sc[rp[lo]:rp[hi]])


@pytest.mark.parametrize("pinned", [False, True])
def test_host_buffers_pageable_and_pinned(tmp_path_factory, pinned):
    """BVG_OUT_HOST into the caller's memory: many chunks (the graph is decoded 4 M arcs at a time), each crossing PCIe while
    the next one is decoded; pageable destinations are filled from a pinned ring by host threads."""
    from webgraph_amd import bvgraph as B
    base, rowptr, succ = make_graph(tmp_path_factory, "hostbuf", 1_500_000, 30_000_000, seed=11, p_copy=0.6)
    g = B.BVGraph.load(base)
    n, m = g.numNodes(), succ.size
    rp = np.empty(n + 1, dtype=np.int64)
    if pinned:
        p = C.c_void_p()
        assert B.lib().bvg_host_alloc(4 * m, C.byref(p)) == 0
        sc = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(m,))
    else:
        sc = np.empty(m, dtype=np.int32)
    assert g.decode_range_into(0, n, rp, sc) == m
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    lo, hi = 123_457, 1_400_001                                    # a sub-range: referents before it come through the halo
    sc[:] = -7
    a = g.decode_range_into(lo, hi, rp[:hi - lo + 1], sc)
    assert a == rowptr[hi] - rowptr[lo] and np.array_equal(sc[:a], succ[rowptr[lo]:rowptr[hi]]) and sc[a] == -7
    # count only, and a buffer that is too small
    assert g.decode_range_into(0, n, rp, None) == m
    with pytest.raises(B.BvgError) as e:
        g.decode_range_into(0, n, rp, sc[:m - 1])
    assert e.value.code == B.BVG_ECAP
    if pinned:
        del sc
        B.lib().bvg_host_free(p)
    g.close()


def test_scan_checksum(cnr_gpu, cnr_oracle):
    """ImmutableGraph.hashCode() (ImmutableGraph.java:757-770) without materialising a list for the caller; ranges compose."""
    og, rp, sc = cnr_oracle
    n = cnr_gpu.numNodes()
    assert cnr_gpu.scan_checksum() == (1711395807, 3216152)
    assert cnr_gpu.hashCode() == 1711395807
    h, tot = -1, 0
    for lo, hi in [(0, 7), (7, 7), (7, 100000), (100000, 100001), (100001, n)]:
        h, a = cnr_gpu.scan_checksum(lo, hi, h)
        assert a == rp[hi] - rp[lo]
        tot += a
    assert (h, tot) == (1711395807, 3216152)
    with pytest.raises(ValueError):
        cnr_gpu.scan_checksum(5, n + 1)


@pytest.mark.parametrize("env", [{}, {"BVGPU_HASH_MATERIALISE": "1"}, {"BVGPU_COOP_MIN": "16", "BVGPU_GIANT_MIN": "64"}, {"BVGPU_COOP_MIN": "2147483647"}, {"BVGPU_SCAN_PIECE": "90000"},
                                 {"BVGPU_OVERLAP": "0"}, {"BVGPU_COPY_MID_MIN": "8", "BVGPU_COOP_MIN": "300", "BVGPU_GIANT_MIN": "3000"}],
                         ids=["fold_in_scan", "materialise_then_fold", "small_thresholds", "lanes_only", "pieces", "serial", "mid_thresholds"])
def test_scan_checksum_folds_inside_the_scan(cnr_oracle, monkeypatch, env):
    """Round 5: the hash is folded by the scan itself -- the node numbers while the row starts are written, the rows without a reference by the one-lane
    parse as it decodes them (writing only the rows some other row copies from), the lane class of the copy pass as it merges, the rows of the wave / group classes from
    memory through their work lists.  Every split of the rows between these (thresholds), every piece size and the old decode-then-fold path give the reference's
    hashCode (ImmutableGraph.java:757-770) on every range.  (The knobs are read when a handle is created: a handle of its own per case.)"""
    from webgraph_amd.bvgraph import BVGraph
    og, rp, sc = cnr_oracle
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = BVGraph.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cnr-2000"))
    n = g.numNodes()
    assert g.scan_checksum() == (1711395807, 3216152)
    for lo, hi, h0 in [(0, 1, 7), (3, 4, -1), (0, 2000, 0), (1999, 64001, 12345), (200000, n, -1), (n - 1, n, 5)]:
        h, a = g.scan_checksum(lo, hi, h0)
        assert a == rp[hi] - rp[lo]
        assert h == og.scan(lo, hi, want_succ=False, want_hash=True, h0=h0)[3], (lo, hi)
    g.set_option("hash_materialise", 1)  # the same handle the other way (bvg_set_option on a live handle)
    assert g.scan_checksum() == (1711395807, 3216152)
    with pytest.raises(ValueError):
        g.set_option("no_such_knob", 1)
    g.close()


def test_scans_in_pieces(cnr_gpu, cnr_oracle, monkeypatch):
    """The device-side scans cut a range into pieces of bounded scratch (256 M arcs by default): same answers in 20 pieces."""
    cnr_gpu.set_option("scan_piece", 170000)
    n = cnr_gpu.numNodes()
    assert cnr_gpu.scan_checksum() == (1711395807, 3216152)
    whole = cnr_gpu.scan_stats(0, n)
    cnr_gpu.set_option("scan_piece", 0)
    assert cnr_gpu.scan_stats(0, n) == whole


def test_split_iterators_drained_by_concurrent_threads(cnr_gpu, cnr_oracle):
    """The reference hands the iterators of splitNodeIterators to one thread each (BVGraph.java:2471-2477).  Each copy
    decodes through its own bvg_clone; small batches so that every thread makes many calls while the others do."""
    og, rp, sc = cnr_oracle
    n = cnr_gpu.numNodes()
    its = cnr_gpu.splitNodeIterators(6)
    for it in its:
        if hasattr(it, "_batch"):
            it._batch = 4096
    out, errs = [None] * len(its), []

    def drain(k):
        try:
            it = its[k]
            nodes, lists = [], []
            while it.hasNext():
                nodes.append(it.nextInt())
                lists.append(np.array(it.successorArray(), copy=True))
            out[k] = (nodes, lists)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=drain, args=(k,)) for k in range(len(its))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    seen = 0
    for nodes, lists in out:
        for x, l in zip(nodes, lists):
            assert x == seen and np.array_equal(l, sc[rp[x]:rp[x + 1]])
            seen += 1
    assert seen == n
    # a copy made in the middle of a scan continues from the same position, independently of the original
    it = cnr_gpu.nodeIterator(1000)
    for _ in range(10):
        it.nextInt()
    cp = it.copy(1200)
    a = [it.nextInt() for _ in range(5)]
    b = [cp.nextInt() for _ in range(5)]
    assert a == b == list(range(1010, 1015)) and np.array_equal(cp.successorArray(), sc[rp[1014]:rp[1015]])
    while cp.hasNext():
        last = cp.nextInt()
    assert last == 1199
