"""EFGraph, the reference's second on-disk format (src/it/unimi/dsi/webgraph/EFGraph.java), CPU side:
the writer (webgraph_amd.tools.store_ef) against a record worked out by hand from the format description, and the oracle's
reader (oracle/efg_oracle.c) against what was stored.  The reference ships no EFGraph fixture: parity is unpinned."""
import numpy as np
import pytest


# n = 4, upper bound 4, quantum 256 (no forward pointers).  Per node: gamma(outdegree) from the low bit up (msb zeros, a one,
# then the msb low bits of outdegree + 1), then -- for the successors followed by the terminator 4 -- l lower bits each and
# the upper bits in negated unary (element i sets bit (value >> l) + i), l = msb(4 / (outdegree + 1)) or 0:
#   node 0 -> {1, 3}      gamma(2) = 0 1 1   l = 0  upper ones at 1, 4, 6        0 1 0 0 1 0 1
#   node 1 -> {}          gamma(0) = 1       l = 2  lower 00  upper one at 1      0 0 | 0 1
#   node 2 -> {0,1,2,3}   gamma(4) = 0 0 1 1 0  l = 0  upper ones at 0,2,4,6,8    1 0 1 0 1 0 1 0 1
#   node 3 -> {2}         gamma(1) = 0 1 0   l = 1  lower 0 0  upper ones at 1, 3  0 1 0 1
KAT_BITS = [0, 1, 1, 0, 1, 0, 0, 1, 0, 1,
            1, 0, 0, 0, 1,
            0, 0, 1, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1,
            0, 1, 0, 0, 0, 0, 1, 0, 1]
KAT_ROWS = [[1, 3], [], [0, 1, 2, 3], [2]]
KAT_OFFSETS = [0, 10, 15, 29, 38]


def _csr(rows):
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    return rowptr, np.array([v for r in rows for v in r], dtype=np.int32)


@pytest.mark.parametrize("big", [False, True])
def test_writer_reproduces_the_hand_made_record(tmp_path, big):
    from webgraph_amd import tools as T
    from oracle import oracle as O
    rowptr, succ = _csr(KAT_ROWS)
    base = str(tmp_path / "kat")
    T.store_ef(base, rowptr, succ, log2_quantum=8, big_endian=big)
    word = sum(b << i for i, b in enumerate(KAT_BITS))
    assert open(base + ".graph", "rb").read() == word.to_bytes(8, "big" if big else "little")  # close() writes the last buffer
    assert list(O.decode_offsets(open(base + ".offsets", "rb").read(), 4, coding=O.DELTA)) == KAT_OFFSETS
    props = O.parse_properties(base + ".properties")
    assert props["graphclass"] == "it.unimi.dsi.webgraph.EFGraph" and props["quantum"] == "256" and props["byteorder"] == ("BIG_ENDIAN" if big else "LITTLE_ENDIAN")
    g = O.OracleEFGraph.load(base)
    rp, sc, arcs = g.scan()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ) and arcs == 7


@pytest.mark.parametrize("n,m,lq,big,ub", [(3000, 40000, 8, False, None), (3000, 40000, 2, True, None), (500, 20000, 0, False, 700), (2000, 6000, 8, False, None)])
def test_round_trip_through_the_oracle(tmp_path, n, m, lq, big, ub):
    """Forward pointers present (small quanta), dense and sparse rows, an upper bound above n, both byte orders."""
    from webgraph_amd import tools as T
    from oracle import oracle as O
    rowptr, succ = T.generate(n, m, seed=41 + lq, p_copy=0.5, threads=2)
    base = str(tmp_path / "ef")
    T.store_ef(base, rowptr, succ, upper_bound=ub, log2_quantum=lq, big_endian=big)
    g = O.OracleEFGraph.load(base)
    assert (g.n, g.arcs, g.upper_bound, g.log2_quantum) == (n, succ.size, ub or n, lq)
    rp, sc, _ = g.scan()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    lo, hi = n // 3, n // 3 + 300
    rp, sc, _ = g.scan(lo, hi)
    assert np.array_equal(rp, rowptr[lo:hi + 1] - rowptr[lo]) and np.array_equal(sc, succ[rowptr[lo]:rowptr[hi]])
    # the record lengths in .offsets are what the records take
    assert g.offsets[-1] <= 64 * g.words.size and g.offsets[-1] > 64 * (g.words.size - 2)


def test_rows_must_increase_below_the_bound(tmp_path):
    from webgraph_amd import tools as T
    with pytest.raises(OSError):
        T.store_ef(str(tmp_path / "x"), *_csr([[1, 1], []]))
    with pytest.raises(OSError):
        T.store_ef(str(tmp_path / "x"), *_csr([[0, 5], []]))


@pytest.mark.parametrize("n,m,lq,big,ub", [(3000, 60000, 2, False, None), (3000, 60000, 0, True, None), (800, 40000, 3, False, 1500), (20000, 100000, 8, False, None)])
def test_forward_pointers_take_skip_to_where_a_search_goes(tmp_path, n, m, lq, big, ub):
    """The forward pointers of a record (Accumulator.add, EFGraph.java:502-516) are read by nothing in a scan; here skipTo (:1147-1215) goes THROUGH them --
    more than a quantum of zeros to skip -- and must land on the first successor >= the bound, which a search in the scanned list gives."""
    from webgraph_amd import tools as T
    from oracle import oracle as O
    rowptr, succ = T.generate(n, m, seed=77 + lq, p_copy=0.4, threads=2)
    base = str(tmp_path / "ef")
    T.store_ef(base, rowptr, succ, upper_bound=ub, log2_quantum=lq, big_endian=big)
    g = O.OracleEFGraph.load(base)
    rng = np.random.default_rng(5)
    deg = np.diff(rowptr)
    long_rows = np.nonzero(deg >= 8)[0]
    nodes = np.concatenate([rng.integers(0, n, 3000), rng.choice(long_rows, 3000)]).astype(np.int32)
    bounds = rng.integers(0, ub or n, nodes.size).astype(np.int32)
    got, used = g.skip_to(nodes, bounds)
    want = np.empty_like(got)
    for i, (x, b) in enumerate(zip(nodes, bounds)):
        row = succ[rowptr[x]:rowptr[x + 1]]
        k = np.searchsorted(row, b)
        want[i] = row[k] if k < row.size else -1
    assert np.array_equal(got, want)
    if lq <= 3:
        assert used.sum() > 1000  # the pointers were on the path
