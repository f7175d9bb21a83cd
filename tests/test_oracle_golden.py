"""CPU tests: the oracle (oracle/bvg_oracle.c) against the reference's own known-answer fixture.

These pin the oracle: cnr-2000.{graph,offsets,properties,graph-txt.gz} are the data files of the reference's
BVGraphTest.testLarge (test/it/unimi/dsi/webgraph/BVGraphTest.java:101-119); the hashes are SURVEY.md App. C.
"""
import hashlib

import numpy as np
import pytest

from conftest import CNR
from oracle import oracle as O


def test_fixture_files_are_the_reference_ones():
    want = {
        ".graph": "b7d6b8bdf1218eb21edd77ce4ce091245050252972e687fc4d10a4d587f02db1",
        ".offsets": "c5268e312d8b4395518f85f6bd18f59049bb687e19b4307d45be08b3b81be3be",
        ".properties": "3fcb5ac1b1bd6505a30656726a737c7a13a3cf9e8739c9f4f18631902400dfef",
        ".graph-txt.gz": "ad05bc0dc8f826532a186a56279878eb34bbdd8ae8176e3cb0681bb571110f0a",
    }
    for ext, h in want.items():
        with open(CNR + ext, "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == h


def test_offsets_known_answers(cnr_oracle):
    g, _, _ = cnr_oracle
    assert list(g.offsets[:9]) == [0, 85, 113, 130, 131, 151, 152, 193, 229]
    assert g.offsets[-1] == 11443904 == 8 * 1430488


def test_sequential_scan_equals_ascii_golden(cnr_oracle):
    """BVGraphTest.testLarge, first half: BVGraph == ASCIIGraph(cnr-2000.graph-txt.gz)."""
    g, rowptr, succ = cnr_oracle
    n, rp, sc = O.read_ascii_graph_gz(CNR + ".graph-txt.gz")
    assert n == g.n == 325557
    assert np.array_equal(rowptr, rp) and np.array_equal(succ, sc)


def test_known_hashes(cnr_oracle):
    g, rowptr, succ = cnr_oracle
    assert rowptr[-1] == 3216152
    assert hashlib.sha256(succ.astype("<i4").tobytes()).hexdigest() == "f8830e775ef6087997ef5fae21c3f538f0417e555cd420d4527ffcb5aea52b3e"
    assert hashlib.sha256(rowptr.astype("<i8").tobytes()).hexdigest() == "2b9a18c9ce44dc8bc1d95bee4ed4e3a7625a2167ab12fab6bf50e6d9ea6829a6"
    assert g.hashcode() == 1711395807  # ImmutableGraph.hashCode()


def test_micro_kats(cnr_oracle):
    """records of nodes 0, 1, 3, 7 decoded by hand from the fixture's bits."""
    g, rowptr, succ = cnr_oracle
    row = lambda x: list(succ[rowptr[x]:rowptr[x + 1]])
    assert row(0) == [1] + list(range(342, 352)) + [211284, 223142]
    assert row(1) == [2, 3, 4, 319]
    assert row(3) == []
    assert row(6) == [117, 218, 296]
    assert row(7) == [6, 18, 218, 285, 296]
    assert row(2) == [211284, 223142] and row(5) == []
    assert row(325556) == [122557]
    d = np.diff(rowptr)
    assert d.max() == 2716 and int(d.argmax()) == 46918 and int((d == 0).sum()) == 78056


def test_random_access_equals_sequential(cnr_oracle):
    """BVGraphTest.testLarge, second half: successors(i) of every node (recursive path, BVG:1120)."""
    g, rowptr, succ = cnr_oracle
    rp, sc = g.successors_batch(np.arange(g.n, dtype=np.int32))
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    assert np.array_equal(g.outdegrees(), np.diff(rowptr).astype(np.int32))


@pytest.mark.parametrize("lo,hi", [(0, 0), (1, 9), (1000, 21000), (325000, 325557), (325557, 325557)])
def test_iterator_from_any_node(cnr_oracle, lo, hi):
    """WebGraphTestCase.assertGraph: nodeIterator(from) agrees with random access (window refill, BVG:1173-1183)."""
    g, rowptr, succ = cnr_oracle
    rp, sc, arcs = g.scan(lo, hi)
    assert arcs == rowptr[hi] - rowptr[lo]
    assert np.array_equal(rp, rowptr[lo:hi + 1] - rowptr[lo]) and np.array_equal(sc, succ[rowptr[lo]:rowptr[hi]])


def test_error_codes(cnr_oracle):
    g, _, _ = cnr_oracle
    with pytest.raises(O.OracleError) as e:
        g.successors(325557)
    assert e.value.code == -1  # IllegalArgumentException, BVG:900
    with pytest.raises(O.OracleError):
        g.outdegree(-1)
    seq = O.OracleGraph.load(CNR, with_offsets=False)
    with pytest.raises(O.OracleError) as e:
        seq.successors(3, cap=8)
    assert e.value.code == -3  # UnsupportedOperationException, BVG:901
    rp, sc, arcs = seq.scan()  # sequential access needs no offsets
    assert arcs == 3216152
