"""GPU decode of the .offsets stream against the host decoder and the golden offsets.

OffsetsLongIterator (BVGraph.java:907-935): n+1 gamma-coded gaps, running sum = bit offset of every record.
"""
import numpy as np
import pytest

from conftest import CNR, make_graph

pytestmark = pytest.mark.gpu


def test_cnr2000_offsets_on_device():
    from webgraph_amd.bvgraph import decode_offsets_device, decode_offsets_host
    raw = open(CNR + ".offsets", "rb").read()
    n = 325557
    dev = decode_offsets_device(raw, n)
    host = decode_offsets_host(raw, n)
    assert np.array_equal(dev, host)
    assert list(dev[:9]) == [0, 85, 113, 130, 131, 151, 152, 193, 229]
    assert dev[-1] <= 1430488 * 8


@pytest.mark.parametrize("n,m,seed", [(1, 0, 1), (3, 2, 2), (64, 300, 3), (5000, 60000, 4), (300000, 9000000, 5)])
def test_synthetic_offsets_on_device(tmp_path_factory, n, m, seed):
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import BVGraph, decode_offsets_device, decode_offsets_host
    if n <= 3:  # hand-made: the generator does not go this small
        rowptr = np.array([0, 0] if n == 1 else [0, 2, 2, 2], dtype=np.int64)
        succ = np.array([] if n == 1 else [1, 2], dtype=np.int32)
        base = str(tmp_path_factory.mktemp("offs%d" % n) / "g")
        T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=4)
    else:
        base, rowptr, succ = make_graph(tmp_path_factory, "offs%d" % n, n, m, seed, 0.5, window=7, max_ref_count=3, min_interval=4)
    raw = open(base + ".offsets", "rb").read()
    assert np.array_equal(decode_offsets_device(raw, n), decode_offsets_host(raw, n))
    g = BVGraph.load(base)
    assert g.info.offsets_on_device == 1  # the load path used the kernels
    rp, sc = g.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    g.close()


@pytest.mark.parametrize("n,m,seed", [(64, 300, 13), (5000, 60000, 14), (300000, 9000000, 15)])
def test_delta_coded_offsets_on_device(tmp_path_factory, n, m, seed):
    """OFFSETS_DELTA (CompressionFlags.java, the offset_coding field of BVG:1317-1325): the same kernels with the delta decoder."""
    from webgraph_amd.bvgraph import BVGraph, decode_offsets_device, decode_offsets_host, flags_from_string
    base, rowptr, succ = make_graph(tmp_path_factory, "offd%d" % n, n, m, seed, 0.5, window=7, max_ref_count=3, min_interval=4, flags=flags_from_string("OFFSETS_DELTA"))
    raw = open(base + ".offsets", "rb").read()
    host = decode_offsets_host(raw, n, 1)
    assert np.array_equal(decode_offsets_device(raw, n, coding=1), host)
    g = BVGraph.load(base)
    assert g.info.offset_coding == 1 and g.info.offsets_on_device == 1
    rp, sc = g.decode_range()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    g.close()


def test_offsets_rejects_wrong_count_and_delta():
    from webgraph_amd.bvgraph import decode_offsets_device
    raw = open(CNR + ".offsets", "rb").read()
    with pytest.raises(Exception):
        decode_offsets_device(raw, 325557 + 5)  # more values asked than the stream holds
    with pytest.raises(Exception):
        decode_offsets_device(raw, 1000)        # fewer: the count must match exactly
    with pytest.raises(Exception):
        decode_offsets_device(raw, 325557, coding=1)  # a gamma stream read as delta codes does not hold n + 1 of them


def test_host_offsets_knob(monkeypatch):
    from webgraph_amd.bvgraph import BVGraph
    monkeypatch.setenv("BVGPU_OFFSETS", "host")
    g = BVGraph.load(CNR)
    assert g.info.offsets_on_device == 0
    assert g.hashCode() == 1711395807
    g.close()
