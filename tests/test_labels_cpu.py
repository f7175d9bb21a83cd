"""Arc labels, CPU side: the writer (webgraph_amd.tools.store_labels), the oracle decoder and
the host-only property parser of libbvgpu agree on the BitStreamArcLabelledImmutableGraph file format
(labelling/BitStreamArcLabelledImmutableGraph.java:60-135, :650-695).  The reference ships no label fixture and tests
this format by round trip only (test/.../labelling/BitStreamArcLabelledGraphTest.java): parity is unpinned here too."""
import numpy as np
import pytest


def _graph(tmp_path, n=3000, m=40000, seed=9):
    from webgraph_amd import tools as T
    rowptr, succ = T.generate(n, m, seed=seed, p_copy=0.5, threads=2)
    base = str(tmp_path / "g")
    T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=4)
    return base, rowptr, succ


@pytest.mark.parametrize("kind,width", [("gamma", 0), ("fixed", 10), ("fixed", 1), ("fixed", 32), ("fixed", 0)])
def test_labels_round_trip_through_the_oracle(tmp_path, kind, width):
    from webgraph_amd import tools as T
    from oracle import oracle as O
    base, rowptr, succ = _graph(tmp_path)
    n, m = rowptr.size - 1, succ.size
    rng = np.random.Generator(np.random.PCG64(3))
    if kind == "gamma":
        labels = (rng.pareto(1.0, size=m) * 3).astype(np.int64).clip(0, 2**31 - 2).astype(np.int32)
    else:
        labels = (rng.integers(0, 2**width, size=m, dtype=np.int64) if width else np.zeros(m, dtype=np.int64)).astype(np.uint32).view(np.int32)
    lbase = str(tmp_path / "lab")
    T.store_labels(lbase, "g", rowptr, labels, kind=kind, width=width, key="WEIGHT")
    props = O.parse_properties(lbase + ".properties")
    assert props["underlyinggraph"] == "g" and "BitStreamArcLabelledImmutableGraph" in props["graphclass"]
    assert O.parse_labelspec(props["labelspec"]) == ((1, -1, "WEIGHT") if kind == "gamma" else (2, width, "WEIGHT"))
    d = np.diff(rowptr).astype(np.int32)
    assert np.array_equal(O.labels_decode(lbase, n, d), labels)
    lo, hi = 1000, 1777
    assert np.array_equal(O.labels_decode(lbase, n, d[lo:hi], lo, hi), labels[rowptr[lo]:rowptr[hi]])
    # the label offsets are a gamma gap stream like .offsets: n+1 values, first 0, last = bits used
    off = O.decode_offsets(open(lbase + ".labeloffsets", "rb").read(), n)
    assert off[0] == 0 and off[-1] <= 8 * len(open(lbase + ".labels", "rb").read())
    if kind == "fixed":
        assert np.array_equal(off, rowptr * width)


def test_labels_properties_host_parser(tmp_path):
    import ctypes as C
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import BvgLabelsInfo, lib
    base, rowptr, succ = _graph(tmp_path, 200, 1500, 4)
    T.store_labels(str(tmp_path / "lab"), "g", rowptr, np.zeros(succ.size, dtype=np.int32), kind="fixed", width=7, key="K")
    info, err = BvgLabelsInfo(), C.create_string_buffer(256)
    assert lib().bvg_labels_parse_properties(str(tmp_path / "lab").encode(), C.byref(info), err, 256) == 0
    assert (info.kind, info.width, info.key) == (2, 7, b"K") and info.underlying.decode() == str(tmp_path / "g")
    # a BVGraph property file is not a labelled graph
    assert lib().bvg_labels_parse_properties(base.encode(), C.byref(info), err, 256) != 0
    # unsupported label classes are reported as such
    open(str(tmp_path / "lab") + ".properties", "w").write(
        "graphclass = it.unimi.dsi.webgraph.labelling.BitStreamArcLabelledImmutableGraph\nunderlyinggraph = g\n"
        "labelspec = it.unimi.dsi.webgraph.labelling.FixedWidthIntListLabel(K,3)\n")
    assert lib().bvg_labels_parse_properties(str(tmp_path / "lab").encode(), C.byref(info), err, 256) == 0
    assert (info.kind, info.width, info.key) == (3, 3, b"K")
    open(str(tmp_path / "lab") + ".properties", "w").write(
        "graphclass = it.unimi.dsi.webgraph.labelling.BitStreamArcLabelledImmutableGraph\nunderlyinggraph = g\n"
        "labelspec = org.example.SomeOtherLabel(K,3)\n")
    assert lib().bvg_labels_parse_properties(str(tmp_path / "lab").encode(), C.byref(info), err, 256) == -3


def _lists(rng, m, width, mean=1.5):
    lens = rng.poisson(mean, size=m).astype(np.int64)
    lens[rng.integers(0, m, size=max(m // 500, 1))] += rng.integers(50, 400)  # a few long lists
    listptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nv = int(listptr[-1])
    values = (rng.integers(0, 2**width, size=nv, dtype=np.int64) if width else np.zeros(nv, dtype=np.int64)).astype(np.uint32).view(np.int32)
    return listptr, values


@pytest.mark.parametrize("width", [0, 1, 9, 32])
def test_label_lists_round_trip_through_the_oracle(tmp_path, width):
    """FixedWidthIntListLabel (FixedWidthIntListLabel.java:107-119): writer and oracle decoder agree; lists may be empty."""
    from webgraph_amd import tools as T
    from oracle import oracle as O
    base, rowptr, succ = _graph(tmp_path)
    n, m = rowptr.size - 1, succ.size
    listptr, values = _lists(np.random.Generator(np.random.PCG64(11 + width)), m, width)
    lbase = str(tmp_path / "lab")
    T.store_label_lists(lbase, "g", rowptr, listptr, values, width, key="L")
    props = O.parse_properties(lbase + ".properties")
    assert O.parse_labelspec(props["labelspec"]) == (3, width, "L")
    d = np.diff(rowptr).astype(np.int32)
    lp, vals = O.label_lists_decode(lbase, n, d)
    assert np.array_equal(lp, listptr) and np.array_equal(vals, values)
    lo, hi = 1000, 1777
    lp, vals = O.label_lists_decode(lbase, n, d[lo:hi], lo, hi)
    a0, a1 = rowptr[lo], rowptr[hi]
    assert np.array_equal(lp, listptr[a0:a1 + 1] - listptr[a0]) and np.array_equal(vals, values[listptr[a0]:listptr[a1]])
    # a wrong outdegree vector does not end on the node boundaries
    d2 = d.copy(); d2[5] += 1
    with pytest.raises(O.OracleError):
        O.label_lists_decode(lbase, n, d2)
