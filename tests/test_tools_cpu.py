"""CPU tests of the BVGraph writer / generator (libbvgtools) and of the host-only half of libbvgpu."""
import filecmp
import os

import numpy as np
import pytest

from conftest import CNR, make_graph
from oracle import oracle as O


def test_writer_reproduces_reference_bytes(tmp_path, cnr_oracle):
    """Recompressing cnr-2000 with its own parameters gives back the reference-produced .graph / .offsets bit for bit."""
    from webgraph_amd import tools as T
    _, rowptr, succ = cnr_oracle
    base = str(tmp_path / "cnr")
    st = T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=3, zeta_k=3)
    assert filecmp.cmp(base + ".graph", CNR + ".graph", shallow=False)
    assert filecmp.cmp(base + ".offsets", CNR + ".offsets", shallow=False)
    # BVGraphTest.testCompression's invariants (test/it/unimi/dsi/webgraph/BVGraphTest.java:59-72)
    bits = st["bits_outdegrees"] + st["bits_references"] + st["bits_blocks"] + st["bits_intervals"] + st["bits_residuals"]
    assert os.path.getsize(base + ".graph") == (bits + 7) // 8
    assert st["copied_arcs"] + st["intervalised_arcs"] + st["residual_arcs"] == rowptr[-1]
    assert (st["copied_arcs"], st["intervalised_arcs"], st["residual_arcs"]) == (2130833, 361894, 723425)
    assert st["max_ref_chain"] == 3
    og = cnr_oracle[0]
    c = og.copied()  # (the oracle's own count of the ids every record takes from its referent: what bench.py prices the copy pass by)
    assert int(c.sum()) == 2130833 and np.all(c <= np.diff(rowptr)) and not np.any((c > 0) & (og.references() == 0))


@pytest.mark.parametrize("w,r,i", [(0, 0, 0), (1, 1, 2), (2, 2, 3), (7, 3, 4), (3, 100, 1)])
def test_roundtrip_through_oracle(tmp_path_factory, w, r, i):
    """BVGraphTest.testCompression: store -> load -> equal, for several (window, maxRefCount, minIntervalLength)."""
    base, rowptr, succ = make_graph(tmp_path_factory, "rt", 3000, 40000, 5 + w, 0.7, window=w, max_ref_count=r, min_interval=i, threads=2)
    g = O.OracleGraph.load(base)
    rp, sc, arcs = g.scan()
    assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
    q = np.arange(0, 3000, 7, dtype=np.int32)
    brp, bsc = g.successors_batch(q)
    for j, x in enumerate(q):
        assert np.array_equal(bsc[brp[j]:brp[j + 1]], succ[rowptr[x]:rowptr[x + 1]])


def test_generator_is_deterministic_and_exact():
    from webgraph_amd import tools as T
    a = T.generate(20000, 300000, seed=99, threads=1)
    b = T.generate(20000, 300000, seed=99, threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])  # independent of the thread count
    rowptr, succ = a
    assert rowptr[-1] == 300000 and succ.size == 300000
    d = np.diff(rowptr)
    assert 0.15 < (d == 0).mean() < 0.25  # 20 % empty nodes
    for x in range(0, 20000, 500):
        row = succ[rowptr[x]:rowptr[x + 1]]
        assert np.all(np.diff(row) > 0) and (row.size == 0 or (row[0] >= 0 and row[-1] < 20000))


def test_library_exports_every_declared_symbol():
    """include/bvgpu.h <-> libbvgpu.so: every declared entry point is exported; no compute call here."""
    import re
    from webgraph_amd import bvgraph as B
    L = B.lib()
    hdr = open(os.path.join(os.path.dirname(CNR), "..", "..", "include", "bvgpu.h")).read()
    declared = set(re.findall(r"\b(bvg_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(B.EXPORTS)
    import ctypes
    T = ctypes.CDLL(os.path.join(os.path.dirname(B.__file__), "libbvgtools.so"))
    thdr = open(os.path.join(os.path.dirname(CNR), "..", "..", "include", "bvgtools.h")).read()
    tdecl = set(re.findall(r"\b(bvt_[a-z_]+)\s*\(", thdr))
    assert {"bvt_store", "bvt_generate", "bvt_free", "bvt_store_labels"} <= tdecl
    for name in tdecl:
        assert hasattr(T, name), name


def test_no_gpu_means_loud_failure():
    """There is no CPU fallback: opening a graph without a HIP device fails (this container has no GPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from webgraph_amd.bvgraph import BVGraph, BvgError
    with pytest.raises(BvgError) as e:
        BVGraph.load(CNR)
    assert e.value.code == -6


def test_parse_properties_like_the_reference(tmp_path):
    from webgraph_amd import bvgraph as B
    info = B.parse_properties(CNR)
    assert (info.nodes, info.arcs, info.window_size, info.max_ref_count, info.min_interval_length, info.zeta_k) == (325557, 3216152, 7, 3, 3, 3)
    assert (info.outdegree_coding, info.block_coding, info.residual_coding, info.reference_coding, info.block_count_coding, info.offset_coding) == (2, 2, 6, 5, 2, 2)

    def write(name, text):
        p = tmp_path / (name + ".properties")
        p.write_text(text)
        return str(tmp_path / name)

    ok = "graphclass=it.unimi.dsi.webgraph.BVGraph\nversion=0\nnodes=5\narcs=7\nwindowsize=7\nmaxrefcount=3\nminintervallength=4\n"
    i2 = B.parse_properties(write("a", ok + "compressionflags=RESIDUALS_GAMMA | OFFSETS_DELTA\n"))
    assert i2.residual_coding == 2 and i2.offset_coding == 1 and i2.zeta_k == 3
    # the `big` spelling is accepted (BVG:1528); ':' and blank separators are java.util.Properties syntax
    i3 = B.parse_properties(write("b", ok.replace("it.unimi.dsi.webgraph", "it.unimi.dsi.big.webgraph").replace("nodes=5", "nodes : 5").replace("arcs=7", "arcs 7")))
    assert i3.nodes == 5 and i3.arcs == 7
    with pytest.raises(NotImplementedError):  # version > 0 (BVG:1534)
        B.parse_properties(write("c", ok.replace("version=0", "version=1")))
    with pytest.raises(NotImplementedError):  # missing version (BVG:1533)
        B.parse_properties(write("d", ok.replace("version=0\n", "")))
    with pytest.raises(NotImplementedError):  # another graph class (BVG:1528)
        B.parse_properties(write("e", ok.replace("BVGraph", "ASCIIGraph")))
    # EFGraph is the second format this library reads (EFGraph.loadInternal, EFGraph.java:709-750)
    ef = "graphclass=it.unimi.dsi.webgraph.EFGraph\nversion=0\nnodes=5\narcs=7\nquantum=256\nbyteorder=LITTLE_ENDIAN\n"
    i4 = B.parse_properties(write("f", ef))
    assert (i4.format, i4.nodes, i4.arcs, i4.ef_upper_bound, i4.ef_log2_quantum, i4.ef_big_endian, i4.offset_coding) == (B.BVG_FORMAT_EF, 5, 7, 5, 8, 0, 1)
    i5 = B.parse_properties(write("g", ef.replace("LITTLE", "BIG") + "upperbound=9\n"))
    assert (i5.ef_upper_bound, i5.ef_big_endian) == (9, 1)
    # the binding's class name in graphclass (what INTEGRATION.md tells a user to put there): the other properties say which format
    i6 = B.parse_properties(write("j", ok.replace("it.unimi.dsi.webgraph.BVGraph", "it.unimi.dsi.webgraph.gpu.GpuBVGraph")))
    i7 = B.parse_properties(write("k", ef.replace("it.unimi.dsi.webgraph.EFGraph", "it.unimi.dsi.webgraph.gpu.GpuBVGraph")))
    assert (i6.format, i6.window_size, i7.format, i7.ef_log2_quantum) == (B.BVG_FORMAT_BV, 7, B.BVG_FORMAT_EF, 8)
    with pytest.raises(ValueError):  # "Illegal quantum (must be a power of 2)", :745
        B.parse_properties(write("h", ef.replace("quantum=256", "quantum=255")))
    with pytest.raises(ValueError):  # "Unknown byte order", :750
        B.parse_properties(write("i", ef.replace("LITTLE_ENDIAN", "PDP")))
    with pytest.raises(ValueError):  # nodes >= 2^31 (BVG:1537)
        B.parse_properties(write("f", ok.replace("nodes=5", "nodes=2147483648")))
    with pytest.raises(NotImplementedError):  # unknown flag name (BVG:1361)
        B.parse_properties(write("g", ok + "compressionflags=RESIDUALS_FOO\n"))
    with pytest.raises(IOError):
        B.parse_properties(str(tmp_path / "missing"))


def test_flags_and_offsets_host_logic(cnr_oracle):
    from webgraph_amd import bvgraph as B
    assert B.flags_from_string("") == 0
    assert B.flags_from_string("OUTDEGREES_DELTA|BLOCKS_DELTA | RESIDUALS_NIBBLE|REFERENCES_GAMMA| BLOCK_COUNT_UNARY |OFFSETS_DELTA") == 1 | 1 << 4 | 7 << 8 | 2 << 12 | 5 << 16 | 1 << 20
    with pytest.raises(IOError):
        B.flags_from_string("BLOCKS_UNARY")  # not a public constant of BVGraph (BVG:475-523)
    g, _, _ = cnr_oracle
    with open(CNR + ".offsets", "rb") as f:
        offs = B.decode_offsets_host(f.read(), g.n)
    assert np.array_equal(offs, g.offsets)  # product's own decoder == oracle's


def test_cpp_host_mirror_compiles(tmp_path):
    """The C++ mirror of the reference API (webgraph_amd/host/bvgraph.hpp) builds against the C ABI; without a GPU it fails loudly."""
    import subprocess
    import torch
    from conftest import ROOT
    exe = str(tmp_path / "host_mirror_test")
    pkg = os.path.join(ROOT, "webgraph_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"),
                           "-L" + pkg, "-lbvgpu", "-Wl,-rpath," + pkg, "-L/opt/rocm/lib", "-lamdhip64"])
    if not torch.cuda.is_available():
        p = subprocess.run([exe, CNR, "1711395807", "3216152"], capture_output=True, text=True)
        assert p.returncode != 0 and "no HIP device" in p.stderr


def _props(path):
    return dict(l.strip().split("=", 1) for l in open(path) if "=" in l and not l.startswith("#"))


def _gap_keys(bins):
    """The three keys BVGraph.java:2592-2632 derives from one histogram, restated with Python's exact arithmetic."""
    import math
    from decimal import ROUND_HALF_EVEN, Decimal
    used = [i for i, b in enumerate(bins) if b]
    l = used[-1] if used else -1
    gaps = sum(bins[:l + 1])
    tot = sum((3 * (1 << i) - 1) * bins[i] for i in range(l + 1))
    tot_log = 0.0
    for i in range(l + 1):
        tot_log += (math.log(float(3 * (1 << i) + 1)) / 0.6931471805599453 - 1) * bins[i]
    exp = ",".join(str(b) for b in bins[:l + 1])
    if gaps == 0:
        return exp, "0", "0"
    return exp, str((Decimal(tot) / Decimal(2 * gaps)).quantize(Decimal("0.001"), rounding=ROUND_HALF_EVEN)), repr(tot_log / gaps)


def _bins_of_lists(nodes, lists):
    """updateBins (BVGraph.java:1940-1944): gaps by most significant bit; the first element by int2nat(first - node), skipped when 0."""
    bins = [0] * 32
    for x, v in zip(nodes, lists):
        if len(v) == 0:
            continue
        d = int(v[0]) - int(x)
        g = 2 * d if d >= 0 else -2 * d - 1
        if g:
            bins[g.bit_length() - 1] += 1
        for a, b in zip(v[:-1], v[1:]):
            bins[(int(b) - int(a)).bit_length() - 1] += 1
    return bins


@pytest.mark.parametrize("I", [0, 3])
def test_properties_carry_the_gap_statistics(tmp_path, I):
    """successorexpstats / residualexpstats and their averages (BVGraph.java:2592-2632), against a restatement of updateBins; without references
    the residuals are what intervalize (:1631-1654) leaves of the successor list."""
    from webgraph_amd import tools as T
    rowptr, succ = T.generate(3000, 40000, seed=99, p_copy=0.3)
    base = str(tmp_path / "g")
    st = T.store(base, rowptr, succ, window=0, max_ref_count=0, min_interval=I)
    p = _props(base + ".properties")
    lists = [succ[rowptr[x]:rowptr[x + 1]] for x in range(3000)]
    sb = _bins_of_lists(range(3000), lists)
    exp, avg, avglog = _gap_keys(sb)
    assert (p["successorexpstats"], p["successoravggap"], p["successoravgloggap"]) == (exp, avg, avglog)
    res = []
    for v in lists:
        v = [int(t) for t in v]
        out, i = [], 0
        while i < len(v):
            j = i
            while j + 1 < len(v) and v[j + 1] == v[j] + 1:
                j += 1
            if I == 0 or j - i + 1 < I:
                out += v[i:j + 1]
            i = j + 1
        res.append(out)
    assert sum(len(r) for r in res) == st["residual_arcs"]
    rb = _bins_of_lists(range(3000), res)
    exp, avg, avglog = _gap_keys(rb)
    assert (p["residualexpstats"], p["residualavggap"], p["residualavgloggap"]) == (exp, avg, avglog)
    n, m = 3000, int(rowptr[-1])
    for k, bits in (("outdegrees", st["bits_outdegrees"]), ("references", st["bits_references"]), ("blocks", st["bits_blocks"]),
                    ("residuals", st["bits_residuals"]), ("intervals", st["bits_intervals"])):
        want = ("%.3f" % (bits / n)).rstrip("0").rstrip(".")
        assert p["avgbitsfor" + k] == want, k
    assert "compratio" in p


def test_host_walk_of_gamma_labels(tmp_path):
    """bvh::decode_gammas (bv_host.cpp): the host walk that stands behind the device decoder of gamma-coded labels (streams that never re-synchronise; streams that do not
    hold one label per arc) -- tests/cpp/host_bits_test.cpp writes streams bit by bit and reads them back from any offset."""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "host_bits_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_bits_test.cpp"), os.path.join(ROOT, "webgraph_amd", "csrc", "bv_host.cpp")])
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr
