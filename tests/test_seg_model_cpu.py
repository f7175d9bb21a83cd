"""The segment pipeline's logic on the CPU: tests/cpp/seg_model.cpp compiles bv_seg.hpp (the bodies that bv_seg.hip runs on the GPU:
struct, A1, A2, B, expand) with g++ and drives them lane after lane; what they decode is compared with the CPU oracle
(BVGraph.java:1032-1133 restated in oracle/bvg_oracle.c).  Rows without a reference must equal the oracle's rows; for a row with a
reference the kernels' contract is "extras in row[copied..d)" -- the extras must be a strictly increasing subset of the oracle's row,
and the ids of the row that are not extras must all come from the referent's row."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import CNR, ROOT, make_graph


def build_model(dirname, seg_bits_log2=None, any_always=False):
    so = os.path.join(str(dirname), "libsegmodel%s%s.so" % (seg_bits_log2 or "", "a" if any_always else ""))
    cmd = ["g++", "-std=c++17", "-O2", "-g", "-shared", "-fPIC", "-D_GLIBCXX_ASSERTIONS", "-o", so, os.path.join(ROOT, "tests", "cpp", "seg_model.cpp")]
    if seg_bits_log2:
        cmd.insert(1, "-DSEG_BITS_LOG2_=%d" % seg_bits_log2)
    if any_always:
        cmd.insert(1, "-DSG_ANY_ALWAYS")
    subprocess.check_call(cmd)
    L = C.CDLL(so)
    for f in (L.seg_model_run,):
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                      C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]
    return L


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    return build_model(tmp_path_factory.mktemp("seg_model"))


@pytest.fixture(scope="module")
def model_always(tmp_path_factory):
    """on the GPU a lane refills its window and tops up its ring whenever ANY lane of its wave needs to: here, always"""
    return build_model(tmp_path_factory.mktemp("seg_modela"), 9, True)


@pytest.fixture(scope="module")
def model_small(tmp_path_factory):
    """pieces of 128 bits: every record of a few successors is cut into several segments"""
    return build_model(tmp_path_factory.mktemp("seg_model7"), 7)


def run_model(L, base, lo=0, hi=None, dmin=1, dmax=1 << 30):
    from oracle import oracle as O
    og = O.OracleGraph.load(base)
    n = og.n
    hi = n if hi is None else hi
    raw = open(base + ".graph", "rb").read()
    graph = np.frombuffer(raw + b"\0" * 64, dtype=np.uint8)
    offsets = og.offsets
    rowptr, succ, arcs = og.scan(lo, hi)
    outd = np.diff(rowptr).astype(np.int32)
    ref = og.references(lo, hi).astype(np.uint16)
    cnt = hi - lo
    got = np.full(max(arcs, 1), -7, dtype=np.int32)
    esc = np.zeros(max(cnt, 1), dtype=np.int32)
    nesc = C.c_int32(0)
    cop = np.zeros(max(cnt, 1), dtype=np.int32)
    stats = np.zeros(8, dtype=np.int64)
    p = og.params
    rc = L.seg_model_run(graph.ctypes.data, len(raw), offsets.ctypes.data, lo, cnt, outd.ctypes.data, ref.ctypes.data, rowptr.ctypes.data,
                         p.window, p.min_interval, p.zeta_k, dmin, dmax, got.ctypes.data, esc.ctypes.data, C.byref(nesc), cop.ctypes.data, stats.ctypes.data)
    assert rc == 0
    return dict(rowptr=rowptr, succ=succ, outd=outd, ref=ref.astype(np.int64), got=got, esc=esc[:nesc.value], cop=cop, stats=stats, cnt=cnt)


def check(r, dmin, dmax, max_escapes=0):
    rowptr, succ, got, outd, ref, cop = r["rowptr"], r["succ"], r["got"], r["outd"], r["ref"], r["cop"]
    escaped = set(int(s) for s in r["esc"])
    # (a sub-range without its halo: the first rows may refer to nodes before it -- the real pipeline never shows the kernels such a row)
    outside = set(int(s) for s in np.nonzero(ref > np.arange(ref.size))[0])
    assert len(escaped - outside) <= max_escapes, "flagged: %s" % sorted(escaped - outside)[:20]
    escaped |= outside
    work = (outd >= max(dmin, 1)) & (outd < dmax)
    nchecked = 0
    for s in np.nonzero(work)[0]:
        s = int(s)
        if s in escaped:
            continue
        a, b = int(rowptr[s]), int(rowptr[s + 1])
        want = succ[a:b]
        c = int(cop[s])
        assert 0 <= c <= b - a, "node %d: copied %d of %d" % (s, c, b - a)
        if ref[s] == 0:
            assert c == 0
            assert np.array_equal(got[a:b], want), "node %d (no reference): %s vs %s" % (s, got[a:b][:12], want[:12])
        else:
            extras = got[a + c:b]
            assert np.all(np.diff(extras) > 0), "node %d: extras not increasing" % s
            assert np.all(np.isin(extras, want)), "node %d: extras outside the row" % s
            rest = np.setdiff1d(want, extras)
            assert rest.size == c, "node %d: %d ids left, %d copied" % (s, rest.size, c)
            t = s - int(ref[s])
            assert np.all(np.isin(rest, succ[int(rowptr[t]):int(rowptr[t + 1])])), "node %d: copied ids not in the referent's row" % s
        nchecked += 1
    # nothing else was touched
    other = np.nonzero(~work)[0]
    for s in other[:2000]:
        a, b = int(rowptr[s]), int(rowptr[s + 1])
        assert np.all(got[a:b] == -7)
    return nchecked


def test_cnr2000_every_record(model):
    """every non-empty record of the fixture through the pipeline (the class bounds are only a choice of speed)"""
    r = run_model(model, CNR)
    n = check(r, 1, 1 << 30, max_escapes=2)
    assert n > 240000
    st = r["stats"]
    assert st[1] > st[0] // 2 and st[5] > 40000  # segments, intervals


def test_cnr2000_small_pieces(model_small):
    """pieces of 128 bits: thousands of records are cut, chains must meet within a dozen codewords or the record is flagged"""
    r = run_model(model_small, CNR, dmin=8)
    st = r["stats"]
    n = check(r, 8, 1 << 30, max_escapes=int(st[0]) // 4)
    assert st[1] > st[0] and st[2] > 0 and n > 50000  # some chains do not meet within 128 bits


def test_cnr2000_wave_synchronised_steps_are_harmless(model_always):
    r = run_model(model_always, CNR)
    n = check(r, 1, 1 << 30, max_escapes=int(r["stats"][0]) // 50)
    assert n > 240000


@pytest.mark.parametrize("lo,hi", [(1000, 21000), (300000, 325557)])
def test_cnr2000_subranges(model, lo, hi):
    r = run_model(model, CNR, lo, hi)
    check(r, 1, 1 << 30, max_escapes=2)


@pytest.mark.parametrize("kw", [dict(window=7, max_ref_count=3, min_interval=4, zeta_k=3), dict(window=7, max_ref_count=3, min_interval=2, zeta_k=5),
                                dict(window=0, max_ref_count=0, min_interval=0, zeta_k=1), dict(window=3, max_ref_count=8, min_interval=0, zeta_k=2),
                                dict(window=16, max_ref_count=30, min_interval=3, zeta_k=7)],
                         ids=lambda kw: "w%d_m%d_i%d_z%d" % (kw["window"], kw["max_ref_count"], kw["min_interval"], kw["zeta_k"]))
def test_synthetic_parameters(model, model_small, model_always, tmp_path_factory, kw):
    base, rowptr, succ = make_graph(tmp_path_factory, "sm", 60000, 1500000, 4242, 0.6, **kw)
    r = run_model(model, base)
    assert np.array_equal(r["succ"], succ)
    n = check(r, 1, 1 << 30, max_escapes=8)  # (chains that do not meet inside a piece: rarer than one in 10^5 pieces of 2 048 bits)
    assert n > 30000 and r["stats"][6] > 10  # a record of more than ten pieces
    r = run_model(model_always, base)
    check(r, 1, 1 << 30, max_escapes=int(r["stats"][0]) // 20)
    r = run_model(model_small, base, dmin=64)
    n = check(r, 64, 1 << 30, max_escapes=int(r["stats"][0]))  # (most chains do not meet within 128 bits of these codes: those records are the cooperative kernel's)
    assert n > 100
