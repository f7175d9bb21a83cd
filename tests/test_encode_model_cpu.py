"""The device compressor's per-node logic (webgraph_amd/csrc/bv_encode.hpp: pair costs, chunked selection of the reference,
record emission with ORed words, the .offsets stream) compiled for the host and compared, byte for byte, with the CPU
writer -- and, on cnr-2000, with the files the reference itself produced."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import CNR, ROOT

FLAG = {"outd": 0, "blk": 4, "res": 8, "ref": 12, "bc": 16, "off": 20}
DELTA, GAMMA, GOLOMB, UNARY, ZETA, NIBBLE = 1, 2, 3, 5, 6, 7


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("encmodel") / "encode_model.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "cpp", "encode_model.cpp")])
    L = C.CDLL(so)
    L.bve_model_compress.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                     C.POINTER(C.c_int32)]
    return L


def codings_of(flags):
    d = [GAMMA, GAMMA, ZETA, UNARY, GAMMA, GAMMA]  # outdegree, block, residual, reference, block count, offset
    for i, k in enumerate(["outd", "blk", "res", "ref", "bc", "off"]):
        if (flags >> FLAG[k]) & 0xF:
            d[i] = (flags >> FLAG[k]) & 0xF
    return np.array(d, dtype=np.int32)


def run_model(L, rowptr, succ, W, R, I, K, flags=0, parts=1, chunk=64, span=16):
    n = rowptr.size - 1
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    succ = np.ascontiguousarray(succ, dtype=np.int32)
    cod = codings_of(flags)
    cap = int(succ.size) * 3 + n + 1024
    words = np.zeros(cap, dtype=np.uint32)
    owords = np.zeros(n * 3 + 1024, dtype=np.uint32)
    off = np.zeros(n + 1, dtype=np.int64)
    bits, obits, rounds = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
    stats = np.zeros(11, dtype=np.uint64)
    per = (n + parts - 1) // parts if n else 0
    rc = L.bve_model_compress(n, rowptr.ctypes.data, succ.ctypes.data, W, R, I, K, cod.ctypes.data, per, chunk, span, words.ctypes.data, words.size, C.byref(bits),
                              off.ctypes.data, owords.ctypes.data, owords.size, C.byref(obits), stats.ctypes.data, C.byref(rounds))
    assert rc == 0, rc
    return words.tobytes()[:(bits.value + 7) // 8], owords.tobytes()[:(obits.value + 7) // 8], off, stats, rounds.value


def test_model_reproduces_reference_bytes(model, cnr_oracle):
    _, rowptr, succ = cnr_oracle
    graph, offs, off, stats, rounds = run_model(model, rowptr, succ, 7, 3, 3, 3)
    assert graph == open(CNR + ".graph", "rb").read()
    assert offs == open(CNR + ".offsets", "rb").read()
    assert (int(stats[5]), int(stats[6]), int(stats[7])) == (2130833, 361894, 723425) and int(stats[10]) == 3
    assert 2 <= rounds <= 16
    assert run_model(model, rowptr, succ, 7, 3, 3, 3, chunk=64, span=1)[:2] == (graph, offs)


@pytest.mark.parametrize("W,R,I,K,flags,parts,chunk", [
    (7, 3, 4, 3, 0, 1, 64), (7, 3, 4, 3, 0, 3, 64), (7, 3, 4, 3, 0, 1, 5), (0, 0, 0, 3, 0, 1, 64), (1, 1, 2, 3, 0, 1, 64), (3, 100, 1, 2, 0, 2, 16),
    (7, 3, 0, 3, 0, 1, 64), (16, 2, 3, 5, 0, 1, 8),
    (7, 3, 4, 3, (DELTA << FLAG["outd"]) | (DELTA << FLAG["blk"]) | (DELTA << FLAG["res"]) | (GAMMA << FLAG["ref"]) | (DELTA << FLAG["bc"]) | (DELTA << FLAG["off"]), 1, 64),
    (7, 3, 4, 3, (UNARY << FLAG["blk"]) | (NIBBLE << FLAG["res"]) | (DELTA << FLAG["ref"]) | (UNARY << FLAG["bc"]), 1, 64),
    (7, 3, 4, 5, (GOLOMB << FLAG["res"]), 1, 64), (7, 3, 4, 3, (GAMMA << FLAG["res"]), 2, 64)])
def test_model_matches_cpu_writer(tmp_path, model, W, R, I, K, flags, parts, chunk):
    from webgraph_amd import tools as T
    rowptr, succ = T.generate(6000, 90000, seed=17 + W + chunk, p_copy=0.7, threads=2)
    base = str(tmp_path / "g")
    st = T.store(base, rowptr, succ, window=W, max_ref_count=R, min_interval=I, zeta_k=K, flags=flags, threads=parts)
    graph, offs, off, stats, rounds = run_model(model, rowptr, succ, W, R, I, K, flags, parts, chunk)
    assert graph == open(base + ".graph", "rb").read()
    assert offs == open(base + ".offsets", "rb").read()
    keys = ["bits_outdegrees", "bits_references", "bits_blocks", "bits_intervals", "bits_residuals", "copied_arcs", "intervalised_arcs", "residual_arcs", "tot_ref", "tot_dist",
            "max_ref_chain"]
    assert [int(v) for v in stats] == [int(st[k]) for k in keys]


def test_model_edge_rows(model, tmp_path):
    """Empty graph, empty rows, one giant row, rows that are one interval, a row equal to its predecessor."""
    from webgraph_amd import tools as T
    rows = [[], [1, 2, 3, 4, 5, 6], [1, 2, 3, 4, 5, 6], [], list(range(0, 3000, 3)), list(range(0, 3000, 3)) + [5000], [0], [7], list(range(10, 2000)), []]
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    succ = np.array([v for r in rows for v in r], dtype=np.int32)
    for W, R, I in [(7, 3, 4), (2, 1, 2), (0, 0, 0)]:
        base = str(tmp_path / ("e%d" % W))
        T.store(base, rowptr, succ, window=W, max_ref_count=R, min_interval=I)
        graph, offs, _, _, _ = run_model(model, rowptr, succ, W, R, I, 3, chunk=4)
        assert graph == open(base + ".graph", "rb").read() and offs == open(base + ".offsets", "rb").read()
    graph, offs, _, _, _ = run_model(model, np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), 7, 3, 4, 3)
    assert graph == b"" and offs == b"\x80"  # gamma(0)


def test_model_chains_that_never_forget(model, tmp_path):
    """Identical rows: every chunk-boundary guess is wrong; the rounds must still end at the sequential result."""
    from webgraph_amd import tools as T
    n = 5000
    rowptr = np.arange(n + 1, dtype=np.int64) * 10
    succ = np.tile(np.arange(5, 105, 10, dtype=np.int32), n)
    base = str(tmp_path / "same")
    T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=4, threads=1)
    for chunk, span in [(64, 16), (4, 2), (7, 1)]:
        graph, offs, _, _, rounds = run_model(model, rowptr, succ, 7, 3, 4, 3, chunk=chunk, span=span)
        assert graph == open(base + ".graph", "rb").read() and offs == open(base + ".offsets", "rb").read()
        assert rounds > 2
