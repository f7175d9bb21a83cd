"""ctypes front-end of libbvgtools.so: the CPU BVGraph writer and the seeded synthetic generator.

Host-only (no GPU).  See include/bvgtools.h for the C ABI and the reference lines it follows.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libbvgtools.so")
_SRC = os.path.join(_HERE, "csrc", "host", "bvg_tools.cpp")

# BVGraph flag constants (BVGraph.java:475-523): coding id << (4 * field)
DELTA, GAMMA, GOLOMB, SKEWED_GOLOMB, UNARY, ZETA, NIBBLE = 1, 2, 3, 4, 5, 6, 7
OUTDEGREES, BLOCKS, RESIDUALS, REFERENCES, BLOCK_COUNT, OFFSETS = 0, 4, 8, 12, 16, 20


def flag(field_shift, coding):
    return coding << field_shift


class StoreStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "written_bits", "offsets_bits", "bits_outdegrees", "bits_references", "bits_blocks", "bits_intervals",
        "bits_residuals", "copied_arcs", "intervalised_arcs", "residual_arcs", "tot_ref", "tot_dist")] + [
        ("max_ref_chain", C.c_int32), ("threads", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", _LIB, _SRC])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError("libbvgtools.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(_LIB)
        L.bvt_store.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_uint32, C.c_int, C.POINTER(StoreStats)]
        L.bvt_generate.argtypes = [C.c_int32, C.c_int64, C.c_uint64, C.c_double, C.c_int,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.bvt_generate_ex.argtypes = [C.c_int32, C.c_int64, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_int,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.bvt_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def store(basename, rowptr, succ, window=7, max_ref_count=3, min_interval=4, zeta_k=3, flags=0, threads=1):
    """BVGraph.store(g, basename, window, maxRefCount, minIntervalLength, zetaK, flags) for a CSR graph."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    succ = np.ascontiguousarray(succ, dtype=np.int32)
    n = rowptr.size - 1
    st = StoreStats()
    rc = lib().bvt_store(os.fsencode(basename), n, rowptr.ctypes.data, succ.ctypes.data, window, max_ref_count,
                         min_interval, zeta_k, flags, threads, C.byref(st))
    if rc:
        raise OSError(-rc, "bvt_store failed: %s" % os.strerror(-rc))
    return st.as_dict()


def generate(n, m, seed=0x5EEDB5E70001, p_copy=0.5, threads=None, p_same=0.0, p_keep=0.7):
    """Seeded power-law / copy-model graph; returns (rowptr int64[n+1], succ int32[m]).
    p_same / p_keep: the C5 knobs (runs of equal outdegrees copying from their predecessor; include/bvgtools.h)."""
    threads = threads or os.cpu_count() or 1
    rp, sp = C.c_void_p(), C.c_void_p()
    rc = lib().bvt_generate_ex(n, m, seed, p_copy, p_same, p_keep, threads, C.byref(rp), C.byref(sp))
    if rc:
        raise OSError(-rc, "bvt_generate failed: %s" % os.strerror(-rc))
    try:
        rowptr = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
        succ = np.ctypeslib.as_array(C.cast(sp, C.POINTER(C.c_int32)), shape=(max(m, 1),))[:m].copy()
    finally:
        lib().bvt_free(rp)
        lib().bvt_free(sp)
    return rowptr, succ


def random_nodes(n, count, seed=0x5EEDB5E70004):
    """SpeedTest's random-access ids (xoroshiro128+ re-seeded per repetition, SpeedTest.java:98-111): int32[count]."""
    out = np.empty(max(count, 1), dtype=np.int32)
    L = lib()
    L.bvt_random_nodes.argtypes = [C.c_uint64, C.c_int32, C.c_int64, C.c_void_p]
    rc = L.bvt_random_nodes(seed, n, count, out.ctypes.data)
    if rc:
        raise OSError(-rc, "bvt_random_nodes failed")
    return out[:count]


def store_labels(basename, underlying, rowptr, labels, kind="gamma", width=0, key="FOO"):
    """BitStreamArcLabelledImmutableGraph.store for int labels given per arc in CSR order (include/bvgtools.h)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    L = lib()
    L.bvt_store_labels.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_char_p]
    rc = L.bvt_store_labels(os.fsencode(basename), os.fsencode(underlying), rowptr.size - 1, rowptr.ctypes.data, labels.ctypes.data,
                            1 if kind == "gamma" else 2, width, key.encode("ascii"))
    if rc:
        raise OSError(-rc, "bvt_store_labels failed: %s" % os.strerror(-rc))


def store_label_lists(basename, underlying, rowptr, listptr, values, width, key="FOO"):
    """BitStreamArcLabelledImmutableGraph.store for FixedWidthIntListLabel: arc a carries values[listptr[a]:listptr[a+1]]."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    listptr = np.ascontiguousarray(listptr, dtype=np.int64)
    values = np.ascontiguousarray(values, dtype=np.int32)
    L = lib()
    L.bvt_store_label_lists.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p]
    rc = L.bvt_store_label_lists(os.fsencode(basename), os.fsencode(underlying), rowptr.size - 1, rowptr.ctypes.data, listptr.ctypes.data,
                                 values.ctypes.data, int(width), key.encode())
    if rc:
        raise OSError(-rc, "bvt_store_label_lists failed: %s" % os.strerror(-rc))


def store_ef(basename, rowptr, succ, upper_bound=None, log2_quantum=8, big_endian=False):
    """EFGraph.store(graph, basename) for a CSR graph (include/bvgtools.h): the quasi-succinct second format."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    succ = np.ascontiguousarray(succ, dtype=np.int32)
    n = rowptr.size - 1
    L = lib()
    L.bvt_store_ef.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int]
    rc = L.bvt_store_ef(os.fsencode(basename), n, rowptr.ctypes.data, succ.ctypes.data, n if upper_bound is None else upper_bound, log2_quantum, 1 if big_endian else 0)
    if rc:
        raise OSError(-rc, "bvt_store_ef failed: %s" % os.strerror(-rc))
