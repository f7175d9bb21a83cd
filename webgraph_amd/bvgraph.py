"""Host-side mirror of the reference's graph API for the BVGraph decode path, over the libbvgpu C ABI.

Names, argument meaning and error behaviour follow it.unimi.dsi.webgraph.ImmutableGraph / BVGraph /
NodeIterator / LazyIntIterator (src/it/unimi/dsi/webgraph/ImmutableGraph.java:169-772, BVGraph.java,
NodeIterator.java:34-107, LazyIntIterator.java:28-43) so that parity tests read like the reference's own
(WebGraphTestCase.assertGraph).  All decoding happens in the HIP kernels behind include/bvgpu.h; there is no
CPU fallback here -- without the library or without a GPU, loading raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("BVGPU_LIB") or os.path.join(_HERE, "libbvgpu.so")  # BVGPU_LIB: tuning builds

BVG_OK, BVG_EARG, BVG_ESTATE, BVG_EUNSUPPORTED, BVG_EIO, BVG_ENOMEM, BVG_EHIP, BVG_EFORMAT, BVG_ECAP = 0, -1, -2, -3, -4, -5, -6, -7, -8
BVG_OUT_HOST, BVG_OUT_DEVICE, BVG_ASYNC = 0, 1, 2
BVG_FORMAT_BV, BVG_FORMAT_EF = 0, 1


class BvgInfo(C.Structure):
    _fields_ = [("nodes", C.c_int32), ("arcs", C.c_int64), ("window_size", C.c_int32), ("max_ref_count", C.c_int32),
                ("min_interval_length", C.c_int32), ("zeta_k", C.c_int32), ("flags", C.c_uint32),
                ("outdegree_coding", C.c_int32), ("block_coding", C.c_int32), ("residual_coding", C.c_int32),
                ("reference_coding", C.c_int32), ("block_count_coding", C.c_int32), ("offset_coding", C.c_int32),
                ("graph_bytes", C.c_uint64), ("device", C.c_int32), ("offsets_on_device", C.c_int32),
                ("shard_from", C.c_int32), ("shard_to", C.c_int32), ("staged_from", C.c_int32),
                ("format", C.c_int32), ("ef_upper_bound", C.c_int32), ("ef_log2_quantum", C.c_int32), ("ef_big_endian", C.c_int32)]


class BvgScanStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("nodes", "arcs", "loops", "dangling", "terminal", "num_gaps", "tot_gap", "tot_loc")] + [
        ("min_outdegree", C.c_int32), ("max_outdegree", C.c_int32), ("min_outdegree_node", C.c_int32), ("max_outdegree_node", C.c_int32),
        ("successor_delta_stats", C.c_uint64 * 32)]


class BvgStoreStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("written_bits", "offsets_bits", "bits_outdegrees", "bits_references", "bits_blocks", "bits_intervals", "bits_residuals",
                                           "copied_arcs", "intervalised_arcs", "residual_arcs", "tot_ref", "tot_dist")] + [
        ("max_ref_chain", C.c_int32), ("threads", C.c_int32), ("selection_rounds", C.c_int32), ("reserved", C.c_int32),
        ("successor_gap_bins", C.c_uint64 * 32), ("residual_gap_bins", C.c_uint64 * 32)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, t in self._fields_ if k != "reserved" and not k.endswith("_bins")}
        d["successor_gap_bins"] = [int(v) for v in self.successor_gap_bins]  # updateBins, BVGraph.java:1940-1944
        d["residual_gap_bins"] = [int(v) for v in self.residual_gap_bins]
        return d


class BvgCompressed(C.Structure):
    _fields_ = [("device", C.c_int32), ("reserved", C.c_int32), ("graph_dev", C.c_void_p), ("graph_bits", C.c_uint64), ("offsets_stream_dev", C.c_void_p),
                ("offsets_bits", C.c_uint64), ("bit_offsets_dev", C.c_void_p), ("stats", BvgStoreStats)]


class BvgLabelsInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_int32), ("nodes", C.c_int32), ("device", C.c_int32), ("labels_bytes", C.c_uint64),
                ("labels_bits", C.c_uint64), ("underlying", C.c_char * 1024), ("key", C.c_char * 128)]


EXPORTS = ["bvg_open", "bvg_open_shard", "bvg_clone", "bvg_close", "bvg_info", "bvg_last_error", "bvg_set_stream", "bvg_sync",
           "bvg_outdegrees", "bvg_decode_range", "bvg_decode_range_view", "bvg_host_alloc", "bvg_host_free", "bvg_scan_checksum", "bvg_equal_range", "bvg_scan_stats", "bvg_bfs_expand", "bvg_hyperball_step", "bvg_successors_batch", "bvg_csr_hashcode", "bvg_shard_bounds",
           "bvg_parse_properties", "bvg_flags_from_string", "bvg_decode_offsets_host", "bvg_decode_offsets_device", "bvg_labels_open", "bvg_labels_close", "bvg_labels_info",
           "bvg_labels_last_error", "bvg_labels_parse_properties", "bvg_labels_decode_range", "bvg_labels_decode_lists", "bvg_compress", "bvg_compressed_free", "bvg_compressed_copy", "bvg_store", "bvg_recompress", "bvg_store_ef", "bvg_recompress_ef", "bvg_cache_as_efgraph", "bvg_set_option", "bvg_set_profile", "bvg_get_profile", "bvg_debug_stats", "bvg_last_thresholds"]

_lib = None


def lib():
    """Loads libbvgpu.so (built in-tree by __graft_entry__.build()); raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise RuntimeError("libbvgpu.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7, and a second copy loaded next
        # to it sees no GPU.  Importing torch first makes libbvgpu's NEEDED libamdhip64.so.7 bind to that copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIBPATH)
        vp, i32, i64, u64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t
        L.bvg_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.bvg_open_shard.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.bvg_clone.argtypes = [vp, C.POINTER(vp)]
        L.bvg_close.argtypes = [vp]
        L.bvg_info.argtypes = [vp, C.POINTER(BvgInfo)]
        L.bvg_last_error.argtypes = [vp]
        L.bvg_last_error.restype = C.c_char_p
        L.bvg_set_stream.argtypes = [vp, vp]
        L.bvg_sync.argtypes = [vp, C.POINTER(u64)]
        L.bvg_outdegrees.argtypes = [vp, i32, i32, vp, C.c_int]
        L.bvg_decode_range.argtypes = [vp, i32, i32, vp, vp, sz, C.POINTER(u64), C.c_int]
        L.bvg_decode_range_view.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
        L.bvg_host_alloc.argtypes = [sz, C.POINTER(vp)]
        L.bvg_host_free.argtypes = [vp]
        L.bvg_host_free.restype = None
        L.bvg_scan_checksum.argtypes = [vp, i32, i32, C.POINTER(i32), C.POINTER(u64)]
        L.bvg_equal_range.argtypes = [vp, vp, C.c_int32, C.c_int32, C.POINTER(C.c_int)]
        L.bvg_scan_stats.argtypes = [vp, i32, i32, C.POINTER(BvgScanStats), vp]
        L.bvg_bfs_expand.argtypes = [vp, vp, sz, vp, i32, C.c_int, vp, sz, C.POINTER(u64)]
        L.bvg_hyperball_step.argtypes = [vp, i32, i32, C.c_int, vp, vp, vp, vp, C.POINTER(u64)]
        L.bvg_successors_batch.argtypes = [vp, vp, sz, vp, vp, sz, C.POINTER(u64), C.c_int]
        L.bvg_csr_hashcode.argtypes = [vp, i32, i32, vp, vp, C.POINTER(i32)]
        L.bvg_shard_bounds.argtypes = [vp, C.c_int, vp]
        L.bvg_parse_properties.argtypes = [C.c_char_p, C.POINTER(BvgInfo), C.c_char_p, sz]
        L.bvg_flags_from_string.argtypes = [C.c_char_p]
        L.bvg_flags_from_string.restype = i64
        L.bvg_decode_offsets_host.argtypes = [vp, sz, i32, C.c_int, vp]
        L.bvg_decode_offsets_device.argtypes = [C.c_int, vp, sz, i32, C.c_int, vp]
        L.bvg_labels_open.argtypes = [C.c_char_p, i32, C.c_int, C.POINTER(vp)]
        L.bvg_labels_close.argtypes = [vp]
        L.bvg_labels_close.restype = None
        L.bvg_labels_info.argtypes = [vp, C.POINTER(BvgLabelsInfo)]
        L.bvg_labels_last_error.argtypes = [vp]
        L.bvg_labels_last_error.restype = C.c_char_p
        L.bvg_labels_parse_properties.argtypes = [C.c_char_p, C.POINTER(BvgLabelsInfo), C.c_char_p, sz]
        L.bvg_labels_decode_range.argtypes = [vp, i32, i32, u64, vp, C.c_int]
        L.bvg_labels_decode_lists.argtypes = [vp, i32, i32, u64, vp, vp, u64, C.POINTER(u64), C.c_int]
        L.bvg_compress.argtypes = [C.c_int, i32, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(BvgCompressed), C.c_char_p, sz]
        L.bvg_compressed_free.argtypes = [C.POINTER(BvgCompressed)]
        L.bvg_compressed_free.restype = None
        L.bvg_compressed_copy.argtypes = [C.POINTER(BvgCompressed), i32, vp, vp, vp]
        L.bvg_store.argtypes = [C.c_char_p, C.c_int, i32, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(BvgStoreStats), C.c_char_p, sz]
        L.bvg_recompress.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(BvgStoreStats), C.c_char_p, sz]
        L.bvg_store_ef.argtypes = [C.c_char_p, C.c_int, i32, vp, vp, C.c_int, i32, C.c_int, C.c_int, C.c_char_p, sz]
        L.bvg_recompress_ef.argtypes = [vp, C.c_char_p, i32, C.c_int, C.c_int, C.c_char_p, sz]
        L.bvg_cache_as_efgraph.argtypes = [vp]
        L.bvg_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.bvg_set_profile.argtypes = [vp, C.c_int]
        L.bvg_get_profile.argtypes = [vp, C.POINTER(C.c_float)]
        L.bvg_last_thresholds.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.bvg_debug_stats.argtypes = [vp, vp, C.c_int]
        _lib = L
    return _lib


class BvgError(Exception):
    def __init__(self, code, msg=""):
        super().__init__("bvgpu error %d: %s" % (code, msg))
        self.code = code


def _raise(code, msg):
    """Maps a bvg_status to the exception class the reference throws at the same place (include/bvgpu.h)."""
    if code == BVG_EARG:
        raise ValueError(msg or "Node index out of range")                # IllegalArgumentException
    if code == BVG_ESTATE:
        raise RuntimeError(msg or "illegal state")                        # IllegalStateException
    if code == BVG_EUNSUPPORTED:
        raise NotImplementedError(msg or "unsupported")                   # UnsupportedOperationException
    if code == BVG_EIO:
        raise IOError(msg or "I/O error")                                 # IOException
    if code == BVG_ENOMEM:
        raise MemoryError(msg)
    raise BvgError(code, msg)


def _csr_ptrs(rowptr, succ):
    """(n, rowptr pointer, succ pointer, in_flags, keepalive): numpy arrays are host pointers, anything with data_ptr() (torch) a device pointer."""
    if hasattr(rowptr, "data_ptr"):
        return int(rowptr.numel()) - 1, rowptr.data_ptr(), succ.data_ptr(), BVG_OUT_DEVICE, (rowptr, succ)
    rp = np.ascontiguousarray(rowptr, dtype=np.int64)
    sc = np.ascontiguousarray(succ, dtype=np.int32)
    return rp.size - 1, rp.ctypes.data, sc.ctypes.data if sc.size else None, BVG_OUT_HOST, (rp, sc)


def store(rowptr, succ, basename, windowSize=7, maxRefCount=3, minIntervalLength=4, zetaK=3, flags=0, numberOfThreads=1, device=0):
    """BVGraph.store(graph, basename, windowSize, maxRefCount, minIntervalLength, zetaK, flags, numberOfThreads)
    (BVGraph.java:1679-1730) for a CSR graph, compressed on the GPU: numpy arrays (host) or torch tensors on `device`
    (int64 rowptr, int32 successors).  Returns the counters of the .properties file."""
    n, rp, sp, fl, keep = _csr_ptrs(rowptr, succ)
    st = BvgStoreStats()
    err = C.create_string_buffer(512)
    rc = lib().bvg_store(os.fsencode(basename), device, n, rp, sp, fl, windowSize, maxRefCount, minIntervalLength, zetaK, flags, numberOfThreads, C.byref(st), err, 512)
    del keep
    if rc:
        _raise(rc, err.value.decode("utf-8", "replace"))
    return st.as_dict()


def store_ef(rowptr, succ, basename, upperBound=None, log2Quantum=8, bigEndian=False, device=0):
    """EFGraph.store(graph, upperBound, basename, log2Quantum, cacheSize, byteOrder, pl) (EFGraph.java:812-889) for a CSR graph, on the GPU."""
    n, rp, sp, fl, keep = _csr_ptrs(rowptr, succ)
    err = C.create_string_buffer(512)
    rc = lib().bvg_store_ef(os.fsencode(basename), device, n, rp, sp, fl, 0 if upperBound is None else upperBound, log2Quantum, 1 if bigEndian else 0, err, 512)
    del keep
    if rc:
        _raise(rc, err.value.decode("utf-8", "replace"))


def compress(rowptr, succ, windowSize=7, maxRefCount=3, minIntervalLength=4, zetaK=3, flags=0, numberOfThreads=1, device=0):
    """The compression of store() alone: returns (graph bytes, offsets-file bytes, bit offsets int64[n+1], counters), copied to the host."""
    n, rp, sp, fl, keep = _csr_ptrs(rowptr, succ)
    c = BvgCompressed()
    err = C.create_string_buffer(512)
    rc = lib().bvg_compress(device, n, rp, sp, fl, windowSize, maxRefCount, minIntervalLength, zetaK, flags, numberOfThreads, C.byref(c), err, 512)
    del keep
    if rc:
        _raise(rc, err.value.decode("utf-8", "replace"))
    try:
        graph = np.empty((c.graph_bits + 7) // 8, dtype=np.uint8)
        offs = np.empty((c.offsets_bits + 7) // 8, dtype=np.uint8)
        bitoff = np.empty(n + 1, dtype=np.int64)
        rc = lib().bvg_compressed_copy(C.byref(c), n, graph.ctypes.data, offs.ctypes.data, bitoff.ctypes.data)
        if rc:
            _raise(rc, "copying the compressed streams back failed")
        return graph.tobytes(), offs.tobytes(), bitoff, c.stats.as_dict()
    finally:
        lib().bvg_compressed_free(C.byref(c))


def parse_properties(basename):
    """[host-only] BVGraph.loadInternal's property parsing (BVGraph.java:1528-1543)."""
    info = BvgInfo()
    buf = C.create_string_buffer(512)
    rc = lib().bvg_parse_properties(os.fsencode(basename), C.byref(info), buf, 512)
    if rc:
        _raise(rc, buf.value.decode("latin-1"))
    return info


def flags_from_string(s):
    """[host-only] BVGraph.string2Flags (BVGraph.java:1352-1366); IOError on an unknown constant name."""
    v = lib().bvg_flags_from_string(s.encode("latin-1"))
    if v < 0:
        raise IOError("Compression flag unknown in %r" % s)
    return int(v)


def decode_offsets_host(offset_bytes, nodes, coding=2):
    """[host-only] OffsetsLongIterator (BVGraph.java:907-935)."""
    b = np.frombuffer(offset_bytes, dtype=np.uint8)
    out = np.empty(nodes + 1, dtype=np.int64)
    rc = lib().bvg_decode_offsets_host(b.ctypes.data, b.size, nodes, coding, out.ctypes.data)
    if rc:
        _raise(rc, "cannot decode offsets")
    return out


def decode_offsets_device(offset_bytes, nodes, coding=2, device=0):
    """[device] the same decode by the GPU kernels (what BVGraph.load uses for gamma-coded offsets)."""
    b = np.frombuffer(offset_bytes, dtype=np.uint8)
    out = np.empty(nodes + 1, dtype=np.int64)
    rc = lib().bvg_decode_offsets_device(device, b.ctypes.data, b.size, nodes, coding, out.ctypes.data)
    if rc:
        _raise(rc, "cannot decode offsets on the device")
    return out


class LazyIntIterator:
    """LazyIntIterator.java:28-43 over a decoded array: increasing ids, then -1 forever."""

    def __init__(self, arr):
        self._a = arr
        self._i = 0

    def nextInt(self):
        if self._i >= len(self._a):
            return -1
        v = int(self._a[self._i])
        self._i += 1
        return v

    def skip(self, n):
        k = min(n, len(self._a) - self._i)
        self._i += k
        return k


class NodeIterator:
    """BVGraphNodeIterator (BVGraph.java:1136-1281): sequential scan served from GPU-decoded batches."""

    def __init__(self, graph, from_, upper_bound=2**31 - 1, batch_nodes=1 << 20, owns_graph=False):
        n = graph.numNodes()
        if from_ < 0 or from_ > n:
            raise ValueError("Node index out of range: %d" % from_)   # BVGraph.java:1165
        self._g = graph
        self._owns = owns_graph  # a copy()/split iterator decodes through a flyweight handle of its own (bvg_clone)
        self._from = from_
        self._curr = from_ - 1
        self._limit = min(upper_bound, n) - 1                         # hasNextLimit, BVGraph.java:1185
        self._batch = batch_nodes
        self._lo = self._hi = from_
        self._rowptr = None
        self._succ = None

    def hasNext(self):
        return self._curr < self._limit                               # BVGraph.java:1216

    def close(self):
        if self._owns and self._g is not None:
            self._g.close()
            self._owns = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def nextInt(self):
        if not self.hasNext():
            raise StopIteration                                       # NoSuchElementException
        self._curr += 1
        if self._curr >= self._hi:
            self._lo = self._curr
            self._hi = min(self._lo + self._batch, self._limit + 1)
            self._rowptr, self._succ = self._g.decode_range(self._lo, self._hi)
        return self._curr

    def _row(self):
        if self._curr == self._from - 1:
            raise RuntimeError("nextInt() not called yet")            # IllegalStateException, BVGraph.java:1220
        i = self._curr - self._lo
        return self._succ[self._rowptr[i]:self._rowptr[i + 1]]

    def outdegree(self):
        return len(self._row())

    def successorArray(self):
        return self._row()

    def successors(self):
        return LazyIntIterator(self._row())

    def copy(self, upper_bound=2**31 - 1):
        """NodeIterator.copy(upperBound) (BVGraph.java:1253-1260): same position, never returns nodes >= upperBound.
        The reference's copy owns a bit stream of its own and may be drained by another thread (BVGraph.java:2471-2477,
        ImmutableGraph.java:379-409); here it owns a flyweight handle (bvg_clone): a bvg_t is not thread-safe."""
        return NodeIterator(self._g.copy(), self._curr + 1, upper_bound, self._batch, owns_graph=True)


class _EmptyNodeIterator:
    """NodeIterator.EMPTY"""

    def hasNext(self):
        return False

    def nextInt(self):
        raise StopIteration


class BVGraph:
    """ImmutableGraph / BVGraph surface backed by libbvgpu (one HIP device)."""

    def __init__(self, handle, basename):
        self._h = handle
        self._basename = basename
        info = BvgInfo()
        rc = lib().bvg_info(self._h, C.byref(info))
        if rc:
            _raise(rc, "bvg_info")
        self.info = info

    # -- loaders (ImmutableGraph.load / loadMapped / loadOffline all stage the graph in HBM here)
    @classmethod
    def load(cls, basename, device=0):
        h = C.c_void_p()
        rc = lib().bvg_open(os.fsencode(basename), device, C.byref(h))
        if rc:
            msg = lib().bvg_last_error(h).decode("latin-1") if h else ""
            if h:
                lib().bvg_close(h)
            _raise(rc, msg)
        return cls(h, basename)

    loadMapped = load
    loadOffline = load

    @classmethod
    def load_shard(cls, basename, part, parts, device=0):
        """One GPU's share of a graph scanned by `parts` GPUs (bvg_open_shard): only the slice of the bit stream and of the
        offset table that nodes [shard_bounds(parts)[part], ...[part+1]) need is staged; info.shard_from / shard_to say which."""
        h = C.c_void_p()
        rc = lib().bvg_open_shard(os.fsencode(basename), device, part, parts, C.byref(h))
        if rc:
            msg = lib().bvg_last_error(h).decode("latin-1") if h else ""
            if h:
                lib().bvg_close(h)
            _raise(rc, msg)
        return cls(h, basename)

    def close(self):
        if self._h:
            lib().bvg_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            _raise(rc, lib().bvg_last_error(self._h).decode("latin-1"))

    # -- ImmutableGraph.java:254-268, BVGraph.java:591-620
    def numNodes(self):
        return self.info.nodes

    def numArcs(self):
        return self.info.arcs

    def randomAccess(self):
        return True

    def hasCopiableIterators(self):
        return True

    def basename(self):
        return self._basename

    def windowSize(self):
        return self.info.window_size

    def maxRefCount(self):
        return self.info.max_ref_count

    def copy(self):
        """BVGraph.copy() (BVGraph.java:552-577): flyweight sharing the staged graph."""
        h = C.c_void_p()
        rc = lib().bvg_clone(self._h, C.byref(h))
        if rc:
            if h:
                lib().bvg_close(h)
            _raise(rc, "bvg_clone")
        return BVGraph(h, self._basename)

    # -- the hot path
    def outdegrees(self, lo=0, hi=None):
        hi = self.numNodes() if hi is None else hi
        out = np.empty(max(hi - lo, 0), dtype=np.int32)
        self._check(lib().bvg_outdegrees(self._h, lo, hi, out.ctypes.data, BVG_OUT_HOST))
        return out

    def outdegree(self, x):
        if x < 0 or x >= self.numNodes():
            raise ValueError("Node index out of range: %d" % x)       # BVGraph.java:860
        return int(self.outdegrees(x, x + 1)[0])

    def decode_range(self, lo=0, hi=None):
        """CSR of nodes [lo,hi): (rowptr int64[hi-lo+1], succ int32[arcs]) in host memory."""
        hi = self.numNodes() if hi is None else hi
        rp, sc = self.decode_range_view(lo, hi)
        return rp.copy(), sc.copy()

    def decode_range_view(self, lo=0, hi=None):
        """One call, no counting call first: (rowptr, succ) as views of the handle's pinned result buffers, valid until
        the next call on this graph (bvg_decode_range_view)."""
        hi = self.numNodes() if hi is None else hi
        rp, sp, arcs = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
        self._check(lib().bvg_decode_range_view(self._h, lo, hi, C.byref(rp), C.byref(sp), C.byref(arcs)))
        rowptr = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_int64)), shape=(max(hi - lo, 0) + 1,))
        succ = np.ctypeslib.as_array(C.cast(sp, C.POINTER(C.c_int32)), shape=(max(arcs.value, 1),))[:arcs.value]
        return rowptr, succ

    def decode_range_into(self, lo, hi, rowptr, succ):
        """bvg_decode_range with BVG_OUT_HOST into caller-owned numpy arrays (pageable or pinned); returns the arc count."""
        arcs = C.c_uint64(0)
        self._check(lib().bvg_decode_range(self._h, lo, hi, rowptr.ctypes.data, succ.ctypes.data if succ is not None else None,
                                           succ.size if succ is not None else 0, C.byref(arcs), BVG_OUT_HOST))
        return arcs.value

    def scan_checksum(self, lo=0, hi=None, h=-1):
        """(hash, arcs) of nodes [lo, hi): ImmutableGraph.hashCode() continued from h, nothing materialised for the caller."""
        hi = self.numNodes() if hi is None else hi
        hh, arcs = C.c_int32(h), C.c_uint64(0)
        self._check(lib().bvg_scan_checksum(self._h, lo, hi, C.byref(hh), C.byref(arcs)))
        return hh.value, arcs.value

    def store(self, basename, windowSize=7, maxRefCount=3, minIntervalLength=4, zetaK=3, flags=0, numberOfThreads=1):
        """BVGraph.store(this, basename, ...) (BVGraph.java:1679-1730): decode and recompress without leaving the device."""
        st = BvgStoreStats()
        err = C.create_string_buffer(512)
        rc = lib().bvg_recompress(self._h, os.fsencode(basename), windowSize, maxRefCount, minIntervalLength, zetaK, flags, numberOfThreads, C.byref(st), err, 512)
        if rc:
            _raise(rc, err.value.decode("utf-8", "replace"))
        return st.as_dict()

    def cache_as_efgraph(self):
        """Re-encodes the handle's lists as an EFGraph image in HBM and decodes from it from now on (bvg_cache_as_efgraph): same lists, faster."""
        self._check(lib().bvg_cache_as_efgraph(self._h))
        rc = lib().bvg_info(self._h, C.byref(self.info))
        if rc:
            _raise(rc, "bvg_info")

    def store_ef(self, basename, upperBound=None, log2Quantum=8, bigEndian=False):
        """EFGraph.store(this, basename, ...): decode and re-encode as an EFGraph without leaving the device."""
        err = C.create_string_buffer(512)
        rc = lib().bvg_recompress_ef(self._h, os.fsencode(basename), 0 if upperBound is None else upperBound, log2Quantum, 1 if bigEndian else 0, err, 512)
        if rc:
            _raise(rc, err.value.decode("utf-8", "replace"))

    def decode_range_device(self, lo, hi, rowptr_ptr, succ_ptr, succ_cap, asynchronous=False):
        """Device-pointer form: rowptr_ptr / succ_ptr are raw device addresses (e.g. torch.Tensor.data_ptr())."""
        arcs = C.c_uint64(0)
        fl = BVG_OUT_DEVICE | (BVG_ASYNC if asynchronous else 0)
        self._check(lib().bvg_decode_range(self._h, lo, hi, rowptr_ptr, succ_ptr, succ_cap, C.byref(arcs), fl))
        return arcs.value

    def sync(self):
        arcs = C.c_uint64(0)
        self._check(lib().bvg_sync(self._h, C.byref(arcs)))
        return arcs.value

    def set_stream(self, hip_stream):
        self._check(lib().bvg_set_stream(self._h, hip_stream))

    PHASES = ("headers", "scan", "lists", "parse_giant", "parse_big", "parse_short", "copy", "tail")

    def set_option(self, name, value):
        """A tuning / debug knob of this handle (bvg_set_option): the BVGPU_<NAME> environment variables are only read when a handle is created."""
        self._check(lib().bvg_set_option(self._h, str(name).encode(), str(value).encode()))

    def set_profile(self, on):
        self._check(lib().bvg_set_profile(self._h, 1 if on else 0))

    def get_profile(self):
        """Per-phase milliseconds of the last profiled range decode (HIP events on the decode stream)."""
        ms = (C.c_float * len(self.PHASES))()
        self._check(lib().bvg_get_profile(self._h, ms))
        return dict(zip(self.PHASES, [float(x) for x in ms]))

    def last_thresholds(self):
        """(coop_min, giant_min) of the last range decode: outdegrees from which a record is decoded by a wave / a group of waves."""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(lib().bvg_last_thresholds(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def debug_stats(self, reset=True):
        out = np.zeros(64, dtype=np.uint64)
        self._check(lib().bvg_debug_stats(self._h, out.ctypes.data, 1 if reset else 0))
        return out

    def scan_stats(self, lo=0, hi=None, indegree_ptr=None):
        """The scan of Stats.run (Stats.java:111-160) over nodes [lo, hi) on the device; indegree_ptr: device int32[n] or None."""
        hi = self.numNodes() if hi is None else hi
        st = BvgScanStats()
        self._check(lib().bvg_scan_stats(self._h, lo, hi, C.byref(st), indegree_ptr))
        d = {k: getattr(st, k) for k, _ in BvgScanStats._fields_ if k != "successor_delta_stats"}
        d["successor_delta_stats"] = list(st.successor_delta_stats)
        return d

    def hyperball_step(self, log2m, regs_in_ptr, regs_out_ptr, modified_in_ptr, modified_out_ptr, lo=0, hi=None):
        """One standard iteration of HyperBall over nodes [lo, hi) (HyperBall.java:875-915) on the device: register-wise maximum of every
        node's counter with its successors' (device pointers; modified_in_ptr None: every counter counts); returns the number of counters that changed."""
        hi = self.numNodes() if hi is None else hi
        cnt = C.c_uint64(0)
        self._check(lib().bvg_hyperball_step(self._h, lo, hi, log2m, regs_in_ptr, regs_out_ptr, modified_in_ptr, modified_out_ptr, C.byref(cnt)))
        return cnt.value

    def bfs_expand(self, frontier_ptr, q, marker_ptr, round_, parent, out_ptr, out_cap):
        """One round of ParallelBreadthFirstVisit (device pointers); returns the size of the next frontier."""
        cnt = C.c_uint64(0)
        self._check(lib().bvg_bfs_expand(self._h, frontier_ptr, q, marker_ptr, round_, 1 if parent else 0, out_ptr, out_cap, C.byref(cnt)))
        return cnt.value

    def bfs(self, start, parent=False, round_=0, marker=None):
        """ParallelBreadthFirstVisit.visit(start) (ParallelBreadthFirstVisit.java:205-247) with the frontier expansion on the
        device: returns (queue, cutPoints, marker) -- the nodes of queue[cutPoints[d]:cutPoints[d+1]] are at distance d."""
        import torch
        n = self.numNodes()
        dev = torch.device("cuda", self.info.device)
        if marker is None:
            marker = torch.full((n,), -1, dtype=torch.int32, device=dev)
        if int(marker[start]) != -1:
            return torch.empty(0, dtype=torch.int32, device=dev), [0], marker
        marker[start] = start if parent else round_
        queue = torch.empty(n, dtype=torch.int32, device=dev)
        queue[0] = start
        cut = [0, 1]
        while cut[-1] > cut[-2]:
            lo, hi = cut[-2], cut[-1]
            got = self.bfs_expand(queue.data_ptr() + 4 * lo, hi - lo, marker.data_ptr(), round_, parent, queue.data_ptr() + 4 * hi, n - hi)
            cut.append(hi + got)
        return queue[:cut[-1]], cut[:-1], marker

    def successors_batch(self, nodes):
        """Concatenated successorArray(nodes[i]) (random access, BVGraph.java:897-904)."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        rowptr = np.empty(nodes.size + 1, dtype=np.int64)
        arcs = C.c_uint64(0)
        self._check(lib().bvg_successors_batch(self._h, nodes.ctypes.data, nodes.size, rowptr.ctypes.data, None, 0, C.byref(arcs), BVG_OUT_HOST))
        succ = np.empty(max(arcs.value, 1), dtype=np.int32)
        self._check(lib().bvg_successors_batch(self._h, nodes.ctypes.data, nodes.size, rowptr.ctypes.data, succ.ctypes.data, succ.size, C.byref(arcs), BVG_OUT_HOST))
        return rowptr, succ[:arcs.value]

    def successorArray(self, x):
        if x < 0 or x >= self.numNodes():
            raise ValueError("Node index out of range: %d" % x)       # BVGraph.java:900
        rp, sc = self.successors_batch(np.array([x], dtype=np.int32))
        return sc

    def successors(self, x):
        return LazyIntIterator(self.successorArray(x))

    def nodeIterator(self, from_=0):
        return NodeIterator(self, from_)

    def splitNodeIterators(self, how_many):
        """ImmutableGraph.splitNodeIterators (ImmutableGraph.java:379-409), random-access branch."""
        n = self.numNodes()
        if n == 0 and how_many == 0:
            return []
        if how_many < 1:
            raise ValueError("howMany < 1")
        m = -(-n // how_many)
        res = []
        f = 0
        while f < n:
            res.append(self.nodeIterator(f).copy(f + m))
            f += m
        res += [_EmptyNodeIterator()] * (how_many - len(res))
        return res

    def shard_bounds(self, parts):
        b = np.empty(parts + 1, dtype=np.int32)
        self._check(lib().bvg_shard_bounds(self._h, parts, b.ctypes.data))
        return b

    def csr_hashcode(self, lo, hi, rowptr_ptr, succ_ptr, h=-1):
        hh = C.c_int32(h)
        self._check(lib().bvg_csr_hashcode(self._h, lo, hi, rowptr_ptr, succ_ptr, C.byref(hh)))
        return hh.value

    def hashCode(self):
        """ImmutableGraph.hashCode() (ImmutableGraph.java:757-770): a checksum scan on the device (bvg_scan_checksum)."""
        return self.scan_checksum(0, self.numNodes(), -1)[0]

    def equal_range(self, other, lo=0, hi=None):
        """bvg_equal_range: do both handles give every node of [lo, hi) the same successors?  Compared on the device."""
        hi = self.numNodes() if hi is None else hi
        eq = C.c_int(0)
        self._check(lib().bvg_equal_range(self._h, other._h, lo, hi, C.byref(eq)))
        return bool(eq.value)

    def equals(self, other):
        """ImmutableGraph.equals() (ImmutableGraph.java:731-749): same number of nodes and the same successor list for
        every node; both graphs are scanned in lock step, a batch of nodes at a time."""
        if not hasattr(other, "numNodes") or not hasattr(other, "decode_range"):
            return False
        n = self.numNodes()
        if n != other.numNodes():
            return False
        if isinstance(other, BVGraph) and getattr(other, "_h", None) and self._h:  # two handles of this library: compared on the device (bvg_equal_range)
            eq = C.c_int(0)
            rc = lib().bvg_equal_range(self._h, other._h, 0, n, C.byref(eq))
            if rc == 0:
                return bool(eq.value)
            if rc != BVG_EARG:  # (BVG_EARG: e.g. handles on different devices -- the host comparison below serves them)
                self._check(rc)
        step = 1 << 22
        for lo in range(0, n, step):
            hi = min(lo + step, n)
            rp1, sc1 = self.decode_range(lo, hi)
            rp2, sc2 = other.decode_range(lo, hi)
            if not (np.array_equal(rp1, rp2) and np.array_equal(sc1, sc2)):
                return False
        return True


class EFGraph(BVGraph):
    """EFGraph (src/it/unimi/dsi/webgraph/EFGraph.java), the quasi-succinct second format, behind the same handle: every
    method of BVGraph applies.  load() insists that the files are an EFGraph, as EFGraph.loadInternal does (:716-718)."""

    @classmethod
    def load(cls, basename, device=0):
        g = super().load(basename, device)
        if g.info.format != BVG_FORMAT_EF:
            g.close()
            raise IOError("This class (EFGraph) cannot load a graph stored using another class")
        return g

    loadMapped = load
    loadOffline = load

    def upperBound(self):
        return self.info.ef_upper_bound


class ArcLabelledBVGraph:
    """BitStreamArcLabelledImmutableGraph over a BVGraph (labelling/BitStreamArcLabelledImmutableGraph.java:383-470), int
    labels (GammaCodedIntLabel / FixedWidthIntLabel) and int-list labels (FixedWidthIntListLabel): the underlying graph and the
    label stream both live in HBM."""

    def __init__(self, graph, handle, basename):
        self.graph = graph
        self._h = handle
        self._basename = basename
        self.info = BvgLabelsInfo()
        rc = lib().bvg_labels_info(self._h, C.byref(self.info))
        if rc:
            _raise(rc, "bvg_labels_info")

    @classmethod
    def load(cls, basename, device=0):
        info = BvgLabelsInfo()
        err = C.create_string_buffer(512)
        rc = lib().bvg_labels_parse_properties(os.fsencode(basename), C.byref(info), err, 512)
        if rc:
            _raise(rc, err.value.decode("utf-8", "replace"))
        g = BVGraph.load(os.fsdecode(info.underlying), device=device)
        h = C.c_void_p()
        rc = lib().bvg_labels_open(os.fsencode(basename), g.numNodes(), device, C.byref(h))
        if rc:
            msg = lib().bvg_labels_last_error(h).decode("utf-8", "replace") if h else ""
            if h:
                lib().bvg_labels_close(h)
            g.close()
            _raise(rc, msg)
        return cls(g, h, basename)

    def close(self):
        if self._h:
            lib().bvg_labels_close(self._h)
            self._h = None
        if self.graph is not None:
            self.graph.close()
            self.graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def numNodes(self):
        return self.graph.numNodes()

    def decode_range(self, lo=0, hi=None):
        """(rowptr, successors, labels) of nodes [lo, hi): labels[k] belongs to arc k of the CSR."""
        hi = self.numNodes() if hi is None else hi
        rowptr, succ = self.graph.decode_range(lo, hi)
        labels = np.empty(max(succ.size, 1), dtype=np.int32)
        rc = lib().bvg_labels_decode_range(self._h, lo, hi, int(rowptr[-1]), labels.ctypes.data, BVG_OUT_HOST)
        if rc:
            _raise(rc, lib().bvg_labels_last_error(self._h).decode("utf-8", "replace"))
        return rowptr, succ, labels[:succ.size]

    def decode_label_lists(self, lo=0, hi=None):
        """(rowptr, successors, listptr, values) of nodes [lo, hi) for FixedWidthIntListLabel: arc k carries values[listptr[k]:listptr[k+1]]."""
        hi = self.numNodes() if hi is None else hi
        rowptr, succ = self.graph.decode_range(lo, hi)
        arcs = int(rowptr[-1])
        listptr = np.empty(arcs + 1, dtype=np.int64)
        nv = C.c_uint64(0)
        rc = lib().bvg_labels_decode_lists(self._h, lo, hi, arcs, listptr.ctypes.data, None, 0, C.byref(nv), BVG_OUT_HOST)  # sizing call
        if rc not in (0, -8):
            _raise(rc, lib().bvg_labels_last_error(self._h).decode("utf-8", "replace"))
        values = np.empty(max(nv.value, 1), dtype=np.int32)
        if nv.value:
            rc = lib().bvg_labels_decode_lists(self._h, lo, hi, arcs, listptr.ctypes.data, values.ctypes.data, values.size, C.byref(nv), BVG_OUT_HOST)
            if rc:
                _raise(rc, lib().bvg_labels_last_error(self._h).decode("utf-8", "replace"))
        return rowptr, succ, listptr, values[:nv.value]

    def decode_labels_device(self, lo, hi, arcs, labels_ptr):
        rc = lib().bvg_labels_decode_range(self._h, lo, hi, int(arcs), labels_ptr, BVG_OUT_DEVICE)
        if rc:
            _raise(rc, lib().bvg_labels_last_error(self._h).decode("utf-8", "replace"))
