"""ctypes front-end of the CPU parity oracle (oracle/bvg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by anything under webgraph_amd/.  See the header of bvg_oracle.c for the parity status
(pinned by the reference's cnr-2000 fixture for the default codings).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libbvgoracle.so")

# CompressionFlags.java:26-44
DELTA, GAMMA, GOLOMB, SKEWED_GOLOMB, UNARY, ZETA, NIBBLE = 1, 2, 3, 4, 5, 6, 7
_CODING = {"DELTA": DELTA, "GAMMA": GAMMA, "GOLOMB": GOLOMB, "SKEWED_GOLOMB": SKEWED_GOLOMB,
           "UNARY": UNARY, "ZETA": ZETA, "NIBBLE": NIBBLE}
# BVGraph.java:475-523: field name prefix -> bit shift inside the flag word (BVGraph.java:1317-1325)
_FIELD_SHIFT = {"OUTDEGREES": 0, "BLOCKS": 4, "RESIDUALS": 8, "REFERENCES": 12, "BLOCK_COUNT": 16, "OFFSETS": 20}


class Params(C.Structure):
    _fields_ = [("n", C.c_int32), ("window", C.c_int32), ("min_interval", C.c_int32), ("zeta_k", C.c_int32),
                ("outdegree_coding", C.c_int32), ("block_coding", C.c_int32), ("residual_coding", C.c_int32),
                ("reference_coding", C.c_int32), ("block_count_coding", C.c_int32), ("offset_coding", C.c_int32)]


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("bvg_oracle.c", "efg_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.bvo_open.restype = C.c_void_p
        L.bvo_open.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Params), C.c_void_p]
        L.bvo_close.argtypes = [C.c_void_p]
        L.bvo_decode_offsets.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int, C.c_void_p]
        L.bvo_labels_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.c_int, C.c_int, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.bvo_outdegree.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.bvo_outdegrees.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.bvo_successors.restype = C.c_int64
        L.bvo_successors.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]
        L.bvo_scan.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                               C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
        L.bvo_successors_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        L.bvo_references.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.bvo_copied.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


def parse_properties(path):
    """Minimal java.util.Properties reader (key=value, '#'/'!' comments); enough for BVGraph.java:1528-1543."""
    props = {}
    with open(path, "r", encoding="latin-1") as f:
        for line in f:
            line = line.strip()
            if not line or line[0] in "#!":
                continue
            for sep in "=:":
                if sep in line:
                    k, v = line.split(sep, 1)
                    props[k.strip()] = v.strip()
                    break
    return props


def flags_from_string(s):
    """BVGraph.string2Flags (BVGraph.java:1352-1366): 'A | B' of constant names -> flag word."""
    flags = 0
    if s:
        for tok in s.split("|"):
            tok = tok.strip()
            if not tok:
                continue
            for prefix, shift in _FIELD_SHIFT.items():
                if tok.startswith(prefix + "_") and tok[len(prefix) + 1:] in _CODING:
                    flags |= _CODING[tok[len(prefix) + 1:]] << shift
                    break
            else:
                raise IOError("Compression flag %s unknown." % tok)
    return flags


def params_from_properties(props):
    """BVGraph.loadInternal (BVGraph.java:1528-1543) + setFlags (:1317-1325)."""
    gc = props.get("graphclass", "").replace("it.unimi.dsi.big.webgraph", "it.unimi.dsi.webgraph")
    if gc != "it.unimi.dsi.webgraph.BVGraph":
        raise IOError("cannot load a graph stored using class " + gc)
    if "version" not in props:
        raise IOError("Missing format version information")
    if int(props["version"]) > 0:
        raise IOError("unsupported format version")
    n = int(props["nodes"])
    if n > 2**31 - 1:
        raise ValueError("too many nodes")
    flags = flags_from_string(props.get("compressionflags", ""))
    p = Params()
    p.n = n
    p.window = int(props["windowsize"])
    p.min_interval = int(props["minintervallength"])
    p.zeta_k = int(props.get("zetak", 3))
    p.outdegree_coding = (flags & 0xF) or GAMMA
    p.block_coding = ((flags >> 4) & 0xF) or GAMMA
    p.residual_coding = ((flags >> 8) & 0xF) or ZETA
    p.reference_coding = ((flags >> 12) & 0xF) or UNARY
    p.block_count_coding = ((flags >> 16) & 0xF) or GAMMA
    p.offset_coding = ((flags >> 20) & 0xF) or GAMMA
    return p, int(props["arcs"])


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d" % code)
        self.code = code


class OracleGraph:
    """Restatement of an in-memory BVGraph (ImmutableGraph.load(basename))."""

    def __init__(self, graph_bytes, params, offsets=None, arcs=None):
        self.params = params
        self.n = params.n
        self.arcs = arcs
        self._graph = np.frombuffer(graph_bytes, dtype=np.uint8)
        self.offsets = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.int64)
        optr = None if self.offsets is None else self.offsets.ctypes.data
        self._h = lib().bvo_open(self._graph.ctypes.data, self._graph.size, C.byref(params), optr)
        if not self._h:
            raise OracleError(-3)

    @classmethod
    def load(cls, basename, with_offsets=True):
        props = parse_properties(basename + ".properties")
        p, arcs = params_from_properties(props)
        with open(basename + ".graph", "rb") as f:
            g = f.read()
        offs = None
        if with_offsets:
            with open(basename + ".offsets", "rb") as f:
                ob = f.read()
            offs = decode_offsets(ob, p.n, p.offset_coding)
        return cls(g, p, offs, arcs)

    def close(self):
        if self._h:
            lib().bvo_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def outdegree(self, x):
        d = C.c_int32()
        rc = lib().bvo_outdegree(self._h, x, C.byref(d))
        if rc:
            raise OracleError(rc)
        return d.value

    def outdegrees(self, lo=0, hi=None):
        hi = self.n if hi is None else hi
        out = np.empty(hi - lo, dtype=np.int32)
        rc = lib().bvo_outdegrees(self._h, lo, hi, out.ctypes.data)
        if rc:
            raise OracleError(rc)
        return out

    def successors(self, x, cap=None):
        if cap is None:
            cap = max(self.outdegree(x), 1) if 0 <= x < self.n and self.offsets is not None else 1
        out = np.empty(cap, dtype=np.int32)
        d = lib().bvo_successors(self._h, x, out.ctypes.data, cap)
        if d < 0:
            raise OracleError(int(d))
        return out[:d].copy()

    def scan(self, lo=0, hi=None, want_succ=True, want_hash=False, cap=None, h0=-1):
        """nodeIterator(lo).copy(hi) drained: returns (rowptr[hi-lo+1], succ, arcs[, hash]); the hash continues from h0."""
        hi = self.n if hi is None else hi
        rowptr = np.empty(max(hi - lo, 0) + 1, dtype=np.int64)
        arcs = C.c_uint64(0)
        h = C.c_int32(h0)
        if want_succ:
            if cap is None:
                # first a counting pass
                rc = lib().bvo_scan(self._h, lo, hi, rowptr.ctypes.data, None, 0, C.byref(arcs), None)
                if rc:
                    raise OracleError(rc)
                cap = arcs.value
            succ = np.empty(max(cap, 1), dtype=np.int32)
            rc = lib().bvo_scan(self._h, lo, hi, rowptr.ctypes.data, succ.ctypes.data, cap, C.byref(arcs),
                                C.byref(h) if want_hash else None)
            if rc:
                raise OracleError(rc)
            succ = succ[:arcs.value]
        else:
            succ = None
            rc = lib().bvo_scan(self._h, lo, hi, rowptr.ctypes.data, None, 0, C.byref(arcs),
                                C.byref(h) if want_hash else None)
            if rc:
                raise OracleError(rc)
        if want_hash:
            return rowptr, succ, arcs.value, h.value
        return rowptr, succ, arcs.value

    def references(self, lo=0, hi=None):
        """Reference field of every node of [lo, hi) (0 = none)."""
        hi = self.n if hi is None else hi
        out = np.empty(hi - lo, dtype=np.int32)
        rc = lib().bvo_references(self._h, lo, hi, out.ctypes.data)
        if rc:
            raise OracleError(rc)
        return out

    def copied(self, lo=0, hi=None, threads=None):
        """How many successors every record of [lo, hi) copies from its referent (BVGraph.java:1058-1071), 0 without a reference."""
        hi = self.n if hi is None else hi
        out = np.empty(hi - lo, dtype=np.int32)
        threads = threads or min(os.cpu_count() or 1, 64)
        cuts = [lo + (hi - lo) * k // threads for k in range(threads + 1)]

        def one(k):
            a, b = cuts[k], cuts[k + 1]
            rc = lib().bvo_copied(self._h, a, b, out[a - lo:].ctypes.data) if b > a else 0
            if rc:
                raise OracleError(rc)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(one, range(threads)))
        return out

    def chain_depths(self):
        """Length of every node's reference chain (0 = no reference): what maxrefcount bounds (BVGraph.java:2315-2326)."""
        ref = self.references().astype(np.int64)
        idx = np.arange(self.n, dtype=np.int64)
        depth = np.zeros(self.n, dtype=np.int32)
        has = ref > 0
        for _ in range(1 << 16):  # one pass per level
            nd = np.where(has, depth[idx - ref] + 1, 0).astype(np.int32)
            if np.array_equal(nd, depth):
                break
            depth = nd
        return depth

    def scan_mt(self, lo=0, hi=None, threads=None, want_succ=True):
        """scan(lo, hi) on `threads` host threads: contiguous node ranges holding the same share of the bit stream, the
        window of each refilled through the random-access path (as ImmutableGraph.splitNodeIterators does,
        ImmutableGraph.java:379-409; ctypes releases the GIL).  Returns (rowptr, succ or None, arcs)."""
        import concurrent.futures as cf
        hi = self.n if hi is None else hi
        T = max(1, min(threads or (os.cpu_count() or 1), 256, max(hi - lo, 1)))
        off = self.offsets
        cuts = [lo]
        for k in range(1, T):
            t = int(off[lo]) + (int(off[hi]) - int(off[lo])) * k // T
            cuts.append(max(cuts[-1], min(hi, int(np.searchsorted(off[lo:hi], t, side="left")) + lo)))
        cuts.append(hi)
        rngs = [(cuts[k], cuts[k + 1]) for k in range(T) if cuts[k + 1] > cuts[k]] or [(lo, hi)]
        with cf.ThreadPoolExecutor(max_workers=len(rngs)) as ex:
            parts = list(ex.map(lambda ab: self.scan(ab[0], ab[1], want_succ=want_succ), rngs))
        rowptr = np.empty(hi - lo + 1, dtype=np.int64)
        rowptr[0] = 0
        base = 0
        for (a, b), (rp, sc, arcs) in zip(rngs, parts):
            rowptr[a - lo + 1:b - lo + 1] = rp[1:] + base
            base += arcs
        succ = np.concatenate([p[1] for p in parts]) if want_succ else None
        return rowptr, succ, base

    def hashcode_mt(self, threads=None):
        """ImmutableGraph.hashCode() on all host cores: every thread scans a contiguous node range twice, from h = 0 and from
        h = 1 -- the range acts on the running hash as h -> A*h + B over Z/2^32 (a chain of h = 31*h + v,
        ImmutableGraph.java:757-770), so B = f(0), A = f(1) - f(0) -- and the maps are folded in node order from -1."""
        import concurrent.futures as cf
        T = max(1, min(threads or (os.cpu_count() or 1), 256, max(self.n, 1)))
        off = self.offsets
        cuts = [0]
        for k in range(1, T):
            cuts.append(max(cuts[-1], min(self.n, int(np.searchsorted(off[:self.n], int(off[self.n]) * k // T, side="left")))))
        cuts.append(self.n)
        rngs = [(cuts[k], cuts[k + 1]) for k in range(T) if cuts[k + 1] > cuts[k]]

        def one(ab):
            f0 = self.scan(ab[0], ab[1], want_succ=False, want_hash=True, h0=0)[3]
            f1 = self.scan(ab[0], ab[1], want_succ=False, want_hash=True, h0=1)[3]
            return (f1 - f0) & 0xFFFFFFFF, f0 & 0xFFFFFFFF
        with cf.ThreadPoolExecutor(max_workers=max(len(rngs), 1)) as ex:
            maps = list(ex.map(one, rngs))
        h = 0xFFFFFFFF
        for a, b in maps:
            h = (a * h + b) & 0xFFFFFFFF
        return h - (1 << 32) if h & 0x80000000 else h

    def hashcode(self):
        """ImmutableGraph.hashCode() (ImmutableGraph.java:757-770)."""
        return self.scan(0, self.n, want_succ=False, want_hash=True)[3]

    def successors_batch(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        rowptr = np.empty(nodes.size + 1, dtype=np.int64)
        rc = lib().bvo_successors_batch(self._h, nodes.ctypes.data, nodes.size, rowptr.ctypes.data, None, 0)
        if rc:
            raise OracleError(rc)
        succ = np.empty(max(int(rowptr[-1]), 1), dtype=np.int32)
        rc = lib().bvo_successors_batch(self._h, nodes.ctypes.data, nodes.size, rowptr.ctypes.data,
                                        succ.ctypes.data, succ.size)
        if rc:
            raise OracleError(rc)
        return rowptr, succ[:int(rowptr[-1])]


def decode_offsets(offset_bytes, n, coding=GAMMA):
    """OffsetsLongIterator (BVGraph.java:907-935): n+1 running sums of gamma/delta coded gaps."""
    b = np.frombuffer(offset_bytes, dtype=np.uint8)
    out = np.empty(n + 1, dtype=np.int64)
    rc = lib().bvo_decode_offsets(b.ctypes.data, b.size, n, coding, out.ctypes.data)
    if rc:
        raise OracleError(rc)
    return out


def read_ascii_graph_gz(path):
    """ASCIIGraph text format (ASCIIGraph.java:57-61): first line n, then one line of successors per node."""
    import gzip
    with gzip.open(path, "rt") as f:
        n = int(f.readline())
        rowptr = np.zeros(n + 1, dtype=np.int64)
        chunks = []
        for i in range(n):
            line = f.readline()
            a = np.array(line.split(), dtype=np.int32) if line.strip() else np.empty(0, dtype=np.int32)
            chunks.append(a)
            rowptr[i + 1] = rowptr[i] + a.size
    succ = np.concatenate(chunks) if chunks else np.empty(0, dtype=np.int32)
    return n, rowptr, succ


def parse_labelspec(spec):
    """'<class>(KEY[,WIDTH])' -> (kind, width, key): kind 1 = GammaCodedIntLabel, 2 = FixedWidthIntLabel, 3 = FixedWidthIntListLabel (Label.toSpec())."""
    cls, args = spec.split("(", 1)
    args = [a.strip() for a in args.rsplit(")", 1)[0].split(",")]
    name = cls.strip().rsplit(".", 1)[-1]
    if name == "GammaCodedIntLabel" and len(args) == 1:
        return 1, -1, args[0]
    if name == "FixedWidthIntLabel" and len(args) == 2:
        return 2, int(args[1]), args[0]
    if name == "FixedWidthIntListLabel" and len(args) == 2:
        return 3, int(args[1]), args[0]
    raise ValueError("unsupported label class: " + spec)


def labels_decode(basename, n, outd, lo=0, hi=None):
    """Labels of the arcs of nodes [lo, hi) of a BitStreamArcLabelledImmutableGraph (test oracle): `outd` = outdegrees of those nodes."""
    props = parse_properties(basename + ".properties")
    kind, width, _ = parse_labelspec(props["labelspec"])
    hi = n if hi is None else hi
    lab = np.frombuffer(open(basename + ".labels", "rb").read(), dtype=np.uint8)
    lof = np.frombuffer(open(basename + ".labeloffsets", "rb").read(), dtype=np.uint8)
    outd = np.ascontiguousarray(outd, dtype=np.int32)
    out = np.empty(max(int(outd.sum()), 1), dtype=np.int32)
    cnt = C.c_uint64(0)
    rc = lib().bvo_labels_decode(lab.ctypes.data if lab.size else None, lab.size, lof.ctypes.data, lof.size, n, kind, max(width, 0), lo, hi,
                                 outd.ctypes.data, out.ctypes.data, out.size, C.byref(cnt))
    if rc:
        raise OracleError(rc)
    return out[:cnt.value]


def label_lists_decode(basename, n, outd, lo=0, hi=None):
    """(listptr, values) of the arcs of nodes [lo, hi) of a graph labelled with FixedWidthIntListLabel (test oracle)."""
    props = parse_properties(basename + ".properties")
    kind, width, _ = parse_labelspec(props["labelspec"])
    assert kind == 3
    hi = n if hi is None else hi
    lab = np.frombuffer(open(basename + ".labels", "rb").read(), dtype=np.uint8)
    lof = np.frombuffer(open(basename + ".labeloffsets", "rb").read(), dtype=np.uint8)
    outd = np.ascontiguousarray(outd, dtype=np.int32)
    arcs = int(outd.sum())
    f = lib().bvo_labels_decode_lists
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                  C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    nl, nv = C.c_uint64(0), C.c_uint64(0)
    args = (lab.ctypes.data if lab.size else None, lab.size, lof.ctypes.data, lof.size, n, width, lo, hi, outd.ctypes.data)
    rc = f(*args, None, 0, None, 0, C.byref(nl), C.byref(nv))  # sizing pass
    if rc:
        raise OracleError(rc)
    listptr = np.empty(arcs + 1, dtype=np.int64)
    values = np.empty(max(nv.value, 1), dtype=np.int32)
    rc = f(*args, listptr.ctypes.data, listptr.size, values.ctypes.data, values.size, C.byref(nl), C.byref(nv))
    if rc:
        raise OracleError(rc)
    return listptr, values[:nv.value]


class OracleEFGraph:
    """EFGraph (the reference's quasi-succinct second format) read by the CPU oracle (oracle/efg_oracle.c; parity unpinned)."""

    def __init__(self, words, offsets, n, arcs, upper_bound, log2_quantum):
        self.words, self.offsets, self.n, self.arcs, self.upper_bound, self.log2_quantum = words, offsets, n, arcs, upper_bound, log2_quantum

    @classmethod
    def load(cls, basename):
        props = parse_properties(basename + ".properties")
        if props.get("graphclass", "").replace("it.unimi.dsi.big.webgraph", "it.unimi.dsi.webgraph") != "it.unimi.dsi.webgraph.EFGraph":
            raise ValueError("not an EFGraph: " + props.get("graphclass", ""))
        n, m = int(props["nodes"]), int(props["arcs"])
        ub = int(props.get("upperbound", n))
        q = int(props["quantum"])
        lq = q.bit_length() - 1
        assert 1 << lq == q
        raw = open(basename + ".graph", "rb").read()
        raw += b"\0" * (-len(raw) % 8)
        words = np.frombuffer(raw, dtype=">u8" if props["byteorder"] == "BIG_ENDIAN" else "<u8").astype(np.uint64)
        offsets = decode_offsets(open(basename + ".offsets", "rb").read(), n, coding=DELTA)
        return cls(np.ascontiguousarray(words), offsets, n, m, ub, lq)

    def scan(self, lo=0, hi=None, want_succ=True):
        hi = self.n if hi is None else hi
        f = lib().efo_scan
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        rowptr = np.empty(hi - lo + 1, dtype=np.int64)
        arcs = C.c_uint64(0)
        rc = f(self.words.ctypes.data, self.words.size, self.offsets.ctypes.data, self.n, self.upper_bound, self.log2_quantum, lo, hi, rowptr.ctypes.data, None, 0, C.byref(arcs))
        if rc:
            raise OracleError(rc)
        if not want_succ:
            return rowptr, None, arcs.value
        succ = np.empty(max(arcs.value, 1), dtype=np.int32)
        rc = f(self.words.ctypes.data, self.words.size, self.offsets.ctypes.data, self.n, self.upper_bound, self.log2_quantum, lo, hi, rowptr.ctypes.data, succ.ctypes.data, succ.size,
               C.byref(arcs))
        if rc:
            raise OracleError(rc)
        return rowptr, succ[:arcs.value], arcs.value

    def skip_to(self, nodes, bounds):
        """EliasFanoSuccessorReader.skipTo on a fresh reader per (node, bound) pair, THROUGH the forward pointers where the reference uses them (efo_skip_to):
        (answers, -1 at the end of a list; which queries went through a pointer)."""
        f = lib().efo_skip_to
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        bounds = np.ascontiguousarray(bounds, dtype=np.int32)
        out = np.empty(nodes.size, dtype=np.int32)
        used = np.zeros(nodes.size, dtype=np.uint8)
        rc = f(self.words.ctypes.data, self.words.size, self.offsets.ctypes.data, self.n, self.upper_bound, self.log2_quantum, nodes.ctypes.data, bounds.ctypes.data, nodes.size,
               out.ctypes.data, used.ctypes.data)
        if rc:
            raise OracleError(rc)
        return out, used.astype(bool)

