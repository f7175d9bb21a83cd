"""Malformed inputs: the library must report an error (or decode garbage consistently) -- never crash, hang or write
out of bounds ("errors": BVG_EFORMAT / BVG_ESTATE / BVG_ECAP instead of the reference's
exceptions, BVGraph.java:705, :1037)."""
import os
import shutil

import numpy as np
import pytest

from conftest import CNR

pytestmark = pytest.mark.gpu


def _errors():
    from webgraph_amd.bvgraph import BvgError
    return (ValueError, IOError, RuntimeError, MemoryError, OSError, NotImplementedError, BvgError)


def _copy_fixture(tmp_path, name):
    base = str(tmp_path / name)
    for ext in (".graph", ".offsets", ".properties"):
        shutil.copy(CNR + ext, base + ext)
    return base


@pytest.mark.parametrize("seed", range(12))
def test_bit_flips_never_crash(tmp_path, seed):
    from webgraph_amd.bvgraph import BVGraph
    base = _copy_fixture(tmp_path, "flip%d" % seed)
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    raw = bytearray(open(base + ".graph", "rb").read())
    for _ in range(1 + seed % 4):  # a few flipped bits, early in the file so that many records are affected downstream
        pos = int(rng.integers(0, len(raw) // (1 + seed % 3)))
        raw[pos] ^= 1 << int(rng.integers(0, 8))
    open(base + ".graph", "wb").write(bytes(raw))
    g = BVGraph.load(base)
    try:
        rowptr, succ = g.decode_range()
    except _errors():
        rowptr = None  # an error status is a fine answer
    if rowptr is not None:  # decoded "something": it must at least be a well-formed CSR
        assert rowptr[0] == 0 and np.all(np.diff(rowptr) >= 0) and rowptr[-1] == succ.size
    # the handle survives and still answers other requests (or errors again) without crashing
    try:
        g.successors_batch(np.array([0, 1, 325556], dtype=np.int32))
    except _errors():
        pass
    # a dense batch goes through the masked scan + gather: same promise
    q = rng.integers(0, g.numNodes(), 40000).astype(np.int32)
    try:
        rp, sc = g.successors_batch(q)
        assert rp[0] == 0 and np.all(np.diff(rp) >= 0) and rp[-1] == sc.size
    except _errors():
        pass
    g.close()


def test_truncated_graph_is_an_io_error(tmp_path):
    from webgraph_amd.bvgraph import BVGraph
    base = _copy_fixture(tmp_path, "trunc")
    raw = open(base + ".graph", "rb").read()
    open(base + ".graph", "wb").write(raw[: len(raw) // 2])
    with pytest.raises(_errors()):
        BVGraph.load(base)


def test_garbage_offsets_are_rejected(tmp_path):
    from webgraph_amd.bvgraph import BVGraph
    base = _copy_fixture(tmp_path, "badoffs")
    raw = bytearray(open(base + ".offsets", "rb").read())
    raw[1000] ^= 0x55
    raw[20000] ^= 0xaa
    open(base + ".offsets", "wb").write(bytes(raw))
    try:
        g = BVGraph.load(base)
    except _errors():
        return  # rejected at load: fine
    try:  # or accepted (the corrupted stream may still hold n+1 monotone values): decoding must not crash
        g.decode_range(0, 2000)
    except _errors():
        pass
    g.close()


@pytest.mark.timeout(120)
def test_understated_arcs_property(tmp_path, monkeypatch):
    """The `arcs` property is only what numArcs() reports (ImmutableGraph.java:254-260): a file that understates it
    still decodes, in the reference and here -- scratch (interval arena, copy queues) is sized by the outdegrees the
    stream really holds, summed once at load time.  (Sized by the property, the arena slices of most long records fell
    outside; the records were skipped by a `continue` behind `if (threadIdx.x == 0) ...` in a loop that hands out work
    through shared memory, and in a one-wave block the lanes never met again at the barrier: a livelock.)"""
    from webgraph_amd import tools as T
    from webgraph_amd.bvgraph import BVGraph
    rowptr, succ = T.generate(60000, 1500000, seed=77, p_copy=0.6)
    base = str(tmp_path / "fewarcs")
    T.store(base, rowptr, succ, window=7, max_ref_count=3, min_interval=3, zeta_k=5)
    props = open(base + ".properties").read()
    assert "arcs=%d" % succ.size in props
    open(base + ".properties", "w").write(props.replace("arcs=%d" % succ.size, "arcs=3000"))
    q = np.arange(0, 60000, 3, dtype=np.int32)
    for env in ({"BVGPU_COOP_MIN": "64", "BVGPU_GIANT_MIN": "2000"}, {}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = BVGraph.load(base)
        assert g.numArcs() == 3000
        rp, sc = g.decode_range()
        assert np.array_equal(rp, rowptr) and np.array_equal(sc, succ)
        rp, sc = g.successors_batch(q)
        deg = (rowptr[q.astype(np.int64) + 1] - rowptr[q]).astype(np.int64)
        assert np.array_equal(np.diff(rp), deg) and np.array_equal(sc[:deg[0]], succ[rowptr[q[0]]:rowptr[q[0] + 1]])
        g.close()
        for k in env:
            monkeypatch.delenv(k)


def _plain_row(first, d, x):
    """record of node x: d successors first, first+2, ... as residuals, no reference, no intervals"""
    from bitio import int2nat

    def rec(w):
        w.gamma(d)
        w.unary(0)
        if d:
            w.zeta(int2nat(first - x))
            for _ in range(d - 1):
                w.zeta(1)  # gap 2
    return rec


@pytest.mark.parametrize("d", [3, 200, 700, 3000])
@pytest.mark.parametrize("dense", ["0", "1"])
def test_block_lengths_that_wrap_are_rejected(tmp_path, monkeypatch, d, dense):
    """Copy-block lengths are gamma codes of up to 2^64 - 2: with len0 = 2^62, len1 = 1, len2 = 2^64 - 2^62 the 64-bit sums
    of a careless decoder wrap to total = 1 <= dref and copied = 0, every later check passes, and the copy pass walks 2^62
    ids past the referent's row.  Each length is checked against what is left of the referent (MaskedIntIterator would
    simply run off the referent's list in the reference)."""
    from bitio import write_graph, int2nat
    from webgraph_amd.bvgraph import BVGraph
    monkeypatch.setenv("BVGPU_BATCH_DENSE", dense)

    def evil(x, blocks):
        def rec(w):
            w.gamma(d)
            w.unary(1)
            w.gamma(len(blocks))
            for i, b in enumerate(blocks):
                w.gamma(b if i == 0 else b - 1)
            w.zeta(int2nat(5 - x))
            for _ in range(d - 1):
                w.zeta(0)
        return rec

    cases = [[1 << 62, 1, (1 << 64) - (1 << 62)],      # total wraps to 1, copied to 0
             [(1 << 64) - 2],                            # a single huge block
             [1, (1 << 64) - 1 - 1, 1],                  # the skipped block wraps the index
             [0, 1, (1 << 63), 1, (1 << 63)]]            # two halves that cancel
    recs = [_plain_row(10, d, 0)]
    for i, blocks in enumerate(cases):
        recs.append(evil(2 * i + 1, blocks))
        recs.append(_plain_row(10, d, 2 * i + 2))
    base = str(tmp_path / "wrap")
    write_graph(base, recs, arcs=d * len(recs))
    g = BVGraph.load(base)
    with pytest.raises(_errors()):
        g.decode_range()
    with pytest.raises(_errors()):
        g.successors_batch(np.arange(len(recs), dtype=np.int32))
    # the well-formed rows are still served
    rp, sc = g.successors_batch(np.array([0, 2], dtype=np.int32))
    assert list(np.diff(rp)) == [d, d] and sc[0] == 10 and sc[d - 1] == 10 + 2 * (d - 1)
    g.close()


def test_stats_scan_rejects_successors_outside_the_graph(tmp_path):
    """A stream can name any id (ids wrap in Java ints, BVG:954): the statistics scan counts indegrees by successor, so an id
    outside [0, n) must end the scan with an error instead of an increment at an address the file chose (the
    reference throws ArrayIndexOutOfBoundsException at Stats.java:130)."""
    import torch
    from bitio import write_graph
    from webgraph_amd.bvgraph import BVGraph
    n = 50
    recs = [_plain_row(3, 4, x) for x in range(n)]
    recs[17] = _plain_row(1 << 20, 4, 17)      # far past the last node
    recs[31] = _plain_row(-5000, 3, 31)        # negative ids
    base = str(tmp_path / "oob")
    write_graph(base, recs, arcs=4 * n - 1)
    g = BVGraph.load(base)
    rp, sc = g.decode_range()                  # the plain scan hands the ids out as they are, like the reference
    assert sc[rp[17]] == 1 << 20 and sc[rp[31]] == -5000
    guard = torch.zeros(n + 4096, dtype=torch.int32, device="cuda")
    with pytest.raises(_errors()):
        g.scan_stats(0, n, indegree_ptr=guard.data_ptr())
    torch.cuda.synchronize()
    assert int(guard[n:].sum()) == 0 and int(guard[:n].sum()) == 4 * n - 1 - 7
    with pytest.raises(_errors()):
        g.scan_stats(0, n)
    st = g.scan_stats(0, 17)                   # well-formed ranges are still served
    assert st["arcs"] == 4 * 17
    g.close()


@pytest.mark.parametrize("knobs", [{}, {"BVGPU_TILE": "1"}, {"BVGPU_COOP_MIN": "2147483647"}])
def test_residuals_inside_intervals_are_emitted_once(tmp_path, monkeypatch, knobs):
    """A residual that equals an id of an interval: MergedIntIterator.java:69-72 emits the two equal heads once, so the list is
    shorter than its outdegree and BVGraph.java:1210 would pad the array with -1.  No writer produces such a record; the
    one-lane decoders (straight-line and general loop, lane windows and tiles) must still agree with the oracle on it."""
    from bitio import write_graph, int2nat
    from webgraph_amd.bvgraph import BVGraph
    from oracle import oracle as O
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    MI = 3

    def rec_of(x, ivs, res):
        d = sum(n for _, n in ivs) + len(res)

        def rec(w):
            w.gamma(d)
            w.unary(0)
            w.gamma(len(ivs))
            prev = None
            for (l, n) in ivs:
                w.gamma(int2nat(l - x) if prev is None else l - prev - 1)
                w.gamma(n - MI)
                prev = l + n
            pv = None
            for r in res:
                w.zeta(int2nat(r - x) if pv is None else r - pv - 1)
                pv = r
        return rec, d

    rng = np.random.default_rng(5)
    specs = [([(10, 4)], [12, 20]), ([(10, 4), (30, 5)], [5, 10, 31, 40]), ([(5, 3)], [7]), ([(5, 3)], [1, 9]), ([], [3, 4]), ([(2, 3)], [])]
    for _ in range(40):  # longer rows: a dozen intervals, residuals drawn over the same range (a third of them collide)
        ivs, pos = [], 0
        for _ in range(int(rng.integers(1, 14))):
            pos += int(rng.integers(1, 9))
            n = int(rng.integers(MI, 9))
            ivs.append((pos, n))
            pos += n
        res = sorted(set(int(v) for v in rng.integers(0, pos + 10, size=int(rng.integers(1, 30)))))
        specs.append((ivs, res))
    recs, arcs = [], 0
    for x, (ivs, res) in enumerate(specs):
        r, d = rec_of(x, ivs, res)
        recs.append(r)
        arcs += d
    base = str(tmp_path / "equalheads")
    write_graph(base, recs, min_interval=MI, arcs=arcs)
    rp0, sc0, _ = O.OracleGraph.load(base).scan()
    assert (sc0 == -1).any()
    g = BVGraph.load(base)
    rp, sc = g.decode_range()
    assert np.array_equal(rp, rp0) and np.array_equal(sc, sc0)
    q = np.arange(len(recs) - 1, -1, -1, dtype=np.int32)
    rpb, scb = g.successors_batch(q)
    for i, x in enumerate(q):
        assert np.array_equal(scb[rpb[i]:rpb[i + 1]], sc0[rp0[x]:rp0[x + 1]])
    g.close()


@pytest.mark.parametrize("knobs", [{}, {"BVGPU_COPY_LOOP": "0"}, {"BVGPU_TILE": "1"}, {"BVGPU_TILE": "0", "BVGPU_COPY_VEC": "1"}, {"BVGPU_COPY_TABLES": "0"}, {"BVGPU_COPY_MID_MIN": "4", "BVGPU_COOP_MIN": "2147483647"}, {"BVGPU_COPY_MID_MIN": "4", "BVGPU_PREWALK": "0"}])
def test_extras_that_equal_copied_ids_are_emitted_once(tmp_path, monkeypatch, knobs):
    """The same one level up: a residual (or an id of an interval) that equals an id COPIED from the referent.  MergedIntIterator.java:69-72 emits the equal heads once, the
    list is shorter than its outdegree and the array ends in -1 (BVGraph.java:1210).  The copy pass's merges -- the wave's loop (k_copy_list_w), the lane-by-lane merges, the
    wave class's ranks (k_copy_mid: rows of 128 .. 1023 ids by default, here from 4 on) -- must agree with the oracle; the rows that copy from such a row (its -1 included) too.
    (The group class -- rows of 1 024 ids and more: test_group_class_emits_equal_heads_once below.)"""
    from bitio import write_graph, int2nat
    from webgraph_amd.bvgraph import BVGraph
    from oracle import oracle as O
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(11)

    def row(x, ref, blocks, res, d):
        def rec(w):
            w.gamma(d)
            if d == 0:
                return
            w.unary(ref)
            if ref:
                w.gamma(len(blocks))
                for i, b in enumerate(blocks):
                    w.gamma(b if i == 0 else b - 1)
            if res:
                w.gamma(0)  # no intervals (the count is there only when the row has extras: BVG:1073)
            pv = None
            for r in res:
                w.zeta(int2nat(r - x) if pv is None else r - pv - 1)
                pv = r
        return rec

    recs, arcs = [], 0
    protos = {}
    for x in range(0, 400, 4):
        base_ids = sorted(set(int(v) for v in rng.integers(0, 3000, size=int(rng.integers(6, 60)))))
        protos[x] = base_ids
        recs.append(row(x, 0, [], base_ids, len(base_ids))); arcs += len(base_ids)
        # x + 1 copies everything and adds residuals of which some collide
        coll = [base_ids[i] for i in sorted(set(int(v) for v in rng.integers(0, len(base_ids), size=3)))]
        fresh = [int(v) for v in rng.integers(0, 3000, size=4) if int(v) not in base_ids]
        res = sorted(set(coll + fresh))
        d1 = len(base_ids) + len(res)
        recs.append(row(x + 1, 1, [], res, d1)); arcs += d1
        # x + 2 copies the first three of x, skips two, copies the rest; a colliding residual in each part
        if len(base_ids) >= 8:
            kept = base_ids[:3] + base_ids[5:]
            res2 = sorted(set([base_ids[1], base_ids[6], 2999 + x]))
            d2 = len(kept) + len(res2)
            recs.append(row(x + 2, 2, [3, 2], res2, d2)); arcs += d2
        else:
            recs.append(row(x + 2, 0, [], [], 0))
        # x + 3 copies x + 1 whole (a row that ends in -1): no extras
        recs.append(row(x + 3, 2, [], [], d1)); arcs += d1
    base = str(tmp_path / "equalcopied")
    write_graph(base, recs, min_interval=2, arcs=arcs)
    rp0, sc0, _ = O.OracleGraph.load(base).scan()
    assert (sc0 == -1).any()
    g = BVGraph.load(base)
    rp, sc = g.decode_range()
    assert np.array_equal(rp, rp0) and np.array_equal(sc, sc0)
    rp, sc = g.decode_range(101, 303)
    assert np.array_equal(sc, sc0[rp0[101]:rp0[303]])
    g.close()
    for dense in ("0", "1"):  # the batch entry point: chains decoded per query (k_bcopy_coop / the lane merges) or a masked scan
        monkeypatch.setenv("BVGPU_BATCH_DENSE", dense)
        g = BVGraph.load(base)
        q = np.arange(len(recs) - 1, -1, -1, dtype=np.int32)
        rpb, scb = g.successors_batch(q)
        for i, x in enumerate(q):
            assert np.array_equal(scb[rpb[i]:rpb[i + 1]], sc0[rp0[x]:rp0[x + 1]]), (dense, x)
        g.close()


@pytest.mark.parametrize("shape", ["lds", "chunks", "stream"])
@pytest.mark.parametrize("knobs", [{}, {"BVGPU_PREWALK": "0"}, {"BVGPU_COOP_MIN": "2147483647"}])
def test_group_class_emits_equal_heads_once(tmp_path, monkeypatch, shape, knobs):
    """The same for the rows the GROUP class of the copy pass merges (1 024 ids and more, k_copy_big), through its three merges: both sets in LDS ("lds": a referent of 3 000 ids,
    2 000 extras), the extras moved in place chunk by chunk ("chunks": 9 000 extras), tables and copied ids in global scratch with the output cut into tiles ("stream": a referent of
    10 000 ids; one equal pair sits astride the first cut at 8 192, one is the last copied id).  The merges are stable and the group closes the gaps afterwards (dedupe_row)."""
    from bitio import write_graph, int2nat
    from webgraph_amd.bvgraph import BVGraph
    from oracle import oracle as O
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(23)

    def row(x, ref, blocks, res, d):
        def rec(w):
            w.gamma(d)
            if d == 0:
                return
            w.unary(ref)
            if ref:
                w.gamma(len(blocks))
                for i, b in enumerate(blocks):
                    w.gamma(b if i == 0 else b - 1)
            if res:
                w.gamma(0)  # no intervals
            pv = None
            for r in res:
                w.zeta(int2nat(r - x) if pv is None else r - pv - 1)
                pv = r
        return rec

    nref, nextra = {"lds": (3000, 2000), "chunks": (3000, 9000), "stream": (10000, 3000)}[shape]
    recs, arcs = [], 0
    for fam in range(3):
        x = 4 * fam
        proto = [2 * t for t in range(nref)]  # the referent: even numbers
        recs.append(row(x, 0, [], proto, nref)); arcs += nref
        if fam == 0:  # copies everything; an equal pair at merged positions (8 191, 8 192) when the referent is long enough, the last copied id, a few more
            dups = [proto[min(8191, nref - 7)], proto[-1], proto[5], proto[nref // 2]]
            blocks, kept = [], proto
        elif fam == 1:  # copies every second stretch of 50
            blocks = [50] * (nref // 50 - 1)
            kept = [v for i, v in enumerate(proto) if (i // 50) % 2 == 0]
            dups = [kept[0], kept[-1], kept[len(kept) // 3]] + [kept[int(i)] for i in rng.integers(0, len(kept), size=20)]
        else:  # skips the first 10, copies the rest
            blocks = [0, 10]
            kept = proto[10:]
            dups = [kept[int(i)] for i in rng.integers(0, len(kept), size=200)]
        odd = sorted(set(int(v) for v in rng.integers(2 * min(8192, nref - 6), 2 * nref + 60000, size=3 * nextra) if v % 2 == 1))[:nextra]
        res = sorted(set(dups + odd))
        d1 = len(kept) + len(res)
        recs.append(row(x + 1, 1, blocks, res, d1)); arcs += d1
        recs.append(row(x + 2, 1, [], [], d1)); arcs += d1  # copies the row that ends in -1, whole
        recs.append(row(x + 3, 0, [], [], 0))
    base = str(tmp_path / ("groupdup_" + shape))
    write_graph(base, recs, min_interval=2, arcs=arcs)
    rp0, sc0, _ = O.OracleGraph.load(base).scan()
    assert (sc0 == -1).sum() >= 3 * 2 * 3
    g = BVGraph.load(base)
    rp, sc = g.decode_range()
    assert np.array_equal(rp, rp0)
    for x in range(len(recs)):
        assert np.array_equal(sc[rp0[x]:rp0[x + 1]], sc0[rp0[x]:rp0[x + 1]]), (shape, x)
    g.close()
    for dense in ("0", "1"):
        monkeypatch.setenv("BVGPU_BATCH_DENSE", dense)
        g = BVGraph.load(base)
        q = np.array([1, 2, 5, 6, 9, 10, 0], dtype=np.int32)
        rpb, scb = g.successors_batch(q)
        for i, x in enumerate(q):
            assert np.array_equal(scb[rpb[i]:rpb[i + 1]], sc0[rp0[x]:rp0[x + 1]]), (shape, dense, x)
        g.close()
