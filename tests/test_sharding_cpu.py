"""CPU tests of the multi-GPU path: shard planning and the host-side (arcs, hashCode) reduction, world_size 2 on gloo.

No GPU here, so each rank's shard is decoded by the CPU oracle as a stand-in; what is under test is the product's
partitioning rule and its reduction (webgraph_amd/parallel.py), i.e. everything of the N>1 path except the kernels.
"""
import os
import socket

import numpy as np
import pytest

from conftest import CNR, ROOT


def test_shard_bounds_rule(cnr_oracle):
    from webgraph_amd.parallel import shard_bounds_from_offsets
    g, rowptr, _ = cnr_oracle
    off = g.offsets
    for parts in (1, 2, 3, 8, 64):
        b = shard_bounds_from_offsets(off, parts)
        assert b[0] == 0 and b[-1] == g.n and np.all(np.diff(b) >= 0)
        for k in range(1, parts):
            target = int(off[-1]) * k // parts
            assert off[b[k]] >= target and (b[k] == 0 or off[b[k] - 1] < target)
        bits = np.diff(off[b])
        assert bits.max() - bits.min() <= 2 * 5989 + 1  # balanced up to the longest record (5989 bits)


def test_affine_fold_equals_sequential_hash(cnr_oracle):
    from webgraph_amd.parallel import affine_from_two_hashes, fold_affine, shard_bounds_from_offsets
    g, _, _ = cnr_oracle
    b = shard_bounds_from_offsets(g.offsets, 5)
    pairs = []
    for k in range(5):
        h0 = g.scan(int(b[k]), int(b[k + 1]), want_succ=False, want_hash=True)[3]  # starts from -1 ...
        # ... so build f(0) and f(1) explicitly through the oracle's hash_io
        pairs.append(_affine(g, int(b[k]), int(b[k + 1])))
    assert fold_affine(pairs) == 1711395807


def _affine(g, lo, hi):
    import ctypes as C
    from oracle import oracle as O
    from webgraph_amd.parallel import affine_from_two_hashes
    out = []
    for h0 in (0, 1):
        h = C.c_int32(h0)
        arcs = C.c_uint64(0)
        rp = np.empty(hi - lo + 1, dtype=np.int64)
        rc = O.lib().bvo_scan(g._h, lo, hi, rp.ctypes.data, None, 0, C.byref(arcs), C.byref(h))
        assert rc == 0
        out.append(h.value)
    return affine_from_two_hashes(out[0], out[1])


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from webgraph_amd.parallel import reduce_scan, shard_bounds_from_offsets
        g = O.OracleGraph.load(CNR)
        b = shard_bounds_from_offsets(g.offsets, world)
        lo, hi = int(b[rank]), int(b[rank + 1])
        _, _, arcs = g.scan(lo, hi, want_succ=False)  # stand-in for bvg_decode_range on this rank's GPU
        total, h = reduce_scan(arcs, _affine(g, lo, hi))
        q.put((rank, lo, hi, arcs, total, h))
    finally:
        dist.destroy_process_group()


def test_two_rank_scan_reduction_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, a0, t0, h0), (r1, lo1, hi1, a1, t1, h1) = res
    assert lo0 == 0 and hi0 == lo1 and hi1 == 325557
    assert a0 + a1 == 3216152 and t0 == t1 == 3216152
    assert h0 == h1 == 1711395807  # ImmutableGraph.hashCode() of the whole graph from two shards
