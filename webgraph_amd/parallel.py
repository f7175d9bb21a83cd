"""Multi-GPU plumbing of the scan: contiguous node-range shards, one process per GPU, no data-path collective.

SURVEY.md section 8(e): `offsets[x]` is an independent entry point for every node, so a full sequential scan
partitions into node ranges; each rank decodes its range (referents before the range come through the halo of
bvg_decode_range) and only (arcs, fingerprint) pairs are reduced on the host side.  The reference's analogue is
ImmutableGraph.splitNodeIterators (ImmutableGraph.java:379-409), which splits by node count; shards here are
balanced by compressed bits instead (same idea as HyperBall's arc-granular chunks, algo/HyperBall.java:865-869).
"""
import numpy as np

MASK = 0xFFFFFFFF


def shard_bounds_from_offsets(offsets, parts):
    """bounds[k] = min{x : off[x] >= k*off[n]/parts}; bounds[0] = 0, bounds[parts] = n (== bvg_shard_bounds)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    total = int(offsets[-1])
    b = np.zeros(parts + 1, dtype=np.int32)
    for k in range(1, parts):
        target = total * k // parts
        x = int(np.searchsorted(offsets[:n], target, side="left"))
        b[k] = max(x, b[k - 1])
    b[parts] = n
    return b


def affine_from_two_hashes(h0, h1):
    """A scan segment acts on the running ImmutableGraph.hashCode as h -> A*h + B over Z/2^32
    (ImmutableGraph.java:757-770 is a chain of h = 31*h + v).  Given f(0) and f(1): B = f(0), A = f(1) - f(0)."""
    b = h0 & MASK
    a = (h1 - h0) & MASK
    return a, b


def fold_affine(pairs, h=-1):
    """Applies the shards' maps in rank order to the initial value of hashCode() (-1); returns a Java int."""
    h &= MASK
    for a, b in pairs:
        h = (a * h + b) & MASK
    return h - (1 << 32) if h & 0x80000000 else h


def reduce_scan(local_arcs, local_affine, group=None):
    """Host-side reduction over the ranks of a torch.distributed group: total arcs and the whole graph's hashCode.

    Exchanges three integers per rank (all_gather); this is the only communication of a sharded scan."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.tensor([int(local_arcs), int(local_affine[0]), int(local_affine[1])], dtype=torch.int64, device=dev)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    rows = [t.cpu().tolist() for t in allv]
    arcs = sum(r[0] for r in rows)
    return arcs, fold_affine([(r[1], r[2]) for r in rows])


def scan_shard(graph, rank, world, rowptr_ptr=None, succ_ptr=None, succ_cap=0):
    """Decodes this rank's shard of `graph` (a webgraph_amd.bvgraph.BVGraph) into device buffers; returns
    (lo, hi, arcs).  With rowptr_ptr None only the bounds are returned."""
    b = graph.shard_bounds(world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    if rowptr_ptr is None:
        return lo, hi, None
    arcs = graph.decode_range_device(lo, hi, rowptr_ptr, succ_ptr, succ_cap)
    return lo, hi, arcs
