"""Consumers that never hand the caller a successor array, against CPU restatements of the
reference loops they stand in for: the scan of Stats.run (src/it/unimi/dsi/webgraph/Stats.java:111-160) and the rounds of
ParallelBreadthFirstVisit (src/it/unimi/dsi/webgraph/algo/ParallelBreadthFirstVisit.java:146-247)."""
import numpy as np
import pytest

from conftest import CNR, make_graph

pytestmark = pytest.mark.gpu


def stats_restated(rp, sc, lo, hi):
    """Stats.java:111-160 over nodes [lo, hi) of the CSR (rp, sc), in numpy."""
    d = np.diff(rp[lo:hi + 1]).astype(np.int64)
    a0, a1 = int(rp[lo]), int(rp[hi])
    succ = sc[a0:a1].astype(np.int64)
    curr = np.repeat(np.arange(lo, hi, dtype=np.int64), d)
    out = {"nodes": hi - lo, "arcs": a1 - a0, "loops": int((succ == curr).sum()), "dangling": int((d == 0).sum())}
    first = np.zeros(hi - lo, dtype=np.int64)
    last = np.zeros(hi - lo, dtype=np.int64)
    nz = d > 0
    starts = (rp[lo:hi] - a0)[nz]
    first[nz] = succ[starts]
    last[nz] = succ[(rp[lo + 1:hi + 1] - a0)[nz] - 1]
    nodes = np.arange(lo, hi, dtype=np.int64)
    out["terminal"] = int((d == 0).sum() + ((d == 1) & (first == nodes)).sum())
    big = d > 1
    diff = (first - nodes)[big]
    out["num_gaps"] = int(d[big].sum())
    out["tot_gap"] = int((last - first)[big].sum() + np.where(diff >= 0, 2 * diff, -2 * diff - 1).sum())
    out["tot_loc"] = int(np.abs(succ - curr).sum())
    if hi > lo:
        out["min_outdegree"], out["min_outdegree_node"] = int(d.min()), lo + int(np.argmin(d))
        out["max_outdegree"] = int(d.max())
        out["max_outdegree_node"] = lo + int(np.argmax(d)) if d.max() > 0 else 0
    else:
        out.update(min_outdegree=0x7fffffff, min_outdegree_node=0, max_outdegree=0, max_outdegree_node=0)
    ad = np.abs(curr - succ)[succ != curr]
    msb = np.floor(np.log2(ad)).astype(np.int64) if ad.size else np.empty(0, dtype=np.int64)
    out["successor_delta_stats"] = [int(x) for x in np.bincount(msb, minlength=32)[:32]]
    return out


@pytest.fixture(scope="module")
def cnr_gpu():
    from webgraph_amd.bvgraph import BVGraph
    g = BVGraph.load(CNR)
    yield g
    g.close()


@pytest.mark.parametrize("lo,hi", [(0, 325557), (1000, 21000), (46000, 47000), (5, 6), (7, 7)])
def test_scan_stats_like_stats_run(cnr_gpu, cnr_oracle, lo, hi):
    og, rp, sc = cnr_oracle
    assert cnr_gpu.scan_stats(lo, hi) == stats_restated(rp, sc, lo, hi)


def test_scan_stats_indegrees_and_long_rows(tmp_path_factory):
    """Indegrees (Stats.java:130) accumulated on the device, on a graph with rows of tens of thousands of successors and loops."""
    import torch
    from webgraph_amd.bvgraph import BVGraph
    base, rowptr, succ = make_graph(tmp_path_factory, "stats", 300_000, 6_000_000, seed=5, p_copy=0.6)
    g = BVGraph.load(base)
    n = g.numNodes()
    indeg = torch.zeros(n, dtype=torch.int32, device="cuda")
    st = g.scan_stats(0, n, indeg.data_ptr())
    assert st == stats_restated(rowptr, succ, 0, n)
    assert np.array_equal(indeg.cpu().numpy(), np.bincount(succ, minlength=n).astype(np.int32))
    g.close()


def bfs_restated(rp, sc, start):
    n = rp.size - 1
    dist = np.full(n, -1, dtype=np.int64)
    dist[start] = 0
    frontier = np.array([start], dtype=np.int64)
    levels = [frontier]
    while frontier.size:
        nxt = np.unique(np.concatenate([sc[rp[x]:rp[x + 1]] for x in frontier]).astype(np.int64)) if frontier.size else frontier
        nxt = nxt[dist[nxt] == -1]
        dist[nxt] = len(levels)
        frontier = nxt
        if nxt.size:
            levels.append(nxt)
    return dist, levels


@pytest.mark.parametrize("parent", [False, True])
def test_breadth_first_visit(cnr_gpu, cnr_oracle, parent):
    """visit(start): queue[cutPoints[d]:cutPoints[d+1]] holds exactly the nodes at distance d; marker holds the round number
    or a parent in the visit tree (ParallelBreadthFirstVisit.java:46-66)."""
    og, rp, sc = cnr_oracle
    start = 100000
    dist, levels = bfs_restated(rp, sc, start)
    queue, cut, marker = cnr_gpu.bfs(start, parent=parent, round_=3)
    q = queue.cpu().numpy()
    assert cut[0] == 0 and cut[-1] == q.size == int((dist >= 0).sum()) and len(cut) - 1 == len(levels)
    for dlev, want in enumerate(levels):
        assert np.array_equal(np.sort(q[cut[dlev]:cut[dlev + 1]]), want)
    mk = marker.cpu().numpy()
    assert np.array_equal(mk == -1, dist == -1)
    if not parent:
        assert np.all(mk[dist >= 0] == 3)
    else:
        assert mk[start] == start
        vis = np.nonzero((dist > 0))[0]
        par = mk[vis].astype(np.int64)
        assert np.all(dist[par] == dist[vis] - 1)                           # the parent is one level up ...
        for x in vis[:: max(1, vis.size // 2000)]:                          # ... and really has the node among its successors
            p = int(mk[x])
            assert x in sc[rp[p]:rp[p + 1]]
    # a second visit from a marked node visits nothing (visit() returns 0 when marker[start] != -1)
    q2, cut2, _ = cnr_gpu.bfs(int(q[-1]), parent=parent, round_=4, marker=marker)
    assert q2.numel() == 0


def hyperball_restated(rp, sc, regs, modified, lo, hi):
    """HyperBall.java:875-915 for a standard iteration: register-wise max with the successors whose counter changed, self-loops skipped."""
    out = regs.copy()
    for x in range(lo, hi):
        s = sc[rp[x]:rp[x + 1]].astype(np.int64)
        s = s[s != x]
        if modified is not None:
            s = s[modified[s] != 0]
        if s.size:
            out[x] = np.maximum(regs[x], regs[s].max(axis=0))
    mod = np.zeros(regs.shape[0], dtype=np.uint8)
    mod[lo:hi] = (out[lo:hi] != regs[lo:hi]).any(axis=1)
    return out, mod


@pytest.mark.parametrize("log2m", [4, 7])
def test_hyperball_iterations(cnr_gpu, cnr_oracle, log2m):
    """Three iterations of the register-max step on the fixture (random 6-bit registers), every register and every `modified` flag against a
    numpy restatement; the second and third iterations only look at counters that changed, as HyperBall.java:909 does."""
    import torch
    og, rp, sc = cnr_oracle
    n, m = og.n, 1 << log2m
    rng = np.random.Generator(np.random.PCG64(99 + log2m))
    regs = (rng.integers(0, 64, size=(n, m)) * (rng.random((n, m)) < 0.2)).astype(np.uint8)
    d_in = torch.from_numpy(regs).cuda()
    d_out = d_in.clone()
    d_mod_in, d_mod_out = None, torch.zeros(n, dtype=torch.uint8, device="cuda")
    mod = None
    for it in range(3):
        changed = cnr_gpu.hyperball_step(log2m, d_in.data_ptr(), d_out.data_ptr(), d_mod_in.data_ptr() if d_mod_in is not None else None, d_mod_out.data_ptr())
        want, wmod = hyperball_restated(rp, sc, regs, mod, 0, n)
        assert np.array_equal(d_out.cpu().numpy(), want), "iteration %d" % it
        assert np.array_equal(d_mod_out.cpu().numpy(), wmod) and changed == int(wmod.sum())
        regs, mod = want, wmod
        d_in, d_out = d_out, d_in.clone()
        d_in = torch.from_numpy(regs).cuda()
        d_out = d_in.clone()
        d_mod_in, d_mod_out = torch.from_numpy(mod).cuda(), torch.zeros(n, dtype=torch.uint8, device="cuda")
    # a sub-range only writes its own counters
    d_in = torch.from_numpy(regs).cuda()
    d_out = torch.full_like(d_in, 255)
    cnr_gpu.hyperball_step(log2m, d_in.data_ptr(), d_out.data_ptr(), None, d_mod_out.data_ptr(), 1000, 3000)
    o = d_out.cpu().numpy()
    assert np.all(o[:1000] == 255) and np.all(o[3000:] == 255)
    assert np.array_equal(o[1000:3000], hyperball_restated(rp, sc, regs, None, 1000, 3000)[0][1000:3000])


@pytest.mark.parametrize("piece", ["0", "500000"])
def test_hyperball_and_bfs_with_runs_of_long_rows(tmp_path_factory, piece):
    """Rows of thousands of successors in runs of neighbours (what the deep-chain recipe is made of): the long rows of a piece are listed and dealt to the groups of sixteen
    waves (k_hyperball_big), the others go to a wave each; one iteration with every counter counting, one with flags, whole and in pieces, against the numpy restatement;
    a round of the visit from every 7th node appends the same set of nodes as the restatement (a block's winners leave in one append)."""
    import torch
    from webgraph_amd.bvgraph import BVGraph
    from webgraph_amd import tools as T
    base = str(tmp_path_factory.mktemp("hbruns") / "hbruns")
    rowptr, succ = T.generate(60_000, 3_000_000, seed=11, p_copy=0.85, p_same=0.95, p_keep=0.95)
    T.store(base, rowptr, succ)
    od = np.diff(rowptr)
    assert (od >= 2048).sum() >= 8 and ((od[1:] >= 2048) & (od[:-1] >= 2048)).any()
    g = BVGraph.load(base)
    g.set_option("scan_piece", piece)
    n, log2m = g.numNodes(), 6
    m = 1 << log2m
    rng = np.random.Generator(np.random.PCG64(7))
    regs = (rng.integers(0, 64, size=(n, m)) * (rng.random((n, m)) < 0.1)).astype(np.uint8)
    mod = None
    for it in range(2):
        d_in = torch.from_numpy(regs).cuda()
        d_out = d_in.clone()
        d_mod_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
        d_mod_in = torch.from_numpy(mod).cuda() if mod is not None else None
        changed = g.hyperball_step(log2m, d_in.data_ptr(), d_out.data_ptr(), d_mod_in.data_ptr() if d_mod_in is not None else None, d_mod_out.data_ptr())
        want, wmod = hyperball_restated(rowptr, succ, regs, mod, 0, n)
        assert np.array_equal(d_out.cpu().numpy(), want) and np.array_equal(d_mod_out.cpu().numpy(), wmod) and changed == int(wmod.sum()), "iteration %d" % it
        regs, mod = want, wmod
    frontier = np.arange(0, n, 7, dtype=np.int32)
    marker = np.full(n, -1, dtype=np.int32)
    marker[frontier] = 0
    d_marker = torch.from_numpy(marker).cuda()
    d_front = torch.from_numpy(frontier).cuda()
    d_out = torch.empty(n, dtype=torch.int32, device="cuda")
    cnt = g.bfs_expand(d_front.data_ptr(), frontier.size, d_marker.data_ptr(), 1, False, d_out.data_ptr(), n)
    reach = np.unique(np.concatenate([succ[rowptr[x]:rowptr[x + 1]] for x in frontier]))
    reach = reach[marker[reach] == -1]
    got = np.sort(d_out[:cnt].cpu().numpy())
    assert cnt == reach.size and np.array_equal(got, reach)
    assert np.array_equal(np.nonzero(d_marker.cpu().numpy() == 1)[0], reach)
    g.close()
