// bv_kernels.hip -- HIP kernels of the BVGraph decode path for gfx950 (MI355X).
//
// Pipeline of one range decode (nodes [from,to), optional halo [lo,from) of referents); DESIGN.md section 3:
//   k_headers        one lane per node: outdegree + reference fields                   (BVG:1048-1054)
//   k_mark_halo      transitive closure of the referents that live before `from`       (replaces the random-access
//                    window refill of BVGraphNodeIterator, BVG:1173-1183)
//   k_scan_*         exclusive prefix sum of the outdegrees -> CSR rowptr
//   k_depth_keys,    chain depth of every node (the file never stores it; maxrefcount is not trusted), compact
//   k_scatter_keys   work lists (per level for the copy pass, per work bin for the parse) and the copy-pass queues
//   k_classify,      records with >= coop_min / >= giant_min successors -> two queues, longest first
//   k_sort_desc
//   k_parse_list     one lane per record (lane-private LDS stream windows, bv_lanewin.hpp)
//   k_parse_big<1>   one wave per record   } cooperative decoder, bv_coop.hpp
//   k_parse_big<8>   eight waves per record}
//                    every parse kernel writes the record's "extra" successors (intervals + residuals, merged) to
//                    the TAIL of its CSR row                                            (BVG:1058-1100, :939-991)
//   k_copy_list_w,   for l = 1..maxdepth: rows whose chain depth is l merge the masked copy of their referent's
//   k_copy_mid,      (already final) row with their extras, in place: one lane / one wave / one 1024-thread group
//   k_copy_big       per row                                                 (MaskedIntIterator / MergedIntIterator)
//                    (k_copy_list_w: the 64 rows of a wave merged as ONE loop from the tables the parse left; k_copy_list: lane by lane)
//   k_chain_*, k_bparse, k_bcopy   the same for batches of random-access queries (slots of reference chains)
//   k_hash_*         ImmutableGraph.hashCode of a decoded CSR
// Older single-purpose variants kept as fallbacks behind knobs: k_parse, k_copy (node-order sweeps).  Tile kernels behind
// knobs: bv_tile.hpp, bv_tile2.hpp, bv_ctile.hpp.  All arithmetic is integer; nothing here is MFMA-shaped.
#include "bv_device.hpp"
#include "bv_launch.hpp"
#include "bv_coop.hpp"
#include "bv_lanewin.hpp"
#include "bv_tile.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace bv {

// A thread's item among up to 2^31 - 1 (a graph may have that many nodes, BVGraph.java:1537): the sum of block, tile and thread index is formed without sign and
// capped at INT32_MAX, which no count exceeds -- `s < cnt` then holds for real items only (as a plain int32 the last block's idle threads went negative and passed it).
__device__ __forceinline__ int32_t item_of(uint32_t i) { return (int32_t)(i < 0x7fffffffu ? i : 0x7fffffffu); }

typedef long long i64x2_a8 __attribute__((ext_vector_type(2), aligned(8))); // two neighbouring int64 (row starts, offsets) in one load
constexpr int TPB = 256;
constexpr int GIANT_NW = COOP_GIANT_NW; // waves per giant record
#ifndef COOPG_STATIC_LDS
#define COOPG_DYNLDS 1
#endif
#ifdef COOPG_DYNLDS
constexpr unsigned GIANT_DYN_LDS = CoopLds<COOP_GIANT_NW>::WORDS * 4;
#else
constexpr unsigned GIANT_DYN_LDS = 0;
#endif

template <int DEF, bool HASH = false>
__device__ __forceinline__ void copy_node(const GraphDev &g, int32_t x, int32_t d, int64_t dref, int32_t *__restrict__ row, const int32_t *__restrict__ src, int *__restrict__ err, uint32_t *hacc = nullptr, uint32_t hw = 0);
template <int DEF>
__device__ __forceinline__ void parse_node(const GraphDev &g, int32_t x, int32_t d, bool hasRef, int64_t dref, int32_t *__restrict__ row, int *__restrict__ err);
template <int DEF>
__device__ __forceinline__ void copy_node_v(const GraphDev &g, int32_t x, int32_t d, int32_t dref, int32_t *__restrict__ row, const int32_t *__restrict__ src, int *__restrict__ err);
template <bool VEC>
__device__ __forceinline__ void copy_node_tab(int32_t d, int32_t dref, int32_t *__restrict__ row, const int32_t *__restrict__ src, const int32_t *__restrict__ tabEnd, int4 hd);
__device__ __forceinline__ void copy_rows_tab(bool have, int32_t d, int32_t dref, int32_t *__restrict__ row, const int32_t *__restrict__ src, int32_t limE, int32_t limS, const int32_t *__restrict__ tabEnd, int4 hd);

// ------------------------------------------------------------------------------------------------ headers
__device__ __forceinline__ int32_t record_bin(uint64_t bitsLen);
template <int DEF>
__global__ void __launch_bounds__(TPB) k_headers(GraphDev g, int32_t lo, int32_t cnt, int32_t *__restrict__ outd,
                                                 uint16_t *__restrict__ ref, int *__restrict__ err, int32_t *__restrict__ part, uint8_t *__restrict__ mark,
                                                 uint16_t *__restrict__ pkey16, int32_t *__restrict__ phist, int32_t pwindows) {
	const int32_t s = item_of(blockIdx.x * TPB + threadIdx.x);
	uint64_t d = 0;
	uint64_t off0 = 0;
	if (s < cnt) {
		const int32_t x = lo + s;
		BitReader br;
		br.init(g.bits, g.nwords);
		off0 = (uint64_t)g.offsets[x];
		br.seek(off0);
		d = Fields<DEF>::outdegree(br, g);
		uint64_t r = 0;
		int e = 0;
		if (d > 0x7fffffffull) { e |= E_FORMAT; d = 0; }
		if (d > 0 && g.W > 0) {
			r = Fields<DEF>::reference(br, g);
			if (r > (uint64_t)g.W) { e |= E_REF; r = 0; }       // BVG:705
			else if (r > (uint64_t)x) { e |= E_FORMAT; r = 0; } // referent before node 0
		}
		e |= br.err;
		outd[s] = (int32_t)d;
		ref[s] = (uint16_t)r;
		if (mark && r > 0 && (uint64_t)s >= r) mark[s - (int32_t)r] = 1; // (bvg_scan_checksum: the rows somebody copies from are the ones that must exist in memory)
		if (e) atomicOr(err, e);
	}
	// How many of the block's records have >= 128, 256, ..., 8192 successors: k_pick_coop adds the blocks up and picks the
	// job's wave-class threshold.  (Per-block slots, no atomics on shared counters and no fences: a streaming kernel of its
	// own with a last-block-done ticket took 85 us on C2 -- its __threadfence() writes back an L2 full of fresh outdegrees.)
	if (part) {
		__shared__ int32_t s_c[PICK_LEVELS];
		if (threadIdx.x < PICK_LEVELS) s_c[threadIdx.x] = 0;
		__syncthreads();
		const int32_t dd = (int32_t)d;
		if (__ballot(dd >= 128)) {
#pragma unroll
			for (int k = 0; k < PICK_LEVELS; k++) { const int n = __popcll(__ballot(dd >= (128 << k))); if ((threadIdx.x & 63) == 0 && n) atomicAdd(&s_c[k], n); }
		}
		__syncthreads();
		if (threadIdx.x < PICK_LEVELS) part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_c[threadIdx.x];
	}
	// The parse list's keys (work bin inside a window of nodes: k_depth_keys with noBin = 6 / 2) while the header and the record's bounds are at hand: k_depth_keys re-read
	// 24 bytes per node for them, 70-80 us on C2 on the path to every parse kernel; the histogram of the keys is k_key_hist's, 2 bytes per node.  (Counted here -- per
	// block in LDS, one global addition per non-empty bin -- the blocks of a window, which run together, queued on the same two dozen words: k_headers 68 -> 540 us.)
	if (pkey16 && s < cnt) {
		uint16_t key = KEY_NONE;
		if (d > 0) {
			const uint64_t bitsLen = (uint64_t)g.offsets[lo + s + 1] - off0;
			const int32_t bin = record_bin(max(bitsLen, d * 8));
			const int32_t win = !pwindows ? 0 : bin >= PARSE_LONG_BIN ? MAXLVL - 1 : (int32_t)(((int64_t)s * MAXLVL) / cnt);
			key = (uint16_t)(win * NBIN + bin);
		}
		pkey16[s] = key;
	}
}

// ------------------------------------------------------------------------------------------------ halo closure
__global__ void k_mark_halo(int32_t nh, int32_t cnt, int32_t W, const int32_t *__restrict__ outd,
                            const uint16_t *__restrict__ ref, uint8_t *__restrict__ need, int *__restrict__ err) {
	const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= W || nh + t >= cnt) return;
	int32_t y = nh + t;
	while (outd[y] > 0 && ref[y] > 0) {
		const int32_t y2 = y - (int32_t)ref[y];
		if (y2 < 0) { atomicOr(err, E_ESCAPED); break; } // chain leaves the halo window: caller retries with a larger one
		if (y2 < nh) need[y2] = 1;
		y = y2;
	}
}

__global__ void k_apply_need(int32_t nh, const uint8_t *__restrict__ need, int32_t *__restrict__ outd, uint16_t *__restrict__ ref) {
	const int32_t s = item_of(blockIdx.x * blockDim.x + threadIdx.x);
	if (s >= nh) return;
	if (!need[s]) { outd[s] = 0; ref[s] = 0; }
	else if ((int32_t)ref[s] > s) ref[s] = 0; // a needed chain leaves the window (k_mark_halo raised E_ESCAPED): nothing before it may be touched
}

// ------------------------------------------------------------------------------------------------ dense batches
// A batch of random-access queries that touches a good part of the graph is decoded as a MASKED scan: every queried
// node and every node on its reference chain is marked, the others get outdegree 0 (k_apply_need) and drop out of
// the scan; each needed record is then decoded ONCE, however many queries (or chains) want it, and the rows are
// gathered into the caller's order at the end (k_gather_rows).
__global__ void k_query_mark(const int32_t *__restrict__ nodes, int64_t q, int32_t n, const int32_t *__restrict__ outd, const uint16_t *__restrict__ ref,
                             uint8_t *need, int32_t *__restrict__ qoutd, int walk, int *__restrict__ err) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= q) return;
	const int32_t x = nodes[i];
	if ((uint32_t)x >= (uint32_t)n) { atomicOr(err, E_ARG); qoutd[i] = 0; return; } // BVG:900
	qoutd[i] = outd[x];
	if (!need) return; // count only
	if (!walk) { need[x] = 1; return; } // (the closure follows in streaming passes, k_need_prop)
	// few queries: walk the chain; whoever marks a node first goes on from there, so a marked node ends the walk
	int32_t y = x;
	while (!need[y]) {
		need[y] = 1;
		if (outd[y] <= 0 || ref[y] == 0) break;
		y -= (int32_t)ref[y]; // >= 0: k_headers drops references before node 0
	}
}
// One step of the closure of the marks under "is copied from": every marked node marks its referent.  The passes
// stream the three arrays (a chain walk per query is three random accesses per step: 0.6 ms for 10 M queries,
// against 25 us per pass here); `changed` tells the host whether the closure had not been reached before this pass.
__global__ void k_need_prop(int32_t n, const int32_t *__restrict__ outd, const uint16_t *__restrict__ ref, uint8_t *need, int32_t *__restrict__ changed) {
	const int32_t s = item_of(blockIdx.x * blockDim.x + threadIdx.x);
	if (s >= n || !need[s] || outd[s] <= 0) return;
	const int32_t r = ref[s];
	if (r == 0) return;
	const int32_t t = s - r; // >= 0: k_headers drops references before node 0
	if (!need[t]) { need[t] = 1; if (changed) *changed = 1; }
}

constexpr int GATHER_ROWS = 256;
__global__ __launch_bounds__(GATHER_ROWS) void k_gather_rows(const int32_t *__restrict__ nodes, int64_t q, const int64_t *__restrict__ rowstart, const int32_t *__restrict__ arena,
                                                             const int64_t *__restrict__ rowptr, int32_t *__restrict__ succ) {
	__shared__ int64_t src[GATHER_ROWS], dst[GATHER_ROWS + 1];
	const int64_t base = (int64_t)blockIdx.x * GATHER_ROWS;
	const int t = threadIdx.x;
	const int64_t i = min(base + t, q - 1);
	src[t] = rowstart[nodes[i]];
	dst[t] = rowptr[min(base + t, q)];
	if (t == 0) dst[GATHER_ROWS] = rowptr[min(base + GATHER_ROWS, q)];
	__syncthreads();
	// four consecutive ids per lane and step, on 16-byte boundaries of the output: one search per four ids
	const int64_t d0 = dst[0], d1 = dst[GATHER_ROWS];
	const int64_t mis = (int64_t)(((uintptr_t)succ >> 2) & 3); // the caller's buffer need not be 16-byte aligned
	// (a few queries for very long rows: gridDim.y blocks share the ids of these rows, each a contiguous run of quads)
	const int64_t A = ((d0 + mis) & ~(int64_t)3) - mis, quads = (d1 - A + 3) >> 2, qper = (quads + gridDim.y - 1) / gridDim.y;
	const int64_t qend = min(quads, (int64_t)(blockIdx.y + 1) * qper);
	for (int64_t qd = (int64_t)blockIdx.y * qper + t; qd < qend; qd += GATHER_ROWS) {
		const int64_t pos = A + 4 * qd;
		const int64_t first = max(pos, d0);
		int j = 0; // last row of the block that starts at or before `first`
#pragma unroll
		for (int step = GATHER_ROWS / 2; step > 0; step >>= 1) if (dst[j + step] <= first) j += step;
		int32_t val[4];
		if (pos >= d0 && dst[j + 1] >= pos + 4) { // all four from row j: one 16-byte load (dword-aligned is enough for global loads)
			typedef int32_t int4u __attribute__((ext_vector_type(4), aligned(4)));
			const int4u q4 = *(const int4u *)(arena + src[j] + (pos - dst[j]));
			*(int4 *)(succ + pos) = int4{ q4.x, q4.y, q4.z, q4.w };
			continue;
		}
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int64_t p = pos + k;
			val[k] = 0;
			if (p >= d0 && p < d1) {
				while (dst[j + 1] <= p) j++; // (dst[GATHER_ROWS] = d1 > p ends it)
				val[k] = arena[src[j] + (p - dst[j])];
			}
		}
		if (pos >= d0 && pos + 4 <= d1) *(int4 *)(succ + pos) = int4{ val[0], val[1], val[2], val[3] };
		else {
#pragma unroll
			for (int k = 0; k < 4; k++) if (pos + k >= d0 && pos + k < d1) succ[pos + k] = val[k];
		}
	}
}

// ------------------------------------------------------------------------------------------------ scan
// Three-phase exclusive scan int32 -> int64 (block sums, scan of the sums, block scan + carry).
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = TPB * SCAN_ITEMS;

__device__ __forceinline__ int64_t wave_incl_scan(int64_t v) {
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int64_t t = __shfl_up(v, o, 64);
		if (lane >= o) v += t;
	}
	return v;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int64_t block_excl_scan(int64_t v, int64_t *total) {
	__shared__ int64_t wsum[TPB / 64];
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const int64_t inc = wave_incl_scan(v);
	if (lane == 63) wsum[wid] = inc;
	__syncthreads();
	int64_t base = 0, tot = 0;
#pragma unroll
	for (int i = 0; i < TPB / 64; i++) { if (i < wid) base += wsum[i]; tot += wsum[i]; }
	__syncthreads();
	*total = tot;
	return base + inc - v;
}

__global__ void __launch_bounds__(TPB) k_scan_sums(const int32_t *__restrict__ in, int64_t n, int64_t *__restrict__ sums) {
	const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
	int64_t v = 0;
#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; i++) { const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i; if (j < n) v += in[j]; }
	int64_t tot;
	block_excl_scan(v, &tot);
	if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single block: exclusive scan of nb block sums in place (loops in tiles of TPB with a carry)
__global__ void __launch_bounds__(TPB) k_scan_top(int64_t *__restrict__ sums, int64_t nb) {
	// one block, one barrier round: every thread owns a contiguous run of the tile sums (the kernel sits between two
	// grid-wide kernels of every scan: a loop of block scans, two barriers per 256 sums, made it last 30 us)
	const int64_t per = (nb + TPB - 1) / TPB, lo = min(per * threadIdx.x, nb), hi = min(lo + per, nb);
	int64_t mine = 0;
	for (int64_t j = lo; j < hi; j++) mine += sums[j];
	int64_t tot;
	int64_t run = block_excl_scan(mine, &tot);
	for (int64_t j = lo; j < hi; j++) { const int64_t v = sums[j]; sums[j] = run; run += v; }
}

// The same for many sums (ranges of tens of millions of nodes: the kernel above grows with nb -- 342 us for the 48 829 sums of a 50 M-node scan, on the
// chain in front of every parse kernel): tiles of 1 024 sums with a carry, coalesced loads, the next tile's in flight while this one is scanned.
// (Not for small ranges: there the scan of the outdegrees must not end before the parse list is built -- the giants start behind it and their
// groups, a CU each, starve k_scatter_keys; profiles/r4_experiments.txt section 9.)
constexpr int SCAN_TOP_T = 1024, SCAN_TOP_I = 4, SCAN_TOP_TILED_MIN = 1024; // (round 6: from 1 M nodes on -- the giants now WAIT for the parse list, giants_after_list, so a faster scan no longer lets them starve its scatter)
__global__ void __launch_bounds__(SCAN_TOP_T) k_scan_top_tiled(int64_t *__restrict__ sums, int64_t nb) {
	__shared__ int64_t wsum[SCAN_TOP_T / 64];
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	constexpr int TILE = SCAN_TOP_T * SCAN_TOP_I; // (a thread owns SCAN_TOP_I consecutive sums of a tile: 32 bytes)
	int64_t carry = 0;
	int64_t vn[SCAN_TOP_I];
#pragma unroll
	for (int i = 0; i < SCAN_TOP_I; i++) { const int64_t j = (int64_t)threadIdx.x * SCAN_TOP_I + i; vn[i] = j < nb ? sums[j] : 0; }
	for (int64_t base = 0; base < nb; base += TILE) {
		int64_t v[SCAN_TOP_I], mine = 0;
#pragma unroll
		for (int i = 0; i < SCAN_TOP_I; i++) { v[i] = vn[i]; mine += v[i]; }
#pragma unroll
		for (int i = 0; i < SCAN_TOP_I; i++) { const int64_t j = base + TILE + (int64_t)threadIdx.x * SCAN_TOP_I + i; vn[i] = j < nb ? sums[j] : 0; }
		const int64_t inc = wave_incl_scan(mine);
		if (lane == 63) wsum[wid] = inc;
		__syncthreads();
		int64_t wbase = 0, tot = 0;
#pragma unroll
		for (int i = 0; i < SCAN_TOP_T / 64; i++) { const int64_t w = wsum[i]; if (i < wid) wbase += w; tot += w; }
		__syncthreads();
		int64_t run = carry + wbase + inc - mine;
#pragma unroll
		for (int i = 0; i < SCAN_TOP_I; i++) { const int64_t j = base + (int64_t)threadIdx.x * SCAN_TOP_I + i; if (j < nb) sums[j] = run; run += v[i]; }
		carry += tot;
	}
}

template <bool HASH>
__global__ void __launch_bounds__(TPB) k_scan_apply(const int32_t *__restrict__ in, int64_t n, const int64_t *__restrict__ sums, int64_t *__restrict__ out, const HashCtx *__restrict__ hxp, int32_t lo, int32_t nh) {
	const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
	uint32_t hacc = 0;
	int64_t vals[SCAN_ITEMS];
	int64_t v = 0;
#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; i++) { const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i; vals[i] = j < n ? in[j] : 0; v += vals[i]; }
	int64_t tot;
	int64_t ex = block_excl_scan(v, &tot) + sums[blockIdx.x];
#pragma unroll
	for (int i = 0; i < SCAN_ITEMS; i++) {
		const int64_t j = base + (int64_t)threadIdx.x * SCAN_ITEMS + i;
		if (j < n) out[j] = ex;
		if (HASH && j < n && j >= nh) hacc += (uint32_t)(lo + (int32_t)j) * hash_upow(hxp->ptab, (uint64_t)(1 + ex + j)); // (bvg_scan_checksum: the node's own number, HashCtx)
		ex += vals[i];
		if (j == n - 1) out[n] = ex;
	}
	if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
	if (HASH) {
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) hacc += (uint32_t)__shfl_xor((int)hacc, o, 64);
		if ((threadIdx.x & 63) == 0) hash_add(*hxp, hacc);
	}
}

// ------------------------------------------------------------------------------------------------ chain depth
__global__ void k_rebase(int32_t nh, int32_t cnt, const int64_t *__restrict__ rowstart, int64_t *__restrict__ out) {
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j <= (int64_t)(cnt - nh)) out[j] = rowstart[nh + j] - rowstart[nh];
}

// ------------------------------------------------------------------------------------------------ parse
// One lane per node.  Decodes everything that does not depend on the referent's CONTENT:
//   copied   = how many successors will come from the referent (needs only the referent's outdegree, BVG:1069)
//   extras   = intervals U residuals, merged, written to row[copied .. d)
// Nodes without a reference are final after this kernel.
// Both cursors refill 32 bits at a time straight from HBM (L1/L2-cached).  Measured alternatives on C2 -- a 16-byte
// prefetching cursor and private per-lane LDS windows -- tripled the register count (148-188 VGPRs) and were slower.
template <int DEF>
__device__ __forceinline__ void parse_node(const GraphDev &g, int32_t x, int32_t d, bool hasRef, int64_t dref, int32_t *__restrict__ row, int *__restrict__ err) {
	BitReader br;
	br.init(g.bits, g.nwords);
	br.seek((uint64_t)g.offsets[x]);
	(void)Fields<DEF>::outdegree(br, g);
	if (g.W > 0) (void)Fields<DEF>::reference(br, g);

	int e = 0;
	int64_t copied = 0;
	if (hasRef) {
		const uint64_t bc = Fields<DEF>::block_count(br, g);
		int64_t total = 0;
		if (bc > (uint64_t)dref + 1) e |= E_FORMAT;
		else {
			for (uint64_t b = 0; b < bc; b++) {
				int64_t len;
				if (!block_len_ok(Fields<DEF>::block(br, g), b == 0, total, dref, len)) { e |= E_FORMAT; break; }
				total += len;
				if (!(b & 1)) copied += len;
			}
			if (!(bc & 1)) copied += dref - total;
		}
	}
	const int64_t extra = (int64_t)d - copied;
	if (extra < 0 || copied < 0) e |= E_FORMAT;
	if (e || br.err) { atomicOr(err, e | br.err); return; }
	if (extra == 0) return;

	// interval section: skip-parse to find the residual section and the number of residuals
	int64_t nIntervals = 0, intervalArcs = 0;
	BitReader bi; // second cursor, re-reads the interval section lazily during the merge
	bi.init(g.bits, g.nwords);
	if (g.minInt != 0) {
		nIntervals = (int64_t)br.gamma();
		if (nIntervals > extra) { atomicOr(err, E_FORMAT); return; }
		if (nIntervals) {
			bi.seek(br.pos());
			for (int64_t i = 0; i < nIntervals; i++) {
				(void)br.gamma();
				const uint64_t len = br.gamma();
				if (len > (uint64_t)extra) { br.err |= E_FORMAT; break; } // (any 64-bit value in a malformed stream: kept out of the sum)
				intervalArcs += (int64_t)len + g.minInt;
			}
		}
	}
	const int64_t nRes = extra - intervalArcs;
	if (nRes < 0 || br.err) { atomicOr(err, E_FORMAT | br.err); return; }

	// merge(intervals, residuals) -> row[copied ..)
	int32_t *out = row + copied;
	int64_t k = 0;
	const int64_t head = min<int64_t>(extra, (int64_t)(((16u - ((uint32_t)(uintptr_t)out & 15u)) & 15u) >> 2)); // scalar stores up to the first 16-byte boundary
	int32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, on = 0;
	int64_t ivLeft = 0, ivRem = 0, ivPrev = 0; // current interval: next value, values left; end of the previous interval
	int64_t ivTodo = nIntervals;
	bool firstIv = true;
	int64_t resTodo = nRes;
	int64_t resVal = 0;
	if (resTodo) resVal = (int64_t)(int32_t)((int64_t)x + nat2int(Fields<DEF>::residual(br, g))); // BVG:954
	while (k < extra) {
		if (ivRem == 0 && ivTodo) { // load the next interval (BVG:1084-1093)
			if (firstIv) { ivLeft = (int64_t)(int32_t)((int64_t)x + nat2int(bi.gamma())); firstIv = false; }
			else ivLeft = ivPrev + (int64_t)bi.gamma() + 1;
			ivRem = (int64_t)bi.gamma() + g.minInt;
			ivPrev = ivLeft + ivRem;
			ivTodo--;
		}
		int32_t val;
		if (ivRem && (!resTodo || ivLeft < resVal)) { val = (int32_t)ivLeft; ivLeft++; ivRem--; }
		else if (resTodo) {
			val = (int32_t)resVal;
			if (ivRem && ivLeft == resVal) { ivLeft++; ivRem--; } // equal heads are emitted once (MergedIntIterator.java:69-72)
			if (--resTodo) resVal += (int64_t)Fields<DEF>::residual(br, g) + 1; // BVG:966
		} else val = -1; // malformed: fewer values than the outdegree promises (BVG:1210 would store -1)
		// 16-byte stores where the row allows it: the 64 lanes of a wave write 64 different rows, and with half a
		// million rows in flight a 4-byte store per successor lets the L2 evict every line several times before it
		// is complete (rocprof: WRITE_SIZE 6x the algorithmic bytes of this kernel)
		if (k < head) { out[k++] = val; continue; }
		o0 = o1; o1 = o2; o2 = o3; o3 = val; k++;
		if (++on == 4) { *(int4 *)(out + k - 4) = int4{ o0, o1, o2, o3 }; on = 0; }
	}
	if (on == 3) { out[k - 3] = o1; out[k - 2] = o2; out[k - 1] = o3; }
	else if (on == 2) { out[k - 2] = o2; out[k - 1] = o3; }
	else if (on == 1) out[k - 1] = o3;
	if (br.err | bi.err) atomicOr(err, br.err | bi.err);
}

__device__ __forceinline__ int32_t record_bin(uint64_t bitsLen) { // half-octave steps (BIN_SUB = 1) from 16 bits up: the lanes of a wave differ by < 1.5x
	const int lg = 63 - __clzll((long long)(bitsLen | 1));
	const int h = (lg << BIN_SUB) + (lg >= BIN_SUB ? (int)((bitsLen >> (lg - BIN_SUB)) & ((1u << BIN_SUB) - 1u)) : 0);
	constexpr int BASE = 4 << BIN_SUB; // (16 bits: lg = 4)
	return h < BASE ? 0 : h - BASE >= NBIN ? NBIN - 1 : h - BASE;
}

// A row that copies from a very long referent can have a block list of thousands of codes, whatever its own length:
// walked by one lane (or one wave) it would be the tail of its kernel, so it goes to the group class, whose walk
// is cooperative (coop_block_walk).
constexpr int COPY_REF_BIG = 8192;
__device__ __forceinline__ int copy_class_of(int32_t d, int32_t dref, int32_t midMin, int32_t bigMin) {
	if (d >= bigMin || (dref >= COPY_REF_BIG && bigMin != 0x7fffffff)) return 3;
	return d >= midMin ? 2 : 1;
}
constexpr int WINDOWED_BINS = PARSE_LONG_BIN; // work < 2048 bits (bin 14): binned per window of nodes (noBin & 4)
constexpr int LIST_ITEMS = 16, LIST_TILE = TPB * LIST_ITEMS; // slots per block: few blocks -> few same-address atomics (~88 M/s each)

__global__ void __launch_bounds__(TPB) k_depth_keys(GraphDev g, int32_t lo, int32_t cnt, const int32_t *__restrict__ outd, const uint16_t *__restrict__ ref,
                                                    uint64_t giantBits, int32_t noBin, int32_t *__restrict__ depth, uint16_t *__restrict__ key16,
                                                    int32_t *__restrict__ hist, int32_t *__restrict__ ctl, int32_t *__restrict__ maxdepth,
                                                    int32_t *__restrict__ bigQ, int32_t bigCap, int32_t *__restrict__ midQ, int32_t midCap, int32_t midMin, int32_t bigMin) {
	__shared__ int32_t s_hist[NKEYS + 1];
	__shared__ int32_t s_q[LIST_TILE], s_qn[2], s_qbase[2]; // rows for the copy queues: wave class from the front, group class from the back
	for (int k = threadIdx.x; k <= NKEYS; k += TPB) s_hist[k] = 0;
	if (threadIdx.x < 2) s_qn[threadIdx.x] = 0;
	__syncthreads();
	for (int it = 0; it < LIST_ITEMS; it++) {
		const int32_t s = item_of(blockIdx.x * LIST_TILE + it * TPB + threadIdx.x);
		if (s >= cnt) break;
		int32_t dd = 0;
		if (!(noBin & 2)) { // (a list that ignores the level needs no depth: `depth` may be NULL)
			int32_t y = s;
			for (;;) { const int32_t r = ref[y]; if (r == 0) break; y -= r; dd++; } // ref[] is 0 for empty / unneeded nodes
			depth[s] = dd;
		}
		uint16_t key = KEY_NONE;
		if (outd[s] > 0) {
			// work of a record ~ codes to parse + successors to emit: a short record can still expand to a huge
			// list through intervals and copy blocks, so weigh the outdegree in (8 bits per successor)
			const uint64_t bitsLen = (uint64_t)(g.offsets[lo + s + 1] - g.offsets[lo + s]);
			const uint64_t work = max(bitsLen, (uint64_t)outd[s] * 8);
			if (work >= giantBits) key = KEY_GIANT;
#ifdef BV_EXP_DROP_LO // (ablation builds, scripts/ablate.sh: the records of an outdegree range are not decoded at all -- what the range costs inside the overlapped scan)
			else if ((noBin & 4) && outd[s] >= BV_EXP_DROP_LO && outd[s] < BV_EXP_DROP_HI) key = KEY_NONE;
#endif
			else if (noBin & 4) {
				// parse list: short records keep their neighbourhood -- sorted by work bin inside MAXLVL windows of
				// consecutive nodes (a sweep of the parse kernel then touches a few windows of stream and rows, not
				// the whole graph); the long ones (bin >= WINDOWED_BINS) stay together at the end: they are swept first
				const int32_t bin = record_bin(work);
				const int32_t win = bin >= WINDOWED_BINS ? MAXLVL - 1 : (int32_t)(((int64_t)s * MAXLVL) / cnt);
				key = (uint16_t)(win * NBIN + bin);
			} else key = (uint16_t)(((noBin & 2) ? 0 : min(dd, MAXLVL - 1)) * NBIN + ((noBin & 1) ? 0 : record_bin(work))); // noBin: bit 0 = ignore the length, bit 1 = ignore the level
			atomicAdd(&s_hist[key == KEY_GIANT ? NKEYS : key], 1);
			if (dd >= MAXLVL - 1 && dd > __builtin_nontemporal_load(maxdepth)) atomicMax(maxdepth, dd); // only very deep chains get here
		}
		key16[s] = key;
		// rows with a reference that the copy pass will merge with a wave (ctl[6]) or a whole group (ctl[5]) each are
		// queued once, for all levels; collected per block first (same-address global atomics run at ~88 M/s)
		if (bigQ && ref[s]) {
			const int cls = copy_class_of(outd[s], outd[s - ref[s]], midMin, bigMin);
			if (cls == 3) s_q[LIST_TILE - 1 - atomicAdd(&s_qn[1], 1)] = s;
			else if (cls == 2) s_q[atomicAdd(&s_qn[0], 1)] = s;
		}
	}
	__syncthreads();
	if (bigQ) {
		if (threadIdx.x < 2 && s_qn[threadIdx.x]) s_qbase[threadIdx.x] = atomicAdd(&ctl[threadIdx.x ? 5 : 6], s_qn[threadIdx.x]);
		__syncthreads();
		for (int k = threadIdx.x; k < s_qn[0]; k += TPB) if (s_qbase[0] + k < midCap) midQ[s_qbase[0] + k] = s_q[k]; // (the caps of bvgpu_api.cpp cannot be exceeded)
		for (int k = threadIdx.x; k < s_qn[1]; k += TPB) if (s_qbase[1] + k < bigCap) bigQ[s_qbase[1] + k] = s_q[LIST_TILE - 1 - k];
	}
	__syncthreads();
	for (int k = threadIdx.x; k <= NKEYS; k += TPB) { const int32_t c = s_hist[k]; if (c) atomicAdd(k == NKEYS ? &ctl[1] : &hist[k], c); }
}

// single block: keyBase = exclusive scan of hist; cursor = keyBase; deepest level present -> maxdepth
__global__ void __launch_bounds__(TPB) k_key_offsets(const int32_t *__restrict__ hist, int32_t *__restrict__ keyBase, int32_t *__restrict__ cursor, int32_t *__restrict__ maxdepth) {
	__shared__ int32_t s_top;
	if (threadIdx.x == 0) s_top = 0;
	__syncthreads();
	constexpr int PER = (NKEYS + TPB - 1) / TPB; // every thread owns a run of keys: one barrier round
	const int lo = min(PER * (int)threadIdx.x, NKEYS), hi = min(lo + PER, NKEYS);
	int64_t mine = 0;
	for (int k = lo; k < hi; k++) mine += hist[k];
	int64_t tot;
	int64_t run = block_excl_scan(mine, &tot);
	for (int k = lo; k < hi; k++) {
		const int32_t v = hist[k];
		keyBase[k] = (int32_t)run; cursor[k] = (int32_t)run;
		if (v) atomicMax(&s_top, k / NBIN);
		run += v;
	}
	__syncthreads();
	if (threadIdx.x == 0) { keyBase[NKEYS] = (int32_t)tot; atomicMax(maxdepth, s_top); }
}

// histogram of the keys that k_headers wrote (the parse list of a scan without a halo)
__global__ void __launch_bounds__(TPB) k_key_hist(int32_t cnt, const uint16_t *__restrict__ key16, int32_t *__restrict__ hist) {
	__shared__ int32_t s_hist[NKEYS];
	for (int k = threadIdx.x; k < NKEYS; k += TPB) s_hist[k] = 0;
	__syncthreads();
#pragma unroll
	for (int it = 0; it < LIST_ITEMS; it++) {
		const int32_t s = item_of(blockIdx.x * LIST_TILE + it * TPB + threadIdx.x);
		const uint16_t key = s < cnt ? key16[s] : KEY_NONE;
		if (key < NKEYS) atomicAdd(&s_hist[key], 1);
	}
	__syncthreads();
	for (int k = threadIdx.x; k < NKEYS; k += TPB) { const int32_t c = s_hist[k]; if (c) atomicAdd(&hist[k], c); }
}

// packRef: the list's entries carry min(ref, LIST_REF_ESC) in bits 28 .. 31 (the parse list of a job of fewer than 2^28 slots: k_parse_list)
constexpr int32_t LIST_SLOT_MASK = 0x0fffffff, LIST_REF_ESC = 15;
__global__ void __launch_bounds__(TPB) k_scatter_keys(int32_t cnt, const uint16_t *__restrict__ key16, int32_t *__restrict__ cursor, int32_t *__restrict__ list,
                                                      int32_t *__restrict__ giantlist, int32_t giantCap, int32_t *__restrict__ ctl, const uint16_t *__restrict__ packRef = nullptr) {
	__shared__ int32_t s_cnt[NKEYS + 1], s_base[NKEYS + 1];
	for (int k = threadIdx.x; k <= NKEYS; k += TPB) s_cnt[k] = 0;
	__syncthreads();
	uint16_t keys[LIST_ITEMS];
	int32_t local[LIST_ITEMS];
#pragma unroll
	for (int it = 0; it < LIST_ITEMS; it++) {
		const int32_t s = item_of(blockIdx.x * LIST_TILE + it * TPB + threadIdx.x);
		keys[it] = s < cnt ? key16[s] : KEY_NONE;
		local[it] = keys[it] != KEY_NONE ? atomicAdd(&s_cnt[keys[it] == KEY_GIANT ? NKEYS : keys[it]], 1) : 0;
	}
	__syncthreads();
	for (int k = threadIdx.x; k <= NKEYS; k += TPB) { const int32_t c = s_cnt[k]; if (c) s_base[k] = atomicAdd(k == NKEYS ? &ctl[4] : &cursor[k], c); }
	__syncthreads();
#pragma unroll
	for (int it = 0; it < LIST_ITEMS; it++) {
		const int32_t s = item_of(blockIdx.x * LIST_TILE + it * TPB + threadIdx.x);
		if (keys[it] == KEY_GIANT) { const int32_t k = s_base[NKEYS] + local[it]; if (k < giantCap) giantlist[k] = s; }
		else if (keys[it] != KEY_NONE) list[s_base[keys[it]] + local[it]] = packRef ? (int32_t)((uint32_t)s | ((uint32_t)min((int32_t)packRef[s], LIST_REF_ESC) << 28)) : s;
	}
}

// The copy pass of one chain level runs as three kernels side by side over the level's compact list, each picking
// the rows of its class: rows with fewer than midMin successors are merged by one lane each (k_copy_list), rows
// with fewer than COPY_BIG_MIN by one wave each (k_copy_mid), longer ones by a 1024-thread group each (k_copy_big).
// A lane-serial merge of a long row would be the tail of the whole scan.
constexpr int COPY_BIG_MIN = 1024;
// class of row s at this level: 0 nothing to do, 1 one lane, 2 one wave, 3 one group
__device__ __forceinline__ int copy_class(const RangeView &v, const int32_t *__restrict__ depth, int32_t level, int32_t s, int32_t midMin, int32_t bigMin) {
	if (level >= MAXLVL - 1 && depth[s] != level) return 0; // shared overflow bucket
	if (v.ref[s] == 0) return 0;
	if (!v.fits(s) || !v.fits(s - v.ref[s])) return 0; // E_CAP / E_HALO already raised by the parse kernel
	return copy_class_of(v.outd[s], v.outd[s - v.ref[s]], midMin, bigMin);
}
// TAB: the rows that the one-lane parse decoded (fewer than coopMin successors) carry their copy blocks as a table at the end of their slice of the interval arena
// (parse_node_lwc / parse_node_tile, bv_lanewin.hpp): the merge reads neither the stream nor the offsets and runs no bit reader (copy_node_tab).
template <int DEF, bool VEC, bool HASH = false, bool TAB = false>
__global__ void __launch_bounds__(TPB) k_copy_list(GraphDev g, RangeView v, const int32_t *__restrict__ depth, const int32_t *__restrict__ list,
                                                   const int32_t *__restrict__ keyBase, int32_t level, int32_t midMin, int32_t bigMin, int *__restrict__ err,
                                                   const IvEntry *__restrict__ arena = nullptr, int64_t arenaCap = 0, const CopyTab *__restrict__ ctab = nullptr) {
	const int32_t bucket = min(level, MAXLVL - 1);
	const int32_t lo = keyBase[bucket * NBIN], hi = keyBase[(bucket + 1) * NBIN];
	const int64_t rsNh = v.rowstart[v.nh];
	const int32_t coopMin = TAB ? v.coopmin() : 0;
	// HASH (bvg_scan_checksum): the rows merged here are added to the job's hash as they are written (copy_node<., true>)
	const HashCtx hx = HASH ? *v.hx : HashCtx{};
	uint32_t hacc = 0;
	for (int32_t idx = hi - 1 - (blockIdx.x * TPB + threadIdx.x); idx >= lo; idx -= gridDim.x * TPB) {
		// (copy_class and RangeView::row / ::fits spelled out: every load of this kernel goes to a line of its own, so each is issued once --
		// the outdegrees are differences of the row starts, which are needed anyway)
		const int32_t s = list[idx];
		const int32_t r = v.ref[s];
		if (r == 0 || (level >= MAXLVL - 1 && depth[s] != level)) continue;
		const int32_t t = s - r;
		const i64x2_a8 rsp = *(const i64x2_a8 *)(v.rowstart + s), rtp = *(const i64x2_a8 *)(v.rowstart + t); // (a 16-byte load per pair of row starts)
		const int64_t rs0 = rsp.x, rs1 = rsp.y, rt0 = rtp.x, rt1 = rtp.y;
		if (!(s >= v.nh ? (uint64_t)(rs1 - rsNh) <= v.succ_cap : (uint64_t)rs1 <= v.halo_cap) || !(t >= v.nh ? (uint64_t)(rt1 - rsNh) <= v.succ_cap : (uint64_t)rt1 <= v.halo_cap)) continue; // E_CAP / E_HALO already raised by the parse kernel
		const int32_t d = (int32_t)(rs1 - rs0), dref = (int32_t)(rt1 - rt0);
		if (copy_class_of(d, dref, midMin, bigMin) != 1) continue;
		int32_t *row = s < v.nh ? v.halo + rs0 : v.succ + (rs0 - rsNh);
		const int32_t *src = t < v.nh ? v.halo + rt0 : v.succ + (rt0 - rsNh);
		if (TAB && d < coopMin) {
			const int4 hd = *(const int4 *)(ctab + s);
			const uint32_t kept = (uint32_t)hd.w & 0xffffu;
			if (kept != CT_NONE) {
				const int32_t *ovfEnd = nullptr;
				bool ok = true;
				if (kept > 3) { // the kept blocks from the fourth on: the end of the record's own part of the interval arena
					int64_t abase = 0; int32_t an = 0;
					if (g.minInt > 0) arena_slice(g.minInt, rs0, d, abase, an);
					ok = g.minInt > 0 && abase >= 0 && abase + an <= arenaCap && (int32_t)kept - 3 <= 4 * (an - 1);
					ovfEnd = (const int32_t *)(arena + abase + (an - 1));
				}
				if (ok) { copy_node_tab<VEC>(d, dref, row, src, ovfEnd, hd); continue; }
			}
		}
		if (HASH) copy_node<DEF, true>(g, v.lo + s, d, (int64_t)dref, row, src, err, &hacc, s >= v.nh ? hash_upow(hx.ptab, (uint64_t)(1 + rs1 + (int64_t)s)) : 0u);
		else if (VEC) copy_node_v<DEF>(g, v.lo + s, d, dref, row, src, err);
		else copy_node<DEF>(g, v.lo + s, d, (int64_t)dref, row, src, err);
	}
	if (HASH) {
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) hacc += (uint32_t)__shfl_xor((int)hacc, o, 64);
		if ((threadIdx.x & 63) == 0) hash_add(hx, hacc);
	}
}

// k_copy_list<., ., false, true> with the merge as a loop the 64 lanes of a wave walk TOGETHER (copy_rows_tab): in copy_node_tab every lane loads when ITS buffer runs dry, so a
// wave waits for memory at nearly every id of its longest row -- a level of cnr-2000 x 30 lasted 215 us for rows of 12 ids, whatever the class bound.  Here a pass is: every lane
// loads its next four copied ids, the first four of its next kept block and its next four extras (three loads in flight at once, ONE wait), then four trips that emit min(copied
// head, extra head) each.  Rows without a table (CT_NONE, or an overflow that does not fit) take the old merges, lane by lane, before the wave's loop.
#ifndef COPY_W_MINWAVES // (tuning builds)
#define COPY_W_MINWAVES 1
#endif
template <int DEF, bool VEC>
__global__ void __launch_bounds__(TPB, COPY_W_MINWAVES) k_copy_list_w(GraphDev g, RangeView v, const int32_t *__restrict__ depth, const int32_t *__restrict__ list,
                                                     const int32_t *__restrict__ keyBase, int32_t level, int32_t midMin, int32_t bigMin, int *__restrict__ err,
                                                     const IvEntry *__restrict__ arena, int64_t arenaCap, const CopyTab *__restrict__ ctab) {
	const int32_t bucket = min(level, MAXLVL - 1);
	const int32_t lo = keyBase[bucket * NBIN], hi = keyBase[(bucket + 1) * NBIN];
	const int64_t rsNh = v.rowstart[v.nh];
	const int32_t coopMin = v.coopmin();
	for (int32_t idx = hi - 1 - (blockIdx.x * TPB + threadIdx.x); wave_any(idx >= lo); idx -= gridDim.x * TPB) {
		bool have = false;
		int32_t d = 0, dref = 0, limE = 0, limS = 0;
		int32_t *row = nullptr;
		const int32_t *src = nullptr, *ovfEnd = nullptr;
		int4 hd = int4{ 0, 0, 0, 0 };
		if (idx >= lo) do {
			const int32_t s = list[idx];
			const int32_t r = v.ref[s];
			if (r == 0 || (level >= MAXLVL - 1 && depth[s] != level)) break;
			const int32_t t = s - r;
			const i64x2_a8 rsp = *(const i64x2_a8 *)(v.rowstart + s), rtp = *(const i64x2_a8 *)(v.rowstart + t);
			const int64_t rs0 = rsp.x, rs1 = rsp.y, rt0 = rtp.x, rt1 = rtp.y;
			if (!(s >= v.nh ? (uint64_t)(rs1 - rsNh) <= v.succ_cap : (uint64_t)rs1 <= v.halo_cap) || !(t >= v.nh ? (uint64_t)(rt1 - rsNh) <= v.succ_cap : (uint64_t)rt1 <= v.halo_cap)) break; // E_CAP / E_HALO already raised by the parse kernel
			d = (int32_t)(rs1 - rs0); dref = (int32_t)(rt1 - rt0);
			if (copy_class_of(d, dref, midMin, bigMin) != 1) break;
			row = s < v.nh ? v.halo + rs0 : v.succ + (rs0 - rsNh);
			src = t < v.nh ? v.halo + rt0 : v.succ + (rt0 - rsNh);
			// ints that may be READ from the row's start on (a 16-byte load at the end of a row reads into its neighbours, never past the buffer)
			// (unsigned: the rows fit, so the differences are >= d / dref)
			limE = (int32_t)min<uint64_t>(s < v.nh ? v.halo_cap - (uint64_t)rs0 : v.succ_cap - (uint64_t)(rs0 - rsNh), 0x7fffffffull);
			limS = (int32_t)min<uint64_t>(t < v.nh ? v.halo_cap - (uint64_t)rt0 : v.succ_cap - (uint64_t)(rt0 - rsNh), 0x7fffffffull);
			if (d < coopMin) {
				hd = *(const int4 *)(ctab + s);
				const uint32_t kept = (uint32_t)hd.w & 0xffffu;
				if (kept != CT_NONE) {
					bool ok = true;
					if (kept > 3) { // the kept blocks from the fourth on: the end of the record's own part of the interval arena
						int64_t abase = 0; int32_t an = 0;
						if (g.minInt > 0) arena_slice(g.minInt, rs0, d, abase, an);
						ok = g.minInt > 0 && abase >= 0 && abase + an <= arenaCap && (int32_t)kept - 3 <= 4 * (an - 1);
						ovfEnd = (const int32_t *)(arena + abase + (an - 1));
					}
					if (ok) { have = true; break; }
				}
			}
			if (VEC) copy_node_v<DEF>(g, v.lo + s, d, dref, row, src, err);
			else copy_node<DEF>(g, v.lo + s, d, (int64_t)dref, row, src, err);
		} while (false);
		if (BV_TIMING(g, 0x20000)) continue; // (timing experiments only: the rows' metadata and nothing else)
		copy_rows_tab(have, d, dref, row, src, limE, limS, ovfEnd, hd);
	}
}

// One wave per row of fewer than COPY_BIG_MIN successors with a reference.  The block list is walked once (by
// every lane: it is short) into two LDS tables -- for the j-th copied block, the number of ids copied up to its
// end and the offset between an id's index in the referent's row and its index among the copied ids.  Then the
// copied ids and the row's extras are loaded into LDS with coalesced loads, and every id finds its final
// position by ONE binary search in the other set (the sets are disjoint in a valid file): copied id t goes to
// t + #(extras smaller), extra e to e + #(copied ids smaller).
constexpr int COPY_MID_WAVES = 4;
template <int DEF>
__global__ void __launch_bounds__(64 * COPY_MID_WAVES) k_copy_mid(GraphDev g, RangeView v, const int32_t *__restrict__ depth, const int32_t *__restrict__ queue,
                                                                  const int32_t *__restrict__ count, int32_t cap, int32_t level, int *__restrict__ err, const int4 *__restrict__ pre,
                                                                  const IvEntry *__restrict__ arena = nullptr, int64_t arenaCap = 0, const CopyTab *__restrict__ ctab = nullptr) { // ctab / arena: the tables the one-lane parse left (null: none)
	__shared__ int32_t s_vals[COPY_MID_WAVES][COPY_BIG_MIN], s_kend[COPY_MID_WAVES][COPY_BIG_MIN + 1], s_delta[COPY_MID_WAVES][COPY_BIG_MIN + 1];
	const int32_t coopMinTab = ctab ? v.coopmin() : 0;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int32_t *vals = s_vals[wave], *kend = s_kend[wave], *delta = s_delta[wave];
	auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); };
	// the queue holds the rows of this class of ALL levels: a wave takes the entries of this level among its share
	const int32_t nq = min(*count, cap);
	const int64_t rsNh = v.rowstart[v.nh];
	// The queue holds the rows of ALL levels, and a row's data sit behind three dependent loads (queue -> row -> referent).  A wave's entries are W apart (its
	// share is as even as with one entry per iteration); it looks at up to 64 of them at once, one per lane -- levels, descriptors, row starts and referents'
	// row starts in three round trips for all of them -- and then works through the ones of this level with everything at hand.
	const int32_t W = (int32_t)(gridDim.x * COPY_MID_WAVES), w = (int32_t)(blockIdx.x * COPY_MID_WAVES + wave);
	for (int64_t k0 = 0; w + k0 * W < nq; k0 += 64) { // (uniform)
		const int64_t qiL64 = w + (k0 + lane) * (int64_t)W;
		const bool inL = qiL64 < nq;
		const int32_t qiL = inL ? (int32_t)qiL64 : 0;
		const int32_t sL = inL ? queue[qiL] : 0;
		const int4 pwL = inL && pre ? pre[qiL] : int4{ -1, 0, 0, 0 };
		const int32_t depL = inL ? depth[sL] : -1, rL = inL ? (int32_t)v.ref[sL] : 0;
		const int64_t rs0L = inL ? v.rowstart[sL] : 0, rs1L = inL ? v.rowstart[sL + 1] : 0;
		const int32_t tL = sL - rL;
		const bool wantL = inL && depL == level && pwL.x != -2 && rL != 0;
		const int64_t rt0L = wantL ? v.rowstart[tL] : 0, rt1L = wantL ? v.rowstart[tL + 1] : 0;
		const bool fitL = wantL && (sL >= v.nh ? (uint64_t)(rs1L - rsNh) <= v.succ_cap : (uint64_t)rs1L <= v.halo_cap) && (tL >= v.nh ? (uint64_t)(rt1L - rsNh) <= v.succ_cap : (uint64_t)rt1L <= v.halo_cap); // (else: E_CAP / E_HALO already raised by the parse kernel)
	for (unsigned long long todo = __ballot(fitL); todo; todo &= todo - 1) {
		const int bsel = __builtin_ctzll(todo);
		const int32_t s = __shfl(sL, bsel, 64), t0 = __shfl(tL, bsel, 64);
		const int4 pw = int4{ __shfl(pwL.x, bsel, 64), __shfl(pwL.y, bsel, 64), __shfl(pwL.z, bsel, 64), __shfl(pwL.w, bsel, 64) };
		const int64_t rs0 = shfl_i64(rs0L, bsel), rs1 = shfl_i64(rs1L, bsel), rt0 = shfl_i64(rt0L, bsel), rt1 = shfl_i64(rt1L, bsel);
		const int32_t d = (int32_t)(rs1 - rs0);
		const int64_t dref = rt1 - rt0;
		int32_t *row = s < v.nh ? v.halo + rs0 : v.succ + (rs0 - rsNh);
		const int32_t *src = t0 < v.nh ? v.halo + rt0 : v.succ + (rt0 - rsNh);
		int64_t total = 0, copied = 0;
		int32_t nKept = 0;
		bool bad = false;
		// header + blocks (uniform)
		BitReader br;
		br.init(g.bits, g.nwords);
		uint64_t bc = 0;
		bool fromTab = false;
		if (pw.x < 0 && d < coopMinTab) {
			// Not walked by the pre-walk (a list of COPY_COOP_WALK_MIN codes and more), but decoded by the one-lane parse, which left the row's kept blocks as a table (CopyTab:
			// first index in the referent's row << 16 | length; three in the slot, the others at the end of the record's own part of the arena): the wave turns it into kend /
			// delta by a prefix sum -- a microsecond instead of the walk of up to a thousand codes by all 64 lanes side by side, 150 cycles each, which WAS this kernel's tail
			// (C2, level 1: its waves worked 4 us on average and the kernel lasted 330).  The table is read as data: every entry is checked against the two rows.
			const int4 hd = *(const int4 *)(ctab + s); // (uniform)
			const uint32_t keptT = (uint32_t)hd.w & 0xffffu;
			if (keptT != CT_NONE && keptT <= (uint32_t)COPY_BIG_MIN) {
				const int32_t *tabEnd = nullptr;
				bool ok = true;
				if (keptT > 3) {
					int64_t abase = 0; int32_t an = 0;
					if (g.minInt > 0) arena_slice(g.minInt, rs0, d, abase, an);
					ok = arena != nullptr && g.minInt > 0 && abase >= 0 && abase + an <= arenaCap && (int32_t)keptT - 3 <= 4 * (an - 1);
					tabEnd = (const int32_t *)(arena + abase + (an - 1));
				}
				if (ok) {
					int32_t carry = 0;
					bool okE = true;
					for (int32_t j0 = 0; j0 < (int32_t)keptT; j0 += 64) { // (uniform)
						const int32_t j = j0 + lane;
						uint32_t ent = 0;
						if (j < (int32_t)keptT) ent = (uint32_t)(j == 0 ? hd.z : j == 1 ? hd.y : j == 2 ? hd.x : tabEnd[2 - j]);
						const int32_t len = (int32_t)(ent & 0xffffu), start = (int32_t)(ent >> 16);
						int32_t inc = len;
#pragma unroll
						for (int o = 1; o < 64; o <<= 1) { const int32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
						if (j < (int32_t)keptT) {
							if (len == 0 || (int64_t)start + len > dref || (int64_t)carry + inc > d) okE = false;
							kend[j] = carry + inc; delta[j] = start - (carry + inc - len);
						}
						carry += __shfl(inc, 63, 64);
					}
					if (!__any(!okE) && carry == (int32_t)((uint32_t)hd.w >> 16)) { fromTab = true; nKept = (int32_t)keptT; copied = carry; }
				}
			}
		}
		if (fromTab) {}
		else if (pw.x >= 0) { // walked already: the tables are in GraphDev::walktab
			const int32_t kM = (pw.w >> 1) + 1;
			nKept = pw.y; copied = pw.z;
			for (int32_t k = lane; k < nKept; k += 64) { kend[k] = g.walktab[pw.x + k]; delta[k] = g.walktab[pw.x + kM + k]; }
		} else {
			br.seek((uint64_t)g.offsets[v.lo + s]);
			(void)Fields<DEF>::outdegree(br, g);
			(void)Fields<DEF>::reference(br, g);
			bc = Fields<DEF>::block_count(br, g);
			if (bc > (uint64_t)dref + 1) continue; // flagged by the parse kernel
		}
		for (uint64_t b = 0; !fromTab && pw.x < 0 && b <= bc; b++) {
			int64_t len;
			if (b < bc) len = (int64_t)Fields<DEF>::block(br, g) + (b ? 1 : 0);
			else len = dref - total; // implicit last block (copied when the block count is even)
			if (len < 0 || total + len > dref) { bad = true; break; }
			if (!(b & 1)) {
				if (copied + len > d || nKept > COPY_BIG_MIN) { bad = true; break; }
				if (lane == (nKept & 63)) { kend[nKept] = (int32_t)(copied + len); delta[nKept] = (int32_t)(total - copied); }
				nKept++;
				copied += len;
			}
			total += len;
		}
		if (bad || br.err || copied == 0) continue; // malformed (flagged by the parse kernel) or nothing to merge
		const int32_t nExtra = d - (int32_t)copied, nc = (int32_t)copied;
		wave_sync();
		// gather: copied ids -> vals[0 .. nc), extras -> vals[nc .. d)
		for (int32_t t = lane; t < nc; t += 64) {
			int32_t lo = 0, hi = nKept; // first kept block with kend > t
			while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (kend[mid] <= t) lo = mid + 1; else hi = mid; }
			vals[t] = src[t + delta[lo]];
		}
		for (int32_t e = lane; e < nExtra; e += 64) vals[nc + e] = row[nc + e];
		wave_sync();
		// final positions
		bool dup = false;
		for (int32_t t = lane; t < d; t += 64) {
			const int32_t val = vals[t];
			int32_t lo, hi;
			if (t < nc) { lo = nc; hi = d; } // copied id: count the extras below it
			else { lo = 0; hi = nc; }        // extra: count the copied ids below it
			while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (vals[mid] < val) lo = mid + 1; else hi = mid; }
			if (t >= nc && lo < nc && vals[lo] == val) dup = true;
			row[t - nc + lo] = val;
		}
		// an extra that equals a copied id (never in a valid file: the two land on one position and leave a hole): MergedIntIterator.java:69-72 emits equal heads once and
		// BVG:1210 pads with -1 -- one lane merges the two sets again, from LDS, the way copy_node does
		if (__any(dup)) {
			if (lane == 0) {
				int32_t i = 0, j = nc, k = 0;
				while (i < nc || j < d) {
					int32_t val;
					if (j >= d || (i < nc && vals[i] <= vals[j])) { val = vals[i]; if (j < d && vals[j] == val) j++; i++; }
					else val = vals[j++];
					row[k++] = val;
				}
				while (k < d) row[k++] = -1;
			}
		}
		wave_sync(); // the tables are reused by the next row
	}
	}
}

// The block lists of the rows of the group class (and, since they cost nothing more, their headers), walked BEFORE the copy pass, all
// levels at once, one wave per row, beside the parse kernels: a block list is a serial chain of codes that depends on nothing but the
// stream, and walked inside k_copy_big -- by one wave of a 1024-thread group while the fifteen others wait, two groups per CU -- it was
// 71 % of that kernel's time on the C5 shard (14 000 rows of 590 codes: 600 ticks per code walked by one lane, 105 per code by the
// wave; profiles/r4_experiments.txt).  Here sixteen waves per CU walk (9.4 KB of LDS each: WalkLds), each its own row.  The tables (kend, delta, as in k_copy_mid)
// go to the bump arena GraphDev::walktab, (bc >> 1) + 1 entries each; desc[qi] = (offset of the tables | -1 not walked: k_copy_big
// walks the list itself | -2 nothing to merge or malformed, number of copied blocks, copied ids, block count).
#ifndef PREWALK_COOP_MIN_
#define PREWALK_COOP_MIN_ 64
#endif
constexpr int PREWALK_WAVES = 4, PREWALK_LONG_MIN = 2048, PWL_NW = 4, PREWALK_COOP_MIN = PREWALK_COOP_MIN_; // (lists of PREWALK_LONG_MIN codes and more: k_copy_prewalk_long, below)
template <int DEF>
__global__ void __launch_bounds__(64 * PREWALK_WAVES) k_copy_prewalk(GraphDev g, RangeView v, const int32_t *__restrict__ queue, const int32_t *__restrict__ count, int32_t cap, int4 *__restrict__ desc, uint32_t longMin) {
	static_assert(DEF != 0, "default codings");
	// 9.4 KB of LDS per wave (WalkLds: the cooperative walk's window, exchange slots and cached codes; the header and the short lists go through the generic
	// reader, all lanes alike): sixteen walking waves per CU
	__shared__ __attribute__((aligned(16))) uint32_t cwin[PREWALK_WAVES][WalkLds::WORDS];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int32_t nq = min(*count, cap);
	// (as k_copy_mid: a wave looks at 64 of its strided queue entries at once, one per lane -- row, reference, outdegrees, record bounds in three round
	// trips for all of them -- and then walks them one by one)
	const int32_t W = (int32_t)(gridDim.x * PREWALK_WAVES), w = (int32_t)(blockIdx.x * PREWALK_WAVES + wave);
	for (int64_t k0 = 0; w + k0 * W < nq; k0 += 64) { // (uniform)
		const int64_t qiL64 = w + (k0 + lane) * (int64_t)W;
		const bool inL = qiL64 < nq;
		const int32_t sL = inL ? queue[inL ? (int32_t)qiL64 : 0] : 0;
		const int32_t rL = inL ? (int32_t)v.ref[sL] : 0, dL = inL ? v.outd[sL] : 0;
		const bool okL = inL && rL != 0; // (no RangeView::fits here: the walk touches no row -- k_copy_big / k_copy_mid check the rows themselves --, and a tile job runs this kernel beside the scan that writes rowstart: ADVICE r5)
		const int32_t drefL = okL ? v.outd[sL - rL] : 0;
		const int64_t off0L = okL ? g.offsets[v.lo + sL] : 0, off1L = okL ? g.offsets[v.lo + sL + 1] : 0;
	for (unsigned long long todo = __ballot(inL); todo; todo &= todo - 1) {
		const int bsel = __builtin_ctzll(todo);
		const int32_t qi = (int32_t)(w + (k0 + bsel) * (int64_t)W);
		int4 out = int4{ -1, 0, 0, 0 };
		bool leave = false;
		const int32_t d = __shfl(dL, bsel, 64);
		if (__shfl((int)okL, bsel, 64)) {
			const int64_t dref = __shfl(drefL, bsel, 64), off0 = shfl_i64(off0L, bsel), off1 = shfl_i64(off1L, bsel);
			BitReader br;
			br.init(g.bits, g.nwords);
			br.seek((uint64_t)off0);
			(void)Fields<DEF>::outdegree(br, g);
			(void)Fields<DEF>::reference(br, g);
			const uint64_t bc = Fields<DEF>::block_count(br, g);
			if (br.err || bc > (uint64_t)dref + 1) out.x = -2; // flagged by the parse kernel
			else if (d >= g.walkMin && bc >= (uint64_t)COPY_GROUP_WALK_MIN) {} // a giant record's long list: its tables fall out of its parse (coop_parse_node, bv_coop.hpp)
			else if (bc >= (uint64_t)longMin) leave = true; // k_copy_prewalk_long's
			else {
				const uint64_t kMax = (bc >> 1) + 1, need = 2 * kMax;
				int64_t off = -1;
				if (lane == 0 && need <= g.walkCap) { const uint32_t a = atomicAdd(g.walkCursor, (uint32_t)need); if ((uint64_t)a + need <= g.walkCap) off = a; }
				off = shfl_i64(off, 0);
				if (off >= 0) {
					int32_t *kend = g.walktab + off, *dlt = kend + kMax;
					int64_t total = 0, copied = 0;
					int32_t nKept = 0;
					int bad = 0;
					if (bc >= PREWALK_COOP_MIN) coop_block_walk<WalkLds>(g, br.pos(), (uint64_t)off1, (int64_t)bc, dref, d, kend, dlt, (int32_t)kMax, cwin[wave], total, copied, nKept, bad);
					else {
						for (uint64_t b = 0; b <= bc; b++) { // (every lane walks: the list is short)
							int64_t len;
							if (b < bc) len = (int64_t)Fields<DEF>::block(br, g) + (b ? 1 : 0);
							else len = dref - total; // implicit last block (copied when the block count is even)
							if (len < 0 || total + len > dref) { bad = 1; break; }
							if (!(b & 1)) {
								if (lane == (nKept & 63)) { kend[nKept] = (int32_t)min<int64_t>(copied + len, 0x7fffffff); dlt[nKept] = (int32_t)(total - copied); }
								nKept++;
								copied += len;
							}
							total += len;
						}
						bad |= br.err;
					}
					if (bad || copied > d || copied == 0) out.x = -2;
					else out = int4{ (int32_t)off, nKept, (int32_t)copied, (int32_t)min<uint64_t>(bc, 0x7fffffff) };
				}
			}
		}
		if (lane == 0 && !leave) desc[qi] = out;
	}
	}
}

// The lists of PREWALK_LONG_MIN codes and more are walked by a GROUP of four waves each (coop_block_walk_nw), in a kernel of their own that is
// launched first: one wave needs ~105 ticks per code, the longest list of the C5 shard has 12 500 -- 0.6 ms, which was how long k_copy_prewalk
// lasted, and what the copy pass waited for behind the parse kernels.  Every block reads the headers of its share of the queue (generic
// reader, all threads alike: the walk is a collective) and walks the long lists among them; k_copy_prewalk leaves those entries alone.
template <int DEF>
__global__ void __launch_bounds__(64 * PWL_NW) k_copy_prewalk_long(GraphDev g, RangeView v, const int32_t *__restrict__ queue, const int32_t *__restrict__ count, int32_t cap, int4 *__restrict__ desc) {
	static_assert(DEF != 0, "default codings");
	__shared__ __attribute__((aligned(16))) uint32_t win[CoopLds<PWL_NW>::WIN_WORDS];
	__shared__ int64_t xch[3 * PWL_NW + 8 + 1];
	const int32_t nq = min(*count, cap);
	for (int32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) { // (uniform)
		const int32_t s = queue[qi];
		const int32_t r = v.ref[s], d = v.outd[s];
		if (r == 0) continue; // (no RangeView::fits: see k_copy_prewalk)
		const int64_t dref = v.outd[s - r];
		if (dref + 1 < PREWALK_LONG_MIN) continue; // (bc <= dref + 1)
		BitReader br;
		br.init(g.bits, g.nwords);
		br.seek((uint64_t)g.offsets[v.lo + s]);
		(void)Fields<DEF>::outdegree(br, g);
		(void)Fields<DEF>::reference(br, g);
		const uint64_t bc = Fields<DEF>::block_count(br, g);
		if (br.err || bc > (uint64_t)dref + 1 || bc < (uint64_t)PREWALK_LONG_MIN) continue; // (short, or flagged: k_copy_prewalk's)
		if (d >= g.walkMin && bc >= (uint64_t)COPY_GROUP_WALK_MIN) continue;               // (a giant record's long list: its parse keeps the tables)
		const uint64_t kMax = (bc >> 1) + 1, need = 2 * kMax;
		__syncthreads(); // xch is free
		if (threadIdx.x == 0) {
			int64_t off = -1;
			if (need <= g.walkCap) { const uint32_t a = atomicAdd(g.walkCursor, (uint32_t)need); if ((uint64_t)a + need <= g.walkCap) off = a; }
			xch[3 * PWL_NW + 8] = off;
		}
		__syncthreads();
		const int64_t off = xch[3 * PWL_NW + 8];
		int4 out = int4{ -1, 0, 0, 0 };
		if (off >= 0) {
			int64_t total = 0, copied = 0;
			int32_t nKept = 0;
			int bad = 0;
			coop_block_walk_nw<PWL_NW>(g, br.pos(), (uint64_t)g.offsets[v.lo + s + 1], (int64_t)bc, dref, d, g.walktab + off, g.walktab + off + kMax, (int32_t)kMax, win, xch,
			                           CoopCfg<PWL_NW>::B_MAX, total, copied, nKept, bad);
			if (bad || copied > d || copied == 0) out.x = -2;
			else out = int4{ (int32_t)off, nKept, (int32_t)copied, (int32_t)min<uint64_t>(bc, 0x7fffffff) };
		}
		if (threadIdx.x == 0) desc[qi] = out;
	}
}

// The same for the rows of the wave class (k_copy_mid), one LANE per row: their lists are short (a row of 320 ids that keeps 95 % of
// its referent's has 32 codes), and k_copy_mid walked each with all 64 lanes side by side -- the serial part of the row, 600 ticks
// per code.  Lists of COPY_COOP_WALK_MIN codes and more are left to k_copy_mid (desc = -1).
template <int DEF>
__global__ void __launch_bounds__(LW_STRIDE) k_copy_prewalk_lanes(GraphDev g, RangeView v, const int32_t *__restrict__ queue, const int32_t *__restrict__ count, int32_t cap, int4 *__restrict__ desc) {
	static_assert(DEF != 0, "default codings");
	__shared__ uint32_t lwin[LW_MAIN * LW_STRIDE];
	const int lane = threadIdx.x & 63;
	const int32_t nq = min(*count, cap);
	for (int64_t q0 = (int64_t)blockIdx.x * LW_STRIDE; q0 < nq; q0 += (int64_t)gridDim.x * LW_STRIDE) {
		const int32_t qi = item_of((uint32_t)q0 + threadIdx.x);
		int4 out = int4{ -1, 0, 0, 0 };
		uint64_t bc = 0, need = 0;
		int64_t dref = 0;
		int32_t d = 0;
		int e = 0;
		bool mine = false;
		LaneWin<LW_MAIN> lw;
		lw.col = lwin + threadIdx.x;
		if (qi < nq) {
			const int32_t s = queue[qi];
			const int32_t r = v.ref[s];
			d = v.outd[s];
			if (r != 0) { // (no RangeView::fits: see k_copy_prewalk)
				dref = v.outd[s - r];
				lw.vlast = min((((uint64_t)g.offsets[v.lo + s + 1] >> 5) + 2) & ~(uint64_t)3, (g.nwords + 4) & ~(uint64_t)3);
				lw.seek(g, (uint64_t)g.offsets[v.lo + s]);
				(void)lw.template code<1>(g, e);
				(void)lw.template code<2>(g, e);
				bc = lw.template code<1>(g, e);
				if (e || bc > (uint64_t)dref + 1) out.x = -2; // flagged by the parse kernel
				else if (bc < COPY_COOP_WALK_MIN) { mine = true; need = 2 * ((bc >> 1) + 1); }
			}
		}
		// one bump of the arena per wave
		const int64_t incl = wave_incl_scan_i64((int64_t)need), tot = shfl_i64(incl, 63);
		int64_t base = -1;
		if (lane == 0 && tot > 0 && (uint64_t)tot <= g.walkCap) { const uint32_t a = atomicAdd(g.walkCursor, (uint32_t)tot); if ((uint64_t)a + (uint64_t)tot <= g.walkCap) base = a; }
		base = shfl_i64(base, 0);
		if (mine && base >= 0) {
			const int64_t off = base + incl - (int64_t)need, kMax = (int64_t)(bc >> 1) + 1;
			int32_t *kend = g.walktab + off, *dlt = kend + kMax;
			int64_t total = 0, copied = 0;
			int32_t nKept = 0;
			int bad = 0;
			for (uint64_t b = 0; b <= bc; b++) {
				int64_t len;
				if (b < bc) len = (int64_t)lw.template code<1>(g, e) + (b ? 1 : 0);
				else len = dref - total; // implicit last block (copied when the block count is even)
				if (len < 0 || total + len > dref) { bad = 1; break; }
				if (!(b & 1)) {
					kend[nKept] = (int32_t)min<int64_t>(copied + len, 0x7fffffff); dlt[nKept] = (int32_t)(total - copied);
					nKept++;
					copied += len;
				}
				total += len;
			}
			if (bad || e || copied > d || copied == 0 || nKept > COPY_BIG_MIN) out.x = -2;
			else out = int4{ (int32_t)off, nKept, (int32_t)copied, (int32_t)bc };
		}
		if (qi < nq) desc[qi] = out;
	}
}

#ifndef COPY_BIG_THREADS_
#define COPY_BIG_THREADS_ 1024
#define COPY_BIG_CAP_ 6144
#endif
#ifndef COPY_BIG_LEAN_
#define COPY_BIG_LEAN_ 1
#endif
// (512 threads and LDS tables for 2048 copied ids -- 38 KB of LDS, four groups per CU instead of one -- changed nothing on C2 and
// cnr-2000 x30 and cost 4 % on C5: a level of k_copy_big lasts as long as its longest row, not as long as its rows in sum.)
constexpr int COPY_BIG_THREADS = COPY_BIG_THREADS_, COPY_BIG_CAP = COPY_BIG_CAP_, COPY_BIG_ITEMS = 8, COPY_BIG_FIRST = 16384, COPY_BIG_TAKE = 8, COPY_BIG_GRID = 256 * (2048 / COPY_BIG_THREADS);
// One 1024-thread group per long row with a reference.  The block list is walked once, without memory traffic
// (it is the serial part of a row with thousands of blocks), into the same two LDS tables as in k_copy_mid; the
// copied ids (<= COPY_BIG_CAP of them) are then gathered into LDS and ranked among the row's extras
// row[copied..d) by binary search.  Then the extras move LEFT in place, chunk by chunk: extra e goes to
// e + #(copied ids smaller than it), which is never to the right of where it sits, and never onto an extra with
// a larger index -- so a chunk may be written once every extra up to its end has been read, and the NEXT chunk
// is already being read while this one is written.  The copied ids drop into the gaps at the end
// (MergedIntIterator semantics for the disjoint sets of a valid file).  Rows copying more than COPY_BIG_CAP ids
// fall back to one lane.
template <int DEF>
__global__ void __launch_bounds__(COPY_BIG_THREADS) k_copy_big(GraphDev g, RangeView v, const int32_t *__restrict__ depth, const int32_t *__restrict__ queue,
                                                               const int32_t *__restrict__ count, int32_t cap, int32_t level, int32_t *__restrict__ tmp, uint32_t tmpCap,
                                                               uint32_t *__restrict__ tmpCursor, int32_t *__restrict__ qhead, int32_t *__restrict__ nextPair, int *__restrict__ err, const int4 *__restrict__ pre) {
	if (blockIdx.x == 0 && threadIdx.x < 2) nextPair[threadIdx.x] = 0;
	__shared__ int32_t tabs[3 * COPY_BIG_CAP + 2];
	__shared__ int32_t s_ext[COPY_BIG_CAP]; // the row's extras, when they fit (registers hold the kernel to one group per CU: its LDS is free)
	int32_t *const cval = tabs, *const cpos = tabs + COPY_BIG_CAP, *const delta = tabs + 2 * COPY_BIG_CAP + 1;
	__shared__ int32_t s_b[2];
	// (since k_copy_prewalk the group walks a list itself only when the pre-walk could not -- its arena was full, it is off, a giant row whose parse kernel kept no
	// tables --: one lane through the generic reader then, and no 29 KB of LDS for the lane window and the cooperative walk's tile held by every group)
	constexpr bool OWN_WALK = DEF != 0 && !COPY_BIG_LEAN_;
	__shared__ uint32_t lwin[OWN_WALK ? LW_MAIN * LW_STRIDE : 1]; // stream window of the wave that walks the block list
	__shared__ __attribute__((aligned(16))) uint32_t cwin[OWN_WALK ? CoopLds<1>::WORDS : 4]; // tile of the cooperative walk of a long block list
	__shared__ int64_t s_copied, s_tmp, s_kmax, s_desc;
	__shared__ int32_t s_kept, s_bad;
	__shared__ int32_t s_dup, s_wsum[COPY_BIG_THREADS / 64]; // an extra that equals a copied id was seen in this row (never in a valid file): dedupe_row below
	// The queue holds the long rows of ALL levels; a group takes the next entry that is of this level and of this pass from a
	// shared head (rows differ by 200x in length: fixed shares left most groups idle while a few worked through several
	// giant rows).  Two passes, the rows of >= COPY_BIG_FIRST ids first, so that the longest merges start at once.
	// (entries are taken COPY_BIG_TAKE at a time: one atomic and two barriers per entry made the kernel 0.56 ms on C2, whose queue
	// holds 20 000 rows of all levels, most of them skipped)
	__shared__ int32_t s_qi;
	const int32_t nq = min(*count, cap);
	// A take is COPY_BIG_TAKE entries a stride apart, not neighbours: long rows come in runs of similar neighbours (C5), which must not
	// land in one group.
	const int32_t stride = (2 * nq + COPY_BIG_TAKE - 1) / COPY_BIG_TAKE;
	int32_t qbase = 0, qoff = COPY_BIG_TAKE;
	for (;;) {
		if (qoff >= COPY_BIG_TAKE) {
			__syncthreads(); // everybody has read s_qi
			if (threadIdx.x == 0) s_qi = atomicAdd(&qhead[0], 1);
			__syncthreads();
			qbase = s_qi; qoff = 0;
			if (qbase >= stride) break;
		}
		const int32_t qraw = qbase + stride * qoff++;
		if (qraw >= 2 * nq) continue;
		const int32_t qi = qraw >= nq ? qraw - nq : qraw;
		const int32_t s = queue[qi];
		if (depth[s] != level || (v.outd[s] >= COPY_BIG_FIRST) != (qraw < nq) || copy_class(v, depth, level, s, 0, 0x7fffffff) == 0) continue;
		const int32_t d = v.outd[s], r = v.ref[s];
		const int64_t dref = v.outd[s - r];
		int32_t *row = v.row(s);
		const int32_t *src = v.row(s - r);
		unsigned long long tk = (g.stats && (g.dbg & 16)) ? __builtin_readcyclecounter() : 0;
		const unsigned long long tkRow = tk;
#define CT(slot) do { if (g.stats && (g.dbg & 16)) { const unsigned long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g.stats[24 + slot], now_ - tk); tk = now_; } } while (0)
		// header + blocks, by the first wave only (sixteen waves walking the list side by side would only slow each
		// other down); default codings read through a lane window in LDS with the short-code decoders.  Fills the
		// tables (kend, delta) up to tabCap entries and publishes (copied, number of copied blocks, bad).
		auto walk_row = [&](int32_t *kend, int32_t *dlt, int32_t tabCap) {
			__syncthreads(); // the tables are free
			if (threadIdx.x < 64) {
				int64_t total = 0, copied = 0;
				int32_t nKept = 0;
				int bad = 0;
				unsigned long long wt0 = (g.stats && (g.dbg & 16)) ? __builtin_readcyclecounter() : 0;
				auto walk = [&](auto &&next_gamma, uint64_t bc) {
					if (bc > (uint64_t)dref + 1) { bad = 1; return; } // flagged by the parse kernel
					for (uint64_t b = 0; b <= bc; b++) {
						int64_t len;
						if (b < bc) len = (int64_t)next_gamma() + (b ? 1 : 0);
						else len = dref - total; // implicit last block (copied when the block count is even)
						if (len < 0 || total + len > dref) { bad = 1; break; }
						if (!(b & 1)) {
							if (nKept < tabCap && (int32_t)threadIdx.x == (nKept & 63)) { kend[nKept] = (int32_t)min<int64_t>(copied + len, 0x7fffffff); dlt[nKept] = (int32_t)(total - copied); }
							nKept++;
							copied += len;
						}
						total += len;
					}
				};
				if (OWN_WALK) {
					LaneWin<LW_MAIN> lw;
					lw.col = lwin + threadIdx.x;
					lw.vlast = min((((uint64_t)g.offsets[v.lo + s + 1] >> 5) + 2) & ~(uint64_t)3, (g.nwords + 4) & ~(uint64_t)3);
					lw.seek(g, (uint64_t)g.offsets[v.lo + s]);
					int e = 0;
					(void)lw.template code<1>(g, e);
					(void)lw.template code<2>(g, e);
					const uint64_t bc = lw.template code<1>(g, e);
					unsigned long long wt1 = (g.stats && (g.dbg & 16)) ? __builtin_readcyclecounter() : 0;
					if (bc >= COPY_COOP_WALK_MIN && bc <= (uint64_t)dref + 1 && !e)
						coop_block_walk(g, lw.pos(), (uint64_t)g.offsets[v.lo + s + 1], (int64_t)bc, dref, d, kend, dlt, tabCap, cwin, total, copied, nKept, bad);
					else walk([&] { return lw.template code<1>(g, e); }, bc);
					if (g.stats && (g.dbg & 16) && threadIdx.x == 0) { const unsigned long long wt2 = __builtin_readcyclecounter(); const int c = bc >= COPY_COOP_WALK_MIN ? 1 : 0;
						atomicAdd(&g.stats[40], wt1 - wt0); atomicAdd(&g.stats[41 + c], wt2 - wt1); atomicAdd(&g.stats[43 + c], 1ull); atomicAdd(&g.stats[45 + c], (unsigned long long)bc); }
					bad |= e;
				} else {
					BitReader br;
					br.init(g.bits, g.nwords);
					br.seek((uint64_t)g.offsets[v.lo + s]);
					(void)Fields<DEF>::outdegree(br, g);
					(void)Fields<DEF>::reference(br, g);
					const uint64_t bc = Fields<DEF>::block_count(br, g);
					walk([&] { return Fields<DEF>::block(br, g); }, bc);
					bad |= br.err;
				}
				if (threadIdx.x == 0) { s_copied = copied; s_kept = nKept; s_bad = bad; }
			}
			__syncthreads();
		};
		// An extra that equals a copied id (no writer produces one): MergedIntIterator.java:69-72 emits equal heads once and the row ends in -1 (BVG:1210).  The merges below
		// are STABLE -- a copied id goes to t + #(extras smaller), an extra to e + #(copied ids smaller OR EQUAL): the pair lands side by side, nothing is left unwritten -- and
		// raise s_dup where they see such a pair; the group then closes the gaps in place, a chunk of the row at a time (an id moves left only), and pads the row.
		auto dedupe_row = [&]() {
			__syncthreads(); // the row is complete, s_dup is final
			if (!s_dup) return; // (uniform)
			int32_t carry = 0;
			for (int32_t base = 0; base < d; base += COPY_BIG_THREADS) {
				const int32_t i = base + (int32_t)threadIdx.x;
				const int32_t x = i < d ? row[i] : 0;
				const bool gone = i > 0 && i < d && x == row[i - 1]; // (row[base - 1] still holds what the merge put there: ids of earlier chunks moved LEFT of it, or not at all)
				const unsigned long long m = __ballot(gone);
				__syncthreads(); // every id of this chunk has been read; the last chunk's sums too
				if ((threadIdx.x & 63) == 0) s_wsum[threadIdx.x >> 6] = __popcll(m);
				__syncthreads();
				int32_t before = 0, total = 0;
				for (int k = 0; k < COPY_BIG_THREADS / 64; k++) { const int32_t c = s_wsum[k]; if (k < (int)(threadIdx.x >> 6)) before += c; total += c; }
				const int32_t shift = carry + before + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
				if (i < d && !gone && shift > 0) row[i - shift] = x;
				carry += total;
			}
			__syncthreads();
			for (int32_t i = d - carry + (int32_t)threadIdx.x; i < d; i += COPY_BIG_THREADS) row[i] = -1;
		};
		// The merge proper, on tables that live in LDS (the usual case) or in global scratch (rows copying more ids
		// than the LDS tables hold): gather the copied ids, rank them among the extras (still at row[nc..d)), move the
		// extras left chunk by chunk, drop the copied ids into the gaps.  kpos doubles as kend during the gather.
		auto merge_row = [&](const int32_t *kend, const int32_t *dlt, int32_t *cv_, int32_t *cp_, int32_t nc, int32_t nKept) {
			const int32_t nExtra = d - nc;
			if (threadIdx.x == 0) s_dup = 0; // (a barrier follows before anybody raises it)
			for (int32_t t = threadIdx.x; t < nc; t += COPY_BIG_THREADS) {
				int32_t lo = 0, hi = nKept; // first copied block with kend > t (a block of length 0 is possible only in first position)
				while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (kend[mid] <= t) lo = mid + 1; else hi = mid; }
				cv_[t] = src[t + dlt[lo]];
			}
			if (nExtra <= COPY_BIG_CAP) {
				// both sets in LDS: every id finds its place by one search in the other set and goes straight to the row -- no search in global memory (the ranks
				// of the copied ids among the extras: a quarter of this kernel), no chunk-wise move of the extras around their own unread tail
				for (int32_t e = threadIdx.x; e < nExtra; e += COPY_BIG_THREADS) s_ext[e] = row[nc + e];
				__syncthreads();
				CT(1);
				for (int32_t t = threadIdx.x; t < nc; t += COPY_BIG_THREADS) {
					const int32_t cv = cv_[t];
					int32_t lo = 0, hi = nExtra;
					while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (s_ext[mid] < cv) lo = mid + 1; else hi = mid; }
					row[t + lo] = cv;
				}
				CT(2);
				for (int32_t e = threadIdx.x; e < nExtra; e += COPY_BIG_THREADS) {
					const int32_t ev = s_ext[e];
					int32_t lo = 0, hi = nc; // copied ids not larger than this extra
					while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (cv_[mid] <= ev) lo = mid + 1; else hi = mid; }
					if (lo > 0 && cv_[lo - 1] == ev) s_dup = 1;
					if (lo < nc) row[e + lo] = ev; // (the extras behind the last copied id stay where they are)
				}
				CT(3);
				dedupe_row();
				return;
			}
			__syncthreads();
			CT(1);
			for (int32_t t = threadIdx.x; t < nc; t += COPY_BIG_THREADS) {
				const int32_t cv = cv_[t];
				int32_t lo = 0, hi = nExtra;
				while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (row[nc + mid] < cv) lo = mid + 1; else hi = mid; }
				cp_[t] = t + lo;
			}
			__syncthreads();
			CT(2);
			// extras up to the one following the last copied id move; the rest stay where they are
			const int32_t nMove = cp_[nc - 1] - (nc - 1);
			if (threadIdx.x == 0 && nMove < nExtra && row[nc + nMove] == cv_[nc - 1]) s_dup = 1; // (the first extra that stays equals the last copied id)
			constexpr int32_t CHUNK = COPY_BIG_THREADS * COPY_BIG_ITEMS;
			int32_t cur[COPY_BIG_ITEMS], nxt[COPY_BIG_ITEMS];
#pragma unroll
			for (int u = 0; u < COPY_BIG_ITEMS; u++) { const int32_t e = u * COPY_BIG_THREADS + (int32_t)threadIdx.x; nxt[u] = e < nMove ? row[nc + e] : 0; }
			for (int32_t e0 = 0; e0 < nMove; e0 += CHUNK) {
#pragma unroll
				for (int u = 0; u < COPY_BIG_ITEMS; u++) cur[u] = nxt[u];
				__syncthreads(); // every extra up to the end of this chunk has been read
#pragma unroll
				for (int u = 0; u < COPY_BIG_ITEMS; u++) { const int32_t e = e0 + CHUNK + u * COPY_BIG_THREADS + (int32_t)threadIdx.x; nxt[u] = e < nMove ? row[nc + e] : 0; }
#pragma unroll
				for (int u = 0; u < COPY_BIG_ITEMS; u++) {
					const int32_t e = e0 + u * COPY_BIG_THREADS + (int32_t)threadIdx.x;
					if (e < nMove) {
						int32_t lo = 0, hi = nc; // copied ids not larger than this extra
						while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (cv_[mid] <= cur[u]) lo = mid + 1; else hi = mid; }
						if (lo > 0 && cv_[lo - 1] == cur[u]) s_dup = 1;
						if (lo < nc) row[e + lo] = cur[u];
					}
				}
			}
			__syncthreads(); // all extras are in place
			CT(3);
			for (int32_t t = threadIdx.x; t < nc; t += COPY_BIG_THREADS) row[cp_[t]] = cv_[t];
			CT(4);
			dedupe_row();
		};
		// The same merge for a row whose tables (kend, dlt) and copied ids (cv_) live in global scratch.  Nothing here searches
		// global memory element by element (a row of 300 000 ids spent 4 ms in such searches): the copied ids are gathered
		// tile by tile with the tile's slice of the block tables in LDS; then the OUTPUT is cut into tiles of STREAM_TS ids --
		// one diagonal search per tile boundary, all boundaries at once, says how many copied ids precede it (merge path) --
		// and each tile's copied ids and extras are ranked against each other in LDS.  In place: tile k overwrites
		// row[p0..p1), which held extras with index < p1 - nc <= j1, all read by then; tile k + 1 is on its way meanwhile.
		auto merge_row_stream = [&](const int32_t *kend, const int32_t *dlt, int32_t *cv_, int32_t nc, int32_t nKept) {
			constexpr int32_t TS = COPY_BIG_THREADS * 8, GT = TS, ITEMS = TS / COPY_BIG_THREADS; // (a round of the gather is a chain of latencies -- table, search, id, store, barriers --: as many ids per round as the LDS tables allow)
			static_assert(2 * (GT + 1) <= 3 * COPY_BIG_CAP + 2 && TS + COPY_BIG_THREADS + 1 <= 3 * COPY_BIG_CAP + 2, "the tiles of the streaming merge live in the LDS tables");
			const int32_t nExtra = d - nc;
			if (threadIdx.x == 0) s_dup = 0; // (barriers follow before anybody raises it)
			int32_t *bufK = tabs, *bufD = tabs + GT + 1;
			// b0 = the block that holds id t0.  Blocks after the first are non-empty, so the blocks of the GT ids of a tile are among
			// the GT + 1 table entries from b0 on: they are loaded as they lie, and the next tile's b0 is found in LDS.
			int32_t b0 = (nKept > 1 && kend[0] == 0) ? 1 : 0;
			for (int32_t t0 = 0; t0 < nc; t0 += GT) {
				const int32_t t1 = min(nc, t0 + GT), nbMax = min(nKept - b0, GT + 1);
				__syncthreads(); // the buffers are free
				// (the tile's blocks, a thousand table entries at a time until one ends at t1 or later: a tile of 4 096 ids of the C5 shard has ~340 blocks,
				// and loading all GT + 1 entries it COULD have was twice the loads of the ids themselves)
				int32_t nb = 0;
				for (;;) {
					const int32_t upto = min(nbMax, nb + COPY_BIG_THREADS);
					for (int32_t k = nb + (int32_t)threadIdx.x; k < upto; k += COPY_BIG_THREADS) { bufK[k] = kend[b0 + k]; bufD[k] = dlt[b0 + k]; }
					__syncthreads();
					nb = upto;
					if (nb >= nbMax || bufK[nb - 1] >= t1) break; // (uniform)
				}
				{
					// the thread's ids of the tile searched side by side, branch-free (one search after the other is a chain of a dozen dependent LDS reads each)
					constexpr int GI = GT / COPY_BIG_THREADS;
					int32_t lo[GI];
#pragma unroll
					for (int u = 0; u < GI; u++) lo[u] = 0;
					const int32_t m = nb - 1; // the answer lies in [0, m]: the number of entries among the first m with kend <= t
					for (int32_t step = m > 0 ? 1 << (31 - __clz(m)) : 0; step > 0; step >>= 1) {
#pragma unroll
						for (int u = 0; u < GI; u++) {
							const int32_t t = t0 + u * COPY_BIG_THREADS + (int32_t)threadIdx.x, nx = lo[u] + step;
							if (nx <= m && bufK[nx - 1] <= t) lo[u] = nx;
						}
					}
#pragma unroll
					for (int u = 0; u < GI; u++) {
						const int32_t t = t0 + u * COPY_BIG_THREADS + (int32_t)threadIdx.x;
						if (t >= t1) continue;
						const int64_t si = (int64_t)t + bufD[lo[u]]; // (inside the referent's row for tables of a valid walk; tables taken over from the parse
						cv_[t] = si >= 0 && si < dref ? src[si] : 0; //  kernel of a record it flagged could hold anything: never read outside the row)
						if (t == t1 - 1) s_b[0] = b0 + lo[u] + (bufK[lo[u]] <= t1 ? 1 : 0); // the block of id t1 (bufK[lo] > t1 - 1: it ends at t1 or later)
					}
				}
				__syncthreads();
				b0 = min(s_b[0], nKept - 1);
			}
			CT(5);
			int32_t *buf = tabs, *splits = tabs + TS;
			int32_t iBase = 0;
			for (int32_t base = 0; base < d; base += COPY_BIG_THREADS * TS) {
				const int32_t ntl = (int32_t)min<int64_t>(COPY_BIG_THREADS, ((int64_t)d - base + TS - 1) / TS);
				__syncthreads(); // cv_ is complete (first round) / the last tile of the previous round is out
				for (int32_t k = threadIdx.x; k <= ntl; k += COPY_BIG_THREADS) {
					const int32_t p = (int32_t)min<int64_t>(d, (int64_t)base + (int64_t)k * TS);
					// copied ids among the first p of the merge; extras below index base - iBase are gone: i(p) <= iBase + (p - base)
					int32_t lo = max(max(0, p - nExtra), iBase), hi = min(min(p, nc), iBase + (p - base));
					while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (cv_[mid] <= row[nc + (p - 1 - mid)]) lo = mid + 1; else hi = mid; } // (<=: of equal heads the copied id comes first)
					if (lo > 0 && p - lo < nExtra && cv_[lo - 1] == row[nc + (p - lo)]) s_dup = 1; // an equal pair astride this cut: no tile sees both
					splits[k] = lo;
				}
				__syncthreads();
				CT(6);
				int32_t x[ITEMS];
				auto fetch = [&](int32_t k) {
					const int32_t p0 = base + k * TS, p1 = (int32_t)min<int64_t>(d, (int64_t)p0 + TS);
					const int32_t i0 = splits[k], ni = splits[k + 1] - i0, j0 = p0 - i0, tot = p1 - p0;
#pragma unroll
					for (int u = 0; u < ITEMS; u++) { const int32_t t = u * COPY_BIG_THREADS + (int32_t)threadIdx.x; x[u] = t < ni ? cv_[i0 + t] : t < tot ? row[nc + j0 + (t - ni)] : 0; }
				};
				if (splits[0] < nc) fetch(0);
				for (int32_t k = 0; k < ntl; k++) {
					const int32_t p0 = base + k * TS, p1 = (int32_t)min<int64_t>(d, (int64_t)p0 + TS);
					const int32_t i0 = splits[k], ni = splits[k + 1] - i0, tot = p1 - p0;
					if (i0 >= nc) break; // every copied id is placed: the remaining extras are where they belong
					__syncthreads(); // the previous tile's searches are done
#pragma unroll
					for (int u = 0; u < ITEMS; u++) buf[u * COPY_BIG_THREADS + threadIdx.x] = x[u];
					__syncthreads();
					int32_t y[ITEMS];
#pragma unroll
					for (int u = 0; u < ITEMS; u++) y[u] = x[u];
					if (k + 1 < ntl && splits[k + 1] < nc) fetch(k + 1); // reads only what no tile before it overwrites
#pragma unroll
					for (int u = 0; u < ITEMS; u++) {
						const int32_t t = u * COPY_BIG_THREADS + (int32_t)threadIdx.x;
						if (t >= tot) continue;
						int32_t lo, hi, self;
						if (t < ni) { lo = ni; hi = tot; self = t - ni; while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (buf[mid] < y[u]) lo = mid + 1; else hi = mid; } } // copied id: the extras smaller
						else { // extra: the copied ids not larger
							lo = 0; hi = ni; self = t - ni;
							while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (buf[mid] <= y[u]) lo = mid + 1; else hi = mid; }
							if (lo > 0 && buf[lo - 1] == y[u]) s_dup = 1;
						}
						row[p0 + self + lo] = y[u];
					}
				}
				iBase = splits[ntl];
				CT(7);
			}
			dedupe_row();
		};
		// Where the tables live is decided BEFORE the walk (a long block list is the serial part of the row: it is walked once):
		// at most bc / 2 + 1 blocks are copied and at most min(dref, d) ids, so a referent of up to COPY_BIG_CAP ids with a
		// block list of up to 2 * COPY_BIG_CAP codes fits the LDS tables for sure; everything else gets tables in global
		// scratch (bump allocator), sized by those bounds.
		const int4 pw = pre ? pre[qi] : int4{ -1, 0, 0, 0 }; // (uniform)
		if (pw.x == -2) continue; // k_copy_prewalk: nothing to merge, or flagged by the parse kernel
		if (pw.x >= 0) {
			// the list was walked by k_copy_prewalk: tables in GraphDev::walktab (bounds as below; values of our own kernel)
			const int64_t kM = ((int64_t)pw.w >> 1) + 1, cMax = dref < (int64_t)d ? dref : (int64_t)d;
			const bool inLds = !(dref > COPY_BIG_CAP || kM > COPY_BIG_CAP + 1);
			__syncthreads(); // the tables and s_tmp are free
			if (inLds) {
				for (int32_t k = threadIdx.x; k < pw.y; k += COPY_BIG_THREADS) { cpos[k] = g.walktab[pw.x + k]; delta[k] = g.walktab[pw.x + kM + k]; }
			} else if (threadIdx.x == 0) {
				s_tmp = -1;
				if (tmp && (uint64_t)cMax <= tmpCap) { const uint32_t o = atomicAdd(tmpCursor, (uint32_t)cMax); if ((uint64_t)o + (uint64_t)cMax <= tmpCap) s_tmp = o; } // room for the copied ids
			}
			__syncthreads();
			if (g.stats && threadIdx.x == 0) { stat_add(g, 8, 1); stat_add(g, 9, (unsigned long long)pw.y); stat_max(g, 15, (unsigned long long)pw.y); stat_add(g, 4, (unsigned long long)d); }
			CT(0);
			if (inLds) merge_row(cpos, delta, cval, cpos, pw.z, pw.y);
			else {
				const int64_t where = s_tmp;
				__syncthreads(); // everybody has read s_tmp
				if (where < 0) { if (threadIdx.x == 0) copy_node<DEF>(g, v.lo + s, d, dref, row, src, err); }
				else merge_row_stream(g.walktab + pw.x, g.walktab + pw.x + kM, tmp + where, pw.z, pw.y);
			}
			continue;
		}
		if (threadIdx.x == 0) {
			BitReader hb;
			hb.init(g.bits, g.nwords);
			hb.seek((uint64_t)g.offsets[v.lo + s]);
			(void)Fields<DEF>::outdegree(hb, g);
			(void)Fields<DEF>::reference(hb, g);
			const uint64_t bc = Fields<DEF>::block_count(hb, g);
			s_tmp = -2; // LDS tables
			s_desc = -1;
			if (bc <= (uint64_t)dref + 1 && g.walktab && d >= g.walkMin && row[0] == -2) {
				// the parse kernel walked this list with eight waves and kept the tables (GraphDev::walktab): nothing to walk here
				const int64_t off = row[1], nk = row[2], cp = row[3], kM = (int64_t)(bc >> 1) + 1, cMax = dref < (int64_t)d ? dref : (int64_t)d;
				if (off >= 0 && (uint64_t)off + 2 * (uint64_t)kM <= g.walkCap && nk >= 1 && nk <= kM && cp >= 4 && cp <= cMax && tmp && (uint64_t)cMax <= tmpCap) {
					const uint32_t o = atomicAdd(tmpCursor, (uint32_t)cMax); // room for the copied ids
					if ((uint64_t)o + (uint64_t)cMax <= tmpCap) { s_desc = off; s_tmp = o; s_kmax = kM; s_copied = cp; s_kept = (int32_t)nk; s_bad = 0; }
				}
			}
			if (s_desc >= 0) {}
			else if (bc > (uint64_t)dref + 1) s_tmp = -3; // flagged by the parse kernel
			else if (dref > COPY_BIG_CAP || (bc >> 1) + 1 > (uint64_t)COPY_BIG_CAP + 1) {
				const uint64_t kMax = (bc >> 1) + 1, cMax = (uint64_t)(dref < (int64_t)d ? dref : (int64_t)d), need = 2 * kMax + 2 * cMax;
				s_tmp = -1; // one lane does the row
				if (tmp && need <= tmpCap) { const uint32_t o = atomicAdd(tmpCursor, (uint32_t)need); if ((uint64_t)o + need <= tmpCap) s_tmp = o; }
				s_kmax = (int64_t)kMax;
			}
		}
		__syncthreads();
		const int64_t where = s_tmp, kMax = s_kmax, desc = s_desc;
		__syncthreads();
		if (g.stats && (g.dbg & 16) && threadIdx.x == 0) atomicAdd(&g.stats[47], __builtin_readcyclecounter() - tkRow);
		if (desc >= 0) { // (copied >= 4, so there is something to merge; bounds checked above)
			CT(0);
			merge_row_stream(g.walktab + desc, g.walktab + desc + kMax, tmp + where, (int32_t)s_copied, s_kept);
			continue;
		}
		if (where == -3) continue;
		if (where == -1) {
			if (threadIdx.x == 0) copy_node<DEF>(g, v.lo + s, d, dref, row, src, err);
			continue;
		}
		int32_t *tabK = where == -2 ? cpos : tmp + where, *tabD = where == -2 ? delta : tabK + kMax;
		walk_row(tabK, tabD, where == -2 ? COPY_BIG_CAP + 1 : (int32_t)min<int64_t>(kMax, 0x7fffffff));
		const int64_t copied = s_copied;
		const int32_t nKept = s_kept;
		if (s_bad || copied > d || copied == 0) continue; // malformed (flagged by the parse kernel) / nothing to merge: the extras already fill the row
		if (g.stats && threadIdx.x == 0) { stat_add(g, 8, 1); stat_add(g, 9, (unsigned long long)nKept); stat_max(g, 15, (unsigned long long)nKept); stat_add(g, 4, (unsigned long long)d); }
		const unsigned long long tRow0 = tk;
		CT(0);
		const unsigned long long tWalk = tk - tRow0;
		if (where == -2) { merge_row(cpos, delta, cval, cpos, (int32_t)copied, nKept); continue; } // (copied <= dref <= COPY_BIG_CAP, nKept <= COPY_BIG_CAP + 1)
		const int64_t cMaxRow = dref < (int64_t)d ? dref : (int64_t)d;
		if (nKept > kMax || copied > cMaxRow) continue; // (cannot happen: the bounds above)
		merge_row_stream(tabK, tabD, tabD + kMax, (int32_t)copied, nKept);
#undef CT
	}
}

// parse pass over a list sorted by work bin only (all chain levels together): 64 records of similar length per wave.
// The list is sorted longest first.  Thread T takes entries T, 2G-1-T, 2G+T, 4G-1-T, ... (G = threads in the grid): a snake, so that the
// threads that got the longest records of one sweep get the shortest of the next.  The loads that lead to a record are issued ahead of it:
// a sweep of short records is three dependent round trips (list entry -> outdegree / reference / row start / offsets -> the referent's
// outdegree and the stream words) in front of ~5 us of decoding; the entry is fetched two sweeps ahead and what hangs on it one sweep
// ahead, so a sweep waits for the last trip only.  Default codings: parse_node_lwb (bv_lanewin.hpp); others: the generic reader.
template <int DEF, bool HASH = false, bool LWC = true> // LWC: the round-6 loop (parse_node_lwc: leaves the copy blocks as tables); false: round 4's (parse_node_lwb; knob lane_loop = 0)
#ifndef PARSE_LIST_MINWAVES // (tuning builds: blocks per CU the compiler is asked to leave registers for)
#define PARSE_LIST_MINWAVES 1
#endif
__global__ void __launch_bounds__(TPB, PARSE_LIST_MINWAVES) k_parse_list(GraphDev g, RangeView v, const int32_t *__restrict__ list, const int32_t *__restrict__ keyBase, int32_t binLo, int32_t binHi,
                                                    IvEntry *__restrict__ arena, int64_t arenaCap, int *__restrict__ err, CopyTab *__restrict__ ctab = nullptr, int packed = 0) {
	static_assert(!HASH || DEF != 0, "the hash fold rides on the default codings' loop");
	__shared__ uint32_t lw[DEF ? (LW_MAIN + 2 * LW_RING) * LW_STRIDE : 1]; // per lane: a window of the stream and a ring of intervals (default codings)
	const int32_t lo = keyBase[binLo], hi = keyBase[binHi], coopMin = v.coopmin();
	// HASH: the rows without a reference are added to the job's hash as they are decoded and written only where a row of the view copies from them
	const HashCtx hx = HASH ? *v.hx : HashCtx{};
	uint32_t hacc = 0;
	const int64_t G = (int64_t)gridDim.x * TPB, T = (int64_t)blockIdx.x * TPB + threadIdx.x, N = (int64_t)hi - lo;
	const int64_t rs0 = v.rowstart[v.nh];
	auto entry = [&](int64_t sweep) -> int32_t {
		const int64_t off = sweep * G + ((sweep & 1) ? G - 1 - T : T);
		return sweep * G < N && off < N ? list[hi - 1 - off] : -1;
	};
	int32_t sCur = entry(0), sNext = entry(1);
	int32_t d = 0, r = 0; int64_t ra = 0, rb = 0; uint64_t oa = 0, ob = 0;
	// (rowstart[s], rowstart[s + 1] and offsets[x], offsets[x + 1] by ONE 16-byte load each: every load of this kernel goes to a line of its own and costs its CU as much
	// whatever it carries -- scripts/ubench_lines.hip)
	// The outdegree is the difference of the row starts, and the reference rides in the list entry (packed: bits 28 .. 31 of an entry hold min(ref, 15), LIST_REF_ESC = look it
	// up; a job of 2^28 slots and more has plain entries): two lines less per record -- the counters put this kernel at the rate at which its CU takes scattered lines
	// (172 M line accesses per scan of C2 for 10 M records, 12 per record of 20 ids: 5 stores and 7 of these loads).
	auto fetch_meta = [&](int32_t ec) { // ec: the list entry (!= -1)
		const int32_t sc = packed ? (ec & LIST_SLOT_MASK) : ec, code = (int32_t)((uint32_t)ec >> 28);
		r = packed && code != LIST_REF_ESC ? code : (int32_t)v.ref[sc];
		const i64x2_a8 rr = *(const i64x2_a8 *)(v.rowstart + sc), oo = *(const i64x2_a8 *)(g.offsets + (v.lo + sc));
		ra = rr.x; rb = rr.y; oa = (uint64_t)oo.x; ob = (uint64_t)oo.y;
		d = (int32_t)(rb - ra);
	};
	if (sCur != -1) fetch_meta(sCur);
	for (int64_t sweep = 0; sweep * G < N; sweep++) {
		const int32_t s = sCur == -1 ? -1 : packed ? (sCur & LIST_SLOT_MASK) : sCur, dC = d, rC = r; const int64_t raC = ra, rbC = rb; const uint64_t oaC = oa, obC = ob;
		const int32_t drefC = s >= 0 && rC > 0 ? v.outd[s - rC] : 0;
		sCur = sNext; sNext = entry(sweep + 2);
		d = 0;
		if (sCur != -1) fetch_meta(sCur);
		if (s < 0 || dC >= coopMin || dC == 0) continue; // decoded by whole waves (k_parse_big) / nothing to decode
		const bool fits = s >= v.nh ? (uint64_t)(rbC - rs0) <= v.succ_cap : (uint64_t)rbC <= v.halo_cap; // (RangeView::fits)
		if (!fits) { atomicOr(err, s >= v.nh ? E_CAP : E_HALO); continue; }
		int32_t *const row = s < v.nh ? v.halo + raC : v.succ + (raC - rs0); // (RangeView::row)
		if (DEF) {
			// the record's slice of the interval arena (the same slices as the cooperative kernels': floor(rowstart / minInt), d / minInt + 1 entries)
			int64_t abase = 0;
			int32_t aslice = 0;
			if (g.minInt > 0) arena_slice(g.minInt, raC, dC, abase, aslice);
			if (g.minInt > 0 && (abase < 0 || abase + aslice > arenaCap)) { atomicOr(err, E_FORMAT); continue; }
			if (LWC) {
				CopyTab *const ct = ctab ? ctab + s : nullptr; // the slot's table of copy blocks, for the copy pass
				const int32_t own = g.minInt > 0 ? aslice - 1 : 0;
				uint32_t *const ring = lw + LW_MAIN * LW_STRIDE + threadIdx.x; // the lane's ring of intervals, behind the stream windows
				auto open_window = [&](LaneWin<LW_MAIN> &br, uint64_t off0, uint64_t off1, int32_t d_, int32_t r_) { // the lane's window of the stream, from the record's block count on
					br.col = lw + threadIdx.x;
					br.vlast = min(((off1 >> 5) + 2) & ~(uint64_t)3, (g.nwords + 4) & ~(uint64_t)3);
					const uint64_t at = record_body(g, off0, d_, r_);
					br.seek_short(g, at, off1 > at ? off1 - at : 0);
				};
				if (HASH) {
					const bool mine = s >= v.nh && rC == 0;
					const uint32_t hw = mine ? hash_upow(hx.ptab, (uint64_t)(1 + rbC + (int64_t)s)) : 0u;
					const bool keep = !mine || hx.mark[s] != 0;
					LaneWin<LW_MAIN> br;
					open_window(br, oaC, obC, dC, rC);
					parse_node_lwc<DEF == 1 ? 3 : 0, true>(g, br, ring, v.lo + s, dC, rC, drefC, row, (int2 *)(arena + abase), own, ct, err, &hacc, hw, keep);
				}
				else {
					LaneWin<LW_MAIN> br;
					open_window(br, oaC, obC, dC, rC);
					parse_node_lwc<DEF == 1 ? 3 : 0>(g, br, ring, v.lo + s, dC, rC, drefC, row, (int2 *)(arena + abase), own, ct, err);
				}
			}
			else if (HASH) {
				const bool mine = s >= v.nh && rC == 0; // hashed here; the others (rows with a reference, halo rows) are written as ever
				const uint32_t hw = mine ? hash_upow(hx.ptab, (uint64_t)(1 + rbC + (int64_t)s)) : 0u; // successor j of slot s weighs u^(1 + rowstart[s + 1] + s) * 31^j
				const bool keep = !mine || hx.mark[s] != 0;
				parse_node_lwb<DEF == 1 ? 3 : 0, true>(g, v.lo + s, dC, rC > 0, (int64_t)drefC, row, lw, (int2 *)(arena + abase), err, oaC, obC, &hacc, hw, keep);
			}
			else parse_node_lwb<DEF == 1 ? 3 : 0>(g, v.lo + s, dC, rC > 0, (int64_t)drefC, row, lw, (int2 *)(arena + abase), err, oaC, obC);
		}
		else parse_node<DEF>(g, v.lo + s, dC, rC > 0, (int64_t)drefC, row, err);
	}
	if (HASH) {
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) hacc += (uint32_t)__shfl_xor((int)hacc, o, 64);
		if ((threadIdx.x & 63) == 0) hash_add(hx, hacc);
	}
}

// bvg_scan_checksum, what the kernels that decode do not add themselves.  The one-lane parse adds the rows without a reference that it decodes (HASH above), the
// lane class of the copy pass the rows it merges (copy_node<., true>); here, from memory: the node numbers, the rows without a reference that the wave / group
// classes decoded (or every such row when the parse did not hash: codings other than the default set) -- `what` bit 0, launched beside the copy pass --, and the rows
// with a reference that the wave / group classes of the copy pass merged (or all of them when the lane class did not hash) -- bit 1, behind the copy pass.
// A block takes 256 consecutive nodes: a lane adds its node's number and, if its row is short, the row; longer rows are cut into pieces of 4 096 ids that go to a
// queue; k_hash_pieces gives every piece to a block: coalesced loads, the weights stepping by 31^256 per lane.
constexpr int HASH_ROW_LANE = 96, HASH_PIECE = 4096;
__global__ void __launch_bounds__(TPB) k_hash_rest(RangeView v, int what, bool inParse, bool inCopy, int32_t midMin, int32_t bigMin, int2 *__restrict__ pieceq, int32_t *__restrict__ npieces, int32_t cap) {
	__shared__ uint32_t s_part[TPB / 64];
	const HashCtx hx = *v.hx;
	const int64_t r0 = v.rowstart[v.nh];
	const int32_t coopMin = v.coopmin();
	const int32_t s = item_of((uint32_t)v.nh + blockIdx.x * TPB + threadIdx.x);
	uint32_t acc = 0;
	if (s < v.cnt) {
		const int64_t a = v.rowstart[s] - r0, b = v.rowstart[s + 1] - r0;
		const int32_t d = (int32_t)(b - a), r = v.ref[s];
		const uint64_t ebase = (uint64_t)(1 + v.rowstart[s + 1] + (int64_t)s);
		if (what & 4) acc = (uint32_t)(v.lo + s) * hash_upow(hx.ptab, ebase - (uint64_t)d); // the node's own number (when k_scan_apply did not add it)
		bool mine = d > 0 && (uint64_t)b <= v.succ_cap;
		if (r == 0) mine = mine && (what & 1) && (!inParse || d >= coopMin);
		else mine = mine && (what & 2) && (!inCopy || copy_class_of(d, v.outd[s - r], midMin, bigMin) != 1);
		if (mine) {
			if (d < HASH_ROW_LANE) {
				const int32_t *row = v.succ + a;
				uint32_t w = hash_upow(hx.ptab, ebase);
				for (int32_t j = 0; j < d; j++) { acc += (uint32_t)row[j] * w; w *= 31u; }
			} else {
				const int32_t np = (d + HASH_PIECE - 1) / HASH_PIECE, q0 = atomicAdd(npieces, np);
				for (int32_t q = 0; q < np; q++) if (q0 + q < cap) pieceq[q0 + q] = int2{ s, q };
			}
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o, 64);
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < TPB / 64; k++) t += s_part[k]; hash_add(hx, t); }
}
// the same for the rows of a work list (the wave / group classes' parse lists: rows without a reference only, wantRef false; the copy pass's queues of the wave and
// group classes: wantRef true): the lists exist anyway, so nobody has to look at every node to find these rows
__global__ void __launch_bounds__(TPB) k_hash_queue(RangeView v, const int32_t *__restrict__ queue, const int32_t *__restrict__ count, int32_t qcap, bool wantRef,
                                                    int2 *__restrict__ pieceq, int32_t *__restrict__ npieces, int32_t cap) {
	const HashCtx hx = *v.hx;
	const int64_t r0 = v.rowstart[v.nh];
	const int32_t n = min(*count, qcap);
	uint32_t acc = 0;
	for (int32_t i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
		const int32_t s = queue[i];
		if (s < v.nh || (v.ref[s] != 0) != wantRef) continue;
		const int64_t a = v.rowstart[s] - r0, b = v.rowstart[s + 1] - r0;
		const int32_t d = (int32_t)(b - a);
		if (d <= 0 || (uint64_t)b > v.succ_cap) continue;
		if (d < HASH_ROW_LANE) {
			const int32_t *row = v.succ + a;
			uint32_t w = hash_upow(hx.ptab, (uint64_t)(1 + v.rowstart[s + 1] + (int64_t)s));
			for (int32_t j = 0; j < d; j++) { acc += (uint32_t)row[j] * w; w *= 31u; }
		} else {
			const int32_t np = (d + HASH_PIECE - 1) / HASH_PIECE, q0 = atomicAdd(npieces, np);
			for (int32_t q = 0; q < np; q++) if (q0 + q < cap) pieceq[q0 + q] = int2{ s, q };
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o, 64);
	if ((threadIdx.x & 63) == 0) hash_add(hx, acc);
}
__global__ void __launch_bounds__(TPB) k_hash_pieces(RangeView v, const int2 *__restrict__ pieceq, const int32_t *__restrict__ npieces, int32_t cap) {
	__shared__ uint32_t s_part[TPB / 64];
	const HashCtx hx = *v.hx;
	const int64_t r0 = v.rowstart[v.nh];
	const int32_t n = min(*npieces, cap);
	uint32_t step = 1; // 31^256
	for (int k = 0; k < TPB; k++) step *= 31u;
	uint32_t acc = 0;
	for (int32_t k = blockIdx.x; k < n; k += gridDim.x) {
		const int2 e = pieceq[k];
		const int64_t a = v.rowstart[e.x] - r0, b = v.rowstart[e.x + 1] - r0;
		const int32_t d = (int32_t)(b - a), j0 = e.y * HASH_PIECE, j1 = min(d, j0 + HASH_PIECE);
		const int32_t *row = v.succ + a;
		uint32_t w = hash_upow(hx.ptab, (uint64_t)(1 + v.rowstart[e.x + 1] + (int64_t)e.x - (int64_t)(j0 + (int32_t)threadIdx.x))); // u^E * 31^j = u^(E - j), E > j
		for (int32_t j = j0 + threadIdx.x; j < j1; j += TPB) { acc += (uint32_t)row[j] * w; w *= step; }
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o, 64);
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < TPB / 64; k++) t += s_part[k]; hash_add(hx, t); }
}

// ------------------------------------------------------------------------------------------------ long records
// ctl[0] = #big, ctl[1] = #giant, ctl[2] / ctl[3] = heads of the two work queues
constexpr int CLASSIFY_ITEMS = 16; // nodes per thread: long records are rare, most blocks only stream outdegrees
__global__ void __launch_bounds__(TPB) k_classify(int32_t cnt, const int32_t *__restrict__ outd, const int32_t *__restrict__ coopPtr, int32_t coopMin, int32_t giantMin,
                                                  int32_t *__restrict__ biglist, int32_t *__restrict__ giantlist, int32_t giantCap, int32_t *__restrict__ ctl) {
	if (coopPtr) coopMin = min(*coopPtr, giantMin);
	// block-aggregated append: one atomic per block and list instead of one per long record
	__shared__ int32_t s_cnt[2], s_base[2];
	const uint32_t base = blockIdx.x * (TPB * CLASSIFY_ITEMS) + threadIdx.x;
	int32_t d[CLASSIFY_ITEMS];
	bool any = false;
#pragma unroll
	for (int it = 0; it < CLASSIFY_ITEMS; it++) {
		const int32_t s = item_of(base + it * TPB);
		d[it] = s < cnt ? outd[s] : 0;
#ifdef BV_EXP_DROP_LO
		if (d[it] >= BV_EXP_DROP_LO && d[it] < BV_EXP_DROP_HI) d[it] = 0;
#endif
		any |= d[it] >= coopMin;
	}
	if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
	if (!__syncthreads_or(any)) return; // (also orders the zeroing above before the atomics below)
	int32_t local[CLASSIFY_ITEMS];
#pragma unroll
	for (int it = 0; it < CLASSIFY_ITEMS; it++) if (d[it] >= coopMin) local[it] = atomicAdd(&s_cnt[d[it] >= giantMin ? 1 : 0], 1);
	__syncthreads();
	if (threadIdx.x < 2 && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&ctl[threadIdx.x], s_cnt[threadIdx.x]);
	__syncthreads();
#pragma unroll
	for (int it = 0; it < CLASSIFY_ITEMS; it++) {
		const int32_t s = base + it * TPB;
		if (d[it] >= giantMin) { const int32_t k = s_base[1] + local[it]; if (k < giantCap) giantlist[k] = s; } // giantCap >= arcs / giantMin: always fits
		else if (d[it] >= coopMin) biglist[s_base[0] + local[it]] = s;
	}
}

// Longest first: the work queues of the long records are ordered by outdegree, descending, so that the records that
// take longest start first and the short ones fill the gaps at the end (the kernel's duration is otherwise the
// start time of an unlucky long record plus its own length).  One block, counting sort in LDS on the top bits of
// the outdegree (exponent + 3 mantissa bits: order inside a bin does not matter); a list longer than the LDS
// table keeps its (node) order.
constexpr int SORT_CAP = 16384, SORT_BINS = 256;
__global__ void __launch_bounds__(1024) k_sort_desc(int32_t *__restrict__ listA, const int32_t *__restrict__ countA, int32_t capA,
                                                    int32_t *__restrict__ listB, const int32_t *__restrict__ countB, int32_t capB, const int32_t *__restrict__ outd,
                                                    int32_t *__restrict__ started) {
	__shared__ int32_t s_out[SORT_CAP], s_cur[SORT_BINS];
	int32_t *__restrict__ list = blockIdx.x ? listB : listA; // one block per queue, side by side
	const int32_t n = blockIdx.x ? min(*countB, capB) : min(*countA, capA);
	if (blockIdx.x == 0 && threadIdx.x == 0 && started) *started = 0; // (k_wait_giants: groups of the giants' kernel that have started, this job)
	if (n <= 1 || n > SORT_CAP) return;
	auto key = [](int32_t d) { // larger outdegree -> smaller key
		const uint32_t u = (uint32_t)max(d, 1);
		const int lg = 31 - __clz((int)u);
		const uint32_t m = lg >= 3 ? (u >> (lg - 3)) & 7u : (u << (3 - lg)) & 7u;
		return SORT_BINS - 1 - (int)((uint32_t)lg * 8u + m);
	};
	for (int i = threadIdx.x; i < SORT_BINS; i += 1024) s_cur[i] = 0;
	__syncthreads();
	for (int32_t i = threadIdx.x; i < n; i += 1024) atomicAdd(&s_cur[key(outd[list[i]])], 1);
	__syncthreads();
	if (threadIdx.x == 0) { int32_t acc = 0; for (int b = 0; b < SORT_BINS; b++) { const int32_t c = s_cur[b]; s_cur[b] = acc; acc += c; } }
	__syncthreads();
	for (int32_t i = threadIdx.x; i < n; i += 1024) { const int32_t s = list[i]; s_out[atomicAdd(&s_cur[key(outd[s])], 1)] = s; }
	__syncthreads();
	for (int32_t i = threadIdx.x; i < n; i += 1024) list[i] = s_out[i];
}

// One group of NW waves per long record, pulled from a device-side queue.  NW = 1 serves the "big" list,
// NW = GIANT_NW the "giant" list (records so long that a single wave would be the tail of the whole scan).
// What the cooperative decoder needs to know about one long record, for a scan (slot of a node range) and for a
// random-access batch (slot of a query's reference chain).  `prefix` = successors of all the slots before this one in
// one common numbering: it places the record's slice of the interval arena.
struct LongRec { int32_t x, d; bool hasRef; int64_t dref; int32_t *row; int64_t prefix; int capErr; }; // capErr: 0, or the flag of the buffer the row does not fit
__device__ __forceinline__ LongRec long_rec(const RangeView &v, int32_t s) {
	const int32_t r = v.ref[s];
	return LongRec{ v.lo + s, v.outd[s], r > 0, r > 0 ? (int64_t)v.outd[s - r] : 0, v.row(s), v.rowstart[s],
	                v.fits(s) ? 0 : (s >= v.nh ? E_CAP : E_HALO) };
}
__device__ __forceinline__ LongRec long_rec(const BatchView &v, int32_t s) {
	const int32_t qi = v.qidx[s];
	const bool hasRef = v.depth[s] > 0;
	return LongRec{ v.node[s], v.outd[s], hasRef, hasRef ? (int64_t)v.outd[s + 1] : 0, v.row(s), qi >= 0 ? v.arow[v.cnt] + v.rowptr[qi] : v.arow[s],
	                (qi >= 0 && (uint64_t)v.rowptr[qi + 1] > v.succ_cap) ? E_CAP : 0 };
}

template <int DEF, int NW, class View>
#ifndef COOP1_MINWAVES
#define COOP1_MINWAVES 4
#endif
#ifndef COOPG_MINWAVES
#define COOPG_MINWAVES 3 // (round 6, with the groups' tile in dynamic LDS: 168 registers and 128 bytes of scratch per lane instead of 241 -- a group no longer takes every register of its CU)
#endif
__device__ __forceinline__ void parse_big_body(const GraphDev &g, const View &v, const int32_t *__restrict__ list, int32_t *__restrict__ ctl, int which,
                                                       IvEntry *__restrict__ arena, int64_t arenaCap, int *__restrict__ err) {
#ifdef COOPG_DYNLDS
	// The groups' tile in DYNAMIC LDS (round 6; -DCOOPG_STATIC_LDS: as before): with 99 KB of static LDS the compiler sees one group per CU and takes every register that leaves
	// (241), whatever __launch_bounds__ asks for -- and a group of eight waves at 241 registers leaves its CU nothing to run beside it.  With the
	// size out of its sight it honours COOPG_MINWAVES = 3: 168 registers, the kernel alone 0.77 -> 0.82 ms, the scan of C2 2.94 -> 2.83, of the C5 shard 5.35 -> 5.23
	// (profiles/r6_experiments.txt section 6)
	extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
	__shared__ __attribute__((aligned(16))) uint32_t lds_st[NW == 1 ? CoopLds<NW>::WORDS : 4];
	uint32_t *const lds = NW == 1 ? lds_st : lds_dyn;
#else
	__shared__ __attribute__((aligned(16))) uint32_t lds[CoopLds<NW>::WORDS];
#endif
	__shared__ int32_t s_idx;
	const int32_t count = ctl[which]; // (the giant list is sized for arcs / giantMin entries, which bounds their number)
	if (count <= 0) return; // (an empty list -- the usual state of the strip kernel's escape list -- costs no atomics)
	if (std::is_same<View, RangeView>::value && NW != 1 && which == 1 && threadIdx.x == 0) atomicAdd(&ctl[CTL_GIANT_STARTED], 1); // (k_wait_giants; scans only)
	for (;;) {
		if (threadIdx.x == 0) s_idx = atomicAdd(&ctl[2 + which], 1);
		__syncthreads();
		const int32_t idx = s_idx;
		__syncthreads();
		if (idx >= count) break;
		const LongRec rec = long_rec(v, list[idx]);
		// arena slice of this record: interval counts are bounded by d / minIntervalLength, and
		// floor(a/k) + floor(b/k) <= floor((a+b)/k) keeps the slices of different records disjoint
		const int64_t abase = g.minInt > 0 ? rec.prefix / g.minInt : 0;
		// (one path to the end of the loop body: a `continue` behind `if (threadIdx.x == 0) ...` left the lanes of a
		// one-wave block apart at the barrier at the top -- lane 0 late with the next index, the others reading the old one for ever)
		const int bad = rec.capErr ? rec.capErr : (g.minInt > 0 && (abase < 0 || abase + rec.d / g.minInt + 1 > arenaCap)) ? E_FORMAT : 0;
		const unsigned long long t0 = g.stats ? __builtin_readcyclecounter() : 0;
		// (scans only) the residual section of the record is handed to the segment pipeline: descriptor idx of this queue's part
		bvsg::RecDesc *segOut = nullptr;
		// (only records of the long work bins, >= 2 048 bits of work: the pipeline's scratch is sized for those)
		if (std::is_same<View, RangeView>::value && DEF != 0 && which < 2 && g.segDesc && idx < g.segCap[which] && rec.d >= g.segMinD &&
		    ((uint64_t)rec.d * 8 >= 2048 || (uint64_t)(g.offsets[rec.x + 1] - g.offsets[rec.x]) >= 2048)) {
			segOut = (bvsg::RecDesc *)g.segDesc + g.segOff[which] + idx;
			if (threadIdx.x == 0) { *segOut = bvsg::RecDesc{ 0, list[idx], 0, 0, 0, bvsg::RF_SKIP, 0 }; g.segFlag[g.segOff[which] + idx] = 0; }
		}
		if (bad) { if (threadIdx.x == 0) atomicOr(err, bad); }
		else coop_parse_node<DEF, NW>(g, rec.x, rec.d, rec.hasRef, rec.dref, rec.row, arena + abase, lds, err, segOut);
		if (segOut && threadIdx.x == 0) {
			const int32_t ns = bvsg::seg_count(*segOut, (uint64_t)g.offsets[rec.x + 1]); // (0 when the record was decoded here after all)
			// residuals handed over but no piece to decode them from: offsets that put the section at or behind the record's end -- nobody would decode them
			if (ns == 0 && segOut->flags == 0 && segOut->nres > 0) atomicOr(err, E_FORMAT);
			g.segNseg[g.segOff[which] + idx] = ns;
		}
		if (g.stats) { const unsigned long long dt = __builtin_readcyclecounter() - t0; stat_add(g, 5, 1); stat_add(g, 6, dt); stat_max(g, 7, dt); }
	}
}
template <int DEF, int NW, class View>
__global__ void __launch_bounds__(64 * NW, NW == 1 ? COOP1_MINWAVES : COOPG_MINWAVES) k_parse_big(GraphDev g, View v, const int32_t *__restrict__ list, int32_t *__restrict__ ctl, int which,
                                                       IvEntry *__restrict__ arena, int64_t arenaCap, int *__restrict__ err) {
	parse_big_body<DEF, NW, View>(g, v, list, ctl, which, arena, arenaCap, err);
}

// ------------------------------------------------------------------------------------------------ copy
// One lane per node of chain depth `level`: merge the masked copy of the referent's final row with the
// node's extras (sitting at row[copied..d)), forward and in place.  The write index never overtakes the
// extras read index: k = (#copied so far) + (j - copied) <= j.
// HASH (bvg_scan_checksum): every id of the final row is also added to *hacc with weight hw, hw * 31, ... (HashCtx, bv_launch.hpp) -- the merged ones as they are
// written, the extras that are already in place by one more pass over them; a row that is left alone (malformed: an error is raised elsewhere) adds nothing.
template <int DEF, bool HASH>
__device__ __forceinline__ void copy_node(const GraphDev &g, int32_t x, int32_t d, int64_t dref, int32_t *__restrict__ row, const int32_t *__restrict__ src, int *__restrict__ err, uint32_t *hacc, uint32_t hw) {
	BitReader br;
	br.init(g.bits, g.nwords);
	br.seek((uint64_t)g.offsets[x]);
	(void)Fields<DEF>::outdegree(br, g);
	(void)Fields<DEF>::reference(br, g);
	const uint64_t bc = Fields<DEF>::block_count(br, g);
	if (bc > (uint64_t)dref + 1) return; // flagged in k_parse
	const uint64_t blocksPos = br.pos();
	int64_t total = 0, copied = 0;
	for (uint64_t b = 0; b < bc; b++) {
		int64_t len;
		if (!block_len_ok(Fields<DEF>::block(br, g), b == 0, total, dref, len)) return; // flagged by the parse kernel
		total += len;
		if (!(b & 1)) copied += len;
	}
	if (!(bc & 1)) copied += dref - total;
	if (copied > d) return;
	br.seek(blocksPos);

	int64_t i = 0;      // index in the referent row
	int64_t k = 0;      // write index
	int64_t j = copied; // extras read index
	int32_t ev = j < d ? row[j] : 0;
	for (uint64_t b = 0; b <= bc; b++) {
		int64_t len;
		if (b < bc) len = (int64_t)Fields<DEF>::block(br, g) + (b ? 1 : 0);
		else len = dref - i; // implicit last block: the rest of the referent
		if (b & 1) { i += len; continue; } // skip block
		for (int64_t t = 0; t < len && i < dref && k < d; t++) { // (the bounds hold by the checks above: belt and braces)
			const int32_t cv = src[i++];
			while (j < d && ev < cv) { if (HASH) { *hacc += (uint32_t)ev * hw; hw *= 31u; } row[k++] = ev; j++; if (j < d) ev = row[j]; }
			if (j < d && ev == cv) { j++; if (j < d) ev = row[j]; } // equal heads emitted once (never in a valid file)
			if (HASH) { *hacc += (uint32_t)cv * hw; hw *= 31u; }
			row[k++] = cv;
		}
	}
	// remaining extras row[j..d) are already in place when k == j; a malformed duplicate leaves a gap: pad with -1
	if (k != j) { while (j < d) { const int32_t t = row[j++]; if (HASH) { *hacc += (uint32_t)t * hw; hw *= 31u; } row[k++] = t; } while (k < d) { if (HASH) { *hacc -= hw; hw *= 31u; } row[k++] = -1; } }
	else if (HASH) { if (j < d) { *hacc += (uint32_t)ev * hw; hw *= 31u; j++; } for (; j < d; j++) { *hacc += (uint32_t)row[j] * hw; hw *= 31u; } } // (ev = row[j] is at hand)
	if (br.err) atomicOr(err, br.err);
}

// copy_node for the lane class of the copy pass (k_copy_list), with a quarter of its memory instructions.  A wave of k_copy_list
// touches 64 different rows with every load and store, and its counters put it at the rate at which the texture path takes scattered
// cache lines (11.4 M memory instructions per launch on the C5 shard, ~1.6 lines per cycle and CU, full occupancy): what it issues
// per id is what it costs.  Here the referent's ids and the row's extras are read 16 bytes at a time into registers (unaligned
// dwordx4 loads: gfx950 takes them), the merged ids leave 16 bytes at a time once the write index is 16-byte aligned, and the first
// four block lengths stay in registers from the first walk of the list (a second walk only for longer lists, from the fifth code).
// In place, as copy_node: a store goes to k - 4 .. k - 1 with k <= j, and every extra below j has been read (the buffered ones at
// j .. j + 3 too) by then.  Results identical to copy_node's.
typedef int32_t i32x4_u __attribute__((ext_vector_type(4), aligned(4)));
template <int DEF>
__device__ __forceinline__ void copy_node_v(const GraphDev &g, int32_t x, int32_t d, int32_t dref, int32_t *__restrict__ row, const int32_t *__restrict__ src, int *__restrict__ err) {
	BitReader br;
	br.init(g.bits, g.nwords);
	br.seek((uint64_t)g.offsets[x]);
	(void)Fields<DEF>::outdegree(br, g);
	(void)Fields<DEF>::reference(br, g);
	const uint64_t bc64 = Fields<DEF>::block_count(br, g);
	if (bc64 > (uint64_t)dref + 1) return; // flagged in k_parse
	const int32_t bc = (int32_t)bc64;
	int32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
	uint64_t pos4 = 0;
	int64_t total = 0, copied64 = 0;
	for (int32_t b = 0; b < bc; b++) {
		if (b == 4) pos4 = br.pos();
		int64_t len;
		if (!block_len_ok(Fields<DEF>::block(br, g), b == 0, total, (int64_t)dref, len)) return; // flagged by the parse kernel
		if (b == 0) l0 = (int32_t)len; else if (b == 1) l1 = (int32_t)len; else if (b == 2) l2 = (int32_t)len; else if (b == 3) l3 = (int32_t)len;
		total += len;
		if (!(b & 1)) copied64 += len;
	}
	if (!(bc & 1)) copied64 += dref - total;
	if (copied64 > d) return;
	if (copied64 == 0) { if (br.err) atomicOr(err, br.err); return; } // the extras are the row
	if (bc > 4) br.seek(pos4);
	const int32_t copied = (int32_t)copied64;

	// the extras row[copied .. d), four at a time
	int32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, en = 0, ej = copied; // buffered extras (e0 is the head), how many, index of the next one to load
	auto ext_fill = [&] {
		if (ej + 4 <= d) { const i32x4_u q = *(const i32x4_u *)(row + ej); e0 = q.x; e1 = q.y; e2 = q.z; e3 = q.w; en = 4; ej += 4; }
		else if (ej < d) { e0 = row[ej++]; en = 1; }
	};
	auto ext_pop = [&] { e0 = e1; e1 = e2; e2 = e3; if (--en == 0) ext_fill(); };
	ext_fill();
	// the referent's ids, four at a time inside a copied block
	int32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, sn = 0;
	// output: scalar stores up to the first 16-byte boundary, then 16 bytes at a time
	int32_t k = 0, o0 = 0, o1 = 0, o2 = 0, o3 = 0, on = 0;
	const int32_t head = min(d, (int32_t)(((16u - ((uint32_t)(uintptr_t)row & 15u)) & 15u) >> 2));
	auto emit = [&](int32_t val) {
		if (k < head) { row[k++] = val; return; }
		o0 = o1; o1 = o2; o2 = o3; o3 = val; k++;
		if (++on == 4) { *(int4 *)(row + k - 4) = int4{ o0, o1, o2, o3 }; on = 0; }
	};
	int32_t i = 0; // index in the referent's row
	for (int32_t b = 0; b <= bc; b++) {
		int32_t len;
		if (b >= bc) len = dref - i; // implicit last block: the rest of the referent
		else if (b == 0) len = l0; else if (b == 1) len = l1; else if (b == 2) len = l2; else if (b == 3) len = l3;
		else len = (int32_t)Fields<DEF>::block(br, g) + 1;
		if (b & 1) { i += len; continue; } // skip block
		const int32_t end = min(i + len, dref);
		sn = 0;
		while (i < end && k < d) { // (the bounds hold by the checks above: belt and braces)
			if (sn == 0) {
				if (i + 4 <= end) { const i32x4_u q = *(const i32x4_u *)(src + i); s0 = q.x; s1 = q.y; s2 = q.z; s3 = q.w; sn = 4; }
				else { s0 = src[i]; sn = 1; }
			}
			const int32_t cv = s0;
			s0 = s1; s1 = s2; s2 = s3; sn--; i++;
			while (en && e0 < cv && k < d) { emit(e0); ext_pop(); }
			if (en && e0 == cv) ext_pop(); // equal heads emitted once (never in a valid file)
			emit(cv);
		}
	}
	// the pending ids; the remaining extras are in place already when no duplicate was dropped (k + remaining == d)
	const int32_t left = en + (d - ej); // extras not emitted yet
	if (k + left != d) { // a malformed duplicate left a gap: move the rest down, pad with -1 (as copy_node)
		while (en && k < d) { emit(e0); ext_pop(); }
		while (k < d) emit(-1);
	}
	if (on == 3) { row[k - 3] = o1; row[k - 2] = o2; row[k - 1] = o3; }
	else if (on == 2) { row[k - 2] = o2; row[k - 1] = o3; }
	else if (on == 1) row[k - 1] = o3;
	if (br.err) atomicOr(err, br.err);
}

// The same merges from a row's TABLE of copied blocks (k_copy_list<., ., ., true>): hd = the slot's CopyTab -- header (copied << 16 | kept blocks) and the first three
// entries; entry j = first index in the referent's row << 16 | length, entries from the fourth on at tabEnd[2 - j] (the end of the record's own part of the interval arena).
// The table is the parse kernel's own (its blocks passed block_len_ok), but it is read as data: no index leaves the two rows whatever it holds.
template <bool VEC>
__device__ __forceinline__ void copy_node_tab(int32_t d, int32_t dref, int32_t *__restrict__ row, const int32_t *__restrict__ src, const int32_t *__restrict__ tabEnd, int4 hd) {
	const int32_t kept = (int32_t)((uint32_t)hd.w & 0xffffu), copied = (int32_t)((uint32_t)hd.w >> 16);
	if (copied == 0 || copied > d) return; // the extras are the row
	auto entry = [&](int32_t b) -> uint32_t { return (uint32_t)(b == 0 ? hd.z : b == 1 ? hd.y : b == 2 ? hd.x : tabEnd[2 - b]); };
	// The merged ids leave four at a time, in ONE 16-byte store at row + k - 4 whatever its alignment (gfx950 takes dwordx4 on 4-byte boundaries: scripts/ubench_store.hip) -- a
	// memory instruction of this kernel touches 64 lines whatever it carries, so what it issues per id is what it costs.  In place: the store covers k - 4 .. k - 1 with k <= j,
	// the index of the first extra not yet consumed, and every extra below j (the buffered ones too) has been read by then.
	int32_t k = 0, o0 = 0, o1 = 0, o2 = 0, o3 = 0;
	auto emit = [&](int32_t val) {
		o0 = o1; o1 = o2; o2 = o3; o3 = val; k++;
		if ((k & 3) == 0) *(i32x4_u *)(row + k - 4) = i32x4_u{ o0, o1, o2, o3 };
	};
	// the extras row[copied .. d): VEC four at a time (the lane class of a web-shaped graph: long enough rows), else one by one
	int32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, en = 0, ej = copied; // buffered extras (e0 is the head), how many, index of the next one to load
	auto ext_fill = [&] {
		if (VEC && ej + 4 <= d) { const i32x4_u q = *(const i32x4_u *)(row + ej); e0 = q.x; e1 = q.y; e2 = q.z; e3 = q.w; en = 4; ej += 4; }
		else if (ej < d) { e0 = row[ej++]; en = 1; }
	};
	auto ext_pop = [&] { if (VEC) { e0 = e1; e1 = e2; e2 = e3; } if (--en == 0) ext_fill(); };
	ext_fill();
	int32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, sn = 0; // the referent's ids, VEC: four at a time inside a copied block
	for (int32_t b = 0; b < kept; b++) {
		const uint32_t ent = entry(b);
		int32_t i = (int32_t)(ent >> 16);
		const int32_t end = min(i + (int32_t)(ent & 0xffffu), dref);
		sn = 0;
		while (i < end && k < d) { // (the bounds hold for a table of the parse kernel's: belt and braces)
			if (sn == 0) {
				if (VEC && i + 4 <= end) { const i32x4_u q = *(const i32x4_u *)(src + i); s0 = q.x; s1 = q.y; s2 = q.z; s3 = q.w; sn = 4; }
				else { s0 = src[i]; sn = 1; }
			}
			const int32_t cv = s0;
			if (VEC) { s0 = s1; s1 = s2; s2 = s3; }
			sn--; i++;
			while (en && e0 < cv && k < d) { emit(e0); ext_pop(); }
			if (en && e0 == cv) ext_pop(); // equal heads emitted once (never in a valid file)
			emit(cv);
		}
	}
	// the remaining extras are in place already when no duplicate was dropped (k + remaining == d)
	const int32_t left = en + (d - ej); // extras not emitted yet
	if (k + left != d) { // a malformed duplicate left a gap: move the rest down, pad with -1 (as copy_node)
		while (en && k < d) { emit(e0); ext_pop(); }
		while (k < d) emit(-1);
	}
	const int32_t on = k & 3; // the ids still in the buffer
	if (on == 3) { row[k - 3] = o1; row[k - 2] = o2; row[k - 1] = o3; }
	else if (on == 2) { row[k - 2] = o2; row[k - 1] = o3; }
	else if (on == 1) row[k - 1] = o3;
}

// copy_node_tab for all 64 rows of a wave at once (k_copy_list_w).  `have`: this lane has a row with a table; limE / limS: ints readable from the start of the row / of its
// referent's row (>= d / dref: how far a 16-byte load may reach).  Per pass three windows of four ids each are loaded side by side -- A: the referent's ids from index i on
// (the current kept block), B: the first ids of the NEXT kept block (only when A ends inside this pass), E: the row's next extras (they sit at row[copied ..), where the parse
// kernel left them) -- and four trips follow, each emitting min(head of A, head of E) (unsigned; 0xffffffff = nothing left on that side: a row that lost a duplicate -- never in a
// valid file -- pads itself with -1).  A lane that crosses a second block boundary within a pass (or whose windows were single ids at the end of the buffer) sits out the rest of the
// pass.  The merged ids leave 16 bytes at a time at row + k - 4 (in place: k <= copied + the extras consumed, and every extra of the window is in a register by then).
// Memory safety does not depend on the table: i < end <= dref, k < d, and the extras are read below d.
__device__ __forceinline__ void copy_rows_tab(bool have, int32_t d, int32_t dref, int32_t *__restrict__ row, const int32_t *__restrict__ src, int32_t limE, int32_t limS, const int32_t *__restrict__ tabEnd, int4 hd) {
	constexpr uint32_t SENT = 0xffffffffu;
	const int32_t kept = (int32_t)((uint32_t)hd.w & 0xffffu), copied = (int32_t)((uint32_t)hd.w >> 16);
	bool done = !have || kept == 0 || copied == 0 || copied > d; // (copied == 0: the extras are the row)
	// kept blocks 3 .. 6 of the table: one 16-byte load below the end of the record's own part of the arena (entry j at tabEnd[2 - j]; at least one IvEntry there when kept > 3)
	i32x4_u t4 = i32x4_u{ 0, 0, 0, 0 };
	if (!done && kept > 3) t4 = *(const i32x4_u *)(tabEnd - 4);
	auto entry = [&](int32_t j) -> uint32_t {
		if (j >= 7) return (uint32_t)tabEnd[2 - j];
		int32_t e = hd.z;
		e = j == 1 ? hd.y : e; e = j == 2 ? hd.x : e; e = j == 3 ? t4.w : e; e = j == 4 ? t4.z : e; e = j == 5 ? t4.y : e; e = j == 6 ? t4.x : e;
		return (uint32_t)e;
	};
	int32_t b = 0, i = 0, end = 0, k = 0, ec = 0; // kept block, index in the referent's row and the block's end there, ids emitted, extras consumed
	if (!done) { const uint32_t e0 = (uint32_t)hd.z; i = (int32_t)(e0 >> 16); end = min(i + (int32_t)(e0 & 0xffffu), dref); }
	uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;
#ifndef COPY_W_WIDE // 1: windows of eight ids (two 16-byte loads side by side) and eight trips per pass -- half the round trips of a row; 0: four and four (tuning builds)
#define COPY_W_WIDE 1
#endif
	constexpr int WN = COPY_W_WIDE ? 8 : 4; // ids per window = trips per pass
	uint32_t E0 = SENT, E1 = SENT, E2 = SENT, E3 = SENT, E4 = SENT, E5 = SENT, E6 = SENT, E7 = SENT; // the window of extras lives across the passes: a row has few of them (2.8 on cnr-2000), and every load of this kernel is a cache line of its own
	int32_t en = 0;
	while (wave_any(!done)) {
		uint32_t A0 = SENT, A1 = SENT, A2 = SENT, A3 = SENT, A4 = SENT, A5 = SENT, A6 = SENT, A7 = SENT, B0 = SENT, B1 = SENT, B2 = SENT, B3 = SENT;
		int32_t An = 0, Bn = 0, nI = 0, nEnd = 0;
		bool stall = false;
		if (!done) {
			if (i >= end && b + 1 < kept) { b++; const uint32_t e = entry(b); i = (int32_t)(e >> 16); end = min(i + (int32_t)(e & 0xffffu), dref); } // a crossing left over from the last pass
			An = max(0, min(WN, end - i));
			if (An > 0) {
				if (i + 4 <= limS) { const i32x4_u q = *(const i32x4_u *)(src + i); A0 = (uint32_t)q.x; A1 = (uint32_t)q.y; A2 = (uint32_t)q.z; A3 = (uint32_t)q.w; }
				else { A0 = (uint32_t)src[i]; An = 1; }
				if (WN == 8 && An > 4) {
					if (i + 8 <= limS) { const i32x4_u q = *(const i32x4_u *)(src + i + 4); A4 = (uint32_t)q.x; A5 = (uint32_t)q.y; A6 = (uint32_t)q.z; A7 = (uint32_t)q.w; }
					else An = 4;
				}
			}
			if (An < WN && b + 1 < kept) {
				const uint32_t e = entry(b + 1);
				nI = (int32_t)(e >> 16); nEnd = min(nI + (int32_t)(e & 0xffffu), dref);
				Bn = max(0, min(4, nEnd - nI));
				if (Bn > 0) {
					if (nI + 4 <= limS) { const i32x4_u q = *(const i32x4_u *)(src + nI); B0 = (uint32_t)q.x; B1 = (uint32_t)q.y; B2 = (uint32_t)q.z; B3 = (uint32_t)q.w; }
					else { B0 = (uint32_t)src[nI]; Bn = 1; }
				}
			}
			const int32_t pe = copied + ec;
			if (en < 2 && pe + en < d) { // (re)load the window from its head on
				en = min(WN, d - pe);
				if (pe + 4 <= limE) { const i32x4_u q = *(const i32x4_u *)(row + pe); E0 = (uint32_t)q.x; E1 = (uint32_t)q.y; E2 = (uint32_t)q.z; E3 = (uint32_t)q.w; }
				else { E0 = (uint32_t)row[pe]; en = 1; }
				if (WN == 8 && en > 4) {
					if (pe + 8 <= limE) { const i32x4_u q = *(const i32x4_u *)(row + pe + 4); E4 = (uint32_t)q.x; E5 = (uint32_t)q.y; E6 = (uint32_t)q.z; E7 = (uint32_t)q.w; }
					else en = 4;
				}
			}
		}
#pragma unroll
		for (int tr = 0; tr < WN; tr++) {
			const bool act = !done && !stall;
			const uint32_t cv = An > 0 ? A0 : SENT, ev = en > 0 ? E0 : SENT;
			const uint32_t val = min(cv, ev);
			const bool takeC = act && An > 0 && cv <= ev, takeE = act && en > 0 && ev <= cv; // (equal heads: emitted once)
			if (act) {
				o0 = o1; o1 = o2; o2 = o3; o3 = val; k++;
				if ((k & 3) == 0) *(i32x4_u *)(row + k - 4) = i32x4_u{ (int32_t)o0, (int32_t)o1, (int32_t)o2, (int32_t)o3 };
			}
			if (takeE) { E0 = E1; E1 = E2; E2 = E3; if (WN == 8) { E3 = E4; E4 = E5; E5 = E6; E6 = E7; } en--; ec++; if (en == 0 && copied + ec < d) stall = true; } // (more extras, none in a register: a single-id window, or the pass's last trip)
			if (takeC) {
				A0 = A1; A1 = A2; A2 = A3; if (WN == 8) { A3 = A4; A4 = A5; A5 = A6; A6 = A7; } An--; i++;
				if (An == 0) {
					if (i >= end && Bn > 0) { A0 = B0; A1 = B1; A2 = B2; A3 = B3; An = Bn; Bn = 0; b++; i = nI; end = nEnd; }
					else if (i < end || b + 1 < kept) stall = true; // more copied ids, none of them in a register
				}
			}
			// the row is full, or no copied id is left and the remaining extras are where they belong
			if (act && (k >= d || (An == 0 && !stall && k == copied + ec))) done = true;
		}
	}
	if (have) {
		const int32_t on = k & 3; // the ids still in the ring: one store over the last four ids (the ring holds them) when there are four
		if (on != 0 && k >= 4) *(i32x4_u *)(row + k - 4) = i32x4_u{ (int32_t)o0, (int32_t)o1, (int32_t)o2, (int32_t)o3 };
		else if (on == 3) { row[k - 3] = (int32_t)o1; row[k - 2] = (int32_t)o2; row[k - 1] = (int32_t)o3; }
		else if (on == 2) { row[k - 2] = (int32_t)o2; row[k - 1] = (int32_t)o3; }
		else if (on == 1) row[k - 1] = (int32_t)o3;
	}
}

template <int DEF>
__device__ __forceinline__ void read_header(const GraphDev &g, int32_t x, int32_t &d, int32_t &r, int &e) {
	BitReader br;
	br.init(g.bits, g.nwords);
	br.seek((uint64_t)g.offsets[x]);
	uint64_t dd = Fields<DEF>::outdegree(br, g), rr = 0;
	if (dd > 0x7fffffffull) { e |= E_FORMAT; dd = 0; }
	if (dd > 0 && g.W > 0) {
		rr = Fields<DEF>::reference(br, g);
		if (rr > (uint64_t)g.W) { e |= E_REF; rr = 0; }
		else if (rr > (uint64_t)x) { e |= E_FORMAT; rr = 0; }
	}
	e |= br.err;
	d = (int32_t)dd; r = (int32_t)rr;
}

template <int DEF>
__global__ void __launch_bounds__(TPB) k_chain_len(GraphDev g, const int32_t *__restrict__ nodes, int64_t q, int32_t *__restrict__ chainlen,
                                                   int32_t *__restrict__ maxlen, int *__restrict__ err) {
	const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
	int32_t L = 0;
	if (j < q) {
		int32_t y = nodes[j];
		int e = 0;
		if (y < 0 || y >= g.n) e |= E_ARG; // BVG:900
		else for (;;) {
			int32_t d, r;
			read_header<DEF>(g, y, d, r, e);
			L++;
			if (d == 0 || r == 0) break;
			y -= r;
		}
		chainlen[j] = L;
		if (e) atomicOr(err, e);
	}
	int32_t m = L;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
	if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(maxlen, m);
}

template <int DEF>
__global__ void __launch_bounds__(TPB) k_chain_fill(GraphDev g, const int32_t *__restrict__ nodes, int64_t q, const int64_t *__restrict__ slotbase,
                                                    int32_t *__restrict__ snode, int32_t *__restrict__ soutd, int32_t *__restrict__ sdepth,
                                                    int32_t *__restrict__ sq, int32_t *__restrict__ aoutd, int32_t *__restrict__ qoutd) {
	const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
	if (j >= q) return;
	const int64_t base = slotbase[j];
	const int32_t L = (int32_t)(slotbase[j + 1] - base);
	int32_t y = nodes[j];
	if (L == 0) { qoutd[j] = 0; return; }
	for (int32_t t = 0; t < L; t++) {
		int32_t d, r; int e = 0;
		read_header<DEF>(g, y, d, r, e);
		const int64_t s = base + t;
		snode[s] = y; soutd[s] = d; sdepth[s] = L - 1 - t;
		sq[s] = t == 0 ? (int32_t)j : -1;
		aoutd[s] = t == 0 ? 0 : d;
		if (t == 0) qoutd[j] = d;
		y -= r;
	}
}

template <int DEF>
__global__ void __launch_bounds__(TPB) k_bparse(GraphDev g, BatchView v, int *__restrict__ err) {
	const int64_t s = (int64_t)blockIdx.x * TPB + threadIdx.x;
	if (s >= v.cnt) return;
	const int32_t d = v.outd[s];
	if (d == 0) return;
	const int32_t qi = v.qidx[s];
	if (qi >= 0 && (uint64_t)v.rowptr[qi + 1] > v.succ_cap) { atomicOr(err, E_CAP); return; }
	if (d >= v.coop_min) return; // decoded by whole waves (k_parse_big over the batch's slots)
	const bool hasRef = v.depth[s] > 0;
	parse_node<DEF>(g, v.node[s], d, hasRef, hasRef ? (int64_t)v.outd[s + 1] : 0, v.row(s), err);
}

// Slots of BCOPY_WAVE_MIN .. COPY_BIG_MIN - 1 successors are merged by a wave each, slots of up to BCOPY_GROUP_CAP - 1 by a group of four waves (k_bcopy_coop, the method of
// k_copy_mid: block list -> two LDS tables, copied ids and extras gathered into LDS, every id placed by one binary search in the other set); a lane merges ~1 id per
// microsecond, and a batch of 100 000 random ids of C2 ended 0.9 ms after its other slots, behind ONE row of a few thousand ids (round 5).  Longer slots stay with their
// lane (a batch that holds many of them takes the masked scan).
constexpr int BCOPY_WAVE_MIN = 96, BCOPY_GROUP_CAP = 8192;
__device__ __forceinline__ int bcopy_class(int32_t d, int32_t dref) { return (d < BCOPY_WAVE_MIN || dref >= 65536) ? 0 : d < COPY_BIG_MIN ? 1 : d < BCOPY_GROUP_CAP ? 2 : 0; } // 0: a lane, 1: a wave, 2: a group
template <int DEF>
__global__ void __launch_bounds__(TPB) k_bcopy(GraphDev g, BatchView v, int32_t level, int *__restrict__ err) {
	const int64_t s = (int64_t)blockIdx.x * TPB + threadIdx.x;
	if (s >= v.cnt) return;
	if (v.depth[s] != level) return;
	const int32_t qi = v.qidx[s];
	if (qi >= 0 && (uint64_t)v.rowptr[qi + 1] > v.succ_cap) return;
	if (bcopy_class(v.outd[s], v.outd[s + 1]) != 0) return; // k_bcopy_coop
	copy_node<DEF>(g, v.node[s], v.outd[s], (int64_t)v.outd[s + 1], v.row(s), v.row(s + 1), err);
}
// CLS 1: NT = 64, CAP = COPY_BIG_MIN, four groups per block; CLS 2: NT = 256, CAP = BCOPY_GROUP_CAP, one group per block
template <int DEF, int CLS>
__global__ void __launch_bounds__(256) k_bcopy_coop(GraphDev g, BatchView v, int32_t level, int *__restrict__ err) {
	constexpr int NT = CLS == 1 ? 64 : 256, NG = 256 / NT, CAP = CLS == 1 ? COPY_BIG_MIN : BCOPY_GROUP_CAP;
	__shared__ int32_t s_vals[NG][CAP], s_kend[NG][CAP + 1], s_delta[NG][CAP + 1];
	__shared__ int32_t s_list[NG][NT], s_n[NG];
	const int tid = threadIdx.x % NT, grp = threadIdx.x / NT;
	int32_t *vals = s_vals[grp], *kend = s_kend[grp], *delta = s_delta[grp];
	auto group_sync = [] {
		if (NT == 64) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }
		else __syncthreads();
	};
	const int64_t nGroups = (int64_t)gridDim.x * NG, w = (int64_t)blockIdx.x * NG + grp;
	for (int64_t s0 = w * NT; s0 < v.cnt; s0 += nGroups * NT) { // (uniform in the group: NT slots at a time, a thread looks at one)
		if (tid == 0) s_n[grp] = 0;
		group_sync();
		const int64_t sL = s0 + tid;
		bool mine = sL < v.cnt && v.depth[sL] == level;
		if (mine) { const int32_t qi = v.qidx[sL]; mine = !(qi >= 0 && (uint64_t)v.rowptr[qi + 1] > v.succ_cap) && bcopy_class(v.outd[sL], v.outd[sL + 1]) == CLS; }
		if (mine) s_list[grp][atomicAdd(&s_n[grp], 1)] = tid;
		group_sync();
		const int32_t nl = s_n[grp];
		for (int32_t q = 0; q < nl; q++) {
			const int64_t s = s0 + s_list[grp][q];
			const int32_t d = v.outd[s];
			const int64_t dref = v.outd[s + 1];
			int32_t *row = v.row(s);
			const int32_t *src = v.row(s + 1);
			// header + blocks, every thread alike (BVG:1058-1071)
			BitReader br;
			br.init(g.bits, g.nwords);
			br.seek((uint64_t)g.offsets[v.node[s]]);
			(void)Fields<DEF>::outdegree(br, g);
			(void)Fields<DEF>::reference(br, g);
			const uint64_t bc = Fields<DEF>::block_count(br, g);
			int64_t total = 0, copied = 0;
			int32_t nKept = 0;
			bool bad = bc > (uint64_t)dref + 1, full = false; // (bad: flagged by the parse kernel)
			for (uint64_t b = 0; !bad && b <= bc; b++) {
				int64_t len;
				if (b < bc) len = (int64_t)Fields<DEF>::block(br, g) + (b ? 1 : 0);
				else len = dref - total; // implicit last block (copied when the block count is even)
				if (len < 0 || total + len > dref) { bad = true; break; }
				if (!(b & 1)) {
					if (copied + len > d) { bad = true; break; }
					if (nKept > CAP) { full = true; break; } // (the tables hold CAP + 1 kept blocks: a valid record of CAP - 1 ids copied one by one behind an empty first block and in front of an empty last one has that many -- ADVICE r5; more: not malformed, the lane-serial merge takes the row)
					if (tid == (nKept % NT)) { kend[nKept] = (int32_t)(copied + len); delta[nKept] = (int32_t)(total - copied); }
					nKept++;
					copied += len;
				}
				total += len;
			}
			if (br.err && tid == 0) atomicOr(err, br.err);
			const bool skip = bad || br.err || full || copied == 0; // malformed (flagged by the parse kernel), nothing to merge, or more kept blocks than the tables hold (uniform)
			const int32_t nExtra = d - (int32_t)copied, nc = (int32_t)copied;
			group_sync();
			if (!skip) {
				for (int32_t t = tid; t < nc; t += NT) { // copied ids -> vals[0 .. nc)
					int32_t lo = 0, hi = nKept; // first kept block with kend > t
					while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (kend[mid] <= t) lo = mid + 1; else hi = mid; }
					vals[t] = src[t + delta[lo]];
				}
				for (int32_t e = tid; e < nExtra; e += NT) vals[nc + e] = row[nc + e]; // extras -> vals[nc .. d)
			}
			group_sync();
			bool dup = full && !bad && !br.err; // an id in both sets (never in a valid file): the lane-serial merge emits equal heads once and pads the row (copy_node); it also takes the rows whose tables overflowed
			int32_t place[CAP / NT];
#pragma unroll
			for (int k = 0; k < CAP / NT; k++) {
				const int32_t t = tid + NT * k;
				place[k] = -1;
				if (!skip && t < d) {
					const int32_t val = vals[t];
					int32_t lo, hi;
					if (t < nc) { lo = nc; hi = d; } else { lo = 0; hi = nc; }
					const int32_t end = hi;
					while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (vals[mid] < val) lo = mid + 1; else hi = mid; }
					dup |= lo < end && vals[lo] == val;
					place[k] = t < nc ? t + lo - nc : t - nc + lo;
				}
			}
			const bool anyDup = NT == 64 ? (bool)__any(dup) : (bool)__syncthreads_or(dup);
			if (anyDup) { if (tid == 0) copy_node<DEF>(g, v.node[s], d, dref, row, src, err); }
			else {
#pragma unroll
				for (int k = 0; k < CAP / NT; k++) if (place[k] >= 0) row[place[k]] = vals[tid + NT * k];
			}
			group_sync(); // (the tables are reused by the next row)
		}
	}
}

// ------------------------------------------------------------------------------------------------ hashCode
// ImmutableGraph.hashCode (ImmutableGraph.java:757-770): h = -1; for x: h = 31h + x; for j = d-1..0: h = 31h + s[j].
// Each step is the affine map h -> 31h + v over Z/2^32; maps compose associatively, so a node contributes
// (A_x, B_x) = (31^(d+1), ...) and blocks combine in order.  One lane per node, block-level ordered reduce.
struct Affine { uint32_t a, b; }; // h -> a*h + b
__device__ __forceinline__ Affine compose(Affine f, Affine g2) { return Affine{ f.a * g2.a, g2.a * f.b + g2.b }; } // g2 after f

__device__ __forceinline__ uint32_t pow31(uint64_t e) { // 31^e mod 2^32
	uint32_t r = 1u, m = 31u;
	while (e) { if (e & 1) r *= m; m *= m; e >>= 1; }
	return r;
}
// The scan order of hashCode() is ONE sequence of n + m values -- node x, then its successors from the last to the first -- and the
// position of node x's header in it is P(x) = (rowptr[x] - rowptr[0]) + x.  The sequence is cut into chunks of HASH_CHUNK positions,
// whatever the rows are (one lane per node made the kernel 1.6 ms on C2, one block per 1 024 nodes 70 ms on C5, where thirty
// neighbouring rows hold 5.6 M ids): k_hash_bounds finds the node that holds every chunk's first position; a block loads the rowptr
// slice of its chunk into LDS and its threads walk the chunk's positions backwards with stride TPB -- so the weight 31^(values
// that follow in the chunk) of a thread's next value is its last one times 31^TPB -- looking each position's node up in LDS.
// Successors are read in descending order of index: coalesced.
constexpr int HASH_CHUNK = 4096;
__global__ void __launch_bounds__(TPB) k_hash_bounds(int32_t cnt, const int64_t *__restrict__ rowptr, int64_t nchunks, int32_t *__restrict__ bounds) {
	const int64_t c = (int64_t)blockIdx.x * TPB + threadIdx.x;
	if (c > nchunks) return;
	if (c == nchunks) { bounds[c] = cnt - 1; return; }
	const int64_t target = c * HASH_CHUNK, r0 = rowptr[0];
	int32_t lo = 0, hi = cnt; // last x with P(x) <= target (P(0) = 0)
	while (hi - lo > 1) { const int32_t mid = lo + ((hi - lo) >> 1); if ((rowptr[mid] - r0) + mid <= target) lo = mid; else hi = mid; }
	bounds[c] = lo;
}
__global__ void __launch_bounds__(TPB) k_hash_nodes(int32_t from, int32_t cnt, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ,
                                                    const int32_t *__restrict__ bounds, int64_t L, uint32_t *__restrict__ outA, uint32_t *__restrict__ outB) {
	__shared__ int64_t R[HASH_CHUNK + 4];
	__shared__ int32_t coarse[HASH_CHUNK / 64 + 1];
	__shared__ uint32_t s_part[TPB];
	const int64_t c = blockIdx.x, start = c * HASH_CHUNK, end = min(L, start + HASH_CHUNK), r0 = rowptr[0];
	const int32_t xa = bounds[c], xb = bounds[c + 1], nx = xb - xa + 1; // (nodes xa .. xb: at most HASH_CHUNK + 1 of them start in [start, end])
	for (int32_t k = threadIdx.x; k <= nx; k += TPB) R[k] = rowptr[xa + k];
	__syncthreads();
	auto P = [&](int32_t k) { return (R[k] - r0) + (xa + k); }; // position of node xa + k's header
	auto node_of = [&](int64_t p, int32_t lo, int32_t hi) { // last k in [lo, hi) with P(k) <= p
		while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (P(mid) <= p) lo = mid; else hi = mid; }
		return lo;
	};
	// the node of every 64th position, by binary search, once; a value then finds its node a few headers further on (a search of
	// thirteen dependent LDS reads per value made the kernel 0.96 ms on C2)
	constexpr int HASH_STRIDE = 64, HASH_SCAN = 12;
	const int32_t ncoarse = (int32_t)((end - start + HASH_STRIDE - 1) / HASH_STRIDE);
	for (int32_t t = threadIdx.x; t < ncoarse; t += TPB) coarse[t] = node_of(start + (int64_t)t * HASH_STRIDE, 0, nx);
	__syncthreads();
	// where value p lives: a node's own number (from + x, flagged by a negative index) or an index into succ
	auto locate = [&](int64_t p) -> int64_t {
		const int32_t t = (int32_t)((p - start) / HASH_STRIDE);
		int32_t lo = coarse[t], n = 0;
		while (lo + 1 < nx && P(lo + 1) <= p) { lo++; if (++n == HASH_SCAN) { lo = node_of(p, lo, t + 1 < ncoarse ? coarse[t + 1] + 1 : nx); break; } } // (a run of empty nodes)
		// position 0 of a node is its header; position q + 1 its successor d - 1 - q (ImmutableGraph.java:757-770 runs over them backwards)
		const int64_t pos = p - P(lo);
		return pos == 0 ? -1 - (int64_t)(xa + lo) : R[lo + 1] - pos;
	};
	uint32_t acc = 0, w = pow31((uint64_t)threadIdx.x);
	const uint32_t step = pow31(TPB), step2 = step * step, step3 = step2 * step, step4 = step2 * step2;
	int64_t p = end - 1 - threadIdx.x;
	for (; p - 3 * TPB >= start; p -= 4 * TPB, w *= step4) { // four loads in flight per lane (one at a time: 0.87 ms on C2, the lanes waiting)
		const int64_t a0 = locate(p), a1 = locate(p - TPB), a2 = locate(p - 2 * TPB), a3 = locate(p - 3 * TPB);
		const uint32_t v0 = a0 < 0 ? (uint32_t)(from - 1 - a0) : (uint32_t)succ[a0], v1 = a1 < 0 ? (uint32_t)(from - 1 - a1) : (uint32_t)succ[a1];
		const uint32_t v2 = a2 < 0 ? (uint32_t)(from - 1 - a2) : (uint32_t)succ[a2], v3 = a3 < 0 ? (uint32_t)(from - 1 - a3) : (uint32_t)succ[a3];
		acc += v0 * w + v1 * (w * step) + v2 * (w * step2) + v3 * (w * step3);
	}
	for (; p >= start; p -= TPB, w *= step) {
		const int64_t a = locate(p);
		acc += (a < 0 ? (uint32_t)(from - 1 - a) : (uint32_t)succ[a]) * w;
	}
	s_part[threadIdx.x] = acc;
	__syncthreads();
	for (int o = TPB / 2; o > 0; o >>= 1) { if (threadIdx.x < o) s_part[threadIdx.x] += s_part[threadIdx.x + o]; __syncthreads(); }
	if (threadIdx.x == 0) { outA[c] = pow31((uint64_t)(end - start)); outB[c] = s_part[0]; }
}

// 256 consecutive maps -> one (the single block below needed 0.24 ms for the 51 000 chunk maps of a C2 scan: 200 rounds of a tree with eight barriers)
__global__ void __launch_bounds__(TPB) k_hash_reduce(const uint32_t *__restrict__ A, const uint32_t *__restrict__ B, int64_t nb, uint32_t *__restrict__ A2, uint32_t *__restrict__ B2) {
	__shared__ Affine sw[TPB / 64];
	const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
	const int lane = threadIdx.x & 63;
	Affine v = j < nb ? Affine{ A[j], B[j] } : Affine{ 1u, 0u };
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { // inclusive scan in order: lane l = maps of lanes 0 .. l composed
		const uint32_t pa = __shfl_up(v.a, o), pb = __shfl_up(v.b, o);
		if (lane >= o) v = compose(Affine{ pa, pb }, v);
	}
	if (lane == 63) sw[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) { Affine all = sw[0]; for (int w = 1; w < TPB / 64; w++) all = compose(all, sw[w]); A2[blockIdx.x] = all.a; B2[blockIdx.x] = all.b; }
}

// single block: fold nb block maps in order and apply to *hash
__global__ void __launch_bounds__(TPB) k_hash_fold(const uint32_t *__restrict__ A, const uint32_t *__restrict__ B, int64_t nb, int32_t *__restrict__ hash) {
	__shared__ Affine sh[TPB];
	Affine acc{ 1u, 0u };
	for (int64_t base = 0; base < nb; base += TPB) {
		const int64_t j = base + threadIdx.x;
		sh[threadIdx.x] = j < nb ? Affine{ A[j], B[j] } : Affine{ 1u, 0u };
		__syncthreads();
		for (int o = 1; o < TPB; o <<= 1) {
			if ((threadIdx.x % (2 * o)) == 0 && threadIdx.x + o < TPB) sh[threadIdx.x] = compose(sh[threadIdx.x], sh[threadIdx.x + o]);
			__syncthreads();
		}
		if (threadIdx.x == 0) acc = compose(acc, sh[0]);
		__syncthreads();
	}
	if (threadIdx.x == 0) *hash = (int32_t)(acc.a * (uint32_t)*hash + acc.b);
}

} // namespace bv

// ------------------------------------------------------------------------------------------------ launchers

namespace bv {

static inline unsigned nblk(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

void launch_headers(const GraphDev &g, int def, int32_t lo, int32_t cnt, int32_t *outd, uint16_t *ref, int *err, hipStream_t st, int32_t *part, uint8_t *mark,
                    uint16_t *pkey16, int32_t *phist, bool pwindows) {
	if (cnt <= 0) return;
	if (mark) (void)hipMemsetAsync(mark, 0, (size_t)cnt, st);
	if (def == 1) hipLaunchKernelGGL(k_headers<1>, dim3(nblk(cnt, TPB)), dim3(TPB), 0, st, g, lo, cnt, outd, ref, err, part, mark, pkey16, phist, pwindows ? 1 : 0);
	else if (def == 2) hipLaunchKernelGGL(k_headers<2>, dim3(nblk(cnt, TPB)), dim3(TPB), 0, st, g, lo, cnt, outd, ref, err, part, mark, pkey16, phist, pwindows ? 1 : 0);
	else hipLaunchKernelGGL(k_headers<0>, dim3(nblk(cnt, TPB)), dim3(TPB), 0, st, g, lo, cnt, outd, ref, err, part, mark, pkey16, phist, pwindows ? 1 : 0);
}
// the parse list from keys and a histogram that k_headers left (launch_headers with pkey16): the two kernels that remain of launch_build_lists
void launch_scatter_lists(int32_t cnt, const uint16_t *key16, const int32_t *hist, int32_t *keyBase, int32_t *cursor, int32_t *list, int32_t *giantlist, int32_t *ctl, int32_t *maxdepth, hipStream_t st, const uint16_t *packRef) {
	if (cnt <= 0) return;
	(void)hipMemsetAsync((void *)hist, 0, sizeof(int32_t) * NKEYS, st);
	hipLaunchKernelGGL(k_key_hist, dim3(nblk(cnt, LIST_TILE)), dim3(TPB), 0, st, cnt, key16, (int32_t *)hist);
	hipLaunchKernelGGL(k_key_offsets, dim3(1), dim3(TPB), 0, st, hist, keyBase, cursor, maxdepth);
	hipLaunchKernelGGL(k_scatter_keys, dim3(nblk(cnt, LIST_TILE)), dim3(TPB), 0, st, cnt, key16, cursor, list, giantlist, 0, ctl, packRef);
}
__global__ void __launch_bounds__(HASH_ACC_SLOTS) k_hash_sum(const HashCtx *__restrict__ hx, int32_t *__restrict__ out) {
	__shared__ uint32_t s_part[HASH_ACC_SLOTS / 64];
	uint32_t acc = hx->acc[threadIdx.x];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o, 64);
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < HASH_ACC_SLOTS / 64; k++) t += s_part[k]; *out = (int32_t)t; }
}
void launch_hash_sum(const HashCtx *hx, int32_t *out, hipStream_t st) { hipLaunchKernelGGL(k_hash_sum, dim3(1), dim3(HASH_ACC_SLOTS), 0, st, hx, out); }
void copy_thresholds(int32_t midMinKnob, bool bigGroups, int32_t &midMin, int32_t &bigMin);
void launch_hash_rest(const RangeView &v, int what, bool inParse, bool inCopy, int32_t midMinKnob, bool bigGroups, void *pieceq, int32_t *npieces, int32_t cap, hipStream_t st,
                      const int32_t *qA, const int32_t *nA, int32_t capA, const int32_t *qB, const int32_t *nB, int32_t capB, bool wantRef) {
	if (v.cnt <= v.nh) return;
	int32_t midMin, bigMin;
	copy_thresholds(midMinKnob, bigGroups, midMin, bigMin);
	(void)hipMemsetAsync(npieces, 0, sizeof(int32_t), st);
	if (what) hipLaunchKernelGGL(k_hash_rest, dim3(nblk((int64_t)v.cnt - v.nh, TPB)), dim3(TPB), 0, st, v, what, inParse, inCopy, midMin, bigMin, (int2 *)pieceq, npieces, cap);
	if (qA && capA > 0) hipLaunchKernelGGL(k_hash_queue, dim3(256), dim3(TPB), 0, st, v, qA, nA, capA, wantRef, (int2 *)pieceq, npieces, cap);
	if (qB && capB > 0) hipLaunchKernelGGL(k_hash_queue, dim3(256), dim3(TPB), 0, st, v, qB, nB, capB, wantRef, (int2 *)pieceq, npieces, cap);
	hipLaunchKernelGGL(k_hash_pieces, dim3(2048), dim3(TPB), 0, st, v, (const int2 *)pieceq, npieces, cap);
}
int64_t headers_blocks(int32_t cnt) { return cnt > 0 ? (int64_t)nblk(cnt, TPB) : 0; }

void launch_mark_halo(int32_t nh, int32_t cnt, int32_t W, int32_t *outd, uint16_t *ref, uint8_t *need, int *err, hipStream_t st) {
	if (nh <= 0 || W <= 0) return;
	(void)hipMemsetAsync(need, 0, (size_t)nh, st);
	hipLaunchKernelGGL(k_mark_halo, dim3(nblk(W, 64)), dim3(64), 0, st, nh, cnt, W, outd, ref, need, err);
	hipLaunchKernelGGL(k_apply_need, dim3(nblk(nh, TPB)), dim3(TPB), 0, st, nh, need, outd, ref);
}

void launch_scan(const int32_t *in, int64_t n, int64_t *out, int64_t *sums, hipStream_t st, const HashCtx *hx, int32_t lo, int32_t nh, long long topTiledMin) {
	const int64_t nb = n > 0 ? (n + SCAN_TILE - 1) / SCAN_TILE : 1;
	hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(TPB), 0, st, in, n, sums);
	const int64_t tiledMin = topTiledMin > 0 ? (int64_t)topTiledMin : (int64_t)SCAN_TOP_TILED_MIN;
	if (nb >= tiledMin) hipLaunchKernelGGL(k_scan_top_tiled, dim3(1), dim3(SCAN_TOP_T), 0, st, sums, nb);
	else hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(TPB), 0, st, sums, nb);
	if (hx) hipLaunchKernelGGL(k_scan_apply<true>, dim3((unsigned)nb), dim3(TPB), 0, st, in, n, sums, out, hx, lo, nh);
	else hipLaunchKernelGGL(k_scan_apply<false>, dim3((unsigned)nb), dim3(TPB), 0, st, in, n, sums, out, hx, lo, nh);
}
int64_t scan_num_sums(int64_t n) { return n > 0 ? (n + SCAN_TILE - 1) / SCAN_TILE : 1; }


bool launch_query_mark(const int32_t *nodes, int64_t q, int32_t n, int32_t *outd, uint16_t *ref, uint8_t *need, int32_t *qoutd, int passes, int32_t *changed, int *err, hipStream_t st) {
	if (need) (void)hipMemsetAsync(need, 0, (size_t)n, st);
	// many queries: marks closed by streaming passes over the nodes (25 us each on 10 M nodes); few: a chain walk per query
	const bool walk = q * 4 < (int64_t)n;
	hipLaunchKernelGGL(k_query_mark, dim3(nblk(q, TPB)), dim3(TPB), 0, st, nodes, q, n, outd, ref, need, qoutd, walk ? 1 : 0, err);
	if (need && !walk) launch_need_prop(n, outd, ref, need, passes, changed, st);
	return need && !walk; // true: *changed must be looked at (and more passes run) before the marks are used
}
// the marks again from scratch, a chain walk per query (for chains too deep for streaming passes)
void launch_query_walk(const int32_t *nodes, int64_t q, int32_t n, int32_t *outd, uint16_t *ref, uint8_t *need, int32_t *qoutd, int *err, hipStream_t st) {
	(void)hipMemsetAsync(need, 0, (size_t)n, st);
	hipLaunchKernelGGL(k_query_mark, dim3(nblk(q, TPB)), dim3(TPB), 0, st, nodes, q, n, outd, ref, need, qoutd, 1, err);
}
// `passes` steps of the closure, then one more that reports into *changed whether it still found something to mark
void launch_need_prop(int32_t n, const int32_t *outd, const uint16_t *ref, uint8_t *need, int passes, int32_t *changed, hipStream_t st) {
	for (int p = 0; p < passes; p++) hipLaunchKernelGGL(k_need_prop, dim3(nblk(n, TPB)), dim3(TPB), 0, st, n, outd, ref, need, (int32_t *)nullptr);
	hipLaunchKernelGGL(k_need_prop, dim3(nblk(n, TPB)), dim3(TPB), 0, st, n, outd, ref, need, changed);
}
void launch_apply_need(int32_t n, const uint8_t *need, int32_t *outd, uint16_t *ref, hipStream_t st) {
	hipLaunchKernelGGL(k_apply_need, dim3(nblk(n, TPB)), dim3(TPB), 0, st, n, need, outd, ref);
}
void launch_gather_rows(const int32_t *nodes, int64_t q, int64_t arcs, const int64_t *rowstart, const int32_t *arena, const int64_t *rowptr, int32_t *succ, hipStream_t st) {
	if (q <= 0) return;
	const unsigned bx = nblk(q, GATHER_ROWS);
	const unsigned by = (unsigned)std::min<int64_t>(std::max<int64_t>(arcs / bx / 16384, 1), 32768); // ~16 K ids per block
	hipLaunchKernelGGL(k_gather_rows, dim3(bx, by), dim3(GATHER_ROWS), 0, st, nodes, q, rowstart, arena, rowptr, succ);
}
void launch_rebase(int32_t nh, int32_t cnt, const int64_t *rowstart, int64_t *out, hipStream_t st) {
	hipLaunchKernelGGL(k_rebase, dim3(nblk((int64_t)cnt - nh + 1, TPB)), dim3(TPB), 0, st, nh, cnt, rowstart, out);
}



int64_t hash_chunks(int32_t cnt, int64_t arcs) { return ((int64_t)cnt + arcs + HASH_CHUNK - 1) / HASH_CHUNK; }
// A, B: hash_chunks(cnt, arcs) entries each; bounds: one more.  arcs = rowptr[cnt] - rowptr[0].
void launch_hash(int32_t from, int32_t cnt, int64_t arcs, const int64_t *rowptr, const int32_t *succ, uint32_t *A, uint32_t *B, int32_t *bounds, int32_t *hash, hipStream_t st) {
	if (cnt <= 0) return;
	const int64_t nc = hash_chunks(cnt, arcs);
	hipLaunchKernelGGL(k_hash_bounds, dim3(nblk(nc + 1, TPB)), dim3(TPB), 0, st, cnt, rowptr, nc, bounds);
	hipLaunchKernelGGL(k_hash_nodes, dim3((unsigned)nc), dim3(TPB), 0, st, from, cnt, rowptr, succ, bounds, (int64_t)cnt + arcs, A, B);
	if (nc >= 4 * TPB) { // the chunk maps are reduced 256 to one first; `bounds` (nc + 1 words, done with) holds the reduced maps
		const int64_t n2 = (nc + TPB - 1) / TPB;
		uint32_t *A2 = (uint32_t *)bounds, *B2 = A2 + n2;
		hipLaunchKernelGGL(k_hash_reduce, dim3((unsigned)n2), dim3(TPB), 0, st, A, B, nc, A2, B2);
		hipLaunchKernelGGL(k_hash_fold, dim3(1), dim3(TPB), 0, st, A2, B2, n2, hash);
	} else hipLaunchKernelGGL(k_hash_fold, dim3(1), dim3(TPB), 0, st, A, B, nc, hash);
}

void launch_chain_len(const GraphDev &g, int def, const int32_t *nodes, int64_t q, int32_t *chainlen, int32_t *maxlen, int *err, hipStream_t st) {
	if (q <= 0) return;
	if (def == 1) hipLaunchKernelGGL(k_chain_len<1>, dim3(nblk(q, TPB)), dim3(TPB), 0, st, g, nodes, q, chainlen, maxlen, err);
	else if (def == 2) hipLaunchKernelGGL(k_chain_len<2>, dim3(nblk(q, TPB)), dim3(TPB), 0, st, g, nodes, q, chainlen, maxlen, err);
	else hipLaunchKernelGGL(k_chain_len<0>, dim3(nblk(q, TPB)), dim3(TPB), 0, st, g, nodes, q, chainlen, maxlen, err);
}

void launch_chain_fill(const GraphDev &g, int def, const int32_t *nodes, int64_t q, const int64_t *slotbase, int32_t *snode, int32_t *soutd,
                       int32_t *sdepth, int32_t *sq, int32_t *aoutd, int32_t *qoutd, hipStream_t st) {
	if (q <= 0) return;
	if (def == 1) hipLaunchKernelGGL(k_chain_fill<1>, dim3(nblk(q, TPB)), dim3(TPB), 0, st, g, nodes, q, slotbase, snode, soutd, sdepth, sq, aoutd, qoutd);
	else if (def == 2) hipLaunchKernelGGL(k_chain_fill<2>, dim3(nblk(q, TPB)), dim3(TPB), 0, st, g, nodes, q, slotbase, snode, soutd, sdepth, sq, aoutd, qoutd);
	else hipLaunchKernelGGL(k_chain_fill<0>, dim3(nblk(q, TPB)), dim3(TPB), 0, st, g, nodes, q, slotbase, snode, soutd, sdepth, sq, aoutd, qoutd);
}

void launch_bparse(const GraphDev &g, int def, const BatchView &v, int *err, hipStream_t st) {
	if (v.cnt <= 0) return;
	if (def == 1) hipLaunchKernelGGL(k_bparse<1>, dim3(nblk(v.cnt, TPB)), dim3(TPB), 0, st, g, v, err);
	else if (def == 2) hipLaunchKernelGGL(k_bparse<2>, dim3(nblk(v.cnt, TPB)), dim3(TPB), 0, st, g, v, err);
	else hipLaunchKernelGGL(k_bparse<0>, dim3(nblk(v.cnt, TPB)), dim3(TPB), 0, st, g, v, err);
}

void launch_bcopy(const GraphDev &g, int def, const BatchView &v, int32_t level, int *err, hipStream_t st) {
	if (v.cnt <= 0) return;
	if (def == 1) hipLaunchKernelGGL(k_bcopy<1>, dim3(nblk(v.cnt, TPB)), dim3(TPB), 0, st, g, v, level, err);
	else if (def == 2) hipLaunchKernelGGL(k_bcopy<2>, dim3(nblk(v.cnt, TPB)), dim3(TPB), 0, st, g, v, level, err);
	else hipLaunchKernelGGL(k_bcopy<0>, dim3(nblk(v.cnt, TPB)), dim3(TPB), 0, st, g, v, level, err);
	// (the slots that a wave or a group merges: the same level, other rows)
	const unsigned wb = (unsigned)std::min<int64_t>(nblk(v.cnt, 256), 1024), gb = (unsigned)std::min<int64_t>(nblk(v.cnt, 256), 256);
	if (def == 1) { hipLaunchKernelGGL((k_bcopy_coop<1, 1>), dim3(wb), dim3(256), 0, st, g, v, level, err); hipLaunchKernelGGL((k_bcopy_coop<1, 2>), dim3(gb), dim3(256), 0, st, g, v, level, err); }
	else if (def == 2) { hipLaunchKernelGGL((k_bcopy_coop<2, 1>), dim3(wb), dim3(256), 0, st, g, v, level, err); hipLaunchKernelGGL((k_bcopy_coop<2, 2>), dim3(gb), dim3(256), 0, st, g, v, level, err); }
	else { hipLaunchKernelGGL((k_bcopy_coop<0, 1>), dim3(wb), dim3(256), 0, st, g, v, level, err); hipLaunchKernelGGL((k_bcopy_coop<0, 2>), dim3(gb), dim3(256), 0, st, g, v, level, err); }
}

// Which records leave the one-lane decoder for a wave: counted, not guessed.  A lane decodes ~0.6 us per successor whatever its
// neighbours do, so the lane class lasts as long as its longest record unless it holds enough records of that length to fill
// whole waves with them (the parse list is sorted by length); a wave costs ~30 us per record but 3 072 of them run side by side.
// The threshold is the smallest of 128 .. 8192 that sends at most `budget` records to the waves (C2: 7 121 records >= 2 048,
// 15 410 >= 1 024 -> 2 048; cnr-2000 x 30: 11 250 >= 128 -> 128, 3.06 -> 2.67 ms; the same generator at 50 M nodes / 1 B arcs:
// 8 192, 14.7 -> 13.3 ms -- a scan five times as long hides a lane four times as long; measured optimum in all three cases).
constexpr int PICK_THREADS = 1024;
__global__ void __launch_bounds__(PICK_THREADS) k_pick_coop(const int32_t *__restrict__ part, int32_t nblocks, int32_t budget, int32_t *__restrict__ ctl, int32_t *__restrict__ counts) {
	__shared__ int32_t s_cnt[PICK_LEVELS];
	if (threadIdx.x < PICK_LEVELS) s_cnt[threadIdx.x] = 0;
	if (!counts && threadIdx.x >= 64 && threadIdx.x < 64 + 12) ctl[4 + (threadIdx.x - 64)] = 0; // counters of the level lists, copy queues and copy levels of this job
	__syncthreads();
#pragma unroll
	for (int k = 0; k < PICK_LEVELS; k++) {
		int32_t t = 0;
		const int32_t *pk = part + (size_t)k * nblocks;
		int32_t b = threadIdx.x;
		for (; b + 7 * PICK_THREADS < nblocks; b += 8 * PICK_THREADS) { // eight loads in flight
			const int32_t x0 = pk[b], x1 = pk[b + PICK_THREADS], x2 = pk[b + 2 * PICK_THREADS], x3 = pk[b + 3 * PICK_THREADS], x4 = pk[b + 4 * PICK_THREADS],
			              x5 = pk[b + 5 * PICK_THREADS], x6 = pk[b + 6 * PICK_THREADS], x7 = pk[b + 7 * PICK_THREADS];
			t += ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
		}
		for (; b < nblocks; b += PICK_THREADS) t += pk[b];
		for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
		if ((threadIdx.x & 63) == 0 && t) atomicAdd(&s_cnt[k], t);
	}
	__syncthreads();
	if (counts) { if (threadIdx.x < PICK_LEVELS) counts[threadIdx.x] = s_cnt[threadIdx.x]; return; } // (load time: the whole graph's counts, for the host)
	if (threadIdx.x != 0) return;
	int32_t pick = 128 << (PICK_LEVELS - 1);
	for (int k = PICK_LEVELS - 1; k >= 0; k--) { if (s_cnt[k] <= budget) pick = 128 << k; else break; }
	ctl[CTL_COOP] = pick;
}
void launch_pick_coop(const int32_t *part, int32_t nblocks, int32_t budget, int32_t *ctl, hipStream_t st, int32_t *counts) {
	hipLaunchKernelGGL(k_pick_coop, dim3(1), dim3(PICK_THREADS), 0, st, part, nblocks, budget, ctl, counts);
}

void launch_classify(int32_t cnt, const int32_t *outd, const int32_t *coopPtr, int32_t coopMin, int32_t giantMin, int32_t *biglist, int32_t *giantlist, int32_t giantCap, int32_t *ctl, hipStream_t st) {
	if (cnt <= 0) return;
	hipLaunchKernelGGL(k_classify, dim3(nblk(cnt, TPB * CLASSIFY_ITEMS)), dim3(TPB), 0, st, cnt, outd, coopPtr, coopMin, giantMin, biglist, giantlist, giantCap, ctl);
	hipLaunchKernelGGL(k_sort_desc, dim3(2), dim3(1024), 0, st, giantlist, ctl + 1, giantCap, biglist, ctl + 0, cnt, outd, ctl + CTL_GIANT_STARTED);
}

// The giants' groups need a CU each (eight waves at 241 VGPRs, 99 KB of LDS) and find one only while the other parse kernels are not there yet: launched
// in the same microsecond as k_parse_list and the wave class (a range of tens of millions of nodes: the scan of the outdegrees and the parse list end
// together) they waited for those to drain -- 8.2 ms for 0.95 ms of work at 1 B arcs, the last kernel of the parse phase.  One wave in front of each of
// the two other kernels holds its stream until the giants' groups have counted themselves in (or 30 us have passed: then they are not coming soon).
__global__ void __launch_bounds__(64) k_wait_giants(const int32_t *__restrict__ ctl, int32_t groups) {
	if (threadIdx.x) return;
	const int32_t want = min(groups, ctl[1]);
	const uint64_t t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
	while (__hip_atomic_load(ctl + CTL_GIANT_STARTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
		if (__builtin_amdgcn_s_memrealtime() - t0 > 3000) break;
		__builtin_amdgcn_s_sleep(16);
	}
}
void launch_wait_giants(const int32_t *ctl, int giantGroups, hipStream_t st) {
	hipLaunchKernelGGL(k_wait_giants, dim3(1), dim3(64), 0, st, ctl, (int32_t)giantGroups);
}

// (COOPG_DYNLDS builds) the giants' tile is dynamic LDS of more than 64 KB: every instantiation is told so once
static void giant_lds_attr() {
#ifdef COOPG_DYNLDS
	static const bool done = [] {
		const int bytes = (int)GIANT_DYN_LDS;
		(void)hipFuncSetAttribute((const void *)k_parse_big<1, GIANT_NW, RangeView>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
		(void)hipFuncSetAttribute((const void *)k_parse_big<2, GIANT_NW, RangeView>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
		(void)hipFuncSetAttribute((const void *)k_parse_big<0, GIANT_NW, RangeView>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
		(void)hipFuncSetAttribute((const void *)k_parse_big<1, GIANT_NW, BatchView>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
		(void)hipFuncSetAttribute((const void *)k_parse_big<2, GIANT_NW, BatchView>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
		(void)hipFuncSetAttribute((const void *)k_parse_big<0, GIANT_NW, BatchView>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
		return true;
	}();
	(void)done;
#endif
}
void launch_parse_big(const GraphDev &g, int def, const RangeView &v, const int32_t *biglist, const int32_t *giantlist, int32_t *ctl, void *arena, int64_t arenaCap,
                      int waves, int giantGroups, int *err, hipStream_t stGiant, hipStream_t stBig, bool waitGiants) {
	giant_lds_attr();
	if (v.cnt <= 0) return;
	if (giantGroups <= 0) {} // (the caller knows that the job has no giant record)
	else if (def == 1) hipLaunchKernelGGL((k_parse_big<1, GIANT_NW, RangeView>), dim3(giantGroups), dim3(64 * GIANT_NW), GIANT_DYN_LDS, stGiant, g, v, giantlist, ctl, 1, (IvEntry *)arena, arenaCap, err);
	else if (def == 2) hipLaunchKernelGGL((k_parse_big<2, GIANT_NW, RangeView>), dim3(giantGroups), dim3(64 * GIANT_NW), GIANT_DYN_LDS, stGiant, g, v, giantlist, ctl, 1, (IvEntry *)arena, arenaCap, err);
	else hipLaunchKernelGGL((k_parse_big<0, GIANT_NW, RangeView>), dim3(giantGroups), dim3(64 * GIANT_NW), GIANT_DYN_LDS, stGiant, g, v, giantlist, ctl, 1, (IvEntry *)arena, arenaCap, err);
	if (stBig != stGiant && waitGiants && giantGroups > 0) launch_wait_giants(ctl, giantGroups, stBig);
	if (def == 1) hipLaunchKernelGGL((k_parse_big<1, 1, RangeView>), dim3(waves), dim3(64), 0, stBig, g, v, biglist, ctl, 0, (IvEntry *)arena, arenaCap, err);
	else if (def == 2) hipLaunchKernelGGL((k_parse_big<2, 1, RangeView>), dim3(waves), dim3(64), 0, stBig, g, v, biglist, ctl, 0, (IvEntry *)arena, arenaCap, err);
	else hipLaunchKernelGGL((k_parse_big<0, 1, RangeView>), dim3(waves), dim3(64), 0, stBig, g, v, biglist, ctl, 0, (IvEntry *)arena, arenaCap, err);
}

void copy_thresholds(int32_t midMinKnob, bool bigGroups, int32_t &midMin, int32_t &bigMin);
// long records of a random-access batch: same classification, queues and cooperative kernels as a scan, over slots
void launch_bparse_big(const GraphDev &g, int def, const BatchView &v, int32_t coopMin, int32_t giantMin, int32_t *biglist, int32_t *giantlist, int32_t giantCap, int32_t *ctl,
                       void *arena, int64_t arenaCap, int waves, int giantGroups, int *err, hipStream_t st, hipStream_t stGiant, hipStream_t stBig, hipEvent_t evFork, hipEvent_t evGiant, hipEvent_t evBig) {
	giant_lds_attr();
	if (v.cnt <= 0) return;
	launch_classify((int32_t)v.cnt, v.outd, nullptr, coopMin, giantMin, biglist, giantlist, giantCap, ctl, st);
	if (stGiant != st) { (void)hipEventRecord(evFork, st); (void)hipStreamWaitEvent(stGiant, evFork, 0); (void)hipStreamWaitEvent(stBig, evFork, 0); }
	if (def == 1) hipLaunchKernelGGL((k_parse_big<1, GIANT_NW, BatchView>), dim3(giantGroups), dim3(64 * GIANT_NW), GIANT_DYN_LDS, stGiant, g, v, giantlist, ctl, 1, (IvEntry *)arena, arenaCap, err);
	else if (def == 2) hipLaunchKernelGGL((k_parse_big<2, GIANT_NW, BatchView>), dim3(giantGroups), dim3(64 * GIANT_NW), GIANT_DYN_LDS, stGiant, g, v, giantlist, ctl, 1, (IvEntry *)arena, arenaCap, err);
	else hipLaunchKernelGGL((k_parse_big<0, GIANT_NW, BatchView>), dim3(giantGroups), dim3(64 * GIANT_NW), GIANT_DYN_LDS, stGiant, g, v, giantlist, ctl, 1, (IvEntry *)arena, arenaCap, err);
	if (def == 1) hipLaunchKernelGGL((k_parse_big<1, 1, BatchView>), dim3(waves), dim3(64), 0, stBig, g, v, biglist, ctl, 0, (IvEntry *)arena, arenaCap, err);
	else if (def == 2) hipLaunchKernelGGL((k_parse_big<2, 1, BatchView>), dim3(waves), dim3(64), 0, stBig, g, v, biglist, ctl, 0, (IvEntry *)arena, arenaCap, err);
	else hipLaunchKernelGGL((k_parse_big<0, 1, BatchView>), dim3(waves), dim3(64), 0, stBig, g, v, biglist, ctl, 0, (IvEntry *)arena, arenaCap, err);
	if (stGiant != st) { (void)hipEventRecord(evGiant, stGiant); (void)hipEventRecord(evBig, stBig); }
}

void launch_build_lists(const GraphDev &g, const RangeView &v, uint64_t giantBits, int32_t noBin, int32_t *depth, uint16_t *key16, int32_t *hist, int32_t *keyBase, int32_t *cursor,
                        int32_t *list, int32_t *giantlist, int32_t giantCap, int32_t *ctl, int32_t *maxdepth, hipStream_t st,
                        int32_t *bigQ, int32_t bigCap, int32_t *midQ, int32_t midCap, int32_t midMinKnob, bool bigGroups, const uint16_t *packRef) {
	if (v.cnt <= 0) return;
	(void)hipMemsetAsync(hist, 0, sizeof(int32_t) * NKEYS, st);
	int32_t midMin, bigMin;
	copy_thresholds(midMinKnob, bigGroups, midMin, bigMin);
	hipLaunchKernelGGL(k_depth_keys, dim3(nblk(v.cnt, LIST_TILE)), dim3(TPB), 0, st, g, v.lo, v.cnt, v.outd, v.ref, giantBits, noBin, depth, key16, hist, ctl, maxdepth,
	                   bigQ, bigCap, midQ, midCap, midMin, bigMin);
	hipLaunchKernelGGL(k_key_offsets, dim3(1), dim3(TPB), 0, st, hist, keyBase, cursor, maxdepth);
	hipLaunchKernelGGL(k_scatter_keys, dim3(nblk(v.cnt, LIST_TILE)), dim3(TPB), 0, st, v.cnt, key16, cursor, list, giantlist, giantCap, ctl, packRef);
}

void launch_parse_giants(const GraphDev &g, int def, const RangeView &v, const int32_t *giantlist, int32_t *ctl, void *arena, int64_t arenaCap, int giantGroups, int *err, hipStream_t st) {
	giant_lds_attr();
	if (v.cnt <= 0) return;
	if (def == 1) hipLaunchKernelGGL((k_parse_big<1, GIANT_NW, RangeView>), dim3(giantGroups), dim3(64 * GIANT_NW), GIANT_DYN_LDS, st, g, v, giantlist, ctl, 1, (IvEntry *)arena, arenaCap, err);
	else if (def == 2) hipLaunchKernelGGL((k_parse_big<2, GIANT_NW, RangeView>), dim3(giantGroups), dim3(64 * GIANT_NW), GIANT_DYN_LDS, st, g, v, giantlist, ctl, 1, (IvEntry *)arena, arenaCap, err);
	else hipLaunchKernelGGL((k_parse_big<0, GIANT_NW, RangeView>), dim3(giantGroups), dim3(64 * GIANT_NW), GIANT_DYN_LDS, st, g, v, giantlist, ctl, 1, (IvEntry *)arena, arenaCap, err);
}

void launch_parse_waves(const GraphDev &g, int def, const RangeView &v, const int32_t *biglist, int32_t *ctl, void *arena, int64_t arenaCap, int waves, int *err, hipStream_t st) {
	if (v.cnt <= 0) return;
	if (def == 1) hipLaunchKernelGGL((k_parse_big<1, 1, RangeView>), dim3(waves), dim3(64), 0, st, g, v, biglist, ctl, 0, (IvEntry *)arena, arenaCap, err);
	else if (def == 2) hipLaunchKernelGGL((k_parse_big<2, 1, RangeView>), dim3(waves), dim3(64), 0, st, g, v, biglist, ctl, 0, (IvEntry *)arena, arenaCap, err);
	else hipLaunchKernelGGL((k_parse_big<0, 1, RangeView>), dim3(waves), dim3(64), 0, st, g, v, biglist, ctl, 0, (IvEntry *)arena, arenaCap, err);
}

// One chain level of the copy pass: three kernels, one per row class.  With side streams they run next to each
// other (evFork forks, evMid / evBig join back into st); with stMid == stBig == st they run one after the other.
// class thresholds of the copy pass (shared by the queue builder and the level kernels)
void copy_thresholds(int32_t midMinKnob, bool bigGroups, int32_t &midMin, int32_t &bigMin) {
	bigMin = bigGroups ? COPY_BIG_MIN : 0x7fffffff; // !bigGroups: every row is merged by one lane
	midMin = (midMinKnob <= 0 || midMinKnob > bigMin || !bigGroups) ? bigMin : midMinKnob; // = bigMin: no wave-per-row class
}
// walks the block lists of the rows in the group class's queue (all levels); desc: 16 bytes per queue entry
void launch_copy_prewalk(const GraphDev &g, int def, const RangeView &v, const int32_t *bigQ, int32_t bigCap, const int32_t *ctl, void *desc, int blocks, hipStream_t st, int32_t midCap, hipStream_t stLong, bool longKernel, hipStream_t stWalk) {
	if (v.cnt <= 0 || !g.walktab) return;
	const uint32_t longMin = longKernel ? (uint32_t)PREWALK_LONG_MIN : 0xffffffffu;
	// the long lists first, on a stream of their own if the caller has one (ordered behind the queues by the caller)
	// (the wave class's queue follows the group class's, and so do its descriptors)
	if (!longKernel) {}
	else if (def == 1) hipLaunchKernelGGL(k_copy_prewalk_long<1>, dim3(blocks), dim3(64 * PWL_NW), 0, stLong, g, v, bigQ, ctl + 5, bigCap, (int4 *)desc);
	else if (def == 2) hipLaunchKernelGGL(k_copy_prewalk_long<2>, dim3(blocks), dim3(64 * PWL_NW), 0, stLong, g, v, bigQ, ctl + 5, bigCap, (int4 *)desc);
	if (midCap > 0 && def == 1) hipLaunchKernelGGL(k_copy_prewalk_lanes<1>, dim3(blocks), dim3(LW_STRIDE), 0, st, g, v, bigQ + bigCap, ctl + 6, midCap, (int4 *)desc + bigCap);
	else if (midCap > 0 && def == 2) hipLaunchKernelGGL(k_copy_prewalk_lanes<2>, dim3(blocks), dim3(LW_STRIDE), 0, st, g, v, bigQ + bigCap, ctl + 6, midCap, (int4 *)desc + bigCap);
	if (def == 1) hipLaunchKernelGGL(k_copy_prewalk<1>, dim3(blocks), dim3(64 * PREWALK_WAVES), 0, stWalk, g, v, bigQ, ctl + 5, bigCap, (int4 *)desc, longMin);
	else if (def == 2) hipLaunchKernelGGL(k_copy_prewalk<2>, dim3(blocks), dim3(64 * PREWALK_WAVES), 0, stWalk, g, v, bigQ, ctl + 5, bigCap, (int4 *)desc, longMin);
}
void launch_copy_level(const GraphDev &g, int def, const RangeView &v, const int32_t *depth, const int32_t *list, const int32_t *keyBase, int32_t level, int blocks,
                       int32_t midMinKnob, bool bigGroups, const int32_t *bigQ, int32_t bigCap, const int32_t *midQ, int32_t midCap, int32_t *ctl, int32_t *tmp, uint32_t tmpCap, int *err,
                       hipStream_t st, hipStream_t stMid, hipStream_t stBig, hipEvent_t evFork, hipEvent_t evMid, hipEvent_t evBig, const void *preDesc, bool preMid, int listMode,
                       const void *tabArena, int64_t tabArenaCap, const void *copyTab) {
	const bool vecList = (listMode & 1) != 0; // listMode: 1 = 16-byte loads and stores in the lane class's merges, 2 = the table merges as a loop of the whole wave (k_copy_list_w), 16 = k_copy_mid takes the parse's tables for the rows the pre-walk left
	if (v.cnt <= 0) return;
#ifdef BV_EXP_NOCOPY // (ablation builds: the scan without its copy pass, or without one of its three row classes)
	return;
#endif
	blocks = (int)std::min<int64_t>(blocks, nblk(v.cnt, TPB)); // (a thread per row at most: a small range does not launch thousands of idle blocks)
	const int4 *pre = (const int4 *)preDesc;
	int32_t midMin, bigMin;
	copy_thresholds(midMinKnob, bigGroups, midMin, bigMin);
	const bool split = stMid != st || stBig != st;
	// The groups of k_copy_big need a whole CU's LDS and 1 024 thread slots each: launched behind the list kernel they wait for its
	// blocks to drain and end 0.13 ms after it (C2, level 1).  So they go first, on the level's own stream, and the list kernel takes
	// the side stream.
	const hipStream_t stList = bigGroups ? stBig : st;
	if (split) {
		(void)hipEventRecord(evFork, st);
		if (stMid != st) (void)hipStreamWaitEvent(stMid, evFork, 0);
		if (stBig != st) (void)hipStreamWaitEvent(stBig, evFork, 0);
	}
#ifdef BV_EXP_NOCOPY_BIG
	if (false) {
#else
	if (bigGroups) {
#endif
		// ctl[8 + 2 (level & 3)], ctl[9 + 2 (level & 3)]: this level's bump pointer into the scratch tables (the previous level's are free
		// again) and the head of its work queue.  Level l zeroes the pair of level l + 1 (no memset launches between the levels);
		// the job's set-up zeroes all four pairs.
		if (def == 1) hipLaunchKernelGGL(k_copy_big<1>, dim3(COPY_BIG_GRID), dim3(COPY_BIG_THREADS), 0, st, g, v, depth, bigQ, ctl + 5, bigCap, level, tmp, tmpCap, (uint32_t *)(ctl + 8 + 2 * (level & 3)), ctl + 9 + 2 * (level & 3), ctl + 8 + 2 * ((level + 1) & 3), err, pre);
		else if (def == 2) hipLaunchKernelGGL(k_copy_big<2>, dim3(COPY_BIG_GRID), dim3(COPY_BIG_THREADS), 0, st, g, v, depth, bigQ, ctl + 5, bigCap, level, tmp, tmpCap, (uint32_t *)(ctl + 8 + 2 * (level & 3)), ctl + 9 + 2 * (level & 3), ctl + 8 + 2 * ((level + 1) & 3), err, pre);
		else hipLaunchKernelGGL(k_copy_big<0>, dim3(COPY_BIG_GRID), dim3(COPY_BIG_THREADS), 0, st, g, v, depth, bigQ, ctl + 5, bigCap, level, tmp, tmpCap, (uint32_t *)(ctl + 8 + 2 * (level & 3)), ctl + 9 + 2 * (level & 3), ctl + 8 + 2 * ((level + 1) & 3), err, pre);
	}
#ifdef BV_EXP_NOCOPY_MID
	if (false) {
#else
	if (midMin < bigMin) {
#endif
		if (def == 1) hipLaunchKernelGGL(k_copy_mid<1>, dim3(1024), dim3(64 * COPY_MID_WAVES), 0, stMid, g, v, depth, midQ, ctl + 6, midCap, level, err, pre && preMid && midQ == bigQ + bigCap ? pre + bigCap : nullptr, (listMode & 16) ? (const IvEntry *)tabArena : nullptr, tabArenaCap, (listMode & 16) ? (const CopyTab *)copyTab : nullptr);
		else if (def == 2) hipLaunchKernelGGL(k_copy_mid<2>, dim3(1024), dim3(64 * COPY_MID_WAVES), 0, stMid, g, v, depth, midQ, ctl + 6, midCap, level, err, pre && preMid && midQ == bigQ + bigCap ? pre + bigCap : nullptr, (listMode & 16) ? (const IvEntry *)tabArena : nullptr, tabArenaCap, (listMode & 16) ? (const CopyTab *)copyTab : nullptr);
		else hipLaunchKernelGGL(k_copy_mid<0>, dim3(1024), dim3(64 * COPY_MID_WAVES), 0, stMid, g, v, depth, midQ, ctl + 6, midCap, level, err, pre && preMid && midQ == bigQ + bigCap ? pre + bigCap : nullptr, (listMode & 16) ? (const IvEntry *)tabArena : nullptr, tabArenaCap, (listMode & 16) ? (const CopyTab *)copyTab : nullptr);
	}
	if (stMid != st) (void)hipEventRecord(evMid, stMid);
#define COPY_LIST(D, V) do { if (copyTab && D != 0 && (listMode & 2)) hipLaunchKernelGGL((k_copy_list_w<(D != 0 ? D : 1), V>), dim3(blocks), dim3(TPB), 0, stList, g, v, depth, list, keyBase, level, midMin, bigMin, err, (const IvEntry *)tabArena, tabArenaCap, (const CopyTab *)copyTab); \
	else if (copyTab && D != 0) hipLaunchKernelGGL((k_copy_list<D, V, false, (D != 0)>), dim3(blocks), dim3(TPB), 0, stList, g, v, depth, list, keyBase, level, midMin, bigMin, err, (const IvEntry *)tabArena, tabArenaCap, (const CopyTab *)copyTab); \
	else hipLaunchKernelGGL((k_copy_list<D, V>), dim3(blocks), dim3(TPB), 0, stList, g, v, depth, list, keyBase, level, midMin, bigMin, err, (const IvEntry *)nullptr, (int64_t)0, (const CopyTab *)nullptr); } while (0)
	if (v.hx && def == 1) hipLaunchKernelGGL((k_copy_list<1, false, true>), dim3(blocks), dim3(TPB), 0, stList, g, v, depth, list, keyBase, level, midMin, bigMin, err, (const IvEntry *)nullptr, (int64_t)0, (const CopyTab *)nullptr); // (the hash fold: ids added as they are merged)
	else if (v.hx && def == 2) hipLaunchKernelGGL((k_copy_list<2, false, true>), dim3(blocks), dim3(TPB), 0, stList, g, v, depth, list, keyBase, level, midMin, bigMin, err, (const IvEntry *)nullptr, (int64_t)0, (const CopyTab *)nullptr);
	else if (v.hx) hipLaunchKernelGGL((k_copy_list<0, false, true>), dim3(blocks), dim3(TPB), 0, stList, g, v, depth, list, keyBase, level, midMin, bigMin, err, (const IvEntry *)nullptr, (int64_t)0, (const CopyTab *)nullptr);
	else
#ifdef BV_EXP_NOCOPY_LIST
	if (true) {}
	else
#endif
	if (def == 1) { if (vecList) COPY_LIST(1, true); else COPY_LIST(1, false); }
	else if (def == 2) { if (vecList) COPY_LIST(2, true); else COPY_LIST(2, false); }
	else { if (vecList) COPY_LIST(0, true); else COPY_LIST(0, false); }
#undef COPY_LIST
	if (stList != st) (void)hipEventRecord(evBig, stList);
	if (stMid != st) (void)hipStreamWaitEvent(st, evMid, 0);
	if (stList != st) (void)hipStreamWaitEvent(st, evBig, 0);
}

int32_t tile_count(int64_t bitSpan, int32_t cnt) { return (int32_t)std::min<int64_t>((bitSpan + (int64_t)TILE_NODE_BITS * cnt) / TILE_SPAN + 1, 0x7ffffff0); }
void launch_tile_bounds(const GraphDev &g, int32_t lo, int32_t cnt, int32_t ntiles, int32_t *tb, hipStream_t st) {
	hipLaunchKernelGGL(k_tile_bounds, dim3(nblk((int64_t)ntiles + 1, 256)), dim3(256), 0, st, g.offsets, lo, cnt, ntiles, tb);
}
void launch_parse_tile(const GraphDev &g, int def, const RangeView &v, const int32_t *tb, int32_t ntiles, int variant, int *err, hipStream_t st, void *tabArena, int64_t tabArenaCap, void *copyTab) {
	if (v.cnt <= 0 || ntiles <= 0) return;
	const int waveLoop = (variant & 0x100) ? 1 : 0; // one lane per record (bv_tile.hpp), through the wave's loop (parse_node_lwc) or the tile's own reader
	IvEntry *a = g.minInt > 0 ? (IvEntry *)tabArena : nullptr;
	if (def == 1) hipLaunchKernelGGL(k_parse_tile<1>, dim3(ntiles), dim3(TILE_T), 0, st, g, v, tb, err, a, tabArenaCap, (CopyTab *)copyTab, waveLoop);
	else hipLaunchKernelGGL(k_parse_tile<2>, dim3(ntiles), dim3(TILE_T), 0, st, g, v, tb, err, a, tabArenaCap, (CopyTab *)copyTab, waveLoop);
}

void launch_parse_listed(const GraphDev &g, int def, const RangeView &v, const int32_t *list, int32_t *ctl, int which, void *arena, int64_t arenaCap, int waves, int *err, hipStream_t st) {
	if (v.cnt <= 0) return;
	if (def == 1) hipLaunchKernelGGL((k_parse_big<1, 1, RangeView>), dim3(waves), dim3(64), 0, st, g, v, list, ctl, which, (IvEntry *)arena, arenaCap, err);
	else if (def == 2) hipLaunchKernelGGL((k_parse_big<2, 1, RangeView>), dim3(waves), dim3(64), 0, st, g, v, list, ctl, which, (IvEntry *)arena, arenaCap, err);
	else hipLaunchKernelGGL((k_parse_big<0, 1, RangeView>), dim3(waves), dim3(64), 0, st, g, v, list, ctl, which, (IvEntry *)arena, arenaCap, err);
}

void launch_parse_list(const GraphDev &g, int def, const RangeView &v, const int32_t *list, const int32_t *keyBase, int blocks, int *err, hipStream_t st, void *arena, int64_t arenaCap, int32_t keyLo, int32_t keyHi, bool lwc, void *copyTab, bool packedList) {
	CopyTab *ct = (CopyTab *)copyTab;
	const int packed = packedList ? 1 : 0;
	if (v.cnt <= 0) return;
	blocks = (int)std::min<int64_t>(blocks, nblk(v.cnt, TPB)); // (a thread per record at most)
	IvEntry *a = (IvEntry *)arena;
	if (!lwc && def != 0) { // round 4's loop (knob lane_loop = 0): no tables for the copy pass
		if (def == 1 && v.hx) hipLaunchKernelGGL((k_parse_list<1, true, false>), dim3(blocks), dim3(TPB), 0, st, g, v, list, keyBase, keyLo, keyHi, a, arenaCap, err, (CopyTab *)nullptr, packed);
		else if (def == 2 && v.hx) hipLaunchKernelGGL((k_parse_list<2, true, false>), dim3(blocks), dim3(TPB), 0, st, g, v, list, keyBase, keyLo, keyHi, a, arenaCap, err, (CopyTab *)nullptr, packed);
		else if (def == 1) hipLaunchKernelGGL((k_parse_list<1, false, false>), dim3(blocks), dim3(TPB), 0, st, g, v, list, keyBase, keyLo, keyHi, a, arenaCap, err, (CopyTab *)nullptr, packed);
		else hipLaunchKernelGGL((k_parse_list<2, false, false>), dim3(blocks), dim3(TPB), 0, st, g, v, list, keyBase, keyLo, keyHi, a, arenaCap, err, (CopyTab *)nullptr, packed);
		return;
	}
	if (def == 1 && v.hx) hipLaunchKernelGGL((k_parse_list<1, true>), dim3(blocks), dim3(TPB), 0, st, g, v, list, keyBase, keyLo, keyHi, a, arenaCap, err, ct, packed);
	else if (def == 2 && v.hx) hipLaunchKernelGGL((k_parse_list<2, true>), dim3(blocks), dim3(TPB), 0, st, g, v, list, keyBase, keyLo, keyHi, a, arenaCap, err, ct, packed);
	else if (def == 1) hipLaunchKernelGGL(k_parse_list<1>, dim3(blocks), dim3(TPB), 0, st, g, v, list, keyBase, keyLo, keyHi, a, arenaCap, err, ct, packed);
	else if (def == 2) hipLaunchKernelGGL(k_parse_list<2>, dim3(blocks), dim3(TPB), 0, st, g, v, list, keyBase, keyLo, keyHi, a, arenaCap, err, ct, packed);
	else hipLaunchKernelGGL(k_parse_list<0>, dim3(blocks), dim3(TPB), 0, st, g, v, list, keyBase, keyLo, keyHi, a, arenaCap, err, ct, packed);
}

} // namespace bv
