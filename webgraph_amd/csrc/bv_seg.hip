// bv_seg.hip -- kernels of the segment pipeline (bodies: bv_seg.hpp): the residual sections of the records too
// long for one lane, decoded in pieces of SEG_BITS bits of stream by one lane per piece.  Every kernel is a grid-stride loop over a
// count that lives on the device (records of the class, segments): nothing here waits for the host.
#include "bv_device.hpp"
#include "bv_launch.hpp"
#include "bv_coop.hpp"
#include "bv_seg.hpp"

#include <algorithm>

namespace bv {

using namespace bvsg;
static_assert(sizeof(SegIv) == sizeof(IvEntry) && offsetof(SegIv, rank) == offsetof(IvEntry, rank) && offsetof(SegIv, len) == offsetof(IvEntry, len) &&
              offsetof(SegIv, pstart) == offsetof(IvEntry, pstart), "arena entries of the cooperative kernels and of the segment pipeline are the same thing");
static_assert(sizeof(RecDesc) == 32, "RecDesc");

constexpr int STPB = 256;

__device__ __forceinline__ SegGraph seg_graph(const GraphDev &g) { return SegGraph{ g.bits, g.nwords, g.offsets, g.W, g.minInt, g.zetaK }; }

// ------------------------------------------------------------------------------------------------ scans
// exclusive scans in three phases (tile sums, scan of the sums by one block, tile scan + carry); out[n] = total
struct U2 { uint32_t x, y; };
__device__ __forceinline__ U2 operator+(U2 a, U2 b) { return U2{ a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ int32_t sg_shfl_up(int32_t v, int o) { return __shfl_up(v, o, 64); }
__device__ __forceinline__ U2 sg_shfl_up(U2 v, int o) { return U2{ (uint32_t)__shfl_up((int)v.x, o, 64), (uint32_t)__shfl_up((int)v.y, o, 64) }; }
template <class T> __device__ __forceinline__ T sg_zero();
template <> __device__ __forceinline__ int32_t sg_zero<int32_t>() { return 0; }
template <> __device__ __forceinline__ U2 sg_zero<U2>() { return U2{ 0, 0 }; }

constexpr int SS_ITEMS = 4, SS_TILE = STPB * SS_ITEMS;
template <class T> __device__ __forceinline__ T sg_block_excl(T v, T *total, T *wsum /* STPB / 64 */) {
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	T inc = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const T t = sg_shfl_up(inc, o); if (lane >= o) inc = inc + t; }
	if (lane == 63) wsum[wid] = inc;
	__syncthreads();
	T base = sg_zero<T>(), tot = sg_zero<T>();
#pragma unroll
	for (int i = 0; i < STPB / 64; i++) { if (i < wid) base = base + wsum[i]; tot = tot + wsum[i]; }
	__syncthreads();
	*total = tot;
	// exclusive = inclusive of the lane before
	T prev = sg_shfl_up(inc, 1);
	if (lane == 0) prev = sg_zero<T>();
	return base + prev;
}
template <class T> __global__ void __launch_bounds__(STPB) k_sg_scan_sums(const T *__restrict__ in, int64_t n, T *__restrict__ sums, const int32_t *__restrict__ nDev) {
	__shared__ T wsum[STPB / 64];
	if (nDev) n = min(n, (int64_t)*nDev); // (the grid covers the capacity; tiles past the end add nothing)
	const int64_t base = (int64_t)blockIdx.x * SS_TILE;
	T v = sg_zero<T>();
#pragma unroll
	for (int i = 0; i < SS_ITEMS; i++) { const int64_t j = base + (int64_t)threadIdx.x * SS_ITEMS + i; if (j < n) v = v + in[j]; }
	T tot;
	(void)sg_block_excl(v, &tot, wsum);
	if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
template <class T> __global__ void __launch_bounds__(STPB) k_sg_scan_top(T *__restrict__ sums, int64_t nb) {
	__shared__ T wsum[STPB / 64];
	const int64_t per = (nb + STPB - 1) / STPB, lo = min(per * (int64_t)threadIdx.x, nb), hi = min(lo + per, nb);
	T mine = sg_zero<T>();
	for (int64_t j = lo; j < hi; j++) mine = mine + sums[j];
	T tot;
	T run = sg_block_excl(mine, &tot, wsum);
	for (int64_t j = lo; j < hi; j++) { const T x = sums[j]; sums[j] = run; run = run + x; }
}
template <class T> __global__ void __launch_bounds__(STPB) k_sg_scan_apply(const T *__restrict__ in, int64_t n, const T *__restrict__ sums, T *__restrict__ out, const int32_t *__restrict__ nDev) {
	__shared__ T wsum[STPB / 64];
	if (nDev) n = min(n, (int64_t)*nDev);
	if ((int64_t)blockIdx.x * SS_TILE >= n + 1) return;
	const int64_t base = (int64_t)blockIdx.x * SS_TILE;
	T vals[SS_ITEMS];
	T v = sg_zero<T>();
#pragma unroll
	for (int i = 0; i < SS_ITEMS; i++) { const int64_t j = base + (int64_t)threadIdx.x * SS_ITEMS + i; vals[i] = j < n ? in[j] : sg_zero<T>(); v = v + vals[i]; }
	T tot;
	T ex = sg_block_excl(v, &tot, wsum) + sums[blockIdx.x];
#pragma unroll
	for (int i = 0; i < SS_ITEMS; i++) {
		const int64_t j = base + (int64_t)threadIdx.x * SS_ITEMS + i;
		if (j < n) out[j] = ex;
		ex = ex + vals[i];
		if (j == n - 1) out[n] = ex;
	}
	if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = sg_zero<T>();
}
template <class T> static void sg_scan(const T *in, int64_t n, T *out, T *sums, hipStream_t st, const int32_t *nDev = nullptr) {
	const int64_t nb = (n + SS_TILE - 1) / SS_TILE;
	hipLaunchKernelGGL(k_sg_scan_sums<T>, dim3((unsigned)nb), dim3(STPB), 0, st, in, n, sums, nDev);
	hipLaunchKernelGGL(k_sg_scan_top<T>, dim3(1), dim3(STPB), 0, st, sums, nb);
	hipLaunchKernelGGL(k_sg_scan_apply<T>, dim3((unsigned)nb), dim3(STPB), 0, st, in, n, sums, out, nDev);
}

// which record every piece belongs to: a record's lane writes its own stretch; a record of more than 64 pieces (a hub has tens of
// thousands: one lane took 0.23 ms) is written by its wave together
__global__ void __launch_bounds__(STPB) k_seg_fill(int32_t Rcap, const int32_t *__restrict__ segbase, int32_t Scap, int32_t *__restrict__ seg2rec) {
	const int lane = threadIdx.x & 63;
	for (int32_t r0 = blockIdx.x * STPB + (threadIdx.x & ~63); r0 < Rcap; r0 += gridDim.x * STPB) { // (wave-uniform)
		const int32_t r = r0 + lane;
		int32_t a = 0, b = 0;
		if (r < Rcap) { a = segbase[r]; b = min(segbase[r + 1], Scap); }
		const bool isLong = b - a > 64;
		if (!isLong) for (int32_t sg = a; sg < b; sg++) seg2rec[sg] = r;
		unsigned long long lm = __ballot(isLong);
		while (lm) {
			const int src = __ffsll((long long)lm) - 1;
			lm &= lm - 1;
			const int32_t A = __shfl(a, src, 64), B = __shfl(b, src, 64);
			for (int32_t sg = A + lane; sg < B; sg += 64) seg2rec[sg] = r0 + src;
		}
	}
}

// ------------------------------------------------------------------------------------------------ order inside a tile
// A lane kernel lasts, wave by wave, as long as the longest of its 64 pieces -- and pieces differ: a record's first and last piece are
// partial, and a piece holds 30 or 300 codes depending on the gaps of its record (A1 as first written: 37 % of the lanes' iterations did
// work).  So every block takes a tile of TILE_P consecutive pieces and orders them by (expected) number of codes, most first, in LDS
// (a counting sort); thread t then processes entries t, t + 256, ...: the 64 lanes of a wave get pieces of like length.
#ifndef TILE_P_
#define TILE_P_ 256
#endif
constexpr int TILE_P = TILE_P_, TILE_ITEMS = TILE_P / STPB, ORD_KEYS = 512; // (a tile of more than one piece per thread is processed in rounds, one after the other: measured slower)
struct TileOrder { uint16_t ord[TILE_P]; int32_t cur[ORD_KEYS]; };
// key[i] of entry i * STPB + threadIdx.x (< 0: no such piece).  On return o.ord[0 .. n) lists the tile's entries by key, descending.
__device__ __forceinline__ int32_t tile_order(TileOrder &o, const int32_t (&key)[TILE_ITEMS]) {
	for (int b = threadIdx.x; b < ORD_KEYS; b += STPB) o.cur[b] = 0;
	__syncthreads();
#pragma unroll
	for (int i = 0; i < TILE_ITEMS; i++) if (key[i] >= 0) atomicAdd(&o.cur[min(key[i], ORD_KEYS - 1)], 1);
	__syncthreads();
	if (threadIdx.x < 64) { // one wave: exclusive scan over the keys from the highest down (8 keys per lane)
		constexpr int PER = ORD_KEYS / 64;
		const int hi = ORD_KEYS - 1 - threadIdx.x * PER; // this lane owns keys hi, hi - 1, ..., hi - PER + 1
		int32_t mine = 0;
#pragma unroll
		for (int j = 0; j < PER; j++) mine += o.cur[hi - j];
		int32_t inc = mine;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const int32_t t = __shfl_up(inc, d, 64); if ((int)threadIdx.x >= d) inc += t; }
		int32_t run = inc - mine;
#pragma unroll
		for (int j = 0; j < PER; j++) { const int32_t c = o.cur[hi - j]; o.cur[hi - j] = run; run += c; }
	}
	__syncthreads();
	int32_t n = 0;
#pragma unroll
	for (int i = 0; i < TILE_ITEMS; i++) if (key[i] >= 0) { o.ord[atomicAdd(&o.cur[min(key[i], ORD_KEYS - 1)], 1)] = (uint16_t)(i * STPB + threadIdx.x); n = 1; }
	__syncthreads();
	// how many entries the tile has: the cursor of key 0 ends at the total
	return o.cur[0] | (n & 0);
}

// ------------------------------------------------------------------------------------------------ A1
template <int ZK>
__global__ void __launch_bounds__(STPB) k_seg_a1(GraphDev g, RangeView v, const RecDesc *__restrict__ desc, const int32_t *__restrict__ segbase, int32_t Rtot, int32_t Scap,
                                                 const int32_t *__restrict__ seg2rec, SegA1 *__restrict__ a1, int32_t *__restrict__ flag) {
	__shared__ uint32_t lds[WIN_WORDS * STPB];
	__shared__ TileOrder ord;
	const SegGraph sg = seg_graph(g);
	const int32_t S = min(segbase[Rtot], Scap);
	for (int32_t kb = blockIdx.x * TILE_P; kb < S; kb += gridDim.x * TILE_P) {
		int32_t key[TILE_ITEMS];
#pragma unroll
		for (int it = 0; it < TILE_ITEMS; it++) { // codes a piece is expected to hold: its bits times the codes per bit of its record's section
			const int32_t k = kb + it * STPB + threadIdx.x;
			key[it] = -1;
			if (k < S) {
				const int32_t r = seg2rec[k];
				const RecDesc d = desc[r];
				uint64_t cellBit; uint32_t a, b;
				const uint64_t recEnd = (uint64_t)g.offsets[v.lo + d.slot + 1];
				seg_span(d, recEnd, k - segbase[r], cellBit, a, b);
				const uint64_t sec = recEnd - (uint64_t)d.rpos;
				key[it] = (int32_t)(((uint64_t)(b - a) * (uint64_t)d.nres) / (sec ? sec : 1));
			}
		}
		const int32_t n = tile_order(ord, key);
		for (int32_t e = threadIdx.x; e < n; e += STPB) {
			const int32_t k = kb + ord.ord[e];
			const int32_t r = seg2rec[k], i = k - segbase[r];
			const RecDesc d = desc[r];
			const int32_t x = v.lo + d.slot;
			uint64_t cellBit;
			uint32_t a, b;
			seg_span(d, (uint64_t)g.offsets[x + 1], i, cellBit, a, b);
			SegA1 o;
			seg_a1<ZK, STPB>(sg, lds + threadIdx.x, x, cellBit, a, b, i == 0, nullptr, ~0u, o);
			a1[k] = o;
			if (i == 0 && o.badIdx != ~0u) flag[r] = 1;
		}
		__syncthreads(); // (the next tile's order overwrites this one's)
	}
}

// ------------------------------------------------------------------------------------------------ A2
// fin[k]: the piece's true start, count and sum (SegFin); pair[k] = (count, sum) for the scan.  A piece whose chains have not met after
// FIX_CODES codes of the true chain is left to the fix pass (pendlist: block-aggregated, one atomic per block and tile)
template <int ZK>
__global__ void __launch_bounds__(STPB) k_seg_a2(GraphDev g, RangeView v, const RecDesc *__restrict__ desc, const int32_t *__restrict__ segbase, int32_t Rtot, int32_t Scap,
                                                 const int32_t *__restrict__ seg2rec, const SegA1 *__restrict__ a1, SegFin *__restrict__ fin, U2 *__restrict__ pair, uint8_t *__restrict__ miss,
                                                 int32_t *__restrict__ pendlist, int32_t *__restrict__ ctl) {
	__shared__ uint32_t lds[WIN_WORDS * STPB];
	__shared__ int32_t s_n, s_base;
	const SegGraph sg = seg_graph(g);
	const int32_t S = min(segbase[Rtot], Scap);
	for (int32_t kb = blockIdx.x * STPB; kb < S; kb += gridDim.x * STPB) {
		const int32_t k = kb + threadIdx.x;
		if (threadIdx.x == 0) s_n = 0;
		__syncthreads();
		bool pending = false;
		if (k < S) {
			const int32_t r = seg2rec[k], i = k - segbase[r];
			const SegA1 me = a1[k];
			const RecDesc d = desc[r];
			const int32_t x = v.lo + d.slot;
			uint64_t cellBit;
			uint32_t a, b;
			seg_span(d, (uint64_t)g.offsets[x + 1], i, cellBit, a, b);
			SegFin o{ a, me.cnt, me.sum, 0, 0, 0, 0, me.badIdx != ~0u ? 2u : 0u };
			if (i > 0) seg_a2<ZK, STPB>(sg, lds + threadIdx.x, cellBit, a1[k - 1].outRel - SEG_BITS, b, me, nullptr, ~0u, nullptr, false, false, o);
			// (mode 2 is no verdict yet: this piece's start may itself be wrong -- then the fix pass comes by, or B's check of the chain fails)
			pending = (o.mode & 3) == 3;
			miss[k] = 0;
			fin[k] = o;
			pair[k] = U2{ o.cnt, o.sum };
		}
		int32_t mySlot = -1;
		if (pending) mySlot = atomicAdd(&s_n, 1);
		__syncthreads();
		if (threadIdx.x == 0 && s_n) s_base = atomicAdd(&ctl[CTL_SEG + 3], s_n);
		__syncthreads();
		if (pending) pendlist[s_base + mySlot] = k;
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------ fix
// Phase 1, one lane per piece that A2 left open: the true chain followed to the end of the piece.  If it ends where A1's chain did
// (they met after all) nothing else changes; if not, the piece after it took a wrong start: miss[k] = 1 and an entry in the fix list.
template <int ZK>
__global__ void __launch_bounds__(64) k_seg_follow(GraphDev g, RangeView v, const RecDesc *__restrict__ desc, const int32_t *__restrict__ segbase, const int32_t *__restrict__ seg2rec,
                                                   const SegA1 *__restrict__ a1, SegFin *__restrict__ fin, U2 *__restrict__ pair, uint8_t *__restrict__ miss,
                                                   const int32_t *__restrict__ pendlist, int32_t *__restrict__ fixlist, int32_t *__restrict__ ctl) {
	__shared__ uint32_t lds[WIN_WORDS * 64];
	const SegGraph sg = seg_graph(g);
	const int32_t n = ctl[CTL_SEG + 3];
	for (int32_t e = blockIdx.x * 64 + threadIdx.x; e < n; e += gridDim.x * 64) {
		const int32_t k = pendlist[e], r = seg2rec[k];
		const RecDesc d = desc[r];
		const int32_t x = v.lo + d.slot;
		uint64_t cellBit;
		uint32_t a, b;
		seg_span(d, (uint64_t)g.offsets[x + 1], k - segbase[r], cellBit, a, b);
		SegFin o = fin[k];
		Win<64> w;
		w.init(sg, lds + threadIdx.x, cellBit + b);
		uint32_t q = w.seek(cellBit + o.inRel), badIdx;
		const uint32_t qend = q + (b > o.inRel ? b - o.inRel : 0u);
		decode_run<ZK, 64>(sg, w, q, qend, false, 0, nullptr, ~0u, o.cnt, o.sum, badIdx);
		o.tRel = (uint32_t)(w.pos(q) - cellBit);
		o.mode = badIdx != ~0u ? 2u : 1u;
		fin[k] = o;
		pair[k] = U2{ o.cnt, o.sum };
		if (o.mode == 1 && k + 1 != segbase[r + 1] && o.tRel != a1[k].outRel) { miss[k] = 1; fixlist[atomicAdd(&ctl[CTL_SEG + 1], 1)] = k; }
	}
}
// Phase 2, one lane per piece whose true chain ended elsewhere than A1's: the next piece again with its true start, and on along the
// record while chains keep missing each other (or the next piece had missed on its own).  A piece whose predecessor missed too is
// not a start: the lane that began further up comes by.  Best effort (FIX_MAX pieces; two runs may collide): B checks every start.
template <int ZK>
__global__ void __launch_bounds__(64) k_seg_fix(GraphDev g, RangeView v, const RecDesc *__restrict__ desc, const int32_t *__restrict__ segbase, const int32_t *__restrict__ seg2rec,
                                                const SegA1 *__restrict__ a1, SegFin *__restrict__ fin, U2 *__restrict__ pair, const uint8_t *__restrict__ miss,
                                                const int32_t *__restrict__ fixlist, const int32_t *__restrict__ ctl, int32_t *__restrict__ flag) {
	__shared__ uint32_t lds[WIN_WORDS * 64];
	const SegGraph sg = seg_graph(g);
	const int32_t n = ctl[CTL_SEG + 1];
	for (int32_t e = blockIdx.x * 64 + threadIdx.x; e < n; e += gridDim.x * 64) {
		const int32_t k0 = fixlist[e], r = seg2rec[k0];
		if (k0 > segbase[r] && miss[k0 - 1]) continue;
		const RecDesc d = desc[r];
		const int32_t x = v.lo + d.slot, kEnd = segbase[r + 1];
		const uint64_t recEnd = (uint64_t)g.offsets[x + 1];
		uint32_t inRel = fin[k0].tRel - SEG_BITS;
		for (int32_t k = k0 + 1, steps = 0; k < kEnd; k++, steps++) {
			if (steps >= FIX_MAX) { flag[r] = 1; break; }
			uint64_t cellBit;
			uint32_t a, b;
			seg_span(d, recEnd, k - segbase[r], cellBit, a, b);
			const SegA1 me = a1[k];
			SegFin o;
			seg_a2<ZK, 64>(sg, lds + threadIdx.x, cellBit, inRel, b, me, nullptr, ~0u, nullptr, false, true, o);
			fin[k] = o;
			pair[k] = U2{ o.cnt, o.sum };
			if ((o.mode & 3) == 2) break; // (B flags the record)
			if ((o.mode & 3) == 1 && o.tRel != me.outRel) { inRel = o.tRel - SEG_BITS; continue; } // missed again: on to the next piece with the true end
			// A1's end of this piece is a true boundary, and that is what the next piece started from: its own walk was right.  If it missed, its
			// entry in the list stood down when this piece had missed (with the wrong start) before: take it along
			if (k + 1 < kEnd && miss[k] && miss[k + 1]) { inRel = me.outRel - SEG_BITS; continue; }
			break;
		}
	}
}

// ------------------------------------------------------------------------------------------------ B
// One lane per piece, the pieces of a tile ordered by their number of codes.  The proof that every piece starts on a codeword boundary:
// the record's first piece does, and every piece's codes end where the next piece says it starts.
template <int ZK>
__global__ void __launch_bounds__(STPB) k_seg_b(GraphDev g, RangeView v, const RecDesc *__restrict__ desc, const int32_t *__restrict__ segbase, int32_t Rtot, int32_t Scap,
                                                const int32_t *__restrict__ seg2rec, const SegFin *__restrict__ fin, const U2 *__restrict__ pre,
                                                IvEntry *__restrict__ arena, int32_t *__restrict__ flag) {
	__shared__ uint32_t lds[(WIN_WORDS + 2 * RING) * STPB];
	__shared__ TileOrder ord;
	const SegGraph sg = seg_graph(g);
	const int32_t S = min(segbase[Rtot], Scap);
	for (int32_t kb = blockIdx.x * TILE_P; kb < S; kb += gridDim.x * TILE_P) {
		int32_t key[TILE_ITEMS];
#pragma unroll
		for (int it = 0; it < TILE_ITEMS; it++) { const int32_t k = kb + it * STPB + threadIdx.x; key[it] = k < S ? (int32_t)min(fin[k].cnt, 0x7fffu) : -1; }
		const int32_t n = tile_order(ord, key);
		for (int32_t e = threadIdx.x; e < n; e += STPB) {
			const int32_t k = kb + ord.ord[e];
			const int32_t r = seg2rec[k];
			if (flag[r]) continue;
			const int32_t k0 = segbase[r], i = k - k0;
			const RecDesc d = desc[r];
			const int32_t s = d.slot, x = v.lo + s;
			const U2 p0 = pre[k0], p = pre[k];
			const SegFin me = fin[k];
			const bool last = k + 1 == segbase[r + 1];
			bool ok = (me.mode & 3) < 2;
			if (last) ok = ok && p.x + me.cnt - p0.x == (uint32_t)d.nres; // the codes of the section add up to the residuals the header promises
			if (ok) {
				const uint64_t cellBit = (((uint64_t)d.rpos >> SEG_BITS_LOG2) + (uint64_t)i) << SEG_BITS_LOG2;
				const int64_t abase = g.minInt > 0 ? v.rowstart[s] / g.minInt : 0;
				uint32_t endRel;
				ok = seg_b<ZK, STPB>(sg, lds + threadIdx.x, lds + WIN_WORDS * STPB + threadIdx.x, x, cellBit, me.inRel, me.cnt, p.x - p0.x, (int32_t)(p.y - p0.y), i == 0,
				                     v.row(s) + d.copied, v.outd[s] - d.copied, (SegIv *)(arena + abase), d.nIv, endRel);
				if (!last) ok = ok && endRel == fin[k + 1].inRel + SEG_BITS;
			}
			if (!ok) flag[r] = 1;
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------ expand
// The intervals of a record are shared out evenly among the lanes of its segments.  Every lane of a wave takes its k-th interval in
// the same iteration: short ones it writes itself, long ones are written by the whole wave, one after the other.
__global__ void __launch_bounds__(STPB) k_seg_expand(RangeView v, int32_t minInt, const RecDesc *__restrict__ desc, const int32_t *__restrict__ segbase, int32_t Rcap, int32_t Scap,
                                                     const int32_t *__restrict__ seg2rec, const IvEntry *__restrict__ arena, const int32_t *__restrict__ flag) {
	const int32_t S = min(segbase[Rcap], Scap);
	const int32_t G = gridDim.x * STPB;
	for (int32_t k0 = blockIdx.x * STPB + (threadIdx.x & ~63); k0 < S; k0 += G) { // (wave-uniform)
		const int32_t k = k0 + (threadIdx.x & 63);
		int32_t lo = 0, hi = 0, nres = 0, extra = 0;
		int32_t *out = nullptr;
		const IvEntry *iv = nullptr;
		if (k < S) {
			const int32_t r = seg2rec[k];
			if (!flag[r]) {
				const RecDesc d = desc[r];
				const int32_t ns = segbase[r + 1] - segbase[r], i = k - segbase[r];
				lo = (int32_t)((int64_t)d.nIv * i / ns); hi = (int32_t)((int64_t)d.nIv * (i + 1) / ns);
				nres = d.nres; extra = v.outd[d.slot] - d.copied;
				out = v.row(d.slot) + d.copied;
				iv = arena + (minInt > 0 ? v.rowstart[d.slot] / minInt : 0);
			}
		}
		int32_t most = hi - lo;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) most = max(most, __shfl_xor(most, o, 64));
		for (int32_t t = 0; t < most; t++) {
			IvEntry e{ 0, 0, 0, 0 };
			if (lo + t < hi) e = iv[lo + t];
			const bool isLong = e.len > 32;
			if (!isLong && e.len > 0) expand_interval(SegIv{ e.left, e.pstart, e.rank, e.len }, nres, out, extra);
			unsigned long long lm = __ballot(isLong);
			while (lm) {
				const int src = __ffsll((long long)lm) - 1;
				lm &= lm - 1;
				const int32_t L = __shfl(e.left, src, 64), N = __shfl(e.len, src, 64), X = __shfl(extra, src, 64);
				const int64_t P = shfl_i64((int64_t)e.pstart + (e.rank < 0 ? nres : e.rank), src);
				int32_t *O = (int32_t *)shfl_i64((int64_t)(uintptr_t)out, src);
				for (int32_t u = threadIdx.x & 63; u < N; u += 64) if (P + u < (int64_t)X) O[P + u] = (int32_t)((uint32_t)L + (uint32_t)u);
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------ flagged records
// -> the list of the cooperative one-wave kernel (k_parse_big<1> with which = CTL_SEG): it decodes them from scratch, whatever the
// pipeline left in their rows and arena slices
__global__ void __launch_bounds__(STPB) k_seg_collect(int32_t Rtot, int32_t Scap, const RecDesc *__restrict__ desc, const int32_t *__restrict__ nseg, const int32_t *__restrict__ segbase,
                                                      const int32_t *__restrict__ flag, int32_t *__restrict__ fblist, int32_t *__restrict__ ctl) {
	for (int32_t r = blockIdx.x * STPB + threadIdx.x; r < Rtot; r += gridDim.x * STPB) {
		// the records whose residuals were handed over
		if (nseg[r] > 0 && (flag[r] || segbase[r + 1] > Scap)) fblist[atomicAdd(&ctl[CTL_SEG], 1)] = desc[r].slot; // (pieces beyond the scratch: cannot happen while the sizing holds)
	}
}

// ------------------------------------------------------------------------------------------------ launch
int32_t seg_bits_log2() { return SEG_BITS_LOG2; }

// Records of the pipeline: the entries of the giants' queue that hand their residual sections over (their descriptors come from k_parse_big).
namespace {
struct SegPtrs {
	RecDesc *desc; int32_t *nseg, *segbase, *flag, *sumsR, *fblist, *seg2rec, *fixlist, *pendlist;
	SegA1 *a1; SegFin *fin; uint8_t *miss; U2 *pair, *pre, *sumsS;
	int32_t *cells; uint32_t *fixbuf;
	size_t bytes;
};
SegPtrs seg_ptrs(void *scratch, int32_t Rtot, int32_t Scap, uint32_t cap) {
	auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
	char *p = (char *)scratch;
	auto take = [&](size_t bytes) { char *q = p; p += up(bytes); return (void *)q; };
	const size_t nbR = ((size_t)Rtot + SS_TILE - 1) / SS_TILE + 1, nbS = ((size_t)Scap + SS_TILE - 1) / SS_TILE + 1;
	SegPtrs o;
	o.desc = (RecDesc *)take(sizeof(RecDesc) * (size_t)Rtot);
	o.nseg = (int32_t *)take(sizeof(int32_t) * ((size_t)Rtot + 1));
	o.segbase = (int32_t *)take(sizeof(int32_t) * ((size_t)Rtot + 1));
	o.flag = (int32_t *)take(sizeof(int32_t) * ((size_t)Rtot + 1));
	o.sumsR = (int32_t *)take(sizeof(int32_t) * nbR);
	o.fblist = (int32_t *)take(sizeof(int32_t) * (size_t)Rtot);
	o.seg2rec = (int32_t *)take(sizeof(int32_t) * ((size_t)Scap + 1));
	o.fixlist = (int32_t *)take(sizeof(int32_t) * ((size_t)Scap + 1));
	o.pendlist = (int32_t *)take(sizeof(int32_t) * ((size_t)Scap + 1));
	o.a1 = (SegA1 *)take(sizeof(SegA1) * ((size_t)Scap + 1));
	o.fin = (SegFin *)take(sizeof(SegFin) * ((size_t)Scap + 1));
	o.miss = (uint8_t *)take((size_t)Scap + 1);
	o.pair = (U2 *)take(sizeof(U2) * ((size_t)Scap + 1));
	o.pre = (U2 *)take(sizeof(U2) * ((size_t)Scap + 1));
	o.sumsS = (U2 *)take(sizeof(U2) * nbS);
	o.cells = nullptr; o.fixbuf = nullptr; (void)cap;
	o.bytes = (size_t)(p - (char *)scratch);
	return o;
}
}
// the most codes a piece can hold (a zeta_k codeword has at least k bits), rounded up to whole 16-byte stores
uint32_t seg_cell_cap(int zetaK) { return (uint32_t)((SEG_BITS / (uint32_t)(zetaK < 1 ? 1 : zetaK) + 2 + 3) & ~3u); }
size_t seg_scratch_bytes(int32_t Rtot, int32_t Scap, int zetaK) { return seg_ptrs(nullptr, Rtot, Scap, seg_cell_cap(zetaK)).bytes; }

// Sizing at load time: how many records have >= 2 048 bits of work (max(bits, 8 successors): the long bins of the parse list) and how many
// bits they hold -- every piece belongs to one of them.  out[0] += records, out[1] += bits (device memory, zeroed by the caller)
__global__ void __launch_bounds__(STPB) k_seg_sizing(const int64_t *__restrict__ offsets, int32_t lo, int32_t n, const int32_t *__restrict__ outd, const uint16_t *__restrict__ ref, unsigned long long *__restrict__ out) {
	__shared__ unsigned long long s_acc[4], s_oct[2 * SIZING_OCTAVES];
	if (threadIdx.x < 4) s_acc[threadIdx.x] = 0;
	if (threadIdx.x < 2 * SIZING_OCTAVES) s_oct[threadIdx.x] = 0;
	__syncthreads();
	unsigned long long recs = 0, bits = 0, lrows = 0, lids = 0; // (lrows, lids: the rows of the copy pass's lane class -- a reference, fewer than 128 successors -- and their ids)
	int32_t mx = 0;
	for (int64_t s64 = (int64_t)blockIdx.x * STPB + threadIdx.x; s64 < n; s64 += (int64_t)gridDim.x * STPB) { // (n may be 2^31 - 1: an int32 index would step past it and come back negative)
		const int32_t s = (int32_t)s64;
		mx = max(mx, outd[s]);
		const uint64_t b = (uint64_t)(offsets[lo + s + 1] - offsets[lo + s]);
		if (outd[s] > 0 && (b >= 2048 || (uint64_t)outd[s] * 8 >= 2048)) { recs++; bits += b; }
		if (ref && ref[s] != 0 && outd[s] > 0 && outd[s] < 128) { lrows++; lids += (unsigned long long)outd[s]; }
		if (outd[s] >= 128) { // records and arcs per octave of the outdegree, from 2^7 up: what the class thresholds are chosen from (pick_thresholds)
			const int k = min(31 - __clz(outd[s]) - 7, SIZING_OCTAVES - 1);
			atomicAdd(&s_oct[2 * k], 1ull); atomicAdd(&s_oct[2 * k + 1], (unsigned long long)outd[s]);
		}
	}
	for (int o = 32; o > 0; o >>= 1) { recs += __shfl_down(recs, o, 64); bits += __shfl_down(bits, o, 64); lrows += __shfl_down(lrows, o, 64); lids += __shfl_down(lids, o, 64); mx = max(mx, __shfl_down(mx, o, 64)); }
	mx = (threadIdx.x & 63) == 0 ? mx : 0;
	if ((threadIdx.x & 63) == 0) { atomicAdd(&s_acc[0], recs); atomicAdd(&s_acc[1], bits); atomicAdd(&s_acc[2], lrows); atomicAdd(&s_acc[3], lids); }
	__syncthreads();
	if (threadIdx.x < 2 && s_acc[threadIdx.x]) atomicAdd(&out[threadIdx.x], s_acc[threadIdx.x]);
	if (threadIdx.x >= 2 && threadIdx.x < 4 && s_acc[threadIdx.x]) atomicAdd(&out[threadIdx.x + 1], s_acc[threadIdx.x]); // out[3], out[4]
	if (mx) atomicMax(&out[2], (unsigned long long)mx); // (the longest record: out[2])
	if (threadIdx.x < 2 * SIZING_OCTAVES && s_oct[threadIdx.x]) atomicAdd(&out[8 + threadIdx.x], s_oct[threadIdx.x]);
}
void launch_seg_sizing(const int64_t *offsets, int32_t lo, int32_t n, const int32_t *outd, const uint16_t *ref, unsigned long long *out5, hipStream_t st) {
	if (n > 0) hipLaunchKernelGGL(k_seg_sizing, dim3(1024), dim3(STPB), 0, st, offsets, lo, n, outd, ref, out5);
}

// the hand-over slots of the cooperative kernels (GraphDev::segDesc ...); the counts of their parts are zeroed on `st`
void seg_handover(GraphDev &g, void *scratch, int32_t capGiant, int32_t Scap, int32_t minD, hipStream_t st) {
	g.segMinD = minD;
	const SegPtrs P = seg_ptrs(scratch, capGiant, Scap, seg_cell_cap(g.zetaK));
	g.segDesc = P.desc; g.segNseg = P.nseg; g.segFlag = P.flag;
	g.segOff[0] = 0; g.segCap[0] = 0; // (the wave class keeps its residuals)
	g.segOff[1] = 0; g.segCap[1] = capGiant;
	if (capGiant > 0) (void)hipMemsetAsync(P.nseg, 0, sizeof(int32_t) * (size_t)capGiant, st);
}

// everything behind the descriptors of the cooperative kernel
void launch_seg_chain(const GraphDev &g, int def, const RangeView &v, int32_t Rtot, int32_t Scap, void *scratch, void *arena, int64_t arenaCap, int32_t *ctl, int blocks, int *err, hipStream_t st) {
	if (v.cnt <= 0 || Rtot <= 0 || def == 0) return;
	(void)hipMemsetAsync(ctl + CTL_SEG, 0, 4 * sizeof(int32_t), st); // the counters of this job's list of flagged records
	const uint32_t cap = seg_cell_cap(g.zetaK);
	const SegPtrs P = seg_ptrs(scratch, Rtot, Scap, cap);
	const dim3 grid((unsigned)blocks), blk(STPB);
	IvEntry *a = (IvEntry *)arena;
	GraphDev g0 = g; g0.segDesc = nullptr; // (the kernel of the flagged records decodes whole records)
	sg_scan<int32_t>(P.nseg, Rtot, P.segbase, P.sumsR, st);
	hipLaunchKernelGGL(k_seg_fill, grid, blk, 0, st, Rtot, P.segbase, Scap, P.seg2rec);
	if (def == 1) hipLaunchKernelGGL(k_seg_a1<3>, grid, blk, 0, st, g0, v, P.desc, P.segbase, Rtot, Scap, P.seg2rec, P.a1, P.flag);
	else hipLaunchKernelGGL(k_seg_a1<0>, grid, blk, 0, st, g0, v, P.desc, P.segbase, Rtot, Scap, P.seg2rec, P.a1, P.flag);
	if (def == 1) hipLaunchKernelGGL(k_seg_a2<3>, grid, blk, 0, st, g0, v, P.desc, P.segbase, Rtot, Scap, P.seg2rec, P.a1, P.fin, P.pair, P.miss, P.pendlist, ctl);
	else hipLaunchKernelGGL(k_seg_a2<0>, grid, blk, 0, st, g0, v, P.desc, P.segbase, Rtot, Scap, P.seg2rec, P.a1, P.fin, P.pair, P.miss, P.pendlist, ctl);
	if (def == 1) hipLaunchKernelGGL(k_seg_follow<3>, dim3(1024), dim3(64), 0, st, g0, v, P.desc, P.segbase, P.seg2rec, P.a1, P.fin, P.pair, P.miss, P.pendlist, P.fixlist, ctl);
	else hipLaunchKernelGGL(k_seg_follow<0>, dim3(1024), dim3(64), 0, st, g0, v, P.desc, P.segbase, P.seg2rec, P.a1, P.fin, P.pair, P.miss, P.pendlist, P.fixlist, ctl);
	if (def == 1) hipLaunchKernelGGL(k_seg_fix<3>, dim3(64), dim3(64), 0, st, g0, v, P.desc, P.segbase, P.seg2rec, P.a1, P.fin, P.pair, P.miss, P.fixlist, ctl, P.flag);
	else hipLaunchKernelGGL(k_seg_fix<0>, dim3(64), dim3(64), 0, st, g0, v, P.desc, P.segbase, P.seg2rec, P.a1, P.fin, P.pair, P.miss, P.fixlist, ctl, P.flag);
	sg_scan<U2>(P.pair, Scap, P.pre, P.sumsS, st, P.segbase + Rtot);
	if (def == 1) hipLaunchKernelGGL(k_seg_b<3>, grid, blk, 0, st, g0, v, P.desc, P.segbase, Rtot, Scap, P.seg2rec, P.fin, P.pre, a, P.flag);
	else hipLaunchKernelGGL(k_seg_b<0>, grid, blk, 0, st, g0, v, P.desc, P.segbase, Rtot, Scap, P.seg2rec, P.fin, P.pre, a, P.flag);
	hipLaunchKernelGGL(k_seg_expand, grid, blk, 0, st, v, g.minInt, P.desc, P.segbase, Rtot, Scap, P.seg2rec, a, P.flag);
	hipLaunchKernelGGL(k_seg_collect, grid, blk, 0, st, Rtot, Scap, P.desc, P.nseg, P.segbase, P.flag, P.fblist, ctl);
	launch_parse_listed(g0, def, v, P.fblist, ctl, CTL_SEG, arena, arenaCap, 256, err, st);
}

} // namespace bv
