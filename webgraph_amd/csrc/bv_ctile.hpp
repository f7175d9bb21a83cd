// bv_ctile.hpp -- the copy pass of a CONTIGUOUS tile of rows resolved in LDS, every chain level in one kernel (gfx950).
//
// A row copies from one of the W rows before it (BVG:1056-1071), so a tile of consecutive rows plus a few rows before
// it holds every list its reference chains need -- the window of BVGraphNodeIterator (BVG:1201-1213) turned into a tile.
// The tile's CSR span is loaded into LDS with coalesced loads (the parse kernels have left each row's extras in the
// tail of the row), the referents that live before the tile are loaded next to it and resolved again here (nobody waits
// for a neighbouring work-group), then the rows are merged level by level of their chains -- one lane per row, the masked
// copy of the referent's list (MaskedIntIterator.java:65-97) merged with the extras (MergedIntIterator.java:50-74),
// forward and in place, all in LDS -- and the span goes back with coalesced stores.  The block lists are read from an
// LDS image of the tile's slice of the bit stream (bv_tile.hpp).
//
// Neighbouring tiles must not run side by side: a tile reads the rows before it as the parse kernels left them while their
// own tile rewrites them in place.  So the kernel runs twice, over the even tiles and then over the odd ones, and a tile
// looks back no further than the tile before it: for an even tile those rows are untouched (ref2 == ref: resolved again
// here), for an odd tile they are either final (ref2 == 0) or untouched.  ref2[] doubles as the references of the rows
// before the tile: a resolved row ends every chain that runs through it.
//
// What the tile cannot resolve stays as the parse kernels left it and keeps its entry in ref2[]: rows of CT_MAXD or more
// successors, rows whose chain runs through such a row or leaves the rows the tile looks at, chains deeper than CT_MAXL.
// The level-wise kernels of bv_kernels.hip finish those, ordered by the depth of what is LEFT of their chains.
#pragma once
#include "bv_tile.hpp"

namespace bv {

#ifndef CT_MAXD_V
#define CT_MAXD_V 128
#endif
constexpr int CT_T = 256;
constexpr int CT_SPAN = 4096;                  // tile weight: successors of the rows that start in it + CT_NODE_W per row
constexpr int CT_NODE_W = 8;
constexpr int CT_ROWS = CT_SPAN / CT_NODE_W;   // rows per tile at most
constexpr int CT_HALO_ROWS = 32;               // rows before the tile that are looked at (the window size must not exceed it)
constexpr int CT_MAXD = CT_MAXD_V;                  // rows of at least this many successors are left to the level-wise kernels
constexpr int CT_HOUT = 1024;                  // LDS words for the rows before the tile
constexpr int CT_OUT = CT_HOUT + CT_SPAN + CT_MAXD;
constexpr int CT_NL = CT_ROWS + CT_HALO_ROWS;
constexpr int CT_WIN = 1536;                   // staged words of the tile's bits
constexpr int CT_MAXL = 8;
constexpr uint32_t CF_INLDS = 1u << 12, CF_NEED = 1u << 13, CF_FINAL = 1u << 14, CF_UNFIT = 1u << 15, CF_LVL_SHIFT = 8, CF_LVL_MASK = 15u << 8;

// tile t = the rows s with  t * CT_SPAN <= rowstart[s] + CT_NODE_W * s < (t+1) * CT_SPAN
__global__ void __launch_bounds__(256) k_ctile_bounds(const int64_t *__restrict__ rowstart, int32_t cnt, int32_t ntiles, int32_t *__restrict__ tb) {
	const int32_t t = blockIdx.x * 256 + threadIdx.x;
	if (t > ntiles) return;
	const int64_t target = (int64_t)t * CT_SPAN;
	int32_t a = 0, b = cnt;
	while (a < b) {
		const int32_t mid = (int32_t)(((int64_t)a + b) >> 1);
		if (rowstart[mid] + (int64_t)CT_NODE_W * mid < target) a = mid + 1; else b = mid;
	}
	tb[t] = a;
}

template <int DEF>
__global__ void __launch_bounds__(CT_T) k_copy_tile(GraphDev g, RangeView v, const int32_t *__restrict__ tb, int32_t parity, uint16_t *ref2, int *__restrict__ err) {
	__shared__ int32_t s_out[CT_OUT];
	__shared__ __attribute__((aligned(16))) uint32_t s_win[CT_WIN];
	__shared__ uint16_t s_ro[CT_NL], s_d[CT_NL];
	__shared__ uint32_t s_fl[CT_NL]; // reference (bits 0-7) | chain level (8-11) | CF_*
	__shared__ int32_t s_maxl, s_hfirst, s_modlo, s_modhi, s_hn, s_hp;
	__shared__ int64_t s_hrs[CT_HALO_ROWS];          // row starts of the needed rows before the tile
	__shared__ uint16_t s_hj[CT_HALO_ROWS];          // ... and their local indices
	__shared__ int32_t s_lcnt[16];                   // rows per chain level, then the start of each level's run in s_llist
	__shared__ uint16_t s_llist[CT_NL];              // the rows to merge, sorted by chain level
	__shared__ uint32_t s_lboff[CT_NL];              // ... and where their records start (bits from the first staged word)
	const int tid = threadIdx.x;
	const int32_t tile = 2 * (int32_t)blockIdx.x + parity;
	const int32_t a = tb[tile], b = tb[tile + 1];
	if (a >= b) return;
	const int32_t hs = max(tile > 0 ? tb[tile - 1] : 0, a - CT_HALO_ROWS), nht = a - hs, nloc = b - hs;
	const int32_t cmax = CT_MAXD;
	const int64_t E0 = v.rowstart[a], hsplit = v.rowstart[v.nh];
	auto gaddr = [&](int64_t e) -> int32_t * { return e < hsplit ? v.halo + e : v.succ + (e - hsplit); }; // element e of the view's rows (halo rows | caller's rows)
	if (tid == 0) { s_maxl = 0; s_hfirst = nht; s_modlo = 0x7fffffff; s_modhi = 0; }
	if (tid < 16) s_lcnt[tid] = 0;
	// ---- 1. what the tile looks at
	constexpr int RPT = (CT_NL + CT_T - 1) / CT_T;
	bool work = false, unfit = false;
#pragma unroll
	for (int k = 0; k < RPT; k++) {
		const int32_t i = tid + k * CT_T;
		if (i < nloc) {
			const int32_t s = hs + i, d = v.outd[s];
			const uint32_t r = i >= nht ? v.ref[s] : ref2[s];
			const bool in = d < cmax;
			s_d[i] = (uint16_t)min(d, 0xffff);
			const int64_t rs = v.rowstart[s];
			const bool fit = d == 0 || v.fits(s);
			s_fl[i] = r | (in ? CF_INLDS : 0u) | ((in && r == 0 && i >= nht) ? CF_FINAL : 0u) | (fit ? 0u : CF_UNFIT);
			if (i >= nht) {
				s_ro[i] = (uint16_t)(in ? CT_HOUT + (int32_t)(rs - E0) : 0);
				if (in && r > 0 && d > 0) work = true;
				if (!fit) unfit = true;
			} else s_hrs[i] = rs;
		}
	}
	if (!__syncthreads_or(work)) return;   // no short row with a reference starts here
	if (__syncthreads_or(unfit)) return;   // rows past the caller's capacity: the level-wise kernels report it
	// ---- 2. chain levels; the rows before the tile that the chains run through
	auto walk = [&](int32_t i) -> uint32_t {
		uint32_t lvl = 0;
		int32_t j = i;
		for (;;) {
			const int32_t r = (int32_t)(s_fl[j] & 0xffu);
			if (r == 0) break;
			if (j < r) return 15u; // the chain leaves the rows the tile looks at
			j -= r;
			if (!(s_fl[j] & CF_INLDS)) return 15u; // a long referent: merged elsewhere, later
			if (j < nht) atomicOr(&s_fl[j], CF_NEED);
			if (++lvl > CT_MAXL) return 15u;
		}
		return lvl;
	};
#pragma unroll
	for (int k = 0; k < RPT; k++) {
		const int32_t i = tid + k * CT_T;
		if (i >= nht && i < nloc && (s_fl[i] & CF_INLDS) && (s_fl[i] & 0xffu) && s_d[i]) {
			const uint32_t lvl = walk(i);
			atomicOr(&s_fl[i], lvl << CF_LVL_SHIFT);
			if (lvl < 15u) atomicMax(&s_maxl, (int32_t)lvl);
		}
	}
	__syncthreads();
	if (tid < nht && (s_fl[tid] & CF_NEED)) {
		atomicMin(&s_hfirst, tid);
		const uint32_t lvl = (s_fl[tid] & 0xffu) ? walk(tid) : 0u; // (its ancestors were marked by the walks above)
		atomicOr(&s_fl[tid], (lvl << CF_LVL_SHIFT) | (lvl == 0 ? CF_FINAL : 0u));
	}
	__syncthreads();
	if (tid == 0) { // LDS rows of the needed rows before the tile (at most CT_HALO_ROWS of them)
		int32_t hp = 0, hn = 0;
		for (int32_t j = s_hfirst; j < nht; j++) {
			const uint32_t fl = s_fl[j];
			if (!(fl & CF_NEED)) continue;
			const int32_t d = s_d[j];
			if ((fl & CF_INLDS) && !(fl & CF_UNFIT) && hp + d <= CT_HOUT) { s_ro[j] = (uint16_t)hp; s_hj[hn++] = (uint16_t)j; hp += d; }
			else s_fl[j] = fl & ~(CF_INLDS | CF_FINAL); // does not fit: what copies from it is left to the level-wise kernels
		}
		s_hn = hn; s_hp = hp;
	}
	// the rows to merge, sorted by chain level: a level is then ONE sweep with a row per lane (in node order a wave would
	// meet the few rows of a level one by one, slot by slot)
	int32_t lpos[RPT];
#pragma unroll
	for (int k = 0; k < RPT; k++) {
		const int32_t i = tid + k * CT_T;
		lpos[k] = -1;
		if (i < nloc) {
			const uint32_t fl = s_fl[i], lvl = (fl & CF_LVL_MASK) >> CF_LVL_SHIFT;
			if (lvl >= 1 && lvl <= (uint32_t)CT_MAXL && (fl & CF_INLDS) && (i >= nht || (fl & CF_NEED))) lpos[k] = atomicAdd(&s_lcnt[lvl], 1);
		}
	}
	__syncthreads();
	if (tid == 0) { int32_t acc = 0; for (int l = 0; l < 16; l++) { const int32_t c = s_lcnt[l]; s_lcnt[l] = acc; acc += c; } }
	__syncthreads();
	// ---- 3. rows and bits -> LDS: every load of a lane is in flight before its first LDS store (one memory round trip)
	const int32_t last = nloc - 1;
	const int32_t span = (int32_t)(v.rowstart[hs + last] - E0) + ((s_fl[last] & CF_INLDS) ? (int32_t)s_d[last] : 0);
	const uint64_t p0 = (uint64_t)g.offsets[v.lo + hs + min(s_hfirst, nht)], p1 = (uint64_t)g.offsets[v.lo + b];
	const uint64_t w0 = (p0 >> 5) & ~(uint64_t)3;
	const uint32_t nw = (uint32_t)min<uint64_t>(CT_WIN, (((p1 + 31) >> 5) - w0 + 3 + 3) & ~(uint64_t)3);
	{
		constexpr int NS = (CT_SPAN + CT_MAXD + CT_T - 1) / CT_T, NH = CT_HOUT / CT_T, NB = (CT_WIN / 4 + CT_T - 1) / CT_T;
		int32_t q[NS], hq[NH];
		uint4 bq[NB];
		int64_t ro[RPT];
#pragma unroll
		for (int k = 0; k < RPT; k++) ro[k] = lpos[k] >= 0 ? g.offsets[v.lo + hs + tid + k * CT_T] : 0;
#pragma unroll
		for (int k = 0; k < NS; k++) { const int32_t e = tid + k * CT_T; q[k] = e < span ? *gaddr(E0 + e) : 0; }
		const int32_t hn = s_hn, hp = s_hp;
#pragma unroll
		for (int k = 0; k < NH; k++) {
			const int32_t h = tid + k * CT_T;
			hq[k] = 0;
			if (h < hp) {
				int32_t c = 0;
				while (c + 1 < hn && (int32_t)s_ro[s_hj[c + 1]] <= h) c++; // the row of LDS word h (at most CT_HALO_ROWS of them)
				const int32_t j = s_hj[c];
				hq[k] = *gaddr(s_hrs[j] + (h - (int32_t)s_ro[j]));
			}
		}
		const uint4 *src4 = (const uint4 *)(g.bits + w0);
		const uint64_t lim4 = (g.nwords + 8 - w0) / 4; // the image is followed by >= 8 zero words
#pragma unroll
		for (int k = 0; k < NB; k++) { const uint32_t i = (uint32_t)tid + (uint32_t)k * CT_T; bq[k] = (i < nw / 4 && i < lim4) ? src4[i] : uint4{ 0u, 0u, 0u, 0u }; }
#pragma unroll
		for (int k = 0; k < RPT; k++) if (lpos[k] >= 0) {
			const int32_t i = tid + k * CT_T, at = s_lcnt[(s_fl[i] & CF_LVL_MASK) >> CF_LVL_SHIFT] + lpos[k];
			s_llist[at] = (uint16_t)i;
			s_lboff[at] = (uint32_t)((uint64_t)ro[k] - (w0 << 5));
		}
#pragma unroll
		for (int k = 0; k < NS; k++) { const int32_t e = tid + k * CT_T; if (e < span) s_out[CT_HOUT + e] = q[k]; }
#pragma unroll
		for (int k = 0; k < NH; k++) { const int32_t h = tid + k * CT_T; if (h < hp) s_out[h] = hq[k]; }
#pragma unroll
		for (int k = 0; k < NB; k++) {
			const uint32_t i = (uint32_t)tid + (uint32_t)k * CT_T;
			if (i < nw / 4) ((uint4 *)s_win)[i] = uint4{ __builtin_bswap32(bq[k].x), __builtin_bswap32(bq[k].y), __builtin_bswap32(bq[k].z), __builtin_bswap32(bq[k].w) };
		}
	}
	__syncthreads();
	const TWin tw{ (const lds_u32 *)s_win, nw, w0, g.bits, g.nwords };
	constexpr int ZK = DEF == 1 ? 3 : 0;
	const uint32_t zk = ZK == 3 ? 3u : (uint32_t)g.zetaK;
	// ---- 4. level by level
	const int32_t maxl = s_maxl;
	for (int32_t l = 1; l <= maxl; l++) {
		for (int32_t idx = s_lcnt[l] + tid; idx < s_lcnt[l + 1]; idx += CT_T) {
			const int32_t i = s_llist[idx];
			const uint32_t fl = s_fl[i];
			if (!(fl & CF_INLDS)) continue; // (a row before the tile that found no room in LDS)
			const int32_t r = (int32_t)(fl & 0xffu), y = i - r;
			if (!(s_fl[y] & CF_FINAL)) continue; // its referent could not be resolved here: neither can it
			const int32_t d = s_d[i], dref = s_d[y];
			const int32_t rowo = s_ro[i], srco = s_ro[y];
			// block list: totals first (BVG:1058-1071), then the merge proper; forward and in place: the write index
			// never overtakes the read index of the extras (k = copied so far + extras so far <= copied + extras so far)
			TCur br;
			br.k0 = s_lboff[idx] >> 5;
			br.q = s_lboff[idx] & 31u;
			int e = 0;
			(void)br.code<1, ZK>(tw, zk, e);
			(void)br.code<2, ZK>(tw, zk, e);
			const uint64_t bc = br.code<1, ZK>(tw, zk, e);
			if (bc > (uint64_t)dref + 1 || e) continue; // flagged by the parse kernel
			const TCur blocks = br;
			int64_t total = 0, copied = 0;
			bool bad = false;
			for (uint64_t bb = 0; bb < bc; bb++) {
				int64_t len;
				if (!block_len_ok(br.code<1, ZK>(tw, zk, e), bb == 0, total, dref, len)) { bad = true; break; }
				total += len;
				if (!(bb & 1)) copied += len;
			}
			if (bad || e) continue;
			if (!(bc & 1)) copied += dref - total;
			if (copied > d) continue;
			br = blocks;
			// ONE flat loop, an output id per iteration: nested loops (blocks x ids of a block x extras in front of an id) would
			// cost a wave the PRODUCT of its lanes' longest trip counts.  State: si = next index in the referent's row, left = ids
			// still to copy from the current block, bb = blocks read so far, jj / ev = next extra.
			int32_t si = 0, left = 0, jj = (int32_t)copied, todo = (int32_t)copied;
			uint32_t bb = 0;
			const uint32_t nb = (uint32_t)bc;
			int32_t ev = jj < d ? s_out[rowo + jj] : 0;
			bool haveC = false;
			int32_t cv = 0;
			for (int32_t kk = 0; kk < d; kk++) {
				if (!haveC && todo > 0) {
					// the next copied id: blocks alternate copy / skip and only the first may be empty, so three reads at most
#pragma unroll 1
					for (int t2 = 0; t2 < 4 && left == 0; t2++) {
						int32_t len;
						if (bb < nb) len = (int32_t)br.code<1, ZK>(tw, zk, e) + (bb ? 1 : 0);
						else len = dref - si; // implicit last block: the rest of the referent
						if (bb & 1) si += len; else left = len;
						bb++;
					}
					if (left > 0 && si < dref) { cv = s_out[srco + si]; si++; left--; todo--; haveC = true; }
					else todo = 0; // (cannot happen after the totals above)
				}
				int32_t val;
				if (haveC && (jj >= d || cv <= ev)) {
					val = cv; haveC = false;
					if (jj < d && ev == cv) { jj++; if (jj < d) ev = s_out[rowo + jj]; } // equal heads emitted once (never in a valid file)
				} else if (jj < d) { val = ev; jj++; if (jj < d) ev = s_out[rowo + jj]; }
				else val = -1; // a malformed duplicate left a gap (BVG:1210 would store -1 too)
				s_out[rowo + kk] = val;
			}
			s_fl[i] = fl | CF_FINAL;
			if (i >= nht) { atomicMin(&s_modlo, rowo - CT_HOUT); atomicMax(&s_modhi, rowo - CT_HOUT + d); ref2[hs + i] = 0; }
		}
		__syncthreads();
	}
	// ---- 5. what changed goes back (rows in between are rewritten with what was loaded)
	const int32_t mlo = s_modlo, mhi = s_modhi;
	if (mlo < mhi) for (int32_t e = mlo + tid; e < mhi; e += CT_T) *gaddr(E0 + e) = s_out[CT_HOUT + e];
}

} // namespace bv
