// bv_tile.hpp -- the records of a CONTIGUOUS tile of nodes decoded by one work-group from one LDS image of their bits (gfx950).
//
// Consecutive nodes are consecutive in the bit stream and in the CSR.  A work-group owns the nodes whose records START in
// one slice of the stream (TILE_SPAN bits, counting 32 bits per node so that a run of empty nodes cannot make a tile of a
// million rows): it loads the slice ONCE, with coalesced 16-byte loads, byte-swaps it into LDS, and every lane then decodes
// whole records from that shared image with the stateless short-code decoders of bv_coop.hpp -- no per-lane stream
// windows, no refills, no global load inside the decode loop (gfx950 counts loads and stores in one counter: a lane that
// loads while it has stores in flight waits for all of them).  The rows of a tile are neighbours in the CSR, so the 16-byte
// stores of its lanes fill whole cache lines in the L2 before they leave for HBM.
//
// Balance inside the tile: its records are sorted by work (max(bits, 8 * outdegree), half-octave bins, counting sort in
// LDS) and handed to the lanes longest first, so that the 64 lanes of a wave hold records of similar length.
//
// Record grammar and semantics: BVG:1032-1133; same contract as parse_node / parse_node_lw: the
// record's extras (intervals merged with residuals, IntIntervalSequenceIterator + ResidualIntIterator under a
// MergedIntIterator, BVG:1103-1110) go to the TAIL of its CSR row, row[copied..d).  Default codings only (DEF 1 / 2).
#pragma once
#include "bv_coop.hpp"
#include "bv_lanewin.hpp"
#include "bv_launch.hpp"

namespace bv {

constexpr int TILE_T = 256;                    // threads per tile
constexpr int TILE_SPAN = 1 << 16;             // tile weight: bits of the records that start in it + TILE_NODE_BITS per node
constexpr int TILE_NODE_BITS = 32;             // => at most TILE_SPAN / 32 = 2048 nodes per tile
constexpr int TILE_NODES = TILE_SPAN / TILE_NODE_BITS;
constexpr int TILE_WIN_WORDS = TILE_SPAN / 32 + 256; // staged words: the slice, 1 KB of overhang for the last record, look-ahead
constexpr int TILE_NBIN = 32;

struct TWin {
	const lds_u32 *win; // staged words, byte-swapped (first stream bit = bit 31)
	uint32_t nw;        // staged words
	uint64_t w0;        // absolute index of win[0] (multiple of 4)
	const uint32_t *gbits;
	uint64_t gnw;
};
// a word the tile did not stage (the tail of a record that overhangs the slice by more than the slack): straight from HBM
__device__ __attribute__((noinline)) uint32_t tw_word_slow(const uint32_t *gbits, uint64_t gnw, uint64_t j) { return j < gnw ? __builtin_bswap32(gbits[j]) : 0u; }
__device__ __forceinline__ uint32_t tw_word(const TWin &t, uint32_t k) { // k: word index relative to w0
	if (__builtin_expect(k < t.nw, 1)) return t.win[k];
	return tw_word_slow(t.gbits, t.gnw, t.w0 + k);
}

// One record's cursor: k0 = its first word (relative to the window), q = bit offset from that word.
struct TCur {
	uint32_t k0, q;
	__device__ __forceinline__ uint64_t abs_pos(const TWin &t) const { return ((t.w0 + k0) << 5) + q; }
	__device__ __forceinline__ void set_abs(const TWin &t, uint64_t p) { q = (uint32_t)(p - ((t.w0 + k0) << 5)); }
	// KIND 0: zeta_k (ZK = 3 folded in, 0: zk at run time), 1: gamma, 2: unary
	template <int KIND, int ZK> __device__ __forceinline__ uint64_t code(const TWin &t, uint32_t zk, int &err) {
		const uint32_t j = k0 + (q >> 5), sh = q & 31u;
		const uint32_t a = tw_word(t, j), b = tw_word(t, j + 1);
		const uint64_t ab = ((uint64_t)a << 32) | b;
		const uint32_t W = (uint32_t)((ab << sh) >> 32);
		uint32_t v, len;
		if (KIND == 2) {
			if (__builtin_expect(W != 0, 1)) { const uint32_t z = (uint32_t)__clz((int)W); q += z + 1; return z; }
		} else if (__builtin_expect(KIND == 1 ? fast_gamma32(W, v, len) : fast_zeta_32<ZK>(W, zk, v, len), 1)) { q += len; return v; }
		const uint32_t c = tw_word(t, j + 2);
		const uint64_t W64 = sh ? (ab << sh) | ((uint64_t)c >> (32u - sh)) : ab;
		uint64_t v64;
		if (KIND == 2) {
			if (W64) { const uint32_t z = (uint32_t)__clzll((long long)W64); q += z + 1; return z; }
		} else if (KIND == 1 ? fast_gamma(W64, v64, len) : fast_zeta<ZK>(W64, zk, v64, len)) { q += len; return v64; }
		const SlowAbs sa = (KIND == 0 && ZK != 3) ? lane_zeta_slow(t.gbits, t.gnw, abs_pos(t), (int)zk) : lane_code_slow<KIND>(t.gbits, t.gnw, abs_pos(t));
		err |= sa.err;
		set_abs(t, sa.pos);
		return sa.v;
	}
};

// ---- the straight-line reader ------------------------------------------------------------------------------------------
// TCur::code above checks every word against the window and falls through three decoder tiers: ~40 branches per merged
// successor, and on gfx950 a wave pays for every one of them (measured: ~2 000 cycles per code).  TFast::code is for code
// that KNOWS its words are staged (the caller checks the record once): two LDS words, one branch-free 32-bit decode of
// the short codes that make up almost all of a record (gamma < 2^16, zeta_3 < 2^21, unary < 32), ONE branch to an
// out-of-line function for everything else.  q = bit offset from the first staged word.
struct SlowRel { uint64_t v; uint32_t q; int err; };
template <int KIND, int ZK>
__device__ __attribute__((noinline)) SlowRel tfast_slow(const uint32_t *gbits, uint64_t gnw, uint64_t w0, uint32_t q, uint32_t zk) {
	const uint64_t base = w0 << 5;
	const SlowAbs sa = (KIND == 0 && ZK != 3) ? lane_zeta_slow(gbits, gnw, base + q, (int)zk) : lane_code_slow<KIND>(gbits, gnw, base + q);
	return SlowRel{ sa.v, (uint32_t)min<uint64_t>(sa.pos - base, 0x7fffff00ull), sa.err };
}
struct TFast {
	uint32_t q;
	// KIND 0: zeta_k (ZK = 3 folded in, 0: zk at run time), 1: gamma, 2: unary
	template <int KIND, int ZK> __device__ __forceinline__ uint64_t code(const TWin &t, uint32_t zk, int &err) {
		const uint32_t j = q >> 5, sh = q & 31u;
		const uint32_t a = t.win[j], b = t.win[j + 1];
		const uint32_t W = (uint32_t)(((((uint64_t)a << 32) | b) << sh) >> 32);
		const uint32_t h = (uint32_t)__clz((int)W); // 32 for W == 0
		uint32_t v, len;
		bool ok;
		if (KIND == 2) { ok = W != 0; v = h; len = h + 1; }
		else if (KIND == 1) { ok = h < 16; len = 2 * h + 1; v = (W >> ((31u - 2 * h) & 31u)) - 1; }
		else {
			const uint32_t k = ZK == 3 ? 3u : zk;
			const uint32_t nb = k * h + k - 1;                    // payload bits of the short codeword
			ok = ZK == 3 ? h < 7 : (h + 2 + nb <= 32u && nb != 0); // (zeta_1 with h = 0 has no payload: left to the slow path)
			const uint32_t mm = (W << ((h + 1) & 31u)) >> ((31u - nb) & 31u); // the nb payload bits plus the extra bit of the long codeword
			const uint32_t m = mm >> 1, left = 1u << ((k * h) & 31u);
			const bool lng = m >= left;
			v = lng ? mm - 1 : m + left - 1;
			len = h + 1 + nb + (lng ? 1u : 0u);
		}
		if (__builtin_expect(ok, 1)) { q += len; return v; }
		const SlowRel sr = tfast_slow<KIND, ZK>(t.gbits, t.gnw, t.w0, q, zk);
		q = sr.q; err |= sr.err;
		return sr.v;
	}
};

// The tile's image as a reader of parse_node_lwc (bv_lanewin.hpp; round 6: the wave's loop -- sentinels, four trips per pass, one store per pass -- for the tile kernel too):
// q counts from the first staged word, a word is one LDS read, nothing is ever refilled; a code the 32-bit decoders cannot take goes through TCur (checked words, all tiers).
// Only for records whose bits AND the decoders' look-ahead are staged (k_parse_tile checks once per record); the others keep parse_node_tile.
struct TileRd {
	const TWin *tw;
	uint32_t q;
	__device__ __forceinline__ uint32_t word(uint32_t j) const { return tw->win[j]; }
	__device__ __forceinline__ bool low(uint32_t) const { return false; }
	template <int MARGIN> __device__ __forceinline__ void wave_refill(const GraphDev &) {}
	template <int KIND, int ZK = 3> __device__ __forceinline__ uint64_t code(const GraphDev &g, int &err) {
		TCur c;
		c.k0 = 0; c.q = q;
		const uint64_t v = c.template code<KIND, ZK>(*tw, ZK == 3 ? 3u : (uint32_t)g.zetaK, err);
		q = c.q;
		return v;
	}
};

// Same contract as parse_node_lw, reading the record from the tile's window.
// ctab: the slot's table (null: none) -- a record with a reference leaves its copy blocks there for the lane class of the copy pass, as parse_node_lwc does
// (bv_lanewin.hpp: the format); this reader never touches the arena otherwise.
template <int ZK>
__device__ __forceinline__ void parse_node_tile(const GraphDev &g, const TWin &tw, int32_t x, int32_t d, bool hasRef, int64_t dref, int32_t *__restrict__ row, int *__restrict__ err,
                                                CopyTab *__restrict__ ctab = nullptr, int32_t *__restrict__ ovfEnd = nullptr, int32_t ovfCap = 0) { // ovfEnd / ovfCap: the end of the record's own part of the interval arena, in ints
	const uint64_t p0 = (uint64_t)g.offsets[x];
	TCur br, bi;
	br.k0 = (uint32_t)((p0 >> 5) - tw.w0);
	br.q = (uint32_t)p0 & 31u;
	const uint32_t zk = ZK == 3 ? 3u : (uint32_t)g.zetaK;
	int e = 0;
	(void)br.code<1, ZK>(tw, zk, e);              // outdegree (known from k_headers)
	if (g.W > 0) (void)br.code<2, ZK>(tw, zk, e); // reference
	int64_t copied = 0;
	const bool tab = hasRef && ctab != nullptr;
	bool tabOk = tab && d < LW_TAB_D && dref < 65536;
	uint32_t t0 = 0, t1 = 0, t2 = 0, kept = 0;
	auto push = [&](uint32_t en) {
		if (kept == 0) t0 = en; else if (kept == 1) t1 = en; else if (kept == 2) t2 = en;
		else if ((int32_t)kept - 2 <= ovfCap) ovfEnd[2 - (int32_t)kept] = (int32_t)en; else tabOk = false;
		kept++;
	};
	if (hasRef) { // BVG:1058-1071
		const uint64_t bc = br.code<1, ZK>(tw, zk, e);
		int64_t total = 0;
		if (bc > (uint64_t)dref + 1) e |= E_FORMAT;
		else {
			for (uint64_t b = 0; b < bc; b++) {
				int64_t len;
				if (!block_len_ok(br.code<1, ZK>(tw, zk, e), b == 0, total, dref, len)) { e |= E_FORMAT; break; }
				if (!(b & 1)) { if (tabOk && len) push(((uint32_t)total << 16) | (uint32_t)len); copied += len; }
				total += len;
			}
			if (!(bc & 1) && !e) { if (tabOk && dref > total) push(((uint32_t)total << 16) | (uint32_t)(dref - total)); copied += dref - total; }
		}
	}
	const int64_t extra = (int64_t)d - copied;
	if (extra < 0 || copied < 0) e |= E_FORMAT;
	// (the table's header is written on every path: the copy pass must never read a previous job's)
	if (tab && !BV_TIMING(g, 0x200000)) *(int4 *)ctab = int4{ (int32_t)t2, (int32_t)t1, (int32_t)t0, tabOk && !e ? (int32_t)(((uint32_t)copied << 16) | kept) : (int32_t)CT_NONE };
	if (e) { atomicOr(err, e); return; }
	if (extra == 0) return;

	int64_t nIntervals = 0, intervalArcs = 0;
	bi = br;
	if (g.minInt != 0) { // BVG:1073-1096: skip-parse to find the residual section and the number of residuals
		nIntervals = (int64_t)br.code<1, ZK>(tw, zk, e);
		if (nIntervals > extra) { atomicOr(err, E_FORMAT); return; }
		bi = br;
		for (int64_t i = 0; i < nIntervals; i++) {
			(void)br.code<1, ZK>(tw, zk, e);
			const uint64_t len = br.code<1, ZK>(tw, zk, e);
			if (len > (uint64_t)extra) { e |= E_FORMAT; break; }
			intervalArcs += (int64_t)len + g.minInt;
		}
	}
	const int64_t nRes = extra - intervalArcs;
	if (nRes < 0 || e) { atomicOr(err, E_FORMAT | e); return; }

	// merge(intervals, residuals) -> row[copied ..), in 16-byte stores where the row allows it.  Ids are Java ints:
	// 32-bit wrapping arithmetic throughout (BVG:954, :966, :1084-1093 compute in int).
	int32_t *out = row + copied;
	const int32_t nExtra = (int32_t)extra;
	int32_t k = 0;
	int32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0; // (four ids per store, at out + k - 4 whatever its alignment: gfx950 takes dwordx4 on 4-byte boundaries -- no scalar head, round 6)
	int32_t ivLeft = 0, ivRem = 0, ivPrev = 0;
	int32_t ivTodo = (int32_t)nIntervals;
	bool firstIv = true;
	int32_t resTodo = (int32_t)nRes;
	int32_t resVal = 0;
	if (resTodo) resVal = (int32_t)((int64_t)x + nat2int(br.code<0, ZK>(tw, zk, e))); // BVG:954
	while (k < nExtra) {
		if (ivRem == 0 && ivTodo) { // BVG:1084-1093
			if (firstIv) { ivLeft = (int32_t)((int64_t)x + nat2int(bi.code<1, ZK>(tw, zk, e))); firstIv = false; }
			else ivLeft = ivPrev + (int32_t)bi.code<1, ZK>(tw, zk, e) + 1;
			ivRem = (int32_t)bi.code<1, ZK>(tw, zk, e) + g.minInt;
			ivPrev = ivLeft + ivRem;
			ivTodo--;
		}
		int32_t val;
		if (ivRem && (!resTodo || ivLeft < resVal)) { val = ivLeft; ivLeft++; ivRem--; }
		else if (resTodo) {
			val = resVal;
			if (ivRem && ivLeft == resVal) { ivLeft++; ivRem--; } // equal heads are emitted once (MergedIntIterator.java:69-72)
			if (--resTodo) resVal += (int32_t)br.code<0, ZK>(tw, zk, e) + 1; // BVG:966
		} else val = -1; // malformed: fewer values than the outdegree promises (BVG:1210 would store -1)
		o0 = o1; o1 = o2; o2 = o3; o3 = val; k++;
		if ((k & 3) == 0 && !BV_TIMING(g, 0x100000)) *(i32x4_a4 *)(out + k - 4) = i32x4_a4{ o0, o1, o2, o3 };
	}
	const int32_t on = BV_TIMING(g, 0x100000) ? ((o0 ^ o1 ^ o2 ^ o3) == 0x12345678 ? 1 : 0) : (k & 3);
	if (on == 3) { out[k - 3] = o1; out[k - 2] = o2; out[k - 1] = o3; }
	else if (on == 2) { out[k - 2] = o2; out[k - 1] = o3; }
	else if (on == 1) out[k - 1] = o3;
	if (e) atomicOr(err, e);
}

// tile t = the slots s of the view with  t * TILE_SPAN <= (offsets[lo+s] - offsets[lo]) + TILE_NODE_BITS * s < (t+1) * TILE_SPAN
__global__ void __launch_bounds__(256) k_tile_bounds(const int64_t *__restrict__ offsets, int32_t lo, int32_t cnt, int32_t ntiles, int32_t *__restrict__ tb) {
	const int32_t t = blockIdx.x * 256 + threadIdx.x;
	if (t > ntiles) return;
	const int64_t target = (int64_t)t * TILE_SPAN, base = offsets[lo];
	int32_t a = 0, b = cnt; // first s in [0, cnt) with weight(s) >= target, cnt if none
	while (a < b) {
		const int32_t mid = (int32_t)(((int64_t)a + b) >> 1);
		if ((offsets[lo + mid] - base) + (int64_t)TILE_NODE_BITS * mid < target) a = mid + 1; else b = mid;
	}
	tb[t] = a;
}

#ifndef TILE_WAVE_LOOP // 1: the records of a tile through parse_node_lwc (0: tuning builds without its rings)
#define TILE_WAVE_LOOP 1
#endif
#ifndef TILE_RING_ // intervals a lane's ring holds (a power of two; the lane kernel's is LW_RING = 8: 16 KB per block -- a web-shaped record has two or three)
#define TILE_RING_ 4
#endif
constexpr int TILE_RING = TILE_RING_;
#ifndef TILE_MINWAVES // blocks per CU the compiler is asked to leave registers for
#define TILE_MINWAVES 6
#endif
template <int DEF>
__global__ void __launch_bounds__(TILE_T, TILE_MINWAVES) k_parse_tile(GraphDev g, RangeView v, const int32_t *__restrict__ tb, int *__restrict__ err, IvEntry *__restrict__ arena, int64_t arenaCap, CopyTab *__restrict__ ctab, int waveLoop) { // ctab (null: none), arena: the copy blocks' tables for the copy pass; waveLoop: parse_node_lwc (knob tile_loop)
	__shared__ __attribute__((aligned(16))) uint32_t s_win[TILE_WIN_WORDS];
	__shared__ uint16_t s_list[TILE_NODES];
	__shared__ int32_t s_hist[TILE_NBIN], s_n;
	__shared__ uint32_t s_ring[TILE_WAVE_LOOP ? 2 * TILE_RING * TILE_T : 1]; // the lanes' rings of intervals (parse_node_lwc), [word][lane]
	static_assert(TILE_T == LW_STRIDE, "the ring's layout is the lane kernel's");
	const int tid = threadIdx.x;
	const int32_t a = tb[blockIdx.x], b = tb[blockIdx.x + 1];
	if (a >= b) return;
	// ---- the tile's slice of the stream -> LDS (all loads of a lane in flight together, then the byte-swapped stores)
	const uint64_t p0 = (uint64_t)g.offsets[v.lo + a], p1 = (uint64_t)g.offsets[v.lo + b];
	const uint64_t w0 = (p0 >> 5) & ~(uint64_t)3;
	const uint32_t nw = (uint32_t)min<uint64_t>(TILE_WIN_WORDS, (((p1 + 31) >> 5) - w0 + 3 + 3) & ~(uint64_t)3); // + look-ahead of the short-code decoders
	{
		constexpr int NV = (TILE_WIN_WORDS / 4 + TILE_T - 1) / TILE_T;
		const uint4 *src4 = (const uint4 *)(g.bits + w0);
		const uint64_t lim4 = (g.nwords + 8 - w0) / 4; // the image is followed by >= 8 zero words
		uint4 q4[NV];
#pragma unroll
		for (int k = 0; k < NV; k++) {
			const uint32_t i = (uint32_t)tid + (uint32_t)k * TILE_T;
			q4[k] = (i < nw / 4 && i < lim4) ? src4[i] : uint4{ 0u, 0u, 0u, 0u };
		}
#pragma unroll
		for (int k = 0; k < NV; k++) {
			const uint32_t i = (uint32_t)tid + (uint32_t)k * TILE_T;
			if (i < nw / 4) ((uint4 *)s_win)[i] = uint4{ __builtin_bswap32(q4[k].x), __builtin_bswap32(q4[k].y), __builtin_bswap32(q4[k].z), __builtin_bswap32(q4[k].w) };
		}
	}
	// ---- the records this tile decodes, sorted by work, longest first (counting sort on half-octave bins)
	if (tid < TILE_NBIN) s_hist[tid] = 0;
	__syncthreads();
	constexpr int RPT = TILE_NODES / TILE_T; // rows per thread
	int32_t bin[RPT], pos[RPT];
#pragma unroll
	for (int k = 0; k < RPT; k++) {
		// (a tile holds <= TILE_NODES slots: the offset is compared, not the sum -- for a graph of 2^31 - 1 nodes `a + tid + k * TILE_T` of the LAST tile passes
		// INT32_MAX, the compiler forms the address from the unwrapped sum, and the block read outdegrees -- and then offsets and row starts -- of slots that do not
		// exist: the illegal access of the slow-test shape)
		const int32_t o = tid + k * TILE_T;
		const int32_t s = a + min(o, b - a);
		bin[k] = -1;
		if (o < b - a) {
			const int32_t d = v.outd[s];
			if (d > 0 && d < v.coopmin()) {
				const uint64_t bitsLen = (uint64_t)(g.offsets[v.lo + s + 1] - g.offsets[v.lo + s]);
				const uint64_t work = max(bitsLen, (uint64_t)d * 8);
				const int lg = 63 - __clzll((long long)(work | 1));
				const int h = 2 * lg + (lg > 0 ? (int)((work >> (lg - 1)) & 1) : 0);
				bin[k] = TILE_NBIN - 1 - min(max(h - 8, 0), TILE_NBIN - 1); // bin 0 = the longest
				pos[k] = atomicAdd(&s_hist[bin[k]], 1);
			}
		}
	}
	__syncthreads();
	if (tid < 64) { // exclusive scan of the 32 bins by one wave
		int32_t c = tid < TILE_NBIN ? s_hist[tid] : 0, inc = c;
#pragma unroll
		for (int o = 1; o < TILE_NBIN; o <<= 1) { const int32_t t2 = __shfl_up(inc, o, 64); if (tid >= o) inc += t2; }
		if (tid < TILE_NBIN) s_hist[tid] = inc - c;
		if (tid == TILE_NBIN - 1) s_n = inc;
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < RPT; k++) if (bin[k] >= 0) s_list[s_hist[bin[k]] + pos[k]] = (uint16_t)(tid + k * TILE_T);
	__syncthreads();
	const TWin tw{ (const lds_u32 *)s_win, nw, w0, g.bits, g.nwords };
	const int32_t nList = BV_TIMING(g, 0x40000) ? 0 : s_n; // (0x40000 / 0x80000: timing experiments only -- the tile's stage and sort alone / with the records' metadata)
	for (int32_t idx = tid; idx < nList; idx += TILE_T) {
		const int32_t s = a + (int32_t)s_list[idx];
		const int32_t d = v.outd[s], r = v.ref[s];
		if (!v.fits(s)) { atomicOr(err, s >= v.nh ? E_CAP : E_HALO); continue; }
		int32_t *ovfEnd = nullptr;
		int32_t ovfCap = 0;
		if (ctab && arena && r > 0 && d >= 4) { // (only a row of four copied blocks or more needs the room)
			int64_t abase; int32_t an;
			arena_slice(g.minInt, v.rowstart[s], d, abase, an);
			if (abase >= 0 && abase + an <= arenaCap) { ovfEnd = (int32_t *)(arena + abase + (an - 1)); ovfCap = 4 * (an - 1); }
		}
		if (TILE_WAVE_LOOP && waveLoop) {
			// the wave's loop where the record and the decoders' look-ahead (a trip reads two words at its cursor whether it uses them or not: the cursor never passes the
			// record's end by more than a code) are staged, and the record has its slice of the arena (interval lists of TILE_RING entries and more go through it)
			const uint64_t off0 = (uint64_t)g.offsets[v.lo + s], off1 = (uint64_t)g.offsets[v.lo + s + 1];
			int64_t abase = 0; int32_t an = 0;
			bool okA = true;
			if (g.minInt > 0) { arena_slice(g.minInt, v.rowstart[s], d, abase, an); okA = arena != nullptr && abase >= 0 && abase + an <= arenaCap; }
			const bool staged = ((off1 + 31) >> 5) - w0 + 4 <= (uint64_t)nw;
			if (staged && okA) {
				TileRd br{ &tw, (uint32_t)(record_body(g, off0, d, r) - (w0 << 5)) };
				parse_node_lwc<DEF == 1 ? 3 : 0, false, TILE_RING>(g, br, s_ring + tid, v.lo + s, d, r, r > 0 ? v.outd[s - r] : 0, v.row(s), (int2 *)(arena + abase), g.minInt > 0 ? an - 1 : 0, ctab ? ctab + s : nullptr, err);
				continue;
			}
		}
		if (BV_TIMING(g, 0x80000)) { if (((int64_t)v.outd[r > 0 ? s - r : s] + (int64_t)(uintptr_t)v.row(s) + g.offsets[v.lo + s] + (int64_t)(uintptr_t)ovfEnd) == 0x123456789ll) atomicOr(err, 1); continue; }
		parse_node_tile<DEF == 1 ? 3 : 0>(g, tw, v.lo + s, d, r > 0, r > 0 ? (int64_t)v.outd[s - r] : 0, v.row(s), err, ctab ? ctab + s : nullptr, ovfEnd, ovfCap);
	}
}

} // namespace bv
