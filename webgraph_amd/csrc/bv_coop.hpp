// bv_coop.hpp -- cooperative decoding of ONE long BVGraph record by a group of NW wavefronts (gfx950).
//
// Variable-length codes are serial within a record, so a long record cannot simply be split between lanes.
// The group decodes a section of the bit stream in tiles of N x B bits (N = 64*NW lanes): the tile's words are
// staged in LDS with coalesced loads, then lane i speculatively parses the codes that START inside its B-bit
// segment, beginning at a guessed boundary; every lane then adopts its left neighbour's end position as its
// start and re-parses, until no start changes (fixed point).  Lane 0 always starts on a true boundary, so at
// the fixed point every lane does (induction); universal codes re-synchronise within a few codes, so the loop
// usually stops after 2-4 rounds.  Gap sequences then become absolute ids through prefix sums over the group
// (__shfl_up scans inside a wave, LDS across waves) -- the "ballot/prefix" step of the north star.
//
// Record grammar and semantics: BVG:1032-1133.  Layout of the work:
//   phase A  header + copy-block totals      uniform (every lane computes the same scalars)
//   phase I  interval section (gamma pairs)  cooperative -> list entries {left, pstart, rank, len} in a scratch arena
//   phase R  residual section (zeta_k ...)   cooperative -> residual ids written at their final position in the
//                                            row tail; tells each interval how many residuals precede it
//   phase X  interval expansion              cooperative -> interval ids written at their final position
// The result is the same "extras merged into row[copied..d)" that the one-lane parse_node produces
// (IntIntervalSequenceIterator + ResidualIntIterator under a MergedIntIterator, BVG:1103-1110).
#pragma once
#include "bv_device.hpp"
#include "bv_seg.hpp"

namespace bv {

// A tile is N x B bits of stream, one B-bit segment per lane.  B adapts to the average code length of the
// section (known from the record's bit length and its code count) so that a segment holds ~COOP_CODES_PER_SEG
// codes: speculative parses only re-synchronise with the true parse after a handful of codes, and a segment
// that ends before they did costs one more round of the fixed-point loop per lane it delays.
constexpr int COOP_B_MIN = 96, COOP_CODES_PER_SEG = 12;

// Tile geometry (compile-time; -D overrides are for tuning builds).  One wave per "big" record, COOP_GIANT_NW
// waves per "giant" record.  The LDS footprint (tile + staged intervals) bounds the groups resident per CU.
#ifndef COOP1_BMAX
#define COOP1_BMAX 512
#endif
#ifndef COOP1_IVCAP
#define COOP1_IVCAP 512
#endif
#ifndef COOPG_BMAX
#define COOPG_BMAX 1024
#endif
#ifndef COOPG_IVCAP
#define COOPG_IVCAP 4096
#endif
#ifndef COOP_RUNIN_MAX
#define COOP_RUNIN_MAX 256
#endif
#ifndef COOP_RUNIN_SHIFT
#define COOP_RUNIN_SHIFT 1
#endif
#ifndef COOP_GIANT_NW
#define COOP_GIANT_NW 8
#endif
#ifndef COOP1_CK
#define COOP1_CK 20 // gaps a lane of the one-wave decoder keeps from its first parse (a segment holds ~12 codes)
#endif
template <int NW> struct CoopCfg { static constexpr int N = 64 * NW, B_MAX = NW == 1 ? COOP1_BMAX : COOPG_BMAX, IVCAP = NW == 1 ? COOP1_IVCAP : COOPG_IVCAP; };

template <int NW> struct CoopLds { // LDS layout of one group, in 32-bit words
	static constexpr int WIN_WORDS = CoopCfg<NW>::N * CoopCfg<NW>::B_MAX / 32 + 12; // staged tile bits (+ alignment and look-ahead slack), multiple of 4
	static constexpr int XCH_WORDS = 2 * (3 * NW + 8);                               // int64 exchange slots
	static constexpr int OFF_WIN = 0, OFF_IVL = WIN_WORDS, OFF_XCH = ((OFF_IVL + 2 * (CoopCfg<NW>::IVCAP + 1) + 1) & ~1); // staged intervals: left[], pstart[]
	static constexpr int OFF_CACHE = OFF_XCH + XCH_WORDS;                            // one wave per record: the gaps of the lane's segment, [slot][lane]
	static constexpr int CACHE_WORDS = NW == 1 ? COOP1_CK * 64 : 0;
	static constexpr int WORDS = OFF_CACHE + CACHE_WORDS;
};

// the part of CoopLds<1> that a block-list walk needs (coop_block_walk: window, exchange slots, the lanes' cached codes -- no staged intervals): 9.4 KB per wave
struct WalkLds {
	static constexpr int WIN_WORDS = CoopLds<1>::WIN_WORDS, XCH_WORDS = CoopLds<1>::XCH_WORDS;
	static constexpr int OFF_WIN = 0, OFF_XCH = (WIN_WORDS + 1) & ~1, OFF_CACHE = OFF_XCH + XCH_WORDS, WORDS = (OFF_CACHE + COOP1_CK * 64 + 3) & ~3;
};

struct IvEntry { int32_t left; int32_t pstart; int32_t rank; int32_t len; }; // one interval in the scratch arena

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) { return (uint64_t)__shfl_up((long long)v, d, 64); }
__device__ __forceinline__ int64_t shfl_i64(int64_t v, int l) { return (int64_t)__shfl((long long)v, l, 64); }

__device__ __forceinline__ int64_t wave_incl_scan_i64(int64_t v) {
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int64_t t = (int64_t)__shfl_up((long long)v, o, 64);
		if (lane >= o) v += t;
	}
	return v;
}

// Group-wide primitives.  Every member is a collective: all N threads must call it under uniform control flow.
template <int NW> struct Grp {
	static constexpr int N = 64 * NW;
	int64_t *xch; // LDS, NW + 8 slots
	// (a one-wave group may be any wave of its block: its thread index is the lane)
	__device__ __forceinline__ int tid() const { return NW == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x; }
	__device__ __forceinline__ int lane() const { return threadIdx.x & 63; }
	__device__ __forceinline__ int wave() const { return NW == 1 ? 0 : (int)(threadIdx.x >> 6); }
	// LDS hand-off inside the group.  One wave: the LDS executes a wave's DS instructions in issue order, so
	// keeping the program order is enough.  Several waves: a workgroup barrier.
	__device__ __forceinline__ void sync() const {
		if (NW == 1) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }
		else __syncthreads();
	}
	// also orders this group's GLOBAL writes before its later reads (waits for outstanding stores: once per phase)
	__device__ __forceinline__ void sync_global() const {
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		if (NW == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
	}
	__device__ __forceinline__ bool any(bool p) const { return NW == 1 ? (bool)__any(p) : (bool)__syncthreads_or(p); }
	__device__ __forceinline__ int64_t incl_scan(int64_t v, int64_t &total) const {
		const int64_t inc = wave_incl_scan_i64(v);
		if (NW == 1) { total = shfl_i64(inc, 63); return inc; }
		if (lane() == 63) xch[wave()] = inc;
		__syncthreads();
		int64_t base = 0, tot = 0;
#pragma unroll
		for (int i = 0; i < NW; i++) { const int64_t t = xch[i]; if (i < wave()) base += t; tot += t; }
		__syncthreads();
		total = tot;
		return base + inc;
	}
	// two scans for the price of one (the barriers are what a multi-wave scan costs)
	__device__ __forceinline__ void incl_scan2(int64_t a, int64_t b, int64_t &ia, int64_t &ib, int64_t &totA, int64_t &totB) const {
		const int64_t inA = wave_incl_scan_i64(a), inB = wave_incl_scan_i64(b);
		if (NW == 1) { totA = shfl_i64(inA, 63); totB = shfl_i64(inB, 63); ia = inA; ib = inB; return; }
		if (lane() == 63) { xch[wave()] = inA; xch[NW + wave()] = inB; }
		__syncthreads();
		int64_t baseA = 0, baseB = 0, tA = 0, tB = 0;
#pragma unroll
		for (int i = 0; i < NW; i++) { const int64_t x = xch[i], y = xch[NW + i]; if (i < wave()) { baseA += x; baseB += y; } tA += x; tB += y; }
		__syncthreads();
		totA = tA; totB = tB; ia = baseA + inA; ib = baseB + inB;
	}
	// the values a, b held by the LAST thread with p set (some thread must have it set), in two barriers
	__device__ __forceinline__ void last2(bool p, int64_t a, int64_t b, int64_t &oa, int64_t &ob) const {
		const unsigned long long m = __ballot(p);
		const int l = m ? 63 - __clzll((long long)m) : -1;
		if (NW == 1) { oa = shfl_i64(a, l); ob = shfl_i64(b, l); return; }
		if (lane() == (l < 0 ? 0 : l)) { xch[wave()] = l >= 0; xch[NW + wave()] = a; xch[2 * NW + wave()] = b; }
		__syncthreads();
		int w = NW - 1;
		while (w > 0 && xch[w] == 0) w--;
		oa = xch[NW + w]; ob = xch[2 * NW + w];
		__syncthreads();
	}
	// the value held by thread tid-1; thread 0 gets `first`
	__device__ __forceinline__ uint64_t prev(uint64_t e, uint64_t first) const {
		uint64_t v = shfl_up_u64(e, 1);
		if (NW > 1) {
			if (lane() == 63) xch[wave()] = (int64_t)e;
			__syncthreads();
			if (lane() == 0 && wave() > 0) v = (uint64_t)xch[wave() - 1];
			__syncthreads();
		}
		return tid() == 0 ? first : v;
	}
	__device__ __forceinline__ int64_t bcast(int64_t v, int srcTid) const {
		if (NW == 1) return shfl_i64(v, srcTid);
		if (tid() == srcTid) xch[NW] = v;
		__syncthreads();
		const int64_t r = xch[NW];
		__syncthreads();
		return r;
	}
	// largest tid with p set, -1 if none
	__device__ __forceinline__ int last_set(bool p) const {
		const unsigned long long m = __ballot(p);
		int w = m ? wave() * 64 + 63 - __clzll((long long)m) : -1;
		if (NW == 1) return w;
		if (lane() == 0) xch[wave()] = w;
		__syncthreads();
		int r = -1;
#pragma unroll
		for (int i = 0; i < NW; i++) r = max(r, (int)xch[i]);
		__syncthreads();
		return r;
	}
	// smallest tid with p set, N if none
	__device__ __forceinline__ int first_set(bool p) const {
		const unsigned long long m = __ballot(p);
		int w = m ? wave() * 64 + __ffsll((long long)m) - 1 : N;
		if (NW == 1) return w;
		if (lane() == 0) xch[wave()] = w;
		__syncthreads();
		int r = N;
#pragma unroll
		for (int i = 0; i < NW; i++) r = min(r, (int)xch[i]);
		__syncthreads();
		return r;
	}
};

// Segment width: at least ~COOP_CODES_PER_SEG codes (so that speculative parses re-synchronise inside their
// segment), and wide enough to spread the whole section over the group's lanes in one tile when it fits:
// fewer, longer segments need fewer rounds of the fixed-point loop per code.
__device__ __forceinline__ uint32_t coop_pick_B(uint64_t sectionBits, uint64_t codes, uint32_t lanes, uint32_t bmax) {
	const uint64_t avg = (sectionBits + codes - 1) / (codes ? codes : 1);
	uint64_t b = max(avg * COOP_CODES_PER_SEG, (sectionBits + lanes - 1) / lanes);
	b = b < COOP_B_MIN ? COOP_B_MIN : b > bmax ? bmax : b;
	// an ODD number of 32-bit words per segment: lane i starts reading at word i * (B/32), and with an even
	// stride the 64 lanes of a wave would keep hitting the same few LDS banks (16 words -> 4 banks)
	uint32_t w = (uint32_t)((b + 31) >> 5) | 1u;
	if (w * 32 > bmax) w -= 2;
	return w * 32;
}

// Stages the words of the tile [pos0, pos0 + N*B) (+ slack) into LDS, byte-swapped.  All the 16-byte loads of
// a lane are issued before the first LDS write, so the tile costs one HBM round trip, not one per iteration.
template <int NW>
__device__ __forceinline__ WindowSrc stage_tile(const Grp<NW> &G, const GraphDev &g, uint32_t *win, uint64_t pos0, uint32_t B) {
	constexpr int N = Grp<NW>::N;
	constexpr int MAXIT = (CoopLds<NW>::WIN_WORDS / 4 + N - 1) / N; // uint4 loads per lane for the widest tile
	const uint64_t w0 = (pos0 >> 5) & ~(uint64_t)3;                // 16-byte aligned window start
	const uint32_t nw4 = (uint32_t)(N * (B >> 5) + 8 + 3) / 4;     // uint4s to stage (<= WIN_WORDS / 4)
	const uint4 *src4 = (const uint4 *)(g.bits + w0);
	const uint64_t lim4 = (g.nwords + 8 - w0) / 4;                 // the image is followed by >= 8 zero words
	uint4 v[MAXIT];
#pragma unroll
	for (int k = 0; k < MAXIT; k++) {
		const uint32_t i = (uint32_t)G.tid() + (uint32_t)k * N;
		v[k] = (i < nw4 && i < lim4) ? src4[i] : uint4{ 0u, 0u, 0u, 0u };
	}
	G.sync(); // the previous tile's readers are done
#pragma unroll
	for (int k = 0; k < MAXIT; k++) {
		const uint32_t i = (uint32_t)G.tid() + (uint32_t)k * N;
		if (i < nw4) ((uint4 *)win)[i] = uint4{ __builtin_bswap32(v[k].x), __builtin_bswap32(v[k].y), __builtin_bswap32(v[k].z), __builtin_bswap32(v[k].w) };
	}
	G.sync();
	return WindowSrc{ win, w0, nw4 * 4, GlobalSrc{ g.bits, g.nwords } };
}

// ---- stateless decoding straight from the staged window --------------------------------------------------
// With the tile in LDS a code can be decoded from its bit position alone: three LDS words give the 64 bits
// starting at `pos`, and gamma / zeta_3 codes of up to 64 bits (values < 2^31 resp. < 2^45) decode with a
// handful of shifts -- no register-buffered reader state, ~5x fewer instructions than the generic reader.
// Anything longer, or outside the window, takes the generic reader for that one code.
__device__ __forceinline__ bool win_peek64(const WindowSrc &src, uint64_t pos, uint64_t &W) {
	const uint64_t j = (pos >> 5) - src.w0;
	if (j + 2 >= (uint64_t)src.nw) return false;
	const uint32_t a = src.win[j], b = src.win[j + 1], c = src.win[j + 2];
	const uint32_t sh = (uint32_t)pos & 31u;
	const uint64_t ab = ((uint64_t)a << 32) | b;
	W = sh ? (ab << sh) | ((uint64_t)c >> (32u - sh)) : ab;
	return true;
}
__device__ __forceinline__ bool fast_gamma(uint64_t W, uint64_t &v, uint32_t &len) {
	if (W == 0) return false;
	const uint32_t m = (uint32_t)__clzll((long long)W);
	if (m > 31) return false;
	v = ((W << m) >> (63u - m)) - 1;
	len = 2 * m + 1;
	return true;
}
// zeta_k from a 64-bit window: unary h, then h*k + k - 1 bits, one more if that is not a short codeword (SURVEY.md
// App. B).  K = 3 (the default, constants folded in) or 0 = the graph's zetak at run time.  Fails (generic reader)
// when the codeword does not fit the window.
template <int K>
__device__ __forceinline__ bool fast_zeta(uint64_t W, uint32_t krt, uint64_t &v, uint32_t &len) {
	if (W == 0) return false;
	const uint32_t h = (uint32_t)__clzll((long long)W);
	if (K == 3) { // the default: constants folded in
		if (h > 15) return false;
		const uint32_t nb = 3 * h + 2;
		const uint64_t W2 = W << (h + 1);
		const uint64_t m = W2 >> (64u - nb);
		const uint64_t left = (uint64_t)1 << (3 * h);
		if (m < left) { v = m + left - 1; len = h + 1 + nb; }
		else { v = ((m << 1) | ((W2 >> (63u - nb)) & 1)) - 1; len = h + 2 + nb; }
		return true;
	}
	const uint32_t k = krt;
	const uint32_t nb = k * h + k - 1;
	if (nb == 0) { v = 0; len = 1; return true; } // zeta_1, h = 0: the codeword "1" has no payload and means 0
	if (h + 2 + nb > 64u) return false;
	const uint64_t W2 = W << (h + 1);
	const uint64_t m = W2 >> (64u - nb);
	const uint64_t left = (uint64_t)1 << (k * h);
	if (m < left) { v = m + left - 1; len = h + 1 + nb; }
	else { v = ((m << 1) | ((W2 >> (63u - nb)) & 1)) - 1; len = h + 2 + nb; }
	return true;
}
// one code at bit position p of the staged window; advances p.  KIND 0: residual code, KIND 1 / 2: gamma code, KIND 3: delta code
template <int DEF, int KIND>
__device__ __forceinline__ uint64_t win_code(const GraphDev &g, const WindowSrc &src, uint64_t &p, int &err) {
	uint64_t W, v; uint32_t len;
	if (KIND != 3 && (KIND != 0 || DEF) && win_peek64(src, p, W) && (KIND != 0 ? fast_gamma(W, v, len) : fast_zeta<DEF == 1 ? 3 : 0>(W, DEF == 1 ? 3u : (uint32_t)g.zetaK, v, len))) { p += len; return v; }
	WinReader br; br.init_src(src, g.nwords);
	br.seek(p);
	v = KIND == 3 ? br.delta() : KIND != 0 ? br.gamma() : Fields<DEF>::residual(br, g);
	p = br.pos();
	err |= br.err;
	return v;
}

// ---- tile-relative decoding ------------------------------------------------------------------------------
// Inside a tile, positions are bit offsets from the first staged word (`base` = src.w0 * 32): they fit 32 bits,
// and the short codes that make up almost all of a record (gamma < 2^16, zeta_3 < 2^21) decode from two LDS
// words with a dozen 32-bit instructions.  Everything else takes the 64-bit path above.
__device__ __forceinline__ bool fast_gamma32(uint32_t W, uint32_t &v, uint32_t &len) {
	if (W < (1u << 16)) return false; // more than 15 leading zeros: longer than 31 bits
	const uint32_t m = (uint32_t)__clz((int)W);
	len = 2 * m + 1;
	v = (W >> (31u - 2 * m)) - 1;
	return true;
}
// delta: gamma(L) then the L low bits of value + 1 (its top bit is implied); codes of up to 32 bits (values < 2^12 .. 2^13)
__device__ __forceinline__ bool fast_delta32(uint32_t W, uint32_t &v, uint32_t &len) {
	if (W < (1u << 27)) return false; // gamma part longer than 9 bits: L > 30
	const uint32_t m = (uint32_t)__clz((int)W), lg = 2 * m + 1;
	const uint32_t L = (W >> (31u - 2 * m)) - 1;
	if (lg + L > 32u) return false;
	v = L ? ((1u << L) | ((W << lg) >> (32u - L))) - 1 : 0;
	len = lg + L;
	return true;
}
template <int K>
__device__ __forceinline__ bool fast_zeta_32(uint32_t W, uint32_t krt, uint32_t &v, uint32_t &len) {
	if (K == 3) {
		if (W < (1u << 25)) return false; // h > 6: longer than 28 bits
		const uint32_t h = (uint32_t)__clz((int)W);
		const uint32_t nb = 3 * h + 2;
		const uint32_t mm = (W << (h + 1)) >> (31u - nb); // the nb bits of the short codeword plus the extra bit of the long one
		const uint32_t m = mm >> 1, left = 1u << (3 * h);
		const bool lng = m >= left;
		v = lng ? mm - 1 : m + left - 1;
		len = 4 * h + 3 + (lng ? 1u : 0u);
		return true;
	}
	const uint32_t k = krt;
	const uint32_t h = (uint32_t)__clz((int)W); // (32 for W == 0: rejected below)
	const uint32_t nb = k * h + k - 1;
	if (h + 2 + nb > 32u) return false; // longer than the window
	if (nb == 0) { v = 0; len = 1; return true; } // zeta_1, h = 0
	const uint32_t mm = (W << (h + 1)) >> (31u - nb);
	const uint32_t m = mm >> 1, left = 1u << (k * h);
	const bool lng = m >= left;
	v = lng ? mm - 1 : m + left - 1;
	len = h + 1 + nb + (lng ? 1u : 0u);
	return true;
}
// the rare long codeword: kept out of line (and fed by value) so that the hot loops stay small
struct SlowCode { uint64_t v; uint32_t q; int err; };
template <int DEF, int KIND>
__device__ __attribute__((noinline)) SlowCode win_code_slow(const GraphDev *gp, const uint32_t *win, uint64_t w0, uint32_t nw, uint32_t q) {
	const GraphDev &g = *gp;
	const WindowSrc src{ win, w0, nw, GlobalSrc{ g.bits, g.nwords } };
	const uint64_t base = w0 << 5;
	uint64_t p = base + q;
	int err = 0;
	const uint64_t v = win_code<DEF, KIND>(g, src, p, err);
	return SlowCode{ v, (uint32_t)min(p - base, (uint64_t)0x7fffff00u), err };
}
// One code at tile-relative position q, which must lie inside the tile proper (then the two words read here
// are staged: the window extends 8 words past the tile).  Advances q.
template <int DEF, int KIND>
__device__ __forceinline__ uint64_t win_code_rel(const GraphDev &g, const WindowSrc &src, uint32_t &q, int &err) {
	if (KIND != 0 || DEF) {
		const uint32_t j = q >> 5;
		const uint64_t ab = ((uint64_t)src.win[j] << 32) | src.win[j + 1];
		const uint32_t W = (uint32_t)((ab << (q & 31u)) >> 32);
		uint32_t v, len;
		if (__builtin_expect(KIND == 3 ? fast_delta32(W, v, len) : KIND != 0 ? fast_gamma32(W, v, len) : fast_zeta_32<DEF == 1 ? 3 : 0>(W, DEF == 1 ? 3u : (uint32_t)g.zetaK, v, len), 1)) { q += len; return v; }
	}
	const SlowCode sc = win_code_slow<DEF, KIND>(&g, src.win, src.w0, src.nw, q);
	q = sc.q;
	err |= sc.err;
	return sc.v;
}

// One speculative parse of the codes starting in [s, segEnd): end position, count and the sum of the decoded
// contributions.  KIND 0: residual codes (gap+1 each; the first code of the section is the zig-zag value);
// KIND 1: gamma codes, positions only; KIND 2 / 3: gamma / delta codes, the sum of their values (offset gaps, bv_offsets.hip).
template <int DEF, int KIND>
__device__ __forceinline__ void spec_parse(const GraphDev &g, const WindowSrc &src, uint64_t base, uint32_t s, uint32_t segEnd, bool firstOfSection, uint32_t &e, uint32_t &c, int64_t &sum) {
	c = 0; sum = 0;
	uint32_t p = s;
	int err = 0; // a speculative parse may run through garbage: errors only stop it
	if (KIND == 0 && firstOfSection && p < segEnd) { sum = nat2int(win_code_rel<DEF, KIND>(g, src, p, err)); c = 1; }
	while (p < segEnd && !err) {
		const uint64_t v = win_code_rel<DEF, KIND>(g, src, p, err);
		if (KIND == 0) sum += (int64_t)v + 1;
		if (KIND >= 2 && !err) sum += (int64_t)v;
		if (KIND >= 2 && err) break; // (a gap stream ends in zero padding: not a code)
		c++;
	}
	e = err ? 0x7fffff00u : p;
}

// The lane's start moved from `so` to `sn`.  Universal codes re-synchronise after a few codewords, so instead of
// parsing the segment again the old and the new chain of codewords are walked in lock step (always the one that
// is behind) until they meet: from there on they are the same chain, and only the difference of the two
// prefixes is applied to (c, sum).  If they do not meet inside the segment the new chain defines the end.
template <int DEF, int KIND>
__device__ __forceinline__ void spec_resync(const GraphDev &g, const WindowSrc &src, uint64_t base, uint32_t so, uint32_t sn, uint32_t segEnd, uint32_t &e, uint32_t &c, int64_t &sum) {
	uint32_t po = so, pn = sn;
	int32_t dc = 0;
	int64_t ds = 0;
	while (po != pn && min(po, pn) < segEnd) {
		const bool adv = pn < po;
		uint32_t q = adv ? pn : po;
		int err = 0;
		const uint64_t v = win_code_rel<DEF, KIND>(g, src, q, err);
		if (err) q = 0x7fffff00u; // garbage: this chain ends here (as in spec_parse)
		const int64_t w = KIND == 0 ? (int64_t)v + 1 : KIND >= 2 ? (int64_t)v : 0;
		if (KIND >= 2 && err) { if (adv) pn = q; else po = q; continue; } // padding: ends the chain, counts nothing
		if (adv) { pn = q; dc++; ds += w; }
		else { po = q; dc--; ds -= w; }
	}
	c += (uint32_t)dc;
	sum += ds;
	if (pn >= segEnd) e = pn;
}

// Fixed-point iteration over one tile starting at the true code boundary pos0.  On return every lane holds
// the true start `s` (tile-relative) of the first code it owns, the number `c` of codes starting in its
// segment, their contribution sum, and E = end of the tile's last code (absolute, uniform).
template <int DEF, int KIND, int NW>
__device__ __forceinline__ void spec_tile(const Grp<NW> &G, const GraphDev &g, const WindowSrc &src, uint64_t pos0, uint64_t secEnd, uint32_t B, bool firstTile,
                                          int64_t needCodes, uint32_t &s, uint32_t &c, int64_t &sum, uint64_t &E, uint64_t anchor = ~0ull) {
	// anchor: where the segment grid starts (default: at pos0).  A caller whose tiles have FIXED nominal boundaries
	// passes the nominal tile start <= pos0 < anchor + B: lane 0 still starts at the true boundary pos0.
	const int tid = G.tid(), lane = G.lane();
	const uint64_t base = src.w0 << 5;
	const uint32_t p0 = (uint32_t)(pos0 - base);
	const uint32_t a0 = anchor == ~0ull ? p0 : (uint32_t)(anchor - base);
	const uint32_t secEndR = (uint32_t)min(secEnd - base, (uint64_t)0x7fffff00u);
	const uint32_t segEnd = min(a0 + (uint32_t)(tid + 1) * B, secEndR);
	s = tid == 0 ? min(p0, secEndR) : min(a0 + (uint32_t)tid * B, secEndR);
	// Run-in: a lane starts a little BEFORE its segment, so that its parse has usually locked onto the true code
	// boundaries by the time it enters the segment; its start is then the first boundary inside the segment.
	// Two neighbours that both locked on agree on that boundary at once, and the rounds below only repair the few
	// that did not (each repair costs the wave the longest resync distance among its lanes).
	if (COOP_RUNIN_MAX > 0 && tid > 0 && s < secEndR) {
		const uint32_t R = min((uint32_t)COOP_RUNIN_MAX, B >> COOP_RUNIN_SHIFT);
		uint32_t p = s - min(R, s - p0);
		int err = 0;
		while (p < s && !err) (void)win_code_rel<DEF, KIND>(g, src, p, err);
		s = err ? s : min(p, secEndR);
	}
	uint32_t e;
	int rounds = 1;
	unsigned long long tq = (g.stats && NW != 1 && KIND == 0) ? __builtin_readcyclecounter() : 0;
#define QT(slot) do { if (g.stats && NW != 1 && KIND == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g.stats[slot], now_ - tq); tq = now_; } } while (0)
	spec_parse<DEF, KIND>(g, src, base, s, segEnd, firstTile && tid == 0, e, c, sum);
	QT(12);
	// a parse that runs past the section end is wrong anyway; clamping keeps the lanes behind the end quiet
	// instead of handing the overshoot down one lane per round
	e = min(e, secEndR);
	// Fixed point inside one wave, with `waveStart` as lane 0's start: no barrier, only shuffles.
	auto wave_rounds = [&](uint32_t waveStart) {
		for (int round = 0; round < 66; round++) {
			uint32_t ns = (uint32_t)__shfl_up((int)e, 1, 64);
			if (lane == 0) ns = waveStart;
			const bool dirty = ns != s;
			if (dirty) {
				spec_resync<DEF, KIND>(g, src, base, s, ns, segEnd, e, c, sum);
				e = min(e, secEndR);
				s = ns;
			}
			rounds++;
			bool stop = !__any(dirty);
			if (!stop && KIND == 1 && NW == 1) {
				// The section ends after needCodes codes, somewhere inside the tile: lanes past that point
				// parse the NEXT section's bits as gamma codes and need not converge.  The clean prefix of
				// lanes is exact, so it suffices that the lanes before the one reaching needCodes are clean.
				const int firstDirty = __ffsll((long long)__ballot(dirty)) - 1; // >= 1: lane 0 is never dirty
				const int64_t cincl = wave_incl_scan_i64((int64_t)c);
				stop = shfl_i64(cincl, firstDirty - 1) >= needCodes;
			}
			if (stop) break;
		}
	};
	if (NW == 1) wave_rounds(p0);
	else {
		// Several waves: every wave first converges on its own from a guessed start (its nominal segment
		// boundary), then the waves exchange end positions; a wave whose start moved re-converges (only the
		// lanes up to the point where the old and the new parse meet do any work).  Barriers only per exchange.
		uint32_t waveStart = G.wave() == 0 ? p0 : s; // s of lane 0 = nominal boundary
		waveStart = (uint32_t)__shfl((int)waveStart, 0, 64);
		wave_rounds(waveStart);
		QT(13);
		for (int xr = 0; xr < NW + 1; xr++) {
			if (lane == 63) G.xch[G.wave()] = (int64_t)e;
			__syncthreads();
			const uint32_t ns = G.wave() == 0 ? p0 : (uint32_t)G.xch[G.wave() - 1];
			const bool changed = ns != waveStart;
			__syncthreads();
			bool stop = !__syncthreads_or(changed);
			if (!stop && KIND == 1) {
				// waves before the first changed one are final; enough if they already hold needCodes codes
				const int firstChanged = G.first_set(changed) >> 6; // wave index, >= 1
				int64_t tot;
				const int64_t cincl = G.incl_scan((int64_t)c, tot);
				stop = G.bcast(cincl, firstChanged * 64 - 1) >= needCodes;
			}
			if (stop) break;
			if (changed) { waveStart = ns; wave_rounds(waveStart); }
			if (KIND == 0) stat_add(g, 10, 1);
		}
		if (KIND == 0) stat_add(g, 11, 1);
		QT(14);
	}
#undef QT
	stat_add(g, KIND == 0 ? 1 : 3, (unsigned long long)rounds);
	stat_add(g, KIND == 0 ? 0 : 2, 1);
	E = base + (uint64_t)G.bcast((int64_t)e, Grp<NW>::N - 1);
}

// ---------------------------------------------------------------------------------------------- one wave, lean: gamma lists
// A tile of a list of gamma codes (copy blocks, interval pairs) by ONE wave with every codeword decoded once: straight-line
// decoder (two LDS words, branch-free for values < 2^16, one branch to the generic reader otherwise; values saturate at
// 2^32 - 1, which every caller rejects), the raw values of the lane's first COOP1_CK codes kept in LDS ([slot][lane]) for the
// passes that interpret them.  Same fixed point as spec_tile: run-in, then every lane adopts its left neighbour's end until
// the clean prefix of lanes holds the codes the section still needs.
template <int DEF>
__device__ __forceinline__ uint32_t w1_gamma(const GraphDev &g, const lds_u32 *lw, const WindowSrc &src, uint32_t &q, int &err) {
	const uint32_t j = q >> 5, sh = q & 31u;
	const uint32_t a = lw[j], b = lw[j + 1];
	const uint32_t W = (uint32_t)(((((uint64_t)a << 32) | b) << sh) >> 32);
	const uint32_t h = (uint32_t)__clz((int)W); // 32 for W == 0
	if (__builtin_expect(h < 16, 1)) { q += 2 * h + 1; return (W >> ((31u - 2 * h) & 31u)) - 1; }
	const SlowCode sc = win_code_slow<DEF, 1>(&g, src.win, src.w0, src.nw, q);
	q = sc.q; err |= sc.err;
	return (uint32_t)min<uint64_t>(sc.v, 0xffffffffull);
}
struct W1Tile { WindowSrc src; uint32_t s, c, pCK; uint64_t E; }; // start, codes, position behind the cached ones; end of the tile's last code (uniform)
template <int DEF, class L = CoopLds<1>>
__device__ __forceinline__ W1Tile w1_gamma_tile(const Grp<1> &G, const GraphDev &g, uint32_t *lds, uint64_t pos, uint64_t secEnd, uint32_t B, int64_t needCodes) {
	constexpr int CK = COOP1_CK;
	const int lane = threadIdx.x & 63;
	uint32_t *win = lds + L::OFF_WIN;
	const lds_u32 *lw = (const lds_u32 *)win;
	lds_u32 *cache = (lds_u32 *)(lds + L::OFF_CACHE);
	W1Tile t;
	t.src = stage_tile<1>(G, g, win, pos, B);
	const uint64_t base = t.src.w0 << 5;
	const uint32_t p0 = (uint32_t)(pos - base), secEndR = (uint32_t)min(secEnd - base, (uint64_t)0x7fffff00u);
	const uint32_t segEnd = min(p0 + (uint32_t)(lane + 1) * B, secEndR);
	uint32_t s = lane == 0 ? min(p0, secEndR) : min(p0 + (uint32_t)lane * B, secEndR);
	const uint32_t R = min((uint32_t)COOP_RUNIN_MAX, B >> COOP_RUNIN_SHIFT);
	if (lane > 0 && s < secEndR && R) { // run-in: lock onto the code boundaries before the segment starts
		uint32_t p = s - min(R, s - p0);
		int e2 = 0;
		while (p < s && !e2) (void)w1_gamma<DEF>(g, lw, t.src, p, e2);
		s = e2 ? s : min(p, secEndR);
	}
	uint32_t e, c, pCK;
	auto parse = [&]() {
		c = 0;
		uint32_t p = s;
		pCK = s;
		int e2 = 0; // a speculative parse may run through garbage: errors only stop it
		while (p < segEnd && !e2) {
			const uint32_t v = w1_gamma<DEF>(g, lw, t.src, p, e2);
			if (c < (uint32_t)CK) { cache[c * 64 + lane] = v; pCK = p; }
			c++;
		}
		e = e2 ? secEndR : min(p, secEndR);
	};
	parse();
	for (int round = 0; round < 66; round++) {
		const uint32_t ns = (uint32_t)__shfl_up((int)e, 1, 64);
		const bool dirty = lane > 0 && ns != s;
		const unsigned long long dm = __ballot(dirty);
		if (!dm) break;
		{ // the section ends after needCodes codes, somewhere inside the tile: the lanes behind that point parse the NEXT section's
		  // bits and need not converge; enough that the clean prefix of lanes already holds needCodes codes
			int32_t ci = (int32_t)c;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const int32_t t2 = __shfl_up(ci, o, 64); if (lane >= o) ci += t2; }
			const int firstDirty = __ffsll((long long)dm) - 1; // >= 1: lane 0 is never dirty
			if ((int64_t)__shfl(ci, firstDirty - 1, 64) >= needCodes) break;
		}
		if (dirty) { s = ns; if (s < segEnd) parse(); else { c = 0; e = s; pCK = s; } }
	}
	t.s = s; t.c = c; t.pCK = pCK;
	t.E = base + (uint64_t)(uint32_t)__shfl((int)e, 63, 64);
	return t;
}
// position behind the lane's first n codes (one lane asks, for the last code of a section)
template <int DEF>
__device__ __forceinline__ uint32_t w1_gamma_skip(const GraphDev &g, const lds_u32 *lw, const WindowSrc &src, uint32_t s, uint32_t n) {
	uint32_t p = s;
	int e2 = 0;
	for (uint32_t k = 0; k < n && !e2; k++) (void)w1_gamma<DEF>(g, lw, src, p, e2);
	return p;
}

// ---------------------------------------------------------------------------------------------- phase I
// Decodes the 2*ic gamma codes of the interval section starting at `pos` into arena entries (BVG:1077-1095);
// returns the bit position after them (start of the residual section) and the number of intervalised arcs.
template <int DEF, int NW>
__device__ __forceinline__ void coop_intervals(const Grp<NW> &G, const GraphDev &g, int32_t x, uint64_t pos, uint64_t recEnd, int64_t ic, int64_t extra, uint32_t B,
                                               IvEntry *__restrict__ list, uint32_t *lds, uint64_t &posAfter, int64_t &intervalArcs, int &err) {
	uint32_t *win = lds + CoopLds<NW>::OFF_WIN;
	int64_t codesDone = 0;              // uniform
	const int64_t codesAll = 2 * ic;
	int64_t cursor = x;                 // end of the previous interval; the first left is x + nat2int(v)
	int64_t pcount = 0;                 // intervalised arcs so far
	while (codesDone < codesAll) {
		const WindowSrc src = stage_tile<NW>(G, g, win, pos, B);
		uint64_t E; uint32_t s, c; int64_t unused;
		const uint64_t base = src.w0 << 5;
		spec_tile<DEF, 1, NW>(G, g, src, pos, recEnd, B, false, codesAll - codesDone, s, c, unused, E);
		int64_t tileTotal;
		const int64_t cincl = G.incl_scan((int64_t)c, tileTotal);
		const int64_t cb = cincl - c;
		const int64_t rem = codesAll - codesDone;
		if (cb >= rem) c = 0; else if (cb + c > rem) c = (uint32_t)(rem - cb);
		const int64_t total = min(rem, tileTotal); // codes this tile contributes
		if (total <= 0) { err |= E_FORMAT; break; }
		// pass 1: what my codes add to the cursor and to the arc count.  Code q of the section is a left gap
		// (q even) or a length (q odd).
		int64_t dcur = 0, dp = 0;
		uint32_t myEnd = s;
		{
			uint32_t p = s;
			for (uint32_t k = 0; k < c; k++) {
				const int64_t q = codesDone + cb + k;
				const uint64_t v = win_code_rel<DEF, 1>(g, src, p, err);
				if (q & 1) {
					if (v > (uint64_t)extra) err |= E_FORMAT; // (any 64-bit value in a malformed stream: the sums below must not wrap)
					const int64_t len = (int64_t)(v & 0x7fffffffu) + g.minInt; dcur += len; dp += len;
				}
				else dcur += q == 0 ? nat2int(v) : (int64_t)v + 1;
			}
			myEnd = p;
		}
		int64_t curTot, pTot;
		int64_t icur, ip;
		G.incl_scan2(dcur, dp, icur, ip, curTot, pTot);
		int64_t cur = cursor + icur - dcur, pc = pcount + ip - dp;
		// pass 2: write the entries
		{
			uint32_t p = s;
			int e2 = 0;
			for (uint32_t k = 0; k < c; k++) {
				const int64_t q = codesDone + cb + k;
				const uint64_t v = win_code_rel<DEF, 1>(g, src, p, e2);
				if (q & 1) { const int64_t len = (int64_t)(v & 0x7fffffffu) + g.minInt; list[q >> 1].pstart = (int32_t)pc; list[q >> 1].len = (int32_t)len; cur += len; pc += len; }
				else { cur += q == 0 ? nat2int(v) : (int64_t)v + 1; list[q >> 1].left = (int32_t)cur; }
			}
		}
		// the lane owning the last contributed code knows where the section really continues
		int64_t endRel, unused2;
		G.last2(c > 0, (int64_t)myEnd, 0, endRel, unused2);
		const uint64_t endPos = base + (uint64_t)endRel;
		cursor += curTot;
		pcount += pTot;
		codesDone += total;
		pos = codesDone >= codesAll ? endPos : E;
	}
	posAfter = pos;
	intervalArcs = pcount;
}

// Phase I by one wave, lean (default codings): the 2 * ic gamma codes of the interval section -> arena entries, every
// codeword decoded once (w1_gamma_tile keeps the values); same results as coop_intervals.
template <int DEF>
__device__ __forceinline__ void coop_intervals_w1(const Grp<1> &G, const GraphDev &g, int32_t x, uint64_t pos, uint64_t recEnd, int64_t ic, int64_t extra, uint32_t B,
                                                  IvEntry *__restrict__ list, uint32_t *lds, uint64_t &posAfter, int64_t &intervalArcs, int &err) {
	constexpr int CK = COOP1_CK;
	const int lane = threadIdx.x & 63;
	const lds_u32 *lw = (const lds_u32 *)(lds + CoopLds<1>::OFF_WIN);
	const lds_u32 *cache = (const lds_u32 *)(lds + CoopLds<1>::OFF_CACHE);
	int64_t codesDone = 0;
	const int64_t codesAll = 2 * ic;
	int64_t cursor = x, pcount = 0; // end of the previous interval (the first left is x + nat2int(v)); intervalised arcs so far
	while (codesDone < codesAll) {
		const W1Tile t = w1_gamma_tile<DEF>(G, g, lds, pos, recEnd, B, codesAll - codesDone);
		const uint64_t base = t.src.w0 << 5;
		uint32_t c = t.c;
		int64_t tileTotal;
		const int64_t cincl = G.incl_scan((int64_t)c, tileTotal);
		const int64_t cb = cincl - c, rem = codesAll - codesDone;
		if (cb >= rem) c = 0; else if (cb + c > rem) c = (uint32_t)(rem - cb);
		const int64_t total = min(rem, tileTotal);
		if (total <= 0) { err |= E_FORMAT; break; }
		// pass 1 (kept values): what my codes add to the cursor and to the arc count; code q is a left gap (q even) or a length (q odd)
		int64_t dcur = 0, dp = 0;
		{
			uint32_t p = t.pCK;
			for (uint32_t k = 0; k < c; k++) {
				const int64_t q = codesDone + cb + k;
				const uint32_t v = k < (uint32_t)CK ? cache[k * 64 + lane] : w1_gamma<DEF>(g, lw, t.src, p, err);
				if (q & 1) {
					if ((uint64_t)v > (uint64_t)extra) err |= E_FORMAT; // (any value in a malformed stream: the sums below must not wrap)
					const int64_t len = (int64_t)(v & 0x7fffffffu) + g.minInt; dcur += len; dp += len;
				}
				else dcur += q == 0 ? nat2int((uint64_t)v) : (int64_t)v + 1;
			}
		}
		int64_t curTot, pTot, icur, ip;
		G.incl_scan2(dcur, dp, icur, ip, curTot, pTot);
		int64_t cur = cursor + icur - dcur, pc = pcount + ip - dp;
		{ // pass 2: the entries
			uint32_t p = t.pCK;
			int e2 = 0;
			for (uint32_t k = 0; k < c; k++) {
				const int64_t q = codesDone + cb + k;
				const uint32_t v = k < (uint32_t)CK ? cache[k * 64 + lane] : w1_gamma<DEF>(g, lw, t.src, p, e2);
				if (q & 1) { const int64_t len = (int64_t)(v & 0x7fffffffu) + g.minInt; list[q >> 1].pstart = (int32_t)pc; list[q >> 1].len = (int32_t)len; cur += len; pc += len; }
				else { cur += q == 0 ? nat2int((uint64_t)v) : (int64_t)v + 1; list[q >> 1].left = (int32_t)cur; }
			}
		}
		cursor += curTot;
		pcount += pTot;
		codesDone += total;
		if (codesDone >= codesAll) { // the lane that owns the last code knows where the residual section starts
			const int lastTid = G.last_set(c > 0);
			uint32_t myEnd = 0;
			if (lane == lastTid) myEnd = w1_gamma_skip<DEF>(g, lw, t.src, t.s, c);
			pos = base + (uint64_t)G.bcast((int64_t)myEnd, lastTid);
		} else pos = t.E;
		G.sync(); // the window and the value slots are reused by the next tile
	}
	posAfter = pos;
	intervalArcs = pcount;
}

__device__ __attribute__((noinline)) int32_t iv_field_slow(const IvEntry *__restrict__ list, int64_t i, int which) { return which ? list[i].pstart : list[i].left; }

// ---------------------------------------------------------------------------------------------- phases R + X
// out = row + copied.  Residual j (0-based in the section) goes to out[j + (arcs of the intervals whose left
// extreme is smaller)], interval i to out[pstart_i + rank_i ..) where rank_i = residuals smaller than its left.
// Every lane decodes the run of residuals it owns and walks the (sorted) interval list alongside: it adds up
// the arcs of the intervals it passes and tells each of them its rank.  The intervals relevant to a tile are
// staged in LDS first (they are a contiguous slice of the list).
template <int DEF, int NW>
__device__ __forceinline__ void coop_residuals(const Grp<NW> &G, const GraphDev &g, int32_t x, uint64_t pos, uint64_t recEnd, int64_t nRes, int64_t ic, int64_t intervalArcs,
                                               IvEntry *__restrict__ list, int32_t *__restrict__ out, uint32_t *lds, int &err) {
	constexpr int N = Grp<NW>::N, IVCAP = CoopCfg<NW>::IVCAP;
	const int tid = G.tid();
	uint32_t *win = lds + CoopLds<NW>::OFF_WIN;
	int32_t *ivLeft = (int32_t *)lds + CoopLds<NW>::OFF_IVL, *ivP = ivLeft + (IVCAP + 1);
	int64_t resDone = 0;       // residuals written so far (uniform)
	int64_t baseVal = x;       // previous residual (BVG:954: the first one is x + nat2int(code))
	int64_t ia = 0;            // first interval not yet ranked (uniform)
	bool firstTile = true;
	const uint32_t B = coop_pick_B(recEnd > pos ? recEnd - pos : 0, (uint64_t)nRes, N, CoopCfg<NW>::B_MAX); // the residual section ends with the record
	unsigned long long tk = g.stats ? __builtin_readcyclecounter() : 0;
#define RT(slot) do { if (g.stats && NW != 1 && !(g.dbg & 16)) { const unsigned long long now_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g.stats[24 + slot], now_ - tk); tk = now_; } } while (0)
	while (resDone < nRes) {
		RT(7);
		const WindowSrc src = stage_tile<NW>(G, g, win, pos, B);
		RT(0);
		uint64_t E; uint32_t s, c; int64_t sum;
		spec_tile<DEF, 0, NW>(G, g, src, pos, recEnd, B, firstTile, nRes - resDone, s, c, sum, E);
		RT(1);
		// no more codes than the section still has
		int64_t tileTotal, sumTot, cincl, sincl;
		G.incl_scan2((int64_t)c, sum, cincl, sincl, tileTotal, sumTot);
		const int64_t cb = cincl - c;
		const int64_t lim = nRes - resDone;
		if (cb >= lim) c = 0; else if (cb + c > lim) c = (uint32_t)(lim - cb);
		const bool lastTile = tileTotal >= lim;
		const int64_t T = min(lim, tileTotal);
		if (T <= 0) { err |= E_FORMAT; break; }
		int64_t val = baseVal + sincl - sum; // the residual before my first one (x for the very first lane)
		// ---- stage the intervals that can fall among this tile's residuals: [ia, first left >= last value)
		int64_t staged = 0;
		if (ic > ia) {
			const int64_t hiVal = lastTile ? INT64_MAX : baseVal + sumTot;
			for (int64_t base = 0; base < IVCAP; base += N) {
				const int64_t i = ia + base + tid;
				const bool valid = i < ic && base + tid < IVCAP;
				int32_t l = 0x7fffffff, pp = (int32_t)intervalArcs;
				if (valid) { l = list[i].left; pp = list[i].pstart; }
				if (base + tid < IVCAP) { ivLeft[base + tid] = l; ivP[base + tid] = pp; }
				int64_t nt;
				(void)G.incl_scan(valid && (int64_t)l < hiVal ? 1 : 0, nt);
				staged = min<int64_t>(base + N, IVCAP);
				if (nt < N) break;
			}
			staged = min(staged, ic - ia);
			G.sync();
		}
		// (the un-staged case is out of line on purpose: merged into one load it would become a FLAT load, whose
		// wait also drains every successor store in flight)
		auto iv_left = [&](int64_t i) -> int64_t { const int64_t o = i - ia; if (__builtin_expect(o < staged, 1)) return (int64_t)ivLeft[o]; return (int64_t)iv_field_slow(list, i, 0); };
		auto iv_p = [&](int64_t i) -> int64_t { if (i >= ic) return intervalArcs; const int64_t o = i - ia; if (__builtin_expect(o < staged, 1)) return (int64_t)ivP[o]; return (int64_t)iv_field_slow(list, i, 1); };
		RT(2);
		// ---- my run of residuals, merged with the interval list
		int64_t i = ia;
		if (c && ic > ia && !(firstTile && tid == 0)) { // first interval with left > val (the residual before mine)
			int64_t lo = ia, hi = ic;
			while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (iv_left(mid) < val) lo = mid + 1; else hi = mid; }
			i = lo;
		}
		{
			uint32_t p = s;
			int64_t j = resDone + cb;
			int64_t arcsBefore = ic ? iv_p(i) : 0;
			const bool zigzag = firstTile && tid == 0;
			int64_t nextLeft = (c && i < ic) ? iv_left(i) : INT64_MAX; // left extreme of the next interval not yet passed
			auto emit = [&](int64_t add) {
				val += add;
				if (nextLeft < val) {
					do { list[i].rank = (int32_t)j; i++; nextLeft = i < ic ? iv_left(i) : INT64_MAX; } while (nextLeft < val); // interval i sits after j residuals
					arcsBefore = iv_p(i);
				}
				out[j + arcsBefore] = (int32_t)val;
				j++;
			};
			uint32_t k = 0;
			if (c && zigzag) { emit(nat2int(win_code_rel<DEF, 0>(g, src, p, err))); k = 1; } // BVG:954
			for (; k < c; k++) emit((int64_t)win_code_rel<DEF, 0>(g, src, p, err) + 1);      // BVG:966
			if (c == 0) i = 0;
		}
		RT(3);
		int64_t lastVal;
		G.last2(c > 0, val, i, lastVal, ia); // ia: everything before it has been ranked
		if (NW == 1) G.sync(); // staged intervals / window are reused by the next tile (last2 ends with a barrier)
		resDone += T;
		baseVal = lastVal;
		pos = E; // a cut only happens on the last tile
		firstTile = false;
		RT(4);
	}
	if (ic == 0) return;
	G.sync_global(); // ranks written above are read by other lanes below
	RT(5);
	// phase X: interval i occupies out[pstart + rank .. + len)  (IntIntervalSequenceIterator.java:64-78)
	for (int64_t i0 = 0; i0 < ic; i0 += N) {
		const int64_t i = i0 + tid;
		int32_t left = 0, len = 0; int64_t p = 0;
		if (i < ic) { const IvEntry en = list[i]; left = en.left; len = en.len; p = (int64_t)en.pstart + en.rank; }
		const bool isLong = len > 16;
		if (!isLong) for (int32_t t = 0; t < len; t++) out[p + t] = left + t;
		unsigned long long lm = __ballot(isLong);
		while (lm) { // long intervals: a whole wave fills one at a time
			const int srcl = __ffsll((long long)lm) - 1;
			lm &= lm - 1;
			const int32_t L = __shfl(left, srcl, 64), Nn = __shfl(len, srcl, 64);
			const int64_t P = shfl_i64(p, srcl);
			for (int32_t t = G.lane(); t < Nn; t += 64) out[P + t] = L + t;
		}
	}
	RT(6);
#undef RT
}

// A long copy-block list (gamma codes, default coding) walked by ONE wave cooperatively instead of code by code: the
// same speculative tile decode as an interval section (coop_intervals in bv_coop.hpp), block b playing the part of
// a (copied, skipped) alternation.  Fills kend[j] / delta[j] for the j-th copied block exactly as the serial walk
// does, including the implicit last block, and returns the totals.  Called by the 64 lanes of wave 0 only.
constexpr int COPY_COOP_WALK_MIN = 192; // below this many blocks the serial walk is as fast
template <class L = CoopLds<1>>
__device__ __forceinline__ void coop_block_walk(const GraphDev &g, uint64_t pos, uint64_t recEnd, int64_t bc, int64_t dref, int32_t d, int32_t *kend, int32_t *delta, int32_t tabCap,
                                                uint32_t *lds, int64_t &totalOut, int64_t &copiedOut, int32_t &nKeptOut, int &bad, uint64_t *posAfter = nullptr) {
	Grp<1> G{ (int64_t *)(lds + L::OFF_XCH) };
	constexpr int CK = COOP1_CK;
	const int lane = threadIdx.x & 63;
	const lds_u32 *lw = (const lds_u32 *)(lds + L::OFF_WIN);
	const lds_u32 *cache = (const lds_u32 *)(lds + L::OFF_CACHE);
	const uint32_t B = coop_pick_B(min<uint64_t>(recEnd > pos ? recEnd - pos : 0, (uint64_t)bc * 8), (uint64_t)bc, 64, CoopCfg<1>::B_MAX);
	int64_t done = 0, total = 0, copied = 0; // uniform
	int err = 0;
	while (done < bc) {
		const W1Tile t = w1_gamma_tile<1, L>(G, g, lds, pos, recEnd, B, bc - done);
		const uint64_t base = t.src.w0 << 5;
		uint32_t c = t.c;
		int64_t tileTotal;
		const int64_t cincl = G.incl_scan((int64_t)c, tileTotal);
		const int64_t cb = cincl - c, rem = bc - done;
		if (cb >= rem) c = 0; else if (cb + c > rem) c = (uint32_t)(rem - cb);
		const int64_t n = min(rem, tileTotal);
		if (n <= 0) { err = 1; break; }
		// pass 1 (from the kept values): what my codes add to the referent index and to the number of copied ids
		int64_t dAll = 0, dEven = 0;
		{
			uint32_t p = t.pCK;
			for (uint32_t k = 0; k < c; k++) {
				const int64_t q = done + cb + k;
				const uint32_t v = k < (uint32_t)CK ? cache[k * 64 + lane] : w1_gamma<1>(g, lw, t.src, p, err);
				if ((uint64_t)v > (uint64_t)dref) err |= 1; // (any value in a malformed stream: the sums below must not wrap)
				const int64_t len = (int64_t)(v & 0x7fffffffu) + (q ? 1 : 0);
				dAll += len;
				if (!(q & 1)) dEven += len;
			}
		}
		int64_t allTot, evenTot;
		const int64_t iAll = G.incl_scan(dAll, allTot), iEven = G.incl_scan(dEven, evenTot);
		int64_t tt = total + iAll - dAll, cp = copied + iEven - dEven;
		// pass 2: the table entries of my copied blocks
		if (tabCap > 0) {
			uint32_t p = t.pCK;
			int e2 = 0;
			for (uint32_t k = 0; k < c; k++) {
				const int64_t q = done + cb + k;
				const uint32_t v = k < (uint32_t)CK ? cache[k * 64 + lane] : w1_gamma<1>(g, lw, t.src, p, e2);
				const int64_t len = (int64_t)(v & 0x7fffffffu) + (q ? 1 : 0);
				if (!(q & 1)) {
					const int64_t j = q >> 1;
					if (j < tabCap) { kend[j] = (int32_t)min<int64_t>(cp + len, 0x7fffffff); delta[j] = (int32_t)(tt - cp); }
					cp += len;
				}
				tt += len;
			}
		}
		total += allTot;
		copied += evenTot;
		done += n;
		if (done >= bc) { // the lane that owns the last code of the list knows where the next section starts
			const int lastTid = G.last_set(c > 0);
			uint32_t myEnd = 0;
			if (lane == lastTid) myEnd = w1_gamma_skip<1>(g, lw, t.src, t.s, c);
			pos = base + (uint64_t)G.bcast((int64_t)myEnd, lastTid);
		} else pos = t.E;
		G.sync(); // the window and the value slots are reused by the next tile
		if (total > dref || copied > d) { err = 1; break; } // (uniform)
	}
	if (G.any(err != 0)) { bad = 1; return; }
	// implicit last block: the rest of the referent's row, copied when the block count is even
	const int64_t rest = dref - total;
	if (rest < 0) { bad = 1; return; }
	if (!(bc & 1)) {
		const int64_t j = bc >> 1;
		if (j < tabCap && lane == 0) { kend[j] = (int32_t)min<int64_t>(copied + rest, 0x7fffffff); delta[j] = (int32_t)(total - copied); }
		copied += rest;
	}
	total += rest;
	totalOut = total;
	copiedOut = copied;
	if (posAfter) *posAfter = pos;
	nKeptOut = (int32_t)min<int64_t>((bc >> 1) + 1, 0x7fffffff);
}

// The same walk by ALL the waves of a group of NW (its tile is NW times as wide; every codeword is decoded three times -- speculation,
// sums, table entries -- instead of once, and every scan costs two barriers): for block lists of tens of thousands of codes, where
// one wave's walk is the longest thing the record or the row does (77 000 codes: 0.9 ms).  `win` holds N * bmax / 32 + 12 words.
constexpr int COPY_GROUP_WALK_MIN = 4096;
template <int NW>
__device__ __forceinline__ void coop_block_walk_nw(const GraphDev &g, uint64_t pos, uint64_t recEnd, int64_t bc, int64_t dref, int32_t d, int32_t *kend, int32_t *delta, int32_t tabCap,
                                                   uint32_t *win, int64_t *xch, uint32_t bmax, int64_t &totalOut, int64_t &copiedOut, int32_t &nKeptOut, int &bad, uint64_t *posAfter = nullptr) {
	Grp<NW> G{ xch };
	const uint32_t B = coop_pick_B(min<uint64_t>(recEnd > pos ? recEnd - pos : 0, (uint64_t)bc * 8), (uint64_t)bc, Grp<NW>::N, bmax);
	int64_t done = 0, total = 0, copied = 0; // uniform
	int err = 0;
	while (done < bc) {
		const WindowSrc src = stage_tile<NW>(G, g, win, pos, B);
		const uint64_t base = src.w0 << 5;
		uint64_t E; uint32_t s, c; int64_t unused;
		spec_tile<1, 1, NW>(G, g, src, pos, recEnd, B, false, bc - done, s, c, unused, E);
		int64_t tileTotal;
		const int64_t cincl = G.incl_scan((int64_t)c, tileTotal);
		const int64_t cb = cincl - c, rem = bc - done;
		if (cb >= rem) c = 0; else if (cb + c > rem) c = (uint32_t)(rem - cb);
		const int64_t n = min(rem, tileTotal);
		if (n <= 0) { err = 1; break; } // (uniform)
		// pass 1: what my codes add to the referent index and to the number of copied ids
		int64_t dAll = 0, dEven = 0;
		uint32_t myEnd = s;
		{
			uint32_t p = s;
			for (uint32_t k = 0; k < c; k++) {
				const int64_t q = done + cb + k;
				const uint64_t v = win_code_rel<1, 1>(g, src, p, err);
				if (v > (uint64_t)dref) err |= 1; // (any 64-bit value in a malformed stream: the sums below must not wrap)
				const int64_t len = (int64_t)(v & 0x7fffffffu) + (q ? 1 : 0);
				dAll += len;
				if (!(q & 1)) dEven += len;
			}
			myEnd = p;
		}
		int64_t iAll, iEven, allTot, evenTot;
		G.incl_scan2(dAll, dEven, iAll, iEven, allTot, evenTot);
		// pass 2: the table entries of my copied blocks
		if (tabCap > 0) {
			int64_t t = total + iAll - dAll, cp = copied + iEven - dEven;
			uint32_t p = s;
			int e2 = 0;
			for (uint32_t k = 0; k < c; k++) {
				const int64_t q = done + cb + k;
				const int64_t len = (int64_t)(win_code_rel<1, 1>(g, src, p, e2) & 0x7fffffffu) + (q ? 1 : 0);
				if (!(q & 1)) {
					const int64_t j = q >> 1;
					if (j < tabCap) { kend[j] = (int32_t)min<int64_t>(cp + len, 0x7fffffff); delta[j] = (int32_t)(t - cp); }
					cp += len;
				}
				t += len;
			}
		}
		const int lastTid = G.last_set(c > 0);
		const uint64_t endPos = base + (uint64_t)G.bcast((int64_t)myEnd, lastTid);
		total += allTot;
		copied += evenTot;
		done += n;
		pos = done >= bc ? endPos : E;
		if (total > dref || copied > d) { err = 1; break; } // (uniform)
	}
	if (G.any(err != 0)) { bad = 1; return; }
	// implicit last block: the rest of the referent's row, copied when the block count is even
	const int64_t rest = dref - total;
	if (rest < 0) { bad = 1; return; }
	if (!(bc & 1)) {
		const int64_t j = bc >> 1;
		if (j < tabCap && threadIdx.x == 0) { kend[j] = (int32_t)min<int64_t>(copied + rest, 0x7fffffff); delta[j] = (int32_t)(total - copied); }
		copied += rest;
	}
	total += rest;
	totalOut = total;
	copiedOut = copied;
	if (posAfter) *posAfter = pos;
	nKeptOut = (int32_t)min<int64_t>((bc >> 1) + 1, 0x7fffffff);
}


// ---------------------------------------------------------------------------------------------- phases R, one wave, lean
// The residual section of a record decoded by ONE wave with every codeword decoded once where the general version above
// decodes it up to three times (run-in, counting parse, value pass): the lane keeps the gaps of its first parse in LDS
// ([slot][lane], COOP1_CK of them) and the value pass only adds them up.  The decoder is straight-line: two LDS words, a
// branch-free decode of the short codewords (zeta_3 < 2^21, or zeta_k that fits 32 bits), ONE branch to the generic reader
// for the rest; counts and sums are 32-bit (ids are Java ints: BVG:954, :966 compute in int).  Default codings only.
template <int DEF>
__device__ __forceinline__ uint32_t w1_residual(const GraphDev &g, const lds_u32 *lw, const WindowSrc &src, uint32_t &q, int &err) {
	const uint32_t j = q >> 5, sh = q & 31u;
	const uint32_t a = lw[j], b = lw[j + 1];
	const uint32_t W = (uint32_t)(((((uint64_t)a << 32) | b) << sh) >> 32);
	const uint32_t h = (uint32_t)__clz((int)W); // 32 for W == 0
	const uint32_t k = DEF == 1 ? 3u : (uint32_t)g.zetaK;
	const uint32_t nb = k * h + k - 1;
	const bool ok = DEF == 1 ? h < 7 : (h + 2 + nb <= 32u && nb != 0);
	const uint32_t mm = (W << ((h + 1) & 31u)) >> ((31u - nb) & 31u); // the nb payload bits plus the extra bit of the long codeword
	const uint32_t m = mm >> 1, left = 1u << ((k * h) & 31u);
	const bool lng = m >= left;
	if (__builtin_expect(ok, 1)) { q += h + 1 + nb + (lng ? 1u : 0u); return lng ? mm - 1 : m + left - 1; }
	const SlowCode sc = win_code_slow<DEF, 0>(&g, src.win, src.w0, src.nw, q);
	q = sc.q; err |= sc.err;
	return (uint32_t)sc.v;
}

template <int DEF>
__device__ __forceinline__ void coop_residuals_w1(const GraphDev &g, int32_t x, uint64_t pos, uint64_t recEnd, int64_t nRes64, int64_t ic64, int64_t intervalArcs,
                                                  IvEntry *__restrict__ list, int32_t *__restrict__ out, uint32_t *lds, int &err) {
	constexpr int IVCAP = CoopCfg<1>::IVCAP, CK = COOP1_CK;
	const Grp<1> G{ (int64_t *)(lds + CoopLds<1>::OFF_XCH) };
	const int lane = threadIdx.x & 63;
	uint32_t *win = lds + CoopLds<1>::OFF_WIN;
	const lds_u32 *lw = (const lds_u32 *)win;
	lds_u32 *ivLeft = (lds_u32 *)(lds + CoopLds<1>::OFF_IVL), *ivP = ivLeft + (IVCAP + 1), *cache = (lds_u32 *)(lds + CoopLds<1>::OFF_CACHE);
	const int32_t nRes = (int32_t)nRes64, ic = (int32_t)ic64;
	int32_t ia = 0; // first interval that no residual has passed yet (uniform)
	const uint32_t B = coop_pick_B(recEnd > pos ? recEnd - pos : 0, (uint64_t)nRes, 64, CoopCfg<1>::B_MAX);
	const uint32_t R = min((uint32_t)COOP_RUNIN_MAX, B >> COOP_RUNIN_SHIFT);
	int32_t resDone = 0, baseVal = x;
	bool firstTile = true;
	// (BVGPU_STATS=1: the wave class's ticks by step, slots 48 ..: stage, run-in, first parse, rounds, scans + interval staging, search + values, tail; 55 tiles, 58 sum of B, 59 codes)
	unsigned long long wk = g.stats ? __builtin_readcyclecounter() : 0;
#define WT(slot) do { if (g.stats) { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g.stats[48 + slot], now_ - wk); wk = now_; } } while (0)
	if (g.stats && lane == 0) { atomicAdd(&g.stats[58], (unsigned long long)B); atomicAdd(&g.stats[59], (unsigned long long)nRes); atomicAdd(&g.stats[57], 1ull); }
	auto wscan = [&](int32_t vv) { // inclusive scan over the wave
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const int32_t t = __shfl_up(vv, o, 64); if (lane >= o) vv += t; }
		return vv;
	};
	while (resDone < nRes) {
		WT(6);
		const WindowSrc src = stage_tile<1>(G, g, win, pos, B); // (ends with a wave sync: the interval table above is in place too)
		WT(0);
		if (g.stats && lane == 0) atomicAdd(&g.stats[55], 1ull);
		const uint64_t base = src.w0 << 5;
		const uint32_t p0 = (uint32_t)(pos - base), secEndR = (uint32_t)min(recEnd - base, (uint64_t)0x7fffff00u);
		const uint32_t segEnd = min(p0 + (uint32_t)(lane + 1) * B, secEndR);
		uint32_t s = lane == 0 ? min(p0, secEndR) : min(p0 + (uint32_t)lane * B, secEndR);
		if (BV_TIMING(g, 0x800)) { resDone = nRes; break; } // (timing experiments only, scripts/r6g.sh: the tile's stage and nothing else)
		if (lane > 0 && s < secEndR && R && !BV_TIMING(g, 0x1000)) { // run-in: lock onto the code boundaries before the segment starts
			uint32_t p = s - min(R, s - p0);
			int e2 = 0;
			while (p < s && !e2) (void)w1_residual<DEF>(g, lw, src, p, e2);
			s = e2 ? s : min(p, secEndR);
		}
		WT(1);
		uint32_t e, c, pCK; int32_t sum;
		auto parse = [&]() { // the codes that start in [s, segEnd): count, what they add to the running id, end; the first CK gaps kept
			c = 0; sum = 0;
			uint32_t p = s;
			pCK = s;
			int e2 = 0; // a speculative parse may run through garbage: errors only stop it
			while (p < segEnd && !e2) {
				const uint32_t cv = w1_residual<DEF>(g, lw, src, p, e2);
				const int32_t add = (firstTile && lane == 0 && c == 0) ? (int32_t)nat2int(cv) : (int32_t)cv + 1; // BVG:954 / :966
				if (c < (uint32_t)CK) { cache[c * 64 + lane] = (uint32_t)add; pCK = p; }
				sum += add; c++;
			}
			e = e2 ? secEndR : min(p, secEndR);
		};
		parse();
		WT(2);
		for (int round = 0; round < 66; round++) { // a segment starts where its left neighbour ended
			uint32_t ns = (uint32_t)__shfl_up((int)e, 1, 64);
			const bool dirty = lane > 0 && ns != s;
			if (!__any(dirty) || BV_TIMING(g, 0x2000)) break;
			if (g.stats && lane == 0) atomicAdd(&g.stats[56], 1ull);
			if (dirty) { s = ns; if (s < segEnd) parse(); else { c = 0; sum = 0; e = s; } }
		}
		WT(3);
		const int32_t cincl = wscan((int32_t)c), sincl = wscan(sum);
		const int32_t tileTotal = __shfl(cincl, 63, 64);
		const int32_t cb = cincl - (int32_t)c, lim = nRes - resDone;
		int32_t cn = (int32_t)c;
		if (cb >= lim) cn = 0; else if (cb + cn > lim) cn = lim - cb; // no more codes than the section still has
		const int32_t T = min(lim, tileTotal);
		if (T <= 0) { err |= E_FORMAT; break; }
		int32_t val = baseVal + sincl - sum; // the residual before my first one
		// ---- the intervals that can fall among this tile's residuals -> LDS: [ia, first left >= the tile's last value), IVCAP at most
		int32_t staged = 0;
		if (ic > ia) {
			const bool lastTile = tileTotal >= lim;
			const int32_t hiVal = lastTile ? 0x7fffffff : baseVal + __shfl(sincl, 63, 64);
			for (int32_t b0 = 0; b0 < IVCAP; b0 += 64) {
				const int32_t i = ia + b0 + lane;
				int32_t l = 0x7fffffff, pp = (int32_t)intervalArcs;
				if (i < ic) { l = list[i].left; pp = list[i].pstart; }
				ivLeft[b0 + lane] = (uint32_t)l; ivP[b0 + lane] = (uint32_t)pp;
				staged = b0 + 64;
				if (__popcll(__ballot(i < ic && l < hiVal)) < 64) break;
			}
			staged = min(staged, ic - ia);
			G.sync();
		}
		auto iv_left = [&](int32_t i) -> int32_t { if (i >= ic) return 0x7fffffff; const int32_t o = i - ia; if (__builtin_expect(o < staged, 1)) return (int32_t)ivLeft[o]; return iv_field_slow(list, i, 0); };
		auto iv_p = [&](int32_t i) -> int32_t { if (i >= ic) return (int32_t)intervalArcs; const int32_t o = i - ia; if (__builtin_expect(o < staged, 1)) return (int32_t)ivP[o]; return iv_field_slow(list, i, 1); };
		// ---- my run of residuals at their final places, between the intervals
		int32_t jj = resDone + cb, ii = ia;
		if (cn && ic > ia && !(firstTile && lane == 0)) { // first interval that the residuals before mine have not passed
			int32_t lo2 = ia, hi2 = ic;
			while (lo2 < hi2) { const int32_t mid = (lo2 + hi2) >> 1; if (iv_left(mid) < val) lo2 = mid + 1; else hi2 = mid; }
			ii = lo2;
		}
		int32_t before = ic ? iv_p(ii) : 0, nextLeft = (cn && ic) ? iv_left(ii) : 0x7fffffff;
		uint32_t p = pCK;
		WT(4);
		// (the ids go straight to the row, a run per lane: handing them to the stores through LDS -- eight lanes' runs of eight per instruction, ~10 lines touched instead of 64 --
		// made the kernel 6 % SLOWER, and without its stores the loop is 3 % faster: the value pass is a third of the kernel for what it issues, not for what it stores;
		// profiles/r6_experiments.txt section 3)
		if (!BV_TIMING(g, 128)) // (timing experiments only)
		for (int32_t k2 = 0; k2 < cn; k2++) {
			const int32_t add = k2 < CK ? (int32_t)cache[k2 * 64 + lane] : (int32_t)w1_residual<DEF>(g, lw, src, p, err) + 1;
			val += add;
			if (nextLeft < val) {
				do { list[ii].rank = jj; ii++; nextLeft = iv_left(ii); } while (nextLeft < val); // interval ii sits after jj residuals
				before = iv_p(ii);
			}
			out[jj + before] = val;
			jj++;
		}
		const unsigned long long has = __ballot(cn > 0);
		const int lastL = has ? 63 - __clzll((long long)has) : 0;
		baseVal = __shfl(val, lastL, 64);
		ia = has ? __shfl(ii, lastL, 64) : ia; // everything before it has been ranked
		resDone += T;
		pos = base + (uint64_t)(uint32_t)__shfl((int)e, 63, 64);
		firstTile = false;
		WT(5);
		G.sync(); // the window and the gap slots are reused by the next tile
	}
#undef WT
}

// The whole record of node x by one group.  Same contract as parse_node: extras merged into row[copied..d).
template <int DEF, int NW>
__device__ __forceinline__ void coop_parse_node(const GraphDev &g, int32_t x, int32_t d, bool hasRef, int64_t dref, int32_t *__restrict__ row,
                                                IvEntry *__restrict__ list, uint32_t *lds, int *__restrict__ errOut, bvsg::RecDesc *segOut = nullptr) {
	Grp<NW> G{ (int64_t *)(lds + CoopLds<NW>::OFF_XCH) };
	const int tid = G.tid();
	const uint64_t recEnd = (uint64_t)g.offsets[x + 1];
	int err = 0;
	const int sb = NW == 1 ? 16 : 20; // stats slots: ticks of phase A, I, R, X
	unsigned long long tk = g.stats ? __builtin_readcyclecounter() : 0;
#define COOP_TICK(slot) do { if (g.stats) { const unsigned long long now_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g.stats[sb + slot], now_ - tk); tk = now_; } } while (0)
	// phase A (uniform): outdegree, reference, copy blocks (BVG:1048-1071)
	BitReader br;
	br.init(g.bits, g.nwords);
	br.seek((uint64_t)g.offsets[x]);
	(void)Fields<DEF>::outdegree(br, g);
	if (g.W > 0) (void)Fields<DEF>::reference(br, g);
	int64_t copied = 0;
	int64_t woff = -1; // where this record's block tables went (NW > 1, long lists), if anywhere
	int32_t wKept = 0;
	if (hasRef) {
		const uint64_t bc = Fields<DEF>::block_count(br, g);
		int64_t total = 0;
		if (bc > (uint64_t)dref + 1) err |= E_FORMAT;
		else if (DEF && bc >= COPY_COOP_WALK_MIN && !br.err) {
			// A long block list (a long row copying a long row: thousands of codes) walked code by code would be the serial
			// part of the whole record: the first wave decodes it cooperatively, totals only (no tables), and tells the others.
			int64_t cp = 0;
			int32_t nKept = 0;
			int bad = 0;
			uint64_t after = br.pos();
			const bool allWaves = NW > 1 && bc >= COPY_GROUP_WALK_MIN; // (uniform)
			const int64_t kMaxW = (int64_t)(bc >> 1) + 1;
			if (allWaves && g.walktab && d >= g.walkMin) { // the copy pass will want this list's tables: they fall out of the walk (GraphDev::walktab)
				if (tid == 0) {
					const uint64_t need = 2 * (uint64_t)kMaxW;
					int64_t o = -1;
					if (need <= g.walkCap) { const uint32_t a = atomicAdd(g.walkCursor, (uint32_t)need); if ((uint64_t)a + need <= g.walkCap) o = a; }
					G.xch[NW + 4] = o;
				}
				__syncthreads();
				woff = G.xch[NW + 4];
				__syncthreads();
			}
			if (allWaves) coop_block_walk_nw<NW>(g, br.pos(), recEnd, (int64_t)bc, dref, d, woff >= 0 ? g.walktab + woff : (int32_t *)nullptr, woff >= 0 ? g.walktab + woff + kMaxW : (int32_t *)nullptr,
			                                     woff >= 0 ? (int32_t)min<int64_t>(kMaxW, 0x7fffffff) : 0, lds + CoopLds<NW>::OFF_WIN,
			                                     (int64_t *)(lds + CoopLds<NW>::OFF_XCH), CoopCfg<NW>::B_MAX, total, cp, nKept, bad, &after);
			else if (NW == 1 || G.wave() == 0) coop_block_walk(g, br.pos(), recEnd, (int64_t)bc, dref, d, (int32_t *)nullptr, (int32_t *)nullptr, 0, lds, total, cp, nKept, bad, &after);
			if (NW > 1 && !allWaves) {
				if (tid == 0) { G.xch[NW + 1] = cp; G.xch[NW + 2] = (int64_t)after; G.xch[NW + 3] = bad; }
				__syncthreads();
				cp = G.xch[NW + 1]; after = (uint64_t)G.xch[NW + 2]; bad = (int)G.xch[NW + 3];
				__syncthreads();
			}
			if (bad) err |= E_FORMAT;
			copied = cp;
			wKept = nKept;
			br.seek(after);
		}
		else {
			for (uint64_t b = 0; b < bc; b++) {
				int64_t len;
				if (!block_len_ok(Fields<DEF>::block(br, g), b == 0, total, dref, len)) { err |= E_FORMAT; break; }
				total += len;
				if (!(b & 1)) copied += len;
			}
			if (!(bc & 1)) copied += dref - total;
		}
	}
	const int64_t extra = (int64_t)d - copied;
	if (extra < 0 || copied < 0) err |= E_FORMAT;
	err |= br.err;
	if (err) { if (tid == 0) atomicOr(errOut, err); return; }
	if (NW > 1 && hasRef && copied >= 1 && tid == 0 && g.walktab && d >= g.walkMin) { // (row[0 .. copied) is free until the copy pass)
		if (woff >= 0 && copied >= 4) { row[0] = -2; row[1] = (int32_t)woff; row[2] = wKept; row[3] = (int32_t)copied; }
		else row[0] = -1;
	}
	COOP_TICK(0);
	if (extra == 0) return;
	int64_t ic = 0, intervalArcs = 0;
	uint64_t pos = br.pos();
	if (g.minInt != 0) {
		ic = (int64_t)br.gamma();
		pos = br.pos();
		if (br.err || ic > extra / g.minInt) { if (tid == 0) atomicOr(errOut, E_FORMAT | br.err); return; }
		if (ic) {
			// codes left in the record: 2 per interval + at most extra - ic*minInt residuals -> lower bound of the
			// mean code length; the interval section itself is estimated as 2*ic codes of that length
			const uint64_t restBits = recEnd > pos ? recEnd - pos : 0, restCodes = (uint64_t)(2 * ic + (extra - ic * g.minInt));
			const uint64_t secEst = min(restBits, (restBits * (uint64_t)(2 * ic) + restCodes - 1) / restCodes * 2); // x2: interval gaps are longer than residual gaps
			const uint32_t B = coop_pick_B(secEst, (uint64_t)(2 * ic), Grp<NW>::N, CoopCfg<NW>::B_MAX);
			if constexpr (NW == 1 && DEF != 0) coop_intervals_w1<DEF>(G, g, x, pos, recEnd, ic, extra, B, list, lds, pos, intervalArcs, err);
			else coop_intervals<DEF, NW>(G, g, x, pos, recEnd, ic, extra, B, list, lds, pos, intervalArcs, err);
			if (G.any(err != 0)) { if (err) atomicOr(errOut, err); return; } // group-uniform exit
		}
	}
	const int64_t nRes = extra - intervalArcs;
	if (nRes < 0) { if (tid == 0) atomicOr(errOut, E_FORMAT); return; }
	COOP_TICK(1);
	// default rank: the interval follows every residual (phase R fixes up the others)
	for (int64_t i = tid; i < ic; i += Grp<NW>::N) list[i].rank = (int32_t)nRes;
	if (segOut && nRes > 0 && nRes < 0x7fffffff && ic < 0x7fffffff) { // the residual section goes to the segment pipeline, cut into pieces (bv_seg.hip); it also expands the intervals
		if (tid == 0) { segOut->rpos = (int64_t)pos; segOut->nres = (int32_t)nRes; segOut->copied = (int32_t)copied; segOut->nIv = (int32_t)ic; segOut->ivArcs = (int32_t)intervalArcs; segOut->flags = 0; }
		return;
	}
	G.sync_global();
	if (NW == 1 && DEF != 0 && ic < 0x7fffffff && nRes < 0x7fffffff) { // one wave, default codings: every codeword decoded once
		if (nRes > 0 && !BV_TIMING(g, 0x4000)) coop_residuals_w1<DEF>(g, x, pos, recEnd, nRes, ic, intervalArcs, list, row + copied, lds, err);
		COOP_TICK(2);
		if (ic > 0 && !BV_TIMING(g, 0x8000)) { // phase X: interval i occupies out[pstart + rank .. + len)  (IntIntervalSequenceIterator.java:64-78)
			G.sync_global();
			int32_t *out = row + copied;
			for (int64_t i0 = 0; i0 < ic; i0 += 64) {
				const int64_t i = i0 + tid;
				int32_t left = 0, len = 0; int64_t p = 0;
				if (i < ic) { const IvEntry en = list[i]; left = en.left; len = en.len; p = (int64_t)en.pstart + en.rank; }
				const bool isLong = len > 16;
				if (!isLong) for (int32_t t = 0; t < len; t++) out[p + t] = left + t; // (the short ones flat over the wave -- id f of their concatenation by lane f mod 64 -- cost 3 % more: r6_experiments.txt)
				unsigned long long lm = __ballot(isLong);
				while (lm) {
					const int srcl = __ffsll((long long)lm) - 1;
					lm &= lm - 1;
					const int32_t L = __shfl(left, srcl, 64), Nn = __shfl(len, srcl, 64);
					const int64_t P = shfl_i64(p, srcl);
					for (int32_t t = G.lane(); t < Nn; t += 64) out[P + t] = L + t;
				}
			}
		}
	}
	else coop_residuals<DEF, NW>(G, g, x, pos, recEnd, nRes, ic, intervalArcs, list, row + copied, lds, err);
	COOP_TICK(NW == 1 ? 3 : 2); // (one wave: slot 2 = the residuals, slot 3 = the expansion of the intervals)
#undef COOP_TICK
	if (err) atomicOr(errOut, err);
}

} // namespace bv
