// bv_lanewin.hpp -- one lane decodes one record through a lane-private window of the stream in LDS (gfx950).
//
// gfx950 counts outstanding global loads AND stores in one counter (vmcnt), so a lane that fetches its next 32
// bits from HBM waits for every successor store it has in flight: a record of 2000 successors decoded by one lane
// straight from HBM took ~1.3 us per successor, all of it latency.  Here a lane fetches 64 bytes of its record
// with four 16-byte loads, parks them in its own LDS column and decodes from there with the stateless 32-bit
// decoders of bv_coop.hpp: one wait per ~500 bits of stream instead of one per 32.
// Default codings only (gamma / unary / zeta_3); the generic reader handles the rest (parse_node).
#pragma once
#include "bv_coop.hpp"

namespace bv {

constexpr int LW_STRIDE = 256;               // threads per block: word k of lane t lives at lds[k * LW_STRIDE + t] (one bank per lane)
#ifndef LW_MAIN_
#define LW_MAIN_ 16
#endif
constexpr int LW_MAIN = LW_MAIN_, LW_SIDE = 16;    // words per lane: main cursor / interval cursor
constexpr int LW_LDS_WORDS = (LW_MAIN + LW_SIDE) * LW_STRIDE;

struct SlowAbs { uint64_t v, pos; int err; };
// codewords longer than 64 bits (or garbage): the generic reader, out of line
template <int KIND>
__device__ __attribute__((noinline)) SlowAbs lane_code_slow(const uint32_t *bits, uint64_t nwords, uint64_t pos) {
	BitReader br;
	br.init(bits, nwords);
	br.seek(pos);
	const uint64_t v = KIND == 2 ? br.unary() : KIND == 1 ? br.gamma() : br.template zeta_k<3>(3);
	return SlowAbs{ v, br.pos(), br.err };
}
// zeta_k with the graph's k (kept apart: one more live argument across the call costs the zeta_3 kernels registers)
__device__ __attribute__((noinline)) SlowAbs lane_zeta_slow(const uint32_t *bits, uint64_t nwords, uint64_t pos, int zetaK) {
	BitReader br;
	br.init(bits, nwords);
	br.seek(pos);
	const uint64_t v = br.template zeta_k<0>(zetaK);
	return SlowAbs{ v, br.pos(), br.err };
}

template <int NWORDS> struct LaneWin {
	uint32_t *col;  // this lane's LDS column
	uint64_t w0;    // absolute index of window word 0 (multiple of 4: 16-byte loads)
	uint32_t q;     // bit offset of the cursor from word w0
	uint64_t vlast; // first word of the last 16-byte vector worth fetching (record end + look-ahead, inside the padded image)

	__device__ __forceinline__ void fill(const GraphDev &g) {
		constexpr int NV = NWORDS / 4;
		uint4 v[NV];
#pragma unroll
		for (int k = 0; k < NV; k++) {
			// branch-free (all the loads of a refill must be in flight together): past the record, or past the
			// padded image (which ends with >= 8 zero words), the last wanted vector is simply read again
			const uint64_t i = min(w0 + 4 * k, vlast);
			v[k] = *(const uint4 *)(g.bits + i);
		}
#pragma unroll
		for (int k = 0; k < NV; k++) {
			col[(4 * k + 0) * LW_STRIDE] = __builtin_bswap32(v[k].x);
			col[(4 * k + 1) * LW_STRIDE] = __builtin_bswap32(v[k].y);
			col[(4 * k + 2) * LW_STRIDE] = __builtin_bswap32(v[k].z);
			col[(4 * k + 3) * LW_STRIDE] = __builtin_bswap32(v[k].w);
		}
	}
	__device__ __forceinline__ void seek(const GraphDev &g, uint64_t pos) {
		w0 = (pos >> 5) & ~(uint64_t)3;
		q = (uint32_t)(pos - (w0 << 5));
		fill(g);
	}
	// Called where the wave is converged, for records of `bits` bits from pos on: when EVERY lane's record (and the two words its decoders look ahead) ends inside the first half
	// of the window, only that half is fetched -- two 16-byte loads per lane instead of four (each goes to a line of its own: scripts/ubench_lines.hip); the other half keeps
	// what it held, which nothing reads (no cursor gets there: a record that ends in the first half never triggers a refill).
	__device__ __forceinline__ void seek_short(const GraphDev &g, uint64_t pos, uint64_t bits) {
		w0 = (pos >> 5) & ~(uint64_t)3;
		q = (uint32_t)(pos - (w0 << 5));
		if (__builtin_amdgcn_ballot_w64((uint64_t)q + bits + 64 > (uint64_t)NWORDS * 16) != 0) { fill(g); return; }
		constexpr int NV = NWORDS / 8;
		uint4 v[NV];
#pragma unroll
		for (int k = 0; k < NV; k++) v[k] = *(const uint4 *)(g.bits + min(w0 + 4 * k, vlast));
#pragma unroll
		for (int k = 0; k < NV; k++) {
			col[(4 * k + 0) * LW_STRIDE] = __builtin_bswap32(v[k].x);
			col[(4 * k + 1) * LW_STRIDE] = __builtin_bswap32(v[k].y);
			col[(4 * k + 2) * LW_STRIDE] = __builtin_bswap32(v[k].z);
			col[(4 * k + 3) * LW_STRIDE] = __builtin_bswap32(v[k].w);
		}
	}
	__device__ __forceinline__ uint64_t pos() const { return (w0 << 5) + q; }
	// (what parse_node_lwc asks of a reader -- the tile kernel's is TileRd, bv_tile.hpp)
	__device__ __forceinline__ uint32_t word(uint32_t j) const { return col[j * LW_STRIDE]; } // word j of the window
	__device__ __forceinline__ bool low(uint32_t margin) const { return (q >> 5) + margin >= (uint32_t)NWORDS; } // fewer than `margin` words left behind the cursor
	// Called where the wave is converged: if ANY lane is about to run out of window, ALL lanes move theirs up to
	// their cursor.  Left to themselves the lanes would each stall the whole wave for a memory round trip at a
	// different iteration (64 lanes, one refill every ~40 codes each: a stall in almost every iteration).
	template <int MARGIN> __device__ __forceinline__ void wave_refill(const GraphDev &g) { // MARGIN: words the next step may need
		if (__any((q >> 5) + MARGIN >= (uint32_t)NWORDS)) {
			const uint32_t adv = (q >> 5) & ~3u;
			w0 += adv;
			q -= adv << 5;
			fill(g);
		}
	}
	// KIND 0: zeta_k (ZK = 3: the default, folded in; 0: the graph's zetak at run time), 1: gamma, 2: unary
	template <int KIND, int ZK = 3> __device__ __forceinline__ uint64_t code(const GraphDev &g, int &err) {
		if (__builtin_expect((q >> 5) + 2 >= (uint32_t)NWORDS, 0)) { // keep three words ahead of the cursor inside the window
			const uint32_t adv = (q >> 5) & ~3u;
			w0 += adv;
			q -= adv << 5;
			fill(g);
		}
		const uint32_t j = q >> 5, sh = q & 31u;
		const uint64_t ab = ((uint64_t)col[j * LW_STRIDE] << 32) | col[(j + 1) * LW_STRIDE];
		const uint32_t W = (uint32_t)((ab << sh) >> 32);
		uint32_t v, len;
		if (KIND == 2) {
			if (__builtin_expect(W != 0, 1)) { const uint32_t z = (uint32_t)__clz((int)W); q += z + 1; return z; }
		} else if (__builtin_expect(KIND == 1 ? fast_gamma32(W, v, len) : fast_zeta_32<ZK>(W, ZK == 3 ? 3u : (uint32_t)g.zetaK, v, len), 1)) { q += len; return v; }
		// up to 64 bits
		const uint32_t c = col[(j + 2) * LW_STRIDE];
		const uint64_t W64 = sh ? (ab << sh) | ((uint64_t)c >> (32u - sh)) : ab;
		uint64_t v64;
		if (KIND == 2) {
			if (W64) { const uint32_t z = (uint32_t)__clzll((long long)W64); q += z + 1; return z; }
		} else if (KIND == 1 ? fast_gamma(W64, v64, len) : fast_zeta<ZK>(W64, ZK == 3 ? 3u : (uint32_t)g.zetaK, v64, len)) { q += len; return v64; }
		const SlowAbs sa = (KIND == 0 && ZK != 3) ? lane_zeta_slow(g.bits, g.nwords, pos(), g.zetaK) : lane_code_slow<KIND>(g.bits, g.nwords, pos());
		err |= sa.err;
		seek(g, sa.pos);
		return sa.v;
	}
};

// Same contract as parse_node<true>: the extras (intervals merged with residuals) of node x go to row[copied..d).
// iv = the record's slice of the interval arena (>= d / minInt + 1 entries of 8 bytes): the reading of the interval section keeps what it
// decodes -- (left, length) per interval -- in a ring of LW_RING entries in the lane's LDS column and, when there are more, in the arena;
// the merge takes an interval from the ring with two LDS reads instead of two gamma decodes, and the rings are topped up from the arena
// by all lanes together, like the stream windows.  (Loading the intervals back from the arena one ahead, without a ring, was 30 % slower:
// a load in the merge loop waits for the loop's stores.  Round 3's loop -- a trip per successor with a branch per case, ~250
// wave-instructions per trip -- and the variant that read the interval section twice are tag r4-experiments.)
#ifndef LW_RING_
#define LW_RING_ 8
#endif
constexpr int LW_RING = LW_RING_; // (a power of two; 2 * LW_RING <= LW_SIDE words of the lane's column)
// The record by a loop that the 64 lanes of a wave walk in step (round 4; default codings -- zeta_3, or ZK = 0: the graph's zeta_k --, interval arena).  A merge loop with a
// branch per case executes ~250 wave-instructions per trip, a third of them scalar: every `if` of a lane is an exec-mask region
// (s_and_saveexec / s_cbranch / s_or) that the wave runs through as soon as ONE lane takes it, and every code() carries a refill
// check and three fallbacks of its own.  Here a trip is straight-line: the next gap and the next ring entry are decoded speculatively
// by every lane and kept or dropped by selects; what is rare (window refill, ring top-up, a codeword of more than 28 bits, the
// unaligned head of the row) sits behind ONE wave-uniform vote each.  The sections in front of the residuals are read by loops the
// wave walks together too (code_w).  Semantics: BVG:1040-1126 (equal heads once, MergedIntIterator.java:69-72).
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; } // (one s_cmp on the mask; __any goes through a 0 / 1 value per lane)
template <int KIND, int ZK = 3> __device__ __forceinline__ bool lane_fast_code(uint32_t W, uint32_t &v, uint32_t &len, uint32_t zk = 3u) { // branch-free; v / len are junk when the result is false
	if (KIND == 2) { const uint32_t z = (uint32_t)__clz((int)(W | 1u)); v = z; len = z + 1; return W != 0; }
	if (KIND == 1) { const uint32_t m = (uint32_t)__clz((int)(W | (1u << 16))); len = 2 * m + 1; v = (W >> (31u - 2 * m)) - 1; return W >= (1u << 16); }
	if (ZK == 3) {
		const uint32_t h = (uint32_t)__clz((int)(W | (1u << 25))); // zeta_3, h <= 6
		const uint32_t nb = 3 * h + 2;
		const uint32_t mm = (W << (h + 1)) >> (31u - nb);
		const uint32_t m = mm >> 1, left = 1u << (3 * h);
		const bool lng = m >= left;
		v = lng ? mm - 1 : m + left - 1;
		len = 4 * h + 3 + (lng ? 1u : 0u);
		return W >= (1u << 25);
	}
	// zeta_k with the graph's k (1 .. 16): unary h, then k h + k - 1 bits, one more if that is not a short codeword; codewords of up to 32 bits
	const uint32_t h = (uint32_t)__clz((int)(W | 1u)), nb = zk * h + zk - 1;
	const bool ok = W != 0 && h + 2 + nb <= 32u; // (then h <= 30 and nb <= 30: the shifts below are in range; otherwise they are masked and the result dropped)
	const uint32_t mm = (W << ((h + 1) & 31u)) >> ((31u - nb) & 31u);
	const uint32_t m = mm >> 1, left = 1u << ((zk * h) & 31u);
	const bool lng = m >= left;
	v = lng ? mm - 1 : m + left - 1; // (zeta_1, h = 0: nb = 0, mm = the extra bit, m = 0 < left = 1: v = 0, len = 1)
	len = h + 1 + nb + (lng ? 1u : 0u);
	return ok;
}
// Called where the wave is converged: the lanes with `want` consume one code, the others keep their cursor.
template <int KIND, int ZK = 3> __device__ __forceinline__ uint64_t code_w(LaneWin<LW_MAIN> &br, const GraphDev &g, bool want, int &err) {
	br.template wave_refill<3>(g);
	const uint32_t j = br.q >> 5, sh = br.q & 31u;
	const uint64_t ab = ((uint64_t)br.col[j * LW_STRIDE] << 32) | br.col[(j + 1) * LW_STRIDE];
	uint32_t v, len;
	const bool ok = lane_fast_code<KIND, ZK>((uint32_t)((ab << sh) >> 32), v, len, (uint32_t)g.zetaK);
	uint64_t r = v;
	if (wave_any(want && !ok)) {
		if (want && !ok) r = br.template code<KIND, ZK>(g, err);
		else if (want) br.q += len;
	} else br.q += want ? len : 0u;
	return r;
}
// off0 / off1: the record's first bit and the next record's (g.offsets[x], g.offsets[x + 1]; the caller fetched them a sweep ahead)
// HASH (bvg_scan_checksum, HashCtx in bv_launch.hpp): every id the loop emits is also added to hacc with weight hw (then hw *= 31: the ids of a row are consecutive in the
// hashed sequence; hw = 0 for a row that is hashed elsewhere), and the row is written only if hstore (somebody copies from it, or it is not hashed here).
template <int ZK, bool HASH = false>
__device__ __forceinline__ void parse_node_lwb(const GraphDev &g, int32_t x, int32_t d, bool hasRef, int64_t dref, int32_t *__restrict__ row, uint32_t *lds, int2 *__restrict__ iv, int *__restrict__ err, uint64_t off0, uint64_t off1,
                                               uint32_t *hacc = nullptr, uint32_t hw = 0, bool hstore = true) {
	LaneWin<LW_MAIN> br;
	br.col = lds + threadIdx.x;
	uint32_t *const ring = lds + LW_MAIN * LW_STRIDE + threadIdx.x; // entry j of the ring: ring[2 j * LW_STRIDE] = left, ring[(2 j + 1) * LW_STRIDE] = length
	br.vlast = min(((off1 >> 5) + 2) & ~(uint64_t)3, (g.nwords + 4) & ~(uint64_t)3);
	br.seek(g, off0);
	int e = 0;
	(void)code_w<1>(br, g, true, e);              // outdegree (known from k_headers)
	if (g.W > 0) (void)code_w<2>(br, g, true, e); // reference
	int64_t copied = 0;
	{ // BVG:1058-1071
		uint64_t bc = code_w<1>(br, g, hasRef, e);
		if (!hasRef) bc = 0;
		if (bc > (uint64_t)dref + 1) { e |= E_FORMAT; bc = 0; }
		int64_t total = 0;
		const uint32_t nb = (uint32_t)bc; // (<= dref + 1 <= 2^31)
		for (uint32_t b = 0; wave_any(b < nb && !e); b++) {
			const bool w = b < nb && !e;
			const uint64_t c = code_w<1>(br, g, w, e);
			int64_t len = 0;
			const bool good = block_len_ok(c, b == 0, total, dref, len);
			if (w && !good) e |= E_FORMAT;
			if (w && good) { total += len; if (!(b & 1)) copied += len; }
		}
		if (hasRef && !e && !(bc & 1)) copied += dref - total;
	}
	const int64_t extra = (int64_t)d - copied;
	if (extra < 0 || copied < 0) e |= E_FORMAT;
	if (e) { atomicOr(err, e); return; }
	if (extra == 0) return;

	int32_t nIntervals = 0;
	int64_t intervalArcs = 0;
	if (g.minInt != 0) { // BVG:1073-1096: the interval section, kept as (left, length) in the ring / the arena
		const uint64_t ni = code_w<1>(br, g, true, e);
		if (ni > (uint64_t)extra / (uint64_t)g.minInt) { atomicOr(err, E_FORMAT); return; } // (an interval holds >= minInt ids; the arena slice has d / minInt + 1 entries -- ADVICE r4)
		nIntervals = (int32_t)ni;
		int32_t prevEnd = 0;
		for (int32_t i = 0; wave_any(i < nIntervals && !e); i++) {
			const bool w = i < nIntervals && !e;
			const uint64_t a = code_w<1>(br, g, w, e);
			const uint64_t len = code_w<1>(br, g, w, e);
			if (w) {
				if (len > (uint64_t)extra) e |= E_FORMAT; // (any 64-bit value in a malformed stream: kept out of the sum)
				else {
					intervalArcs += (int64_t)len + g.minInt;
					const int32_t left = i == 0 ? (int32_t)((int64_t)x + nat2int(a)) : prevEnd + (int32_t)a + 1, n = (int32_t)len + g.minInt; // BVG:1084-1093, in Java ints
					prevEnd = left + n;
					if (i < LW_RING) { ring[(2 * i) * LW_STRIDE] = (uint32_t)left; ring[(2 * i + 1) * LW_STRIDE] = (uint32_t)n; }
					if (nIntervals > LW_RING) iv[i] = int2{ left, n };
				}
			}
		}
	}
	const int64_t nRes = extra - intervalArcs;
	if (nRes < 0 || e) { atomicOr(err, E_FORMAT | e); return; }

	// merge(intervals, residuals) -> row[copied ..), 16 bytes at a time behind an unaligned head.  Ids are Java ints (BVG:954, :966, :1084-1093).
	int32_t *const out = row + copied;
	const int32_t nExtra = (int32_t)extra;
	const int32_t head = min(nExtra, (int32_t)(((16u - ((uint32_t)(uintptr_t)out & 15u)) & 15u) >> 2));
	int32_t k = 0, o0 = 0, o1 = 0, o2 = 0, o3 = 0, on = 0;
	int32_t ivLeft = 0, ivRem = 0, ivTodo = nIntervals;
	int32_t ivIdx = 0, ivBase = 0, ivLoaded = min(ivTodo, LW_RING); // next interval; oldest one in the ring; intervals [ivBase, ivLoaded) are in the ring
	int32_t resTodo = (int32_t)nRes;
	int32_t resVal = (int32_t)((int64_t)x + nat2int(code_w<0, ZK>(br, g, resTodo != 0, e))); // BVG:954
	while (k < nExtra) {
		const bool lowRing = ivLoaded < nIntervals && ivIdx - ivBase >= LW_RING - 2;
		if (wave_any(lowRing | ((br.q >> 5) + 3 >= (uint32_t)LW_MAIN))) {
			br.template wave_refill<3>(g);
			if (wave_any(lowRing)) { // some lane's ring runs low: every lane tops its own up from the arena
				const int32_t cnt = min((ivIdx - ivBase) & ~1, nIntervals - ivLoaded); // (ivLoaded stays even until the last top-up)
#pragma unroll
				for (int p = 0; p < LW_RING / 2; p++) {
					if (2 * p < cnt) {
						const int4 t = *(const int4 *)(iv + ivLoaded + 2 * p); // (the slice has room for twice the entries: reading one past the last is harmless)
						const int j0 = (ivLoaded + 2 * p) & (LW_RING - 1);
						ring[(2 * j0) * LW_STRIDE] = (uint32_t)t.x; ring[(2 * j0 + 1) * LW_STRIDE] = (uint32_t)t.y;
						ring[(2 * j0 + 2) * LW_STRIDE] = (uint32_t)t.z; ring[(2 * j0 + 3) * LW_STRIDE] = (uint32_t)t.w;
					}
				}
				if (cnt > 0) { ivBase += cnt; ivLoaded += cnt; }
			}
		}
		// the ring's next entry and the stream's next gap, read by every lane whether it will use them or not
		const int jr = ivIdx & (LW_RING - 1);
		const int32_t rl = (int32_t)ring[(2 * jr) * LW_STRIDE], rn = (int32_t)ring[(2 * jr + 1) * LW_STRIDE];
		const uint32_t jw = br.q >> 5, sh = br.q & 31u;
		const uint64_t ab = ((uint64_t)br.col[jw * LW_STRIDE] << 32) | br.col[(jw + 1) * LW_STRIDE];
		uint32_t gap, len;
		const bool ok = lane_fast_code<0, ZK>((uint32_t)((ab << sh) >> 32), gap, len, (uint32_t)g.zetaK);
		const bool fetch = ivRem == 0 && ivTodo != 0;
		ivLeft = fetch ? rl : ivLeft; ivRem = fetch ? rn : ivRem; ivIdx += fetch; ivTodo -= fetch;
		const bool haveRes = resTodo != 0;
		const bool takeIv = ivRem != 0 && (!haveRes || ivLeft < resVal);
		const int32_t val = takeIv ? ivLeft : haveRes ? resVal : -1; // (-1: malformed, fewer values than the outdegree promises; BVG:1210 would store -1)
		const bool ivAdv = takeIv || (haveRes && ivRem != 0 && ivLeft == resVal); // equal heads are emitted once (MergedIntIterator.java:69-72)
		ivLeft += ivAdv; ivRem -= ivAdv;
		const bool useRes = !takeIv && haveRes;
		resTodo -= useRes;
		const bool adv = useRes && resTodo != 0;
		if (wave_any(adv && !ok)) {
			if (adv && !ok) resVal += (int32_t)br.template code<0, ZK>(g, e) + 1;
			else if (adv) { resVal += (int32_t)gap + 1; br.q += len; }
		} else { resVal += adv ? (int32_t)gap + 1 : 0; br.q += adv ? len : 0u; } // BVG:966
		if (HASH) { *hacc += (uint32_t)val * hw; hw *= 31u; }
		const bool inHead = k < head;
		if (wave_any(inHead)) { if (inHead && (!HASH || hstore)) out[k] = val; }
		k++;
		o0 = o1; o1 = o2; o2 = o3; o3 = val;
		on += !inHead;
		if (on == 4) { if (!HASH || hstore) *(int4 *)(out + k - 4) = int4{ o0, o1, o2, o3 }; on = 0; }
	}
	if (!HASH || hstore) {
		if (on == 3) { out[k - 3] = o1; out[k - 2] = o2; out[k - 1] = o3; }
		else if (on == 2) { out[k - 2] = o2; out[k - 1] = o3; }
		else if (on == 1) out[k - 1] = o3;
	}
	if (e) atomicOr(err, e);
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the same record by a loop with less in it (parse_node_lwc).  What changed against parse_node_lwb, and why:
//  * the outdegree and the reference are not decoded again: k_headers did, and a gamma code of d takes 2 floor(log2(d + 1)) + 1 bits, a unary
//    reference r + 1 -- the window is opened behind them;
//  * the sums of the block section are 32-bit (block_len_ok keeps 0 <= total <= dref <= 2^31 - 1), the interval arcs are checked as they
//    accumulate (no sum can pass `extra`);
//  * the merge state carries SENTINELS instead of flags: an exhausted residual section is resVal = 0xffffffff, the entry behind the last interval
//    is (0xffffffff, 1) -- in the ring and, for lists that do not fit it, in the arena --, so a trip is val = min(ivLeft, resVal) (unsigned: ids
//    are Java ints >= 0 in every valid file), `both equal` advances both (MergedIntIterator.java:69-72), and a row that runs out of values pads
//    itself with -1 (BVG:1210) with no case of its own.  ~17 vector and one scalar instruction of merge logic per id instead of ~20 + 12;
//  * the trip counter is the WAVE's: lane l's k-th id is decoded in trip k by every lane, finished lanes idle on their sentinels (no exec masks
//    inside the trip), four trips per pass; the four ids leave in ONE 16-byte store at out + k, whatever its alignment (gfx950 takes
//    dwordx4 stores on 4-byte boundaries; scripts/ubench_store.hip: +15 % on the store itself) -- no unaligned head, no per-lane phase of the
//    store, no register shuffle; the refill vote is once per pass;
//  * the copy blocks leave as a TABLE for the copy pass's lane class: 16 bytes per slot in a scratch array of its own (CopyTab: header =
//    copied << 16 | kept blocks -- CT_NONE in the low half: no table, walk the stream -- and the first three kept blocks, each first index in the referent's
//    row << 16 | length); kept blocks from the fourth on go to the END of the part of the interval arena that is the record's alone (16 floor(d / minInt)
//    bytes from floor(rowstart / minInt): its front holds the record's own intervals while it is parsed), entry j at ovfEnd[2 - j].
//    k_copy_list then touches neither the stream nor the offsets and runs no bit reader.
constexpr uint32_t CT_NONE = 0xffffu;
constexpr int32_t LW_TAB_D = 1024; // rows with fewer successors get a table (the lane class of the copy pass ends at copy_mid_min <= COPY_BIG_MIN = 1024)
typedef int32_t i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// the record's slice of the interval arena: entries [abase, abase + n) of 16 bytes, abase = floor(rowstart / minInt), n = d / minInt + 1 (minInt > 0).  The LAST entry is
// shared with the next record (floor(a / k) + floor(b / k) <= floor((a + b) / k): only the first n - 1 are the record's alone).
__device__ __forceinline__ void arena_slice(int32_t minInt, int64_t rowstart, int32_t d, int64_t &abase, int32_t &n) {
	if ((minInt & (minInt - 1)) == 0) { const int sh = 31 - __clz(minInt); abase = rowstart >> sh; n = (d >> sh) + 1; } // (uniform: the usual 4 costs two shifts, not a 64-bit division)
	else { abase = rowstart / minInt; n = d / minInt + 1; }
}
struct CopyTab { int32_t t2, t1, t0, hdr; }; // one per slot (16 bytes, read and written as an int4)

// zeta_3 from a 32-bit window, branch-free: what the code ADDS to the running id (gap + 1 = value + 1) and its length; false: longer than 28 bits
__device__ __forceinline__ bool lane_zeta3_add(uint32_t W, uint32_t &add, uint32_t &len) {
	const uint32_t h = (uint32_t)__clz((int)(W | (1u << 25))); // h <= 6
	const uint32_t h3 = 3u * h;
	const uint32_t mm = (W << (h + 1u)) >> (29u - h3); // 3 h + 3 bits: the short codeword's 3 h + 2 and the extra bit of the long one
	const uint32_t m = mm >> 1, left = 1u << h3;
	const bool lng = m >= left;
	add = lng ? mm : m + left; // value + 1
	len = 4u * h + 3u + (lng ? 1u : 0u);
	return W >= (1u << 25);
}
// As code_w, 32 bits wide: values past 2^32 - 1 (a malformed stream) saturate, which every caller rejects.
template <int KIND, int ZK = 3, class RD> __device__ __forceinline__ uint32_t code_w32(RD &br, const GraphDev &g, bool want, int &err) {
	br.template wave_refill<3>(g);
	const uint32_t j = br.q >> 5, sh = br.q & 31u;
	const uint64_t ab = ((uint64_t)br.word(j) << 32) | br.word(j + 1);
	uint32_t v, len;
	const bool ok = lane_fast_code<KIND, ZK>((uint32_t)((ab << sh) >> 32), v, len, (uint32_t)g.zetaK);
	if (wave_any(want && !ok)) {
		if (want && !ok) { const uint64_t t = br.template code<KIND, ZK>(g, err); v = KIND == 0 ? (uint32_t)t : (uint32_t)min<uint64_t>(t, 0xffffffffull); } // (residuals are Java ints: truncated, BVG:954)
		else if (want) br.q += len;
	} else br.q += want ? len : 0u;
	return v;
}

// off0 / off1: the record's first bit and the next record's; r: its reference (0: none), dref: the referent's outdegree; iv: the record's slice of the interval arena
// (ivOwn 16-byte entries of it are the record's alone); ctab: the slot's table (null: none).  HASH: as parse_node_lwb.
// where the record's block count starts: behind the outdegree (gamma: 2 floor(log2(d + 1)) + 1 bits) and the reference (unary: r + 1 bits) that k_headers read (BVG:1048-1054)
__device__ __forceinline__ uint64_t record_body(const GraphDev &g, uint64_t off0, int32_t d, int32_t r) {
	return off0 + (2u * (31u - (uint32_t)__clz((int)((uint32_t)d + 1u))) + 1u) + (g.W > 0 ? (uint32_t)r + 1u : 0u);
}
// br: the reader, positioned at record_body() (LaneWin: the lane's own window of the stream; TileRd: the tile's shared image); ring: the lane's ring of RING intervals in LDS (a power
// of two), entry j = ring[2 j * LW_STRIDE] (left), ring[(2 j + 1) * LW_STRIDE] (length)
template <int ZK, bool HASH = false, int RING = LW_RING, class RD>
__device__ __forceinline__ void parse_node_lwc(const GraphDev &g, RD &br, uint32_t *ring, int32_t x, int32_t d, int32_t r, int32_t dref, int32_t *__restrict__ row, int2 *__restrict__ iv, int32_t ivOwn,
                                               CopyTab *__restrict__ ctab, int *__restrict__ err, uint32_t *hacc = nullptr, uint32_t hw = 0, bool hstore = true) {
	const bool hasRef = r > 0;
	int e = 0;
	int32_t copied = 0;
	const bool tab = hasRef && ctab != nullptr;
	bool tabOk = tab && d < LW_TAB_D && dref < 65536;
	uint32_t t0 = 0, t1 = 0, t2 = 0, kept = 0;
	int32_t *const ovfEnd = (int32_t *)(iv + 2 * (int64_t)ivOwn); // kept blocks from the fourth on: down from the end of the record's own part of the arena
	auto push = [&](uint32_t en) { // kept block `kept` of the table
		t0 = kept == 0 ? en : t0; t1 = kept == 1 ? en : t1; t2 = kept == 2 ? en : t2;
		if (kept >= 3) { if ((int32_t)kept - 2 <= 4 * ivOwn) ovfEnd[2 - (int32_t)kept] = (int32_t)en; else tabOk = false; }
		kept++;
	};
	{ // BVG:1058-1071
		uint32_t bc = code_w32<1>(br, g, hasRef, e);
		if (!hasRef) bc = 0;
		if (bc > (uint32_t)dref + 1u) { e |= E_FORMAT; bc = 0; }
		int32_t total = 0;
		for (uint32_t b = 0; wave_any(b < bc && !e); b++) {
			const bool w = b < bc && !e;
			const uint32_t c = code_w32<1>(br, g, w, e);
			const uint32_t room = (uint32_t)(dref - total), len = c + (b ? 1u : 0u);
			const bool good = c <= room && len <= room; // (block_len_ok)
			if (w && !good) e |= E_FORMAT;
			if (w && good) {
				if (!(b & 1)) { if (tabOk && len) push(((uint32_t)total << 16) | len); copied += (int32_t)len; }
				total += (int32_t)len;
			}
		}
		if (hasRef && !e && !(bc & 1)) { const int32_t rest = dref - total; if (tabOk && rest) push(((uint32_t)total << 16) | (uint32_t)rest); copied += rest; }
	}
	const int32_t extra = d - copied;
	if (extra < 0) e |= E_FORMAT;
	auto leave_table = [&](bool ok) { if (tab) *(int4 *)ctab = int4{ (int32_t)t2, (int32_t)t1, (int32_t)t0, ok ? (int32_t)(((uint32_t)copied << 16) | kept) : (int32_t)CT_NONE }; };
	if (e) { leave_table(false); atomicOr(err, e); return; }
	if (extra == 0) { leave_table(tabOk); return; }

	int32_t nIv = 0, ivArcs = 0;
	const uint32_t SENT = 0xffffffffu;
	if (g.minInt != 0) { // BVG:1073-1096: the interval section, kept as (left, length) in the ring / the arena
		const uint32_t ni = code_w32<1>(br, g, true, e);
		if ((uint64_t)ni * (uint32_t)g.minInt > (uint64_t)(uint32_t)extra) { leave_table(false); atomicOr(err, E_FORMAT); return; } // (an interval holds >= minInt ids; the arena slice has d / minInt + 1 entries)
		nIv = (int32_t)ni;
		const bool spill = nIv >= RING; // with the sentinel the list does not fit the ring: all of it goes to the arena too
		int32_t prevEnd = 0;
		for (int32_t i = 0; wave_any(i < nIv && !e); i++) {
			const bool w = i < nIv && !e;
			const uint32_t a = code_w32<1>(br, g, w, e);
			const uint32_t len = code_w32<1>(br, g, w, e);
			if (w) {
				if (len > (uint32_t)(extra - ivArcs) || len + (uint32_t)g.minInt > (uint32_t)(extra - ivArcs)) e |= E_FORMAT; // (the intervals' ids are among the `extra`: no sum passes it)
				else {
					const int32_t n = (int32_t)len + g.minInt;
					ivArcs += n;
					const int32_t left = i == 0 ? (int32_t)((int64_t)x + nat2int(a)) : prevEnd + (int32_t)a + 1; // BVG:1084-1093, in Java ints
					prevEnd = left + n;
					if (i < RING) { ring[(2 * i) * LW_STRIDE] = (uint32_t)left; ring[(2 * i + 1) * LW_STRIDE] = (uint32_t)n; }
					if (spill) iv[i] = int2{ left, n };
				}
			}
		}
		if (nIv < RING) { ring[(2 * nIv) * LW_STRIDE] = SENT; ring[(2 * nIv + 1) * LW_STRIDE] = 1u; }
		else iv[nIv] = int2{ -1, 1 };
		if (spill && tabOk && kept > 3 && 8 * ((int64_t)nIv + 2) + 4 * ((int64_t)kept - 3) > 16 * (int64_t)ivOwn) tabOk = false; // (the list reached into the table's end of the record's part)
	} else { ring[0] = SENT; ring[LW_STRIDE] = 1u; }
	leave_table(tabOk && !e);
	if (e) { atomicOr(err, E_FORMAT | e); return; }
	const int32_t nRes = extra - ivArcs; // >= 0

	// merge(intervals, residuals) -> row[copied ..).  Ids are Java ints (BVG:954, :966, :1084-1093).
	int32_t *const out = row + copied;
	uint32_t ivLeft = ring[0], ivRem = ring[LW_STRIDE]; // the first interval (or the sentinel)
	int32_t ivIdx = min(1, nIv);                          // the next entry to take: never past the sentinel's
	int32_t ivLoaded = RING;                           // (lists in the arena) entries [ivLoaded - RING, ivLoaded) are in the ring; even
	const bool spillLane = nIv >= RING;
	const bool anySpill = wave_any(spillLane);
	uint32_t resVal = SENT;
	int32_t resLeft = 0; // codes of the residual section not read yet
	{
		const uint32_t first = code_w32<0, ZK>(br, g, nRes != 0, e);
		if (nRes != 0) { resVal = (uint32_t)((int64_t)x + nat2int(first)); resLeft = nRes - 1; } // BVG:954
	}
	auto trip = [&](int32_t k) -> int32_t {
		// the ring's next entry and the stream's next gap, read by every lane whether it will use them or not
		const int jr = ivIdx & (RING - 1);
		const uint32_t rl = ring[(2 * jr) * LW_STRIDE], rn = ring[(2 * jr + 1) * LW_STRIDE];
		const uint32_t jw = br.q >> 5, sh = br.q & 31u;
		const uint64_t ab = ((uint64_t)br.word(jw) << 32) | br.word(jw + 1);
		const uint32_t W = (uint32_t)((ab << sh) >> 32);
		uint32_t add, len;
		bool ok;
		if (ZK == 3) ok = lane_zeta3_add(W, add, len);
		else { ok = lane_fast_code<0, ZK>(W, add, len, (uint32_t)g.zetaK); add += 1u; }
		const uint32_t val = min(ivLeft, resVal);
		const bool aI = ivLeft == val, aR = resVal == val; // equal heads are emitted once (MergedIntIterator.java:69-72)
		ivLeft += aI ? 1u : 0u; ivRem -= aI ? 1u : 0u;
		const bool need = ivRem == 0;
		ivLeft = need ? rl : ivLeft; ivRem = need ? rn : ivRem;
		ivIdx = min(ivIdx + (need ? 1 : 0), nIv);
		const bool more = resLeft != 0, adv = aR && more;
		uint32_t nv = resVal + add; // BVG:966
		if (wave_any(adv && !ok)) {
			if (adv && !ok) nv = resVal + (uint32_t)br.template code<0, ZK>(g, e) + 1u;
			else if (adv) br.q += len;
		} else br.q += adv ? len : 0u;
		resVal = aR ? (more ? nv : SENT) : resVal;
		resLeft -= adv ? 1 : 0;
		if (HASH) { *hacc += k < extra ? val * hw : 0u; hw *= 31u; }
		return (int32_t)val;
	};
	int32_t p1 = 0, p2 = 0, p3 = 0; // the last pass's last three ids
	for (int32_t k0 = 0; wave_any(k0 < extra); k0 += 4) {
		// four codes of <= 32 bits behind the cursor, four entries of the ring: or ALL lanes move their windows / top their rings up
		const bool low = spillLane && ivLoaded <= nIv && ivLoaded - ivIdx < 4;
		if (wave_any(low | br.low(5))) {
			br.template wave_refill<5>(g);
			if (anySpill && wave_any(low)) {
				// entries [ivLoaded, upto) replace consumed ones (index - RING < ivIdx), two per 16-byte load (ivLoaded is even; one entry past the sentinel may be read: the slice has the room)
				const int32_t upto = spillLane ? min((ivIdx + RING) & ~1, (nIv + 2) & ~1) : 0;
#pragma unroll
				for (int p = 0; p < RING / 2; p++) {
					const int32_t i0 = ivLoaded + 2 * p;
					if (i0 < upto) {
						const int4 t = *(const int4 *)(iv + i0);
						const int j0 = i0 & (RING - 1);
						ring[(2 * j0) * LW_STRIDE] = (uint32_t)t.x; ring[(2 * j0 + 1) * LW_STRIDE] = (uint32_t)t.y;
						ring[(2 * j0 + 2) * LW_STRIDE] = (uint32_t)t.z; ring[(2 * j0 + 3) * LW_STRIDE] = (uint32_t)t.w;
					}
				}
				ivLoaded = max(ivLoaded, upto);
			}
		}
		const int32_t v0 = trip(k0), v1 = trip(k0 + 1), v2 = trip(k0 + 2), v3 = trip(k0 + 3);
		if ((!HASH || hstore) && !BV_TIMING(g, 0x10000)) { // (0x10000: timing experiments only)
			const int32_t left = extra - k0;
			if (left >= 4) *(i32x4_a4 *)(out + k0) = i32x4_a4{ v0, v1, v2, v3 };
			else if (left > 0) {
				// the last one to three ids: ONE 16-byte store over the row's last four ids (the first of them are the last pass's, still at hand) when the row has four --
				// a store instruction of this kernel costs its CU a line per lane whatever it carries, and three 4-byte stores were three of them
				if (k0 > 0) *(i32x4_a4 *)(out + extra - 4) = left == 1 ? i32x4_a4{ p1, p2, p3, v0 } : left == 2 ? i32x4_a4{ p2, p3, v0, v1 } : i32x4_a4{ p3, v0, v1, v2 };
				else { out[k0] = v0; if (left > 1) out[k0 + 1] = v1; if (left > 2) out[k0 + 2] = v2; }
			}
		}
		p1 = v1; p2 = v2; p3 = v3;
	}
	if (e) atomicOr(err, e);
}

} // namespace bv
