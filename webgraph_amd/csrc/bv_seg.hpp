// bv_seg.hpp -- the segment pipeline: residual sections of long records decoded in pieces of SEG_BITS bits of stream, one lane per
// piece, whatever record the piece belongs to (gfx950).  Shared by the device kernels (bv_seg.hip) and a host model compiled with g++
// (tests/cpp/seg_model.cpp), which runs the same bodies lane after lane against the CPU oracle before a GPU sees them.
//
// Why.  A record's codes are a serial chain, and the kernels that gave a whole record to one lane / one wave / one group of waves
// ended when their longest record did (rounds 1-3: a lane with 2 047 successors was 1.2 of k_parse_list's 1.4 ms; the one record of
// 347 500 successors most of the giant kernel).  Here the unit of work is bounded: the residual section of every record of the class
// is cut at the multiples of SEG_BITS of the stream's bit positions, and every piece is one work item of about a hundred codes.
//
//   struct   one lane per record: reference, copy blocks (only their count of copied ids), intervals -> arena entries
//            {left, pstart, rank, len}; leaves where the residual section starts and how many residuals it holds   (BVG:1048-1096)
//   A1       one lane per segment: decodes the codes that start in its piece.  Segment 0 of a record starts on a true codeword
//            boundary; the others start at the grid point -- usually not a boundary: zeta codes re-synchronise after a few codewords,
//            so the lane's END is, with overwhelming probability, a true boundary, while its count and gap sum include a false prefix
//   A2       one lane per segment: takes its predecessor's end as its true start and walks the true chain and its own false chain in
//            lock step (always the one that is behind) until they meet; only the difference of the two prefixes is applied to (count,
//            sum).  If they meet inside the piece, the end found by A1 was a true boundary -- by induction from segment 0 every start
//            is then exact.  If not (about one piece in 2 000 of 1 024 bits), the lane has followed the true chain to the end of
//            its piece and knows where the NEXT piece really starts: it says so in a list
//   fix      one lane per entry of that list: the same walk for the next piece with its true start, and on along the record for
//            as long as chains keep missing each other.  Best effort: what proves the starts right is B, which checks that the
//            codes it decodes end exactly where the next piece is said to start
//   scan     counts and sums -> index of the segment's first residual in its record, value of the residual before it
//   B        one WAVE per segment, one lane per residual and per interval: no codeword is decoded a second time -- A1 left the running
//            sums of its chain in a scratch cell per piece, A2 the few sums of the true chain before the meeting point; a residual is
//            base + running sum (BVG:954, :966).  Residuals and the intervals they pass are ranked against each other by binary
//            searches in LDS: residual j goes to out[j + #(interval ids below it)], interval i to out[pstart + #(residuals below it)]
//            (MergedIntIterator.java:50-74, IntIntervalSequenceIterator.java:64-78); neighbouring lanes write neighbouring ids
//
// Same contract as the other parse kernels: the extras (intervals merged with residuals) of node x end up in row[copied..d); the
// copy pass fills in the rest.  Default codings only (gamma / unary / zeta_k).  Anything unusual -- a codeword longer than 64 bits, a
// value that does not fit 32, chains that do not meet, a count that does not add up, a residual inside an interval (a malformed
// file: MergedIntIterator emits equal heads once) -- is not handled here: the record is flagged, appended to a list and decoded
// afterwards by the cooperative one-wave kernel (k_parse_big), which also owns all error reporting.  The bodies below have no error
// plumbing; they only have to be memory-safe on garbage.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SG_D __device__ __forceinline__
#define SG_ANY(p) __any(p)
#else
#define SG_D inline
#if defined(SG_ANY_ALWAYS) // model: as if some other lane of the wave always needed it -- whatever SG_ANY guards must be harmless at any time
#define SG_ANY(p) ((void)(p), true)
#else
#define SG_ANY(p) (p)
#endif
#endif

namespace bvsg {

#ifndef SEG_BITS_LOG2_
#define SEG_BITS_LOG2_ 10
#endif
constexpr int SEG_BITS_LOG2 = SEG_BITS_LOG2_;
constexpr uint32_t SEG_BITS = 1u << SEG_BITS_LOG2; // a piece of stream: [c * SEG_BITS, (c + 1) * SEG_BITS), "cell" c of the grid; a multiple of 128 bits (16-byte loads)
constexpr int WIN_WORDS = 16;                      // a lane's window of the stream in LDS
constexpr int RING = 8;                            // intervals a lane of B keeps at hand (2 words each)
constexpr int FIX_MAX = 64;                        // pieces one lane of the fix kernel follows a run of missed meetings for

struct SegGraph { // what the bodies need of bv::GraphDev
	const uint32_t *bits;   // .graph bytes as big-endian words (byte-swapped on load), padded with >= 8 zero words
	uint64_t nwords;
	const int64_t *offsets;
	int32_t W, minInt, zetaK;
};
struct SegIv { int32_t left, pstart, rank, len; };                                  // = bv::IvEntry (bv_coop.hpp)
struct RecDesc { int64_t rpos; int32_t slot, nres, copied, nIv, flags, ivArcs; };   // one record of the class (32 bytes)
enum { RF_FALLBACK = 1, RF_SKIP = 2 }; // flagged: the cooperative kernel decodes it; not this pipeline's record at all
// positions inside a piece are relative to the start of its cell (c << SEG_BITS_LOG2): they fit 32 bits
struct SegA1 { uint32_t outRel, cnt, sum, badIdx; };  // A1: where the chain that started at the piece's nominal start leaves it; its codes; their sum; index of its first "codeword" of more than 64 bits (~0: none)
// A2 / fix: the piece's true start; true count and sum; where the true chain ends (mode 1 only).  Residual t of the piece is
// base + (t < cb ? fix[t] : cell[t - cb + ca] + delta): cb sums of the true chain before it joins A1's chain at that chain's code ca.
// mode 0: as said; 1: the chains did not meet within FIX_CODES codes -- the lane decoded the whole piece again and the cell holds the
// true chain (ca = cb = delta = 0); 2: a codeword this decoder does not take; 3: not known yet (the fix pass follows the true chain).
// Bit 2 (SEG_REWRITTEN): the cell no longer holds A1's chain
struct SegFin { uint32_t inRel, cnt, sum, tRel, ca, cb, delta, mode; };
constexpr uint32_t FIX_CODES = 32, SEG_REWRITTEN = 4;

#if defined(__HIPCC__)
#define SG_ASSERT(c) ((void)0)
#define SG_LIKELY(c) __builtin_expect(!!(c), 1)
#define SG_UNLIKELY(c) __builtin_expect(!!(c), 0)
#else
#include <assert.h>
#define SG_ASSERT(c) assert(c)
#define SG_LIKELY(c) (c)
#define SG_UNLIKELY(c) (c)
#endif

SG_D uint32_t clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__clz((int)x); // 32 for 0
#else
	return x ? (uint32_t)__builtin_clz(x) : 32u;
#endif
}
SG_D uint32_t clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__clzll((long long)x);
#else
	return x ? (uint32_t)__builtin_clzll(x) : 64u;
#endif
}
SG_D int32_t zigzag32(uint32_t v) { return (int32_t)(v >> 1) ^ -(int32_t)(v & 1); } // Fast.nat2int, truncated to a Java int
SG_D uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
SG_D uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// A lane's window of WIN_WORDS words of the stream: word k lives at p[k * STRIDE] (one LDS bank per lane on the device, STRIDE =
// threads of the block; a plain array in the model).  Refilled with four 16-byte loads that are all in flight together.  Cursors
// are bit offsets from window word 0.  A codeword may be decoded at cursor q when q <= Q_OK (it reads words q / 32 .. q / 32 + 2);
// a decoder advances its cursor by at most 65, and every decode below is preceded by a refill check: the reads stay inside the
// window whatever the stream holds.
constexpr uint32_t Q_OK = (WIN_WORDS - 3) * 32 + 31;
template <int STRIDE> struct Win {
	uint32_t *p;
	const uint32_t *bits;
	uint64_t w0;    // absolute index of window word 0 (a multiple of 4)
	uint64_t vlast; // first word of the last 16-byte vector worth fetching: past it the last vector is simply read again (branch-free)
	SG_D uint32_t word(uint32_t j) const { SG_ASSERT(j < (uint32_t)WIN_WORDS); return p[j * STRIDE]; }
	SG_D void fill() {
#if defined(__HIP_DEVICE_COMPILE__)
		uint4 v[WIN_WORDS / 4];
#pragma unroll
		for (int k = 0; k < WIN_WORDS / 4; k++) v[k] = *(const uint4 *)(bits + umin64(w0 + 4 * k, vlast));
#pragma unroll
		for (int k = 0; k < WIN_WORDS / 4; k++) {
			p[(4 * k + 0) * STRIDE] = __builtin_bswap32(v[k].x); p[(4 * k + 1) * STRIDE] = __builtin_bswap32(v[k].y);
			p[(4 * k + 2) * STRIDE] = __builtin_bswap32(v[k].z); p[(4 * k + 3) * STRIDE] = __builtin_bswap32(v[k].w);
		}
#else
		for (int k = 0; k < WIN_WORDS / 4; k++) {
			uint32_t t[4];
			memcpy(t, bits + umin64(w0 + 4 * k, vlast), 16);
			for (int e = 0; e < 4; e++) p[(4 * k + e) * STRIDE] = __builtin_bswap32(t[e]);
		}
#endif
	}
	SG_D void init(const SegGraph &g, uint32_t *col, uint64_t lastBit) { // lastBit: no codeword that matters starts after it
		p = col; bits = g.bits;
		vlast = umin64((((lastBit + 64) >> 5) + 2) & ~(uint64_t)3, (g.nwords + 4) & ~(uint64_t)3);
	}
	SG_D uint32_t seek(uint64_t pos) { w0 = (pos >> 5) & ~(uint64_t)3; fill(); return (uint32_t)(pos - (w0 << 5)); }
	SG_D uint64_t pos(uint32_t q) const { return (w0 << 5) + q; }
	SG_D uint32_t slide(uint32_t qmin) { const uint32_t adv = (qmin >> 5) & ~3u; w0 += adv; fill(); return adv << 5; } // returns the bits every cursor moves down by
	SG_D uint32_t peek32(uint32_t q) const {
		const uint32_t j = q >> 5;
		const uint64_t ab = ((uint64_t)word(j) << 32) | word(j + 1);
		return (uint32_t)((ab << (q & 31u)) >> 32);
	}
	SG_D uint64_t peek64(uint32_t q) const {
		const uint32_t j = q >> 5, sh = q & 31u;
		const uint64_t ab = ((uint64_t)word(j) << 32) | word(j + 1);
		return sh ? (ab << sh) | ((uint64_t)word(j + 2) >> (32u - sh)) : ab;
	}
	// The decoders: the common codewords (gamma < 2^16, zeta_3 < 2^21, unary < 32) from one 32-bit peek, a dozen instructions; longer
	// ones out of line from a 64-bit peek; `bad` when a codeword does not fit 64 bits or its value 32 (then q advances by one bit).
	SG_D uint32_t gamma_slow(uint32_t &q, bool &bad) const {
		const uint64_t W64 = peek64(q);
		const uint32_t m = clz64(W64);
		if (m > 31) { bad = true; q += 1; return 0; }
		q += 2 * m + 1;
		return (uint32_t)(((W64 << m) >> (63u - m)) - 1);
	}
	SG_D uint32_t gamma(uint32_t &q, bool &bad) const { // the value (x, not x + 1)
		const uint32_t W = peek32(q);
		if (SG_LIKELY(W >= (1u << 16))) { const uint32_t h = clz32(W); q += 2 * h + 1; return (W >> (31u - 2 * h)) - 1; }
		return gamma_slow(q, bad);
	}
	SG_D uint32_t unary(uint32_t &q, bool &bad) const {
		uint32_t z = clz32(peek32(q));
		if (SG_UNLIKELY(z >= 32)) { z = clz64(peek64(q)); if (z >= 64) { bad = true; z = 0; } }
		q += z + 1;
		return z;
	}
	SG_D uint32_t zeta_slow(uint32_t &q, uint32_t k, bool &bad) const {
		const uint64_t W64 = peek64(q);
		const uint32_t h = clz64(W64);
		const uint32_t nb = k * h + k - 1;
		if (h + 2 + nb > 64u || k * h > 32u) { bad = true; q += 1; return 0; }
		if (nb == 0) { q += 1; return 0; } // zeta_1, h = 0: the codeword "1" has no payload and means 0
		const uint64_t mm = (W64 << (h + 1)) >> (63u - nb); // nb payload bits plus the extra bit of a long codeword
		const uint64_t m = mm >> 1, left = (uint64_t)1 << (k * h);
		const bool lng = m >= left;
		const uint64_t vv = lng ? mm - 1 : m + left - 1;
		if (vv > 0xffffffffull) bad = true;
		q += h + 1 + nb + (lng ? 1u : 0u);
		return (uint32_t)vv;
	}
	template <int K> SG_D uint32_t zeta(uint32_t &q, uint32_t krt, bool &bad) const { // K = 3 folded in; K = 0: k at run time (1 <= k <= 16)
		const uint32_t W = peek32(q);
		if (K == 3) {
			if (SG_LIKELY(W >= (1u << 25))) { // h <= 6: at most 28 bits
				const uint32_t h = clz32(W), h3 = 3 * h;
				const uint32_t mm = (W << (h + 1)) >> (29u - h3); // 3h + 2 payload bits plus the extra bit of a long codeword
				const uint32_t m = mm >> 1, left = 1u << h3;
				const bool lng = m >= left;
				q += 4 * h + 3 + (lng ? 1u : 0u);
				return lng ? mm - 1 : m + left - 1;
			}
			return zeta_slow(q, 3, bad);
		}
		const uint32_t k = krt, h = clz32(W), nb = k * h + k - 1;
		if (SG_LIKELY(h + 2 + nb <= 32u && nb != 0)) {
			const uint32_t mm = (W << (h + 1)) >> (31u - nb);
			const uint32_t m = mm >> 1, left = 1u << (k * h);
			const bool lng = m >= left;
			q += h + 1 + nb + (lng ? 1u : 0u);
			return lng ? mm - 1 : m + left - 1;
		}
		return zeta_slow(q, k, bad);
	}
};

// wave-synchronised refill: if ANY lane of the wave is about to run out of window, ALL (active) lanes move theirs up to their cursor
// (left to themselves the lanes would each stall the whole wave for a memory round trip at a different iteration)
#define SG_REFILL(w, q) do { if (SG_ANY((q) > Q_OK)) q -= w.slide(q); } while (0)

// segments of a record: the cells of the grid that its residual section [rpos, recEnd) touches
SG_D int32_t seg_count(const RecDesc &r, uint64_t recEnd) {
	if ((r.flags & (RF_FALLBACK | RF_SKIP)) || r.nres <= 0) return 0;
	const uint64_t c0 = (uint64_t)r.rpos >> SEG_BITS_LOG2, c1 = (recEnd - 1) >> SEG_BITS_LOG2;
	const uint64_t n = c1 - c0 + 1;
	return n > 0x3fffffffull ? 0 : (int32_t)n;
}
// segment i of the record: its cell's first bit, and where the codes that are its own start: [cell + startRel, cell + endRel)
SG_D void seg_span(const RecDesc &r, uint64_t recEnd, int32_t i, uint64_t &cell, uint32_t &startRel, uint32_t &endRel) {
	const uint64_t c = ((uint64_t)r.rpos >> SEG_BITS_LOG2) + (uint64_t)i;
	cell = c << SEG_BITS_LOG2;
	startRel = i == 0 ? (uint32_t)((uint64_t)r.rpos - cell) : 0u;
	endRel = (uint32_t)(umin64(cell + SEG_BITS, recEnd) - cell);
}

// ------------------------------------------------------------------------------------------------ A1
// A run of residual codes from cursor q up to qend (cursors relative to the window), the running sum of their contributions (gap + 1
// each; the first code of a record is the zig-zag value relative to x, BVG:954) written to cell[0 ..) after every code, 16 bytes at a
// time.  Sums are Java ints: they wrap.  A chain that starts off a codeword boundary reads garbage until it locks on, and garbage can
// look like a codeword of more than 64 bits: such a "codeword" is stepped over as one bit and its index reported -- A2 knows whether
// it lay before the point where the true chain joins this one (harmless) or behind it (the record is flagged).
template <int ZK, int STRIDE>
SG_D void decode_run(const SegGraph &g, Win<STRIDE> &w, uint32_t &q, uint32_t qend, bool firstOfRecord, int32_t x, int32_t *cell, uint32_t cap, uint32_t &cntOut, uint32_t &sumOut, uint32_t &badIdx) {
	uint32_t cnt = 0, sum = 0;
	uint32_t o1 = 0, o2 = 0, o3 = 0;
	badIdx = ~0u;
	if (firstOfRecord && q < qend) {
		bool bad = false;
		const uint32_t v = w.template zeta<ZK>(q, (uint32_t)g.zetaK, bad);
		if (bad) badIdx = 0;
		sum = (uint32_t)x + (uint32_t)zigzag32(v);
		o3 = sum;
		cnt = 1;
	}
	while (q < qend) {
		if (SG_ANY(q > Q_OK)) { const uint32_t dn = w.slide(q); q -= dn; qend -= dn; }
		bool bad = false;
		const uint32_t v = w.template zeta<ZK>(q, (uint32_t)g.zetaK, bad);
		if (SG_UNLIKELY(bad)) badIdx = umin32(badIdx, cnt);
		sum += v + 1u;
		const uint32_t o0 = o1; o1 = o2; o2 = o3; o3 = sum;
		cnt++;
		if (cell && (cnt & 3u) == 0 && cnt <= cap) { // (cell is 16-byte aligned)
#if defined(__HIP_DEVICE_COMPILE__)
			*(int4 *)(cell + cnt - 4) = int4{ (int32_t)o0, (int32_t)o1, (int32_t)o2, (int32_t)o3 };
#else
			cell[cnt - 4] = (int32_t)o0; cell[cnt - 3] = (int32_t)o1; cell[cnt - 2] = (int32_t)o2; cell[cnt - 1] = (int32_t)o3;
#endif
		}
	}
	if (!cell) { if (cnt > cap) badIdx = 0; }
	else if (cnt <= cap) {
		const uint32_t m = cnt & 3u;
		if (m >= 3) cell[cnt - 3] = (int32_t)o1;
		if (m >= 2) cell[cnt - 2] = (int32_t)o2;
		if (m >= 1) cell[cnt - 1] = (int32_t)o3;
	} else badIdx = 0; // (more codes than the shortest codeword allows: cannot happen)
	cntOut = cnt; sumOut = sum;
}

// The codes that start in [startRel, endRel) of the cell, from startRel.
template <int ZK, int STRIDE>
SG_D void seg_a1(const SegGraph &g, uint32_t *col, int32_t x, uint64_t cellBit, uint32_t startRel, uint32_t endRel, bool firstOfRecord, int32_t *cell, uint32_t cap, SegA1 &o) {
	Win<STRIDE> w;
	w.init(g, col, cellBit + endRel);
	uint32_t q = w.seek(cellBit + startRel);
	const uint32_t qend = q + (endRel - startRel);
	decode_run<ZK, STRIDE>(g, w, q, qend, firstOfRecord, x, cell, cap, o.cnt, o.sum, o.badIdx);
	o.outRel = (uint32_t)(w.pos(q) - cellBit);
}

// ------------------------------------------------------------------------------------------------ A2
// A piece that does not start its record: A1 started at the cell's first bit; the true chain enters the piece at inRel (the end of the
// piece before, minus SEG_BITS).  Walks both chains in lock step (always the one that is behind) until they meet, keeping the running
// sums of the true chain's codes in fix[]; see SegFin for what it leaves.
template <int ZK, int STRIDE>
SG_D void seg_a2(const SegGraph &g, uint32_t *col, uint64_t cellBit, uint32_t inRel, uint32_t endRel, const SegA1 &a1, int32_t *cell, uint32_t cap, uint32_t *fix, bool cellRewritten, bool follow, SegFin &o) {
	// cellRewritten (the fix pass, a piece it or A2 visited before): the cell no longer holds A1's chain -- the whole piece again, whatever the chains do
	o = SegFin{ inRel, a1.cnt, a1.sum, 0, 0, 0, 0, cellRewritten ? SEG_REWRITTEN : 0u };
	if (inRel == 0 && !cellRewritten) { if (a1.badIdx != ~0u) o.mode = 2; return; }
	if (inRel > 128) { o.mode |= 2; return; } // (a codeword of the piece before cannot reach that far)
	Win<STRIDE> w;
	w.init(g, col, cellBit + endRel);
	uint32_t qa = w.seek(cellBit);  // the false chain (a cell starts on a multiple of 128 bits: qa = 0)
	uint32_t qb = qa + inRel;       // the true chain
	uint32_t qend = qa + endRel;
	uint32_t ca = 0, cb = 0, sa = 0, sb = 0;
	bool met = !cellRewritten, badB = false;
	while (qa != qb && !cellRewritten) {
		const bool aBehind = qa < qb;
		uint32_t q = aBehind ? qa : qb;
		if (q >= qend || (!aBehind && cb == FIX_CODES)) { met = false; break; } // the chain that is behind has left the piece (and so has the other) -- or this takes too long
		if (SG_ANY((aBehind ? qb : qa) > Q_OK)) { const uint32_t dn = w.slide(q); q -= dn; qa -= dn; qb -= dn; qend -= dn; }
		bool bad = false;
		const uint32_t inc = w.template zeta<ZK>(q, (uint32_t)g.zetaK, bad) + 1u;
		if (aBehind) { qa = q; ca++; sa += inc; }
		else { qb = q; sb += inc; if (fix) fix[cb] = sb; cb++; badB |= bad; }
	}
	if (badB) { o.mode |= 2; return; }
	if (met) {
		if (a1.badIdx != ~0u && a1.badIdx >= ca) { o.mode = 2; return; } // the codeword A1 could not take lies on the true chain
		o.cnt = a1.cnt - ca + cb; o.sum = a1.sum - sa + sb;
		o.ca = ca; o.cb = cb; o.delta = sb - sa;
		return;
	}
	// no meeting point in sight.  !follow (A2, where a lane that went on would hold up its whole wave): left to the fix pass (mode 3)
	if (!follow) { o.mode = 3; return; }
	// the whole piece again, from its true start (into the cell, if one is kept)
	uint32_t q = w.seek(cellBit + inRel), badIdx;
	qend = q + (endRel > inRel ? endRel - inRel : 0u);
	decode_run<ZK, STRIDE>(g, w, q, qend, false, 0, cell, cap, o.cnt, o.sum, badIdx);
	o.tRel = (uint32_t)(w.pos(q) - cellBit);
	o.mode = (badIdx != ~0u ? 2u : 1u) | SEG_REWRITTEN;
}

// ------------------------------------------------------------------------------------------------ B
// Decodes the cnt codes of a piece from its true start and stores every residual at its place among the record's extras: residual j
// (value r) goes to out[j + #(interval ids below r)].  v0 = the residual before the piece's first one, j0 = its index + 1.
// Intervals that the piece's residuals pass learn their rank (= residuals before them).  endRel: where the last code ended (the
// caller checks it against the start of the next piece).  false: the record must be flagged.
//
// What a wave executes per iteration is what counts (all of these kernels are bound by instruction issue): the loop proper is the
// decode, one range test and one store.  The lane's next intervals wait in a ring in LDS -- (left, ids up to its end) -- that is
// topped up, four entries at a time and without a branch per entry, whenever the wave refills its stream windows anyway: no loads of
// their own in the loop (a load waits for every store before it: the GPU counts both in one counter).  A lane that runs out of ring
// all the same takes its next interval straight from the arena.
template <int STRIDE> struct IvRing {
	uint32_t *ring; const SegIv *iv; int32_t nIv, idx, loaded; // intervals [idx, loaded) are in the ring, entry k at slot k & (RING - 1)
	SG_D void top_up() { // four more, if this lane has the room and the record has them
		if (RING - (loaded - idx) >= 4 && loaded < nIv) {
			const int32_t last = nIv - 1;
#if defined(__HIP_DEVICE_COMPILE__)
			int4 e[4];
#pragma unroll
			for (int k = 0; k < 4; k++) e[k] = *(const int4 *)(iv + (loaded + k < last ? loaded + k : last)); // (past the record's last interval: read again, never used)
#pragma unroll
			for (int k = 0; k < 4; k++) { const uint32_t s = (uint32_t)(loaded + k) & (RING - 1); ring[(2 * s) * STRIDE] = (uint32_t)e[k].x; ring[(2 * s + 1) * STRIDE] = (uint32_t)(e[k].y + e[k].w); }
#else
			for (int k = 0; k < 4; k++) { const SegIv e = iv[loaded + k < last ? loaded + k : last]; const uint32_t s = (uint32_t)(loaded + k) & (RING - 1); ring[(2 * s) * STRIDE] = (uint32_t)e.left; ring[(2 * s + 1) * STRIDE] = (uint32_t)(e.pstart + e.len); }
#endif
			loaded = loaded + 4 < nIv ? loaded + 4 : nIv;
		}
	}
	SG_D void get(int32_t &left, int32_t &cum) { // interval idx (idx < nIv)
		if (idx < loaded) { const uint32_t s = (uint32_t)idx & (RING - 1); left = (int32_t)ring[(2 * s) * STRIDE]; cum = (int32_t)ring[(2 * s + 1) * STRIDE]; }
		else { const SegIv e = iv[idx]; left = e.left; cum = e.pstart + e.len; loaded = idx; } // (the ring is empty: straight from the arena)
	}
};
template <int ZK, int STRIDE>
SG_D bool seg_b(const SegGraph &g, uint32_t *col, uint32_t *ring, int32_t x, uint64_t cell, uint32_t inRel, uint32_t cnt, uint32_t j0, int32_t v0, bool firstOfRecord,
                int32_t *out, int32_t extra, SegIv *iv, int32_t nIv, uint32_t &endRel) {
#if defined(SG_DBG_NOIV)
	nIv = 0;
#endif
	Win<STRIDE> w;
	w.init(g, col, cell + SEG_BITS);
	uint32_t q = w.seek(cell + inRel);
	bool bad = false;
	// intervals [0, idx) lie below v0 (passed by the pieces before): a binary search in the record's arena slice
	IvRing<STRIDE> R{ ring, iv, nIv, 0, 0 };
	if (!firstOfRecord && nIv > 0) {
		int32_t lo = 0, hi = nIv;
		while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (iv[mid].left <= v0) lo = mid + 1; else hi = mid; }
		R.idx = lo;
	}
	R.loaded = R.idx;
	// between the end of the last interval passed and the start of the next one a residual simply goes to out[j + before]
	int32_t before = 0, prevEnd = (int32_t)0x80000000, nl = 0x7fffffff, ncum = 0; // ids of the intervals passed; [prevEnd, nl): the free stretch; ids up to the end of the next interval
	if (R.idx > 0) { const SegIv e = iv[R.idx - 1]; before = e.pstart + e.len; prevEnd = (int32_t)((uint32_t)e.left + (uint32_t)e.len); }
	R.top_up(); R.top_up();
	if (R.idx < nIv) R.get(nl, ncum);
	uint32_t span = (uint32_t)nl - (uint32_t)prevEnd;
	uint32_t j = j0, val = (uint32_t)v0;
	const uint32_t xtr = (uint32_t)extra;
	for (uint32_t t = 0; t < cnt; t++) {
		if (SG_ANY(q > Q_OK)) { q -= w.slide(q); R.top_up(); }
		const uint32_t v = w.template zeta<ZK>(q, (uint32_t)g.zetaK, bad);
		val = (firstOfRecord && t == 0) ? (uint32_t)x + (uint32_t)zigzag32(v) : val + v + 1u; // BVG:954, :966
		if (SG_UNLIKELY(val - (uint32_t)prevEnd >= span)) { // not in the free stretch: the residual passes intervals -- or sits inside one
			const int32_t sv = (int32_t)val;
			if (sv < prevEnd) bad = true; // inside an interval: equal heads are emitted once (MergedIntIterator.java:69-72) -- not here
			while (R.idx < nIv && nl < sv) { // the residual passes interval idx: j residuals precede it
#if !defined(SG_DBG_NORANK)
				iv[R.idx].rank = (int32_t)j;
#endif
				prevEnd = (int32_t)((uint32_t)nl + (uint32_t)(ncum - before));
				before = ncum;
				R.idx++;
				if (R.idx < nIv) { R.get(nl, ncum); if (nl < prevEnd) bad = true; }
				else nl = 0x7fffffff;
			}
			if ((R.idx < nIv && nl == sv) || sv < prevEnd) bad = true;
			span = (uint32_t)nl - (uint32_t)prevEnd;
		}
		const uint32_t p = j + (uint32_t)before;
#if defined(SG_DBG_NOSTORE)
		if (p >= xtr) bad = true;
#else
		if (p < xtr) out[p] = (int32_t)val; else bad = true;
#endif
		j++;
	}
	endRel = (uint32_t)(w.pos(q) - cell);
	return !bad;
}

// ------------------------------------------------------------------------------------------------ expand
SG_D void expand_interval(const SegIv e, int32_t nres, int32_t *out, int32_t extra) {
	const int64_t p = (int64_t)e.pstart + (e.rank < 0 ? nres : e.rank);
	for (int32_t t = 0; t < e.len; t++) if (p + t < (int64_t)extra) out[p + t] = (int32_t)((uint32_t)e.left + (uint32_t)t);
}

} // namespace bvsg
