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
//            is then exact; if not, the record is flagged
//   scan     counts and sums -> index of the segment's first residual in its record, value of the residual before it
//   B        one lane per segment: decodes its codes again, adds up the gaps (BVG:954, :966) and stores every residual at its final
//            place among the record's extras, walking the record's interval list alongside (a small ring in LDS) to count the interval
//            ids that precede it -- which also tells every interval its rank                      (MergedIntIterator.java:50-74)
//   expand   one lane per interval: left .. left + len - 1 at pstart + rank                (IntIntervalSequenceIterator.java:64-78)
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
#define SG_ANY(p) (p)
#endif

namespace bvsg {

#ifndef SEG_BITS_LOG2_
#define SEG_BITS_LOG2_ 11
#endif
constexpr int SEG_BITS_LOG2 = SEG_BITS_LOG2_;
constexpr uint64_t SEG_BITS = (uint64_t)1 << SEG_BITS_LOG2; // a piece of stream; its start is a multiple of 128 bits (16-byte loads)
constexpr int WIN_WORDS = 16;                               // a lane's window of the stream in LDS
constexpr int RING = 8;                                     // intervals a lane of B keeps at hand (2 words each)

struct SegGraph { // what the bodies need of bv::GraphDev
	const uint32_t *bits;   // .graph bytes as big-endian words (byte-swapped on load), padded with >= 8 zero words
	uint64_t nwords;
	const int64_t *offsets;
	int32_t W, minInt, zetaK;
};
struct SegIv { int32_t left, pstart, rank, len; };                                  // = bv::IvEntry (bv_coop.hpp)
struct RecDesc { int64_t rpos; int32_t slot, nres, copied, nIv, flags, ivArcs; };   // one record of the class (32 bytes)
enum { RF_FALLBACK = 1 };

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

template <int STRIDE> struct Col { // word k of a lane's column: one LDS bank per lane on the device (STRIDE = threads of the block), plain array in the model
	uint32_t *p;
	SG_D uint32_t get(uint32_t k) const { return p[k * STRIDE]; }
	SG_D void set(uint32_t k, uint32_t v) const { p[k * STRIDE] = v; }
};

// A lane's window of WIN_WORDS words of the stream, refilled with four 16-byte loads that are all in flight together.
template <int STRIDE> struct Win {
	Col<STRIDE> c;
	const uint32_t *bits;
	uint64_t w0;    // absolute index of window word 0 (a multiple of 4)
	uint64_t vlast; // first word of the last 16-byte vector worth fetching: past it the last vector is simply read again (branch-free)
	SG_D void fill() {
#if defined(__HIP_DEVICE_COMPILE__)
		uint4 v[WIN_WORDS / 4];
#pragma unroll
		for (int k = 0; k < WIN_WORDS / 4; k++) v[k] = *(const uint4 *)(bits + umin64(w0 + 4 * k, vlast));
#pragma unroll
		for (int k = 0; k < WIN_WORDS / 4; k++) {
			c.set(4 * k + 0, __builtin_bswap32(v[k].x)); c.set(4 * k + 1, __builtin_bswap32(v[k].y));
			c.set(4 * k + 2, __builtin_bswap32(v[k].z)); c.set(4 * k + 3, __builtin_bswap32(v[k].w));
		}
#else
		for (int k = 0; k < WIN_WORDS / 4; k++) {
			uint32_t t[4];
			memcpy(t, bits + umin64(w0 + 4 * k, vlast), 16);
			for (int e = 0; e < 4; e++) c.set(4 * k + e, __builtin_bswap32(t[e]));
		}
#endif
	}
	SG_D void init(const SegGraph &g, Col<STRIDE> col, uint64_t lastBit) { // lastBit: no codeword that matters starts after it
		c = col; bits = g.bits;
		vlast = umin64((((lastBit + 64) >> 5) + 2) & ~(uint64_t)3, (g.nwords + 4) & ~(uint64_t)3);
	}
	SG_D uint32_t seek(uint64_t pos) { w0 = (pos >> 5) & ~(uint64_t)3; fill(); return (uint32_t)(pos - (w0 << 5)); }
	SG_D uint64_t pos(uint32_t q) const { return (w0 << 5) + q; }
	// a cursor at q may decode one codeword of up to 64 bits when q's word index is <= WIN_WORDS - 3
	SG_D static bool low(uint32_t q) { return (q >> 5) + 3 > (uint32_t)WIN_WORDS; }
	SG_D uint32_t slide(uint32_t qmin) { const uint32_t adv = (qmin >> 5) & ~3u; w0 += adv; fill(); return adv << 5; } // returns the bits every cursor moves down by
	SG_D uint32_t peek32(uint32_t q) const {
		const uint32_t j = (q >> 5) & (WIN_WORDS - 1), sh = q & 31u; // (masked: memory-safe on garbage)
		const uint64_t ab = ((uint64_t)c.get(j) << 32) | c.get((j + 1) & (WIN_WORDS - 1));
		return (uint32_t)((ab << sh) >> 32);
	}
	SG_D uint64_t peek64(uint32_t q) const {
		const uint32_t j = (q >> 5) & (WIN_WORDS - 1), sh = q & 31u;
		const uint64_t ab = ((uint64_t)c.get(j) << 32) | c.get((j + 1) & (WIN_WORDS - 1));
		return sh ? (ab << sh) | ((uint64_t)c.get((j + 2) & (WIN_WORDS - 1)) >> (32u - sh)) : ab;
	}
	// The decoders: the common codewords (gamma < 2^16, zeta_3 < 2^21, unary < 32) from one 32-bit peek, longer ones from a 64-bit peek
	// behind one rarely taken branch; `bad` when a codeword does not fit 64 bits or its value 32.  q advances by at most 64.
	SG_D uint32_t gamma(uint32_t &q, bool &bad) const { // the value (x, not x + 1)
		const uint32_t W = peek32(q);
		const uint32_t h = clz32(W);
		uint32_t len = 2 * h + 1, v = (W >> ((31u - 2 * h) & 31u)) - 1;
		if (__builtin_expect(h >= 16, 0)) {
			const uint64_t W64 = peek64(q);
			const uint32_t m = clz64(W64);
			if (m > 31) { bad = true; len = 1; v = 0; }
			else { len = 2 * m + 1; const uint64_t vv = ((W64 << m) >> (63u - m)) - 1; v = (uint32_t)vv; }
		}
		q += len;
		return v;
	}
	SG_D uint32_t unary(uint32_t &q, bool &bad) const {
		uint32_t z = clz32(peek32(q));
		if (__builtin_expect(z >= 32, 0)) { z = clz64(peek64(q)); if (z >= 64) { bad = true; z = 0; } }
		q += z + 1;
		return z;
	}
	template <int K> SG_D uint32_t zeta(uint32_t &q, uint32_t krt, bool &bad) const { // K = 3 folded in; K = 0: k at run time (1 <= k <= 16)
		const uint32_t k = K ? (uint32_t)K : krt;
		const uint32_t W = peek32(q);
		const uint32_t h = clz32(W);
		const uint32_t nb = k * h + k - 1;                 // payload bits of the short codeword
		const bool fits = h + 2 + nb <= 32u;
		const uint32_t mm = nb ? (W << ((h + 1) & 31u)) >> ((31u - nb) & 31u) : 0u; // nb payload bits plus the extra bit of a long codeword (shifts masked: only used when it fits)
		const uint32_t m = mm >> 1, left = 1u << ((k * h) & 31u);
		const bool lng = nb != 0 && m >= left;             // (zeta_1, h = 0: the codeword "1" has no payload and means 0)
		uint32_t v = lng ? mm - 1 : m + left - 1;
		uint32_t len = h + 1 + nb + (lng ? 1u : 0u);
		if (__builtin_expect(!fits, 0)) {
			const uint64_t W64 = peek64(q);
			const uint32_t h2 = clz64(W64);
			const uint32_t nb2 = k * h2 + k - 1;
			if (h2 + 2 + nb2 > 64u || k * h2 > 32u) { bad = true; v = 0; len = 1; }
			else {
				const uint64_t mm2 = (W64 << (h2 + 1)) >> (63u - nb2);
				const uint64_t m2 = mm2 >> 1, left2 = (uint64_t)1 << (k * h2);
				const bool lng2 = m2 >= left2;
				const uint64_t vv = lng2 ? mm2 - 1 : m2 + left2 - 1;
				if (vv > 0xffffffffull) bad = true;
				v = (uint32_t)vv;
				len = h2 + 1 + nb2 + (lng2 ? 1u : 0u);
			}
		}
		q += len;
		return v;
	}
};

// wave-synchronised refill: if ANY lane of the wave is about to run out of window, ALL (active) lanes move theirs up to their cursor
// (left to themselves the lanes would each stall the whole wave for a memory round trip at a different iteration)
#define SG_REFILL(w, q) do { if (SG_ANY(w.low(q))) q -= w.slide(q); } while (0)

// ------------------------------------------------------------------------------------------------ struct
// The gamma-coded front of record x (outdegree d; referent's outdegree dref if it has a reference): BVG:1048-1096.
template <int STRIDE>
SG_D void struct_lane(const SegGraph &g, Col<STRIDE> col, int32_t x, int32_t d, bool hasRef, int64_t dref, SegIv *iv, RecDesc &o) {
	Win<STRIDE> w;
	const uint64_t recEnd = (uint64_t)g.offsets[x + 1];
	w.init(g, col, recEnd);
	uint32_t q = w.seek((uint64_t)g.offsets[x]);
	bool bad = false;
	(void)w.gamma(q, bad);               // outdegree (known from k_headers)
	if (g.W > 0) (void)w.unary(q, bad);  // reference
	int64_t copied = 0;
	if (hasRef) { // BVG:1058-1071
		SG_REFILL(w, q);
		const uint32_t bc = w.gamma(q, bad);
		int64_t total = 0;
		if ((int64_t)bc > dref + 1) bad = true;
		for (uint32_t b = 0; b < bc && !bad; b++) {
			SG_REFILL(w, q);
			const int64_t code = (int64_t)w.gamma(q, bad);
			if (code > dref - total) { bad = true; break; } // (a code of a malformed stream is rejected before it reaches a sum)
			const int64_t len = code + (b ? 1 : 0);
			if (total + len > dref) { bad = true; break; }
			total += len;
			if (!(b & 1)) copied += len;
		}
		if (!(bc & 1)) copied += dref - total;
	}
	const int64_t extra = (int64_t)d - copied;
	if (extra < 0 || copied < 0) bad = true;
	int64_t nIv = 0, ivArcs = 0;
	if (!bad && extra > 0 && g.minInt != 0) { // BVG:1073-1096
		SG_REFILL(w, q);
		nIv = (int64_t)w.gamma(q, bad);
		if (nIv > extra / g.minInt) { bad = true; nIv = 0; }
		int32_t prevEnd = 0;
		for (int64_t i = 0; i < nIv && !bad; i++) {
			SG_REFILL(w, q);
			const uint32_t a = w.gamma(q, bad);
			SG_REFILL(w, q);
			const uint32_t l = w.gamma(q, bad);
			if ((int64_t)l > extra) { bad = true; break; }
			const int32_t left = i == 0 ? (int32_t)((uint32_t)x + (uint32_t)zigzag32(a)) : (int32_t)((uint32_t)prevEnd + a + 1u), n = (int32_t)l + g.minInt; // in Java ints (BVG:1084-1093)
			prevEnd = (int32_t)((uint32_t)left + (uint32_t)n);
			iv[i] = SegIv{ left, (int32_t)ivArcs, -1, n }; // rank -1: behind every residual, unless B says otherwise
			ivArcs += n;
			if (ivArcs > extra) { bad = true; break; }
		}
	}
	// (the zig-zag value of the first interval is a long in the file: one that does not fit 33 bits made gamma() say bad)
	const int64_t nres = extra - ivArcs;
	if (nres < 0) bad = true;
	o.rpos = (int64_t)w.pos(q);
	o.nres = bad ? 0 : (int32_t)nres;
	o.copied = bad ? 0 : (int32_t)copied;
	o.nIv = bad ? 0 : (int32_t)nIv;
	o.ivArcs = bad ? 0 : (int32_t)ivArcs;
	o.flags = bad ? RF_FALLBACK : 0;
	if (!bad && nres > 0 && (uint64_t)o.rpos >= recEnd) o.flags = RF_FALLBACK; // residuals past the record's end (offsets that disagree with the stream)
}

// segments of a record: the pieces of the SEG_BITS grid that its residual section [rpos, recEnd) touches
SG_D int32_t seg_count(const RecDesc &r, uint64_t recEnd) {
	if ((r.flags & RF_FALLBACK) || r.nres <= 0) return 0;
	const uint64_t c0 = (uint64_t)r.rpos >> SEG_BITS_LOG2, c1 = (recEnd - 1) >> SEG_BITS_LOG2;
	const uint64_t n = c1 - c0 + 1;
	return n > 0x3fffffffull ? 0 : (int32_t)n;
}
SG_D void seg_span(const RecDesc &r, uint64_t recEnd, int32_t i, uint64_t &start, uint64_t &end) { // codes of segment i start in [start, end)
	const uint64_t c = ((uint64_t)r.rpos >> SEG_BITS_LOG2) + (uint64_t)i;
	start = i == 0 ? (uint64_t)r.rpos : c << SEG_BITS_LOG2;
	end = umin64((c + 1) << SEG_BITS_LOG2, recEnd);
}

// ------------------------------------------------------------------------------------------------ A1
// The codes that start in [start, end), from `start`: where the chain leaves the piece, how many codes, the sum of their contributions
// (gap + 1 each; the first code of a record is the zig-zag value relative to x, BVG:954).  Sums are Java ints: they wrap.
// A chain that starts off a codeword boundary reads garbage until it locks on, and garbage can look like a codeword of more than 64
// bits: such a "codeword" is stepped over as one bit and its position reported in badAt (~0: none) -- A2 knows whether it lay before
// the point where the true chain joins this one (harmless) or behind it (the record is flagged).
template <int ZK, int STRIDE>
SG_D void seg_a1(const SegGraph &g, Col<STRIDE> col, int32_t x, uint64_t start, uint64_t end, bool firstOfRecord, uint64_t &out, uint32_t &cnt, uint32_t &sum, uint64_t &badAt) {
	Win<STRIDE> w;
	w.init(g, col, end);
	uint32_t q = w.seek(start);
	cnt = 0; sum = 0; badAt = ~(uint64_t)0;
	while (w.pos(q) < end) {
		SG_REFILL(w, q);
		bool bad = false;
		const uint64_t p0 = w.pos(q);
		const uint32_t v = w.template zeta<ZK>(q, (uint32_t)g.zetaK, bad);
		if (bad && badAt == ~(uint64_t)0) badAt = p0;
		sum = (firstOfRecord && cnt == 0) ? (uint32_t)x + (uint32_t)zigzag32(v) : sum + v + 1u;
		cnt++;
	}
	out = w.pos(q);
}

// ------------------------------------------------------------------------------------------------ A2
// Segment i >= 1: A1 started at the grid point `gstart`; the true chain enters the piece at `in` (>= gstart: the end of the segment
// before).  Walks both chains in lock step until they meet, correcting (cnt, sum).  false: they did not meet inside the piece, so the
// end A1 found is not known to be a true boundary -- or the true chain holds a codeword this decoder does not take.  The last segment
// of a record has no successor that would take its end on trust: there the true chain is simply followed to the end of the record.
template <int ZK, int STRIDE>
SG_D bool seg_a2(const SegGraph &g, Col<STRIDE> col, uint64_t gstart, uint64_t in, uint64_t end, bool last, uint64_t badAt, uint32_t &cnt, uint32_t &sum) {
	if (in == gstart) return badAt == ~(uint64_t)0;
	if (in < gstart || in - gstart > 128) return false; // (a codeword of the segment before cannot reach that far)
	Win<STRIDE> w;
	w.init(g, col, end);
	uint32_t qa = w.seek(gstart);               // the false chain
	uint32_t qb = qa + (uint32_t)(in - gstart); // the true chain
	uint32_t ca = 0, cb = 0, sa = 0, sb = 0;
	bool met = true, badB = false;
	while (qa != qb) {
		const bool aBehind = qa < qb;
		uint32_t &q = aBehind ? qa : qb;
		if (w.pos(q) >= end) { met = false; break; } // the chain that is behind has left the piece (and so has the other): no meeting point
		if (SG_ANY(w.low(qa > qb ? qa : qb))) { const uint32_t dn = w.slide(qa < qb ? qa : qb); qa -= dn; qb -= dn; }
		bool bad = false;
		const uint32_t v = w.template zeta<ZK>(q, (uint32_t)g.zetaK, bad);
		if (aBehind) { ca++; sa += v + 1u; } else { cb++; sb += v + 1u; badB |= bad; }
	}
	if (badB) return false;
	if (!met) {
		if (!last) return false;
		cnt = cb; sum = sb; // both chains are past the end: the true one has been followed all the way
		return true;
	}
	if (badAt != ~(uint64_t)0 && badAt >= w.pos(qa)) return false; // the codeword A1 could not take lies on the true chain
	cnt = cnt - ca + cb;
	sum = sum - sa + sb;
	return true;
}

// ------------------------------------------------------------------------------------------------ B
// Decodes the cnt codes of a segment from its true start `in` and stores every residual at its place among the record's extras:
// residual j (value r) goes to out[j + #(interval ids below r)].  v0 = the residual before the segment's first one, j0 = its index + 1.
// Intervals that the segment's residuals pass learn their rank (= residuals before them).  false: the record must be flagged.
template <int ZK, int STRIDE>
SG_D bool seg_b(const SegGraph &g, Col<STRIDE> col, Col<STRIDE> ring, int32_t x, uint64_t in, uint64_t end, uint32_t cnt, uint32_t j0, int32_t v0, bool firstOfRecord,
                int32_t *out, int32_t extra, SegIv *iv, int32_t nIv) {
	Win<STRIDE> w;
	w.init(g, col, end);
	uint32_t q = w.seek(in);
	bool bad = false;
	// intervals [0, idx) lie below v0 (passed by the segments before): a binary search in the record's arena slice
	int32_t idx = 0;
	if (!firstOfRecord && nIv > 0) {
		int32_t lo = 0, hi = nIv;
		while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (iv[mid].left <= v0) lo = mid + 1; else hi = mid; }
		idx = lo;
	}
	int32_t before = 0, prevEnd = 0; // interval ids below the cursor; end of the last interval passed
	bool havePrev = false;
	if (idx > 0) { const SegIv e = iv[idx - 1]; before = e.pstart + e.len; prevEnd = (int32_t)((uint32_t)e.left + (uint32_t)e.len); havePrev = true; }
	// ring entry k & (RING - 1) holds interval k: (left, pstart + len); intervals [idx, loaded) are in the ring
	int32_t loaded = idx;
	{
		const int32_t n = nIv - idx < RING ? nIv - idx : RING;
		for (int32_t k = 0; k < n; k++) { const SegIv e = iv[idx + k]; const uint32_t s = (uint32_t)(idx + k) & (RING - 1); ring.set(2 * s, (uint32_t)e.left); ring.set(2 * s + 1, (uint32_t)(e.pstart + e.len)); }
		loaded = idx + n;
	}
	int32_t nl = 0, ncum = 0; // the next interval: left, ids up to its end
	if (idx < nIv) { const uint32_t s = (uint32_t)idx & (RING - 1); nl = (int32_t)ring.get(2 * s); ncum = (int32_t)ring.get(2 * s + 1); }
	uint32_t j = j0;
	int32_t val = v0;
	for (uint32_t t = 0; t < cnt; t++) {
		SG_REFILL(w, q);
		if (SG_ANY(loaded < nIv && loaded - idx <= 2)) { // some lane's ring runs low: every lane tops its own up (the wave waits once)
			const int32_t room = RING / 2, n = nIv - loaded < room ? nIv - loaded : room; // (at most half a ring at a time: registers)
#if defined(__HIP_DEVICE_COMPILE__)
			int4 e[RING / 2];
#pragma unroll
			for (int k = 0; k < RING / 2; k++) if (k < n) e[k] = *(const int4 *)(iv + loaded + k);
#pragma unroll
			for (int k = 0; k < RING / 2; k++) if (k < n) { const uint32_t s = (uint32_t)(loaded + k) & (RING - 1); ring.set(2 * s, (uint32_t)e[k].x); ring.set(2 * s + 1, (uint32_t)(e[k].y + e[k].w)); }
#else
			for (int k = 0; k < n; k++) { const SegIv e = iv[loaded + k]; const uint32_t s = (uint32_t)(loaded + k) & (RING - 1); ring.set(2 * s, (uint32_t)e.left); ring.set(2 * s + 1, (uint32_t)(e.pstart + e.len)); }
#endif
			if (n > 0) {
				if (loaded == idx) { const uint32_t s = (uint32_t)idx & (RING - 1); nl = (int32_t)ring.get(2 * s); ncum = (int32_t)ring.get(2 * s + 1); }
				loaded += n;
			}
		}
		const uint32_t v = w.template zeta<ZK>(q, (uint32_t)g.zetaK, bad);
		val = (firstOfRecord && t == 0) ? (int32_t)((uint32_t)x + (uint32_t)zigzag32(v)) : (int32_t)((uint32_t)val + v + 1u); // BVG:954, :966
		while (idx < nIv && nl < val) { // the residual passes interval idx: j residuals precede it
			iv[idx].rank = (int32_t)j;
			prevEnd = (int32_t)((uint32_t)nl + (uint32_t)(ncum - before)); havePrev = true;
			before = ncum;
			idx++;
			if (idx < nIv) {
				if (idx < loaded) { const uint32_t s = (uint32_t)idx & (RING - 1); nl = (int32_t)ring.get(2 * s); ncum = (int32_t)ring.get(2 * s + 1); }
				else { const SegIv e = iv[idx]; nl = e.left; ncum = e.pstart + e.len; loaded = idx; } // (more than RING intervals between two residuals: straight from the arena)
			}
		}
		if ((idx < nIv && nl == val) || (havePrev && val < prevEnd)) bad = true; // a residual inside an interval: equal heads are emitted once (MergedIntIterator.java:69-72) -- not here
		const int64_t p = (int64_t)j + before;
		if (p < (int64_t)extra) out[p] = val; else bad = true;
		j++;
	}
	return !bad;
}

// ------------------------------------------------------------------------------------------------ expand
SG_D void expand_interval(const SegIv e, int32_t nres, int32_t *out, int32_t extra) {
	const int64_t p = (int64_t)e.pstart + (e.rank < 0 ? nres : e.rank);
	for (int32_t t = 0; t < e.len; t++) if (p + t < (int64_t)extra) out[p + t] = (int32_t)((uint32_t)e.left + (uint32_t)t);
}

} // namespace bvsg
