// bv_seg.hpp -- the segment decoder of ONE record of middle length (about a thousand to a few thousand successors) by one wavefront
// from an LDS image of the record (gfx950).  This header holds what is the same on the device and in the host-side model of the kernel
// (tests/cpp/seg_model.cpp runs the phase bodies below lane after lane on the CPU, against the oracle): the carve-up of the wave's LDS
// pool, the code decoders, and the body of every phase as a function of ONE work item.  bv_seg.hip adds what only exists on the GPU.
//
// Where it comes from.  Round 3 built "strip" kernels on these phases for ALL records below the giant class (tag
// r3-strip-kernel-and-merge-experiments, profiles/r3_experiments.txt): for the millions of short records they lose to the bin-sorted
// one-lane kernel (too few lanes have work in a strip), but a record of 500-2000 successors cost them 15-19 us against 60-170 us in the
// one-wave cooperative decoder (k_parse_big<1>), whose every record pays a dozen dependent passes over its tiles.  So the phases stay for
// the class where they win: records too long for one lane (the tail of k_parse_list) and short enough for one LDS image.
//   phase S (structure)  lane 0 -- reference, copy blocks, intervals: the gamma-coded front of the record; leaves where the residual
//                        section starts and how many residuals it holds
//   phase A (anchors)    the residual section is cut at nominal boundaries every SEG_BITS; one lane per boundary runs in from RUNIN_BITS
//                        before it (zeta codes re-synchronise within a few codewords) and reports (first code start, end, count, sum of
//                        gaps) of its segment
//   phase B (chain)      lane 0 checks end[k] == start[k+1] along the segments (re-decoding the rare segment whose run-in had not locked
//                        on) and turns counts and sums into first index / base value
//   phase R (residuals)  one lane per segment: decode, prefix-add, and store every residual at its final place in the CSR row -- the
//                        interval list is walked alongside (two LDS reads per interval) to count the interval ids that precede it,
//                        which also tells every interval where it starts
//   phase X (intervals)  one lane per interval: expand it in place
// Record grammar and semantics: BVG:1032-1133 (successors(x, ibs, window, outd)), ResidualIntIterator BVG:939-991,
// IntIntervalSequenceIterator.java:64-78, MergedIntIterator.java:50-74.  Same contract as the other parse kernels:
// the record's extras (intervals merged with residuals) end up in row[copied..d); the copy pass fills row[0..copied) and merges.
// Default codings only (gamma / unary / zeta_k).
//
// Anything unusual -- a codeword longer than 64 bits, a record that does not fit the wave's LDS budget, a count that does not add up --
// is not handled here: the record is appended to the kernel's escape list and decoded by the cooperative one-wave kernel (k_parse_big)
// afterwards, which also owns all error reporting.  The loops below have no error plumbing; they only have to be memory-safe on garbage.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BVS_HD __host__ __device__ __forceinline__
#else
#define BVS_HD inline
#endif
#ifndef BVS_WHY
#define BVS_WHY(k) ((void)0)
#endif

namespace bvs {

// ---- geometry ----------------------------------------------------------------------------------------------------------
constexpr int WPOOL_WORDS = 3072;   // LDS pool of a wave: 12 KB (13 waves per CU)
constexpr int WIN_MAX_WORDS = 1664; // at most this much stream is staged (52 Kbit); a longer record escapes
constexpr int SEG_BITS = 256, SEG_SHORT_BITS = 384, RUNIN_BITS = 256;
constexpr int MAX_BLOCKS = 8191, MAX_INTERVALS = 8191; // per record; more: escape
constexpr int LONG_INTERVAL = 48;   // intervals at least this long are expanded by the whole wave
constexpr int MID_MIN_DEFAULT = 1024, MID_MAX_DEFAULT = 4096; // successors of the records this decoder takes (BVGPU_MID_MIN / BVGPU_MID_MAX)

BVS_HD uint32_t clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__clz((int)x); // 32 for 0
#else
	return x ? (uint32_t)__builtin_clz(x) : 32u;
#endif
}
BVS_HD uint32_t clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__clzll((long long)x);
#else
	return x ? (uint32_t)__builtin_clzll(x) : 64u;
#endif
}
BVS_HD int32_t nat2int32(uint32_t v) { return (int32_t)(v >> 1) ^ -(int32_t)(v & 1); } // Fast.nat2int, in Java ints

// ---- decoders ----------------------------------------------------------------------------------------------------------
// `w` = the strip's slice of the stream, byte-swapped (first stream bit of a word = bit 31); q = bit offset from w[0].
// Every decoder reads at most w[(q >> 5) + 2]: q is clamped to qmax = (staged words - 3) * 32 after every code, so a decoder
// that runs through garbage stays inside the window.  The common codewords (gamma < 2^16, zeta_3 < 2^21, unary < 32) decode
// from one 32-bit peek without a branch; longer ones take ONE rarely taken branch to a 64-bit peek.  `bad` is set when a
// codeword does not fit 64 bits or a value does not fit 32.
template <class WP> BVS_HD uint32_t peek32(WP w, uint32_t q) {
	const uint32_t j = q >> 5, sh = q & 31u;
	const uint64_t ab = ((uint64_t)w[j] << 32) | w[j + 1];
	return (uint32_t)((ab << sh) >> 32);
}
template <class WP> BVS_HD uint64_t peek64(WP w, uint32_t q) {
	const uint32_t j = q >> 5, sh = q & 31u;
	const uint64_t ab = ((uint64_t)w[j] << 32) | w[j + 1];
	return sh ? (ab << sh) | ((uint64_t)w[j + 2] >> (32u - sh)) : ab;
}
BVS_HD uint32_t advance(uint32_t q, uint32_t len, uint32_t qmax) { const uint32_t n = q + len; return n < qmax ? n : qmax; }
// gamma: returns the value (x, not x + 1)
template <class WP> BVS_HD uint32_t gamma(WP w, uint32_t &q, uint32_t qmax, bool &bad) {
	const uint32_t W = peek32(w, q);
	const uint32_t h = clz32(W);
	uint32_t len = 2 * h + 1;
	uint32_t v = (W >> ((31u - 2 * h) & 31u)) - 1;
	if (__builtin_expect(h >= 16, 0)) {
		const uint64_t W64 = peek64(w, q);
		const uint32_t m = clz64(W64);
		if (m > 31) { bad = true; len = 1; v = 0; }
		else { len = 2 * m + 1; v = (uint32_t)(((W64 << m) >> (63u - m)) - 1); }
	}
	q = advance(q, len, qmax);
	return v;
}
template <class WP> BVS_HD uint32_t unary(WP w, uint32_t &q, uint32_t qmax, bool &bad) {
	const uint32_t W = peek32(w, q);
	uint32_t z = clz32(W);
	if (__builtin_expect(z >= 32, 0)) { const uint64_t W64 = peek64(w, q); z = clz64(W64); if (z >= 64) { bad = true; z = 0; } }
	q = advance(q, z + 1, qmax);
	return z;
}
// zeta_k: K = 3 folded in, K = 0: k at run time (1 <= k <= 16)
template <int K, class WP> BVS_HD uint32_t zeta(WP w, uint32_t &q, uint32_t qmax, uint32_t krt, bool &bad) {
	const uint32_t k = K ? (uint32_t)K : krt;
	const uint32_t W = peek32(w, q);
	const uint32_t h = clz32(W);
	const uint32_t nb = k * h + k - 1;                 // payload bits of the short codeword
	const bool fits = h + 2 + nb <= 32u;
	// (shift amounts are masked: the result is only used when the codeword fits the 32-bit peek)
	const uint32_t mm = nb ? (W << ((h + 1) & 31u)) >> ((31u - nb) & 31u) : 0u; // nb payload bits plus the extra bit of a long codeword
	const uint32_t m = mm >> 1, left = 1u << ((k * h) & 31u);
	const bool lng = nb != 0 && m >= left;             // (zeta_1, h = 0: the codeword "1" has no payload and means 0)
	uint32_t v = lng ? mm - 1 : m + left - 1;
	uint32_t len = h + 1 + nb + (lng ? 1u : 0u);
	if (__builtin_expect(!fits, 0)) {
		const uint64_t W64 = peek64(w, q);
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
	q = advance(q, len, qmax);
	return v;
}

// ---- a strip in a wave's LDS pool --------------------------------------------------------------------------------------
// e = index of a segment of a residual section, j = index into the interval arena.  Row positions are 32-bit offsets
// from the strip's first row in the buffer it is written to.
template <class U32P, class U16P, class I32P> struct StripT {
	U32P win;                 // [nw] stream slice
	// segment table.  Before phase A (long sections only): start = start of the section, out = its end, i0 = index of the segment in it.
	U32P seg_start;           // bit offset of the segment's first codeword
	I32P seg_base;            // value of the residual before its first one (node id for a section's first segment); phase A: sum of its gaps
	U32P seg_out;             // row position of the record's extras + index of the segment's first residual; phase A: end of its last codeword
	U16P seg_cnt, seg_i0;     // codewords in the segment; index of its first residual in the section (bit 15: last segment of the section)
	U16P seg_ivb, seg_ive;    // the record's slice of the interval arena
	// interval arena
	I32P iv_left;             // left end
	U32P iv_out;              // row position of the record's extras
	U16P iv_len, iv_cum, iv_rb; // length; lengths of the record's earlier intervals; residuals of the record below `left`
	int32_t segCap, ivCap;
};
constexpr uint16_t SEG_LAST = 0x8000;
struct StripLayout { int nw, oWin, oSeg, oIv, segCap, ivCap; };
// words [0, nw) the stream; then the segment table (5 words per entry) and the interval arena (4 words per entry, the last half word unused)
BVS_HD StripLayout strip_layout(int64_t nwWant) {
	StripLayout L;
	L.nw = (int)(nwWant < (int64_t)WIN_MAX_WORDS ? nwWant : (int64_t)WIN_MAX_WORDS) & ~3;
	if (L.nw < 8) L.nw = 8;
	L.oWin = 0;
	const int rest = WPOOL_WORDS - L.nw;
	const int cap = (rest / 9) & ~1; // as many segments as intervals (a record of d successors: ~d / 25 segments, ~d / 16 intervals)
	L.segCap = cap; L.ivCap = cap;
	L.oSeg = L.nw;
	L.oIv = L.oSeg + 5 * cap;
	return L;
}
template <class S, class PoolP> BVS_HD void strip_bind(S &st, PoolP pool, const StripLayout &L) {
	st.win = (decltype(st.win))(pool + L.oWin);
	st.seg_start = (decltype(st.seg_start))(pool + L.oSeg);
	st.seg_base = (decltype(st.seg_base))(pool + L.oSeg + L.segCap);
	st.seg_out = (decltype(st.seg_out))(pool + L.oSeg + 2 * L.segCap);
	st.seg_cnt = (decltype(st.seg_cnt))(pool + L.oSeg + 3 * L.segCap);
	st.seg_i0 = st.seg_cnt + L.segCap; st.seg_ivb = st.seg_i0 + L.segCap; st.seg_ive = st.seg_ivb + L.segCap;
	st.iv_left = (decltype(st.iv_left))(pool + L.oIv);
	st.iv_out = (decltype(st.iv_out))(pool + L.oIv + L.ivCap);
	st.iv_len = (decltype(st.iv_len))(pool + L.oIv + 2 * L.ivCap);
	st.iv_cum = st.iv_len + L.ivCap; st.iv_rb = st.iv_cum + L.ivCap;
	st.segCap = L.segCap; st.ivCap = L.ivCap;
}

struct Job {
	int32_t W, minInt;
	uint32_t zk; // zeta k
};

// What phase S leaves in the registers of a record's lane.
struct Rec {
	uint32_t q;      // cursor: after the head, the interval count; after the intervals, the start of the residual section
	uint32_t sbits;  // bits of the residual section (to the end of the record)
	int32_t copied, extra, nIv, ivb, nRes;
	bool ok;
};

// ---- phase S, first half: outdegree, reference, copy blocks, interval count (BVG:1058-1075) ------------------------------
// q0 = start of the record, d = its outdegree (> 0), r = its reference, dref = outdegree of the referent.
template <class S> BVS_HD Rec structure_head(const S &st, const Job &job, uint32_t qmax, uint32_t q0, int32_t d, int32_t r, int64_t dref) {
	Rec R; R.q = q0; R.sbits = 0; R.copied = 0; R.extra = 0; R.nIv = 0; R.ivb = 0; R.nRes = 0; R.ok = false;
	uint32_t q = q0;
	bool bad = false;
	(void)gamma(st.win, q, qmax, bad);                // outdegree (k_headers decoded it)
	if (job.W > 0) (void)unary(st.win, q, qmax, bad); // reference
	int32_t copied = 0;
	if (r > 0) {
		const uint32_t bc = gamma(st.win, q, qmax, bad);
		if (bad || bc > (uint32_t)MAX_BLOCKS || (int64_t)bc > dref + 1) return BVS_WHY(1), R;
		int64_t total = 0;
		for (uint32_t b = 0; b < bc; b++) {
			const uint32_t code = gamma(st.win, q, qmax, bad);
			if (bad || (int64_t)code > dref - total) return BVS_WHY(3), R;
			const int64_t len = (int64_t)code + (b == 0 ? 0 : 1);
			if (total + len > dref) return BVS_WHY(4), R;
			total += len;
			if (!(b & 1)) copied += (int32_t)len;
		}
		if (!(bc & 1)) copied += (int32_t)(dref - total);
	}
	const int32_t extra = d - copied;
	if (extra < 0) return BVS_WHY(5), R;
	int32_t nIv = 0;
	if (extra > 0 && job.minInt != 0) {
		const uint32_t ni = gamma(st.win, q, qmax, bad);
		if (bad || ni > (uint32_t)MAX_INTERVALS || (int32_t)ni > extra) return BVS_WHY(6), R;
		nIv = (int32_t)ni;
	}
	if (bad) return BVS_WHY(7), R;
	R.q = q; R.copied = copied; R.extra = extra; R.nIv = nIv; R.ok = true;
	return R;
}
// ---- phase S, second half: the intervals into the arena slice [ivb, ivb + nIv) (BVG:1076-1096); the residual section -----
// x = node id, rowOut = row position of the record's extras (row start + copied), recEnd = end of the record.
template <class S> BVS_HD void structure_intervals(const S &st, const Job &job, uint32_t qmax, Rec &R, int32_t x, uint32_t rowOut, uint32_t recEnd) {
	uint32_t q = R.q;
	bool bad = false;
	int32_t ivArcs = 0, prevEnd = 0;
	for (int32_t j = 0; j < R.nIv; j++) {
		const uint32_t a = gamma(st.win, q, qmax, bad);
		const uint32_t l = gamma(st.win, q, qmax, bad);
		if (bad || l > (uint32_t)R.extra) { R.ok = false; BVS_WHY(8); return; }
		const int32_t left = j == 0 ? x + nat2int32(a) : prevEnd + (int32_t)a + 1; // BVG:1084-1093, in Java ints
		const int32_t len = (int32_t)l + job.minInt;
		if (ivArcs + len > R.extra) { R.ok = false; BVS_WHY(9); return; }
		st.iv_left[R.ivb + j] = left;
		st.iv_out[R.ivb + j] = rowOut;
		st.iv_len[R.ivb + j] = (uint16_t)len;
		st.iv_cum[R.ivb + j] = (uint16_t)ivArcs;
		st.iv_rb[R.ivb + j] = 0;
		ivArcs += len;
		prevEnd = left + len;
	}
	R.nRes = R.extra - ivArcs;
	R.q = q;
	if (R.nRes < 0) { R.ok = false; BVS_WHY(10); return; }
	if (R.nRes > 0 && (q >= recEnd || recEnd - q > 0xffffu)) { R.ok = false; BVS_WHY(11); return; } // (a residual is at least one bit)
	R.sbits = R.nRes > 0 ? recEnd - q : 0;
}

// segments a residual section needs
BVS_HD int32_t segments_of(int32_t nRes, uint32_t sbits) {
	if (nRes <= 0) return 0;
	return sbits <= (uint32_t)SEG_SHORT_BITS ? 1 : (int32_t)((sbits + SEG_BITS - 1) / SEG_BITS);
}
// the single segment of a short section
template <class S> BVS_HD void segment_short(const S &st, int32_t e, const Rec &R, int32_t x, uint32_t rowOut) {
	st.seg_start[e] = R.q; st.seg_base[e] = x; st.seg_out[e] = rowOut; st.seg_cnt[e] = (uint16_t)R.nRes; st.seg_i0[e] = SEG_LAST;
	st.seg_ivb[e] = (uint16_t)R.ivb; st.seg_ive[e] = (uint16_t)(R.ivb + R.nIv);
}
// segment k of a long section, as phase A wants it
template <class S> BVS_HD void segment_nominal(const S &st, int32_t e, uint32_t r0, uint32_t rEnd, int32_t k) {
	st.seg_start[e] = r0; st.seg_out[e] = rEnd; st.seg_i0[e] = (uint16_t)k; st.seg_cnt[e] = 0;
}

// ---- phase A: one segment of a LONG section, from its nominal boundary ----------------------------------------------
// in: seg_start[e] = start of the section, seg_out[e] = its end, seg_i0[e] = k.  out: start, seg_out = end, seg_base = sum, cnt.
template <int ZK, class S>
BVS_HD void phase_anchor(const S &st, const Job &job, uint32_t qmax, int32_t e) {
	const uint32_t r0 = st.seg_start[e], rEnd = st.seg_out[e], k = st.seg_i0[e];
	const uint32_t b0 = r0 + k * (uint32_t)SEG_BITS, b1 = b0 + (uint32_t)SEG_BITS < rEnd ? b0 + (uint32_t)SEG_BITS : rEnd;
	bool bad = false;
	uint32_t q = k == 0 ? r0 : (b0 - r0 > (uint32_t)RUNIN_BITS ? b0 - (uint32_t)RUNIN_BITS : r0);
	while (q < b0) (void)zeta<ZK>(st.win, q, qmax, job.zk, bad);
	const uint32_t s = q;
	uint32_t cnt = 0, sum = 0;
	if (k == 0 && q < b1) { sum = (uint32_t)nat2int32(zeta<ZK>(st.win, q, qmax, job.zk, bad)); cnt = 1; } // BVG:954
	while (q < b1) { sum += zeta<ZK>(st.win, q, qmax, job.zk, bad) + 1u; cnt++; }                     // BVG:966
	st.seg_start[e] = s;
	st.seg_out[e] = q;
	st.seg_base[e] = (int32_t)sum;
	st.seg_cnt[e] = (uint16_t)(cnt < 0xffffu ? cnt : 0xffffu);
}

// ---- phase B: chain the m segments [e0, e0 + m) of one long section ----------------------------------------------------
// R = the record (R.q = start of the section), x the node, rowOut the row position of its extras.
// false: the counts do not add up (malformed, or a codeword the decoders reject): the record escapes.
template <int ZK, class S>
BVS_HD bool phase_chain(const S &st, const Job &job, uint32_t qmax, int32_t e0, int32_t m, const Rec &R, int32_t x, uint32_t rowOut) {
	const uint32_t r0 = R.q, rEnd = r0 + R.sbits;
	uint32_t expect = r0, idx = 0;
	int32_t val = x;
	for (int32_t k = 0; k < m; k++) {
		const int32_t e = e0 + k;
		uint32_t s = st.seg_start[e], en = st.seg_out[e], cnt = st.seg_cnt[e], sum = (uint32_t)st.seg_base[e];
		if (s != expect) { // the run-in had not locked on: decode this segment from the true boundary
			const uint32_t b1 = r0 + (uint32_t)(k + 1) * SEG_BITS < rEnd ? r0 + (uint32_t)(k + 1) * SEG_BITS : rEnd;
			bool bad = false;
			uint32_t q = expect;
			cnt = 0; sum = 0;
			while (q < b1 && cnt < 0x7fffu) { sum += zeta<ZK>(st.win, q, qmax, job.zk, bad) + 1u; cnt++; if (q >= qmax) break; }
			if (bad) return false;
			s = expect; en = q;
		}
		if (idx + cnt > (uint32_t)R.nRes) return false;
		st.seg_start[e] = s;
		st.seg_i0[e] = (uint16_t)(idx | (idx + cnt == (uint32_t)R.nRes ? (uint32_t)SEG_LAST : 0u));
		st.seg_out[e] = rowOut + idx;
		st.seg_base[e] = val;
		st.seg_cnt[e] = (uint16_t)cnt;
		st.seg_ivb[e] = (uint16_t)R.ivb;
		st.seg_ive[e] = (uint16_t)(R.ivb + R.nIv);
		idx += cnt;
		val += (int32_t)sum;
		expect = en;
	}
	return idx == (uint32_t)R.nRes && expect == rEnd;
}

// ---- phase R: the residuals of one segment, stored at their final place ------------------------------------------------
// rows = the buffer the strip's rows live in, seen from the strip's first row.
template <int ZK, class S, class ROWS>
BVS_HD bool phase_residuals(const S &st, const Job &job, uint32_t qmax, ROWS rows, int32_t e) {
	uint32_t q = st.seg_start[e];
	const uint32_t i0f = st.seg_i0[e];
	const int32_t i0 = (int32_t)(i0f & 0x7fffu), cnt = (int32_t)st.seg_cnt[e];
	int32_t val = st.seg_base[e];
	const int32_t jEnd = (int32_t)st.seg_ive[e];
	int32_t j = (int32_t)st.seg_ivb[e];
	const int32_t ivTotal = jEnd > j ? (int32_t)st.iv_cum[jEnd - 1] + (int32_t)st.iv_len[jEnd - 1] : 0;
	if (i0 > 0) { // the intervals below the residual before this segment belong to earlier segments
		int32_t lo = j, hi = jEnd;
		while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (st.iv_left[mid] < val) lo = mid + 1; else hi = mid; }
		j = lo;
	}
	int32_t cum = j < jEnd ? (int32_t)st.iv_cum[j] : ivTotal;
	int32_t nextLeft = j < jEnd ? st.iv_left[j] : 0x7fffffff;
	const uint32_t out = st.seg_out[e];
	bool bad = false;
	for (int32_t t = 0; t < cnt; t++) {
		const uint32_t c = zeta<ZK>(st.win, q, qmax, job.zk, bad);
		val = (i0 + t == 0) ? val + nat2int32(c) : val + (int32_t)c + 1; // BVG:954, :966 (Java ints)
		while (j < jEnd && nextLeft < val) { // the intervals between the previous residual and this one: this residual is the first above them
			st.iv_rb[j] = (uint16_t)(i0 + t);
			cum += (int32_t)st.iv_len[j];
			j++;
			nextLeft = j < jEnd ? st.iv_left[j] : 0x7fffffff;
		}
		rows[out + (uint32_t)(t + cum)] = val;
	}
	if (i0f & SEG_LAST) for (; j < jEnd; j++) st.iv_rb[j] = (uint16_t)(i0 + cnt); // intervals above the last residual
	return !bad;
}

// ---- phase X: one interval ----------------------------------------------------------------------------------------------
template <class S, class ROWS> BVS_HD void phase_interval(const S &st, ROWS rows, int32_t j, int32_t t0, int32_t step) {
	const uint32_t out = st.iv_out[j] + (uint32_t)st.iv_cum[j] + (uint32_t)st.iv_rb[j];
	const int32_t left = st.iv_left[j], len = (int32_t)st.iv_len[j];
	for (int32_t t = t0; t < len; t += step) rows[out + (uint32_t)t] = left + t;
}

} // namespace bvs
