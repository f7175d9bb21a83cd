// bv_strip.hpp -- the strip decoder: all records that START in one slice of the bit stream, decoded by one work-group
// entirely in LDS (gfx950).  This header holds the part that is the same on the device and in the host-side model of
// the kernel (tests/cpp/strip_model.cpp runs the phases below lane after lane on the CPU, against the oracle): layout of
// a strip in LDS, the code decoders, and the body of every phase as a function of ONE work item.  bv_strip.hip adds what
// only exists on the GPU: staging, block-wide scans and sorts, the hand-out of work items to wavefronts, barriers.
//
// Why strips.  The records of consecutive nodes are consecutive in the stream and in the CSR, so a group that owns a
// slice of the stream reads it once with coalesced 16-byte loads and writes its rows once, whole cache lines at a time;
// in between nothing touches HBM.  What made earlier tile kernels slow was not the memory side but balance: one lane per
// record means a wave lasts as long as its longest record.  Here the unit of work is never a record:
//   phase S (structure)  one lane per record, records sorted by outdegree -- reference, copy blocks, intervals: the
//                        short, gamma-coded front of a record; leaves where its residual section starts
//   phase A (anchors)    residual sections longer than SEG_SHORT_BITS are cut at nominal boundaries every SEG_BITS; one
//                        lane per boundary runs in from RUNIN bits before it (zeta codes re-synchronise within a few
//                        codewords) and reports (first code start, end, count, sum of gaps) for its segment
//   phase B (chain)      one lane per long section checks end[k] == start[k+1] along its segments (re-decoding the rare
//                        segment whose run-in had not locked on) and turns counts and sums into first index / base value
//   phase R (residuals)  one lane per SEGMENT (<= ~64 codes), segments sorted by length: decode, prefix-add, and write
//                        every residual at its final place in the row -- the interval list of the record is walked
//                        alongside (two LDS reads per interval) to count the interval ids that precede it
//   phase X (intervals)  one lane per interval: expand it at the place phase R worked out for it
//   phase W (write-out)  the rows leave LDS for the CSR, 16 lanes per row
// Record grammar and semantics: BVG:1032-1133 (successors(x, ibs, window, outd)), ResidualIntIterator BVG:939-991,
// IntIntervalSequenceIterator.java:64-78, MergedIntIterator.java:50-74.  Same contract as the other
// parse kernels: the record's extras (intervals merged with residuals) end up in row[copied..d); the copy pass fills
// row[0..copied) and merges.  Default codings only (gamma / unary / zeta_k).
//
// Anything unusual -- a codeword longer than 64 bits, a record that does not fit the strip's LDS budget, a count that
// does not add up -- is not handled here: the record is appended to the strip kernel's escape list and decoded by the
// cooperative one-wave kernel (k_parse_big) afterwards, which also owns all error reporting.  The hot loops below
// therefore have no slow paths and no error plumbing; they only have to be memory-safe on garbage.
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
constexpr int STRIP_T = 512;                 // threads per strip
constexpr int POOL_WORDS = 19 * 1024;        // LDS pool per strip: 76 KB (two strips per CU)
// A strip is the set of slots s whose weight(s) = bits(s) + NODE_W * s + ARC_W * rowstart(s) falls into one window of
// SPAN_W: each resource is bounded, and so is their weighted sum -- the LDS layout is carved per strip from what it holds.
constexpr int NODE_BYTES = 28;               // per-record fields (below)
constexpr int NODE_W = NODE_BYTES * 8, ARC_W = 32; // weight = 8 x the LDS bytes a unit needs (1 bit of stream = 1/8 byte)
constexpr int MAIN_BYTES = 40 * 1024;        // stream + rows + record fields of the slots inside the window ...
constexpr int64_t SPAN_W = (int64_t)MAIN_BYTES * 8;
constexpr int STRIP_MAX_DEFAULT = 512;       // records with more successors are not strip work (cooperative kernels)
constexpr int OVERHANG_WORDS = 640;          // ... plus the part of the last record's bits (<= 20 Kbit staged) and row beyond the window
constexpr int MAX_NODES = (int)(SPAN_W / NODE_W) + 1;
constexpr int SEG_BITS = 256, SEG_SHORT_BITS = 384, RUNIN_BITS = 192;
constexpr int MAX_BLOCKS = 255, MAX_INTERVALS = 1023; // per record; more: escape
constexpr int LONG_INTERVAL = 48;            // intervals at least this long are expanded by a whole wave

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
// `win` = the strip's slice of the stream, byte-swapped (first stream bit of a word = bit 31); q = bit offset from win[0].
// Every decoder reads at most win[(q >> 5) + 2]: the window is followed by >= 3 readable words, and q is clamped to qmax
// after every code (a decoder that runs through garbage stays inside the window).  `bad` is set when a codeword does not
// fit 64 bits or a value does not fit 32.
struct Win {
	const uint32_t *w;
	uint32_t qmax; // largest cursor value: (staged words - 3) * 32
};
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
// gamma: returns value (x, not x+1)
template <class WP> BVS_HD uint32_t gamma(WP w, uint32_t &q, uint32_t qmax, bool &bad) {
	const uint32_t W = peek32(w, q);
	const uint32_t h = clz32(W);
	uint32_t v, len;
	if (h < 16) { len = 2 * h + 1; v = (W >> (31u - 2 * h)) - 1; }
	else {
		const uint64_t W64 = peek64(w, q);
		const uint32_t m = clz64(W64);
		if (m > 31) { bad = true; len = 1; v = 0; }
		else { len = 2 * m + 1; v = (uint32_t)(((W64 << m) >> (63u - m)) - 1); }
	}
	q = q + len < qmax ? q + len : qmax;
	return v;
}
template <class WP> BVS_HD uint32_t unary(WP w, uint32_t &q, uint32_t qmax, bool &bad) {
	const uint32_t W = peek32(w, q);
	uint32_t z = clz32(W);
	if (z >= 32) { const uint64_t W64 = peek64(w, q); z = clz64(W64); if (z >= 64) { bad = true; z = 0; } }
	q = q + z + 1 < qmax ? q + z + 1 : qmax;
	return z;
}
// zeta_k: K = 3 folded in, K = 0: k at run time (1 <= k <= 16)
template <int K, class WP> BVS_HD uint32_t zeta(WP w, uint32_t &q, uint32_t qmax, uint32_t krt, bool &bad) {
	const uint32_t k = K ? (uint32_t)K : krt;
	const uint32_t W = peek32(w, q);
	const uint32_t h = clz32(W);
	const uint32_t nb = k * h + k - 1; // payload bits of the short codeword
	uint32_t v, len;
	if (h + 2 + nb <= 32u) {
		if (K == 0 && nb == 0) { v = 0; len = 1; } // zeta_1, h = 0: "1" means 0
		else {
			const uint32_t mm = (W << (h + 1)) >> (31u - nb); // nb payload bits plus the extra bit of a long codeword
			const uint32_t m = mm >> 1, left = 1u << (k * h);
			const bool lng = m >= left;
			v = lng ? mm - 1 : m + left - 1;
			len = h + 1 + nb + (lng ? 1u : 0u);
		}
	} else {
		const uint64_t W64 = peek64(w, q);
		const uint32_t h2 = clz64(W64);
		const uint32_t nb2 = k * h2 + k - 1;
		if (h2 + 2 + nb2 > 64u || k * h2 > 32u) { bad = true; v = 0; len = 1; }
		else {
			const uint64_t mm = (W64 << (h2 + 1)) >> (63u - nb2);
			const uint64_t m = mm >> 1, left = (uint64_t)1 << (k * h2);
			const bool lng = m >= left;
			const uint64_t vv = lng ? mm - 1 : m + left - 1;
			if (vv > 0xffffffffull) bad = true;
			v = (uint32_t)vv;
			len = h2 + 1 + nb2 + (lng ? 1u : 0u);
		}
	}
	q = q + len < qmax ? q + len : qmax;
	return v;
}

// ---- a strip in LDS ----------------------------------------------------------------------------------------------------
// One segment of a residual section (16 bytes).  Before phase A: k = index of the segment inside its section.
struct Seg {
	uint32_t start;  // bit offset of the segment's first codeword (phase A: candidate, phase B: true)
	uint32_t end;    // phase A: end of its last codeword; phase B on: index of its first residual in the section
	int32_t base;    // phase A: sum of its gaps; phase B on: value of the residual before its first one (node id for the first segment)
	uint16_t cnt;    // codewords in the segment
	uint16_t rec;    // local index of the record
};

// Field arrays of a strip.  i = local index of a record (slot - first slot of the strip), j = index into the interval arena.
template <class U32P, class U16P, class I32P, class SEGP> struct StripT {
	U32P win;              // [nw] stream slice
	I32P rows;             // [narcs] the rows of the strip's own records, in CSR order, back to back
	U32P m_bit;            // record start (bits from win[0]); after phase S: start of its residual section
	U16P m_d, m_off;       // outdegree (0: not strip work); row offset in `rows`
	U16P m_cop, m_nres;    // ids copied from the referent; residuals
	U16P m_sbits;          // bits of the residual section (to the end of the record)
	U16P m_ivb, m_niv;     // the record's slice of the interval arena
	U16P m_blkb, m_nblk;   // the record's slice of the block arena (kept for referents inside the strip)
	U16P m_ref;            // reference distance
	U16P m_seg0;           // first segment of the record
	I32P iv_left;          // interval arena: left end,
	U16P iv_len, iv_cum;   //   length, lengths of the record's earlier intervals,
	U16P iv_rb, iv_base;   //   residuals of the record below `left`, row offset of the record's extras in `rows`
	U16P blk;              // block arena: copy-block lengths as coded (+1 from the second on)
	SEGP seg;              // segment table
	U16P list, listB;      // work lists (records / segments, sorted; long sections)
	int32_t ivCap, blkCap, segCap, listLen;
};

// ---- carve-up of the LDS pool -------------------------------------------------------------------------------------------
// All offsets in 32-bit words from the start of the pool.  The rows, the record fields and the stream slice come first
// (their sum is bounded by the strip's weight); what is left is split between the block arena, the segment table with its
// two work lists, and the interval arena.
struct StripLayout { int nw, oWin, oRows, oBit, oF16, f16Stride, oList, oListB, listLen, oBlk, oSeg, oIv, ivCap, blkCap, segCap; bool ok; };
constexpr int AUX_MIN_WORDS = 6144; // words always kept for the arenas and lists (24 KB)
BVS_HD StripLayout strip_layout(int n, int narcs, int64_t nwWant, int minInt) {
	StripLayout L;
	const int n2 = (n + 1) >> 1; // words of a uint16 field
	L.f16Stride = n2;
	int o = 0;
	L.oRows = o; o += (narcs + 3) & ~3;
	L.oBit = o; o += n;
	L.oF16 = o; o += 11 * n2;
	o = (o + 3) & ~3;
	L.oWin = o;
	const int winMax = POOL_WORDS - AUX_MIN_WORDS - o;
	L.ok = winMax >= 64;
	L.nw = (int)(nwWant < (int64_t)winMax ? nwWant : (int64_t)winMax) & ~3;
	if (L.nw < 8) L.nw = 8;
	o += L.nw;
	int rest = POOL_WORDS - o;
	L.blkCap = 2048;             // uint16 entries: 1024 words
	L.oBlk = o; o += L.blkCap / 2; rest -= L.blkCap / 2;
	// A segment costs 5 words (16 bytes + a uint16 slot in each of the two lists), an interval 3.  What the strip can need at
	// most is known: a segment per record plus one per SEG_BITS of stream, an interval per minInt arcs.  Both get that when it
	// fits (then nothing escapes for lack of room); otherwise the segments get theirs first, up to three fifths of the space.
	const int segWant = n + (L.nw * 32) / SEG_BITS + 8;
	const int ivWant = minInt > 0 ? narcs / minInt + 1 : 0;
	L.segCap = segWant;
	if (5 * segWant + 3 * ivWant + 8 > rest && 5 * segWant > rest / 5 * 3) L.segCap = (rest / 5 * 3) / 5;
	if (L.segCap > 0x7ff0) L.segCap = 0x7ff0;
	if (L.segCap < 1) { L.segCap = 1; L.ok = false; }
	L.listLen = L.segCap > n ? L.segCap : n;
	L.oSeg = o; o += 4 * L.segCap;
	L.oList = o; o += (L.listLen + 1) >> 1;
	L.oListB = o; o += (L.listLen + 1) >> 1;
	rest = POOL_WORDS - o;
	L.oIv = o;
	L.ivCap = rest > 0 ? rest / 3 : 0;
	if (L.ivCap > ivWant) L.ivCap = ivWant;
	if (L.ivCap > 0x7ff0) L.ivCap = 0x7ff0;
	if (rest < 0) L.ok = false;
	return L;
}
// Binds the field arrays of a strip to its pool.  `pool` = pointer to the first word (LDS-qualified on the device).
template <class S, class PoolP> BVS_HD void strip_bind(S &st, PoolP pool, const StripLayout &L) {
	st.win = (decltype(st.win))(pool + L.oWin);
	st.rows = (decltype(st.rows))(pool + L.oRows);
	st.m_bit = (decltype(st.m_bit))(pool + L.oBit);
	const decltype(st.m_d) f16 = (decltype(st.m_d))(pool + L.oF16);
	const int fs = 2 * L.f16Stride;
	st.m_d = f16; st.m_off = f16 + fs; st.m_cop = f16 + 2 * fs; st.m_nres = f16 + 3 * fs; st.m_sbits = f16 + 4 * fs; st.m_ivb = f16 + 5 * fs;
	st.m_niv = f16 + 6 * fs; st.m_blkb = f16 + 7 * fs; st.m_nblk = f16 + 8 * fs; st.m_ref = f16 + 9 * fs; st.m_seg0 = f16 + 10 * fs;
	st.blk = (decltype(st.blk))(pool + L.oBlk);
	st.seg = (decltype(st.seg))(pool + L.oSeg);
	st.list = (decltype(st.list))(pool + L.oList);
	st.listB = (decltype(st.listB))(pool + L.oListB);
	st.iv_left = (decltype(st.iv_left))(pool + L.oIv);
	st.iv_len = (decltype(st.iv_len))(pool + L.oIv + L.ivCap);
	st.iv_cum = st.iv_len + L.ivCap; st.iv_rb = st.iv_cum + L.ivCap; st.iv_base = st.iv_rb + L.ivCap;
	st.ivCap = L.ivCap; st.blkCap = L.blkCap; st.segCap = L.segCap; st.listLen = L.listLen;
}

// What a phase needs to know about the job.
struct Job {
	int32_t W, minInt;
	uint32_t zk;       // zeta k
	int32_t stripMax;  // records with at least this many successors are not strip work
	int32_t x0;        // node id of local record 0
};

enum : uint32_t { ESC_NONE = 0, ESC_BAD = 1 };

// ---- phase S: the front of one record ---------------------------------------------------------------------------------
// Returns false when the record escapes (the caller appends it to the escape list and clears m_d[i]).
// drefOf(i, r): outdegree of the referent (slot of record i minus r), wherever it lives.
// ivAlloc(n) / blkAlloc(n): bump allocation in the arenas (atomicAdd on an LDS counter), < 0 when full.
template <int ZK, class S, class FDref, class FIv, class FBlk>
BVS_HD bool phase_structure(const S &st, const Job &job, uint32_t qmax, int32_t i, uint32_t recEnd, FDref drefOf, FIv ivAlloc, FBlk blkAlloc, bool keepBlocks) {
	const int32_t x = job.x0 + i;
	const int32_t d = (int32_t)st.m_d[i];
	uint32_t q = st.m_bit[i];
	bool bad = false;
	(void)gamma(st.win, q, qmax, bad);                     // outdegree (k_headers decoded it)
	if (job.W > 0) (void)unary(st.win, q, qmax, bad);      // reference
	const int32_t r = (int32_t)st.m_ref[i];
	int32_t copied = 0;
	st.m_nblk[i] = 0; st.m_blkb[i] = 0;
	if (r > 0) { // BVG:1058-1071
		const int64_t dref = drefOf(i, r);
		const uint32_t bc = gamma(st.win, q, qmax, bad);
		if (bad || bc > (uint32_t)MAX_BLOCKS || (int64_t)bc > dref + 1) return BVS_WHY(1), false;
		int32_t bb = -1;
		if (keepBlocks && bc > 0 && dref <= 0xffff) { bb = blkAlloc((int32_t)bc); if (bb >= 0) { st.m_blkb[i] = (uint16_t)bb; st.m_nblk[i] = (uint16_t)bc; } } // (arena full: the list is simply not kept)
		int64_t total = 0;
		for (uint32_t b = 0; b < bc; b++) {
			const uint32_t code = gamma(st.win, q, qmax, bad);
			if (bad || (int64_t)code > dref - total) return BVS_WHY(3), false;
			const int64_t len = (int64_t)code + (b == 0 ? 0 : 1);
			if (total + len > dref) return BVS_WHY(4), false;
			if (bb >= 0) st.blk[bb + (int32_t)b] = (uint16_t)len;
			total += len;
			if (!(b & 1)) copied += (int32_t)len;
		}
		if (!(bc & 1)) copied += (int32_t)(dref - total);
	}
	const int32_t extra = d - copied;
	if (extra < 0) return BVS_WHY(5), false;
	int32_t nIv = 0, ivArcs = 0, ivb = 0;
	if (extra > 0 && job.minInt != 0) { // BVG:1073-1096
		const uint32_t ni = gamma(st.win, q, qmax, bad);
		if (bad || ni > (uint32_t)MAX_INTERVALS || (int32_t)ni > extra) return BVS_WHY(6), false;
		nIv = (int32_t)ni;
		if (nIv) {
			ivb = ivAlloc(nIv);
			if (ivb < 0) return BVS_WHY(7), false;
			int32_t prevEnd = 0;
			const uint16_t base = (uint16_t)(st.m_off[i] + copied);
			for (int32_t j = 0; j < nIv; j++) {
				const uint32_t a = gamma(st.win, q, qmax, bad);
				const uint32_t l = gamma(st.win, q, qmax, bad);
				if (bad || l > (uint32_t)extra) return BVS_WHY(8), false;
				const int32_t left = j == 0 ? x + nat2int32(a) : prevEnd + (int32_t)a + 1; // BVG:1084-1093, in Java ints
				const int32_t len = (int32_t)l + job.minInt;
				if (ivArcs + len > extra) return BVS_WHY(9), false;
				st.iv_left[ivb + j] = left;
				st.iv_len[ivb + j] = (uint16_t)len;
				st.iv_cum[ivb + j] = (uint16_t)ivArcs;
				st.iv_rb[ivb + j] = 0;
				st.iv_base[ivb + j] = base;
				ivArcs += len;
				prevEnd = left + len;
			}
		}
	}
	const int32_t nRes = extra - ivArcs;
	if (bad || nRes < 0) return BVS_WHY(10), false;
	if (nRes > 0 && (q >= recEnd || recEnd - q > 0xffffu)) return BVS_WHY(11), false; // (a residual is at least one bit)
	st.m_cop[i] = (uint16_t)copied;
	st.m_nres[i] = (uint16_t)nRes;
	st.m_ivb[i] = (uint16_t)ivb;
	st.m_niv[i] = (uint16_t)nIv;
	st.m_bit[i] = q;
	st.m_sbits[i] = (uint16_t)(nRes > 0 ? recEnd - q : 0);
	return true;
}

// segments a record's residual section needs
BVS_HD int32_t segments_of(uint32_t nRes, uint32_t sbits) {
	if (nRes == 0) return 0;
	return sbits <= (uint32_t)SEG_SHORT_BITS ? 1 : (int32_t)((sbits + SEG_BITS - 1) / SEG_BITS);
}

// ---- phase A: one segment of a LONG section, from its nominal boundary ----------------------------------------------
template <int ZK, class S>
BVS_HD void phase_anchor(const S &st, const Job &job, uint32_t qmax, int32_t e) {
	const int32_t i = (int32_t)st.seg[e].rec;
	const uint32_t k = st.seg[e].end; // segment index inside the section
	const uint32_t r0 = st.m_bit[i], rEnd = r0 + (uint32_t)st.m_sbits[i];
	const uint32_t b0 = r0 + k * (uint32_t)SEG_BITS, b1 = b0 + (uint32_t)SEG_BITS < rEnd ? b0 + (uint32_t)SEG_BITS : rEnd;
	bool bad = false;
	uint32_t q = k == 0 ? r0 : (b0 - r0 > (uint32_t)RUNIN_BITS ? b0 - (uint32_t)RUNIN_BITS : r0);
	while (q < b0) (void)zeta<ZK>(st.win, q, qmax, job.zk, bad);
	const uint32_t s = q;
	uint32_t cnt = 0, sum = 0;
	if (k == 0 && q < b1) { sum = (uint32_t)nat2int32(zeta<ZK>(st.win, q, qmax, job.zk, bad)); cnt = 1; } // BVG:954
	while (q < b1) { sum += zeta<ZK>(st.win, q, qmax, job.zk, bad) + 1u; cnt++; }                     // BVG:966
	st.seg[e].start = s;
	st.seg[e].end = q;
	st.seg[e].base = (int32_t)sum;
	st.seg[e].cnt = (uint16_t)(cnt < 0xffffu ? cnt : 0xffffu);
}

// ---- phase B: chain the segments of one long section ------------------------------------------------------------------
// false: the counts do not add up (malformed, or a codeword that the decoders reject): the record escapes.
template <int ZK, class S>
BVS_HD bool phase_chain(const S &st, const Job &job, uint32_t qmax, int32_t i) {
	const int32_t e0 = (int32_t)st.m_seg0[i];
	const uint32_t r0 = st.m_bit[i], sbits = st.m_sbits[i], rEnd = r0 + sbits, nRes = st.m_nres[i];
	const int32_t m = segments_of(nRes, sbits);
	uint32_t expect = r0, idx = 0;
	int32_t val = job.x0 + i;
	for (int32_t k = 0; k < m; k++) {
		const int32_t e = e0 + k;
		uint32_t s = st.seg[e].start, en = st.seg[e].end, cnt = st.seg[e].cnt, sum = (uint32_t)st.seg[e].base;
		if (s != expect) { // the run-in had not locked on: decode this segment from the true boundary
			const uint32_t b1 = r0 + (uint32_t)(k + 1) * SEG_BITS < rEnd ? r0 + (uint32_t)(k + 1) * SEG_BITS : rEnd;
			bool bad = false;
			uint32_t q = expect;
			cnt = 0; sum = 0;
			while (q < b1 && cnt < 0xffffu) { sum += zeta<ZK>(st.win, q, qmax, job.zk, bad) + 1u; cnt++; if (q >= qmax) break; }
			if (bad) return false;
			s = expect; en = q;
		}
		st.seg[e].start = s;
		st.seg[e].end = idx;
		st.seg[e].base = val;
		st.seg[e].cnt = (uint16_t)cnt;
		idx += cnt;
		val += (int32_t)sum;
		expect = en;
	}
	return idx == nRes && expect == rEnd;
}

// ---- phase R: the residuals of one segment --------------------------------------------------------------------------
template <int ZK, class S>
BVS_HD bool phase_residuals(const S &st, const Job &job, uint32_t qmax, int32_t e) {
	const int32_t i = (int32_t)st.seg[e].rec;
	uint32_t q = st.seg[e].start;
	const int32_t i0 = (int32_t)st.seg[e].end, cnt = (int32_t)st.seg[e].cnt, nRes = (int32_t)st.m_nres[i];
	int32_t val = st.seg[e].base;
	const int32_t jEnd = (int32_t)st.m_ivb[i] + (int32_t)st.m_niv[i];
	int32_t j = (int32_t)st.m_ivb[i];
	const int32_t ivTotal = jEnd > j ? (int32_t)st.iv_cum[jEnd - 1] + (int32_t)st.iv_len[jEnd - 1] : 0; // (from the arena: m_d is cleared when a record escapes)
	if (i0 > 0) { // the intervals below the residual before this segment belong to earlier segments
		int32_t lo = j, hi = jEnd;
		while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (st.iv_left[mid] < val) lo = mid + 1; else hi = mid; }
		j = lo;
	}
	int32_t cum = j < jEnd ? (int32_t)st.iv_cum[j] : ivTotal;
	int32_t nextLeft = j < jEnd ? st.iv_left[j] : 0x7fffffff;
	const int32_t out = (int32_t)st.m_off[i] + (int32_t)st.m_cop[i] + i0;
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
		st.rows[out + t + cum] = val;
	}
	if (i0 + cnt == nRes) for (; j < jEnd; j++) st.iv_rb[j] = (uint16_t)nRes; // intervals above the last residual
	return !bad;
}

// ---- phase X: one interval ----------------------------------------------------------------------------------------------
template <class S> BVS_HD void phase_interval(const S &st, int32_t j, int32_t t0, int32_t step) {
	const int32_t out = (int32_t)st.iv_base[j] + (int32_t)st.iv_cum[j] + (int32_t)st.iv_rb[j], left = st.iv_left[j], len = (int32_t)st.iv_len[j];
	for (int32_t t = t0; t < len; t += step) st.rows[out + t] = left + t;
}

} // namespace bvs
