// bv_encode.hpp -- the BVGraph compressor's per-node work as code shared by the device kernels (bv_encode.hip) and a host
// model compiled by g++ (tests/cpp/encode_model.cpp).  SURVEY.md section 8 row f1.
//
// What the reference does per node x (CompressionThread.call, BVGraph.java:2222-2386): for every candidate reference
// x - r, r = 0..W, whose chain is not too long (:2313-2327), run diffComp (:2049-2219) against a bit-counting stream and
// keep the cheapest; then run it again for real.  diffComp walks the two sorted lists once (:2072-2121) producing copy
// blocks and "extras"; intervalize (:1631-1654) splits the extras into intervals and residuals; then the fields are
// written in the order of the record grammar (:2123-2217).
//
// Here that becomes four data-parallel phases (bv_encode.hip):
//   A  cost of every (node, candidate) pair  -- independent, the bulk of the work: diff_walk into a bit-counting visitor
//   B  the choice of the reference           -- a recurrence over the nodes through the chain lengths (select_chunk)
//   C  record lengths -> bit offsets         -- a scan
//   D  emission                              -- independent per node: diff_walk again, writing at the record's offset
// The walk is ONE template (diff_walk) instantiated with a counting and a writing visitor, so that what phase A prices
// is what phase D writes.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BVE_HD __host__ __device__ __forceinline__
#else
#define BVE_HD inline
#endif

namespace bve {

// CompressionFlags.java:26-44
enum { C_DELTA = 1, C_GAMMA = 2, C_GOLOMB = 3, C_SKEWED_GOLOMB = 4, C_UNARY = 5, C_ZETA = 6, C_NIBBLE = 7 };

struct Params {
	int32_t W, R, I, K;                                   // windowsize, maxrefcount, minintervallength, zetak
	int32_t c_outd, c_blk, c_res, c_ref, c_bc, c_off;     // codings (setFlags, BVGraph.java:1317-1325)
	int32_t per;                                          // nodes per independently compressed part (the reference's threads start with an empty window, :2471-2550)
};

BVE_HD int msb64(uint64_t v) { return 63 - __builtin_clzll(v); }
BVE_HD uint64_t int2nat(int64_t x) { return x >= 0 ? (uint64_t)x << 1 : (uint64_t)(-x) * 2 - 1; } // Fast.int2nat

// ---- sinks: `zeros(n)` advances over n zero bits (the stream starts zeroed), `put(v, n)` appends the low n bits of v (n <= 64)
struct LenSink {
	uint64_t bits = 0;
	BVE_HD void zeros(uint64_t n) { bits += n; }
	BVE_HD void put(uint64_t, int n) { bits += (uint64_t)n; }
};

#if defined(__HIP_DEVICE_COMPILE__)
#define BVE_OR32(p, v) atomicOr((unsigned int *)(p), (unsigned int)(v))
#else
#define BVE_OR32(p, v) (*(p) |= (v))
#endif

// MSB-first bit writer over 32-bit words stored big-endian (= the file's bytes).  Records of different lanes share
// words at their ends, so every word goes out with an OR; a lane gathers the bits of one word before it does.
struct WordSink {
	uint32_t *words;
	uint64_t pos;
	uint32_t cur = 0;
	BVE_HD WordSink(uint32_t *w, uint64_t p) : words(w), pos(p) {}
	BVE_HD void flush() {
		if (cur) { BVE_OR32(words + ((pos - 1) >> 5), __builtin_bswap32(cur)); cur = 0; }
	}
	BVE_HD void zeros(uint64_t n) {
		if (n == 0) return;
		if ((pos & 31u) + n >= 32u) { if (pos & 31u) flush(); }
		pos += n;
	}
	BVE_HD void put(uint64_t v, int n) {
		while (n > 0) {
			const int room = 32 - (int)(pos & 31u);
			const int t = n < room ? n : room;
			const uint32_t chunk = (uint32_t)((v >> (n - t)) & (t == 32 ? 0xffffffffull : ((1ull << t) - 1)));
			cur |= t == 32 ? chunk : chunk << (room - t);
			pos += (uint64_t)t;
			n -= t;
			if ((pos & 31u) == 0) flush();
		}
	}
	BVE_HD void finish() { if (pos & 31u) flush(); }
};

// ---- universal codes (dsiutils OutputBitStream: writeUnary/Gamma/Delta/Zeta/Golomb/Nibble; the lengths are what the
// bit-counting run of diffComp adds up)
template <class S> BVE_HD void w_unary(S &s, uint64_t x) { s.zeros(x); s.put(1, 1); }
template <class S> BVE_HD void w_gamma(S &s, uint64_t x) { const int m = msb64(x + 1); s.zeros((uint64_t)m); s.put(x + 1, m + 1); }
template <class S> BVE_HD void w_delta(S &s, uint64_t x) { const int m = msb64(x + 1); w_gamma(s, (uint64_t)m); s.put(x + 1, m); }
template <class S> BVE_HD void w_zeta(S &s, uint64_t x, int k) {
	const uint64_t v = x + 1;
	const int h = msb64(v) / k;
	w_unary(s, (uint64_t)h);
	const uint64_t left = (uint64_t)1 << (h * k);
	if (v - left < left) s.put(v - left, h * k + k - 1); else s.put(v, h * k + k);
}
template <class S> BVE_HD void w_golomb(S &s, uint64_t x, int b) {
	if (b == 0) return;
	const uint64_t q = x / (uint64_t)b, r = x % (uint64_t)b;
	w_unary(s, q);
	const int l2 = msb64((uint64_t)b);
	const uint64_t mm = ((uint64_t)1 << (l2 + 1)) - (uint64_t)b;
	if (r < mm) s.put(r, l2); else s.put(r + mm, l2 + 1);
}
template <class S> BVE_HD void w_nibble(S &s, uint64_t x) {
	if (x == 0) { s.put(8, 4); return; }
	int h = msb64(x) / 3;
	do { s.put(h == 0 ? 1 : 0, 1); s.put((x >> (h * 3)) & 7, 3); } while (h-- != 0);
}
template <class S> BVE_HD void w_code(S &s, int coding, uint64_t x, int k) {
	switch (coding) {
	case C_GAMMA: w_gamma(s, x); break;
	case C_DELTA: w_delta(s, x); break;
	case C_UNARY: w_unary(s, x); break;
	case C_ZETA: w_zeta(s, x, k); break;
	case C_GOLOMB: w_golomb(s, x, k); break;
	case C_NIBBLE: w_nibble(s, x); break;
	default: break;
	}
}

// the five fields of a record whose coding the flags choose; DEF = the default codings with zetak = 3, resolved at compile time
BVE_HD bool default_codings(const Params &p) { return p.c_outd == C_GAMMA && p.c_blk == C_GAMMA && p.c_res == C_ZETA && p.K == 3 && p.c_ref == C_UNARY && p.c_bc == C_GAMMA; }
template <bool DEF, class S> BVE_HD void f_outd(S &s, const Params &p, uint64_t x) { if (DEF) w_gamma(s, x); else w_code(s, p.c_outd, x, 0); }
template <bool DEF, class S> BVE_HD void f_ref(S &s, const Params &p, uint64_t x) { if (DEF) w_unary(s, x); else w_code(s, p.c_ref, x, 0); }
template <bool DEF, class S> BVE_HD void f_bc(S &s, const Params &p, uint64_t x) { if (DEF) w_gamma(s, x); else w_code(s, p.c_bc, x, 0); }
template <bool DEF, class S> BVE_HD void f_blk(S &s, const Params &p, uint64_t x) { if (DEF) w_gamma(s, x); else w_code(s, p.c_blk, x, 0); }
template <bool DEF, class S> BVE_HD void f_res(S &s, const Params &p, uint64_t x) { if (DEF) w_zeta(s, x, 3); else w_code(s, p.c_res, x, p.K); }

// ---- the walk.  V sees, in stream order within each section: block(run) for every copy block, interval(left, len) and
// residual(v) for the extras.  (Blocks, intervals and residuals are three sections of the record; a visitor that writes
// keeps a cursor per section.)
template <class V>
struct Extras { // intervalize (BVGraph.java:1631-1654) as a stream: maximal runs of consecutive extras of length >= max(2, I) are intervals
	V &v;
	int32_t I;
	int32_t runStart = 0, runLen = 0;
	BVE_HD Extras(V &v_, int32_t I_) : v(v_), I(I_) {}
	BVE_HD void flush() {
		if (runLen >= 2 && I != 0 && runLen >= I) v.interval(runStart, runLen);
		else for (int32_t t = 0; t < runLen; t++) v.residual(runStart + t);
		runLen = 0;
	}
	BVE_HD void add(int32_t x) {
		if (runLen > 0 && x == runStart + runLen) { runLen++; return; }
		flush();
		runStart = x; runLen = 1;
	}
};

// cur[0..d): the node's successors; ref[0..dr): the candidate's (dr = 0: no reference).  Returns the number of extras.
template <class V>
BVE_HD int32_t diff_walk(const int32_t *__restrict__ cur, int32_t d, const int32_t *__restrict__ ref, int32_t dr, int32_t I, V &v) {
	Extras<V> ex(v, I);
	int32_t j = 0, k = 0, run = 0, nextra = 0;
	bool copying = true;
	if (dr > 0 && d > 0) {
		int32_t c = cur[0], f = ref[0];
		for (;;) { // one comparison per turn (BVGraph.java:2072-2121)
			if (c == f) {
				if (copying) { j++; k++; run++; if (j == d || k == dr) break; c = cur[j]; f = ref[k]; }
				else { v.block(run); copying = true; run = 0; }
			} else if (c < f) {
				ex.add(c); nextra++; j++;
				if (j == d) break;
				c = cur[j];
			} else if (copying) { v.block(run); copying = false; run = 0; }
			else { k++; run++; if (k == dr) break; f = ref[k]; }
		}
		if (copying && k < dr) v.block(run);
	}
	for (; j < d; j++) { ex.add(cur[j]); nextra++; }
	ex.flush();
	return nextra;
}

// ---- visitors
// section sizes of one (node, candidate) description: what the forReal = false run of diffComp measures
template <bool DEF>
struct CountVisitor {
	const Params &p;
	int32_t node;
	uint64_t bitsB = 0, bitsI = 0, bitsR = 0;
	uint32_t nb = 0, ni = 0, nr = 0;
	int64_t prevEnd = 0, prevRes = 0;
	uint64_t ivArcs = 0;
	BVE_HD CountVisitor(const Params &p_, int32_t node_) : p(p_), node(node_) {}
	BVE_HD void block(int32_t run) { LenSink s; f_blk<DEF>(s, p, (uint64_t)(nb == 0 ? run : run - 1)); bitsB += s.bits; nb++; }
	BVE_HD void interval(int32_t left, int32_t len) {
		LenSink s;
		w_gamma(s, ni == 0 ? int2nat((int64_t)left - node) : (uint64_t)((int64_t)left - prevEnd - 1));
		w_gamma(s, (uint64_t)(len - p.I));
		bitsI += s.bits; ni++; prevEnd = (int64_t)left + len; ivArcs += (uint64_t)len;
	}
	BVE_HD void residual(int32_t x) {
		LenSink s;
		f_res<DEF>(s, p, nr == 0 ? int2nat((int64_t)x - node) : (uint64_t)((int64_t)x - prevRes - 1));
		bitsR += s.bits; nr++; prevRes = x;
	}
	// bits of the description after the outdegree (reference field, block section, interval section, residual section)
	BVE_HD uint64_t bits_ref(int r) const { LenSink s; if (p.W > 0) f_ref<DEF>(s, p, (uint64_t)r); return s.bits; }
	BVE_HD uint64_t bits_blocks(int r) const { LenSink s; if (r != 0) f_bc<DEF>(s, p, nb); return s.bits + (r != 0 ? bitsB : 0); }
	BVE_HD uint64_t bits_intervals(int32_t nextra) const { LenSink s; if (nextra > 0 && p.I != 0) w_gamma(s, ni); return s.bits + bitsI; }
	BVE_HD uint64_t total(int r, int32_t nextra) const { return bits_ref(r) + bits_blocks(r) + bits_intervals(nextra) + bitsR; }
};

// updateBins (BVGraph.java:1940-1944) for one residual, from the value its code carries: the first of a node by the most significant bit of int2nat(first - node)
// (nothing when that is 0), a later one by that of its distance from the one before (= coded value + 1).  `bins`: 32 counters of the block or wave, in LDS on the device.
BVE_HD void res_bin(unsigned long long *bins, bool first, uint64_t coded) {
	if (!bins) return;
	const uint64_t g = first ? coded : coded + 1;
	if (g == 0) return;
	const int b = msb64(g);
#if defined(__HIP_DEVICE_COMPILE__)
	atomicAdd(&bins[b < 31 ? b : 31], 1ull);
#else
	bins[b < 31 ? b : 31]++;
#endif
}

// writes the three sections through three cursors
template <bool DEF>
struct EmitVisitor {
	const Params &p;
	int32_t node;
	WordSink sB, sI, sR;
	uint32_t nb = 0, ni = 0, nr = 0;
	int64_t prevEnd = 0, prevRes = 0;
	unsigned long long *rbins; // the block's histogram of the residuals' gaps (res_bin), or null
	BVE_HD EmitVisitor(const Params &p_, int32_t node_, uint32_t *words, uint64_t posB, uint64_t posI, uint64_t posR, unsigned long long *rbins_)
	    : p(p_), node(node_), sB(words, posB), sI(words, posI), sR(words, posR), rbins(rbins_) {}
	BVE_HD void block(int32_t run) { f_blk<DEF>(sB, p, (uint64_t)(nb == 0 ? run : run - 1)); nb++; }
	BVE_HD void interval(int32_t left, int32_t len) {
		w_gamma(sI, ni == 0 ? int2nat((int64_t)left - node) : (uint64_t)((int64_t)left - prevEnd - 1));
		w_gamma(sI, (uint64_t)(len - p.I));
		ni++; prevEnd = (int64_t)left + len;
	}
	BVE_HD void residual(int32_t x) {
		const uint64_t v = nr == 0 ? int2nat((int64_t)x - node) : (uint64_t)((int64_t)x - prevRes - 1);
		f_res<DEF>(sR, p, v);
		res_bin(rbins, nr == 0, v);
		nr++; prevRes = x;
	}
	BVE_HD void finish() { sB.finish(); sI.finish(); sR.finish(); }
};

constexpr uint32_t COST_NONE = 0xffffffffu; // candidate not available (before the part's first node, or an empty list)
constexpr uint64_t COST_MAX = 0x7fffffffull; // a record of 2^31 bits or more is refused

// first node of the part x belongs to
BVE_HD int32_t part_lo(const Params &p, int32_t x) { return p.per > 0 ? (x / p.per) * p.per : 0; }

// Phase A for one pair: cost in bits of describing x through x - r (r = 0: no reference), COST_NONE if the pair is not a
// candidate whatever the chain lengths are (BVGraph.java:2313-2327 also asks refCount < maxRefCount: phase B).
template <bool DEF>
BVE_HD uint32_t pair_cost(const Params &p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int32_t x, int r, int *err) {
	const int64_t a = rowptr[x];
	const int32_t d = (int32_t)(rowptr[x + 1] - a);
	if (d == 0) return COST_NONE;
	const int32_t y = x - r;
	if (y < part_lo(p, x)) return COST_NONE;
	const int64_t b = rowptr[y];
	const int32_t dr = r == 0 ? 0 : (int32_t)(rowptr[y + 1] - b);
	if (r != 0 && dr == 0) return COST_NONE;
	if (r == 0) for (int32_t j = 1; j < d; j++) if (succ[a + j] <= succ[a + j - 1]) *err |= 1; // the lists must increase strictly (every non-empty list has this pair)
	if (r == 0 && (succ[a] < 0 || succ[a + d - 1] == INT32_MAX)) *err |= 1; // node ids: never negative, and 2^31 - 1 is the padding of the wave path's tiles
	CountVisitor<DEF> v(p, x);
	const int32_t nextra = diff_walk(succ + a, d, succ + b, dr, p.I, v);
	const uint64_t t = v.total(r, nextra);
	if (t > COST_MAX) { *err |= 2; return COST_NONE; }
	return (uint32_t)t;
}

// Phase B for the nodes [lo, hi) of one chunk: the reference's choice (first candidate of minimal cost among those whose
// chain is shorter than maxRefCount, :2313-2327) given the chain lengths of the W nodes before lo in `inState`
// (inState[W - t] = chain length of node lo - t).  Writes best[x] (0 = none) and refc[x] for every node with successors.
struct alignas(16) Cost4 { uint32_t v[4]; };
BVE_HD Cost4 load_cost4(const uint32_t *q) { // q is 16-byte aligned on the device (hipMalloc + rows of 32 bytes); the host model may get any pointer
#if defined(__HIP_DEVICE_COMPILE__)
	return *(const Cost4 *)q;
#else
	Cost4 c;
	for (int i = 0; i < 4; i++) c.v[i] = q[i];
	return c;
#endif
}
// windowsize 7 (the reference's default: 8 candidates, a row of prices is two 16-byte loads): the chain lengths of the last 7 nodes
// live in registers and the next node's prices are on their way while this node chooses -- the walk is a chain of dependent steps,
// and with a global load per candidate it took about a microsecond per node.  A node without successors is one whose r = 0 price
// is COST_NONE.  Same choices as the general loop below.
BVE_HD void select_chunk_w7(const Params &p, const uint32_t *__restrict__ cost, int32_t lo, int32_t hi, const int32_t *__restrict__ inState, uint8_t *__restrict__ best,
                            int32_t *__restrict__ refc) {
	int32_t w0 = inState[6], w1 = inState[5], w2 = inState[4], w3 = inState[3], w4 = inState[2], w5 = inState[1], w6 = inState[0]; // w_t: chain length of node x - 1 - t
	const int32_t R = p.R;
	if (lo >= hi) return;
	Cost4 a = load_cost4(cost + (int64_t)lo * 8), b = load_cost4(cost + (int64_t)lo * 8 + 4);
	for (int32_t x = lo; x < hi; x++) {
		const Cost4 ca = a, cb = b;
		if (x + 1 < hi) { a = load_cost4(cost + (int64_t)(x + 1) * 8); b = load_cost4(cost + (int64_t)(x + 1) * 8 + 4); }
		int bestR = 0;
		int32_t rc = 0;
		if (ca.v[0] != COST_NONE) {
			uint32_t bc = ca.v[0];
			if (ca.v[1] < bc && w0 < R) { bc = ca.v[1]; bestR = 1; rc = w0 + 1; }
			if (ca.v[2] < bc && w1 < R) { bc = ca.v[2]; bestR = 2; rc = w1 + 1; }
			if (ca.v[3] < bc && w2 < R) { bc = ca.v[3]; bestR = 3; rc = w2 + 1; }
			if (cb.v[0] < bc && w3 < R) { bc = cb.v[0]; bestR = 4; rc = w3 + 1; }
			if (cb.v[1] < bc && w4 < R) { bc = cb.v[1]; bestR = 5; rc = w4 + 1; }
			if (cb.v[2] < bc && w5 < R) { bc = cb.v[2]; bestR = 6; rc = w5 + 1; }
			if (cb.v[3] < bc && w6 < R) { bc = cb.v[3]; bestR = 7; rc = w6 + 1; }
		}
		best[x] = (uint8_t)bestR;
		refc[x] = rc;
		w6 = w5; w5 = w4; w4 = w3; w3 = w2; w2 = w1; w1 = w0; w0 = rc;
	}
}

BVE_HD void select_chunk(const Params &p, const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ cost, int32_t lo, int32_t hi,
                         const int32_t *__restrict__ inState, uint8_t *__restrict__ best, int32_t *__restrict__ refc) {
	if (p.W == 7) { select_chunk_w7(p, cost, lo, hi, inState, best, refc); return; }
	const int cyc = p.W + 1;
	for (int32_t x = lo; x < hi; x++) {
		if (rowptr[x + 1] == rowptr[x]) { best[x] = 0; refc[x] = 0; continue; }
		const uint32_t *c = cost + (int64_t)x * cyc;
		uint32_t bestCost = c[0];
		int bestR = 0;
		for (int r = 1; r < cyc; r++) {
			const uint32_t t = c[r];
			if (t >= bestCost) continue; // COST_NONE included
			const int32_t y = x - r;
			const int32_t rc = y >= lo ? refc[y] : inState[p.W - (lo - y)];
			if (rc < p.R) { bestCost = t; bestR = r; }
		}
		const int32_t y = x - bestR;
		best[x] = (uint8_t)bestR;
		refc[x] = bestR == 0 ? 0 : (y >= lo ? refc[y] : inState[p.W - (lo - y)]) + 1;
	}
}

// Phase B, one round for the chunks [c0, c1) of `chunk` nodes each, walked in order by one lane.  state*[c] holds the chain
// lengths of the last W nodes of chunk c as of the previous / this round, used[c] the in-state chunk c last ran with: a
// chunk whose in-state is what it last ran with is skipped (its results stand).  The in-state of c0 comes from the
// previous round (zeros in round 0: a guess), that of the others from the walk itself.  `in`: scratch of W values.
// Returns whether some chunk's final state moved.  After a round in which nothing moved, every chunk ran with the true
// final state of its predecessor (chunk 0 always does), so best[] / refc[] are the sequential compressor's.
BVE_HD bool select_span(const Params &p, const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ cost, int32_t n, int32_t chunk, int64_t c0, int64_t c1, int round,
                        const int32_t *__restrict__ statePrev, int32_t *__restrict__ stateNew, int32_t *__restrict__ used, uint8_t *__restrict__ best,
                        int32_t *__restrict__ refc, int32_t *in) {
	const int W = p.W;
	for (int t = 0; t < W; t++) in[t] = c0 > 0 && round > 0 ? statePrev[(c0 - 1) * W + t] : 0;
	bool moved = false;
	for (int64_t c = c0; c < c1; c++) {
		bool same = round > 0;
		for (int t = 0; t < W && same; t++) same = used[c * W + t] == in[t];
		if (same) {
			for (int t = 0; t < W; t++) { const int32_t v = statePrev[c * W + t]; stateNew[c * W + t] = v; in[t] = v; }
			continue;
		}
		for (int t = 0; t < W; t++) used[c * W + t] = in[t];
		const int32_t lo = (int32_t)(c * chunk), hi = (int32_t)((int64_t)lo + chunk < n ? lo + chunk : n);
		select_chunk(p, rowptr, cost, lo, hi, in, best, refc);
		for (int t = 1; t <= W; t++) {
			const int32_t y = hi - t;
			const int32_t v = y >= lo ? refc[y] : (lo - y <= W ? in[W - (lo - y)] : 0);
			if (round == 0 || statePrev[c * W + (W - t)] != v) moved = true;
			stateNew[c * W + (W - t)] = v;
		}
		for (int t = 0; t < W; t++) in[t] = stateNew[c * W + t];
	}
	return moved;
}

// What a node contributes to the counters of the .properties file (BVGraph.java:2558-2600)
struct NodeStats {
	uint64_t bitsOutd = 0, bitsRef = 0, bitsBlocks = 0, bitsIntervals = 0, bitsResiduals = 0;
	uint64_t copied = 0, intervalised = 0, residuals = 0;
};

// Phase D for one node: write the record at bit `pos` of `words` (zeroed beforehand).  Returns its length in bits.
template <bool DEF>
BVE_HD uint64_t emit_node(const Params &p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int32_t x, int r, uint32_t *words,
                          uint64_t pos, NodeStats *st, unsigned long long *rbins = nullptr) {
	const int64_t a = rowptr[x];
	const int32_t d = (int32_t)(rowptr[x + 1] - a);
	WordSink head(words, pos);
	f_outd<DEF>(head, p, (uint64_t)d);
	const uint64_t afterOutd = head.pos;
	if (st) st->bitsOutd = afterOutd - pos;
	if (d == 0) { head.finish(); return afterOutd - pos; }
	const int32_t y = x - r;
	const int64_t b = rowptr[y];
	const int32_t dr = r == 0 ? 0 : (int32_t)(rowptr[y + 1] - b);
	// sizes of the sections first: the counts precede the items in the stream
	CountVisitor<DEF> cv(p, x);
	const int32_t nextra = diff_walk(succ + a, d, succ + b, dr, p.I, cv);
	if (p.W > 0) f_ref<DEF>(head, p, (uint64_t)r);
	const uint64_t afterRef = head.pos;
	if (r != 0) f_bc<DEF>(head, p, cv.nb);
	const uint64_t posB = head.pos;
	head.finish();
	const uint64_t startI = posB + (r != 0 ? cv.bitsB : 0);
	WordSink ic(words, startI);
	if (nextra > 0 && p.I != 0) w_gamma(ic, cv.ni);
	const uint64_t posI = ic.pos;
	ic.finish();
	const uint64_t posR = posI + cv.bitsI;
	EmitVisitor<DEF> ev(p, x, words, posB, posI, posR, rbins);
	(void)diff_walk(succ + a, d, succ + b, dr, p.I, ev);
	ev.finish();
	const uint64_t end = posR + cv.bitsR;
	if (st) {
		st->bitsRef = afterRef - afterOutd;
		st->bitsBlocks = startI - afterRef;
		st->bitsIntervals = posR - startI;
		st->bitsResiduals = cv.bitsR;
		st->copied = (uint64_t)(d - nextra);
		st->intervalised = cv.ivArcs;
		st->residuals = cv.nr;
	}
	return end - pos;
}

} // namespace bve
