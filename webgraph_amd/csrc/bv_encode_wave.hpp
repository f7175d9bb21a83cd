// bv_encode_wave.hpp -- the compressor's list walk (bv_encode.hpp: diff_walk + its visitors) done by a whole wave, for
// pairs too long for one lane.  Device only (gfx950, wave64).
//
// diffComp walks the node's list `cur` and the candidate's list `ref` in step (BVGraph.java:2072-2121).  What comes out of
// that walk depends on two bit vectors only: for every element of ref, is it in cur (the copy blocks are the run lengths of
// that vector, last run dropped); for every element of cur, is it in ref (those that are not are the extras, whose maximal
// runs of consecutive integers of length >= minIntervalLength become intervals, the others residuals).  A wave takes the
// next 64 elements of each list, settles the elements not above the smaller of the two tiles' last values -- their
// membership is decided inside the tiles, by a binary search over the lanes -- and derives blocks, intervals and residuals
// of that stretch with ballots; what a stretch needs from the stretches before it (the run in progress on either side, the
// previous interval's end, the previous residual) travels in a few wave-uniform values.  Every round consumes at least one
// whole tile.
//
// COUNT mode adds up the section sizes (the forReal = false run of diffComp); EMIT mode writes the codes: lengths ->
// exclusive scan over the lanes -> every lane ORs its codes in at its own bit position.
#pragma once
#include "bv_encode.hpp"

namespace bvw {

using bve::Params;

__device__ __forceinline__ uint64_t lt_mask(int j) { return j >= 64 ? ~0ull : ((1ull << j) - 1); } // lanes below j
__device__ __forceinline__ uint64_t le_mask(int j) { return j >= 63 ? ~0ull : ((2ull << j) - 1); } // lanes up to j
__device__ __forceinline__ int hibit(uint64_t m) { return 63 - __builtin_clzll(m); }               // m != 0
__device__ __forceinline__ int lobit(uint64_t m) { return __builtin_ctzll(m); }                     // m != 0

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int lane) {
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const uint64_t t = __shfl_up(v, o); if (lane >= o) v += t; }
	return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) { // the default codings: 64 codes are far below 2^32 bits (a unary code may not be)
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(v, o); if (lane >= o) v += t; }
	return v;
}
__device__ __forceinline__ uint64_t wave_total(uint64_t v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

// is `key` among the 64 sorted values the lanes hold in `tile` (lanes past the end hold INT32_MAX)?
__device__ __forceinline__ bool lane_search(int32_t key, int32_t tile) {
	int lo = 0, hi = 64; // lower bound: 7 halvings empty a range of 64
#pragma unroll
	for (int s = 0; s < 7; s++) {
		const int mid = (lo + hi) >> 1;
		const int32_t v = __shfl(tile, mid & 63);
		if (lo < hi) { if (v < key) lo = mid + 1; else hi = mid; }
	}
	const int32_t v = __shfl(tile, lo & 63);
	return lo < 64 && v == key;
}

struct WaveTotals { // what CountVisitor holds after a walk (the same in every lane)
	uint64_t bitsB, bitsI, bitsR, ivArcs;
	uint32_t nb, ni, nr;
	int32_t nextra;
	int bad; // a list that is not strictly increasing
};

// SEG (with EMIT = false): the walk is over a SEGMENT of a pair whose predecessors' state is not known yet: it starts in the state of
// its own first element and prices every item but the three that depend on what came before -- the block that ends at the first
// change of the membership flag, the first residual, the left extreme of the first interval -- and reports those instead.
template <bool DEF, bool EMIT, bool SEG = false>
struct WaveWalk {
	const Params &p;
	const int32_t node;
	const int lane;
	uint32_t *words;
	uint64_t posB, posI, posR;                 // EMIT: cursors of the three sections
	const uint64_t posB0, posI0, posR0;
	uint64_t accB = 0, accI = 0, accR = 0, accArcs = 0; // COUNT: per-lane partial sums
	uint32_t nb = 0, ni = 0, nr = 0;
	int64_t prevEnd = 0, prevRes = 0;
	uint32_t prevFlag = 1;   // the walk starts in a copy run ...
	int64_t runStart = 0;    // ... that began at index 0 of ref (a later segment of a cut pair: where the run in progress began, relative to the segment)
	unsigned long long *rbins = nullptr; // EMIT: the block's histogram of the residuals' gaps (bve::res_bin), set by the caller
	int32_t firstFlag = -1;  // SEG: membership flag of the segment's first element of ref (-1: none), index of the first change, first residual, first interval
	int64_t firstChange = -1, firstRes = 0, firstLeft = 0;

	__device__ __forceinline__ WaveWalk(const Params &p_, int32_t node_, uint32_t *w, uint64_t pb, uint64_t pi, uint64_t pr)
	    : p(p_), node(node_), lane((int)(threadIdx.x & 63)), words(w), posB(pb), posI(pi), posR(pr), posB0(pb), posI0(pi), posR0(pr) {}

	// the lanes with `on` hold one block each (`val`, already the value to code), in lane order
	__device__ __forceinline__ void blocks(bool on, uint64_t val) {
		bve::LenSink s;
		if (on) bve::f_blk<DEF>(s, p, val);
		if (!EMIT) { accB += s.bits; return; }
		const uint64_t inc = DEF ? (uint64_t)wave_incl_scan((uint32_t)s.bits, lane) : wave_incl_scan(s.bits, lane);
		if (on) { bve::WordSink w(words, posB + inc - s.bits); bve::f_blk<DEF>(w, p, val); w.finish(); }
		posB += __shfl(inc, 63);
	}
	// ... one interval each: (v1, v2) = (coded left extreme, length - minIntervalLength)
	__device__ __forceinline__ void intervals(bool on, uint64_t v1, uint64_t v2, bool noV1 = false) { // noV1 (SEG): the left extreme is priced by whoever knows the previous interval
		bve::LenSink s;
		if (on) { if (!noV1) bve::w_gamma(s, v1); bve::w_gamma(s, v2); }
		if (!EMIT) { accI += s.bits; return; }
		const uint64_t inc = DEF ? (uint64_t)wave_incl_scan((uint32_t)s.bits, lane) : wave_incl_scan(s.bits, lane);
		if (on) { bve::WordSink w(words, posI + inc - s.bits); bve::w_gamma(w, v1); bve::w_gamma(w, v2); w.finish(); }
		posI += __shfl(inc, 63);
	}
	// ... one residual each (`first`: the lane's is the node's first residual -- for the histogram of the gaps only)
	__device__ __forceinline__ void residuals(bool on, uint64_t val, bool first) {
		bve::LenSink s;
		if (on) bve::f_res<DEF>(s, p, val);
		if (!EMIT) { accR += s.bits; return; }
		const uint64_t inc = DEF ? (uint64_t)wave_incl_scan((uint32_t)s.bits, lane) : wave_incl_scan(s.bits, lane);
		if (on) { bve::WordSink w(words, posR + inc - s.bits); bve::f_res<DEF>(w, p, val); w.finish(); bve::res_bin(rbins, first, val); }
		posR += __shfl(inc, 63);
	}

	// a run of T consecutive extras start .. start + T - 1 that began in an earlier round ends here (wave-uniform arguments)
	__device__ __forceinline__ void close_run(int64_t start, int64_t T) {
		if (T >= 2 && T >= p.I) { // p.I != 0: runs are only carried when intervals exist
			const uint64_t v1 = ni == 0 ? bve::int2nat(start - node) : (uint64_t)(start - prevEnd - 1);
			intervals(lane == 0, v1, (uint64_t)(T - p.I), SEG && ni == 0);
			if (SEG && ni == 0) firstLeft = start;
			if (lane == 0) accArcs += (uint64_t)T;
			prevEnd = start + T; ni++;
		} else { // T residuals: the first one's gap, then gaps of 0
			const uint64_t v0 = nr == 0 ? bve::int2nat(start - node) : (uint64_t)(start - prevRes - 1);
			const bool skipFirst = SEG && nr == 0;
			if (skipFirst) firstRes = start;
			for (int64_t t0 = 0; t0 < T; t0 += 64) residuals(t0 + lane < T && !(skipFirst && t0 + lane == 0), t0 + lane == 0 ? v0 : 0, nr == 0 && t0 + lane == 0);
			prevRes = start + T - 1; nr += (uint32_t)T;
		}
	}

	__device__ __forceinline__ void run(const int32_t *__restrict__ cur, int32_t d, const int32_t *__restrict__ ref, int32_t dr, WaveTotals &tot) {
		const int32_t I = p.I;
		int64_t j0 = 0, k0 = 0;
		int64_t openStart = 0, openLen = 0; // run of consecutive extras in progress at the end of the consumed part of cur
		int64_t nextra = 0;
		int64_t lastA = INT64_MIN; // last consumed element of cur: the list must increase strictly
		int bad = 0;
		int32_t a = lane < d ? cur[lane] : INT32_MAX, b = lane < dr ? ref[lane] : INT32_MAX;
		while (j0 < d || k0 < dr) {
			const int na = d - j0 < 64 ? (int)(d - j0) : 64, nv = dr - k0 < 64 ? (int)(dr - k0) : 64;
			const int32_t boundA = j0 + 64 < d ? __shfl(a, 63) : INT32_MAX;
			const int32_t boundB = k0 + 64 < dr ? __shfl(b, 63) : INT32_MAX;
			const int32_t limit = boundA < boundB ? boundA : boundB;
			const uint64_t VA = __ballot(lane < na && a <= limit), VB = __ballot(lane < nv && b <= limit);
			const int ca = __popcll(VA), cb = __popcll(VB);
			// the next tiles are known now: their loads fly while this stretch is worked on
			const int32_t an = j0 + ca + lane < d ? cur[j0 + ca + lane] : INT32_MAX, bn = k0 + cb + lane < dr ? ref[k0 + cb + lane] : INT32_MAX;
			const bool both = na > 0 && nv > 0; // (wave-uniform) nothing to look up in an empty tile
			const bool inRef = both && lane_search(a, b), inCur = both && lane_search(b, a);
			const uint64_t E = __ballot(!inRef) & VA, M = __ballot(inCur) & VB;
			const int32_t ap = __shfl_up(a, 1);
			if (__ballot(lane < na && (lane == 0 ? (int64_t)a <= lastA : a <= ap))) bad = 1;
			if (ca) lastA = __shfl(a, ca - 1);

			// ---- copy blocks: one per change of the membership flag along ref
			if (cb > 0) {
				const uint64_t maskB = lt_mask(cb);
				if (SEG && firstFlag < 0) { firstFlag = (int32_t)(M & 1); prevFlag = (uint32_t)firstFlag; } // no change at the segment's own start
				const uint64_t Bd = (M ^ ((M << 1) | prevFlag)) & maskB;
				if (Bd) {
					const uint64_t below = Bd & lt_mask(lane);
					const int64_t len = k0 + lane - (below ? k0 + hibit(below) : runStart);
					const bool firstInternal = SEG && nb == 0 && !below; // its length depends on where the run began: priced by the stitching
					if (SEG && nb == 0) firstChange = k0 + lobit(Bd);
					blocks(((Bd >> lane) & 1) && !firstInternal, (uint64_t)(nb == 0 && !below ? len : len - 1));
					nb += (uint32_t)__popcll(Bd);
					runStart = k0 + hibit(Bd);
				}
				prevFlag = (uint32_t)((M >> (cb - 1)) & 1);
			}

			// ---- extras
			if (ca > 0) {
				const bool finA = j0 + ca == d;
				nextra += __popcll(E);
				uint64_t link = 0; // lane is an extra, and so is its predecessor, one less
				if (I != 0) {
					const bool lk = lane == 0 ? (openLen > 0 && (int64_t)a == openStart + openLen) : (int64_t)a == (int64_t)ap + 1;
					link = __ballot(lk) & E & ((E << 1) | (openLen > 0 ? 1ull : 0ull));
				}
				int c = 0; // leading lanes that continue the run in progress
				bool whole = false;
				if (openLen > 0) {
					c = ~link ? lobit(~link) : 64;
					if (c == ca && !finA) { openLen += ca; whole = true; } // the whole stretch lies inside it
					else { close_run(openStart, openLen + c); openLen = 0; }
				}
				if (!whole) {
					uint64_t R = E & ~lt_mask(c);  // extras of runs that start in this stretch
					uint64_t S = R & ~link;        // their first elements
					if (I != 0 && !finA && ((R >> (ca - 1)) & 1)) { // the run that reaches the end of the stretch stays in progress
						const int s = hibit(S);
						openStart = __shfl(a, s); openLen = ca - s;
						R &= lt_mask(s); S &= lt_mask(s);
					}
					if (R) {
						const bool inR = (R >> lane) & 1;
						const uint64_t sb = S & le_mask(lane);
						const int sj = sb ? hibit(sb) : 0;
						const uint64_t brk = (~link | ~R) & ~le_mask(lane);
						const int ej = brk ? lobit(brk) - 1 : 63;
						const int len = ej - sj + 1;
						const bool iv = I != 0 && len >= 2 && len >= I;
						const uint64_t IS = __ballot(inR && lane == sj && iv), RS = __ballot(inR && !iv);
						if (IS) {
							const uint64_t below = IS & lt_mask(lane);
							const int pl = below ? hibit(below) : 0;
							const int64_t pe = (int64_t)__shfl(a, pl) + __shfl(len, pl); // (every lane takes part in a shuffle: not inside the conditional)
							const int64_t pend = below ? pe : prevEnd;
							const bool on = (IS >> lane) & 1;
							const uint64_t v1 = ni == 0 && !below ? bve::int2nat((int64_t)a - node) : (uint64_t)((int64_t)a - pend - 1);
							if (SEG && ni == 0) firstLeft = __shfl(a, lobit(IS));
							intervals(on, v1, (uint64_t)(len - I), SEG && ni == 0 && !below);
							if (on) accArcs += (uint64_t)len;
							const int last = hibit(IS);
							prevEnd = (int64_t)__shfl(a, last) + __shfl(len, last);
							ni += (uint32_t)__popcll(IS);
						}
						if (RS) {
							const uint64_t below = RS & lt_mask(lane);
							const int pl = below ? hibit(below) : 0;
							const int64_t pa = __shfl(a, pl);
							const int64_t prv = below ? pa : prevRes;
							const uint64_t v = nr == 0 && !below ? bve::int2nat((int64_t)a - node) : (uint64_t)((int64_t)a - prv - 1);
							if (SEG && nr == 0) firstRes = __shfl(a, lobit(RS));
							residuals(((RS >> lane) & 1) && !(SEG && nr == 0 && !below), v, nr == 0 && !below);
							prevRes = __shfl(a, hibit(RS));
							nr += (uint32_t)__popcll(RS);
						}
					}
				}
			}
			j0 += ca; k0 += cb;
			a = an; b = bn;
		}
		if (EMIT) { tot.bitsB = posB - posB0; tot.bitsI = posI - posI0; tot.bitsR = posR - posR0; }
		else { tot.bitsB = wave_total(accB); tot.bitsI = wave_total(accI); tot.bitsR = wave_total(accR); }
		tot.ivArcs = wave_total(accArcs);
		tot.nb = nb; tot.ni = ni; tot.nr = nr; tot.nextra = (int32_t)nextra; tot.bad = bad;
	}
};

// the cost of one pair, as bve::pair_cost computes it, by a wave (the pair is known to be a candidate)
template <bool DEF>
__device__ __forceinline__ uint32_t wave_pair_cost(const Params &p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int32_t x, int r, int *err, WaveTotals &t) {
	const int64_t a = rowptr[x], b = rowptr[x - r];
	const int32_t d = (int32_t)(rowptr[x + 1] - a), dr = r == 0 ? 0 : (int32_t)(rowptr[x - r + 1] - b);
	WaveWalk<DEF, false> w(p, x, nullptr, 0, 0, 0);
	w.run(succ + a, d, succ + b, dr, t);
	bve::LenSink s;
	if (p.W > 0) bve::f_ref<DEF>(s, p, (uint64_t)r);
	if (r != 0) bve::f_bc<DEF>(s, p, t.nb);
	if (t.nextra > 0 && p.I != 0) bve::w_gamma(s, t.ni);
	const uint64_t total = s.bits + (r != 0 ? t.bitsB : 0) + t.bitsI + t.bitsR;
	if (t.bad || (r == 0 && d > 0 && (succ[a] < 0 || succ[a + d - 1] == INT32_MAX))) *err |= 1; // (as pair_cost: ids in [0, 2^31 - 1))
	if (total > bve::COST_MAX) { *err |= 2; return bve::COST_NONE; }
	return (uint32_t)total;
}

// what the pricing of a pair leaves for its emission: the sizes that precede the items in the stream
struct PairInfo { uint32_t nb, bitsB, ni, bitsI; }; // ni: bit 31 = the list has extras (an interval count is written)

// the record of node x written by a wave, as bve::emit_node does, given what the pricing of the pair (x, r) found;
// `st` is filled in lane 0 only
template <bool DEF>
__device__ __forceinline__ void wave_emit_node(const Params &p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int32_t x, int r, const PairInfo info,
                                               uint32_t *words, uint64_t pos, bve::NodeStats &st, unsigned long long *rbins) {
	const int lane = (int)(threadIdx.x & 63);
	const int64_t a = rowptr[x], b = rowptr[x - r];
	const int32_t d = (int32_t)(rowptr[x + 1] - a), dr = r == 0 ? 0 : (int32_t)(rowptr[x - r + 1] - b);
	const uint32_t ni = info.ni & 0x7fffffffu;
	const bool extras = (info.ni >> 31) != 0;
	bve::LenSink h;
	bve::f_outd<DEF>(h, p, (uint64_t)d);
	const uint64_t afterOutd = pos + h.bits;
	if (p.W > 0) bve::f_ref<DEF>(h, p, (uint64_t)r);
	const uint64_t afterRef = pos + h.bits;
	if (r != 0) bve::f_bc<DEF>(h, p, info.nb);
	const uint64_t posB = pos + h.bits;
	const uint64_t startI = posB + (r != 0 ? info.bitsB : 0);
	bve::LenSink ic;
	if (extras && p.I != 0) bve::w_gamma(ic, ni);
	const uint64_t posI = startI + ic.bits, posR = posI + info.bitsI;
	if (lane == 0) {
		bve::WordSink hw(words, pos);
		bve::f_outd<DEF>(hw, p, (uint64_t)d);
		if (p.W > 0) bve::f_ref<DEF>(hw, p, (uint64_t)r);
		if (r != 0) bve::f_bc<DEF>(hw, p, info.nb);
		hw.finish();
		if (extras && p.I != 0) { bve::WordSink wi(words, startI); bve::w_gamma(wi, ni); wi.finish(); }
	}
	WaveWalk<DEF, true> w(p, x, words, posB, posI, posR);
	w.rbins = rbins;
	WaveTotals t;
	w.run(succ + a, d, succ + b, dr, t);
	if (lane == 0) {
		st.bitsOutd = afterOutd - pos; st.bitsRef = afterRef - afterOutd; st.bitsBlocks = startI - afterRef; st.bitsIntervals = posR - startI;
		st.bitsResiduals = t.bitsR; st.copied = (uint64_t)(d - t.nextra); st.intervalised = t.ivArcs; st.residuals = t.nr;
	}
}

} // namespace bvw
