// bv_encode.hip -- BVGraph.store on the GPU (gfx950): a CSR in HBM -> the .graph bit stream, the bit offsets and the
// .offsets stream, byte for byte what the reference's single-threaded compressor writes.
//
// The reference compresses node after node: per node, up to W + 1 runs of diffComp against a bit-counting stream, the
// cheapest admissible one again for real (CompressionThread.call, BVGraph.java:2222-2386).  Only the admissibility --
// the length of the reference chain, :2313-2327 -- links a node to its predecessors; the W + 1 costs do not.  So:
//   A  k_enc_cost    one lane per (node, candidate) pair: the pair's cost in bits.  The pairs are taken from a list
//                    grouped by size (k_enc_hist / k_enc_scatter), biggest first.                    [the bulk of the work]
//   B  k_enc_select  the chain-length recurrence, cut into chunks of SEL_CHUNK nodes: every chunk runs from a guessed
//                    state of the W nodes before it, then again only if its predecessor's final state turned out
//                    different.  On a copy-model graph the choice forgets its past within a few nodes (a node that
//                    takes no reference, or whose cheapest candidate is admissible either way); on a web graph runs of
//                    similar pages carry the phase of their chains for thousands of nodes (cnr-2000: 7 000), so after
//                    round 0 a lane walks SEL_SPAN chunks in order (twice as many after every batch of rounds that
//                    did not settle), skipping those whose in-state did not move.  Exact:
//                    the loop runs until a round moves nothing (bve::select_span).
//   C  k_enc_reclen + scan: record lengths -> bit offsets (what the reference's .offsets file holds).
//   D  k_enc_emit    one lane per node writes its record at its offset; words shared by two records are ORed.
//   E  the .offsets stream (gamma / delta coded gaps) the same way: lengths, scan, emit.
// The per-node logic is bv_encode.hpp, shared with the host model that the CPU tests compare with the CPU writer.
#include "bv_encode.hpp"
#include "bv_encode_wave.hpp"
#include "bv_launch.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace bv {

using bve::Params;

constexpr int SEL_CHUNK = 64; // nodes per chunk of the selection recurrence
constexpr int SEL_SPAN = 16, SEL_BATCH = 8;
constexpr int ENC_MAX_W = 63; // state of a chunk boundary: W chain lengths

struct EncStatsDev { // bitsOutd, bitsRef, bitsBlocks, bitsIntervals, bitsResiduals, copied, intervalised, residuals, totRef, totDist, maxRef, -
	unsigned long long v[12];
	unsigned long long resBins[32];                  // gaps between residuals by their most significant bit (bve::res_bin; residualGapStats, BVGraph.java:2196)
	unsigned long long succBins[32], succBinsOff[32]; // the same over whole successor lists (successorGapStats, :2303): counted over the array, less what k_enc_succ_bins_rows takes back
};

// the histogram of a block, folded into the global one
__device__ __forceinline__ void flush_bins(const unsigned long long *s_bins, unsigned long long *bins) {
	__syncthreads();
	if (threadIdx.x < 32 && s_bins[threadIdx.x]) atomicAdd(&bins[threadIdx.x], s_bins[threadIdx.x]);
}

// successorGapStats (updateBins, BVGraph.java:1940-1944) in two sweeps.  Over the ARRAY of successors: every positive difference between neighbours is binned, row starts included ...
constexpr int SB_GRID = 2048;
__global__ void __launch_bounds__(256) k_enc_succ_bins(const int32_t *__restrict__ succ, int64_t m, EncStatsDev *__restrict__ stats) {
	__shared__ unsigned long long s_b[32];
	if (threadIdx.x < 32) s_b[threadIdx.x] = 0;
	__syncthreads();
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x + 1; i < m; i += (int64_t)SB_GRID * 256) {
		const int64_t d = (int64_t)succ[i] - succ[i - 1];
		if (d > 0) atomicAdd(&s_b[bve::msb64((uint64_t)d)], 1ull);
	}
	flush_bins(s_b, stats->succBins);
}
// ... over the ROWS: what the first sweep counted across a row's start is taken back (succBinsOff), and the row's first successor is binned by int2nat(first - node)
__global__ void __launch_bounds__(256) k_enc_succ_bins_rows(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int32_t n, EncStatsDev *__restrict__ stats) {
	__shared__ unsigned long long s_b[64];
	if (threadIdx.x < 64) s_b[threadIdx.x] = 0;
	__syncthreads();
	for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < n; x += (int64_t)SB_GRID * 256) {
		const int64_t a = rowptr[x];
		if (rowptr[x + 1] == a) continue;
		const int32_t f = succ[a];
		if (a > 0) { const int64_t d = (int64_t)f - succ[a - 1]; if (d > 0) atomicAdd(&s_b[32 + bve::msb64((uint64_t)d)], 1ull); }
		const uint64_t g = bve::int2nat((int64_t)f - x);
		if (g) atomicAdd(&s_b[bve::msb64(g) < 31 ? bve::msb64(g) : 31], 1ull);
	}
	flush_bins(s_b, stats->succBins);
	if (threadIdx.x < 32 && s_b[32 + threadIdx.x]) atomicAdd(&stats->succBinsOff[threadIdx.x], s_b[32 + threadIdx.x]);
}

// a CSR handed over in device memory is checked like one from the host: rowptr must not decrease (every list then lies inside succ[0, rowptr[n]))
__global__ void __launch_bounds__(256) k_enc_check_rowptr(const int64_t *__restrict__ rowptr, int32_t n, int *__restrict__ err) {
	const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (x < n && (rowptr[x + 1] < rowptr[x] || rowptr[x + 1] - rowptr[x] > 0x7fffffffll)) atomicOr(err, 8);
}

// ---- work lists: the items of a phase (pairs, nodes) grouped by the log2 of their size, biggest first.  A lane walks its
// item alone, so a wave lasts as long as its longest item: waves of like-sized items waste no lane-time, and the long ones
// start first.  Within a bin the items stay in (nearly) node order: the pairs of a node share its successor list.
constexpr int ENC_NBIN = 32, SORT_ITEMS = 8, SORT_TILE = 256 * SORT_ITEMS;
constexpr int BIG_BIN = 8;         // items of 2^(BIG_BIN - 1) = 128 elements or more are walked by a wave
constexpr int WAVE_BLOCKS = 2048; // 4 waves each, striding over the head of the list
__device__ __forceinline__ int size_bin(uint64_t s) { return s == 0 ? 0 : (s >> 30 ? 31 : 32 - __clz((uint32_t)s)); }

struct PairItems { // item q <-> (node q / (W + 1), candidate q % (W + 1)); size = successors of the node + successors of the candidate
	Params p;
	const int64_t *rowptr;
	int64_t count;
	__device__ __forceinline__ int bin(int64_t q) const {
		const int cyc = p.W + 1;
		const int32_t x = (int32_t)(q / cyc);
		const int r = (int)(q - (int64_t)x * cyc);
		const int64_t a = rowptr[x], d = rowptr[x + 1] - a;
		const int32_t y = x - r;
		if (d == 0 || y < bve::part_lo(p, x)) return -1;
		const int64_t dr = r == 0 ? 0 : rowptr[y + 1] - rowptr[y];
		if (r != 0 && dr == 0) return -1;
		return size_bin((uint64_t)(d + dr));
	}
};
struct NodeItems { // item x <-> node x; size = its successors + those of the chosen reference
	Params p;
	const int64_t *rowptr;
	const uint8_t *best;
	int64_t count;
	int bigBin; // nodes whose chosen pair is that big are written by the waves, straight from the list of pairs
	__device__ __forceinline__ int bin(int64_t x) const {
		const int64_t d = rowptr[x + 1] - rowptr[x];
		const int r = d ? best[x] : 0;
		const int b = size_bin((uint64_t)(d + (r ? rowptr[x - r + 1] - rowptr[x - r] : 0)));
		return b >= bigBin ? -1 : b;
	}
};

template <class Items>
__global__ void __launch_bounds__(256) k_enc_hist(const Items it, uint32_t *__restrict__ hist) {
	__shared__ uint32_t s_cnt[ENC_NBIN];
	if (threadIdx.x < ENC_NBIN) s_cnt[threadIdx.x] = 0;
	__syncthreads();
#pragma unroll
	for (int i = 0; i < SORT_ITEMS; i++) {
		const int64_t q = (int64_t)blockIdx.x * SORT_TILE + i * 256 + threadIdx.x;
		if (q < it.count) { const int b = it.bin(q); if (b >= 0) atomicAdd(&s_cnt[b], 1u); }
	}
	__syncthreads();
	if (threadIdx.x < ENC_NBIN && s_cnt[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_cnt[threadIdx.x]);
}
// cursor[b] = first list slot of bin b, bins in descending order; cursor[ENC_NBIN] = number of listed items
// cursor[ENC_NBIN + 1] = items of the bins >= bigBin (the head of the list): those go to whole waves
// cursor[ENC_NBIN + 2] = of those, the items of the bins >= segBin (the very head): pairs that are cut into segments
__global__ void k_enc_bases(const uint32_t *__restrict__ hist, uint32_t *__restrict__ cursor, int bigBin, int segBin) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	uint32_t run = 0;
	cursor[ENC_NBIN + 2] = 0;
	for (int b = ENC_NBIN - 1; b >= 0; b--) { if (b == bigBin - 1) cursor[ENC_NBIN + 1] = run; if (b == segBin - 1) cursor[ENC_NBIN + 2] = run; cursor[b] = run; run += hist[b]; }
	cursor[ENC_NBIN] = run;
	if (bigBin <= 0) cursor[ENC_NBIN + 1] = run;
	if (segBin <= 0 || segBin < bigBin) cursor[ENC_NBIN + 2] = 0; // (only pairs the waves take can be cut)
}
template <class Items>
__global__ void __launch_bounds__(256) k_enc_scatter(const Items it, uint32_t *__restrict__ cursor, uint32_t *__restrict__ list, uint32_t *__restrict__ none) {
	__shared__ uint32_t s_cnt[ENC_NBIN], s_base[ENC_NBIN];
	if (threadIdx.x < ENC_NBIN) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	int bin[SORT_ITEMS];
	uint32_t rank[SORT_ITEMS];
#pragma unroll
	for (int i = 0; i < SORT_ITEMS; i++) {
		const int64_t q = (int64_t)blockIdx.x * SORT_TILE + i * 256 + threadIdx.x;
		bin[i] = q < it.count ? it.bin(q) : -2;
		rank[i] = bin[i] >= 0 ? atomicAdd(&s_cnt[bin[i]], 1u) : 0;
		if (bin[i] == -1 && none) none[q] = bve::COST_NONE; // not a candidate: its cost is known without a walk
	}
	__syncthreads();
	if (threadIdx.x < ENC_NBIN && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], s_cnt[threadIdx.x]);
	__syncthreads();
#pragma unroll
	for (int i = 0; i < SORT_ITEMS; i++)
		if (bin[i] >= 0) list[s_base[bin[i]] + rank[i]] = (uint32_t)((int64_t)blockIdx.x * SORT_TILE + i * 256 + threadIdx.x);
}

template <bool DEF>
__global__ void __launch_bounds__(256) k_enc_cost(const Params p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, const uint32_t *__restrict__ list,
                                                  const uint32_t *__restrict__ total, uint32_t *__restrict__ cost, int *__restrict__ err) {
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x + total[1]; // the lanes take what the waves leave
	if (t >= *total) return;
	const int64_t q = list[t];
	const int cyc = p.W + 1;
	const int32_t x = (int32_t)(q / cyc);
	const int r = (int)(q - (int64_t)x * cyc);
	int e = 0;
	cost[q] = bve::pair_cost<DEF>(p, rowptr, succ, x, r, &e);
	if (e) atomicOr(err, e);
}

// the pairs at the head of the list, one wave each (bv_encode_wave.hpp)
template <bool DEF>
__global__ void __launch_bounds__(256) k_enc_cost_wave(const Params p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, const uint32_t *__restrict__ list,
                                                       const uint32_t *__restrict__ total, uint32_t *__restrict__ cost, bvw::PairInfo *__restrict__ info, int *__restrict__ err) {
	const int64_t nbig = total[1], stride = (int64_t)gridDim.x * 4;
	for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6) + total[2]; t < nbig; t += stride) { // (the first total[2] pairs are cut into segments: k_seg_*)
		const int64_t q = list[t];
		const int cyc = p.W + 1;
		const int32_t x = (int32_t)(q / cyc);
		const int r = (int)(q - (int64_t)x * cyc);
		int e = 0;
		bvw::WaveTotals wt;
		const uint32_t c = bvw::wave_pair_cost<DEF>(p, rowptr, succ, x, r, &e, wt);
		if ((threadIdx.x & 63) == 0) {
			cost[q] = c;
			info[t] = bvw::PairInfo{ wt.nb, (uint32_t)wt.bitsB, wt.ni | (wt.nextra > 0 ? 0x80000000u : 0u), (uint32_t)wt.bitsI }; // (sections of 2^31 bits: c = COST_NONE, error raised)
			if (e) atomicOr(err, e);
		}
	}
}

// ---- the longest pairs, cut into segments.  A pair of 2^15 elements or more is some thousands of rounds of one wave (C2 has
// pairs of 4 * 10^5: 13 ms of pricing, 18 ms of emission, each the tail of its kernel).  Its lists are cut at values where no
// run of consecutive successors crosses (k_seg_plan): then a segment is an independent walk except for what it inherits -- the
// copy run in progress, the previous residual, the previous interval's end, and how much of each section precedes it.  Every
// segment is first walked in SEG mode (k_seg_count: everything priced but the three items that depend on the predecessors), a
// lane stitches the segments of a pair in order (k_seg_compose: the pair's price and section sizes, and every segment's
// inherited state and section offsets), and the emission walks the segments again with that state (k_seg_emit).
constexpr int SEG_ELEMS = 8192, SEG_BIN = 16, SEG_MAX = 64;
struct Seg { uint32_t pair; int32_t ja, jb, ka, kb; }; // segment = cur[ja, jb) x ref[ka, kb) of pair list[pair]
struct SegSum {
	int32_t firstFlag, lastFlag, nextra, bad;
	uint32_t nb, nr, ni, pad;
	int64_t firstChange, lastChange, firstRes, lastRes, firstLeft, lastEnd; // (changes: indices relative to ka)
	uint64_t bitsB, bitsI, bitsR, ivArcs;
};
struct SegIn { uint32_t prevFlag, nb, nr, ni; int64_t runStart, prevRes, prevEnd; uint64_t offB, offI, offR; }; // runStart relative to ka
struct SegPair { uint32_t segBase, nseg, nr; int32_t nextra; uint64_t bitsR, ivArcs; };

__global__ void __launch_bounds__(256) k_seg_plan(const Params p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, const uint32_t *__restrict__ list,
                                                  const uint32_t *__restrict__ total, int segElems, Seg *__restrict__ segs, uint32_t segCap, uint32_t *__restrict__ nsegs, SegPair *__restrict__ pairs, int *__restrict__ err) {
	const int lane = threadIdx.x & 63;
	const int64_t ng = total[2], stride = (int64_t)gridDim.x * 4;
	const int cyc = p.W + 1;
	for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < ng; t += stride) {
		const int64_t q = list[t];
		const int32_t x = (int32_t)(q / cyc);
		const int r = (int)(q - (int64_t)x * cyc);
		const int64_t a = rowptr[x], b = rowptr[x - r];
		const int32_t d = (int32_t)(rowptr[x + 1] - a), dr = r == 0 ? 0 : (int32_t)(rowptr[x - r + 1] - b);
		const int32_t *cur = succ + a, *ref = succ + b;
		const int64_t size = (int64_t)d + dr;
		const int nseg = (int)((size + segElems - 1) / segElems < SEG_MAX ? (size + segElems - 1) / segElems : SEG_MAX);
		// cut s (lane s, 0 < s < nseg): the first index at or after the nominal one where cur does not continue a run of consecutive ids
		int32_t j = lane == 0 ? 0 : d, k = lane == 0 ? 0 : dr;
		if (lane > 0 && lane < nseg) {
			j = (int32_t)((int64_t)d * lane / nseg);
			while (j > 0 && j < d && (int64_t)cur[j] == (int64_t)cur[j - 1] + 1) j++;
			if (j < d) { int32_t lo = 0, hi = dr; const int32_t v = cur[j]; while (lo < hi) { const int32_t mid = (int32_t)(((int64_t)lo + hi) >> 1); if (ref[mid] < v) lo = mid + 1; else hi = mid; } k = lo; }
		}
		const int32_t jn = __shfl_down(j, 1), kn = __shfl_down(k, 1); // lanes >= nseg hold (d, dr): lane nseg - 1 ends there
		uint32_t base = 0;
		if (lane == 0) base = atomicAdd(nsegs, (uint32_t)nseg);
		base = __shfl(base, 0);
		if ((uint64_t)base + nseg > segCap) { if (lane == 0) { atomicOr(err, 4); pairs[t] = SegPair{ 0, 0, 0, 0, 0, 0 }; } continue; }
		if (lane < nseg) segs[base + lane] = Seg{ (uint32_t)t, j, lane == nseg - 1 ? d : jn, k, lane == nseg - 1 ? dr : kn };
		if (lane == 0) pairs[t] = SegPair{ base, (uint32_t)nseg, 0, 0, 0, 0 };
	}
}

template <bool DEF>
__global__ void __launch_bounds__(256) k_seg_count(const Params p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, const uint32_t *__restrict__ list,
                                                   const Seg *__restrict__ segs, const uint32_t *__restrict__ nsegs, uint32_t segCap, SegSum *__restrict__ sums) {
	const int64_t n = *nsegs < segCap ? *nsegs : segCap, stride = (int64_t)gridDim.x * 4;
	const int cyc = p.W + 1;
	for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += stride) {
		const Seg sg = segs[i];
		const int64_t q = list[sg.pair];
		const int32_t x = (int32_t)(q / cyc);
		const int r = (int)(q - (int64_t)x * cyc);
		const int32_t *cur = succ + rowptr[x] + sg.ja, *ref = succ + rowptr[x - r] + sg.ka;
		bvw::WaveWalk<DEF, false, true> w(p, x, nullptr, 0, 0, 0);
		bvw::WaveTotals t;
		w.run(cur, sg.jb - sg.ja, ref, r == 0 ? 0 : sg.kb - sg.ka, t);
		if ((threadIdx.x & 63) == 0)
			sums[i] = SegSum{ w.firstFlag, (int32_t)w.prevFlag, t.nextra, t.bad, t.nb, t.nr, t.ni, 0, w.firstChange, w.runStart, w.firstRes, w.prevRes, w.firstLeft, w.prevEnd,
			                  t.bitsB, t.bitsI, t.bitsR, t.ivArcs };
	}
}

// one lane per cut pair: its segments in order
template <bool DEF>
__global__ void __launch_bounds__(64) k_seg_compose(const Params p, const uint32_t *__restrict__ list, const uint32_t *__restrict__ total, const Seg *__restrict__ segs,
                                                    const SegSum *__restrict__ sums, SegIn *__restrict__ ins, SegPair *__restrict__ pairs, uint32_t *__restrict__ cost,
                                                    bvw::PairInfo *__restrict__ info, int *__restrict__ err) {
	const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
	if (t >= total[2]) return;
	const int cyc = p.W + 1;
	const int64_t q = list[t];
	const int32_t x = (int32_t)(q / cyc);
	const int r = (int)(q - (int64_t)x * cyc);
	SegPair sp = pairs[t];
	uint32_t prevFlag = 1, nb = 0, nr = 0, ni = 0;
	int64_t runStart = 0, prevRes = 0, prevEnd = 0, nextra = 0;
	uint64_t bitsB = 0, bitsI = 0, bitsR = 0, ivArcs = 0;
	int bad = sp.nseg == 0 ? 4 : 0;
	auto lenBlk = [&](int64_t len) { bve::LenSink s; bve::f_blk<DEF>(s, p, (uint64_t)(nb == 0 ? len : len - 1)); return s.bits; };
	for (uint32_t i = 0; i < sp.nseg; i++) {
		const Seg sg = segs[sp.segBase + i];
		const SegSum sm = sums[sp.segBase + i];
		ins[sp.segBase + i] = SegIn{ prevFlag, nb, nr, ni, runStart - sg.ka, prevRes, prevEnd, bitsB, bitsI, bitsR };
		if (r != 0 && sg.kb > sg.ka) {
			if ((uint32_t)sm.firstFlag != prevFlag) { bitsB += lenBlk((int64_t)sg.ka - runStart); nb++; runStart = sg.ka; } // the flag changes where the segment begins
			if (sm.nb > 0) {
				bitsB += lenBlk((int64_t)sg.ka + sm.firstChange - runStart); nb++;
				nb += sm.nb - 1; bitsB += sm.bitsB;
				runStart = (int64_t)sg.ka + sm.lastChange;
			}
			prevFlag = (uint32_t)sm.lastFlag;
		}
		if (sm.ni > 0) {
			bve::LenSink s;
			bve::w_gamma(s, ni == 0 ? bve::int2nat(sm.firstLeft - x) : (uint64_t)(sm.firstLeft - prevEnd - 1));
			bitsI += s.bits + sm.bitsI; ni += sm.ni; prevEnd = sm.lastEnd;
		}
		if (sm.nr > 0) {
			bve::LenSink s;
			bve::f_res<DEF>(s, p, nr == 0 ? bve::int2nat(sm.firstRes - x) : (uint64_t)(sm.firstRes - prevRes - 1));
			bitsR += s.bits + sm.bitsR; nr += sm.nr; prevRes = sm.lastRes;
		}
		nextra += sm.nextra; ivArcs += sm.ivArcs; bad |= sm.bad;
	}
	bve::LenSink s;
	if (p.W > 0) bve::f_ref<DEF>(s, p, (uint64_t)r);
	if (r != 0) bve::f_bc<DEF>(s, p, nb);
	if (nextra > 0 && p.I != 0) bve::w_gamma(s, ni);
	const uint64_t tot = s.bits + (r != 0 ? bitsB : 0) + bitsI + bitsR;
	if (bad & 1) atomicOr(err, 1);
	if (bad & 4) atomicOr(err, 4);
	if (tot > bve::COST_MAX) { atomicOr(err, 2); cost[q] = bve::COST_NONE; }
	else cost[q] = (uint32_t)tot;
	info[t] = bvw::PairInfo{ nb, (uint32_t)bitsB, ni | (nextra > 0 ? 0x80000000u : 0u), (uint32_t)bitsI };
	sp.nr = nr; sp.nextra = (int32_t)nextra; sp.bitsR = bitsR; sp.ivArcs = ivArcs;
	pairs[t] = sp;
}

// emission of the cut pairs that were chosen: one wave per segment, with the state the stitching handed it
template <bool DEF>
__global__ void __launch_bounds__(256) k_seg_emit(const Params p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, const uint8_t *__restrict__ best,
                                                  const int32_t *__restrict__ refc, const int64_t *__restrict__ off, const uint32_t *__restrict__ list, const Seg *__restrict__ segs,
                                                  const uint32_t *__restrict__ nsegs, uint32_t segCap, const SegIn *__restrict__ ins, const SegPair *__restrict__ pairs,
                                                  const bvw::PairInfo *__restrict__ info, uint32_t *__restrict__ words, EncStatsDev *__restrict__ stats) {
	const int lane = threadIdx.x & 63;
	const int64_t n = *nsegs < segCap ? *nsegs : segCap, stride = (int64_t)gridDim.x * 4;
	const int cyc = p.W + 1;
	__shared__ unsigned long long s_rb[32];
	if (threadIdx.x < 32) s_rb[threadIdx.x] = 0;
	__syncthreads();
	for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += stride) {
		const Seg sg = segs[i];
		const int64_t q = list[sg.pair];
		const int32_t x = (int32_t)(q / cyc);
		const int r = (int)(q - (int64_t)x * cyc);
		if (best[x] != r) continue;
		const SegPair sp = pairs[sg.pair];
		const bvw::PairInfo pi = info[sg.pair];
		const SegIn in = ins[i];
		const int64_t a = rowptr[x], b = rowptr[x - r];
		const int32_t d = (int32_t)(rowptr[x + 1] - a);
		const uint32_t ni = pi.ni & 0x7fffffffu;
		const bool extras = (pi.ni >> 31) != 0;
		const uint64_t pos = (uint64_t)off[x];
		bve::LenSink h;
		bve::f_outd<DEF>(h, p, (uint64_t)d);
		const uint64_t afterOutd = pos + h.bits;
		if (p.W > 0) bve::f_ref<DEF>(h, p, (uint64_t)r);
		const uint64_t afterRef = pos + h.bits;
		if (r != 0) bve::f_bc<DEF>(h, p, pi.nb);
		const uint64_t posB = pos + h.bits, startI = posB + (r != 0 ? pi.bitsB : 0);
		bve::LenSink ic;
		if (extras && p.I != 0) bve::w_gamma(ic, ni);
		const uint64_t posI = startI + ic.bits, posR = posI + pi.bitsI;
		if ((uint32_t)i == sp.segBase && lane == 0) { // the record's fixed fields and its contribution to the counters, once
			bve::WordSink hw(words, pos);
			bve::f_outd<DEF>(hw, p, (uint64_t)d);
			if (p.W > 0) bve::f_ref<DEF>(hw, p, (uint64_t)r);
			if (r != 0) bve::f_bc<DEF>(hw, p, pi.nb);
			hw.finish();
			if (extras && p.I != 0) { bve::WordSink wi(words, startI); bve::w_gamma(wi, ni); wi.finish(); }
			const unsigned long long v[10] = { afterOutd - pos, afterRef - afterOutd, startI - afterRef, posR - startI, sp.bitsR, (unsigned long long)(d - sp.nextra), sp.ivArcs, sp.nr,
			                                   (unsigned long long)refc[x], (unsigned long long)r };
			for (int k = 0; k < 10; k++) if (v[k]) atomicAdd(&stats->v[k], v[k]);
			atomicMax(&stats->v[10], (unsigned long long)refc[x]);
		}
		bvw::WaveWalk<DEF, true> w(p, x, words, posB + in.offB, posI + in.offI, posR + in.offR);
		w.prevFlag = in.prevFlag; w.runStart = in.runStart; w.nb = in.nb; w.nr = in.nr; w.ni = in.ni; w.prevRes = in.prevRes; w.prevEnd = in.prevEnd;
		w.rbins = s_rb;
		bvw::WaveTotals t;
		w.run(succ + a + sg.ja, sg.jb - sg.ja, succ + b + sg.ka, r == 0 ? 0 : sg.kb - sg.ka, t);
	}
	flush_bins(s_rb, stats->resBins);
}

// self-check (BVGPU_ENC_VERIFY, used by the tests): the lane walk over the pairs the waves took; mismatches -> dbg[0] = count, then (q, wave, lane) triples
template <bool DEF>
__global__ void __launch_bounds__(256) k_enc_verify(const Params p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, const uint32_t *__restrict__ list,
                                                    const uint32_t *__restrict__ total, const uint32_t *__restrict__ cost, unsigned long long *__restrict__ dbg) {
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (t >= total[1]) return;
	const int64_t q = list[t];
	const int cyc = p.W + 1;
	const int32_t x = (int32_t)(q / cyc);
	int e = 0;
	const uint32_t c = bve::pair_cost<DEF>(p, rowptr, succ, x, (int)(q - (int64_t)x * cyc), &e);
	if (c != cost[q]) { const unsigned long long k = atomicAdd(dbg, 1ull); if (k < 16) { dbg[1 + 3 * k] = (unsigned long long)q; dbg[2 + 3 * k] = cost[q]; dbg[3 + 3 * k] = c; } }
}

// one round of bve::select_span: lane l walks the chunks [l * span, (l + 1) * span)
__global__ void __launch_bounds__(64) k_enc_select(const Params p, const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ cost, int32_t n, int64_t nchunks, int span, int round,
                                                   const int32_t *__restrict__ statePrev, int32_t *__restrict__ stateNew, int32_t *__restrict__ used,
                                                   uint8_t *__restrict__ best, int32_t *__restrict__ refc, int *__restrict__ moved) {
	const int64_t c0 = ((int64_t)blockIdx.x * 64 + threadIdx.x) * span;
	if (c0 >= nchunks) return;
	int32_t in[ENC_MAX_W + 1];
	const int64_t c1 = c0 + span < nchunks ? c0 + span : nchunks;
	if (bve::select_span(p, rowptr, cost, n, SEL_CHUNK, c0, c1, round, statePrev, stateNew, used, best, refc, in)) *moved = 1;
}

__global__ void __launch_bounds__(256) k_enc_reclen(const Params p, const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ cost, const uint8_t *__restrict__ best, int32_t n, int32_t *__restrict__ reclen, int *__restrict__ err) {
	const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (x >= n) return;
	const int64_t d = rowptr[x + 1] - rowptr[x];
	bve::LenSink s;
	bve::w_code(s, p.c_outd, (uint64_t)d, 0);
	uint64_t t = s.bits;
	if (d > 0) t += cost[x * (p.W + 1) + best[x]];
	if (t > bve::COST_MAX) { atomicOr(err, 2); t = 0; }
	reclen[x] = (int32_t)t;
}


__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

template <bool DEF>
__global__ void __launch_bounds__(256) k_enc_emit(const Params p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, const uint8_t *__restrict__ best, const int32_t *__restrict__ refc,
                                                  const int64_t *__restrict__ off, const uint32_t *__restrict__ list, const uint32_t *__restrict__ total, int32_t n, uint32_t *__restrict__ words, EncStatsDev *__restrict__ stats) {
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	bve::NodeStats st;
	unsigned long long totRef = 0, totDist = 0, chain = 0;
	__shared__ unsigned long long s_rb[32];
	if (threadIdx.x < 32) s_rb[threadIdx.x] = 0;
	__syncthreads();
	if (t < total[0]) {
		const int64_t x = list[t];
		const int r = best[x];
		(void)bve::emit_node<DEF>(p, rowptr, succ, (int32_t)x, r, words, (uint64_t)off[x], &st, s_rb);
		if (rowptr[x + 1] > rowptr[x]) { totRef = (unsigned long long)refc[x]; totDist = (unsigned long long)r; chain = totRef; }
	}
	// (the counters of a block joined in LDS first: 156 000 waves x 10 additions to the same ten words were ~2 ms of C2's emission, same-address atomics run at ~88 M/s)
	__shared__ unsigned long long s_acc[11];
	if (threadIdx.x < 11) s_acc[threadIdx.x] = 0;
	__syncthreads();
	const unsigned long long vals[10] = { st.bitsOutd, st.bitsRef, st.bitsBlocks, st.bitsIntervals, st.bitsResiduals, st.copied, st.intervalised, st.residuals, totRef, totDist };
#pragma unroll
	for (int i = 0; i < 10; i++) {
		const unsigned long long s = wave_sum(vals[i]);
		if ((threadIdx.x & 63) == 0 && s) atomicAdd(&s_acc[i], s);
	}
	for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(chain, o); chain = t > chain ? t : chain; }
	if ((threadIdx.x & 63) == 0 && chain) atomicMax(&s_acc[10], chain);
	__syncthreads();
	if (threadIdx.x < 10 && s_acc[threadIdx.x]) atomicAdd(&stats->v[threadIdx.x], s_acc[threadIdx.x]);
	if (threadIdx.x == 10 && s_acc[10]) atomicMax(&stats->v[10], s_acc[10]);
	flush_bins(s_rb, stats->resBins);
}

// the nodes whose chosen pair is at the head of the list of pairs: one wave each, with the sizes the pricing left
template <bool DEF>
__global__ void __launch_bounds__(256) k_enc_emit_wave(const Params p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, const uint8_t *__restrict__ best, const int32_t *__restrict__ refc,
                                                       const int64_t *__restrict__ off, const uint32_t *__restrict__ pairList, const uint32_t *__restrict__ pairTotal,
                                                       const bvw::PairInfo *__restrict__ info, uint32_t *__restrict__ words, EncStatsDev *__restrict__ stats) {
	const int64_t nbig = pairTotal[1], stride = (int64_t)gridDim.x * 4;
	const int cyc = p.W + 1;
	unsigned long long acc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, chain = 0; // lane 0 of the wave
	__shared__ unsigned long long s_rb[32];
	if (threadIdx.x < 32) s_rb[threadIdx.x] = 0;
	__syncthreads();
	for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6) + pairTotal[2]; t < nbig; t += stride) { // (the cut pairs: k_seg_emit)
		const int64_t q = pairList[t];
		const int32_t x = (int32_t)(q / cyc);
		const int r = (int)(q - (int64_t)x * cyc);
		if (best[x] != r) continue;
		bve::NodeStats st;
		bvw::wave_emit_node<DEF>(p, rowptr, succ, x, r, info[t], words, (uint64_t)off[x], st, s_rb);
		acc[0] += st.bitsOutd; acc[1] += st.bitsRef; acc[2] += st.bitsBlocks; acc[3] += st.bitsIntervals; acc[4] += st.bitsResiduals;
		acc[5] += st.copied; acc[6] += st.intervalised; acc[7] += st.residuals; acc[8] += (unsigned long long)refc[x]; acc[9] += (unsigned long long)r;
		if ((unsigned long long)refc[x] > chain) chain = (unsigned long long)refc[x];
	}
	if ((threadIdx.x & 63) == 0) {
		for (int i = 0; i < 10; i++) if (acc[i]) atomicAdd(&stats->v[i], acc[i]);
		if (chain) atomicMax(&stats->v[10], chain);
	}
	flush_bins(s_rb, stats->resBins);
}

// the .offsets stream: code 0 is the offset of node 0, code i the length of record i - 1 (BVGraph.java:2285, :2369)
__global__ void __launch_bounds__(256) k_enc_offlen(const Params p, const int32_t *__restrict__ reclen, int32_t n, int32_t *__restrict__ len) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i > n) return;
	bve::LenSink s;
	bve::w_code(s, p.c_off, i == 0 ? 0 : (uint64_t)reclen[i - 1], 0);
	len[i] = (int32_t)s.bits;
}
__global__ void __launch_bounds__(256) k_enc_offemit(const Params p, const int32_t *__restrict__ reclen, const int64_t *__restrict__ at, int32_t n, uint32_t *__restrict__ words) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i > n) return;
	bve::WordSink s(words, (uint64_t)at[i]);
	bve::w_code(s, p.c_off, i == 0 ? 0 : (uint64_t)reclen[i - 1], 0);
	s.finish();
}

// ---------------------------------------------------------------------------------------------------------------- host
// the .offsets stream of n record lengths (code 0: the offset of node 0), gamma or delta coded; *d_words_out is hipMalloc'ed
int offsets_stream_device(int coding, const int32_t *d_reclen, int32_t n, uint32_t **d_words_out, uint64_t *bits_out, hipStream_t st) {
	*d_words_out = nullptr; *bits_out = 0;
	Params p{};
	p.c_off = coding;
	int32_t *offlen = nullptr;
	int64_t *offat = nullptr, *sums = nullptr;
	const size_t nn = (size_t)n + 1;
	auto done = [&](int rc) { for (void *q : { (void *)offlen, (void *)offat, (void *)sums }) if (q) (void)hipFree(q); if (rc && *d_words_out) { (void)hipFree(*d_words_out); *d_words_out = nullptr; } return rc; };
	if (hipMalloc((void **)&offlen, sizeof(int32_t) * nn) != hipSuccess || hipMalloc((void **)&offat, sizeof(int64_t) * (nn + 1)) != hipSuccess ||
	    hipMalloc((void **)&sums, sizeof(int64_t) * (size_t)(scan_num_sums((int64_t)nn) + 1)) != hipSuccess) return done(-5);
	const dim3 grid((unsigned)((nn + 255) / 256));
	hipLaunchKernelGGL(k_enc_offlen, grid, dim3(256), 0, st, p, d_reclen, n, offlen);
	launch_scan(offlen, (int64_t)nn, offat, sums, st);
	int64_t offBits = 0;
	if (hipMemcpyAsync(&offBits, offat + nn, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return done(-6);
	const size_t ow = (size_t)((offBits + 31) / 32) + 8;
	if (hipMalloc((void **)d_words_out, ow * 4) != hipSuccess) return done(-5);
	(void)hipMemsetAsync(*d_words_out, 0, ow * 4, st);
	hipLaunchKernelGGL(k_enc_offemit, grid, dim3(256), 0, st, p, d_reclen, offat, n, *d_words_out);
	if (hipStreamSynchronize(st) != hipSuccess) return done(-6);
	*bits_out = (uint64_t)offBits;
	return done(0);
}

void encode_free(EncodeOut &o) {
	for (void *q : { (void *)o.graph_words, (void *)o.off_words, (void *)o.offsets }) if (q) (void)hipFree(q);
	o.graph_words = nullptr; o.off_words = nullptr; o.offsets = nullptr;
}

int encode_device(const Params &p, int32_t n, const int64_t *d_rowptr, const int32_t *d_succ, uint64_t m, EncodeOut &out, std::string &err, hipStream_t st) {
	out = EncodeOut{};
	if (p.W < 0 || p.W > ENC_MAX_W) { err = "windowsize above 63 is not supported by the device compressor"; return -3; }
	const bool trace = bv_env("BVGPU_ENC_TRACE") != nullptr;
	const int cyc = p.W + 1;
	const int64_t npairs = (int64_t)n * cyc;
	const int64_t nchunks = ((int64_t)n + SEL_CHUNK - 1) / SEL_CHUNK;
	const int64_t ns = scan_num_sums((int64_t)n + 1);
	uint32_t *cost = nullptr;
	uint8_t *best = nullptr;
	int32_t *refc = nullptr, *reclen = nullptr, *offlen = nullptr, *state = nullptr, *used = nullptr;
	int64_t *sums = nullptr, *offat = nullptr;
	uint32_t *list = nullptr, *nlist = nullptr, *bins = nullptr, *nbins = nullptr; // lists of pairs / of nodes; bins: [0, 32) histogram, [32, 64) cursors, [64] listed items, [65] of them for the waves
	bvw::PairInfo *info = nullptr;
	Seg *segs = nullptr; SegSum *ssums = nullptr; SegIn *sins = nullptr; SegPair *spairs = nullptr; uint32_t *nsegs = nullptr; // the cut pairs
	uint32_t segCap = 0, ngiant = 0;
	int *flags = nullptr, *moved = nullptr; // flags[0]: error bits; moved[i]: did round i of the batch change a chunk's final state
	EncStatsDev *dstats = nullptr;
	std::vector<hipEvent_t> ev;
	auto mark = [&]() { if (trace) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, st); ev.push_back(e); } };
	auto cleanup = [&](int rc) {
		for (void *q : { (void *)cost, (void *)best, (void *)refc, (void *)reclen, (void *)offlen, (void *)state, (void *)used, (void *)sums, (void *)offat, (void *)list, (void *)nlist, (void *)bins, (void *)nbins, (void *)info, (void *)segs, (void *)ssums, (void *)sins, (void *)spairs, (void *)nsegs, (void *)flags, (void *)moved, (void *)dstats })
			if (q) (void)hipFree(q);
		for (auto e : ev) (void)hipEventDestroy(e);
		if (rc) { encode_free(out); (void)hipGetLastError(); }
		return rc;
	};
	auto alloc = [&](void **q, size_t bytes) { return hipMalloc(q, bytes ? bytes : 16) == hipSuccess; };
	const size_t nn = (size_t)n + 1;
	if (npairs >= 0xffffffffll) { err = "too many (node, candidate) pairs for one call"; return cleanup(-3); }
	const bool def = bve::default_codings(p);
	const int bigBin = bv_env("BVGPU_ENC_BIGBIN") ? atoi(bv_env("BVGPU_ENC_BIGBIN")) : BIG_BIN; // experiment: 32 = everything lane by lane
	const int segElems = bv_env("BVGPU_ENC_SEGELEMS") ? std::max(1, atoi(bv_env("BVGPU_ENC_SEGELEMS"))) : SEG_ELEMS; // tests: short segments
	const int segBin = bv_env("BVGPU_ENC_SEGBIN") ? atoi(bv_env("BVGPU_ENC_SEGBIN")) : SEG_BIN; // experiment / tests: 32 = no pair is cut, 8 = every pair the waves take
	if (!alloc((void **)&cost, sizeof(uint32_t) * (size_t)npairs) || !alloc((void **)&best, nn) || !alloc((void **)&refc, sizeof(int32_t) * nn) ||
	    !alloc((void **)&reclen, sizeof(int32_t) * nn) || !alloc((void **)&offlen, sizeof(int32_t) * nn) || !alloc((void **)&state, sizeof(int32_t) * 2 * (size_t)nchunks * (size_t)(p.W ? p.W : 1)) ||
	    !alloc((void **)&used, sizeof(int32_t) * (size_t)nchunks * (size_t)(p.W ? p.W : 1)) || !alloc((void **)&sums, sizeof(int64_t) * (size_t)(ns + 1)) ||
	    !alloc((void **)&offat, sizeof(int64_t) * (nn + 1)) || !alloc((void **)&list, sizeof(uint32_t) * (size_t)npairs) || !alloc((void **)&nlist, sizeof(uint32_t) * nn) || !alloc((void **)&bins, sizeof(uint32_t) * (2 * ENC_NBIN + 3)) || !alloc((void **)&nbins, sizeof(uint32_t) * (2 * ENC_NBIN + 3)) || !alloc((void **)&flags, 2 * sizeof(int)) || !alloc((void **)&moved, SEL_BATCH * sizeof(int)) ||
	    !alloc((void **)&dstats, sizeof(EncStatsDev)) || !alloc((void **)&out.offsets, sizeof(int64_t) * nn)) { err = "device allocation failed"; return cleanup(-5); }
	(void)hipMemsetAsync(flags, 0, 2 * sizeof(int), st);
	(void)hipMemsetAsync(dstats, 0, sizeof(EncStatsDev), st);
	mark();
	auto blocks = [](int64_t items, int per) { return dim3((unsigned)((items + per - 1) / per > 0 ? (items + per - 1) / per : 1)); };
	if (n) hipLaunchKernelGGL(k_enc_check_rowptr, blocks(n, 256), dim3(256), 0, st, d_rowptr, n, flags);
	{
		int h0 = 0;
		if (hipMemcpyAsync(&h0, flags, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "the compressor kernels failed"; return cleanup(-6); }
		if (h0 & 8) { err = "rowptr must start at 0 and be monotone"; return cleanup(-1); }
	}
	// A
	if (npairs) {
		const PairItems items{ p, d_rowptr, npairs };
		(void)hipMemsetAsync(bins, 0, sizeof(uint32_t) * (2 * ENC_NBIN + 3), st);
		hipLaunchKernelGGL(k_enc_hist<PairItems>, blocks(npairs, SORT_TILE), dim3(256), 0, st, items, bins);
		hipLaunchKernelGGL(k_enc_bases, dim3(1), dim3(64), 0, st, bins, bins + ENC_NBIN, bigBin, segBin);
		uint32_t hb[2 * ENC_NBIN + 3]; // the pricing of the long pairs leaves their section sizes for the emission: one entry per such pair
		if (hipMemcpyAsync(hb, bins, sizeof hb, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "the list kernels failed"; return cleanup(-6); }
		const uint32_t nbig = hb[2 * ENC_NBIN + 1];
		ngiant = hb[2 * ENC_NBIN + 2];
		if (!alloc((void **)&info, sizeof(bvw::PairInfo) * (size_t)nbig)) { err = "device allocation failed"; return cleanup(-5); }
		hipLaunchKernelGGL(k_enc_scatter<PairItems>, blocks(npairs, SORT_TILE), dim3(256), 0, st, items, bins + ENC_NBIN, list, cost);
		if (ngiant) { // segments of the cut pairs: at most SEG_MAX each, and a pair of bin b has fewer than 2^b elements
			uint64_t cap = 0;
			for (int b = segBin; b < ENC_NBIN; b++) cap += (uint64_t)hb[b] * std::min<uint64_t>(SEG_MAX, ((1ull << b) + segElems - 1) / segElems);
			segCap = (uint32_t)std::min<uint64_t>(cap, 0x7fffffffu);
			if (!alloc((void **)&segs, sizeof(Seg) * (size_t)segCap) || !alloc((void **)&ssums, sizeof(SegSum) * (size_t)segCap) || !alloc((void **)&sins, sizeof(SegIn) * (size_t)segCap) ||
			    !alloc((void **)&spairs, sizeof(SegPair) * (size_t)ngiant) || !alloc((void **)&nsegs, sizeof(uint32_t))) { err = "device allocation failed"; return cleanup(-5); }
			(void)hipMemsetAsync(nsegs, 0, sizeof(uint32_t), st);
			hipLaunchKernelGGL(k_seg_plan, dim3(256), dim3(256), 0, st, p, d_rowptr, d_succ, list, bins + 2 * ENC_NBIN, segElems, segs, segCap, nsegs, spairs, flags);
			if (def) {
				hipLaunchKernelGGL(k_seg_count<true>, dim3(WAVE_BLOCKS), dim3(256), 0, st, p, d_rowptr, d_succ, list, segs, nsegs, segCap, ssums);
				hipLaunchKernelGGL(k_seg_compose<true>, blocks(ngiant, 64), dim3(64), 0, st, p, list, bins + 2 * ENC_NBIN, segs, ssums, sins, spairs, cost, info, flags);
			} else {
				hipLaunchKernelGGL(k_seg_count<false>, dim3(WAVE_BLOCKS), dim3(256), 0, st, p, d_rowptr, d_succ, list, segs, nsegs, segCap, ssums);
				hipLaunchKernelGGL(k_seg_compose<false>, blocks(ngiant, 64), dim3(64), 0, st, p, list, bins + 2 * ENC_NBIN, segs, ssums, sins, spairs, cost, info, flags);
			}
		}
		if (def) {
			hipLaunchKernelGGL(k_enc_cost_wave<true>, dim3(WAVE_BLOCKS), dim3(256), 0, st, p, d_rowptr, d_succ, list, bins + 2 * ENC_NBIN, cost, info, flags);
			hipLaunchKernelGGL(k_enc_cost<true>, blocks(npairs, 256), dim3(256), 0, st, p, d_rowptr, d_succ, list, bins + 2 * ENC_NBIN, cost, flags);
		} else {
			hipLaunchKernelGGL(k_enc_cost_wave<false>, dim3(WAVE_BLOCKS), dim3(256), 0, st, p, d_rowptr, d_succ, list, bins + 2 * ENC_NBIN, cost, info, flags);
			hipLaunchKernelGGL(k_enc_cost<false>, blocks(npairs, 256), dim3(256), 0, st, p, d_rowptr, d_succ, list, bins + 2 * ENC_NBIN, cost, flags);
		}
	}
	bool verifyBad = false; // BVGPU_ENC_VERIFY (tests): the pairs the waves priced, priced again lane by lane
	if (npairs && bv_env("BVGPU_ENC_VERIFY")) {
		unsigned long long *dbg = nullptr, h[49] = { 0 };
		if (hipMalloc((void **)&dbg, sizeof h) == hipSuccess) {
			(void)hipMemsetAsync(dbg, 0, sizeof h, st);
			if (def) hipLaunchKernelGGL(k_enc_verify<true>, blocks(npairs, 256), dim3(256), 0, st, p, d_rowptr, d_succ, list, bins + 2 * ENC_NBIN, cost, dbg);
			else hipLaunchKernelGGL(k_enc_verify<false>, blocks(npairs, 256), dim3(256), 0, st, p, d_rowptr, d_succ, list, bins + 2 * ENC_NBIN, cost, dbg);
			(void)hipMemcpyAsync(h, dbg, sizeof h, hipMemcpyDeviceToHost, st);
			(void)hipStreamSynchronize(st);
			(void)hipFree(dbg);
			if (h[0]) fprintf(stderr, "[bvgpu enc] verify: %llu pairs priced differently by the waves\n", h[0]);
			verifyBad = h[0] != 0;
			std::vector<int64_t> rp((size_t)n + 1);
			(void)hipMemcpy(rp.data(), d_rowptr, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost);
			for (unsigned long long k = 0; k < h[0] && k < 16; k++) {
				const int64_t q = (int64_t)h[1 + 3 * k];
				const int32_t x = (int32_t)(q / cyc); const int r = (int)(q % cyc);
				fprintf(stderr, "[bvgpu enc]   node %d ref %d: d %lld dr %lld wave %llu lane %llu\n", x, r, (long long)(rp[(size_t)x + 1] - rp[(size_t)x]),
				        (long long)(rp[(size_t)(x - r) + 1] - rp[(size_t)(x - r)]), h[2 + 3 * k], h[3 + 3 * k]);
			}
		}
	}
	if (verifyBad) { err = "the wave walk and the lane walk price some pair differently"; return cleanup(-6); }
	if (trace && npairs) {
		uint32_t h[ENC_NBIN];
		if (hipMemcpyAsync(h, bins, sizeof h, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
			fprintf(stderr, "[bvgpu enc] pairs by size bin (bin b: < 2^b elements):");
			for (int b = 0; b < ENC_NBIN; b++) if (h[b]) fprintf(stderr, " %d:%u", b, h[b]);
			fprintf(stderr, "\n");
		}
	}
	mark();
	// B: round 0 one chunk per lane, then SEL_SPAN chunks per lane; SEL_BATCH rounds are enqueued between two looks at the flags
	int rounds = 0;
	const size_t stateHalf = (size_t)nchunks * (size_t)(p.W ? p.W : 1);
	int spanNow = SEL_SPAN; // doubled after every batch that did not settle: a stretch of any length is crossed in O(log) batches
	for (bool settled = nchunks == 0; !settled;) {
		(void)hipMemsetAsync(moved, 0, SEL_BATCH * sizeof(int), st);
		const int first = rounds;
		for (int i = 0; i < SEL_BATCH; i++, rounds++) {
			const int span = rounds == 0 ? 1 : spanNow;
			const int64_t lanes = (nchunks + span - 1) / span;
			int32_t *sPrev = state + (size_t)((rounds + 1) & 1) * stateHalf, *sNew = state + (size_t)(rounds & 1) * stateHalf;
			hipLaunchKernelGGL(k_enc_select, blocks(lanes, 64), dim3(64), 0, st, p, d_rowptr, cost, n, nchunks, span, rounds, sPrev, sNew, used, best, refc, moved + i);
			if (p.W == 0 || p.R == 0) { rounds++; break; } // no references at all: nothing to settle
		}
		int h[SEL_BATCH];
		if (hipMemcpyAsync(h, moved, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "the selection kernel failed"; return cleanup(-6); }
		if (p.W == 0 || p.R == 0) break;
		for (int i = 0; i < rounds - first; i++) if (first + i > 0 && !h[i]) { settled = true; rounds = first + i + 1; break; }
		if (!settled && (int64_t)rounds > nchunks + 2 * SEL_BATCH) { err = "the selection did not settle"; return cleanup(-6); }
		if (!settled && (int64_t)spanNow < nchunks) spanNow *= 2;
	}
	mark();
	// C
	int herr = 0;
	if (n) hipLaunchKernelGGL(k_enc_reclen, blocks(n, 256), dim3(256), 0, st, p, d_rowptr, cost, best, n, reclen, flags);
	launch_scan(reclen, n, out.offsets, sums, st);
	int64_t totalBits = 0;
	if (hipMemcpyAsync(&totalBits, out.offsets + n, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&herr, flags, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "the compressor kernels failed"; return cleanup(-6); }
	if (herr & 1) { err = "successor lists must be strictly increasing, with ids in [0, 2^31 - 1)"; return cleanup(-1); }
	if (herr & 4) { err = "the segment table of the cut pairs overflowed"; return cleanup(-6); }
	if (herr) { err = "a record of 2^31 bits or more"; return cleanup(-3); }
	mark();
	// D
	out.graph_bits = (uint64_t)totalBits;
	const size_t gw = (size_t)((totalBits + 31) / 32) + 8;
	if (!alloc((void **)&out.graph_words, gw * 4)) { err = "device allocation failed"; return cleanup(-5); }
	(void)hipMemsetAsync(out.graph_words, 0, gw * 4, st);
	if (n) {
		const NodeItems items{ p, d_rowptr, best, n, bigBin };
		(void)hipMemsetAsync(nbins, 0, sizeof(uint32_t) * (2 * ENC_NBIN + 3), st);
		hipLaunchKernelGGL(k_enc_hist<NodeItems>, blocks(n, SORT_TILE), dim3(256), 0, st, items, nbins);
		hipLaunchKernelGGL(k_enc_bases, dim3(1), dim3(64), 0, st, nbins, nbins + ENC_NBIN, ENC_NBIN, 0);
		hipLaunchKernelGGL(k_enc_scatter<NodeItems>, blocks(n, SORT_TILE), dim3(256), 0, st, items, nbins + ENC_NBIN, nlist, (uint32_t *)nullptr);
		if (ngiant) {
			if (def) hipLaunchKernelGGL(k_seg_emit<true>, dim3(WAVE_BLOCKS), dim3(256), 0, st, p, d_rowptr, d_succ, best, refc, out.offsets, list, segs, nsegs, segCap, sins, spairs, info, out.graph_words, dstats);
			else hipLaunchKernelGGL(k_seg_emit<false>, dim3(WAVE_BLOCKS), dim3(256), 0, st, p, d_rowptr, d_succ, best, refc, out.offsets, list, segs, nsegs, segCap, sins, spairs, info, out.graph_words, dstats);
		}
		if (def) {
			if (npairs) hipLaunchKernelGGL(k_enc_emit_wave<true>, dim3(WAVE_BLOCKS), dim3(256), 0, st, p, d_rowptr, d_succ, best, refc, out.offsets, list, bins + 2 * ENC_NBIN, info, out.graph_words, dstats);
			hipLaunchKernelGGL(k_enc_emit<true>, blocks(n, 256), dim3(256), 0, st, p, d_rowptr, d_succ, best, refc, out.offsets, nlist, nbins + 2 * ENC_NBIN, n, out.graph_words, dstats);
		} else {
			if (npairs) hipLaunchKernelGGL(k_enc_emit_wave<false>, dim3(WAVE_BLOCKS), dim3(256), 0, st, p, d_rowptr, d_succ, best, refc, out.offsets, list, bins + 2 * ENC_NBIN, info, out.graph_words, dstats);
			hipLaunchKernelGGL(k_enc_emit<false>, blocks(n, 256), dim3(256), 0, st, p, d_rowptr, d_succ, best, refc, out.offsets, nlist, nbins + 2 * ENC_NBIN, n, out.graph_words, dstats);
		}
	}
	mark();
	// E
	hipLaunchKernelGGL(k_enc_offlen, blocks((int64_t)n + 1, 256), dim3(256), 0, st, p, reclen, n, offlen);
	launch_scan(offlen, (int64_t)n + 1, offat, sums, st);
	int64_t offBits = 0;
	if (hipMemcpyAsync(&offBits, offat + n + 1, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "the offsets kernels failed"; return cleanup(-6); }
	out.off_bits = (uint64_t)offBits;
	const size_t ow = (size_t)((offBits + 31) / 32) + 8;
	if (!alloc((void **)&out.off_words, ow * 4)) { err = "device allocation failed"; return cleanup(-5); }
	(void)hipMemsetAsync(out.off_words, 0, ow * 4, st);
	hipLaunchKernelGGL(k_enc_offemit, blocks((int64_t)n + 1, 256), dim3(256), 0, st, p, reclen, offat, n, out.off_words);
	if (m > 1) hipLaunchKernelGGL(k_enc_succ_bins, dim3(SB_GRID), dim3(256), 0, st, d_succ, (int64_t)m, dstats);
	if (n > 0 && m > 0) hipLaunchKernelGGL(k_enc_succ_bins_rows, dim3(SB_GRID), dim3(256), 0, st, d_rowptr, d_succ, n, dstats);
	EncStatsDev hs{};
	if (hipMemcpyAsync(&hs, dstats, sizeof hs, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "the emission kernels failed"; return cleanup(-6); }
	mark();
	out.bits_outdegrees = hs.v[0]; out.bits_references = hs.v[1]; out.bits_blocks = hs.v[2]; out.bits_intervals = hs.v[3]; out.bits_residuals = hs.v[4];
	out.copied_arcs = hs.v[5]; out.intervalised_arcs = hs.v[6]; out.residual_arcs = hs.v[7]; out.tot_ref = hs.v[8]; out.tot_dist = hs.v[9];
	out.max_ref_chain = (int32_t)hs.v[10];
	for (int i = 0; i < 32; i++) { out.residual_gap_bins[i] = hs.resBins[i]; out.successor_gap_bins[i] = hs.succBins[i] - hs.succBinsOff[i]; }
	out.rounds = rounds;
	if (trace && ev.size() == 6) {
		static const char *names[] = { "A cost", "B select", "C lengths+scan", "D emit", "E offsets" };
		float total = 0;
		for (int i = 0; i < 5; i++) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[(size_t)i], ev[(size_t)i + 1]); total += ms; fprintf(stderr, "[bvgpu enc] %-16s %8.3f ms\n", names[i], ms); }
		fprintf(stderr, "[bvgpu enc] total %.3f ms, %d selection rounds, %llu bits\n", total, rounds, (unsigned long long)totalBits);
	}
	return cleanup(0);
}

} // namespace bv
