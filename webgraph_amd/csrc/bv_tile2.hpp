// bv_tile2.hpp -- a contiguous tile of records decoded by one work-group, the long residual sections SEGMENT BY SEGMENT (gfx950).
//
// bv_tile.hpp gave every record of the tile to one lane: the tile then lasts as long as its longest record while most lanes
// idle, and the merge loop of a whole record (interval or residual? refill? flush?) makes the 64 lanes of a wave execute
// every branch one after the other.  Here the work of a tile is cut into pieces of equal size with the same loop body:
//
//   H  one lane per record, node order: header, copy-block totals, interval section (BVG:1048-1096).  Records with few
//      successors are finished on the spot (the fused merge of bv_tile.hpp, at most T2_JOB_MIN - 1 iterations).  The others
//      become JOBS: their intervals go to an LDS pool as {left, arcs before it, length}, their residual section
//      [q, qend) -- it ends with the record -- is left for the next phases.
//   S  the residual section of every job is cut into segments of ~12 codes (at least 128 bits; wider if the tile would
//      have more than T2_SEGS of them), numbered consecutively, job after job.
//   R1 one lane per segment: decode the codes that START in the segment (count, sum of the gaps), beginning at a guessed
//      boundary found by a short run-in -- universal codes re-synchronise within a few codewords (the idea of the
//      cooperative decoder of bv_coop.hpp, flattened over all records of the tile).
//   R2 a segment must start where its left neighbour ended: whoever disagrees parses again from there, until nobody does
//      (the first segment of a section starts on a true boundary, so the fixed point is exact).
//   R3 prefix sums over the segments: index and value of the first residual of every segment.
//   R4 one lane per segment again: the residuals at their final places in the tail of the row, between the intervals
//      (residual j goes to j + arcs of the intervals left of it); each interval learns how many residuals precede it.
//   X  one lane per interval: the interval's ids at their final place (IntIntervalSequenceIterator.java:64-78).
//
// Same contract as every parse kernel: the record's extras (intervals merged with residuals, MergedIntIterator.java:50-74)
// end up in row[copied..d).  Default codings only (DEF 1 / 2).  Anything that does not fit the LDS tables of a tile (jobs,
// intervals) is decoded by one lane in one go, like the short records.
#pragma once
#include "bv_tile.hpp"

namespace bv {

constexpr int T2_T = 256;
constexpr int T2_JOBS = 512;    // records per tile whose residual section is decoded segment by segment
constexpr int T2_SEGS = 1024;   // segments per tile
constexpr int T2_SPL = T2_SEGS / T2_T;
constexpr int T2_IVS = 1024;    // interval entries per tile
constexpr int T2_JOB_MIN = 24;  // records with fewer successors are decoded by one lane in one go
constexpr int T2_B_MIN = 128;   // segment width in bits, at least
constexpr int T2_RUNIN = 64;    // bits a segment's parse starts before its nominal boundary

// Header of a record whose cursor `br` stands at its first bit: skips the outdegree and the reference, walks the copy blocks.
// Returns false (and flags the stream) when the record is malformed.  On return br stands behind the block list.
template <int ZK>
__device__ __forceinline__ bool tile_header(const GraphDev &g, const TWin &tw, TFast &br, uint32_t zk, int32_t d, bool hasRef, int64_t dref, int64_t &copied, int &e) {
	(void)br.code<1, ZK>(tw, zk, e);              // outdegree (known from k_headers)
	if (g.W > 0) (void)br.code<2, ZK>(tw, zk, e); // reference
	copied = 0;
	if (hasRef) { // BVG:1058-1071
		const uint64_t bc = br.code<1, ZK>(tw, zk, e);
		int64_t total = 0;
		if (bc > (uint64_t)dref + 1) e |= E_FORMAT;
		else {
			for (uint64_t b = 0; b < bc; b++) {
				int64_t len;
				if (!block_len_ok(br.code<1, ZK>(tw, zk, e), b == 0, total, dref, len)) { e |= E_FORMAT; break; }
				total += len;
				if (!(b & 1)) copied += len;
			}
			if (!(bc & 1)) copied += dref - total;
		}
	}
	if ((int64_t)d - copied < 0) e |= E_FORMAT;
	return e == 0;
}

// The rest of a record by one lane: interval section + residual section merged into out[0..extra) (the fused loop of
// parse_node_tile).  br stands behind the block list.
template <int ZK>
__device__ __forceinline__ void tile_extras_inline(const GraphDev &g, const TWin &tw, TFast br, uint32_t zk, int32_t x, int32_t nExtra, int32_t *__restrict__ out, int &e) {
	int64_t nIntervals = 0, intervalArcs = 0;
	TFast bi = br;
	if (g.minInt != 0) { // BVG:1073-1096: skip-parse to find the residual section and the number of residuals
		nIntervals = (int64_t)br.code<1, ZK>(tw, zk, e);
		if (nIntervals > nExtra) { e |= E_FORMAT; return; }
		bi = br;
		for (int64_t i = 0; i < nIntervals; i++) {
			(void)br.code<1, ZK>(tw, zk, e);
			const uint64_t len = br.code<1, ZK>(tw, zk, e);
			if (len > (uint64_t)nExtra) { e |= E_FORMAT; break; }
			intervalArcs += (int64_t)len + g.minInt;
		}
	}
	const int64_t nRes = (int64_t)nExtra - intervalArcs;
	if (nRes < 0 || e) { e |= E_FORMAT; return; }
	int32_t k = 0;
	const int32_t head = min(nExtra, (int32_t)(((16u - ((uint32_t)(uintptr_t)out & 15u)) & 15u) >> 2));
	int32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, on = 0;
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
		if (k < head) { out[k++] = val; continue; }
		o0 = o1; o1 = o2; o2 = o3; o3 = val; k++;
		if (++on == 4) { *(int4 *)(out + k - 4) = int4{ o0, o1, o2, o3 }; on = 0; }
	}
	if (on == 3) { out[k - 3] = o1; out[k - 2] = o2; out[k - 1] = o3; }
	else if (on == 2) { out[k - 2] = o2; out[k - 1] = o3; }
	else if (on == 1) out[k - 1] = o3;
}

// The codes of a residual section that START in [s, segEnd): how many, the sum of what they add to the running id (the
// zig-zag value for the very first code of the section, gap + 1 for the others), and where the last of them ends.
template <int ZK>
__device__ __forceinline__ void seg_parse(const TWin &tw, uint32_t zk, uint32_t s, uint32_t segEnd, uint32_t secStart, uint32_t secEnd, uint32_t &e, uint32_t &c, int32_t &sum) {
	TFast p{ s };
	c = 0; sum = 0;
	int err = 0; // a speculative parse may run through garbage: errors only stop it
	if (p.q < segEnd && p.q == secStart) { sum = (int32_t)nat2int(p.code<0, ZK>(tw, zk, err)); c = 1; }
	while (p.q < segEnd && !err) { sum += (int32_t)p.code<0, ZK>(tw, zk, err) + 1; c++; }
	e = err ? secEnd : min(p.q, secEnd);
}

template <int ZK>
__device__ __attribute__((noinline)) void parse_node_tile_unstaged(GraphDev g, TWin tw, int32_t x, int32_t d, bool hasRef, int64_t dref, int32_t *row, int *err) {
	parse_node_tile<ZK>(g, tw, x, d, hasRef, dref, row, err); // (arguments by value: a pointer to the caller's copies would put them into scratch memory)
}

template <int DEF>
__global__ void __launch_bounds__(T2_T) k_parse_tile2(GraphDev g, RangeView v, const int32_t *__restrict__ tb, int *__restrict__ err) {
	__shared__ __attribute__((aligned(16))) uint32_t s_win[TILE_WIN_WORDS];
	__shared__ uint32_t j_q[T2_JOBS], j_qend[T2_JOBS], j_row[T2_JOBS]; // residual section [q, qend) in bits from the first staged word; element index of row[copied] from the tile's first row
	__shared__ int32_t j_x[T2_JOBS];
	__shared__ uint16_t j_nres[T2_JOBS], j_ic[T2_JOBS], j_iv[T2_JOBS], j_seg[T2_JOBS + 2], j_B[T2_JOBS], j_R[T2_JOBS]; // ..., segment width and run-in of the job's section
	__shared__ uint32_t sg_s[T2_SEGS], sg_e[T2_SEGS], sg_c[T2_SEGS + 1]; // start, end; codes (then: codes before the segment)
	__shared__ int32_t sg_sum[T2_SEGS + 1];                            // sum (then: sum before the segment)
	__shared__ int32_t iv_left[T2_IVS];
	__shared__ uint16_t iv_p[T2_IVS], iv_len[T2_IVS], iv_rank[T2_IVS], iv_job[T2_IVS];
	__shared__ int32_t s_njobs, s_nivs, s_S, s_wsum[2][T2_T / 64 + 1];
	__shared__ unsigned long long s_bits;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int32_t a = tb[blockIdx.x], b = tb[blockIdx.x + 1];
	if (a >= b) return;
	constexpr int ZK = DEF == 1 ? 3 : 0;
	const uint32_t zk = ZK == 3 ? 3u : (uint32_t)g.zetaK;
	// ---- the tile's slice of the stream -> LDS (all loads of a lane in flight together, then the byte-swapped stores)
	const uint64_t p0 = (uint64_t)g.offsets[v.lo + a], p1 = (uint64_t)g.offsets[v.lo + b];
	const uint64_t w0 = (p0 >> 5) & ~(uint64_t)3;
	const uint32_t nw = (uint32_t)min<uint64_t>(TILE_WIN_WORDS, (((p1 + 31) >> 5) - w0 + 3 + 3) & ~(uint64_t)3);
	{
		constexpr int NV = (TILE_WIN_WORDS / 4 + T2_T - 1) / T2_T;
		const uint4 *src4 = (const uint4 *)(g.bits + w0);
		const uint64_t lim4 = (g.nwords + 8 - w0) / 4; // the image is followed by >= 8 zero words
		uint4 q4[NV];
#pragma unroll
		for (int k = 0; k < NV; k++) { const uint32_t i = (uint32_t)tid + (uint32_t)k * T2_T; q4[k] = (i < nw / 4 && i < lim4) ? src4[i] : uint4{ 0u, 0u, 0u, 0u }; }
#pragma unroll
		for (int k = 0; k < NV; k++) {
			const uint32_t i = (uint32_t)tid + (uint32_t)k * T2_T;
			if (i < nw / 4) ((uint4 *)s_win)[i] = uint4{ __builtin_bswap32(q4[k].x), __builtin_bswap32(q4[k].y), __builtin_bswap32(q4[k].z), __builtin_bswap32(q4[k].w) };
		}
	}
	if (tid == 0) { s_njobs = 0; s_nivs = 0; s_bits = 0; }
	__syncthreads();
	// BVGPU_STATS=1: clock ticks per phase, summed over the tiles (slots 0..7: stage H S R1 R2 R3 R4 X; 8: tiles, 9: R2 rounds, 10: jobs, 11: segments)
	unsigned long long tk = g.stats ? __builtin_readcyclecounter() : 0;
#define T2_TICK(slot) do { if (g.stats) { const unsigned long long now_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g.stats[slot], now_ - tk); tk = now_; } } while (0)
	T2_TICK(0);
	const TWin tw{ (const lds_u32 *)s_win, nw, w0, g.bits, g.nwords };
	const int64_t E0 = v.rowstart[a], hsplit = v.rowstart[v.nh];
	auto gaddr = [&](int64_t el) -> int32_t * { return el < hsplit ? v.halo + el : v.succ + (el - hsplit); }; // element el of the view's rows (halo rows | caller's rows)

	// ---- H: one lane per record
	constexpr int RPT = TILE_NODES / T2_T;
#pragma unroll 1
	for (int k = 0; k < RPT; k++) {
		const int32_t s = a + tid + k * T2_T;
		if (s >= b) break;
		const int32_t d = v.outd[s];
		if (d == 0 || d >= v.coopmin()) continue; // nothing to decode / decoded by whole waves (k_parse_big)
		if (!v.fits(s)) { atomicOr(err, s >= v.nh ? E_CAP : E_HALO); continue; }
		const int32_t r = v.ref[s], x = v.lo + s;
		const int64_t dref = r > 0 ? (int64_t)v.outd[s - r] : 0;
		const uint64_t px = (uint64_t)g.offsets[x], pend = (uint64_t)g.offsets[x + 1];
		if (((pend - (w0 << 5)) >> 5) + 4 > (uint64_t)nw) { // the record overhangs the staged slice: the checked reader, one lane, out of line
			parse_node_tile_unstaged<ZK>(g, tw, x, d, r > 0, dref, v.row(s), err);
			continue;
		}
		TFast br{ (uint32_t)(px - (w0 << 5)) };
		int e = 0;
		int64_t copied;
		if (!tile_header<ZK>(g, tw, br, zk, d, r > 0, dref, copied, e)) { atomicOr(err, e | E_FORMAT); continue; }
		const int32_t extra = d - (int32_t)copied;
		if (extra == 0) continue;
		const int64_t rs = v.rowstart[s];
		const uint64_t relEnd = pend - (w0 << 5), relRow = (uint64_t)(rs - E0) + (uint64_t)copied;
		bool job = d >= T2_JOB_MIN && d < 16384 && relEnd < 0x7fffffffull && relRow < 0x7fffffffull; // (16-bit fields in the tile's tables)
		int32_t slot = -1, ivb = 0;
		uint64_t ic = 0;
		TFast bj = br;
		if (job) {
			if (g.minInt != 0) {
				ic = bj.code<1, ZK>(tw, zk, e);
				if (ic > (uint64_t)extra) { atomicOr(err, E_FORMAT); continue; }
			}
			if (ic) { ivb = atomicAdd(&s_nivs, (int32_t)ic); if (ivb + (int64_t)ic > T2_IVS) job = false; }
			if (job) { slot = atomicAdd(&s_njobs, 1); if (slot >= T2_JOBS) job = false; }
			if (!job && ic) for (int64_t i = ivb; i < min<int64_t>(ivb + (int64_t)ic, T2_IVS); i++) iv_len[i] = 0; // pool entries taken in vain: nothing to expand
		}
		if (!job) { // short record (or no room in the tile's tables): one lane, one go
			tile_extras_inline<ZK>(g, tw, br, zk, x, extra, gaddr(rs + copied), e);
			if (e) atomicOr(err, e);
			continue;
		}
		// intervals -> pool (BVG:1084-1093); ids are Java ints: 32-bit wrapping arithmetic
		int64_t intervalArcs = 0;
		int32_t prev = 0;
		bool bad = false;
		for (uint64_t i = 0; i < ic; i++) {
			const uint64_t gl = bj.code<1, ZK>(tw, zk, e), ln = bj.code<1, ZK>(tw, zk, e);
			if (ln > (uint64_t)extra) { bad = true; break; }
			const int32_t left = i == 0 ? (int32_t)((int64_t)x + nat2int(gl)) : prev + (int32_t)gl + 1;
			const int32_t len = (int32_t)ln + g.minInt;
			iv_left[ivb + i] = left; iv_p[ivb + i] = (uint16_t)intervalArcs; iv_len[ivb + i] = (uint16_t)len; iv_rank[ivb + i] = 0xffffu; iv_job[ivb + i] = (uint16_t)slot;
			prev = left + len;
			intervalArcs += len;
			if (intervalArcs > extra) { bad = true; break; }
		}
		const int64_t nRes = (int64_t)extra - intervalArcs;
		const uint32_t q = bj.q;
		if (bad || e || nRes < 0 || q > (uint32_t)relEnd) {
			atomicOr(err, e | E_FORMAT);
			// the job slot stays, emptied: no segments, no intervals to expand
			for (uint64_t i = 0; i < ic; i++) iv_len[ivb + i] = 0;
			j_nres[slot] = 0; j_ic[slot] = 0; j_q[slot] = 0; j_qend[slot] = 0; j_row[slot] = 0; j_x[slot] = x; j_iv[slot] = 0;
			continue;
		}
		j_q[slot] = q; j_qend[slot] = (uint32_t)relEnd; j_row[slot] = (uint32_t)relRow; j_x[slot] = x;
		j_nres[slot] = (uint16_t)nRes; j_ic[slot] = (uint16_t)ic; j_iv[slot] = (uint16_t)ivb;
		if (nRes) atomicAdd(&s_bits, (unsigned long long)((uint32_t)relEnd - q));
	}
	__syncthreads();
	T2_TICK(1);
	const int32_t njobs = min(s_njobs, T2_JOBS), nivs = min(s_nivs, T2_IVS);
	if (g.stats && tid == 0) { atomicAdd(&g.stats[8], 1ull); atomicAdd(&g.stats[10], (unsigned long long)njobs); }
	if (njobs == 0) return;
	// ---- S: segments.  A section is cut into pieces of ~12 codes (at least T2_B_MIN bits): a speculative parse locks onto the
	// code boundaries within a few codewords, so the run-in before a piece (5 codes) must be short against the piece.  If
	// the tile would have more than T2_SEGS pieces, every job's width is scaled up.
	auto block_incl_scan = [&](int32_t vv, int which) -> int32_t { // inclusive scan over the block's threads, thread order
		int32_t inc = vv;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const int32_t t2 = __shfl_up(inc, o, 64); if (lane >= o) inc += t2; }
		if (lane == 63) s_wsum[which][wave] = inc;
		__syncthreads();
		int32_t base = 0;
#pragma unroll
		for (int w = 0; w < T2_T / 64; w++) if (w < wave) base += s_wsum[which][w];
		__syncthreads();
		return base + inc;
	};
	{
		constexpr int JPT = T2_JOBS / T2_T;
		uint32_t bits[JPT], width[JPT], avg[JPT];
		int32_t ns[JPT], tot = 0;
#pragma unroll
		for (int i = 0; i < JPT; i++) {
			const int32_t j = tid * JPT + i;
			bits[i] = (j < njobs && j_nres[j]) ? j_qend[j] - j_q[j] : 0u;
			avg[i] = bits[i] ? (bits[i] + j_nres[j] - 1) / j_nres[j] : 1u;
			width[i] = min(max(12u * avg[i], (uint32_t)T2_B_MIN), 8192u);
			ns[i] = bits[i] ? (int32_t)((bits[i] + width[i] - 1) / width[i]) : 0;
			tot += ns[i];
		}
		int32_t inc = block_incl_scan(tot, 0);
		if (tid == T2_T - 1) s_S = inc;
		__syncthreads();
		if (s_S > T2_SEGS) { // (uniform) too many pieces: wider ones
			const uint32_t f = (uint32_t)((s_S + (T2_SEGS - njobs) - 1) / max(T2_SEGS - njobs, 1)) + 1u;
			tot = 0;
#pragma unroll
			for (int i = 0; i < JPT; i++) { width[i] = min(width[i] * f, 65535u); ns[i] = bits[i] ? (int32_t)((bits[i] + width[i] - 1) / width[i]) : 0; tot += ns[i]; }
			__syncthreads();
			inc = block_incl_scan(tot, 0);
			if (tid == T2_T - 1) s_S = inc;
		}
		int32_t run = inc - tot;
#pragma unroll
		for (int i = 0; i < JPT; i++) {
			const int32_t j = tid * JPT + i;
			if (j <= njobs && j < T2_JOBS + 1) j_seg[j] = (uint16_t)min(run, T2_SEGS);
			if (j < njobs) { j_B[j] = (uint16_t)width[i]; j_R[j] = (uint16_t)min(max(64u, 5u * avg[i]), width[i]); }
			run += ns[i];
		}
	}
	__syncthreads();
	const int32_t S = min(s_S, T2_SEGS);
	if (tid == 0) j_seg[njobs] = (uint16_t)S;
	__syncthreads();
	auto job_of = [&](int32_t gs) -> int32_t { // the job whose run of segments holds segment gs
		int32_t lo2 = 0, hi2 = njobs; // last j with j_seg[j] <= gs
		while (hi2 - lo2 > 1) { const int32_t mid = (lo2 + hi2) >> 1; if ((int32_t)j_seg[mid] <= gs) lo2 = mid; else hi2 = mid; }
		return lo2;
	};
	T2_TICK(2);
	if (g.stats && tid == 0) atomicAdd(&g.stats[11], (unsigned long long)S);
	// ---- R1: every segment parsed from a guessed boundary
	int32_t myJob[T2_SPL];
#pragma unroll
	for (int i = 0; i < T2_SPL; i++) {
		const int32_t gs = tid + i * T2_T;
		myJob[i] = -1;
		if (gs >= S) continue;
		const int32_t j = job_of(gs), k = gs - (int32_t)j_seg[j];
		myJob[i] = j;
		const uint32_t q = j_q[j], qend = j_qend[j];
		const uint32_t B = j_B[j];
		const uint32_t nominal = q + (uint32_t)k * B, segEnd = min(nominal + B, qend);
		uint32_t s = nominal;
		if (k > 0) { // run-in: start a little before the segment, so that the parse has locked onto the code boundaries when it enters it
			TFast p{ nominal - min((uint32_t)j_R[j], nominal - q) };
			int e2 = 0;
			while (p.q < nominal && !e2) (void)p.code<0, ZK>(tw, zk, e2);
			s = e2 ? nominal : min(p.q, qend);
		}
		uint32_t e, c; int32_t sum;
		seg_parse<ZK>(tw, zk, s, segEnd, q, qend, e, c, sum);
		sg_s[gs] = s; sg_e[gs] = e; sg_c[gs] = c; sg_sum[gs] = sum;
	}
	__syncthreads();
	T2_TICK(3);
	// ---- R2: a segment starts where its left neighbour ended
	for (int round = 0; round < T2_SEGS + 2; round++) {
		uint32_t want[T2_SPL];
#pragma unroll
		for (int i = 0; i < T2_SPL; i++) {
			const int32_t gs = tid + i * T2_T;
			want[i] = 0xffffffffu;
			if (gs < S && myJob[i] >= 0 && gs > (int32_t)j_seg[myJob[i]]) want[i] = sg_e[gs - 1];
		}
		__syncthreads();
		bool changed = false;
#pragma unroll
		for (int i = 0; i < T2_SPL; i++) {
			const int32_t gs = tid + i * T2_T;
			if (want[i] == 0xffffffffu || want[i] == sg_s[gs]) continue;
			const int32_t j = myJob[i], k = gs - (int32_t)j_seg[j];
			const uint32_t q = j_q[j], qend = j_qend[j];
			const uint32_t segEnd = min(q + (uint32_t)(k + 1) * (uint32_t)j_B[j], qend);
			uint32_t e = want[i], c = 0; int32_t sum = 0; // (the neighbour's last code may run past this whole segment)
			if (want[i] < segEnd) seg_parse<ZK>(tw, zk, want[i], segEnd, q, qend, e, c, sum);
			sg_s[gs] = want[i]; sg_e[gs] = e; sg_c[gs] = c; sg_sum[gs] = sum;
			changed = true;
		}
		if (g.stats && tid == 0) atomicAdd(&g.stats[9], 1ull);
		if (!__syncthreads_or(changed)) break;
	}
	T2_TICK(4);
	// ---- R3: codes and sums before every segment (exclusive prefix in segment order: chunk i = segments [i * T2_T, (i + 1) * T2_T))
	{
		int32_t carryC = 0, carryS = 0;
#pragma unroll
		for (int i = 0; i < T2_SPL; i++) {
			const int32_t gs = tid + i * T2_T;
			if (i * T2_T > S) break; // (uniform)
			const int32_t c = gs < S ? (int32_t)sg_c[gs] : 0, sm = gs < S ? sg_sum[gs] : 0;
			const int32_t ci = block_incl_scan(c, 0), si = block_incl_scan(sm, 1);
			if (gs <= S) { sg_c[gs] = (uint32_t)(carryC + ci - c); sg_sum[gs] = carryS + si - sm; }
			if (tid == T2_T - 1) { s_wsum[0][T2_T / 64] = carryC + ci; s_wsum[1][T2_T / 64] = carryS + si; }
			__syncthreads();
			carryC = s_wsum[0][T2_T / 64]; carryS = s_wsum[1][T2_T / 64];
			if (tid == 0 && (i + 1) * T2_T == T2_SEGS) { sg_c[T2_SEGS] = (uint32_t)carryC; sg_sum[T2_SEGS] = carryS; } // (the entry behind the last segment of a full tile)
			__syncthreads();
		}
	}
	__syncthreads();
	T2_TICK(5);
	// ---- R4: the residuals at their final places; every interval learns how many residuals precede it
#pragma unroll
	for (int i = 0; i < T2_SPL; i++) {
		const int32_t gs = tid + i * T2_T;
		if (gs >= S) continue;
		const int32_t j = myJob[i], g0 = (int32_t)j_seg[j], g1 = (int32_t)j_seg[j + 1];
		const int32_t nRes = j_nres[j];
		const uint32_t cnt = sg_c[gs + 1] - sg_c[gs];
		int32_t jj = (int32_t)(sg_c[gs] - sg_c[g0]);
		if (gs == g1 - 1 && jj + (int32_t)cnt != nRes) atomicOr(err, E_FORMAT); // the section does not hold the residuals the header promises
		int32_t val = j_x[j] + (sg_sum[gs] - sg_sum[g0]);
		const int32_t ic = j_ic[j], ivb = j_iv[j];
		const int64_t rowEl = E0 + (int64_t)j_row[j];
		const int32_t ivArcs = ic ? (int32_t)iv_p[ivb + ic - 1] + (int32_t)iv_len[ivb + ic - 1] : 0;
		int32_t ii = 0;
		if (ic && jj > 0) { // first interval that the residuals before this segment have not passed: left >= the residual before mine
			int32_t lo2 = 0, hi2 = ic;
			while (lo2 < hi2) { const int32_t mid = (lo2 + hi2) >> 1; if (iv_left[ivb + mid] < val) lo2 = mid + 1; else hi2 = mid; }
			ii = lo2;
		}
		int32_t before = ic ? (ii < ic ? (int32_t)iv_p[ivb + ii] : ivArcs) : 0;
		int32_t nextLeft = ii < ic ? iv_left[ivb + ii] : 0x7fffffff;
		TFast p{ sg_s[gs] };
		const uint32_t secStart = j_q[j];
		int e2 = 0;
		for (uint32_t t2 = 0; t2 < cnt && jj < nRes; t2++, jj++) {
			const bool first = p.q == secStart;
			const uint64_t cv = p.code<0, ZK>(tw, zk, e2);
			val += first ? (int32_t)nat2int(cv) : (int32_t)cv + 1; // BVG:954, :966
			while (nextLeft < val && ii < ic) { // interval ii sits after jj residuals
				iv_rank[ivb + ii] = (uint16_t)jj; ii++;
				nextLeft = ii < ic ? iv_left[ivb + ii] : 0x7fffffff;
				before = ii < ic ? (int32_t)iv_p[ivb + ii] : ivArcs;
			}
			*gaddr(rowEl + jj + before) = val;
		}
		if (e2) atomicOr(err, e2);
	}
	__syncthreads();
	T2_TICK(6);
	// ---- X: interval ids at their final places: interval i occupies [arcs before it + residuals before it, + len)
	for (int32_t base = 0; base < nivs; base += T2_T) {
		const int32_t idx = base + tid;
		int32_t left = 0, len = 0;
		int64_t el = 0;
		if (idx < nivs && iv_len[idx]) {
			const int32_t j = iv_job[idx];
			const int32_t rk = iv_rank[idx] == 0xffffu ? (int32_t)j_nres[j] : (int32_t)iv_rank[idx];
			left = iv_left[idx]; len = iv_len[idx];
			el = E0 + (int64_t)j_row[j] + (int32_t)iv_p[idx] + rk;
		}
		const bool isLong = len > 16;
		if (!isLong) for (int32_t t2 = 0; t2 < len; t2++) *gaddr(el + t2) = left + t2;
		unsigned long long lm = __ballot(isLong);
		while (lm) { // long intervals: a whole wave fills one at a time
			const int srcl = __ffsll((long long)lm) - 1;
			lm &= lm - 1;
			const int32_t L = __shfl(left, srcl, 64), Nn = __shfl(len, srcl, 64);
			const int64_t P = shfl_i64(el, srcl);
			for (int32_t t2 = lane; t2 < Nn; t2 += 64) *gaddr(P + t2) = L + t2;
		}
	}
	T2_TICK(7);
#undef T2_TICK
}

} // namespace bv
