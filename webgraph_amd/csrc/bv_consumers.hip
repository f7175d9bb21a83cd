// bv_consumers.hip -- consumers of decoded rows that never hand a successor array to the caller.
//
// Every user of the decode path in the reference reduces a successor list as soon as it has it.  Two of them, restated over
// rows that the decode kernels have just written to scratch that stays on the die (a chunk of the graph at a time):
//   k_stats_*     Stats.run's scan (src/it/unimi/dsi/webgraph/Stats.java:111-160): arcs, loops, dangling / terminal nodes,
//                 min / max outdegree and the first node that has it, gap and locality sums, the exponentially binned
//                 histogram of |successor - node|, optionally the indegrees
//   k_bfs_expand  one round of ParallelBreadthFirstVisit (src/it/unimi/dsi/webgraph/algo/ParallelBreadthFirstVisit.java:
//                 146-170): for every node of the frontier and every successor s, marker.compareAndSet(s, -1, mark)
//                 and, where that wins, s joins the next frontier
#include "bv_launch.hpp"

#include <hip/hip_runtime.h>
#include <algorithm>

namespace bv {

constexpr int CS_T = 256, CS_ARCS = 2048; // arcs per block of the per-arc kernels

// the row of arc `a` among rows [rlo, rhi] whose starts are staged in LDS (s_rp[k] = rowptr[rlo + k])
__device__ __forceinline__ int32_t row_of(const int64_t *s_rp, int32_t nrows, int64_t a) {
	int32_t lo = 0, hi = nrows; // last k with s_rp[k] <= a
	while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (s_rp[mid] <= a) lo = mid; else hi = mid; }
	return lo;
}
// rows whose arcs [a0, a1) belong to: first row holding a0 .. row holding a1 - 1; their starts go to LDS (at most CS_ARCS + 1
// rows hold arcs of the slice; empty rows in between make it longer: then the search runs on the global array)
struct RowSlice { int32_t rlo, n; const int64_t *rp; };
__device__ __forceinline__ RowSlice stage_rows(const int64_t *__restrict__ rowptr, int32_t cnt, int64_t a0, int64_t a1, int64_t *s_rp, int32_t *s_b) {
	if (threadIdx.x < 2) {
		const int64_t a = threadIdx.x ? a1 - 1 : a0;
		int32_t lo = 0, hi = cnt; // last row with rowptr[row] <= a  (rowptr[cnt] = arcs > a)
		while (hi - lo > 1) { const int32_t mid = (int32_t)(((int64_t)lo + hi) >> 1); if (rowptr[mid] <= a) lo = mid; else hi = mid; }
		s_b[threadIdx.x] = lo;
	}
	__syncthreads();
	const int32_t rlo = s_b[0], n = s_b[1] - s_b[0] + 1;
	if (n <= CS_ARCS + 1) {
		for (int32_t k = threadIdx.x; k < n; k += CS_T) s_rp[k] = rowptr[rlo + k];
		__syncthreads();
		return RowSlice{ rlo, n, s_rp };
	}
	return RowSlice{ rlo, n, rowptr + rlo };
}

struct StatsDev { // accumulated with atomics
	unsigned long long arcs, loops, dangling, terminal, num_gaps, tot_loc, tot_gap, min_key, max_key, delta[32];
	unsigned long long bad; // arcs whose successor is outside [0, n): a malformed stream (the reference would throw ArrayIndexOutOfBounds at Stats.java:130)
};

// Both kernels run as a few thousand blocks that walk the range and keep their sums to themselves until the end: ONE atomic per block and field.  (Round 5: with a
// block per 256 nodes / 2 048 arcs and an atomic per wave and field, C2's 200 M arcs were 490 000 additions to the same eight words -- same-address atomics run at
// ~88 M/s -- and bvg_scan_stats took 22.3 ms for a graph that scans in 3.0; scripts/stats_time.py.)
constexpr int CS_GRID = 2048;
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *s_w) { // (all threads; the result is valid in thread 0)
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
	__syncthreads();
	unsigned long long t = 0;
	if (threadIdx.x == 0) for (int k = 0; k < CS_T / 64; k++) t += s_w[k];
	return t;
}
__global__ void __launch_bounds__(CS_T) k_stats_nodes(int32_t from, int32_t cnt, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, StatsDev *st) {
	__shared__ unsigned long long s_w[CS_T / 64], s_mn, s_mx;
	unsigned long long dang = 0, term = 0, gaps = 0, totgap = 0, mn = ~0ull, mx = 0;
	if (threadIdx.x == 0) { s_mn = ~0ull; s_mx = 0; }
	for (int64_t s = (int64_t)blockIdx.x * CS_T + threadIdx.x; s < cnt; s += (int64_t)gridDim.x * CS_T) {
		const int64_t lo = rowptr[s], hi = rowptr[s + 1];
		const int64_t d = hi - lo;
		const int32_t curr = from + (int32_t)s;
		if (d == 0) { dang++; term++; }                                       // Stats.java:133-136
		if (d == 1 && succ[lo] == curr) term++;                               // :138
		if (d > 1) {                                                          // :119-123
			const int32_t a = succ[lo], z = succ[hi - 1];
			const int32_t diff = a - curr;
			totgap += (unsigned long long)(int64_t)(z - a) + (unsigned long long)(diff >= 0 ? 2ll * diff : -2ll * diff - 1); // Fast.int2nat
			gaps += (unsigned long long)d;
		}
		mn = min(mn, ((unsigned long long)d << 32) | (uint32_t)curr);                  // smallest outdegree, then the first node that has it (:140-143)
		mx = max(mx, ((unsigned long long)d << 32) | (0xffffffffu - (uint32_t)curr));  // largest outdegree, then the first node that has it (:145-148)
	}
	for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (unsigned long long)__shfl_xor(mn, o, 64)); mx = max(mx, (unsigned long long)__shfl_xor(mx, o, 64)); }
	__syncthreads();
	if ((threadIdx.x & 63) == 0) { atomicMin(&s_mn, mn); atomicMax(&s_mx, mx); }
	dang = block_sum_u64(dang, s_w); term = block_sum_u64(term, s_w); gaps = block_sum_u64(gaps, s_w); totgap = block_sum_u64(totgap, s_w);
	if (threadIdx.x == 0) {
		if (dang) atomicAdd(&st->dangling, dang);
		if (term) atomicAdd(&st->terminal, term);
		if (gaps) { atomicAdd(&st->num_gaps, gaps); atomicAdd(&st->tot_gap, totgap); }
		if (s_mn != ~0ull) { atomicMin(&st->min_key, s_mn); atomicMax(&st->max_key, s_mx); }
	}
}

__global__ void __launch_bounds__(CS_T) k_stats_arcs(int32_t from, int32_t cnt, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, StatsDev *st, int32_t *__restrict__ indegree, int32_t n) {
	__shared__ int64_t s_rp[CS_ARCS + 2];
	__shared__ int32_t s_b[2];
	__shared__ unsigned long long s_delta[32], s_w[CS_T / 64];
	const int64_t arcs = rowptr[cnt];
	if (threadIdx.x < 32) s_delta[threadIdx.x] = 0;
	unsigned long long loops = 0, loc = 0, bad = 0, seen = 0;
	for (int64_t a0 = (int64_t)blockIdx.x * CS_ARCS; a0 < arcs; a0 += (int64_t)gridDim.x * CS_ARCS) { // (uniform in the block)
		const int64_t a1 = min(a0 + CS_ARCS, arcs);
		__syncthreads(); // (the staged rows of the slice before)
		const RowSlice rs = stage_rows(rowptr, cnt, a0, a1, s_rp, s_b);
		__syncthreads();
		for (int64_t a = a0 + threadIdx.x; a < a1; a += CS_T) {
			const int32_t curr = from + rs.rlo + row_of(rs.rp, rs.n, a), sx = succ[a];
			const int64_t dist = (int64_t)sx - curr;
			const unsigned long long ad = (unsigned long long)(dist < 0 ? -dist : dist);
			loc += ad;                                                            // Stats.java:125
			if (sx != curr) atomicAdd(&s_delta[63 - __clzll((long long)ad)], 1ull); // :127  Fast.mostSignificantBit
			else loops++;                                                         // :128
			if ((uint32_t)sx >= (uint32_t)n) bad++;                               // never index with an id the stream made up
			else if (indegree) atomicAdd(&indegree[sx], 1);                       // :130
			seen++;
		}
	}
	loops = block_sum_u64(loops, s_w); loc = block_sum_u64(loc, s_w); bad = block_sum_u64(bad, s_w); seen = block_sum_u64(seen, s_w);
	if (threadIdx.x == 0) { if (loops) atomicAdd(&st->loops, loops); if (loc) atomicAdd(&st->tot_loc, loc); if (bad) atomicAdd(&st->bad, bad); if (seen) atomicAdd(&st->arcs, seen); }
	__syncthreads();
	if (threadIdx.x < 32 && s_delta[threadIdx.x]) atomicAdd(&st->delta[threadIdx.x], s_delta[threadIdx.x]);
}

// One round of the visit over the rows of the frontier's nodes (rowptr / succ = bvg_successors_batch of the frontier).  A block keeps the winners of its 2 048 arcs in
// LDS and appends them with ONE addition to the frontier's counter (round 5: an addition per wave was 195 000 same-address atomics for a frontier of 12 M arcs, half the round).
__global__ void __launch_bounds__(CS_T) k_bfs_expand(const int32_t *__restrict__ frontier, int32_t q, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ,
                                                     int32_t *marker, int32_t n, int32_t round, int parent, int32_t *__restrict__ out, unsigned long long outCap, unsigned long long *outCount) {
	__shared__ int64_t s_rp[CS_ARCS + 2];
	__shared__ int32_t s_b[2], s_won[CS_ARCS], s_n;
	__shared__ unsigned long long s_at;
	const int64_t arcs = rowptr[q], a0 = (int64_t)blockIdx.x * CS_ARCS, a1 = min(a0 + CS_ARCS, arcs);
	if (a0 >= a1) return;
	if (threadIdx.x == 0) s_n = 0;
	const RowSlice rs = stage_rows(rowptr, q, a0, a1, s_rp, s_b);
	__syncthreads();
	for (int64_t a = a0 + threadIdx.x; a < a1; a += CS_T) {
		const int32_t sx = succ[a];
		const int32_t mark = parent ? frontier[rs.rlo + row_of(rs.rp, rs.n, a)] : round;    // ParallelBreadthFirstVisit.java:162
		if ((uint32_t)sx < (uint32_t)n && atomicCAS(&marker[sx], -1, mark) == -1) s_won[atomicAdd(&s_n, 1)] = sx; // :165 marker.compareAndSet(s, -1, mark)
	}
	__syncthreads();
	const int32_t nw = s_n;
	if (nw == 0) return;
	if (threadIdx.x == 0) s_at = atomicAdd(outCount, (unsigned long long)nw);
	__syncthreads();
	const unsigned long long at = s_at;
	for (int32_t k = threadIdx.x; k < nw; k += CS_T) if (at + k < outCap) out[at + k] = s_won[k];
}

void launch_stats(int32_t from, int32_t cnt, const int64_t *rowptr, const int32_t *succ, int64_t arcsUpper, void *statsDev, int32_t *indegree, int32_t n, hipStream_t st) {
	if (cnt <= 0) return;
	hipLaunchKernelGGL(k_stats_nodes, dim3((unsigned)std::min<int64_t>(((int64_t)cnt + CS_T - 1) / CS_T, CS_GRID)), dim3(CS_T), 0, st, from, cnt, rowptr, succ, (StatsDev *)statsDev);
	if (arcsUpper > 0) hipLaunchKernelGGL(k_stats_arcs, dim3((unsigned)std::min<int64_t>((arcsUpper + CS_ARCS - 1) / CS_ARCS, CS_GRID)), dim3(CS_T), 0, st, from, cnt, rowptr, succ, (StatsDev *)statsDev, indegree, n);
}
size_t stats_dev_bytes() { return sizeof(StatsDev); }
// bvg_equal_range: rows of the same nodes decoded from two handles; *differ |= 1 where the row starts or the successors differ (the successors are only looked at
// where both row-start arrays agree up to there: a lane compares position i of both arrays and stays inside both)
__global__ void __launch_bounds__(CS_T) k_rows_differ(int32_t cnt, const int64_t *__restrict__ rpA, const int64_t *__restrict__ rpB, const int32_t *__restrict__ scA, const int32_t *__restrict__ scB, int *__restrict__ differ) {
	const int64_t arcsA = rpA[cnt] - rpA[0], arcsB = rpB[cnt] - rpB[0], arcs = arcsA < arcsB ? arcsA : arcsB, stride = (int64_t)gridDim.x * CS_T;
	bool bad = arcsA != arcsB;
	for (int64_t i = (int64_t)blockIdx.x * CS_T + threadIdx.x; i <= cnt; i += stride) bad |= rpA[i] - rpA[0] != rpB[i] - rpB[0];
	for (int64_t i = (int64_t)blockIdx.x * CS_T + threadIdx.x; i < arcs; i += stride) bad |= scA[i] != scB[i];
	if (__any(bad) && (threadIdx.x & 63) == 0) *differ = 1; // (a plain store: every writer writes the same value)
}
void launch_rows_differ(int32_t cnt, const int64_t *rpA, const int64_t *rpB, const int32_t *scA, const int32_t *scB, int *differ, hipStream_t st) {
	if (cnt < 0) return;
	hipLaunchKernelGGL(k_rows_differ, dim3(CS_GRID), dim3(CS_T), 0, st, cnt, rpA, rpB, scA, scB, differ);
}
void launch_bfs_expand(const int32_t *frontier, int32_t q, const int64_t *rowptr, const int32_t *succ, int64_t arcs, int32_t *marker, int32_t n, int32_t round, int parent,
                       int32_t *out, uint64_t outCap, unsigned long long *outCount, hipStream_t st) {
	if (q <= 0 || arcs <= 0) return;
	hipLaunchKernelGGL(k_bfs_expand, dim3((unsigned)((arcs + CS_ARCS - 1) / CS_ARCS)), dim3(CS_T), 0, st, frontier, q, rowptr, succ, marker, n, round, parent, out, (unsigned long long)outCap, outCount);
}

// One standard (non-systolic) iteration of HyperBall over rows decoded into scratch (src/it/unimi/dsi/webgraph/algo/HyperBall.java:875-915):
// t = counter[node]; for every successor s != node whose counter changed in the previous iteration, t = max(t, counter[s]) register by
// register (:907-913, max() is a register-wise maximum); a counter that changed is stored in the result array and flagged (:972-978).  A counter
// is m = 2^log2m registers, one byte each here (the reference packs registerSize bits into longwords and maximises them broadword: same values).
// One wave per node, lane r takes registers r, r + 64, ...: the successors' counters are read 64 bytes at a time.
// Round 5: rows of fewer than HB_BIG successors by a wave each -- the waves walk the range (one addition to `changed` per block), a row's successors 64 at a time (ids and
// modified flags fetched by the 64 lanes), the counters of the ones that count four at a time in flight; longer rows by a group of sixteen waves each (k_hyperball_big).  Before: a
// wave per node walking its successors one dependent load after the other, an atomic per changed counter -- C2 (m = 64): 219 ms for a graph that scans in 3, its row of 347 500
// successors alone a third of a second of one wave.
constexpr int HB_BIG = 2048, HB_GRID = 4096, HB_BIG_T = 1024;
__device__ __forceinline__ uint8_t hb_max4(uint8_t t, uint8_t a, uint8_t b, uint8_t c, uint8_t d) { const uint8_t x = a > b ? a : b, y = c > d ? c : d, z = x > y ? x : y; return z > t ? z : t; }
// the counters of the successors in [a0, a0 + 64) of a row, register r of each, folded into t (wave-uniform control flow)
__device__ __forceinline__ uint8_t hb_chunk(uint8_t t, int64_t a0, int64_t hi, int32_t node, int32_t n, int32_t m, int32_t r, const int32_t *__restrict__ succ, const uint8_t *__restrict__ regsIn, const uint8_t *__restrict__ modIn) {
	const int lane = threadIdx.x & 63;
	const int64_t a = a0 + lane;
	const int32_t sx = a < hi ? succ[a] : -1;
	const bool use = a < hi && sx != node && (uint32_t)sx < (uint32_t)n && (!modIn || modIn[sx]); // neither self-loops nor unmodified counters influence the computation (:909)
	unsigned long long mask = __ballot(use);
	while (mask) {
		const int k0 = __ffsll((long long)mask) - 1; mask &= mask - 1;
		const int k1 = mask ? __ffsll((long long)mask) - 1 : k0; mask &= mask - 1;
		const int k2 = mask ? __ffsll((long long)mask) - 1 : k1; mask &= mask - 1;
		const int k3 = mask ? __ffsll((long long)mask) - 1 : k2; mask &= mask - 1;
		const int32_t s0 = __shfl(sx, k0, 64), s1 = __shfl(sx, k1, 64), s2 = __shfl(sx, k2, 64), s3 = __shfl(sx, k3, 64);
		if (r < m) {
			const uint8_t u0 = regsIn[(size_t)s0 * m + r], u1 = regsIn[(size_t)s1 * m + r], u2 = regsIn[(size_t)s2 * m + r], u3 = regsIn[(size_t)s3 * m + r];
			t = hb_max4(t, u0, u1, u2, u3);
		}
	}
	return t;
}
__global__ void __launch_bounds__(CS_T) k_hyperball(int32_t from, int32_t cnt, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int32_t n, int32_t m,
                                                    const uint8_t *__restrict__ regsIn, uint8_t *__restrict__ regsOut, const uint8_t *__restrict__ modIn, uint8_t *__restrict__ modOut,
                                                    unsigned long long *__restrict__ changed, int32_t *__restrict__ bigRows, int32_t bigCap, int32_t *__restrict__ bigCount) {
	__shared__ unsigned long long s_w[CS_T / 64];
	const int lane = threadIdx.x & 63;
	const int64_t nWaves = (int64_t)gridDim.x * (CS_T / 64);
	unsigned long long nch = 0;
	for (int64_t row = (int64_t)blockIdx.x * (CS_T / 64) + (threadIdx.x >> 6); row < cnt; row += nWaves) {
		const int64_t lo = rowptr[row], hi = rowptr[row + 1];
		if (hi - lo >= HB_BIG) { // a long row: listed for k_hyperball_big (long rows come in runs of neighbours: a list deals them to all groups)
			if (lane == 0) { const int32_t k = atomicAdd(bigCount, 1); if (k < bigCap) bigRows[k] = (int32_t)row; }
			continue;
		}
		const int32_t node = from + (int32_t)row;
		bool any = false;
		for (int32_t r0 = 0; r0 < m; r0 += 64) {
			const int32_t r = r0 + lane;
			const uint8_t t0 = r < m ? regsIn[(size_t)node * m + r] : 0;
			uint8_t t = t0;
			for (int64_t a0 = lo; a0 < hi; a0 += 64) t = hb_chunk(t, a0, hi, node, n, m, r, succ, regsIn, modIn);
			if (r < m) regsOut[(size_t)node * m + r] = t;
			any |= t != t0;
		}
		const bool rowChanged = __any(any);
		if (lane == 0) { modOut[node] = rowChanged ? 1 : 0; nch += rowChanged ? 1 : 0; }
	}
	nch = block_sum_u64(nch, s_w);
	if (threadIdx.x == 0 && nch) atomicAdd(changed, nch);
}
// the rows of HB_BIG successors and more: a group of sixteen waves per row, every wave a share of the row's chunks of 64 successors, the waves' maxima joined in LDS
__global__ void __launch_bounds__(HB_BIG_T) k_hyperball_big(int32_t from, int32_t cnt, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int32_t n, int32_t m,
                                                            const uint8_t *__restrict__ regsIn, uint8_t *__restrict__ regsOut, const uint8_t *__restrict__ modIn, uint8_t *__restrict__ modOut,
                                                            unsigned long long *__restrict__ changed, const int32_t *__restrict__ bigRows, int32_t bigCap, const int32_t *__restrict__ bigCount) {
	constexpr int NWV = HB_BIG_T / 64;
	__shared__ uint8_t s_t[NWV][64];
	__shared__ int32_t s_any;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	unsigned long long nch = 0;
	{
		const int32_t nb = min(*bigCount, bigCap); // (bigCap >= arcs / HB_BIG: every long row is listed)
		for (int32_t q = blockIdx.x; q < nb; q += gridDim.x) { // (uniform in the block)
			const int32_t rw = bigRows[q], node = from + rw;
			const int64_t lo = rowptr[rw], hi = rowptr[rw + 1];
			if (threadIdx.x == 0) s_any = 0;
			for (int32_t r0 = 0; r0 < m; r0 += 64) {
				const int32_t r = r0 + lane;
				const uint8_t t0 = r < m ? regsIn[(size_t)node * m + r] : 0;
				uint8_t t = t0;
				for (int64_t a0 = lo + 64 * wv; a0 < hi; a0 += 64 * NWV) t = hb_chunk(t, a0, hi, node, n, m, r, succ, regsIn, modIn);
				s_t[wv][lane] = t;
				__syncthreads();
				if (wv == 0) {
					uint8_t x = t;
					for (int k = 1; k < NWV; k++) { const uint8_t y = s_t[k][lane]; x = y > x ? y : x; }
					if (r < m) regsOut[(size_t)node * m + r] = x;
					if (__any(x != t0) && lane == 0) s_any = 1;
				}
				__syncthreads();
			}
			if (threadIdx.x == 0) { modOut[node] = s_any ? 1 : 0; nch += s_any ? 1 : 0; }
			__syncthreads();
		}
	}
	if (threadIdx.x == 0 && nch) atomicAdd(changed, nch);
}
int64_t hyperball_big_cap(int64_t arcs) { return arcs / HB_BIG + 1; }
// bigRows: hyperball_big_cap(arcs of the piece) ints, bigCount: one int (zeroed here)
void launch_hyperball(int32_t from, int32_t cnt, const int64_t *rowptr, const int32_t *succ, int32_t n, int32_t m, const uint8_t *regsIn, uint8_t *regsOut, const uint8_t *modIn, uint8_t *modOut,
                      unsigned long long *changed, int32_t *bigRows, int32_t bigCap, int32_t *bigCount, hipStream_t st) {
	if (cnt <= 0) return;
	(void)hipMemsetAsync(bigCount, 0, sizeof(int32_t), st);
	hipLaunchKernelGGL(k_hyperball, dim3((unsigned)std::min<int64_t>(((int64_t)cnt + CS_T / 64 - 1) / (CS_T / 64), HB_GRID)), dim3(CS_T), 0, st, from, cnt, rowptr, succ, n, m, regsIn, regsOut, modIn, modOut, changed, bigRows, bigCap, bigCount);
	hipLaunchKernelGGL(k_hyperball_big, dim3(512), dim3(HB_BIG_T), 0, st, from, cnt, rowptr, succ, n, m, regsIn, regsOut, modIn, modOut, changed, bigRows, bigCap, bigCount);
}

} // namespace bv
