// bv_strip.hip -- the strip kernel: device side of bv_strip.hpp (gfx950).  One work-group of STRIP_T threads owns the
// records that start in one slice of the stream; see bv_strip.hpp for the phases and for why the unit of work is a
// segment and never a record.  What lives here: the strip bounds, the carve-up of the LDS pool, staging, the block-wide
// scan and the counting sorts, the hand-out of work items to wavefronts (64 at a time from an LDS counter, longest first),
// the barriers between the phases, and the write-out.
#include "bv_strip.hpp"
#include "bv_launch.hpp"

namespace bv {
using namespace bvs;

typedef __attribute__((address_space(3))) uint32_t l_u32; // LDS-qualified: accesses through these are ds_* instructions, never flat ones
typedef __attribute__((address_space(3))) uint16_t l_u16;
typedef __attribute__((address_space(3))) int32_t l_i32;
typedef __attribute__((address_space(3))) Seg l_seg;
using StripL = StripT<l_u32 *, l_u16 *, l_i32 *, l_seg *>;

// counters of a strip (LDS)
enum : int { C_NARCS = 0, C_NIV, C_NBLK, C_NSEG, C_IVLIM, C_SEGLIM, C_NLSEG, C_NLREC, C_NLIV, C_FETCH, C_FETCH2, C_HIST = 16, C_WSUM = 48, C_TOTAL = 64 };

__device__ __forceinline__ int lds_add(l_i32 *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_min(l_i32 *p, int v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// a wave takes the next 64 items of a list; returns the first one's index (uniform)
__device__ __forceinline__ int fetch64(l_i32 *ctr) {
	int b = 0;
	if ((threadIdx.x & 63) == 0) b = lds_add(ctr, 64);
	return __builtin_amdgcn_readfirstlane(b);
}

// strip t = the slots s of the view with  t * SPAN_W <= weight(s) < (t+1) * SPAN_W,
// weight(s) = (offsets[lo+s] - offsets[lo]) + NODE_W * s + ARC_W * (rowstart[s] - rowstart[0])
__global__ void __launch_bounds__(256) k_strip_bounds(const int64_t *__restrict__ offsets, const int64_t *__restrict__ rowstart, int32_t lo, int32_t cnt, int32_t ntiles,
                                                      int32_t *__restrict__ tb, int32_t *__restrict__ escCtl, int *__restrict__ err) {
	const int32_t t = blockIdx.x * 256 + threadIdx.x;
	if (t == 0) { escCtl[0] = 0; escCtl[2] = 0; } // the escape list of the strip kernel behind this one: count, queue head
	if (t > ntiles) return;
	const int64_t target = (int64_t)t * SPAN_W, base = offsets[lo], r0 = rowstart[0];
	int32_t a = 0, b = cnt; // first s in [0, cnt) with weight(s) >= target, cnt if none
	while (a < b) {
		const int32_t mid = (int32_t)(((int64_t)a + b) >> 1);
		if ((offsets[lo + mid] - base) + (int64_t)NODE_W * mid + (int64_t)ARC_W * (rowstart[mid] - r0) < target) a = mid + 1; else b = mid;
	}
	tb[t] = a;
	// the grid was sized from an upper bound of the job's arcs (the capacity of the caller's buffer): rows beyond it do not fit that buffer
	if (t == ntiles && a < cnt) atomicOr(err, E_CAP);
}

constexpr int RPT = (MAX_NODES + STRIP_T - 1) / STRIP_T; // records per thread (consecutive)

template <int ZK>
__global__ void __launch_bounds__(STRIP_T, 4) k_strip(GraphDev g, RangeView v, const int32_t *__restrict__ tb, int32_t stripMax, int32_t *__restrict__ esc,
                                                        int32_t *__restrict__ escCtl, int32_t escCap, int *__restrict__ err) {
	__shared__ __attribute__((aligned(16))) uint32_t pool_[POOL_WORDS];
	__shared__ int32_t ctr_[C_TOTAL];
	l_u32 *pool = (l_u32 *)pool_;
	l_i32 *ctr = (l_i32 *)ctr_;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int32_t a = tb[blockIdx.x], b = tb[blockIdx.x + 1];
	if (a >= b) return;
	// BVGPU_STATS=1: clock ticks (100 MHz) per phase, summed over the strips: stats[32 + phase]; stats[32 + 15] = strips
	unsigned long long tPrev = g.stats ? wall_clock64() : 0;
	int tPhase = 0;
#define STRIP_TICK() do { if (g.stats) { if (tid == 0) { const unsigned long long tn_ = wall_clock64(); atomicAdd(&g.stats[32 + tPhase], tn_ - tPrev); tPrev = tn_; } tPhase++; } } while (0)
	const int32_t n = min(b - a, (int32_t)MAX_NODES); // (b - a <= MAX_NODES by construction of the bounds)
	if (tid < C_TOTAL) ctr[tid] = tid == C_IVLIM || tid == C_SEGLIM ? 0x7fffffff : 0;
	__syncthreads();

	// ---- the strip's records: outdegree, reference, position (registers until the layout is known)
	int32_t rd[RPT], rr[RPT];
	int64_t ro[RPT], rn[RPT];
	int32_t mine = 0;
#pragma unroll
	for (int k = 0; k < RPT; k++) {
		const int32_t i = tid * RPT + k;
		rd[k] = 0; rr[k] = 0; ro[k] = 0; rn[k] = 0;
		if (i < n) {
			const int32_t s = a + i;
			const int32_t d = v.outd[s];
			ro[k] = g.offsets[v.lo + s]; rn[k] = g.offsets[v.lo + s + 1];
			if (d > 0 && d < stripMax) { rd[k] = d; rr[k] = v.ref[s]; mine += d; }
		}
	}
	// exclusive scan of the outdegrees over the strip (threads hold consecutive records)
	int32_t inc = mine;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const int32_t t2 = __shfl_up(inc, o, 64); if (lane >= o) inc += t2; }
	if (lane == 63) ctr[C_WSUM + wave] = inc;
	__syncthreads();
	int32_t wbase = 0, narcs = 0;
#pragma unroll
	for (int w = 0; w < STRIP_T / 64; w++) { const int32_t t2 = ctr[C_WSUM + w]; if (w < wave) wbase += t2; narcs += t2; }
	int32_t rowOff = wbase + inc - mine;

	STRIP_TICK(); // 0: record loads + scan
	// ---- layout of the pool, staging of the stream slice
	const int64_t p0 = g.offsets[v.lo + a], p1 = g.offsets[v.lo + b];
	const uint64_t w0 = ((uint64_t)p0 >> 5) & ~(uint64_t)3;
	const int64_t base = (int64_t)(w0 << 5);
	const StripLayout L = strip_layout(n, narcs, ((p1 - base + 31) >> 5) + 8, g.minInt);
	StripL st;
	strip_bind(st, pool, L);
	l_u16 *const listB = st.listB;
	const uint32_t nw = (uint32_t)L.nw;
	const uint32_t qmax = (nw - 3) * 32;
	{
		const uint4 *src4 = (const uint4 *)(g.bits + w0);
		const uint64_t lim4 = (g.nwords + 8 - w0) / 4; // the image is followed by >= 8 zero words
		for (uint32_t i4 = (uint32_t)tid; i4 < nw / 4; i4 += STRIP_T) {
			const uint4 q4 = i4 < lim4 ? src4[i4] : uint4{ 0u, 0u, 0u, 0u };
			st.win[4 * i4 + 0] = __builtin_bswap32(q4.x); st.win[4 * i4 + 1] = __builtin_bswap32(q4.y);
			st.win[4 * i4 + 2] = __builtin_bswap32(q4.z); st.win[4 * i4 + 3] = __builtin_bswap32(q4.w);
		}
	}
	auto escape = [&](int32_t i) { // the record is decoded by the cooperative kernel after this one
		st.m_d[i] = 0;
		const int32_t k = atomicAdd(&escCtl[0], 1);
		if (k < escCap) esc[k] = a + i; else atomicOr(err, E_FORMAT);
	};
	// ---- record fields; records sorted by outdegree, longest first (counting sort on the bit length)
	int32_t bin[RPT], pos[RPT];
#pragma unroll
	for (int k = 0; k < RPT; k++) {
		const int32_t i = tid * RPT + k;
		bin[k] = -1;
		if (i < n) {
			int32_t d = rd[k];
			const int64_t q0 = ro[k] - base, q1 = rn[k] - base;
			st.m_ref[i] = (uint16_t)rr[k];
			st.m_off[i] = (uint16_t)rowOff;
			rowOff += d;
			if (d > 0 && (!L.ok || q1 > (int64_t)qmax || q1 - q0 > 0xffff || q1 <= q0)) { st.m_d[i] = (uint16_t)d; escape(i); d = 0; } // does not fit the staged slice
			st.m_d[i] = (uint16_t)d;
			st.m_bit[i] = d ? (uint32_t)q0 : 0u;
			st.m_sbits[i] = d ? (uint16_t)(q1 - q0) : 0; // until phase S: length of the record
			st.m_nres[i] = 0; st.m_niv[i] = 0; st.m_cop[i] = 0; st.m_seg0[i] = 0xffff;
			if (d) { bin[k] = (int)clz32((uint32_t)d) - 16; pos[k] = lds_add(&ctr[C_HIST + bin[k]], 1); } // d < 2^16: bins 0 (longest) .. 15
		}
	}
	// (records beyond MAX_NODES cannot exist; if the bounds were ever wrong they would be silently skipped: make it loud)
	if (tid == 0 && b - a > (int32_t)MAX_NODES) atomicOr(err, E_FORMAT);
	__syncthreads();
	if (tid < 64) { // exclusive scan of the 16 bins by one wave
		const int32_t c = tid < 16 ? ctr[C_HIST + tid] : 0;
		int32_t in2 = c;
#pragma unroll
		for (int o = 1; o < 16; o <<= 1) { const int32_t t2 = __shfl_up(in2, o, 64); if (tid >= o) in2 += t2; }
		if (tid < 16) ctr[C_HIST + tid] = in2 - c;
		if (tid == 15) ctr[C_HIST + 16] = in2; // records with work
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < RPT; k++) if (bin[k] >= 0) st.list[ctr[C_HIST + bin[k]] + pos[k]] = (uint16_t)(tid * RPT + k);
	__syncthreads();
	STRIP_TICK(); // 1: staging, fields, sort
	const int32_t nRec = ctr[C_HIST + 16];
	Job job;
	job.W = g.W; job.minInt = g.minInt; job.zk = (uint32_t)g.zetaK; job.stripMax = stripMax; job.x0 = v.lo + a;

	// ---- phase S: structure, one lane per record
	{
		auto drefOf = [&](int32_t i, int32_t r) -> int64_t { return a + i - r >= 0 ? (int64_t)v.outd[a + i - r] : -1; }; // (referents before the view: k_apply_need clears such references)
		auto ivAlloc = [&](int32_t cnt) -> int32_t { const int32_t o = lds_add(&ctr[C_NIV], cnt); if (o + cnt > st.ivCap) { lds_min(&ctr[C_IVLIM], o); return -1; } return o; };
		auto blkAlloc = [&](int32_t cnt) -> int32_t { const int32_t o = lds_add(&ctr[C_NBLK], cnt); return o + cnt > st.blkCap ? -1 : o; };
		for (;;) {
			const int32_t b0 = fetch64(&ctr[C_FETCH]);
			if (b0 >= nRec) break;
			const int32_t idx = b0 + lane;
			if (idx < nRec) {
				const int32_t i = (int32_t)st.list[idx];
				const uint32_t recEnd = st.m_bit[i] + (uint32_t)st.m_sbits[i];
				if (!phase_structure<ZK>(st, job, qmax, i, recEnd, drefOf, ivAlloc, blkAlloc, false)) escape(i);
			}
		}
	}
	__syncthreads();
	STRIP_TICK(); // 2: phase S
	// ---- segments of the residual sections (static: one lane per record)
	for (int32_t i = tid; i < n; i += STRIP_T) {
		const uint32_t nRes = st.m_nres[i];
		if (st.m_d[i] == 0 || nRes == 0) continue;
		const int32_t m = segments_of(nRes, st.m_sbits[i]);
		const int32_t e0 = lds_add(&ctr[C_NSEG], m);
		if (e0 + m > st.segCap) { lds_min(&ctr[C_SEGLIM], e0); escape(i); continue; }
		st.m_seg0[i] = (uint16_t)e0;
		if (m == 1) { st.seg[e0].start = st.m_bit[i]; st.seg[e0].end = 0; st.seg[e0].base = job.x0 + i; st.seg[e0].cnt = (uint16_t)nRes; st.seg[e0].rec = (uint16_t)i; }
		else {
			const int32_t la = lds_add(&ctr[C_NLSEG], m);
			for (int32_t k = 0; k < m; k++) { st.seg[e0 + k].end = (uint32_t)k; st.seg[e0 + k].rec = (uint16_t)i; st.seg[e0 + k].cnt = 0; listB[la + k] = (uint16_t)(e0 + k); }
			st.list[st.listLen - 1 - lds_add(&ctr[C_NLREC], 1)] = (uint16_t)i; // long sections: from the back of the (now free) record list
		}
	}
	__syncthreads();
	STRIP_TICK(); // 3: segment allocation
	const int32_t nSeg = min(ctr[C_NSEG], ctr[C_SEGLIM]);
	const int32_t nLongSeg = ctr[C_NLSEG], nLongRec = ctr[C_NLREC];
	const int32_t listEnd = st.listLen;
	// ---- phase A: anchors of the long sections, one lane per nominal segment
	for (int32_t t = tid; t < nLongSeg; t += STRIP_T) phase_anchor<ZK>(st, job, qmax, (int32_t)listB[t]);
	__syncthreads();
	STRIP_TICK(); // 4: phase A
	// ---- phase B: chain the segments of each long section
	for (int32_t t = tid; t < nLongRec; t += STRIP_T) {
		const int32_t i = (int32_t)st.list[listEnd - 1 - t];
		if (st.m_d[i] != 0 && !phase_chain<ZK>(st, job, qmax, i)) escape(i);
	}
	if (tid < 33) ctr[C_HIST + tid] = 0;
	__syncthreads();
	STRIP_TICK(); // 5: phase B
	// ---- segments sorted by length, longest first
	constexpr int SPT = 8; // a thread sorts up to SPT segments (segCap <= POOL: a few thousand)
	int32_t sbin[SPT], spos[SPT];
#pragma unroll
	for (int k = 0; k < SPT; k++) {
		const int32_t e = tid + k * STRIP_T;
		sbin[k] = -1;
		if (e < nSeg) {
			const int32_t i = (int32_t)st.seg[e].rec;
			const int32_t c = st.m_d[i] ? (int32_t)st.seg[e].cnt : 0;
			if (c > 0) { sbin[k] = 31 - min(c >> 2, 31); spos[k] = lds_add(&ctr[C_HIST + sbin[k]], 1); }
		}
	}
	__syncthreads();
	if (tid < 64) {
		const int32_t c = tid < 32 ? ctr[C_HIST + tid] : 0;
		int32_t in2 = c;
#pragma unroll
		for (int o = 1; o < 32; o <<= 1) { const int32_t t2 = __shfl_up(in2, o, 64); if (tid >= o) in2 += t2; }
		if (tid < 32) ctr[C_HIST + tid] = in2 - c;
		if (tid == 31) ctr[C_HIST + 32] = in2;
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < SPT; k++) if (sbin[k] >= 0) listB[ctr[C_HIST + sbin[k]] + spos[k]] = (uint16_t)(tid + k * STRIP_T);
	// (segments beyond SPT * STRIP_T cannot exist: segCap < 4096)
	__syncthreads();
	STRIP_TICK(); // 6: segment sort
	const int32_t nWork = ctr[C_HIST + 32];
	// ---- phase R: residuals, one lane per segment
	for (;;) {
		const int32_t b0 = fetch64(&ctr[C_FETCH2]);
		if (b0 >= nWork) break;
		const int32_t idx = b0 + lane;
		if (idx < nWork) {
			const int32_t e = (int32_t)listB[idx];
			if (!phase_residuals<ZK>(st, job, qmax, e)) { const int32_t i = (int32_t)st.seg[e].rec; if (st.m_d[i]) escape(i); }
		}
	}
	__syncthreads();
	STRIP_TICK(); // 7: phase R
	// ---- phase X: intervals, one lane each; the long ones by a wave each
	const int32_t nIv = min(ctr[C_NIV], ctr[C_IVLIM]);
	for (int32_t j = tid; j < nIv; j += STRIP_T) {
		if ((int32_t)st.iv_len[j] >= LONG_INTERVAL) { const int32_t k = lds_add(&ctr[C_NLIV], 1); st.list[k] = (uint16_t)j; }
		else phase_interval(st, j, 0, 1);
	}
	__syncthreads();
	{
		const int32_t nLongIv = ctr[C_NLIV];
		for (int32_t t = wave; t < nLongIv; t += STRIP_T / 64) phase_interval(st, (int32_t)st.list[t], lane, 64);
	}
	__syncthreads();
	STRIP_TICK(); // 8: phase X
	// ---- phase W: the rows leave for the CSR, 16 lanes per row
	for (int32_t i = tid >> 4; i < n; i += STRIP_T / 16) {
		const int32_t d = (int32_t)st.m_d[i];
		if (d == 0) continue;
		const int32_t s = a + i;
		if (!v.fits(s)) { if ((tid & 15) == 0) atomicOr(err, s >= v.nh ? E_CAP : E_HALO); continue; }
		int32_t *__restrict__ dst = v.row(s);
		const int32_t off = (int32_t)st.m_off[i];
		for (int32_t t = tid & 15; t < d; t += 16) dst[t] = st.rows[off + t];
	}
	if (g.stats) { __syncthreads(); STRIP_TICK(); if (tid == 0) { atomicAdd(&g.stats[32 + 15], 1ull); atomicAdd(&g.stats[32 + 14], (unsigned long long)n); atomicAdd(&g.stats[32 + 13], (unsigned long long)nSeg); atomicAdd(&g.stats[32 + 12], (unsigned long long)nLongSeg); atomicAdd(&g.stats[32 + 11], (unsigned long long)narcs); } } // 9: phase W
#undef STRIP_TICK
}

int32_t strip_count(int64_t bitSpan, int32_t cnt, int64_t arcsBound) {
	const long double w = (long double)bitSpan + (long double)NODE_W * cnt + (long double)ARC_W * (long double)arcsBound;
	const long double t = w / (long double)SPAN_W + 2;
	return (int32_t)(t < 0x7ffffff0 ? t : 0x7ffffff0);
}
void launch_strip_bounds(const GraphDev &g, const RangeView &v, int32_t ntiles, int32_t *tb, int32_t *escCtl, int *err, hipStream_t st) {
	hipLaunchKernelGGL(k_strip_bounds, dim3((unsigned)(((int64_t)ntiles + 1 + 255) / 256)), dim3(256), 0, st, g.offsets, v.rowstart, v.lo, v.cnt, ntiles, tb, escCtl, err);
}
void launch_strips(const GraphDev &g, int def, const RangeView &v, const int32_t *tb, int32_t ntiles, int32_t stripMax, int32_t *esc, int32_t *escCtl, int32_t escCap, int *err, hipStream_t st) {
	if (v.cnt <= 0 || ntiles <= 0) return;
	if (def == 1) hipLaunchKernelGGL(k_strip<3>, dim3(ntiles), dim3(STRIP_T), 0, st, g, v, tb, stripMax, esc, escCtl, escCap, err);
	else hipLaunchKernelGGL(k_strip<0>, dim3(ntiles), dim3(STRIP_T), 0, st, g, v, tb, stripMax, esc, escCtl, escCap, err);
}
int32_t strip_max_default() { return STRIP_MAX_DEFAULT; }

} // namespace bv
