// bv_strip.hip -- the strip kernel: device side of bv_strip.hpp (gfx950).  One WAVEFRONT owns the records that start in one
// small slice of the stream (at most 64: a lane per record in the structure phase); see bv_strip.hpp for the phases and for
// why a wave never waits for another one.  What lives here: the strip bounds, staging of the slice, the wave-wide scans that
// hand out arena and segment slots, the loops over work items, and the escape list.
#include "bv_strip.hpp"
#include "bv_launch.hpp"

namespace bv {
using namespace bvs;

typedef __attribute__((address_space(3))) uint32_t l_u32; // LDS-qualified: accesses through these are ds_* instructions, never flat ones
typedef __attribute__((address_space(3))) uint16_t l_u16;
typedef __attribute__((address_space(3))) int32_t l_i32;
using StripL = StripT<l_u32 *, l_u16 *, l_i32 *>;

// LDS hand-off inside the wave: the LDS executes a wave's DS instructions in issue order, so keeping the program order is enough
__device__ __forceinline__ void wsync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ int32_t wave_excl_scan(int32_t v, int lane, int32_t &total) {
	int32_t inc = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const int32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
	total = __shfl(inc, 63, 64);
	return inc - v;
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

struct RowsMasked { int32_t *p; uint32_t mask; __device__ __forceinline__ int32_t &operator[](uint32_t i) const { return p[i & mask]; } }; // mask = ~0; BVGPU_DBG=512 (timing experiment, wrong results): every store lands in the strip's first 64 ids

template <int ZK>
__global__ void __launch_bounds__(64) k_strip(GraphDev g, RangeView v, const int32_t *__restrict__ tb, int32_t stripMin, int32_t stripMax, int32_t *__restrict__ esc,
                                              int32_t *__restrict__ escCtl, int32_t escCap, int *__restrict__ err) {
	__shared__ __attribute__((aligned(16))) uint32_t pool_[WPOOL_WORDS];
	__shared__ int32_t hist_[SORT_BINS + 1];
	l_u32 *pool = (l_u32 *)pool_;
	l_i32 *hist = (l_i32 *)hist_;
	const int lane = threadIdx.x;
	const int32_t a = tb[blockIdx.x], b = tb[blockIdx.x + 1];
	if (a >= b) return;
	const int32_t n = min(b - a, (int32_t)STRIP_NODES); // (b - a <= STRIP_NODES by construction of the bounds;
	if (b - a > (int32_t)STRIP_NODES && lane == 0) atomicOr(err, E_FORMAT); // if that were ever wrong, records would be skipped: make it loud)
	// BVGPU_STATS=1: clock ticks (100 MHz) per phase of one strip in 64, summed: stats[32 + phase]; stats[32 + 15] = strips sampled
	const bool tSample = g.stats && (blockIdx.x & 63) == 0; // (the atomics of every strip would be what is measured)
	unsigned long long tPrev = tSample ? wall_clock64() : 0;
	int tPhase = 0;
#define STRIP_TICK() do { if (tSample) { if (lane == 0) { const unsigned long long tn_ = wall_clock64(); atomicAdd(&g.stats[32 + tPhase], tn_ - tPrev); tPrev = tn_; } tPhase++; } } while (0)

	// ---- the strip: rows, stream slice, layout of the pool
	// rows: in the caller's buffer from slot nh on, in the halo scratch before; a strip that straddles nh leaves its halo records to the escape path
	const int64_t rowNh = v.rowstart[v.nh], rowA = v.rowstart[a];
	const bool inHalo = b <= v.nh;
	const int64_t rowFirst = inHalo ? rowA : (a >= v.nh ? rowA : rowNh);
	const RowsMasked rows{ inHalo ? v.halo + rowA : v.succ + (rowFirst - rowNh), (g.dbg & 512) ? 63u : 0xffffffffu };
	const int64_t p0 = g.offsets[v.lo + a], p1 = g.offsets[v.lo + a + n];
	const uint64_t w0 = ((uint64_t)p0 >> 5) & ~(uint64_t)3;
	const int64_t base = (int64_t)(w0 << 5);
	const StripLayout L = strip_layout(((p1 - base + 31) >> 5) + 8);
	StripL st;
	strip_bind(st, pool, L);
	const uint32_t nw = (uint32_t)L.nw;
	const uint32_t qmax = (nw - 3) * 32;
	{
		const uint4 *src4 = (const uint4 *)(g.bits + w0);
		const uint64_t lim4 = (g.nwords + 8 - w0) / 4; // the image is followed by >= 8 zero words
		for (uint32_t i4 = (uint32_t)lane; i4 < nw / 4; i4 += 64) {
			const uint4 q4 = i4 < lim4 ? src4[i4] : uint4{ 0u, 0u, 0u, 0u };
			st.win[4 * i4 + 0] = __builtin_bswap32(q4.x); st.win[4 * i4 + 1] = __builtin_bswap32(q4.y);
			st.win[4 * i4 + 2] = __builtin_bswap32(q4.z); st.win[4 * i4 + 3] = __builtin_bswap32(q4.w);
		}
	}
	Job job;
	job.W = g.W; job.minInt = g.minInt; job.zk = (uint32_t)g.zetaK;

	// ---- the strip's records: KREC per lane (record k * 64 + lane in pass k)
	Rec R[KREC];
	int32_t x[KREC], m[KREC], eFirst[KREC];
	uint32_t rowOut[KREC], recEnd[KREC];
	bool own[KREC], escNow[KREC], isLong[KREC];
	int32_t dk[KREC], rk[KREC];
	int64_t drefk[KREC];
	uint32_t q0k[KREC];
#pragma unroll
	for (int k = 0; k < KREC; k++) {
		const int32_t i = k * 64 + lane, s = a + i;
		const bool have = i < n;
		const int32_t d = have ? v.outd[s] : 0;
		const int64_t o0 = have ? g.offsets[v.lo + s] : 0, o1 = have ? g.offsets[v.lo + s + 1] : 0;
		const int64_t rs = have ? v.rowstart[s] : 0, rsn = have ? v.rowstart[s + 1] : 0;
		own[k] = d >= stripMin && d > 0 && d < stripMax; escNow[k] = false; isLong[k] = false; m[k] = 0; eFirst[k] = 0;
		const int32_t r = own[k] ? (int32_t)v.ref[s] : 0;
		drefk[k] = r > 0 ? (s - r >= 0 ? (int64_t)v.outd[s - r] : -1) : 0; // (referents before the view: k_apply_need clears such references)
		dk[k] = d; rk[k] = r; x[k] = v.lo + s;
		if (own[k]) {
			if (!inHalo && s < v.nh) { escNow[k] = true; own[k] = false; }
			else if (!(inHalo ? (uint64_t)rsn <= v.halo_cap : (uint64_t)(rsn - rowNh) <= v.succ_cap)) { atomicOr(err, inHalo ? E_HALO : E_CAP); own[k] = false; }
			else if (rs - rowFirst + d > 0x7fffffffll) { escNow[k] = true; own[k] = false; }
		}
		rowOut[k] = (uint32_t)(rs - rowFirst);
		const int64_t q0 = o0 - base, q1 = o1 - base;
		if (own[k] && (q1 > (int64_t)qmax || q1 <= q0)) { escNow[k] = true; own[k] = false; } // the record overhangs the staged slice
		q0k[k] = (uint32_t)q0; recEnd[k] = (uint32_t)q1;
		R[k].q = 0; R[k].sbits = 0; R[k].copied = 0; R[k].extra = 0; R[k].nIv = 0; R[k].ivb = 0; R[k].nRes = 0; R[k].ok = false;
	}
	wsync();
	STRIP_TICK(); // 0: loads, staging

	// ---- phase S: structure, KREC passes of one lane per record
	int32_t ivBase = 0; // arena slots handed out so far
#pragma unroll
	for (int k = 0; k < KREC; k++) {
		if (k * 64 >= n) break;
		if (own[k]) { R[k] = structure_head(st, job, qmax, q0k[k], dk[k], rk[k], drefk[k]); if (!R[k].ok) { escNow[k] = true; own[k] = false; R[k].nIv = 0; } }
		int32_t ivTotal;
		R[k].ivb = ivBase + wave_excl_scan(own[k] ? R[k].nIv : 0, lane, ivTotal);
		if (own[k] && R[k].ivb + R[k].nIv > st.ivCap) { escNow[k] = true; own[k] = false; R[k].nIv = 0; } // no room for its intervals (nothing of it is in the arena)
		const int32_t ivEnd = min(ivBase + ivTotal, st.ivCap);
		// (an arena slot that no lane fills -- the slice of a record that escapes -- may hold what an earlier strip left there: phase X skips length 0)
		for (int32_t j = ivBase + lane; j < ivEnd; j += 64) st.iv_len[j] = 0;
		wsync();
		ivBase = ivEnd;
		rowOut[k] += (uint32_t)R[k].copied;
		if (own[k]) {
			structure_intervals(st, job, qmax, R[k], x[k], rowOut[k], recEnd[k]);
			if (!R[k].ok) { escNow[k] = true; own[k] = false; }
		}
		if (!own[k]) for (int32_t j = 0; j < R[k].nIv; j++) st.iv_len[R[k].ivb + j] = 0; // (a record that escaped half-way through its intervals)
	}
	const int32_t nIvAll = ivBase;
	STRIP_TICK(); // 1: phase S
	// ---- segments of the residual sections: the short sections first (one each), then the nominal segments of the long ones
	int32_t nShort = 0;
#pragma unroll
	for (int k = 0; k < KREC; k++) {
		m[k] = own[k] ? segments_of(R[k].nRes, R[k].sbits) : 0;
		const unsigned long long sm = __ballot(m[k] == 1);
		if (m[k] == 1) { eFirst[k] = nShort + __popcll(sm & ((1ull << lane) - 1)); if (eFirst[k] < st.segCap) segment_short(st, eFirst[k], R[k], x[k], rowOut[k]); else { escNow[k] = true; own[k] = false; m[k] = 0; for (int32_t j = 0; j < R[k].nIv; j++) st.iv_len[R[k].ivb + j] = 0; } }
		nShort += __popcll(sm);
	}
	nShort = min(nShort, st.segCap);
	int32_t nSeg = nShort;
#pragma unroll
	for (int k = 0; k < KREC; k++) {
		int32_t longTotal;
		const int32_t exL = wave_excl_scan(m[k] > 1 ? m[k] : 0, lane, longTotal);
		if (m[k] > 1) eFirst[k] = nSeg + exL;
		isLong[k] = m[k] > 1;
		if (isLong[k] && eFirst[k] + m[k] > st.segCap) { escNow[k] = true; own[k] = false; isLong[k] = false; for (int32_t j = 0; j < R[k].nIv; j++) st.iv_len[R[k].ivb + j] = 0; } // no room for its segments
		unsigned long long lm = __ballot(isLong[k]);
		while (lm) {
			const int Ls = __ffsll((long long)lm) - 1;
			lm &= lm - 1;
			const int32_t e0 = __shfl(eFirst[k], Ls, 64), mL = __shfl(m[k], Ls, 64);
			const uint32_t r0 = (uint32_t)__shfl((int)R[k].q, Ls, 64), sb = (uint32_t)__shfl((int)R[k].sbits, Ls, 64);
			for (int32_t kk = lane; kk < mL; kk += 64) segment_nominal(st, e0 + kk, r0, r0 + sb, kk);
			nSeg = e0 + mL; // (the lanes that fit are a prefix of the long ones: their entries are contiguous)
		}
		if (__ballot(m[k] > 1 && !isLong[k])) break; // the table is full: the later passes' long sections do not fit either (their records escape below)
	}
#pragma unroll
	for (int k = 0; k < KREC; k++) if (m[k] > 1 && !isLong[k] && own[k]) { escNow[k] = true; own[k] = false; for (int32_t j = 0; j < R[k].nIv; j++) st.iv_len[R[k].ivb + j] = 0; }
	wsync();
	// ---- phase A: anchors of the long sections, one lane per nominal segment
	for (int32_t e = nShort + lane; e < nSeg; e += 64) phase_anchor<ZK>(st, job, qmax, e);
	wsync();
	STRIP_TICK(); // 2: segments + phase A
	// ---- phase B: the record's lane chains its segments
#pragma unroll
	for (int k = 0; k < KREC; k++) {
		if (isLong[k] && !phase_chain<ZK>(st, job, qmax, eFirst[k], m[k], R[k], x[k], rowOut[k])) {
			escNow[k] = true; own[k] = false;
			for (int32_t kk = 0; kk < m[k]; kk++) st.seg_cnt[eFirst[k] + kk] = 0;
			for (int32_t j = 0; j < R[k].nIv; j++) st.iv_len[R[k].ivb + j] = 0;
		}
	}
	wsync();
	STRIP_TICK(); // 3: phase B
	// ---- the segments sorted by length, longest first (counting sort, bins of 4 codewords): the lanes of a round of phase R get segments of about the same length
	if (lane <= SORT_BINS) hist[lane] = 0;
	wsync();
	for (int32_t e0 = 0; e0 < nSeg; e0 += 64) {
		const int32_t e = e0 + lane;
		if (e < nSeg) { const int32_t c = (int32_t)st.seg_cnt[e]; __hip_atomic_fetch_add(&hist[c ? SORT_BINS - 1 - min(c >> 2, SORT_BINS - 1) : SORT_BINS], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
	}
	wsync();
	{
		const int32_t c = lane <= SORT_BINS ? hist[lane] : 0;
		int32_t tot;
		const int32_t ex = wave_excl_scan(c, lane, tot);
		wsync();
		if (lane <= SORT_BINS) hist[lane] = ex;
	}
	wsync();
	const int32_t nWork = hist[SORT_BINS]; // segments with codewords (the empty ones -- escaped records -- sort last)
	for (int32_t e0 = 0; e0 < nSeg; e0 += 64) {
		const int32_t e = e0 + lane;
		if (e < nSeg) { const int32_t c = (int32_t)st.seg_cnt[e]; const int32_t at = __hip_atomic_fetch_add(&hist[c ? SORT_BINS - 1 - min(c >> 2, SORT_BINS - 1) : SORT_BINS], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); st.order[at] = (uint16_t)e; }
	}
	wsync();
	// ---- phase R: residuals, one lane per segment, stored straight into the rows
	bool badR = false;
	for (int32_t t = lane; t < nWork; t += 64) { if (!phase_residuals<ZK>(st, job, qmax, rows, (int32_t)st.order[t])) badR = true; }
	wsync();
	STRIP_TICK(); // 4: sort + phase R
	if (__any(badR)) { // (a codeword the decoders reject: malformed -- every record of the strip is decoded again by the escape path, which reports it)
#pragma unroll
		for (int k = 0; k < KREC; k++) if (own[k]) { escNow[k] = true; own[k] = false; }
	}
	else {
		// ---- phase X: intervals, one lane each; the long ones by the whole wave
		for (int32_t j0 = 0; j0 < nIvAll; j0 += 64) {
			const int32_t j = j0 + lane;
			const int32_t len = j < nIvAll ? (int32_t)st.iv_len[j] : 0;
			if (len > 0 && len < LONG_INTERVAL) phase_interval(st, rows, j, 0, 1);
			unsigned long long lm = __ballot(len >= LONG_INTERVAL);
			while (lm) { const int Ls = __ffsll((long long)lm) - 1; lm &= lm - 1; phase_interval(st, rows, j0 + Ls, lane, 64); }
		}
	}
	STRIP_TICK(); // 5: phase X
	// ---- the records this strip leaves to the cooperative kernel
#pragma unroll
	for (int k = 0; k < KREC; k++) {
		const unsigned long long em = __ballot(escNow[k]);
		if (em) {
			int32_t k0 = 0;
			if (lane == 0) k0 = atomicAdd(&escCtl[0], __popcll(em));
			k0 = __shfl(k0, 0, 64);
			if (escNow[k]) { const int32_t at = k0 + __popcll(em & ((1ull << lane) - 1)); if (at < escCap) esc[at] = a + k * 64 + lane; else atomicOr(err, E_FORMAT); }
		}
	}
	if (tSample && lane == 0) { atomicAdd(&g.stats[32 + 15], 1ull); atomicAdd(&g.stats[32 + 14], (unsigned long long)n); atomicAdd(&g.stats[32 + 13], (unsigned long long)nSeg); atomicAdd(&g.stats[32 + 12], (unsigned long long)(nSeg - nShort)); atomicAdd(&g.stats[32 + 11], (unsigned long long)nIvAll); }
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
void launch_strips(const GraphDev &g, int def, const RangeView &v, const int32_t *tb, int32_t ntiles, int32_t stripMin, int32_t stripMax, int32_t *esc, int32_t *escCtl, int32_t escCap, int *err, hipStream_t st) {
	if (v.cnt <= 0 || ntiles <= 0) return;
	if (def == 1) hipLaunchKernelGGL(k_strip<3>, dim3(ntiles), dim3(64), 0, st, g, v, tb, stripMin, stripMax, esc, escCtl, escCap, err);
	else hipLaunchKernelGGL(k_strip<0>, dim3(ntiles), dim3(64), 0, st, g, v, tb, stripMin, stripMax, esc, escCtl, escCap, err);
}
int32_t strip_max_default() { return STRIP_MAX_DEFAULT; }

} // namespace bv
