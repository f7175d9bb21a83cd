// bv_seg.hip -- the segment decoder: device side of bv_seg.hpp (gfx950).  One WAVEFRONT per record of the middle class, taken from
// the list that k_classify builds for the cooperative decoders (longest first); see bv_seg.hpp for the phases.  What lives here: staging
// of the record's bits, the loops over segments and intervals, and the escape list.
#include "bv_seg.hpp"
#include "bv_launch.hpp"

namespace bv {
using namespace bvs;

typedef __attribute__((address_space(3))) uint32_t l_u32; // LDS-qualified: accesses through these are ds_* instructions, never flat ones
typedef __attribute__((address_space(3))) uint16_t l_u16;
typedef __attribute__((address_space(3))) int32_t l_i32;
using SegL = StripT<l_u32 *, l_u16 *, l_i32 *>;

// LDS hand-off inside the wave: the LDS executes a wave's DS instructions in issue order, so keeping the program order is enough
__device__ __forceinline__ void wsync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }

// list / count: the records with coopMin <= successors < giantMin (k_classify); this kernel takes those with fewer than midMax, block b the b-th of the list
template <int ZK>
__global__ void __launch_bounds__(64) k_mid(GraphDev g, RangeView v, const int32_t *__restrict__ list, const int32_t *__restrict__ count, int32_t midMax,
                                            int32_t *__restrict__ esc, int32_t *__restrict__ escCtl, int32_t escCap, int *__restrict__ err) {
	__shared__ __attribute__((aligned(16))) uint32_t pool_[WPOOL_WORDS];
	l_u32 *pool = (l_u32 *)pool_;
	const int lane = threadIdx.x;
	if ((int32_t)blockIdx.x >= *count) return;
	const int32_t s = list[blockIdx.x];
	const int32_t d = v.outd[s];
	if (d <= 0 || d >= midMax) return; // (the longer ones: k_parse_big<1> over the same list)
	const int32_t x = v.lo + s;
	const int32_t r = (int32_t)v.ref[s];
	const int64_t dref = r > 0 ? (s - r >= 0 ? (int64_t)v.outd[s - r] : -1) : 0;
	if (!v.fits(s)) { if (lane == 0) atomicOr(err, s >= v.nh ? E_CAP : E_HALO); return; }
	int32_t *const rows = v.row(s);
	bool escape = false;

	// ---- the record's bits -> LDS
	const int64_t o0 = g.offsets[x], o1 = g.offsets[x + 1];
	const uint64_t w0 = ((uint64_t)o0 >> 5) & ~(uint64_t)3;
	const int64_t base = (int64_t)(w0 << 5);
	const int64_t nwWant = ((o1 - base + 31) >> 5) + 8;
	const StripLayout L = strip_layout(nwWant);
	SegL st;
	strip_bind(st, pool, L);
	const uint32_t nw = (uint32_t)L.nw;
	const uint32_t qmax = (nw - 3) * 32;
	const int64_t q0 = o0 - base, q1 = o1 - base;
	if (q1 > (int64_t)qmax || q1 <= q0) escape = true; // longer than the pool allows
	Rec R; R.q = 0; R.sbits = 0; R.copied = 0; R.extra = 0; R.nIv = 0; R.ivb = 0; R.nRes = 0; R.ok = false;
	int32_t m = 0;
	if (!escape) {
		const uint4 *src4 = (const uint4 *)(g.bits + w0);
		const uint64_t lim4 = (g.nwords + 8 - w0) / 4; // the image is followed by >= 8 zero words
		for (uint32_t i4 = (uint32_t)lane; i4 < nw / 4; i4 += 64) {
			const uint4 q4 = i4 < lim4 ? src4[i4] : uint4{ 0u, 0u, 0u, 0u };
			st.win[4 * i4 + 0] = __builtin_bswap32(q4.x); st.win[4 * i4 + 1] = __builtin_bswap32(q4.y);
			st.win[4 * i4 + 2] = __builtin_bswap32(q4.z); st.win[4 * i4 + 3] = __builtin_bswap32(q4.w);
		}
		wsync();
		Job job;
		job.W = g.W; job.minInt = g.minInt; job.zk = (uint32_t)g.zetaK;
		// ---- phase S: the front of the record, lane 0
		if (lane == 0) {
			R = structure_head(st, job, qmax, (uint32_t)q0, d, r, dref);
			if (R.ok && R.nIv > st.ivCap) R.ok = false; // no room for its intervals
			if (R.ok) structure_intervals(st, job, qmax, R, x, (uint32_t)R.copied, (uint32_t)q1);
			if (R.ok) { m = segments_of(R.nRes, R.sbits); if (m > st.segCap) R.ok = false; }
		}
		// (everything the other lanes need of the record)
		R.ok = __shfl((int)R.ok, 0, 64) != 0;
		R.q = (uint32_t)__shfl((int)R.q, 0, 64); R.sbits = (uint32_t)__shfl((int)R.sbits, 0, 64);
		R.copied = __shfl(R.copied, 0, 64); R.nIv = __shfl(R.nIv, 0, 64); R.nRes = __shfl(R.nRes, 0, 64);
		m = __shfl(m, 0, 64);
		if (!R.ok) escape = true;
		wsync();
		if (!escape) {
			const uint32_t rowOut = (uint32_t)R.copied;
			if (m == 1) { if (lane == 0) segment_short(st, 0, R, x, rowOut); }
			else if (m > 1) {
				for (int32_t k = lane; k < m; k += 64) segment_nominal(st, k, R.q, R.q + R.sbits, k);
				wsync();
				// ---- phase A: anchors, one lane per nominal segment
				for (int32_t e = lane; e < m; e += 64) phase_anchor<ZK>(st, job, qmax, e);
				wsync();
				// ---- phase B: lane 0 chains the segments
				bool okB = true;
				if (lane == 0) okB = phase_chain<ZK>(st, job, qmax, 0, m, R, x, rowOut);
				if (!__shfl((int)okB, 0, 64)) escape = true;
			}
			wsync();
			if (!escape) {
				// ---- phase R: residuals, one lane per segment, stored straight into the row
				bool badR = false;
				for (int32_t e = lane; e < m; e += 64) if (st.seg_cnt[e] != 0 && !phase_residuals<ZK>(st, job, qmax, rows, e)) badR = true;
				wsync();
				if (__any(badR)) escape = true; // (a codeword the decoders reject: the cooperative kernel decodes the record again and reports it)
				else {
					// ---- phase X: intervals, one lane each; the long ones by the whole wave
					for (int32_t j0 = 0; j0 < R.nIv; j0 += 64) {
						const int32_t j = j0 + lane;
						const int32_t len = j < R.nIv ? (int32_t)st.iv_len[j] : 0;
						if (len > 0 && len < LONG_INTERVAL) phase_interval(st, rows, j, 0, 1);
						unsigned long long lm = __ballot(len >= LONG_INTERVAL);
						while (lm) { const int Ls = __ffsll((long long)lm) - 1; lm &= lm - 1; phase_interval(st, rows, j0 + Ls, lane, 64); }
					}
				}
			}
		}
	}
	if (escape && lane == 0) { const int32_t at = atomicAdd(&escCtl[0], 1); if (at < escCap) esc[at] = s; else atomicOr(err, E_FORMAT); }
}

// the escape list's counters (count, queue head) are zeroed in front of the kernel
__global__ void k_mid_reset(int32_t *__restrict__ escCtl) { escCtl[0] = 0; escCtl[2] = 0; }

void launch_mid(const GraphDev &g, int def, const RangeView &v, const int32_t *list, const int32_t *count, int32_t listCap, int32_t midMax, int32_t *esc, int32_t *escCtl, int32_t escCap, int *err, hipStream_t st) {
	if (v.cnt <= 0 || listCap <= 0) return;
	hipLaunchKernelGGL(k_mid_reset, dim3(1), dim3(1), 0, st, escCtl);
	if (def == 1) hipLaunchKernelGGL(k_mid<3>, dim3(listCap), dim3(64), 0, st, g, v, list, count, midMax, esc, escCtl, escCap, err);
	else hipLaunchKernelGGL(k_mid<0>, dim3(listCap), dim3(64), 0, st, g, v, list, count, midMax, esc, escCtl, escCap, err);
}
int32_t mid_min_default() { return MID_MIN_DEFAULT; }
int32_t mid_max_default() { return MID_MAX_DEFAULT; }

} // namespace bv
