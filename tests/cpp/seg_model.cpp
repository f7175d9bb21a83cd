// seg_model.cpp -- host-side model of the segment decoder (webgraph_amd/csrc/bv_seg.hip): the SAME phase bodies, decoders and LDS
// carve-up (bv_seg.hpp, compiled here for the CPU) driven lane after lane, record after record, in the order the wavefront runs them.
// Test infrastructure: tests/test_seg_model_cpu.py builds it with g++ and compares what it decodes with the CPU oracle, so that the
// logic of the kernel is checked in the `-m "not gpu"` suite before it ever runs on a GPU.  Not part of the product.
#include <cstdio>
#include <cstdlib>
static int g_why = 0;
#define BVS_WHY(k) (g_why = (k))
#include "../../webgraph_amd/csrc/bv_seg.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

using namespace bvs;
typedef StripT<uint32_t *, uint16_t *, int32_t *> SegH;

extern "C" {

// graph: the .graph bytes followed by >= 64 zero bytes (nbytes = file size).  View = nodes [lo, lo + cnt), no halo.
// outd / ref / rowstart as the kernels' RangeView holds them.  succ[rowstart[cnt]]: rows (only the records of the class are written).
// esc[cnt]: escaped slots, *nEsc their number.  cop[cnt]: ids copied from the referent (-1: not this decoder's work).
// stats[8]: records, max pool words used, segments, long-section segments, re-decoded segments, intervals, long intervals, -
int seg_model_run(const uint8_t *graph, uint64_t nbytes, const int64_t *offsets, int32_t lo, int32_t cnt, const int32_t *outd, const uint16_t *ref,
                  const int64_t *rowstart, int W, int minInt, int zk, int midMin, int midMax, int32_t *succ, int32_t *esc, int32_t *nEsc, int32_t *cop, int64_t *stats) {
	const uint64_t nwords = (nbytes + 3) / 4;
	const bool trace = getenv("SEG_MODEL_TRACE") != nullptr;
	*nEsc = 0;
	for (int s = 0; s < cnt; s++) cop[s] = -1;
	for (int k = 0; k < 8; k++) stats[k] = 0;
	std::vector<uint32_t> pool((size_t)WPOOL_WORDS + 64);
	Job job; job.W = W; job.minInt = minInt; job.zk = (uint32_t)zk;
	for (int32_t s = 0; s < cnt; s++) {
		const int32_t d = outd[s];
		if (d < midMin || d <= 0 || d >= midMax) continue;
		stats[0]++;
		const int32_t x = lo + s, r = ref[s];
		const int64_t dref = r > 0 ? (s - r >= 0 ? (int64_t)outd[s - r] : -1) : 0;
		int32_t *rows = succ + (rowstart[s] - rowstart[0]);
		// (the pool is NOT cleared between records: a wave inherits what the previous one left in LDS)
		const int64_t o0 = offsets[x], o1 = offsets[x + 1];
		const uint64_t w0 = ((uint64_t)o0 >> 5) & ~(uint64_t)3;
		const int64_t base = (int64_t)(w0 << 5);
		const StripLayout L = strip_layout(((o1 - base + 31) >> 5) + 8);
		if (L.oIv + 4 * L.ivCap > WPOOL_WORDS || L.oSeg + 5 * L.segCap > L.oIv || L.oWin + L.nw > L.oSeg) { fprintf(stderr, "seg_model: layout overflow\n"); return -2; }
		stats[1] = std::max<int64_t>(stats[1], L.oIv + 4 * L.ivCap);
		SegH st;
		strip_bind(st, pool.data(), L);
		const uint32_t nw = (uint32_t)L.nw, qmax = (nw - 3) * 32;
		const int64_t q0 = o0 - base, q1 = o1 - base;
		const char *stage = "window";
		auto escape = [&]() { if (trace) fprintf(stderr, "escape(why %d): slot %d at %s: d %d ref %d ivCap %d segCap %d nw %d\n", g_why, s, stage, d, r, st.ivCap, st.segCap, L.nw); esc[(*nEsc)++] = s; };
		if (q1 > (int64_t)qmax || q1 <= q0) { g_why = 100; escape(); continue; }
		for (uint32_t k = 0; k < nw; k++) {
			const uint64_t wi = w0 + k;
			uint32_t word = 0;
			if (wi < nwords + 8) { uint8_t bts[4] = { 0, 0, 0, 0 }; for (int q = 0; q < 4; q++) { const uint64_t bi = wi * 4 + q; bts[q] = bi < nbytes ? graph[bi] : 0; } word = ((uint32_t)bts[0] << 24) | ((uint32_t)bts[1] << 16) | ((uint32_t)bts[2] << 8) | bts[3]; }
			st.win[k] = word;
		}
		stage = "structure";
		Rec R = structure_head(st, job, qmax, (uint32_t)q0, d, r, dref);
		if (R.ok && R.nIv > st.ivCap) { R.ok = false; g_why = 101; }
		if (R.ok) structure_intervals(st, job, qmax, R, x, (uint32_t)R.copied, (uint32_t)q1);
		int32_t m = 0;
		if (R.ok) { m = segments_of(R.nRes, R.sbits); if (m > st.segCap) { R.ok = false; g_why = 102; } }
		if (!R.ok) { escape(); continue; }
		const uint32_t rowOut = (uint32_t)R.copied;
		stats[2] += m;
		if (m == 1) segment_short(st, 0, R, x, rowOut);
		else if (m > 1) {
			stats[3] += m;
			for (int32_t k = 0; k < m; k++) segment_nominal(st, k, R.q, R.q + R.sbits, k);
			for (int32_t e = 0; e < m; e++) { if (zk == 3) phase_anchor<3>(st, job, qmax, e); else phase_anchor<0>(st, job, qmax, e); }
			for (int32_t k = 1; k < m; k++) if (st.seg_start[k] != st.seg_out[k - 1]) stats[4]++;
			stage = "chain";
			const bool ok = zk == 3 ? phase_chain<3>(st, job, qmax, 0, m, R, x, rowOut) : phase_chain<0>(st, job, qmax, 0, m, R, x, rowOut);
			if (!ok) { g_why = 103; escape(); continue; }
		}
		stage = "residuals";
		bool badR = false;
		for (int32_t e = 0; e < m; e++) if (st.seg_cnt[e] != 0) { const bool ok = zk == 3 ? phase_residuals<3>(st, job, qmax, rows, e) : phase_residuals<0>(st, job, qmax, rows, e); if (!ok) badR = true; }
		if (badR) { g_why = 104; escape(); continue; }
		stats[5] += R.nIv;
		for (int32_t j = 0; j < R.nIv; j++) {
			const int32_t len = st.iv_len[j];
			if (len >= LONG_INTERVAL) { stats[6]++; for (int l = 0; l < 64; l++) phase_interval(st, rows, j, l, 64); }
			else if (len > 0) phase_interval(st, rows, j, 0, 1);
		}
		cop[s] = R.copied;
	}
	return 0;
}
}
