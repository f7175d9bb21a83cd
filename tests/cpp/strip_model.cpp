// strip_model.cpp -- host-side model of the strip kernel (webgraph_amd/csrc/bv_strip.hip): the SAME phase bodies, decoders
// and LDS carve-up (bv_strip.hpp, compiled here for the CPU) driven lane after lane, strip after strip, in the order the
// wavefront runs them.  Test infrastructure: tests/test_strip_model_cpu.py builds it with g++ and compares what it decodes
// with the CPU oracle, so that the logic of the kernel is checked in the `-m "not gpu"` suite before it ever runs on a GPU.
// Not part of the product.
#include <cstdio>
#include <cstdlib>
static int g_why = 0;
#define BVS_WHY(k) (g_why = (k))
#include "../../webgraph_amd/csrc/bv_strip.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

using namespace bvs;
typedef StripT<uint32_t *, uint16_t *, int32_t *> StripH;

extern "C" {

// graph: the .graph bytes followed by >= 64 zero bytes (nbytes = file size).  View = nodes [lo, lo + cnt), no halo.
// outd / ref / rowstart as the kernels' RangeView holds them.  succ[rowstart[cnt]]: rows (only the strip work is written).
// esc[cnt]: escaped slots, *nEsc their number.  cop[cnt]: ids copied from the referent (-1: not strip work).
// stats[8]: strips, max pool words used, segments, long segments, re-decoded segments, intervals, long intervals, records
int strip_model_run(const uint8_t *graph, uint64_t nbytes, const int64_t *offsets, int32_t lo, int32_t cnt, const int32_t *outd, const uint16_t *ref,
                    const int64_t *rowstart, int W, int minInt, int zk, int stripMax, int32_t *succ, int32_t *esc, int32_t *nEsc, int32_t *cop, int64_t *stats) {
	const uint64_t nwords = (nbytes + 3) / 4;
	const bool trace = getenv("STRIP_MODEL_TRACE") != nullptr;
	*nEsc = 0;
	for (int s = 0; s < cnt; s++) cop[s] = -1;
	for (int k = 0; k < 8; k++) stats[k] = 0;
	auto weight = [&](int32_t s) { return (offsets[lo + s] - offsets[lo]) + (int64_t)NODE_W * s + (int64_t)ARC_W * (rowstart[s] - rowstart[0]); };
	const int64_t ntiles = weight(cnt) / SPAN_W + 2;
	std::vector<uint32_t> pool((size_t)WPOOL_WORDS + 64);
	Job job; job.W = W; job.minInt = minInt; job.zk = (uint32_t)zk;
	for (int64_t t = 0; t < ntiles; t++) {
		auto bound = [&](int64_t tt) { const int64_t target = tt * SPAN_W; int32_t a = 0, b = cnt; while (a < b) { const int32_t mid = (int32_t)(((int64_t)a + b) >> 1); if (weight(mid) < target) a = mid + 1; else b = mid; } return a; };
		const int32_t a = bound(t), b = bound(t + 1);
		if (a >= b) continue;
		if (b - a > STRIP_NODES) { fprintf(stderr, "strip_model: %d nodes in a strip\n", b - a); return -1; }
		const int32_t n = b - a;
		stats[0]++;
		// (the pool is NOT cleared between strips: a wave inherits what the previous one left in LDS)
		const int64_t p0 = offsets[lo + a], p1 = offsets[lo + b];
		const uint64_t w0 = ((uint64_t)p0 >> 5) & ~(uint64_t)3;
		const int64_t base = (int64_t)(w0 << 5);
		const StripLayout L = strip_layout(((p1 - base + 31) >> 5) + 8);
		if (L.oIv + 4 * L.ivCap > WPOOL_WORDS || L.oSeg + 5 * L.segCap > L.oOrd || L.oOrd + (L.segCap + 1) / 2 > L.oIv || L.oWin + L.nw > L.oSeg) { fprintf(stderr, "strip_model: layout overflow\n"); return -2; }
		stats[1] = std::max<int64_t>(stats[1], L.oIv + 4 * L.ivCap);
		StripH st;
		strip_bind(st, pool.data(), L);
		const uint32_t nw = (uint32_t)L.nw, qmax = (nw - 3) * 32;
		for (uint32_t k = 0; k < nw; k++) {
			const uint64_t wi = w0 + k;
			uint32_t word = 0;
			if (wi < nwords + 8) { uint8_t bts[4] = { 0, 0, 0, 0 }; for (int q = 0; q < 4; q++) { const uint64_t bi = wi * 4 + q; bts[q] = bi < nbytes ? graph[bi] : 0; } word = ((uint32_t)bts[0] << 24) | ((uint32_t)bts[1] << 16) | ((uint32_t)bts[2] << 8) | bts[3]; }
			st.win[k] = word;
		}
		int32_t *rows = succ + (rowstart[a] - rowstart[0]);
		// per-record state: record i = k * 64 + lane lives in lane (i & 63), pass (i >> 6)
		struct Lane { bool own = false, esc = false, isLong = false; int32_t d = 0, r = 0, x = 0, m = 0, eFirst = 0; int64_t dref = 0, q0 = 0, q1 = 0; uint32_t rowOut = 0; Rec R{}; };
		std::vector<Lane> ln(64 * KREC);
		const char *stage = "fields";
		auto escape = [&](int l) { if (trace) fprintf(stderr, "escape(why %d): strip %lld slot %d (record %d of %d) at %s: d %d ref %d nres %d niv %d ivCap %d segCap %d nw %d\n", g_why, (long long)t, a + l, l, n, stage, ln[l].d, ln[l].r, ln[l].R.nRes, ln[l].R.nIv, st.ivCap, st.segCap, L.nw); ln[l].own = false; ln[l].esc = true; };
		for (int l = 0; l < n; l++) {
			Lane &z = ln[l];
			const int32_t s = a + l;
			z.d = outd[s]; z.x = lo + s;
			z.own = z.d > 0 && z.d < stripMax;
			z.r = z.own ? ref[s] : 0;
			z.dref = z.r > 0 ? (s - z.r >= 0 ? (int64_t)outd[s - z.r] : -1) : 0;
			z.rowOut = (uint32_t)(rowstart[s] - rowstart[a]);
			z.q0 = offsets[lo + s] - base; z.q1 = offsets[lo + s + 1] - base;
			if (z.own) stats[7]++;
			if (z.own && (z.q1 > (int64_t)qmax || z.q1 <= z.q0)) { g_why = 100; escape(l); }
		}
		int32_t ivBase = 0;
		for (int k = 0; k < KREC && k * 64 < n; k++) {
			stage = "head";
			for (int l = k * 64; l < k * 64 + 64; l++) { Lane &z = ln[l]; if (z.own) { z.R = structure_head(st, job, qmax, (uint32_t)z.q0, z.d, z.r, z.dref); if (!z.R.ok) { escape(l); z.R.nIv = 0; } } }
			int32_t ivTotal = 0;
			for (int l = k * 64; l < k * 64 + 64; l++) { Lane &z = ln[l]; z.R.ivb = ivBase + ivTotal; ivTotal += z.own ? z.R.nIv : 0; }
			for (int l = k * 64; l < k * 64 + 64; l++) { Lane &z = ln[l]; if (z.own && z.R.ivb + z.R.nIv > st.ivCap) { g_why = 101; escape(l); z.R.nIv = 0; } }
			const int32_t ivEnd = std::min(ivBase + ivTotal, st.ivCap);
			for (int32_t j = ivBase; j < ivEnd; j++) st.iv_len[j] = 0;
			ivBase = ivEnd;
			stage = "intervals";
			for (int l = k * 64; l < k * 64 + 64; l++) {
				Lane &z = ln[l];
				z.rowOut += (uint32_t)z.R.copied;
				if (z.own) { structure_intervals(st, job, qmax, z.R, z.x, z.rowOut, (uint32_t)z.q1); if (!z.R.ok) escape(l); }
				if (!z.own) for (int32_t j = 0; j < z.R.nIv; j++) st.iv_len[z.R.ivb + j] = 0;
			}
		}
		const int32_t nIvAll = ivBase;
		stage = "segments";
		int32_t nShort = 0;
		for (int l = 0; l < 64 * KREC; l++) {
			Lane &z = ln[l];
			z.m = z.own ? segments_of(z.R.nRes, z.R.sbits) : 0;
			if (z.m == 1) {
				z.eFirst = nShort++;
				if (z.eFirst < st.segCap) segment_short(st, z.eFirst, z.R, z.x, z.rowOut);
				else { g_why = 105; escape(l); z.m = 0; for (int32_t j = 0; j < z.R.nIv; j++) st.iv_len[z.R.ivb + j] = 0; }
			}
		}
		nShort = std::min(nShort, st.segCap);
		int32_t nSeg = nShort;
		bool full = false;
		for (int k = 0; k < KREC && !full; k++) {
			int32_t run = nSeg;
			for (int l = k * 64; l < k * 64 + 64; l++) { Lane &z = ln[l]; if (z.m > 1) { z.eFirst = run; run += z.m; } }
			for (int l = k * 64; l < k * 64 + 64; l++) {
				Lane &z = ln[l];
				z.isLong = z.m > 1;
				if (z.isLong && z.eFirst + z.m > st.segCap) { g_why = 102; escape(l); z.isLong = false; full = true; for (int32_t j = 0; j < z.R.nIv; j++) st.iv_len[z.R.ivb + j] = 0; }
				if (z.isLong) { for (int32_t kk = 0; kk < z.m; kk++) segment_nominal(st, z.eFirst + kk, z.R.q, z.R.q + z.R.sbits, kk); nSeg = z.eFirst + z.m; }
			}
		}
		for (int l = 0; l < 64 * KREC; l++) { Lane &z = ln[l]; if (z.m > 1 && !z.isLong && z.own) { g_why = 106; escape(l); for (int32_t j = 0; j < z.R.nIv; j++) st.iv_len[z.R.ivb + j] = 0; } }
		stats[2] += nSeg; stats[3] += nSeg - nShort;
		for (int32_t e = nShort; e < nSeg; e++) { if (zk == 3) phase_anchor<3>(st, job, qmax, e); else phase_anchor<0>(st, job, qmax, e); }
		stage = "chain";
		for (int l = 0; l < 64 * KREC; l++) {
			Lane &z = ln[l];
			if (!z.isLong) continue;
			for (int32_t kk = 1; kk < z.m; kk++) if (st.seg_start[z.eFirst + kk] != st.seg_out[z.eFirst + kk - 1]) stats[4]++;
			const bool ok = zk == 3 ? phase_chain<3>(st, job, qmax, z.eFirst, z.m, z.R, z.x, z.rowOut) : phase_chain<0>(st, job, qmax, z.eFirst, z.m, z.R, z.x, z.rowOut);
			if (!ok) { g_why = 103; escape(l); for (int32_t kk = 0; kk < z.m; kk++) st.seg_cnt[z.eFirst + kk] = 0; for (int32_t j = 0; j < z.R.nIv; j++) st.iv_len[z.R.ivb + j] = 0; }
		}
		stage = "residuals";
		// the segments in the order of phase R: counting sort on bins of 4 codewords, longest first, the empty ones last
		int32_t histo[SORT_BINS + 1] = { 0 };
		auto binOf = [&](int32_t c) { return c ? SORT_BINS - 1 - std::min(c >> 2, SORT_BINS - 1) : SORT_BINS; };
		for (int32_t e = 0; e < nSeg; e++) histo[binOf(st.seg_cnt[e])]++;
		{ int32_t acc = 0; for (int bI = 0; bI <= SORT_BINS; bI++) { const int32_t c = histo[bI]; histo[bI] = acc; acc += c; } }
		const int32_t nWork = histo[SORT_BINS];
		for (int32_t e = 0; e < nSeg; e++) st.order[histo[binOf(st.seg_cnt[e])]++] = (uint16_t)e;
		bool badR = false;
		for (int32_t tt = 0; tt < nWork; tt++) { const int32_t e = st.order[tt]; const bool ok = zk == 3 ? phase_residuals<3>(st, job, qmax, rows, e) : phase_residuals<0>(st, job, qmax, rows, e); if (!ok) badR = true; }
		if (badR) { g_why = 104; for (int l = 0; l < n; l++) if (ln[l].own) escape(l); }
		else {
			stats[5] += nIvAll;
			for (int32_t j = 0; j < nIvAll; j++) {
				const int32_t len = st.iv_len[j];
				if (len >= LONG_INTERVAL) { stats[6]++; for (int l = 0; l < 64; l++) phase_interval(st, rows, j, l, 64); }
				else if (len > 0) phase_interval(st, rows, j, 0, 1);
			}
		}
		for (int l = 0; l < n; l++) { if (ln[l].esc) esc[(*nEsc)++] = a + l; else if (ln[l].own) cop[a + l] = ln[l].R.copied; }
	}
	return 0;
}
}
