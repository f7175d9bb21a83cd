// strip_model.cpp -- host-side model of the strip kernel (webgraph_amd/csrc/bv_strip.hip): the SAME phase bodies, decoders
// and LDS carve-up (bv_strip.hpp, compiled here for the CPU) driven lane after lane, strip after strip.  Test
// infrastructure: tests/test_strip_model_cpu.py builds it with g++ and compares what it decodes with the CPU oracle, so
// that the logic of the kernel is checked in the `-m "not gpu"` suite before it ever runs on a GPU.  Not part of the product.
static int g_why = 0;
#define BVS_WHY(k) (g_why = (k))
#include "../../webgraph_amd/csrc/bv_strip.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace bvs;
typedef StripT<uint32_t *, uint16_t *, int32_t *, Seg *> StripH;

extern "C" {

// graph: the .graph bytes followed by >= 64 zero bytes (nbytes = file size).  View = nodes [lo, lo + cnt), no halo.
// outd / ref / rowstart as the kernels' RangeView holds them.  succ[rowstart[cnt]]: rows (only the strip work is written).
// esc[cnt]: escaped slots, *nEsc their number.  cop[cnt]: ids copied from the referent (-1: not strip work).
// stats[8]: strips, max pool words used, segments, long segments, re-decoded segments, intervals, long intervals, records
int strip_model_run(const uint8_t *graph, uint64_t nbytes, const int64_t *offsets, int32_t lo, int32_t cnt, const int32_t *outd, const uint16_t *ref,
                    const int64_t *rowstart, int W, int minInt, int zk, int stripMax, int32_t *succ, int32_t *esc, int32_t *nEsc, int32_t *cop, int64_t *stats) {
	const uint64_t nwords = (nbytes + 3) / 4;
	*nEsc = 0;
	for (int s = 0; s < cnt; s++) cop[s] = -1;
	for (int k = 0; k < 8; k++) stats[k] = 0;
	auto weight = [&](int32_t s) { return (offsets[lo + s] - offsets[lo]) + (int64_t)NODE_W * s + (int64_t)ARC_W * (rowstart[s] - rowstart[0]); };
	const int64_t ntiles = weight(cnt) / SPAN_W + 2;
	std::vector<uint32_t> pool((size_t)POOL_WORDS + 64);
	for (int64_t t = 0; t < ntiles; t++) {
		auto bound = [&](int64_t tt) { const int64_t target = tt * SPAN_W; int32_t a = 0, b = cnt; while (a < b) { const int32_t mid = (int32_t)(((int64_t)a + b) >> 1); if (weight(mid) < target) a = mid + 1; else b = mid; } return a; };
		const int32_t a = bound(t), b = bound(t + 1);
		if (a >= b) continue;
		if (b - a > MAX_NODES) { fprintf(stderr, "strip_model: %d nodes in a strip\n", b - a); return -1; }
		const int32_t n = b - a;
		stats[0]++;
		std::fill(pool.begin(), pool.end(), 0xdeadbeefu);
		int32_t narcs = 0;
		std::vector<int32_t> rd(n), rr(n), roff(n);
		for (int i = 0; i < n; i++) {
			const int32_t d = outd[a + i];
			rd[i] = (d > 0 && d < stripMax) ? d : 0;
			rr[i] = rd[i] ? ref[a + i] : 0;
			roff[i] = narcs;
			narcs += rd[i];
		}
		const int64_t p0 = offsets[lo + a], p1 = offsets[lo + b];
		const uint64_t w0 = ((uint64_t)p0 >> 5) & ~(uint64_t)3;
		const int64_t base = (int64_t)(w0 << 5);
		const StripLayout L = strip_layout(n, narcs, ((p1 - base + 31) >> 5) + 8, minInt);
		{ // every region inside the pool?
			const int ends[] = { L.oRows + narcs, L.oBit + n, L.oF16 + 11 * L.f16Stride, L.oWin + L.nw, L.oBlk + L.blkCap / 2, L.oSeg + 4 * L.segCap, L.oList + (L.listLen + 1) / 2, L.oListB + (L.listLen + 1) / 2, L.oIv + 3 * L.ivCap };
			for (int e : ends) { if (L.ok && e > POOL_WORDS) { fprintf(stderr, "strip_model: layout overflow %d (n %d narcs %d nw %d)\n", e, n, narcs, L.nw); return -2; } stats[1] = std::max<int64_t>(stats[1], e); }
			if (L.ok && !(L.oRows + narcs <= L.oBit && L.oBit + n <= L.oF16 && L.oF16 + 11 * L.f16Stride <= L.oWin && L.oWin + L.nw <= L.oBlk && L.oBlk + L.blkCap / 2 <= L.oSeg &&
			              L.oSeg + 4 * L.segCap <= L.oList && L.oList + (L.listLen + 1) / 2 <= L.oListB && L.oListB + (L.listLen + 1) / 2 <= L.oIv)) { fprintf(stderr, "strip_model: regions overlap\n"); return -3; }
		}
		StripH st;
		strip_bind(st, pool.data(), L);
		const uint32_t nw = (uint32_t)L.nw, qmax = (nw - 3) * 32;
		for (uint32_t k = 0; k < nw; k++) {
			const uint64_t wi = w0 + k;
			uint32_t word = 0;
			if (wi < nwords + 8) { uint8_t bts[4] = { 0, 0, 0, 0 }; for (int q = 0; q < 4; q++) { const uint64_t bi = wi * 4 + q; bts[q] = bi < nbytes ? graph[bi] : 0; } word = ((uint32_t)bts[0] << 24) | ((uint32_t)bts[1] << 16) | ((uint32_t)bts[2] << 8) | bts[3]; }
			st.win[k] = word;
		}
		const char *stage = "fields";
		auto escape = [&](int32_t i) { if (getenv("STRIP_MODEL_TRACE")) fprintf(stderr, "escape(why %d): strip %lld slot %d (i %d of %d) at %s: d %d ref %d nres %d niv %d ivCap %d segCap %d nw %d narcs %d\n", g_why, (long long)t, a + i, i, n, stage, (int)st.m_d[i], (int)st.m_ref[i], (int)st.m_nres[i], (int)st.m_niv[i], st.ivCap, st.segCap, L.nw, narcs); st.m_d[i] = 0; esc[(*nEsc)++] = a + i; };
		std::vector<int32_t> order;
		for (int i = 0; i < n; i++) {
			int32_t d = rd[i];
			const int64_t q0 = offsets[lo + a + i] - base, q1 = offsets[lo + a + i + 1] - base;
			st.m_ref[i] = (uint16_t)rr[i];
			st.m_off[i] = (uint16_t)roff[i];
			if (d > 0 && (!L.ok || q1 > (int64_t)qmax || q1 - q0 > 0xffff || q1 <= q0)) { escape(i); d = 0; }
			st.m_d[i] = (uint16_t)d;
			st.m_bit[i] = d ? (uint32_t)q0 : 0u;
			st.m_sbits[i] = d ? (uint16_t)(q1 - q0) : 0;
			st.m_nres[i] = 0; st.m_niv[i] = 0; st.m_cop[i] = 0; st.m_seg0[i] = 0xffff;
			if (d) order.push_back(i);
		}
		std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return clz32(st.m_d[x]) < clz32(st.m_d[y]); });
		stats[7] += (int64_t)order.size();
		Job job; job.W = W; job.minInt = minInt; job.zk = (uint32_t)zk; job.stripMax = stripMax; job.x0 = lo + a;
		int32_t cNiv = 0, cNblk = 0, ivLim = 0x7fffffff;
		auto drefOf = [&](int32_t i, int32_t r) -> int64_t { return a + i - r >= 0 ? (int64_t)outd[a + i - r] : -1; };
		auto ivAlloc = [&](int32_t c) -> int32_t { const int32_t o = cNiv; cNiv += c; if (o + c > st.ivCap) { ivLim = std::min(ivLim, o); return -1; } return o; };
		auto blkAlloc = [&](int32_t c) -> int32_t { const int32_t o = cNblk; cNblk += c; return o + c > st.blkCap ? -1 : o; };
		stage = "structure";
		for (int32_t i : order) {
			const uint32_t recEnd = st.m_bit[i] + (uint32_t)st.m_sbits[i];
			const bool ok = zk == 3 ? phase_structure<3>(st, job, qmax, i, recEnd, drefOf, ivAlloc, blkAlloc, true) : phase_structure<0>(st, job, qmax, i, recEnd, drefOf, ivAlloc, blkAlloc, true);
			if (!ok) escape(i);
		}
		stage = "segments";
		int32_t cNseg = 0, segLim = 0x7fffffff, nLongSeg = 0, nLongRec = 0;
		for (int32_t i = 0; i < n; i++) {
			const uint32_t nRes = st.m_nres[i];
			if (st.m_d[i] == 0 || nRes == 0) continue;
			const int32_t m = segments_of(nRes, st.m_sbits[i]);
			const int32_t e0 = cNseg; cNseg += m;
			if (e0 + m > st.segCap) { segLim = std::min(segLim, e0); escape(i); continue; }
			st.m_seg0[i] = (uint16_t)e0;
			if (m == 1) { st.seg[e0].start = st.m_bit[i]; st.seg[e0].end = 0; st.seg[e0].base = job.x0 + i; st.seg[e0].cnt = (uint16_t)nRes; st.seg[e0].rec = (uint16_t)i; }
			else {
				for (int32_t k = 0; k < m; k++) { st.seg[e0 + k].end = (uint32_t)k; st.seg[e0 + k].rec = (uint16_t)i; st.seg[e0 + k].cnt = 0; st.listB[nLongSeg + k] = (uint16_t)(e0 + k); }
				nLongSeg += m;
				st.list[st.listLen - 1 - nLongRec++] = (uint16_t)i;
			}
		}
		const int32_t nSeg = std::min(cNseg, segLim);
		stats[2] += nSeg; stats[3] += nLongSeg;
		// phase A, then count how many anchors had not locked on (phase B re-decodes those)
		for (int32_t t2 = 0; t2 < nLongSeg; t2++) { if (zk == 3) phase_anchor<3>(st, job, qmax, (int32_t)st.listB[t2]); else phase_anchor<0>(st, job, qmax, (int32_t)st.listB[t2]); }
		stage = "chain";
		for (int32_t t2 = 0; t2 < nLongRec; t2++) {
			const int32_t i = (int32_t)st.list[st.listLen - 1 - t2];
			if (st.m_d[i] == 0) continue;
			const int32_t e0 = st.m_seg0[i], m = segments_of(st.m_nres[i], st.m_sbits[i]);
			for (int32_t k = 1; k < m; k++) if (st.seg[e0 + k].start != st.seg[e0 + k - 1].end) stats[4]++;
			const bool ok = zk == 3 ? phase_chain<3>(st, job, qmax, i) : phase_chain<0>(st, job, qmax, i);
			if (!ok) escape(i);
		}
		std::vector<int32_t> work;
		for (int32_t e = 0; e < nSeg; e++) { const int32_t i = st.seg[e].rec; if (st.m_d[i] && st.seg[e].cnt > 0) work.push_back(e); }
		std::stable_sort(work.begin(), work.end(), [&](int x, int y) { return std::min<int>(st.seg[x].cnt >> 2, 31) > std::min<int>(st.seg[y].cnt >> 2, 31); });
		for (size_t k = 0; k < work.size(); k++) st.listB[k] = (uint16_t)work[k];
		stage = "residuals";
		for (size_t k = 0; k < work.size(); k++) {
			const int32_t e = st.listB[k];
			const bool ok = zk == 3 ? phase_residuals<3>(st, job, qmax, e) : phase_residuals<0>(st, job, qmax, e);
			if (!ok) { const int32_t i = st.seg[e].rec; if (st.m_d[i]) escape(i); }
		}
		const int32_t nIv = std::min(cNiv, ivLim);
		stats[5] += nIv;
		for (int32_t j = 0; j < nIv; j++) {
			if ((int32_t)st.iv_len[j] >= LONG_INTERVAL) { stats[6]++; for (int l = 0; l < 64; l++) phase_interval(st, j, l, 64); }
			else phase_interval(st, j, 0, 1);
		}
		for (int32_t i = 0; i < n; i++) {
			const int32_t d = st.m_d[i];
			if (!d) continue;
			cop[a + i] = st.m_cop[i];
			int32_t *dst = succ + (rowstart[a + i] - rowstart[0]);
			for (int32_t t2 = 0; t2 < d; t2++) dst[t2] = st.rows[st.m_off[i] + t2];
		}
	}
	return 0;
}
}
