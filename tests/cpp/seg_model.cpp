// seg_model.cpp -- host-side model of the segment pipeline (webgraph_amd/csrc/bv_seg.hip): the SAME bodies (bv_seg.hpp, compiled here
// for the CPU) driven lane after lane in the order the kernels run them -- struct, A1, A2, scan, B, expand.  Test infrastructure:
// tests/test_seg_model_cpu.py builds it with g++ and compares what it decodes with the CPU oracle, so that the logic of the kernels is
// checked in the `-m "not gpu"` suite before it ever runs on a GPU.  Not part of the product.
#include "../../webgraph_amd/csrc/bv_seg.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace bvsg;

extern "C" {

// graph: the .graph bytes followed by >= 64 zero bytes (nbytes = file size).  View = nodes [lo, lo + cnt), no halo.
// outd / ref / rowstart as the kernels' RangeView holds them.  succ[rowstart[cnt]]: rows (only the records of the class are written).
// esc[cnt]: flagged slots (the cooperative kernel's work), *nEsc their number.  cop[cnt]: ids copied from the referent (-1: not this class).
// stats[8]: records, segments, segments whose chains did not meet, flagged records, codes walked by A2, intervals, max segments of a record, -
int seg_model_run(const uint8_t *graph, uint64_t nbytes, const int64_t *offsets, int32_t lo, int32_t cnt, const int32_t *outd, const uint16_t *ref,
                  const int64_t *rowstart, int W, int minInt, int zk, int dmin, int dmax, int32_t *succ, int32_t *esc, int32_t *nEsc, int32_t *cop, int64_t *stats) {
	SegGraph g{ (const uint32_t *)graph, (nbytes + 3) / 4, offsets, W, minInt, zk };
	*nEsc = 0;
	for (int s = 0; s < cnt; s++) cop[s] = -1;
	for (int k = 0; k < 8; k++) stats[k] = 0;
	std::vector<uint32_t> lds(WIN_WORDS + 2 * RING);
	Col<1> col{ lds.data() }, ring{ lds.data() + WIN_WORDS };
	const int64_t arcs = rowstart[cnt] - rowstart[0];
	std::vector<SegIv> arena((size_t)(minInt > 0 ? arcs / minInt + cnt + 2 : 1));
	// the class
	std::vector<int32_t> list;
	for (int32_t s = 0; s < cnt; s++) if (outd[s] >= std::max(dmin, 1) && outd[s] < dmax) list.push_back(s);
	const size_t R = list.size();
	stats[0] = (int64_t)R;
	std::vector<RecDesc> desc(R);
	std::vector<int32_t> segbase(R + 1, 0);
	auto iv_of = [&](int32_t s) { return arena.data() + (minInt > 0 ? (rowstart[s] - rowstart[0]) / minInt : 0); };
	// struct
	for (size_t r = 0; r < R; r++) {
		const int32_t s = list[r], x = lo + s, rf = ref[s];
		RecDesc d{};
		if (rf > s) { d.flags = RF_FALLBACK; } // (a sub-range without its halo: the real pipeline never shows the kernels such a row)
		else struct_lane<1>(g, col, x, outd[s], rf > 0, rf > 0 ? (int64_t)outd[s - rf] : 0, iv_of(s), d);
		d.slot = s;
		desc[r] = d;
		segbase[r + 1] = segbase[r] + seg_count(d, (uint64_t)offsets[x + 1]);
		stats[6] = std::max<int64_t>(stats[6], segbase[r + 1] - segbase[r]);
	}
	const int32_t S = segbase[R];
	stats[1] = S;
	std::vector<uint64_t> segOut((size_t)S + 1), segBad((size_t)S + 1);
	std::vector<uint32_t> segCnt((size_t)S + 1), segSum((size_t)S + 1), pc((size_t)S + 1), ps((size_t)S + 1);
	std::vector<int32_t> seg2rec((size_t)S + 1);
	std::vector<uint8_t> flag(R, 0);
	// A1
	for (size_t r = 0; r < R; r++) {
		const int32_t x = lo + desc[r].slot;
		for (int32_t i = 0; i < segbase[r + 1] - segbase[r]; i++) {
			const int32_t sg = segbase[r] + i;
			uint64_t a, b;
			seg_span(desc[r], (uint64_t)offsets[x + 1], i, a, b);
			if (zk == 3) seg_a1<3, 1>(g, col, x, a, b, i == 0, segOut[sg], segCnt[sg], segSum[sg], segBad[sg]);
			else seg_a1<0, 1>(g, col, x, a, b, i == 0, segOut[sg], segCnt[sg], segSum[sg], segBad[sg]);
			if (i == 0 && segBad[sg] != ~(uint64_t)0) flag[r] = 1;
			seg2rec[sg] = (int32_t)r;
		}
	}
	// A2 (reads the ends A1 wrote, writes counts and sums of its own segment only)
	for (size_t r = 0; r < R; r++) {
		const int32_t x = lo + desc[r].slot, ns = segbase[r + 1] - segbase[r];
		for (int32_t i = 1; i < ns; i++) {
			const int32_t sg = segbase[r] + i;
			uint64_t a, b;
			seg_span(desc[r], (uint64_t)offsets[x + 1], i, a, b);
			bool bad = false;
			const uint32_t c0 = segCnt[sg];
			const bool ok = zk == 3 ? seg_a2<3, 1>(g, col, a, segOut[sg - 1], b, i == ns - 1, segBad[sg], segCnt[sg], segSum[sg]) : seg_a2<0, 1>(g, col, a, segOut[sg - 1], b, i == ns - 1, segBad[sg], segCnt[sg], segSum[sg]);
			(void)c0;
			if (!ok || bad) { flag[r] = 1; stats[2]++; if (getenv("SEG_MODEL_TRACE")) fprintf(stderr, "a2 fail: slot %d seg %d/%d gstart %llu in %llu end %llu bad %d cnt0 %u\n", desc[r].slot, i, ns, (unsigned long long)a, (unsigned long long)segOut[sg - 1], (unsigned long long)b, (int)bad, c0); }
		}
	}
	// scan
	pc[0] = ps[0] = 0;
	for (int32_t sg = 0; sg < S; sg++) { pc[sg + 1] = pc[sg] + segCnt[sg]; ps[sg + 1] = ps[sg] + segSum[sg]; }
	for (size_t r = 0; r < R; r++) if (segbase[r + 1] > segbase[r] && pc[segbase[r + 1]] - pc[segbase[r]] != (uint32_t)desc[r].nres) flag[r] = 1;
	// B
	for (size_t r = 0; r < R; r++) {
		if (flag[r] || (desc[r].flags & RF_FALLBACK)) continue;
		const int32_t s = desc[r].slot, x = lo + s, ns = segbase[r + 1] - segbase[r];
		int32_t *out = succ + (rowstart[s] - rowstart[0]) + desc[r].copied;
		const int32_t extra = outd[s] - desc[r].copied;
		for (int32_t i = 0; i < ns; i++) {
			const int32_t sg = segbase[r] + i;
			uint64_t a, b;
			seg_span(desc[r], (uint64_t)offsets[x + 1], i, a, b);
			const uint64_t in = i == 0 ? a : segOut[sg - 1];
			const uint32_t j0 = pc[sg] - pc[segbase[r]];
			const int32_t v0 = (int32_t)(ps[sg] - ps[segbase[r]]);
			const bool ok = zk == 3 ? seg_b<3, 1>(g, col, ring, x, in, b, segCnt[sg], j0, v0, i == 0, out, extra, iv_of(s), desc[r].nIv)
			                        : seg_b<0, 1>(g, col, ring, x, in, b, segCnt[sg], j0, v0, i == 0, out, extra, iv_of(s), desc[r].nIv);
			if (!ok) flag[r] = 1;
		}
	}
	// expand
	for (size_t r = 0; r < R; r++) {
		const int32_t s = desc[r].slot;
		if (flag[r] || (desc[r].flags & RF_FALLBACK)) { esc[(*nEsc)++] = s; stats[3]++; continue; }
		cop[s] = desc[r].copied;
		int32_t *out = succ + (rowstart[s] - rowstart[0]) + desc[r].copied;
		const int32_t extra = outd[s] - desc[r].copied;
		stats[5] += desc[r].nIv;
		for (int32_t i = 0; i < desc[r].nIv; i++) expand_interval(iv_of(s)[i], desc[r].nres, out, extra);
	}
	return 0;
}

}
