// seg_model.cpp -- host-side model of the segment pipeline (webgraph_amd/csrc/bv_seg.hip): the SAME bodies (bv_seg.hpp, compiled here
// for the CPU) driven lane after lane in the order the kernels run them -- struct (here: see below), A1, A2, scan, B, expand.  Test infrastructure:
// tests/test_seg_model_cpu.py builds it with g++ and compares what it decodes with the CPU oracle, so that the logic of the kernels is
// checked in the `-m "not gpu"` suite before it ever runs on a GPU.  Not part of the product.
#include "../../webgraph_amd/csrc/bv_seg.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace bvsg;

// The structure of a record as one lane reads it (round 4's k_seg_struct, which gave the pipeline records of its own, is tag r4-experiments;
// in the product the descriptors of the pipeline's records come from the cooperative kernel, bv_coop.hpp).  Kept here: the model's source of descriptors.
namespace bvsg {
// The gamma-coded front of record x (outdegree d; referent's outdegree dref if it has a reference): BVG:1048-1096.
template <int STRIDE>
SG_D void struct_lane(const SegGraph &g, uint32_t *col, int32_t x, int32_t d, bool hasRef, int64_t dref, SegIv *iv, RecDesc &o) {
	Win<STRIDE> w;
	const uint64_t recEnd = (uint64_t)g.offsets[x + 1];
	w.init(g, col, recEnd);
	uint32_t q = w.seek((uint64_t)g.offsets[x]);
	bool bad = false;
	(void)w.gamma(q, bad);               // outdegree (known from k_headers)
	if (g.W > 0) { SG_REFILL(w, q); (void)w.unary(q, bad); } // reference
	int64_t copied = 0;
	if (hasRef) { // BVG:1058-1071
		SG_REFILL(w, q);
		const uint32_t bc = w.gamma(q, bad);
		int64_t total = 0;
		if ((int64_t)bc > dref + 1) bad = true;
		for (uint32_t b = 0; b < bc && !bad; b++) {
			SG_REFILL(w, q);
			const int64_t code = (int64_t)w.gamma(q, bad);
			if (code > dref - total) { bad = true; break; } // (a code of a malformed stream is rejected before it reaches a sum)
			const int64_t len = code + (b ? 1 : 0);
			if (total + len > dref) { bad = true; break; }
			total += len;
			if (!(b & 1)) copied += len;
		}
		if (!(bc & 1)) copied += dref - total;
	}
	const int64_t extra = (int64_t)d - copied;
	if (extra < 0 || copied < 0) bad = true;
	int32_t nIv = 0, ivArcs = 0;
	if (!bad && extra > 0 && g.minInt != 0) { // BVG:1073-1096
		SG_REFILL(w, q);
		const uint32_t ic = w.gamma(q, bad);
		const int32_t xtr = (int32_t)extra, minInt = g.minInt;
		if ((int64_t)ic > extra / minInt) bad = true; else nIv = (int32_t)ic;
		int32_t prevEnd = 0;
		for (int32_t i = 0; i < nIv && !bad; i++) {
			SG_REFILL(w, q);
			const uint32_t a = w.gamma(q, bad);
			SG_REFILL(w, q);
			const uint32_t l = w.gamma(q, bad);
			if (l > (uint32_t)xtr) { bad = true; break; }
			const int32_t left = i == 0 ? (int32_t)((uint32_t)x + (uint32_t)zigzag32(a)) : (int32_t)((uint32_t)prevEnd + a + 1u), n = (int32_t)l + minInt; // in Java ints (BVG:1084-1093)
			if (i > 0 && left < prevEnd) { bad = true; break; } // intervals that wrap around: not here
			prevEnd = (int32_t)((uint32_t)left + (uint32_t)n);
			iv[i] = SegIv{ left, ivArcs, -1, n }; // rank -1: behind every residual, unless B says otherwise
			ivArcs += n;                          // (<= extra + minInt: no overflow)
			if (ivArcs > xtr) { bad = true; break; }
		}
	}
	// (the zig-zag value of the first interval is a long in the file: one that does not fit 33 bits made gamma() say bad)
	const int64_t nres = extra - ivArcs;
	if (nres < 0) bad = true;
	o.rpos = (int64_t)w.pos(q);
	o.nres = bad ? 0 : (int32_t)nres;
	o.copied = bad ? 0 : (int32_t)copied;
	o.nIv = bad ? 0 : nIv;
	o.ivArcs = bad ? 0 : ivArcs;
	o.flags = bad ? RF_FALLBACK : 0;
	if (!bad && nres > 0 && (uint64_t)o.rpos >= recEnd) o.flags = RF_FALLBACK; // residuals past the record's end (offsets that disagree with the stream)
}

}

extern "C" {

// graph: the .graph bytes followed by >= 64 zero bytes (nbytes = file size).  View = nodes [lo, lo + cnt), no halo.
// outd / ref / rowstart as the kernels' RangeView holds them.  succ[rowstart[cnt]]: rows (only the records of the class are written).
// esc[cnt]: flagged slots (the cooperative kernel's work), *nEsc their number.  cop[cnt]: ids copied from the referent (-1: not this class).
// stats[8]: records, segments, segments whose chains did not meet, flagged records, pieces the fix pass walked again, intervals, max segments of a record, -
int seg_model_run(const uint8_t *graph, uint64_t nbytes, const int64_t *offsets, int32_t lo, int32_t cnt, const int32_t *outd, const uint16_t *ref,
                  const int64_t *rowstart, int W, int minInt, int zk, int dmin, int dmax, int32_t *succ, int32_t *esc, int32_t *nEsc, int32_t *cop, int64_t *stats) {
	SegGraph g{ (const uint32_t *)graph, (nbytes + 3) / 4, offsets, W, minInt, zk };
	*nEsc = 0;
	for (int s = 0; s < cnt; s++) cop[s] = -1;
	for (int k = 0; k < 8; k++) stats[k] = 0;
	std::vector<uint32_t> lds(WIN_WORDS);
	uint32_t *col = lds.data();
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
	const uint32_t cap = (uint32_t)((SEG_BITS / (uint32_t)(zk < 1 ? 1 : zk) + 2 + 3) & ~3u); // = bv::seg_cell_cap
	std::vector<SegA1> a1((size_t)S + 1);
	std::vector<SegFin> fin((size_t)S + 1);
	std::vector<uint32_t> pc((size_t)S + 1), ps((size_t)S + 1);
	std::vector<int32_t> seg2rec((size_t)S + 1), fixlist, pendlist;
	std::vector<uint8_t> flag(R, 0), miss((size_t)S + 1, 0);
	auto span = [&](int32_t sg, uint64_t &cell, uint32_t &a, uint32_t &b) {
		const int32_t r = seg2rec[sg];
		seg_span(desc[r], (uint64_t)offsets[lo + desc[r].slot + 1], sg - segbase[r], cell, a, b);
	};
	auto a2 = [&](int32_t sg, uint32_t inRel, bool follow, SegFin &o) {
		uint64_t cell; uint32_t a, b;
		span(sg, cell, a, b);
		if (zk == 3) seg_a2<3, 1>(g, col, cell, inRel, b, a1[sg], (int32_t *)nullptr, cap, (uint32_t *)nullptr, false, follow, o);
		else seg_a2<0, 1>(g, col, cell, inRel, b, a1[sg], (int32_t *)nullptr, cap, (uint32_t *)nullptr, false, follow, o);
	};
	// A1
	for (size_t r = 0; r < R; r++) {
		const int32_t x = lo + desc[r].slot;
		for (int32_t i = 0; i < segbase[r + 1] - segbase[r]; i++) {
			const int32_t sg = segbase[r] + i;
			seg2rec[sg] = (int32_t)r;
			uint64_t cell; uint32_t a, b;
			span(sg, cell, a, b);
			if (zk == 3) seg_a1<3, 1>(g, col, x, cell, a, b, i == 0, (int32_t *)nullptr, cap, a1[sg]);
			else seg_a1<0, 1>(g, col, x, cell, a, b, i == 0, (int32_t *)nullptr, cap, a1[sg]);
			if (i == 0 && a1[sg].badIdx != ~0u) flag[r] = 1;
		}
	}
	// A2 (reads what A1 wrote of this piece and the one before; writes fin / miss of its own piece only)
	for (int32_t sg = 0; sg < S; sg++) {
		const int32_t r = seg2rec[sg], i = sg - segbase[r];
		uint64_t cell; uint32_t a, b;
		span(sg, cell, a, b);
		SegFin o{ a, a1[sg].cnt, a1[sg].sum, 0, 0, 0, 0, a1[sg].badIdx != ~0u ? 2u : 0u };
		if (i > 0) a2(sg, a1[sg - 1].outRel - SEG_BITS, false, o);
		if ((o.mode & 3) == 3) pendlist.push_back(sg);
		fin[sg] = o;
	}
	// fix, phase 1: the true chain of the pieces A2 left open, followed to the end of the piece
	for (int32_t k : pendlist) {
		const int32_t r = seg2rec[k];
		uint64_t cell; uint32_t a, b;
		span(k, cell, a, b);
		SegFin o = fin[k];
		Win<1> w;
		w.init(g, col, cell + b);
		uint32_t q = w.seek(cell + o.inRel), badIdx;
		const uint32_t qend = q + (b > o.inRel ? b - o.inRel : 0u);
		if (zk == 3) decode_run<3, 1>(g, w, q, qend, false, 0, nullptr, ~0u, o.cnt, o.sum, badIdx);
		else decode_run<0, 1>(g, w, q, qend, false, 0, nullptr, ~0u, o.cnt, o.sum, badIdx);
		o.tRel = (uint32_t)(w.pos(q) - cell);
		o.mode = badIdx != ~0u ? 2u : 1u;
		fin[k] = o;
		stats[2]++;
		if (o.mode == 1 && k + 1 != segbase[r + 1] && o.tRel != a1[k].outRel) { miss[k] = 1; fixlist.push_back(k); }
	}
	// fix
	for (int32_t k0 : fixlist) {
		const int32_t r = seg2rec[k0], kEnd = segbase[r + 1];
		if (k0 > segbase[r] && miss[k0 - 1]) continue;
		uint32_t inRel = fin[k0].tRel - SEG_BITS;
		for (int32_t k = k0 + 1, steps = 0; k < kEnd; k++, steps++) {
			if (steps >= FIX_MAX) { flag[r] = 1; break; }
			SegFin o;
			a2(k, inRel, true, o);
			stats[4]++;
			fin[k] = o;
			if ((o.mode & 3) == 2) break;
			if ((o.mode & 3) == 1 && o.tRel != a1[k].outRel) { inRel = o.tRel - SEG_BITS; continue; }
			if (k + 1 < kEnd && miss[k] && miss[k + 1]) { inRel = a1[k].outRel - SEG_BITS; continue; }
			break;
		}
	}
	// scan
	pc[0] = ps[0] = 0;
	for (int32_t sg = 0; sg < S; sg++) { pc[sg + 1] = pc[sg] + fin[sg].cnt; ps[sg + 1] = ps[sg] + fin[sg].sum; }
	// B
	std::vector<uint32_t> ringv(2 * RING);
	for (int32_t sg = 0; sg < S; sg++) {
		const int32_t r = seg2rec[sg];
		if (flag[r]) continue; // (on the GPU a record may be flagged while its other pieces are already being written: harmless, the cooperative kernel rewrites the row)
		const int32_t k0 = segbase[r], i = sg - k0, s = desc[r].slot;
		const bool last = sg + 1 == segbase[r + 1];
		const SegFin me = fin[sg];
		bool ok = (me.mode & 3) < 2;
		if (last) ok = ok && pc[sg] + me.cnt - pc[k0] == (uint32_t)desc[r].nres;
		if (ok) {
			uint64_t cell; uint32_t a, b;
			span(sg, cell, a, b);
			int32_t *out = succ + (rowstart[s] - rowstart[0]) + desc[r].copied;
			uint32_t endRel;
			ok = zk == 3 ? seg_b<3, 1>(g, col, ringv.data(), lo + s, cell, me.inRel, me.cnt, pc[sg] - pc[k0], (int32_t)(ps[sg] - ps[k0]), i == 0, out, outd[s] - desc[r].copied, iv_of(s), desc[r].nIv, endRel)
			             : seg_b<0, 1>(g, col, ringv.data(), lo + s, cell, me.inRel, me.cnt, pc[sg] - pc[k0], (int32_t)(ps[sg] - ps[k0]), i == 0, out, outd[s] - desc[r].copied, iv_of(s), desc[r].nIv, endRel);
			if (!last) ok = ok && endRel == fin[sg + 1].inRel + SEG_BITS;
		}
		if (!ok) { flag[r] = 1; if (getenv("SEG_MODEL_TRACE")) fprintf(stderr, "B: slot %d piece %d/%d mode %u cnt %u\n", s, i, segbase[r + 1] - k0, me.mode, me.cnt); }
	}
	// expand
	for (size_t r = 0; r < R; r++) {
		if (flag[r] || (desc[r].flags & RF_FALLBACK) || desc[r].nres <= 0) continue;
		const int32_t s = desc[r].slot;
		int32_t *out = succ + (rowstart[s] - rowstart[0]) + desc[r].copied;
		for (int32_t i = 0; i < desc[r].nIv; i++) expand_interval(iv_of(s)[i], desc[r].nres, out, outd[s] - desc[r].copied);
	}
	// what is left: flagged records -> the cooperative kernel; records without residuals have their intervals expanded by the struct lane
	for (size_t r = 0; r < R; r++) {
		const int32_t s = desc[r].slot;
		if (flag[r] || (desc[r].flags & RF_FALLBACK)) { esc[(*nEsc)++] = s; stats[3]++; continue; }
		cop[s] = desc[r].copied;
		stats[5] += desc[r].nIv;
		if (desc[r].nres == 0) {
			int32_t *out = succ + (rowstart[s] - rowstart[0]) + desc[r].copied;
			for (int32_t i = 0; i < desc[r].nIv; i++) expand_interval(iv_of(s)[i], 0, out, outd[s] - desc[r].copied);
		}
	}
	return 0;
}

}
