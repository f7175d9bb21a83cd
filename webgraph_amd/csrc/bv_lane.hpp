// bv_lane.hpp -- one lane decodes one whole record (gfx950).
//
// The work-horse for short and medium records: 64 records per wavefront, each lane walking its own record.
// It fuses everything the reference does lazily per successor (BVG:1032-1133): the masked copy of the
// referent's list (MaskedIntIterator.java:65-97), the interval expansion (IntIntervalSequenceIterator.java:64-78)
// and the residual gaps (BVG:939-991) are merged three ways (MergedIntIterator.java:50-74) straight into the
// CSR row, so every successor is written exactly once.  The referent's row must already be final, which is why
// records are decoded level by level of their reference chain (k_decode_level in bv_kernels.hip).
//
// Three cursors walk the same record: the main one (header, then the residual section) streams through HBM
// with 16-byte prefetching loads; the copy-block cursor and the interval cursor re-read short, already cached
// parts of the record with plain loads.
#pragma once
#include "bv_device.hpp"

namespace bv {

// gfx950 has ONE counter (vmcnt) for outstanding global loads and stores, so every load whose result is needed
// drains the stores issued before it.  A loop that stores one id and loads one referent id per iteration runs
// at memory latency per successor.  Hence both directions go through small per-lane LDS buffers and touch
// HBM in bursts: LANE_BURST ids per store burst, LANE_BURST referent ids per load burst.
constexpr int LANE_BURST = 16;
constexpr int LANE_LDS_INTS_PER_THREAD = 2 * LANE_BURST; // out buffer + copy buffer, laid out [slot][thread]

template <int DEF, bool HAS_REF>
__device__ __forceinline__ void decode_node_full(const GraphDev &g, int32_t x, int32_t d, int32_t r, int64_t dref, const int32_t *__restrict__ src,
                                                 int32_t *__restrict__ row, int32_t *lds, int *__restrict__ errOut) {
	// lds: LANE_LDS_INTS_PER_THREAD * blockDim.x ints; slot j of this thread is lds[j * blockDim.x + threadIdx.x]
	int32_t *obuf = lds + threadIdx.x;
	int32_t *cbuf = lds + LANE_BURST * blockDim.x + threadIdx.x;
	const int32_t lstride = blockDim.x;
	constexpr int32_t INF = 0x7fffffff; // node ids are < 2^31 - 1
	int err = 0;
	PReader pr;
	pr.init(g.bits, g.nwords);
	pr.seek((uint64_t)g.offsets[x]);
	(void)Fields<DEF>::outdegree(pr, g);
	if (g.W > 0) (void)Fields<DEF>::reference(pr, g);

	// All counters are 32-bit: they are bounded by the outdegree (< 2^31), and ids use Java's int arithmetic.
	// ---- copy blocks: totals now, the blocks themselves again lazily through `bb` (BVG:1058-1071)
	const uint32_t drefU = (uint32_t)dref;
	uint32_t copied = 0, bc = 0;
	BitReader bb;
	bb.init(g.bits, g.nwords);
	if (HAS_REF && r > 0) {
		const uint64_t bc64 = Fields<DEF>::block_count(pr, g);
		const uint64_t blocksPos = pr.pos();
		uint64_t total = 0, cp = 0;
		if (bc64 > (uint64_t)drefU + 1) err |= E_FORMAT;
		else {
			bc = (uint32_t)bc64;
			for (uint32_t b = 0; b < bc; b++) {
				const uint64_t len = Fields<DEF>::block(pr, g) + (b ? 1 : 0);
				total += len;
				if (!(b & 1)) cp += len;
			}
			if (total > drefU) err |= E_FORMAT;
			else if (!(bc & 1)) cp += drefU - total;
		}
		if (cp > (uint64_t)d) err |= E_FORMAT;
		copied = (uint32_t)cp;
		if (bc) bb.seek(blocksPos);
	}
	if (err | pr.err) { atomicOr(errOut, err | pr.err); return; }
	const uint32_t extra = (uint32_t)d - copied;

	// ---- intervals: skip-parse to find the residual section; the values again lazily through `bi` (BVG:1073-1096)
	uint32_t ivTodo = 0;
	uint64_t intervalArcs = 0;
	BitReader bi;
	bi.init(g.bits, g.nwords);
	if (extra > 0 && g.minInt != 0) {
		const uint64_t ic = pr.gamma();
		if (ic > extra) { atomicOr(errOut, E_FORMAT); return; }
		ivTodo = (uint32_t)ic;
		if (ivTodo) {
			bi.seek(pr.pos());
			for (uint32_t i = 0; i < ivTodo; i++) {
				(void)pr.gamma();
				intervalArcs += pr.gamma() + (uint64_t)g.minInt;
			}
		}
	}
	if (intervalArcs > extra || pr.err) { atomicOr(errOut, E_FORMAT | pr.err); return; }
	uint32_t resTodo = extra - (uint32_t)intervalArcs;

	// ---- the three streams
	uint32_t cTodo = copied, cLeft = 0, ci = 0; // arcs still to copy, arcs left in the current copy block, index in the referent row
	uint32_t bIdx = 0;
	int32_t cN = INF;                           // head of the copy stream
	uint32_t cbBase = 0, cbFill = 0;            // cbuf holds src[cbBase .. cbBase + cbFill)
	uint32_t ivRem = 0, ivCur = 0, ivPrevEnd = 0; // interval stream (ids as Java ints: wrapping 32-bit arithmetic)
	bool ivFirst = true;
	int32_t iN = INF;
	bool resFirst = true;                       // residual stream
	uint32_t rPrev = 0;
	int32_t rN = INF;

	auto next_copy = [&]() {
		if (!HAS_REF || cTodo == 0) { cN = INF; return; }
		while (cLeft == 0) {
			if (bIdx < bc) {
				const uint32_t len = (uint32_t)Fields<DEF>::block(bb, g) + (bIdx ? 1u : 0u);
				if (bIdx & 1) ci += len; else cLeft = len;
				bIdx++;
			} else cLeft = drefU - ci; // implicit last block: the rest of the referent (block count even)
		}
		if (ci - cbBase >= cbFill) { // burst-load the next referent ids (all loads in flight together, one wait)
			cbBase = ci;
			cbFill = min((uint32_t)LANE_BURST, drefU - ci);
			int32_t tmp[LANE_BURST];
#pragma unroll
			for (int j = 0; j < LANE_BURST; j++) tmp[j] = (uint32_t)j < cbFill ? src[ci + j] : 0;
#pragma unroll
			for (int j = 0; j < LANE_BURST; j++) cbuf[j * lstride] = tmp[j];
		}
		cN = cbuf[(ci - cbBase) * lstride];
		ci++; cLeft--; cTodo--;
	};
	auto next_iv = [&]() {
		if (ivRem == 0) {
			if (ivTodo == 0) { iN = INF; return; }
			if (ivFirst) { ivCur = (uint32_t)((int64_t)x + nat2int(bi.gamma())); ivFirst = false; } // BVG:1084
			else ivCur = ivPrevEnd + (uint32_t)bi.gamma() + 1u;                                    // BVG:1090
			ivRem = (uint32_t)bi.gamma() + (uint32_t)g.minInt;
			ivPrevEnd = ivCur + ivRem;
			ivTodo--;
		}
		iN = (int32_t)ivCur;
		ivCur++; ivRem--;
	};
	auto next_res = [&]() {
		if (resTodo == 0) { rN = INF; return; }
		const uint64_t v = Fields<DEF>::residual(pr, g);
		rPrev = resFirst ? (uint32_t)((int64_t)x + nat2int(v)) : rPrev + (uint32_t)v + 1u; // BVG:954, :966
		resFirst = false;
		rN = (int32_t)rPrev;
		resTodo--;
	};
	next_copy(); next_iv(); next_res();
	// Successors are collected in LDS and leave in bursts of LANE_BURST ids, as 16-byte stores where the row
	// is 16-byte aligned (a wave's 64 lanes write 64 different rows: every store instruction is 64 requests).
	auto flush = [&](int32_t kEnd, int32_t cnt) { // writes obuf[0..cnt) to row[kEnd-cnt .. kEnd)
		int32_t *dst = row + (kEnd - cnt);
		int32_t j = 0;
		while (j < cnt && (((uintptr_t)(dst + j)) & 15u)) { dst[j] = obuf[j * lstride]; j++; }
		for (; j + 4 <= cnt; j += 4) *(int4 *)(dst + j) = int4{ obuf[j * lstride], obuf[(j + 1) * lstride], obuf[(j + 2) * lstride], obuf[(j + 3) * lstride] };
		for (; j < cnt; j++) dst[j] = obuf[j * lstride];
	};
	int32_t on = 0;
	for (int32_t k = 0; k < d; k++) {
		const int32_t m = min(cN, min(iN, rN));
		obuf[on * lstride] = m == INF ? -1 : m; // -1: fewer values than the outdegree promises (malformed; BVG:1210 stores -1 too)
		// equal heads are emitted once (MergedIntIterator.java:69-72)
		if (HAS_REF && cN == m) next_copy();
		if (iN == m) next_iv();
		if (rN == m) next_res();
		if (++on == LANE_BURST) { flush(k + 1, on); on = 0; }
	}
	if (on) flush(d, on);
	err |= pr.err | bb.err | bi.err;
	if (err) atomicOr(errOut, err);
}

} // namespace bv
