// bv_encode.hip -- BVGraph.store on the GPU (gfx950): a CSR in HBM -> the .graph bit stream, the bit offsets and the
// .offsets stream, byte for byte what the reference's single-threaded compressor writes.
//
// The reference compresses node after node: per node, up to W + 1 runs of diffComp against a bit-counting stream, the
// cheapest admissible one again for real (CompressionThread.call, BVGraph.java:2222-2386).  Only the admissibility --
// the length of the reference chain, :2313-2327 -- links a node to its predecessors; the W + 1 costs do not.  So:
//   A  k_enc_cost    one lane per (node, candidate) pair: the pair's cost in bits.  The 8 candidates of a node sit in
//                    neighbouring lanes (same successor list, similar trip counts).                [the bulk of the work]
//   B  k_enc_select  the chain-length recurrence, cut into chunks of SEL_CHUNK nodes: every chunk runs from a guessed
//                    state of the W nodes before it, then again only if its predecessor's final state turned out
//                    different.  On a copy-model graph the choice forgets its past within a few nodes (a node that
//                    takes no reference, or whose cheapest candidate is admissible either way); on a web graph runs of
//                    similar pages carry the phase of their chains for thousands of nodes (cnr-2000: 7 000), so after
//                    round 0 a lane walks SEL_SPAN chunks in order, skipping those whose in-state did not move.  Exact:
//                    the loop runs until a round moves nothing (bve::select_span).
//   C  k_enc_reclen + scan: record lengths -> bit offsets (what the reference's .offsets file holds).
//   D  k_enc_emit    one lane per node writes its record at its offset; words shared by two records are ORed.
//   E  the .offsets stream (gamma / delta coded gaps) the same way: lengths, scan, emit.
// The per-node logic is bv_encode.hpp, shared with the host model that the CPU tests compare with the CPU writer.
#include "bv_encode.hpp"
#include "bv_launch.hpp"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace bv {

using bve::Params;

constexpr int SEL_CHUNK = 64; // nodes per chunk of the selection recurrence
constexpr int SEL_SPAN = 16, SEL_BATCH = 8;
constexpr int ENC_MAX_W = 63; // state of a chunk boundary: W chain lengths

__global__ void __launch_bounds__(256) k_enc_check(const int32_t *__restrict__ succ, int64_t m, unsigned long long *__restrict__ viol) {
	const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
	const bool bad = a >= 1 && a < m && succ[a] <= succ[a - 1];
	const unsigned long long cnt = __popcll(__ballot(bad));
	if (cnt && (threadIdx.x & 63) == 0) atomicAdd(viol, cnt);
}
// descents at the first successor of a row are not violations: count them too (row starts are distinct positions)
__global__ void __launch_bounds__(256) k_enc_check_rows(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int32_t n, unsigned long long *__restrict__ viol) {
	const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
	bool hit = false;
	if (x < n) {
		const int64_t a = rowptr[x];
		hit = rowptr[x + 1] > a && a >= 1 && succ[a] <= succ[a - 1];
	}
	const unsigned long long cnt = __popcll(__ballot(hit));
	if (cnt && (threadIdx.x & 63) == 0) atomicAdd(viol + 1, cnt);
}

__global__ void __launch_bounds__(256) k_enc_cost(const Params p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int64_t npairs, uint32_t *__restrict__ cost, int *__restrict__ err) {
	const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (q >= npairs) return;
	const int cyc = p.W + 1;
	const int32_t x = (int32_t)(q / cyc);
	const int r = (int)(q - (int64_t)x * cyc);
	int e = 0;
	cost[q] = bve::pair_cost(p, rowptr, succ, x, r, &e);
	if (e) atomicOr(err, e);
}

// one round of bve::select_span: lane l walks the chunks [l * span, (l + 1) * span)
__global__ void __launch_bounds__(64) k_enc_select(const Params p, const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ cost, int32_t n, int64_t nchunks, int span, int round,
                                                   const int32_t *__restrict__ statePrev, int32_t *__restrict__ stateNew, int32_t *__restrict__ used,
                                                   uint8_t *__restrict__ best, int32_t *__restrict__ refc, int *__restrict__ moved) {
	const int64_t c0 = ((int64_t)blockIdx.x * 64 + threadIdx.x) * span;
	if (c0 >= nchunks) return;
	int32_t in[ENC_MAX_W + 1];
	const int64_t c1 = c0 + span < nchunks ? c0 + span : nchunks;
	if (bve::select_span(p, rowptr, cost, n, SEL_CHUNK, c0, c1, round, statePrev, stateNew, used, best, refc, in)) *moved = 1;
}

__global__ void __launch_bounds__(256) k_enc_reclen(const Params p, const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ cost, const uint8_t *__restrict__ best, int32_t n, int32_t *__restrict__ reclen, int *__restrict__ err) {
	const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (x >= n) return;
	const int64_t d = rowptr[x + 1] - rowptr[x];
	bve::LenSink s;
	bve::w_code(s, p.c_outd, (uint64_t)d, 0);
	uint64_t t = s.bits;
	if (d > 0) t += cost[x * (p.W + 1) + best[x]];
	if (t > bve::COST_MAX) { atomicOr(err, 2); t = 0; }
	reclen[x] = (int32_t)t;
}

struct EncStatsDev { unsigned long long v[12]; }; // bitsOutd, bitsRef, bitsBlocks, bitsIntervals, bitsResiduals, copied, intervalised, residuals, totRef, totDist, maxRef, -

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

__global__ void __launch_bounds__(256) k_enc_emit(const Params p, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, const uint8_t *__restrict__ best, const int32_t *__restrict__ refc,
                                                  const int64_t *__restrict__ off, int32_t n, uint32_t *__restrict__ words, EncStatsDev *__restrict__ stats) {
	const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
	bve::NodeStats st;
	unsigned long long totRef = 0, totDist = 0, chain = 0;
	if (x < n) {
		const int r = best[x];
		(void)bve::emit_node(p, rowptr, succ, (int32_t)x, r, words, (uint64_t)off[x], &st);
		if (rowptr[x + 1] > rowptr[x]) { totRef = (unsigned long long)refc[x]; totDist = (unsigned long long)r; chain = totRef; }
	}
	const unsigned long long vals[10] = { st.bitsOutd, st.bitsRef, st.bitsBlocks, st.bitsIntervals, st.bitsResiduals, st.copied, st.intervalised, st.residuals, totRef, totDist };
#pragma unroll
	for (int i = 0; i < 10; i++) {
		const unsigned long long s = wave_sum(vals[i]);
		if ((threadIdx.x & 63) == 0 && s) atomicAdd(&stats->v[i], s);
	}
	for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(chain, o); chain = t > chain ? t : chain; }
	if ((threadIdx.x & 63) == 0 && chain) atomicMax(&stats->v[10], chain);
}

// the .offsets stream: code 0 is the offset of node 0, code i the length of record i - 1 (BVGraph.java:2285, :2369)
__global__ void __launch_bounds__(256) k_enc_offlen(const Params p, const int32_t *__restrict__ reclen, int32_t n, int32_t *__restrict__ len) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i > n) return;
	bve::LenSink s;
	bve::w_code(s, p.c_off, i == 0 ? 0 : (uint64_t)reclen[i - 1], 0);
	len[i] = (int32_t)s.bits;
}
__global__ void __launch_bounds__(256) k_enc_offemit(const Params p, const int32_t *__restrict__ reclen, const int64_t *__restrict__ at, int32_t n, uint32_t *__restrict__ words) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i > n) return;
	bve::WordSink s(words, (uint64_t)at[i]);
	bve::w_code(s, p.c_off, i == 0 ? 0 : (uint64_t)reclen[i - 1], 0);
	s.finish();
}

// ---------------------------------------------------------------------------------------------------------------- host
void encode_free(EncodeOut &o) {
	for (void *q : { (void *)o.graph_words, (void *)o.off_words, (void *)o.offsets }) if (q) (void)hipFree(q);
	o.graph_words = nullptr; o.off_words = nullptr; o.offsets = nullptr;
}

int encode_device(const Params &p, int32_t n, const int64_t *d_rowptr, const int32_t *d_succ, uint64_t m, EncodeOut &out, std::string &err, hipStream_t st) {
	out = EncodeOut{};
	if (p.W < 0 || p.W > ENC_MAX_W) { err = "windowsize above 63 is not supported by the device compressor"; return -3; }
	const bool trace = getenv("BVGPU_ENC_TRACE") != nullptr;
	const int cyc = p.W + 1;
	const int64_t npairs = (int64_t)n * cyc;
	const int64_t nchunks = ((int64_t)n + SEL_CHUNK - 1) / SEL_CHUNK;
	const int64_t ns = scan_num_sums((int64_t)n + 1);
	uint32_t *cost = nullptr;
	uint8_t *best = nullptr;
	int32_t *refc = nullptr, *reclen = nullptr, *offlen = nullptr, *state = nullptr, *used = nullptr;
	int64_t *sums = nullptr, *offat = nullptr;
	int *flags = nullptr, *moved = nullptr; // flags[0]: error bits; moved[i]: did round i of the batch change a chunk's final state
	unsigned long long *viol = nullptr;
	EncStatsDev *dstats = nullptr;
	std::vector<hipEvent_t> ev;
	auto mark = [&]() { if (trace) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, st); ev.push_back(e); } };
	auto cleanup = [&](int rc) {
		for (void *q : { (void *)cost, (void *)best, (void *)refc, (void *)reclen, (void *)offlen, (void *)state, (void *)used, (void *)sums, (void *)offat, (void *)flags, (void *)moved, (void *)viol, (void *)dstats })
			if (q) (void)hipFree(q);
		for (auto e : ev) (void)hipEventDestroy(e);
		if (rc) { encode_free(out); (void)hipGetLastError(); }
		return rc;
	};
	auto alloc = [&](void **q, size_t bytes) { return hipMalloc(q, bytes ? bytes : 16) == hipSuccess; };
	const size_t nn = (size_t)n + 1;
	if (!alloc((void **)&cost, sizeof(uint32_t) * (size_t)npairs) || !alloc((void **)&best, nn) || !alloc((void **)&refc, sizeof(int32_t) * nn) ||
	    !alloc((void **)&reclen, sizeof(int32_t) * nn) || !alloc((void **)&offlen, sizeof(int32_t) * nn) || !alloc((void **)&state, sizeof(int32_t) * 2 * (size_t)nchunks * (size_t)(p.W ? p.W : 1)) ||
	    !alloc((void **)&used, sizeof(int32_t) * (size_t)nchunks * (size_t)(p.W ? p.W : 1)) || !alloc((void **)&sums, sizeof(int64_t) * (size_t)(ns + 1)) ||
	    !alloc((void **)&offat, sizeof(int64_t) * (nn + 1)) || !alloc((void **)&flags, 2 * sizeof(int)) || !alloc((void **)&moved, SEL_BATCH * sizeof(int)) || !alloc((void **)&viol, 2 * sizeof(unsigned long long)) ||
	    !alloc((void **)&dstats, sizeof(EncStatsDev)) || !alloc((void **)&out.offsets, sizeof(int64_t) * nn)) { err = "device allocation failed"; return cleanup(-5); }
	(void)hipMemsetAsync(flags, 0, 2 * sizeof(int), st);
	(void)hipMemsetAsync(viol, 0, 2 * sizeof(unsigned long long), st);
	(void)hipMemsetAsync(dstats, 0, sizeof(EncStatsDev), st);
	mark();
	auto blocks = [](int64_t items, int per) { return dim3((unsigned)((items + per - 1) / per > 0 ? (items + per - 1) / per : 1)); };
	// rows must be strictly increasing (the reference's iterators guarantee it; a CSR from elsewhere may not)
	hipLaunchKernelGGL(k_enc_check, blocks((int64_t)m, 256), dim3(256), 0, st, d_succ, (int64_t)m, viol);
	hipLaunchKernelGGL(k_enc_check_rows, blocks(n, 256), dim3(256), 0, st, d_rowptr, d_succ, n, viol);
	// A
	if (npairs) hipLaunchKernelGGL(k_enc_cost, blocks(npairs, 256), dim3(256), 0, st, p, d_rowptr, d_succ, npairs, cost, flags);
	mark();
	// B: round 0 one chunk per lane, then SEL_SPAN chunks per lane; SEL_BATCH rounds are enqueued between two looks at the flags
	int rounds = 0;
	const size_t stateHalf = (size_t)nchunks * (size_t)(p.W ? p.W : 1);
	for (bool settled = nchunks == 0; !settled;) {
		(void)hipMemsetAsync(moved, 0, SEL_BATCH * sizeof(int), st);
		const int first = rounds;
		for (int i = 0; i < SEL_BATCH; i++, rounds++) {
			const int span = rounds == 0 ? 1 : SEL_SPAN;
			const int64_t lanes = (nchunks + span - 1) / span;
			int32_t *sPrev = state + (size_t)((rounds + 1) & 1) * stateHalf, *sNew = state + (size_t)(rounds & 1) * stateHalf;
			hipLaunchKernelGGL(k_enc_select, blocks(lanes, 64), dim3(64), 0, st, p, d_rowptr, cost, n, nchunks, span, rounds, sPrev, sNew, used, best, refc, moved + i);
			if (p.W == 0 || p.R == 0) { rounds++; break; } // no references at all: nothing to settle
		}
		int h[SEL_BATCH];
		if (hipMemcpyAsync(h, moved, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "the selection kernel failed"; return cleanup(-6); }
		if (p.W == 0 || p.R == 0) break;
		for (int i = 0; i < rounds - first; i++) if (first + i > 0 && !h[i]) { settled = true; rounds = first + i + 1; break; }
		if (!settled && (int64_t)rounds > nchunks + 2 * SEL_BATCH) { err = "the selection did not settle"; return cleanup(-6); }
	}
	mark();
	// C
	unsigned long long hviol[2] = { 0, 0 };
	int herr = 0;
	if (n) hipLaunchKernelGGL(k_enc_reclen, blocks(n, 256), dim3(256), 0, st, p, d_rowptr, cost, best, n, reclen, flags);
	launch_scan(reclen, n, out.offsets, sums, st);
	int64_t totalBits = 0;
	if (hipMemcpyAsync(&totalBits, out.offsets + n, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&herr, flags, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
	    hipMemcpyAsync(hviol, viol, sizeof hviol, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "the compressor kernels failed"; return cleanup(-6); }
	if (hviol[0] != hviol[1]) { err = "successor lists must be strictly increasing"; return cleanup(-1); }
	if (herr) { err = "a record of 2^31 bits or more"; return cleanup(-3); }
	mark();
	// D
	out.graph_bits = (uint64_t)totalBits;
	const size_t gw = (size_t)((totalBits + 31) / 32) + 8;
	if (!alloc((void **)&out.graph_words, gw * 4)) { err = "device allocation failed"; return cleanup(-5); }
	(void)hipMemsetAsync(out.graph_words, 0, gw * 4, st);
	if (n) hipLaunchKernelGGL(k_enc_emit, blocks(n, 256), dim3(256), 0, st, p, d_rowptr, d_succ, best, refc, out.offsets, n, out.graph_words, dstats);
	mark();
	// E
	hipLaunchKernelGGL(k_enc_offlen, blocks((int64_t)n + 1, 256), dim3(256), 0, st, p, reclen, n, offlen);
	launch_scan(offlen, (int64_t)n + 1, offat, sums, st);
	int64_t offBits = 0;
	if (hipMemcpyAsync(&offBits, offat + n + 1, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "the offsets kernels failed"; return cleanup(-6); }
	out.off_bits = (uint64_t)offBits;
	const size_t ow = (size_t)((offBits + 31) / 32) + 8;
	if (!alloc((void **)&out.off_words, ow * 4)) { err = "device allocation failed"; return cleanup(-5); }
	(void)hipMemsetAsync(out.off_words, 0, ow * 4, st);
	hipLaunchKernelGGL(k_enc_offemit, blocks((int64_t)n + 1, 256), dim3(256), 0, st, p, reclen, offat, n, out.off_words);
	EncStatsDev hs{};
	if (hipMemcpyAsync(&hs, dstats, sizeof hs, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { err = "the emission kernels failed"; return cleanup(-6); }
	mark();
	out.bits_outdegrees = hs.v[0]; out.bits_references = hs.v[1]; out.bits_blocks = hs.v[2]; out.bits_intervals = hs.v[3]; out.bits_residuals = hs.v[4];
	out.copied_arcs = hs.v[5]; out.intervalised_arcs = hs.v[6]; out.residual_arcs = hs.v[7]; out.tot_ref = hs.v[8]; out.tot_dist = hs.v[9];
	out.max_ref_chain = (int32_t)hs.v[10];
	out.rounds = rounds;
	if (trace && ev.size() == 6) {
		static const char *names[] = { "A cost", "B select", "C lengths+scan", "D emit", "E offsets" };
		float total = 0;
		for (int i = 0; i < 5; i++) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[(size_t)i], ev[(size_t)i + 1]); total += ms; fprintf(stderr, "[bvgpu enc] %-16s %8.3f ms\n", names[i], ms); }
		fprintf(stderr, "[bvgpu enc] total %.3f ms, %d selection rounds, %llu bits\n", total, rounds, (unsigned long long)totalBits);
	}
	return cleanup(0);
}

} // namespace bv
