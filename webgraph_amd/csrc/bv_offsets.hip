// bv_offsets.hip -- the .offsets file decoded on the GPU (gfx950): n+1 gamma-coded gaps -> int64 bit offsets.
//
// Replaces, at load time, the sequential loop the reference calls "long and tedious" (OffsetsLongIterator,
// BVG:907-935; the offsets list built at BVG:1577-1601): the file is ONE stream of universal codes, so its decode
// is the cooperative decode of bv_coop.hpp taken to the whole grid.  The stream is cut into chunks of 64 x 256
// bits with FIXED nominal boundaries; a wave per chunk stages its chunk in LDS and finds the codeword boundaries
// by speculation with run-in (spec_tile), starting from a guess of the chunk's first boundary.  Chunk c's true
// first boundary is the end of chunk c-1's last codeword: the kernel is simply run again, every chunk taking its
// predecessor's end from the previous round and parsing again only if that moved its start -- universal codes
// re-synchronise within a few codewords, so round 1 repairs a handful of chunks and round 2 finds nothing to do.
// Two scans over the chunks (codes, gap sums) and a value pass then write off[i] = sum of the first i+1 gaps.
// gamma-coded offsets (the default, A.4 of SURVEY.md) and delta-coded ones (OFFSETS_DELTA): the codeword decoder is a template
// argument (KIND 2 / 3 of bv_coop.hpp), the speculation is the same -- both codes are self-delimiting and re-synchronise.
#include "bv_coop.hpp"
#include "bv_launch.hpp"

namespace bv {

constexpr int OFF_SEG = 256, OFF_CHUNK = 64 * OFF_SEG, OFF_RUNIN = 128, OFF_STAGE_B = 288; // bits; staging covers run-in + chunk + look-ahead

struct OffLane { uint32_t s, c; int64_t v; }; // per lane of a chunk: first owned code (bits from the chunk's nominal start), codes, sum of gaps

__device__ __forceinline__ GraphDev offsets_stream(const uint32_t *words, uint64_t nwords) {
	GraphDev g{};
	g.bits = words; g.nwords = nwords; g.offsets = nullptr; g.stats = nullptr; g.dbg = 0;
	return g;
}

// round 0: every chunk guesses its start by run-in; round r > 0: start = end of the previous chunk in round r-1
template <int KIND>
__global__ void __launch_bounds__(64) k_off_parse(const uint32_t *__restrict__ words, uint64_t nwords, uint64_t startBit, uint64_t totalBits, int64_t nchunks, int round,
                                                  const uint64_t *__restrict__ endPrev, uint64_t *__restrict__ endNew, uint64_t *__restrict__ startUsed,
                                                  uint32_t *__restrict__ cnt, int64_t *__restrict__ gapsum, OffLane *__restrict__ lanes, int *__restrict__ changed) {
	__shared__ __attribute__((aligned(16))) uint32_t lds[CoopLds<1>::WORDS];
	const GraphDev g = offsets_stream(words, nwords);
	Grp<1> G{ (int64_t *)(lds + CoopLds<1>::OFF_XCH) };
	const int64_t c = blockIdx.x;
	if (c >= nchunks) return;
	const uint64_t anchor = startBit + (uint64_t)c * OFF_CHUNK; // (startBit is a true codeword boundary)
	uint64_t start = startBit;
	if (c > 0 && round > 0) {
		start = endPrev[c - 1];
		if (start == startUsed[c]) { if (threadIdx.x == 0) endNew[c] = endPrev[c]; return; } // nothing moved
	}
	const uint64_t posW = anchor >= OFF_RUNIN ? anchor - OFF_RUNIN : 0;
	const WindowSrc src = stage_tile<1>(G, g, lds + CoopLds<1>::OFF_WIN, posW, OFF_STAGE_B);
	if (c > 0 && round == 0) { // guess: the first boundary at or after the nominal start, parsing from a little before it
		const uint64_t base = src.w0 << 5;
		uint32_t p = (uint32_t)(posW - base);
		const uint32_t a0 = (uint32_t)(anchor - base);
		int err = 0;
		while (p < a0 && !err) (void)win_code_rel<1, KIND>(g, src, p, err);
		start = err ? anchor : base + p;
	}
	// a start that is not in [anchor, anchor + 64 + OFF_SEG) cannot come from a valid stream: keep the lanes in range
	if (start < anchor) start = anchor;
	if (start > anchor + OFF_SEG) start = anchor + OFF_SEG;
	uint32_t s, n; int64_t sum; uint64_t E;
	spec_tile<1, KIND, 1>(G, g, src, start, totalBits, OFF_SEG, false, INT64_MAX, s, n, sum, E, anchor);
	int64_t ntot, stot;
	(void)G.incl_scan((int64_t)n, ntot);
	(void)G.incl_scan(sum, stot);
	lanes[c * 64 + threadIdx.x] = OffLane{ (uint32_t)(((src.w0 << 5) + s) - anchor), n, sum };
	if (threadIdx.x == 0) {
		const bool moved = round == 0 || E != endPrev[c];
		startUsed[c] = start; endNew[c] = E; cnt[c] = (uint32_t)ntot; gapsum[c] = stot;
		if (moved && round > 0) atomicOr(changed, 1);
	}
}

// exclusive scans over the chunks: code index and gap sum at the start of every chunk; total number of codes
__global__ void __launch_bounds__(1024) k_off_scan(const uint32_t *__restrict__ cnt, const int64_t *__restrict__ gapsum, int64_t nchunks,
                                                   int64_t *__restrict__ cntBase, int64_t *__restrict__ sumBase, int64_t *__restrict__ total) {
	__shared__ int64_t s_c[1024], s_s[1024];
	int64_t carryC = 0, carryS = 0;
	for (int64_t b = 0; b < nchunks; b += 1024) {
		const int64_t i = b + threadIdx.x;
		const int64_t vc = i < nchunks ? (int64_t)cnt[i] : 0, vs = i < nchunks ? gapsum[i] : 0;
		s_c[threadIdx.x] = vc; s_s[threadIdx.x] = vs;
		__syncthreads();
		for (int o = 1; o < 1024; o <<= 1) { // Hillis-Steele: a few thousand chunks, once per load
			const int64_t tc = threadIdx.x >= o ? s_c[threadIdx.x - o] : 0, ts = threadIdx.x >= o ? s_s[threadIdx.x - o] : 0;
			__syncthreads();
			s_c[threadIdx.x] += tc; s_s[threadIdx.x] += ts;
			__syncthreads();
		}
		if (i < nchunks) { cntBase[i] = carryC + s_c[threadIdx.x] - vc; sumBase[i] = carryS + s_s[threadIdx.x] - vs; }
		carryC += s_c[1023]; carryS += s_s[1023];
		__syncthreads();
	}
	if (threadIdx.x == 0) *total = carryC;
}

// value pass: every lane decodes the codes it owns again and writes the running sums
template <bool PREFIX, class T, int KIND>
__global__ void __launch_bounds__(64) k_off_values(const uint32_t *__restrict__ words, uint64_t nwords, uint64_t startBit, int64_t nchunks, const OffLane *__restrict__ lanes,
                                                   const int64_t *__restrict__ cntBase, const int64_t *__restrict__ sumBase, int64_t nOut, T *__restrict__ out) {
	__shared__ __attribute__((aligned(16))) uint32_t lds[CoopLds<1>::WORDS];
	const GraphDev g = offsets_stream(words, nwords);
	Grp<1> G{ (int64_t *)(lds + CoopLds<1>::OFF_XCH) };
	const int64_t c = blockIdx.x;
	if (c >= nchunks) return;
	const uint64_t anchor = startBit + (uint64_t)c * OFF_CHUNK;
	const uint64_t posW = anchor >= OFF_RUNIN ? anchor - OFF_RUNIN : 0;
	const WindowSrc src = stage_tile<1>(G, g, lds + CoopLds<1>::OFF_WIN, posW, OFF_STAGE_B);
	const OffLane me = lanes[c * 64 + threadIdx.x];
	int64_t tot;
	const int64_t cincl = G.incl_scan((int64_t)me.c, tot), vincl = G.incl_scan(me.v, tot);
	int64_t idx = cntBase[c] + cincl - me.c, acc = sumBase[c] + vincl - me.v;
	uint32_t p = (uint32_t)(anchor + me.s - (src.w0 << 5));
	int err = 0;
	for (uint32_t k = 0; k < me.c; k++) {
		const int64_t v = (int64_t)win_code_rel<1, KIND>(g, src, p, err);
		acc += v;
		if (idx < nOut) out[idx] = (T)(PREFIX ? acc : v); // offsets: running sum of the gaps; labels: the values themselves
		idx++;
	}
}

// fixed-width labels (FixedWidthIntLabel.java:70-73, readInt(width)): label a of the range sits at startBit + a * width
__global__ void __launch_bounds__(256) k_fixed_width(const uint32_t *__restrict__ words, uint64_t nwords, uint64_t startBit, int32_t width, int64_t count, int32_t *__restrict__ out) {
	const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (a >= count) return;
	const uint64_t pos = startBit + (uint64_t)a * (uint64_t)width;
	const uint64_t w = pos >> 5;
	const uint32_t sh = (uint32_t)pos & 31u;
	const uint64_t ab = ((uint64_t)__builtin_bswap32(words[w]) << 32) | __builtin_bswap32(words[w + 1]); // (the image ends with >= 8 zero words)
	out[a] = width == 0 ? 0 : (int32_t)(uint32_t)(((ab << sh) >> 32) >> (32u - (uint32_t)width));
}

// host side -------------------------------------------------------------------------------------------------
// d_words: the file's bytes as 32-bit words followed by >= 8 zero words; scratch is allocated and freed here
// (load-time code).  Returns 0, or -1 when the stream does not hold exactly `nodes + 1` codes (caller falls
// back to the host decoder, which produces the precise error).
// One contiguous stream of gamma codes in [startBit, endBit) holding exactly nOut codes: running sums -> int64 (offsets)
// or the values themselves -> int32 (gamma-coded labels).  startBit must be a codeword boundary.
template <bool PREFIX, class T, int KIND>
static int code_stream_decode(const uint32_t *d_words, uint64_t nwords, uint64_t startBit, uint64_t totalBits, int64_t nOut, T *d_out, hipStream_t st, int maxRounds) {
	if (nOut == 0) return totalBits == startBit ? 0 : -1;
	const int64_t nchunks = totalBits > startBit ? (int64_t)((totalBits - startBit + OFF_CHUNK - 1) / OFF_CHUNK) : 0;
	if (nchunks <= 0 || nchunks > 0x7fffffff) return -1;
	uint64_t *ends = nullptr, *startUsed = nullptr;
	uint32_t *cnt = nullptr;
	int64_t *gapsum = nullptr, *cntBase = nullptr, *sumBase = nullptr, *total = nullptr;
	OffLane *lanes = nullptr;
	int *changed = nullptr;
	int rc = -1;
	auto ok = [](hipError_t e) { return e == hipSuccess; };
	if (ok(hipMalloc((void **)&ends, sizeof(uint64_t) * 2 * nchunks)) && ok(hipMalloc((void **)&startUsed, sizeof(uint64_t) * nchunks)) &&
	    ok(hipMalloc((void **)&cnt, sizeof(uint32_t) * nchunks)) && ok(hipMalloc((void **)&gapsum, sizeof(int64_t) * nchunks)) &&
	    ok(hipMalloc((void **)&cntBase, sizeof(int64_t) * nchunks)) && ok(hipMalloc((void **)&sumBase, sizeof(int64_t) * nchunks)) &&
	    ok(hipMalloc((void **)&total, sizeof(int64_t))) && ok(hipMalloc((void **)&lanes, sizeof(OffLane) * 64 * nchunks)) && ok(hipMalloc((void **)&changed, sizeof(int)))) {
		int cur = 0;
		bool fine = true;
		// (a round only repeats while some chunk's end still moves: 2-3 rounds.  A stretch of equal codes never re-synchronises -- thousands of empty nodes in a row are
		// 010 010 010 ..., and a chain that starts one bit late reads 1, 00100, 1, 00100, ... for ever --, so the true boundaries advance one chunk per round there:
		// after maxRounds the caller's host decoder takes over -- 12 for .offsets: 64 rounds were 15 s of a 2^31-node load that the host decodes in 6; the label streams have no host
		// decoder behind them and keep 64)
		for (int round = 0; round < maxRounds && fine; round++) {
			fine = ok(hipMemsetAsync(changed, 0, sizeof(int), st));
			hipLaunchKernelGGL(k_off_parse<KIND>, dim3((unsigned)nchunks), dim3(64), 0, st, d_words, nwords, startBit, totalBits, nchunks, round, ends + (size_t)cur * nchunks,
			                   ends + (size_t)(cur ^ 1) * nchunks, startUsed, cnt, gapsum, lanes, changed);
			cur ^= 1;
			int h = 0;
			fine = fine && ok(hipMemcpyAsync(&h, changed, sizeof(int), hipMemcpyDeviceToHost, st)) && ok(hipStreamSynchronize(st));
			if (round > 0 && h == 0) break;
			if (round == maxRounds - 1) fine = false;
		}
		if (fine) {
			hipLaunchKernelGGL(k_off_scan, dim3(1), dim3(1024), 0, st, cnt, gapsum, nchunks, cntBase, sumBase, total);
			int64_t h = -1;
			if (ok(hipMemcpyAsync(&h, total, sizeof(int64_t), hipMemcpyDeviceToHost, st)) && ok(hipStreamSynchronize(st)) && h == nOut) {
				hipLaunchKernelGGL((k_off_values<PREFIX, T, KIND>), dim3((unsigned)nchunks), dim3(64), 0, st, d_words, nwords, startBit, nchunks, lanes, cntBase, sumBase, nOut, d_out);
				if (ok(hipStreamSynchronize(st))) rc = 0;
			}
		}
	}
	for (void *p : { (void *)ends, (void *)startUsed, (void *)cnt, (void *)gapsum, (void *)cntBase, (void *)sumBase, (void *)total, (void *)lanes, (void *)changed }) if (p) (void)hipFree(p);
	return rc;
}

int offsets_decode_device(const uint32_t *d_words, uint64_t nwords, uint64_t totalBits, int32_t nodes, int64_t *d_out, hipStream_t st, bool deltaCoded) {
	return deltaCoded ? code_stream_decode<true, int64_t, 3>(d_words, nwords, 0, totalBits, (int64_t)nodes + 1, d_out, st, 12)
	                  : code_stream_decode<true, int64_t, 2>(d_words, nwords, 0, totalBits, (int64_t)nodes + 1, d_out, st, 12);
}

// labels of `count` consecutive arcs, stored in [startBit, endBit) of the .labels stream (GammaCodedIntLabel.java:60-64)
int gamma_labels_decode_device(const uint32_t *d_words, uint64_t nwords, uint64_t startBit, uint64_t endBit, int64_t count, int32_t *d_out, hipStream_t st) {
	return code_stream_decode<false, int32_t, 2>(d_words, nwords, startBit, endBit, count, d_out, st, 64);
}

int fixed_labels_decode_device(const uint32_t *d_words, uint64_t nwords, uint64_t startBit, int32_t width, int64_t count, int32_t *d_out, hipStream_t st) {
	if (count <= 0) return 0;
	if (width < 0 || width > 32) return -1;
	hipLaunchKernelGGL(k_fixed_width, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, d_words, nwords, startBit, width, count, d_out);
	return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
}

// ---- FixedWidthIntListLabel (FixedWidthIntListLabel.java:107-112, fromBitStream): every arc carries a LIST, stored as
// gamma(length) followed by length x readInt(width).  The lengths make this a mixed stream, not a stream of universal
// codes, so the grid-wide speculation above does not apply; the label offsets give a true boundary per NODE, and a lane
// walks its node's lists from there: pass 1 counts lists and values per node, two scans place them, pass 2 writes.
template <bool FILL>
__global__ void __launch_bounds__(256) k_label_lists(const uint32_t *__restrict__ words, uint64_t nwords, const int64_t *__restrict__ off, int32_t from, int32_t cnt,
                                                     int32_t width, int32_t *__restrict__ nlists, int32_t *__restrict__ nvals, const int64_t *__restrict__ listBase,
                                                     const int64_t *__restrict__ valBase, int64_t *__restrict__ listptr, int32_t *__restrict__ values, int *__restrict__ err) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= cnt) return;
	const uint64_t end = (uint64_t)off[from + i + 1];
	uint64_t pos = (uint64_t)off[from + i];
	BitReader br;
	br.init(words, nwords);
	int64_t lists = 0, vals = 0;
	int64_t lb = 0, vb = 0;
	if (FILL) { lb = listBase[i]; vb = valBase[i]; }
	bool bad = false;
	while (pos < end) {
		br.seek(pos);
		const uint64_t len = br.gamma();
		const uint64_t p1 = br.pos();
		// the list must fit in what is left of the node's bits (also bounds `len` for width > 0)
		if (br.err || p1 > end || len > 0x7fffffffull || (width > 0 && len > (end - p1) / (uint64_t)width)) { bad = true; break; }
		if (FILL) {
			listptr[lb + lists] = vb + vals;
			for (uint64_t j = 0; j < len; j++) values[vb + vals + (int64_t)j] = (int32_t)br.bits((uint32_t)width);
		}
		lists++;
		vals += (int64_t)len;
		if (vals > 0x7fffffffll) { bad = true; break; }
		pos = p1 + len * (uint64_t)width;
	}
	if (bad || pos != end) { atomicOr(err, 1); lists = 0; vals = 0; }
	if (!FILL) { nlists[i] = (int32_t)lists; nvals[i] = (int32_t)vals; }
	else if (i == cnt - 1) listptr[listBase[cnt]] = valBase[cnt];
}

// 0 ok; -1 malformed / not `arcs` lists; -2 more than valuesCap values (*nvalues tells how many); -3 HIP / memory
int label_lists_decode_device(const uint32_t *d_words, uint64_t nwords, const int64_t *d_off, int32_t from, int32_t cnt, int32_t width, uint64_t arcs,
                              int64_t *d_listptr, int32_t *d_values, uint64_t valuesCap, uint64_t *nvalues, hipStream_t st) {
	if (nvalues) *nvalues = 0;
	if (cnt <= 0) return arcs == 0 ? 0 : -1;
	if (width < 0 || width > 32) return -1;
	const int64_t ns = scan_num_sums(cnt);
	int32_t *cnts = nullptr;
	int64_t *bases = nullptr, *sums = nullptr;
	int *err = nullptr;
	auto done = [&](int rc) { for (void *p : { (void *)cnts, (void *)bases, (void *)sums, (void *)err }) if (p) (void)hipFree(p); return rc; };
	if (hipMalloc((void **)&cnts, sizeof(int32_t) * 2 * (size_t)cnt) != hipSuccess || hipMalloc((void **)&bases, sizeof(int64_t) * 2 * ((size_t)cnt + 1)) != hipSuccess ||
	    hipMalloc((void **)&sums, sizeof(int64_t) * (size_t)(ns + 1)) != hipSuccess || hipMalloc((void **)&err, sizeof(int)) != hipSuccess) return done(-3);
	int32_t *nl = cnts, *nv = cnts + cnt;
	int64_t *lb = bases, *vb = bases + cnt + 1;
	(void)hipMemsetAsync(err, 0, sizeof(int), st);
	const dim3 grid((unsigned)(((int64_t)cnt + 255) / 256));
	hipLaunchKernelGGL(k_label_lists<false>, grid, dim3(256), 0, st, d_words, nwords, d_off, from, cnt, width, nl, nv, nullptr, nullptr, nullptr, nullptr, err);
	launch_scan(nl, cnt, lb, sums, st);
	launch_scan(nv, cnt, vb, sums, st);
	int herr = 0;
	int64_t tot[2] = { 0, 0 };
	if (hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&tot[0], lb + cnt, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
	    hipMemcpyAsync(&tot[1], vb + cnt, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return done(-3);
	if (herr || (uint64_t)tot[0] != arcs) return done(-1);
	if (nvalues) *nvalues = (uint64_t)tot[1];
	if ((uint64_t)tot[1] > valuesCap) return done(-2);
	hipLaunchKernelGGL(k_label_lists<true>, grid, dim3(256), 0, st, d_words, nwords, d_off, from, cnt, width, nullptr, nullptr, lb, vb, d_listptr, d_values, err);
	if (hipStreamSynchronize(st) != hipSuccess) return done(-3);
	return done(0);
}

} // namespace bv
