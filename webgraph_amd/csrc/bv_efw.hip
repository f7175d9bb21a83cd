// bv_efw.hip -- EFGraph.store on the GPU (gfx950): a CSR in HBM -> the quasi-succinct .graph stream and its record lengths.
//
// EFGraph.store (src/it/unimi/dsi/webgraph/EFGraph.java:812-889) with its Accumulator (:420-552): per node gamma(outdegree),
// then -- for the successors followed by the terminator upperBound -- forward pointers, lower bits, upper bits, all taken from
// the LOW end of 64-bit words (LongWordOutputBitStream, :298-418).  Every piece of a record has a position that follows from
// the outdegree and the value alone:
//   sizes        l, pointer width and count from the outdegree (:145-171) -> record length -> a scan gives every record's offset
//   value i      its l lower bits at lowerStart + i * l, its one at upperStart + (value >> l) + i          [one lane per arc]
//   pointer k    the bit after the (k * quantum)-th zero of the upper bits (:516-522) = k * quantum + the number of values
//                whose upper part is below k * quantum: a binary search in the list                          [one lane per pointer]
// so nothing is sequential.  Words are shared between neighbours: the stream starts zeroed and every piece is ORed in.
#include "bv_launch.hpp"

namespace bv {

struct EfwRec { int l, ps; uint64_t np, ptrStart, lowerStart, upperStart, bits; }; // positions relative to the record's first bit
__device__ __forceinline__ EfwRec efw_layout(uint64_t d, uint64_t ub, int lq) {
	EfwRec r;
	const uint64_t v = d + 1;                                  // gamma(d) = non-zero gamma of d + 1 (:398-410)
	const int msb = 63 - __builtin_clzll(v);
	const uint64_t len = d + 1;                                // the terminator counts (:494)
	const uint32_t q = (uint32_t)ub / (uint32_t)len;
	r.l = q == 0 ? 0 : 31 - __builtin_clz(q);
	const uint64_t hi = ub >> r.l, x = len + hi;
	const int ps = x <= 2 ? (int)x - 1 : 64 - __builtin_clzll(x - 1);
	r.ps = ps < 0 ? 0 : ps;
	r.np = hi >> lq;
	r.ptrStart = (uint64_t)(2 * msb + 1);
	r.lowerStart = r.ptrStart + (uint64_t)r.ps * r.np;
	r.upperStart = r.lowerStart + (uint64_t)r.l * len;
	r.bits = r.upperStart + hi + d + 1;                       // the terminator's one is the last bit: position (ub >> l) + d
	return r;
}
// ORs the low `width` bits of v in at bit `pos`.  `words` is the stream in HBM, or -- with w0 -- the image in LDS of its words from w0 on:
// a block whose records fit assembles them there and writes whole words afterwards
__device__ __forceinline__ void efw_put(unsigned long long *words, uint64_t pos, uint64_t v, int width, uint64_t w0 = 0) {
	if (width == 0) return;
	const uint64_t i = (pos >> 6) - w0;
	const int b = (int)(pos & 63);
	atomicOr(words + i, (unsigned long long)(v << b));
	if (b + width > 64) atomicOr(words + i + 1, (unsigned long long)(v >> (64 - b)));
}

__global__ void __launch_bounds__(256) k_efw_sizes(const int64_t *__restrict__ rowptr, int32_t n, uint64_t ub, int lq, int32_t *__restrict__ reclen, int *__restrict__ err) {
	const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (x >= n) return;
	const int64_t d = rowptr[x + 1] - rowptr[x];
	if (d < 0 || (uint64_t)d > ub) { atomicOr(err, 1); reclen[x] = 0; return; }
	const EfwRec r = efw_layout((uint64_t)d, ub, lq);
	if (r.bits > 0x7fffffffull) { atomicOr(err, 2); reclen[x] = 0; return; }
	reclen[x] = (int32_t)r.bits;
}

// 256 nodes per block: headers (gamma, terminator, pointers of the nodes that have few), then one lane per arc of the tile
constexpr int EFW_TILE = 256, EFW_LANE_PTRS = 32, EFW_IMG_WORDS = 4096; // 32 KB of LDS for the tile's image
__global__ void __launch_bounds__(EFW_TILE) k_efw_emit(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int32_t n, uint64_t ub, int lq, const int64_t *__restrict__ off,
                                                       unsigned long long *__restrict__ words, int *__restrict__ err) {
	__shared__ uint64_t s_lower[EFW_TILE], s_upper[EFW_TILE];
	__shared__ int64_t s_row[EFW_TILE + 1];
	__shared__ int32_t s_l[EFW_TILE];
	__shared__ unsigned long long s_img[EFW_IMG_WORDS + 1];
	const int t = threadIdx.x;
	const int64_t x0 = (int64_t)blockIdx.x * EFW_TILE, x = x0 + t;
	// the tile's records are one stretch of the stream: words [iw0, iw0 + inw)
	const int64_t xEnd = x0 + EFW_TILE < n ? x0 + EFW_TILE : n;
	const uint64_t iw0 = (uint64_t)off[x0] >> 6, inw = (((uint64_t)off[xEnd] + 63) >> 6) - iw0;
	const bool img = inw <= EFW_IMG_WORDS; // (wave-uniform)
	unsigned long long *const out = img ? s_img : words;
	const uint64_t ow0 = img ? iw0 : 0;
	if (img) { for (uint64_t j = t; j <= inw; j += EFW_TILE) s_img[j] = 0; __syncthreads(); }
	if (x <= n) s_row[t] = rowptr[x];
	if (t == 0) s_row[EFW_TILE] = rowptr[x0 + EFW_TILE < n ? x0 + EFW_TILE : n];
	if (x < n) {
		const int64_t a = rowptr[x];
		const uint64_t d = (uint64_t)(rowptr[x + 1] - a);
		const EfwRec r = efw_layout(d, ub, lq);
		const uint64_t p = (uint64_t)off[x];
		// gamma(d): the unary part 1 << msb on msb + 1 bits, then the msb low bits of d + 1
		const uint64_t v = d + 1;
		const int msb = 63 - __builtin_clzll(v);
		efw_put(out, p, 1ull << msb, msb + 1, ow0);
		efw_put(out, p + msb + 1, v ^ (1ull << msb), msb, ow0);
		// the terminator: value ub at index d
		efw_put(out, p + r.lowerStart + d * (uint64_t)r.l, r.l ? ub & ((1ull << r.l) - 1) : 0, r.l, ow0);
		efw_put(out, p + r.upperStart + (ub >> r.l) + d, 1, 1, ow0);
		s_lower[t] = p + r.lowerStart; s_upper[t] = p + r.upperStart; s_l[t] = r.l;
		if (r.np <= EFW_LANE_PTRS) { // pointer k: k * quantum + the values whose upper part is below k * quantum
			for (uint64_t k = 1; k <= r.np; k++) {
				const uint64_t z = k << lq;
				uint64_t lo = 0, hi = d; // first index with (succ >> l) >= z
				while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (((uint64_t)(uint32_t)succ[a + mid] >> r.l) < z) lo = mid + 1; else hi = mid; }
				efw_put(out, p + r.ptrStart + (k - 1) * (uint64_t)r.ps, z + lo, r.ps, ow0);
			}
		}
	}
	__syncthreads();
	const int cntNodes = (int)(x0 + EFW_TILE < n ? EFW_TILE : n - x0);
	const int64_t a0 = s_row[0], a1 = s_row[cntNodes < EFW_TILE ? cntNodes : EFW_TILE];
	for (int64_t a = a0 + t; a < a1; a += EFW_TILE) {
		int lo = 0, hi = cntNodes; // last node of the tile whose row starts at or before a (empty rows repeat their successor's start: the last one owns a)
#pragma unroll
		for (int step = 0; step < 8; step++) { const int mid = (lo + hi) >> 1; if (lo < hi - 0 && s_row[mid] <= a) lo = mid; else hi = mid; }
		const uint64_t i = (uint64_t)(a - s_row[lo]);
		const int32_t sv = succ[a];
		if (sv < 0 || (uint64_t)sv >= ub || (i > 0 && sv <= succ[a - 1])) { atomicOr(err, 1); continue; } // strictly increasing, below the bound (:510-513)
		const int l = s_l[lo];
		efw_put(out, s_lower[lo] + i * (uint64_t)l, l ? (uint64_t)sv & ((1ull << l) - 1) : 0, l, ow0);
		efw_put(out, s_upper[lo] + ((uint64_t)sv >> l) + i, 1, 1, ow0);
	}
	if (img) { // whole words go out as they are; the first and the last are shared with the neighbouring tiles
		__syncthreads();
		for (uint64_t j2 = t; j2 < inw; j2 += EFW_TILE) {
			const unsigned long long v = s_img[j2];
			if (j2 == 0 || j2 + 1 == inw) { if (v) atomicOr(words + iw0 + j2, v); }
			else words[iw0 + j2] = v;
		}
	}
}

// the pointers of the nodes that have many: one wave per node, found by ballots over the outdegrees
__global__ void __launch_bounds__(256) k_efw_pointers(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ succ, int32_t n, uint64_t ub, int lq, const int64_t *__restrict__ off,
                                                      unsigned long long *__restrict__ words) {
	const int lane = threadIdx.x & 63;
	const int64_t groups = ((int64_t)n + 63) / 64, stride = (int64_t)gridDim.x * 4;
	for (int64_t gi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); gi < groups; gi += stride) {
		const int64_t xl = gi * 64 + lane;
		bool many = false;
		if (xl < n) { const int64_t d = rowptr[xl + 1] - rowptr[xl]; many = d >= 0 && (uint64_t)d <= ub && efw_layout((uint64_t)d, ub, lq).np > EFW_LANE_PTRS; }
		for (uint64_t m = __ballot(many); m; m &= m - 1) {
			const int64_t x = gi * 64 + __builtin_ctzll(m);
			const int64_t a = rowptr[x];
			const uint64_t d = (uint64_t)(rowptr[x + 1] - a);
			const EfwRec r = efw_layout(d, ub, lq);
			const uint64_t p = (uint64_t)off[x];
			for (uint64_t k = 1 + lane; k <= r.np; k += 64) {
				const uint64_t z = k << lq;
				uint64_t lo = 0, hi = d;
				while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (((uint64_t)(uint32_t)succ[a + mid] >> r.l) < z) lo = mid + 1; else hi = mid; }
				efw_put(words, p + r.ptrStart + (k - 1) * (uint64_t)r.ps, z + lo, r.ps);
			}
		}
	}
}

// 0 ok; -1 bad lists (not strictly increasing / not below the bound); -3 a record of 2^31 bits; -5 memory; -6 HIP.
// *d_words_out: ceil(bits / 64) + 1 words (close() writes the last buffer whatever it holds, :412-417), host order; *d_reclen_out: int32[n]
int ef_encode_device(int32_t n, const int64_t *d_rowptr, const int32_t *d_succ, uint64_t ub, int lq, uint64_t **d_words_out, uint64_t *nwords_out, uint64_t *bits_out,
                     int32_t **d_reclen_out, int64_t **d_off_out, hipStream_t st) {
	*d_words_out = nullptr; *d_reclen_out = nullptr; *d_off_out = nullptr; *nwords_out = 0; *bits_out = 0;
	int32_t *reclen = nullptr;
	int64_t *off = nullptr, *sums = nullptr;
	int *err = nullptr;
	unsigned long long *words = nullptr;
	const size_t nn = (size_t)n + 1;
	auto done = [&](int rc) {
		for (void *q : { (void *)sums, (void *)err }) if (q) (void)hipFree(q);
		if (rc) { for (void *q : { (void *)reclen, (void *)off, (void *)words }) if (q) (void)hipFree(q); (void)hipGetLastError(); }
		else { *d_words_out = (uint64_t *)words; *d_reclen_out = reclen; *d_off_out = off; }
		return rc;
	};
	if (hipMalloc((void **)&reclen, sizeof(int32_t) * nn) != hipSuccess || hipMalloc((void **)&off, sizeof(int64_t) * nn) != hipSuccess ||
	    hipMalloc((void **)&sums, sizeof(int64_t) * (size_t)(scan_num_sums(n) + 1)) != hipSuccess || hipMalloc((void **)&err, sizeof(int)) != hipSuccess) return done(-5);
	(void)hipMemsetAsync(err, 0, sizeof(int), st);
	const dim3 gridN((unsigned)(((int64_t)n + 255) / 256 > 0 ? ((int64_t)n + 255) / 256 : 1));
	hipLaunchKernelGGL(k_efw_sizes, gridN, dim3(256), 0, st, d_rowptr, n, ub, lq, reclen, err);
	launch_scan(reclen, n, off, sums, st);
	int64_t bits = 0;
	int herr = 0;
	if (hipMemcpyAsync(&bits, off + n, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
	    hipStreamSynchronize(st) != hipSuccess) return done(-6);
	if (herr & 1) return done(-1);
	if (herr & 2) return done(-3);
	const uint64_t nw = (uint64_t)bits / 64 + 1;
	if (hipMalloc((void **)&words, (size_t)(nw + 2) * 8) != hipSuccess) return done(-5);
	(void)hipMemsetAsync(words, 0, (size_t)(nw + 2) * 8, st);
	if (n > 0) {
		hipLaunchKernelGGL(k_efw_emit, dim3((unsigned)(((int64_t)n + EFW_TILE - 1) / EFW_TILE)), dim3(EFW_TILE), 0, st, d_rowptr, d_succ, n, ub, lq, off, words, err);
		hipLaunchKernelGGL(k_efw_pointers, dim3(512), dim3(256), 0, st, d_rowptr, d_succ, n, ub, lq, off, words);
	}
	if (hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return done(-6);
	if (herr) return done(-1);
	*nwords_out = nw; *bits_out = (uint64_t)bits;
	return done(0);
}

} // namespace bv
