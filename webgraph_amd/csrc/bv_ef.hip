// bv_ef.hip -- EFGraph, the reference's quasi-succinct second format, decoded on the GPU (gfx950).  SURVEY.md section 8 row f4.
//
// src/it/unimi/dsi/webgraph/EFGraph.java: the record of node x starts at bit offsets[x] of a stream of 64-bit words read
// from the low bit up (LongWordBitReader, :892-1033): gamma(outdegree) (:1024-1032), then the Elias-Fano representation of
// outdegree + 1 increasing values -- the successors and the terminator upperBound: forward pointers (skipped here: their
// number and width follow from the outdegree, :156-171), l lower bits per value, and the upper bits in negated unary, where
// value i sets bit (value >> l) + i (EliasFanoSuccessorReader, :1103-1145).  No record refers to another one: the scan is
// outdegrees (k_ef_outdeg) -> scan -> one pass that writes every list (k_ef_decode, k_ef_decode_wave), with nothing to wait
// for between nodes.  Successor i is  ((position of the i-th one) - i) << l | lower_i : a select in the upper bits, which a
// lane does by walking the ones of its words and a wave by a prefix sum over the popcounts of 64 words.
#include "bv_launch.hpp"

namespace bv {

__device__ __forceinline__ uint64_t ef_ld(const EfDev &g, uint64_t i) { return i < g.nwords ? g.words[i] : 0ull; } // (the image is followed by zero words; a malformed offset may point anywhere)
// `width` bits (0..64) from bit `pos`, low bits first
__device__ __forceinline__ uint64_t ef_get(const EfDev &g, uint64_t pos, int width) {
	if (width == 0) return 0;
	const uint64_t i = pos >> 6;
	const int b = (int)(pos & 63);
	uint64_t v = ef_ld(g, i) >> b;
	if (b + width > 64) v |= ef_ld(g, i + 1) << (64 - b);
	return width == 64 ? v : v & ((1ull << width) - 1);
}
struct EfRecord { uint32_t d; int l; uint64_t lowerStart, upperStart; };
// header of the record at bit `pos`: gamma(outdegree), then the sizes that follow from it (EFGraph.java:145-171, :1110-1115)
__device__ __forceinline__ bool ef_header(const EfDev &g, uint64_t pos, EfRecord &r) {
	// readUnary: zeros up to the next one (a gamma code of a valid outdegree has at most 31 of them)
	uint64_t i = pos >> 6;
	uint64_t w = ef_ld(g, i) & (~0ull << (pos & 63));
	if (w == 0) { w = ef_ld(g, ++i); if (w == 0) return false; }
	const uint64_t one = i * 64 + (uint64_t)__builtin_ctzll(w);
	const uint64_t msb = one - pos;
	if (msb > 31) return false;
	const uint64_t v = (ef_get(g, one + 1, (int)msb) | (1ull << msb)) - 1;
	if (v > g.ub) return false; // more successors than values below the bound
	const uint64_t after = one + 1 + msb;
	r.d = (uint32_t)v;
	const uint64_t len = v + 1, q = g.ub / len;
	r.l = q == 0 ? 0 : 63 - __builtin_clzll(q);
	const uint64_t hi = g.ub >> r.l, x = len + hi;
	const int ps = x <= 2 ? (int)x - 1 : 64 - __builtin_clzll(x - 1); // Fast.ceilLog2
	r.lowerStart = after + (uint64_t)(ps < 0 ? 0 : ps) * (hi >> g.lq);
	r.upperStart = r.lowerStart + (uint64_t)r.l * len;
	return true;
}

// slot s <-> node nodes[s] (a batch) or lo + s (a range)
__global__ void __launch_bounds__(256) k_ef_outdeg(const EfDev g, const int32_t *__restrict__ nodes, int32_t lo, int64_t cnt, int32_t bigMin, int32_t *__restrict__ outd,
                                                   int32_t *__restrict__ biglist, int32_t *__restrict__ nbig, int *__restrict__ err) {
	const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (s >= cnt) return;
	const int32_t x = nodes ? nodes[s] : (int32_t)(lo + s);
	EfRecord r;
	if (x < 0 || x >= g.n || !ef_header(g, (uint64_t)g.offsets[x], r)) { outd[s] = 0; atomicOr(err, x < 0 || x >= g.n ? E_REF : E_FORMAT); return; }
	outd[s] = (int32_t)r.d;
	if (biglist && (int32_t)r.d >= bigMin) biglist[atomicAdd(nbig, 1)] = (int32_t)s;
}

// one lane per list of fewer than bigMin successors
__global__ void __launch_bounds__(256) k_ef_decode(const EfDev g, const int32_t *__restrict__ nodes, int32_t lo, int64_t cnt, int32_t bigMin, const int64_t *__restrict__ rowstart,
                                                   int32_t *__restrict__ succ, uint64_t cap, int *__restrict__ err) {
	const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (s >= cnt) return;
	const int64_t base = rowstart[s];
	const uint32_t d = (uint32_t)(rowstart[s + 1] - base);
	if (d == 0 || (int32_t)d >= bigMin) return;
	if ((uint64_t)(base + d) > cap) { atomicOr(err, E_CAP); return; }
	const int32_t x = nodes ? nodes[s] : (int32_t)(lo + s);
	EfRecord r;
	if (!ef_header(g, (uint64_t)g.offsets[x], r) || r.d != d) { atomicOr(err, E_FORMAT); return; }
	uint64_t wi = r.upperStart >> 6;
	uint64_t w = ef_ld(g, wi) & (~0ull << (r.upperStart & 63));
	uint64_t lp = r.lowerStart;
	for (uint32_t i = 0; i < d; i++) {
		while (w == 0) { if (++wi >= g.nwords) { atomicOr(err, E_FORMAT); return; } w = g.words[wi]; }
		const uint64_t high = wi * 64 + (uint64_t)__builtin_ctzll(w) - r.upperStart - i;
		w &= w - 1;
		succ[base + i] = (int32_t)((high << r.l) | ef_get(g, lp, r.l));
		lp += (uint64_t)r.l;
	}
}

// one wave per long list: 64 words of upper bits per round, a prefix sum over their popcounts gives every one its index
__global__ void __launch_bounds__(256) k_ef_decode_wave(const EfDev g, const int32_t *__restrict__ nodes, int32_t lo, const int32_t *__restrict__ biglist, const int32_t *__restrict__ nbig,
                                                        const int64_t *__restrict__ rowstart, int32_t *__restrict__ succ, uint64_t cap, int *__restrict__ err) {
	const int lane = threadIdx.x & 63;
	const int64_t n = *nbig, stride = (int64_t)gridDim.x * 4;
	for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < n; t += stride) {
		const int64_t s = biglist[t];
		const int64_t base = rowstart[s];
		const uint32_t d = (uint32_t)(rowstart[s + 1] - base);
		if ((uint64_t)(base + d) > cap) { if (lane == 0) atomicOr(err, E_CAP); continue; }
		const int32_t x = nodes ? nodes[s] : (int32_t)(lo + s);
		EfRecord r;
		if (!ef_header(g, (uint64_t)g.offsets[x], r) || r.d != d) { if (lane == 0) atomicOr(err, E_FORMAT); continue; }
		uint64_t done = 0;
		for (uint64_t w0 = r.upperStart >> 6; done < d; w0 += 64) {
			if (w0 >= g.nwords) { if (lane == 0) atomicOr(err, E_FORMAT); break; }
			uint64_t w = ef_ld(g, w0 + lane);
			if (w0 + lane == (r.upperStart >> 6)) w &= ~0ull << (r.upperStart & 63);
			uint32_t inc = (uint32_t)__popcll(w);
			const uint32_t mine = inc;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
			uint64_t i = done + inc - mine; // index of this lane's first one
			const uint64_t bit0 = (w0 + lane) * 64 - r.upperStart;
			while (w && i < d) {
				const uint64_t high = bit0 + (uint64_t)__builtin_ctzll(w) - i;
				w &= w - 1;
				succ[base + i] = (int32_t)((high << r.l) | ef_get(g, r.lowerStart + i * (uint64_t)r.l, r.l));
				i++;
			}
			done += __shfl(inc, 63);
		}
	}
}

void launch_ef_outdeg(const EfDev &g, const int32_t *nodes, int32_t lo, int64_t cnt, int32_t bigMin, int32_t *outd, int32_t *biglist, int32_t *nbig, int *err, hipStream_t st) {
	if (cnt > 0) hipLaunchKernelGGL(k_ef_outdeg, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, g, nodes, lo, cnt, bigMin, outd, biglist, nbig, err);
}
void launch_ef_decode(const EfDev &g, const int32_t *nodes, int32_t lo, int64_t cnt, int32_t bigMin, const int32_t *biglist, const int32_t *nbig, const int64_t *rowstart, int32_t *succ,
                      uint64_t cap, int *err, hipStream_t st) {
	if (cnt <= 0) return;
	hipLaunchKernelGGL(k_ef_decode_wave, dim3(1024), dim3(256), 0, st, g, nodes, lo, biglist, nbig, rowstart, succ, cap, err);
	hipLaunchKernelGGL(k_ef_decode, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, g, nodes, lo, cnt, bigMin, rowstart, succ, cap, err);
}

} // namespace bv
