// bv_ef.hip -- EFGraph, the reference's quasi-succinct second format, decoded on the GPU (gfx950).  SURVEY.md section 8 row f4.
//
// src/it/unimi/dsi/webgraph/EFGraph.java: the record of node x starts at bit offsets[x] of a stream of 64-bit words read
// from the low bit up (LongWordBitReader, :892-1033): gamma(outdegree) (:1024-1032), then the Elias-Fano representation of
// outdegree + 1 increasing values -- the successors and the terminator upperBound: forward pointers (skipped here: their
// number and width follow from the outdegree, :156-171), l lower bits per value, and the upper bits in negated unary, where
// value i sets bit (value >> l) + i (EliasFanoSuccessorReader, :1103-1145).  No record refers to another one: the scan is
// outdegrees (k_ef_outdeg) -> scan -> one pass that writes every list (k_ef_decode, k_ef_decode_wave), with nothing to wait
// for between nodes.  Successor i is  ((position of the i-th one) - i) << l | lower_i : a select in the upper bits, done per
// successor by popcounts for the short lists and by a prefix sum over the popcounts of 64 words for the long ones.
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
// gamma(outdegree) at bit `pos` (readGamma, EFGraph.java:1024-1032); `after`: the bit that follows it
__device__ __forceinline__ bool ef_outdegree(const EfDev &g, uint64_t pos, uint32_t &d, uint64_t &after) {
	// readUnary: zeros up to the next one (a gamma code of a valid outdegree has at most 31 of them)
	uint64_t i = pos >> 6;
	uint64_t w = ef_ld(g, i) & (~0ull << (pos & 63));
	if (w == 0) { w = ef_ld(g, ++i); if (w == 0) return false; }
	const uint64_t one = i * 64 + (uint64_t)__builtin_ctzll(w);
	const uint64_t msb = one - pos;
	if (msb > 31) return false;
	const uint64_t v = (ef_get(g, one + 1, (int)msb) | (1ull << msb)) - 1;
	if (v > g.ub) return false; // more successors than values below the bound
	after = one + 1 + msb;
	d = (uint32_t)v;
	return true;
}
// header of the record at bit `pos`: the outdegree, then the sizes that follow from it (EFGraph.java:145-171, :1110-1115)
__device__ __forceinline__ bool ef_header(const EfDev &g, uint64_t pos, EfRecord &r) {
	uint64_t after;
	if (!ef_outdegree(g, pos, r.d, after)) return false;
	const uint64_t len = (uint64_t)r.d + 1;
	const uint32_t q = (uint32_t)g.ub / (uint32_t)len; // (both below 2^31 + 1: a 32-bit division)
	r.l = q == 0 ? 0 : 31 - __builtin_clz(q);
	const uint64_t hi = g.ub >> r.l, x = len + hi;
	const int ps = x <= 2 ? (int)x - 1 : 64 - __builtin_clzll(x - 1); // Fast.ceilLog2
	r.lowerStart = after + (uint64_t)(ps < 0 ? 0 : ps) * (hi >> g.lq);
	r.upperStart = r.lowerStart + (uint64_t)r.l * len;
	return true;
}

// slot s <-> node nodes[s] (a batch) or lo + s (a range)
__global__ void __launch_bounds__(256) k_ef_outdeg(const EfDev g, const int32_t *__restrict__ nodes, int32_t lo, int64_t cnt, int32_t *__restrict__ outd, int *__restrict__ err) {
	const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
	uint32_t d = 0;
	if (s < cnt) {
		const int32_t x = nodes ? nodes[s] : (int32_t)(lo + s);
		uint64_t after;
		if (x < 0 || x >= g.n || !ef_outdegree(g, (uint64_t)g.offsets[x], d, after)) { d = 0; atomicOr(err, x < 0 || x >= g.n ? E_REF : E_FORMAT); }
		outd[s] = (int32_t)d;
	}
}

// ImmutableGraph.hashCode() (ImmutableGraph.java:757-770) without the lists ever being written: h -> 31 h + x, then 31 h + s for the
// successors from the last to the first, i.e. node x maps h to 31^(d+1) h + x 31^d + sum_i s_i 31^i (s_0 the smallest).  In HASH
// mode a decoded successor goes, times its power of 31, into its node's accumulator instead of into the CSR; the nodes' maps are
// composed in order per block (k_ef_decode) and the blocks' maps by k_ef_hash_fold.
__device__ __forceinline__ uint32_t pow31(uint32_t e) { uint32_t r = 1, b = 31; while (e) { if (e & 1) r *= b; b *= b; e >>= 1; } return r; }
struct Affine { uint32_t m, b; }; // h -> m h + b
__device__ __forceinline__ Affine then(Affine first, Affine second) { return Affine{ second.m * first.m, second.m * first.b + second.b }; }
__device__ __forceinline__ Affine wave_compose(Affine v, int lane) { // inclusive scan: lane l holds the composition of lanes 0 .. l, in order
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t pm = __shfl_up(v.m, o), pb = __shfl_up(v.b, o);
		if (lane >= o) v = then(Affine{ pm, pb }, v);
	}
	return v;
}

// Lists of fewer than bigMin successors, 256 slots per block.  Phase 1: one lane per slot reads the header of its record (l,
// where the lower and the upper bits start) into LDS, a block scan numbers the successors of the tile.  Phase 2: one lane per
// SUCCESSOR -- its slot by a search in the tile's prefix sums, the position of its one in the upper bits by popcounts (a select
// within a few words: a short list has about two upper bits per successor), its lower bits by one extraction -- so that
// neighbouring lanes read neighbouring bits and write neighbouring ids, whatever the lengths of the lists are.
constexpr int EF_TILE = 256;
// HASH: `succ` is the array of per-slot sums (the short lists' are written here, the long-list kernels add theirs)
template <bool HASH>
__global__ void __launch_bounds__(EF_TILE) k_ef_decode(const EfDev g, const int32_t *__restrict__ nodes, int32_t lo, int64_t cnt, int32_t bigMin, const int64_t *__restrict__ rowstart,
                                                       int32_t *__restrict__ succ, uint64_t cap, int *__restrict__ err) {
	__shared__ uint64_t s_lower[EF_TILE], s_upper[EF_TILE];
	__shared__ int64_t s_row[EF_TILE];
	__shared__ int32_t s_first[EF_TILE + 1]; // successors of the tile's short lists before slot t
	__shared__ int32_t s_l[EF_TILE], s_wsum[EF_TILE / 64];
	__shared__ uint32_t s_acc[HASH ? EF_TILE : 1], s_pow[HASH ? EF_TILE : 1]; // HASH: the slots' sums; 31^i for i < 256 (a short list has fewer successors)
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	if (HASH) { s_acc[t] = 0; s_pow[t] = pow31((uint32_t)t); }
	const int64_t s = (int64_t)blockIdx.x * EF_TILE + t;
	int32_t d = 0;
	if (s < cnt) {
		const int64_t base = rowstart[s];
		d = (int32_t)(rowstart[s + 1] - base);
		if (d >= bigMin) d = 0; // a wave's
		else if (d > 0) {
			if ((uint64_t)(base + d) > cap) { atomicOr(err, E_CAP); d = 0; }
			else {
				EfRecord r;
				const int32_t x = nodes ? nodes[s] : (int32_t)(lo + s);
				if (!ef_header(g, (uint64_t)g.offsets[x], r) || (int32_t)r.d != d) { atomicOr(err, E_FORMAT); d = 0; }
				else { s_lower[t] = r.lowerStart; s_upper[t] = r.upperStart; s_l[t] = r.l; s_row[t] = base; }
			}
		}
	}
	// exclusive scan of d over the block
	int32_t inc = d;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const int32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
	if (lane == 63) s_wsum[wv] = inc;
	__syncthreads();
	int32_t before = 0;
	for (int w = 0; w < wv; w++) before += s_wsum[w];
	s_first[t] = before + inc - d;
	if (t == EF_TILE - 1) s_first[EF_TILE] = before + inc;
	__syncthreads();
	const int32_t total = s_first[EF_TILE];
	for (int32_t k0 = 0; k0 < total; k0 += EF_TILE) { // (uniform trip count: the HASH reduction shuffles)
		const int32_t k = k0 + t;
		uint32_t hv = 0;
		int ha = -1;
		if (k < total) {
		int a = 0, b = EF_TILE; // last slot with s_first <= k (slots without short lists repeat their successor's value: the last of them is the one)
#pragma unroll
		for (int step = 0; step < 8; step++) { const int mid = (a + b) >> 1; if (s_first[mid] <= k) a = mid; else b = mid; }
		const uint32_t i = (uint32_t)(k - s_first[a]);
		const uint64_t up = s_upper[a];
		const int l = s_l[a];
		// select: the i-th one at or after bit `up`
		uint64_t wi = up >> 6;
		uint64_t w = ef_ld(g, wi) & (~0ull << (up & 63));
		uint32_t r = i;
		bool bad = false;
		for (uint32_t c = (uint32_t)__popcll(w); r >= c; c = (uint32_t)__popcll(w)) { r -= c; if (++wi >= g.nwords) { bad = true; break; } w = ef_ld(g, wi); }
		if (bad) atomicOr(err, E_FORMAT);
		else {
		uint32_t bit = 0;
#pragma unroll
		for (int sh = 32; sh > 0; sh >>= 1) { const uint32_t c = (uint32_t)__popcll(w & ((1ull << sh) - 1)); if (r >= c) { r -= c; w >>= sh; bit += sh; } }
		const uint64_t high = wi * 64 + bit - up - i;
		const uint32_t val = (uint32_t)((high << l) | ef_get(g, s_lower[a] + (uint64_t)i * (uint64_t)l, l));
		if (!HASH) succ[s_row[a] + i] = (int32_t)val;
		else hv = val * (i < (uint32_t)EF_TILE ? s_pow[i] : pow31(i)), ha = a;
		}
		}
		if (HASH) { // neighbouring lanes hold neighbouring successors: the lanes of one list add up among themselves, its first lane adds to the slot
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const uint32_t v2 = __shfl_down(hv, o); const int a2 = __shfl_down(ha, o); if (lane + o < 64 && a2 == ha) hv += v2; }
			const int ap = __shfl_up(ha, 1);
			if (ha >= 0 && (lane == 0 || ap != ha)) atomicAdd(&s_acc[ha], hv);
		}
	}
	if (HASH) { // the sums of the short lists: the long-list kernels add theirs to the same array
		__syncthreads();
		if (s < cnt && (int32_t)(rowstart[s + 1] - rowstart[s]) < bigMin) ((uint32_t *)succ)[s] = s_acc[t];
	}
}
// HASH: the nodes of a block of 256 slots, in order, as one map (after every sum is in)
__global__ void __launch_bounds__(EF_TILE) k_ef_hash_nodes(int32_t lo, int64_t cnt, const int64_t *__restrict__ rowstart, const uint32_t *__restrict__ acc, Affine *__restrict__ maps) {
	__shared__ Affine s_wmap[EF_TILE / 64];
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	const int64_t s = (int64_t)blockIdx.x * EF_TILE + t;
	Affine mine{ 1, 0 };
	if (s < cnt) {
		const uint32_t pd = pow31((uint32_t)(rowstart[s + 1] - rowstart[s]));
		mine = Affine{ pd * 31u, (uint32_t)(lo + s) * pd + acc[s] };
	}
	const Affine inc = wave_compose(mine, lane);
	if (lane == 63) s_wmap[wv] = inc;
	__syncthreads();
	if (t == 0) { Affine all = s_wmap[0]; for (int w = 1; w < EF_TILE / 64; w++) all = then(all, s_wmap[w]); maps[blockIdx.x] = all; }
}
// the blocks' maps in order, applied to *h
__global__ void __launch_bounds__(256) k_ef_hash_fold(const Affine *__restrict__ maps, int64_t n, int32_t *__restrict__ h) {
	__shared__ Affine s_w[4];
	const int t = threadIdx.x, lane = t & 63;
	const int64_t per = (n + 255) / 256, a = (int64_t)t * per, e = a + per < n ? a + per : n;
	Affine mine{ 1, 0 };
	for (int64_t i = a; i < e; i++) mine = then(mine, maps[i]);
	const Affine inc = wave_compose(mine, lane);
	if (lane == 63) s_w[t >> 6] = inc;
	__syncthreads();
	if (t == 0) { Affine all = s_w[0]; for (int w = 1; w < 4; w++) all = then(all, s_w[w]); *h = (int32_t)(all.m * (uint32_t)*h + all.b); }
}

// ---- long lists.  A round takes 64 words of upper bits: a prefix sum over their popcounts gives every one its index; the lower
// bits of the round's successors are one contiguous stretch of the stream, staged in LDS so that the walk over a word's ones
// waits for nothing.  Lists below giantMin are decoded by one wave, round after round (k_ef_decode_wave).  A giant list (C2 has
// one of 3.5 * 10^5 successors: 240 rounds one after the other, 2.3 ms) is first measured -- k_ef_rank: popcounts only, the
// number of ones before every round -- which makes its rounds independent work items for as many waves (k_ef_decode_chunks).
constexpr int EF_LDS_WORDS = 512; // per wave: 32 768 lower bits
struct EfChunk { int32_t slot; uint32_t round; uint64_t rank; }; // round `round` of the list in slot `slot` starts with successor `rank`

__device__ __forceinline__ void ef_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }

// words [w0, w0 + 64) of the upper bits of record r, `done` successors before them; returns the ones in these words
// HASH: `succ` is the array of per-slot accumulators and `base` the slot
template <bool HASH>
__device__ __forceinline__ uint32_t ef_round(const EfDev &g, const EfRecord &r, int64_t base, uint32_t d, uint64_t w0, uint64_t done, uint64_t *low, int lane,
                                             int32_t *__restrict__ succ) {
	uint64_t w = ef_ld(g, w0 + lane);
	if (w0 + lane == (r.upperStart >> 6)) w &= ~0ull << (r.upperStart & 63);
	uint32_t inc = (uint32_t)__popcll(w);
	const uint32_t mine = inc;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
	const uint32_t total = __shfl(inc, 63);
	if (done >= d) return total;
	const uint64_t todo = d - done < total ? d - done : total; // successors of this round
	// their lower bits: [lb, lb + todo * l)
	const uint64_t lb = r.lowerStart + done * (uint64_t)r.l, lw0 = lb >> 6;
	const uint64_t nlw = r.l ? ((lb + todo * (uint64_t)r.l + 63) >> 6) - lw0 : 0;
	const bool staged = nlw <= EF_LDS_WORDS;
	if (staged) {
		ef_wave_sync(); // (the previous round's reads are over)
		for (uint64_t j = lane; j < nlw + 1; j += 64) low[j] = ef_ld(g, lw0 + j);
		ef_wave_sync();
	}
	uint64_t i = done + inc - mine; // index of this lane's first one
	const uint64_t bit0 = (w0 + lane) * 64 - r.upperStart;
	uint32_t hsum = 0, pw = HASH ? pow31((uint32_t)i) : 0; // a lane's successors are consecutive: 31^i runs along
	while (w && i < d) {
		const uint64_t high = bit0 + (uint64_t)__builtin_ctzll(w) - i;
		w &= w - 1;
		uint64_t lowv = 0;
		if (r.l) {
			const uint64_t p = r.lowerStart + i * (uint64_t)r.l;
			if (staged) {
				const uint64_t q = p - lw0 * 64, k = q >> 6;
				const int b = (int)(q & 63);
				lowv = low[k] >> b;
				if (b + r.l > 64) lowv |= low[k + 1] << (64 - b);
				lowv &= (1ull << r.l) - 1;
			} else lowv = ef_get(g, p, r.l);
		}
		const uint32_t val = (uint32_t)((high << r.l) | lowv);
		if (HASH) { hsum += val * pw; pw *= 31u; }
		else succ[base + i] = (int32_t)val;
		i++;
	}
	if (HASH && hsum) atomicAdd((uint32_t *)succ + base, hsum);
	return total;
}

// The long lists are found where they are: a wave looks at 64 slots at a time (their lengths are neighbouring words of
// rowstart) and takes the ones in its range -- no list of them is built (an atomic append per long list cost more than their
// decoding: 35 000 atomics on one counter, 0.5 ms).
template <bool HASH>
__global__ void __launch_bounds__(256) k_ef_decode_wave(const EfDev g, const int32_t *__restrict__ nodes, int32_t lo, int64_t cnt, int32_t bigMin, int32_t giantMin,
                                                        const int64_t *__restrict__ rowstart, int32_t *__restrict__ succ, uint64_t cap, int *__restrict__ err) {
	__shared__ uint64_t s_low[4][EF_LDS_WORDS + 2];
	const int lane = threadIdx.x & 63;
	uint64_t *low = s_low[threadIdx.x >> 6];
	const int64_t groups = (cnt + 63) / 64, stride = (int64_t)gridDim.x * 4;
	for (int64_t gi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); gi < groups; gi += stride) {
		const int64_t sl = gi * 64 + lane;
		const int64_t dl = sl < cnt ? rowstart[sl + 1] - rowstart[sl] : 0;
		for (uint64_t m = __ballot(dl >= bigMin && dl < giantMin); m; m &= m - 1) {
			const int64_t s = gi * 64 + __builtin_ctzll(m);
			const int64_t base = rowstart[s];
			const uint32_t d = (uint32_t)(rowstart[s + 1] - base);
			if ((uint64_t)(base + d) > cap) { if (lane == 0) atomicOr(err, E_CAP); continue; }
			const int32_t x = nodes ? nodes[s] : (int32_t)(lo + s);
			EfRecord r;
			if (!ef_header(g, (uint64_t)g.offsets[x], r) || r.d != d) { if (lane == 0) atomicOr(err, E_FORMAT); continue; }
			uint64_t done = 0;
			for (uint64_t w0 = r.upperStart >> 6; done < d; w0 += 64) {
				if (w0 >= g.nwords) { if (lane == 0) atomicOr(err, E_FORMAT); break; }
				done += ef_round<HASH>(g, r, HASH ? s : base, d, w0, done, low, lane, succ);
			}
		}
	}
}

// the giant lists: ones before every round -> work items.  chunks[] is a bump allocation: *nchunks slots are in use.
template <bool HASH>
__global__ void __launch_bounds__(256) k_ef_rank(const EfDev g, const int32_t *__restrict__ nodes, int32_t lo, int64_t cnt, int32_t giantMin, const int64_t *__restrict__ rowstart,
                                                 int32_t *__restrict__ succ, uint64_t cap, EfChunk *__restrict__ chunks, uint32_t chunkCap, uint32_t *__restrict__ nchunks,
                                                 int *__restrict__ err) {
	__shared__ uint64_t s_low[4][EF_LDS_WORDS + 2];
	uint64_t *low = s_low[threadIdx.x >> 6];
	const int lane = threadIdx.x & 63;
	const int64_t groups = (cnt + 63) / 64, stride = (int64_t)gridDim.x * 4;
	for (int64_t gi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); gi < groups; gi += stride) {
	const int64_t sl = gi * 64 + lane;
	const int64_t dl = sl < cnt ? rowstart[sl + 1] - rowstart[sl] : 0;
	for (uint64_t m = __ballot(dl >= giantMin); m; m &= m - 1) {
		const int64_t s = gi * 64 + __builtin_ctzll(m);
		const int64_t base = rowstart[s];
		const uint32_t d = (uint32_t)(rowstart[s + 1] - base);
		if ((uint64_t)(base + d) > cap) { if (lane == 0) atomicOr(err, E_CAP); continue; }
		const int32_t x = nodes ? nodes[s] : (int32_t)(lo + s);
		EfRecord r;
		if (!ef_header(g, (uint64_t)g.offsets[x], r) || r.d != d) { if (lane == 0) atomicOr(err, E_FORMAT); continue; }
		// the record ends with the terminator's one: its upper bits are the words up to offsets[x + 1]
		const uint64_t wFirst = r.upperStart >> 6, wEnd = ((uint64_t)g.offsets[x + 1] + 63) >> 6;
		if (wEnd <= wFirst || wEnd > g.nwords + 1) { if (lane == 0) atomicOr(err, E_FORMAT); continue; }
		const uint32_t rounds = (uint32_t)((wEnd - wFirst + 63) / 64);
		uint32_t at = 0;
		if (lane == 0) at = atomicAdd(nchunks, rounds);
		at = __shfl(at, 0);
		if ((uint64_t)at + rounds > chunkCap) { // no room (a batch that asks for the same giant list many times): this wave decodes it, round after round
			for (uint64_t j = (uint64_t)at + lane; j < chunkCap; j += 64) chunks[j].slot = -1;
			uint64_t done = 0;
			for (uint64_t w0 = wFirst; done < d && w0 < wEnd; w0 += 64) done += ef_round<HASH>(g, r, HASH ? s : base, d, w0, done, low, lane, succ);
			if (done < d && lane == 0) atomicOr(err, E_FORMAT);
			continue;
		}
		uint64_t rank = 0;
		for (uint32_t rd = 0; rd < rounds; rd += 4) { // four rounds of loads in flight
			uint32_t c[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const uint64_t wi = wFirst + (uint64_t)(rd + u) * 64 + lane;
				uint64_t w = rd + u < rounds && wi < wEnd ? ef_ld(g, wi) : 0;
				if (wi == wFirst) w &= ~0ull << (r.upperStart & 63);
				c[u] = (uint32_t)__popcll(w);
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) c[u] += __shfl_xor(c[u], o);
				if (rd + u < rounds) { if (lane == 0) chunks[at + rd + u] = EfChunk{ (int32_t)s, rd + u, rank }; rank += c[u]; }
			}
		}
		if (rank < d && lane == 0) atomicOr(err, E_FORMAT); // fewer ones than successors before the record ends: part of the row would stay unwritten
	}
}
}
template <bool HASH>
__global__ void __launch_bounds__(256) k_ef_decode_chunks(const EfDev g, const int32_t *__restrict__ nodes, int32_t lo, const EfChunk *__restrict__ chunks, const uint32_t *__restrict__ nchunks,
                                                          uint32_t chunkCap, const int64_t *__restrict__ rowstart, int32_t *__restrict__ succ, int *__restrict__ err) {
	__shared__ uint64_t s_low[4][EF_LDS_WORDS + 2];
	const int lane = threadIdx.x & 63;
	uint64_t *low = s_low[threadIdx.x >> 6];
	const int64_t n = *nchunks < chunkCap ? *nchunks : chunkCap, stride = (int64_t)gridDim.x * 4;
	for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < n; t += stride) {
		const EfChunk c = chunks[t];
		if (c.slot < 0) continue; // slots given up by a list that did not fit
		const int64_t base = rowstart[c.slot];
		const uint32_t d = (uint32_t)(rowstart[c.slot + 1] - base);
		if (c.rank >= d) continue; // nothing but the terminator (or padding) in this round
		const int32_t x = nodes ? nodes[c.slot] : (int32_t)(lo + c.slot);
		EfRecord r;
		if (!ef_header(g, (uint64_t)g.offsets[x], r) || r.d != d) { if (lane == 0) atomicOr(err, E_FORMAT); continue; }
		(void)ef_round<HASH>(g, r, HASH ? (int64_t)c.slot : base, d, (r.upperStart >> 6) + (uint64_t)c.round * 64, c.rank, low, lane, succ);
	}
}

void launch_ef_outdeg(const EfDev &g, const int32_t *nodes, int32_t lo, int64_t cnt, int32_t *outd, int *err, hipStream_t st) {
	if (cnt > 0) hipLaunchKernelGGL(k_ef_outdeg, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, g, nodes, lo, cnt, outd, err);
}
void launch_ef_decode(const EfDev &g, const int32_t *nodes, int32_t lo, int64_t cnt, int32_t bigMin, const int64_t *rowstart, int32_t *succ, uint64_t cap, int *err, int32_t giantMin,
                      void *chunks, uint32_t chunkCap, uint32_t *nchunks, hipStream_t st, hipStream_t stLong, hipStream_t stGiant) {
	if (cnt <= 0) return;
	hipLaunchKernelGGL(k_ef_rank<false>, dim3(256), dim3(256), 0, stGiant, g, nodes, lo, cnt, giantMin, rowstart, succ, cap, (EfChunk *)chunks, chunkCap, nchunks, err);
	hipLaunchKernelGGL(k_ef_decode_chunks<false>, dim3(1024), dim3(256), 0, stGiant, g, nodes, lo, (const EfChunk *)chunks, nchunks, chunkCap, rowstart, succ, err);
	hipLaunchKernelGGL(k_ef_decode_wave<false>, dim3(1024), dim3(256), 0, stLong, g, nodes, lo, cnt, bigMin, giantMin, rowstart, succ, cap, err);
	hipLaunchKernelGGL(k_ef_decode<false>, dim3((unsigned)((cnt + EF_TILE - 1) / EF_TILE)), dim3(EF_TILE), 0, st, g, nodes, lo, cnt, bigMin, rowstart, succ, cap, err);
}
// hashCode() of the nodes lo .. lo + cnt - 1 folded into *h (device): acc = uint32[cnt] zeroed, maps = ef_hash_blocks(cnt) * 8 bytes.  The three
// decode kernels run side by side (st, stLong, stGiant); the caller joins the streams before launch_ef_hash_fold
void launch_ef_hash(const EfDev &g, int32_t lo, int64_t cnt, int32_t bigMin, const int64_t *rowstart, uint32_t *acc, int *err, int32_t giantMin, void *chunks, uint32_t chunkCap,
                    uint32_t *nchunks, hipStream_t st, hipStream_t stLong, hipStream_t stGiant) {
	if (cnt <= 0) return;
	const uint64_t cap = ~0ull;
	hipLaunchKernelGGL(k_ef_rank<true>, dim3(256), dim3(256), 0, stGiant, g, (const int32_t *)nullptr, lo, cnt, giantMin, rowstart, (int32_t *)acc, cap, (EfChunk *)chunks, chunkCap, nchunks, err);
	hipLaunchKernelGGL(k_ef_decode_chunks<true>, dim3(1024), dim3(256), 0, stGiant, g, (const int32_t *)nullptr, lo, (const EfChunk *)chunks, nchunks, chunkCap, rowstart, (int32_t *)acc, err);
	hipLaunchKernelGGL(k_ef_decode_wave<true>, dim3(1024), dim3(256), 0, stLong, g, (const int32_t *)nullptr, lo, cnt, bigMin, giantMin, rowstart, (int32_t *)acc, cap, err);
	hipLaunchKernelGGL(k_ef_decode<true>, dim3((unsigned)((cnt + EF_TILE - 1) / EF_TILE)), dim3(EF_TILE), 0, st, g, (const int32_t *)nullptr, lo, cnt, bigMin, rowstart, (int32_t *)acc, cap, err);
}
void launch_ef_hash_fold(int32_t lo, int64_t cnt, const int64_t *rowstart, const uint32_t *acc, void *maps, int32_t *h, hipStream_t st) {
	if (cnt <= 0) return;
	const int64_t nb = (cnt + EF_TILE - 1) / EF_TILE;
	hipLaunchKernelGGL(k_ef_hash_nodes, dim3((unsigned)nb), dim3(EF_TILE), 0, st, lo, cnt, rowstart, acc, (Affine *)maps);
	hipLaunchKernelGGL(k_ef_hash_fold, dim3(1), dim3(256), 0, st, (const Affine *)maps, nb, h);
}
int64_t ef_hash_blocks(int64_t cnt) { return (cnt + EF_TILE - 1) / EF_TILE; }
size_t ef_chunk_bytes() { return sizeof(EfChunk); }

} // namespace bv
