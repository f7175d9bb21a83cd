// bv_device.hpp -- device-side bit reader and universal-code decoders for the BVGraph bit stream (gfx950).
//
// The .graph image is staged in HBM byte-for-byte as it is on disk (MSB-first bit order)
// plus zero padding; kernels view it as 32-bit words and byte-swap on load (one v_perm_b32), so that the
// first stream bit of a word is its bit 31.  All readers are bounds-guarded: past the padded end they see
// zeros and raise a sticky error instead of touching memory.
//
// Code definitions: dsiutils InputBitStream (not in the reference repo; restated in SURVEY.md App. B, pinned
// through the cnr-2000 fixture for unary / gamma / zeta_3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bv {

typedef __attribute__((address_space(3))) uint32_t lds_u32; // a word in LDS: accesses through it are ds_* instructions, never flat ones

// CompressionFlags.java:26-44
enum : int { C_DELTA = 1, C_GAMMA = 2, C_GOLOMB = 3, C_SKEWED_GOLOMB = 4, C_UNARY = 5, C_ZETA = 6, C_NIBBLE = 7 };

// sticky device error bits
enum : int { E_REF = 1, E_FORMAT = 2, E_CAP = 4, E_UNSUP = 8, E_ESCAPED = 16, E_ARG = 32, E_HALO = 64 }; // E_HALO: a halo row past the scratch capacity (the host retries)

struct GraphDev {
	const uint32_t *bits;   // .graph bytes viewed as big-endian 32-bit words
	uint64_t nwords;        // valid words (file bytes rounded up to 4); the allocation has >= 4 more zero words
	const int64_t *offsets; // n+1 bit offsets
	int32_t n;
	int32_t W, minInt, zetaK;
	int32_t c_outd, c_ref, c_bc, c_blk, c_res;
	unsigned long long *stats; // optional tuning counters (BVGPU_STATS=1), NULL otherwise
	int32_t dbg;               // BVGPU_DBG: selects which tick counters BVGPU_STATS collects (bit 4: copy kernels); never changes results
	// The block tables of the giant records (outdegree >= walkMin), written by the parse kernel that walks their lists anyway and read
	// by k_copy_big instead of walking them again: walktab[walkCap] ints, bump-allocated through *walkCursor (NULL: not kept).
	// A record's tables are announced in the free head of its row: row[0] = -2, row[1] = offset, row[2] = copied blocks, row[3] = copied
	// ids; row[0] = -1: none (every giant record with a reference that copies something gets one or the other).
	int32_t *walktab;
	uint32_t walkCap;
	uint32_t *walkCursor;
	int32_t walkMin;
	// Hand-over of the long records' residual sections to the segment pipeline (bv_seg.hip): a cooperative kernel that serves queue
	// `which` (0: the wave class, 1: the giants) parses the structure of its idx-th record only and leaves a descriptor in
	// segDesc[segOff[which] + idx] (segNseg: how many pieces its residual section has).  NULL: it decodes the residuals itself.
	void *segDesc;
	int32_t *segNseg, *segFlag;
	int32_t segOff[2], segCap[2];
	int32_t segMinD; // only records with at least this many successors are handed over
};

// Timing switches (scripts/r6g.sh, profiles/r6_wave_class_ablation.txt): BVGPU_DBG bits that leave a step of a kernel OUT to see what it costs -- the results are garbage, so
// they exist only in -DBV_EXP_TIMING tuning builds; in the library every knob is a choice of speed, never of results.
#ifdef BV_EXP_TIMING
#define BV_TIMING(g, bit) (((g).dbg & (bit)) != 0)
#else
#define BV_TIMING(g, bit) false
#endif
// tuning counters: 0 tiles(residual) 1 rounds(residual) 2 tiles(interval) 3 rounds(interval) 4 lane-parses 5 big nodes 6 clock ticks in coop nodes 7 max ticks of one node
__device__ __forceinline__ void stat_add(const GraphDev &g, int i, unsigned long long v) { if (g.stats && (threadIdx.x & 63) == 0) atomicAdd(&g.stats[i], v); }
__device__ __forceinline__ void stat_max(const GraphDev &g, int i, unsigned long long v) { if (g.stats && (threadIdx.x & 63) == 0) atomicMax(&g.stats[i], v); }

// Word sources of the bit reader.  Words are indexed from the start of the .graph image and returned with
// the first stream bit in bit 31.
struct GlobalSrc { // straight from HBM (cached in L1/L2), one word at a time
	const uint32_t *__restrict__ w;
	uint64_t nwords;
	__device__ __forceinline__ void start(uint64_t) {}
	__device__ __forceinline__ uint32_t ld(uint64_t i) const { return i < nwords ? __builtin_bswap32(w[i]) : 0u; }
};
// Sequential reader over HBM with 16-byte loads and one vector always in flight: the words of the current
// vector are handed out from registers while the next vector is already on its way, so a lane streaming
// through a long record waits for memory once per 128 bits instead of once per 32, and mostly not at all.
// ld(i) must be called with consecutive i after start(i) -- exactly what BitReaderT does.
struct PrefetchSrc {
	const uint32_t *__restrict__ w; // image; the allocation is padded to a multiple of 16 bytes plus >= 8 zero words
	uint64_t nwords;
	uint64_t vnext;                 // next 16-byte vector to request
	uint4 pend;                     // the vector in flight
	uint4 cur;                      // the vector being handed out (already byte-swapped)
	int qn;                         // words of `cur` still to hand out
	__device__ __forceinline__ uint4 ld4(uint64_t vi) const {
		return vi * 4 < nwords + 4 ? ((const uint4 *)w)[vi] : uint4{ 0u, 0u, 0u, 0u };
	}
	__device__ __forceinline__ void take() {
		cur = uint4{ __builtin_bswap32(pend.x), __builtin_bswap32(pend.y), __builtin_bswap32(pend.z), __builtin_bswap32(pend.w) };
		qn = 4;
		pend = ld4(vnext++);
	}
	__device__ __forceinline__ void start(uint64_t i) {
		vnext = i >> 2;
		pend = ld4(vnext++);
		take();
		qn = 4 - (int)(i & 3);
	}
	__device__ __forceinline__ uint32_t ld(uint64_t) {
		if (qn == 0) take();
		const uint32_t r = qn == 4 ? cur.x : qn == 3 ? cur.y : qn == 2 ? cur.z : cur.w;
		qn--;
		return r;
	}
};
struct WindowSrc { // a window of the stream staged in LDS (already byte-swapped); reads outside fall back to HBM
	const uint32_t *win;
	uint64_t w0;
	uint32_t nw;
	GlobalSrc g;
	__device__ __forceinline__ void start(uint64_t) {}
	__device__ __forceinline__ uint32_t ld(uint64_t i) const { const uint64_t j = i - w0; return j < (uint64_t)nw ? win[j] : g.ld(i); }
};

// Register-buffered MSB-first reader: 64-bit window, refilled 32 bits at a time.
template <class Src> struct BitReaderT {
	Src src;
	uint64_t widx;  // next word to load
	uint64_t buf;   // valid bits are the top `nbits`; everything below is zero
	uint32_t nbits;
	int err;
	uint64_t nwords; // words of the image (end-of-stream detection)

	__device__ __forceinline__ uint32_t ld(uint64_t i) { return src.ld(i); }

	__device__ __forceinline__ void init(const uint32_t *words, uint64_t nw) { src = Src{ words, nw }; nwords = nw; err = 0; widx = 0; buf = 0; nbits = 0; }
	__device__ __forceinline__ void init_src(const Src &s_, uint64_t nw) { src = s_; nwords = nw; err = 0; widx = 0; buf = 0; nbits = 0; }

	__device__ __forceinline__ void seek(uint64_t pos) {
		widx = pos >> 5;
		src.start(widx);
		const uint32_t s = (uint32_t)pos & 31u;
		const uint64_t hi = ld(widx);
		const uint64_t lo = ld(widx + 1);
		widx += 2;
		buf = ((hi << 32) | lo) << s;
		nbits = 64u - s;
	}
	__device__ __forceinline__ uint64_t pos() const { return widx * 32u - nbits; }

	// after refill(): nbits >= 33
	__device__ __forceinline__ void refill() {
		if (nbits <= 32u) {
			buf |= (uint64_t)ld(widx) << (32u - nbits);
			widx++;
			nbits += 32u;
		}
	}
	// n in 0..32
	__device__ __forceinline__ uint32_t bits(uint32_t n) {
		refill();
		const uint32_t v = n ? (uint32_t)(buf >> (64u - n)) : 0u;
		buf <<= n;
		nbits -= n;
		return v;
	}
	// n in 0..64
	__device__ __forceinline__ uint64_t bits64(uint32_t n) {
		if (n <= 32u) return bits(n);
		const uint64_t hi = bits(n - 32u);
		return (hi << 32) | bits(32u);
	}
	// number of zeros before the first one; the one is consumed
	__device__ __forceinline__ uint64_t unary() {
		uint64_t z = 0;
		for (;;) {
			refill();
			if (buf) {
				const uint32_t c = (uint32_t)__clzll((long long)buf);
				z += c;
				buf = (buf << c) << 1;
				nbits -= c + 1u;
				return z;
			}
			z += nbits;
			nbits = 0;
			if (widx >= nwords + 2) { err |= E_FORMAT; return z; }
		}
	}
	__device__ __forceinline__ uint64_t gamma() {
		// fast path: the whole code (<= 31 bits, values < 2^15) sits in the top word of the window
		refill();
		const uint32_t hi = (uint32_t)(buf >> 32);
		if (hi >= (1u << 16)) {
			const uint32_t m = (uint32_t)__clz((int)hi);
			const uint32_t len = 2 * m + 1;
			const uint32_t v = (hi >> (32u - len)) - 1;
			buf <<= len; nbits -= len;
			return v;
		}
		const uint64_t m = unary();
		if (m > 63) { err |= E_FORMAT; return 0; }
		return (((uint64_t)1 << m) | bits64((uint32_t)m)) - 1;
	}
	__device__ __forceinline__ uint64_t delta() {
		const uint64_t m = gamma();
		if (m > 63) { err |= E_FORMAT; return 0; }
		return (((uint64_t)1 << m) | bits64((uint32_t)m)) - 1;
	}
	template <int K> __device__ __forceinline__ uint64_t zeta_k(int k) {
		const int kk = K > 0 ? K : k;
		if (K == 3) {
			// fast path for zeta_3: h <= 7, i.e. the whole code (<= 32 bits, values < 2^24 - 1) sits in the top word
			refill();
			const uint32_t hi = (uint32_t)(buf >> 32);
			if (hi >= (1u << 24)) {
				const uint32_t h = (uint32_t)__clz((int)hi);
				const uint32_t nb = 3 * h + 2;
				const uint32_t left = 1u << (3 * h);
				const uint32_t rest = hi << (h + 1);       // payload at the top
				const uint32_t m = rest >> (32u - nb);
				uint32_t v, len;
				if (m < left) { v = m + left - 1; len = h + 1 + nb; }
				else { v = ((m << 1) | ((rest >> (31u - nb)) & 1u)) - 1; len = h + 2 + nb; }
				buf <<= len; nbits -= len;
				return v;
			}
		}
		const uint64_t h = unary();
		const uint64_t nb = h * (uint64_t)kk + (uint64_t)kk - 1;
		if (nb > 63) { err |= E_FORMAT; return 0; }
		const uint32_t hk = (uint32_t)h * (uint32_t)kk;
		const uint64_t left = (uint64_t)1 << hk;
		const uint64_t m = bits64((uint32_t)nb);
		if (m < left) return m + left - 1;
		return ((m << 1) | bits(1)) - 1;
	}
	__device__ __forceinline__ uint64_t golomb(int b) {
		if (b == 0) return 0;
		const uint32_t l2 = 31u - (uint32_t)__clz(b);
		const uint64_t q = unary();
		const uint64_t mm = ((uint64_t)1 << (l2 + 1)) - (uint64_t)b;
		const uint64_t x = bits(l2);
		const uint64_t r = x < mm ? x : ((x << 1) | bits(1)) - mm;
		return q * (uint64_t)b + r;
	}
	__device__ __forceinline__ uint64_t nibble() {
		uint64_t x = 0;
		uint32_t stop;
		int guard = 0;
		do {
			x <<= 3;
			stop = bits(1);
			x |= bits(3);
		} while (!stop && ++guard < 22);
		if (!stop) err |= E_FORMAT;
		return x;
	}
	// run-time dispatch (wave-uniform); DEF folds it at compile time in the callers
	__device__ __forceinline__ uint64_t coded(int coding, int k) {
		switch (coding) {
		case C_GAMMA: return gamma();
		case C_DELTA: return delta();
		case C_UNARY: return unary();
		case C_ZETA: return zeta_k<0>(k);
		case C_GOLOMB: return golomb(k);
		case C_NIBBLE: return nibble();
		default: err |= E_UNSUP; return 0;
		}
	}
};

// Fast.nat2int
__device__ __forceinline__ int64_t nat2int(uint64_t v) { return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }

// One copy block (BVG:1063-1069) checked against what is left of the referent's row.  `total` = blocks so far, with the
// invariant 0 <= total <= dref.  A code of a malformed stream can be any 64-bit value: it is rejected here, before it
// reaches a sum (a wrapped sum would pass every later "total <= dref" test and send the copy pass out of bounds).
__device__ __forceinline__ bool block_len_ok(uint64_t code, bool first, int64_t total, int64_t dref, int64_t &len) {
	if (code > (uint64_t)(dref - total)) return false;
	len = (int64_t)code + (first ? 0 : 1);
	return total + len <= dref;
}

using BitReader = BitReaderT<GlobalSrc>;
using WinReader = BitReaderT<WindowSrc>;
using PReader = BitReaderT<PrefetchSrc>;

// Field readers.  DEF == true: the default coding set (gamma outdegrees / block counts / blocks, unary
// references, zeta_3 residuals -- BVG:525-541 and DEFAULT_ZETA_K) is resolved at compile time.
template <int DEF> struct Fields {
	template <class R> static __device__ __forceinline__ uint64_t outdegree(R &br, const GraphDev &g) { return DEF ? br.gamma() : br.coded(g.c_outd, 0); }
	template <class R> static __device__ __forceinline__ uint64_t reference(R &br, const GraphDev &g) { return DEF ? br.unary() : br.coded(g.c_ref, 0); }
	template <class R> static __device__ __forceinline__ uint64_t block_count(R &br, const GraphDev &g) { return DEF ? br.gamma() : br.coded(g.c_bc, 0); }
	template <class R> static __device__ __forceinline__ uint64_t block(R &br, const GraphDev &g) { return DEF ? br.gamma() : br.coded(g.c_blk, 0); }
	template <class R> static __device__ __forceinline__ uint64_t residual(R &br, const GraphDev &g) { return DEF == 1 ? br.template zeta_k<3>(3) : DEF == 2 ? br.template zeta_k<0>(g.zetaK) : br.coded(g.c_res, g.zetaK); }
};

} // namespace bv
