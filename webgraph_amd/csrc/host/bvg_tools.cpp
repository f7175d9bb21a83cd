// bvg_tools.cpp -- CPU-side BVGraph writer + seeded synthetic graph generator (libbvgtools.so).
//
// The writer produces files in the reference's on-disk format so that the same .graph/.offsets/
// .properties load both here and in it.unimi.dsi.webgraph.BVGraph.  It follows the *decisions* of the
// reference compressor (which candidate reference is chosen, how copy blocks / intervals / residuals are
// formed: BVGraph.java:2222-2386 `call`, :2049-2219 `diffComp`, :1631-1654 `intervalize`), but is organised
// differently: candidate costs are computed arithmetically from code lengths instead of writing to a
// bit-counting stream, per-thread streams live in memory and are spliced bit-exactly at the end.
#include "../../../include/bvgtools.h"
#include "bv_props.hpp"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// CompressionFlags.java:26-44
enum { DELTA = 1, GAMMA = 2, GOLOMB = 3, SKEWED_GOLOMB = 4, UNARY = 5, ZETA = 6, NIBBLE = 7 };

inline int msb64(uint64_t v) { return 63 - __builtin_clzll(v); }

// ---------------------------------------------------------------- code lengths (bits)
inline uint64_t len_unary(uint64_t x) { return x + 1; }
inline uint64_t len_gamma(uint64_t x) { return 2 * (uint64_t)msb64(x + 1) + 1; }
inline uint64_t len_delta(uint64_t x) { int m = msb64(x + 1); return (uint64_t)m + len_gamma((uint64_t)m); }
inline uint64_t len_zeta(uint64_t x, int k) {
	uint64_t v = x + 1; int h = msb64(v) / k; uint64_t left = (uint64_t)1 << (h * k);
	return (uint64_t)h + 1 + (v - left < left ? (uint64_t)(h * k + k - 1) : (uint64_t)(h * k + k));
}
inline uint64_t len_golomb(uint64_t x, int b) {
	if (b == 0) return 0;
	uint64_t q = x / b, r = x % b; int l2 = msb64((uint64_t)b); uint64_t mm = ((uint64_t)1 << (l2 + 1)) - b;
	return q + 1 + (r < mm ? l2 : l2 + 1);
}
inline uint64_t len_nibble(uint64_t x) { return x == 0 ? 4 : 4 * (uint64_t)(msb64(x) / 3 + 1); }

inline uint64_t code_len(int coding, uint64_t x, int k) {
	switch (coding) {
	case GAMMA: return len_gamma(x);
	case DELTA: return len_delta(x);
	case UNARY: return len_unary(x);
	case ZETA: return len_zeta(x, k);
	case GOLOMB: return len_golomb(x, k);
	case NIBBLE: return len_nibble(x);
	default: return 0;
	}
}

// ---------------------------------------------------------------- MSB-first bit sink
struct BitSink {
	std::vector<uint64_t> w; // big-endian bit order inside each word: first stream bit = bit 63 of w[0]
	uint64_t acc = 0; int fill = 0; uint64_t bits = 0;
	void put(uint64_t v, int n) { // appends the low n bits of v, n in 0..64
		if (n == 0) return;
		bits += (uint64_t)n;
		if (n < 64) v &= (((uint64_t)1 << n) - 1);
		const int room = 64 - fill; // 1..64
		if (n < room) { acc |= v << (room - n); fill += n; return; }
		const int rest = n - room;  // 0..63
		acc |= rest ? v >> rest : v;
		w.push_back(acc);
		acc = rest ? v << (64 - rest) : 0; fill = rest;
	}
	void zeros(uint64_t n) { while (n >= 64) { put(0, 64); n -= 64; } put(0, (int)n); }
	void unary(uint64_t x) { zeros(x); put(1, 1); }
	void gamma(uint64_t x) { int m = msb64(x + 1); zeros((uint64_t)m); put(x + 1, m + 1); }
	void delta(uint64_t x) { int m = msb64(x + 1); gamma((uint64_t)m); put(x + 1, m); }
	void zeta(uint64_t x, int k) {
		uint64_t v = x + 1; int h = msb64(v) / k; unary((uint64_t)h);
		uint64_t left = (uint64_t)1 << (h * k);
		if (v - left < left) put(v - left, h * k + k - 1); else put(v, h * k + k);
	}
	void golomb(uint64_t x, int b) {
		if (b == 0) return;
		uint64_t q = x / b, r = x % b; unary(q);
		int l2 = msb64((uint64_t)b); uint64_t mm = ((uint64_t)1 << (l2 + 1)) - b;
		if (r < mm) put(r, l2); else put(r + mm, l2 + 1);
	}
	void nibble(uint64_t x) {
		if (x == 0) { put(8, 4); return; }
		int h = msb64(x) / 3;
		do { put(h == 0 ? 1 : 0, 1); put((x >> (h * 3)) & 7, 3); } while (h-- != 0);
	}
	void code(int coding, uint64_t x, int k) {
		switch (coding) {
		case GAMMA: gamma(x); break;
		case DELTA: delta(x); break;
		case UNARY: unary(x); break;
		case ZETA: zeta(x, k); break;
		case GOLOMB: golomb(x, k); break;
		case NIBBLE: nibble(x); break;
		}
	}
	// append all bits of another sink
	void append(const BitSink &o) {
		for (uint64_t v : o.w) put(v, 64);
		if (o.fill) put(o.acc >> (64 - o.fill), o.fill);
	}
	bool write_file(const std::string &path) const {
		FILE *f = fopen(path.c_str(), "wb");
		if (!f) return false;
		std::vector<uint8_t> buf; buf.reserve(1 << 20);
		auto flush = [&]() { if (!buf.empty()) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } };
		for (uint64_t v : w) { for (int s = 56; s >= 0; s -= 8) buf.push_back((uint8_t)(v >> s)); if (buf.size() >= (1 << 20)) flush(); }
		int nb = (fill + 7) / 8;
		for (int i = 0; i < nb; i++) buf.push_back((uint8_t)(acc >> (56 - 8 * i)));
		flush();
		bool ok = !ferror(f);
		fclose(f);
		return ok;
	}
};

inline uint64_t int2nat(int64_t x) { return x >= 0 ? (uint64_t)x << 1 : (uint64_t)(-x) * 2 - 1; } // Fast.int2nat

struct Codings { int outdegree = GAMMA, block = GAMMA, residual = ZETA, reference = UNARY, block_count = GAMMA, offset = GAMMA; };

struct Cfg { int W, R, I, K; Codings c; };

// One candidate's differential description (what diffComp builds before writing, BVGraph.java:2049-2172).
struct Diff {
	std::vector<int32_t> blocks, extras, left, len, residuals;
	void build(const int32_t *ref, int refLen, const int32_t *cur, int curLen, int minInterval) {
		blocks.clear(); extras.clear();
		int j = 0, k = 0, run = 0; bool copying = true;
		while (j < curLen && k < refLen) {
			if (copying) {
				if (cur[j] > ref[k]) { blocks.push_back(run); copying = false; run = 0; }
				else if (cur[j] < ref[k]) extras.push_back(cur[j++]);
				else { j++; k++; run++; }
			} else if (cur[j] < ref[k]) extras.push_back(cur[j++]);
			else if (cur[j] > ref[k]) { k++; run++; }
			else { blocks.push_back(run); copying = true; run = 0; }
		}
		if (copying && k < refLen) blocks.push_back(run);
		while (j < curLen) extras.push_back(cur[j++]);
		left.clear(); len.clear(); residuals.clear();
		if (minInterval != 0) { // intervalize, BVGraph.java:1631-1654
			const int vl = (int)extras.size(); const int32_t *v = extras.data();
			for (int i = 0; i < vl; i++) {
				int q = 0;
				if (i < vl - 1 && v[i] + 1 == v[i + 1]) {
					do q++; while (i + q < vl - 1 && v[i + q] + 1 == v[i + q + 1]);
					q++;
					if (q >= minInterval) { left.push_back(v[i]); len.push_back(q); i += q - 1; }
				}
				if (q < minInterval) residuals.push_back(v[i]);
			}
		} else residuals = extras;
	}
	// bits this description takes (the forReal=false run of diffComp)
	uint64_t cost(int32_t node, int ref, const Cfg &g) const {
		uint64_t t = 0;
		if (g.W > 0) t += code_len(g.c.reference, (uint64_t)ref, 0);
		if (ref != 0) {
			t += code_len(g.c.block_count, blocks.size(), 0);
			for (size_t i = 0; i < blocks.size(); i++) t += code_len(g.c.block, (uint64_t)(i == 0 ? blocks[0] : blocks[i] - 1), 0);
		}
		if (!extras.empty()) {
			if (g.I != 0) {
				t += len_gamma(left.size());
				int32_t prev = 0;
				for (size_t i = 0; i < left.size(); i++) {
					if (i == 0) t += len_gamma(int2nat((int64_t)left[0] - node));
					else t += len_gamma((uint64_t)(left[i] - prev - 1));
					prev = left[i] + len[i];
					t += len_gamma((uint64_t)(len[i] - g.I));
				}
			}
			if (!residuals.empty()) {
				t += code_len(g.c.residual, int2nat((int64_t)residuals[0] - node), g.K);
				for (size_t i = 1; i < residuals.size(); i++) t += code_len(g.c.residual, (uint64_t)(residuals[i] - residuals[i - 1] - 1), g.K);
			}
		}
		return t;
	}
	void emit(BitSink &o, int32_t node, int ref, const Cfg &g, bvt_store_stats &st, uint64_t *resBins) const {
		uint64_t b0 = o.bits;
		if (g.W > 0) { o.code(g.c.reference, (uint64_t)ref, 0); st.bits_references += o.bits - b0; }
		if (ref != 0) {
			b0 = o.bits;
			o.code(g.c.block_count, blocks.size(), 0);
			for (size_t i = 0; i < blocks.size(); i++) o.code(g.c.block, (uint64_t)(i == 0 ? blocks[0] : blocks[i] - 1), 0);
			st.bits_blocks += o.bits - b0;
		}
		if (!extras.empty()) {
			if (g.I != 0) {
				b0 = o.bits;
				o.gamma(left.size());
				int32_t prev = 0;
				for (size_t i = 0; i < left.size(); i++) {
					if (i == 0) o.gamma(int2nat((int64_t)left[0] - node)); else o.gamma((uint64_t)(left[i] - prev - 1));
					prev = left[i] + len[i];
					st.intervalised_arcs += (uint64_t)len[i];
					o.gamma((uint64_t)(len[i] - g.I));
				}
				st.bits_intervals += o.bits - b0;
			}
			if (!residuals.empty()) {
				b0 = o.bits;
				st.residual_arcs += residuals.size();
				for (size_t i = 0; i < residuals.size(); i++) { // updateBins(currNode, residual, residualCount, residualGapStats), BVGraph.java:2196
					const int b = bvprops::gap_bin(i == 0, i == 0 ? node : residuals[i - 1], residuals[i]);
					if (b >= 0) resBins[b]++;
				}
				o.code(g.c.residual, int2nat((int64_t)residuals[0] - node), g.K);
				for (size_t i = 1; i < residuals.size(); i++) o.code(g.c.residual, (uint64_t)(residuals[i] - residuals[i - 1] - 1), g.K);
				st.bits_residuals += o.bits - b0;
			}
		}
	}
};

struct ThreadOut { BitSink graph; std::vector<uint64_t> reclen; bvt_store_stats st{}; uint64_t succBins[32] = {}, resBins[32] = {}; int err = 0; };

// CompressionThread.call for nodes [lo, hi): empty window at lo (BVGraph.java:2222-2386).
void compress_range(int32_t lo, int32_t hi, const int64_t *rowptr, const int32_t *succ, const Cfg &g, ThreadOut &out) {
	const int cyc = g.W + 1;
	std::vector<int32_t> refCount(cyc, 0), listNode(cyc, -1);
	Diff best, cand;
	out.reclen.reserve((size_t)(hi - lo));
	for (int32_t x = lo; x < hi; x++) {
		const uint64_t start = out.graph.bits;
		const int32_t *cur = succ + rowptr[x]; const int d = (int)(rowptr[x + 1] - rowptr[x]);
		const int ci = x % cyc;
		out.graph.code(g.c.outdegree, (uint64_t)d, 0);
		out.st.bits_outdegrees += out.graph.bits - start;
		listNode[ci] = x;
		if (d > 0) {
			for (int i = 1; i < d; i++) if (cur[i] <= cur[i - 1]) { out.err = -EINVAL; return; } // strictly increasing rows only
			for (int i = 0; i < d; i++) { // updateBins(currNode, list[currIndex], outd, successorGapStats), BVGraph.java:2303
				const int b = bvprops::gap_bin(i == 0, i == 0 ? x : cur[i - 1], cur[i]);
				if (b >= 0) out.succBins[b]++;
			}
			uint64_t bestCost = UINT64_MAX; int bestRef = -1, bestCand = -1;
			refCount[ci] = -1;
			for (int r = 0; r < cyc; r++) {
				if (x - r < lo) break; // nothing before the range start: window starts empty
				const int c = (int)(((int64_t)x - r + cyc) % cyc);
				const int y = x - r; const int yl = (int)(rowptr[y + 1] - rowptr[y]);
				if (refCount[c] < g.R && yl != 0) {
					cand.build(succ + rowptr[y], r == 0 ? 0 : yl, cur, d, g.I);
					uint64_t t = cand.cost(x, r, g);
					if (t < bestCost) { bestCost = t; bestRef = r; bestCand = c; std::swap(best, cand); }
				}
			}
			refCount[ci] = refCount[bestCand] + 1;
			best.emit(out.graph, x, bestRef, g, out.st, out.resBins);
			uint64_t copied = (uint64_t)d - best.extras.size();
			out.st.copied_arcs += copied;
			out.st.tot_ref += (uint64_t)refCount[ci];
			out.st.tot_dist += (uint64_t)bestRef;
			if (refCount[ci] > out.st.max_ref_chain) out.st.max_ref_chain = refCount[ci];
		}
		out.reclen.push_back(out.graph.bits - start);
	}
}

// ---------------------------------------------------------------- RNG: splitmix64-seeded xoroshiro128+
struct Rng {
	uint64_t s0, s1;
	static uint64_t splitmix(uint64_t &z) { z += 0x9E3779B97F4A7C15ULL; uint64_t r = z; r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ULL; r = (r ^ (r >> 27)) * 0x94D049BB133111EBULL; return r ^ (r >> 31); }
	explicit Rng(uint64_t seed) { uint64_t z = seed; s0 = splitmix(z); s1 = splitmix(z); if (!(s0 | s1)) s1 = 1; }
	uint64_t next() { uint64_t a = s0, b = s1, r = a + b; b ^= a; s0 = ((a << 24) | (a >> 40)) ^ b ^ (b << 16); s1 = (b << 37) | (b >> 27); return r; }
	double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); } // [0,1)
};

// floor of a Pareto variate on [1, xmax+1) with density ~ x^-(a+1)
inline int64_t pareto_floor(Rng &r, double a, double xmax) {
	double u = r.unit();
	double t = 1.0 - u * (1.0 - std::pow(xmax + 1.0, -a));
	double x = std::pow(t, -1.0 / a);
	int64_t v = (int64_t)x;
	if (v < 1) v = 1;
	if ((double)v > xmax) v = (int64_t)xmax;
	return v;
}

const int GEN_BLOCK = 1 << 16; // nodes per independent generation block

} // namespace

extern "C" int bvt_generate_ex(int32_t n, int64_t m, uint64_t seed, double p_copy, double p_same, double p_keep, int threads, int64_t **rowptr_out, int32_t **succ_out);

extern "C" int bvt_store(const char *basename, int32_t n, const int64_t *rowptr, const int32_t *succ,
                         int window, int max_ref_count, int min_interval, int zeta_k, uint32_t flags, int threads,
                         bvt_store_stats *stats) {
	if (!basename || n < 0 || !rowptr || window < 0 || min_interval < 0 || zeta_k < 1) return -EINVAL;
	Cfg g; g.W = window; g.R = max_ref_count; g.I = min_interval; g.K = zeta_k;
	if (flags & 0xF) g.c.outdegree = flags & 0xF;                 // setFlags, BVGraph.java:1317-1325
	if ((flags >> 4) & 0xF) g.c.block = (flags >> 4) & 0xF;
	if ((flags >> 8) & 0xF) g.c.residual = (flags >> 8) & 0xF;
	if ((flags >> 12) & 0xF) g.c.reference = (flags >> 12) & 0xF;
	if ((flags >> 16) & 0xF) g.c.block_count = (flags >> 16) & 0xF;
	if ((flags >> 20) & 0xF) g.c.offset = (flags >> 20) & 0xF;
	auto in = [](int c, std::initializer_list<int> ok) { for (int v : ok) if (c == v) return true; return false; };
	if (!in(g.c.outdegree, { GAMMA, DELTA }) || !in(g.c.block, { GAMMA, DELTA, UNARY }) || !in(g.c.block_count, { GAMMA, DELTA, UNARY }) ||
	    !in(g.c.reference, { UNARY, GAMMA, DELTA }) || !in(g.c.residual, { GAMMA, ZETA, DELTA, GOLOMB, NIBBLE }) || !in(g.c.offset, { GAMMA, DELTA }))
		return -ENOTSUP;
	if (threads < 1) threads = 1;
	if (threads > n) threads = n > 0 ? n : 1;

	std::vector<ThreadOut> outs((size_t)threads);
	const int32_t per = (int32_t)(((int64_t)n + threads - 1) / threads);
	std::vector<std::thread> pool;
	for (int t = 0; t < threads; t++) {
		int32_t lo = (int32_t)std::min<int64_t>((int64_t)t * per, n), hi = (int32_t)std::min<int64_t>((int64_t)lo + per, n);
		pool.emplace_back([&, lo, hi, t]() { compress_range(lo, hi, rowptr, succ, g, outs[(size_t)t]); });
	}
	for (auto &th : pool) th.join();
	for (auto &o : outs) if (o.err) return o.err;

	BitSink graph, offs;
	bvt_store_stats st{};
	st.threads = threads;
	uint64_t succBins[32] = {}, resBins[32] = {};
	offs.code(g.c.offset, 0, 0); // offset of node 0
	for (auto &o : outs) {
		for (int i = 0; i < 32; i++) { succBins[i] += o.succBins[i]; resBins[i] += o.resBins[i]; }
		if (threads == 1) graph = std::move(o.graph); else graph.append(o.graph);
		for (uint64_t l : o.reclen) offs.code(g.c.offset, l, 0);
		st.bits_outdegrees += o.st.bits_outdegrees; st.bits_references += o.st.bits_references; st.bits_blocks += o.st.bits_blocks;
		st.bits_intervals += o.st.bits_intervals; st.bits_residuals += o.st.bits_residuals;
		st.copied_arcs += o.st.copied_arcs; st.intervalised_arcs += o.st.intervalised_arcs; st.residual_arcs += o.st.residual_arcs;
		st.tot_ref += o.st.tot_ref; st.tot_dist += o.st.tot_dist;
		st.max_ref_chain = std::max(st.max_ref_chain, o.st.max_ref_chain);
	}
	st.written_bits = graph.bits; st.offsets_bits = offs.bits;
	std::string base(basename);
	if (!graph.write_file(base + ".graph") || !offs.write_file(base + ".offsets")) return -EIO;

	const uint64_t m = n ? (uint64_t)rowptr[n] : 0;
	bvprops::Counters cnt{ st.written_bits, st.bits_outdegrees, st.bits_references, st.bits_blocks, st.bits_intervals, st.bits_residuals,
	                       st.copied_arcs, st.intervalised_arcs, st.residual_arcs, st.tot_ref, st.tot_dist, {}, {} };
	for (int i = 0; i < 32; i++) { cnt.successor_gap_bins[i] = succBins[i]; cnt.residual_gap_bins[i] = resBins[i]; }
	if (!bvprops::write(base + ".properties", n, m, window, max_ref_count, min_interval, zeta_k, g.c.residual == ZETA, flags, cnt)) return -EIO;
	if (stats) *stats = st;
	return 0;
}

extern "C" int bvt_generate(int32_t n, int64_t m, uint64_t seed, double p_copy, int threads, int64_t **rowptr_out, int32_t **succ_out) {
	return bvt_generate_ex(n, m, seed, p_copy, 0.0, 0.7, threads, rowptr_out, succ_out);
}

// p_same > 0: outdegrees come in runs (a node repeats its predecessor's raw outdegree with probability p_same), and a
// copying node prefers its immediate predecessor as prototype -- pages of one site with near-identical link lists, the
// shape that gives real web graphs their deep reference chains (cnr-2000: 47.5 % of the non-empty nodes at depth 3).
// p_same = 0, p_keep = 0.7 is the C2 recipe, bit for bit (no extra random draws).
extern "C" int bvt_generate_ex(int32_t n, int64_t m, uint64_t seed, double p_copy, double p_same, double p_keep, int threads, int64_t **rowptr_out, int32_t **succ_out) {
	if (n <= 0 || m < 0 || !rowptr_out || !succ_out) return -EINVAL;
	if (threads < 1) threads = 1;
	const int64_t dcap = std::max<int64_t>(1, n / 4);
	if (m > (int64_t)n * dcap / 2) return -EINVAL;
	const int nblocks = (n + GEN_BLOCK - 1) / GEN_BLOCK;

	// 1. raw power-law outdegrees: P(d) ~ d^-2.1 on [1, 1e5]; 20% of the nodes forced empty.  BVT_DEGREE_ALPHA (default 1.1: the recipe of C2 / C5) moves the
	// exponent for the threshold experiments of round 5 (density ~ d^-(alpha + 1): 0.8 = heavier tail, 1.6 = lighter); the cache names of bench.py carry it.
	const char *eAlpha = getenv("BVT_DEGREE_ALPHA");
	const double degAlpha = eAlpha && atof(eAlpha) > 0 ? atof(eAlpha) : 1.1;
	std::vector<int32_t> raw((size_t)n);
	auto par = [&](auto fn) {
		std::atomic<int> next{0};
		std::vector<std::thread> pool;
		for (int t = 0; t < threads; t++) pool.emplace_back([&]() { for (int b; (b = next.fetch_add(1)) < nblocks;) fn(b); });
		for (auto &th : pool) th.join();
	};
	par([&](int b) {
		Rng r(seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(b + 1)));
		int32_t lo = b * GEN_BLOCK, hi = std::min<int64_t>((int64_t)lo + GEN_BLOCK, n);
		for (int32_t x = lo; x < hi; x++) {
			if (p_same > 0 && x > lo && raw[(size_t)x - 1] && r.unit() < p_same) { raw[(size_t)x] = raw[(size_t)x - 1]; continue; }
			raw[(size_t)x] = r.unit() < 0.2 ? 0 : (int32_t)pareto_floor(r, degAlpha, 1e5);
		}
	});
	// 2. rescale to hit exactly m arcs: d = clamp(round(raw*s), 1, dcap) for raw>0, s by bisection
	auto total = [&](double s) { int64_t t = 0; for (int32_t x = 0; x < n; x++) if (raw[(size_t)x]) t += std::min<int64_t>(dcap, std::max<int64_t>(1, (int64_t)std::llround(raw[(size_t)x] * s))); return t; };
	double lo_s = 0, hi_s = 1;
	while (total(hi_s) < m && hi_s < 1e9) hi_s *= 2;
	for (int it = 0; it < 60; it++) { double mid = 0.5 * (lo_s + hi_s); if (total(mid) < m) lo_s = mid; else hi_s = mid; }
	std::vector<int64_t> deg((size_t)n);
	int64_t tot = 0;
	for (int32_t x = 0; x < n; x++) { deg[(size_t)x] = raw[(size_t)x] ? std::min<int64_t>(dcap, std::max<int64_t>(1, (int64_t)std::llround(raw[(size_t)x] * hi_s))) : 0; tot += deg[(size_t)x]; }
	// fix the remainder on a fixed stride of non-empty nodes
	for (int32_t x = 0; tot != m && x < n; x++) {
		if (tot > m && deg[(size_t)x] > 1) { deg[(size_t)x]--; tot--; }
		else if (tot < m && deg[(size_t)x] > 0 && deg[(size_t)x] < dcap) { deg[(size_t)x]++; tot++; }
	}
	if (tot != m) { for (int32_t x = 0; tot < m && x < n; x++) if (deg[(size_t)x] < dcap) { int64_t a = std::min<int64_t>(dcap - deg[(size_t)x], m - tot); deg[(size_t)x] += a; tot += a; } }
	if (tot != m) return -ERANGE;

	int64_t *rowptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
	int32_t *succ = (int32_t *)malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(m, 1));
	if (!rowptr || !succ) { free(rowptr); free(succ); return -ENOMEM; }
	rowptr[0] = 0;
	for (int32_t x = 0; x < n; x++) rowptr[x + 1] = rowptr[x] + deg[(size_t)x];

	// 3. successor lists: copy model + power-law signed gaps + runs of consecutive ids
	par([&](int b) {
		Rng r(seed ^ (0xA0761D6478BD642FULL * (uint64_t)(b + 1)));
		int32_t lo = b * GEN_BLOCK, hi = std::min<int64_t>((int64_t)lo + GEN_BLOCK, n);
		std::vector<int32_t> cur;
		for (int32_t x = lo; x < hi; x++) {
			const int64_t d = deg[(size_t)x];
			if (!d) continue;
			cur.clear();
			if (x > lo && r.unit() < p_copy) {
				int32_t back = 1 + (int32_t)(r.next() % (uint64_t)std::min(7, x - lo));
				if (p_same > 0 && r.unit() < p_same) back = 1;
				int32_t y = x - back;
				const int32_t *pl = succ + rowptr[y]; int64_t pd = rowptr[y + 1] - rowptr[y];
				for (int64_t i = 0; i < pd && (int64_t)cur.size() < d; i++) if (r.unit() < p_keep) cur.push_back(pl[i]);
			}
			for (int round = 0; (int64_t)cur.size() < d; round++) {
				int64_t need = d - (int64_t)cur.size();
				while (need > 0) {
					int64_t gmag = pareto_floor(r, 0.5, (double)n);
					int64_t t = (r.next() & 1) ? (int64_t)x + gmag : (int64_t)x - gmag;
					if (round > 8) t = (int64_t)(r.next() % (uint64_t)n); // dense rows: fall back to uniform targets
					if (t < 0 || t >= n) continue;
					if (r.unit() < 0.1) { // a run of consecutive ids, geometric length >= 3
						int64_t L = 3; while (r.unit() < 0.5) L++;
						L = std::min(L, need);
						for (int64_t i = 0; i < L && t + i < n; i++) { cur.push_back((int32_t)(t + i)); need--; }
					} else { cur.push_back((int32_t)t); need--; }
				}
				std::sort(cur.begin(), cur.end());
				cur.erase(std::unique(cur.begin(), cur.end()), cur.end());
			}
			if ((int64_t)cur.size() > d) cur.resize((size_t)d);
			memcpy(succ + rowptr[x], cur.data(), sizeof(int32_t) * (size_t)d);
		}
	});
	*rowptr_out = rowptr; *succ_out = succ;
	return 0;
}

extern "C" void bvt_free(void *p) { free(p); }

// The node ids of SpeedTest's random-access leg (src/it/unimi/dsi/webgraph/test/SpeedTest.java:79, :98-111):
// `r.setSeed(seed)` before every repetition, then `r.nextInt(n)` per sample, r an XoRoShiRo128PlusRandom.  That class
// lives in dsiutils (not in the reference repository, version unpinned): restated here as xoroshiro128+ with the 2018
// constants (24, 16, 37), state = two SplitMix64 outputs of the seed, nextInt(n) = nextLong(n) by the high bits for a power
// of two and by rejection on the top 63 bits otherwise.  The exact id stream is therefore "parity unpinned"; the protocol
// (the same ids in every repetition, uniform over [0, n)) is what the reference pins.
extern "C" int bvt_random_nodes(uint64_t seed, int32_t n, int64_t count, int32_t *out) {
	if (n <= 0 || count < 0 || !out) return -EINVAL;
	Rng r(seed);
	const uint64_t nn = (uint64_t)n, nm1 = nn - 1;
	for (int64_t i = 0; i < count; i++) {
		uint64_t t = r.next();
		if ((nn & nm1) == 0) { out[i] = (int32_t)(nm1 ? (t >> __builtin_clzll(nm1)) & nm1 : 0); continue; }
		for (uint64_t u = t >> 1; (int64_t)(u + nm1 - (t = u % nn)) < 0; u = r.next() >> 1) {}
		out[i] = (int32_t)t;
	}
	return 0;
}

// BitStreamArcLabelledImmutableGraph.store (labelling/BitStreamArcLabelledImmutableGraph.java:650-693): the labels of
// all arcs in enumeration order as one bit stream, gamma(0) + gamma(bits of every node's list) as offsets, and the
// three-line property file of saveProperties (:689-695).
extern "C" int bvt_store_labels(const char *basename, const char *underlying, int32_t n, const int64_t *rowptr, const int32_t *labels,
                                int kind, int width, const char *key) {
	if (!basename || !underlying || n < 0 || !rowptr || (kind != 1 && kind != 2) || (kind == 2 && (width < 0 || width > 32))) return -EINVAL;
	BitSink lab, offs;
	offs.gamma(0);
	for (int32_t x = 0; x < n; x++) {
		const uint64_t before = lab.bits;
		for (int64_t a = rowptr[x]; a < rowptr[x + 1]; a++) {
			const uint32_t v = (uint32_t)labels[a];
			if (kind == 1) { if (labels[a] < 0) return -EINVAL; lab.gamma(v); }            // GammaCodedIntLabel.java:74-76
			else { if (width < 32 && (v >> width)) return -EINVAL; lab.put(v, width); }     // FixedWidthIntLabel.java:76-78
		}
		offs.gamma(lab.bits - before);
	}
	std::string base(basename);
	if (!lab.write_file(base + ".labels") || !offs.write_file(base + ".labeloffsets")) return -EIO;
	FILE *f = fopen((base + ".properties").c_str(), "w");
	if (!f) return -EIO;
	fprintf(f, "graphclass = it.unimi.dsi.webgraph.labelling.BitStreamArcLabelledImmutableGraph\n");
	fprintf(f, "underlyinggraph = %s\n", underlying);
	if (kind == 1) fprintf(f, "labelspec = it.unimi.dsi.webgraph.labelling.GammaCodedIntLabel(%s)\n", key ? key : "FOO");
	else fprintf(f, "labelspec = it.unimi.dsi.webgraph.labelling.FixedWidthIntLabel(%s,%d)\n", key ? key : "FOO", width);
	fclose(f);
	return 0;
}

// FixedWidthIntListLabel.toBitStream (FixedWidthIntListLabel.java:114-119): gamma(length), then every element on `width` bits
extern "C" int bvt_store_label_lists(const char *basename, const char *underlying, int32_t n, const int64_t *rowptr, const int64_t *listptr,
                                     const int32_t *values, int width, const char *key) {
	if (!basename || !underlying || n < 0 || !rowptr || !listptr || width < 0 || width > 32) return -EINVAL;
	BitSink lab, offs;
	offs.gamma(0);
	for (int32_t x = 0; x < n; x++) {
		const uint64_t before = lab.bits;
		for (int64_t a = rowptr[x]; a < rowptr[x + 1]; a++) {
			if (listptr[a + 1] < listptr[a]) return -EINVAL;
			lab.gamma((uint64_t)(listptr[a + 1] - listptr[a]));
			for (int64_t i = listptr[a]; i < listptr[a + 1]; i++) {
				const uint32_t v = (uint32_t)values[i];
				if (width < 32 && (v >> width)) return -EINVAL; // "Value too large" (:64)
				lab.put(v, width);
			}
		}
		offs.gamma(lab.bits - before);
	}
	std::string base(basename);
	if (!lab.write_file(base + ".labels") || !offs.write_file(base + ".labeloffsets")) return -EIO;
	FILE *f = fopen((base + ".properties").c_str(), "w");
	if (!f) return -EIO;
	fprintf(f, "graphclass = it.unimi.dsi.webgraph.labelling.BitStreamArcLabelledImmutableGraph\n");
	fprintf(f, "underlyinggraph = %s\n", underlying);
	fprintf(f, "labelspec = it.unimi.dsi.webgraph.labelling.FixedWidthIntListLabel(%s,%d)\n", key ? key : "FOO", width);
	fclose(f);
	return 0;
}

// ---------------------------------------------------------------- EFGraph writer (second format)
// EFGraph.store (EFGraph.java:812-889) with its Accumulator (:420-552) and LongWordOutputBitStream (:298-418): per node
// gamma(outdegree), then the quasi-succinct (Elias-Fano) encoding of the successors followed by the terminator `upperBound`:
// forward pointers, lower bits, upper bits.  Bits are taken from the LOW end of 64-bit words, the words are written in `byteorder`.
namespace {
struct LongWordSink { // LongWordOutputBitStream: value bits go in from bit `64 - free` upwards
	std::vector<uint64_t> w;
	uint64_t buffer = 0; int free = 64; uint64_t bits = 0;
	int append(uint64_t value, int width) { // EFGraph.java:316-340
		if (width == 0) return 0;
		bits += (uint64_t)width;
		buffer |= free == 64 ? value : value << (64 - free);
		if (width < free) free -= width;
		else {
			w.push_back(buffer);
			if (width == free) { buffer = 0; free = 64; }
			else { buffer = value >> free; free = 64 - width + free; }
		}
		return width;
	}
	int gamma(uint64_t value) { // writeGamma -> writeNonZeroGamma(value + 1), :398-410
		const uint64_t v = value + 1; const int msb = msb64(v); const uint64_t unary = (uint64_t)1 << msb;
		append(unary, msb + 1); append(v ^ unary, msb);
		return 2 * msb + 1;
	}
	void unary(uint64_t zeros) { while (zeros >= 64) { append(0, 64); zeros -= 64; } append((uint64_t)1 << zeros, (int)zeros + 1 > 64 ? 64 : (int)zeros + 1); }
	void append_all(const LongWordSink &o) { // append(LongWordCache), :368-378
		for (uint64_t v : o.w) append(v, 64);
		if (o.free != 64) append(o.buffer, 64 - o.free);
	}
	void clear() { w.clear(); buffer = 0; free = 64; bits = 0; }
};
inline int ef_lower_bits(uint64_t length, uint64_t ub) { if (length == 0) return 0; const uint64_t q = ub / length; return q == 0 ? 0 : msb64(q); }          // EFGraph.java:145-147
inline int ef_ceil_log2(uint64_t x) { return x <= 2 ? (int)x - 1 : 64 - __builtin_clzll(x - 1); }                                                     // dsiutils Fast.ceilLog2
inline int ef_pointer_size(uint64_t length, uint64_t ub) { return std::max(0, ef_ceil_log2(length + (ub >> ef_lower_bits(length, ub)))); }             // :156-158
inline uint64_t ef_num_pointers(uint64_t length, uint64_t ub, int lq) { return length == 0 ? 0 : (ub >> ef_lower_bits(length, ub)) >> lq; }          // :168-171
} // namespace

extern "C" int bvt_store_ef(const char *basename, int32_t n, const int64_t *rowptr, const int32_t *succ, int32_t upper_bound, int log2_quantum, int big_endian) {
	if (!basename || n < 0 || !rowptr || log2_quantum < 0 || log2_quantum > 62 || upper_bound < n) return -EINVAL;
	LongWordSink graph, pointers, lower, upper;
	BitSink offs;
	offs.delta(0); // offsets.writeLongDelta(0), :830
	uint64_t bitsOutd = 0, bitsSucc = 0;
	const uint64_t ub = (uint64_t)upper_bound, quantum = (uint64_t)1 << log2_quantum;
	for (int32_t x = 0; x < n; x++) {
		const int64_t a = rowptr[x], d = rowptr[x + 1] - a;
		const int ob = graph.gamma((uint64_t)d);
		bitsOutd += (uint64_t)ob;
		// Accumulator.init(outdegree, upperBound, strict = false, indexZeroes = true, log2Quantum), :483-507
		const uint64_t len = (uint64_t)d + 1;
		const int l = ef_lower_bits(len, ub), ps = ef_pointer_size(len, ub);
		pointers.clear(); lower.clear(); upper.clear();
		int64_t lastOne = -1; uint64_t cur = 0;
		int64_t prev = -1;
		for (int64_t i = 0; i <= d; i++) { // add(), :509-525; the last turn is dump()'s terminator, :529-531
			const uint64_t v = i < d ? (uint64_t)(uint32_t)succ[a + i] : ub;
			if (i < d && (succ[a + i] < 0 || (int64_t)v <= prev || v >= ub)) return -EINVAL; // strictly increasing, below the bound (:510, :513)
			prev = (int64_t)v;
			if (l) lower.append(v & (((uint64_t)1 << l) - 1), l);
			const int64_t one = (int64_t)(v >> l) + (int64_t)cur;
			upper.unary((uint64_t)(one - lastOne - 1));
			int64_t zeroesBefore = lastOne - (int64_t)cur + 1;
			for (int64_t pos = lastOne + (zeroesBefore & -(int64_t)quantum) + (int64_t)quantum - zeroesBefore; pos < one; pos += (int64_t)quantum, zeroesBefore += (int64_t)quantum)
				pointers.append((uint64_t)(pos + 1), ps);
			lastOne = one; cur++;
		}
		graph.append_all(pointers); graph.append_all(lower); graph.append_all(upper);
		const uint64_t sb = pointers.bits + lower.bits + upper.bits;
		bitsSucc += sb;
		offs.delta((uint64_t)ob + sb); // :855
	}
	graph.w.push_back(graph.buffer); // close() writes the buffer whatever it holds, :412-417
	std::string base(basename);
	{
		FILE *f = fopen((base + ".graph").c_str(), "wb");
		if (!f) return -EIO;
		std::vector<uint8_t> buf; buf.reserve(1 << 20);
		for (uint64_t v : graph.w) {
			for (int b = 0; b < 8; b++) buf.push_back((uint8_t)(big_endian ? v >> (56 - 8 * b) : v >> (8 * b)));
			if (buf.size() >= (1 << 20)) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
		}
		if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), f);
		const bool ok = !ferror(f);
		if (fclose(f) != 0 || !ok) return -EIO;
	}
	if (!offs.write_file(base + ".offsets")) return -EIO;
	const uint64_t m = n ? (uint64_t)rowptr[n] : 0, written = (uint64_t)graph.w.size() * 64;
	if (!bvprops::write_ef(base + ".properties", n, m, upper_bound, log2_quantum, big_endian != 0, written, bitsOutd, bitsSucc)) return -EIO;
	return 0;
}
