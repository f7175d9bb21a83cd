// encode_model.cpp -- the device compressor's phases (webgraph_amd/csrc/bv_encode.hpp) run on the host, one "lane" after
// the other, so that the CPU suite can compare them with the CPU writer (libbvgtools) and the reference's cnr-2000 bytes
// before a GPU sees them.  Test infrastructure; built by tests/test_encode_model_cpu.py with g++.
#include "../../webgraph_amd/csrc/bv_encode.hpp"

#include <cstring>
#include <vector>

using namespace bve;

extern "C" int bve_model_compress(int32_t n, const int64_t *rowptr, const int32_t *succ, int W, int R, int I, int K, const int32_t *codings /* outd, blk, res, ref, bc, off */,
                                  int32_t per, int32_t chunk, int32_t span, uint32_t *words, uint64_t cap_words, uint64_t *bits_out, int64_t *off /* n+1 */, uint32_t *off_words,
                                  uint64_t off_cap_words, uint64_t *off_bits_out, uint64_t *stats /* 11 */, int32_t *rounds_out) {
	Params p{ W, R, I, K, codings[0], codings[1], codings[2], codings[3], codings[4], codings[5], per };
	const int cyc = W + 1;
	const bool def = default_codings(p); // the compile-time variant of the default codings is what the device runs for them
	int err = 0;
	// A
	std::vector<uint32_t> cost((size_t)n * cyc);
	for (int64_t q = 0; q < (int64_t)n * cyc; q++) cost[(size_t)q] = def ? pair_cost<true>(p, rowptr, succ, (int32_t)(q / cyc), (int)(q % cyc), &err) : pair_cost<false>(p, rowptr, succ, (int32_t)(q / cyc), (int)(q % cyc), &err);
	if (err) return -3;
	// B: rounds of select_span until nothing moves, as encode_device does (round 0: one chunk per lane; then `span` chunks per lane)
	const int64_t nchunks = ((int64_t)n + chunk - 1) / chunk;
	const int Wn = W ? W : 1;
	std::vector<int32_t> state[2] = { std::vector<int32_t>((size_t)nchunks * Wn, 0), std::vector<int32_t>((size_t)nchunks * Wn, 0) }, used((size_t)nchunks * Wn, 0);
	std::vector<uint8_t> best((size_t)n + 1, 0);
	std::vector<int32_t> refc((size_t)n + 1, 0), in((size_t)Wn, 0);
	int rounds = 0;
	for (int round = 0; nchunks > 0; round++) {
		const int64_t sp = round == 0 ? 1 : span;
		bool changed = false;
		for (int64_t c0 = 0; c0 < nchunks; c0 += sp)
			if (select_span(p, rowptr, cost.data(), n, chunk, c0, c0 + sp < nchunks ? c0 + sp : nchunks, round, state[(round + 1) & 1].data(), state[round & 1].data(), used.data(),
			                best.data(), refc.data(), in.data())) changed = true;
		rounds = round + 1;
		if (round > 0 && !changed) break;
		if (W == 0 || R == 0) break;
	}
	*rounds_out = rounds;
	// C
	off[0] = 0;
	std::vector<uint64_t> reclen((size_t)n);
	for (int32_t x = 0; x < n; x++) {
		const int64_t d = rowptr[x + 1] - rowptr[x];
		LenSink s;
		w_code(s, p.c_outd, (uint64_t)d, 0);
		uint64_t t = s.bits;
		if (d > 0) t += cost[(size_t)x * cyc + best[(size_t)x]];
		reclen[(size_t)x] = t;
		off[x + 1] = off[x] + (int64_t)t;
	}
	*bits_out = (uint64_t)off[n];
	if (((uint64_t)off[n] + 31) / 32 > cap_words) return -8;
	// D
	memset(words, 0, cap_words * 4);
	memset(stats, 0, 11 * sizeof(uint64_t));
	for (int32_t x = 0; x < n; x++) {
		NodeStats st;
		const uint64_t len = def ? emit_node<true>(p, rowptr, succ, x, best[(size_t)x], words, (uint64_t)off[x], &st) : emit_node<false>(p, rowptr, succ, x, best[(size_t)x], words, (uint64_t)off[x], &st);
		if (len != reclen[(size_t)x]) return -100;
		stats[0] += st.bitsOutd; stats[1] += st.bitsRef; stats[2] += st.bitsBlocks; stats[3] += st.bitsIntervals; stats[4] += st.bitsResiduals;
		stats[5] += st.copied; stats[6] += st.intervalised; stats[7] += st.residuals;
		if (rowptr[x + 1] > rowptr[x]) {
			stats[8] += (uint64_t)refc[(size_t)x]; stats[9] += best[(size_t)x];
			if ((uint64_t)refc[(size_t)x] > stats[10]) stats[10] = (uint64_t)refc[(size_t)x];
		}
	}
	// E
	uint64_t at = 0;
	std::vector<uint64_t> offat((size_t)n + 2);
	for (int64_t i = 0; i <= n; i++) { LenSink s; w_code(s, p.c_off, i == 0 ? 0 : reclen[(size_t)i - 1], 0); offat[(size_t)i] = at; at += s.bits; }
	*off_bits_out = at;
	if ((at + 31) / 32 > off_cap_words) return -8;
	memset(off_words, 0, off_cap_words * 4);
	for (int64_t i = 0; i <= n; i++) { WordSink s(off_words, offat[(size_t)i]); w_code(s, p.c_off, i == 0 ? 0 : reclen[(size_t)i - 1], 0); s.finish(); }
	return 0;
}
