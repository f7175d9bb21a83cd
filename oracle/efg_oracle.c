/*
 * efg_oracle.c -- CPU restatement of the EFGraph read path (TEST INFRASTRUCTURE, part of libbvgoracle.so: used by tests/ only).
 *
 * Follows src/it/unimi/dsi/webgraph/EFGraph.java: the record of node x starts at bit offsets[x] of a stream of 64-bit words
 * read from the LOW bit up (LongWordBitReader, :892-1033): gamma(outdegree) (readGamma :1024-1032), then the Elias-Fano
 * encoding of outdegree + 1 values (the successors and the terminator upperBound) -- forward pointers, lower bits, upper
 * bits (EliasFanoSuccessorReader :1103-1145; sizes :145-171).
 *
 * PARITY UNPINNED: the reference holds no EFGraph fixture (test/it/unimi/dsi/webgraph/EFGraphTest.java round-trips only) and
 * cannot be built here (Java).  This file, the writer (bvt_store_ef) and the GPU kernels are checked against each other, and
 * the writer against a record worked out by hand from the format description (tests/test_efgraph_cpu.py).  The forward
 * pointers, which no scan reads, are read by efo_skip_to below the way the reference's skipTo reads them (:1147-1215).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EFO_OK 0
#define EFO_EARG (-1)
#define EFO_ENOMEM (-5)
#define EFO_EFORMAT (-7)

typedef struct { const uint64_t *w; uint64_t nw; int err; } lw_t;

static inline uint64_t lw_word(lw_t *s, uint64_t i) { if (i >= s->nw) { s->err = 1; return 0; } return s->w[i]; }
/* `width` bits (0..64) starting at bit `pos`, low bits first (LongWordBitReader.extract, :960-1000) */
static inline uint64_t lw_get(lw_t *s, uint64_t pos, int width) {
	if (width == 0) return 0;
	const uint64_t i = pos >> 6; const int b = (int)(pos & 63);
	uint64_t v = lw_word(s, i) >> b;
	if (b + width > 64) v |= lw_word(s, i + 1) << (64 - b);
	return width == 64 ? v : v & (((uint64_t)1 << width) - 1);
}
/* zeros up to the next one at or after `pos` (readUnary, :1002-1022) */
static inline uint64_t lw_unary(lw_t *s, uint64_t *pos) {
	uint64_t i = *pos >> 6; const int b = (int)(*pos & 63);
	uint64_t w = lw_word(s, i) & (~(uint64_t)0 << b), z = 0;
	while (w == 0) { z += 64; if (++i >= s->nw) { s->err = 1; return 0; } w = s->w[i]; }
	const uint64_t one = i * 64 + (uint64_t)__builtin_ctzll(w);
	const uint64_t zeros = one - *pos;
	(void)z;
	*pos = one + 1;
	return zeros;
}
static inline uint64_t lw_gamma(lw_t *s, uint64_t *pos) { /* readGamma = readNonZeroGamma - 1 */
	const uint64_t msb = lw_unary(s, pos);
	if (msb > 62) { s->err = 1; return 0; }
	const uint64_t v = lw_get(s, *pos, (int)msb) | ((uint64_t)1 << msb);
	*pos += msb;
	return v - 1;
}
static inline int msb64(uint64_t v) { return 63 - __builtin_clzll(v); }
static inline int ef_lower_bits(uint64_t length, uint64_t ub) { if (length == 0) return 0; const uint64_t q = ub / length; return q == 0 ? 0 : msb64(q); } /* :145-147 */
static inline int ef_ceil_log2(uint64_t x) { return x <= 2 ? (int)x - 1 : 64 - __builtin_clzll(x - 1); }                                        /* dsiutils Fast.ceilLog2 */

/* words: the .graph file as host-order 64-bit words (the caller undoes `byteorder`); offsets: decoded, n + 1 values.
 * rowptr[to - from + 1] (may be NULL), succ[cap] (may be NULL: count only). */
int efo_scan(const uint64_t *words, uint64_t nwords, const int64_t *offsets, int32_t n, int32_t upper_bound, int log2_quantum, int32_t from, int32_t to,
             int64_t *rowptr, int32_t *succ, size_t cap, uint64_t *arcs_out) {
	if (!words || !offsets || from < 0 || to < from || to > n || upper_bound < n || log2_quantum < 0) return EFO_EARG;
	lw_t s = { words, nwords, 0 };
	uint64_t k = 0;
	const uint64_t ub = (uint64_t)upper_bound;
	for (int32_t x = from; x < to; x++) {
		uint64_t pos = (uint64_t)offsets[x];
		const uint64_t d = lw_gamma(&s, &pos); /* outdegree(x), :1056-1061 */
		if (s.err || d > ub) return EFO_EFORMAT;
		if (rowptr) rowptr[x - from] = (int64_t)k;
		const uint64_t len = d + 1;
		const int l = ef_lower_bits(len, ub);                                   /* :1110 */
		const uint64_t np = (ub >> l) >> log2_quantum;                          /* numberOfPointers, :1111 */
		const int ps = ef_ceil_log2(len + (ub >> l)) < 0 ? 0 : ef_ceil_log2(len + (ub >> l)); /* pointerSize, :1112 */
		const uint64_t lowerStart = pos + (uint64_t)ps * np, upperStart = lowerStart + (uint64_t)l * len; /* :1114-1115 */
		uint64_t up = upperStart;
		for (uint64_t i = 0; i < d; i++) { /* nextInt, :1138-1144: position of the i-th one, minus i, are the upper bits */
			const uint64_t zeros = lw_unary(&s, &up);
			(void)zeros;
			const uint64_t high = (up - 1 - upperStart) - i;
			const uint64_t v = (high << l) | lw_get(&s, lowerStart + (uint64_t)l * i, l);
			if (s.err) return EFO_EFORMAT;
			if (succ) { if (k >= cap) return EFO_EARG; succ[k] = (int32_t)v; }
			k++;
		}
	}
	if (rowptr) rowptr[to - from] = (int64_t)k;
	if (arcs_out) *arcs_out = k;
	return s.err ? EFO_EFORMAT : EFO_OK;
}

/* EliasFanoSuccessorReader.skipTo(lowerBound) on a fresh reader of node x (EFGraph.java:1147-1215), for q (node, bound) pairs: the first successor >= bound, or -1
 * at the end of the list.  The forward pointers are used exactly where the reference uses them -- more than `quantum` zeros to skip: block = zeros >> log2Quantum,
 * skip = pointer[block - 1], the reader lands on bit upperBitsStart + skip of the upper bits with skip - (block << log2Quantum) ones behind it (:1164-1173) --, the
 * rest of the way is the plain walk (the reference's broadword select finds the same bit).  used[i] = 1 when query i went through a pointer.  A pointer that does
 * not say what the writer's Accumulator.add (:502-516) must put there sends the reader to the wrong element: the answers are compared with a search in the scanned lists. */
int efo_skip_to(const uint64_t *words, uint64_t nwords, const int64_t *offsets, int32_t n, int32_t upper_bound, int log2_quantum, const int32_t *nodes, const int32_t *bounds,
                size_t q, int32_t *out, uint8_t *used) {
	if (!words || !offsets || !nodes || !bounds || !out || upper_bound < n || log2_quantum < 0) return EFO_EARG;
	lw_t s = { words, nwords, 0 };
	const uint64_t ub = (uint64_t)upper_bound, quantum = (uint64_t)1 << log2_quantum;
	for (size_t i = 0; i < q; i++) {
		const int32_t x = nodes[i];
		if (x < 0 || x >= n || bounds[i] < 0) return EFO_EARG;
		uint64_t pos = (uint64_t)offsets[x];
		const uint64_t d = lw_gamma(&s, &pos);
		if (s.err || d > ub) return EFO_EFORMAT;
		const uint64_t len = d + 1;
		const int l = ef_lower_bits(len, ub);
		const uint64_t np = (ub >> l) >> log2_quantum;
		const int ps = ef_ceil_log2(len + (ub >> l)) < 0 ? 0 : ef_ceil_log2(len + (ub >> l));
		const uint64_t ptrStart = pos, lowerStart = pos + (uint64_t)ps * np, upperStart = lowerStart + (uint64_t)l * len;
		const uint64_t lb = (uint64_t)bounds[i], zeroesToSkip = lb >> l;
		uint64_t up = upperStart, index = 0; /* bit of the upper stream the reader stands on; ones behind it (currentIndex) */
		if (used) used[i] = 0;
		if (zeroesToSkip > quantum && np > 0) { /* delta > quantum with last = Integer.MIN_VALUE, :1164 */
			const uint64_t block = zeroesToSkip >> log2_quantum;
			if (block == 0 || block > np) return EFO_EFORMAT;
			const uint64_t skip = lw_get(&s, ptrStart + (block - 1) * (uint64_t)ps, ps);
			if (skip == 0 || skip < (block << log2_quantum)) return EFO_EFORMAT;
			up = upperStart + skip;
			index = skip - (block << log2_quantum);
			if (used) used[i] = 1;
		}
		int32_t ans = -1;
		while (index < d) { /* nextInt until last >= lowerBound (:1211-1214); the terminator is never returned */
			(void)lw_unary(&s, &up);
			if (s.err) return EFO_EFORMAT;
			const uint64_t high = (up - 1 - upperStart) - index;
			const uint64_t v = (high << l) | lw_get(&s, lowerStart + (uint64_t)l * index, l);
			index++;
			if (v >= lb) { ans = (int32_t)v; break; }
		}
		out[i] = ans;
	}
	return s.err ? EFO_EFORMAT : EFO_OK;
}
