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
 * the writer against a record worked out by hand from the format description (tests/test_efgraph_cpu.py).
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
