/*
 * bvgtools.h -- host-side (CPU, no GPU needed) companions of the decode path: a BVGraph *writer* and
 * a seeded synthetic graph generator.  They exist because the decode path needs .graph/.offsets/
 * .properties inputs and this environment has no JVM to run the reference's BVGraph.store().
 *
 * the writer follows the reference compressor closely enough that its files load
 * in the real it.unimi.dsi.webgraph.BVGraph:
 *   bvt_store         <->  BVGraph.store / storeInternal / CompressionThread.call / diffComp / intervalize
 *                           (src/it/unimi/dsi/webgraph/BVGraph.java:1679, :2436-2650, :2222-2386, :2049-2219, :1631-1654)
 * Built as libbvgtools.so (g++), independent of libbvgpu.so.
 */
#ifndef BVGTOOLS_H
#define BVGTOOLS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Statistics gathered while storing; mirrors the counters persisted by BVGraph.java:2558-2600. */
typedef struct bvt_store_stats {
	uint64_t written_bits, offsets_bits;
	uint64_t bits_outdegrees, bits_references, bits_blocks, bits_intervals, bits_residuals;
	uint64_t copied_arcs, intervalised_arcs, residual_arcs;
	uint64_t tot_ref, tot_dist;
	int32_t  max_ref_chain;        /* longest reference chain actually produced */
	int32_t  threads;
} bvt_store_stats;

/*
 * Compresses the CSR graph (rowptr[n+1], succ[rowptr[n]], each row strictly increasing) into
 * <basename>.graph / .offsets / .properties.
 *   window, max_ref_count, min_interval, zeta_k : as BVGraph.store(g, basename, w, r, i, k, flags)
 *   flags   : BVGraph flag word (BVGraph.java:475-523, layout :1317-1325); 0 = all defaults
 *   threads : >=1.  As in the reference (BVGraph.java:2471-2550) each thread compresses a contiguous
 *             node range starting with an EMPTY window and the bit streams are concatenated.
 * Returns 0, or a negative errno-style code; `stats` may be NULL.
 */
int bvt_store(const char *basename, int32_t n, const int64_t *rowptr, const int32_t *succ,
              int window, int max_ref_count, int min_interval, int zeta_k, uint32_t flags, int threads,
              bvt_store_stats *stats);

/*
 * Seeded synthetic "power-law + copy model" graph; deterministic for a
 * given (n, m, seed, p_copy) whatever the thread count.  Allocates *rowptr_out (n+1) and *succ_out (m);
 * release both with bvt_free.  The exact recipe is documented in DESIGN.md section "Synthetic workload".
 */
int bvt_generate(int32_t n, int64_t m, uint64_t seed, double p_copy, int threads,
                 int64_t **rowptr_out, int32_t **succ_out);

/* The same generator with two more knobs (config C5): p_same = probability that a node repeats its predecessor's raw
 * outdegree and, when it copies, takes that predecessor as its prototype; p_keep = probability that a prototype's successor
 * is kept (0.7 in C2).  bvt_generate(...) == bvt_generate_ex(..., p_same 0, p_keep 0.7, ...). */
int bvt_generate_ex(int32_t n, int64_t m, uint64_t seed, double p_copy, double p_same, double p_keep, int threads,
                    int64_t **rowptr_out, int32_t **succ_out);

void bvt_free(void *p);

/*
 * The node ids of the random-access leg of the reference's SpeedTest (src/it/unimi/dsi/webgraph/test/SpeedTest.java:79,
 * :98-111): XoRoShiRo128PlusRandom.setSeed(seed), then nextInt(n) `count` times (dsiutils' generator restated; see the
 * source for what is pinned).  Config C4 of BASELINE.json uses seed 0x5EEDB5E70004.
 */
int bvt_random_nodes(uint64_t seed, int32_t n, int64_t count, int32_t *out);

/*
 * Arc labels: writes <basename>.labels / .labeloffsets / .properties as
 * BitStreamArcLabelledImmutableGraph.store does (labelling/BitStreamArcLabelledImmutableGraph.java:650-695), for int
 * labels given per arc in CSR order.  kind 1 = GammaCodedIntLabel (labels >= 0), 2 = FixedWidthIntLabel(width).
 * `underlying` is written verbatim as the `underlyinggraph` property (relative to the property file's directory).
 */
int bvt_store_labels(const char *basename, const char *underlying, int32_t n, const int64_t *rowptr, const int32_t *labels,
                     int kind, int width, const char *key);
/* The same for FixedWidthIntListLabel(key, width) (FixedWidthIntListLabel.java:114-119): arc a (CSR order) carries the list
 * values[listptr[a] .. listptr[a+1]); every value must fit in `width` bits.  listptr: int64[rowptr[n] + 1]. */
int bvt_store_label_lists(const char *basename, const char *underlying, int32_t n, const int64_t *rowptr, const int64_t *listptr,
                          const int32_t *values, int width, const char *key);

/*
 * EFGraph.store (src/it/unimi/dsi/webgraph/EFGraph.java:812-889): the same CSR as <basename>.graph (64-bit words, bits from
 * the low end, `big_endian` chooses the byte order of the words: the `byteorder` property), <basename>.offsets (delta-coded
 * record lengths) and <basename>.properties with graphclass = it.unimi.dsi.webgraph.EFGraph.  upper_bound >= n (the
 * reference's default is n), log2_quantum: the reference's default is 8.
 */
int bvt_store_ef(const char *basename, int32_t n, const int64_t *rowptr, const int32_t *succ, int32_t upper_bound, int log2_quantum, int big_endian);

#ifdef __cplusplus
}
#endif
#endif
