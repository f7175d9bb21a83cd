/*
 * bvgpu.h -- C ABI of libbvgpu.so, the MI355X-native BVGraph decompressor.
 *
 * This is the drop-in boundary for ONE hot path of vigna/webgraph: decoding the .graph bit stream of
 * it.unimi.dsi.webgraph.BVGraph into successor lists.  The reference has no FFI (it is pure Java); its
 * plug-in point is the `graphclass` reflection of ImmutableGraph.load (ImmutableGraph.java:647-685).  A Java
 * class `GpuBVGraph extends ImmutableGraph` binds these entry points through JNI (INTEGRATION.md shows the
 * stub); each function below names the reference method it stands in for.  "BVG" =
 * src/it/unimi/dsi/webgraph/BVGraph.java.
 *
 * Conventions
 *   - plain C99 types only; every function returns 0 (BVG_OK) or a negative bvg_status.
 *   - a handle (bvg_t) is NOT thread-safe; bvg_clone() gives a flyweight sharing the immutable device
 *     buffers with its own stream + scratch, like BVGraph.copy() (BVG:552-577, ImmutableGraph.java:157-165).
 *   - output buffers are owned by the caller and may live in host or device memory (BVG_OUT_DEVICE).
 *   - there is NO CPU fallback: every decoding entry point needs a HIP device and fails with BVG_EHIP
 *     without one.  Only the functions marked [host-only] run without a GPU.
 */
#ifndef BVGPU_H
#define BVGPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct bvg_graph bvg_t;

typedef enum bvg_status {
	BVG_OK = 0,
	BVG_EARG = -1,         /* IllegalArgumentException: node / range out of bounds (BVG:860, :900, :1037, :1165) */
	BVG_ESTATE = -2,       /* IllegalStateException: reference > windowsize (BVG:705), no offsets (BVG:869) */
	BVG_EUNSUPPORTED = -3, /* UnsupportedOperationException / IOException: coding id, version > 0, nodes >= 2^31
	                          (BVG:635, :1534, :1537), graphclass mismatch (BVG:1528) */
	BVG_EIO = -4,          /* IOException: missing / short / unreadable files */
	BVG_ENOMEM = -5,       /* host or device allocation failed */
	BVG_EHIP = -6,         /* no HIP device / HIP runtime error */
	BVG_EFORMAT = -7,      /* malformed bit stream detected while decoding (never reads out of bounds) */
	BVG_ECAP = -8          /* caller-provided successor buffer too small (arcs_out still reports the need) */
} bvg_status;

/* coding ids, CompressionFlags.java:26-44 */
enum { BVG_DELTA = 1, BVG_GAMMA = 2, BVG_GOLOMB = 3, BVG_SKEWED_GOLOMB = 4, BVG_UNARY = 5, BVG_ZETA = 6, BVG_NIBBLE = 7 };

/* What BVGraph.loadInternal reads from <basename>.properties (BVG:1528-1543) after setFlags (BVG:1317-1325). */
typedef struct bvg_info {
	int32_t  nodes;            /* numNodes() */
	int64_t  arcs;             /* numArcs() */
	int32_t  window_size;      /* windowSize()  BVG:610 */
	int32_t  max_ref_count;    /* maxRefCount() BVG:618 -- metadata only, never trusted while decoding */
	int32_t  min_interval_length;
	int32_t  zeta_k;
	uint32_t flags;            /* packed compression flags, layout BVG:1317-1325 */
	int32_t  outdegree_coding, block_coding, residual_coding, reference_coding, block_count_coding, offset_coding;
	uint64_t graph_bytes;      /* size of <basename>.graph */
	int32_t  device;           /* HIP device ordinal the handle lives on, -1 for a host-only parse */
	int32_t  offsets_on_device; /* 1: the .offsets stream was decoded by the GPU kernels, 0: by the host decoder */
	int32_t  shard_from, shard_to; /* nodes this handle decodes: [0, nodes) for bvg_open, one slice for bvg_open_shard */
	int32_t  staged_from;          /* first node whose record is staged (shard_from minus the room kept for referents before it) */
	int32_t  format;               /* BVG_FORMAT_BV: graphclass BVGraph; BVG_FORMAT_EF: graphclass EFGraph (the fields from window_size to offset_coding
	                                  do not apply, except offset_coding = delta) */
	int32_t  ef_upper_bound;       /* EFGraph: the `upperbound` property (default: nodes), EFGraph.java:742 */
	int32_t  ef_log2_quantum;      /* EFGraph: log2 of the `quantum` property, :743-745 */
	int32_t  ef_big_endian;        /* EFGraph: `byteorder` = BIG_ENDIAN (the words are swapped once, at load time), :747-750 */
} bvg_info_t;
enum { BVG_FORMAT_BV = 0, BVG_FORMAT_EF = 1 };

/* flags for the *_range / *_batch calls */
enum {
	BVG_OUT_HOST = 0,     /* rowptr / succ / nodes are host pointers (copied through a staging buffer) */
	BVG_OUT_DEVICE = 1,   /* ... are device pointers on the handle's device */
	BVG_ASYNC = 2         /* with BVG_OUT_DEVICE: enqueue on the handle's stream and return without synchronising;
	                         errors and *arcs_out are then delivered by bvg_sync(), which also launches what the
	                         optimistic launch left out (deeper chain levels; a sub-range whose halo was deeper or
	                         larger than guessed is decoded again there): the buffers must stay valid until then */
};

/* ---- lifecycle ------------------------------------------------------------------------------------- */

/* A second on-disk format behind the same handle: when <basename>.properties says
 * graphclass = it.unimi.dsi.webgraph.EFGraph, bvg_open loads the quasi-succinct (Elias-Fano) files EFGraph.store writes
 * (src/it/unimi/dsi/webgraph/EFGraph.java:709-789: 64-bit words in `byteorder`, delta-coded .offsets) and every call of this
 * header that decodes -- bvg_outdegrees, bvg_decode_range[_view], bvg_successors_batch, bvg_scan_checksum, bvg_scan_stats,
 * bvg_bfs_expand, bvg_hyperball_step, bvg_recompress -- works on them: no record of an EFGraph refers to another one, so a
 * range is outdegrees -> scan -> one pass that writes every list (bv_ef.hip).  BVG_ASYNC is accepted and ignored (the call
 * synchronises). */

/* (no counterpart in the reference) Trades HBM for speed: the handle's lists are decoded once, re-encoded on the device as an EFGraph
 * image that stays in HBM (2.3 times the size of a BVGraph stream), and every later call on this handle decodes from that image -- scans
 * 2.3 times, random batches 2.5 times as fast on the C2 lists.  Results are the same lists; bvg_info then reports
 * BVG_FORMAT_EF.  Clones made before the call keep the original image.  BVG_EUNSUPPORTED for a shard handle. */
int bvg_cache_as_efgraph(bvg_t *g);

/* ImmutableGraph.load(basename) -> BVGraph.load -> loadInternal (BVG:1380, :1516-1609): parse .properties,
 * read .graph and .offsets, stage the bit stream and the decoded int64 offset table in HBM on `device`. */
int bvg_open(const char *basename, int device, bvg_t **out);

/* One GPU's share of a graph that is scanned by `parts` GPUs: same files, same bounds as
 * bvg_shard_bounds(parts) -- bounds[k] = min{x : off[x] >= k*off[n]/parts} -- but only the slice of the bit stream and of
 * the offset table that nodes [bounds[part], bounds[part+1]) need is staged in HBM (plus a few thousand nodes before it
 * for the referents of its first rows).  The handle decodes ranges inside its slice (bvg_decode_range, bvg_decode_range_view,
 * bvg_scan_checksum, bvg_outdegrees; node ids stay global); random access needs the whole graph (BVG_EUNSUPPORTED here).
 * There is no exchange step between the parts: a host-side reduction of (arcs, hash) pairs is all a scan needs.
 * The room before the slice is max(4096, 64 x windowsize) nodes: a file whose reference chains run deeper than that (written with a
 * huge or unlimited maxrefcount) decodes through bvg_open only -- a range decode of such a slice returns BVG_EUNSUPPORTED. */
int bvg_open_shard(const char *basename, int device, int part, int parts, bvg_t **out);

/* BVGraph.copy() (BVG:552-577): flyweight sharing the staged graph; own stream and scratch. */
int bvg_clone(const bvg_t *g, bvg_t **out);

int bvg_close(bvg_t *g);

/* numNodes / numArcs / windowSize / maxRefCount ... (ImmutableGraph.java:254-260, BVG:610-620) */
int bvg_info(const bvg_t *g, bvg_info_t *out);

/* message for the last failing call on this handle (never NULL) */
const char *bvg_last_error(const bvg_t *g);

/* Orders this handle's work with the caller's HIP stream (a hipStream_t passed as void*): every call first waits
 * for what the caller has enqueued on it, and the stream waits for the call's results.  The kernels themselves run
 * on the handle's own streams (they overlap better there).  NULL detaches. */
int bvg_set_stream(bvg_t *g, void *hip_stream);

/* Waits for the handle's stream; returns the status of the asynchronous work since the last sync and,
 * if arcs_out != NULL, the arc count of the last range/batch decode. */
int bvg_sync(bvg_t *g, uint64_t *arcs_out);

/* ---- measurement ------------------------------------------------------------------------------------ */

/* Phases of one bvg_decode_range, in stream order (HIP events are recorded between them when profiling is on;
 * the three parse kernels, which normally overlap on forked streams, then run one after the other):
 * headers(+halo closure) | scan | chain depth + work lists | parse of giant records (a group of waves each) |
 * parse of big records (a wave each) | parse of short records (one lane per record) | copy levels |
 * rowptr rebase + totals */
#define BVG_NUM_PHASES 8
/* Enables / disables per-phase HIP-event timing on this handle (off by default; costs a few event records). */
/* Tuning and debug knobs of a handle, by name ("coop_min", "giant_min", "overlap", "copy_mid_min", "hash_materialise", ... -- the names of the BVGPU_<NAME>
 * environment variables, which are read ONCE when a handle is created and never per launch; -DBVGPU_NO_ENV builds ignore the environment altogether and keep only
 * this entry point).  Every knob is a choice of speed, never of results.  Waits for the handle's pending job.  BVG_EARG: unknown name.  No counterpart in the
 * reference (its only run-time knobs are the JVM's). */
int bvg_set_option(bvg_t *g, const char *name, const char *value);
int bvg_set_profile(bvg_t *g, int enable);
/* Milliseconds of each phase of the last range decode issued with profiling on; ms has BVG_NUM_PHASES floats. */
int bvg_get_profile(bvg_t *g, float *ms);

/* The outdegree thresholds the last bvg_decode_range ran with: records with at least coop_min successors were decoded by one
 * wavefront each, with at least giant_min by a group of wavefronts, the others by one lane each.  The first is
 * picked on the device from the job's outdegrees, the second from the job's size; bench.py prices each kernel on its own records. */
int bvg_last_thresholds(bvg_t *g, int32_t *coop_min, int32_t *giant_min);

/* Tuning counters (only when BVGPU_STATS=1 was set at bvg_open): 64 uint64 -- [0, 32) the cooperative decoder's (bv_device.hpp), the rest spare. */
int bvg_debug_stats(bvg_t *g, uint64_t *out16, int reset);

/* ---- the hot path ---------------------------------------------------------------------------------- */

/* outdegree(x) for x in [from, to)  (BVG:858-888).  out has to-from int32. */
int bvg_outdegrees(bvg_t *g, int32_t from, int32_t to, int32_t *out, int flags);

/*
 * Sequential scan of nodes [from, to): what draining nodeIterator(from).copy(to) produces
 * (BVGraphNodeIterator BVG:1136-1281 over successors(x, ibs, window, outd) BVG:1032-1133), in CSR form:
 *   rowptr[to-from+1]  exclusive prefix sums of the outdegrees, rowptr[0] = 0
 *   succ[rowptr[to-from]]  the successor lists, each strictly increasing
 * succ == NULL: count-only (rowptr and *arcs_out).  succ_cap is the capacity of succ in int32 elements; a smaller
 * result is BVG_ECAP (nothing is written past succ_cap, *arcs_out reports the need).  rowptr is required.
 * from > 0: referents before `from` are decoded into library scratch (a halo), never into the caller's buffers.
 */
int bvg_decode_range(bvg_t *g, int32_t from, int32_t to, int64_t *rowptr, int32_t *succ, size_t succ_cap,
                     uint64_t *arcs_out, int flags);

/*
 * The same scan with the results in pinned host memory OWNED BY THE HANDLE: one call, no counting call before it, the
 * successors cross PCIe chunk by chunk while the next chunk is being decoded.  *rowptr_out (to-from+1 entries) and
 * *succ_out (*arcs_out entries) stay valid until the next call on this handle.  What a JNI caller wants: one native call,
 * then NewIntArray + SetIntArrayRegion (or a direct ByteBuffer over the pinned memory).
 * bvg_decode_range with BVG_OUT_HOST runs the same pipeline into the caller's buffers (at PCIe speed when they are
 * pinned -- bvg_host_alloc --, through a pinned ring and host threads when they are pageable).
 */
int bvg_decode_range_view(bvg_t *g, int32_t from, int32_t to, const int64_t **rowptr_out, const int32_t **succ_out, uint64_t *arcs_out);

/* Pinned host memory for output buffers (hipHostMalloc); release with bvg_host_free. */
int bvg_host_alloc(size_t bytes, void **out);
void bvg_host_free(void *p);

/*
 * A scan that hands nothing back but its fingerprint -- the consumer every test of the reference is
 * (ImmutableGraph.equals / hashCode, ImmutableGraph.java:731-770) and the first of the no-materialise modes (SURVEY.md
 * section 8 row f4): continues ImmutableGraph.hashCode() from *hash_io over nodes [from, to) and counts their arcs.  The rows
 * are decoded piece by piece into library scratch that stays on the die; no 4 B/edge array reaches the caller.  On an EFGraph
 * handle the fold happens inside the decode kernels: no successor is written at all (4 B per node of sums instead of 4 B per arc).
 * Shards compose: h(whole) = fold of the shards' maps in node order (webgraph_amd/parallel.py).
 */
int bvg_scan_checksum(bvg_t *g, int32_t from, int32_t to, int32_t *hash_io, uint64_t *arcs_out);

/*
 * ImmutableGraph.equals() (ImmutableGraph.java:731-749) restricted to nodes [from, to): *equal = 1 iff both handles give every node of the range the same
 * outdegree and the same successors.  Both graphs are decoded piece by piece into the scratch of their handles and compared on the device: no row reaches
 * the host (the mirrors' equals() ran two host-bound scans and a host comparison: 329 ms for C2 against 8 here).  The handles must live on the same device
 * and hold at least `to` nodes each (BVG_EARG otherwise); they may be the same handle or clones.
 */
int bvg_equal_range(bvg_t *a, bvg_t *b, int32_t from, int32_t to, int *equal);

/*
 * Two more consumers that never hand the caller a successor array; both work a chunk of the
 * graph at a time on rows decoded into library scratch.
 *
 * bvg_scan_stats: the scan of Stats.run (src/it/unimi/dsi/webgraph/Stats.java:111-160) over nodes [from, to).  tot_gap and
 * tot_loc are BigIntegers in the reference; 64 bits hold them for every graph whose ids fit an int.  indegree_dev: NULL, or a
 * DEVICE array of `nodes` int32 that is incremented once per arc (Stats.java:130; zero it first).
 */
typedef struct bvg_scan_stats {
	uint64_t nodes, arcs, loops, dangling, terminal, num_gaps;
	uint64_t tot_gap;   /* sum over nodes with d > 1 of (last - first) + int2nat(first - node)   Stats.java:119-122 */
	uint64_t tot_loc;   /* sum over arcs of |successor - node|                                   :125 */
	int32_t  min_outdegree, max_outdegree, min_outdegree_node, max_outdegree_node; /* :140-148 (first node in node order) */
	uint64_t successor_delta_stats[32]; /* arcs with successor != node, by mostSignificantBit(|node - successor|)   :127 */
} bvg_scan_stats_t;
int bvg_scan_stats(bvg_t *g, int32_t from, int32_t to, bvg_scan_stats_t *out, int32_t *indegree_dev);

/*
 * One round of ParallelBreadthFirstVisit (src/it/unimi/dsi/webgraph/algo/ParallelBreadthFirstVisit.java:146-170): for
 * every node x of the frontier and every successor s of x, marker.compareAndSet(s, -1, parent ? x : round); the successors
 * that were still unmarked form the next frontier (in no particular order, as in the reference, where it depends on the
 * threads).  All pointers are DEVICE pointers: frontier[q], marker[nodes] (-1 = not enqueued yet), out[out_cap]; *out_count
 * (host) receives the size of the next frontier -- if it exceeds out_cap only out_cap entries were written (BVG_ECAP).
 */
int bvg_bfs_expand(bvg_t *g, const int32_t *frontier_dev, size_t q, int32_t *marker_dev, int32_t round, int parent,
                   int32_t *out_dev, size_t out_cap, uint64_t *out_count);

/*
 * One standard iteration of HyperBall over nodes [from, to) (src/it/unimi/dsi/webgraph/algo/HyperBall.java:875-915, :972-978): for every node x,
 * regs_out[x] = register-wise maximum of regs_in[x] and regs_in[s] over the successors s != x whose counter changed in the previous iteration
 * (modified_in[s] != 0; modified_in == NULL: every counter counts, as in the first iteration), and modified_out[x] = (regs_out[x] != regs_in[x]).
 * A counter is 2^log2m registers of one byte each (HyperLogLog registers are at most 7 bits wide; the reference packs them into longwords).
 * All pointers are DEVICE pointers: regs_in / regs_out [nodes << log2m] bytes, modified_in / modified_out [nodes] bytes; only the entries of
 * [from, to) are written.  *changed (host) receives the number of counters of [from, to) that changed.  The rows are decoded piece by piece
 * into library scratch; no successor array reaches the caller.
 */
int bvg_hyperball_step(bvg_t *g, int32_t from, int32_t to, int log2m, const uint8_t *regs_in_dev, uint8_t *regs_out_dev, const uint8_t *modified_in_dev,
                       uint8_t *modified_out_dev, uint64_t *changed);

/*
 * Random access: concatenation of successorArray(nodes[i]) (BVG:897-904, ImmutableGraph.java:329-333),
 * reference chains resolved on the device.  rowptr has q+1 entries; ids may repeat and come in any order; an id
 * outside [0, n) is BVG_EARG (BVG:900).  succ == NULL: count-only; BVG_ECAP as above, checked before any decode.
 * A batch that touches a good part of the graph is decoded as one masked scan (every needed record once) plus a
 * gather, a sparse one query by query: batch the ids, one call per id is the slow way to use this entry point.
 */
int bvg_successors_batch(bvg_t *g, const int32_t *nodes, size_t q, int64_t *rowptr, int32_t *succ, size_t succ_cap,
                         uint64_t *arcs_out, int flags);

/*
 * Fingerprint of a scan, computed on the device from an already decoded CSR range (device pointers):
 * continues ImmutableGraph.hashCode() (ImmutableGraph.java:757-770) from *hash_io over nodes [from, to)
 * (pass -1 and the whole graph to get hashCode()).
 */
int bvg_csr_hashcode(bvg_t *g, int32_t from, int32_t to, const int64_t *rowptr_dev, const int32_t *succ_dev,
                     int32_t *hash_io);

/* ---- sharding (multi-GPU) -------------------------------------------------- */

/* Splits [0, n) into `parts` contiguous node ranges with balanced compressed bits:
 * bounds[k] = min{x : off[x] >= k*off[n]/parts}, bounds[0] = 0, bounds[parts] = n.  Uses the staged offsets. */
int bvg_shard_bounds(const bvg_t *g, int parts, int32_t *bounds);

/* ---- [host-only] helpers (no GPU needed) ----------------------------------------------------------- */

/* Parses <basename>.properties exactly as BVG:1528-1543 does (including the error cases). */
int bvg_parse_properties(const char *basename, bvg_info_t *out, char *errbuf, size_t errlen);

/* string2Flags (BVG:1352-1366): "A | B" of BVGraph constant names -> flag word; -1 on an unknown name. */
int64_t bvg_flags_from_string(const char *s);

/* OffsetsLongIterator (BVG:907-935): decodes n+1 gamma/delta coded gaps of a .offsets image into running sums. */
int bvg_decode_offsets_host(const uint8_t *offsets_file, size_t len, int32_t nodes, int offset_coding, int64_t *out);

/* [device] The same decode on the GPU (bv_offsets.hip; what bvg_open uses): OffsetsLongIterator (BVG:907-935) as a grid-wide
 * cooperative decode of the gap stream, gamma- or delta-coded.  `out` is a HOST array of nodes + 1 values.
 * BVG_EFORMAT when the stream does not hold exactly nodes + 1 codes. */
int bvg_decode_offsets_device(int device, const uint8_t *offsets_file, size_t len, int32_t nodes, int offset_coding, int64_t *out);

/* ---- the compressor --------------------------------------------------------------
 * BVGraph.store(graph, basename, windowSize, maxRefCount, minIntervalLength, zetaK, flags, numberOfThreads)
 * (BVGraph.java:1679-1730 -> storeInternal :2436-2650; per node CompressionThread.call :2222-2386, diffComp :2049-2219,
 * intervalize :1631-1654) on the GPU.  The graph comes as a CSR: rowptr int64[n + 1] (rowptr[0] = 0), succ
 * int32[rowptr[n]], every row strictly increasing (BVG_EARG otherwise); in_flags: BVG_OUT_HOST (0) = host pointers,
 * BVG_OUT_DEVICE = device pointers on `device` (e.g. what bvg_decode_range just wrote: recompression without a trip to
 * the host).  The streams are the reference's byte for byte: the choice of the reference (first cheapest candidate whose
 * chain is shorter than maxRefCount, :2313-2327), copy blocks, intervals, residuals, codings by `flags` (layout
 * :1317-1325, 0 = defaults); `threads` > 1 reproduces the reference's multi-threaded store (contiguous ranges of
 * ceil(n / threads) nodes, each starting with an empty window, streams concatenated, :2471-2550) -- it does not change how
 * the GPU works.  Limits: windowSize <= 63, records shorter than 2^31 bits (BVG_EUNSUPPORTED). */
typedef struct bvg_store_stats {   /* the counters BVGraph.java:2558-2632 persists in .properties */
	uint64_t written_bits, offsets_bits;
	uint64_t bits_outdegrees, bits_references, bits_blocks, bits_intervals, bits_residuals;
	uint64_t copied_arcs, intervalised_arcs, residual_arcs;
	uint64_t tot_ref, tot_dist;    /* avgref = tot_ref / n, avgdist = tot_dist / n */
	int32_t  max_ref_chain;        /* longest reference chain produced */
	int32_t  threads;
	int32_t  selection_rounds;     /* measurement only: rounds the reference-selection recurrence took to settle */
	int32_t  reserved;
	/* successorGapStats / residualGapStats (updateBins, BVGraph.java:1940-1944; :2303, :2196): the gaps of every successor list, and of
	 * every list of residuals, counted by their most significant bit -- the first element of a list by int2nat(first - node), skipped
	 * when that is 0.  .properties carries them as successorexpstats / residualexpstats and the averages derived from them (:2592-2632). */
	uint64_t successor_gap_bins[32], residual_gap_bins[32];
} bvg_store_stats_t;
typedef struct bvg_compressed {    /* result of bvg_compress, in HBM of `device`; release with bvg_compressed_free */
	int32_t  device, reserved;
	uint8_t *graph_dev;            /* the bytes of <basename>.graph: (graph_bits + 7) / 8 of them, zero padded to 32 more */
	uint64_t graph_bits;
	uint8_t *offsets_stream_dev;   /* the bytes of <basename>.offsets */
	uint64_t offsets_bits;
	int64_t *bit_offsets_dev;      /* int64[n + 1]: bit offset of every record = the decoded .offsets */
	bvg_store_stats_t stats;
} bvg_compressed_t;
int bvg_compress(int device, int32_t n, const int64_t *rowptr, const int32_t *succ, int in_flags, int window, int max_ref_count, int min_interval, int zeta_k,
                 uint32_t flags, int threads, bvg_compressed_t *out, char *errbuf, size_t errlen);
void bvg_compressed_free(bvg_compressed_t *c);
/* Copies the result to host memory: graph_host (graph_bits + 7) / 8 bytes, offsets_host (offsets_bits + 7) / 8 bytes,
 * bit_offsets_host int64[n + 1]; any of the three may be NULL. */
int bvg_compressed_copy(const bvg_compressed_t *c, int32_t n, uint8_t *graph_host, uint8_t *offsets_host, int64_t *bit_offsets_host);
/* bvg_compress + the three files <basename>.graph / .offsets / .properties (the properties as :2558-2632 writes them). */
int bvg_store(const char *basename, int device, int32_t n, const int64_t *rowptr, const int32_t *succ, int in_flags, int window, int max_ref_count,
              int min_interval, int zeta_k, uint32_t flags, int threads, bvg_store_stats_t *stats, char *errbuf, size_t errlen);

/* BVGraph.store(graph, ...) when `graph` is a handle of this library: the whole graph is decoded into HBM (bvg_decode_range,
 * BVG_OUT_DEVICE) and compressed from there with the new parameters; nothing but the three result files leaves the device.
 * BVG_EUNSUPPORTED for a shard handle. */
int bvg_recompress(bvg_t *g, const char *basename, int window, int max_ref_count, int min_interval, int zeta_k, uint32_t flags, int threads,
                   bvg_store_stats_t *stats, char *errbuf, size_t errlen);

/* EFGraph.store(graph, upperBound, basename, log2Quantum, cacheSize, byteOrder, pl) (EFGraph.java:812-889; Accumulator :420-552) on the
 * GPU: the CSR (as for bvg_store) becomes <basename>.graph (64-bit words, bits from the low end, in the byte order asked for),
 * <basename>.offsets (delta-coded record lengths) and <basename>.properties with graphclass = it.unimi.dsi.webgraph.EFGraph.
 * upper_bound: 0 = the number of nodes (the reference's default), otherwise >= n; log2_quantum: the reference's default is 8.
 * Every piece of a record has a position that follows from the outdegree and the value: one lane per arc, one per forward pointer.
 * bvg_recompress_ef: the same for a graph that is a handle of this library, in either format (BVGraph -> EFGraph on the device). */
int bvg_store_ef(const char *basename, int device, int32_t n, const int64_t *rowptr, const int32_t *succ, int in_flags, int32_t upper_bound, int log2_quantum, int big_endian,
                 char *errbuf, size_t errlen);
int bvg_recompress_ef(bvg_t *g, const char *basename, int32_t upper_bound, int log2_quantum, int big_endian, char *errbuf, size_t errlen);

/* ---- arc labels ----------------------------------------------------------------
 * labelling/BitStreamArcLabelledImmutableGraph.java:60-135: <basename>.properties names the underlying graph and the
 * label class (`underlyinggraph`, `labelspec`), <basename>.labels holds the labels of all arcs in enumeration order as
 * one bit stream, <basename>.labeloffsets the gamma-coded lengths of the per-node label lists (:652-671).  Supported
 * label classes: GammaCodedIntLabel (GammaCodedIntLabel.java:60-64), FixedWidthIntLabel (FixedWidthIntLabel.java:70-73) and
 * FixedWidthIntListLabel (FixedWidthIntListLabel.java:107-112, a list of ints per arc: bvg_labels_decode_lists).
 * The labels of the arcs of nodes [from, to) come out in the CSR order of bvg_decode_range on the underlying graph. */
typedef struct bvg_labels bvg_labels_t;
enum { BVG_LABEL_GAMMA = 1, BVG_LABEL_FIXED = 2, BVG_LABEL_FIXED_LIST = 3 };
typedef struct bvg_labels_info {
	int32_t  kind;             /* BVG_LABEL_GAMMA / BVG_LABEL_FIXED / BVG_LABEL_FIXED_LIST */
	int32_t  width;            /* bits per label (BVG_LABEL_FIXED) or per list element (BVG_LABEL_FIXED_LIST) */
	int32_t  nodes;            /* nodes of the underlying graph */
	int32_t  device;
	uint64_t labels_bytes;     /* size of <basename>.labels */
	uint64_t labels_bits;      /* last label offset = bits actually used */
	char     underlying[1024]; /* resolved basename of the underlying graph (open it with bvg_open) */
	char     key[128];         /* the label's key (first constructor argument) */
} bvg_labels_info_t;

/* BitStreamArcLabelledImmutableGraph.load (:383-470): parse the properties, stage .labels and the decoded label offsets
 * in HBM.  `nodes` must be the underlying graph's node count (the label files do not record it). */
int bvg_labels_open(const char *basename, int32_t nodes, int device, bvg_labels_t **out);
void bvg_labels_close(bvg_labels_t *h);
int bvg_labels_info(const bvg_labels_t *h, bvg_labels_info_t *out);
const char *bvg_labels_last_error(const bvg_labels_t *h);
/* [host-only] the properties of a labelled graph without touching a GPU: fills kind / width / underlying / key. */
int bvg_labels_parse_properties(const char *basename, bvg_labels_info_t *out, char *errbuf, size_t errlen);
/* Labels of the `arcs` arcs of nodes [from, to) (arcs = rowptr[to] - rowptr[from] of the underlying graph; checked against
 * the stream: BVG_EFORMAT if the stream holds another number of labels).  flags: BVG_OUT_HOST or BVG_OUT_DEVICE. */
int bvg_labels_decode_range(bvg_labels_t *h, int32_t from, int32_t to, uint64_t arcs, int32_t *labels, int flags);
/* FixedWidthIntListLabel.fromBitStream (FixedWidthIntListLabel.java:107-112: value = new int[readGamma()], then readInt(width)
 * each): the lists of the `arcs` arcs of nodes [from, to) as a CSR over the arcs -- the list of arc k (CSR order of
 * bvg_decode_range) is values[list_ptr[k] .. list_ptr[k+1]).  list_ptr: int64[arcs + 1]; values: int32[values_cap].
 * *nvalues reports the number of values of the range; BVG_ECAP if it exceeds values_cap (nothing is written to `values`
 * then: call with values_cap = 0 to size the buffer).  BVG_EFORMAT if the stream does not hold exactly `arcs` lists ending on
 * the node boundaries of .labeloffsets; BVG_EUNSUPPORTED for the one-int-per-arc label classes (and bvg_labels_decode_range
 * answers BVG_EUNSUPPORTED for this one).  flags: BVG_OUT_HOST or BVG_OUT_DEVICE (both arrays). */
int bvg_labels_decode_lists(bvg_labels_t *h, int32_t from, int32_t to, uint64_t arcs, int64_t *list_ptr, int32_t *values, uint64_t values_cap,
                            uint64_t *nvalues, int flags);

#ifdef __cplusplus
}
#endif
#endif
