// bv_launch.hpp -- host-visible launch interface of bv_kernels.hip (internal to libbvgpu.so).
#pragma once
#include "bv_device.hpp"
#include "bv_encode.hpp"

#include <string>

#include <cstdlib>
// the only door to the environment (debug / tuning knobs; see apply_option in bvgpu_api.cpp): -DBVGPU_NO_ENV closes it
inline const char *bv_env(const char *name) {
#if defined(BVGPU_NO_ENV)
	(void)name; return nullptr;
#else
	return getenv(name);
#endif
}

namespace bv {

// work-list keys (bv_kernels.hip, "work lists")
#ifndef BIN_SUB_ // (tuning builds) bits of a work bin below the octave: 1 = half-octave steps (the lanes of a wave differ by < 1.41x), 2 = quarter-octave steps (< 1.19x)
#define BIN_SUB_ 1
#endif
constexpr int BIN_SUB = BIN_SUB_, NBIN = 12 << BIN_SUB, MAXLVL = 64, NKEYS = NBIN * MAXLVL;
constexpr uint16_t KEY_NONE = 0xffff, KEY_GIANT = 0xfffe;

// A scan that folds ImmutableGraph.hashCode() (ImmutableGraph.java:757-770) instead of handing the rows to anybody (bvg_scan_checksum).  The hash of the
// sequence "node, its successors backwards, next node ..." (L values) is 31^L h0 + sum(value_i * 31^(L - 1 - i)) mod 2^32.  With u = 31^-1 mod 2^32 that is
// 31^L h0 + 31^(L + c) * sum(value_i * u^(i + c + 1)) for any c: taking c = rowstart[nh] + nh, a value's weight depends on nothing but the row starts as the kernels see
// them -- successor j (ascending) of slot s weighs u^(1 + rowstart[s + 1] + s) * 31^j, the node's own number u^(1 + rowstart[s] + s) -- and any kernel can add to the
// sum in any order: k_scan_apply adds the node numbers while it writes the row starts, the one-lane parse the rows without a reference as it decodes them (writing
// only those that some row of the view copies from: mark), the lane class of the copy pass the rows it merges; the rows of the wave / group classes are added from
// memory, found through their work lists (k_hash_queue).  The host multiplies by 31^(L + c) when the job's status comes home.
struct HashCtx {
	const uint8_t *mark;  // [cnt] 1: some row of the view copies from this row (k_headers)
	uint32_t *acc;        // the sum, in HASH_ACC_SLOTS parts (zeroed per job; a wave adds to part (block + wave) mod HASH_ACC_SLOTS: same-address atomics run at ~88 M/s, and 40 000 waves adding to ONE word made k_scan_apply last 0.45 ms instead of 0.05)
	const uint32_t *ptab; // u^e for e mod 2^30 (the order of any odd number divides 2^30): ptab[e & 1023] * ptab[1024 + ((e >> 10) & 1023)] * ptab[2048 + ((e >> 20) & 1023)]
};
constexpr int HASH_ACC_SLOTS = 256;
__device__ __forceinline__ void hash_add(const HashCtx &hx, uint32_t v) { if (v) atomicAdd(hx.acc + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (HASH_ACC_SLOTS - 1)), v); }
__device__ __forceinline__ uint32_t hash_upow(const uint32_t *__restrict__ ptab, uint64_t e) {
	return ptab[e & 1023u] * ptab[1024u + ((e >> 10) & 1023u)] * ptab[2048u + ((e >> 20) & 1023u)];
}

// A decode job over consecutive nodes: slot s <-> node lo+s; slots [0,nh) are halo nodes whose rows live
// in `halo`, slots [nh,cnt) are the caller's nodes whose rows live in `succ`.
struct RangeView {
	int32_t lo, cnt, nh;
	int32_t *outd;          // [cnt] effective outdegree (0 for halo nodes nobody needs)
	uint16_t *ref;          // [cnt] reference distance, 0 = none
	int64_t *rowstart;      // [cnt+1] exclusive scan of outd
	int32_t *succ;          // caller rows
	int32_t *halo;          // halo rows
	uint64_t succ_cap;      // capacity of succ in elements
	int32_t coop_min;       // records with outdegree >= coop_min are decoded by whole waves (k_parse_big) ...
	const int32_t *coop_ptr; // ... unless the job picks the threshold on the device (k_pick_coop): then it is read from here
	uint64_t halo_cap;      // capacity of halo in elements (a sub-range is decoded before the size of its halo is known on the host)
	const HashCtx *hx;      // null, or (device memory) the hash fold of bvg_scan_checksum: see HashCtx
	// does row s lie inside the buffer it belongs to?  (rows are laid out in node order: if s fits, so does every row before it in the same buffer)
	__device__ __forceinline__ bool fits(int32_t s) const {
		return s >= nh ? (uint64_t)(rowstart[s + 1] - rowstart[nh]) <= succ_cap : (uint64_t)rowstart[s + 1] <= halo_cap;
	}
	__device__ __forceinline__ int32_t *row(int32_t s) const {
		const int64_t o = rowstart[s];
		return s < nh ? halo + o : succ + (o - rowstart[nh]);
	}
	__device__ __forceinline__ int32_t coopmin() const { return coop_ptr ? *coop_ptr : coop_min; }
};


// A batch of random-access queries: slot s holds one node of a query's reference chain (see bv_kernels.hip).
struct BatchView {
	int64_t cnt;
	const int32_t *node, *outd, *depth, *qidx; // [cnt]
	const int64_t *arow;                       // [cnt+1] arena row starts (0-length for query slots)
	const int64_t *rowptr;                     // [q+1] caller rows
	int32_t *succ, *arena;
	uint64_t succ_cap;
	int32_t coop_min;                          // slots with outdegree >= coop_min are decoded by whole waves (k_parse_big)
	__device__ __forceinline__ int32_t *row(int64_t s) const { const int32_t qi = qidx[s]; return qi >= 0 ? succ + rowptr[qi] : arena + arow[s]; }
};

void launch_headers(const GraphDev &g, int def, int32_t lo, int32_t cnt, int32_t *outd, uint16_t *ref, int *err, hipStream_t st, int32_t *part = nullptr, uint8_t *mark = nullptr, // mark[cnt] (zeroed): set for every referent
                    uint16_t *pkey16 = nullptr, int32_t *phist = nullptr, bool pwindows = true); // pkey16 / phist: the parse list's keys and their histogram (zeroed here), as k_depth_keys with noBin = 6 (pwindows) / 2 writes them
void launch_scatter_lists(int32_t cnt, const uint16_t *key16, const int32_t *hist, int32_t *keyBase, int32_t *cursor, int32_t *list, int32_t *giantlist, int32_t *ctl, int32_t *maxdepth, hipStream_t st, const uint16_t *packRef = nullptr); // packRef: bvg ref[] -> entries carry min(ref, 15) in bits 28 .. 31 (k_parse_list's list, fewer than 2^28 slots)
int64_t headers_blocks(int32_t cnt); // part: 5 counts per block of k_headers, [5][headers_blocks(cnt)] (input of k_pick_coop)
void launch_mark_halo(int32_t nh, int32_t cnt, int32_t W, int32_t *outd, uint16_t *ref, uint8_t *need, int *err, hipStream_t st);
void launch_scan(const int32_t *in, int64_t n, int64_t *out, int64_t *sums, hipStream_t st, const HashCtx *hx = nullptr, int32_t lo = 0, int32_t nh = 0, long long topTiledMin = -1); // topTiledMin: block sums from which the top level runs tiled (-1: default) // hx: the node numbers of slots >= nh are added to the hash (HashCtx)
int64_t scan_num_sums(int64_t n);
void launch_rebase(int32_t nh, int32_t cnt, const int64_t *rowstart, int64_t *out, hipStream_t st);
bool launch_query_mark(const int32_t *nodes, int64_t q, int32_t n, int32_t *outd, uint16_t *ref, uint8_t *need, int32_t *qoutd, int passes, int32_t *changed, int *err, hipStream_t st);
void launch_query_walk(const int32_t *nodes, int64_t q, int32_t n, int32_t *outd, uint16_t *ref, uint8_t *need, int32_t *qoutd, int *err, hipStream_t st);
void launch_need_prop(int32_t n, const int32_t *outd, const uint16_t *ref, uint8_t *need, int passes, int32_t *changed, hipStream_t st);
void launch_apply_need(int32_t n, const uint8_t *need, int32_t *outd, uint16_t *ref, hipStream_t st);
void launch_gather_rows(const int32_t *nodes, int64_t q, int64_t arcs, const int64_t *rowstart, const int32_t *arena, const int64_t *rowptr, int32_t *succ, hipStream_t st);
void launch_chain_len(const GraphDev &g, int def, const int32_t *nodes, int64_t q, int32_t *chainlen, int32_t *maxlen, int *err, hipStream_t st);
void launch_chain_fill(const GraphDev &g, int def, const int32_t *nodes, int64_t q, const int64_t *slotbase, int32_t *snode, int32_t *soutd,
                       int32_t *sdepth, int32_t *sq, int32_t *aoutd, int32_t *qoutd, hipStream_t st);
void launch_bparse(const GraphDev &g, int def, const BatchView &v, int *err, hipStream_t st);
void launch_bcopy(const GraphDev &g, int def, const BatchView &v, int32_t level, int *err, hipStream_t st);
void launch_pick_coop(const int32_t *part, int32_t nblocks, int32_t budget, int32_t *ctl, hipStream_t st, int32_t *counts = nullptr);
constexpr int PICK_LEVELS = 7; // outdegree classes counted by k_headers / k_pick_coop: >= 128, 256, ..., 8192 successors
constexpr int CTL_INTS = 32, CTL_COOP = 22, CTL_SEG = 24, CTL_GIANT_STARTED = 31, CTL_TOTAL_INTS = CTL_INTS; // control block (bv_kernels.hip); ctl[CTL_SEG], ctl[CTL_SEG + 2]: records the segment pipeline hands to the cooperative kernel, head of that queue
void launch_classify(int32_t cnt, const int32_t *outd, const int32_t *coopPtr, int32_t coopMin, int32_t giantMin, int32_t *biglist, int32_t *giantlist, int32_t giantCap, int32_t *ctl, hipStream_t st);
void launch_parse_big(const GraphDev &g, int def, const RangeView &v, const int32_t *biglist, const int32_t *giantlist, int32_t *ctl, void *arena, int64_t arenaCap,
                      int waves, int giantGroups, int *err, hipStream_t stGiant, hipStream_t stBig, bool waitGiants = true);
constexpr int ARENA_ENTRY_BYTES = 16;
void launch_build_lists(const GraphDev &g, const RangeView &v, uint64_t giantBits, int32_t noBin, int32_t *depth, uint16_t *key16, int32_t *hist, int32_t *keyBase, int32_t *cursor,
                        int32_t *list, int32_t *giantlist, int32_t giantCap, int32_t *ctl, int32_t *maxdepth, hipStream_t st,
                        int32_t *bigQ = nullptr, int32_t bigCap = 0, int32_t *midQ = nullptr, int32_t midCap = 0, int32_t midMinKnob = 0, bool bigGroups = false, const uint16_t *packRef = nullptr);
void launch_copy_level(const GraphDev &g, int def, const RangeView &v, const int32_t *depth, const int32_t *list, const int32_t *keyBase, int32_t level, int blocks,
                       int32_t midMinKnob, bool bigGroups, const int32_t *bigQ, int32_t bigCap, const int32_t *midQ, int32_t midCap, int32_t *ctl, int32_t *tmp, uint32_t tmpCap, int *err,
                       hipStream_t st, hipStream_t stMid, hipStream_t stBig, hipEvent_t evFork, hipEvent_t evMid, hipEvent_t evBig, const void *preDesc = nullptr, bool preMid = false, int listMode = false,
                       const void *tabArena = nullptr, int64_t tabArenaCap = 0, const void *copyTab = nullptr); // copyTab: 16 bytes per slot, the copy blocks of the rows that the one-lane parse decoded (parse_node_lwc / parse_node_tile; bv_lanewin.hpp), blocks from the fourth on in tabArena = the interval arena; null: the lane class walks the stream // vecList: the lane class merges with 16-byte loads and stores (copy_node_v) // preDesc: launch_copy_prewalk's descriptors
void launch_copy_prewalk(const GraphDev &g, int def, const RangeView &v, const int32_t *bigQ, int32_t bigCap, const int32_t *ctl, void *desc, int blocks, hipStream_t st, int32_t midCap, hipStream_t stLong, bool longKernel, hipStream_t stWalk); // stWalk: the stream of k_copy_prewalk (as stLong) // stLong: the stream of the long lists' kernel (ordered behind the queues by the caller; may be st); // midCap > 0: also the wave class's rows (queue at bigQ + bigCap, descriptors at desc + bigCap)
void launch_parse_list(const GraphDev &g, int def, const RangeView &v, const int32_t *list, const int32_t *keyBase, int blocks, int *err, hipStream_t st, void *arena, int64_t arenaCap, int32_t keyLo = 0, int32_t keyHi = NKEYS, bool lwc = true, void *copyTab = nullptr, bool packed = false); // lwc: round 6's loop (parse_node_lwc), which leaves the tables in copyTab; false: round 4's // the list's keys [keyLo, keyHi); v.hx (default codings only): hash fold
// what the decoding kernels did not add to v.hx->acc, from memory: what bit 2 = the node numbers, bit 0 = every row without a reference that the one-lane parse did not
// hash (a pass over all nodes: only when the rows are not in a list), bit 1 = the same for the rows with a reference and the lane class of the copy pass; qA / qB: work
// lists whose rows are hashed (those with a reference if wantRef, those without otherwise); inParse / inCopy: the one-lane parse / the lane class of the copy pass
// hashed their rows themselves; pieceq: cap entries of 8 bytes, *npieces is zeroed here
void launch_hash_sum(const HashCtx *hx, int32_t *out, hipStream_t st); // *out = the sum of the parts
void launch_hash_rest(const RangeView &v, int what, bool inParse, bool inCopy, int32_t midMinKnob, bool bigGroups, void *pieceq, int32_t *npieces, int32_t cap, hipStream_t st,
                      const int32_t *qA = nullptr, const int32_t *nA = nullptr, int32_t capA = 0, const int32_t *qB = nullptr, const int32_t *nB = nullptr, int32_t capB = 0, bool wantRef = false);
// one wave per record of `list` (ctl[which] entries, queue head ctl[which + 2]): k_parse_big<1>
void launch_wait_giants(const int32_t *ctl, int giantGroups, hipStream_t st); // holds st until the giants' groups are on their CUs (or 30 us have passed)
void launch_parse_listed(const GraphDev &g, int def, const RangeView &v, const int32_t *list, int32_t *ctl, int which, void *arena, int64_t arenaCap, int waves, int *err, hipStream_t st);
constexpr int PARSE_LONG_BIN = 7 << BIN_SUB; // work bins from here up (>= 2048 bits of work) are not windowed (k_depth_keys)
// bv_seg.hip: the segment pipeline -- the residual sections of the hubs (giant records with >= minD successors), handed over by k_parse_big
size_t seg_scratch_bytes(int32_t Rtot, int32_t Scap, int zetaK);
void launch_seg_sizing(const int64_t *offsets, int32_t lo, int32_t n, const int32_t *outd, const uint16_t *ref, unsigned long long *out5, hipStream_t st); // out5 (zeroed by the caller): records and bits of the long bins, longest record, rows and ids of the copy pass's lane class
constexpr int SIZING_OCTAVES = 24, SIZING_WORDS = 8 + 2 * SIZING_OCTAVES; // launch_seg_sizing: out5[8 + 2 k], out5[9 + 2 k] = records and arcs with 2^(7 + k) <= outdegree < 2^(8 + k)
int32_t seg_bits_log2();
void seg_handover(GraphDev &g, void *scratch, int32_t capGiant, int32_t Scap, int32_t minD, hipStream_t st); // capGiant hand-over slots, one per entry of the giants' queue; minD: records with fewer successors are not handed over
void launch_seg_chain(const GraphDev &g, int def, const RangeView &v, int32_t Rtot, int32_t Scap, void *scratch, void *arena, int64_t arenaCap, int32_t *ctl, int blocks, int *err, hipStream_t st);
void launch_parse_waves(const GraphDev &g, int def, const RangeView &v, const int32_t *biglist, int32_t *ctl, void *arena, int64_t arenaCap, int waves, int *err, hipStream_t st);
void launch_parse_giants(const GraphDev &g, int def, const RangeView &v, const int32_t *giantlist, int32_t *ctl, void *arena, int64_t arenaCap, int giantGroups, int *err, hipStream_t st);
// bv_tile.hpp: short records decoded tile by tile from one LDS image of a contiguous slice of the stream
int32_t tile_count(int64_t bitSpan, int32_t cnt);
void launch_tile_bounds(const GraphDev &g, int32_t lo, int32_t cnt, int32_t ntiles, int32_t *tb, hipStream_t st);
void launch_parse_tile(const GraphDev &g, int def, const RangeView &v, const int32_t *tb, int32_t ntiles, int variant, int *err, hipStream_t st, void *tabArena = nullptr, int64_t tabArenaCap = 0, void *copyTab = nullptr); // copyTab / tabArena: the copy blocks' tables (launch_copy_level), null: none
int64_t hash_chunks(int32_t cnt, int64_t arcs);
void launch_hash(int32_t from, int32_t cnt, int64_t arcs, const int64_t *rowptr, const int32_t *succ, uint32_t *A, uint32_t *B, int32_t *bounds, int32_t *hash, hipStream_t st);

void launch_bparse_big(const GraphDev &g, int def, const BatchView &v, int32_t coopMin, int32_t giantMin, int32_t *biglist, int32_t *giantlist, int32_t giantCap, int32_t *ctl,
                       void *arena, int64_t arenaCap, int waves, int giantGroups, int *err, hipStream_t st, hipStream_t stGiant, hipStream_t stBig, hipEvent_t evFork, hipEvent_t evGiant, hipEvent_t evBig);

// bv_consumers.hip: consumers of rows decoded into on-die scratch
void launch_stats(int32_t from, int32_t cnt, const int64_t *rowptr, const int32_t *succ, int64_t arcsUpper, void *statsDev, int32_t *indegree, int32_t n, hipStream_t st);
void launch_rows_differ(int32_t cnt, const int64_t *rpA, const int64_t *rpB, const int32_t *scA, const int32_t *scB, int *differ, hipStream_t st);
size_t stats_dev_bytes();
int64_t hyperball_big_cap(int64_t arcs);
void launch_hyperball(int32_t from, int32_t cnt, const int64_t *rowptr, const int32_t *succ, int32_t n, int32_t m, const uint8_t *regsIn, uint8_t *regsOut, const uint8_t *modIn, uint8_t *modOut,
                      unsigned long long *changed, int32_t *bigRows, int32_t bigCap, int32_t *bigCount, hipStream_t st);
void launch_bfs_expand(const int32_t *frontier, int32_t q, const int64_t *rowptr, const int32_t *succ, int64_t arcs, int32_t *marker, int32_t n, int32_t round, int parent,
                       int32_t *out, uint64_t outCap, unsigned long long *outCount, hipStream_t st);

// bv_offsets.hip: gamma-coded .offsets stream (words + >= 8 zero words in HBM) -> int64 offsets[nodes + 1] in HBM
int offsets_decode_device(const uint32_t *d_words, uint64_t nwords, uint64_t totalBits, int32_t nodes, int64_t *d_out, hipStream_t st, bool deltaCoded = false);
// arc labels (.labels stream in HBM, same padding): `count` consecutive labels starting at bit `startBit`
int gamma_labels_decode_device(const uint32_t *d_words, uint64_t nwords, uint64_t startBit, uint64_t endBit, int64_t count, int32_t *d_out, hipStream_t st);
int fixed_labels_decode_device(const uint32_t *d_words, uint64_t nwords, uint64_t startBit, int32_t width, int64_t count, int32_t *d_out, hipStream_t st);
// lists of fixed-width ints per arc (FixedWidthIntListLabel): nodes [from, from + cnt) through their label offsets -> listptr[arcs + 1], values
int label_lists_decode_device(const uint32_t *d_words, uint64_t nwords, const int64_t *d_off, int32_t from, int32_t cnt, int32_t width, uint64_t arcs,
                              int64_t *d_listptr, int32_t *d_values, uint64_t valuesCap, uint64_t *nvalues, hipStream_t st);

// EFGraph (bv_ef.hip): the .graph image as 64-bit words in host order (low bit first), decoded offsets, upper bound, log2 quantum
struct EfDev { const uint64_t *words; uint64_t nwords; const int64_t *offsets; int32_t n; uint64_t ub; int lq; };
// slot s <-> node nodes[s] (nodes != nullptr) or lo + s.  outd[cnt]
void launch_ef_outdeg(const EfDev &g, const int32_t *nodes, int32_t lo, int64_t cnt, int32_t *outd, int *err, hipStream_t st);
// rowstart[cnt + 1] = exclusive scan of outd; writes succ[rowstart[s] .. rowstart[s + 1]) for every slot whose list fits in `cap` (E_CAP otherwise).
// Lists shorter than bigMin: st; longer ones: stLong; giant ones: stGiant (the kernels touch different lists).  Lists of giantMin successors or more are cut
// into rounds of 64 words of upper bits, one work item each: chunks = chunkCap * ef_chunk_bytes() bytes of scratch, *nchunks zeroed
void launch_ef_decode(const EfDev &g, const int32_t *nodes, int32_t lo, int64_t cnt, int32_t bigMin, const int64_t *rowstart, int32_t *succ, uint64_t cap, int *err, int32_t giantMin,
                      void *chunks, uint32_t chunkCap, uint32_t *nchunks, hipStream_t st, hipStream_t stLong, hipStream_t stGiant);
size_t ef_chunk_bytes();
// ImmutableGraph.hashCode() of nodes lo .. lo + cnt - 1 without writing a successor: launch_ef_hash adds every successor, times its power of 31,
// to its slot's sum (acc = uint32[cnt] zeroed; three kernels on three streams), launch_ef_hash_fold -- after the streams are joined -- composes the
// nodes' maps in order and applies them to *h (device memory); maps = ef_hash_blocks(cnt) * 8 bytes of scratch
void launch_ef_hash(const EfDev &g, int32_t lo, int64_t cnt, int32_t bigMin, const int64_t *rowstart, uint32_t *acc, int *err, int32_t giantMin, void *chunks, uint32_t chunkCap,
                    uint32_t *nchunks, hipStream_t st, hipStream_t stLong, hipStream_t stGiant);
void launch_ef_hash_fold(int32_t lo, int64_t cnt, const int64_t *rowstart, const uint32_t *acc, void *maps, int32_t *h, hipStream_t st);
int64_t ef_hash_blocks(int64_t cnt);

// EFGraph.store on the device (bv_efw.hip): device CSR -> stream words (host order), record lengths, bit offsets; all hipMalloc'ed on success
int ef_encode_device(int32_t n, const int64_t *d_rowptr, const int32_t *d_succ, uint64_t ub, int lq, uint64_t **d_words_out, uint64_t *nwords_out, uint64_t *bits_out,
                     int32_t **d_reclen_out, int64_t **d_off_out, hipStream_t st);
// the .offsets stream of n record lengths (bv_encode.hip), coding = BVG_GAMMA / BVG_DELTA numbering; *d_words_out hipMalloc'ed
int offsets_stream_device(int coding, const int32_t *d_reclen, int32_t n, uint32_t **d_words_out, uint64_t *bits_out, hipStream_t st);

// BVGraph.store on the device (bv_encode.hip): device CSR -> .graph stream, bit offsets, .offsets stream, counters of the .properties file
struct EncodeOut {
	uint32_t *graph_words = nullptr; // big-endian words = the bytes of <basename>.graph, zero padded (+ >= 8 zero words)
	uint64_t graph_bits = 0;
	uint32_t *off_words = nullptr;   // <basename>.offsets
	uint64_t off_bits = 0;
	int64_t *offsets = nullptr;      // [n + 1] bit offset of every record
	uint64_t bits_outdegrees = 0, bits_references = 0, bits_blocks = 0, bits_intervals = 0, bits_residuals = 0;
	uint64_t copied_arcs = 0, intervalised_arcs = 0, residual_arcs = 0, tot_ref = 0, tot_dist = 0;
	int32_t max_ref_chain = 0, rounds = 0;
	uint64_t successor_gap_bins[32] = {}, residual_gap_bins[32] = {}; // successorGapStats / residualGapStats (BVGraph.java:1940-1944, :2196, :2303)
};
int encode_device(const bve::Params &p, int32_t n, const int64_t *d_rowptr, const int32_t *d_succ, uint64_t m, EncodeOut &out, std::string &err, hipStream_t st);
void encode_free(EncodeOut &o);

} // namespace bv
