// bvgpu_api.cpp -- the C ABI of libbvgpu.so (include/bvgpu.h): handle management, staging of the graph in HBM,
// orchestration of the decode pipeline of bv_kernels.hip.  Compiled with hipcc (host side only).
//
// No CPU fallback lives here: without a HIP device every decode entry point returns BVG_EHIP.
#include "../../include/bvgpu.h"
#include "bv_host.hpp"
#include "bv_launch.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

namespace {

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	// grows (never shrinks); contents are NOT preserved
	bool need(size_t bytes) {
		if (bytes <= cap) return true;
		if (p) (void)hipFree(p);
		p = nullptr; cap = 0;
		// BVGPU_EXACT_ALLOC=1 (tests): no slack, so that scripts/guard_alloc.cpp's unmapped page sits right behind what was asked for
		static const bool exact = [] { const char *e = bv_env("BVGPU_EXACT_ALLOC"); return e && atoi(e) != 0; }();
		size_t want = exact ? bytes : bytes + bytes / 8 + 256;
		if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return false; }
		cap = want;
		return true;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
	template <class T> T *as() const { return (T *)p; }
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	~DevBuf() { release(); } // (a buffer that bvg_close's list forgets is still freed with the handle -- ADVICE r4)
};

struct PinBuf { // pinned host memory (grows, never shrinks; contents are NOT preserved)
	void *p = nullptr;
	size_t cap = 0;
	bool need(size_t bytes) {
		if (bytes <= cap) return true;
		if (p) (void)hipHostFree(p);
		p = nullptr; cap = 0;
		const size_t want = bytes + bytes / 8 + 4096;
		if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return false; }
		cap = want;
		return true;
	}
	void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
	template <class T> T *as() const { return (T *)p; }
};

// Immutable, shared between a handle and its clones (BVGraph.copy() shares graphMemory / offsets, BVG:552-577).
// a vector whose resize() leaves the new elements alone: the host copy of the offsets is 8 (n + 1) bytes that are overwritten at once (zero-filling them was 10 ms of C2's load)
template <class T> struct NoInitAlloc : std::allocator<T> {
	template <class U> struct rebind { using other = NoInitAlloc<U>; };
	template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
	template <class U, class... A> void construct(U *p, A &&... a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
using HostOffsets = std::vector<int64_t, NoInitAlloc<int64_t>>;

struct Staged {
	int device = -1;
	bvg_info_t info{};
	uint32_t *d_bits = nullptr;     // word i of the .graph image is d_bits[i]; a shard stages words [word_lo, nwords) only: d_bits = allocation - word_lo
	uint32_t *d_bits_alloc = nullptr;
	uint64_t nwords = 0;
	int64_t *d_offsets = nullptr;   // likewise: d_offsets[x] for x in [stage_lo, node_hi]
	int64_t *d_offsets_alloc = nullptr;
	int32_t node_lo = 0, node_hi = 0, stage_lo = 0; // the nodes this handle decodes / the first node staged
	HostOffsets h_offsets; // host copy: shard bounds and halo sizing
	int64_t arcs_sizing = 0;        // max(arcs property, sum of the outdegrees in the stream): what scratch is sized by
	int64_t max_outdegree = -1; // the longest staged record (counted with arcs_sizing; -1: unknown)
	int64_t lane_rows = 0, lane_ids = 0; // staged rows with a reference and fewer than 128 successors, and their ids (the lane class of the copy pass)
	int64_t seg_long_records = -1, seg_long_bits = -1; // staged records with >= 2 048 bits of work (the parse list's long bins) and their bits (counted with arcs_sizing; -1: unknown): they size the segment pipeline
	int64_t oct_recs[bv::SIZING_OCTAVES] = {}, oct_arcs[bv::SIZING_OCTAVES] = {}; // staged records / their arcs with 2^(7 + k) <= outdegree < 2^(8 + k) (valid when max_outdegree >= 0)
	int32_t deg_counts[bv::PICK_LEVELS] = { -1, -1, -1, -1, -1, -1, -1 }; // staged records with >= 128, 256, ..., 8192 successors (counted with arcs_sizing; -1: unknown)
	int def = 0;                    // kernel variant: 1 default codings with zeta_3, 2 default codings with another zeta_k, 0 generic
	std::string basename;
	~Staged() {
		if (device >= 0) (void)hipSetDevice(device);
		if (d_bits_alloc) (void)hipFree(d_bits_alloc);
		if (d_offsets_alloc) (void)hipFree(d_offsets_alloc);
	}
};

constexpr int32_t COOP_BUDGET = 12288; // records a job sends to the wave class at most (4 x the waves in flight): see k_pick_coop

struct Small { // device <-> host mailbox
	int err;
	int32_t maxdepth;
	int32_t hash;
	int32_t pad;
	int64_t total;     // rowstart[cnt] - rowstart[nh]
	int64_t halo_total;
	int32_t coop_used; // the wave-class threshold the job ran with
	int32_t pad2;
};

struct Pending { // an enqueued range decode whose status has not been collected yet
	bool active = false;
	bv::RangeView view{};
	int32_t levels_done = 0;
	bool want_succ = false;
	int32_t giantCap = 0, bigCap = 0, midCap = 0, walkMin = 0x7fffffff;
	uint32_t tmpCap = 0;
	const void *preDesc = nullptr; // k_copy_prewalk's descriptors (null: not run)
	const void *copyTab = nullptr; // 16 bytes per slot: the copy blocks of the rows that the one-lane parse decoded, as tables (null: the copy pass walks the stream)
	const void *tabArena = nullptr; // the interval arena (their blocks from the fourth on)
	int64_t tabArenaCap = 0;
	// a sub-range decoded before its halo was sized (optimistic): what to repeat if the guess was wrong
	bool optimistic = false;
	int32_t from = 0, to = 0;
	int64_t *rowptr = nullptr;
	int32_t *succ = nullptr;
	size_t succ_cap = 0;
};

} // namespace

struct bvg_graph {
	std::shared_ptr<Staged> st;
	hipStream_t own = nullptr, stream = nullptr; // `stream` is always `own`: the library executes on its own streams
	hipStream_t user = nullptr;                  // caller's stream (bvg_set_stream): work is ordered after it and it waits for our results
	hipEvent_t evIn = nullptr, evOut = nullptr;
	mutable std::string err;
	DevBuf outd, ref, rowstart, depth, sums, need, halo, hashA, hashB, hashBounds, stage_rowptr, stage_succ, stage_nodes, small;
	DevBuf b_chainlen, b_slotbase, b_node, b_qidx, b_aoutd, b_qoutd; // random-access batches
	DevBuf walktab;                                                   // block tables of the giant records (GraphDev::walktab)
	int walk_tables = 1;                                              // BVGPU_WALK_TABLES=0: the copy pass walks every block list itself
	DevBuf pickpart;                                                  // per-block outdegree class counts of k_headers
	DevBuf biglist, giantlist, arena, coopctl;                        // work lists; cooperative decode of giant records
	DevBuf key16, keys;                                               // per-slot list key; hist / keyBase / cursor
	int level_blocks = 16384; // blocks of the list kernels (k_parse_list, k_copy_list), at most one thread per node of the range: 8192 .. 32768 are within 1 % on C2, 2 % faster than 4096 on the C5 shard and cnr-2000 x 30 (profiles/r4_experiments.txt section 9)
	DevBuf lvlist;
	DevBuf plist, pkeys, pkey16;
	DevBuf segbuf;       // scratch of the segment pipeline (bv_seg.hip)
	int seg = 1;         // BVGPU_SEG: the segment pipeline (bv_seg.hip).  0: never; 1 (default): for the hubs of a job -- the giant records with >= seg_hub_min successors hand their
	                     // residual sections over, everything else as before; 3: that, whatever the graph holds.  (Mode 2 of round 4 -- everything it can take -- is tag r4-experiments.)
	int seg_hub_min = 1000000; // BVGPU_SEG_HUB_MIN
	int seg_blocks = 2048;
	int lists_on_b = 0;  // BVGPU_LISTS_ON_B=1: the chain depths / level lists behind the giants (side B) instead of behind the wave class (side A); 0: side B when the graph has no giants; 3: side A
	int level_lists_early = 1; // BVGPU_LEVEL_LISTS_EARLY=0: a job that decodes its lane class from tiles builds its level lists behind the wave class, like the others
	int skip_empty_giants = 1; // BVGPU_SKIP_EMPTY_GIANTS=0: the giants' kernel is launched even when the graph holds no record that long
	int tile = -1;       // -1: automatic (see enqueue_decode); BVGPU_TILE=0: never; BVGPU_TILE=1: short records decoded from contiguous tiles of the stream (k_parse_tile) instead of the bin-sorted parse list (k_parse_list)
	DevBuf tilebounds;
	int copy_big = 1;    // BVGPU_COPY_BIG=0: every row is copied by one lane
	int parse_windows = 1; // BVGPU_PARSE_WINDOWS=0: the parse list is sorted by work bin over the whole range
	int copy_mid_min = 128; // rows with at least this many successors (and fewer than 1024) are copied by one wave each
	int32_t coop_min = 2048, giant_min = 32768;                         // thresholds on the outdegree (BVGPU_COOP_MIN / BVGPU_GIANT_MIN)
	long long scan_top_tiled_min = -1, scan_piece = 0; // (-1 / 0: the defaults of bv_kernels.hip / scan_piece_arcs)
	bool wait_giants = true, hash_materialise = false, ef_hash_materialise = false, want_stats = false, trace_retry = false, trace_err = false, trace_host = false;
	int dbg = 0;
	bool adaptive = true, adaptive_giant = true;                        // smaller jobs lower them (pick_thresholds) unless a knob pins them (each knob pins its own threshold)
	int coop_waves = 4096, giant_groups = 256;
	DevBuf copyq; // rows the copy pass merges with a group / a wave each (all levels), filled while the level lists are built
	DevBuf walkdesc; // k_copy_prewalk: 16 bytes per entry of the group class's queue
	DevBuf copytab;  // 16 bytes per slot: the copy blocks of the rows that the one-lane parse decoded, for the lane class of the copy pass (CopyTab, bv_lanewin.hpp)
	int copy_vec = -1; // BVGPU_COPY_VEC=1|0: the lane class of the copy pass merges with 16-byte loads and stores (copy_node_v) or id by id; -1: by the mean length of its rows (counted at load time)
	int giants_after_list = 1; // BVGPU_GIANTS_AFTER_LIST=0: the giants' kernel does not wait for the parse list
	int keys_in_headers = 1; // BVGPU_KEYS_IN_HEADERS=0: the parse list's keys by k_depth_keys, not by k_headers
	bool keys_ready = false; // (per job) k_headers wrote them
	int level_bins = 1;  // BVGPU_LEVEL_BINS=0: the level lists in node order (round 5), not sorted by the records' work bins inside a level: the wave loop of k_copy_list_w runs as long as its longest row
	int waves_on_b = 1;  // BVGPU_WAVES_ON_B=0: the wave class of a tile job without giants behind the pre-walks on side A (round 5), not on side B
	int pick_aside = 1;  // BVGPU_PICK_ASIDE=0: k_pick_coop in front of the scan of the outdegrees, not beside it
	int list_refs = 1;   // BVGPU_LIST_REFS=0: plain slot numbers in the parse list (k_parse_list looks the reference up)
	int tile_loop = 1;   // BVGPU_TILE_LOOP=0: the tile kernel decodes with its own reader (parse_node_tile), not with the wave's loop of the lane kernel (parse_node_lwc)
	int stream_prio = 1; // BVGPU_STREAM_PRIO (environment only: read when the handle's streams are created): bit 0 = side B (the giants of the parse phase, the wave class of the copy pass) above the other streams, bit 1 = side A below them; 0: all alike
	int mid_tables = 1;  // BVGPU_MID_TABLES=0: k_copy_mid walks the block lists that the pre-walk left (192 codes and more) although the one-lane parse left them as tables
	int copy_loop = 1;   // BVGPU_COPY_LOOP=0: the lane class of the copy pass merges lane by lane (copy_node_tab), not as a loop of the wave (k_copy_list_w)
	int lane_loop = 1;   // BVGPU_LANE_LOOP=0: round 4's one-lane loop (parse_node_lwb) instead of round 6's (parse_node_lwc)
	int copy_tables = 1; // BVGPU_COPY_TABLES=0: the lane class of the copy pass walks the block lists in the stream although the parse left them as tables
	int prewalk_long = 1; // BVGPU_PREWALK_LONG=0: no kernel of their own for the lists of >= 2048 codes; 2: on the lists' stream instead of side B
	int prewalk_blocks = 1024;
	int prewalk = 1; // BVGPU_PREWALK=0: k_copy_big walks its rows' block lists itself
	DevBuf bigtmp; // global scratch tables for rows that copy more ids than the LDS tables of k_copy_big hold
	DevBuf stats; // BVGPU_STATS=1: tuning counters
	// the three parse kernels (giant / big / short records) are independent: they run on forked streams
	hipStream_t sideC = nullptr; // BVGPU_LISTS_ON_B=2: the level lists on a stream of their own
	hipStream_t sideA = nullptr, sideB = nullptr; // (more streams than this share hardware queues with each other: they would serialise)
	bool ctl_clean = false;                       // ctl[4..16) were zeroed by this job's k_pick_coop
	bool host_mode = false;                       // host_scan: sideB carries the PCIe copies, its kernels go to sideA
	hipEvent_t evFork = nullptr, evA = nullptr, evB = nullptr, evC = nullptr, evHdr = nullptr, evHdr0 = nullptr, evP = nullptr, evM = nullptr, evH = nullptr, evL = nullptr;
	bool overlap = true;
	size_t halo_min = (size_t)16 << 20; // bytes of halo scratch an optimistic sub-range decode starts with (BVGPU_HALO_MIN)
	bool force_halo_sync = false;   // (retry of an optimistic sub-range decode: size the halo with a host round trip)
	int64_t *early_rowptr = nullptr; // decode_range_device: caller's rowptr, written on a side stream as soon as the scan is done (set per call)
	int batch_dense = 32;   // a random-access batch of q nodes with q * batch_dense >= n is decoded as a masked scan of the graph (0: never;
	                        // measured crossover on the 10 M-node C2 graph: q = 300 000)
	// host-output scans (BVG_OUT_HOST, bvg_decode_range_view): chunks are decoded into two device buffers in turn and
	// leave over PCIe on a copy stream of their own while the next chunk is being decoded
	DevBuf statsbuf, bfs_rowptr, bfs_succ, bfs_ctr; // consumers (bv_consumers.hip)
	// bvg_scan_checksum without the 4 B / edge round trip (bv::HashCtx): hash_job is set around its range decodes; hashctx = { HashCtx, acc, two counters }
	DevBuf hashmark, hashctx, hashtab, hashq;
	bool hash_job = false, hash_in_parse = false, hash_stale = false; // hash_stale: the copy pass needed more levels than were launched before the rows were folded
	bv::RangeView hash_view{};
	DevBuf hchunk[2];
	PinBuf hring[2];               // pageable destinations: the chunk lands here first and is copied out by host threads
	PinBuf view_rowptr, view_succ; // bvg_decode_range_view: library-owned pinned results
	hipStream_t copyStream = nullptr;
	hipEvent_t evChunk[2] = {}, evCopied[2] = {};
	Small *h_small = nullptr; // pinned
	int32_t levels_hint = 1;
	Pending pend;
	uint64_t last_arcs = 0;
	int32_t last_coop_min = 0, last_giant_min = 0; // thresholds of the last range decode (bvg_last_thresholds)
	// optional per-phase timing (bvg_set_profile): events recorded between the phases of a range decode
	bool profile = false;
	hipEvent_t ev[BVG_NUM_PHASES + 1] = {};
	bool ev_valid = false;
};

namespace {

bv::GraphDev graph_dev0(const Staged &s);
bool copy_vec(const bvg_graph *g);
inline hipStream_t side_b(const bvg_graph *g);
bv::GraphDev graph_dev_h(const bvg_graph *g, const Staged &s) { bv::GraphDev d = graph_dev0(s); d.stats = (unsigned long long *)g->stats.p; d.dbg = g->dbg; return d; }

int fail(const bvg_graph *g, int code, const std::string &msg) { if (g) g->err = msg; return code; }

#define HIPCHK(g, call)                                                                                     \
	do {                                                                                                    \
		hipError_t e_ = (call);                                                                             \
		if (e_ != hipSuccess) return fail(g, e_ == hipErrorOutOfMemory ? BVG_ENOMEM : BVG_EHIP, std::string(#call ": ") + hipGetErrorString(e_)); \
	} while (0)

bv::GraphDev graph_dev0(const Staged &s) {
	bv::GraphDev g{};
	g.bits = s.d_bits; g.nwords = s.nwords; g.offsets = s.d_offsets; g.n = s.info.nodes;
	g.W = s.info.window_size; g.minInt = s.info.min_interval_length; g.zetaK = s.info.zeta_k;
	g.c_outd = s.info.outdegree_coding; g.c_ref = s.info.reference_coding; g.c_bc = s.info.block_count_coding;
	g.c_blk = s.info.block_coding; g.c_res = s.info.residual_coding;
	g.stats = nullptr;
	g.dbg = 0;
	return g;
}

#define graph_dev(S) graph_dev_h(g, S)

int dev_err_to_status(int e) {
	if (e & bv::E_REF) return BVG_ESTATE;
	if (e & bv::E_UNSUP) return BVG_EUNSUPPORTED;
	if (e & bv::E_FORMAT) return BVG_EFORMAT;
	if (e & bv::E_CAP) return BVG_ECAP;
	if (e & bv::E_ESCAPED) return BVG_EFORMAT;
	return BVG_OK;
}

// ---- tuning and debug knobs of a handle.  One table: bvg_set_option(name, value) sets a knob of a live handle, and init_handle applies the environment's
// BVGPU_<NAME> through the same function, once per handle -- nothing reads the environment per launch.  -DBVGPU_NO_ENV compiles the environment out (bv_env):
// a release build has the defaults and bvg_set_option only.  Every knob is a choice of speed, never of results (tests/test_gpu_scan.py::test_tuning_knobs_keep_parity).
int apply_option(bvg_graph *g, const std::string &name, const char *value) {
	const long long v = value ? atoll(value) : 0;
	const int iv = (int)std::max<long long>(std::min<long long>(v, 0x7fffffff), -0x7fffffff);
	if (name == "coop_min") { g->coop_min = std::max(1, iv); g->adaptive = false; }   // 0x7fffffff disables the cooperative path
	else if (name == "giant_min") { g->giant_min = std::max(1, iv); g->adaptive_giant = false; }
	else if (name == "adaptive") { g->adaptive = g->adaptive_giant = iv != 0; }        // 1: both thresholds chosen per job again
	else if (name == "coop_waves") g->coop_waves = std::max(1, iv);
	else if (name == "giant_groups") g->giant_groups = std::max(1, iv);
	else if (name == "level_blocks") g->level_blocks = std::max(1, iv);
	else if (name == "copy_big") g->copy_big = iv;
	else if (name == "parse_windows") g->parse_windows = iv;
	else if (name == "tile") g->tile = iv;
	else if (name == "seg") g->seg = iv;
	else if (name == "seg_hub_min") g->seg_hub_min = std::max(1, iv);
	else if (name == "seg_blocks") g->seg_blocks = std::max(1, iv);
	else if (name == "lists_on_b") g->lists_on_b = iv;
	else if (name == "skip_empty_giants") g->skip_empty_giants = iv;
	else if (name == "level_lists_early") g->level_lists_early = iv;
	else if (name == "walk_tables") g->walk_tables = iv;
	else if (name == "copy_vec") g->copy_vec = iv;
	else if (name == "lane_loop") g->lane_loop = iv;
	else if (name == "copy_loop") g->copy_loop = iv;
	else if (name == "mid_tables") g->mid_tables = iv;
	else if (name == "stream_prio") g->stream_prio = iv;
	else if (name == "tile_loop") g->tile_loop = iv;
	else if (name == "list_refs") g->list_refs = iv;
	else if (name == "pick_aside") g->pick_aside = iv;
	else if (name == "waves_on_b") g->waves_on_b = iv;
	else if (name == "level_bins") g->level_bins = iv;
	else if (name == "keys_in_headers") g->keys_in_headers = iv;
	else if (name == "giants_after_list") g->giants_after_list = iv;
	else if (name == "copy_tables") g->copy_tables = iv;
	else if (name == "prewalk") g->prewalk = iv;
	else if (name == "prewalk_long") g->prewalk_long = iv;
	else if (name == "prewalk_blocks") g->prewalk_blocks = std::max(1, iv);
	else if (name == "copy_mid_min") g->copy_mid_min = std::min(std::max(0, iv), 1024); // 0: no wave-per-row copy
	else if (name == "overlap") g->overlap = iv != 0;
	else if (name == "halo_min") g->halo_min = (size_t)std::max(4, iv);
	else if (name == "batch_dense") g->batch_dense = std::max(0, iv);
	else if (name == "scan_top_tiled_min") g->scan_top_tiled_min = v > 0 ? v : -1;
	else if (name == "wait_giants") g->wait_giants = iv != 0;
	else if (name == "hash_materialise") g->hash_materialise = iv != 0;
	else if (name == "ef_hash_materialise") g->ef_hash_materialise = iv != 0;
	else if (name == "scan_piece") g->scan_piece = v > 0 ? v : 0;
	else if (name == "dbg") g->dbg = iv;
	else if (name == "stats") g->want_stats = iv != 0; // (the counters are allocated by init_handle: environment only)
	else if (name == "trace_retry") g->trace_retry = iv != 0;
	else if (name == "trace_err") g->trace_err = iv != 0;
	else if (name == "trace_host") g->trace_host = iv != 0;
	else return BVG_EARG;
	return BVG_OK;
}
const char *const OPTION_NAMES[] = { "coop_min", "giant_min", "coop_waves", "giant_groups", "level_blocks", "copy_big", "parse_windows", "tile", "seg", "seg_hub_min", "seg_blocks", "lists_on_b", "skip_empty_giants", "level_lists_early",
	"walk_tables", "copy_vec", "lane_loop", "copy_loop", "mid_tables", "stream_prio", "tile_loop", "list_refs", "pick_aside", "waves_on_b", "level_bins", "copy_tables", "giants_after_list", "keys_in_headers", "prewalk", "prewalk_long", "prewalk_blocks", "copy_mid_min", "overlap", "halo_min", "batch_dense", "scan_top_tiled_min", "wait_giants", "hash_materialise",
	"ef_hash_materialise", "scan_piece", "dbg", "stats", "trace_retry", "trace_err", "trace_host" };
void options_from_env(bvg_graph *g) {
	for (const char *n : OPTION_NAMES) {
		std::string e = "BVGPU_";
		for (const char *c = n; *c; c++) e += (char)toupper((unsigned char)*c);
		if (const char *val = bv_env(e.c_str())) (void)apply_option(g, n, val);
	}
}

int init_handle(bvg_graph *g) {
	HIPCHK(g, hipSetDevice(g->st->device));
	HIPCHK(g, hipStreamCreateWithFlags(&g->own, hipStreamNonBlocking));
	g->stream = g->own;
	HIPCHK(g, hipHostMalloc((void **)&g->h_small, sizeof(Small), hipHostMallocDefault));
	if (!g->small.need(sizeof(Small))) return fail(g, BVG_ENOMEM, "device allocation failed");
	const int mr = g->st->info.max_ref_count;
	g->levels_hint = mr < 1 ? 1 : (mr > 8 ? 8 : mr);
	if (!g->coopctl.need(bv::CTL_TOTAL_INTS * sizeof(int32_t)) || !g->keys.need(3 * (bv::NKEYS + 1) * sizeof(int32_t))) return fail(g, BVG_ENOMEM, "device allocation failed");
	HIPCHK(g, hipMemset(g->coopctl.p, 0, bv::CTL_TOTAL_INTS * sizeof(int32_t))); // (k_pick_coop leaves its counters zeroed for the next job)
	options_from_env(g); // every tuning / debug knob of a handle: read here ONCE (and never per launch), or set later through bvg_set_option
	if (g->stream_prio) { // side B above the others: what it carries are few, long-lived blocks (the giants' groups; k_copy_mid's waves) that the many short ones of the list kernels would otherwise keep out of the CUs -- C5 shard 4.76 -> 4.65 ms, C2 and cnr-2000 x 30 even (profiles/r6_experiments.txt section 24)
		int least = 0, greatest = 0;
		HIPCHK(g, hipDeviceGetStreamPriorityRange(&least, &greatest));
		HIPCHK(g, hipStreamCreateWithPriority(&g->sideA, hipStreamNonBlocking, (g->stream_prio & 2) ? least : (least + greatest) / 2));
		HIPCHK(g, hipStreamCreateWithPriority(&g->sideB, hipStreamNonBlocking, (g->stream_prio & 1) ? greatest : (least + greatest) / 2));
	} else {
		HIPCHK(g, hipStreamCreateWithFlags(&g->sideA, hipStreamNonBlocking));
		HIPCHK(g, hipStreamCreateWithFlags(&g->sideB, hipStreamNonBlocking));
	}
	HIPCHK(g, hipStreamCreateWithFlags(&g->sideC, hipStreamNonBlocking));
	g->copyStream = g->sideB; // a fourth stream would share a hardware queue with one of the other three (GPU_MAX_HW_QUEUES = 4, one is the null stream's):
	                          // its copies then hold back the kernels queued behind them -- measured: every chunk of a host scan took decode + copy, 25 ms instead of 17
	for (int i = 0; i < 2; i++) { HIPCHK(g, hipEventCreateWithFlags(&g->evChunk[i], hipEventDisableTiming)); HIPCHK(g, hipEventCreateWithFlags(&g->evCopied[i], hipEventDisableTiming)); }
	HIPCHK(g, hipEventCreateWithFlags(&g->evFork, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evIn, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evOut, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evA, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evB, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evC, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evHdr, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evHdr0, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evP, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evM, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evH, hipEventDisableTiming));
	HIPCHK(g, hipEventCreateWithFlags(&g->evL, hipEventDisableTiming));
	if (g->want_stats) { if (!g->stats.need(64 * sizeof(unsigned long long))) return fail(g, BVG_ENOMEM, "device allocation failed"); HIPCHK(g, hipMemset(g->stats.p, 0, 512)); }
	return BVG_OK;
}

void mark(bvg_graph *g, int i) { if (g->profile) (void)hipEventRecord(g->ev[i], g->stream); }

// ---- bvg_scan_checksum's fold (bv::HashCtx): the device-side context of a hash job; the sum and the two counters of k_hash_rest sit behind it
constexpr size_t HASH_ACC_OFF = 64, HASH_SLOTS_OFF = 128; // bytes: (unused word at +64,) counters of the piece queues at +68, +72; the parts of the sum from +128
uint32_t host_pow31(uint64_t e) { uint32_t r = 1, b = 31; while (e) { if (e & 1) r *= b; b *= b; e >>= 1; } return r; }
int hash_prepare(bvg_graph *g) {
	if (!g->hashtab.p) {
		if (!g->hashtab.need(3 * 1024 * sizeof(uint32_t))) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
		std::vector<uint32_t> t(3 * 1024);
		uint32_t u = 31; // u = 31^-1 mod 2^32 (Newton: every step doubles the correct low bits)
		for (int k = 0; k < 6; k++) u *= 2u - 31u * u;
		for (int k = 0; k < 3; k++) {
			uint32_t step = u; // u^(2^(10 k))
			for (int i = 0; i < 10 * k; i++) step *= step;
			uint32_t w = 1;
			for (int i = 0; i < 1024; i++) { t[(size_t)k * 1024 + i] = w; w *= step; }
		}
		HIPCHK(g, hipMemcpy(g->hashtab.p, t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	static_assert(sizeof(bv::HashCtx) <= HASH_ACC_OFF, "HashCtx");
	if (!g->hashctx.need(HASH_SLOTS_OFF + sizeof(uint32_t) * bv::HASH_ACC_SLOTS)) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	return BVG_OK; // (the context itself is written by hash_publish, from enqueue_structure, which sizes the mark buffer)
}
int hash_publish(bvg_graph *g) { // after enqueue_structure: every pointer of the context is final
	struct Init { bv::HashCtx c; char pad[HASH_SLOTS_OFF - sizeof(bv::HashCtx)]; uint32_t acc[bv::HASH_ACC_SLOTS]; } init{};
	init.c.mark = g->hashmark.as<uint8_t>();
	init.c.acc = (uint32_t *)((char *)g->hashctx.p + HASH_SLOTS_OFF);
	init.c.ptab = g->hashtab.as<uint32_t>();
	HIPCHK(g, hipMemcpyAsync(g->hashctx.p, &init, sizeof(init), hipMemcpyHostToDevice, g->stream)); // (pageable source: the copy is staged before the call returns)
	return BVG_OK;
}

// Does a job over slots [lo, lo + cnt) decode its lane class from tiles (bv_tile.hpp) rather than from the parse list?  (Decided from what the host knows when the job is
// enqueued: enqueue_structure asks -- a tile job needs no parse-list keys from k_headers -- and enqueue_decode acts on it.)
int job_tile_variant(const bvg_graph *g, int32_t lo, int32_t cnt, bool pick) {
	const Staged &s = *g->st;
	int tileVariant = g->tile > 0 ? g->tile : 0;
	if (g->hash_job) tileVariant = 0; // (the hash fold rides on k_parse_list)
	else if (g->tile < 0 && g->adaptive && pick && s.deg_counts[0] >= 0) {
		const double share = (double)(s.h_offsets[(size_t)lo + cnt] - s.h_offsets[(size_t)lo]) / (double)std::max<int64_t>(s.h_offsets[(size_t)s.node_hi] - s.h_offsets[(size_t)s.stage_lo], 1);
		if ((double)s.deg_counts[0] * share <= (double)COOP_BUDGET * (share > 0.999 ? 1.0 : 0.8)) tileVariant = 1; // (a sub-range: an estimate, with a margin)
	}
	return s.def != 0 ? tileVariant : 0;
}

// Enqueues headers (+halo closure) + scan for nodes [from,to) with a halo of nh nodes before `from`.
// On return the view describes the job; rowstart lives in scratch.
int enqueue_structure(bvg_graph *g, int32_t from, int32_t to, int32_t nh, bv::RangeView &v, bool pickCoop = false, int64_t *rowstart_out = nullptr) {
	const Staged &s = *g->st;
	const int32_t lo = from - nh, cnt = to - lo;
	if (!g->outd.need(sizeof(int32_t) * (size_t)cnt) || !g->ref.need(sizeof(uint16_t) * (size_t)cnt) ||
	    !g->rowstart.need(sizeof(int64_t) * ((size_t)cnt + 1)) || !g->sums.need(sizeof(int64_t) * (size_t)bv::scan_num_sums(cnt)) ||
	    (nh && !g->need.need((size_t)nh)))
		return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	v = bv::RangeView{};
	v.halo_cap = ~0ull;
	v.lo = lo; v.cnt = cnt; v.nh = nh;
	v.outd = g->outd.as<int32_t>(); v.ref = g->ref.as<uint16_t>(); v.rowstart = g->rowstart.as<int64_t>();
	if (rowstart_out && nh == 0) v.rowstart = rowstart_out; // without a halo the row starts ARE the caller's rowptr: scanned in place (no k_rebase pass: 160 MB less per C2 scan)
	int *derr = &g->small.as<Small>()->err;
	const bv::GraphDev gd = graph_dev(s);
	mark(g, 0);
	const bool pick = pickCoop && g->adaptive; // the wave-class threshold of this job comes from its outdegrees (k_pick_coop)
	const int64_t hb = bv::headers_blocks(cnt);
	if (pick && !g->pickpart.need(sizeof(int32_t) * bv::PICK_LEVELS * (size_t)hb)) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	if (g->hash_job) { // bvg_scan_checksum: the context of the fold, written in front of the first kernel that adds to it (the scan: node numbers)
		if (!g->hashmark.need((size_t)cnt)) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
		const int rc = hash_publish(g);
		if (rc) return rc;
	}
	// the parse list's keys fall out of the headers (a scan job without a halo that will build a parse list: not a tile job; keys_in_headers = 0: k_depth_keys computes them as before)
	g->keys_ready = false;
	uint16_t *pk16 = nullptr;
	if (g->keys_in_headers && pickCoop && nh == 0 && s.def != 0 && g->overlap && !g->profile && job_tile_variant(g, lo, cnt, pick) == 0 &&
	    g->pkey16.need(sizeof(uint16_t) * (size_t)cnt) && g->pkeys.need(3 * (bv::NKEYS + 1) * sizeof(int32_t))) { pk16 = g->pkey16.as<uint16_t>(); g->keys_ready = true; }
	bv::launch_headers(gd, s.def, lo, cnt, v.outd, v.ref, derr, g->stream, pick ? g->pickpart.as<int32_t>() : nullptr, g->hash_job ? g->hashmark.as<uint8_t>() : nullptr,
	                   pk16, pk16 ? g->pkeys.as<int32_t>() : nullptr, g->parse_windows != 0);
	if (nh) bv::launch_mark_halo(nh, cnt, s.info.window_size, v.outd, v.ref, g->need.as<uint8_t>(), derr, g->stream);
	// k_pick_coop is ONE block that adds up 7 counts per block of k_headers (21 us on C2, 124 us at 50 M nodes) and the scan of the outdegrees does not need it: on side B,
	// beside the scan (pick_aside = 0: in front of it, as until round 6).  evHdr then says "headers final, threshold picked, counters clean"; the job's stream joins it behind the scan.
	const bool pickAside = pick && g->pick_aside && g->overlap && !g->profile;
	if (pick) { // (also zeroes ctl[4..16), the counters of the lists and of the copy levels: a memset behind the scan kernels sat 22 us on the critical path)
		hipStream_t stPick = g->stream;
		if (pickAside) {
			stPick = side_b(g);
			HIPCHK(g, hipEventRecord(g->evHdr0, g->stream));
			HIPCHK(g, hipStreamWaitEvent(stPick, g->evHdr0, 0));
		}
		bv::launch_pick_coop(g->pickpart.as<int32_t>(), (int32_t)hb, COOP_BUDGET, g->coopctl.as<int32_t>(), stPick);
		v.coop_ptr = g->coopctl.as<int32_t>() + bv::CTL_COOP;
		g->ctl_clean = true;
		if (pickAside) HIPCHK(g, hipEventRecord(g->evHdr, stPick));
	}
	if (!pickAside) HIPCHK(g, hipEventRecord(g->evHdr, g->stream)); // outdegrees and references are final: the parse list can be built while the scan runs
	mark(g, 1);
	bv::launch_scan(v.outd, cnt, v.rowstart, g->sums.as<int64_t>(), g->stream, g->hash_job ? g->hashctx.as<bv::HashCtx>() : nullptr, lo, nh, g->scan_top_tiled_min);
	if (pickAside) HIPCHK(g, hipStreamWaitEvent(g->stream, g->evHdr, 0));
	mark(g, 2);
	return BVG_OK;
}

// The caller's stream (if any) only orders the work: kernels run on the library's own streams, which overlap
// better than a stream taken from the caller (measured with PyTorch's: 9.3 ms vs 7.7 ms per C2 scan).
int fork_from_user(bvg_graph *g) {
	if (!g->user) return BVG_OK;
	HIPCHK(g, hipEventRecord(g->evIn, g->user));
	HIPCHK(g, hipStreamWaitEvent(g->stream, g->evIn, 0));
	return BVG_OK;
}
int join_to_user(bvg_graph *g) {
	if (!g->user) return BVG_OK;
	HIPCHK(g, hipEventRecord(g->evOut, g->stream));
	HIPCHK(g, hipStreamWaitEvent(g->user, g->evOut, 0));
	return BVG_OK;
}

inline hipStream_t side_b(const bvg_graph *g) { return g->host_mode ? g->sideA : g->sideB; }

// GraphDev::walktab for this handle's job (walkMin = 0x7fffffff: none); the bump pointer is ctl[7], zeroed with the job's other counters
void set_walk(bv::GraphDev &gd, bvg_graph *g, int32_t walkMin) {
	gd.walkMin = walkMin;
	gd.walktab = walkMin < 0x7fffffff ? g->walktab.as<int32_t>() : nullptr;
	gd.walkCap = walkMin < 0x7fffffff ? (uint32_t)std::min<size_t>(g->walktab.cap / sizeof(int32_t), 0x7fffffff) : 0;
	gd.walkCursor = (uint32_t *)(g->coopctl.as<int32_t>() + 7);
}

int fetch_small(bvg_graph *g) {
	HIPCHK(g, hipMemcpyAsync(g->h_small, g->small.p, sizeof(Small), hipMemcpyDeviceToHost, g->stream));
	HIPCHK(g, hipStreamSynchronize(g->stream));
	return BVG_OK;
}

__global__ void k_totals(const int64_t *rowstart, int32_t nh, int32_t cnt, Small *sm, const int32_t *coopPtr = nullptr, int32_t coopMin = 0) {
	sm->total = rowstart[cnt] - rowstart[nh];
	sm->halo_total = rowstart[nh];
	sm->coop_used = coopPtr ? *coopPtr : coopMin;
}

int decode_range_device(bvg_graph *g, int32_t from, int32_t to, int64_t *rowptr_dev, int32_t *succ_dev, size_t succ_cap, bool async, uint64_t *arcs_out);

// Collects the status of the pending job: runs the reference-chain levels that the optimistic launch did
// not cover, then reports errors / arc count.
int finish_pending(bvg_graph *g, uint64_t *arcs_out) {
	if (!g->pend.active) { if (arcs_out) *arcs_out = g->last_arcs; return BVG_OK; }
	const Staged &s = *g->st;
	int rc = fetch_small(g);
	if (rc) { g->pend = Pending{}; return rc; }
	if (g->pend.optimistic && (g->h_small->err & (bv::E_ESCAPED | bv::E_HALO))) {
		if (g->trace_retry) fprintf(stderr, "[bvgpu] optimistic halo missed (err %d): repeating [%d, %d)\n", g->h_small->err, g->pend.from, g->pend.to);
		// the halo of this sub-range was deeper or larger than guessed: once more, sized with a host round trip
		const Pending p = g->pend;
		g->pend = Pending{};
		g->force_halo_sync = true;
		rc = decode_range_device(g, p.from, p.to, p.rowptr, p.succ, p.succ_cap, false, arcs_out);
		g->force_halo_sync = false;
		return rc;
	}
	if (g->pend.want_succ && !g->h_small->err) {
		bv::GraphDev gd = graph_dev(s);
		set_walk(gd, g, g->pend.walkMin);
		int *derr = &g->small.as<Small>()->err;
		while (g->pend.levels_done < g->h_small->maxdepth) {
			g->hash_stale = g->hash_job; // (rows were folded before these levels ran: bvg_scan_checksum repeats the piece the plain way)
			const int32_t upto = g->h_small->maxdepth;
			int32_t *keyBase = g->keys.as<int32_t>() + (bv::NKEYS + 1);
			for (int32_t l = g->pend.levels_done + 1; l <= upto; l++) {
				{
					const bool ov2 = g->overlap && !g->profile;
					bv::launch_copy_level(gd, s.def, g->pend.view, g->depth.as<int32_t>(), g->lvlist.as<int32_t>(), keyBase, l, g->level_blocks, g->copy_mid_min, g->copy_big != 0,
					                      g->copyq.as<int32_t>(), g->pend.bigCap, g->copyq.as<int32_t>() + g->pend.bigCap, g->pend.midCap, g->coopctl.as<int32_t>(), g->bigtmp.as<int32_t>(), g->pend.tmpCap, derr,
					                      g->stream, ov2 ? side_b(g) : g->stream, ov2 ? g->sideA : g->stream, g->evFork, g->evB, g->evA, g->pend.preDesc, g->prewalk < 2 && g->pend.midCap > 0, (copy_vec(g) ? 1 : 0) | (g->copy_loop ? 2 : 0) | (g->mid_tables ? 16 : 0),
					                      g->pend.tabArena, g->pend.tabArenaCap, g->pend.copyTab);
				}
			}
			g->pend.levels_done = upto;
			rc = fetch_small(g);
			if (rc) { g->pend = Pending{}; return rc; }
		}
		g->levels_hint = std::max(1, std::min<int32_t>(g->h_small->maxdepth, 64));
	}
	g->pend = Pending{}; // (a later job never sees this one's buffers or its "optimistic" flag)
	g->last_arcs = (uint64_t)g->h_small->total;
	g->last_coop_min = g->h_small->coop_used;
	if (arcs_out) *arcs_out = g->last_arcs;
	if (g->h_small->err) {
		const int st = dev_err_to_status(g->h_small->err);
		if (g->trace_err) fprintf(stderr, "[bvgpu] range job: device error bits 0x%x\n", g->h_small->err);
		return fail(g, st, st == BVG_ECAP ? "successor buffer too small" : st == BVG_ESTATE ? "reference incompatible with the window size" : "malformed or unsupported bit stream");
	}
	return BVG_OK;
}

// Enqueues the decode proper for a view whose structure (outdegrees, references, row starts) is in place: parse of
// every record, then `levels_hint` levels of the copy pass (finish_pending launches the levels still missing).
// Which records leave the one-lane decoder for a wave (>= coopMin successors) or a group of waves (>= giantMin).
// A long record is a serial chain (one lane: ~0.5 us per successor, one wave: ~35 ns, a group: ~5 ns); in a scan of
// the whole C2 graph there is enough other work to hide chains of 2048 / 32768 successors, a job of a few million
// arcs ends when its longest chain does.  Steps measured on C2 sub-ranges (scripts/small_range.py).
void pick_thresholds(const bvg_graph *g, int64_t estArcs, int32_t &coopMin, int32_t &giantMin) {
	coopMin = g->coop_min; giantMin = g->giant_min;
	if (g->adaptive) { // (a lane takes ~0.6 us per successor, a wave ~30 us per record: cnr-2000, 3.2 M arcs, 0.80 ms at 512, 0.62 ms at 128)
		if (estArcs < 8000000) coopMin = 128; else if (estArcs < 32000000) coopMin = 512; else if (estArcs < 80000000) coopMin = 1024;
	}
	if (g->adaptive_giant) {
		// The group class is for the records a WAVE could not finish inside the scan: a wave decodes ~25 ns per successor whatever else runs, so the chain it can hide
		// grows with the job -- but a group of eight waves needs a CU to itself, and many records of middling length lose as groups to 3 000 waves side by side (675
		// records of 32 768 .. 65 535 successors: 0.6 ms more as groups).  Fitted on the best fixed threshold of seven workloads (100 M .. 1 B arcs, three outdegree
		// exponents, deep chains, a web shape; profiles/r5_thresholds.txt): 32 768 x (arcs / 10^8)^0.6, rounded to a power of two, from 8 192 to 2^20 --
		// 100 M arcs 32 768, 200 - 400 M 65 536, 1 B 131 072; within 3 % of the best on all seven (round 4's steps by job size: 9 % off at 100 M arcs, 20 % off on a
		// heavier-tailed outdegree distribution).
		const double target = 32768.0 * std::pow((double)std::max<int64_t>(estArcs, 1) / 1e8, 0.6);
		const int lg = (int)std::lround(std::log2(std::max(target, 1.0)));
		giantMin = 1 << std::min(20, std::max(13, lg));
	}
	giantMin = std::max(giantMin, coopMin);
}

// The lane class of the copy pass reads and writes 16 bytes at a time where its rows are long enough to pay for the bookkeeping:
// cnr-2000 (14.4 ids per row of the class, copied in runs of 6.6) gains 13 % on k_copy_list, the synthetic workloads (6.9 and 8.8 ids,
// runs of 4.5 and 2.4) lose 5-8 %.
bool copy_vec(const bvg_graph *g) {
	if (g->copy_vec >= 0) return g->copy_vec != 0;
	const Staged &s = *g->st;
	return s.lane_rows > 0 && s.lane_ids >= 11 * s.lane_rows;
}

int enqueue_decode(bvg_graph *g, bv::RangeView &v, int64_t estArcs, int32_t &levels, int32_t &giantCap, bool hdrEvent = false) {
	const Staged &s = *g->st;
	int32_t coopMin, giantMin;
	pick_thresholds(g, estArcs, coopMin, giantMin);
	const int32_t W = s.info.window_size;
	int *derr = &g->small.as<Small>()->err;
	bv::GraphDev gd = graph_dev(s);
	levels = 0;
	giantCap = 0;
	{
		// the block tables of the giant records, kept by the parse kernel for the copy pass (GraphDev::walktab): on when the job has a giant class
		g->pend.walkMin = 0x7fffffff;
		if (g->walk_tables && coopMin < 0x7fffffff && s.def != 0 && W > 0 && g->copy_big) {
			// (arcs / 4 ints: two per copied block of the rows of the wave and group classes -- lists denser than one block per eight ids of ALL rows do not fit,
			// and the rows that find no room are walked by one lane inside k_copy_big, correct and slow)
			const size_t cap = (size_t)std::min<int64_t>(std::max<int64_t>(s.arcs_sizing / 4, 1 << 20), 0x7fffffff);
			if (g->walktab.need(sizeof(int32_t) * cap)) g->pend.walkMin = giantMin; // (no room: the copy pass walks the lists itself)
		}
		set_walk(gd, g, g->pend.walkMin);
		// default path: depth + per-level lists; cooperative decode of long records (two classes) next to the
		// one-lane decode of the short ones; then the copy pass level by level over compact lists
		const int64_t arcsBound = std::max<int64_t>(s.arcs_sizing, 1);
		giantCap = (int32_t)std::min<int64_t>(arcsBound / giantMin + 2, 0x7fffffff);
		const int64_t arenaCap = s.info.min_interval_length > 0 ? arcsBound / s.info.min_interval_length + 2 : 1;
		if (!g->depth.need(sizeof(int32_t) * (size_t)v.cnt) || !g->key16.need(sizeof(uint16_t) * (size_t)v.cnt) || !g->lvlist.need(sizeof(int32_t) * (size_t)v.cnt) ||
		    !g->biglist.need(sizeof(int32_t) * (size_t)v.cnt) || !g->giantlist.need(sizeof(int32_t) * (size_t)giantCap) || !g->arena.need((size_t)bv::ARENA_ENTRY_BYTES * (size_t)arenaCap))
			return fail(g, BVG_ENOMEM, "device scratch allocation failed");
		int32_t *hist = g->keys.as<int32_t>(), *keyBase = hist + (bv::NKEYS + 1), *cursor = keyBase + (bv::NKEYS + 1);
		int32_t *ctl = g->coopctl.as<int32_t>();

		// (ctl[0..3], the queues of the long records, are zeroed on side B when the classification starts early)
		const bool early = hdrEvent && coopMin < 0x7fffffff && g->overlap && !g->profile;
		if (!early) HIPCHK(g, hipMemsetAsync(ctl, 0, 4 * sizeof(int32_t), g->stream));
		const bool ctlWasClean = g->ctl_clean; // (zeroed in front of the headers' event: a side stream may use ctl[4..16) as soon as it has seen that event)
		if (!g->ctl_clean) HIPCHK(g, hipMemsetAsync(ctl + 4, 0, 12 * sizeof(int32_t), g->stream));
		g->ctl_clean = false;
		// rows with a reference and >= 1024 (resp. >= copy_mid_min) successors: at most arcs / 1024 (resp. / copy_mid_min) of them
		const int32_t bigCap = (int32_t)std::min<int64_t>(arcsBound / 1024 + 2, 0x3fffffff);
		const int32_t midCap = g->copy_mid_min > 0 ? (int32_t)std::min<int64_t>(arcsBound / g->copy_mid_min + 2, 0x3fffffff) : 0;
		if (!g->copyq.need(sizeof(int32_t) * ((size_t)bigCap + (size_t)midCap))) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
		g->pend.bigCap = bigCap; g->pend.midCap = midCap;
		// scratch tables of the rows that copy more ids than k_copy_big's LDS tables hold (bump-allocated per level, ctl[8]; ctl[9] = head of the level's queue of long rows):
		// <= 4 ints per copied id, and a level's long rows copy a fraction of the arcs; a row that does not fit falls back to one lane
		const uint32_t tmpCap = (uint32_t)std::min<int64_t>(std::max<int64_t>(arcsBound, 1 << 22), 0x7fffffff);
		if (g->copy_big && !g->bigtmp.need(sizeof(int32_t) * (size_t)tmpCap)) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
		g->pend.tmpCap = g->copy_big ? tmpCap : 0;
		const bool coop = coopMin < 0x7fffffff;
		const bool ovl = g->overlap && !g->profile; // per-kernel timing needs the kernels one after the other
		// bvg_scan_checksum: k_parse_list (default codings) and the lane class of the copy pass add the rows they produce to the job's hash (v.hx; the other kernels ignore it)
		g->hash_in_parse = g->hash_job && s.def != 0;
		if (g->hash_job) v.hx = g->hashctx.as<bv::HashCtx>();
		v.coop_min = coop ? coopMin : 0x7fffffff;
		g->last_giant_min = giantMin;
		// Three things run next to each other from here on (unless profiling serialises them):
		//   side B: classification of the long records, then the giant ones (a group of waves each) -- the longest
		//           dependency chains of the scan, which need nothing but the outdegrees and the row starts;
		//   side A: chain depths + per-level lists + copy queues (only the copy pass needs them), then the big records (a wave each);
		//   here:   the parse list and the one-lane parse of everything else.
		hipStream_t stLists = g->stream;
		// parse list (every non-empty record, sorted by work bin inside windows of nodes): needs the outdegrees only, so
		// with the headers' event at hand it is built on side A while the scan of the outdegrees still runs
		int32_t *pKeyBase = nullptr;
		// short records of the default codings: contiguous tiles of the stream, one LDS image each (bv_tile.hpp); the tile
		// bounds follow from the offsets alone
		// BVGPU_TILE=1|2 forces them, 0 forbids them; by default (-1) they are taken when the job is expected to keep only short records
		// in the lane class -- its share (by bits) of the staged records with >= 128 successors fits the wave class, so that
		// k_pick_coop will pick 128 --: neighbouring short records are alike, a tile's lanes stay even, and the coalesced tile
		// kernel is 20 % faster than the bins (cnr-2000 x 30: 0.42 against 0.53 ms); with a heavy-tailed lane class it is 2.6x slower (C2).
		const int tileVariant = job_tile_variant(g, v.lo, v.cnt, v.coop_ptr != nullptr);
		const bool tiles = tileVariant != 0 && s.def != 0;
		// the tables of copy blocks that the one-lane parse leaves for the lane class of the copy pass: 16 bytes per slot (a range of more than 2^28 slots walks the stream as before:
		// 4 GB of tables and more; so does a job that finds no room for them, and one that decodes its lane class with round 4's loop, which writes none)
		void *copyTab = nullptr;
		if (g->copy_tables && s.def != 0 && W > 0 && (tiles || g->lane_loop != 0) && v.cnt <= (1 << 28) && g->copytab.need(16 * (size_t)v.cnt)) copyTab = g->copytab.p;
		int32_t ntiles = 0;
		if (tiles) {
			ntiles = bv::tile_count(s.h_offsets[(size_t)v.lo + v.cnt] - s.h_offsets[(size_t)v.lo], v.cnt);
			if (!g->tilebounds.need(sizeof(int32_t) * ((size_t)ntiles + 2))) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
			bv::launch_tile_bounds(gd, v.lo, v.cnt, ntiles, g->tilebounds.as<int32_t>(), ovl && hdrEvent ? g->sideA : g->stream);
			if (ovl && hdrEvent) HIPCHK(g, hipEventRecord(g->evP, g->sideA));
		}
		const bool earlyList = !tiles && ovl && hdrEvent;
		// The segment pipeline (bv_seg.hip): the residual sections of the hubs -- giant records with >= seg_hub_min successors --, cut into pieces of stream, one lane
		// per piece.  The giants' kernel parses the structure of such a record and hands the residuals over (GraphDev::segDesc).
		const bool segAble = g->seg && g->seg != 2 && !tiles && s.def != 0 && coop && s.seg_long_records >= 0 && !g->hash_job;
		const bool segOn = segAble && (g->seg == 3 || (s.max_outdegree >= g->seg_hub_min && estArcs >= 4000000));
		int32_t segScap = 0;
		bool segReady = false;
		if (segOn) {
			const int64_t bits = s.h_offsets[(size_t)v.lo + v.cnt] - s.h_offsets[(size_t)v.lo];
			// every record with pieces is one of the staged records of the long bins, and has at most bits / piece + 2 of them
			const int64_t recsBound = std::min<int64_t>((int64_t)giantCap, s.seg_long_records);
			segScap = (int32_t)std::min<int64_t>(std::min<int64_t>(bits, s.seg_long_bits) / ((int64_t)1 << bv::seg_bits_log2()) + 2 * recsBound + 2, 0x7ffffff0);
			if (giantCap > 0 && g->segbuf.need(bv::seg_scratch_bytes(giantCap, segScap, s.info.zeta_k))) {
				segReady = true;
				bv::seg_handover(gd, g->segbuf.p, giantCap, segScap, g->seg_hub_min, g->stream); // (before the fork: the cooperative kernels start behind it)
			}
		}
		// A graph without a record of giantMin successors (known since load time: every web-shaped graph at this size) has no giants' kernel to launch -- queued behind the
		// tile / list kernel's blocks its groups, a CU each, would only be scheduled when those drain (cnr-2000 x 30: 600 us on side B for nothing) -- and side B is free for
		// the chain depths and level lists, which otherwise wait behind the wave class on side A (lists_on_b = 3: on side A all the same).
		const bool noGiants = g->skip_empty_giants && s.max_outdegree >= 0 && s.max_outdegree < (int64_t)giantMin;
		const bool listsOnB = (g->lists_on_b == 1 || (g->lists_on_b == 0 && noGiants && ovl && coop && !(tiles && g->level_lists_early))) && !segReady; // (with the hand-over side B carries the segment pipeline's chain)
		const bool listsOnC = g->lists_on_b == 2 && ovl;
		// the parse list's entries carry the records' references (two lines less per record in k_parse_list: bv_kernels.hip) when a slot fits in 28 bits
		const uint16_t *packRef = g->list_refs && !tiles && v.cnt < (1 << 28) ? v.ref : nullptr;
		if (!tiles) {
			if (!g->plist.need(sizeof(int32_t) * (size_t)v.cnt) || !g->pkeys.need(3 * (bv::NKEYS + 1) * sizeof(int32_t)) || !g->pkey16.need(sizeof(uint16_t) * (size_t)v.cnt)) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
			pKeyBase = g->pkeys.as<int32_t>() + (bv::NKEYS + 1);
		}
		// chain depths + per-level lists + copy queues, then the pre-walks of the wave and group classes' block lists: on stL; alone: everything on that one stream
		auto build_levels = [&](hipStream_t stLists, bool alone) -> int {
			if (W > 0) bv::launch_build_lists(gd, v, ~0ull, g->level_bins ? 0 : 1, g->depth.as<int32_t>(), g->key16.as<uint16_t>(), hist, keyBase, cursor, g->lvlist.as<int32_t>(),
			                                  g->giantlist.as<int32_t>(), 0, ctl, &g->small.as<Small>()->maxdepth, stLists,
			                                  g->copyq.as<int32_t>(), bigCap, g->copyq.as<int32_t>() + bigCap, midCap, g->copy_mid_min, g->copy_big != 0);
			// the block lists of the group class's rows, walked beside the parse kernels (k_copy_prewalk)
			g->pend.preDesc = nullptr;
			if (W > 0 && g->prewalk && g->copy_big && gd.walktab && s.def != 0 && g->walkdesc.need(16 * ((size_t)bigCap + (size_t)midCap))) {
				// (the long lists' kernel on side B, which has been idle since the giants -- unless it carries the lists themselves, or the segment pipeline's chain)
				hipStream_t stLong = stLists, stWalk = stLists;
				const bool longKernel = g->prewalk_long != 0;
				const bool cross = !alone && g->prewalk_long != 2 && ovl && coop && !segReady;
				const bool longOnB = cross && longKernel && stLists == g->sideA;   // lists behind the wave class: the long lists on side B
				const bool walkOnA = cross && stLists == side_b(g);                // lists behind the giants: the waves' kernel behind the wave class
				if (longOnB || walkOnA) {
					HIPCHK(g, hipEventRecord(g->evL, stLists));
					if (longOnB) { stLong = side_b(g); HIPCHK(g, hipStreamWaitEvent(stLong, g->evL, 0)); }
					else { stWalk = g->sideA; HIPCHK(g, hipStreamWaitEvent(stWalk, g->evL, 0)); }
				}
				bv::launch_copy_prewalk(gd, s.def, v, g->copyq.as<int32_t>(), bigCap, ctl, g->walkdesc.p, g->prewalk_blocks, stLists, g->prewalk >= 2 ? 0 : midCap, stLong, longKernel, stWalk); // (BVGPU_PREWALK=2: the group class only)
				if (longOnB) HIPCHK(g, hipEventRecord(g->evB, stLong)); // (side B is done when this kernel is)
				g->pend.preDesc = g->walkdesc.p;
			}
			return BVG_OK;
		};
		// A job that decodes its lane class from tiles builds no parse list: side A is idle until the classification is done, and the chain depths, level lists, copy queues and
		// pre-walks -- which need the headers and the stream only -- run there during the set-up instead of behind the wave class, beside the tile kernel (level_lists_early = 0: behind).
		const bool earlyLevels = g->level_lists_early && tiles && ovl && hdrEvent && ctlWasClean && W > 0 && !segReady;
		if (earlyLevels) {
			HIPCHK(g, hipStreamWaitEvent(g->sideA, g->evHdr, 0));
			const int rc = build_levels(g->sideA, true);
			if (rc) return rc;
		}
		const bool keysReady = g->keys_ready; // (this job's: a job that did not go through enqueue_structure -- the masked scan of a dense batch -- must not see the last one's)
		g->keys_ready = false;
		if (earlyList && keysReady) { // k_headers left the keys: three small kernels to the list
			HIPCHK(g, hipStreamWaitEvent(g->sideA, g->evHdr, 0));
			bv::launch_scatter_lists(v.cnt, g->pkey16.as<uint16_t>(), g->pkeys.as<int32_t>(), pKeyBase, pKeyBase + (bv::NKEYS + 1), g->plist.as<int32_t>(), g->giantlist.as<int32_t>(), ctl,
			                         &g->small.as<Small>()->pad, g->sideA, packRef);
			HIPCHK(g, hipEventRecord(g->evP, g->sideA));
		}
		else if (earlyList) {
			HIPCHK(g, hipStreamWaitEvent(g->sideA, g->evHdr, 0));
			bv::launch_build_lists(gd, v, ~0ull, g->parse_windows ? 6 : 2, nullptr, g->pkey16.as<uint16_t>(), g->pkeys.as<int32_t>(), pKeyBase, pKeyBase + (bv::NKEYS + 1), g->plist.as<int32_t>(),
			                       g->giantlist.as<int32_t>(), 0, ctl, &g->small.as<Small>()->pad, g->sideA, nullptr, 0, nullptr, 0, 0, false, packRef);
			HIPCHK(g, hipEventRecord(g->evP, g->sideA));
		}
		if (ovl) {
			if (early) { // side B: classification and sort of the long records next to the scan (they need the outdegrees only)
				HIPCHK(g, hipStreamWaitEvent(side_b(g), g->evHdr, 0));
				HIPCHK(g, hipMemsetAsync(ctl, 0, 4 * sizeof(int32_t), side_b(g)));
				bv::launch_classify(v.cnt, v.outd, v.coop_ptr, coopMin, giantMin, g->biglist.as<int32_t>(), g->giantlist.as<int32_t>(), giantCap, ctl, side_b(g));
				HIPCHK(g, hipEventRecord(g->evC, side_b(g)));
			}
			HIPCHK(g, hipEventRecord(g->evFork, g->stream));
			HIPCHK(g, hipStreamWaitEvent(g->sideA, g->evFork, 0));
			HIPCHK(g, hipStreamWaitEvent(side_b(g), g->evFork, 0));
			stLists = listsOnB ? side_b(g) : listsOnC ? g->sideC : g->sideA;
			if (listsOnC) HIPCHK(g, hipStreamWaitEvent(g->sideC, g->evFork, 0));
			if (g->early_rowptr) { // the caller's rowptr needs the scan only: written now, not at the end of the call
				if (v.rowstart != g->early_rowptr) bv::launch_rebase(v.nh, v.cnt, v.rowstart, g->early_rowptr, g->sideA);
				hipLaunchKernelGGL(k_totals, dim3(1), dim3(1), 0, g->sideA, v.rowstart, v.nh, v.cnt, g->small.as<Small>(), v.coop_ptr, v.coop_min);
			}
			if (coop && !early) {
				bv::launch_classify(v.cnt, v.outd, v.coop_ptr, coopMin, giantMin, g->biglist.as<int32_t>(), g->giantlist.as<int32_t>(), giantCap, ctl, side_b(g));
				HIPCHK(g, hipEventRecord(g->evC, side_b(g)));
			}
		}
		// the long records first on both side streams: giants on B, the wave class on A ...
		if (ovl && coop) {
			HIPCHK(g, hipStreamWaitEvent(g->sideA, g->evC, 0));
			// the giants' groups take a CU each: started while the parse list is still being scattered they starve k_scatter_keys of its blocks (profiles/r4_experiments.txt section 9:
			// 48 -> 402 us) -- until round 5 the order rested on the scan of the outdegrees being the slower of the two chains; now the giants WAIT for the list (giants_after_list = 0: as before)
			if (earlyList && g->giants_after_list && !noGiants) HIPCHK(g, hipStreamWaitEvent(side_b(g), g->evP, 0));
			// (a tile job without giants whose level lists and pre-walks went to side A during the set-up: the wave class on side B, which is idle, beside the tile kernel --
			// behind the pre-walks it started when the tile kernel ended and ran 0.2 ms alone on cnr-2000 x 30)
			const bool wavesOnB = g->waves_on_b && noGiants && earlyLevels;
			bv::launch_parse_big(gd, s.def, v, g->biglist.as<int32_t>(), g->giantlist.as<int32_t>(), ctl, g->arena.p, arenaCap, g->coop_waves, noGiants ? 0 : g->giant_groups, derr, side_b(g), wavesOnB ? side_b(g) : g->sideA, g->wait_giants); // (giants, big)
			if (!segReady) HIPCHK(g, hipEventRecord(g->evB, side_b(g))); // (else: behind the segment pipeline's chain, below)
		}
		// ... then, behind the wave class, the chain depth of every record + per-level lists (node order inside a level) +
		// copy queues: only the copy pass needs them, and launched first they would sit in front of the wave class while
		// the one-lane kernel holds every CU
		if (!earlyLevels) { const int rc = build_levels(stLists, false); if (rc) return rc; }
		if (listsOnC) HIPCHK(g, hipEventRecord(g->evL, g->sideC));
		auto hash_phase_a = [&](hipStream_t stH) { // the node numbers and the rows without a reference that the wave / group classes decoded (their lists), or every such row (codings the one-lane parse does not hash)
			const int32_t pcap = (int32_t)std::min<int64_t>((int64_t)v.cnt + arcsBound / 4096 + 2, 0x7ffffff0);
			if (!g->hashq.need(sizeof(int64_t) * (size_t)pcap)) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
			const bool lists = g->hash_in_parse && coop;
			bv::launch_hash_rest(v, g->hash_in_parse ? 0 : 1, g->hash_in_parse, true, g->copy_mid_min, g->copy_big != 0, g->hashq.p, (int32_t *)((char *)g->hashctx.p + HASH_ACC_OFF + 4), pcap, stH,
			                     lists ? g->biglist.as<int32_t>() : nullptr, ctl + 0, v.cnt, lists ? g->giantlist.as<int32_t>() : nullptr, ctl + 1, giantCap, false);
			return (int)BVG_OK;
		};
		// (codings other than the default set: the one-lane parse does not hash, k_hash_rest reads the rows of ITS class from memory too -- behind it, below;
		// beside it the fold read rows that were not written yet: fuzz_params.py, hashCode of graphs with non-default flags)
		if (g->hash_job && ovl && g->hash_in_parse) { // behind side A's chain, once the giants are done too: beside the tail of the one-lane parse
			if (coop) HIPCHK(g, hipStreamWaitEvent(g->sideA, g->evB, 0));
			const int rc = hash_phase_a(g->sideA);
			if (rc) return rc;
		}
		if (ovl) HIPCHK(g, hipEventRecord(g->evA, g->sideA));
		if (ovl && coop && listsOnB) HIPCHK(g, hipEventRecord(g->evB, side_b(g))); // (the lists sit behind the giants: side B is done when they are)
		if (!tiles && !earlyList)
			bv::launch_build_lists(gd, v, ~0ull, g->parse_windows ? 6 : 2, nullptr, g->pkey16.as<uint16_t>(), g->pkeys.as<int32_t>(), pKeyBase, pKeyBase + (bv::NKEYS + 1), g->plist.as<int32_t>(),
			                       g->giantlist.as<int32_t>(), 0, ctl, &g->small.as<Small>()->pad, g->stream, nullptr, 0, nullptr, 0, 0, false, packRef);
		if (earlyList || (tiles && ovl && hdrEvent)) HIPCHK(g, hipStreamWaitEvent(g->stream, g->evP, 0));
		if (coop && !ovl) bv::launch_classify(v.cnt, v.outd, v.coop_ptr, coopMin, giantMin, g->biglist.as<int32_t>(), g->giantlist.as<int32_t>(), giantCap, ctl, g->stream);
		mark(g, 3);
		if (coop && !ovl) bv::launch_parse_giants(gd, s.def, v, g->giantlist.as<int32_t>(), ctl, g->arena.p, arenaCap, g->giant_groups, derr, g->stream);
		mark(g, 4);
		if (coop && !ovl) bv::launch_parse_waves(gd, s.def, v, g->biglist.as<int32_t>(), ctl, g->arena.p, arenaCap, g->coop_waves, derr, g->stream);
		mark(g, 5);
		if (tiles) bv::launch_parse_tile(gd, s.def, v, g->tilebounds.as<int32_t>(), ntiles, tileVariant | (g->tile_loop && g->lane_loop ? 0x100 : 0), derr, g->stream, g->arena.p, arenaCap, copyTab);
		else {
			if (segReady) { // on side B, behind the giants: the pieces of the records that handed their residual sections over
				hipStream_t stChain = g->stream;
				if (ovl) {
					stChain = side_b(g);
					HIPCHK(g, hipEventRecord(g->evM, g->stream)); // (the row starts are ready)
					HIPCHK(g, hipStreamWaitEvent(stChain, g->evM, 0));
				}
				bv::launch_seg_chain(gd, s.def, v, giantCap, segScap, g->segbuf.p, g->arena.p, arenaCap, ctl, g->seg_blocks, derr, stChain);
				if (ovl) HIPCHK(g, hipEventRecord(g->evB, stChain));
			}
			if (ovl && coop) { HIPCHK(g, hipStreamWaitEvent(g->stream, g->evC, 0)); if (g->wait_giants && !noGiants) bv::launch_wait_giants(ctl, g->giant_groups, g->stream); } // (the giants first: k_wait_giants)
			bv::launch_parse_list(gd, s.def, v, g->plist.as<int32_t>(), pKeyBase, g->level_blocks, derr, g->stream, g->arena.p, arenaCap, 0, bv::NKEYS, g->lane_loop != 0, copyTab, packRef != nullptr);
		}
		if (ovl) {
			HIPCHK(g, hipStreamWaitEvent(g->stream, g->evA, 0));
			if (coop) HIPCHK(g, hipStreamWaitEvent(g->stream, g->evB, 0));
			if (listsOnC) HIPCHK(g, hipStreamWaitEvent(g->stream, g->evL, 0));
		}
		mark(g, 6);
		if (g->hash_job && (!ovl || !g->hash_in_parse)) { const int rc = hash_phase_a(g->stream); if (rc) return rc; }
		g->pend.copyTab = g->hash_job ? nullptr : copyTab; // (the hash fold rides on the lane class's stream-walking merge)
		g->pend.tabArena = g->arena.p;
		g->pend.tabArenaCap = arenaCap;
		if (W > 0) {
			levels = g->levels_hint;
			for (int32_t l = 1; l <= levels; l++) {
				bv::launch_copy_level(gd, s.def, v, g->depth.as<int32_t>(), g->lvlist.as<int32_t>(), keyBase, l, g->level_blocks, g->copy_mid_min, g->copy_big != 0,
				                                          g->copyq.as<int32_t>(), bigCap, g->copyq.as<int32_t>() + bigCap, midCap, ctl, g->bigtmp.as<int32_t>(), g->pend.tmpCap, derr,
				                                          g->stream, ovl ? side_b(g) : g->stream, ovl ? g->sideA : g->stream, g->evFork, g->evB, g->evA, g->pend.preDesc, g->prewalk < 2 && g->pend.midCap > 0, (copy_vec(g) ? 1 : 0) | (g->copy_loop ? 2 : 0) | (g->mid_tables ? 16 : 0),
				                                          g->pend.tabArena, g->pend.tabArenaCap, g->pend.copyTab);
			}
		}
		if (g->hash_job) { // the rows that the wave / group classes of the copy pass merged (its queues), then the sum rides home in the job's mailbox
			if (W > 0 && g->copy_big)
				bv::launch_hash_rest(v, 0, g->hash_in_parse, true, g->copy_mid_min, true, g->hashq.p, (int32_t *)((char *)g->hashctx.p + HASH_ACC_OFF + 8), (int32_t)(g->hashq.cap / sizeof(int64_t)), g->stream,
				                     g->copyq.as<int32_t>(), ctl + 5, bigCap, midCap > 0 ? g->copyq.as<int32_t>() + bigCap : nullptr, ctl + 6, midCap, true);
			bv::launch_hash_sum(g->hashctx.as<bv::HashCtx>(), &g->small.as<Small>()->hash, g->stream);
		}
	}
	return BVG_OK;
}

// Device-pointer core of bvg_decode_range.  rowptr_dev: to-from+1 int64; succ_dev may be NULL (count only).
// ---- EFGraph (bv_ef.hip): no references between records, so a job is outdegrees -> scan -> one decode pass
constexpr int32_t EF_BIG_MIN = 256;     // lists of that many successors are decoded by a wave each ...
constexpr int32_t EF_GIANT_MIN = 2048; // ... and from here on by a wave per round of 64 words of upper bits

bv::EfDev ef_dev(const Staged &s) {
	return bv::EfDev{ (const uint64_t *)s.d_bits, (s.nwords + 1) / 2, s.d_offsets, s.info.nodes, (uint64_t)s.info.ef_upper_bound, s.info.ef_log2_quantum };
}

// slots <-> nodes[0..cnt) (device pointer) or from + s; rowptr_dev[cnt + 1] and succ_dev (may be null: outdegrees only) on the device
int ef_job(bvg_graph *g, const int32_t *d_nodes, int32_t from, int64_t cnt, int64_t *rowptr_dev, int32_t *succ_dev, size_t succ_cap, uint64_t *arcs_out) {
	const Staged &s = *g->st;
	Small *dsm = g->small.as<Small>();
	HIPCHK(g, hipMemsetAsync(g->small.p, 0, sizeof(Small), g->stream));
	if (cnt == 0) {
		HIPCHK(g, hipMemsetAsync(rowptr_dev, 0, sizeof(int64_t), g->stream));
		HIPCHK(g, hipStreamSynchronize(g->stream));
		g->last_arcs = 0;
		if (arcs_out) *arcs_out = 0;
		return BVG_OK;
	}
	if (!g->outd.need(sizeof(int32_t) * (size_t)cnt) || !g->sums.need(sizeof(int64_t) * (size_t)(bv::scan_num_sums(cnt) + 1)))
		return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	int32_t *nbig = g->coopctl.as<int32_t>();
	HIPCHK(g, hipMemsetAsync(nbig, 0, 2 * sizeof(int32_t), g->stream)); // [0] rounds of the giant lists (k_ef_rank)
	const bv::EfDev gd = ef_dev(s);
	bv::launch_ef_outdeg(gd, d_nodes, from, cnt, g->outd.as<int32_t>(), &dsm->err, g->stream);
	bv::launch_scan(g->outd.as<int32_t>(), cnt, rowptr_dev, g->sums.as<int64_t>(), g->stream);
	HIPCHK(g, hipMemcpyAsync(&dsm->total, rowptr_dev + cnt, sizeof(int64_t), hipMemcpyDeviceToDevice, g->stream));
	if (succ_dev) { // the long lists on a side stream, next to the short ones
		HIPCHK(g, hipEventRecord(g->evFork, g->stream));
		HIPCHK(g, hipStreamWaitEvent(g->sideA, g->evFork, 0));
		HIPCHK(g, hipStreamWaitEvent(g->sideB, g->evFork, 0));
		// rounds of the giant lists: one per 64 words of the stream plus one per list when every list is asked for once (more queries
		// for giant lists than that fits are decoded the slow way, k_ef_rank)
		const uint64_t chunkCap64 = gd.nwords / 64 + (uint64_t)cnt / 16 + 64;
		const uint32_t chunkCap = (uint32_t)std::min<uint64_t>(chunkCap64, 0x7fffffffu);
		if (!g->arena.need((size_t)chunkCap * bv::ef_chunk_bytes())) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
		bv::launch_ef_decode(gd, d_nodes, from, cnt, EF_BIG_MIN, rowptr_dev, succ_dev, (uint64_t)succ_cap, &dsm->err, EF_GIANT_MIN, g->arena.p, chunkCap, (uint32_t *)nbig, g->stream,
		                     g->sideA, g->sideB);
		HIPCHK(g, hipEventRecord(g->evA, g->sideA));
		HIPCHK(g, hipStreamWaitEvent(g->stream, g->evA, 0));
		HIPCHK(g, hipEventRecord(g->evB, g->sideB));
		HIPCHK(g, hipStreamWaitEvent(g->stream, g->evB, 0));
	}
	HIPCHK(g, hipMemsetAsync(nbig, 0, 2 * sizeof(int32_t), g->stream)); // (the BV jobs expect their control block zeroed)
	int rc = fetch_small(g);
	if (rc) return rc;
	g->last_arcs = (uint64_t)g->h_small->total;
	if (arcs_out) *arcs_out = g->last_arcs;
	if (g->h_small->err & bv::E_REF) return fail(g, BVG_EARG, "Node index out of range");
	if (g->h_small->err & bv::E_FORMAT) return fail(g, BVG_EFORMAT, "malformed bit stream");
	if (g->h_small->err & bv::E_CAP) return fail(g, BVG_ECAP, "successor buffer too small");
	return BVG_OK;
}

int ef_decode_range_device(bvg_graph *g, int32_t from, int32_t to, int64_t *rowptr_dev, int32_t *succ_dev, size_t succ_cap, uint64_t *arcs_out) {
	const Staged &s = *g->st;
	HIPCHK(g, hipSetDevice(s.device));
	if (from < s.node_lo || to > s.node_hi) return fail(g, BVG_EARG, "node range outside the slice this handle stages (bvg_open_shard)");
	{ int rc = fork_from_user(g); if (rc) return rc; }
	int rc = ef_job(g, nullptr, from, (int64_t)to - from, rowptr_dev, succ_dev, succ_cap, arcs_out);
	if (rc) return rc;
	return join_to_user(g);
}

// host outputs: the job runs into staging buffers on the device, the results cross PCIe afterwards
int ef_to_host(bvg_graph *g, const int32_t *nodes_h, int32_t from, int64_t cnt, int64_t *rowptr_h, int32_t *succ_h, size_t succ_cap, uint64_t *arcs_out) {
	const Staged &s = *g->st;
	HIPCHK(g, hipSetDevice(s.device));
	{ int rc = fork_from_user(g); if (rc) return rc; }
	if (!g->stage_rowptr.need(sizeof(int64_t) * ((size_t)cnt + 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
	const int32_t *d_nodes = nullptr;
	if (nodes_h) {
		if (!g->stage_nodes.need(sizeof(int32_t) * std::max<size_t>((size_t)cnt, 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
		if (cnt) HIPCHK(g, hipMemcpyAsync(g->stage_nodes.p, nodes_h, sizeof(int32_t) * (size_t)cnt, hipMemcpyHostToDevice, g->stream));
		d_nodes = g->stage_nodes.as<int32_t>();
	}
	// outdegrees first: the staging buffer is sized by what the range holds
	uint64_t arcs = 0;
	int rc = ef_job(g, d_nodes, from, cnt, g->stage_rowptr.as<int64_t>(), nullptr, 0, &arcs);
	if (rc) return rc;
	if (arcs_out) *arcs_out = arcs;
	HIPCHK(g, hipMemcpy(rowptr_h, g->stage_rowptr.p, sizeof(int64_t) * ((size_t)cnt + 1), hipMemcpyDeviceToHost));
	if (!succ_h) return BVG_OK;
	if (arcs > succ_cap) return fail(g, BVG_ECAP, "successor buffer too small");
	if (!g->stage_succ.need(sizeof(int32_t) * std::max<size_t>((size_t)arcs, 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
	rc = ef_job(g, d_nodes, from, cnt, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), (size_t)arcs, nullptr);
	if (rc) return rc;
	if (arcs) HIPCHK(g, hipMemcpy(succ_h, g->stage_succ.p, sizeof(int32_t) * (size_t)arcs, hipMemcpyDeviceToHost));
	return BVG_OK;
}

// ImmutableGraph.hashCode() over [from, to) of an EFGraph folded into *h without a successor ever being written (bv_ef.hip, HASH mode):
// the write side of the scan is 8 bytes per 256 nodes
int ef_scan_checksum(bvg_graph *g, int32_t from, int32_t to, int32_t *h, uint64_t *arcs_out) {
	const Staged &s = *g->st;
	if (from < s.node_lo || to > s.node_hi) return fail(g, BVG_EARG, "node range outside the slice this handle stages (bvg_open_shard)");
	const int64_t cnt = (int64_t)to - from;
	if (cnt == 0) { if (arcs_out) *arcs_out = 0; return BVG_OK; }
	{ int rc = fork_from_user(g); if (rc) return rc; }
	Small *dsm = g->small.as<Small>();
	HIPCHK(g, hipMemsetAsync(g->small.p, 0, sizeof(Small), g->stream));
	const bv::EfDev gd = ef_dev(s);
	const uint64_t chunkCap64 = gd.nwords / 64 + (uint64_t)cnt / 16 + 64;
	const uint32_t chunkCap = (uint32_t)std::min<uint64_t>(chunkCap64, 0x7fffffffu);
	if (!g->outd.need(sizeof(int32_t) * (size_t)cnt) || !g->sums.need(sizeof(int64_t) * (size_t)(bv::scan_num_sums(cnt) + 1)) || !g->stage_rowptr.need(sizeof(int64_t) * ((size_t)cnt + 1)) ||
	    !g->stage_nodes.need(sizeof(uint32_t) * (size_t)cnt) || !g->hashA.need((size_t)bv::ef_hash_blocks(cnt) * 8) || !g->arena.need((size_t)chunkCap * bv::ef_chunk_bytes()))
		return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	int32_t *nbig = g->coopctl.as<int32_t>();
	HIPCHK(g, hipMemsetAsync(nbig, 0, 2 * sizeof(int32_t), g->stream));
	HIPCHK(g, hipMemsetAsync(g->stage_nodes.p, 0, sizeof(uint32_t) * (size_t)cnt, g->stream)); // the long lists' accumulators
	HIPCHK(g, hipMemcpyAsync(&dsm->hash, h, sizeof(int32_t), hipMemcpyHostToDevice, g->stream));
	int64_t *rowstart = g->stage_rowptr.as<int64_t>();
	bv::launch_ef_outdeg(gd, nullptr, from, cnt, g->outd.as<int32_t>(), &dsm->err, g->stream);
	bv::launch_scan(g->outd.as<int32_t>(), cnt, rowstart, g->sums.as<int64_t>(), g->stream);
	HIPCHK(g, hipMemcpyAsync(&dsm->total, rowstart + cnt, sizeof(int64_t), hipMemcpyDeviceToDevice, g->stream));
	HIPCHK(g, hipEventRecord(g->evFork, g->stream));
	HIPCHK(g, hipStreamWaitEvent(g->sideA, g->evFork, 0));
	HIPCHK(g, hipStreamWaitEvent(g->sideB, g->evFork, 0));
	bv::launch_ef_hash(gd, from, cnt, EF_BIG_MIN, rowstart, g->stage_nodes.as<uint32_t>(), &dsm->err, EF_GIANT_MIN, g->arena.p, chunkCap, (uint32_t *)nbig, g->stream, g->sideA, g->sideB);
	HIPCHK(g, hipEventRecord(g->evA, g->sideA));
	HIPCHK(g, hipStreamWaitEvent(g->stream, g->evA, 0));
	HIPCHK(g, hipEventRecord(g->evB, g->sideB));
	HIPCHK(g, hipStreamWaitEvent(g->stream, g->evB, 0));
	bv::launch_ef_hash_fold(from, cnt, rowstart, g->stage_nodes.as<uint32_t>(), g->hashA.p, &dsm->hash, g->stream);
	HIPCHK(g, hipMemsetAsync(nbig, 0, 2 * sizeof(int32_t), g->stream));
	int rc = fetch_small(g);
	if (rc) return rc;
	if (g->h_small->err) return fail(g, BVG_EFORMAT, "malformed bit stream");
	*h = g->h_small->hash;
	g->last_arcs = (uint64_t)g->h_small->total;
	if (arcs_out) *arcs_out = g->last_arcs;
	return join_to_user(g);
}

int decode_range_device(bvg_graph *g, int32_t from, int32_t to, int64_t *rowptr_dev, int32_t *succ_dev, size_t succ_cap, bool async, uint64_t *arcs_out) {
	const Staged &s = *g->st;
	if (s.info.format == BVG_FORMAT_EF) return ef_decode_range_device(g, from, to, rowptr_dev, succ_dev, succ_cap, arcs_out);
	HIPCHK(g, hipSetDevice(s.device));
	if (g->pend.active) { int rc = finish_pending(g, nullptr); if (rc) return rc; }
	g->ctl_clean = false; // (only this job's own k_pick_coop vouches for the counters: a job that failed half-way leaves them dirty -- ADVICE r2)
	const int32_t W = s.info.window_size;
	if (from < s.node_lo || to > s.node_hi) return fail(g, BVG_EARG, "node range outside the slice this handle stages (bvg_open_shard)");
	{ int rc = fork_from_user(g); if (rc) return rc; }
	HIPCHK(g, hipMemsetAsync(g->small.p, 0, sizeof(Small), g->stream));
	if (g->hash_job) { int rc = hash_prepare(g); if (rc) return rc; }
	if (to == from) {
		HIPCHK(g, hipMemsetAsync(rowptr_dev, 0, sizeof(int64_t), g->stream));
		{ int rc = join_to_user(g); if (rc) return rc; }
		if (!async) HIPCHK(g, hipStreamSynchronize(g->stream));
		g->last_arcs = 0;
		if (arcs_out) *arcs_out = 0;
		return BVG_OK;
	}
	// halo: only needed when successors are wanted and the range does not start at node 0
	int32_t nh = 0;
	if (succ_dev && from > 0 && W > 0) {
		const int64_t mr = s.info.max_ref_count < 1 ? 1 : std::min(s.info.max_ref_count, 64);
		nh = (int32_t)std::min<int64_t>(from - s.stage_lo, (int64_t)W * mr);
	}
	bv::RangeView v;
	// A sub-range needs a halo whose depth and size are only known on the device.  The common case (chains no deeper
	// than maxrefcount says, rows that fit the scratch buffer of the previous calls) is decoded without asking: the
	// kernels check every halo row against the capacity, and finish_pending repeats the call the slow way if a chain
	// escaped or a row did not fit.
	const bool optimistic = nh > 0 && succ_dev && !g->force_halo_sync && g->overlap && !g->profile;
	if (optimistic) {
		if (!g->halo.need(std::max<size_t>(g->halo.cap, g->halo_min))) return fail(g, BVG_ENOMEM, "halo allocation failed");
		int rc = enqueue_structure(g, from, to, nh, v, succ_dev != nullptr);
		if (rc) return rc;
		v.halo_cap = g->halo.cap / sizeof(int32_t);
	}
	else for (;;) {
		int rc = enqueue_structure(g, from, to, nh, v, succ_dev != nullptr, rowptr_dev);
		if (rc) return rc;
		if (nh == 0) break;
		// the halo buffer size and the "chain escaped the window" flag need a round trip
		hipLaunchKernelGGL(k_totals, dim3(1), dim3(1), 0, g->stream, v.rowstart, v.nh, v.cnt, g->small.as<Small>(), (const int32_t *)nullptr, 0);
		rc = fetch_small(g);
		if (rc) return rc;
		if (g->h_small->err & bv::E_ESCAPED) {
			if (nh == from - s.stage_lo) {
				// a valid file with reference chains deeper than the room a shard stages before its slice (files written with a huge or unlimited maxRefCount)
				if (s.stage_lo) return fail(g, BVG_EUNSUPPORTED, "a reference chain of the slice's first rows reaches before the nodes staged for it (max(4096, 64 x window) nodes): open the graph with bvg_open");
				return fail(g, BVG_EFORMAT, "reference chain runs before node 0");
			}
			nh = (int32_t)std::min<int64_t>(from - s.stage_lo, (int64_t)nh * 8);
			HIPCHK(g, hipMemsetAsync(g->small.p, 0, sizeof(Small), g->stream));
			continue;
		}
		if (!g->halo.need(sizeof(int32_t) * (size_t)std::max<int64_t>(g->h_small->halo_total, 1))) return fail(g, BVG_ENOMEM, "halo allocation failed");
		break;
	}
	v.succ = succ_dev; v.halo = g->halo.as<int32_t>(); v.succ_cap = succ_cap;
	if (nh > 0) v.halo_cap = g->halo.cap / sizeof(int32_t); // (the real size in every case: the copy pass's 16-byte windows may read up to three ids past a row, never past the buffer)
	int32_t levels = 0;
	int32_t giantCap = 0;
	g->early_rowptr = succ_dev && g->overlap && !g->profile ? rowptr_dev : nullptr;
	if (succ_dev) {
		// arcs of the job, estimated from its share of the bit stream (the true count is still on the device)
		const int64_t bits = s.h_offsets[to] - s.h_offsets[from - nh], allBits = std::max<int64_t>(s.h_offsets.back(), 1);
		const int64_t estArcs = (int64_t)((double)s.info.arcs * (double)bits / (double)allBits);
		int rc = enqueue_decode(g, v, estArcs, levels, giantCap, true);
		if (rc) return rc;
	}
	if (!succ_dev) { mark(g, 3); mark(g, 4); mark(g, 5); mark(g, 6); }
	mark(g, 7);
	if (!g->early_rowptr) {
		if (v.rowstart != rowptr_dev) bv::launch_rebase(v.nh, v.cnt, v.rowstart, rowptr_dev, g->stream);
		hipLaunchKernelGGL(k_totals, dim3(1), dim3(1), 0, g->stream, v.rowstart, v.nh, v.cnt, g->small.as<Small>(), v.coop_ptr, v.coop_min);
	}
	g->early_rowptr = nullptr;
	mark(g, 8);
	g->ev_valid = g->profile;
	{ int rc = join_to_user(g); if (rc) return rc; }
	HIPCHK(g, hipGetLastError());
	if (g->hash_job) g->hash_view = v;
	g->pend.active = true; g->pend.view = v; g->pend.levels_done = levels; g->pend.want_succ = succ_dev != nullptr; g->pend.giantCap = giantCap;
	g->pend.optimistic = optimistic; g->pend.from = from; g->pend.to = to; g->pend.rowptr = rowptr_dev; g->pend.succ = succ_dev; g->pend.succ_cap = succ_cap;
	if (async) return BVG_OK;
	return finish_pending(g, arcs_out);
}

} // namespace

// ================================================================================================ C ABI

extern "C" int bvg_parse_properties(const char *basename, bvg_info_t *out, char *errbuf, size_t errlen) {
	if (!basename || !out) return BVG_EARG;
	std::string err;
	int rc = bvh::parse_properties(basename, *out, err);
	if (errbuf && errlen) { strncpy(errbuf, err.c_str(), errlen - 1); errbuf[errlen - 1] = 0; }
	return rc;
}

extern "C" int64_t bvg_flags_from_string(const char *s) { return bvh::flags_from_string(s ? s : ""); }

extern "C" int bvg_decode_offsets_host(const uint8_t *offsets_file, size_t len, int32_t nodes, int offset_coding, int64_t *out) {
	if (!offsets_file || !out || nodes < 0) return BVG_EARG;
	return bvh::decode_offsets(offsets_file, len, nodes, offset_coding, out);
}

extern "C" int bvg_decode_offsets_device(int device, const uint8_t *offsets_file, size_t len, int32_t nodes, int offset_coding, int64_t *out) {
	if (!offsets_file || !out || nodes < 0) return BVG_EARG;
	if (offset_coding != BVG_GAMMA && offset_coding != BVG_DELTA) return BVG_EUNSUPPORTED;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BVG_EHIP;
	if (device < 0 || device >= ndev) return BVG_EARG;
	if (hipSetDevice(device) != hipSuccess) return BVG_EHIP;
	const uint64_t ow = (len + 3) / 4;
	uint32_t *d_ow = nullptr;
	int64_t *d_out = nullptr;
	int rc = BVG_ENOMEM;
	if (hipMalloc((void **)&d_ow, (size_t)(ow + 8) * 4) == hipSuccess && hipMalloc((void **)&d_out, sizeof(int64_t) * ((size_t)nodes + 1)) == hipSuccess) {
		rc = BVG_EHIP;
		if (hipMemset(d_ow, 0, (size_t)(ow + 8) * 4) == hipSuccess && (len == 0 || hipMemcpy(d_ow, offsets_file, len, hipMemcpyHostToDevice) == hipSuccess)) {
			if (len > 0 && bv::offsets_decode_device(d_ow, ow, (uint64_t)len * 8, nodes, d_out, nullptr, offset_coding == BVG_DELTA) == 0)
				rc = hipMemcpy(out, d_out, sizeof(int64_t) * ((size_t)nodes + 1), hipMemcpyDeviceToHost) == hipSuccess ? BVG_OK : BVG_EHIP;
			else rc = BVG_EFORMAT;
		}
	}
	if (d_ow) (void)hipFree(d_ow);
	if (d_out) (void)hipFree(d_out);
	return rc;
}

static int open_impl(const char *basename, int device, int part, int parts, bvg_t **out);
extern "C" int bvg_open(const char *basename, int device, bvg_t **out) { return open_impl(basename, device, 0, 1, out); }
extern "C" int bvg_open_shard(const char *basename, int device, int part, int parts, bvg_t **out) {
	if (parts < 1 || part < 0 || part >= parts) { if (out) *out = nullptr; return BVG_EARG; }
	return open_impl(basename, device, part, parts, out);
}
// <path>[lo, hi) -> device memory at dst, through two small pinned buffers: the read of piece k + 1 (page cache -> pinned) runs while piece k crosses PCIe.
// (A std::vector of the whole file and one pageable hipMemcpy cost three passes over the bytes on the host -- zero fill, fread, the runtime's own staging copy --
// one after the other: 190 MB of C2 in 60 ms; this way 25.)  Returns 0, or a BVG_ code with `err` set.
static int stage_file_range(FILE *f, const std::string &path, size_t lo, size_t hi, uint8_t *dst, std::string &err) {
	constexpr size_t PIECE = (size_t)4 << 20;
	if (hi <= lo) return BVG_OK;
	uint8_t *pin[2] = { nullptr, nullptr };
	hipEvent_t ev[2] = { nullptr, nullptr };
	hipStream_t st = nullptr;
	int rc = BVG_OK;
	auto done = [&]() {
		if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
		for (int i = 0; i < 2; i++) { if (ev[i]) (void)hipEventDestroy(ev[i]); if (pin[i]) (void)hipHostFree(pin[i]); }
		return rc;
	};
	const size_t piece = std::min(PIECE, hi - lo);
	if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; (void)hipGetLastError(); err = "hipStreamCreate failed"; rc = BVG_EHIP; return done(); }
	for (int i = 0; i < 2; i++)
		if (hipHostMalloc((void **)&pin[i], piece, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); err = "pinned staging buffers: allocation failed"; rc = BVG_ENOMEM; return done(); }
	if (fseeko(f, (off_t)lo, SEEK_SET) != 0) { err = "cannot seek in " + path; rc = BVG_EIO; return done(); }
	int k = 0;
	for (size_t at = lo; at < hi; k ^= 1) {
		const size_t len = std::min(piece, hi - at);
		if (at >= lo + 2 * piece && hipEventSynchronize(ev[k]) != hipSuccess) { err = "staging " + path + " failed"; rc = BVG_EHIP; return done(); } // the buffer's last piece has left
		if (fread(pin[k], 1, len, f) != len) { err = "short read on " + path; rc = BVG_EIO; return done(); }
		if (hipMemcpyAsync(dst + (at - lo), pin[k], len, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[k], st) != hipSuccess) { err = "staging " + path + " failed"; rc = BVG_EHIP; return done(); }
		at += len;
	}
	if (hipStreamSynchronize(st) != hipSuccess) { err = "staging " + path + " failed"; rc = BVG_EHIP; }
	return done();
}

static int open_impl(const char *basename, int device, int part, int parts, bvg_t **out) {
	if (!basename || !out) return BVG_EARG;
	*out = nullptr;
	auto *g = new bvg_graph();
	*out = g; // returned even on failure so that bvg_last_error works; caller still closes it
	auto st = std::make_shared<Staged>();
	std::string err;
	int rc = bvh::parse_properties(basename, st->info, err);
	if (rc) return fail(g, rc, err);
	const bvg_info_t &in = st->info;
	const bool ef = in.format == BVG_FORMAT_EF;
	auto okc = [](int c, std::initializer_list<int> l) { for (int v : l) if (c == v) return true; return false; };
	// the switch statements of BVG:631-816 accept exactly these
	if (!ef && (!okc(in.outdegree_coding, { BVG_GAMMA, BVG_DELTA }) || !okc(in.block_coding, { BVG_UNARY, BVG_GAMMA, BVG_DELTA }) ||
	    !okc(in.block_count_coding, { BVG_UNARY, BVG_GAMMA, BVG_DELTA }) || !okc(in.reference_coding, { BVG_UNARY, BVG_GAMMA, BVG_DELTA }) ||
	    !okc(in.residual_coding, { BVG_GAMMA, BVG_ZETA, BVG_DELTA, BVG_GOLOMB, BVG_NIBBLE }) || !okc(in.offset_coding, { BVG_GAMMA, BVG_DELTA })))
		return fail(g, BVG_EUNSUPPORTED, "The required coding is not supported");
	st->basename = basename;
	// kernel variant: 1 = every coding is the default one and zeta_3 (constants folded in), 2 = the default codings
	// with another zeta_k (taken at run time), 0 = anything else (generic readers)
	const bool defaults = in.outdegree_coding == BVG_GAMMA && in.block_coding == BVG_GAMMA && in.block_count_coding == BVG_GAMMA &&
	                      in.reference_coding == BVG_UNARY && in.residual_coding == BVG_ZETA;
	st->def = !defaults ? 0 : in.zeta_k == 3 ? 1 : (in.zeta_k >= 1 && in.zeta_k <= 16) ? 2 : 0;

	const bool traceOpen = bv_env("BVGPU_TRACE_OPEN") != nullptr;
	auto tOpen = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!traceOpen) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[bvgpu open] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tOpen).count());
		tOpen = now;
	};
	// the bit stream of a BVGraph goes from the file to HBM in pieces (stage_file_range) once the offsets say which part this handle stages; an EFGraph's words are
	// brought to host order first
	std::vector<uint8_t> graph, offs;
	struct FileCloser { FILE *f = nullptr; ~FileCloser() { if (f) fclose(f); } } gfile;
	size_t graphSize = 0;
	if (ef) { if (!bvh::read_file(st->basename + ".graph", graph, err)) return fail(g, BVG_EIO, err); }
	else {
		const std::string gp = st->basename + ".graph";
		gfile.f = fopen(gp.c_str(), "rb");
		if (!gfile.f) return fail(g, BVG_EIO, "cannot open " + gp + ": " + strerror(errno));
		if (fseeko(gfile.f, 0, SEEK_END) != 0 || ftello(gfile.f) < 0) return fail(g, BVG_EIO, "cannot size " + gp);
		graphSize = (size_t)ftello(gfile.f);
	}
	if (!bvh::read_file(st->basename + ".offsets", offs, err)) return fail(g, BVG_EIO, err);
	st->h_offsets.resize((size_t)in.nodes + 1);
	lap("files opened, .offsets read");
	if (ef) { // 64-bit words (EFGraph.loadLongBigList, EFGraph.java:677-707): padded to a whole word, brought to host order once
		graph.resize((graph.size() + 7) & ~(size_t)7, 0);
		if (in.ef_big_endian) for (size_t i = 0; i + 8 <= graph.size(); i += 8) { std::swap(graph[i], graph[i + 7]); std::swap(graph[i + 1], graph[i + 6]); std::swap(graph[i + 2], graph[i + 5]); std::swap(graph[i + 3], graph[i + 4]); }
		graphSize = graph.size();
	}
	st->info.graph_bytes = graphSize;

	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(g, BVG_EHIP, "no HIP device available (libbvgpu has no CPU fallback)");
	if (device < 0 || device >= ndev) return fail(g, BVG_EARG, "no such HIP device");
	st->device = device;
	st->info.device = device;
	HIPCHK(g, hipSetDevice(device));
	HIPCHK(g, hipMalloc((void **)&st->d_offsets_alloc, sizeof(int64_t) * st->h_offsets.size()));
	st->d_offsets = st->d_offsets_alloc;

	// offsets: gamma-coded gaps (the default) are decoded on the device (bv_offsets.hip) and copied back for shard
	// planning; delta-coded ones, BVGPU_OFFSETS=host, or a stream the device decoder rejects take the host decoder
	// (OffsetsLongIterator, BVG:907-935), which also produces the precise error
	bool onDevice = false;
	const char *offEnv = bv_env("BVGPU_OFFSETS");
	if ((in.offset_coding == BVG_GAMMA || in.offset_coding == BVG_DELTA) && !offs.empty() && !(offEnv && strcmp(offEnv, "host") == 0)) {
		const uint64_t ow = (offs.size() + 3) / 4;
		uint32_t *d_ow = nullptr;
		if (hipMalloc((void **)&d_ow, (size_t)(ow + 8) * 4) == hipSuccess) {
			const bool up = hipMemset(d_ow, 0, (size_t)(ow + 8) * 4) == hipSuccess && hipMemcpy(d_ow, offs.data(), offs.size(), hipMemcpyHostToDevice) == hipSuccess;
			lap("  .offsets to the device");
			const bool dec = up && bv::offsets_decode_device(d_ow, ow, (uint64_t)offs.size() * 8, in.nodes, st->d_offsets, nullptr, in.offset_coding == BVG_DELTA) == 0;
			lap(dec ? "  decoded on the device" : "  device decoder gave up");
			if (dec && hipMemcpy(st->h_offsets.data(), st->d_offsets, sizeof(int64_t) * st->h_offsets.size(), hipMemcpyDeviceToHost) == hipSuccess)
				onDevice = true;
			lap("  table to the host");
			(void)hipFree(d_ow);
		}
		(void)hipGetLastError();
	}
	if (!onDevice) {
		rc = bvh::decode_offsets(offs.data(), offs.size(), in.nodes, in.offset_coding, st->h_offsets.data());
		if (rc) return fail(g, rc, "cannot decode " + st->basename + ".offsets");
		HIPCHK(g, hipMemcpy(st->d_offsets, st->h_offsets.data(), sizeof(int64_t) * st->h_offsets.size(), hipMemcpyHostToDevice));
	}
	st->info.offsets_on_device = onDevice ? 1 : 0;
	lap("offsets decoded");
	if ((uint64_t)st->h_offsets.back() > (uint64_t)graphSize * 8) return fail(g, BVG_EIO, "offsets run past the end of the .graph file");
	for (size_t i = 1; i < st->h_offsets.size(); i++) if (st->h_offsets[i] < st->h_offsets[i - 1]) return fail(g, BVG_EIO, "offsets are not monotone");
	lap("offsets checked");
	// ---- what this handle stages: the whole graph, or one bits-balanced slice of it
	st->node_lo = 0; st->node_hi = in.nodes; st->stage_lo = 0;
	if (parts > 1) {
		const HostOffsets &off = st->h_offsets;
		auto bound = [&](int k) { return k >= parts ? in.nodes : (int32_t)(std::lower_bound(off.begin(), off.begin() + in.nodes, (int64_t)((__int128)off.back() * k / parts)) - off.begin()); };
		st->node_lo = part == 0 ? 0 : bound(part);
		st->node_hi = std::max(st->node_lo, bound(part + 1));
		// room for the referents of the slice's first rows: chains of any realistic depth (window x 64 levels, at least 4096 nodes)
		st->stage_lo = ef ? st->node_lo : (int32_t)std::max<int64_t>(0, (int64_t)st->node_lo - std::max<int64_t>(4096, (int64_t)in.window_size * 64)); // (no record of an EFGraph refers to another)
		// the offsets of [stage_lo, node_hi] only
		int64_t *slice = nullptr;
		const size_t cntOff = (size_t)(st->node_hi - st->stage_lo) + 1;
		HIPCHK(g, hipMalloc((void **)&slice, sizeof(int64_t) * cntOff));
		HIPCHK(g, hipMemcpy(slice, st->d_offsets_alloc + st->stage_lo, sizeof(int64_t) * cntOff, hipMemcpyDeviceToDevice));
		(void)hipFree(st->d_offsets_alloc);
		st->d_offsets_alloc = slice;
		st->d_offsets = (int64_t *)((uintptr_t)slice - sizeof(int64_t) * (size_t)st->stage_lo);
	}
	{ // the bit stream: words [word_lo, nwords) plus >= 8 zero words
		const uint64_t allWords = (graphSize + 3) / 4;
		const uint64_t wordLo = parts > 1 ? ((uint64_t)st->h_offsets[st->stage_lo] >> 5) & ~(uint64_t)3 : 0;
		st->nwords = parts > 1 ? std::min<uint64_t>(allWords, (((uint64_t)st->h_offsets[st->node_hi] + 31) >> 5)) : allWords;
		const size_t words = (size_t)(st->nwords - wordLo);
		HIPCHK(g, hipMalloc((void **)&st->d_bits_alloc, (words + 8) * 4));
		HIPCHK(g, hipMemset(st->d_bits_alloc, 0, (words + 8) * 4));
		const size_t byteLo = (size_t)wordLo * 4, byteHi = std::min<size_t>(graphSize, (size_t)st->nwords * 4);
		if (ef) { if (byteHi > byteLo) HIPCHK(g, hipMemcpy(st->d_bits_alloc, graph.data() + byteLo, byteHi - byteLo, hipMemcpyHostToDevice)); }
		else if (hipStreamSynchronize(nullptr) != hipSuccess /* the zero fill above is ahead of the pieces, which travel on a stream of their own */ ||
		         (rc = stage_file_range(gfile.f, st->basename + ".graph", byteLo, byteHi, (uint8_t *)st->d_bits_alloc, err)) != BVG_OK) return fail(g, rc ? rc : BVG_EHIP, err.empty() ? "staging the bit stream failed" : err);
		st->d_bits = (uint32_t *)((uintptr_t)st->d_bits_alloc - (size_t)wordLo * 4);
	}
	lap("bit stream staged");
	st->info.shard_from = st->node_lo; st->info.shard_to = st->node_hi; st->info.staged_from = st->stage_lo;
	// Scratch (interval arena, copy queues, giant list) is sized by the number of arcs: by what the stream holds, not by
	// what .properties claims -- one pass over the record headers at load time
	st->arcs_sizing = parts > 1 ? 1 : std::max<int64_t>(in.arcs, 1);
	if (ef && parts > 1) { // an EFGraph has no header pass to count with: the slice's share of the file's bits, with a margin (est_arcs takes shares of THIS; ADVICE r3: it was 1)
		const double all = (double)std::max<int64_t>(st->h_offsets.back() - st->h_offsets.front(), 1);
		const double mine = (double)(st->h_offsets[(size_t)st->node_hi] - st->h_offsets[(size_t)st->stage_lo]);
		st->arcs_sizing = std::max<int64_t>((int64_t)((double)std::max<int64_t>(in.arcs, 1) * mine / all * 1.1) + 4096, 1);
	}
	if (!ef && st->node_hi > st->stage_lo) {
		const int32_t n = st->node_hi - st->stage_lo;
		void *p_outd = nullptr, *p_ref = nullptr, *p_rs = nullptr, *p_sums = nullptr, *p_err = nullptr, *p_part = nullptr;
		const bool ok = hipMalloc(&p_outd, sizeof(int32_t) * (size_t)n) == hipSuccess && hipMalloc(&p_ref, sizeof(uint16_t) * (size_t)n) == hipSuccess &&
		                hipMalloc(&p_rs, sizeof(int64_t) * ((size_t)n + bv::SIZING_WORDS)) == hipSuccess /* (also the five counters of the sizing pass) */ && hipMalloc(&p_sums, sizeof(int64_t) * (size_t)bv::scan_num_sums(n)) == hipSuccess &&
		                hipMalloc(&p_err, sizeof(int)) == hipSuccess && hipMalloc(&p_part, sizeof(int32_t) * (bv::PICK_LEVELS * (size_t)bv::headers_blocks(n) + 8)) == hipSuccess;
		int64_t total = 0;
		hipError_t e = ok ? hipMemset(p_err, 0, sizeof(int)) : hipErrorOutOfMemory;
		if (e == hipSuccess) {
			const int64_t hb = bv::headers_blocks(n);
			bv::launch_headers(graph_dev0(*st), st->def, st->stage_lo, n, (int32_t *)p_outd, (uint16_t *)p_ref, (int *)p_err, nullptr, (int32_t *)p_part);
			bv::launch_scan((const int32_t *)p_outd, n, (int64_t *)p_rs, (int64_t *)p_sums, nullptr);
			bv::launch_pick_coop((const int32_t *)p_part, (int32_t)hb, 0, nullptr, nullptr, (int32_t *)p_part + bv::PICK_LEVELS * hb); // how long the records are: the lane class is decoded from tiles when few are long (enqueue_decode)
			e = hipMemcpy(&total, (int64_t *)p_rs + n, sizeof(int64_t), hipMemcpyDeviceToHost);
			if (e == hipSuccess) e = hipMemcpy(st->deg_counts, (int32_t *)p_part + bv::PICK_LEVELS * hb, sizeof(st->deg_counts), hipMemcpyDeviceToHost);
			if (e == hipSuccess && st->def != 0) { // (p_rs is done with: two counters)
				unsigned long long five[bv::SIZING_WORDS] = {};
				e = hipMemset(p_rs, 0, sizeof(five));
				if (e == hipSuccess) { bv::launch_seg_sizing(st->d_offsets, st->stage_lo, n, (const int32_t *)p_outd, (const uint16_t *)p_ref, (unsigned long long *)p_rs, nullptr); e = hipMemcpy(five, p_rs, sizeof(five), hipMemcpyDeviceToHost); }
				if (e == hipSuccess) { st->seg_long_records = (int64_t)five[0]; st->seg_long_bits = (int64_t)five[1]; st->max_outdegree = (int64_t)five[2]; st->lane_rows = (int64_t)five[3]; st->lane_ids = (int64_t)five[4];
					for (int k = 0; k < bv::SIZING_OCTAVES; k++) { st->oct_recs[k] = (int64_t)five[8 + 2 * k]; st->oct_arcs[k] = (int64_t)five[9 + 2 * k]; }
					if (bv_env("BVGPU_TRACE_HIST")) for (int k = 0; k < bv::SIZING_OCTAVES; k++) if (st->oct_recs[k]) fprintf(stderr, "[bvgpu] outdegree >= %d: %lld records, %lld arcs\n", 128 << k, (long long)st->oct_recs[k], (long long)st->oct_arcs[k]); }
			}
		}
		for (void *q : { p_outd, p_ref, p_rs, p_sums, p_err, p_part }) if (q) (void)hipFree(q);
		if (e != hipSuccess) return fail(g, e == hipErrorOutOfMemory ? BVG_ENOMEM : BVG_EHIP, "cannot scan the record headers");
		st->arcs_sizing = std::max<int64_t>(st->arcs_sizing, total);
	}
	lap("record headers counted");
	g->st = st;
	rc = init_handle(g);
	lap("handle set up");
	return rc;
}

extern "C" int bvg_clone(const bvg_t *src, bvg_t **out) {
	if (!src || !out || !src->st) return BVG_EARG;
	auto *g = new bvg_graph();
	*out = g;
	g->st = src->st;
	return init_handle(g);
}

// The handle's own copy of the graph, re-encoded for speed: 288 GB of HBM hold a representation 2.3 times the size of the BVGraph
// stream that scans 2.3 times and answers random batches 2.5 times as fast.  The lists are decoded once, encoded
// as an EFGraph in HBM (bv_efw.hip) and the handle switches to that image; clones made before keep what they had.
extern "C" int bvg_cache_as_efgraph(bvg_t *g) {
	if (!g || !g->st) return BVG_EARG;
	const std::shared_ptr<Staged> old = g->st;
	if (old->info.format == BVG_FORMAT_EF) return BVG_OK;
	if (old->node_lo != 0 || old->node_hi != old->info.nodes) return fail(g, BVG_EUNSUPPORTED, "a shard handle holds a slice of the graph: cache a whole-graph handle");
	HIPCHK(g, hipSetDevice(old->device));
	if (g->pend.active) { int rc = finish_pending(g, nullptr); if (rc) return rc; }
	const int32_t n = old->info.nodes;
	int64_t *d_rowptr = nullptr;
	int32_t *d_succ = nullptr;
	auto drop = [&]() { if (d_rowptr) (void)hipFree(d_rowptr); if (d_succ) (void)hipFree(d_succ); };
	uint64_t arcs = 0;
	HIPCHK(g, hipMalloc((void **)&d_rowptr, sizeof(int64_t) * ((size_t)n + 1)));
	int rc = decode_range_device(g, 0, n, d_rowptr, nullptr, 0, false, &arcs); // sizes the list buffer by what the stream holds
	if (rc == BVG_OK && hipMalloc((void **)&d_succ, sizeof(int32_t) * (size_t)std::max<uint64_t>(arcs, 1)) != hipSuccess) { (void)hipGetLastError(); rc = fail(g, BVG_ENOMEM, "device allocation failed"); }
	if (rc == BVG_OK) rc = decode_range_device(g, 0, n, d_rowptr, d_succ, (size_t)arcs, false, &arcs);
	if (rc) { drop(); return rc; }
	uint64_t *d_words = nullptr, nwords = 0, bits = 0;
	int32_t *d_reclen = nullptr;
	int64_t *d_off = nullptr;
	const int erc = bv::ef_encode_device(n, d_rowptr, d_succ, (uint64_t)n, 8, &d_words, &nwords, &bits, &d_reclen, &d_off, g->stream);
	drop();
	if (erc) { (void)hipGetLastError(); return fail(g, erc == -5 ? BVG_ENOMEM : erc == -3 ? BVG_EUNSUPPORTED : erc == -1 ? BVG_EFORMAT : BVG_EHIP, "re-encoding the lists failed"); }
	(void)hipFree(d_reclen);
	auto st = std::make_shared<Staged>();
	st->device = old->device;
	st->info = old->info;
	st->info.format = BVG_FORMAT_EF; st->info.ef_upper_bound = n; st->info.ef_log2_quantum = 8; st->info.ef_big_endian = 0; st->info.offset_coding = BVG_DELTA;
	st->info.arcs = (int64_t)arcs; st->info.graph_bytes = nwords * 8;
	st->d_bits_alloc = (uint32_t *)d_words; st->d_bits = st->d_bits_alloc; st->nwords = nwords * 2; // (ef_encode_device allocates two words more than it reports: reads past the end see zeros)
	st->d_offsets_alloc = d_off; st->d_offsets = d_off;
	st->node_lo = 0; st->node_hi = n; st->stage_lo = 0;
	st->h_offsets.resize((size_t)n + 1);
	if (hipMemcpy(st->h_offsets.data(), d_off, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyDeviceToHost) != hipSuccess) return fail(g, BVG_EHIP, "copying the offsets back failed"); // (st frees the buffers)
	st->arcs_sizing = std::max<int64_t>((int64_t)arcs, 1);
	st->basename = old->basename;
	g->st = st;
	return BVG_OK;
}

extern "C" int bvg_close(bvg_t *g) {
	if (!g) return BVG_OK;
	if (g->st && g->st->device >= 0) {
		(void)hipSetDevice(g->st->device);
		if (g->own) { (void)hipStreamSynchronize(g->own); (void)hipStreamDestroy(g->own); }
		for (DevBuf *b : { &g->outd, &g->ref, &g->rowstart, &g->depth, &g->sums, &g->need, &g->halo, &g->hashA, &g->hashB, &g->hashBounds, &g->pickpart, &g->walktab, &g->stage_rowptr, &g->stage_succ, &g->stage_nodes, &g->small, &g->b_chainlen, &g->b_slotbase, &g->b_node, &g->b_qidx, &g->b_aoutd, &g->b_qoutd, &g->biglist, &g->giantlist, &g->arena, &g->coopctl, &g->stats, &g->key16, &g->keys, &g->lvlist, &g->plist, &g->pkeys, &g->pkey16, &g->copyq, &g->walkdesc, &g->copytab, &g->bigtmp, &g->tilebounds, &g->segbuf, &g->hashmark, &g->hashctx, &g->hashtab, &g->hashq }) b->release();
		for (DevBuf *b : { &g->hchunk[0], &g->hchunk[1], &g->statsbuf, &g->bfs_rowptr, &g->bfs_succ, &g->bfs_ctr }) b->release();
		for (PinBuf *b : { &g->hring[0], &g->hring[1], &g->view_rowptr, &g->view_succ }) b->release();
		for (hipEvent_t e : { g->evChunk[0], g->evChunk[1], g->evCopied[0], g->evCopied[1] }) if (e) (void)hipEventDestroy(e);
		if (g->h_small) (void)hipHostFree(g->h_small);
		for (auto &e : g->ev) if (e) (void)hipEventDestroy(e);
		for (hipEvent_t e : { g->evFork, g->evA, g->evB, g->evC, g->evHdr, g->evHdr0, g->evP, g->evM, g->evH, g->evL, g->evIn, g->evOut }) if (e) (void)hipEventDestroy(e);
		for (hipStream_t st : { g->sideA, g->sideB, g->sideC }) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
	}
	delete g;
	return BVG_OK;
}

extern "C" int bvg_info(const bvg_t *g, bvg_info_t *out) {
	if (!g || !out || !g->st) return BVG_EARG;
	*out = g->st->info;
	return BVG_OK;
}

extern "C" const char *bvg_last_error(const bvg_t *g) { return g ? g->err.c_str() : "null handle"; }

extern "C" int bvg_set_stream(bvg_t *g, void *hip_stream) {
	if (!g || !g->st) return BVG_EARG;
	if (g->pend.active) { int rc = finish_pending(g, nullptr); if (rc) return rc; }
	g->user = (hipStream_t)hip_stream;
	return BVG_OK;
}

extern "C" int bvg_set_option(bvg_t *g, const char *name, const char *value) {
	if (!g || !name) return BVG_EARG;
	if (g->pend.active) { int rc = finish_pending(g, nullptr); if (rc) return rc; }
	std::string n(name);
	for (char &c : n) c = (char)tolower((unsigned char)c);
	if (n.rfind("bvgpu_", 0) == 0) n = n.substr(6);
	const int rc = apply_option(g, n, value);
	return rc ? fail(g, rc, "unknown option: " + n) : BVG_OK;
}

extern "C" int bvg_set_profile(bvg_t *g, int enable) {
	if (!g || !g->st) return BVG_EARG;
	HIPCHK(g, hipSetDevice(g->st->device));
	if (enable && !g->ev[0]) for (auto &e : g->ev) HIPCHK(g, hipEventCreate(&e));
	g->profile = enable != 0;
	g->ev_valid = false;
	return BVG_OK;
}

extern "C" int bvg_last_thresholds(bvg_t *g, int32_t *coop_min, int32_t *giant_min) {
	if (!g || !g->st) return BVG_EARG;
	if (g->pend.active) { int rc = finish_pending(g, nullptr); if (rc) return rc; }
	if (coop_min) *coop_min = g->last_coop_min;
	if (giant_min) *giant_min = g->last_giant_min;
	return BVG_OK;
}

extern "C" int bvg_get_profile(bvg_t *g, float *ms) {
	if (!g || !g->st || !ms) return BVG_EARG;
	if (!g->ev_valid) return fail(g, BVG_ESTATE, "no profiled range decode has completed");
	if (g->pend.active) { int rc = finish_pending(g, nullptr); if (rc) return rc; }
	HIPCHK(g, hipEventSynchronize(g->ev[BVG_NUM_PHASES]));
	for (int i = 0; i < BVG_NUM_PHASES; i++) HIPCHK(g, hipEventElapsedTime(&ms[i], g->ev[i], g->ev[i + 1]));
	return BVG_OK;
}

extern "C" int bvg_debug_stats(bvg_t *g, uint64_t *out8, int reset) {
	if (!g || !g->st || !out8) return BVG_EARG;
	if (!g->stats.p) return fail(g, BVG_ESTATE, "set BVGPU_STATS=1 before opening the graph");
	HIPCHK(g, hipSetDevice(g->st->device));
	HIPCHK(g, hipDeviceSynchronize());
	HIPCHK(g, hipMemcpy(out8, g->stats.p, 512, hipMemcpyDeviceToHost));
	if (reset) HIPCHK(g, hipMemset(g->stats.p, 0, 512));
	return BVG_OK;
}

extern "C" int bvg_sync(bvg_t *g, uint64_t *arcs_out) {
	if (!g || !g->st) return BVG_EARG;
	HIPCHK(g, hipSetDevice(g->st->device));
	if (g->pend.active) return finish_pending(g, arcs_out);
	HIPCHK(g, hipStreamSynchronize(g->stream));
	if (arcs_out) *arcs_out = g->last_arcs;
	return BVG_OK;
}

extern "C" int bvg_outdegrees(bvg_t *g, int32_t from, int32_t to, int32_t *out, int flags) {
	if (!g || !g->st) return BVG_EARG;
	const Staged &s = *g->st;
	if (from < s.stage_lo || to > s.node_hi || from > to || (!out && to > from)) return fail(g, BVG_EARG, "node range out of bounds"); // BVG:860
	if (to == from) return BVG_OK;
	HIPCHK(g, hipSetDevice(s.device));
	if (g->pend.active) { int rc = finish_pending(g, nullptr); if (rc) return rc; }
	{ int rc = fork_from_user(g); if (rc) return rc; }
	const int32_t cnt = to - from;
	if (s.info.format == BVG_FORMAT_EF) {
		if (!g->outd.need(sizeof(int32_t) * (size_t)cnt)) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
		HIPCHK(g, hipMemsetAsync(g->small.p, 0, sizeof(Small), g->stream));
		bv::launch_ef_outdeg(ef_dev(s), nullptr, from, cnt, g->outd.as<int32_t>(), &g->small.as<Small>()->err, g->stream);
		HIPCHK(g, hipMemcpyAsync(out, g->outd.p, sizeof(int32_t) * (size_t)cnt, (flags & BVG_OUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, g->stream));
		int rc = fetch_small(g);
		if (rc) return rc;
		if (g->h_small->err) return fail(g, BVG_EFORMAT, "malformed bit stream");
		return BVG_OK;
	}
	if (!g->outd.need(sizeof(int32_t) * (size_t)cnt) || !g->ref.need(sizeof(uint16_t) * (size_t)cnt)) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	HIPCHK(g, hipMemsetAsync(g->small.p, 0, sizeof(Small), g->stream));
	bv::launch_headers(graph_dev(s), s.def, from, cnt, g->outd.as<int32_t>(), g->ref.as<uint16_t>(), &g->small.as<Small>()->err, g->stream);
	HIPCHK(g, hipMemcpyAsync(out, g->outd.p, sizeof(int32_t) * (size_t)cnt, (flags & BVG_OUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, g->stream));
	int rc = fetch_small(g);
	if (rc) return rc;
	if (g->h_small->err & ~bv::E_REF) return fail(g, dev_err_to_status(g->h_small->err & ~bv::E_REF), "malformed bit stream");
	return BVG_OK;
}

namespace {

// Is p pinned (or otherwise known to the HIP runtime as host memory)?  Then a device-to-host copy goes straight there at
// PCIe speed; pageable memory is filled by host threads from a pinned ring instead.
bool is_pinned_host(const void *p) {
	hipPointerAttribute_t a{};
	if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
	return a.type == hipMemoryTypeHost;
}

void parallel_memcpy(void *dst, const void *src, size_t bytes) {
	constexpr size_t PIECE = (size_t)8 << 20;
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	const size_t want = std::min<size_t>({ (bytes + PIECE - 1) / PIECE, (size_t)hw, (size_t)16 });
	if (want <= 1) { memcpy(dst, src, bytes); return; }
	std::vector<std::thread> th;
	const size_t per = ((bytes + want - 1) / want + 63) & ~(size_t)63;
	for (size_t k = 0; k < want; k++) {
		const size_t o = std::min(bytes, k * per), e = std::min(bytes, o + per);
		if (e > o) th.emplace_back([=] { memcpy((char *)dst + o, (const char *)src + o, e - o); });
	}
	for (auto &t : th) t.join();
}

// Arcs of nodes [a, e), estimated from their share of the bits this handle stages: arcs_sizing counts the staged records only (a
// bvg_open_shard handle stages one slice), so the share is taken of the staged span, not of the whole file (buffers of a
// shard's scans were under-sized by the number of shards and every first scan was decoded twice).
double est_arcs(const Staged &s, int32_t a, int32_t e) {
	const int64_t staged = std::max<int64_t>(s.h_offsets[(size_t)s.node_hi] - s.h_offsets[(size_t)s.stage_lo], 1);
	return (double)std::max<int64_t>(s.arcs_sizing, 1) * (double)(s.h_offsets[(size_t)e] - s.h_offsets[(size_t)a]) / (double)staged;
}

// Cuts [from, to) into pieces of roughly `target` arcs each, by the share of the bit stream they hold (host offsets).
std::vector<int32_t> plan_chunks_by_bits(const Staged &s, int32_t from, int32_t to, int64_t target) {
	const int64_t bits = s.h_offsets[to] - s.h_offsets[from];
	const double estArcs = est_arcs(s, from, to);
	const int64_t parts = std::max<int64_t>(1, (int64_t)(estArcs / (double)std::max<int64_t>(target, 1) + 0.999));
	std::vector<int32_t> b{ from };
	for (int64_t k = 1; k < parts; k++) {
		const int64_t t = s.h_offsets[from] + (int64_t)((__int128)bits * k / parts);
		int32_t x = (int32_t)(std::lower_bound(s.h_offsets.begin() + from, s.h_offsets.begin() + to, t) - s.h_offsets.begin());
		x = std::min(std::max(x, b.back()), to);
		if (x > b.back()) b.push_back(x);
	}
	if (to > b.back() || b.size() == 1) b.push_back(to);
	return b;
}

// Arcs per piece of the scans that keep their rows on the device (checksum, statistics).  BVGPU_SCAN_PIECE: tests only.
// arcs per piece of the scans whose rows stay in the library (checksum, statistics, equality, HyperBall): a piece is one job, and a job four times as long hides more of its set-up and
// tails than four jobs do -- 1 G arcs (4.4 GB of scratch rows) where the device has the memory: the checksum of a 1 G-arc graph 16.3 -> 12.5 ms, below the 12.8 ms of the scan that hands the
// rows to the caller; 256 M arcs (1 GB) on a device of less than 96 GB
int64_t scan_piece_arcs(const bvg_graph *g) {
	if (g->scan_piece > 0) return (int64_t)g->scan_piece;
	// (by the memory of the handle's OWN device, asked once per device: a process may hold handles on devices of different sizes -- ADVICE r5)
	static std::atomic<int> roomy[64]; // 0 unknown, 1 small, 2 roomy
	const int dev = g->st ? g->st->device : 0, slot = dev >= 0 && dev < 64 ? dev : 0;
	int r = roomy[slot].load(std::memory_order_relaxed);
	if (r == 0) {
		hipDeviceProp_t prop;
		r = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.totalGlobalMem >= ((size_t)96 << 30) ? 2 : 1;
		roomy[slot].store(r, std::memory_order_relaxed);
	}
	return r == 2 ? (int64_t)1 << 30 : (int64_t)256 << 20;
}

// Host-output scan in ONE pass: the structure (outdegrees, CSR row starts) of the whole range first -- that is the
// rowptr the caller gets and the exact chunk plan --, then the successors chunk by chunk: chunk k leaves over PCIe on the
// copy stream while chunk k+1 is being decoded.  rowptr_h: to-from+1 entries (host); succ_h may be NULL (count only).
int host_scan(bvg_graph *g, int32_t from, int32_t to, int64_t *rowptr_h, int32_t *succ_h, size_t succ_cap, uint64_t *arcs_out) {
	const Staged &s = *g->st;
	HIPCHK(g, hipSetDevice(s.device));
	const size_t nrow = (size_t)(to - from) + 1;
	const bool trace = g->trace_host;
	const auto t0 = std::chrono::steady_clock::now();
	auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
	if (!g->stage_rowptr.need(sizeof(int64_t) * nrow)) return fail(g, BVG_ENOMEM, "staging allocation failed");
	uint64_t arcs = 0;
	int rc = decode_range_device(g, from, to, g->stage_rowptr.as<int64_t>(), nullptr, 0, false, &arcs);
	if (rc) return rc;
	if (arcs_out) *arcs_out = arcs;
	if (trace) fprintf(stderr, "[host_scan] %.2f ms structure pass done\n", ms());
	HIPCHK(g, hipMemcpy(rowptr_h, g->stage_rowptr.p, sizeof(int64_t) * nrow, hipMemcpyDeviceToHost));
	if (trace) fprintf(stderr, "[host_scan] %.2f ms rowptr on the host\n", ms());
	if (!succ_h || arcs == 0) return BVG_OK;
	if (arcs > succ_cap) return fail(g, BVG_ECAP, "successor buffer too small");
	// chunks of ~1/8 of the range, between 4 M and 32 M arcs: cut where the (exact) row starts say
	const uint64_t target = std::min<uint64_t>(std::max<uint64_t>(arcs / 8, (uint64_t)4 << 20), (uint64_t)32 << 20);
	std::vector<int32_t> cut{ from };
	while (cut.back() < to) {
		const int64_t base = rowptr_h[cut.back() - from];
		const int64_t *lo = rowptr_h + (cut.back() - from) + 1, *hi = rowptr_h + nrow;
		int32_t nx = from + (int32_t)(std::upper_bound(lo, hi, base + (int64_t)target) - rowptr_h) - 1; // last node whose row still ends inside the target
		nx = std::max(nx, cut.back() + 1);                                                             // (a single row longer than the target is a chunk of its own)
		cut.push_back(std::min(nx, to));
	}
	uint64_t maxChunk = 0;
	for (size_t k = 0; k + 1 < cut.size(); k++) maxChunk = std::max<uint64_t>(maxChunk, (uint64_t)(rowptr_h[cut[k + 1] - from] - rowptr_h[cut[k] - from]));
	const bool pinned = is_pinned_host(succ_h);
	for (int b = 0; b < 2; b++) {
		if (!g->hchunk[b].need(sizeof(int32_t) * (size_t)std::max<uint64_t>(maxChunk, 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
		if (!pinned && !g->hring[b].need(sizeof(int32_t) * (size_t)std::max<uint64_t>(maxChunk, 1))) return fail(g, BVG_ENOMEM, "pinned staging allocation failed");
	}
	int32_t maxNodes = 0;
	for (size_t k = 0; k + 1 < cut.size(); k++) maxNodes = std::max(maxNodes, cut[k + 1] - cut[k]);
	if (!g->stage_rowptr.need(sizeof(int64_t) * ((size_t)maxNodes + 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
	struct Out { int32_t *dst; const int32_t *src; size_t bytes; bool live = false; } out[2];
	struct HostMode { // the copies own sideB until the last one has landed (also on the error paths)
		bvg_graph *g;
		explicit HostMode(bvg_graph *g_) : g(g_) { g->host_mode = true; }
		~HostMode() { (void)hipStreamSynchronize(g->sideB); g->host_mode = false; }
	} hostMode(g);
	for (size_t k = 0; k + 1 < cut.size(); k++) {
		const int b = (int)(k & 1);
		const int32_t a = cut[k], e = cut[k + 1];
		const uint64_t want = (uint64_t)(rowptr_h[e - from] - rowptr_h[a - from]);
		if (out[b].live) HIPCHK(g, hipEventSynchronize(g->evCopied[b])); // chunk k-2 has left device buffer b
		if (trace) fprintf(stderr, "[host_scan] %.2f ms chunk %zu: buffer free, decode starts (%llu arcs)\n", ms(), k, (unsigned long long)want);
		rc = decode_range_device(g, a, e, g->stage_rowptr.as<int64_t>(), g->hchunk[b].as<int32_t>(), (size_t)std::max<uint64_t>(want, 1), true, nullptr);
		if (rc) return rc;
		// while the GPU decodes: chunk k-2 goes from the ring to a pageable destination (host threads)
		if (out[b].live && !pinned) parallel_memcpy(out[b].dst, out[b].src, out[b].bytes);
		out[b].live = false;
		uint64_t got = 0;
		rc = finish_pending(g, &got);
		if (rc) return rc;
		if (got != want) return fail(g, BVG_EFORMAT, "a chunk decoded to another number of arcs than the scan counted");
		if (trace) fprintf(stderr, "[host_scan] %.2f ms chunk %zu decoded\n", ms(), k);
		if (want == 0) continue;
		HIPCHK(g, hipEventRecord(g->evChunk[b], g->stream));
		HIPCHK(g, hipStreamWaitEvent(g->copyStream, g->evChunk[b], 0));
		int32_t *dst = succ_h + (rowptr_h[a - from] - rowptr_h[0]);
		void *land = pinned ? (void *)dst : g->hring[b].p;
		HIPCHK(g, hipMemcpyAsync(land, g->hchunk[b].p, sizeof(int32_t) * (size_t)want, hipMemcpyDeviceToHost, g->copyStream)); // overlaps the next chunk's decode
		HIPCHK(g, hipEventRecord(g->evCopied[b], g->copyStream));
		out[b] = Out{ dst, (const int32_t *)g->hring[b].p, sizeof(int32_t) * (size_t)want, true };
	}
	for (size_t k = cut.size() - 1, i = 0; i < 2; i++, k++) { // the last two chunks, in order
		const int b = (int)(k & 1);
		if (!out[b].live) continue;
		HIPCHK(g, hipEventSynchronize(g->evCopied[b]));
		if (!pinned) parallel_memcpy(out[b].dst, out[b].src, out[b].bytes);
		out[b].live = false;
		if (trace) fprintf(stderr, "[host_scan] %.2f ms chunk %zu on the host\n", ms(), k);
	}
	g->last_arcs = arcs;
	return BVG_OK;
}

} // namespace

extern "C" int bvg_decode_range(bvg_t *g, int32_t from, int32_t to, int64_t *rowptr, int32_t *succ, size_t succ_cap, uint64_t *arcs_out, int flags) {
	if (!g || !g->st) return BVG_EARG;
	const Staged &s = *g->st;
	if (from < 0 || from > s.info.nodes || to < from || to > s.info.nodes || !rowptr) return fail(g, BVG_EARG, "node range out of bounds"); // BVG:1165
	if (flags & BVG_OUT_DEVICE) return decode_range_device(g, from, to, rowptr, succ, succ_cap, (flags & BVG_ASYNC) != 0, arcs_out);
	if (s.info.format == BVG_FORMAT_EF) {
		if (from < s.node_lo || to > s.node_hi) return fail(g, BVG_EARG, "node range outside the slice this handle stages (bvg_open_shard)");
		return ef_to_host(g, nullptr, from, (int64_t)to - from, rowptr, succ, succ_cap, arcs_out);
	}
	return host_scan(g, from, to, rowptr, succ, succ_cap, arcs_out);
}

extern "C" int bvg_decode_range_view(bvg_t *g, int32_t from, int32_t to, const int64_t **rowptr_out, const int32_t **succ_out, uint64_t *arcs_out) {
	if (!g || !g->st) return BVG_EARG;
	const Staged &s = *g->st;
	if (from < 0 || from > s.info.nodes || to < from || to > s.info.nodes || !rowptr_out || !succ_out) return fail(g, BVG_EARG, "node range out of bounds"); // BVG:1165
	HIPCHK(g, hipSetDevice(s.device));
	const size_t nrow = (size_t)(to - from) + 1;
	if (!g->view_rowptr.need(sizeof(int64_t) * nrow)) return fail(g, BVG_ENOMEM, "pinned result allocation failed");
	// the successor buffer is sized by the range's share of the bit stream first (no counting pass of its own)
	const uint64_t guess = (uint64_t)(est_arcs(s, from, to) * 1.05) + 1024;
	if (!g->view_succ.need(sizeof(int32_t) * (size_t)guess)) return fail(g, BVG_ENOMEM, "pinned result allocation failed");
	uint64_t arcs = 0;
	auto scan = [&]() { return s.info.format == BVG_FORMAT_EF ? ef_to_host(g, nullptr, from, (int64_t)to - from, g->view_rowptr.as<int64_t>(), g->view_succ.as<int32_t>(), g->view_succ.cap / sizeof(int32_t), &arcs)
	                                                        : host_scan(g, from, to, g->view_rowptr.as<int64_t>(), g->view_succ.as<int32_t>(), g->view_succ.cap / sizeof(int32_t), &arcs); };
	int rc = scan();
	if (rc == BVG_ECAP) { // more arcs than the share of the stream suggested: now the count is known
		if (!g->view_succ.need(sizeof(int32_t) * (size_t)std::max<uint64_t>(arcs, 1))) return fail(g, BVG_ENOMEM, "pinned result allocation failed");
		rc = scan();
	}
	if (rc) return rc;
	*rowptr_out = g->view_rowptr.as<int64_t>();
	*succ_out = g->view_succ.as<int32_t>();
	if (arcs_out) *arcs_out = arcs;
	return BVG_OK;
}

extern "C" int bvg_host_alloc(size_t bytes, void **out) {
	if (!out) return BVG_EARG;
	*out = nullptr;
	if (hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return BVG_ENOMEM; }
	return BVG_OK;
}
extern "C" void bvg_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int bvg_scan_checksum(bvg_t *g, int32_t from, int32_t to, int32_t *hash_io, uint64_t *arcs_out) {
	if (!g || !g->st || !hash_io) return BVG_EARG;
	const Staged &s = *g->st;
	if (from < 0 || from > s.info.nodes || to < from || to > s.info.nodes) return fail(g, BVG_EARG, "node range out of bounds"); // BVG:1165
	HIPCHK(g, hipSetDevice(s.device));
	if (s.info.format == BVG_FORMAT_EF && !g->ef_hash_materialise) return ef_scan_checksum(g, from, to, hash_io, arcs_out); // (the knob: the decode-then-fold path below, for comparison)
	// The rows never reach the caller: they are decoded piece by piece into one scratch buffer and folded into the running
	// hash there.  Pieces of <= 1 G arcs (scan_piece_arcs): smaller ones that would stay in the Infinity Cache (32 M arcs)
	// cost more in per-call set-up than they save (C2: 12.2 ms in 7 pieces, 4 ms in one).
	const std::vector<int32_t> cut = plan_chunks_by_bits(s, from, to, scan_piece_arcs(g));
	uint64_t total = 0;
	int32_t h = *hash_io;
	// Since round 5 the fold is part of the scan (bv::HashCtx): the one-lane parse adds the rows without a reference to the sum as it decodes them and writes only
	// those that a row of the piece copies from; k_hash_rest adds the node numbers and the rows that are in memory anyway.  BVGPU_HASH_MATERIALISE=1: decode every
	// row, then fold from memory (round 2's path, kept for comparison and for the tests).
	const bool materialise = g->hash_materialise || s.info.format == BVG_FORMAT_EF; // (an EFGraph gets here only through BVGPU_EF_HASH_MATERIALISE: the fold of bv_ef.hip is the other path)
	for (size_t k = 0; k + 1 < cut.size(); k++) {
		const int32_t a = cut[k], e = cut[k + 1];
		if (e == a) continue;
		if (!g->stage_rowptr.need(sizeof(int64_t) * ((size_t)(e - a) + 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
		const uint64_t guess = (uint64_t)(est_arcs(s, a, e) * 1.1) + 4096;
		if (!g->stage_succ.need(sizeof(int32_t) * (size_t)guess)) return fail(g, BVG_ENOMEM, "staging allocation failed");
		uint64_t arcs = 0;
		g->hash_job = !materialise;
		g->hash_stale = false;
		int rc = decode_range_device(g, a, e, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), g->stage_succ.cap / sizeof(int32_t), false, &arcs);
		if (rc == BVG_ECAP) {
			if (!g->stage_succ.need(sizeof(int32_t) * (size_t)std::max<uint64_t>(arcs, 1))) { g->hash_job = false; return fail(g, BVG_ENOMEM, "staging allocation failed"); }
			g->hash_stale = false;
			rc = decode_range_device(g, a, e, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), g->stage_succ.cap / sizeof(int32_t), false, &arcs);
		}
		const bool folded = g->hash_job && !g->hash_stale;
		if (!rc && g->hash_job && g->hash_stale) { // reference chains deeper than the levels launched ahead (a file that understates maxrefcount): every row is final now -- fold from memory
			g->hash_job = false;
			rc = decode_range_device(g, a, e, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), g->stage_succ.cap / sizeof(int32_t), false, &arcs);
		}
		g->hash_job = false;
		if (rc) return rc;
		if (folded) { // the piece's map h -> 31^L h + 31^(L + c) sum, c = rowstart[nh] + nh (HashCtx); the sum and rowstart[nh] came home with the job's status
			const uint64_t L = (uint64_t)(e - a) + arcs, c = (uint64_t)g->h_small->halo_total + (uint64_t)g->hash_view.nh;
			h = (int32_t)(host_pow31(L) * (uint32_t)h + host_pow31(L + c) * (uint32_t)g->h_small->hash);
		}
		else rc = bvg_csr_hashcode(g, a, e, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), &h);
		if (rc) return rc;
		total += arcs;
	}
	*hash_io = h;
	if (arcs_out) *arcs_out = total;
	g->last_arcs = total;
	return BVG_OK;
}

extern "C" int bvg_successors_batch(bvg_t *g, const int32_t *nodes, size_t q, int64_t *rowptr, int32_t *succ, size_t succ_cap, uint64_t *arcs_out, int flags) {
	if (!g || !g->st) return BVG_EARG;
	if (!rowptr || (!nodes && q)) return fail(g, BVG_EARG, "null argument");
	const Staged &s = *g->st;
	if (s.node_lo != 0 || s.node_hi != s.info.nodes) return fail(g, BVG_EUNSUPPORTED, "random access needs the whole graph: open it with bvg_open");
	HIPCHK(g, hipSetDevice(s.device));
	if (g->pend.active) { int rc = finish_pending(g, nullptr); if (rc) return rc; }
	g->ctl_clean = false; // (see decode_range_device)
	{ int rc = fork_from_user(g); if (rc) return rc; }
	const bool dev = (flags & BVG_OUT_DEVICE) != 0;
	if (s.info.format == BVG_FORMAT_EF) { // every list is decoded from its own record: a batch is a range with an indirection
		if (q > 0x7fffffffu) return fail(g, BVG_EARG, "too many queries");
		if (!dev) return ef_to_host(g, nodes, 0, (int64_t)q, rowptr, succ, succ_cap, arcs_out);
		int rc = ef_job(g, nodes, 0, (int64_t)q, rowptr, succ, succ_cap, arcs_out);
		if (rc) return rc;
		return join_to_user(g);
	}
	const bv::GraphDev gd = graph_dev(s);
	Small *dsm = g->small.as<Small>();
	HIPCHK(g, hipMemsetAsync(g->small.p, 0, sizeof(Small), g->stream));
	// inputs / outputs on the device
	const int32_t *d_nodes = nodes;
	int64_t *d_rowptr = rowptr;
	if (!dev) {
		if (!g->stage_nodes.need(sizeof(int32_t) * std::max<size_t>(q, 1)) || !g->stage_rowptr.need(sizeof(int64_t) * (q + 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
		if (q) HIPCHK(g, hipMemcpyAsync(g->stage_nodes.p, nodes, sizeof(int32_t) * q, hipMemcpyHostToDevice, g->stream));
		d_nodes = g->stage_nodes.as<int32_t>();
		d_rowptr = g->stage_rowptr.as<int64_t>();
	}
	if (q == 0) {
		HIPCHK(g, hipMemsetAsync(d_rowptr, 0, sizeof(int64_t), g->stream));
		if (!dev) HIPCHK(g, hipMemcpyAsync(rowptr, d_rowptr, sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
		HIPCHK(g, hipStreamSynchronize(g->stream));
		if (arcs_out) *arcs_out = 0;
		return BVG_OK;
	}
	auto run_dense = [&]() -> int {
		// Dense batch: a masked scan of the whole graph (every needed record is decoded once, however many queries or
		// reference chains want it), then a gather of the rows into the caller's order.
		const int32_t n = s.info.nodes;
		if (!g->outd.need(sizeof(int32_t) * (size_t)n) || !g->ref.need(sizeof(uint16_t) * (size_t)n) || !g->rowstart.need(sizeof(int64_t) * ((size_t)n + 1)) ||
		    !g->sums.need(sizeof(int64_t) * (size_t)bv::scan_num_sums((int64_t)std::max<size_t>((size_t)n, q))) || !g->b_qoutd.need(sizeof(int32_t) * q) || (succ && !g->need.need((size_t)n)))
			return fail(g, BVG_ENOMEM, "device scratch allocation failed");
		bv::RangeView v{};
		v.halo_cap = ~0ull;
		v.lo = 0; v.cnt = n; v.nh = n; // every row lives in the arena ("halo" rows of the scan)
		v.outd = g->outd.as<int32_t>(); v.ref = g->ref.as<uint16_t>(); v.rowstart = g->rowstart.as<int64_t>();
		g->keys_ready = false; // (no parse-list keys from these headers: the marks change the outdegrees behind them)
		bv::launch_headers(gd, s.def, 0, n, v.outd, v.ref, &dsm->err, g->stream);
		// queries mark their nodes; the marks are closed under "is copied from" in streaming passes (as many as chains
		// were deep last time, plus one that reports whether it still found something)
		const bool streamed = bv::launch_query_mark(d_nodes, (int64_t)q, n, v.outd, v.ref, succ ? g->need.as<uint8_t>() : nullptr, g->b_qoutd.as<int32_t>(), g->levels_hint, &dsm->pad, &dsm->err, g->stream);
		if (succ) {
			for (int round = 0; streamed; round++) { // (the reporting pass found nothing in the common case: one round trip)
				int rc0 = fetch_small(g);
				if (rc0) return rc0;
				if (!g->h_small->pad) break;
				HIPCHK(g, hipMemsetAsync(&dsm->pad, 0, sizeof(int32_t), g->stream));
				if (round == 6) { // chains of dozens of levels (no maxrefcount, or a corrupt file): a pass per level would never end -- walk them
					bv::launch_query_walk(d_nodes, (int64_t)q, n, v.outd, v.ref, g->need.as<uint8_t>(), g->b_qoutd.as<int32_t>(), &dsm->err, g->stream);
					break;
				}
				bv::launch_need_prop(n, v.outd, v.ref, g->need.as<uint8_t>(), 4, &dsm->pad, g->stream);
			}
			bv::launch_apply_need(n, g->need.as<uint8_t>(), v.outd, v.ref, g->stream);
		}
		bv::launch_scan(g->b_qoutd.as<int32_t>(), (int64_t)q, d_rowptr, g->sums.as<int64_t>(), g->stream);
		HIPCHK(g, hipMemcpyAsync(&dsm->total, d_rowptr + q, sizeof(int64_t), hipMemcpyDeviceToDevice, g->stream));
		if (succ) {
			bv::launch_scan(v.outd, n, v.rowstart, g->sums.as<int64_t>(), g->stream);
			HIPCHK(g, hipMemcpyAsync(&dsm->halo_total, v.rowstart + n, sizeof(int64_t), hipMemcpyDeviceToDevice, g->stream));
		}
		int rc = fetch_small(g);
		if (rc) return rc;
		if (g->h_small->err) {
			const int e = g->h_small->err;
			if (e & bv::E_ARG) return fail(g, BVG_EARG, "Node index out of range");
			return fail(g, dev_err_to_status(e), "malformed bit stream");
		}
		const uint64_t arcs = (uint64_t)g->h_small->total;
		g->last_arcs = arcs;
		if (arcs_out) *arcs_out = arcs;
		if (!dev) HIPCHK(g, hipMemcpyAsync(rowptr, d_rowptr, sizeof(int64_t) * (q + 1), hipMemcpyDeviceToHost, g->stream));
		if (!succ) { HIPCHK(g, hipStreamSynchronize(g->stream)); return BVG_OK; }
		if (arcs > succ_cap) { HIPCHK(g, hipStreamSynchronize(g->stream)); return fail(g, BVG_ECAP, "successor buffer too small"); }
		int32_t *d_succ = succ;
		if (!dev) {
			if (!g->stage_succ.need(sizeof(int32_t) * (size_t)std::max<uint64_t>(arcs, 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
			d_succ = g->stage_succ.as<int32_t>();
		}
		if (!g->halo.need(sizeof(int32_t) * (size_t)std::max<int64_t>(g->h_small->halo_total, 1))) return fail(g, BVG_ENOMEM, "arena allocation failed");
		v.succ = nullptr; v.halo = g->halo.as<int32_t>(); v.succ_cap = 0;
		v.halo_cap = g->halo.cap / sizeof(int32_t); // (the arena's real size: the copy pass's 16-byte windows may read up to three ids past a row, never past the buffer)
		int32_t levels = 0, giantCap = 0;
		// (the headers' event: outdegrees and references have been final since the round trip above -- with it the parse list is built and the long records are classified
		// side by side BEFORE the cooperative kernels start, as in a range job; without it the list's three kernels ran beside the giants and the wave class, starved: 0.8 of C4's 4.3 ms)
		HIPCHK(g, hipEventRecord(g->evHdr, g->stream));
		rc = enqueue_decode(g, v, g->h_small->halo_total, levels, giantCap, true);
		if (rc) return rc;
		HIPCHK(g, hipGetLastError());
		g->pend.active = true; g->pend.view = v; g->pend.levels_done = levels; g->pend.want_succ = true; g->pend.giantCap = giantCap;
		g->pend.optimistic = false; // (nothing to repeat: the arena was sized by a round trip)
		rc = finish_pending(g, nullptr); // the levels of the copy pass still missing, errors
		if (rc) return rc;
		bv::launch_gather_rows(d_nodes, (int64_t)q, (int64_t)arcs, v.rowstart, v.halo, d_rowptr, d_succ, g->stream);
		if (!dev && arcs) HIPCHK(g, hipMemcpyAsync(succ, d_succ, sizeof(int32_t) * (size_t)arcs, hipMemcpyDeviceToHost, g->stream));
		HIPCHK(g, hipStreamSynchronize(g->stream));
		HIPCHK(g, hipGetLastError());
		return BVG_OK;
	};
	// many queries, or (below) few queries for a good part of the arcs: the masked scan
	if (g->batch_dense > 0 && (uint64_t)q * (uint64_t)g->batch_dense >= (uint64_t)s.info.nodes) return run_dense();
	// 1. chain lengths -> slot bases
	if (!g->b_chainlen.need(sizeof(int32_t) * q) || !g->b_slotbase.need(sizeof(int64_t) * (q + 1)) || !g->sums.need(sizeof(int64_t) * (size_t)bv::scan_num_sums((int64_t)q)))
		return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	bv::launch_chain_len(gd, s.def, d_nodes, (int64_t)q, g->b_chainlen.as<int32_t>(), &dsm->maxdepth, &dsm->err, g->stream);
	bv::launch_scan(g->b_chainlen.as<int32_t>(), (int64_t)q, g->b_slotbase.as<int64_t>(), g->sums.as<int64_t>(), g->stream);
	HIPCHK(g, hipMemcpyAsync(&dsm->total, g->b_slotbase.as<int64_t>() + q, sizeof(int64_t), hipMemcpyDeviceToDevice, g->stream));
	int rc = fetch_small(g);
	if (rc) return rc;
	if (g->h_small->err) {
		const int e = g->h_small->err;
		if (e & bv::E_ARG) return fail(g, BVG_EARG, "Node index out of range");
		return fail(g, dev_err_to_status(e), "malformed bit stream");
	}
	const int64_t S = g->h_small->total;
	const int32_t maxlen = g->h_small->maxdepth;
	// 2. slots, caller rowptr, arena rows
	const size_t Sz = (size_t)std::max<int64_t>(S, 1);
	if (!g->b_node.need(4 * Sz) || !g->outd.need(4 * Sz) || !g->depth.need(4 * Sz) || !g->b_qidx.need(4 * Sz) || !g->b_aoutd.need(4 * Sz) ||
	    !g->b_qoutd.need(4 * q) || !g->rowstart.need(8 * (Sz + 1)) || !g->sums.need(sizeof(int64_t) * (size_t)bv::scan_num_sums((int64_t)std::max<size_t>(Sz, q))))
		return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	bv::launch_chain_fill(gd, s.def, d_nodes, (int64_t)q, g->b_slotbase.as<int64_t>(), g->b_node.as<int32_t>(), g->outd.as<int32_t>(), g->depth.as<int32_t>(),
	                      g->b_qidx.as<int32_t>(), g->b_aoutd.as<int32_t>(), g->b_qoutd.as<int32_t>(), g->stream);
	bv::launch_scan(g->b_qoutd.as<int32_t>(), (int64_t)q, d_rowptr, g->sums.as<int64_t>(), g->stream);
	bv::launch_scan(g->b_aoutd.as<int32_t>(), S, g->rowstart.as<int64_t>(), g->sums.as<int64_t>(), g->stream);
	HIPCHK(g, hipMemcpyAsync(&dsm->total, d_rowptr + q, sizeof(int64_t), hipMemcpyDeviceToDevice, g->stream));
	HIPCHK(g, hipMemcpyAsync(&dsm->halo_total, g->rowstart.as<int64_t>() + S, sizeof(int64_t), hipMemcpyDeviceToDevice, g->stream));
	rc = fetch_small(g);
	if (rc) return rc;
	const uint64_t arcs = (uint64_t)g->h_small->total;
	g->last_arcs = arcs;
	if (arcs_out) *arcs_out = arcs;
	if (!dev) HIPCHK(g, hipMemcpyAsync(rowptr, d_rowptr, sizeof(int64_t) * (q + 1), hipMemcpyDeviceToHost, g->stream));
	if (!succ) { HIPCHK(g, hipStreamSynchronize(g->stream)); return BVG_OK; }
	if (arcs > succ_cap) { HIPCHK(g, hipStreamSynchronize(g->stream)); return fail(g, BVG_ECAP, "successor buffer too small"); }
	// the rows wanted (with their ancestors) hold a good part of the graph's arcs: long rows, decoded once each by the masked scan
	if (g->batch_dense > 0 && ((uint64_t)arcs + (uint64_t)g->h_small->halo_total) * 8 >= (uint64_t)std::max<int64_t>(s.info.arcs, 1)) {
		HIPCHK(g, hipStreamSynchronize(g->stream));
		HIPCHK(g, hipMemsetAsync(g->small.p, 0, sizeof(Small), g->stream));
		return run_dense();
	}
	int32_t *d_succ = succ;
	if (!dev) {
		if (!g->stage_succ.need(sizeof(int32_t) * (size_t)std::max<uint64_t>(arcs, 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
		d_succ = g->stage_succ.as<int32_t>();
	}
	if (!g->halo.need(sizeof(int32_t) * (size_t)std::max<int64_t>(g->h_small->halo_total, 1))) return fail(g, BVG_ENOMEM, "arena allocation failed");
	// 3. decode: parse every slot, then resolve the chains level by level
	bv::BatchView v{};
	v.cnt = S; v.node = g->b_node.as<int32_t>(); v.outd = g->outd.as<int32_t>(); v.depth = g->depth.as<int32_t>(); v.qidx = g->b_qidx.as<int32_t>();
	v.arow = g->rowstart.as<int64_t>(); v.rowptr = d_rowptr; v.succ = d_succ; v.arena = g->halo.as<int32_t>(); v.succ_cap = succ_cap;
	int32_t coopMin, giantMin;
	pick_thresholds(g, (int64_t)arcs + g->h_small->halo_total, coopMin, giantMin);
	const bool coop = coopMin < 0x7fffffff && S <= 0x7fffffff;
	v.coop_min = coop ? coopMin : 0x7fffffff;
	const bool ovl = coop && g->overlap && !g->profile;
	if (coop) { // long records: the cooperative kernels of the scan path, over the batch's slots
		const int64_t arcsTot = (int64_t)arcs + g->h_small->halo_total;
		const int32_t giantCap = (int32_t)std::min<int64_t>(arcsTot / giantMin + 2, 0x7fffffff);
		const int64_t arenaCap = s.info.min_interval_length > 0 ? arcsTot / s.info.min_interval_length + 2 : 1;
		if (!g->biglist.need(4 * Sz) || !g->giantlist.need(sizeof(int32_t) * (size_t)giantCap) || !g->arena.need((size_t)bv::ARENA_ENTRY_BYTES * (size_t)arenaCap))
			return fail(g, BVG_ENOMEM, "device scratch allocation failed");
		int32_t *ctl = g->coopctl.as<int32_t>();
		HIPCHK(g, hipMemsetAsync(ctl, 0, 16 * sizeof(int32_t), g->stream));
		bv::launch_bparse_big(gd, s.def, v, coopMin, giantMin, g->biglist.as<int32_t>(), g->giantlist.as<int32_t>(), giantCap, ctl, g->arena.p, arenaCap,
		                      g->coop_waves, g->giant_groups, &dsm->err, g->stream, ovl ? side_b(g) : g->stream, ovl ? g->sideA : g->stream, g->evFork, g->evB, g->evA);
	}
	bv::launch_bparse(gd, s.def, v, &dsm->err, g->stream);
	if (ovl) {
		HIPCHK(g, hipStreamWaitEvent(g->stream, g->evA, 0));
		HIPCHK(g, hipStreamWaitEvent(g->stream, g->evB, 0));
	}
	for (int32_t l = 1; l < maxlen; l++) bv::launch_bcopy(gd, s.def, v, l, &dsm->err, g->stream);
	if (!dev && arcs) HIPCHK(g, hipMemcpyAsync(succ, d_succ, sizeof(int32_t) * (size_t)arcs, hipMemcpyDeviceToHost, g->stream));
	rc = fetch_small(g);
	if (rc) return rc;
	if (g->h_small->err) return fail(g, dev_err_to_status(g->h_small->err), "malformed or unsupported bit stream");
	return BVG_OK;
}

extern "C" int bvg_csr_hashcode(bvg_t *g, int32_t from, int32_t to, const int64_t *rowptr_dev, const int32_t *succ_dev, int32_t *hash_io) {
	if (!g || !g->st || !hash_io || from < 0 || to < from) return BVG_EARG;
	const Staged &s = *g->st;
	HIPCHK(g, hipSetDevice(s.device));
	if (g->pend.active) { int rc = finish_pending(g, nullptr); if (rc) return rc; }
	{ int rc = fork_from_user(g); if (rc) return rc; }
	const int32_t cnt = to - from;
	if (cnt == 0) return BVG_OK;
	// the fold runs over chunks of the n + m values of the scan order: their number depends on the arcs, which only the device knows
	int64_t ends[2] = { 0, 0 };
	HIPCHK(g, hipMemcpyAsync(&ends[0], rowptr_dev, sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
	HIPCHK(g, hipMemcpyAsync(&ends[1], rowptr_dev + cnt, sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
	HIPCHK(g, hipStreamSynchronize(g->stream));
	const int64_t arcs = ends[1] - ends[0];
	if (arcs < 0) return fail(g, BVG_EARG, "rowptr is not a CSR row pointer array");
	const size_t nb = (size_t)bv::hash_chunks(cnt, arcs);
	if (!g->hashA.need(sizeof(uint32_t) * nb) || !g->hashB.need(sizeof(uint32_t) * nb) || !g->hashBounds.need(sizeof(int32_t) * (nb + 1))) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	g->h_small->hash = *hash_io;
	int32_t *dh = &g->small.as<Small>()->hash;
	HIPCHK(g, hipMemcpyAsync(dh, &g->h_small->hash, sizeof(int32_t), hipMemcpyHostToDevice, g->stream));
	bv::launch_hash(from, cnt, arcs, rowptr_dev, succ_dev, g->hashA.as<uint32_t>(), g->hashB.as<uint32_t>(), g->hashBounds.as<int32_t>(), dh, g->stream);
	int rc = fetch_small(g);
	if (rc) return rc;
	*hash_io = g->h_small->hash;
	return BVG_OK;
}

namespace {
struct StatsHost { unsigned long long arcs, loops, dangling, terminal, num_gaps, tot_loc, tot_gap, min_key, max_key, delta[32], bad; }; // = bv::StatsDev
}

extern "C" int bvg_scan_stats(bvg_t *g, int32_t from, int32_t to, bvg_scan_stats_t *out, int32_t *indegree_dev) {
	if (!g || !g->st || !out) return BVG_EARG;
	const Staged &s = *g->st;
	if (from < 0 || from > s.info.nodes || to < from || to > s.info.nodes) return fail(g, BVG_EARG, "node range out of bounds"); // BVG:1165
	HIPCHK(g, hipSetDevice(s.device));
	if (bv::stats_dev_bytes() != sizeof(StatsHost) || !g->statsbuf.need(sizeof(StatsHost))) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	StatsHost h{};
	h.min_key = ~0ull;
	HIPCHK(g, hipMemcpy(g->statsbuf.p, &h, sizeof(h), hipMemcpyHostToDevice));
	const std::vector<int32_t> cut = plan_chunks_by_bits(s, from, to, scan_piece_arcs(g));
	for (size_t k = 0; k + 1 < cut.size(); k++) {
		const int32_t a = cut[k], e = cut[k + 1];
		if (e == a) continue;
		if (!g->stage_rowptr.need(sizeof(int64_t) * ((size_t)(e - a) + 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
		const uint64_t guess = (uint64_t)(est_arcs(s, a, e) * 1.1) + 4096;
		if (!g->stage_succ.need(sizeof(int32_t) * (size_t)guess)) return fail(g, BVG_ENOMEM, "staging allocation failed");
		uint64_t arcs = 0;
		int rc = decode_range_device(g, a, e, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), g->stage_succ.cap / sizeof(int32_t), false, &arcs);
		if (rc == BVG_ECAP) {
			if (!g->stage_succ.need(sizeof(int32_t) * (size_t)std::max<uint64_t>(arcs, 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
			rc = decode_range_device(g, a, e, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), g->stage_succ.cap / sizeof(int32_t), false, &arcs);
		}
		if (rc) return rc;
		bv::launch_stats(a, e - a, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), (int64_t)arcs, g->statsbuf.p, indegree_dev, s.info.nodes, g->stream);
		HIPCHK(g, hipStreamSynchronize(g->stream)); // the scratch rows are reused by the next chunk
	}
	HIPCHK(g, hipMemcpy(&h, g->statsbuf.p, sizeof(h), hipMemcpyDeviceToHost));
	memset(out, 0, sizeof(*out));
	if (h.bad) return fail(g, BVG_EFORMAT, "malformed bit stream: successor outside [0, nodes)");
	out->nodes = (uint64_t)(to - from); out->arcs = h.arcs; out->loops = h.loops; out->dangling = h.dangling; out->terminal = h.terminal;
	out->num_gaps = h.num_gaps; out->tot_gap = h.tot_gap; out->tot_loc = h.tot_loc;
	for (int i = 0; i < 32; i++) out->successor_delta_stats[i] = h.delta[i];
	// the reference starts from mind = Integer.MAX_VALUE, maxd = 0, both nodes 0, and moves on strict comparisons (Stats.java:98, :140-148)
	out->min_outdegree = 0x7fffffff; out->min_outdegree_node = 0; out->max_outdegree = 0; out->max_outdegree_node = 0;
	if (to > from) {
		out->min_outdegree = (int32_t)(h.min_key >> 32); out->min_outdegree_node = (int32_t)(uint32_t)h.min_key;
		out->max_outdegree = (int32_t)(h.max_key >> 32);
		out->max_outdegree_node = out->max_outdegree > 0 ? (int32_t)(0xffffffffu - (uint32_t)h.max_key) : 0;
	}
	return BVG_OK;
}

extern "C" int bvg_equal_range(bvg_t *a, bvg_t *b, int32_t from, int32_t to, int *equal) {
	if (!a || !b || !a->st || !b->st || !equal) return BVG_EARG;
	*equal = 0;
	const Staged &sa = *a->st, &sb = *b->st;
	if (sa.device != sb.device) return fail(a, BVG_EARG, "the two handles live on different devices");
	if (from < 0 || to < from || to > sa.info.nodes || to > sb.info.nodes) return fail(a, BVG_EARG, "node range out of bounds"); // BVG:1165
	HIPCHK(a, hipSetDevice(sa.device));
	if (!a->bfs_ctr.need(sizeof(unsigned long long))) return fail(a, BVG_ENOMEM, "device scratch allocation failed");
	HIPCHK(a, hipMemset(a->bfs_ctr.p, 0, sizeof(unsigned long long)));
	int differ = 0;
	const std::vector<int32_t> cut = plan_chunks_by_bits(sa, from, to, scan_piece_arcs(a));
	for (size_t k = 0; k + 1 < cut.size() && !differ; k++) {
		const int32_t lo = cut[k], hi = cut[k + 1];
		if (hi == lo) continue;
		uint64_t arcs[2] = { 0, 0 };
		bvg_t *h[2] = { a, b };
		for (int w = 0; w < 2; w++) { // (the same handle twice: its scratch holds one piece at a time -- compared with itself it is equal)
			bvg_t *g = h[w];
			if (w == 1 && b == a) break;
			const Staged &s = *g->st;
			if (!g->stage_rowptr.need(sizeof(int64_t) * ((size_t)(hi - lo) + 1))) return fail(a, BVG_ENOMEM, "staging allocation failed");
			const uint64_t guess = (uint64_t)(est_arcs(s, lo, hi) * 1.1) + 4096;
			if (!g->stage_succ.need(sizeof(int32_t) * (size_t)guess)) return fail(a, BVG_ENOMEM, "staging allocation failed");
			int rc = decode_range_device(g, lo, hi, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), g->stage_succ.cap / sizeof(int32_t), false, &arcs[w]);
			if (rc == BVG_ECAP) {
				if (!g->stage_succ.need(sizeof(int32_t) * (size_t)std::max<uint64_t>(arcs[w], 1))) return fail(a, BVG_ENOMEM, "staging allocation failed");
				rc = decode_range_device(g, lo, hi, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), g->stage_succ.cap / sizeof(int32_t), false, &arcs[w]);
			}
			if (rc) { if (g != a) fail(a, rc, b->err); return rc; }
		}
		if (b == a) continue;
		HIPCHK(a, hipStreamSynchronize(b->stream)); // b's rows are in place; the comparison runs on a's stream, behind a's decode
		bv::launch_rows_differ(hi - lo, a->stage_rowptr.as<int64_t>(), b->stage_rowptr.as<int64_t>(), a->stage_succ.as<int32_t>(), b->stage_succ.as<int32_t>(), (int *)a->bfs_ctr.p, a->stream);
		HIPCHK(a, hipMemcpyAsync(&differ, a->bfs_ctr.p, sizeof(int), hipMemcpyDeviceToHost, a->stream));
		HIPCHK(a, hipStreamSynchronize(a->stream)); // (the scratch rows are reused by the next piece)
	}
	*equal = differ ? 0 : 1;
	return BVG_OK;
}

extern "C" int bvg_hyperball_step(bvg_t *g, int32_t from, int32_t to, int log2m, const uint8_t *regs_in_dev, uint8_t *regs_out_dev, const uint8_t *modified_in_dev,
                                  uint8_t *modified_out_dev, uint64_t *changed) {
	if (!g || !g->st || !regs_in_dev || !regs_out_dev || !modified_out_dev || !changed) return BVG_EARG;
	const Staged &s = *g->st;
	if (log2m < 0 || log2m > 16) return fail(g, BVG_EARG, "log2m out of range");
	if (from < 0 || from > s.info.nodes || to < from || to > s.info.nodes) return fail(g, BVG_EARG, "node range out of bounds"); // BVG:1165
	HIPCHK(g, hipSetDevice(s.device));
	*changed = 0;
	if (!g->bfs_ctr.need(sizeof(unsigned long long))) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	HIPCHK(g, hipMemset(g->bfs_ctr.p, 0, sizeof(unsigned long long)));
	const std::vector<int32_t> cut = plan_chunks_by_bits(s, from, to, scan_piece_arcs(g));
	for (size_t k = 0; k + 1 < cut.size(); k++) {
		const int32_t a = cut[k], e = cut[k + 1];
		if (e == a) continue;
		if (!g->stage_rowptr.need(sizeof(int64_t) * ((size_t)(e - a) + 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
		const uint64_t guess = (uint64_t)(est_arcs(s, a, e) * 1.1) + 4096;
		if (!g->stage_succ.need(sizeof(int32_t) * (size_t)guess)) return fail(g, BVG_ENOMEM, "staging allocation failed");
		uint64_t arcs = 0;
		int rc = decode_range_device(g, a, e, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), g->stage_succ.cap / sizeof(int32_t), false, &arcs);
		if (rc == BVG_ECAP) {
			if (!g->stage_succ.need(sizeof(int32_t) * (size_t)std::max<uint64_t>(arcs, 1))) return fail(g, BVG_ENOMEM, "staging allocation failed");
			rc = decode_range_device(g, a, e, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), g->stage_succ.cap / sizeof(int32_t), false, &arcs);
		}
		if (rc) return rc;
		const int64_t bigCap = std::min<int64_t>(bv::hyperball_big_cap((int64_t)arcs), 0x7ffffff0);
		if (!g->bfs_rowptr.need(sizeof(int32_t) * ((size_t)bigCap + 1))) return fail(g, BVG_ENOMEM, "device scratch allocation failed"); // (the list of the piece's long rows, its count behind it)
		bv::launch_hyperball(a, e - a, g->stage_rowptr.as<int64_t>(), g->stage_succ.as<int32_t>(), s.info.nodes, 1 << log2m, regs_in_dev, regs_out_dev, modified_in_dev, modified_out_dev,
		                     g->bfs_ctr.as<unsigned long long>(), g->bfs_rowptr.as<int32_t>(), (int32_t)bigCap, g->bfs_rowptr.as<int32_t>() + bigCap, g->stream);
		HIPCHK(g, hipStreamSynchronize(g->stream)); // the scratch rows are reused by the next piece
	}
	unsigned long long c = 0;
	HIPCHK(g, hipMemcpy(&c, g->bfs_ctr.p, sizeof(c), hipMemcpyDeviceToHost));
	*changed = (uint64_t)c;
	return BVG_OK;
}

extern "C" int bvg_bfs_expand(bvg_t *g, const int32_t *frontier_dev, size_t q, int32_t *marker_dev, int32_t round, int parent, int32_t *out_dev, size_t out_cap, uint64_t *out_count) {
	if (!g || !g->st || !marker_dev || (!frontier_dev && q) || (!out_dev && out_cap) || !out_count) return BVG_EARG;
	const Staged &s = *g->st;
	*out_count = 0;
	if (q == 0) return BVG_OK;
	if (q > 0x7fffffffull) return fail(g, BVG_EARG, "frontier too long");
	HIPCHK(g, hipSetDevice(s.device));
	if (!g->bfs_rowptr.need(sizeof(int64_t) * (q + 1)) || !g->bfs_ctr.need(sizeof(unsigned long long))) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	uint64_t arcs = 0;
	int rc = bvg_successors_batch(g, frontier_dev, q, g->bfs_rowptr.as<int64_t>(), nullptr, 0, &arcs, BVG_OUT_DEVICE);
	if (rc) return rc;
	if (arcs == 0) return BVG_OK;
	if (!g->bfs_succ.need(sizeof(int32_t) * (size_t)arcs)) return fail(g, BVG_ENOMEM, "device scratch allocation failed");
	rc = bvg_successors_batch(g, frontier_dev, q, g->bfs_rowptr.as<int64_t>(), g->bfs_succ.as<int32_t>(), (size_t)arcs, &arcs, BVG_OUT_DEVICE);
	if (rc) return rc;
	HIPCHK(g, hipMemsetAsync(g->bfs_ctr.p, 0, sizeof(unsigned long long), g->stream));
	bv::launch_bfs_expand(frontier_dev, (int32_t)q, g->bfs_rowptr.as<int64_t>(), g->bfs_succ.as<int32_t>(), (int64_t)arcs, marker_dev, s.info.nodes, round, parent,
	                      out_dev, (uint64_t)out_cap, g->bfs_ctr.as<unsigned long long>(), g->stream);
	unsigned long long cnt = 0;
	HIPCHK(g, hipMemcpyAsync(&cnt, g->bfs_ctr.p, sizeof(cnt), hipMemcpyDeviceToHost, g->stream));
	HIPCHK(g, hipStreamSynchronize(g->stream));
	*out_count = cnt;
	if (cnt > out_cap) return fail(g, BVG_ECAP, "next frontier larger than the output buffer");
	return BVG_OK;
}

extern "C" int bvg_shard_bounds(const bvg_t *g, int parts, int32_t *bounds) {
	if (!g || !g->st || parts < 1 || !bounds) return BVG_EARG;
	const HostOffsets &off = g->st->h_offsets;
	const int32_t n = g->st->info.nodes;
	const int64_t totalBits = off.back();
	bounds[0] = 0;
	for (int k = 1; k < parts; k++) {
		// bounds[k] = min{x : off[x] >= k*total/parts}
		const int64_t target = (int64_t)((__int128)totalBits * k / parts);
		int32_t x = (int32_t)(std::lower_bound(off.begin(), off.begin() + n, target) - off.begin());
		bounds[k] = std::max(x, bounds[k - 1]);
	}
	bounds[parts] = n;
	return BVG_OK;
}
