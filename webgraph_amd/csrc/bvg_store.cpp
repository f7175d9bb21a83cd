// bvg_store.cpp -- BVGraph.store on the GPU behind the C ABI (include/bvgpu.h).
//
// BVGraph.store(graph, basename, windowSize, maxRefCount, minIntervalLength, zetaK, flags, numberOfThreads)
// (BVGraph.java:1679-1730 -> storeInternal :2436-2650) writes <basename>.graph / .offsets / .properties.  bvg_compress is
// the compression alone, from a CSR in host or device memory to streams in HBM (bv_encode.hip); bvg_store adds the three
// files.  No CPU fallback: without a HIP device both fail with BVG_EHIP.
#include "bv_host.hpp"
#include "bv_launch.hpp"
#include "host/bv_props.hpp"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>

namespace {

int sfail(char *errbuf, size_t errlen, int rc, const std::string &msg) {
	if (errbuf && errlen) { strncpy(errbuf, msg.c_str(), errlen - 1); errbuf[errlen - 1] = 0; }
	return rc;
}

bool one_of(int c, std::initializer_list<int> ok) { for (int v : ok) if (c == v) return true; return false; }

// setFlags (BVGraph.java:1317-1325) + the codings the writer accepts (writeOutdegree/writeReference/... :1839-2030)
int make_params(int window, int max_ref_count, int min_interval, int zeta_k, uint32_t flags, int threads, int32_t n, bve::Params &p, std::string &err) {
	if (window < 0 || max_ref_count < 0 || min_interval < 0 || zeta_k < 1) { err = "negative window / maxrefcount / minintervallength or zetak < 1"; return BVG_EARG; }
	if (threads <= 0) threads = 1; // numberOfThreads <= 0 means "choose" in the reference (BVGraph.java:2446-2450: the available processors); here the parts only bound the candidates: one
	p.W = window; p.R = max_ref_count; p.I = min_interval; p.K = zeta_k;
	p.c_outd = bve::C_GAMMA; p.c_blk = bve::C_GAMMA; p.c_res = bve::C_ZETA; p.c_ref = bve::C_UNARY; p.c_bc = bve::C_GAMMA; p.c_off = bve::C_GAMMA;
	if (flags & 0xF) p.c_outd = flags & 0xF;
	if ((flags >> 4) & 0xF) p.c_blk = (flags >> 4) & 0xF;
	if ((flags >> 8) & 0xF) p.c_res = (flags >> 8) & 0xF;
	if ((flags >> 12) & 0xF) p.c_ref = (flags >> 12) & 0xF;
	if ((flags >> 16) & 0xF) p.c_bc = (flags >> 16) & 0xF;
	if ((flags >> 20) & 0xF) p.c_off = (flags >> 20) & 0xF;
	using namespace bve;
	if (!one_of(p.c_outd, { C_GAMMA, C_DELTA }) || !one_of(p.c_blk, { C_GAMMA, C_DELTA, C_UNARY }) || !one_of(p.c_bc, { C_GAMMA, C_DELTA, C_UNARY }) ||
	    !one_of(p.c_ref, { C_UNARY, C_GAMMA, C_DELTA }) || !one_of(p.c_res, { C_GAMMA, C_ZETA, C_DELTA, C_GOLOMB, C_NIBBLE }) || !one_of(p.c_off, { C_GAMMA, C_DELTA })) {
		err = "The required coding is not supported"; // UnsupportedOperationException, BVGraph.java:1846 and siblings
		return BVG_EUNSUPPORTED;
	}
	if (threads > n) threads = n > 0 ? n : 1;
	p.per = (int32_t)(((int64_t)n + threads - 1) / threads);
	return BVG_OK;
}

void fill_stats(const bv::EncodeOut &o, uint64_t off_bits, int threads, bvg_store_stats_t *st) {
	if (!st) return;
	st->written_bits = o.graph_bits; st->offsets_bits = off_bits;
	st->bits_outdegrees = o.bits_outdegrees; st->bits_references = o.bits_references; st->bits_blocks = o.bits_blocks;
	st->bits_intervals = o.bits_intervals; st->bits_residuals = o.bits_residuals;
	st->copied_arcs = o.copied_arcs; st->intervalised_arcs = o.intervalised_arcs; st->residual_arcs = o.residual_arcs;
	st->tot_ref = o.tot_ref; st->tot_dist = o.tot_dist; st->max_ref_chain = o.max_ref_chain; st->threads = threads; st->selection_rounds = o.rounds;
	for (int i = 0; i < 32; i++) { st->successor_gap_bins[i] = o.successor_gap_bins[i]; st->residual_gap_bins[i] = o.residual_gap_bins[i]; }
}

int compress(int device, int32_t n, const int64_t *rowptr, const int32_t *succ, int in_flags, const bve::Params &p, bv::EncodeOut &out, std::string &err) {
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { err = "no HIP device available (libbvgpu has no CPU fallback)"; return BVG_EHIP; }
	if (device < 0 || device >= ndev) { err = "no such HIP device"; return BVG_EARG; }
	if (hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return BVG_EHIP; }
	const bool dev = (in_flags & BVG_OUT_DEVICE) != 0;
	int64_t *d_rowptr = nullptr;
	int32_t *d_succ = nullptr;
	int64_t m = 0;
	auto release = [&]() { if (!dev) { if (d_rowptr) (void)hipFree(d_rowptr); if (d_succ) (void)hipFree(d_succ); } };
	if (dev) {
		int64_t ends[2] = { 0, 0 };
		if (hipMemcpy(&ends[0], rowptr, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&ends[1], rowptr + n, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) { err = "reading rowptr failed"; return BVG_EHIP; }
		if (ends[0] != 0 || ends[1] < 0) { err = "rowptr must start at 0 and be monotone"; return BVG_EARG; }
		m = ends[1];
		d_rowptr = const_cast<int64_t *>(rowptr); d_succ = const_cast<int32_t *>(succ);
	} else {
		if (rowptr[0] != 0) { err = "rowptr must start at 0 and be monotone"; return BVG_EARG; }
		for (int32_t x = 0; x < n; x++) if (rowptr[x + 1] < rowptr[x]) { err = "rowptr must start at 0 and be monotone"; return BVG_EARG; }
		m = rowptr[n];
		if (m && !succ) { err = "null successor array"; return BVG_EARG; } // (before anything is staged)
		if (hipMalloc((void **)&d_rowptr, sizeof(int64_t) * ((size_t)n + 1)) != hipSuccess || hipMalloc((void **)&d_succ, sizeof(int32_t) * (size_t)(m ? m : 1)) != hipSuccess) {
			release(); (void)hipGetLastError(); err = "device allocation failed"; return BVG_ENOMEM;
		}
		if (hipMemcpy(d_rowptr, rowptr, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice) != hipSuccess || (m && hipMemcpy(d_succ, succ, sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice) != hipSuccess)) {
			release(); err = "staging the graph failed"; return BVG_EHIP;
		}
	}
	if (m && !succ) { release(); err = "null successor array"; return BVG_EARG; }
	const int rc = bv::encode_device(p, n, d_rowptr, d_succ, (uint64_t)m, out, err, nullptr);
	release();
	switch (rc) {
	case 0: return BVG_OK;
	case -1: return BVG_EARG;
	case -3: return BVG_EUNSUPPORTED;
	case -5: return BVG_ENOMEM;
	default: return BVG_EHIP;
	}
}

bool write_bytes(const std::string &path, const std::vector<uint8_t> &b) {
	FILE *f = fopen(path.c_str(), "wb");
	if (!f) return false;
	const bool ok = b.empty() || fwrite(b.data(), 1, b.size(), f) == b.size();
	return fclose(f) == 0 && ok;
}

// The three files of a graph appear together or not at all: each is written next to its place (<file>.tmp.<pid>) and renamed into it once all three are complete,
// .properties last -- a loader starts from it (BVG:1516-1530), so a store that dies half-way leaves the old graph, or none, never a new .graph with old .offsets.
struct AtomicTriple {
	std::string base, tag;
	explicit AtomicTriple(const std::string &b) : base(b), tag(".tmp." + std::to_string((long long)getpid())) {}
	std::string tmp(const char *ext) const { return base + ext + tag; }
	bool commit() const {
		for (const char *ext : { ".graph", ".offsets", ".properties" }) if (rename(tmp(ext).c_str(), (base + ext).c_str()) != 0) { discard(); return false; }
		return true;
	}
	void discard() const { for (const char *ext : { ".graph", ".offsets", ".properties" }) (void)remove(tmp(ext).c_str()); }
};

} // namespace

extern "C" int bvg_compress(int device, int32_t n, const int64_t *rowptr, const int32_t *succ, int in_flags, int window, int max_ref_count, int min_interval, int zeta_k,
                            uint32_t flags, int threads, bvg_compressed_t *out, char *errbuf, size_t errlen) {
	if (!out || n < 0 || !rowptr) return sfail(errbuf, errlen, BVG_EARG, "null argument or negative node count");
	memset(out, 0, sizeof *out);
	bve::Params p{};
	std::string err;
	int rc = make_params(window, max_ref_count, min_interval, zeta_k, flags, threads, n, p, err);
	if (rc) return sfail(errbuf, errlen, rc, err);
	bv::EncodeOut o;
	rc = compress(device, n, rowptr, succ, in_flags, p, o, err);
	if (rc) return sfail(errbuf, errlen, rc, err);
	out->device = device;
	out->graph_dev = (uint8_t *)o.graph_words; out->graph_bits = o.graph_bits;
	out->offsets_stream_dev = (uint8_t *)o.off_words; out->offsets_bits = o.off_bits;
	out->bit_offsets_dev = o.offsets;
	fill_stats(o, o.off_bits, threads, &out->stats);
	return BVG_OK;
}

extern "C" void bvg_compressed_free(bvg_compressed_t *c) {
	if (!c) return;
	if (c->device >= 0 && (c->graph_dev || c->offsets_stream_dev || c->bit_offsets_dev)) (void)hipSetDevice(c->device);
	for (void *q : { (void *)c->graph_dev, (void *)c->offsets_stream_dev, (void *)c->bit_offsets_dev }) if (q) (void)hipFree(q);
	c->graph_dev = nullptr; c->offsets_stream_dev = nullptr; c->bit_offsets_dev = nullptr;
}

extern "C" int bvg_compressed_copy(const bvg_compressed_t *c, int32_t n, uint8_t *graph_host, uint8_t *offsets_host, int64_t *bit_offsets_host) {
	if (!c || n < 0) return BVG_EARG;
	if (hipSetDevice(c->device) != hipSuccess) return BVG_EHIP;
	const size_t gb = (size_t)((c->graph_bits + 7) / 8), ob = (size_t)((c->offsets_bits + 7) / 8);
	if (graph_host && gb && hipMemcpy(graph_host, c->graph_dev, gb, hipMemcpyDeviceToHost) != hipSuccess) return BVG_EHIP;
	if (offsets_host && ob && hipMemcpy(offsets_host, c->offsets_stream_dev, ob, hipMemcpyDeviceToHost) != hipSuccess) return BVG_EHIP;
	if (bit_offsets_host && hipMemcpy(bit_offsets_host, c->bit_offsets_dev, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyDeviceToHost) != hipSuccess) return BVG_EHIP;
	return BVG_OK;
}

extern "C" int bvg_store(const char *basename, int device, int32_t n, const int64_t *rowptr, const int32_t *succ, int in_flags, int window, int max_ref_count,
                         int min_interval, int zeta_k, uint32_t flags, int threads, bvg_store_stats_t *stats, char *errbuf, size_t errlen) {
	if (!basename) return sfail(errbuf, errlen, BVG_EARG, "null basename");
	bvg_compressed_t c;
	int rc = bvg_compress(device, n, rowptr, succ, in_flags, window, max_ref_count, min_interval, zeta_k, flags, threads, &c, errbuf, errlen);
	if (rc) return rc;
	std::vector<uint8_t> graph((size_t)((c.graph_bits + 7) / 8)), offs((size_t)((c.offsets_bits + 7) / 8));
	if ((!graph.empty() && hipMemcpy(graph.data(), c.graph_dev, graph.size(), hipMemcpyDeviceToHost) != hipSuccess) ||
	    (!offs.empty() && hipMemcpy(offs.data(), c.offsets_stream_dev, offs.size(), hipMemcpyDeviceToHost) != hipSuccess)) { bvg_compressed_free(&c); return sfail(errbuf, errlen, BVG_EHIP, "copying the streams back failed"); }
	uint64_t m = 0;
	{
		int64_t last = 0;
		if (in_flags & BVG_OUT_DEVICE) { if (hipMemcpy(&last, rowptr + n, sizeof last, hipMemcpyDeviceToHost) != hipSuccess) { bvg_compressed_free(&c); return sfail(errbuf, errlen, BVG_EHIP, "reading rowptr failed"); } }
		else last = rowptr[n];
		m = (uint64_t)last;
	}
	const bvg_store_stats_t st = c.stats;
	bvg_compressed_free(&c);
	const std::string base(basename);
	const AtomicTriple files(base);
	if (!write_bytes(files.tmp(".graph"), graph) || !write_bytes(files.tmp(".offsets"), offs)) { files.discard(); return sfail(errbuf, errlen, BVG_EIO, "cannot write " + base + ".graph / .offsets"); }
	bvprops::Counters cnt{ st.written_bits, st.bits_outdegrees, st.bits_references, st.bits_blocks, st.bits_intervals, st.bits_residuals,
	                       st.copied_arcs, st.intervalised_arcs, st.residual_arcs, st.tot_ref, st.tot_dist, {}, {} };
	for (int i = 0; i < 32; i++) { cnt.successor_gap_bins[i] = st.successor_gap_bins[i]; cnt.residual_gap_bins[i] = st.residual_gap_bins[i]; }
	const int resCoding = (flags >> 8) & 0xF;
	if (!bvprops::write(files.tmp(".properties"), n, m, window, max_ref_count, min_interval, zeta_k, resCoding == 0 || resCoding == bve::C_ZETA, flags, cnt) || !files.commit())
		{ files.discard(); return sfail(errbuf, errlen, BVG_EIO, "cannot write " + base + ".properties"); }
	if (stats) *stats = st;
	return BVG_OK;
}

// decode -> compress without leaving HBM: BVGraph.store(graph, ...) for a graph that is itself a handle of this library
extern "C" int bvg_recompress(bvg_t *g, const char *basename, int window, int max_ref_count, int min_interval, int zeta_k, uint32_t flags, int threads,
                              bvg_store_stats_t *stats, char *errbuf, size_t errlen) {
	if (!g || !basename) return sfail(errbuf, errlen, BVG_EARG, "null argument");
	bvg_info_t info;
	int rc = bvg_info(g, &info);
	if (rc) return sfail(errbuf, errlen, rc, bvg_last_error(g));
	if (info.shard_from != 0 || info.shard_to != info.nodes) return sfail(errbuf, errlen, BVG_EUNSUPPORTED, "a shard handle holds a slice of the graph: recompress from a whole-graph handle");
	if (hipSetDevice(info.device) != hipSuccess) return sfail(errbuf, errlen, BVG_EHIP, "hipSetDevice failed");
	int64_t *d_rowptr = nullptr;
	int32_t *d_succ = nullptr;
	const size_t m = (size_t)info.arcs;
	if (hipMalloc((void **)&d_rowptr, sizeof(int64_t) * ((size_t)info.nodes + 1)) != hipSuccess || hipMalloc((void **)&d_succ, sizeof(int32_t) * (m ? m : 1)) != hipSuccess) {
		if (d_rowptr) (void)hipFree(d_rowptr);
		(void)hipGetLastError();
		return sfail(errbuf, errlen, BVG_ENOMEM, "device allocation failed");
	}
	uint64_t arcs = 0;
	rc = bvg_decode_range(g, 0, info.nodes, d_rowptr, d_succ, m, &arcs, BVG_OUT_DEVICE);
	if (rc) sfail(errbuf, errlen, rc, bvg_last_error(g));
	else rc = bvg_store(basename, info.device, info.nodes, d_rowptr, d_succ, BVG_OUT_DEVICE, window, max_ref_count, min_interval, zeta_k, flags, threads, stats, errbuf, errlen);
	(void)hipFree(d_rowptr); (void)hipFree(d_succ);
	return rc;
}

// ---- EFGraph.store on the device (bv_efw.hip)
namespace {
int store_ef_impl(const char *basename, int device, int32_t n, const int64_t *d_rowptr, const int32_t *d_succ, uint64_t m, int32_t upper_bound, int log2_quantum, int big_endian,
                  std::string &err) {
	uint64_t *d_words = nullptr, nwords = 0, bits = 0, obits = 0;
	int32_t *d_reclen = nullptr;
	int64_t *d_off = nullptr;
	uint32_t *d_ow = nullptr;
	auto release = [&]() { for (void *q : { (void *)d_words, (void *)d_reclen, (void *)d_off, (void *)d_ow }) if (q) (void)hipFree(q); };
	const auto t0 = std::chrono::steady_clock::now();
	int rc = bv::ef_encode_device(n, d_rowptr, d_succ, (uint64_t)upper_bound, log2_quantum, &d_words, &nwords, &bits, &d_reclen, &d_off, nullptr);
	if (bv_env("BVGPU_ENC_TRACE")) fprintf(stderr, "[bvgpu enc] EFGraph: CSR in HBM -> stream in HBM %.3f ms (%llu bits, rc %d)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), (unsigned long long)bits, rc);
	if (rc) {
		err = rc == -1 ? "successor lists must be strictly increasing, non-negative and below the upper bound" : rc == -3 ? "a record of 2^31 bits or more" : rc == -5 ? "device allocation failed" : "the EFGraph kernels failed";
		return rc == -1 ? BVG_EARG : rc == -3 ? BVG_EUNSUPPORTED : rc == -5 ? BVG_ENOMEM : BVG_EHIP;
	}
	rc = bv::offsets_stream_device(BVG_DELTA, d_reclen, n, &d_ow, &obits, nullptr); // offsets.writeLongDelta, EFGraph.java:830, :855
	if (rc) { release(); err = "the offsets kernels failed"; return rc == -5 ? BVG_ENOMEM : BVG_EHIP; }
	std::vector<uint8_t> graph((size_t)nwords * 8), offs((size_t)((obits + 7) / 8));
	// bits of the outdegrees: gamma(d) = 2 * msb(d + 1) + 1; the rest of every record is successors (:866-888 persists both)
	std::vector<int64_t> rp((size_t)n + 1);
	if (hipMemcpy(graph.data(), d_words, graph.size(), hipMemcpyDeviceToHost) != hipSuccess || (!offs.empty() && hipMemcpy(offs.data(), d_ow, offs.size(), hipMemcpyDeviceToHost) != hipSuccess) ||
	    hipMemcpy(rp.data(), d_rowptr, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost) != hipSuccess) { release(); err = "copying the streams back failed"; return BVG_EHIP; }
	release();
	if (big_endian) for (size_t i = 0; i + 8 <= graph.size(); i += 8) { std::swap(graph[i], graph[i + 7]); std::swap(graph[i + 1], graph[i + 6]); std::swap(graph[i + 2], graph[i + 5]); std::swap(graph[i + 3], graph[i + 4]); }
	uint64_t bitsOutd = 0;
	for (int32_t x = 0; x < n; x++) bitsOutd += 2 * (uint64_t)(63 - __builtin_clzll((unsigned long long)(rp[(size_t)x + 1] - rp[(size_t)x] + 1))) + 1;
	const std::string base(basename);
	const AtomicTriple files(base);
	if (!write_bytes(files.tmp(".graph"), graph) || !write_bytes(files.tmp(".offsets"), offs)) { files.discard(); err = "cannot write " + base + ".graph / .offsets"; return BVG_EIO; }
	if (!bvprops::write_ef(files.tmp(".properties"), n, m, upper_bound, log2_quantum, big_endian != 0, nwords * 64, bitsOutd, bits - bitsOutd) || !files.commit()) { files.discard(); err = "cannot write " + base + ".properties"; return BVG_EIO; }
	return BVG_OK;
}
} // namespace

extern "C" int bvg_store_ef(const char *basename, int device, int32_t n, const int64_t *rowptr, const int32_t *succ, int in_flags, int32_t upper_bound, int log2_quantum, int big_endian,
                            char *errbuf, size_t errlen) {
	if (!basename || n < 0 || !rowptr) return sfail(errbuf, errlen, BVG_EARG, "null argument or negative node count");
	if (upper_bound == 0) upper_bound = n; // EFGraph.store(graph, basename): the number of nodes (:808-810)
	if (upper_bound < n || log2_quantum < 0 || log2_quantum > 62) return sfail(errbuf, errlen, BVG_EARG, "upper bound below the number of nodes, or a negative quantum"); // :814
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sfail(errbuf, errlen, BVG_EHIP, "no HIP device available (libbvgpu has no CPU fallback)");
	if (device < 0 || device >= ndev) return sfail(errbuf, errlen, BVG_EARG, "no such HIP device");
	if (hipSetDevice(device) != hipSuccess) return sfail(errbuf, errlen, BVG_EHIP, "hipSetDevice failed");
	const bool dev = (in_flags & BVG_OUT_DEVICE) != 0;
	int64_t *d_rowptr = const_cast<int64_t *>(rowptr);
	int32_t *d_succ = const_cast<int32_t *>(succ);
	int64_t m = 0;
	auto release = [&]() { if (!dev) { if (d_rowptr) (void)hipFree(d_rowptr); if (d_succ) (void)hipFree(d_succ); } };
	if (dev) {
		int64_t ends[2] = { 0, 0 };
		if (hipMemcpy(&ends[0], rowptr, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&ends[1], rowptr + n, sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) return sfail(errbuf, errlen, BVG_EHIP, "reading rowptr failed");
		if (ends[0] != 0 || ends[1] < 0) return sfail(errbuf, errlen, BVG_EARG, "rowptr must start at 0 and be monotone");
		m = ends[1];
	} else {
		if (rowptr[0] != 0) return sfail(errbuf, errlen, BVG_EARG, "rowptr must start at 0 and be monotone");
		for (int32_t x = 0; x < n; x++) if (rowptr[x + 1] < rowptr[x]) return sfail(errbuf, errlen, BVG_EARG, "rowptr must start at 0 and be monotone");
		m = rowptr[n];
		if (m && !succ) return sfail(errbuf, errlen, BVG_EARG, "null successor array"); // (before anything is staged)
		d_rowptr = nullptr; d_succ = nullptr;
		if (hipMalloc((void **)&d_rowptr, sizeof(int64_t) * ((size_t)n + 1)) != hipSuccess || hipMalloc((void **)&d_succ, sizeof(int32_t) * (size_t)(m ? m : 1)) != hipSuccess) { release(); (void)hipGetLastError(); return sfail(errbuf, errlen, BVG_ENOMEM, "device allocation failed"); }
		if (hipMemcpy(d_rowptr, rowptr, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice) != hipSuccess || (m && hipMemcpy(d_succ, succ, sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice) != hipSuccess)) { release(); return sfail(errbuf, errlen, BVG_EHIP, "staging the graph failed"); }
	}
	if (m && !succ) { release(); return sfail(errbuf, errlen, BVG_EARG, "null successor array"); }
	std::string err;
	const int rc = store_ef_impl(basename, device, n, d_rowptr, d_succ, (uint64_t)m, upper_bound, log2_quantum, big_endian, err);
	release();
	return rc ? sfail(errbuf, errlen, rc, err) : BVG_OK;
}

// EFGraph.store(graph, basename) for a graph that is a handle of this library (either format): decode into HBM, encode from there
extern "C" int bvg_recompress_ef(bvg_t *g, const char *basename, int32_t upper_bound, int log2_quantum, int big_endian, char *errbuf, size_t errlen) {
	if (!g || !basename) return sfail(errbuf, errlen, BVG_EARG, "null argument");
	bvg_info_t info;
	int rc = bvg_info(g, &info);
	if (rc) return sfail(errbuf, errlen, rc, bvg_last_error(g));
	if (info.shard_from != 0 || info.shard_to != info.nodes) return sfail(errbuf, errlen, BVG_EUNSUPPORTED, "a shard handle holds a slice of the graph: recompress from a whole-graph handle");
	if (hipSetDevice(info.device) != hipSuccess) return sfail(errbuf, errlen, BVG_EHIP, "hipSetDevice failed");
	int64_t *d_rowptr = nullptr;
	int32_t *d_succ = nullptr;
	const size_t m = (size_t)info.arcs;
	if (hipMalloc((void **)&d_rowptr, sizeof(int64_t) * ((size_t)info.nodes + 1)) != hipSuccess || hipMalloc((void **)&d_succ, sizeof(int32_t) * (m ? m : 1)) != hipSuccess) {
		if (d_rowptr) (void)hipFree(d_rowptr);
		(void)hipGetLastError();
		return sfail(errbuf, errlen, BVG_ENOMEM, "device allocation failed");
	}
	uint64_t arcs = 0;
	rc = bvg_decode_range(g, 0, info.nodes, d_rowptr, d_succ, m, &arcs, BVG_OUT_DEVICE);
	if (rc) sfail(errbuf, errlen, rc, bvg_last_error(g));
	else rc = bvg_store_ef(basename, info.device, info.nodes, d_rowptr, d_succ, BVG_OUT_DEVICE, upper_bound, log2_quantum, big_endian, errbuf, errlen);
	(void)hipFree(d_rowptr); (void)hipFree(d_succ);
	return rc;
}
