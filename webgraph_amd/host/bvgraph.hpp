// bvgraph.hpp -- C++ host-side mirror of the reference's graph API for the BVGraph decode path, header-only,
// over the libbvgpu C ABI (include/bvgpu.h).
//
// The reference is Java and this build environment has no JVM, so the host side above the C ABI is provided in
// C++ with the reference's names, argument meaning and error behaviour:
//   webgraph::BVGraph        <->  it.unimi.dsi.webgraph.BVGraph / ImmutableGraph   (ImmutableGraph.java:169-772)
//   webgraph::NodeIterator   <->  NodeIterator / BVGraphNodeIterator               (NodeIterator.java:34-107, BVGraph.java:1136-1281)
//   webgraph::LazyIntIterator<->  LazyIntIterator                                  (LazyIntIterator.java:28-43)
// Exceptions: std::invalid_argument = IllegalArgumentException, std::logic_error = IllegalStateException,
// webgraph::unsupported_operation = UnsupportedOperationException, webgraph::io_error = IOException.
// All decoding happens in the HIP kernels; nothing here decodes a bit.
#pragma once
#include "../../include/bvgpu.h"

#include <algorithm>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace webgraph {

struct unsupported_operation : std::runtime_error { using std::runtime_error::runtime_error; };
struct io_error : std::runtime_error { using std::runtime_error::runtime_error; };

namespace detail {
inline void check(int rc, const bvg_t *h) {
	if (rc == BVG_OK) return;
	const std::string msg = h ? bvg_last_error(h) : "bvgpu error";
	switch (rc) {
	case BVG_EARG: throw std::invalid_argument(msg);
	case BVG_ESTATE: throw std::logic_error(msg);
	case BVG_EUNSUPPORTED: throw unsupported_operation(msg);
	case BVG_EIO: throw io_error(msg);
	case BVG_ENOMEM: throw std::bad_alloc();
	default: throw std::runtime_error(msg + " (bvg_status " + std::to_string(rc) + ")");
	}
}
inline void check_msg(int rc, const char *msg) { // entry points that report through a caller's buffer
	if (rc == BVG_OK) return;
	switch (rc) {
	case BVG_EARG: throw std::invalid_argument(msg);
	case BVG_ESTATE: throw std::logic_error(msg);
	case BVG_EUNSUPPORTED: throw unsupported_operation(msg);
	case BVG_EIO: throw io_error(msg);
	case BVG_ENOMEM: throw std::bad_alloc();
	default: throw std::runtime_error(std::string(msg) + " (bvg_status " + std::to_string(rc) + ")");
	}
}
} // namespace detail

// Increasing successor ids, then -1 forever (LazyIntIterator.java:28-43).
class LazyIntIterator {
	const int32_t *p_, *end_;
public:
	LazyIntIterator(const int32_t *p, size_t n) : p_(p), end_(p + n) {}
	int32_t nextInt() { return p_ < end_ ? *p_++ : -1; }
	int skip(int n) { const int k = (int)std::min<ptrdiff_t>(n, end_ - p_); p_ += k; return k; }
};

class BVGraph;

// Sequential scan served from GPU-decoded batches (BVGraphNodeIterator, BVGraph.java:1136-1281).
class NodeIterator {
	const BVGraph *g_;
	std::shared_ptr<const BVGraph> own_; // copy() / split iterators decode through a flyweight handle of their own (bvg_clone):
	                                     // the reference hands them to other threads (BVGraph.java:2471-2477) and a bvg_t is not thread-safe
	int32_t from_, curr_, limit_, lo_ = 0, hi_ = 0, batch_;
	std::vector<int64_t> rowptr_;
	std::vector<int32_t> succ_;
	void fill();
public:
	NodeIterator(const BVGraph *g, int32_t from, int32_t upperBound, int32_t batchNodes = 1 << 20);
	NodeIterator(std::shared_ptr<const BVGraph> own, int32_t from, int32_t upperBound, int32_t batchNodes);
	bool hasNext() const { return curr_ < limit_; }                                  // BVGraph.java:1216
	int32_t nextInt() {
		if (!hasNext()) throw std::out_of_range("NoSuchElementException");
		if (++curr_ >= hi_) fill();
		return curr_;
	}
	int32_t outdegree() const { check(); return (int32_t)(rowptr_[curr_ - lo_ + 1] - rowptr_[curr_ - lo_]); }
	// valid until the next nextInt() crosses a batch boundary (the reference's window row is valid for W+1 calls)
	const int32_t *successorArray() const { check(); return succ_.data() + rowptr_[curr_ - lo_]; }
	LazyIntIterator successors() const { return LazyIntIterator(successorArray(), (size_t)outdegree()); }
	NodeIterator copy(int32_t upperBound) const;                                     // BVGraph.java:1253-1260
private:
	void check() const { if (curr_ == from_ - 1) throw std::logic_error("nextInt() has not been called"); } // BVGraph.java:1220
};

class BVGraph {
	struct Closer { void operator()(bvg_t *h) const { bvg_close(h); } };
	std::unique_ptr<bvg_t, Closer> h_;
	bvg_info_t info_{};
	std::string basename_;
	explicit BVGraph(bvg_t *h, std::string base) : h_(h), basename_(std::move(base)) { detail::check(bvg_info(h, &info_), h); }
public:
	// ImmutableGraph.load / loadMapped / loadOffline(basename): all stage the graph in HBM here (BVGraph.java:1380-1516)
	static BVGraph load(const std::string &basename, int device = 0) {
		bvg_t *h = nullptr;
		const int rc = bvg_open(basename.c_str(), device, &h);
		if (rc) { std::unique_ptr<bvg_t, Closer> guard(h); detail::check(rc, h); }
		return BVGraph(h, basename);
	}
	static BVGraph loadMapped(const std::string &b, int device = 0) { return load(b, device); }
	static BVGraph loadOffline(const std::string &b, int device = 0) { return load(b, device); }

	// BVGraph.store(graph, basename, windowSize, maxRefCount, minIntervalLength, zetaK, flags, numberOfThreads) (BVGraph.java:1679-1730) for a graph
	// that is a handle of this library: decoded and recompressed on the device (bvg_recompress); the reference's defaults (BVGraph.java:455-470)
	static void store(const BVGraph &graph, const std::string &basename, int windowSize = 7, int maxRefCount = 3, int minIntervalLength = 4, int zetaK = 3, uint32_t flags = 0,
	                  int numberOfThreads = 1) {
		char err[512] = "";
		detail::check_msg(bvg_recompress(graph.handle(), basename.c_str(), windowSize, maxRefCount, minIntervalLength, zetaK, flags, numberOfThreads, nullptr, err, sizeof err), err);
	}
	// EFGraph.store(graph, basename) (EFGraph.java:800-810): the quasi-succinct format, written on the device (bvg_recompress_ef)
	static void storeEF(const BVGraph &graph, const std::string &basename, int log2Quantum = 8, bool bigEndian = false) {
		char err[512] = "";
		detail::check_msg(bvg_recompress_ef(graph.handle(), basename.c_str(), 0, log2Quantum, bigEndian ? 1 : 0, err, sizeof err), err);
	}
	bool isEFGraph() const { return info_.format == BVG_FORMAT_EF; }

	int32_t numNodes() const { return info_.nodes; }               // ImmutableGraph.java:254
	int64_t numArcs() const { return info_.arcs; }                 // :260
	bool randomAccess() const { return true; }                     // :268
	bool hasCopiableIterators() const { return true; }             // BVGraph.java:591-599
	const std::string &basename() const { return basename_; }
	int32_t windowSize() const { return info_.window_size; }       // BVGraph.java:610
	int32_t maxRefCount() const { return info_.max_ref_count; }    // BVGraph.java:618
	bvg_t *handle() const { return h_.get(); }

	BVGraph copy() const {                                          // BVGraph.java:552-577 (flyweight)
		bvg_t *h = nullptr;
		const int rc = bvg_clone(h_.get(), &h);
		if (rc) { std::unique_ptr<bvg_t, Closer> guard(h); detail::check(rc, h_.get()); }
		return BVGraph(h, basename_);
	}
	int32_t outdegree(int32_t x) const {                            // BVGraph.java:858-888
		if (x < 0 || x >= numNodes()) throw std::invalid_argument("Node index out of range: " + std::to_string(x));
		int32_t d = 0;
		detail::check(bvg_outdegrees(h_.get(), x, x + 1, &d, BVG_OUT_HOST), h_.get());
		return d;
	}
	// a distinct, exact-length array per call (ImmutableGraph.java:329-333)
	std::vector<int32_t> successorArray(int32_t x) const {
		if (x < 0 || x >= numNodes()) throw std::invalid_argument("Node index out of range: " + std::to_string(x)); // BVGraph.java:900
		int64_t rp[2] = { 0, 0 };
		uint64_t arcs = 0;
		detail::check(bvg_successors_batch(h_.get(), &x, 1, rp, nullptr, 0, &arcs, BVG_OUT_HOST), h_.get());
		std::vector<int32_t> out((size_t)arcs);
		detail::check(bvg_successors_batch(h_.get(), &x, 1, rp, out.data(), out.size(), &arcs, BVG_OUT_HOST), h_.get());
		return out;
	}
	// CSR of nodes [from, to): what draining nodeIterator(from).copy(to) yields.  One call: the results arrive in the
	// handle's pinned buffers (bvg_decode_range_view) and are copied out once.
	void decodeRange(int32_t from, int32_t to, std::vector<int64_t> &rowptr, std::vector<int32_t> &succ) const {
		const int64_t *rp = nullptr; const int32_t *sc = nullptr;
		uint64_t arcs = 0;
		detail::check(bvg_decode_range_view(h_.get(), from, to, &rp, &sc, &arcs), h_.get());
		rowptr.assign(rp, rp + (size_t)std::max(to - from, 0) + 1);
		succ.assign(sc, sc + arcs);
	}
	// hashCode() continued from h over nodes [from, to) on the device, nothing materialised (bvg_scan_checksum)
	int32_t scanChecksum(int32_t from, int32_t to, int32_t h, uint64_t *arcs = nullptr) const {
		detail::check(bvg_scan_checksum(h_.get(), from, to, &h, arcs), h_.get());
		return h;
	}
	NodeIterator nodeIterator(int32_t from = 0) const { return NodeIterator(this, from, INT32_MAX); } // BVGraph.java:1293
	// ImmutableGraph.splitNodeIterators (ImmutableGraph.java:379-409), random-access branch; unused slots are empty iterators
	std::vector<NodeIterator> splitNodeIterators(int howMany) const {
		if (howMany < 1) throw std::invalid_argument("howMany < 1");
		const int32_t n = numNodes();
		const int32_t m = (int32_t)(((int64_t)n + howMany - 1) / howMany);
		std::vector<NodeIterator> res;
		for (int32_t from = 0; from < n; from += m) res.push_back(nodeIterator(from).copy(from + m));
		while ((int)res.size() < howMany) res.emplace_back(this, n, n);
		return res;
	}
	// ImmutableGraph.hashCode() (ImmutableGraph.java:757-770) over a host-side scan
	int32_t hashCode() const {
		uint32_t h = (uint32_t)-1;
		NodeIterator it = nodeIterator();
		for (int32_t n = numNodes(); n-- != 0;) {
			h = h * 31u + (uint32_t)it.nextInt();
			const int32_t *s = it.successorArray();
			for (int32_t d = it.outdegree(); d-- != 0;) h = h * 31u + (uint32_t)s[d];
		}
		return (int32_t)h;
	}
	// ImmutableGraph.equals() (ImmutableGraph.java:731-749): same size, same successor list for every node
	bool equals(const BVGraph &o) const {
		int32_t n = numNodes();
		if (n != o.numNodes()) return false;
		{ // two handles of the library: compared on the device, no row reaches the host (bvg_equal_range)
			int eq = 0;
			const int rc = bvg_equal_range(h_.get(), o.h_.get(), 0, n, &eq);
			if (rc == BVG_OK) return eq != 0;
			if (rc != BVG_EARG) detail::check(rc, h_.get()); // (BVG_EARG: handles on different devices -- the host comparison below serves them)
		}
		NodeIterator i = nodeIterator(), j = o.nodeIterator();
		while (n-- != 0) {
			i.nextInt();
			j.nextInt();
			int32_t d = i.outdegree();
			if (d != j.outdegree()) return false;
			const int32_t *s = i.successorArray(), *t = j.successorArray();
			while (d-- != 0) if (s[d] != t[d]) return false;
		}
		return true;
	}
};

inline NodeIterator::NodeIterator(const BVGraph *g, int32_t from, int32_t upperBound, int32_t batchNodes) : g_(g), from_(from), batch_(batchNodes) {
	if (from < 0 || from > g->numNodes()) throw std::invalid_argument("Node index out of range: " + std::to_string(from)); // BVGraph.java:1165
	curr_ = from - 1;
	limit_ = std::min(upperBound, g->numNodes()) - 1; // hasNextLimit, BVGraph.java:1185
	lo_ = hi_ = from;
}
inline void NodeIterator::fill() {
	lo_ = curr_;
	hi_ = (int32_t)std::min<int64_t>((int64_t)lo_ + batch_, (int64_t)limit_ + 1);
	g_->decodeRange(lo_, hi_, rowptr_, succ_);
}
inline NodeIterator::NodeIterator(std::shared_ptr<const BVGraph> own, int32_t from, int32_t upperBound, int32_t batchNodes) : NodeIterator(own.get(), from, upperBound, batchNodes) { own_ = std::move(own); }
inline NodeIterator NodeIterator::copy(int32_t upperBound) const { return NodeIterator(std::make_shared<const BVGraph>(g_->copy()), curr_ + 1, upperBound, batch_); }

} // namespace webgraph
