// Drives the C++ host mirror (webgraph_amd/host/bvgraph.hpp) the way the reference's tests drive ImmutableGraph
// (WebGraphTestCase.assertGraph / BVGraphTest.testLarge).  Usage: host_mirror_test <basename> <expected hashCode> <expected arcs>
#include "../../webgraph_amd/host/bvgraph.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
	if (argc < 4) { fprintf(stderr, "usage: %s basename hashcode arcs [out-basename minintervallength]\n", argv[0]); return 2; }
	using namespace webgraph;
	try {
		BVGraph g = BVGraph::load(argv[1]);
		const int32_t n = g.numNodes();
		REQUIRE(g.numArcs() == atoll(argv[3]));
		REQUIRE(g.randomAccess() && g.hasCopiableIterators());
		REQUIRE(g.hashCode() == atoi(argv[2]));
		{ BVGraph c = g.copy(); REQUIRE(g.equals(c) && c.equals(g)); } // ImmutableGraph.equals over a flyweight copy
		// sequential vs random access on a sample, including the -1 terminator
		NodeIterator it = g.nodeIterator();
		int64_t arcs = 0;
		for (int32_t x = 0; x < n; x++) {
			REQUIRE(it.nextInt() == x);
			arcs += it.outdegree();
			if (x % 4099 == 0) {
				std::vector<int32_t> a = g.successorArray(x);
				REQUIRE((int32_t)a.size() == it.outdegree() && it.outdegree() == g.outdegree(x));
				LazyIntIterator s = it.successors();
				for (size_t j = 0; j < a.size(); j++) REQUIRE(s.nextInt() == a[j]);
				REQUIRE(s.nextInt() == -1 && s.nextInt() == -1);
			}
		}
		REQUIRE(!it.hasNext() && arcs == g.numArcs());
		// split iterators return every node exactly once
		int32_t seen = 0;
		for (NodeIterator &s : g.splitNodeIterators(5)) while (s.hasNext()) REQUIRE(s.nextInt() == seen++);
		REQUIRE(seen == n);
		// ... and are drained by concurrent threads, as the reference's parallel consumers do (BVGraph.java:2471-2477,
		// ImmutableGraph.java:379-409): every split iterator decodes through a flyweight handle of its own
		{
			std::vector<NodeIterator> parts = g.splitNodeIterators(4);
			std::vector<uint32_t> fa(parts.size(), 1), fb(parts.size(), 0); // the affine map h -> a*h + b of each part
			std::vector<int64_t> parcs(parts.size(), 0);
			std::atomic<int> bad{ 0 };
			std::vector<std::thread> th;
			for (size_t k = 0; k < parts.size(); k++) th.emplace_back([&, k] {
				try {
					NodeIterator &s = parts[k];
					uint32_t a = 1, b = 0;
					while (s.hasNext()) {
						const int32_t x = s.nextInt();
						a *= 31u; b = b * 31u + (uint32_t)x;
						const int32_t *sc = s.successorArray();
						for (int32_t d = s.outdegree(); d-- != 0;) { a *= 31u; b = b * 31u + (uint32_t)sc[d]; }
						parcs[k] += s.outdegree();
					}
					fa[k] = a; fb[k] = b;
				} catch (...) { bad++; }
			});
			for (auto &t : th) t.join();
			REQUIRE(bad == 0);
			uint32_t h = (uint32_t)-1;
			int64_t tot = 0;
			for (size_t k = 0; k < parts.size(); k++) { h = fa[k] * h + fb[k]; tot += parcs[k]; }
			REQUIRE((int32_t)h == atoi(argv[2]) && tot == g.numArcs());
		}
		// the checksum scan (nothing materialised) agrees with the host-side fold
		{ uint64_t a = 0; REQUIRE(g.scanChecksum(0, n, -1, &a) == atoi(argv[2]) && (int64_t)a == g.numArcs()); }
		// BVGraph.store / EFGraph.store of the loaded graph, on the device: with the fixture's own parameters the reference's bytes come back;
		// the EFGraph loads through the same class and equals the graph it was made from
		if (argc > 4) {
			const std::string out = argv[4];
			BVGraph::store(g, out + "_bv", g.windowSize(), g.maxRefCount(), atoi(argv[5]), 3);
			auto slurp = [](const std::string &path) { std::vector<char> b; FILE *f = fopen(path.c_str(), "rb"); if (f) { char buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + k); fclose(f); } return b; };
			REQUIRE(slurp(out + "_bv.graph") == slurp(std::string(argv[1]) + ".graph") && !slurp(out + "_bv.graph").empty());
			REQUIRE(slurp(out + "_bv.offsets") == slurp(std::string(argv[1]) + ".offsets"));
			BVGraph::storeEF(g, out + "_ef");
			BVGraph e = BVGraph::load(out + "_ef");
			REQUIRE(e.isEFGraph() && !g.isEFGraph() && e.numArcs() == g.numArcs());
			REQUIRE(e.equals(g) && e.scanChecksum(0, n, -1) == atoi(argv[2]));
		}
		// flyweight copy
		BVGraph c = g.copy();
		REQUIRE(c.successorArray(n - 1) == g.successorArray(n - 1));
		// error behaviour
		bool threw = false;
		try { g.outdegree(n); } catch (const std::invalid_argument &) { threw = true; }
		REQUIRE(threw);
		threw = false;
		try { g.nodeIterator(n + 1); } catch (const std::invalid_argument &) { threw = true; }
		REQUIRE(threw);
		threw = false;
		try { BVGraph::load(std::string(argv[1]) + ".missing"); } catch (const io_error &) { threw = true; }
		REQUIRE(threw);
	} catch (const std::exception &e) { fprintf(stderr, "exception: %s\n", e.what()); return 1; }
	printf("host mirror ok\n");
	return 0;
}
