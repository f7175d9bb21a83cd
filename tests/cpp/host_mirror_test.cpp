// Drives the C++ host mirror (webgraph_amd/host/bvgraph.hpp) the way the reference's tests drive ImmutableGraph
// (WebGraphTestCase.assertGraph / BVGraphTest.testLarge).  Usage: host_mirror_test <basename> <expected hashCode> <expected arcs>
#include "../../webgraph_amd/host/bvgraph.hpp"

#include <cstdio>
#include <cstdlib>

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
	if (argc < 4) { fprintf(stderr, "usage: %s basename hashcode arcs\n", argv[0]); return 2; }
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
