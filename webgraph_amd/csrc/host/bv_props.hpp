// bv_props.hpp -- the .properties file BVGraph.store writes (BVGraph.java:2558-2600), shared by the CPU writer
// (bvg_tools.cpp) and the device compressor's host side (bvg_store.cpp).
#pragma once
#include <stdint.h>

#include <cstdio>
#include <string>

namespace bvprops {

struct Counters {
	uint64_t written_bits, bits_outdegrees, bits_references, bits_blocks, bits_intervals, bits_residuals;
	uint64_t copied_arcs, intervalised_arcs, residual_arcs, tot_ref, tot_dist;
};

inline std::string flags_to_string(uint32_t flags) { // flags2String, BVGraph.java:1333-1345
	static const char *names[] = { "DEFAULT", "DELTA", "GAMMA", "GOLOMB", "SKEWED_GOLOMB", "UNARY", "ZETA", "NIBBLE" };
	static const char *fields[] = { "OUTDEGREES_", "BLOCKS_", "RESIDUALS_", "REFERENCES_", "BLOCK_COUNT_", "OFFSETS_" };
	std::string s;
	for (int f = 0; f < 6; f++) {
		unsigned c = (flags >> (4 * f)) & 0xF;
		if (c && c < 8) { if (!s.empty()) s += " | "; s += fields[f]; s += names[c]; }
	}
	return s;
}

inline std::string fmt3(double v) { // DecimalFormat("0.###")
	char b[64]; snprintf(b, sizeof b, "%.3f", v);
	std::string s(b);
	while (!s.empty() && s.back() == '0') s.pop_back();
	if (!s.empty() && s.back() == '.') s.pop_back();
	return s;
}

inline bool write(const std::string &path, int32_t n, uint64_t m, int window, int max_ref_count, int min_interval, int zeta_k, bool residuals_zeta, uint32_t flags, const Counters &st) {
	FILE *f = fopen(path.c_str(), "w");
	if (!f) return false;
	fprintf(f, "#BVGraph properties\n");
	fprintf(f, "nodes=%d\narcs=%llu\nwindowsize=%d\nmaxrefcount=%d\nminintervallength=%d\n", n, (unsigned long long)m, window, max_ref_count, min_interval);
	if (residuals_zeta) fprintf(f, "zetak=%d\n", zeta_k);
	fprintf(f, "compressionflags=%s\n", flags_to_string(flags).c_str());
	fprintf(f, "avgref=%s\navgdist=%s\n", fmt3(n ? (double)st.tot_ref / n : 0).c_str(), fmt3(n ? (double)st.tot_dist / n : 0).c_str());
	fprintf(f, "copiedarcs=%llu\nintervalisedarcs=%llu\nresidualarcs=%llu\n", (unsigned long long)st.copied_arcs, (unsigned long long)st.intervalised_arcs, (unsigned long long)st.residual_arcs);
	fprintf(f, "bitsperlink=%s\nbitspernode=%s\n", fmt3(m ? (double)st.written_bits / m : 0).c_str(), fmt3(n ? (double)st.written_bits / n : 0).c_str());
	fprintf(f, "bitsforoutdegrees=%llu\nbitsforreferences=%llu\nbitsforblocks=%llu\nbitsforresiduals=%llu\nbitsforintervals=%llu\n",
	        (unsigned long long)st.bits_outdegrees, (unsigned long long)st.bits_references, (unsigned long long)st.bits_blocks,
	        (unsigned long long)st.bits_residuals, (unsigned long long)st.bits_intervals);
	fprintf(f, "graphclass=it.unimi.dsi.webgraph.BVGraph\nversion=0\n");
	const bool ok = !ferror(f);
	return fclose(f) == 0 && ok;
}

} // namespace bvprops
