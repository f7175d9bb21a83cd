// bv_props.hpp -- the .properties file BVGraph.store writes (BVGraph.java:2558-2600), shared by the CPU writer
// (bvg_tools.cpp) and the device compressor's host side (bvg_store.cpp).
#pragma once
#include <stdint.h>

#include <cmath>
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

// the .properties file EFGraph.store writes (EFGraph.java:866-888)
inline bool write_ef(const std::string &path, int32_t n, uint64_t m, int32_t upper_bound, int log2_quantum, bool big_endian, uint64_t written_bits, uint64_t bits_outdegrees,
                     uint64_t bits_successors) {
	FILE *f = fopen(path.c_str(), "w");
	if (!f) return false;
	auto stirling = [](double v) { return v * std::log(v) - v + 0.5 * std::log(2 * 3.14159265358979323846 * v); }; // :804-806
	fprintf(f, "#EFGraph properties\n");
	fprintf(f, "nodes=%d\narcs=%llu\n", n, (unsigned long long)m);
	if (upper_bound != n) fprintf(f, "upperbound=%d\n", upper_bound);
	fprintf(f, "quantum=%llu\nbyteorder=%s\n", 1ull << log2_quantum, big_endian ? "BIG_ENDIAN" : "LITTLE_ENDIAN");
	fprintf(f, "bitsperlink=%s\n", fmt3(m ? (double)written_bits / m : 0).c_str());
	if (n > 0 && m > 0 && (double)n * n > (double)m)
		fprintf(f, "compratio=%s\n", fmt3(written_bits * std::log(2.0) / (stirling((double)n * n) - stirling((double)m) - stirling((double)n * n - (double)m))).c_str());
	fprintf(f, "bitspernode=%s\navgbitsforoutdegrees=%s\n", fmt3(n ? (double)written_bits / n : 0).c_str(), fmt3(n ? (double)bits_outdegrees / n : 0).c_str());
	fprintf(f, "bitsforoutdegrees=%llu\nbitsforsuccessors=%llu\n", (unsigned long long)bits_outdegrees, (unsigned long long)bits_successors);
	fprintf(f, "graphclass=it.unimi.dsi.webgraph.EFGraph\nversion=0\n");
	const bool ok = !ferror(f);
	return fclose(f) == 0 && ok;
}

} // namespace bvprops
