// bv_props.hpp -- the .properties file BVGraph.store writes (BVGraph.java:2558-2632), shared by the CPU writer
// (bvg_tools.cpp) and the device compressor's host side (bvg_store.cpp).
#pragma once
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace bvprops {

struct Counters {
	uint64_t written_bits, bits_outdegrees, bits_references, bits_blocks, bits_intervals, bits_residuals;
	uint64_t copied_arcs, intervalised_arcs, residual_arcs, tot_ref, tot_dist;
	// gaps binned by their most significant bit (updateBins, BVGraph.java:1940-1944): over every successor list, and over every list of residuals
	uint64_t successor_gap_bins[32], residual_gap_bins[32];
};

// updateBins' bin of one element: the first of a list by int2nat(first - node) (nothing when that is 0), a later one by its distance from its predecessor
inline int gap_bin(bool first, int64_t node_or_prev, int64_t v) {
	const int64_t d = v - node_or_prev;
	const uint64_t g = first ? (d >= 0 ? (uint64_t)d << 1 : (((uint64_t)-d) << 1) - 1) : (uint64_t)d;
	return g == 0 ? -1 : 63 - __builtin_clzll(g);
}

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

// Double.toString for the magnitudes an average log-gap has (10^-3 <= v < 10^7: plain decimals, the shortest that read back as v, at least one after the point)
inline std::string java_double(double v) {
	if (v == 0) return "0.0";
	char b[64];
	for (int prec = 1; prec <= 17; prec++) { snprintf(b, sizeof b, "%.*g", prec, v); if (strtod(b, nullptr) == v) break; }
	std::string s(b);
	const size_t e = s.find('e');
	if (e != std::string::npos) { // outside the plain range: Java's d.dddE[-]x
		std::string m = s.substr(0, e);
		int ex = atoi(s.c_str() + e + 1);
		if (m.find('.') == std::string::npos) m += ".0";
		return m + "E" + std::to_string(ex);
	}
	if (s.find('.') == std::string::npos) s += ".0";
	return s;
}

// the three keys of one histogram (:2592-2632): <name>expstats = the bins up to the last one in use, <name>avggap = sum (3 * 2^i - 1) * bin[i] / (2 * gaps) with three decimals
// (BigDecimal.divide(.., 3, HALF_EVEN)), <name>avgloggap = sum (log2(3 * 2^i + 1) - 1) * bin[i] / gaps as Double.toString prints it
inline void write_gap_stats(FILE *f, const char *name, const uint64_t *bins) {
	int l = 31;
	while (l >= 0 && bins[l] == 0) l--;
	std::string s;
	unsigned __int128 tot = 0;
	double totLog = 0;
	uint64_t gaps = 0, g = 1;
	for (int i = 0; i <= l; i++, g *= 2) {
		if (i) s += ',';
		s += std::to_string((unsigned long long)bins[i]);
		gaps += bins[i];
		tot += (unsigned __int128)(g * 2 + g - 1) * bins[i];
		totLog += (std::log((double)(g * 2 + g + 1)) / 0.6931471805599453 - 1) * (double)bins[i];
	}
	fprintf(f, "%sexpstats=%s\n", name, s.c_str());
	if (gaps == 0) { fprintf(f, "%savggap=0\n%savgloggap=0\n", name, name); return; }
	const unsigned __int128 den = (unsigned __int128)gaps * 2, num = tot * 1000;
	unsigned __int128 q = num / den;
	const unsigned __int128 r2 = (num % den) * 2;
	if (r2 > den || (r2 == den && (q & 1))) q++; // HALF_EVEN
	fprintf(f, "%savggap=%llu.%03llu\n", name, (unsigned long long)(q / 1000), (unsigned long long)(q % 1000));
	fprintf(f, "%savgloggap=%s\n", name, java_double(totLog / (double)gaps).c_str());
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
	if (n > 0 && m > 0 && (double)n * n > (double)m) { // (the reference prints NaN otherwise: Stirling of 0)
		auto stirling = [](double v) { return v * std::log(v) - v + 0.5 * std::log(2 * 3.14159265358979323846 * v); }; // :2652-2654
		fprintf(f, "compratio=%s\n", fmt3(st.written_bits * std::log(2.0) / (stirling((double)n * n) - stirling((double)m) - stirling((double)n * n - (double)m))).c_str());
	}
	fprintf(f, "avgbitsforoutdegrees=%s\navgbitsforreferences=%s\navgbitsforblocks=%s\navgbitsforresiduals=%s\navgbitsforintervals=%s\n",
	        fmt3(n ? (double)st.bits_outdegrees / n : 0).c_str(), fmt3(n ? (double)st.bits_references / n : 0).c_str(), fmt3(n ? (double)st.bits_blocks / n : 0).c_str(),
	        fmt3(n ? (double)st.bits_residuals / n : 0).c_str(), fmt3(n ? (double)st.bits_intervals / n : 0).c_str());
	fprintf(f, "bitsforoutdegrees=%llu\nbitsforreferences=%llu\nbitsforblocks=%llu\nbitsforresiduals=%llu\nbitsforintervals=%llu\n",
	        (unsigned long long)st.bits_outdegrees, (unsigned long long)st.bits_references, (unsigned long long)st.bits_blocks,
	        (unsigned long long)st.bits_residuals, (unsigned long long)st.bits_intervals);
	write_gap_stats(f, "successor", st.successor_gap_bins);
	write_gap_stats(f, "residual", st.residual_gap_bins);
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
