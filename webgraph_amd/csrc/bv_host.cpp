// bv_host.cpp -- see bv_host.hpp.  Host-only logic of the load path (BVGraph.loadInternal, BVG:1516-1609).
#include "bv_host.hpp"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bvh {

bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err) {
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) { err = "cannot open " + path + ": " + strerror(errno); return false; }
	fseek(f, 0, SEEK_END);
	long long sz = ftello(f);
	fseek(f, 0, SEEK_SET);
	if (sz < 0) { fclose(f); err = "cannot size " + path; return false; }
	out.resize((size_t)sz);
	size_t got = sz ? fread(out.data(), 1, (size_t)sz, f) : 0;
	fclose(f);
	if (got != (size_t)sz) { err = "short read on " + path; return false; }
	return true;
}

bool load_properties(const std::string &path, std::vector<std::pair<std::string, std::string>> &out) {
	std::vector<uint8_t> raw;
	std::string err;
	if (!read_file(path, raw, err)) return false;
	// physical lines -> logical lines: a line ending in an odd number of backslashes continues on the next one
	std::vector<std::string> lines;
	{
		std::string cur, line;
		bool cont = false;
		auto flush_line = [&]() {
			if (!line.empty() && line.back() == '\r') line.pop_back();
			size_t p0 = 0;
			while (p0 < line.size() && (line[p0] == ' ' || line[p0] == '\t' || line[p0] == '\f')) p0++;
			line.erase(0, p0);
			if (!cont && (line.empty() || line[0] == '#' || line[0] == '!')) { line.clear(); return; }
			size_t bs = 0;
			while (bs < line.size() && line[line.size() - 1 - bs] == '\\') bs++;
			if (bs & 1) { line.pop_back(); cur += line; cont = true; }
			else { cur += line; lines.push_back(cur); cur.clear(); cont = false; }
			line.clear();
		};
		for (uint8_t c : raw) { if (c == '\n') flush_line(); else line.push_back((char)c); }
		flush_line();
		if (cont && !cur.empty()) lines.push_back(cur);
	}
	auto unesc = [](const std::string &s) {
		std::string o;
		for (size_t k = 0; k < s.size(); k++) {
			if (s[k] != '\\' || k + 1 == s.size()) { o.push_back(s[k]); continue; }
			char c = s[++k];
			switch (c) {
			case 't': o.push_back('\t'); break;
			case 'n': o.push_back('\n'); break;
			case 'r': o.push_back('\r'); break;
			case 'f': o.push_back('\f'); break;
			case 'u': if (k + 4 < s.size()) { unsigned v = (unsigned)strtoul(s.substr(k + 1, 4).c_str(), nullptr, 16); o.push_back((char)(v & 0xff)); k += 4; } break;
			default: o.push_back(c);
			}
		}
		return o;
	};
	for (const std::string &l : lines) {
		size_t k = 0;
		// key: up to the first unescaped '=', ':' or blank
		while (k < l.size()) {
			if (l[k] == '\\') { k += 2; continue; }
			if (l[k] == '=' || l[k] == ':' || l[k] == ' ' || l[k] == '\t' || l[k] == '\f') break;
			k++;
		}
		if (k > l.size()) k = l.size();
		std::string key = unesc(l.substr(0, k));
		while (k < l.size() && (l[k] == ' ' || l[k] == '\t' || l[k] == '\f')) k++;
		if (k < l.size() && (l[k] == '=' || l[k] == ':')) k++;
		while (k < l.size() && (l[k] == ' ' || l[k] == '\t' || l[k] == '\f')) k++;
		out.emplace_back(key, unesc(l.substr(k)));
	}
	return true;
}

int64_t flags_from_string(const std::string &s) {
	// the public flag constants of BVGraph (BVG:475-523); value = coding id << field shift
	static const struct { const char *name; uint32_t v; } K[] = {
		{ "OUTDEGREES_GAMMA", BVG_GAMMA }, { "OUTDEGREES_DELTA", BVG_DELTA },
		{ "BLOCKS_GAMMA", BVG_GAMMA << 4 }, { "BLOCKS_DELTA", BVG_DELTA << 4 },
		{ "RESIDUALS_GAMMA", BVG_GAMMA << 8 }, { "RESIDUALS_ZETA", BVG_ZETA << 8 }, { "RESIDUALS_DELTA", BVG_DELTA << 8 },
		{ "RESIDUALS_NIBBLE", BVG_NIBBLE << 8 }, { "RESIDUALS_GOLOMB", BVG_GOLOMB << 8 },
		{ "REFERENCES_GAMMA", BVG_GAMMA << 12 }, { "REFERENCES_DELTA", BVG_DELTA << 12 }, { "REFERENCES_UNARY", BVG_UNARY << 12 },
		{ "BLOCK_COUNT_GAMMA", BVG_GAMMA << 16 }, { "BLOCK_COUNT_DELTA", BVG_DELTA << 16 }, { "BLOCK_COUNT_UNARY", BVG_UNARY << 16 },
		{ "OFFSETS_GAMMA", BVG_GAMMA << 20 }, { "OFFSETS_DELTA", BVG_DELTA << 20 },
	};
	uint32_t flags = 0;
	if (s.empty()) return 0;
	size_t p = 0;
	while (p <= s.size()) {
		size_t q = s.find('|', p);
		if (q == std::string::npos) q = s.size();
		std::string tok = s.substr(p, q - p);
		size_t a = tok.find_first_not_of(" \t\r\n\f"), b = tok.find_last_not_of(" \t\r\n\f");
		tok = a == std::string::npos ? "" : tok.substr(a, b - a + 1);
		bool found = false;
		for (const auto &k : K) if (tok == k.name) { flags |= k.v; found = true; break; }
		if (!found) return -1; // IOException("Compression flag ... unknown"), BVG:1361
		p = q + 1;
	}
	return flags;
}

static bool parse_ll(const std::string &s, long long &v) {
	if (s.empty()) return false;
	char *end = nullptr;
	errno = 0;
	v = strtoll(s.c_str(), &end, 10);
	return errno == 0 && end && *end == 0;
}

int parse_properties(const std::string &basename, bvg_info_t &info, std::string &err) {
	std::vector<std::pair<std::string, std::string>> kv;
	if (!load_properties(basename + ".properties", kv)) { err = "cannot read " + basename + ".properties"; return BVG_EIO; }
	auto get = [&](const char *k, std::string &v) { bool f = false; for (auto &p : kv) if (p.first == k) { v = p.second; f = true; } return f; }; // last wins
	std::string v;
	memset(&info, 0, sizeof info);
	info.device = -1;
	if (!get("graphclass", v)) { err = "missing graphclass"; return BVG_EUNSUPPORTED; }
	{ // BVG:1528: the big spelling is accepted too
		std::string c = v;
		const std::string big = "it.unimi.dsi.big.webgraph";
		size_t at = c.find(big);
		if (at != std::string::npos) c.replace(at, big.size(), "it.unimi.dsi.webgraph");
		// the binding's own class (graphclass=it.unimi.dsi.webgraph.gpu.GpuBVGraph makes ImmutableGraph.load reflect on
		// it): the files are a BVGraph's or an EFGraph's -- an EFGraph has a byteorder, a BVGraph a windowsize
		if (c == "it.unimi.dsi.webgraph.gpu.GpuBVGraph") { std::string w; c = get("byteorder", w) ? "it.unimi.dsi.webgraph.EFGraph" : "it.unimi.dsi.webgraph.BVGraph"; }
		if (c == "it.unimi.dsi.webgraph.EFGraph") { // EFGraph.loadInternal, EFGraph.java:709-750
			long long t;
			std::string w;
			info.format = BVG_FORMAT_EF;
			if (!get("version", w)) { err = "Missing format version information"; return BVG_EUNSUPPORTED; }
			if (!parse_ll(w, t) || t > 0) { err = "This graph uses format " + w + ", but this library understands only graphs up to format 0"; return BVG_EUNSUPPORTED; }
			if (!get("nodes", w) || !parse_ll(w, t) || t < 0) { err = "bad or missing nodes"; return BVG_EUNSUPPORTED; }
			if (t > 0x7fffffffLL) { err = "cannot handle graphs with " + w + " (>=2^31) nodes"; return BVG_EARG; }
			info.nodes = (int32_t)t;
			if (!get("arcs", w) || !parse_ll(w, t) || t < 0) { err = "bad or missing arcs"; return BVG_EUNSUPPORTED; }
			info.arcs = t;
			info.ef_upper_bound = info.nodes;
			if (get("upperbound", w)) { if (!parse_ll(w, t) || t < info.nodes || t > 0x7fffffffLL) { err = "bad upperbound"; return BVG_EUNSUPPORTED; } info.ef_upper_bound = (int32_t)t; }
			if (!get("quantum", w)) { err = "missing quantum"; return BVG_EUNSUPPORTED; }
			if (!parse_ll(w, t) || t < 1 || (t & (t - 1))) { err = "Illegal quantum (must be a power of 2): " + w; return BVG_EARG; }
			info.ef_log2_quantum = 63 - __builtin_clzll((unsigned long long)t);
			if (!get("byteorder", w)) { err = "missing byteorder"; return BVG_EUNSUPPORTED; }
			if (w != "BIG_ENDIAN" && w != "LITTLE_ENDIAN") { err = "Unknown byte order " + w; return BVG_EARG; }
			info.ef_big_endian = w == "BIG_ENDIAN";
			info.offset_coding = BVG_DELTA; // offsets.writeLongDelta, :830, :855
			return BVG_OK;
		}
		if (c != "it.unimi.dsi.webgraph.BVGraph") { err = "this class cannot load a graph stored using class \"" + v + "\""; return BVG_EUNSUPPORTED; }
	}
	std::string fs;
	get("compressionflags", fs);
	int64_t fl = flags_from_string(fs);
	if (fl < 0) { err = "Compression flag unknown in \"" + fs + "\""; return BVG_EUNSUPPORTED; }
	info.flags = (uint32_t)fl;
	long long t;
	if (!get("version", v)) { err = "Missing format version information"; return BVG_EUNSUPPORTED; } // BVG:1533
	if (!parse_ll(v, t)) { err = "bad version"; return BVG_EUNSUPPORTED; }
	if (t > 0) { err = "This graph uses format " + v + ", but this library understands only graphs up to format 0"; return BVG_EUNSUPPORTED; } // BVG:1534
	if (!get("nodes", v) || !parse_ll(v, t) || t < 0) { err = "bad or missing nodes"; return BVG_EUNSUPPORTED; }
	if (t > 0x7fffffffLL) { err = "cannot handle graphs with " + v + " (>=2^31) nodes"; return BVG_EARG; } // BVG:1537
	info.nodes = (int32_t)t;
	if (!get("arcs", v) || !parse_ll(v, t) || t < 0) { err = "bad or missing arcs"; return BVG_EUNSUPPORTED; }
	info.arcs = t;
	if (!get("windowsize", v) || !parse_ll(v, t) || t < 0 || t > 65535) { err = "bad or missing windowsize"; return BVG_EUNSUPPORTED; }
	info.window_size = (int32_t)t;
	if (!get("maxrefcount", v) || !parse_ll(v, t)) { err = "bad or missing maxrefcount"; return BVG_EUNSUPPORTED; }
	info.max_ref_count = (int32_t)std::min<long long>(t, 0x7fffffffLL);
	if (!get("minintervallength", v) || !parse_ll(v, t) || t < 0 || t > 0x7fffffffLL) { err = "bad or missing minintervallength"; return BVG_EUNSUPPORTED; }
	info.min_interval_length = (int32_t)t;
	info.zeta_k = 3; // DEFAULT_ZETA_K, BVG:469-472
	if (get("zetak", v)) { if (!parse_ll(v, t) || t < 1 || t > 64) { err = "bad zetak"; return BVG_EUNSUPPORTED; } info.zeta_k = (int32_t)t; }
	// setFlags, BVG:1317-1325: an unspecified slot keeps the default
	const uint32_t f = info.flags;
	info.outdegree_coding = (f & 0xF) ? (f & 0xF) : BVG_GAMMA;
	info.block_coding = ((f >> 4) & 0xF) ? ((f >> 4) & 0xF) : BVG_GAMMA;
	info.residual_coding = ((f >> 8) & 0xF) ? ((f >> 8) & 0xF) : BVG_ZETA;
	info.reference_coding = ((f >> 12) & 0xF) ? ((f >> 12) & 0xF) : BVG_UNARY;
	info.block_count_coding = ((f >> 16) & 0xF) ? ((f >> 16) & 0xF) : BVG_GAMMA;
	info.offset_coding = ((f >> 20) & 0xF) ? ((f >> 20) & 0xF) : BVG_GAMMA;
	return BVG_OK;
}

namespace {
// tiny sequential MSB-first reader for the (one-off) offsets decode
struct HostBits {
	const uint8_t *p; uint64_t nbits; uint64_t pos = 0; bool bad = false;
	inline int bit() { if (pos >= nbits) { bad = true; return 1; } int b = (p[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; return b; }
	inline uint64_t unary() {
		uint64_t z = 0;
		// byte-at-a-time fast path
		while (pos < nbits) {
			unsigned rem = 8 - (unsigned)(pos & 7);
			unsigned byte = p[pos >> 3] & ((1u << rem) - 1);
			if (byte) { unsigned lead = (unsigned)__builtin_clz(byte) - (32 - rem); z += lead; pos += lead + 1; if (pos > nbits) bad = true; return z; }
			z += rem; pos += rem;
		}
		bad = true;
		return z;
	}
	inline uint64_t take(unsigned n) { uint64_t v = 0; while (n--) v = (v << 1) | (uint64_t)bit(); return v; }
	inline uint64_t gamma() { uint64_t m = unary(); if (m > 63) { bad = true; return 0; } return (((uint64_t)1 << m) | take((unsigned)m)) - 1; }
	inline uint64_t delta() { uint64_t m = gamma(); if (m > 63) { bad = true; return 0; } return (((uint64_t)1 << m) | take((unsigned)m)) - 1; }
};
} // namespace

int decode_offsets(const uint8_t *p, size_t len, int32_t nodes, int coding, int64_t *out) {
	if (coding != BVG_GAMMA && coding != BVG_DELTA) return BVG_EUNSUPPORTED; // BVG:635
	HostBits hb{ p, (uint64_t)len * 8 };
	int64_t off = 0;
	for (int64_t i = 0; i <= nodes; i++) {
		off += (int64_t)(coding == BVG_GAMMA ? hb.gamma() : hb.delta());
		if (hb.bad) return BVG_EIO;
		out[i] = off;
	}
	return BVG_OK;
}

// `count` gamma codes in bits [lo, hi) of p, one after the other (GammaCodedIntLabel.fromBitStream, labelling/GammaCodedIntLabel.java:60-64): BVG_OK when exactly the count
// fits the stretch, BVG_EFORMAT otherwise
int decode_gammas(const uint8_t *p, uint64_t lo, uint64_t hi, int64_t count, int32_t *out) {
	HostBits hb{ p, hi };
	hb.pos = lo;
	for (int64_t i = 0; i < count; i++) {
		const uint64_t v = hb.gamma();
		if (hb.bad || v > 0x7fffffffull) return BVG_EFORMAT;
		out[i] = (int32_t)v;
	}
	return hb.pos == hi ? BVG_OK : BVG_EFORMAT;
}

} // namespace bvh
