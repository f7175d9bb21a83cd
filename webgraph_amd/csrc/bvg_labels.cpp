// bvg_labels.cpp -- arc labels of a BitStreamArcLabelledImmutableGraph on the GPU (include/bvgpu.h).
//
// labelling/BitStreamArcLabelledImmutableGraph.java: the labels of all arcs, in enumeration order, are ONE bit stream
// (<basename>.labels); <basename>.labeloffsets holds gamma(0) followed by the gamma-coded bit length of every node's
// label list (:652-671, read back by LabelOffsetsLongIterator :340-358), i.e. exactly the structure of a BVGraph
// .offsets file.  Both are therefore decoded by the grid-wide gamma stream decoder of bv_offsets.hip: the offsets as
// running sums, gamma-coded labels (GammaCodedIntLabel.java:60-64) as plain values; fixed-width labels
// (FixedWidthIntLabel.java:70-73) need no decoding at all, label a of a range sits at startBit + a * width.
// FixedWidthIntListLabel (FixedWidthIntListLabel.java:107-112) stores a LIST per arc, gamma(length) + length x width bits:
// a lane per node walks its lists from the node's label offset (k_label_lists), the result is a CSR over the arcs.
#include "bv_host.hpp"
#include "bv_launch.hpp"

#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

struct bvg_labels {
	bvg_labels_info_t info{};
	uint32_t *d_words = nullptr; // .labels bytes as words + >= 8 zero words
	uint64_t nwords = 0;
	int64_t *d_off = nullptr;    // label offsets, int64[nodes + 1]
	std::vector<int64_t> h_off;
	int32_t *stage = nullptr;    // device staging for host outputs
	size_t stageCap = 0;
	mutable std::string err;
};

namespace {

int lfail(const bvg_labels *h, int rc, const std::string &msg) { if (h) h->err = msg; return rc; }

std::string trim(const std::string &s) {
	size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
	return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

// "<class>(KEY[,WIDTH])" -- Label.toSpec() (GammaCodedIntLabel.java:93-95, FixedWidthIntLabel.java:94-96)
int parse_labelspec(const std::string &spec, bvg_labels_info_t &info, std::string &err) {
	const size_t lp = spec.find('('), rp = spec.rfind(')');
	if (lp == std::string::npos || rp == std::string::npos || rp < lp) { err = "malformed labelspec: " + spec; return BVG_EIO; }
	const std::string cls = trim(spec.substr(0, lp)), args = spec.substr(lp + 1, rp - lp - 1);
	std::vector<std::string> a;
	for (size_t p = 0;;) { const size_t c = args.find(',', p); a.push_back(trim(args.substr(p, c == std::string::npos ? std::string::npos : c - p))); if (c == std::string::npos) break; p = c + 1; }
	const std::string simple = cls.substr(cls.rfind('.') == std::string::npos ? 0 : cls.rfind('.') + 1);
	if (simple == "GammaCodedIntLabel" && a.size() == 1) { info.kind = BVG_LABEL_GAMMA; info.width = -1; }
	else if (simple == "FixedWidthIntLabel" && a.size() == 2) {
		info.kind = BVG_LABEL_FIXED;
		char *end = nullptr;
		const long w = strtol(a[1].c_str(), &end, 10);
		if (!end || *end || w < 0 || w > 32) { err = "Width out of range: " + a[1]; return BVG_EARG; } // FixedWidthIntLabel.java:45
		info.width = (int32_t)w;
	} else if (simple == "FixedWidthIntListLabel" && a.size() == 2) {
		info.kind = BVG_LABEL_FIXED_LIST;
		char *end = nullptr;
		const long w = strtol(a[1].c_str(), &end, 10);
		if (!end || *end || w < 0 || w > 32) { err = "Width out of range: " + a[1]; return BVG_EARG; } // FixedWidthIntListLabel.java:48
		info.width = (int32_t)w;
	} else { err = "unsupported label class: " + cls; return BVG_EUNSUPPORTED; }
	strncpy(info.key, a[0].c_str(), sizeof info.key - 1);
	return BVG_OK;
}

int parse_label_properties(const std::string &basename, bvg_labels_info_t &info, std::string &err) {
	std::vector<std::pair<std::string, std::string>> kv;
	if (!bvh::load_properties(basename + ".properties", kv)) { err = "cannot read " + basename + ".properties"; return BVG_EIO; }
	std::string under, spec, gclass;
	for (auto &p : kv) { if (p.first == "underlyinggraph") under = p.second; else if (p.first == "labelspec") spec = p.second; else if (p.first == "graphclass") gclass = p.second; }
	if (gclass.find("BitStreamArcLabelledImmutableGraph") == std::string::npos) { err = "graphclass is not BitStreamArcLabelledImmutableGraph: " + gclass; return BVG_EIO; }
	if (under.empty()) { err = "The property file for " + basename + " does not contain an underlying graph basename"; return BVG_EIO; } // :391
	if (spec.empty()) { err = "The property file for " + basename + " does not contain a label specification"; return BVG_EIO; }      // :409
	// relative to the property file's directory unless absolute (:393-395)
	if (under[0] != '/') { const size_t sl = basename.rfind('/'); if (sl != std::string::npos) under = basename.substr(0, sl + 1) + under; }
	memset(&info, 0, sizeof info);
	info.device = -1;
	strncpy(info.underlying, under.c_str(), sizeof info.underlying - 1);
	return parse_labelspec(spec, info, err);
}

} // namespace

extern "C" int bvg_labels_parse_properties(const char *basename, bvg_labels_info_t *out, char *errbuf, size_t errlen) {
	if (!basename || !out) return BVG_EARG;
	std::string err;
	const int rc = parse_label_properties(basename, *out, err);
	if (errbuf && errlen) { strncpy(errbuf, err.c_str(), errlen - 1); errbuf[errlen - 1] = 0; }
	return rc;
}

extern "C" const char *bvg_labels_last_error(const bvg_labels_t *h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int bvg_labels_info(const bvg_labels_t *h, bvg_labels_info_t *out) {
	if (!h || !out) return BVG_EARG;
	*out = h->info;
	return BVG_OK;
}

extern "C" void bvg_labels_close(bvg_labels_t *h) {
	if (!h) return;
	if (h->info.device >= 0) (void)hipSetDevice(h->info.device);
	for (void *p : { (void *)h->d_words, (void *)h->d_off, (void *)h->stage }) if (p) (void)hipFree(p);
	delete h;
}

extern "C" int bvg_labels_open(const char *basename, int32_t nodes, int device, bvg_labels_t **out) {
	if (!basename || !out || nodes < 0) return BVG_EARG;
	auto *h = new bvg_labels();
	*out = h; // returned even on failure so that bvg_labels_last_error works; the caller still closes it
	std::string err;
	int rc = parse_label_properties(basename, h->info, err);
	if (rc) return lfail(h, rc, err);
	h->info.nodes = nodes;
	std::vector<uint8_t> lab, offs;
	if (!bvh::read_file(std::string(basename) + ".labels", lab, err)) return lfail(h, BVG_EIO, err);
	if (!bvh::read_file(std::string(basename) + ".labeloffsets", offs, err)) return lfail(h, BVG_EIO, err);
	h->info.labels_bytes = lab.size();
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return lfail(h, BVG_EHIP, "no HIP device available (libbvgpu has no CPU fallback)");
	if (device < 0 || device >= ndev) return lfail(h, BVG_EARG, "no such HIP device");
	if (hipSetDevice(device) != hipSuccess) return lfail(h, BVG_EHIP, "hipSetDevice failed");
	h->info.device = device;
	h->nwords = (lab.size() + 3) / 4;
	const size_t padded = (size_t)(h->nwords + 8) * 4;
	if (hipMalloc((void **)&h->d_words, padded) != hipSuccess || hipMalloc((void **)&h->d_off, sizeof(int64_t) * ((size_t)nodes + 1)) != hipSuccess) return lfail(h, BVG_ENOMEM, "device allocation failed");
	if (hipMemset(h->d_words, 0, padded) != hipSuccess || (!lab.empty() && hipMemcpy(h->d_words, lab.data(), lab.size(), hipMemcpyHostToDevice) != hipSuccess)) return lfail(h, BVG_EHIP, "staging the label stream failed");
	// label offsets: the same gamma gap stream as a BVGraph .offsets file -> the same device decoder, host decoder as fallback
	h->h_off.resize((size_t)nodes + 1);
	bool onDevice = false;
	if (!offs.empty()) {
		const uint64_t ow = (offs.size() + 3) / 4;
		uint32_t *d_ow = nullptr;
		if (hipMalloc((void **)&d_ow, (size_t)(ow + 8) * 4) == hipSuccess) {
			if (hipMemset(d_ow, 0, (size_t)(ow + 8) * 4) == hipSuccess && hipMemcpy(d_ow, offs.data(), offs.size(), hipMemcpyHostToDevice) == hipSuccess &&
			    bv::offsets_decode_device(d_ow, ow, (uint64_t)offs.size() * 8, nodes, h->d_off, nullptr) == 0 &&
			    hipMemcpy(h->h_off.data(), h->d_off, sizeof(int64_t) * h->h_off.size(), hipMemcpyDeviceToHost) == hipSuccess)
				onDevice = true;
			(void)hipFree(d_ow);
		}
		(void)hipGetLastError();
	}
	if (!onDevice) {
		rc = bvh::decode_offsets(offs.data(), offs.size(), nodes, BVG_GAMMA, h->h_off.data());
		if (rc) return lfail(h, rc, std::string("cannot decode ") + basename + ".labeloffsets");
		if (hipMemcpy(h->d_off, h->h_off.data(), sizeof(int64_t) * h->h_off.size(), hipMemcpyHostToDevice) != hipSuccess) return lfail(h, BVG_EHIP, "staging the label offsets failed");
	}
	for (size_t i = 1; i < h->h_off.size(); i++) if (h->h_off[i] < h->h_off[i - 1]) return lfail(h, BVG_EIO, "label offsets are not monotone");
	if ((uint64_t)h->h_off.back() > (uint64_t)lab.size() * 8) return lfail(h, BVG_EIO, "label offsets run past the end of the .labels file");
	h->info.labels_bits = (uint64_t)h->h_off.back();
	return BVG_OK;
}

extern "C" int bvg_labels_decode_range(bvg_labels_t *h, int32_t from, int32_t to, uint64_t arcs, int32_t *labels, int flags) {
	if (!h || !h->d_words || from < 0 || to < from || to > h->info.nodes || (arcs && !labels)) return lfail(h, BVG_EARG, "Node index out of range"); // as BVG:1165
	if (h->info.kind == BVG_LABEL_FIXED_LIST) return lfail(h, BVG_EUNSUPPORTED, "FixedWidthIntListLabel carries a list per arc: use bvg_labels_decode_lists");
	if (hipSetDevice(h->info.device) != hipSuccess) return lfail(h, BVG_EHIP, "hipSetDevice failed");
	if (arcs == 0) return h->h_off[to] == h->h_off[from] ? BVG_OK : lfail(h, BVG_EFORMAT, "the label stream holds labels for a range without arcs");
	if (arcs > 0x7fffffffffffull) return lfail(h, BVG_EARG, "too many arcs");
	const bool dev = (flags & BVG_OUT_DEVICE) != 0;
	int32_t *d_out = labels;
	if (!dev) {
		if (h->stageCap < arcs) {
			if (h->stage) (void)hipFree(h->stage);
			h->stage = nullptr; h->stageCap = 0;
			if (hipMalloc((void **)&h->stage, sizeof(int32_t) * (size_t)arcs) != hipSuccess) return lfail(h, BVG_ENOMEM, "staging allocation failed");
			h->stageCap = (size_t)arcs;
		}
		d_out = h->stage;
	}
	const uint64_t b0 = (uint64_t)h->h_off[from], b1 = (uint64_t)h->h_off[to];
	int rc;
	if (h->info.kind == BVG_LABEL_FIXED) {
		if ((uint64_t)h->info.width * arcs != b1 - b0) return lfail(h, BVG_EFORMAT, "the label stream does not hold one fixed-width label per arc");
		rc = bv::fixed_labels_decode_device(h->d_words, h->nwords, b0, h->info.width, (int64_t)arcs, d_out, nullptr);
	} else {
		rc = bv::gamma_labels_decode_device(h->d_words, h->nwords, b0, b1, (int64_t)arcs, d_out, nullptr);
		if (rc) { // the device decoder gives up on streams that do not re-synchronise -- a long stretch of EQUAL labels is one (every label 5: 00110 00110 ...; a chain that starts one
			// bit late never meets the true one) -- and on streams that do not hold `arcs` codes: the stretch is walked on the host, which tells the two apart
			(void)hipGetLastError();
			const uint64_t byteLo = b0 >> 3, byteHi = (b1 + 7) >> 3;
			std::vector<uint8_t> bytes((size_t)(byteHi - byteLo) + 1);
			std::vector<int32_t> vals((size_t)arcs);
			if (byteHi > byteLo && hipMemcpy(bytes.data(), (const uint8_t *)h->d_words + byteLo, (size_t)(byteHi - byteLo), hipMemcpyDeviceToHost) != hipSuccess) return lfail(h, BVG_EHIP, "copying the label stream back failed");
			rc = bvh::decode_gammas(bytes.data(), b0 - byteLo * 8, b1 - byteLo * 8, (int64_t)arcs, vals.data());
			if (!rc && arcs && hipMemcpy(d_out, vals.data(), sizeof(int32_t) * (size_t)arcs, hipMemcpyHostToDevice) != hipSuccess) return lfail(h, BVG_EHIP, "staging the labels failed");
		}
	}
	if (rc) { (void)hipGetLastError(); return lfail(h, BVG_EFORMAT, "the label stream does not hold one label per arc of the range"); }
	if (!dev && hipMemcpy(labels, d_out, sizeof(int32_t) * (size_t)arcs, hipMemcpyDeviceToHost) != hipSuccess) return lfail(h, BVG_EHIP, "copying the labels back failed");
	return BVG_OK;
}

extern "C" int bvg_labels_decode_lists(bvg_labels_t *h, int32_t from, int32_t to, uint64_t arcs, int64_t *list_ptr, int32_t *values, uint64_t values_cap,
                                       uint64_t *nvalues, int flags) {
	if (nvalues) *nvalues = 0;
	if (!h || !h->d_words || from < 0 || to < from || to > h->info.nodes || !list_ptr || (values_cap && !values)) return lfail(h, BVG_EARG, "Node index out of range");
	if (h->info.kind != BVG_LABEL_FIXED_LIST) return lfail(h, BVG_EUNSUPPORTED, "the label class carries one int per arc: use bvg_labels_decode_range");
	if (arcs > 0x7fffffffffffull) return lfail(h, BVG_EARG, "too many arcs");
	if (hipSetDevice(h->info.device) != hipSuccess) return lfail(h, BVG_EHIP, "hipSetDevice failed");
	const bool dev = (flags & BVG_OUT_DEVICE) != 0;
	if (to == from || arcs == 0) {
		if (h->h_off[to] != h->h_off[from]) return lfail(h, BVG_EFORMAT, "the label stream holds labels for a range without arcs");
		if (arcs) return lfail(h, BVG_EFORMAT, "the label stream does not hold one list per arc of the range");
		const int64_t zero = 0;
		if (dev) { if (hipMemcpy(list_ptr, &zero, sizeof zero, hipMemcpyHostToDevice) != hipSuccess) return lfail(h, BVG_EHIP, "writing the list pointer failed"); }
		else list_ptr[0] = 0;
		return BVG_OK;
	}
	int64_t *d_lp = list_ptr;
	int32_t *d_val = values;
	auto release = [&]() { if (!dev) { if (d_lp) (void)hipFree(d_lp); if (d_val) (void)hipFree(d_val); } };
	if (!dev) {
		d_lp = nullptr; d_val = nullptr;
		if (hipMalloc((void **)&d_lp, sizeof(int64_t) * ((size_t)arcs + 1)) != hipSuccess || (values_cap && hipMalloc((void **)&d_val, sizeof(int32_t) * (size_t)values_cap) != hipSuccess)) {
			release();
			(void)hipGetLastError();
			return lfail(h, BVG_ENOMEM, "staging allocation failed");
		}
	}
	uint64_t nv = 0;
	const int rc = bv::label_lists_decode_device(h->d_words, h->nwords, h->d_off, from, to - from, h->info.width, arcs, d_lp, d_val, values_cap, &nv, nullptr);
	if (nvalues) *nvalues = nv;
	if (rc) {
		release();
		(void)hipGetLastError();
		if (rc == -2) return lfail(h, BVG_ECAP, "the value buffer is too small for the lists of the range");
		if (rc == -3) return lfail(h, BVG_EHIP, "decoding the label lists failed");
		return lfail(h, BVG_EFORMAT, "the label stream does not hold one list per arc of the range");
	}
	int out = BVG_OK;
	if (!dev && (hipMemcpy(list_ptr, d_lp, sizeof(int64_t) * ((size_t)arcs + 1), hipMemcpyDeviceToHost) != hipSuccess ||
	             (nv && hipMemcpy(values, d_val, sizeof(int32_t) * (size_t)nv, hipMemcpyDeviceToHost) != hipSuccess)))
		out = lfail(h, BVG_EHIP, "copying the label lists back failed");
	release();
	return out;
}
