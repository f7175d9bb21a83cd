// bv_host.hpp -- host-side pieces of libbvgpu.so that need no GPU: .properties parsing, flag strings,
// .offsets decoding, file slurping.  (BVGraph.loadInternal, BVG:1516-1609.)
#pragma once
#include "../../include/bvgpu.h"

#include <string>
#include <vector>

namespace bvh {

// java.util.Properties subset: key/value separated by '=', ':' or blanks; '#'/'!' comments; '\' escapes and
// line continuations.  Returns false if the file cannot be read.
bool load_properties(const std::string &path, std::vector<std::pair<std::string, std::string>> &out);

// string2Flags (BVG:1352-1366).  Returns -1 for a name that is not a public BVGraph constant (BVG:475-523).
int64_t flags_from_string(const std::string &s);

// BVG:1528-1543 + setFlags BVG:1317-1325.  Fills `info` (device = -1) or returns a negative bvg_status with a message.
int parse_properties(const std::string &basename, bvg_info_t &info, std::string &err);

// OffsetsLongIterator BVG:907-935
int decode_offsets(const uint8_t *p, size_t len, int32_t nodes, int coding, int64_t *out);

bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err);
int decode_gammas(const uint8_t *p, uint64_t lo, uint64_t hi, int64_t count, int32_t *out); // `count` gamma codes in bits [lo, hi), one after the other

} // namespace bvh
