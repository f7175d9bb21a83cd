// host_bits_test.cpp -- bvh::decode_gammas (bv_host.cpp), the host walk behind the device decoder of gamma-coded labels: streams written here bit by bit, MSB first,
// as GammaCodedIntLabel.toBitStream does (labelling/GammaCodedIntLabel.java:66-69; OutputBitStream.writeGamma), read back from any bit offset; a stretch that holds more or
// fewer codes than asked, or ends inside a codeword, is BVG_EFORMAT.  Built with g++ by tests/test_tools_cpu.py.
#include "../../webgraph_amd/csrc/bv_host.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
struct Bits {
	std::vector<uint8_t> b;
	uint64_t n = 0;
	void put(int bit) { if ((n & 7) == 0) b.push_back(0); if (bit) b[n >> 3] |= (uint8_t)(0x80 >> (n & 7)); n++; }
	void gamma(uint64_t x) { // msb(x + 1) zeros, then x + 1 in msb + 1 bits
		const uint64_t v = x + 1;
		int m = 63 - __builtin_clzll(v);
		for (int i = 0; i < m; i++) put(0);
		for (int i = m; i >= 0; i--) put((int)((v >> i) & 1));
	}
};
int fails = 0;
void expect(bool c, const char *what) { if (!c) { fprintf(stderr, "FAILED: %s\n", what); fails++; } }
} // namespace

int main() {
	uint64_t seed = 12345;
	auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return seed >> 33; };
	for (int round = 0; round < 200; round++) {
		Bits s;
		const int lead = (int)(rnd() % 70); // the stretch starts anywhere in a byte
		for (int i = 0; i < lead; i++) s.put((int)(rnd() & 1));
		const uint64_t lo = s.n;
		const int count = (int)(rnd() % 300);
		std::vector<int32_t> want;
		for (int i = 0; i < count; i++) {
			const int kind = (int)(rnd() % 4);
			const uint64_t x = kind == 0 ? 0 : kind == 1 ? 5 : kind == 2 ? rnd() % 1000 : rnd() % 0x7fffffffull;
			want.push_back((int32_t)x);
			s.gamma(x);
		}
		const uint64_t hi = s.n;
		for (int i = 0; i < 40; i++) s.put((int)(rnd() & 1)); // whatever follows
		s.b.push_back(0);
		std::vector<int32_t> got((size_t)count + 2, -7);
		expect(bvh::decode_gammas(s.b.data(), lo, hi, count, got.data()) == BVG_OK, "a stretch with exactly its codes");
		for (int i = 0; i < count; i++) expect(got[(size_t)i] == want[(size_t)i], "value");
		if (count > 0) {
			expect(bvh::decode_gammas(s.b.data(), lo, hi, count - 1, got.data()) == BVG_EFORMAT, "one code fewer than the stretch holds");
			expect(bvh::decode_gammas(s.b.data(), lo, hi, count + 1, got.data()) == BVG_EFORMAT, "one code more than the stretch holds");
			if (hi - lo > 1) expect(bvh::decode_gammas(s.b.data(), lo, hi - 1, count, got.data()) == BVG_EFORMAT, "a stretch that ends inside a codeword");
		} else expect(bvh::decode_gammas(s.b.data(), lo, hi, 0, got.data()) == BVG_OK, "an empty stretch");
	}
	{ // every label equal: the stream the device decoder cannot re-synchronise on
		Bits s;
		for (int i = 0; i < 100000; i++) s.gamma(5);
		s.b.push_back(0);
		std::vector<int32_t> got(100000);
		expect(bvh::decode_gammas(s.b.data(), 0, s.n, 100000, got.data()) == BVG_OK, "100 000 equal labels");
		bool all = true;
		for (int32_t v : got) all = all && v == 5;
		expect(all, "their values");
	}
	{ // a value that does not fit a Java int
		Bits s;
		s.gamma(0x80000000ull);
		s.b.push_back(0);
		int32_t got[2];
		expect(bvh::decode_gammas(s.b.data(), 0, s.n, 1, got) == BVG_EFORMAT, "a label of 2^31");
	}
	if (fails) { fprintf(stderr, "%d checks failed\n", fails); return 1; }
	printf("host_bits_test: ok\n");
	return 0;
}
