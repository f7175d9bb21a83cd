// GPU box: what a scattered store costs on gfx950 (round 6: the one-lane parse kernel writes 64 different rows per instruction).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_store scripts/ubench_store.hip && /tmp/ubench_store
// Every lane owns a row of ROW ints at base + lane_global * ROWSTRIDE; a wave's store instruction touches 64 different rows.
// Variants: 16-byte stores on 16-byte boundaries, 16-byte stores 4 / 8 / 12 bytes off (gfx950 takes unaligned dwordx4 stores: checked against a
// reference fill), 4-byte stores, and 16-byte stores where four neighbouring lanes share a 64-byte piece of a row (16 lines per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k_store(int *__restrict__ base, int rowStride, int groups, int mis, int alu) {
	const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
	int *row = base + t * rowStride + mis;
	int acc = (int)t;
	for (int g = 0; g < groups; g++) {
		// a little dependent ALU between the stores, like a decode loop
		for (int a = 0; a < alu; a++) acc = acc * 1664525 + 1013904223;
		const int v = (int)t * 1000 + 4 * g + (acc & 0);
		if (MODE == 0) *(int4u *)(row + 4 * g) = int4u{ v, v + 1, v + 2, v + 3 };
		else if (MODE == 1) { row[4 * g] = v; row[4 * g + 1] = v + 1; row[4 * g + 2] = v + 2; row[4 * g + 3] = v + 3; }
		else if (MODE == 2) {
			// four lanes share a row piece: lane L writes piece (L & 3) of the row of lane group (L >> 2), four groups of ids per trip
			const long long owner = (t & ~3ll);
			int *r2 = base + (owner + (g & 3)) * rowStride + mis;
			const int gg = (g >> 2) * 4 + (int)(t & 3);
			const int v2 = (int)(owner + (g & 3)) * 1000 + 4 * gg;
			*(int4u *)(r2 + 4 * gg) = int4u{ v2, v2 + 1, v2 + 2, v2 + 3 };
		}
	}
	if (acc == 0x7fffffff) base[0] = acc;
}

int main() {
	const int blocks = 256 * 20, threads = blocks * 256, rowStride = 1024 + 36, groups = 64; // 256 ids per row
	const size_t n = (size_t)threads * rowStride + 64;
	int *d;
	CK(hipMalloc(&d, n * sizeof(int)));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	std::vector<int> h(rowStride * 8);
	auto run = [&](const char *name, auto kern, int mis, int alu) {
		CK(hipMemset(d, 0xff, n * sizeof(int)));
		for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, rowStride, groups, mis, alu);
		CK(hipEventRecord(e0));
		for (int w = 0; w < 5; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, rowStride, groups, mis, alu);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		ms /= 5;
		CK(hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost));
		bool ok = true;
		for (int r = 0; r < 8 && ok; r++) for (int i = 0; i < 4 * groups; i++) if (h[(size_t)r * rowStride + mis + i] != r * 1000 + i) { ok = false; printf("  row %d id %d = %d\n", r, i, h[(size_t)r * rowStride + mis + i]); break; }
		const double instr = (double)threads / 64 * groups; // store instructions (MODE 1: four times as many)
		printf("%-44s mis %d alu %3d: %7.3f ms  %6.1f GB/s  %5.1f ns per wave-store  %s\n", name, mis, alu, ms, (double)threads * groups * 16 / ms / 1e6, ms * 1e6 / instr * 256 * 4 / 1.0 / (double)(256 * 4) * 1.0, ok ? "ok" : "WRONG");
	};
	for (int alu : { 0, 40, 160 }) {
		for (int mis = 0; mis < 4; mis++) run("16-byte stores, 64 rows per instruction", k_store<0>, mis, alu);
		run("4-byte stores", k_store<1>, 0, alu);
		run("4-byte stores", k_store<1>, 1, alu);
		run("16-byte stores, 16 rows x 64 bytes per instr", k_store<2>, 0, alu);
		run("16-byte stores, 16 rows x 64 bytes per instr", k_store<2>, 1, alu);
	}
	return 0;
}
