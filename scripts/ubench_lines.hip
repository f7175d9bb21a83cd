// GPU box: what a store INSTRUCTION costs a CU on gfx950 by the number of cache lines its 64 lanes touch (round 6: the value pass of the wave class writes 64 runs,
// one per lane -- 64 lines per instruction).  Every wave writes its own small region over and over (the footprint stays in the L2 / Infinity Cache: HBM is not the limit).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_lines scripts/ubench_lines.hip && /tmp/ubench_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));

// MODE 0: 4-byte stores, lane stride `stride` ints (stride 1: 256 contiguous bytes; 64: 64 lines); MODE 1: 16-byte stores, lane stride `stride` ints (4: contiguous 1 KB)
// MODE 2: 4-byte LOADS (same addressing), MODE 3: 16-byte loads
template <int MODE>
__global__ void __launch_bounds__(64) k(int *__restrict__ base, int regionInts, int stride, int iters, int alu) {
	int *reg = base + (size_t)blockIdx.x * regionInts;
	const int lane = threadIdx.x;
	int acc = lane, off = 0;
	const int span = MODE & 1 ? 4 : 1;
	for (int i = 0; i < iters; i++) {
		for (int a = 0; a < alu; a++) acc = acc * 1664525 + 1013904223;
		int *p = reg + ((lane * stride + off) % (regionInts - 4));
		if (MODE == 0) *p = acc;
		else if (MODE == 1) *(int4u *)p = int4u{ acc, acc, acc, acc };
		else if (MODE == 2) acc += *(volatile int *)p;
		else { const int4u q = *(volatile int4u *)p; acc += q.x + q.w; }
		off += span; // the next trip writes the next ids of every lane's run
		if (off >= stride && stride >= span) off = 0;
	}
	if (acc == 0x12345678) base[0] = acc;
}

int main() {
	const int waves = 256 * 12, regionInts = 16384; // 64 KB per wave, 200 MB in all... the runs of one trip stay inside 64 x stride ints
	int *d;
	CK(hipMalloc(&d, (size_t)waves * regionInts * sizeof(int)));
	CK(hipMemset(d, 0, (size_t)waves * regionInts * sizeof(int)));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int iters = 4096;
	auto run = [&](const char *name, auto kern, int stride, int alu) {
		hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, d, regionInts, stride, iters, alu);
		CK(hipEventRecord(e0));
		for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, d, regionInts, stride, iters, alu);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		ms /= 3;
		// cycles of one CU per memory instruction: 12 waves per CU share it
		const double perInstrNs = ms * 1e6 / ((double)iters * 12);
		printf("%-26s lane stride %4d ints, alu %3d: %8.3f ms  %7.1f ns = %6.0f cycles (2.4 GHz) per instruction and CU\n", name, stride, alu, ms, perInstrNs, perInstrNs * 2.4);
	};
	for (int alu : { 0, 16 }) {
		for (int stride : { 1, 4, 16, 64, 55, 220 }) run("4-byte stores", k<0>, stride, alu);
		for (int stride : { 4, 16, 64, 55, 220 }) run("16-byte stores", k<1>, stride, alu);
		for (int stride : { 1, 16, 64, 220 }) run("4-byte loads", k<2>, stride, alu);
		for (int stride : { 4, 16, 64, 220 }) run("16-byte loads", k<3>, stride, alu);
	}
	return 0;
}
