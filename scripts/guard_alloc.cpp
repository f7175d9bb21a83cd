// LD_PRELOAD shim for hunting out-of-bounds device accesses: every hipMalloc'ed buffer ENDS at an unmapped page.
//
//   hipcc -O1 -shared -fPIC -o gpurun_out/libguard.so scripts/guard_alloc.cpp -ldl
//   LD_PRELOAD=$PWD/gpurun_out/libguard.so python -u -m pytest tests -m gpu -x -v
//
// hipMalloc normally hands out slices of larger mappings, so a kernel that reads a few words past the end of its
// buffer almost always lands on mapped memory and nobody notices -- until, once in a while, the buffer is the last
// of its mapping and the process dies with "Memory access fault by GPU".  Here every allocation gets its own
// reservation (HIP virtual memory management): `mapped` bytes backed by device memory followed by one granule
// that is reserved and never mapped, and the pointer handed out is placed so that the buffer ends (rounded up to
// GUARD_ALIGN bytes, default 16, env GUARD_ALIGN) where the mapping ends.  GUARD_FILL=<byte> fills every new buffer
// (fresh device memory is zero in practice, which hides reads of words nobody wrote); GUARD_VMM=0 keeps only the fill.  A read or write past the end then
// faults every time, in the test that does it.  Test tool only: nothing in the product links or loads it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

namespace {
struct Rec { void *va; size_t reserved, mapped; hipMemGenericAllocationHandle_t h; };
std::mutex mu;
std::unordered_map<void *, Rec> live;
size_t n_alloc = 0, n_fallback = 0;

using malloc_fn = hipError_t (*)(void **, size_t);
using free_fn = hipError_t (*)(void *);
malloc_fn real_malloc() { static malloc_fn f = (malloc_fn)dlsym(RTLD_NEXT, "hipMalloc"); return f; }
free_fn real_free() { static free_fn f = (free_fn)dlsym(RTLD_NEXT, "hipFree"); return f; }
size_t guard_align() { static size_t a = [] { const char *e = getenv("GUARD_ALIGN"); const long v = e ? atol(e) : 16; return (size_t)(v >= 1 ? v : 16); }(); return a; }
size_t guard_max() { static size_t a = [] { const char *e = getenv("GUARD_MAX_BYTES"); return e ? (size_t)atoll(e) : (size_t)1 << 30; }(); return a; } // larger buffers go to the real hipMalloc
int guard_fill() { static int a = [] { const char *e = getenv("GUARD_FILL"); return e ? (int)strtol(e, nullptr, 0) : -1; }(); return a; } // >= 0: every new buffer is filled with this byte
bool guard_front() { static bool a = [] { const char *e = getenv("GUARD_FRONT"); return e && atoi(e) != 0; }(); return a; } // 1: the unmapped granule sits IN FRONT of the buffer (negative indices fault; overruns do not)
bool guard_vmm() { static bool a = [] { const char *e = getenv("GUARD_VMM"); return !e || atoi(e) != 0; }(); return a; } // 0: plain hipMalloc (only the fill remains)
long guard_only() { static long a = [] { const char *e = getenv("GUARD_ONLY"); return e ? atol(e) : -1L; }(); return a; } // >= 0: only the allocation with this ordinal is guarded (bisection)
size_t n_seen = 0;
bool guard_selected(size_t ordinal) { // GUARD_SET=a,b,c: only these ordinals; GUARD_BELOW=k: only ordinals < k
	if (const char *e = getenv("GUARD_BELOW")) if ((long)ordinal >= atol(e)) return false;
	if (const char *e = getenv("GUARD_SET")) { for (const char *q = e; *q;) { char *end; const long v = strtol(q, &end, 10); if (end == q) break; if ((size_t)v == ordinal) return true; q = *end ? end + 1 : end; } return false; }
	return true;
}
struct Report { ~Report() { if (getenv("GUARD_VERBOSE")) fprintf(stderr, "[guard_alloc] %zu guarded allocations, %zu passed through\n", n_alloc, n_fallback); } } report;
} // namespace

extern "C" hipError_t hipMalloc(void **out, size_t size) {
	if (!out) return hipErrorInvalidValue;
	const size_t ordinal = n_seen++;
	if (getenv("GUARD_TRACE")) fprintf(stderr, "[guard_alloc] #%zu: %zu bytes\n", ordinal, size);
	if (size == 0 || size > guard_max() || !guard_vmm() || (guard_only() >= 0 && (size_t)guard_only() != ordinal) || !guard_selected(ordinal)) {
		n_fallback++;
		const hipError_t e = real_malloc()(out, size);
		if (e == hipSuccess && size && guard_fill() >= 0) { (void)hipMemset(*out, guard_fill(), size); (void)hipDeviceSynchronize(); } // (the fill runs on the null stream: it must not overtake, or be overtaken by, the caller's streams)
		return e;
	}
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return real_malloc()(out, size);
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = dev;
	size_t g = 0;
	if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || g == 0) { (void)hipGetLastError(); n_fallback++; return real_malloc()(out, size); }
	const size_t a = guard_align(), user = (size + a - 1) / a * a, mapped = (user + g - 1) / g * g, reserved = mapped + g;
	Rec r = { nullptr, reserved, mapped, {} };
	void *resv = nullptr;
	if (hipMemAddressReserve(&resv, reserved, g, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); n_fallback++; return real_malloc()(out, size); }
	r.va = guard_front() ? (void *)((char *)resv + g) : resv; // (front mode: the first granule of the reservation stays unmapped)
	if (hipMemCreate(&r.h, mapped, &prop, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemAddressFree(resv, reserved); return hipErrorOutOfMemory; }
	hipMemAccessDesc ad = {};
	ad.location = prop.location;
	ad.flags = hipMemAccessFlagsProtReadWrite;
	if (hipMemMap(r.va, mapped, 0, r.h, 0) != hipSuccess || hipMemSetAccess(r.va, mapped, &ad, 1) != hipSuccess) {
		(void)hipGetLastError();
		(void)hipMemUnmap(r.va, mapped);
		(void)hipMemRelease(r.h);
		(void)hipMemAddressFree(resv, reserved);
		n_fallback++;
		return real_malloc()(out, size);
	}
	if (guard_fill() >= 0) { (void)hipMemset(r.va, guard_fill(), mapped); (void)hipDeviceSynchronize(); }
	void *p = guard_front() ? r.va : (void *)((char *)r.va + (mapped - user));
	if (getenv("GUARD_TRACE")) fprintf(stderr, "[guard_alloc] #%zu: [%p, %p) mapped [%p, %p)\n", ordinal, p, (void *)((char *)p + size), r.va, (void *)((char *)r.va + mapped));
	{
		std::lock_guard<std::mutex> lk(mu);
		live[p] = r;
		n_alloc++;
	}
	*out = p;
	return hipSuccess;
}

extern "C" hipError_t hipFree(void *p) {
	if (!p) return hipSuccess;
	Rec r;
	{
		std::lock_guard<std::mutex> lk(mu);
		auto it = live.find(p);
		if (it == live.end()) return real_free()(p);
		r = it->second;
		live.erase(it);
	}
	(void)hipDeviceSynchronize(); // hipFree's implicit synchronisation
	(void)hipMemUnmap(r.va, r.mapped);
	(void)hipMemRelease(r.h);
	// The address range is NOT given back unless GUARD_REUSE_VA=1: a later reservation would get the same addresses, and
	// on this stack a mapping made at a just-unmapped address can still serve the OLD pages' contents for a while
	// (measured: a 36-byte buffer freed and allocated again read back the words of its previous life).  It also
	// turns a use after free into a fault.
	static const bool reuse = [] { const char *e = getenv("GUARD_REUSE_VA"); return e && atoi(e) != 0; }();
	if (reuse) (void)hipMemAddressFree(guard_front() ? (void *)((char *)r.va - (r.reserved - r.mapped)) : r.va, r.reserved);
	return hipSuccess;
}
