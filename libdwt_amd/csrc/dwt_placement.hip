// dwt_placement.hip -- placement-aware device memory for the sweeps.
//
// Why.  The sweeps run three streams at once: they read the source rows and write the detail
// subbands and the running LL band.  On MI355X the rate of the SAME launch on the SAME virtual
// addresses moves between plateaus (level 0 of 64 images: 5.2 / 5.4 / 5.75 / 6.15 TB/s) with WHERE in
// physical memory the three buffers lie (profiles/r04_placement.md): physical memory falls into
// three classes of coarse regions (tens of GiB; consistent with the three stack-ID ranks of the 12-high
// HBM3E stacks), and streams that run at the same time in regions of the SAME class slow each other
// down -- two write streams by 13 %, a read and a write stream by 3-4 % -- while nothing at all depends
// on fine address bits (offsets of 1 KiB .. 1 GiB inside an allocation, row pitch, image stride).
// Same requests, same TLB misses, more DRAM-credit stalls at the L2's memory side.
//
// The reference hands its callers a placement-aware allocator for the same kind of reason
// (dwt_util_get_opt_stride / dwt_util_get_stride, src/libdwt.c:20641-20707: power-of-two pitches alias
// in the CPU caches).  The analogue here: memory of a chosen CLASS.  Physical memory is taken in chunks
// through HIP's virtual-memory API (hipMemCreate), each chunk is classified by timing a small two-stream
// write kernel against three reference chunks of mutually different classes, and a buffer is a virtual
// range mapped from chunks of one class only.  A caller that keeps source, destination and the
// library's workspace in three different classes sits on the fast plateau by construction.
#include "dwt_backend.h"

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <vector>

namespace dwtb {

// ---- the probe: two (or one) streaming write streams, the way the sweeps store ----------------
// Every wave stores 1 KiB pieces (16 B per lane, non-temporal) alternately to a and b; consecutive
// waves take consecutive pieces, so each stream is one dense sequential write of `bytes`.
__global__ __launch_bounds__(256) void k_probe_streams(char *a, char *b, size_t bytes)
{
	typedef unsigned u4 __attribute__((ext_vector_type(4)));
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const unsigned lane = threadIdx.x & 63;
	const u4 v = {lane, 1u, 2u, 3u};
	for (size_t piece = wave; piece * 1024 < bytes; piece += nwaves) {
		const size_t off = piece * 1024 + lane * 16;
		__builtin_nontemporal_store(v, (u4 *)(a + off));
		if (b)
			__builtin_nontemporal_store(v, (u4 *)(b + off));
	}
}

// Read stream + write stream: a dense copy a -> b, 1 KiB pieces, non-temporal both ways.
__global__ __launch_bounds__(256) void k_probe_copy(const char *a, char *b, size_t bytes)
{
	typedef unsigned u4 __attribute__((ext_vector_type(4)));
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const unsigned lane = threadIdx.x & 63;
	for (size_t piece = wave; piece * 1024 < bytes; piece += nwaves) {
		const size_t off = piece * 1024 + lane * 16;
		const u4 v = __builtin_nontemporal_load((const u4 *)(a + off));
		__builtin_nontemporal_store(v, (u4 *)(b + off));
	}
}

// microseconds of one probe launch (median of `reps` after one warm-up); < 0 on error
static double probe_us(void *a, void *b, size_t bytes, int reps = 5, bool copy = false)
{
	hipEvent_t e0, e1;
	if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
		return -1;
	std::vector<float> t;
	for (int r = 0; r <= reps; r++) {
		hipEventRecord(e0, g.stream);
		if (copy)
			k_probe_copy<<<1024, 256, 0, g.stream>>>((const char *)a, (char *)b, bytes);
		else
			k_probe_streams<<<1024, 256, 0, g.stream>>>((char *)a, (char *)b, bytes);
		hipEventRecord(e1, g.stream);
		if (hipEventSynchronize(e1) != hipSuccess)
			break;
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		if (r)
			t.push_back(ms * 1e3f);
	}
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	if (t.empty())
		return -1;
	std::sort(t.begin(), t.end());
	return t[t.size() / 2];
}

// ---- buffers mapped from physical pieces (HIP virtual-memory API) ---------------------------------
struct VmmBuf {
	size_t bytes = 0, piece = 0;
	std::vector<hipMemGenericAllocationHandle_t> handles;
};
static std::map<void *, VmmBuf> g_vmm;
static std::mutex g_vmm_mu;

static int vmm_release(void *va, VmmBuf &b, size_t mapped_pieces)
{
	for (size_t i = 0; i < mapped_pieces; i++)
		hipMemUnmap((char *)va + i * b.piece, b.piece);
	for (auto h : b.handles)
		hipMemRelease(h);
	if (va)
		hipMemAddressFree(va, b.bytes);
	return 1;
}

// `bytes` of device memory as ONE virtual range mapped from physical pieces of `piece` bytes taken in
// `slices` groups, with `ballast` bytes of ordinary allocations made between the groups (and freed
// before returning) so that the groups come from physical memory far apart; piece i of the range is
// piece i / slices of group i % slices.  slices == 1: plain pieces in creation order.
static void *vmm_alloc(size_t bytes, size_t piece, int slices, size_t ballast)
{
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = g.device;
	size_t gran = 0;
	if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) {
		fail("hipMemGetAllocationGranularity failed");
		return nullptr;
	}
	piece = (piece + gran - 1) / gran * gran;
	const size_t n = (bytes + piece - 1) / piece;
	VmmBuf b;
	b.piece = piece;
	b.bytes = n * piece;
	void *va = nullptr;
	if (hipMemAddressReserve(&va, b.bytes, 0, nullptr, 0) != hipSuccess) {
		fail("hipMemAddressReserve(%zu) failed", b.bytes);
		return nullptr;
	}
	if (slices < 1)
		slices = 1;
	std::vector<std::vector<hipMemGenericAllocationHandle_t>> grp(slices);
	std::vector<void *> ballasts;
	bool ok = true;
	for (int k = 0; k < slices && ok; k++) {
		const size_t cnt = n / slices + ((size_t)k < n % slices ? 1 : 0);
		for (size_t i = 0; i < cnt && ok; i++) {
			hipMemGenericAllocationHandle_t h;
			ok = hipMemCreate(&h, piece, &prop, 0) == hipSuccess;
			if (ok) {
				grp[k].push_back(h);
				b.handles.push_back(h);
			}
		}
		if (ballast && k + 1 < slices) {
			void *p = nullptr;
			if (hipMalloc(&p, ballast) == hipSuccess)
				ballasts.push_back(p);
			else
				(void)hipGetLastError(); // not enough room left for the spacing: go on without it
		}
	}
	for (void *p : ballasts)
		hipFree(p);
	size_t mapped = 0;
	for (size_t i = 0; i < n && ok; i++) {
		ok = hipMemMap((char *)va + i * piece, piece, 0, grp[i % slices][i / slices], 0) == hipSuccess;
		if (ok)
			mapped++;
	}
	if (ok) {
		hipMemAccessDesc acc = {};
		acc.location = prop.location;
		acc.flags = hipMemAccessFlagsProtReadWrite;
		ok = hipMemSetAccess(va, b.bytes, &acc, 1) == hipSuccess;
	}
	if (!ok) {
		fail("mapping %zu bytes from %zu-byte pieces failed: %s", bytes, piece, hipGetErrorString(hipGetLastError()));
		vmm_release(va, b, mapped);
		return nullptr;
	}
	std::lock_guard<std::mutex> lk(g_vmm_mu);
	g_vmm[va] = std::move(b);
	return va;
}

// SPREAD allocation: `bytes` mapped from pieces taken at even distances through ALL the physical memory
// that is free right now ("comb"): between two pieces that are kept, one filler allocation of
// (free - bytes) / pieces bytes is made and released again at the end.  Successive physical allocations
// come from neighbouring physical memory, so the kept pieces sample every region of the card in
// proportion -- every buffer allocated this way is the same mix of the physical classes, whatever else
// is allocated, and any two such buffers have the same relation to each other.
static void *spread_alloc(size_t bytes, size_t piece, size_t reserve)
{
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = g.device;
	size_t gran = 0;
	if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) {
		fail("hipMemGetAllocationGranularity failed");
		return nullptr;
	}
	piece = (piece + gran - 1) / gran * gran;
	const size_t n = (bytes + piece - 1) / piece;
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
		fail("hipMemGetInfo failed");
		return nullptr;
	}
	VmmBuf b;
	b.piece = piece;
	b.bytes = n * piece;
	if (free_b < b.bytes + reserve / 4) {
		fail("spread allocation of %zu bytes: only %zu bytes free", bytes, free_b);
		return nullptr;
	}
	const size_t spare = free_b > b.bytes + reserve ? free_b - b.bytes - reserve : 0;
	const size_t filler = spare / n / gran * gran;
	void *va = nullptr;
	if (hipMemAddressReserve(&va, b.bytes, 0, nullptr, 0) != hipSuccess) {
		fail("hipMemAddressReserve(%zu) failed", b.bytes);
		return nullptr;
	}
	std::vector<hipMemGenericAllocationHandle_t> fillers;
	bool ok = true, fill = filler > 0;
	for (size_t i = 0; i < n && ok; i++) {
		hipMemGenericAllocationHandle_t h;
		ok = hipMemCreate(&h, piece, &prop, 0) == hipSuccess;
		if (!ok && !fillers.empty()) {
			// the fillers ate what was left (another process allocated meanwhile): give them back, go on plainly
			(void)hipGetLastError();
			for (auto f : fillers)
				hipMemRelease(f);
			fillers.clear();
			fill = false;
			ok = hipMemCreate(&h, piece, &prop, 0) == hipSuccess;
		}
		if (!ok)
			break;
		b.handles.push_back(h);
		if (fill && i + 1 < n) {
			hipMemGenericAllocationHandle_t f;
			if (hipMemCreate(&f, filler, &prop, 0) == hipSuccess)
				fillers.push_back(f);
			else {
				(void)hipGetLastError();
				fill = false;
			}
		}
	}
	for (auto f : fillers)
		hipMemRelease(f);
	size_t mapped = 0;
	for (size_t i = 0; i < n && ok; i++) {
		ok = hipMemMap((char *)va + i * piece, piece, 0, b.handles[i], 0) == hipSuccess;
		if (ok)
			mapped++;
	}
	if (ok) {
		hipMemAccessDesc acc = {};
		acc.location = prop.location;
		acc.flags = hipMemAccessFlagsProtReadWrite;
		ok = hipMemSetAccess(va, b.bytes, &acc, 1) == hipSuccess;
	}
	if (!ok) {
		fail("spread allocation of %zu bytes from %zu-byte pieces failed: %s", bytes, piece, hipGetErrorString(hipGetLastError()));
		vmm_release(va, b, mapped);
		return nullptr;
	}
	std::lock_guard<std::mutex> lk(g_vmm_mu);
	g_vmm[va] = std::move(b);
	return va;
}

static bool vmm_free(void *va)
{
	VmmBuf b;
	{
		std::lock_guard<std::mutex> lk(g_vmm_mu);
		auto it = g_vmm.find(va);
		if (it == g_vmm.end())
			return false;
		b = std::move(it->second);
		g_vmm.erase(it);
	}
	vmm_release(va, b, b.handles.size());
	return true;
}


// ---- placed allocation: buffers of two different physical classes -----------------------------------
// The sweep itself is the probe (nothing simpler shows the classes: a dense copy or two dense write
// streams keep their DRAM pages open and do not care; the sweeps' thousands of concurrent row streams
// open a page per access): one level of the float 9/7 on a few 4096^2 images, source in the reference
// chunk, all four subbands into the candidate chunk.  Same class: ~6 % slower.
static double probe_level_us(void *ref, void *cand, size_t chunk)
{
	const int W = 4096, H = 4096;
	dwt::FwdLevelArgs a;
	a.in = ref;
	a.in_pitch = W;
	a.in_bstride = (long)W * H;
	a.out_ll = a.out_h = cand;
	a.ll_pitch = a.h_pitch = W;
	a.ll_bstride = a.h_bstride = (long)W * H;
	a.W = W;
	a.H = H;
	a.batch = (int)(chunk / ((size_t)W * H * 4));
	hipEvent_t e0, e1;
	if (a.batch < 1 || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
		return -1;
	dwt::SweepTuning tune; // the defaults, whatever the caller has set
	float best[4];
	int n = 0;
	for (int r = 0; r < 4; r++) {
		hipEventRecord(e0, g.stream);
		if (dwt::launch_fwd_level(dwt::kCdf97S, a, tune, g.stream) != hipSuccess)
			break;
		hipEventRecord(e1, g.stream);
		if (hipEventSynchronize(e1) != hipSuccess)
			break;
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		if (r)
			best[n++] = ms * 1e3f;
	}
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	if (n < 3)
		return -1;
	std::sort(best, best + n);
	return best[n / 2];
}

struct PlacedStats {
	int walked = 0, same = 0, other = 0;
	double us_lo = 0, us_hi = 0, seconds = 0;
};
static thread_local PlacedStats g_placed_stats;
thread_local int g_placed_prefer = 0; // experiments: 1 = group A must be the reference chunk's class, 2 = the other

// Buffers `bytes_a[0..n_a)` from ONE physical class and `bytes_b[0..n_b)` from ANOTHER.  Physical memory
// is taken chunk by chunk (256 MiB), each chunk timed against the first one; the walk goes on until both
// groups can be filled, then every buffer is one virtual range mapped from chunks of its group and the
// rest is released.  Where the card does not offer two classes within the walk (or is nearly full)
// the groups are filled with what there is: the memory is valid either way.
static int placed_alloc(const size_t *bytes_a, int n_a, const size_t *bytes_b, int n_b, void **out_a, void **out_b)
{
	const size_t C = (size_t)256 << 20;
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = g.device;
	hipMemAccessDesc acc = {};
	acc.location = prop.location;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	size_t need_a = 0, need_b = 0;
	for (int i = 0; i < n_a; i++)
		need_a += (bytes_a[i] + C - 1) / C;
	for (int i = 0; i < n_b; i++)
		need_b += (bytes_b[i] + C - 1) / C;
	size_t free_b = 0, total_b = 0;
	HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	const size_t reserve = (size_t)2 << 30;
	if (free_b < (need_a + need_b) * C + reserve)
		return fail("placed allocation of %zu MiB: only %zu MiB free", (need_a + need_b) * (C >> 20), free_b >> 20);
	const size_t max_walk = (free_b - reserve) / C;
	const auto t_start = std::chrono::steady_clock::now();

	void *pva = nullptr; // probe window: [reference chunk | candidate chunk]
	HIP_TRY(hipMemAddressReserve(&pva, 2 * C, 0, nullptr, 0));
	struct Chunk {
		hipMemGenericAllocationHandle_t h;
		double us;
	};
	std::vector<Chunk> chunks;
	auto release_all = [&]() {
		for (auto &c : chunks)
			hipMemRelease(c.h);
		chunks.clear();
	};
	bool ref_mapped = false, failed = false;
	double lo = 0, hi = 0;
	auto split = [&](size_t &same, size_t &other) {
		// two clusters when the spread exceeds 3 %: chunks slower than the midpoint share the reference's class
		same = other = 0;
		const bool two = hi > lo * 1.03;
		for (auto &c : chunks)
			(!two || c.us > 0.5 * (lo + hi) ? same : other)++;
	};
	while (chunks.size() < max_walk) {
		Chunk c{};
		if (hipMemCreate(&c.h, C, &prop, 0) != hipSuccess) {
			(void)hipGetLastError();
			break;
		}
		if (!ref_mapped) {
			if (hipMemMap(pva, C, 0, c.h, 0) != hipSuccess || hipMemSetAccess(pva, C, &acc, 1) != hipSuccess) {
				hipMemRelease(c.h);
				failed = true;
				break;
			}
			ref_mapped = true;
			c.us = 1e30; // the reference is of its own class by definition
			chunks.push_back(c);
			continue;
		}
		char *cva = (char *)pva + C;
		if (hipMemMap(cva, C, 0, c.h, 0) != hipSuccess || hipMemSetAccess(cva, C, &acc, 1) != hipSuccess) {
			hipMemRelease(c.h);
			failed = true;
			break;
		}
		c.us = probe_level_us(pva, cva, C);
		hipMemUnmap(cva, C);
		if (c.us <= 0) {
			hipMemRelease(c.h);
			failed = true;
			break;
		}
		lo = chunks.size() == 1 ? c.us : std::min(lo, c.us);
		hi = chunks.size() == 1 ? c.us : std::max(hi, c.us);
		chunks.push_back(c);
		size_t same, other;
		split(same, other);
		const bool fit1 = same >= need_a && other >= need_b, fit2 = same >= need_b && other >= need_a;
		if ((fit1 && g_placed_prefer != 2) || (fit2 && g_placed_prefer != 1))
			break;
	}
	if (ref_mapped)
		hipMemUnmap(pva, C);
	hipMemAddressFree(pva, 2 * C);
	if (failed || chunks.size() < need_a + need_b) {
		release_all();
		return fail("placed allocation: walking the physical memory failed (%s)", hipGetErrorString(hipGetLastError()));
	}
	chunks[0].us = hi > lo * 1.03 ? hi : 1e30;
	size_t same, other;
	split(same, other);
	// group A takes the reference's class unless only the other arrangement fits
	const bool fit1 = same >= need_a && other >= need_b, fit2 = same >= need_b && other >= need_a;
	const bool a_is_same = g_placed_prefer == 1 ? true : g_placed_prefer == 2 ? false : (fit1 || !fit2);
	const double mid = hi > lo * 1.03 ? 0.5 * (lo + hi) : -1;
	std::vector<hipMemGenericAllocationHandle_t> pool_same, pool_other;
	for (auto &c : chunks)
		(c.us > mid ? pool_same : pool_other).push_back(c.h);
	std::vector<hipMemGenericAllocationHandle_t> &pa = a_is_same ? pool_same : pool_other, &pb = a_is_same ? pool_other : pool_same;
	auto take = [&](std::vector<hipMemGenericAllocationHandle_t> &own, std::vector<hipMemGenericAllocationHandle_t> &alt) {
		std::vector<hipMemGenericAllocationHandle_t> &from = own.empty() ? alt : own;
		hipMemGenericAllocationHandle_t h = from.front(); // front: keep the walk's order inside a buffer
		from.erase(from.begin());
		return h;
	};
	std::vector<void *> made;
	auto build = [&](size_t bytes, bool group_a, void **out) -> int {
		const size_t n = (bytes + C - 1) / C;
		VmmBuf b;
		b.piece = C;
		b.bytes = n * C;
		void *va = nullptr;
		if (hipMemAddressReserve(&va, b.bytes, 0, nullptr, 0) != hipSuccess)
			return fail("hipMemAddressReserve(%zu) failed", b.bytes);
		size_t mapped = 0;
		bool ok = true;
		for (size_t i = 0; i < n && ok; i++) {
			hipMemGenericAllocationHandle_t h = group_a ? take(pa, pb) : take(pb, pa);
			b.handles.push_back(h);
			ok = hipMemMap((char *)va + i * C, C, 0, h, 0) == hipSuccess;
			if (ok)
				mapped++;
		}
		ok = ok && hipMemSetAccess(va, b.bytes, &acc, 1) == hipSuccess;
		if (!ok) {
			vmm_release(va, b, mapped);
			return fail("mapping a placed buffer of %zu bytes failed: %s", bytes, hipGetErrorString(hipGetLastError()));
		}
		std::lock_guard<std::mutex> lk(g_vmm_mu);
		g_vmm[va] = std::move(b);
		*out = va;
		made.push_back(va);
		return 0;
	};
	int rc = 0;
	for (int i = 0; i < n_a && !rc; i++)
		rc = build(bytes_a[i], true, &out_a[i]);
	for (int i = 0; i < n_b && !rc; i++)
		rc = build(bytes_b[i], false, &out_b[i]);
	for (auto h : pool_same)
		hipMemRelease(h);
	for (auto h : pool_other)
		hipMemRelease(h);
	if (rc) {
		for (void *p : made)
			vmm_free(p);
		return rc;
	}
	g_placed_stats.walked = (int)chunks.size();
	g_placed_stats.same = (int)same;
	g_placed_stats.other = (int)other;
	g_placed_stats.us_lo = lo;
	g_placed_stats.us_hi = hi;
	g_placed_stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	return 0;
}

// device memory of either kind back to the system
void dev_free(void *p)
{
	if (p && !vmm_free(p))
		hipFree(p);
}

} // namespace dwtb

using namespace dwtb;

#pragma GCC visibility push(default)
extern "C" {

// Time of the two-stream probe on device buffers a and b (`bytes` each; b may be NULL: one stream),
// in microseconds; negative on error.  Diagnostic entry (scripts/probes/r04_*).
double dwt_hip_probe_pair_us(void *a, void *b, size_t bytes)
{
	if (check_inited())
		return -1;
	return probe_us(a, b, bytes);
}

double dwt_hip_probe_copy_us(const void *a, void *b, size_t bytes)
{
	if (check_inited())
		return -1;
	return probe_us((void *)a, b, bytes, 5, true);
}

// Experimental: see vmm_alloc.  Freed with dwt_hip_free_mapped.
void *dwt_hip_malloc_mapped(size_t bytes, size_t piece_bytes, int slices, size_t ballast_bytes)
{
	if (check_inited())
		return nullptr;
	return vmm_alloc(bytes, piece_bytes, slices, ballast_bytes);
}

void *dwt_hip_malloc_spread(size_t bytes, size_t piece_bytes)
{
	if (check_inited())
		return nullptr;
	return spread_alloc(bytes, piece_bytes ? piece_bytes : (size_t)2 << 20, (size_t)1 << 30);
}

int dwt_hip_alloc_placed(const size_t *bytes_a, int n_a, const size_t *bytes_b, int n_b, void **out_a, void **out_b)
{
	if (check_inited())
		return 1;
	if (n_a < 0 || n_b < 0 || n_a + n_b < 1 || (n_a && (!bytes_a || !out_a)) || (n_b && (!bytes_b || !out_b)))
		return fail("dwt_hip_alloc_placed: bad argument");
	return placed_alloc(bytes_a, n_a, bytes_b, n_b, out_a, out_b);
}

// what the last dwt_hip_alloc_placed of this thread saw: chunks walked, of the reference's class, of the
// other class, probe time of the fastest and the slowest chunk (us), seconds spent
void dwt_hip_placed_stats(int *walked, int *same, int *other, double *us_lo, double *us_hi, double *seconds)
{
	if (walked) *walked = g_placed_stats.walked;
	if (same) *same = g_placed_stats.same;
	if (other) *other = g_placed_stats.other;
	if (us_lo) *us_lo = g_placed_stats.us_lo;
	if (us_hi) *us_hi = g_placed_stats.us_hi;
	if (seconds) *seconds = g_placed_stats.seconds;
}

void dwt_hip_free_mapped(void *p)
{
	if (p)
		vmm_free(p);
}

} // extern "C"
#pragma GCC visibility pop
