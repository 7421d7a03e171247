// dwt_placement.hip -- placement-aware device memory (profiles/r04_placement.md).
//
// The sweeps run three streams at once -- source rows, detail subbands, running LL band -- and their rate
// depends on WHERE in physical memory those lie relative to each other: physical memory falls into coarse
// regions (16 GiB granules) of a few classes, and streams that run at the same time in regions of the same
// class slow each other down (level 0 of 64 images: 5.2 ... 6.2 TB/s on the same virtual addresses), while
// nothing depends on fine address bits.  The reference hands its callers a placement-aware allocator for the
// same kind of reason (dwt_util_get_opt_stride / dwt_util_get_stride, src/libdwt.c:20641-20707: power-of-two
// pitches alias in the CPU caches).  This file holds
//   - the product's allocators: dwt_hip_alloc_batch / dwt_hip_alloc_volumes over arena_place -- most of the
//     free memory mapped as ONE arena (HIP virtual-memory API, 1 GiB physical chunks), every arrangement of
//     destination and workspace measured with the workload itself, the best kept mapped where it was measured;
//   - the instruments the diagnosis was made with and scripts/archive/probes/r04_*.py call: dense two-stream write /
//     copy probes (dwt_hip_probe_pair_us / _copy_us: they do NOT see the classes the sweeps see -- their DRAM
//     pages stay open), buffers mapped from physical pieces of far-apart groups (dwt_hip_malloc_mapped) or from
//     pieces at even distances through ALL free memory (dwt_hip_malloc_spread: the same mix of the classes for
//     every buffer -- the same rate in every process, but the rate of the mix, 5.7 TB/s, not of the best
//     arrangement, 6.2).
// Buffers of every kind are freed by dwt_hip_free.  (The library's own scratch for callers who bring their
// buffers: place_ll_scratch in dwt_backend.hip.)
#include "dwt_backend.h"

#include <algorithm>
#include <chrono>
#include <functional>
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
struct VmmArena { // one address reservation shared by several buffers (dwt_hip_alloc_batch)
	void *base;
	size_t bytes;
	int live;
};
struct VmmBuf {
	size_t bytes = 0, piece = 0;
	std::vector<hipMemGenericAllocationHandle_t> handles;
	VmmArena *arena = nullptr; // set: the range is part of that reservation, which goes when its last part goes
};
static std::map<void *, VmmBuf> g_vmm;
static std::mutex g_vmm_mu;

// Read-write access for the owning device and for the peers named (devices == nullptr: every device that can reach it
// as a peer): hipDeviceEnablePeerAccess does not cover ranges mapped through the virtual-memory API, so a placed buffer
// that is to be the source / destination of hipMemcpyPeerAsync (dwt_multi.hip) is granted to exactly the devices of that
// call, when the call is made (grant_range).  Buffers are CREATED with access for their owner alone: a process that
// drives one GPU of eight (one rank of bench.py) never touches the other seven.
static std::vector<hipMemAccessDesc> access_descs(int owner, const int *devices, int n_devices)
{
	std::vector<hipMemAccessDesc> v;
	hipMemAccessDesc acc = {};
	acc.location.type = hipMemLocationTypeDevice;
	acc.location.id = owner;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	v.push_back(acc);
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		n = 0;
	auto add = [&](int d) {
		int can = 0;
		if (d == owner || d < 0 || d >= n)
			return;
		for (auto &a : v)
			if (a.location.id == d)
				return;
		if (hipDeviceCanAccessPeer(&can, d, owner) == hipSuccess && can) {
			acc.location.id = d;
			v.push_back(acc);
		}
	};
	if (devices)
		for (int i = 0; i < n_devices; i++)
			add(devices[i]);
	else
		for (int d = 0; d < n; d++)
			add(d);
	(void)hipGetLastError();
	return v;
}

static bool set_access(void *va, size_t bytes, int owner, bool peers)
{
	std::vector<hipMemAccessDesc> v = access_descs(owner, peers ? nullptr : &owner, peers ? 0 : 1);
	if (hipMemSetAccess(va, bytes, v.data(), v.size()) == hipSuccess)
		return true;
	(void)hipGetLastError();
	// the peers were refused: the owner alone (a peer copy of this buffer is then staged by the runtime or fails loudly)
	return v.size() > 1 && hipMemSetAccess(va, bytes, v.data(), 1) == hipSuccess;
}

// A virtual range is never handed out twice.  hipMemAddressReserve without a hint returns the range it returned last time
// once that has been freed -- and a range that was freed, reserved again and mapped to OTHER physical chunks was accessed
// through stale translations: kernels wrote the old chunks, copies read the new ones (round 6, scripts/r06/sharded_stress.py:
// wrong results from the second alloc / free cycle of a placed batch on; never with fresh addresses).  Keeping the ranges
// reserved instead is no cure: the runtime then holds on to the physical memory as well.  So every reservation gets a
// hint below everything this process has reserved before (the runtime honours hints; 47 bits of address space hold
// thousands of arenas), and a reservation that lands on a used range nevertheless is given up for the next hint.
// DWT_HIP_VMM_REUSE_RANGES=1: no hints (the behaviour of rounds 4-5; for the soak that shows why not).
static hipError_t reserve_fresh(void **out, size_t bytes)
{
	static std::mutex mu;
	static std::vector<std::pair<uintptr_t, uintptr_t>> used; // [lo, hi) of every range ever reserved here
	static uintptr_t lowest = 0;
	static const bool reuse = getenv("DWT_HIP_VMM_REUSE_RANGES") && atoi(getenv("DWT_HIP_VMM_REUSE_RANGES")) > 0;
	std::lock_guard<std::mutex> lk(mu);
	const size_t gap = (size_t)1 << 30, al = (size_t)2 << 20;
	auto overlaps = [&](uintptr_t lo, uintptr_t hi) {
		for (auto &r : used)
			if (lo < r.second && r.first < hi)
				return true;
		return false;
	};
	hipError_t e = hipSuccess;
	for (int attempt = 0; attempt < 8; attempt++) {
		void *va = nullptr;
		void *hint = (lowest && !reuse) ? (void *)((lowest - bytes - gap * (attempt + 1)) / al * al) : nullptr;
		e = hipMemAddressReserve(&va, bytes, 0, hint, 0);
		if (e != hipSuccess) {
			(void)hipGetLastError();
			if (!hint)
				return e;
			continue; // (a hint the system cannot serve: the next one, further down)
		}
		const uintptr_t lo = (uintptr_t)va, hi = lo + bytes;
		if (!reuse && overlaps(lo, hi) && attempt < 7) {
			hipMemAddressFree(va, bytes);
			if (!lowest || lo < lowest)
				lowest = lo;
			continue;
		}
		used.push_back({lo, hi});
		if (!lowest || lo < lowest)
			lowest = lo;
		*out = va;
		return hipSuccess;
	}
	return e != hipSuccess ? e : hipErrorOutOfMemory;
}

static int vmm_release(void *va, VmmBuf &b, size_t mapped_pieces)
{
	for (size_t i = 0; i < mapped_pieces; i++)
		hipMemUnmap((char *)va + i * b.piece, b.piece);
	for (auto h : b.handles)
		hipMemRelease(h);
	// (the range goes back to the system; reserve_fresh never asks for these addresses again)
	if (b.arena) {
		std::lock_guard<std::mutex> lk(g_vmm_mu);
		if (--b.arena->live == 0) {
			hipMemAddressFree(b.arena->base, b.arena->bytes);
			delete b.arena;
		}
	} else if (va) {
		hipMemAddressFree(va, b.bytes);
	}
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
	if (reserve_fresh(&va, b.bytes) != hipSuccess) {
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
	if (ok)
		ok = set_access(va, b.bytes, g.device, false);
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
	if (reserve_fresh(&va, b.bytes) != hipSuccess) {
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
	if (ok)
		ok = set_access(va, b.bytes, g.device, false);
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
	// hipFree waits for the device by itself; unmapping does not: work still queued on the range would fault
	(void)hipDeviceSynchronize();
	vmm_release(va, b, b.handles.size());
	return true;
}


// ---- dwt_hip_alloc_batch: source, destination and LL scratch of a resident batch, placed by measurement ---------
// An ARENA -- most of the free memory of the card, mapped from 1 GiB physical chunks into one virtual
// range -- holds every candidate arrangement at once, so trying one costs a transform and nothing else:
//   1. the source at the start of the arena; the destination at every 4 GiB step behind it, timed with
//      ONE level (read stream against write stream, no scratch);
//   2. for the best destinations, the two LL bands at every 4 GiB step that overlaps neither, timed with
//      the `levels`-level transform of the whole batch;
//   3. the chunks under the best (destination, scratch) and under the source stay mapped where they are --
//      the four buffers ARE the arrangement that was measured -- and every other chunk is unmapped and goes
//      back to the system.
struct ArenaStats {
	int chunks = 0, dst_tried = 0, ll_tried = 0, dst_at = 0, ll_at = 0;
	double dst_best_ms = 0, dst_worst_ms = 0, ll_best_ms = 0, ll_worst_ms = 0, final_ms = 0, seconds = 0;
};
static thread_local ArenaStats g_arena;
// why the last dwt_hip_alloc_batch / _volumes of this thread did NOT search ("" = it did): dwt_hip_alloc_batch_note
static thread_local char g_arena_note[256] = "";
static void arena_note(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_arena_note, sizeof(g_arena_note), fmt, ap);
	va_end(ap);
}

static int alloc_batch_plain(size_t total, void **src, void **dst)
{
	*src = *dst = nullptr;
	if (hipMalloc(src, total) != hipSuccess || hipMalloc(dst, total) != hipSuccess) {
		(void)hipGetLastError();
		if (*src)
			hipFree(*src);
		*src = nullptr;
		return fail("hipMalloc(%zu) failed", total);
	}
	return 0;
}

// The search itself, for any workload with one source, one destination and a workspace of `n_ws` parts that
// lie one behind the other: sizes in bytes; `quick(src, dst, &ms)` times the source / destination relation
// alone (may be empty: then every destination position gets the full measurement's first workspace
// position), `full(src, dst, ws_parts, &ms)` the workload itself.  On success out[0] = source, out[1] =
// destination, out[2 ..] = the workspace parts, each a buffer of its own for dwt_hip_free.
// Returns 0, > 0 on error, -1 when the card has no room for a choice (the caller allocates plainly).
struct ArenaJob {
	size_t src_bytes, dst_bytes;
	std::vector<size_t> ws_bytes;
	std::function<int(char *, char *, double *)> quick;
	std::function<int(char *, char *, char *const *, double *)> full;
};

static int arena_place(const ArenaJob &job, void **out)
{
	const size_t C = (size_t)1 << 30;
	const size_t nS = (job.src_bytes + C - 1) / C, nD = (job.dst_bytes + C - 1) / C;
	std::vector<size_t> nW;
	size_t nL = 0;
	for (size_t b : job.ws_bytes) {
		nW.push_back((b + C - 1) / C);
		nL += nW.back();
	}
	size_t free_b = 0, total_b = 0;
	HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	const size_t reserve = (size_t)8 << 30;
	size_t n_chunks = free_b > reserve ? (free_b - reserve) / C : 0;
	// as much as the search can use: four times the workload, at least 96 GiB (the classes come in 16 GiB granules,
	// runs of one class can be 64 GiB long); option "place_max_gib" caps it for cards shared with other tenants
	n_chunks = std::min<size_t>(n_chunks, std::max<size_t>(96, 4 * (nS + nD + nL)));
	if (g.place_max_gib > 0)
		n_chunks = std::min<size_t>(n_chunks, (size_t)g.place_max_gib);
	if (n_chunks < 2 * (nS + nD + nL)) {
		arena_note("the card has %zu GiB to spare for the placement arena, a choice needs %zu: plain allocations", n_chunks, 2 * (nS + nD + nL));
		return -1;
	}
	const auto t_start = std::chrono::steady_clock::now();
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = g.device;
	std::vector<hipMemGenericAllocationHandle_t> chunk;
	for (size_t i = 0; i < n_chunks; i++) {
		hipMemGenericAllocationHandle_t h;
		if (hipMemCreate(&h, C, &prop, 0) != hipSuccess) {
			(void)hipGetLastError();
			break;
		}
		chunk.push_back(h);
	}
	auto release_chunks = [&]() {
		for (auto h : chunk)
			hipMemRelease(h);
		chunk.clear();
	};
	n_chunks = chunk.size();
	if (n_chunks < 2 * (nS + nD + nL)) {
		release_chunks();
		arena_note("only %zu GiB of physical chunks could be created, a choice needs %zu: plain allocations", n_chunks, 2 * (nS + nD + nL));
		return -1;
	}
	char *arena = nullptr;
	if (reserve_fresh((void **)&arena, n_chunks * C) != hipSuccess) {
		release_chunks();
		return fail("hipMemAddressReserve(%zu) failed", n_chunks * C);
	}
	size_t mapped = 0;
	bool ok = true;
	for (size_t i = 0; i < n_chunks && ok; i++) {
		ok = hipMemMap(arena + i * C, C, 0, chunk[i], 0) == hipSuccess;
		if (ok)
			mapped++;
	}
	ok = ok && set_access(arena, n_chunks * C, g.device, false); // (the search runs on this device alone)
	auto drop_arena = [&]() {
		for (size_t i = 0; i < mapped; i++)
			hipMemUnmap(arena + i * C, C);
		hipMemAddressFree(arena, n_chunks * C);
		release_chunks();
	};
	if (!ok) {
		drop_arena();
		return fail("mapping the placement arena failed: %s", hipGetErrorString(hipGetLastError()));
	}
	const double t_mapped = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	int rc = 0;
	const size_t step = 4;
	auto ws_at = [&](size_t P, std::vector<char *> &parts) {
		parts.clear();
		for (size_t n : nW) {
			parts.push_back(arena + P * C);
			P += n;
		}
	};
	// first workspace position that overlaps neither the source nor a destination at D
	auto first_ws = [&](size_t D) {
		for (size_t P = 0; P + nL <= n_chunks; P += step)
			if (P >= nS && !(P < D + nD && D < P + nL))
				return P;
		return n_chunks;
	};
	std::vector<char *> parts;
	// 1. destinations against the source
	std::vector<std::pair<double, size_t>> dsts;
	for (size_t D = nS; D + nD <= n_chunks && !rc; D += step) {
		double ms = 0;
		if (job.quick) {
			rc = job.quick(arena, arena + D * C, &ms);
		} else {
			const size_t P = first_ws(D);
			if (P >= n_chunks)
				continue;
			ws_at(P, parts);
			rc = job.full(arena, arena + D * C, parts.data(), &ms);
		}
		dsts.push_back({ms, D});
	}
	const double t_dst = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	double best_ms = 1e30, worst_ws = 0;
	size_t best_D = nS, best_P = nS + nD;
	int ws_tried = 0;
	if (!rc && !dsts.empty() && nW.empty()) {
		// no workspace: the destination search is the whole search
		for (auto &d : dsts)
			if (d.first < best_ms) {
				best_ms = d.first;
				best_D = d.second;
			}
		best_P = 0;
	} else if (!rc && !dsts.empty()) {
		std::vector<std::pair<double, size_t>> order = dsts;
		std::sort(order.begin(), order.end());
		const size_t top = std::min<size_t>(order.size(), 3);
		for (size_t t = 0; t < top && !rc; t++) {
			const size_t D = order[t].second;
			// 2. the workspace wherever it overlaps neither the source nor this destination
			for (size_t P = 0; P + nL <= n_chunks && !rc; P += step) {
				if (P < nS || (P < D + nD && D < P + nL))
					continue;
				ws_at(P, parts);
				double ms = 0;
				rc = job.full(arena, arena + D * C, parts.data(), &ms);
				ws_tried++;
				worst_ws = std::max(worst_ws, ms);
				if (ms < best_ms) {
					best_ms = ms;
					best_D = D;
					best_P = P;
				}
			}
		}
	}
	hipStreamSynchronize(g.stream);
	if (rc || best_ms >= 1e30) {
		drop_arena();
		if (!rc)
			arena_note("no arrangement could be measured in an arena of %zu GiB: plain allocations", n_chunks);
		return rc ? rc : -1;
	}
	// 3. the chosen chunks stay where they are mapped -- the buffers ARE the arrangement that was measured,
	//    physical chunks and virtual addresses alike (mapping the same chunks again at other addresses gave
	//    other rates: 8.5-8.8 ms for arrangements that had measured 7.5-7.7) -- and every other chunk is
	//    unmapped and returned; the address reservation goes when the last of the buffers is freed
	struct Part {
		size_t first, count;
	};
	std::vector<Part> keep = {{0, nS}, {best_D, nD}};
	{
		size_t P = best_P;
		for (size_t n : nW) {
			keep.push_back({P, n});
			P += n;
		}
	}
	std::vector<char> used(n_chunks, 0);
	VmmArena *ar = new VmmArena{arena, n_chunks * C, (int)keep.size()};
	for (size_t k = 0; k < keep.size(); k++) {
		VmmBuf b;
		b.piece = C;
		b.bytes = keep[k].count * C;
		b.arena = ar;
		for (size_t i = 0; i < keep[k].count; i++) {
			b.handles.push_back(chunk[keep[k].first + i]);
			used[keep[k].first + i] = 1;
		}
		out[k] = arena + keep[k].first * C;
		std::lock_guard<std::mutex> lk(g_vmm_mu);
		g_vmm[out[k]] = std::move(b);
	}
	for (size_t i = 0; i < n_chunks; i++)
		if (!used[i]) {
			hipMemUnmap(arena + i * C, C);
			hipMemRelease(chunk[i]);
		}
	// the arrangement as the caller will use it, timed once more
	double final_ms = 0;
	ws_at(best_P, parts);
	job.full(arena, arena + best_D * C, parts.data(), &final_ms);
	std::sort(dsts.begin(), dsts.end());
	g_arena = ArenaStats();
	g_arena.chunks = (int)n_chunks;
	g_arena.final_ms = final_ms;
	g_arena.dst_tried = (int)dsts.size();
	g_arena.ll_tried = ws_tried;
	g_arena.dst_at = (int)best_D;
	g_arena.ll_at = (int)best_P;
	g_arena.dst_best_ms = dsts.front().first;
	g_arena.dst_worst_ms = dsts.back().first;
	g_arena.ll_best_ms = best_ms;
	g_arena.ll_worst_ms = worst_ws;
	g_arena.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	if (getenv("DWT_HIP_PLACE_VERBOSE"))
		fprintf(stderr, "arena_place: %zu chunks mapped after %.2f s, destinations done after %.2f s, all after %.2f s\n", n_chunks, t_mapped, t_dst,
			g_arena.seconds);
	return 0;
}

// the context's workspace pointed into the arena for one trial (never grown or freed meanwhile)
struct TrialLL {
	TrialLL(char *b0, size_t n0, char *b1, size_t n1)
	{
		g.ll[0] = b0;
		g.ll[1] = b1;
		g.ll_bytes[0] = n0;
		g.ll_bytes[1] = n1;
		g.ll_external = true;
	}
	~TrialLL()
	{
		g.ll_external = false;
		g.ll[0] = g.ll[1] = nullptr;
		g.ll_bytes[0] = g.ll_bytes[1] = 0;
	}
};

static int alloc_batch_arena(Wavelet w, int n_images, int size_x, int size_y, int levels, void **src_out, void **dst_out)
{
	const int es = elem_size(w);
	const size_t pitch = (size_t)size_x * es, img = pitch * size_y, total = img * n_images;
	const Geom ge{size_x, size_y, size_x, size_y};
	ArenaJob job;
	job.src_bytes = job.dst_bytes = total;
	job.ws_bytes = {ll_band_bytes(ge, 0, n_images, es), ll_band_bytes(ge, 1, n_images, es)};
	const size_t C = (size_t)1 << 30;
	const size_t cap0 = (job.ws_bytes[0] + C - 1) / C * C, cap1 = (job.ws_bytes[1] + C - 1) / C * C;
	// the context's own scratch makes room for the arena's candidates
	hipStreamSynchronize(g.stream);
	for (int b = 0; b < 2; b++) {
		if (g.ll[b])
			dev_free(g.ll[b]);
		g.ll[b] = nullptr;
		g.ll_bytes[b] = 0;
	}
	job.quick = [&](char *s, char *d, double *ms) {
		// (a one-level call never touches the scratch; the context is pointed at the source region so that
		// it does not allocate one of its own meanwhile)
		TrialLL t(s, total, s, total);
		return timed_forward(w, Img{s, (long)pitch, es}, Img{d, (long)pitch, es}, ge, 1, n_images, (long)img, (long)img, ms);
	};
	job.full = [&](char *s, char *d, char *const *ws, double *ms) {
		TrialLL t(ws[0], cap0, ws[1], cap1);
		return timed_forward(w, Img{s, (long)pitch, es}, Img{d, (long)pitch, es}, ge, levels, n_images, (long)img, (long)img, ms);
	};
	void *out[4] = {nullptr, nullptr, nullptr, nullptr};
	const int rc = arena_place(job, out);
	if (rc)
		return rc;
	g.ll[0] = out[2];
	g.ll[1] = out[3];
	g.ll_bytes[0] = cap0;
	g.ll_bytes[1] = cap1;
	*src_out = out[0];
	*dst_out = out[1];
	return 0;
}

// workspace bytes (each of the two pools) of an out-of-place 3-D call: the dense level inputs / outputs of levels >= 1
static size_t vol_pool_bytes(int nx, int ny, int nz, int levels)
{
	size_t pool = 0;
	for (int j = 1; j < levels; j++) {
		const size_t lx = ceil_div_pow2(nx, j), ly = ceil_div_pow2(ny, j), lz = ceil_div_pow2(nz, j);
		pool += (size_t)align_up((long)lx, 4) * ly * lz;
	}
	return pool * 4;
}

static int timed_volume_op(const char *s, char *d, size_t sy, size_t sz, int nx, int ny, int nz, int levels, double *ms)
{
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	int rc = 0;
	for (int r = 0; r < 2 && !rc; r++) {
		hipEventRecord(e0, g.stream);
		rc = dwt_hip_transform3d_op(s, d, sy, sz, nx, ny, nz, levels);
		hipEventRecord(e1, g.stream);
	}
	float t = 0;
	if (!rc && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess))
		rc = fail("timing a placement trial failed: %s", hipGetErrorString(hipGetLastError()));
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	*ms = t;
	return rc;
}

static int alloc_volumes_arena(int nx, int ny, int nz, int levels, void **src_out, void **dst_out)
{
	const size_t sy = (size_t)nx * 4, sz = sy * ny, total = sz * nz;
	const size_t pool = vol_pool_bytes(nx, ny, nz, levels);
	const size_t C = (size_t)1 << 30;
	const size_t cap = (pool + C - 1) / C * C;
	ArenaJob job;
	job.src_bytes = job.dst_bytes = total;
	if (pool)
		job.ws_bytes = {pool, pool};
	hipStreamSynchronize(g.stream);
	void **own[2] = {&g.host_a, &g.host_b};
	size_t *own_bytes[2] = {&g.host_a_bytes, &g.host_b_bytes};
	for (int b = 0; b < 2; b++) {
		if (*own[b])
			dev_free(*own[b]);
		*own[b] = nullptr;
		*own_bytes[b] = 0;
	}
	auto point = [&](char *a, char *b, size_t n) {
		g.host_a = a;
		g.host_b = b;
		g.host_a_bytes = g.host_b_bytes = n;
	};
	if (pool) // (a one-level call has no workspace: the quick measurement of the destinations)
		job.quick = [&](char *s, char *d, double *ms) { return timed_volume_op(s, d, sy, sz, nx, ny, nz, 1, ms); };
	job.full = [&](char *s, char *d, char *const *ws, double *ms) {
		if (pool)
			point(ws[0], ws[1], cap);
		const int rc = timed_volume_op(s, d, sy, sz, nx, ny, nz, levels, ms);
		point(nullptr, nullptr, 0);
		return rc;
	};
	void *out[4] = {nullptr, nullptr, nullptr, nullptr};
	const int rc = arena_place(job, out);
	if (rc)
		return rc;
	if (pool)
		point((char *)out[2], (char *)out[3], cap);
	*src_out = out[0];
	*dst_out = out[1];
	return 0;
}

// the mapped buffer that contains p (null: p is not in one)
static bool vmm_find(const void *p, void **base, size_t *bytes)
{
	std::lock_guard<std::mutex> lk(g_vmm_mu);
	auto it = g_vmm.upper_bound((void *)p);
	if (it == g_vmm.begin())
		return false;
	--it;
	if ((const char *)p >= (const char *)it->first + it->second.bytes)
		return false;
	*base = it->first;
	*bytes = it->second.bytes;
	return true;
}

// 0: every device named may now read and write the buffer p lies in (a plain allocation: nothing to do here, the
// devices enable peer access themselves); 1: a mapped buffer could not be granted
int grant_range(const void *p, int owner, const int *devices, int n_devices)
{
	void *base = nullptr;
	size_t bytes = 0;
	if (!vmm_find(p, &base, &bytes))
		return 0;
	std::vector<hipMemAccessDesc> v = access_descs(owner, devices, n_devices);
	if (hipMemSetAccess(base, bytes, v.data(), v.size()) != hipSuccess)
		return fail("hipMemSetAccess for %zu device(s) on a placed buffer failed: %s", v.size(), hipGetErrorString(hipGetLastError()));
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
// in microseconds; negative on error.  Diagnostic entry (scripts/archive/probes/r04_*).
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

// Source and destination of a resident batch, allocated WITH their placement -- the analogue of the
// reference's dwt_util_get_opt_stride / dwt_util_get_stride for its callers (src/libdwt.c:20641-20707).  Dense
// pitch (size_x elements), images size_x * size_y elements apart; the LL scratch stays with the calling
// thread's context.  Batches below "place_min_mib" MiB, "place_tries" < 2, a card too full for a choice:
// plain allocations (the first forward call then runs the library's own scratch search).
int dwt_hip_alloc_batch(int wavelet, int n_images, int size_x, int size_y, int levels, void **src_out, void **dst_out)
{
	if (check_inited())
		return 1;
	if (wavelet < 0 || wavelet > 5 || n_images < 1 || n_images > 65535 || size_x < 1 || size_y < 1 || !src_out || !dst_out)
		return fail("dwt_hip_alloc_batch: bad argument");
	const Wavelet w = (Wavelet)wavelet;
	const int es = elem_size(w);
	g_elems_are_32bit = es == 4;
	const size_t total = (size_t)size_x * es * size_y * n_images;
	const Geom ge{size_x, size_y, size_x, size_y};
	g_arena = ArenaStats();
	g_arena_note[0] = 0;
	const bool search = g.place_tries >= 2 && total >= ((size_t)g.place_min_mib << 20) && ge.Wo(2) >= 2 && ge.Ho(2) >= 2 && !g.ll_external &&
		!g.force_generic && es == 4 && !stream_is_capturing();
	if (search) {
		const int rc = alloc_batch_arena(w, n_images, size_x, size_y, levels, src_out, dst_out);
		if (rc >= 0)
			return rc;
	} else {
		arena_note("no placement search for this batch (%zu MiB; \"place_min_mib\" %d, \"place_tries\" %d, 32-bit elements and at least two levels needed): plain allocations",
			total >> 20, g.place_min_mib, g.place_tries);
	}
	return alloc_batch_plain(total, src_out, dst_out);
}

// The same for the volumes of an out-of-place 3-D call (dwt_hip_transform3d_op / cdf97_3f_op_sep_horizontal_s on
// device volumes): dense strides (rows nx floats, slices nx * ny floats), the workspace of the deeper levels
// placed with them and kept by the calling thread's context.
int dwt_hip_alloc_volumes(int size_x, int size_y, int size_z, int levels, void **src_out, void **dst_out)
{
	if (check_inited())
		return 1;
	if (size_x < 1 || size_y < 1 || size_z < 1 || levels < 1 || levels > 24 || !src_out || !dst_out)
		return fail("dwt_hip_alloc_volumes: bad argument");
	const size_t total = (size_t)size_x * size_y * size_z * 4;
	g_arena = ArenaStats();
	g_arena_note[0] = 0;
	if (g.place_tries >= 2 && total >= ((size_t)g.place_min_mib << 20) && !g.force_generic && !stream_is_capturing()) {
		const int rc = alloc_volumes_arena(size_x, size_y, size_z, levels, src_out, dst_out);
		if (rc >= 0)
			return rc;
	} else {
		arena_note("no placement search for these volumes (%zu MiB; \"place_min_mib\" %d, \"place_tries\" %d): plain allocations", total >> 20,
			g.place_min_mib, g.place_tries);
	}
	return alloc_batch_plain(total, src_out, dst_out);
}

// what the last dwt_hip_alloc_batch of this thread measured (all zero: plain allocations)
void dwt_hip_alloc_batch_report(int *chunks, int *dst_tried, int *ll_tried, int *dst_at, int *ll_at, double *ms /* [5]: dst best, dst worst (one level), scratch best, scratch worst (whole call), the kept arrangement timed again */, double *seconds)
{
	if (chunks) *chunks = g_arena.chunks;
	if (dst_tried) *dst_tried = g_arena.dst_tried;
	if (ll_tried) *ll_tried = g_arena.ll_tried;
	if (dst_at) *dst_at = g_arena.dst_at;
	if (ll_at) *ll_at = g_arena.ll_at;
	if (ms) {
		ms[0] = g_arena.dst_best_ms;
		ms[1] = g_arena.dst_worst_ms;
		ms[2] = g_arena.ll_best_ms;
		ms[3] = g_arena.ll_worst_ms;
		ms[4] = g_arena.final_ms;
	}
	if (seconds) *seconds = g_arena.seconds;
}

// "" when the last dwt_hip_alloc_batch / _volumes of this thread ran its search; otherwise why it fell back to plain allocations
const char *dwt_hip_alloc_batch_note(void) { return g_arena_note; }

// Makes a device buffer reachable from other devices of this process (peer copies): buffers of dwt_hip_alloc_batch /
// _volumes / _malloc_mapped get the access on their mapping (they are created with access for their owner alone;
// dwt_hip_transform2d_batch_sharded grants its slots' devices by itself), plain allocations through
// hipDeviceEnablePeerAccess from each of the devices.  `ptr` may point anywhere into the buffer.  0 = every device
// named can reach it.
int dwt_hip_grant_access(void *ptr, const int *devices, int n_devices)
{
	if (check_inited())
		return 1;
	if (!ptr || !devices || n_devices < 1)
		return fail("dwt_hip_grant_access: bad argument");
	hipPointerAttribute_t at;
	if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
		(void)hipGetLastError();
		return fail("dwt_hip_grant_access: not a device pointer");
	}
	const int owner = at.device;
	int bad = 0;
	for (int i = 0; i < n_devices; i++) {
		int can = devices[i] == owner;
		if (!can && (hipDeviceCanAccessPeer(&can, devices[i], owner) != hipSuccess || !can))
			bad++;
	}
	(void)hipGetLastError();
	if (bad)
		return fail("%d of the %d device(s) cannot reach device %d's memory as peers", bad, n_devices, owner);
	void *base = nullptr;
	size_t bytes = 0;
	if (vmm_find(ptr, &base, &bytes))
		return grant_range(ptr, owner, devices, n_devices);
	// a plain allocation: peer access is a property of the device pair, enabled from each of the devices
	int cur = 0;
	HIP_TRY(hipGetDevice(&cur));
	for (int i = 0; i < n_devices; i++) {
		if (devices[i] == owner)
			continue;
		if (hipSetDevice(devices[i]) == hipSuccess) {
			const hipError_t e = hipDeviceEnablePeerAccess(owner, 0);
			if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
				bad++;
		} else {
			bad++;
		}
		(void)hipGetLastError();
	}
	HIP_TRY(hipSetDevice(cur));
	if (bad)
		return fail("peer access to device %d could not be enabled from %d device(s)", owner, bad);
	return 0;
}

void dwt_hip_free_mapped(void *p)
{
	if (p)
		vmm_free(p);
}

} // extern "C"
#pragma GCC visibility pop
