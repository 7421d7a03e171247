// dwt_multi.hip -- one process, several GPUs: the batch split behind the C-ABI ("single
// process, 8 devices, one host thread per device").
//
// Images are independent, so a batch of B images shards over G devices with NO collective in the
// transform: image b belongs to slot b*G/B, i.e. slot k owns images [ceil(k*B/G), ceil((k+1)*B/G)).
// Two entries:
//   dwt_hip_transform2d_batch_multi   -- the shards are RESIDENT: shard k already lies in the memory of
//       devices[k]; every shard is transformed where it lies, all at once, nothing crosses xGMI.  This is the
//       case the >= 7x scaling claim is about.
//   dwt_hip_transform2d_batch_sharded -- the whole batch lies on the calling thread's device (slot 0, whose
//       shard is transformed where it lies); every other slot pulls its shard over xGMI (hipMemcpyPeerAsync),
//       transforms it and pushes the coefficients back, piece by piece so that the copies run beside the
//       transform: root-egress bound (about 14 ms each way for 16 GiB against 1 ms of transform).
// Every slot but the caller's is a persistent host thread with a context of its own (device binding, stream,
// workspace, measured tile heights, staging -- all kept between calls).
//
// (The Python harness does the same split across PROCESSES with RCCL point-to-point transfers,
// libdwt_amd/batch.py; this file is what a C caller of libdwt.h gets.)
#include "dwt_backend.h"
#include "dwt_host_pools.h"

#include <memory>
#include <string>

namespace dwtb {

// a slot's worker (dwt_host_pools.h: a persistent host thread running one job at a time) with the slot's staging
struct SlotWorker : SlotThread {
	SlotWorker() : SlotThread(dwt_hip_last_error) {}
	// staging of this slot (owned by the worker thread's device): two source and two result pieces
	void *stage[4] = {nullptr, nullptr, nullptr, nullptr};
	size_t stage_bytes[4] = {0, 0, 0, 0};
	hipStream_t cin = nullptr, cout = nullptr;
	hipEvent_t ev_in[2] = {}, ev_done[2] = {}, ev_out[2] = {};
	int device = -1;
};

static std::mutex g_slots_mu;             // one multi-device call at a time (the workers are shared)
static std::vector<SlotWorker *> g_slots; // slot k -> its worker (made on first use, never destroyed)

static SlotWorker *worker(int k)
{
	if ((int)g_slots.size() <= k)
		g_slots.resize(k + 1, nullptr);
	if (!g_slots[k])
		g_slots[k] = new SlotWorker();
	return g_slots[k];
}

// first image of slot k when `batch` images are dealt to G slots, image b -> slot b*G/batch
static int shard_lo(int k, int batch, int G) { return (int)(((long)k * batch + G - 1) / G); }

// the options of the calling thread's context that a slot's context takes over for the call
struct SlotOpts {
	SweepTuning tune;
	int force_generic, fma, tune_tiles, place_tries, place_min_mib, place_max_gib, tune_in_call;
	static SlotOpts of_caller()
	{
		return SlotOpts{g.tune, g.force_generic, g.fma, g.tune_tiles, g.place_tries, g.place_min_mib, g.place_max_gib, may_measure() ? 1 : 0};
	}
	void apply() const
	{
		if (memcmp(&g.tune, &tune, sizeof(tune)) || g.force_generic != force_generic || g.fma != fma)
			g.tile_cache.clear();
		g.tune = tune;
		g.force_generic = force_generic;
		g.fma = fma;
		g.tune_tiles = tune_tiles;
		g.place_tries = place_tries;
		g.place_min_mib = place_min_mib;
		g.place_max_gib = place_max_gib;
		g.tune_in_call = tune_in_call;
	}
};

// a slot of the SHARDED call: n images of the root's batch, through this device and back
static int slot_job(SlotWorker *wk, const SlotOpts &opts, int device, int root_device, int wavelet, int inverse, const char *src, char *dst,
	size_t batch_stride, int n, int stride_x, int size_x, int size_y, int j_in, int *j_out, bool dense)
{
	if (dwt_hip_set_device(device))
		return 1;
	opts.apply();
	if (wk->device != device) { // the slot moved to another device: its staging, streams and events are on the old one
		for (int k = 0; k < 4; k++) {
			if (wk->stage[k])
				dev_free(wk->stage[k]);
			wk->stage[k] = nullptr;
			wk->stage_bytes[k] = 0;
		}
		if (wk->cin) {
			hipStreamDestroy(wk->cin);
			hipStreamDestroy(wk->cout);
			for (int k = 0; k < 2; k++) {
				hipEventDestroy(wk->ev_in[k]);
				hipEventDestroy(wk->ev_done[k]);
				hipEventDestroy(wk->ev_out[k]);
			}
			wk->cin = wk->cout = nullptr;
			(void)hipGetLastError();
		}
		wk->device = device;
		if (device != root_device) {
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, device, root_device) == hipSuccess && can)
				(void)hipDeviceEnablePeerAccess(root_device, 0);
			(void)hipGetLastError(); // already enabled / not possible: the copies are staged by the runtime then
		}
	}
	if (!wk->cin) {
		// all or nothing: a slot with half of its streams and events would use the missing ones for ever
		hipStream_t cin = nullptr, cout = nullptr;
		hipEvent_t ev[6] = {};
		bool ok = hipStreamCreateWithFlags(&cin, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&cout, hipStreamNonBlocking) == hipSuccess;
		for (int k = 0; k < 6 && ok; k++)
			ok = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) == hipSuccess;
		if (!ok) {
			const hipError_t e = hipGetLastError();
			for (hipEvent_t x : ev)
				if (x)
					hipEventDestroy(x);
			if (cin)
				hipStreamDestroy(cin);
			if (cout)
				hipStreamDestroy(cout);
			(void)hipGetLastError();
			return fail("slot on device %d: creating its copy streams / events failed: %s", device, hipGetErrorString(e));
		}
		wk->cin = cin;
		wk->cout = cout;
		for (int k = 0; k < 2; k++) {
			wk->ev_in[k] = ev[3 * k];
			wk->ev_done[k] = ev[3 * k + 1];
			wk->ev_out[k] = ev[3 * k + 2];
		}
	}
	// The shard crosses in up to four pieces through two pairs of staging buffers: piece c+1 comes in and piece c-1
	// goes out while piece c is transformed (round 4 staged the whole shard twice and ran copy in, transform, copy out
	// strictly one after the other).  Larger pieces first, so that the workspace is sized by the first call.
	const int n_pieces = n < 4 ? n : 4;
	const int m = (n + n_pieces - 1) / n_pieces;
	for (int k = 0; k < 4; k++)
		if (grow(&wk->stage[k], &wk->stage_bytes[k], (size_t)m * batch_stride))
			return 1;
	hipStream_t st = g.stream;
	int j = j_in;
	// whatever fails half way: the peer copies still in flight write into the caller's `dst` and read this slot's
	// staging, which the next call reuses -- drain the three streams before the error is reported
	struct Drain {
		hipStream_t a, b, c;
		bool armed = true;
		~Drain()
		{
			if (!armed)
				return;
			(void)hipStreamSynchronize(a);
			(void)hipStreamSynchronize(b);
			(void)hipStreamSynchronize(c);
			(void)hipGetLastError();
		}
	} drain{wk->cin, wk->cout, st};
	for (int c = 0; c < n_pieces; c++) {
		const int a = shard_lo(c, n, n_pieces), cnt = shard_lo(c + 1, n, n_pieces) - a, s = c & 1;
		const size_t off = (size_t)a * batch_stride, bytes = (size_t)cnt * batch_stride;
		char *in = (char *)wk->stage[s], *out = (char *)wk->stage[2 + s];
		if (c >= 2) {
			HIP_TRY(hipStreamWaitEvent(wk->cin, wk->ev_done[s], 0)); // piece c-2 has been read
			HIP_TRY(hipStreamWaitEvent(wk->cin, wk->ev_out[s], 0));  // ... and its result has left
		}
		HIP_TRY(hipMemcpyPeerAsync(in, device, src + off, root_device, bytes, wk->cin));
		if (!dense) // bytes between the frames of the destination keep their values: bring them along
			HIP_TRY(hipMemcpyPeerAsync(out, device, dst + off, root_device, bytes, wk->cin));
		HIP_TRY(hipEventRecord(wk->ev_in[s], wk->cin));
		HIP_TRY(hipStreamWaitEvent(st, wk->ev_in[s], 0));
		if (c >= 2)
			HIP_TRY(hipStreamWaitEvent(st, wk->ev_out[s], 0));
		j = j_in;
		if (dwt_hip_transform2d_batch(wavelet, inverse, in, out, batch_stride, cnt, stride_x, size_x, size_y, &j))
			return 1;
		HIP_TRY(hipEventRecord(wk->ev_done[s], st));
		HIP_TRY(hipStreamWaitEvent(wk->cout, wk->ev_done[s], 0));
		HIP_TRY(hipMemcpyPeerAsync(dst + off, root_device, out, device, bytes, wk->cout));
		HIP_TRY(hipEventRecord(wk->ev_out[s], wk->cout));
	}
	HIP_TRY(hipStreamSynchronize(wk->cout));
	HIP_TRY(hipStreamSynchronize(st));
	drain.armed = false;
	*j_out = j;
	return 0;
}

// a slot of the RESIDENT call: the shard lies on this device already
static int resident_job(const SlotOpts *opts, int device, int wavelet, int inverse, bool tune, const void *src, void *dst, size_t batch_stride, int n,
	int stride_x, int size_x, int size_y, int j_in, int *j_out)
{
	if (dwt_hip_set_device(device))
		return 1;
	if (opts)
		opts->apply();
	for (const void *p : {src, (const void *)dst}) {
		hipPointerAttribute_t at;
		if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice && at.device != device)
			return fail("a buffer of the shard lies in the memory of device %d, not of device %d", at.device, device);
		(void)hipGetLastError(); // (host pointers and the like are refused by the transform entry itself)
	}
	// The shard's producers may have run on any stream of this device (the caller's threads, torch side streams): the
	// slot's own stream is not ordered behind them, so the device is drained first.  Once per shard and call.
	HIP_TRY(hipDeviceSynchronize());
	int j = j_in;
	if (tune) {
		if (dwt_hip_tune(wavelet, inverse, src, dst, batch_stride, n, stride_x, size_x, size_y, j_in))
			return 1;
	} else if (dwt_hip_transform2d_batch(wavelet, inverse, src, dst, batch_stride, n, stride_x, size_x, size_y, &j)) {
		return 1;
	}
	HIP_TRY(hipStreamSynchronize(g.stream));
	*j_out = j;
	return 0;
}

static int batch_multi(bool tune, int wavelet, int inverse, const void *const *srcs, void *const *dsts, const int *counts, const int *devices, int n_shards,
	size_t batch_stride, int stride_x, int size_x, int size_y, int *j)
{
	if (check_inited())
		return 1;
	if (!srcs || !dsts || !counts || !devices || !j || n_shards < 1 || n_shards > 64)
		return fail("dwt_hip_transform2d_batch_multi: bad argument");
	if (wavelet < 0 || wavelet > 5)
		return fail("unknown wavelet %d", wavelet);
	const int ndev = dwt_hip_device_count();
	for (int k = 0; k < n_shards; k++) {
		if (devices[k] < 0 || devices[k] >= ndev)
			return fail("devices[%d] = %d: the process sees %d device(s)", k, devices[k], ndev);
		if (counts[k] < 0 || (counts[k] > 0 && (!srcs[k] || !dsts[k])))
			return fail("shard %d: bad count or null buffer", k);
	}
	std::lock_guard<std::mutex> turn(g_slots_mu);
	// (every slot drains its own device before it reads its shard: resident_job)
	const SlotOpts opts = SlotOpts::of_caller();
	const int j_in = *j;
	std::vector<int> js(n_shards, j_in);
	// the calling thread takes the first shard that lies on its own device; every other shard has a worker
	int mine = -1;
	for (int k = 0; k < n_shards && mine < 0; k++)
		if (counts[k] > 0 && devices[k] == g.device)
			mine = k;
	for (int k = 0; k < n_shards; k++) {
		if (k == mine || counts[k] == 0)
			continue;
		const void *s = srcs[k];
		void *d = dsts[k];
		const int dev = devices[k], n = counts[k];
		int *jo = &js[k];
		worker(k)->submit([=] { return resident_job(&opts, dev, wavelet, inverse, tune, s, d, batch_stride, n, stride_x, size_x, size_y, j_in, jo); });
	}
	int rc = 0;
	std::string first_err;
	if (mine >= 0) {
		rc = resident_job(nullptr, g.device, wavelet, inverse, tune, srcs[mine], dsts[mine], batch_stride, counts[mine], stride_x, size_x, size_y, j_in, &js[mine]);
		if (rc)
			first_err = dwt_hip_last_error();
	}
	for (int k = 0; k < n_shards; k++) {
		if (k == mine || counts[k] == 0)
			continue;
		std::string err;
		const int r = g_slots[k]->wait(err);
		if (r && !rc) {
			rc = r;
			first_err = "shard " + std::to_string(k) + " (device " + std::to_string(devices[k]) + "): " + err;
		}
	}
	if (rc)
		return fail("%s", first_err.c_str());
	for (int k = 0; k < n_shards; k++)
		if (counts[k] > 0) {
			*j = js[k];
			break;
		}
	return 0;
}

} // namespace dwtb

using namespace dwtb;

#pragma GCC visibility push(default)
extern "C" {

int dwt_hip_transform2d_batch_multi(int wavelet, int inverse, const void *const *srcs, void *const *dsts, const int *counts, const int *devices,
	int n_shards, size_t batch_stride, int stride_x, int size_x, int size_y, int *j)
{
	return batch_multi(false, wavelet, inverse, srcs, dsts, counts, devices, n_shards, batch_stride, stride_x, size_x, size_y, j);
}

int dwt_hip_tune_batch_multi(int wavelet, int inverse, const void *const *srcs, void *const *dsts, const int *counts, const int *devices, int n_shards,
	size_t batch_stride, int stride_x, int size_x, int size_y, int levels)
{
	int j = levels;
	return batch_multi(true, wavelet, inverse, srcs, dsts, counts, devices, n_shards, batch_stride, stride_x, size_x, size_y, &j);
}

void dwt_hip_shard_bounds(int batch, int n_slots, int slot, int *first, int *count)
{
	const int G = n_slots < 1 ? 1 : n_slots;
	// a slot outside [0, n_slots) or a negative batch owns nothing: first = batch (clamped at 0), count = 0
	if (batch < 0)
		batch = 0;
	if (slot < 0 || slot >= G) {
		if (first)
			*first = batch;
		if (count)
			*count = 0;
		return;
	}
	const int a = shard_lo(slot, batch, G), b = shard_lo(slot + 1, batch, G);
	if (first)
		*first = a;
	if (count)
		*count = b - a;
}

int dwt_hip_transform2d_batch_sharded(int wavelet, int inverse, const void *src, void *dst, size_t batch_stride, int batch,
	int stride_x, int size_x, int size_y, int *j, const int *devices, int n_devices)
{
	if (check_inited())
		return 1;
	if (!devices || n_devices < 1 || n_devices > 64 || !j || !src || !dst || batch < 1)
		return fail("dwt_hip_transform2d_batch_sharded: bad argument");
	const int root = g.device, ndev = dwt_hip_device_count();
	if (devices[0] != root)
		return fail("devices[0] must be the calling thread's device (%d), which holds the batch; got %d", root, devices[0]);
	for (int k = 0; k < n_devices; k++)
		if (devices[k] < 0 || devices[k] >= ndev)
			return fail("devices[%d] = %d: the process sees %d device(s)", k, devices[k], ndev);
	if (wavelet < 0 || wavelet > 5)
		return fail("unknown wavelet %d", wavelet);
	const int es = elem_size((Wavelet)wavelet);
	const int G = n_devices < batch ? n_devices : batch; // never more slots than images
	const bool dense = (size_t)stride_x == (size_t)size_x * es && batch_stride == (size_t)stride_x * size_y;
	std::lock_guard<std::mutex> turn(g_slots_mu);
	// everything queued on the caller's stream so far (the batch's producers) before the other devices read it
	HIP_TRY(hipStreamSynchronize(g.stream));
	// a batch from dwt_hip_alloc_batch is mapped through the virtual-memory API with access for its owner alone:
	// the slots' devices are granted now (hipDeviceEnablePeerAccess, which the slots call, does not cover such ranges)
	if (grant_range(src, root, devices, G) || grant_range(dst, root, devices, G))
		return 1;
	const SlotOpts opts = SlotOpts::of_caller();
	std::vector<int> js(G, *j);
	const int j_in = *j;
	auto lo = [&](int k) { return shard_lo(k, batch, G); };
	for (int k = 1; k < G; k++) {
		SlotWorker *wk = worker(k);
		const int a = lo(k), n = lo(k + 1) - a, dev = devices[k];
		const char *s = (const char *)src + (size_t)a * batch_stride;
		char *d = (char *)dst + (size_t)a * batch_stride;
		int *jo = &js[k];
		wk->submit([=] { return slot_job(wk, opts, dev, root, wavelet, inverse, s, d, batch_stride, n, stride_x, size_x, size_y, j_in, jo, dense); });
	}
	// slot 0: where the batch lies, on the caller's own context and stream
	int rc = dwt_hip_transform2d_batch(wavelet, inverse, src, dst, batch_stride, lo(1), stride_x, size_x, size_y, &js[0]);
	if (!rc && hipStreamSynchronize(g.stream) != hipSuccess)
		rc = fail("hipStreamSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
	std::string first_err = rc ? dwt_hip_last_error() : "";
	for (int k = 1; k < G; k++) {
		std::string err;
		const int r = g_slots[k]->wait(err);
		if (r && !rc) {
			rc = r;
			first_err = "slot " + std::to_string(k) + " (device " + std::to_string(devices[k]) + "): " + err;
		}
	}
	if (rc)
		return fail("%s", first_err.c_str());
	*j = js[0];
	return 0;
}

} // extern "C"
#pragma GCC visibility pop
