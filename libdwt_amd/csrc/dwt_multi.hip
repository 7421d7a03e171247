// dwt_multi.hip -- one process, several GPUs: the batch split behind the C-ABI ("single
// process, 8 devices, one host thread per device").
//
// Images are independent, so a batch of B images shards over G devices with NO collective in the
// transform: image b belongs to slot b*G/B.  The batch lives in the memory of the calling thread's device
// (slot 0, whose shard is transformed where it lies); every other slot has a persistent host thread with a
// context of its own (device binding, stream, workspace, staging images -- all kept between calls) that
// pulls its shard over xGMI (hipMemcpyPeerAsync), runs the single-GPU batched transform on it and pushes
// the coefficients back.  The only exchange is that split; all slots work at the same time.
//
// (The Python harness does the same split across PROCESSES with RCCL point-to-point transfers,
// libdwt_amd/batch.py; this file is what a C caller of libdwt.h gets.)
#include "dwt_backend.h"

#include <memory>
#include <string>

namespace dwtb {

// A host thread that lives as long as the process and runs one job at a time: its thread-local
// context (dwt_backend.hip's `g`) and its staging buffers persist between the calls.
class SlotWorker {
public:
	SlotWorker() : th_([this] { loop(); }) { th_.detach(); }
	void submit(std::function<int()> job)
	{
		std::lock_guard<std::mutex> lk(m_);
		job_ = std::move(job);
		busy_ = true;
		rc_ = 0;
		cv_.notify_all();
	}
	int wait(std::string &err)
	{
		std::unique_lock<std::mutex> lk(m_);
		cv_.wait(lk, [&] { return !busy_; });
		err = err_;
		return rc_;
	}
	// staging of this slot (owned by the worker thread's device)
	void *stage[2] = {nullptr, nullptr};
	size_t stage_bytes[2] = {0, 0};
	int device = -1;

private:
	void loop()
	{
		for (;;) {
			std::function<int()> job;
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_.wait(lk, [&] { return busy_ && job_; });
				job = std::move(job_);
				job_ = nullptr;
			}
			const int rc = job();
			{
				std::lock_guard<std::mutex> lk(m_);
				rc_ = rc;
				err_ = rc ? dwt_hip_last_error() : "";
				busy_ = false;
			}
			cv_.notify_all();
		}
	}
	std::mutex m_;
	std::condition_variable cv_;
	std::function<int()> job_;
	bool busy_ = false;
	int rc_ = 0;
	std::string err_;
	std::thread th_;
};

static std::mutex g_slots_mu;             // one sharded call at a time (the workers are shared)
static std::vector<SlotWorker *> g_slots; // slot k >= 1 -> its worker (never destroyed)

static int slot_job(SlotWorker *wk, int device, int root_device, int wavelet, int inverse, const char *src, char *dst,
	size_t batch_stride, int n, int stride_x, int size_x, int size_y, int j_in, int *j_out, bool dense)
{
	if (dwt_hip_set_device(device))
		return 1;
	if (wk->device != device) { // the slot moved to another device: its staging is on the old one
		for (int k = 0; k < 2; k++) {
			if (wk->stage[k])
				dev_free(wk->stage[k]);
			wk->stage[k] = nullptr;
			wk->stage_bytes[k] = 0;
		}
		wk->device = device;
		if (device != root_device) {
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, device, root_device) == hipSuccess && can)
				(void)hipDeviceEnablePeerAccess(root_device, 0);
			(void)hipGetLastError(); // already enabled / not possible: the copies are staged by the runtime then
		}
	}
	const size_t bytes = (size_t)n * batch_stride;
	for (int k = 0; k < 2; k++)
		if (grow(&wk->stage[k], &wk->stage_bytes[k], bytes))
			return 1;
	hipStream_t st = g.stream;
	HIP_TRY(hipMemcpyPeerAsync(wk->stage[0], device, src, root_device, bytes, st));
	if (!dense) // bytes between the frames of the destination keep their values: bring them along
		HIP_TRY(hipMemcpyPeerAsync(wk->stage[1], device, dst, root_device, bytes, st));
	int j = j_in;
	if (dwt_hip_transform2d_batch(wavelet, inverse, wk->stage[0], wk->stage[1], batch_stride, n, stride_x, size_x, size_y, &j))
		return 1;
	HIP_TRY(hipMemcpyPeerAsync(dst, root_device, wk->stage[1], device, bytes, st));
	HIP_TRY(hipStreamSynchronize(st));
	*j_out = j;
	return 0;
}

} // namespace dwtb

using namespace dwtb;

#pragma GCC visibility push(default)
extern "C" {

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
	while ((int)g_slots.size() < G)
		g_slots.push_back(g_slots.empty() ? nullptr : new SlotWorker()); // index 0 is the caller itself
	// everything queued on the caller's stream so far (the batch's producers) before the other devices read it
	HIP_TRY(hipStreamSynchronize(g.stream));
	std::vector<int> js(G, *j);
	const int j_in = *j;
	auto lo = [&](int k) { return (int)((long)k * batch / G); };
	for (int k = 1; k < G; k++) {
		SlotWorker *wk = g_slots[k];
		const int a = lo(k), n = lo(k + 1) - a, dev = devices[k];
		const char *s = (const char *)src + (size_t)a * batch_stride;
		char *d = (char *)dst + (size_t)a * batch_stride;
		int *jo = &js[k];
		wk->submit([=] { return slot_job(wk, dev, root, wavelet, inverse, s, d, batch_stride, n, stride_x, size_x, size_y, j_in, jo, dense); });
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
