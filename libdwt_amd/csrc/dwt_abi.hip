// dwt_abi.hip -- the device-level C-ABI of include/libdwt_hip.h: lifecycle, options, memory helpers,
// kernel timing and the 2-D transform entries (argument checks, host / device dispatch).
#include "dwt_backend.h"

using namespace dwtb;

#pragma GCC visibility push(default)
extern "C" {

const char *dwt_hip_last_error(void) { return g_err; }

int dwt_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int dwt_hip_init(void)
{
	if (g.inited)
		return 0;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
		return fail("no HIP device available (%s); libdwt_amd has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "0 devices");
	int dev = 0;
	const char *env = getenv("DWT_HIP_DEVICE");
	if (!env)
		env = getenv("LOCAL_RANK");
	if (g.want_device >= 0) {
		if (g.want_device >= n)
			return fail("dwt_hip_set_device(%d): the process sees %d device(s)", g.want_device, n);
		dev = g.want_device;
	} else if (env) {
		dev = atoi(env) % n;
	}
	HIP_TRY(hipSetDevice(dev));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, dev));
	snprintf(g.devname, sizeof(g.devname), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
	if (!strstr(prop.gcnArchName, "gfx950"))
		return fail("device %d is %s; this library carries gfx950 code only", dev, prop.gcnArchName);
	g.device = dev;
	g.inited = true;
	return 0;
}

int dwt_hip_set_device(int device)
{
	if (device < 0)
		return fail("dwt_hip_set_device(%d): bad device index", device);
	if (g.inited && g.device != device) {
		// rebinding: this thread's workspace lives on the old device
		dwt_hip_finish();
		g.inited = false;
	}
	g.want_device = device;
	return check_inited();
}

int dwt_hip_get_device(void)
{
	return g.inited ? g.device : -1;
}

void dwt_hip_finish(void)
{
	if (!g.inited)
		return;
	hipStreamSynchronize(g.stream);
	if (g.ll_external) {
		g.ll[0] = g.ll[1] = nullptr;
		g.ll_external = false;
	}
	void **bufs[] = {&g.stage_img, &g.ll[0], &g.ll[1], &g.host_a, &g.host_b, &g.vol_out, &g.vol_host[0], &g.vol_host[1]};
	for (void **b : bufs) {
		if (*b)
			dev_free(*b);
		*b = nullptr;
	}
	g.stage_bytes = g.ll_bytes[0] = g.ll_bytes[1] = g.host_a_bytes = g.host_b_bytes = g.vol_out_bytes = 0;
	g.vol_host_bytes[0] = g.vol_host_bytes[1] = 0;
	for (hipEvent_t &e : g.dl_ev) {
		if (e)
			hipEventDestroy(e);
		e = nullptr;
	}
	for (auto &row : g.pipe_ev)
		for (hipEvent_t &e : row) {
			if (e)
				hipEventDestroy(e);
			e = nullptr;
		}
	if (g.up)
		hipStreamDestroy(g.up);
	if (g.down)
		hipStreamDestroy(g.down);
	g.up = g.down = nullptr;
	if (g.pin)
		hipHostFree(g.pin);
	g.pin = nullptr;
	g.pin_bytes = 0;
	if (g.switch_ev)
		hipEventDestroy(g.switch_ev);
	g.switch_ev = nullptr;
	for (auto &ev : g.prof_events) {
		hipEventDestroy(ev.first);
		hipEventDestroy(ev.second);
	}
	g.prof_events.clear();
	g.prof_used = 0;
	g.tile_cache.clear(); // what dwt_hip_tune measured went with the buffers it was measured on
	g.place_n = 0;
	g.place_best = -1;
	// the context stays usable: a later call re-allocates its workspace
}

const char *dwt_hip_device_name(void)
{
	if (check_inited())
		return "";
	return g.devname;
}

// The workspace of the context (LL scratch, staging image, pinned buffer) is shared by whatever stream comes next, so a
// stream change ORDERS the new stream behind everything this context has queued on the old one: an event recorded on the
// old stream at the switch, awaited by the new one.  No cost for calls that stay on one stream; a caller alternating two
// streams on one thread gets the two chains serialised where they share scratch instead of a race on it.  Streams under
// capture are left alone (the wait would pull the new stream into the capture).
void dwt_hip_set_stream(void *s)
{
	hipStream_t ns = (hipStream_t)s;
	if (g.inited && ns != g.stream) {
		hipStreamCaptureStatus a = hipStreamCaptureStatusNone, b = hipStreamCaptureStatusNone;
		const bool plain = hipStreamIsCapturing(g.stream, &a) == hipSuccess && a == hipStreamCaptureStatusNone &&
			hipStreamIsCapturing(ns, &b) == hipSuccess && b == hipStreamCaptureStatusNone;
		if (plain && (g.switch_ev || hipEventCreateWithFlags(&g.switch_ev, hipEventDisableTiming) == hipSuccess) &&
			hipEventRecord(g.switch_ev, g.stream) == hipSuccess)
			(void)hipStreamWaitEvent(ns, g.switch_ev, 0);
		(void)hipGetLastError();
	}
	g.stream = ns;
}

int dwt_hip_set_workspace(void *band0, size_t bytes0, void *band1, size_t bytes1)
{
	if (check_inited())
		return 1;
	HIP_TRY(hipStreamSynchronize(g.stream));
	if (!g.ll_external) {
		for (int k = 0; k < 2; k++) {
			if (g.ll[k])
				dev_free(g.ll[k]); // (may be a range of a placement arena: dwt_hip_alloc_batch)
			g.ll[k] = nullptr;
			g.ll_bytes[k] = 0;
		}
	}
	if (!band0 || !band1) {
		g.ll[0] = g.ll[1] = nullptr;
		g.ll_bytes[0] = g.ll_bytes[1] = 0;
		g.ll_external = false;
		return 0;
	}
	if (!dwt_hip_is_device_pointer(band0) || !dwt_hip_is_device_pointer(band1) || ((uintptr_t)band0 & 15) || ((uintptr_t)band1 & 15))
		return fail("dwt_hip_set_workspace takes two 16-byte aligned device buffers");
	g.ll[0] = band0;
	g.ll[1] = band1;
	g.ll_bytes[0] = bytes0;
	g.ll_bytes[1] = bytes1;
	g.ll_external = true;
	return 0;
}

int dwt_hip_placement_report(double *ms, int n)
{
	for (int i = 0; i < n && i < g.place_n; i++)
		ms[i] = g.place_ms[i];
	return g.place_n;
}


void dwt_hip_sync(void)
{
	if (g.inited)
		hipStreamSynchronize(g.stream);
}

// One table for dwt_hip_set_option / dwt_hip_get_option: the option's name, where it lives in the calling thread's
// context, whether measured tile heights depend on it (they are forgotten when it changes), how a value is normalised.
namespace {
enum OptKind { kPlain, kSweep /* tile heights were measured under it */, kBool, kNonNegative };
struct Opt {
	const char *name;
	int *(*ref)(Ctx &);
	OptKind kind;
};
#define DWT_OPT(name_, member_, kind_) {name_, [](Ctx &c) -> int * { return &c.member_; }, kind_}
const Opt kOpts[] = {
	DWT_OPT("generic", force_generic, kSweep), DWT_OPT("cpt", tune.cpt, kSweep), DWT_OPT("tile_pairs", tune.tile_pairs, kSweep),
	DWT_OPT("waves", tune.waves, kSweep), DWT_OPT("xcd_swizzle", tune.xcd_swizzle, kSweep), DWT_OPT("ring", tune.ring, kSweep),
	DWT_OPT("ring_inv", tune.ring_inv, kSweep), DWT_OPT("inv_ll_temporal", tune.inv_ll_temporal, kSweep), DWT_OPT("inv_pairs", tune.inv_pairs, kSweep), DWT_OPT("probe_fuse1", tune.probe_fuse1, kSweep), DWT_OPT("nt", tune.nt, kSweep), DWT_OPT("nt_auto", tune.nt_auto, kSweep),
	DWT_OPT("fma", fma, kSweep), DWT_OPT("fused_d", fused_d, kPlain), DWT_OPT("ride_copy", ride_copy, kBool), DWT_OPT("ride_mib", ride_mib, kNonNegative), DWT_OPT("il_exact_borders", il_exact_borders, kPlain),
	DWT_OPT("il_inplace_shell", il_inplace_shell, kPlain), DWT_OPT("host_pipeline", host_pipeline, kPlain),
	DWT_OPT("tune_tiles", tune_tiles, kPlain), DWT_OPT("tune_in_call", tune_in_call, kBool), DWT_OPT("place_tries", place_tries, kPlain),
	DWT_OPT("place_min_mib", place_min_mib, kNonNegative), DWT_OPT("place_max_gib", place_max_gib, kNonNegative),
	DWT_OPT("vol_ip_waves", vol.ip_waves, kPlain), DWT_OPT("vol_tile_pairs", vol.tile_pairs, kPlain), DWT_OPT("vol_nt", vol.nt, kPlain),
	DWT_OPT("vol_fused", vol.fused, kPlain), DWT_OPT("vol_direct", vol.direct, kPlain), DWT_OPT("vol_whole", vol.whole, kPlain),
	DWT_OPT("vol_inplace_fused", vol.inplace_fused, kBool), DWT_OPT("vol_swizzle", vol.swizzle, kPlain), DWT_OPT("vol_rows", vol.rows, kPlain),
};
#undef DWT_OPT
const Opt *find_opt(const char *name)
{
	for (const Opt &o : kOpts)
		if (!strcmp(name, o.name))
			return &o;
	return nullptr;
}
} // namespace

int dwt_hip_set_option(const char *name, int value)
{
	const Opt *o = name ? find_opt(name) : nullptr;
	if (!o)
		return fail("unknown option '%s'", name ? name : "(null)");
	if (o->kind == kSweep)
		g.tile_cache.clear();
	*o->ref(g) = o->kind == kBool ? (value ? 1 : 0) : (o->kind == kNonNegative && value < 0) ? 0 : value;
	return 0;
}

int dwt_hip_get_option(const char *name)
{
	if (!name)
		return -1;
	// read-only figures of the calling thread's context
	if (!strcmp(name, "tune_in_call"))
		return may_measure() && !g.tuning ? 1 : 0; // (DWT_HIP_TUNE is read on first use)
	if (!strcmp(name, "place_last_tries")) // candidates the last placement search timed (0: none ran)
		return g.place_n;
	if (!strcmp(name, "place_last_best"))
		return g.place_best;
	if (!strcmp(name, "stat_launches")) // kernel launches of this context's 2-D drivers so far (tests)
		return (int)(g.stat_launches & 0x7fffffff);
	if (!strcmp(name, "stat_allocs"))   // device allocations of this context's 2-D drivers so far (tests)
		return (int)(g.stat_allocs & 0x7fffffff);
	if (!strcmp(name, "tile_cache_size"))
		return (int)g.tile_cache.size();
	const Opt *o = find_opt(name);
	return o ? *o->ref(g) : -1;
}

int dwt_hip_is_device_pointer(const void *p)
{
	hipPointerAttribute_t at;
	hipError_t e = hipPointerGetAttributes(&at, p);
	if (e != hipSuccess) {
		(void)hipGetLastError(); // plain host memory reports an error; clear it
		return 0;
	}
	return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

void *dwt_hip_malloc(size_t bytes)
{
	if (check_inited())
		return nullptr;
	void *p = nullptr;
	if (hipMalloc(&p, bytes) != hipSuccess) {
		fail("hipMalloc(%zu) failed", bytes);
		return nullptr;
	}
	return p;
}

void dwt_hip_free(void *p)
{
	dev_free(p);
}

void *dwt_hip_malloc_host(size_t bytes)
{
	if (check_inited())
		return nullptr;
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
		fail("hipHostMalloc(%zu) failed", bytes);
		return nullptr;
	}
	return p;
}

void dwt_hip_free_host(void *p)
{
	if (p)
		hipHostFree(p);
}

int dwt_hip_memcpy_h2d(void *d, const void *h, size_t n)
{
	if (check_inited())
		return 1;
	HIP_TRY(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	return 0;
}

int dwt_hip_memcpy_d2h(void *h, const void *d, size_t n)
{
	if (check_inited())
		return 1;
	HIP_TRY(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	return 0;
}

void dwt_hip_prof_enable(int on)
{
	if (g.inited)
		prof_drain();
	g.prof_on = on;
	g.prof_ms = 0;
	g.prof_launches = 0;
}

int dwt_hip_prof_read_levels(double *ms_sum, int *launches, int n)
{
	if (prof_drain())
		return 1;
	for (int i = 0; i < n && i < 16; i++) {
		ms_sum[i] = g.prof_level_ms[i];
		launches[i] = g.prof_level_n[i];
		g.prof_level_ms[i] = 0;
		g.prof_level_n[i] = 0;
	}
	g.prof_ms = 0;
	g.prof_launches = 0;
	return 0;
}

int dwt_hip_prof_read(double *ms, int *launches)
{
	if (prof_drain())
		return 1;
	if (ms)
		*ms = g.prof_ms;
	if (launches)
		*launches = g.prof_launches;
	g.prof_ms = 0;
	g.prof_launches = 0;
	for (int i = 0; i < 16; i++) {
		g.prof_level_ms[i] = 0;
		g.prof_level_n[i] = 0;
	}
	return 0;
}
int dwt_hip_transform2d(int wavelet, int inverse, const void *src, void *dst, int stride_x, int stride_y,
	int sox, int soy, int six, int siy, int *j, int decompose_one, int zero_padding)
{
	if (check_inited())
		return 1;
	if (wavelet < 0 || wavelet > 5)
		return fail("unknown wavelet %d", wavelet);
	if (!src || !dst || !j)
		return fail("null pointer argument");
	const int es = elem_size((Wavelet)wavelet);
	g_elems_are_32bit = (es == 4);
	if (sox <= 0 || soy <= 0 || six < 0 || siy < 0 || six > sox || siy > soy)
		return fail("bad sizes: outer %dx%d inner %dx%d", sox, soy, six, siy);
	const Wavelet w = (Wavelet)wavelet;
	const Geom ge{sox, soy, six, siy};
	const bool dev_src = dwt_hip_is_device_pointer(src), dev_dst = dwt_hip_is_device_pointer(dst);
	if (dev_src != dev_dst)
		return fail("src and dst must both be host or both be device pointers");

	if (dev_dst && stride_y == es && stride_x % es == 0 && stride_x >= sox * es && (uintptr_t)src % es == 0 && (uintptr_t)dst % es == 0) {
		Img s{(char *)src, stride_x, es}, d{(char *)dst, stride_x, es};
		if (!inverse && !decompose_one && (*j < 0 || *j >= 2) && place_ll_scratch(w, s, d, ge, *j, 1, 0, 0))
			return 1;
		return inverse ? inverse2d(w, s, d, ge, *j, decompose_one, zero_padding, 1, 0, 0)
		               : forward2d(w, s, d, ge, j, decompose_one, zero_padding, 1, 0, 0);
	}
	// ---- a device image whose elements are not adjacent (one channel of an interleaved multi-channel image,
	// src/cvdwt.cpp:98-135) or not aligned: the reference gathers every line through dwt_util_memcpy_stride_*
	// (src/system.c:102-164); here the frame is packed into a dense image ON THE DEVICE, transformed there and spread
	// back element by element (dwt_strided.hip) -- nothing crosses PCIe, nothing but the image's own elements is written
	if (dev_dst && (stride_y < es || (long)stride_x < (long)(sox - 1) * stride_y + es))
		return fail("device image: stride_y %d must be >= %d and stride_x %d >= (width-1)*stride_y + %d", stride_y, es, stride_x, es);

	// ---- host pointers: stage the whole outer frame through HBM ----
	if (!dev_dst && ge.dense() && stride_y == es && es == 4) {
		const int rc = inverse ? host_inverse_pipelined(w, src, dst, stride_x, sox, soy, *j, decompose_one)
		                       : host_forward_pipelined(w, src, dst, stride_x, sox, soy, j, decompose_one);
		if (rc >= 0)
			return rc;
	}
	const long pitch = align_up((long)sox * es, 256);
	const size_t bytes = (size_t)pitch * soy;
	if (grow(&g.host_a, &g.host_a_bytes, bytes) || grow(&g.host_b, &g.host_b_bytes, bytes))
		return 1;
	const bool s2 = (src != dst);
	auto upload = [&](const void *hp, void *dp) -> int {
		if (!dev_dst)
			return host_upload(hp, stride_x, stride_y, es, sox, soy, dp, pitch);
		const hipError_t e = launch_strided_pack(dp, pitch, hp, stride_x, stride_y, es, sox, soy, g.stream);
		return e == hipSuccess ? 0 : fail("strided pack launch failed: %s", hipGetErrorString(e));
	};
	auto download = [&](void *hp, const void *dp) -> int {
		if (!dev_dst)
			return host_download(hp, stride_x, stride_y, es, sox, soy, dp, pitch);
		const hipError_t e = launch_strided_unpack(hp, stride_x, stride_y, dp, pitch, es, sox, soy, g.stream);
		return e == hipSuccess ? 0 : fail("strided unpack launch failed: %s", hipGetErrorString(e));
	};
	Img A{(char *)g.host_a, pitch, es}, B{(char *)g.host_b, pitch, es};
	if (upload(src, A.p))
		return 1;
	// B receives the result.  It starts as a copy of what the destination holds so
	// that every element the reference leaves untouched keeps its value -- unless the call
	// writes every element of the frame anyway (a dense frame, at least one level: no second
	// trip over PCIe for the out-of-place entries)
	const int so_min = sox < soy ? sox : soy, so_max = sox > soy ? sox : soy;
	const int j_lim = ceil_log2(decompose_one ? so_max : so_min);
	const int j_eff = (*j < 0 || *j > j_lim) ? j_lim : *j;
	const bool writes_all = ge.dense() && j_eff >= 1;
	if (s2 && writes_all) {
		// (nothing to preserve)
	} else if (s2) {
		if (upload(dst, B.p))
			return 1;
	} else {
		if (copy_rect(B, 0, 0, A, 0, 0, sox, soy))
			return 1;
	}
	int rc;
	if (s2 || ge.dense()) {
		// out of place on the device: no in-place detour even for the in-place entry
		rc = inverse ? inverse2d(w, A, B, ge, *j, decompose_one, zero_padding, 1, 0, 0)
		             : forward2d(w, A, B, ge, j, decompose_one, zero_padding, 1, 0, 0);
	} else {
		rc = inverse ? inverse2d(w, B, B, ge, *j, decompose_one, zero_padding, 1, 0, 0)
		             : forward2d(w, B, B, ge, j, decompose_one, zero_padding, 1, 0, 0);
	}
	if (rc)
		return rc;
	return download(dst, B.p);
}

int dwt_hip_transform2d_batch(int wavelet, int inverse, const void *src, void *dst, size_t batch_stride, int batch,
	int stride_x, int size_x, int size_y, int *j)
{
	if (check_inited())
		return 1;
	if (wavelet < 0 || wavelet > 5)
		return fail("unknown wavelet %d", wavelet);
	const int es = elem_size((Wavelet)wavelet);
	g_elems_are_32bit = es == 4;
	if (!src || !dst || !j || batch < 1 || batch > 65535)
		return fail("bad argument (batch must be 1..65535)");
	if (!dwt_hip_is_device_pointer(src) || !dwt_hip_is_device_pointer(dst))
		return fail("batched transforms take device pointers");
	if ((stride_x % es) || stride_x < size_x * es || (batch_stride % es) || batch_stride < (size_t)stride_x * size_y)
		return fail("bad strides");
	if (batch > 1 && src == dst)
		return fail("in-place batches are not supported; use distinct src and dst");
	const Geom ge{size_x, size_y, size_x, size_y};
	Img s{(char *)src, stride_x, es}, d{(char *)dst, stride_x, es};
	if (!inverse && (*j < 0 || *j >= 2) && place_ll_scratch((Wavelet)wavelet, s, d, ge, *j, batch, (long)batch_stride, (long)batch_stride))
		return 1;
	return inverse ? inverse2d((Wavelet)wavelet, s, d, ge, *j, 0, 0, batch, (long)batch_stride, (long)batch_stride)
	               : forward2d((Wavelet)wavelet, s, d, ge, j, 0, 0, batch, (long)batch_stride, (long)batch_stride);
}

int dwt_hip_conv_show(int is_int, const void *src, void *dst, int stride_x, int stride_y, int size_x, int size_y)
{
	if (check_inited())
		return 1;
	if (!dwt_hip_is_device_pointer(src) || !dwt_hip_is_device_pointer(dst))
		return fail("dwt_hip_conv_show takes device images (host images: dwt_util_conv_show_s/_i)");
	if (stride_y != 4 || (stride_x & 3))
		return fail("device images need stride_y == 4 and stride_x a multiple of 4");
	hipError_t e = launch_conv_show(is_int != 0, src, dst, stride_x, size_x, size_y, g.stream);
	if (e != hipSuccess)
		return fail("conv_show launch failed: %s", hipGetErrorString(e));
	return 0;
}

int dwt_hip_compare(int is_int, const void *ptr1, const void *ptr2, int stride_x, int stride_y, int size_x, int size_y)
{
	if (check_inited())
		return -1;
	if (!dwt_hip_is_device_pointer(ptr1) || !dwt_hip_is_device_pointer(ptr2)) {
		fail("dwt_hip_compare takes device images (host images: dwt_util_compare_s/_i)");
		return -1;
	}
	if (stride_y != 4 || (stride_x & 3)) {
		fail("device images need stride_y == 4 and stride_x a multiple of 4");
		return -1;
	}
	static thread_local unsigned *counter = nullptr; // per thread, like the context (and its device)
	if (!counter && hipMalloc((void **)&counter, sizeof(unsigned)) != hipSuccess) {
		fail("hipMalloc failed");
		return -1;
	}
	unsigned host = 0;
	if (hipMemsetAsync(counter, 0, sizeof(unsigned), g.stream) != hipSuccess ||
		launch_compare(is_int != 0, ptr1, ptr2, stride_x, size_x, size_y, counter, g.stream) != hipSuccess ||
		hipMemcpyAsync(&host, counter, sizeof(unsigned), hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
		hipStreamSynchronize(g.stream) != hipSuccess) {
		fail("compare failed: %s", hipGetErrorString(hipGetLastError()));
		return -1;
	}
	return host ? 1 : 0;
}

} // extern "C"
#pragma GCC visibility pop
