// dwt_backend.hip -- device context, workspace, host<->HBM staging and the multi-level
// drivers behind the C-ABI of include/libdwt_hip.h.
//
// Level scheduling (forward, dense frame).  libdwt transforms in place
// (src/libdwt.c:12812-12919): level j reads the LL region of the image and writes
// its four subbands over it.  A fused tile sweep cannot do that (one tile's outputs
// land on another tile's inputs), so the driver keeps the running LL band in a
// small ping-pong scratch instead of the image:
//
//     level 0 : image            -> HL/LH/HH at their final place, LL -> scratch0
//     level j : scratch[(j-1)&1] -> HL/LH/HH at their final place, LL -> scratch[j&1]
//     last    :                     LL -> its final place too
//
// Every level therefore reads its input once and writes its output once: the
// algorithmic traffic 2*sizeof(T)*sum_j(W_j*H_j).  Only when the caller's source
// and destination are the SAME device buffer does level 0 have to detour its detail
// subbands through a staging image and copy them back (the `_s2` entries and every
// host-pointer call avoid that).  The inverse runs the mirror image of this.
//
// Frames with size_o != size_i, zero padding, and levels where a direction has a
// single line follow the reference's exact line-by-line semantics through the
// generic line-pass kernel, out of place per pass.
#include "dwt_backend.h"

namespace dwtb {

thread_local Ctx g;
thread_local char g_err[512] = "";

int fail(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return 1;
}

int grow(void **p, size_t *have, size_t need)
{
	if (*have >= need)
		return 0;
	if (*p) {
		HIP_TRY(hipStreamSynchronize(g.stream));
		dev_free(*p);
		*p = nullptr;
		*have = 0;
	}
	HIP_TRY(hipMalloc(p, need));
	*have = need;
	return 0;
}


bool skip_single(Wavelet w) { return w == kCdf97S; } // only the 9/7 drivers guard on lines > 1

// several rectangles (element coordinates) between two device images in one kernel launch
struct Rect {
	long dx, dy, sx, sy, w, h;
};
int copy_rects_on(hipStream_t st, Img dst, Img src, const Rect *rc, int n, int policy = 3)
{
	CopyRects r{};
	r.n = 0;
	r.policy = policy;
	for (int k = 0; k < n && r.n < 3; k++) {
		if (rc[k].w <= 0 || rc[k].h <= 0)
			continue;
		const int i = r.n++;
		r.src[i] = src.p + rc[k].sy * src.sx + rc[k].sx * src.es;
		r.dst[i] = dst.p + rc[k].dy * dst.sx + rc[k].dx * dst.es;
		r.spitch[i] = src.sx;
		r.dpitch[i] = dst.sx;
		r.wbytes[i] = (int)(rc[k].w * dst.es);
		r.h[i] = (int)rc[k].h;
	}
	hipError_t e = launch_copy_rects(r, st);
	if (e != hipSuccess)
		return fail("rectangle copy launch failed: %s", hipGetErrorString(e));
	return 0;
}

int copy_rect_on(hipStream_t st, Img dst, long dx, long dy, Img src, long sx_, long sy_, long w, long h)
{
	if (w <= 0 || h <= 0)
		return 0;
	HIP_TRY(hipMemcpy2DAsync(dst.p + dy * dst.sx + dx * dst.es, dst.sx, src.p + sy_ * src.sx + sx_ * src.es, src.sx, w * dst.es, h,
		hipMemcpyDeviceToDevice, st));
	return 0;
}

// ---- host images <-> dense device images (host-pointer entries) ----
// hipMemcpy2D from pageable memory falls to a row-by-row path when the host pitch is not
// nicely aligned -- and libdwt's "optimal" strides are primes (2053 B for 512 floats,
// src/libdwt.c:20655-20658): 7.7 ms instead of 0.16 ms for 512^2.  Such images are packed
// into a pinned buffer with the device pitch (parallel row memcpy) and moved by ONE copy.
static int grow_pinned(size_t need)
{
	if (g.pin_bytes >= need)
		return 0;
	if (g.pin) {
		HIP_TRY(hipStreamSynchronize(g.stream));
		HIP_TRY(hipHostFree(g.pin));
		g.pin = nullptr;
		g.pin_bytes = 0;
	}
	HIP_TRY(hipHostMalloc(&g.pin, need, hipHostMallocDefault));
	g.pin_bytes = need;
	return 0;
}

// A small persistent pool for the host-side row repacking (starting 16 threads per call cost more
// than the repacking of a 1080p frame).  Workers sleep on a condition variable between jobs; a job
// is a range of row chunks handed out under the mutex; the caller works too.  One job at a time
// (calls from several host threads take turns).  The pool is created on first use and never
// destroyed (no static-destruction order to get wrong); a forked child builds its own.
class RowPool {
public:
	static RowPool &get()
	{
		static RowPool *p = nullptr;
		static std::mutex mk;
		std::lock_guard<std::mutex> lk(mk);
		if (!p || p->pid_ != getpid())
			p = new RowPool();
		return *p;
	}
	template <class F>
	void run(int rows, int chunk, F f)
	{
		std::lock_guard<std::mutex> turn(turn_);
		std::function<void(int, int)> fn = f;
		{
			std::lock_guard<std::mutex> lk(m_);
			job_ = &fn; rows_ = rows; chunk_ = chunk; next_ = 0; active_ = 0; gen_++;
		}
		cv_job_.notify_all();
		work();
		std::unique_lock<std::mutex> lk(m_);
		cv_done_.wait(lk, [&] { return next_ >= rows_ && active_ == 0; });
		job_ = nullptr;
	}
	int workers() const { return (int)th_.size() + 1; }

private:
	RowPool() : pid_(getpid())
	{
		unsigned n = std::thread::hardware_concurrency();
		n = n > 16 ? 16 : n;
		for (unsigned i = 1; i < n; i++)
			th_.emplace_back([this] { loop(); });
		for (auto &t : th_)
			t.detach();
	}
	void loop()
	{
		unsigned long seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_job_.wait(lk, [&] { return gen_ != seen; });
				seen = gen_;
			}
			work();
		}
	}
	void work()
	{
		for (;;) {
			int a, b;
			const std::function<void(int, int)> *fn;
			{
				std::lock_guard<std::mutex> lk(m_);
				if (!job_ || next_ >= rows_)
					break;
				a = next_; b = a + chunk_ < rows_ ? a + chunk_ : rows_;
				next_ = b; active_++; fn = job_;
			}
			(*fn)(a, b);
			{
				std::lock_guard<std::mutex> lk(m_);
				active_--;
			}
			cv_done_.notify_all();
		}
		cv_done_.notify_all();
	}
	pid_t pid_;
	std::vector<std::thread> th_;
	std::mutex m_, turn_;
	std::condition_variable cv_job_, cv_done_;
	const std::function<void(int, int)> *job_ = nullptr;
	int rows_ = 0, chunk_ = 1, next_ = 0, active_ = 0;
	unsigned long gen_ = 0;
};

template <class F>
static void for_rows_parallel(int rows, size_t bytes_total, F f)
{
	if (bytes_total < (2u << 20) || rows < 64) {
		f(0, rows);
		return;
	}
	RowPool &pool = RowPool::get();
	// about four chunks per worker, so that a slow core does not hold the others up
	const int chunk = std::max(8, rows / (4 * pool.workers()));
	pool.run(rows, chunk, f);
}

// element-strided rows (one channel of an interleaved multi-channel image, src/cvdwt.cpp:98-135):
// fixed-size copies the compiler turns into plain loads and stores (a memcpy with a run-time size
// is a library call per element: 5.6 ms for one 1920 x 1080 channel)
template <int ES>
static void gather_row(char *dense, const char *strided, int w, int stride)
{
	for (int x = 0; x < w; x++)
		memcpy(dense + (size_t)x * ES, strided + (size_t)x * stride, ES);
}

template <int ES>
static void scatter_row(char *strided, const char *dense, int w, int stride)
{
	for (int x = 0; x < w; x++)
		memcpy(strided + (size_t)x * stride, dense + (size_t)x * ES, ES);
}

// a 2-D copy straight from / to the caller's rows runs at the PCIe rate for every pitch that is a multiple of 4 bytes
// (57 GB/s at 8192, 8196 and 8256 B, pageable or pinned) and at 1 GB/s for an odd one (8205 B):
// scripts/probes/r04_oddpitch_probe.py
static bool host_pitch_is_fast(const void *hp, int stride_x, int stride_y, int es)
{
	return stride_y == es && stride_x % 4 == 0 && (uintptr_t)hp % 4 == 0;
}

// w x h elements of `es` bytes at hp (byte strides) -> device image dp with `pitch`
int host_upload(const void *hp, int stride_x, int stride_y, int es, int w, int h, void *dp, long pitch)
{
	if (host_pitch_is_fast(hp, stride_x, stride_y, es)) {
		HIP_TRY(hipMemcpy2DAsync(dp, pitch, hp, stride_x, (size_t)w * es, h, hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
		return 0;
	}
	if (grow_pinned((size_t)pitch * h))
		return 1;
	char *pin = (char *)g.pin;
	// strips: the CPU repacks strip k+1 into the pinned buffer while strip k crosses PCIe
	const int strips = (int)std::min<size_t>(8, std::max<size_t>(1, (size_t)pitch * h / (16u << 20))); // >= 16 MiB each
	const int rows_per = (h + strips - 1) / strips;
	for (int y_a = 0; y_a < h; y_a += rows_per) {
		const int y_b = y_a + rows_per < h ? y_a + rows_per : h;
		for_rows_parallel(y_b - y_a, (size_t)pitch * (y_b - y_a), [=](int r0, int r1) {
			for (int y = y_a + r0; y < y_a + r1; y++) {
				const char *row = (const char *)hp + (long)y * stride_x;
				char *out = pin + (long)y * pitch;
				if (stride_y == es)
					memcpy(out, row, (size_t)w * es);
				else if (es == 4)
					gather_row<4>(out, row, w, stride_y);
				else
					gather_row<8>(out, row, w, stride_y);
			}
		});
		HIP_TRY(hipMemcpyAsync((char *)dp + (long)y_a * pitch, pin + (long)y_a * pitch, (size_t)pitch * (y_b - y_a), hipMemcpyHostToDevice, g.stream));
	}
	HIP_TRY(hipStreamSynchronize(g.stream));
	return 0;
}

int host_download(void *hp, int stride_x, int stride_y, int es, int w, int h, const void *dp, long pitch)
{
	if (host_pitch_is_fast(hp, stride_x, stride_y, es)) {
		HIP_TRY(hipMemcpy2DAsync(hp, stride_x, dp, pitch, (size_t)w * es, h, hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
		return 0;
	}
	if (grow_pinned((size_t)pitch * h))
		return 1;
	char *pin = (char *)g.pin;
	// strips: strip k is spread back into the caller's image while strip k+1 crosses PCIe
	const int strips = (int)std::min<size_t>(8, std::max<size_t>(1, (size_t)pitch * h / (16u << 20))); // >= 16 MiB each
	const int rows_per = (h + strips - 1) / strips;
	// one event per strip, created once per context (they used to be created and destroyed per call,
	// and leaked when a call failed half way)
	hipEvent_t *ev = g.dl_ev;
	int n_ev = 0;
	for (int y_a = 0; y_a < h; y_a += rows_per, n_ev++) {
		const int y_b = y_a + rows_per < h ? y_a + rows_per : h;
		HIP_TRY(hipMemcpyAsync(pin + (long)y_a * pitch, (const char *)dp + (long)y_a * pitch, (size_t)pitch * (y_b - y_a), hipMemcpyDeviceToHost, g.stream));
		if (!ev[n_ev])
			HIP_TRY(hipEventCreateWithFlags(&ev[n_ev], hipEventDisableTiming));
		HIP_TRY(hipEventRecord(ev[n_ev], g.stream));
	}
	int k = 0;
	for (int y_a = 0; y_a < h; y_a += rows_per, k++) {
		const int y_b = y_a + rows_per < h ? y_a + rows_per : h;
		HIP_TRY(hipEventSynchronize(ev[k]));
		for_rows_parallel(y_b - y_a, (size_t)pitch * (y_b - y_a), [=](int r0, int r1) {
			for (int y = y_a + r0; y < y_a + r1; y++) {
				char *row = (char *)hp + (long)y * stride_x;
				const char *in = pin + (long)y * pitch;
				if (stride_y == es)
					memcpy(row, in, (size_t)w * es);
				else if (es == 4)
					scatter_row<4>(row, in, w, stride_y);
				else
					scatter_row<8>(row, in, w, stride_y);
			}
		});
	}
	return 0;
}

// A host volume with awkward strides (libdwt's own "optimal" strides are odd numbers of bytes: a 2-D copy with such a
// pitch runs at 1 GB/s, scripts/probes/r04_oddpitch_probe.py) <-> a device volume: batches of slices of about 32 MiB
// are repacked by the row pool into / out of the halves of a pinned buffer laid out like the device volume, one copy
// per batch, the CPU on batch k+1 while batch k crosses PCIe.  (Round 3 moved slice by slice with a stream
// synchronisation each: 0.8 ns per voxel against 0.14 for the bytes alone.)
int host_volume_xfer(bool to_device, void *dev, size_t d_sy, size_t d_sz, void *host, size_t h_sy, size_t h_sz, int nx, int ny, int nz)
{
	const int zb = (int)std::max<size_t>(1, std::min<size_t>((size_t)nz, ((size_t)32 << 20) / d_sz));
	const size_t half = (size_t)zb * d_sz;
	if (grow_pinned(2 * half))
		return 1;
	hipEvent_t *ev = g.dl_ev;
	for (int k = 0; k < 2; k++)
		if (!ev[k])
			HIP_TRY(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
	const int nb = (nz + zb - 1) / zb;
	auto rows_of = [&](int b, char *buf, bool pack) {
		const int z0 = b * zb, z1 = std::min(nz, z0 + zb);
		for_rows_parallel((z1 - z0) * ny, (size_t)(z1 - z0) * d_sz, [=](int r0, int r1) {
			for (int r = r0; r < r1; r++) {
				const int z = z0 + r / ny, y = r % ny;
				char *h = (char *)host + (size_t)z * h_sz + (size_t)y * h_sy;
				char *p = buf + (size_t)(z - z0) * d_sz + (size_t)y * d_sy;
				if (pack)
					memcpy(p, h, (size_t)nx * 4);
				else
					memcpy(h, p, (size_t)nx * 4);
			}
		});
	};
	if (to_device) {
		for (int b = 0; b < nb; b++) {
			char *buf = (char *)g.pin + (size_t)(b & 1) * half;
			if (b >= 2)
				HIP_TRY(hipEventSynchronize(ev[b & 1])); // the copy that last read this half
			rows_of(b, buf, true);
			const int z0 = b * zb, z1 = std::min(nz, z0 + zb);
			HIP_TRY(hipMemcpyAsync((char *)dev + (size_t)z0 * d_sz, buf, (size_t)(z1 - z0) * d_sz, hipMemcpyHostToDevice, g.stream));
			HIP_TRY(hipEventRecord(ev[b & 1], g.stream));
		}
		HIP_TRY(hipStreamSynchronize(g.stream));
		return 0;
	}
	auto issue = [&](int b) -> int {
		const int z0 = b * zb, z1 = std::min(nz, z0 + zb);
		HIP_TRY(hipMemcpyAsync((char *)g.pin + (size_t)(b & 1) * half, (const char *)dev + (size_t)z0 * d_sz, (size_t)(z1 - z0) * d_sz, hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipEventRecord(ev[b & 1], g.stream));
		return 0;
	};
	if (issue(0))
		return 1;
	for (int b = 0; b < nb; b++) {
		HIP_TRY(hipEventSynchronize(ev[b & 1]));
		if (b + 1 < nb && issue(b + 1)) // (the other half: unpacked an iteration ago)
			return 1;
		rows_of(b, (char *)g.pin + (size_t)(b & 1) * half, false);
	}
	return 0;
}

int copy_rect(Img dst, long dx, long dy, Img src, long sx_, long sy_, long w, long h)
{
	return copy_rect_on(g.stream, dst, dx, dy, src, sx_, sy_, w, h);
}

int zero_rect(Img img, long x, long y, long w, long h)
{
	if (w <= 0 || h <= 0)
		return 0;
	HIP_TRY(hipMemset2DAsync(img.p + y * img.sx + x * img.es, img.sx, 0, w * img.es, h, g.stream));
	return 0;
}

// One generic 1-D pass over the frame of a level (rows: lines are image rows).
// in == out is handled by staging the frame -- pass-through elements included -- in
// an image with the caller's pitch, so the kernel sees one pair of strides.
int generic_pass(Wavelet w, bool inverse, bool rows, Img in, Img out, int frame_w, int frame_h, int n_lines, int N, int hoff)
{
	if (n_lines <= 0 || N <= 0)
		return 0;
	if (N == 1 && (w == kCdf53I || w == kCdf97I)) {
		// the int kernels leave a lone sample as it is (src/libdwt.c:10961); out of place that
		// still means the samples have to arrive in the destination
		if (in.p != out.p)
			return copy_rect(out, 0, 0, in, 0, 0, frame_w, frame_h);
		return 0;
	}
	const bool alias = in.p == out.p;
	Img dst = out;
	if (alias) {
		// staging image with the SAME pitch as the caller's image
		if (grow(&g.stage_img, &g.stage_bytes, (size_t)out.sx * frame_h))
			return 1;
		dst = Img{(char *)g.stage_img, out.sx, out.es};
		if (copy_rect(dst, 0, 0, out, 0, 0, frame_w, frame_h))
			return 1;
	} else if (in.sx != out.sx) {
		return fail("generic pass: source and destination pitches differ (%ld vs %ld)", in.sx, out.sx);
	}
	hipError_t e = launch_line_pass(w, inverse, in.p, dst.p, rows ? in.sx : in.es, rows ? in.es : in.sx, n_lines, N, hoff, !rows, g.stream);
	if (e != hipSuccess)
		return fail("line pass launch failed: %s", hipGetErrorString(e));
	if (alias && copy_rect(out, 0, 0, dst, 0, 0, frame_w, frame_h))
		return 1;
	return 0;
}

void prof_before(int level)
{
	if (!g.prof_on || (g.prof_on == 1 && level != 0))
		return;
	if (g.prof_tag.size() <= g.prof_used)
		g.prof_tag.resize(g.prof_used + 1);
	g.prof_tag[g.prof_used] = level;
	if (g.prof_used == g.prof_events.size()) {
		hipEvent_t a, b;
		hipEventCreate(&a);
		hipEventCreate(&b);
		g.prof_events.push_back({a, b});
	}
	hipEventRecord(g.prof_events[g.prof_used].first, g.stream);
}

void prof_after(int level)
{
	if (!g.prof_on || (g.prof_on == 1 && level != 0))
		return;
	hipEventRecord(g.prof_events[g.prof_used].second, g.stream);
	g.prof_used++;
}

int prof_drain()
{
	if (g.prof_used == 0)
		return 0;
	HIP_TRY(hipStreamSynchronize(g.stream));
	for (size_t i = 0; i < g.prof_used; i++) {
		float ms = 0;
		HIP_TRY(hipEventElapsedTime(&ms, g.prof_events[i].first, g.prof_events[i].second));
		const int lv = g.prof_tag[i] & 15;
		g.prof_level_ms[lv] += ms;
		g.prof_level_n[lv]++;
		if (lv == 0) {
			g.prof_ms += ms;
			g.prof_launches++;
		}
	}
	g.prof_used = 0;
	return 0;
}

long ll_pitch_elems(int w) { return align_up(w, 4); }
static char *ll_band(int k) { return (char *)g.ll[k]; }

size_t ll_band_bytes(const Geom &ge, int k, int batch, int es)
{
	return (size_t)ll_pitch_elems(ge.Wo(k + 1)) * ge.Ho(k + 1) * es * batch + 64;
}

int ensure_ll(const Geom &ge, int batch, int es)
{
	for (int k = 0; k < 2; k++) {
		if (g.ll_external) {
			if (g.ll_bytes[k] < ll_band_bytes(ge, k, batch, es))
				return fail("the caller's workspace (dwt_hip_set_workspace) is too small: band %d needs %zu bytes", k, ll_band_bytes(ge, k, batch, es));
			continue;
		}
		if (grow(&g.ll[k], &g.ll_bytes[k], ll_band_bytes(ge, k, batch, es)))
			return 1;
	}
	return 0;
}

thread_local bool g_elems_are_32bit = true;

// the fused sweeps exist for the 32-bit types and, since round 2, for the double-precision wavelets
// (dwt_sweep2d_d.hip; option "fused_d" = 0 sends those back to the exact line passes)
bool level_fused_ok(const Geom &ge, int j)
{
	return !g.force_generic && (g_elems_are_32bit || g.fused_d) && ge.Wi(j) == ge.Wo(j) && ge.Hi(j) == ge.Ho(j) && ge.Wo(j) >= 2 &&
		ge.Ho(j) >= 2;
}

// ---- tile height of a large forward level: measured, once per shape ------------------------------------
// The launcher's rule (64 row pairs per tile unless that leaves too few tiles) is within 1-2 % of the best
// height for level 0 of most calls, but the best height of a level depends on more than its tile count --
// level 1 of 32 images wants 32 pairs (735 against 765 us), level 0 of 8 images wants 64 (761 against 778),
// both have 8192 tiles of 64 pairs; level 3 of 64 images wants 16 (113 against 141 us).  So a level that
// moves 64 MiB or more is timed ONCE per (wavelet, width, height, batch) with 64, 32 and 16 pairs -- the level
// is idempotent while its input stands, which it does until the next level runs -- and the fastest height is
// remembered by the calling thread's context.  Same bits with every height (tests: tile variants).  Never
// under a stream capture; option "tune_tiles" = 0 turns it off; a forced "tile_pairs" wins.
static int tune_tile_pairs(unsigned long long key, std::initializer_list<int> heights, const std::function<hipError_t(const SweepTuning &)> &launch)
{
	auto it = g.tile_cache.find(key);
	if (it != g.tile_cache.end())
		return it->second;
	if (g.placing || stream_is_capturing())
		return 0; // decided later, by a call that may synchronise
	hipEvent_t e0, e1;
	if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
		return 0;
	int best = 0;
	float best_ms = 0;
	for (int tp : heights) {
		SweepTuning t = g.tune;
		t.tile_pairs = tp;
		float ms = 0;
		bool ok = true;
		for (int r = 0; r < 2 && ok; r++) {
			hipEventRecord(e0, g.stream);
			ok = launch(t) == hipSuccess;
			hipEventRecord(e1, g.stream);
		}
		ok = ok && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
		if (ok && (!best || ms < best_ms)) {
			best = tp;
			best_ms = ms;
		}
	}
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	(void)hipGetLastError();
	g.tile_cache[key] = best;
	return best;
}

static unsigned long long tile_key(Wavelet w, bool inverse, int W, int H, int batch)
{
	return ((unsigned long long)w << 59) ^ ((unsigned long long)inverse << 58) ^ ((unsigned long long)W << 38) ^ ((unsigned long long)H << 18) ^ (unsigned long long)batch;
}

int tuned_tile_pairs(Wavelet w, const FwdLevelArgs &a)
{
	if (!g.tune_tiles || a.interleaved || (size_t)a.W * a.H * a.batch * sizeof(float) < ((size_t)64 << 20) || a.W < 1024 || a.H < 256)
		return 0;
	const Wavelet wk = (g.fma && w == kCdf97S) ? kCdf97SFma : w;
	return tune_tile_pairs(tile_key(w, false, a.W, a.H, a.batch), {64, 32, 16}, [&](const SweepTuning &t) { return launch_fwd_level(wk, a, t, g.stream); });
}

// the inverse levels alike (32 images of 8192^2: 16 pairs 520 against 507-510 Gsamples/s with the rule's 32)
int tuned_tile_pairs(Wavelet w, const InvLevelArgs &a)
{
	if (!g.tune_tiles || a.interleaved || (size_t)a.W * a.H * a.batch * sizeof(float) < ((size_t)64 << 20) || a.W < 1024 || a.H < 256)
		return 0;
	const Wavelet wk = (g.fma && w == kCdf97S) ? kCdf97SFma : w;
	return tune_tile_pairs(tile_key(w, true, a.W, a.H, a.batch), {32, 16, 8}, [&](const SweepTuning &t) { return launch_inv_level(wk, a, t, g.stream); });
}

// ---- forward ---------------------------------------------------------------------
int forward2d(Wavelet w, Img src, Img dst, const Geom &ge, int *jp, int decompose_one, int zero_padding,
	int batch, long src_bstride, long dst_bstride)
{
	const int so_min = ge.sox < ge.soy ? ge.sox : ge.soy, so_max = ge.sox > ge.soy ? ge.sox : ge.soy;
	const int j_limit = ceil_log2(decompose_one ? so_max : so_min);
	if (*jp < 0 || *jp > j_limit)
		*jp = j_limit; // src/libdwt.c:12807-12810
	const int J = *jp;
	if (J == 0)
		return 0;
	const int es = elem_size(w);
	const bool dbl = es == 8;
	if (ensure_ll(ge, batch, es))
		return 1;

	// where the running LL band lives: -1 = in `cur` image (src before level 0, dst after), else scratch index
	int ll_in = -1;
	Img cur = src;
	for (int j = 0; j < J; j++) {
		const int Wo = ge.Wo(j), Ho = ge.Ho(j), Wi = ge.Wi(j), Hi = ge.Hi(j);
		const int Wd = ge.Wo(j + 1), Hd = ge.Ho(j + 1);
		if (level_fused_ok(ge, j)) {
			const bool last = (j == J - 1) || !level_fused_ok(ge, j + 1);
			FwdLevelArgs a;
			a.W = Wo;
			a.H = Ho;
			a.batch = batch;
			bool detour = false;
			if (ll_in < 0) {
				a.in = cur.p;
				a.in_pitch = cur.sx / es;
				a.in_bstride = (cur.p == src.p ? src_bstride : dst_bstride) / es;
				detour = (cur.p == dst.p); // reading the image we also write: stage the outputs
			} else {
				a.in = ll_band(ll_in);
				a.in_pitch = ll_pitch_elems(Wo);
				a.in_bstride = a.in_pitch * Ho;
			}
			Img hdst = dst;
			long h_bstride = dst_bstride;
			if (detour) {
				if (batch != 1)
					return fail("in-place batches are not supported; use distinct src and dst");
				if (grow(&g.stage_img, &g.stage_bytes, (size_t)dst.sx * Ho))
					return 1;
				hdst = Img{(char *)g.stage_img, dst.sx, es};
				h_bstride = 0;
				// the copy-back below reads the staged subbands at once: temporal stores leave them in the 256 MiB
				// Infinity Cache (one 8192^2 image: 218.5 -> 205 us; the copy's own stores stay non-temporal: 204 against 219-225)
				a.temporal = 1;
			}
			a.out_h = hdst.p;
			a.h_pitch = hdst.sx / es;
			a.h_bstride = h_bstride / es;
			// ping-pong: the other buffer than the one read (after a fused pair the parity
			// of the level no longer tells which one that is); band j+1 fits either for j >= 1
			const int ll_out = last ? -1 : (ll_in < 0 ? (j & 1) : 1 - ll_in);
			if (last) {
				a.out_ll = hdst.p;
				a.ll_pitch = a.h_pitch;
				a.ll_bstride = a.h_bstride;
			} else {
				a.out_ll = ll_band(ll_out);
				a.ll_pitch = ll_pitch_elems(Wd);
				a.ll_bstride = a.ll_pitch * Hd;
			}
			SweepTuning tune = g.tune;
			if (!dbl && tune.tile_pairs <= 0)
				tune.tile_pairs = tuned_tile_pairs(w, a); // 0: the launcher's own rule
			prof_before(j);
			hipError_t e = dbl ? launch_fwd_level_d(w, a, tune, g.stream)
			                   : launch_fwd_level((g.fma && w == kCdf97S) ? kCdf97SFma : w, a, tune, g.stream);
			prof_after(j);
			if (e != hipSuccess)
				return fail("forward level %d launch failed: %s", j, hipGetErrorString(e));
			if (detour) {
				// copy the staged subbands to their place: right half, bottom-left, and the LL
				// quadrant too when it was written here (in line: on a side stream beside the deeper
				// levels it measured 8-10 us slower, profiles/r03_entries_summary.md)
				const Rect rc[3] = {{Wd, 0, Wd, 0, Wo - Wd, Ho}, {0, Hd, 0, Hd, Wd, Ho - Hd}, {0, 0, 0, 0, last ? Wd : 0, Hd}};
				if (copy_rects_on(g.stream, dst, hdst, rc, 3))
					return 1;
			}
			ll_in = ll_out;
			cur = dst;
			continue;
		}

		// ---- generic level: exact line semantics, in place on dst ----
		if (batch != 1)
			return fail("batched transforms need dense frames with both sides >= 2 at every level");
		if (ll_in >= 0) {
			// bring the LL band back into the image
			Img s{ll_band(ll_in), ll_pitch_elems(Wo) * es, es};
			if (copy_rect(dst, 0, 0, s, 0, 0, Wo, Ho))
				return 1;
			ll_in = -1;
			cur = dst;
		}
		if (Wi == Wo && Hi == Ho && Wo >= 2 && Ho >= 2) {
			// dense frame: each pass writes every element of the level's frame, so the two passes
			// ping-pong through the staging image (rows: image -> stage, columns: stage -> image)
			// instead of each staging and copying back a frame of its own: 4 instead of 12 frame
			// transfers per level (the double-precision drivers and accel 1 live on these passes)
			if (grow(&g.stage_img, &g.stage_bytes, (size_t)dst.sx * Ho))
				return 1;
			const Img S{(char *)g.stage_img, dst.sx, dst.es};
			if (generic_pass(w, false, true, cur, S, Wo, Ho, Ho, Wi, Wd) || generic_pass(w, false, false, S, dst, Wo, Ho, Wo, Hi, Hd))
				return 1;
			cur = dst;
		} else {
			if (!skip_single(w) || Wo > 1) {
				if (generic_pass(w, false, true, cur, dst, Wo, Ho, Ho, Wi, Wd))
					return 1;
				cur = dst; // src/libdwt.c:12709
			}
			if (!skip_single(w) || Ho > 1) {
				if (generic_pass(w, false, false, cur, dst, Wo, Ho, Wo, Hi, Hd))
					return 1;
				cur = dst; // src/libdwt.c:12742
			}
		}
		if (zero_padding) {
			// dwt_zero_padding_f_stride_* (src/libdwt.c:12079-12131) over rows then columns
			const int nl_x = (Wi + 1) >> 1, nh_x = Wi >> 1, nl_y = (Hi + 1) >> 1, nh_y = Hi >> 1;
			if (zero_rect(dst, nl_x, 0, Wd - nl_x, Ho) || zero_rect(dst, Wd + nh_x, 0, (Wo - Wd) - nh_x, Ho) ||
				zero_rect(dst, 0, nl_y, Wo, Hd - nl_y) || zero_rect(dst, 0, Hd + nh_y, Wo, (Ho - Hd) - nh_y))
				return 1;
		}
	}
	return 0;
}

// ---- inverse ---------------------------------------------------------------------
int inverse2d(Wavelet w, Img src, Img dst, const Geom &ge, int j_max, int decompose_one, int zero_padding,
	int batch, long src_bstride, long dst_bstride)
{
	const int so_min = ge.sox < ge.soy ? ge.sox : ge.soy, so_max = ge.sox > ge.soy ? ge.sox : ge.soy;
	int J = ceil_log2(decompose_one ? so_max : so_min);
	if (j_max >= 0 && j_max < J)
		J = j_max; // src/libdwt.c:17069-17072
	if (J == 0) {
		// dwt_cdf97_2i_s2 still copies the inner region (src/libdwt.c:18001-18008)
		if (src.p != dst.p && copy_rect(dst, 0, 0, src, 0, 0, ge.six, ge.siy))
			return 1;
		return 0;
	}
	const int es = elem_size(w);
	const bool dbl = es == 8;
	if (ensure_ll(ge, batch, es))
		return 1;
	const bool cols_first = (w == kCdf53I || w == kCdf97I); // the int inverses undo columns first

	// reconstruction level j consumes the subbands of size ceil(.,j) and produces the
	// band of size ceil(.,j-1); it is fused when that PRODUCED frame is dense and >= 2
	auto fused_ok = [&](int j) { return level_fused_ok(ge, j - 1); };

	Img cur = src;          // image holding the not-yet-consumed subbands
	long cur_bstride = src_bstride;
	int ll_in = -1;         // -1: LL band is in `cur`; else scratch index
	bool copied = false;
	for (int j = J; j >= 1; j--) {
		const int Ws = ge.Wo(j), Hs = ge.Ho(j);       // subband sizes (= Mallat offsets)
		const int Wo = ge.Wo(j - 1), Ho = ge.Ho(j - 1); // produced frame
		const int Wi = ge.Wi(j - 1), Hi = ge.Hi(j - 1);
		if (fused_ok(j)) {
			InvLevelArgs a;
			a.W = Wo;
			a.H = Ho;
			a.batch = batch;
			a.in_h = cur.p;
			a.h_pitch = cur.sx / es;
			a.h_bstride = cur_bstride / es;
			if (ll_in < 0) {
				a.in_ll = cur.p;
				a.ll_pitch = cur.sx / es;
				a.ll_bstride = cur_bstride / es;
			} else {
				a.in_ll = ll_band(ll_in);
				a.ll_pitch = ll_pitch_elems(Ws);
				a.ll_bstride = a.ll_pitch * Hs;
			}
			const bool last = (j == 1);
			int ll_out = -1;
			if (last) {
				a.out = dst.p;
				a.out_pitch = dst.sx / es;
				a.out_bstride = dst_bstride / es;
				if (cur.p == dst.p) {
					// in place: the final level would overwrite subbands it still reads;
					// move them (right half + bottom-left, and LL if it is still there) aside
					if (batch != 1)
						return fail("in-place batches are not supported; use distinct src and dst");
					if (grow(&g.stage_img, &g.stage_bytes, (size_t)dst.sx * Ho))
						return 1;
					Img st{(char *)g.stage_img, dst.sx, es};
					// (in line, like the forward copy-back: started early on a side stream beside the deeper levels it
					// measured slower)
					const Rect rc[3] = {{Ws, 0, Ws, 0, Wo - Ws, Ho}, {0, Hs, 0, Hs, Ws, Ho - Hs}, {0, 0, 0, 0, ll_in < 0 ? Ws : 0, Hs}};
					if (copy_rects_on(g.stream, st, cur, rc, 3, /* temporal both ways: the final level reads the staged subbands (233 against 237 us) */ 0))
						return 1;
					a.in_h = st.p;
					a.h_bstride = 0;
					if (ll_in < 0)
						a.in_ll = st.p;
				}
			} else {
				ll_out = j & 1; // band of level m = j-1 lives in scratch (m-1)&1, as in the forward driver
				a.out = ll_band(ll_out);
				a.out_pitch = ll_pitch_elems(Wo);
				a.out_bstride = a.out_pitch * Ho;
			}
			SweepTuning tune = g.tune;
			if (!dbl && tune.tile_pairs <= 0)
				tune.tile_pairs = tuned_tile_pairs(w, a);
			prof_before(j - 1);
			hipError_t e = dbl ? launch_inv_level_d(w, a, tune, g.stream)
			                   : launch_inv_level((g.fma && w == kCdf97S) ? kCdf97SFma : w, a, tune, g.stream);
			prof_after(j - 1);
			if (e != hipSuccess)
				return fail("inverse level %d launch failed: %s", j, hipGetErrorString(e));
			ll_in = ll_out;
			continue;
		}

		// ---- generic level, in place on dst ----
		if (batch != 1)
			return fail("batched transforms need dense frames with both sides >= 2 at every level");
		if (src.p != dst.p && !copied) {
			// the `_s2` entry copies the inner region, then works in place (:18001-18008)
			if (copy_rect(dst, 0, 0, src, 0, 0, ge.six, ge.siy))
				return 1;
			copied = true;
		}
		cur = dst;
		cur_bstride = dst_bstride;
		if (ll_in >= 0) {
			// a deeper fused level left its result in scratch: bring it back into the image
			Img s{ll_band(ll_in), ll_pitch_elems(Ws) * es, es};
			if (copy_rect(dst, 0, 0, s, 0, 0, Ws, Hs))
				return 1;
			ll_in = -1;
		}
		if (Wi == Wo && Hi == Ho && Wo >= 2 && Ho >= 2) {
			// dense frame: ping-pong through the staging image, as in the forward driver
			if (grow(&g.stage_img, &g.stage_bytes, (size_t)dst.sx * Ho))
				return 1;
			const Img S{(char *)g.stage_img, dst.sx, dst.es};
			const bool rows_first = !cols_first;
			if (generic_pass(w, true, rows_first, dst, S, Wo, Ho, rows_first ? Ho : Wo, rows_first ? Wi : Hi, rows_first ? Ws : Hs) ||
				generic_pass(w, true, !rows_first, S, dst, Wo, Ho, rows_first ? Wo : Ho, rows_first ? Hi : Wi, rows_first ? Hs : Ws))
				return 1;
		} else
		for (int pass = 0; pass < 2; pass++) {
			const bool rows = cols_first ? (pass == 1) : (pass == 0);
			if (rows) {
				if (!skip_single(w) || Wo > 1)
					if (generic_pass(w, true, true, dst, dst, Wo, Ho, Ho, Wi, Ws))
						return 1;
			} else {
				if (!skip_single(w) || Ho > 1)
					if (generic_pass(w, true, false, dst, dst, Wo, Ho, Wo, Hi, Hs))
						return 1;
			}
		}
		if (zero_padding) {
			// dwt_zero_padding_i_stride_* (src/libdwt.c:12161-12215)
			if (zero_rect(dst, Wi, 0, Wo - Wi, Ho) || zero_rect(dst, 0, Hi, Wo, Ho - Hi))
				return 1;
		}
	}
	return 0;
}

int check_inited()
{
	if (!g.inited && dwt_hip_init())
		return 1;
	return 0;
}

// ---- placement of the LL scratch ------------------------------------------------------------------
// The rate of a forward level depends on where in PHYSICAL memory its three streams lie relative to each
// other -- source rows, detail subbands, running LL band (profiles/r04_placement.md: coarse regions of
// three classes; +13 % when the two write streams are in different ones) -- and nothing finer than that
// matters.  The caller owns source and destination; the LL scratch is the library's.  So the first
// forward call that needs a large scratch tries a few allocations of it, each behind a spacer that pushes
// it into other physical memory, times the call itself on each (it writes exactly what the call will
// write: idempotent for distinct source and destination), and keeps the fastest.  Once per
// size: later calls find the scratch in place, allocate nothing and never synchronise.
int timed_forward(Wavelet w, Img s, Img d, const Geom &ge, int levels, int batch, long sb, long db, double *ms)
{
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	int rc = 0;
	g.placing = true;
	for (int r = 0; r < 2 && !rc; r++) {
		int j = levels;
		hipEventRecord(e0, g.stream);
		rc = forward2d(w, s, d, ge, &j, 0, 0, batch, sb, db);
		hipEventRecord(e1, g.stream);
	}
	g.placing = false;
	float t = 0;
	if (!rc && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess))
		rc = fail("timing a placement trial failed: %s", hipGetErrorString(hipGetLastError()));
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	*ms = t;
	return rc;
}

// spacer in front of candidate k of a placement search
static size_t place_jump(int k)
{
	return k <= 0 ? 0 : ((size_t)14 << 30) << (k > 3 ? 2 : k - 1);
}

bool stream_is_capturing()
{
	hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing(g.stream, &st) != hipSuccess) {
		(void)hipGetLastError();
		return true; // unknown: do nothing that synchronises
	}
	return st != hipStreamCaptureStatusNone;
}

int place_ll_scratch(Wavelet w, Img s, Img d, const Geom &ge, int levels, int batch, long sb, long db)
{
	const int es = elem_size(w);
	const size_t need[2] = {ll_band_bytes(ge, 0, batch, es), ll_band_bytes(ge, 1, batch, es)};
	g.place_n = 0;
	g.place_best = -1;
	if (g.placing || g.ll_external || g.place_tries < 2 || s.p == d.p || (g.ll_bytes[0] >= need[0] && g.ll_bytes[1] >= need[1]) ||
		need[0] + need[1] < ((size_t)g.place_min_mib << 20) || !ge.dense() || ge.Wo(2) < 2 || ge.Ho(2) < 2 || g.force_generic ||
		stream_is_capturing())
		return 0;
	struct Cand {
		void *ll[2], *spacer;
		double ms;
	};
	std::vector<Cand> cands;
	int rc = 0;
	for (int k = 0; k < g.place_tries && k < 8 && !rc; k++) {
		Cand c{{nullptr, nullptr}, nullptr, 0};
		// the spacers stay allocated during the search, so the jumps add up: candidates 14, 44, 104 ... GiB
		// further on (the classes come in 16 GiB granules, runs of one class can be 64 GiB long)
		const size_t jump = place_jump(k);
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need[0] + need[1] + (k ? jump : 0) + ((size_t)2 << 30))
			break;
		if (k && hipMalloc(&c.spacer, jump) != hipSuccess) {
			(void)hipGetLastError();
			break;
		}
		if (hipMalloc(&c.ll[0], need[0]) != hipSuccess || hipMalloc(&c.ll[1], need[1]) != hipSuccess) {
			(void)hipGetLastError();
			for (void *p : {c.ll[0], c.ll[1], c.spacer})
				if (p)
					hipFree(p);
			break;
		}
		// the context works on this candidate for the trial
		if (g.ll[0] || g.ll[1])
			HIP_TRY(hipStreamSynchronize(g.stream));
		for (int b = 0; b < 2; b++) {
			if (cands.empty() && g.ll[b])
				dev_free(g.ll[b]); // the too-small scratch of earlier calls
			g.ll[b] = c.ll[b];
			g.ll_bytes[b] = need[b];
		}
		rc = timed_forward(w, s, d, ge, levels, batch, sb, db, &c.ms);
		cands.push_back(c);
	}
	if (cands.empty()) {
		g.ll[0] = g.ll[1] = nullptr;
		g.ll_bytes[0] = g.ll_bytes[1] = 0;
		return rc; // nothing allocated here: the call allocates plainly
	}
	int best = 0;
	for (size_t k = 0; k < cands.size(); k++) {
		if (cands[k].ms < cands[best].ms)
			best = (int)k;
		g.place_ms[k] = cands[k].ms;
	}
	g.place_n = (int)cands.size();
	g.place_best = best;
	HIP_TRY(hipStreamSynchronize(g.stream));
	for (size_t k = 0; k < cands.size(); k++) {
		if (cands[k].spacer)
			hipFree(cands[k].spacer);
		if ((int)k != best) {
			hipFree(cands[k].ll[0]);
			hipFree(cands[k].ll[1]);
		}
	}
	for (int b = 0; b < 2; b++) {
		g.ll[b] = cands[best].ll[b];
		g.ll_bytes[b] = need[b];
	}
	return rc;
}

} // namespace dwtb

using namespace dwtb;

// ---- C ABI ------------------------------------------------------------------------
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
	for (auto &ev : g.prof_events) {
		hipEventDestroy(ev.first);
		hipEventDestroy(ev.second);
	}
	g.prof_events.clear();
	g.prof_used = 0;
	// the context stays usable: a later call re-allocates its workspace
}

const char *dwt_hip_device_name(void)
{
	if (check_inited())
		return "";
	return g.devname;
}

void dwt_hip_set_stream(void *s) { g.stream = (hipStream_t)s; }

int dwt_hip_set_workspace(void *band0, size_t bytes0, void *band1, size_t bytes1)
{
	if (check_inited())
		return 1;
	HIP_TRY(hipStreamSynchronize(g.stream));
	if (!g.ll_external) {
		for (int k = 0; k < 2; k++) {
			if (g.ll[k])
				HIP_TRY(hipFree(g.ll[k]));
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

int dwt_hip_set_option(const char *name, int value)
{
	g.tile_cache.clear(); // measured tile heights belong to the options they were measured under
	if (!strcmp(name, "generic"))
		g.force_generic = value;
	else if (!strcmp(name, "cpt"))
		g.tune.cpt = value;
	else if (!strcmp(name, "tile_pairs"))
		g.tune.tile_pairs = value;
	else if (!strcmp(name, "waves"))
		g.tune.waves = value;
	else if (!strcmp(name, "xcd_swizzle"))
		g.tune.xcd_swizzle = value;
	else if (!strcmp(name, "ring"))
		g.tune.ring = value;
	else if (!strcmp(name, "nt_auto"))
		g.tune.nt_auto = value;
	else if (!strcmp(name, "il_exact_borders"))
		g.il_exact_borders = value;
	else if (!strcmp(name, "il_inplace_shell"))
		g.il_inplace_shell = value;
	else if (!strcmp(name, "host_pipeline"))
		g.host_pipeline = value;
	else if (!strcmp(name, "vol_ip_waves"))
		g.vol.ip_waves = value;
	else if (!strcmp(name, "nt"))
		g.tune.nt = value;
	else if (!strcmp(name, "ring_inv"))
		g.tune.ring_inv = value;
	else if (!strcmp(name, "fma"))
		g.fma = value;
	else if (!strcmp(name, "fused_d"))
		g.fused_d = value;
	else if (!strcmp(name, "tune_tiles"))
		g.tune_tiles = value;
	else if (!strcmp(name, "place_tries"))
		g.place_tries = value;
	else if (!strcmp(name, "place_min_mib"))
		g.place_min_mib = value < 0 ? 0 : value;
	else if (!strcmp(name, "vol_tile_pairs"))
		g.vol.tile_pairs = value;
	else if (!strcmp(name, "vol_nt"))
		g.vol.nt = value;
	else if (!strcmp(name, "vol_fused"))
		g.vol.fused = value;
	else if (!strcmp(name, "vol_direct"))
		g.vol.direct = value;
	else if (!strcmp(name, "vol_whole"))
		g.vol.whole = value;
	else if (!strcmp(name, "vol_inplace_fused"))
		g.vol.inplace_fused = value ? 1 : 0;
	else if (!strcmp(name, "vol_swizzle"))
		g.vol.swizzle = value;
	else if (!strcmp(name, "vol_rows"))
		g.vol.rows = value;
	else
		return fail("unknown option '%s'", name);
	return 0;
}

int dwt_hip_get_option(const char *name)
{
	if (!strcmp(name, "generic"))
		return g.force_generic;
	if (!strcmp(name, "cpt"))
		return g.tune.cpt;
	if (!strcmp(name, "tile_pairs"))
		return g.tune.tile_pairs;
	if (!strcmp(name, "waves"))
		return g.tune.waves;
	if (!strcmp(name, "xcd_swizzle"))
		return g.tune.xcd_swizzle;
	if (!strcmp(name, "ring"))
		return g.tune.ring;
	if (!strcmp(name, "nt_auto"))
		return g.tune.nt_auto;
	if (!strcmp(name, "il_exact_borders"))
		return g.il_exact_borders;
	if (!strcmp(name, "il_inplace_shell"))
		return g.il_inplace_shell;
	if (!strcmp(name, "host_pipeline"))
		return g.host_pipeline;
	if (!strcmp(name, "vol_ip_waves"))
		return g.vol.ip_waves;
	if (!strcmp(name, "nt"))
		return g.tune.nt;
	if (!strcmp(name, "ring_inv"))
		return g.tune.ring_inv;
	if (!strcmp(name, "vol_swizzle"))
		return g.vol.swizzle;
	if (!strcmp(name, "vol_rows"))
		return g.vol.rows;
	if (!strcmp(name, "fma"))
		return g.fma;
	if (!strcmp(name, "fused_d"))
		return g.fused_d;
	if (!strcmp(name, "vol_tile_pairs"))
		return g.vol.tile_pairs;
	if (!strcmp(name, "vol_nt"))
		return g.vol.nt;
	if (!strcmp(name, "vol_fused"))
		return g.vol.fused;
	if (!strcmp(name, "vol_direct"))
		return g.vol.direct;
	if (!strcmp(name, "vol_whole"))
		return g.vol.whole;
	if (!strcmp(name, "vol_inplace_fused"))
		return g.vol.inplace_fused;
	if (!strcmp(name, "tune_tiles"))
		return g.tune_tiles;
	if (!strcmp(name, "place_tries"))
		return g.place_tries;
	if (!strcmp(name, "place_min_mib"))
		return g.place_min_mib;
	if (!strcmp(name, "place_last_tries")) // candidates the last placement search timed (0: none ran)
		return g.place_n;
	if (!strcmp(name, "place_last_best"))
		return g.place_best;
	return -1;
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

// Pins a caller's host range for the duration of a call, unless it is pinned memory already (hipHostMalloc, or
// registered by the caller): ok() says whether asynchronous copies may address it.
struct HostPin {
	void *p = nullptr;
	bool ours = false, good = false;
	HostPin(const void *ptr, size_t bytes)
	{
		hipPointerAttribute_t at;
		if (hipPointerGetAttributes(&at, ptr) == hipSuccess && at.type == hipMemoryTypeHost) {
			good = true; // the caller's own pinned memory
			return;
		}
		(void)hipGetLastError();
		if (hipHostRegister((void *)ptr, bytes, hipHostRegisterDefault) == hipSuccess) {
			p = (void *)ptr;
			ours = good = true;
		} else {
			(void)hipGetLastError();
		}
	}
	~HostPin()
	{
		if (ours)
			hipHostUnregister(p);
	}
	bool ok() const { return good; }
};

// ---- host-pointer forward call on a large image: level 0 band by band under the transfers ----
// A host-pointer call is bound by PCIe: 8192^2 floats take 4.7 ms each way against 0.15 ms of kernels.  The two
// directions are independent links, so the call is cut into bands of 512 row pairs: while band g+1 is still on its
// way up, band g's tiles of level 0 run (FwdLevelArgs::pair_lo / pair_hi) and their detail rows -- three quarters of
// the result -- travel down.  The caller's image is pinned in place for the call (hipHostRegister, 0.6 ms for
// 256 MiB the first time): every copy is an asynchronous DMA from / to it, no repacking on the CPU.  In place the
// rows Hd + [A, B) that band [A, B)'s LH / HH rows will overwrite are uploaded together with the band itself, so
// that no output lands on input that has not been read.  The deeper levels run on the complete low-pass band at the
// end and its quadrant follows.  Returns 0 done, 1 error, -1 not applicable (the caller takes the plain path).
static int host_forward_pipelined(Wavelet w, const void *src, void *dst, int stride_x, int W, int H, int *jp, int decompose_one)
{
	const Geom ge{W, H, W, H};
	const int Hd = (H + 1) / 2, Wd = (W + 1) / 2, Hh = H / 2;
	// row pairs per band: a multiple of every tile height, at most 16 bands, 256 pairs where that is enough (the first
	// band's upload and the last band's download overlap with nothing; 8192^2 in bands of 256 / 512 / 1024 pairs: forward
	// 7.42 / 7.44 / 7.95 ms, inverse 7.17 / 7.31 / 7.79)
	static const int band_opt = getenv("DWT_HIP_PIPE_BAND") ? atoi(getenv("DWT_HIP_PIPE_BAND")) : 256;
	const int kBand = std::max(band_opt, ((Hd + 15) / 16 + 63) / 64 * 64);
	const int n_bands = (Hd + kBand - 1) / kBand;
	// (decompose_one: the levels past the shorter side's last run on the generic passes, which borrow the buffers used here;
	// a tile height set by hand must divide the band: tiles do not straddle bands)
	if (!g.host_pipeline || decompose_one || kBand % 64 || (g.tune.tile_pairs > 0 && kBand % g.tune.tile_pairs) || elem_size(w) != 4 || !level_fused_ok(ge, 0) || (size_t)W * H * 4 < ((size_t)64 << 20) || n_bands < 2 || n_bands > 16 ||
		stride_x % 4 || stride_x < W * 4)
		return -1;
	const int j_lim = ceil_log2(decompose_one ? (W > H ? W : H) : (W < H ? W : H));
	const int J = (*jp < 0 || *jp > j_lim) ? j_lim : *jp;
	if (J < 1)
		return -1;
	const long pitch = align_up((long)W * 4, 256);
	const size_t span = (size_t)(H - 1) * stride_x + (size_t)W * 4;
	// pin the caller's image(s) where they are
	const bool two = src != dst;
	HostPin pin_src(src, span);
	if (!pin_src.ok())
		return -1;
	HostPin pin_dst(two ? dst : src, two ? span : 0);
	if (two && !pin_dst.ok())
		return -1;
	auto body = [&]() -> int {
		if (grow(&g.host_a, &g.host_a_bytes, (size_t)pitch * H) || grow(&g.host_b, &g.host_b_bytes, (size_t)pitch * H))
			return 1;
		// level 1 follows level 0 band by band too (its details are three quarters of the low-pass quadrant): scratch
		// for both low-pass bands
		const int Wd1 = (Wd + 1) / 2, Hd1 = (Hd + 1) / 2, Hh1 = Hd / 2;
		const long llp = align_up((long)Wd, 64), llp1 = align_up((long)Wd1, 64);
		const bool lvl1 = J > 1 && Wd >= 2 && Hd >= 2;
		if (J > 1 && grow(&g.stage_img, &g.stage_bytes, ((size_t)llp * Hd + (size_t)llp1 * Hd1) * 4))
			return 1;
		float *const ll0 = (float *)g.stage_img, *const ll1 = ll0 + (size_t)llp * Hd;
		int done1 = 0; // level 1: row pairs computed so far
		if (!g.up) {
			HIP_TRY(hipStreamCreateWithFlags(&g.up, hipStreamNonBlocking));
			HIP_TRY(hipStreamCreateWithFlags(&g.down, hipStreamNonBlocking));
		}
		for (auto &row : g.pipe_ev)
			for (int k = 0; k < 16; k++)
				if (!row[k])
					HIP_TRY(hipEventCreateWithFlags(&row[k], hipEventDisableTiming));
		char *A = (char *)g.host_a, *B = (char *)g.host_b;
		// everything queued on the caller's stream so far comes first
		HIP_TRY(hipEventRecord(g.pipe_ev[0][15], g.stream));
		HIP_TRY(hipStreamWaitEvent(g.up, g.pipe_ev[0][15], 0));
		HIP_TRY(hipStreamWaitEvent(g.down, g.pipe_ev[0][15], 0));
		// One copy per band and direction.  (Measured: cutting them into pieces of 2-16 MiB, or plain instead of 2-D
		// copies where the rows lie back to back, made the call slower or erratic -- 7.8-10.8 ms against 7.5.  The two
		// directions overlap only in part on this platform: 256 MiB each way at once from pinned memory take 9.4 ms as two
		// copies, 5.9 ms as 32 + 32; scripts/probes/r04_duplex_probe.py.)
		auto up_rows = [&](int r0, int r1) -> int {
			if (r1 > r0)
				HIP_TRY(hipMemcpy2DAsync(A + (long)r0 * pitch, pitch, (const char *)src + (long)r0 * stride_x, stride_x, (size_t)W * 4, r1 - r0,
					hipMemcpyHostToDevice, g.up));
			return 0;
		};
		auto down_rect = [&](int r0, int r1, int c0, int c1) -> int {
			if (r1 > r0 && c1 > c0)
				HIP_TRY(hipMemcpy2DAsync((char *)dst + (long)r0 * stride_x + (long)c0 * 4, stride_x, B + (long)r0 * pitch + (long)c0 * 4, pitch,
					(size_t)(c1 - c0) * 4, r1 - r0, hipMemcpyDeviceToHost, g.down));
			return 0;
		};
		// (DWT_HIP_PIPE_VERBOSE: when each stream finishes, from the call's start)
		static const bool verbose = getenv("DWT_HIP_PIPE_VERBOSE") != nullptr;
		hipEvent_t tv[4] = {};
		if (verbose) {
			for (auto &e : tv)
				hipEventCreate(&e);
			hipEventRecord(tv[0], g.up);
		}
		int top_end = 0, bot_end = Hd; // rows [0, top_end) and [Hd, bot_end) are on their way up
		// In place a result may only come down onto rows that have gone up: rectangles wait here until they may
		// (level 1's LH / HH rows lie ahead of the upload front for a few bands)
		struct Pending { int r0, r1, c0, c1; };
		Pending pend[64];
		int n_pend = 0;
		auto flush = [&](bool all) -> int {
			int keep = 0;
			for (int i = 0; i < n_pend; i++) {
				const Pending q = pend[i];
				const bool up = all || q.r1 <= top_end || (q.r0 >= Hd && q.r1 <= bot_end) || (top_end >= Hd && q.r1 <= (top_end > bot_end ? top_end : bot_end));
				if (!up)
					pend[keep++] = q;
				else if (down_rect(q.r0, q.r1, q.c0, q.c1))
					return 1;
			}
			n_pend = keep;
			return 0;
		};
		auto later = [&](int r0, int r1, int c0, int c1) {
			if (r1 > r0 && c1 > c0 && n_pend < 64)
				pend[n_pend++] = Pending{r0, r1, c0, c1};
		};
		for (int b = 0; b < n_bands; b++) {
			const int P0 = b * kBand, P1 = (b + 1) * kBand < Hd ? (b + 1) * kBand : Hd;
			// the band's input rows (its tiles read up to row 2 P1 + 2) ...
			int want = P1 == Hd ? H : (2 * P1 + 3 < H ? 2 * P1 + 3 : H);
			if (top_end >= Hd && top_end < bot_end)
				top_end = bot_end; // (those went up as some band's bottom rows)
			if (want > top_end) {
				// rows [Hd, bot_end) inside the range are up already
				if (top_end < Hd && want > Hd) {
					if (up_rows(top_end, Hd) || up_rows(bot_end > Hd ? bot_end : Hd, want > bot_end ? want : bot_end))
						return 1;
					bot_end = want > bot_end ? want : bot_end;
				} else if (up_rows(top_end, want)) {
					return 1;
				}
				top_end = want;
				if (top_end >= Hd && top_end > bot_end)
					bot_end = top_end;
			}
			// ... and the rows its LH / HH rows will land on
			const int b1 = Hd + P1 < H ? Hd + P1 : H;
			if (b1 > bot_end && b1 > top_end) {
				const int from = bot_end > top_end ? bot_end : top_end;
				if (up_rows(from > Hd ? from : Hd, b1))
					return 1;
				bot_end = b1;
			}
			HIP_TRY(hipEventRecord(g.pipe_ev[0][b], g.up));
			HIP_TRY(hipStreamWaitEvent(g.stream, g.pipe_ev[0][b], 0));
			FwdLevelArgs a;
			a.in = A; a.in_pitch = pitch / 4; a.in_bstride = 0;
			a.out_h = B; a.h_pitch = pitch / 4; a.h_bstride = 0;
			if (J > 1) {
				a.out_ll = ll0; a.ll_pitch = llp; a.ll_bstride = 0;
			} else {
				a.out_ll = B; a.ll_pitch = pitch / 4; a.ll_bstride = 0;
			}
			a.W = W; a.H = H; a.batch = 1;
			a.pair_lo = P0; a.pair_hi = P1 == Hd ? Hd + kBand : P1;
			hipError_t e = launch_fwd_level((g.fma && w == kCdf97S) ? kCdf97SFma : w, a, g.tune, g.stream);
			if (e != hipSuccess)
				return fail("forward level 0 (band %d) launch failed: %s", b, hipGetErrorString(e));
			// level 1 on the rows of the low-pass band that are complete now (its tiles read up to row 2 hi + 2)
			int lo1 = done1, hi1 = done1;
			if (lvl1) {
				hi1 = P1 == Hd ? Hd1 : ((P1 - 3) / 2) / 64 * 64;
				if (hi1 > lo1) {
					FwdLevelArgs a1;
					a1.in = ll0; a1.in_pitch = llp; a1.in_bstride = 0;
					a1.out_h = B; a1.h_pitch = pitch / 4; a1.h_bstride = 0;
					if (J > 2) {
						a1.out_ll = ll1; a1.ll_pitch = llp1; a1.ll_bstride = 0;
					} else {
						a1.out_ll = B; a1.ll_pitch = pitch / 4; a1.ll_bstride = 0;
					}
					a1.W = Wd; a1.H = Hd; a1.batch = 1;
					a1.pair_lo = lo1; a1.pair_hi = hi1 == Hd1 ? Hd1 + kBand : hi1;
					e = launch_fwd_level((g.fma && w == kCdf97S) ? kCdf97SFma : w, a1, g.tune, g.stream);
					if (e != hipSuccess)
						return fail("forward level 1 (band %d) launch failed: %s", b, hipGetErrorString(e));
					done1 = hi1;
				} else {
					hi1 = lo1;
				}
			}
			HIP_TRY(hipEventRecord(g.pipe_ev[1][b], g.stream));
			HIP_TRY(hipStreamWaitEvent(g.down, g.pipe_ev[1][b], 0));
			// the band's detail rows: HL beside the low-pass quadrant, LH | HH below it; level 1's alike inside the quadrant
			later(P0, P1, Wd, W);
			later(Hd + P0, Hd + (P1 < Hh ? P1 : Hh), 0, W);
			if (hi1 > lo1) {
				later(lo1, hi1 < Hd1 ? hi1 : Hd1, Wd1, Wd);
				later(Hd1 + lo1, Hd1 + (hi1 < Hh1 ? hi1 : Hh1), 0, Wd);
			}
			// (the download stream has just been made to wait for this band's uploads and kernels)
			if (flush(false))
				return 1;
		}
		if (verbose) {
			hipEventRecord(tv[1], g.up);
			hipEventRecord(tv[2], g.down);
		}
		// the deeper levels on the complete low-pass band of the last banded level, then that band's quadrant
		int qw = Wd, qh = Hd;
		if (lvl1) {
			qw = Wd1; qh = Hd1;
			if (J > 2) {
				int j2 = J - 2;
				const Geom gl{Wd1, Hd1, Wd1, Hd1};
				if (forward2d(w, Img{(char *)ll1, llp1 * 4, 4}, Img{B, pitch, 4}, gl, &j2, decompose_one, 0, 1, 0, 0))
					return 1;
			}
		}
		HIP_TRY(hipEventRecord(g.pipe_ev[2][0], g.stream));
		HIP_TRY(hipStreamWaitEvent(g.down, g.pipe_ev[2][0], 0));
		if (flush(true) || down_rect(0, qh, 0, qw))
			return 1;
		if (verbose)
			hipEventRecord(tv[3], g.down);
		HIP_TRY(hipStreamSynchronize(g.down));
		HIP_TRY(hipStreamSynchronize(g.up));
		if (verbose) {
			float up = 0, dd = 0, all = 0;
			hipEventElapsedTime(&up, tv[0], tv[1]);
			hipEventElapsedTime(&dd, tv[0], tv[2]);
			hipEventElapsedTime(&all, tv[0], tv[3]);
			fprintf(stderr, "host pipeline: uploads done at %.2f ms, detail downloads at %.2f ms, all at %.2f ms\n", up, dd, all);
			for (auto &e : tv)
				hipEventDestroy(e);
		}
		*jp = J;
		return 0;
	};
	const int rc = body();
	if (rc) {
		hipStreamSynchronize(g.up);
		hipStreamSynchronize(g.down);
		hipStreamSynchronize(g.stream);
	}
	return rc;
}

// The inverse likewise: the low-pass quadrant goes up first and the levels >= 1 run on it while the detail bands
// follow; band [P0, P1) of level 0 needs the HL rows up to P1 + 2 and the LH | HH rows up to Hd + P1 + 2, and its
// result -- rows [2 P0, 2 P1) of the image -- comes down at once.  In place that result overwrites coefficient rows:
// every row below 2 P1 goes up before it (a band's uploads run ahead of its own needs by that much).
static int host_inverse_pipelined(Wavelet w, const void *src, void *dst, int stride_x, int W, int H, int j_max, int decompose_one)
{
	const Geom ge{W, H, W, H};
	const int Hd = (H + 1) / 2, Wd = (W + 1) / 2, Hh = H / 2;
	static const int band_opt = getenv("DWT_HIP_PIPE_BAND") ? atoi(getenv("DWT_HIP_PIPE_BAND")) : 256;
	const int kBand = std::max(band_opt, ((Hd + 15) / 16 + 63) / 64 * 64); // (see host_forward_pipelined)
	const int n_bands = (Hd + kBand - 1) / kBand;
	// (decompose_one: the levels past the shorter side's last run on the generic passes, which borrow the buffers used here;
	// a tile height set by hand must divide the band: tiles do not straddle bands)
	if (!g.host_pipeline || decompose_one || kBand % 64 || (g.tune.tile_pairs > 0 && kBand % g.tune.tile_pairs) || elem_size(w) != 4 || !level_fused_ok(ge, 0) || (size_t)W * H * 4 < ((size_t)64 << 20) || n_bands < 2 || n_bands > 16 ||
		stride_x % 4 || stride_x < W * 4)
		return -1;
	int J = ceil_log2(decompose_one ? (W > H ? W : H) : (W < H ? W : H));
	if (j_max >= 0 && j_max < J)
		J = j_max;
	if (J < 1)
		return -1;
	const long pitch = align_up((long)W * 4, 256);
	const size_t span = (size_t)(H - 1) * stride_x + (size_t)W * 4;
	const bool two = src != dst;
	HostPin pin_src(src, span);
	if (!pin_src.ok())
		return -1;
	HostPin pin_dst(two ? dst : src, two ? span : 0);
	if (two && !pin_dst.ok())
		return -1;
	auto body = [&]() -> int {
		if (grow(&g.host_a, &g.host_a_bytes, (size_t)pitch * H) || grow(&g.host_b, &g.host_b_bytes, (size_t)pitch * H))
			return 1;
		const long llp = align_up((long)Wd, 64);
		if (J > 1 && grow(&g.stage_img, &g.stage_bytes, (size_t)llp * Hd * 4))
			return 1;
		if (!g.up) {
			HIP_TRY(hipStreamCreateWithFlags(&g.up, hipStreamNonBlocking));
			HIP_TRY(hipStreamCreateWithFlags(&g.down, hipStreamNonBlocking));
		}
		for (auto &row : g.pipe_ev)
			for (int k = 0; k < 16; k++)
				if (!row[k])
					HIP_TRY(hipEventCreateWithFlags(&row[k], hipEventDisableTiming));
		char *A = (char *)g.host_a, *B = (char *)g.host_b;
		HIP_TRY(hipEventRecord(g.pipe_ev[0][15], g.stream));
		HIP_TRY(hipStreamWaitEvent(g.up, g.pipe_ev[0][15], 0));
		HIP_TRY(hipStreamWaitEvent(g.down, g.pipe_ev[0][15], 0));
		auto up_rect = [&](int r0, int r1, int c0, int c1) -> int {
			if (r1 > r0 && c1 > c0)
				HIP_TRY(hipMemcpy2DAsync(A + (long)r0 * pitch + (long)c0 * 4, pitch, (const char *)src + (long)r0 * stride_x + (long)c0 * 4, stride_x,
					(size_t)(c1 - c0) * 4, r1 - r0, hipMemcpyHostToDevice, g.up));
			return 0;
		};
		// the low-pass quadrant first; the levels >= 1 rebuild the level-0 low-pass band from it
		if (up_rect(0, Hd, 0, Wd))
			return 1;
		HIP_TRY(hipEventRecord(g.pipe_ev[2][1], g.up));
		HIP_TRY(hipStreamWaitEvent(g.stream, g.pipe_ev[2][1], 0));
		const void *ll = A;
		long ll_pitch = pitch / 4;
		if (J > 1) {
			const Geom gl{Wd, Hd, Wd, Hd};
			if (inverse2d(w, Img{A, pitch, 4}, Img{(char *)g.stage_img, llp * 4, 4}, gl, J - 1, decompose_one, 0, 1, 0, 0))
				return 1;
			ll = g.stage_img;
			ll_pitch = llp;
		}
		int top_done = 0, bot_done = Hd; // HL rows [0, top_done) and image rows [Hd, bot_done) are on their way up
		for (int b = 0; b < n_bands; b++) {
			const int P0 = b * kBand, P1 = (b + 1) * kBand < Hd ? (b + 1) * kBand : Hd;
			const bool last = P1 == Hd;
			// what the band reads, and (in place) every row its result will overwrite
			int top_need = last ? Hd : (2 * P1 < Hd ? 2 * P1 : Hd);
			if (!last && top_need < P1 + 2)
				top_need = P1 + 2 < Hd ? P1 + 2 : Hd;
			int bot_need = last ? H : Hd + (P1 + 2 < Hh ? P1 + 2 : Hh);
			if (!last && 2 * P1 > bot_need)
				bot_need = 2 * P1 < H ? 2 * P1 : H;
			if (up_rect(top_done, top_need, Wd, W) || up_rect(bot_done, bot_need, 0, W))
				return 1;
			top_done = top_need > top_done ? top_need : top_done;
			bot_done = bot_need > bot_done ? bot_need : bot_done;
			HIP_TRY(hipEventRecord(g.pipe_ev[0][b], g.up));
			HIP_TRY(hipStreamWaitEvent(g.stream, g.pipe_ev[0][b], 0));
			InvLevelArgs a;
			a.W = W; a.H = H; a.batch = 1;
			a.in_h = A; a.h_pitch = pitch / 4; a.h_bstride = 0;
			a.in_ll = ll; a.ll_pitch = ll_pitch; a.ll_bstride = 0;
			a.out = B; a.out_pitch = pitch / 4; a.out_bstride = 0;
			a.pair_lo = P0; a.pair_hi = last ? Hd + kBand : P1;
			hipError_t e = launch_inv_level((g.fma && w == kCdf97S) ? kCdf97SFma : w, a, g.tune, g.stream);
			if (e != hipSuccess)
				return fail("inverse level 1 (band %d) launch failed: %s", b, hipGetErrorString(e));
			HIP_TRY(hipEventRecord(g.pipe_ev[1][b], g.stream));
			HIP_TRY(hipStreamWaitEvent(g.down, g.pipe_ev[1][b], 0));
			const int r0 = 2 * P0, r1 = last ? H : 2 * P1;
			HIP_TRY(hipMemcpy2DAsync((char *)dst + (long)r0 * stride_x, stride_x, B + (long)r0 * pitch, pitch, (size_t)W * 4, r1 - r0, hipMemcpyDeviceToHost, g.down));
		}
		HIP_TRY(hipStreamSynchronize(g.down));
		HIP_TRY(hipStreamSynchronize(g.up));
		return 0;
	};
	const int rc = body();
	if (rc) {
		hipStreamSynchronize(g.up);
		hipStreamSynchronize(g.down);
		hipStreamSynchronize(g.stream);
	}
	return rc;
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

	if (dev_dst) {
		if (stride_y != es || (stride_x % es) || stride_x < sox * es)
			return fail("device images need stride_y == %d and stride_x a multiple of it >= width*%d (got %d, %d)", es, es, stride_x, stride_y);
		Img s{(char *)src, stride_x, es}, d{(char *)dst, stride_x, es};
		if (!inverse && !decompose_one && (*j < 0 || *j >= 2) && place_ll_scratch(w, s, d, ge, *j, 1, 0, 0))
			return 1;
		return inverse ? inverse2d(w, s, d, ge, *j, decompose_one, zero_padding, 1, 0, 0)
		               : forward2d(w, s, d, ge, j, decompose_one, zero_padding, 1, 0, 0);
	}

	// ---- host pointers: stage the whole outer frame through HBM ----
	if (ge.dense() && stride_y == es && es == 4) {
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
	auto upload = [&](const void *hp, void *dp) -> int { return host_upload(hp, stride_x, stride_y, es, sox, soy, dp, pitch); };
	auto download = [&](void *hp, const void *dp) -> int { return host_download(hp, stride_x, stride_y, es, sox, soy, dp, pitch); };
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
