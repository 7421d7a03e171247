// dwt_host_xfer.hip -- host-pointer calls: host images / volumes <-> dense device images (any byte
// strides, src/system.c:102-164 is the reference's gather / scatter), and the pipelined host-pointer
// calls on large images (level 0 and 1 band by band under their own PCIe transfers).
#include "dwt_backend.h"
#include "dwt_host_pools.h"

namespace dwtb {

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
// scripts/archive/probes/r04_oddpitch_probe.py
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
// pitch runs at 1 GB/s, scripts/archive/probes/r04_oddpitch_probe.py) <-> a device volume: batches of slices of about 32 MiB
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
int host_forward_pipelined(Wavelet w, const void *src, void *dst, int stride_x, int W, int H, int *jp, int decompose_one)
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
		// copies, 5.9 ms as 32 + 32; scripts/archive/probes/r04_duplex_probe.py.)
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
int host_inverse_pipelined(Wavelet w, const void *src, void *dst, int stride_x, int W, int H, int j_max, int decompose_one)
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

} // namespace dwtb
