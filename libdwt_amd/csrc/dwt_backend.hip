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
#include "../../include/libdwt_hip.h"
#include "dwt_kernels.h"

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <utility>
#include <vector>

using namespace dwt;

namespace {

struct Ctx {
	bool inited = false;
	int device = 0;
	hipStream_t stream = nullptr;
	char devname[256] = {0};
	// workspace
	void *stage_img = nullptr; // frame-sized staging image (in-place detour, generic passes)
	size_t stage_bytes = 0;
	void *ll[2] = {nullptr, nullptr}; // LL ping-pong
	size_t ll_bytes[2] = {0, 0};
	void *host_a = nullptr, *host_b = nullptr; // device images for host-pointer calls
	size_t host_a_bytes = 0, host_b_bytes = 0;
	void *pin = nullptr; // pinned host staging for host-pointer calls with awkward strides
	size_t pin_bytes = 0;
	// pipeline lanes for batches: image k runs all its levels on lane k % lanes, so the
	// small tail levels of one image overlap the big levels of the next
	struct Lane {
		hipStream_t stream = nullptr;
		void *ll[2] = {nullptr, nullptr};
		size_t ll_bytes[2] = {0, 0};
		void *stage_img = nullptr;
		size_t stage_bytes = 0;
		hipEvent_t done = nullptr;
	};
	static constexpr int kMaxLanes = 4;
	Lane lanes[kMaxLanes];
	hipEvent_t fork = nullptr;
	// side stream: the copy-back of an in-place level 0 (and the copy-aside of an in-place
	// final inverse level) overlaps the small levels instead of preceding/following them
	hipStream_t side = nullptr;
	hipEvent_t side_a = nullptr, side_b = nullptr;
	bool side_pending = false;
	int pipeline = 0; // 0: one launch per level for the whole batch; n>=2: n lanes
	// options
	SweepTuning tune;
	VolTuning vol;
	int force_generic = 0;
	int fma = 0; // opt-in: contract the float 9/7 lifting steps (not bit-identical to libdwt)
	// profiling
	int prof_on = 0;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
	std::vector<int> prof_tag; // level index of each recorded pair
	size_t prof_used = 0;
	double prof_ms = 0;
	int prof_launches = 0;
	double prof_level_ms[16] = {0};
	int prof_level_n[16] = {0};
};

Ctx g;
thread_local char g_err[512] = "";

int fail(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return 1;
}

#define HIP_TRY(expr)                                                                          \
	do {                                                                                       \
		hipError_t e_ = (expr);                                                                \
		if (e_ != hipSuccess)                                                                  \
			return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

inline int ceil_div_pow2(int i, int j) { return (i + (1 << j) - 1) >> j; } // src/inline.h:455-461
inline int ceil_log2(int x)                                               // src/inline.h:443-448
{
	int n = 0;
	while (n < 31 && (1 << n) < x)
		n++;
	return n;
}
inline long align_up(long v, long a) { return (v + a - 1) / a * a; }

int grow(void **p, size_t *have, size_t need)
{
	if (*have >= need)
		return 0;
	if (*p) {
		HIP_TRY(hipStreamSynchronize(g.stream));
		HIP_TRY(hipFree(*p));
		*p = nullptr;
		*have = 0;
	}
	HIP_TRY(hipMalloc(p, need));
	*have = need;
	return 0;
}

// A device image: element (y,x) at p + y*sx + x*es (dense elements of es = 4 or 8 bytes).
struct Img {
	char *p;
	long sx;    // row pitch in bytes
	int es = 4; // element size in bytes
};

struct Geom {
	int sox, soy, six, siy;
	int Wo(int j) const { return ceil_div_pow2(sox, j); }
	int Ho(int j) const { return ceil_div_pow2(soy, j); }
	int Wi(int j) const { return ceil_div_pow2(six, j); }
	int Hi(int j) const { return ceil_div_pow2(siy, j); }
	bool dense() const { return sox == six && soy == siy; }
};

bool skip_single(Wavelet w) { return w == kCdf97S; } // only the 9/7 drivers guard on lines > 1

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

template <class F>
static void for_rows_parallel(int rows, size_t bytes_total, F f)
{
	unsigned nt = bytes_total >= (8u << 20) ? std::thread::hardware_concurrency() : 1;
	if (nt > 16)
		nt = 16;
	if (nt <= 1 || rows < 64) {
		f(0, rows);
		return;
	}
	std::vector<std::thread> th;
	const int chunk = (rows + (int)nt - 1) / (int)nt;
	for (unsigned t = 0; t < nt; t++) {
		const int a = (int)t * chunk, b = a + chunk < rows ? a + chunk : rows;
		if (a < b)
			th.emplace_back(f, a, b);
	}
	for (auto &x : th)
		x.join();
}

static bool host_pitch_is_fast(const void *hp, int stride_x, int stride_y, int es)
{
	return stride_y == es && stride_x % 64 == 0 && (uintptr_t)hp % 16 == 0;
}

// w x h elements of `es` bytes at hp (byte strides) -> device image dp with `pitch`
static int host_upload(const void *hp, int stride_x, int stride_y, int es, int w, int h, void *dp, long pitch)
{
	if (host_pitch_is_fast(hp, stride_x, stride_y, es)) {
		HIP_TRY(hipMemcpy2DAsync(dp, pitch, hp, stride_x, (size_t)w * es, h, hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
		return 0;
	}
	if (grow_pinned((size_t)pitch * h))
		return 1;
	char *pin = (char *)g.pin;
	for_rows_parallel(h, (size_t)pitch * h, [=](int y0, int y1) {
		for (int y = y0; y < y1; y++) {
			const char *row = (const char *)hp + (long)y * stride_x;
			char *out = pin + (long)y * pitch;
			if (stride_y == es) {
				memcpy(out, row, (size_t)w * es);
			} else {
				for (int x = 0; x < w; x++)
					memcpy(out + (long)x * es, row + (long)x * stride_y, es);
			}
		}
	});
	HIP_TRY(hipMemcpyAsync(dp, pin, (size_t)pitch * h, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	return 0;
}

static int host_download(void *hp, int stride_x, int stride_y, int es, int w, int h, const void *dp, long pitch)
{
	if (host_pitch_is_fast(hp, stride_x, stride_y, es)) {
		HIP_TRY(hipMemcpy2DAsync(hp, stride_x, dp, pitch, (size_t)w * es, h, hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
		return 0;
	}
	if (grow_pinned((size_t)pitch * h))
		return 1;
	char *pin = (char *)g.pin;
	HIP_TRY(hipMemcpyAsync(pin, dp, (size_t)pitch * h, hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	for_rows_parallel(h, (size_t)pitch * h, [=](int y0, int y1) {
		for (int y = y0; y < y1; y++) {
			char *row = (char *)hp + (long)y * stride_x;
			const char *in = pin + (long)y * pitch;
			if (stride_y == es) {
				memcpy(row, in, (size_t)w * es);
			} else {
				for (int x = 0; x < w; x++)
					memcpy(row + (long)x * stride_y, in + (long)x * es, es);
			}
		}
	});
	return 0;
}

int copy_rect(Img dst, long dx, long dy, Img src, long sx_, long sy_, long w, long h)
{
	return copy_rect_on(g.stream, dst, dx, dy, src, sx_, sy_, w, h);
}

// fork: work queued on the side stream from now on starts after everything already
// queued on the main stream; join: the main stream waits for the side stream
int side_fork()
{
	if (!g.side) {
		HIP_TRY(hipStreamCreateWithFlags(&g.side, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&g.side_a, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&g.side_b, hipEventDisableTiming));
	}
	HIP_TRY(hipEventRecord(g.side_a, g.stream));
	HIP_TRY(hipStreamWaitEvent(g.side, g.side_a, 0));
	g.side_pending = true;
	return 0;
}

int side_join()
{
	if (!g.side_pending)
		return 0;
	g.side_pending = false;
	HIP_TRY(hipEventRecord(g.side_b, g.side));
	HIP_TRY(hipStreamWaitEvent(g.stream, g.side_b, 0));
	return 0;
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
		if (in.p != out.p) {
			if (side_join())
				return 1;
			return copy_rect(out, 0, 0, in, 0, 0, frame_w, frame_h);
		}
		return 0;
	}
	if (side_join())
		return 1;
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

void prof_before(int level = 0)
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

void prof_after(int level = 0)
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

// make a lane's stream and scratch the active ones (and back)
void swap_lane(Ctx::Lane &l)
{
	std::swap(g.stream, l.stream);
	std::swap(g.ll[0], l.ll[0]);
	std::swap(g.ll[1], l.ll[1]);
	std::swap(g.ll_bytes[0], l.ll_bytes[0]);
	std::swap(g.ll_bytes[1], l.ll_bytes[1]);
	std::swap(g.stage_img, l.stage_img);
	std::swap(g.stage_bytes, l.stage_bytes);
}

long ll_pitch_elems(int w) { return align_up(w, 4); }

int ensure_ll(const Geom &ge, int batch)
{
	for (int k = 0; k < 2; k++) {
		const int w = ge.Wo(k + 1), h = ge.Ho(k + 1);
		if (grow(&g.ll[k], &g.ll_bytes[k], (size_t)ll_pitch_elems(w) * h * 4 * batch + 64))
			return 1;
	}
	return 0;
}

bool g_elems_are_32bit = true; // set per call: the fused sweeps exist for 4-byte elements only

bool level_fused_ok(const Geom &ge, int j)
{
	return !g.force_generic && g_elems_are_32bit && ge.Wi(j) == ge.Wo(j) && ge.Hi(j) == ge.Ho(j) && ge.Wo(j) >= 2 && ge.Ho(j) >= 2;
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
	if (ensure_ll(ge, batch))
		return 1;

	// where the running LL band lives: -1 = in `cur` image (src before level 0, dst after), else scratch index
	int ll_in = -1;
	Img cur = src;
	for (int j = 0; j < J; j++) {
		const int Wo = ge.Wo(j), Ho = ge.Ho(j), Wi = ge.Wi(j), Hi = ge.Hi(j);
		const int Wd = ge.Wo(j + 1), Hd = ge.Ho(j + 1);
		if (level_fused_ok(ge, j)) {
			// two levels in one sweep when both are fused-ok, nothing is in place and the
			// launch is big enough (fwd2_tile_pairs decides); the LL band between them
			// then never reaches HBM
			if (j + 1 < J && level_fused_ok(ge, j + 1) && !(ll_in < 0 && cur.p == dst.p)) {
				const int W2 = ge.Wo(j + 2), H2 = ge.Ho(j + 2);
				const bool last2 = (j + 1 == J - 1) || !level_fused_ok(ge, j + 2);
				Fwd2LevelArgs f;
				f.W = Wo;
				f.H = Ho;
				f.batch = batch;
				if (ll_in < 0) {
					f.in = cur.p;
					f.in_pitch = cur.sx / 4;
					f.in_bstride = (cur.p == src.p ? src_bstride : dst_bstride) / 4;
				} else {
					f.in = g.ll[ll_in];
					f.in_pitch = ll_pitch_elems(Wo);
					f.in_bstride = f.in_pitch * Ho;
				}
				f.out_h = dst.p;
				f.h_pitch = dst.sx / 4;
				f.h_bstride = dst_bstride / 4;
				const int ll_out2 = last2 ? -1 : ((j + 1) & 1);
				if (last2) {
					f.out_ll2 = dst.p;
					f.ll2_pitch = f.h_pitch;
					f.ll2_bstride = f.h_bstride;
				} else {
					f.out_ll2 = g.ll[ll_out2];
					f.ll2_pitch = ll_pitch_elems(W2);
					f.ll2_bstride = f.ll2_pitch * H2;
				}
				if (fwd2_tile_pairs(f, g.tune) > 0) {
					prof_before(j);
					hipError_t e = launch_fwd2_level((g.fma && w == kCdf97S) ? kCdf97SFma : w, f, g.tune, g.stream);
					prof_after(j);
					if (e != hipSuccess)
						return fail("forward levels %d+%d launch failed: %s", j, j + 1, hipGetErrorString(e));
					ll_in = ll_out2;
					cur = dst;
					j++; // the next level is done too
					continue;
				}
			}
			const bool last = (j == J - 1) || !level_fused_ok(ge, j + 1);
			FwdLevelArgs a;
			a.W = Wo;
			a.H = Ho;
			a.batch = batch;
			bool detour = false;
			if (ll_in < 0) {
				a.in = cur.p;
				a.in_pitch = cur.sx / 4;
				a.in_bstride = (cur.p == src.p ? src_bstride : dst_bstride) / 4;
				detour = (cur.p == dst.p); // reading the image we also write: stage the outputs
			} else {
				a.in = g.ll[ll_in];
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
				hdst = Img{(char *)g.stage_img, dst.sx};
				h_bstride = 0;
			}
			a.out_h = hdst.p;
			a.h_pitch = hdst.sx / 4;
			a.h_bstride = h_bstride / 4;
			const int ll_out = last ? -1 : (j & 1);
			if (last) {
				a.out_ll = hdst.p;
				a.ll_pitch = a.h_pitch;
				a.ll_bstride = a.h_bstride;
			} else {
				a.out_ll = g.ll[ll_out];
				a.ll_pitch = ll_pitch_elems(Wd);
				a.ll_bstride = a.ll_pitch * Hd;
			}
			prof_before(j);
			hipError_t e = launch_fwd_level((g.fma && w == kCdf97S) ? kCdf97SFma : w, a, g.tune, g.stream);
			prof_after(j);
			if (e != hipSuccess)
				return fail("forward level %d launch failed: %s", j, hipGetErrorString(e));
			if (detour) {
				// copy the staged subbands to their place: right half, bottom-left, and the LL
				// quadrant too when it was written here.  On the side stream: the deeper levels
				// only touch the top-left quadrant and run meanwhile.
				if (side_fork())
					return 1;
				if (copy_rect_on(g.side, dst, Wd, 0, hdst, Wd, 0, Wo - Wd, Ho) || copy_rect_on(g.side, dst, 0, Hd, hdst, 0, Hd, Wd, Ho - Hd))
					return 1;
				if (last && copy_rect_on(g.side, dst, 0, 0, hdst, 0, 0, Wd, Hd))
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
			Img s{(char *)g.ll[ll_in], ll_pitch_elems(Wo) * 4};
			if (copy_rect(dst, 0, 0, s, 0, 0, Wo, Ho))
				return 1;
			ll_in = -1;
			cur = dst;
		}
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
		if (zero_padding) {
			// dwt_zero_padding_f_stride_* (src/libdwt.c:12079-12131) over rows then columns
			const int nl_x = (Wi + 1) >> 1, nh_x = Wi >> 1, nl_y = (Hi + 1) >> 1, nh_y = Hi >> 1;
			if (zero_rect(dst, nl_x, 0, Wd - nl_x, Ho) || zero_rect(dst, Wd + nh_x, 0, (Wo - Wd) - nh_x, Ho) ||
				zero_rect(dst, 0, nl_y, Wo, Hd - nl_y) || zero_rect(dst, 0, Hd + nh_y, Wo, (Ho - Hd) - nh_y))
				return 1;
		}
	}
	return side_join();
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
	if (ensure_ll(ge, batch))
		return 1;
	const bool cols_first = (w == kCdf53I || w == kCdf97I); // the int inverses undo columns first

	// reconstruction level j consumes the subbands of size ceil(.,j) and produces the
	// band of size ceil(.,j-1); it is fused when that PRODUCED frame is dense and >= 2
	auto fused_ok = [&](int j) { return level_fused_ok(ge, j - 1); };

	Img cur = src;          // image holding the not-yet-consumed subbands
	long cur_bstride = src_bstride;
	int ll_in = -1;         // -1: LL band is in `cur`; else scratch index
	bool copied = false;
	// In place with every level fused: the detail subbands of level 1 have to be moved
	// aside before the final level overwrites them.  Start that copy now, on the side
	// stream, so that it overlaps the deeper (small) levels.
	bool aside_early = false;
	if (src.p == dst.p && J >= 2 && batch == 1) {
		bool all_fused = true;
		for (int j = 1; j <= J; j++)
			all_fused = all_fused && fused_ok(j);
		if (all_fused) {
			const int Ws = ge.Wo(1), Hs = ge.Ho(1), Wo = ge.Wo(0), Ho = ge.Ho(0);
			if (grow(&g.stage_img, &g.stage_bytes, (size_t)dst.sx * Ho))
				return 1;
			Img st{(char *)g.stage_img, dst.sx};
			if (side_fork())
				return 1;
			if (copy_rect_on(g.side, st, Ws, 0, src, Ws, 0, Wo - Ws, Ho) || copy_rect_on(g.side, st, 0, Hs, src, 0, Hs, Ws, Ho - Hs))
				return 1;
			aside_early = true;
		}
	}
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
			a.h_pitch = cur.sx / 4;
			a.h_bstride = cur_bstride / 4;
			if (ll_in < 0) {
				a.in_ll = cur.p;
				a.ll_pitch = cur.sx / 4;
				a.ll_bstride = cur_bstride / 4;
			} else {
				a.in_ll = g.ll[ll_in];
				a.ll_pitch = ll_pitch_elems(Ws);
				a.ll_bstride = a.ll_pitch * Hs;
			}
			const bool last = (j == 1);
			int ll_out = -1;
			if (last) {
				a.out = dst.p;
				a.out_pitch = dst.sx / 4;
				a.out_bstride = dst_bstride / 4;
				if (cur.p == dst.p) {
					// in place: the final level would overwrite subbands it still reads;
					// move them (right half + bottom-left, and LL if it is still there) aside
					if (batch != 1)
						return fail("in-place batches are not supported; use distinct src and dst");
					if (grow(&g.stage_img, &g.stage_bytes, (size_t)dst.sx * Ho))
						return 1;
					Img st{(char *)g.stage_img, dst.sx};
					if (aside_early) {
						if (side_join()) // the copy started before the deeper levels
							return 1;
					} else {
						if (copy_rect(st, Ws, 0, cur, Ws, 0, Wo - Ws, Ho) || copy_rect(st, 0, Hs, cur, 0, Hs, Ws, Ho - Hs))
							return 1;
						if (ll_in < 0 && copy_rect(st, 0, 0, cur, 0, 0, Ws, Hs))
							return 1;
					}
					a.in_h = st.p;
					a.h_bstride = 0;
					if (ll_in < 0)
						a.in_ll = st.p;
				}
			} else {
				ll_out = j & 1; // band of level m = j-1 lives in scratch (m-1)&1, as in the forward driver
				a.out = g.ll[ll_out];
				a.out_pitch = ll_pitch_elems(Wo);
				a.out_bstride = a.out_pitch * Ho;
			}
			prof_before(j - 1);
			hipError_t e = launch_inv_level((g.fma && w == kCdf97S) ? kCdf97SFma : w, a, g.tune, g.stream);
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
			Img s{(char *)g.ll[ll_in], ll_pitch_elems(Ws) * 4};
			if (copy_rect(dst, 0, 0, s, 0, 0, Ws, Hs))
				return 1;
			ll_in = -1;
		}
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
	return side_join();
}

int check_inited()
{
	if (!g.inited && dwt_hip_init())
		return 1;
	return 0;
}

} // namespace

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
	if (env)
		dev = atoi(env) % n;
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

void dwt_hip_finish(void)
{
	if (!g.inited)
		return;
	hipStreamSynchronize(g.stream);
	void **bufs[] = {&g.stage_img, &g.ll[0], &g.ll[1], &g.host_a, &g.host_b};
	for (void **b : bufs) {
		if (*b)
			hipFree(*b);
		*b = nullptr;
	}
	g.stage_bytes = g.ll_bytes[0] = g.ll_bytes[1] = g.host_a_bytes = g.host_b_bytes = 0;
	if (g.pin)
		hipHostFree(g.pin);
	g.pin = nullptr;
	g.pin_bytes = 0;
	for (auto &l : g.lanes) {
		void **lb[] = {&l.ll[0], &l.ll[1], &l.stage_img};
		for (void **p : lb) {
			if (*p)
				hipFree(*p);
			*p = nullptr;
		}
		l.ll_bytes[0] = l.ll_bytes[1] = l.stage_bytes = 0;
	}
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

void dwt_hip_sync(void)
{
	if (g.inited)
		hipStreamSynchronize(g.stream);
}

int dwt_hip_set_option(const char *name, int value)
{
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
	else if (!strcmp(name, "wave_horiz"))
		g.tune.wave_horiz = value;
	else if (!strcmp(name, "ring"))
		g.tune.ring = value;
	else if (!strcmp(name, "nt"))
		g.tune.nt = value;
	else if (!strcmp(name, "nt_inv"))
		g.tune.nt_inv = value;
	else if (!strcmp(name, "ring_inv"))
		g.tune.ring_inv = value;
	else if (!strcmp(name, "wave_horiz_inv"))
		g.tune.wave_horiz_inv = value;
	else if (!strcmp(name, "fma"))
		g.fma = value;
	else if (!strcmp(name, "fuse2"))
		g.tune.fuse2 = value;
	else if (!strcmp(name, "vol_cpt"))
		g.vol.cpt = value;
	else if (!strcmp(name, "vol_tile_pairs"))
		g.vol.tile_pairs = value;
	else if (!strcmp(name, "vol_nt"))
		g.vol.nt = value;
	else if (!strcmp(name, "vol_fused"))
		g.vol.fused = value;
	else if (!strcmp(name, "vol_swizzle"))
		g.vol.swizzle = value;
	else if (!strcmp(name, "pipeline"))
		g.pipeline = value < 2 ? 0 : (value > Ctx::kMaxLanes ? Ctx::kMaxLanes : value);
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
	if (!strcmp(name, "wave_horiz"))
		return g.tune.wave_horiz;
	if (!strcmp(name, "ring"))
		return g.tune.ring;
	if (!strcmp(name, "nt"))
		return g.tune.nt;
	if (!strcmp(name, "nt_inv"))
		return g.tune.nt_inv;
	if (!strcmp(name, "pipeline"))
		return g.pipeline;
	if (!strcmp(name, "fma"))
		return g.fma;
	if (!strcmp(name, "fuse2"))
		return g.tune.fuse2;
	if (!strcmp(name, "vol_cpt"))
		return g.vol.cpt;
	if (!strcmp(name, "vol_tile_pairs"))
		return g.vol.tile_pairs;
	if (!strcmp(name, "vol_nt"))
		return g.vol.nt;
	if (!strcmp(name, "vol_fused"))
		return g.vol.fused;
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
	if (p)
		hipFree(p);
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

	if (dev_dst) {
		if (stride_y != es || (stride_x % es) || stride_x < sox * es)
			return fail("device images need stride_y == %d and stride_x a multiple of it >= width*%d (got %d, %d)", es, es, stride_x, stride_y);
		Img s{(char *)src, stride_x, es}, d{(char *)dst, stride_x, es};
		return inverse ? inverse2d(w, s, d, ge, *j, decompose_one, zero_padding, 1, 0, 0)
		               : forward2d(w, s, d, ge, j, decompose_one, zero_padding, 1, 0, 0);
	}

	// ---- host pointers: stage the whole outer frame through HBM ----
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
	// that every element the reference leaves untouched keeps its value.
	if (s2) {
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

// ---------------------------------------------------------------------------------
// Interleaved (in-place lifting) layout: libdwt.h dwt_cdf97_2f_inplace_s (src/libdwt.c:12926),
// dwt_cdf97_2i_inplace_s (:17474), dwt_cdf53_2f_inplace_s (:16553), dwt_cdf53_2i_inplace_s
// (:17886) and dwt-simple.h fdwt2_cdf97_* / fdwt2_cdf53_* (src/dwt-simple.c:2224, :2356).
// Level j transforms the stride-2^j lattice of the image in place.  On the device every
// level runs on a DENSE image instead: the forward sweep of level j writes its low-pass
// samples a second time, densely, as the input of level j+1, and the results of the levels
// >= 1 are scattered into the lattice afterwards (deepest last); the inverse gathers the
// lattices first.  Rows are finished before columns at every level; the 9/7 entries of the
// reference interleave the two in phases (prolog / core / epilog), which changes fp32
// rounding in the 8-sample border bands only (tests/test_oracle_interleaved.py).
// ---------------------------------------------------------------------------------
struct IlLevel {
	float *a = nullptr, *b = nullptr; // dense input / output of the level (levels >= 1)
	long pitch = 0;                   // elements
	int lx = 0, ly = 0;
};

// The reference's 9/7 in-place drivers and fdwt2_* cut every line transform into phases --
// SHORT (whole line, lines shorter than `min_phased`), PROLOG, CORE, EPILOG -- and run each
// phase over all rows, then all columns, before the next (src/dwt-simple.c:2266-2350,
// src/libdwt.c:12970-13480, 17517-17594).  This is that order, phase by phase, for
// dwt_util_set_accel(1): bit-identical to the reference, eight passes per level instead of one.
// Index ranges per lifting step: prolog src/dwt-simple.c:580-611, core :981-1029, epilog
// :1469-1528, short :424-510; inverse src/libdwt.c:9591-9668, 7661-7740, 9929-10010.
static void il_phase_ranges(int N, int K, bool inverse, int phase, IlPhase *ph)
{
	// phase: 0 short, 1 prolog, 2 core, 3 epilog
	for (int s = 0; s < 4; s++) {
		ph->lo[s] = 1;
		ph->hi[s] = 0;
	}
	if (phase == 0) {
		for (int s = 0; s < K; s++) {
			ph->lo[s] = 0;
			ph->hi[s] = N - 1;
		}
		ph->sc_lo = 0;
		ph->sc_hi = N - 1;
	} else if (!inverse) {
		const int M = (((N - 1) & ~1) - K) / 2; // core pairs, counted from index 1
		for (int s = 0; s < K; s++) {
			if (phase == 1) { ph->lo[s] = 0; ph->hi[s] = K - 1 - s; }
			else if (phase == 2) { ph->lo[s] = K + 1 - s; ph->hi[s] = K - 1 - s + 2 * M; }
			else { ph->lo[s] = K + 1 - s + 2 * M; ph->hi[s] = N - 1; }
		}
		if (phase == 1) { ph->sc_lo = 0; ph->sc_hi = 0; }
		else if (phase == 2) { ph->sc_lo = 1; ph->sc_hi = 2 * M; }
		else { ph->sc_lo = 2 * M + 1; ph->sc_hi = N - 1; }
	} else {
		const int M = ((N & ~1) - K) / 2; // core pairs, counted from index 0
		for (int s = 0; s < K; s++) {
			if (phase == 1) { ph->lo[s] = 0; ph->hi[s] = K - 2 - s; }
			else if (phase == 2) { ph->lo[s] = K - s; ph->hi[s] = K - 2 - s + 2 * M; }
			else { ph->lo[s] = K - s + 2 * M; ph->hi[s] = N - 1; }
		}
		if (phase == 1) { ph->sc_lo = 0; ph->sc_hi = K - 1; }
		else if (phase == 2) { ph->sc_lo = K; ph->sc_hi = K - 1 + 2 * M; }
		else { ph->sc_lo = K + 2 * M; ph->sc_hi = N - 1; }
	}
}

static bool il_is_phased(Wavelet w) { return w == kCdf97S || w == kCdf97SFma || w == kCdf53SNew; }

static int il_level_phased(Wavelet w, bool inverse, Img in, Img out, int lx, int ly)
{
	if (in.sx != out.sx)
		return fail("interleaved phased level: pitches differ");
	const int K = w == kCdf53SNew ? 2 : 4;
	const int min_phased = w == kCdf53SNew ? 3 : (inverse ? 4 : 5);
	struct Pass { bool rows; int phase; };
	Pass seq[8];
	int n = 0;
	for (int phase = 0; phase < 4; phase++) {
		if (lx > 1 && (phase == 0) == (lx < min_phased))
			seq[n++] = {true, phase};
		if (ly > 1 && (phase == 0) == (ly < min_phased))
			seq[n++] = {false, phase};
	}
	if (n == 0)
		return copy_rect(out, 0, 0, in, 0, 0, lx, ly);
	if (grow(&g.host_b, &g.host_b_bytes, (size_t)in.sx * ly))
		return 1;
	Img tmp{(char *)g.host_b, in.sx, 4};
	// ping-pong so that the last pass writes `out`
	Img cur = in;
	for (int i = 0; i < n; i++) {
		const Img nxt = ((n - 1 - i) % 2 == 0) ? out : tmp;
		IlPhase ph;
		const int N = seq[i].rows ? lx : ly;
		il_phase_ranges(N, K, inverse, seq[i].phase, &ph);
		hipError_t e = launch_il_phase(w == kCdf97SFma ? kCdf97S : w, inverse, cur.p, nxt.p, seq[i].rows ? cur.sx : 4, seq[i].rows ? 4 : cur.sx,
			seq[i].rows ? ly : lx, N, !seq[i].rows, ph, g.stream);
		if (e != hipSuccess)
			return fail("interleaved phase launch failed: %s", hipGetErrorString(e));
		cur = nxt;
	}
	return 0;
}

// one level on dense images with a common pitch: rows completely, then columns
static int il_level(Wavelet w, bool inverse, bool scale_single, Img in, Img out, int lx, int ly, float *ll, long ll_pitch,
	const Img *even_rows = nullptr)
{
	const bool fused = !g.force_generic && lx >= 2 && ly >= 2 && (((uintptr_t)in.p | (uintptr_t)out.p) % 4 == 0);
	if (even_rows && !fused)
		return fail("internal: split rows need the fused sweep");
	if (fused) {
		hipError_t e;
		if (!inverse) {
			FwdLevelArgs a;
			a.in = in.p; a.in_pitch = in.sx / 4; a.in_bstride = 0;
			a.out_ll = ll; a.ll_pitch = ll_pitch; a.ll_bstride = 0;
			a.out_h = out.p; a.h_pitch = out.sx / 4; a.h_bstride = 0;
			a.W = lx; a.H = ly; a.batch = 1; a.interleaved = 1; a.il_ll = ll != nullptr;
			e = launch_fwd_level(w, a, g.tune, g.stream);
		} else {
			InvLevelArgs a;
			// the even rows may live in a buffer of their own (packed), see interleaved2d
			a.in_ll = even_rows ? even_rows->p : in.p; a.ll_pitch = even_rows ? even_rows->sx / 4 : in.sx / 4 * 2; a.ll_bstride = 0;
			a.in_h = in.p + in.sx; a.h_pitch = in.sx / 4 * 2; a.h_bstride = 0;
			a.out = out.p; a.out_pitch = out.sx / 4; a.out_bstride = 0;
			a.W = lx; a.H = ly; a.batch = 1; a.interleaved = 1;
			e = launch_inv_level(w == kCdf53SNew ? kCdf53S : w, a, g.tune, g.stream);
		}
		if (e != hipSuccess)
			return fail("interleaved sweep launch failed: %s", hipGetErrorString(e));
		return 0;
	}
	// generic.  The phase-ordered entries reproduce the reference's order exactly when the
	// generic path was asked for (accel 1); tiny levels of the fused path and the 5/3 _inplace_
	// pair (rows, then columns in the reference too) take two exact line passes.
	if (g.force_generic && il_is_phased(w) && !scale_single) {
		if (il_level_phased(w, inverse, in, out, lx, ly))
			return 1;
		if (ll && !inverse) {
			hipError_t e = launch_lattice_copy((const float *)out.p, 2, out.sx / 4 * 2, 0, ll, 1, ll_pitch, 0, (lx + 1) / 2, (ly + 1) / 2, 1, g.stream);
			if (e != hipSuccess)
				return fail("lattice gather failed: %s", hipGetErrorString(e));
		}
		return 0;
	}
	if (in.sx != out.sx)
		return fail("interleaved generic level: pitches differ");
	if (grow(&g.host_b, &g.host_b_bytes, (size_t)in.sx * ly))
		return 1;
	Img tmp{(char *)g.host_b, in.sx, 4};
	auto pass = [&](bool rows, Img from, Img to) -> int {
		const int N = rows ? lx : ly, lines = rows ? ly : lx;
		if (N == 1 && !scale_single)
			return copy_rect(to, 0, 0, from, 0, 0, lx, ly);
		hipError_t e = launch_line_pass(w, inverse, from.p, to.p, rows ? from.sx : 4, rows ? 4 : from.sx, lines, N, -1, !rows, g.stream);
		if (e != hipSuccess)
			return fail("interleaved line pass launch failed: %s", hipGetErrorString(e));
		return 0;
	};
	if (pass(true, in, tmp) || pass(false, tmp, out))
		return 1;
	if (ll && !inverse) {
		hipError_t e = launch_lattice_copy((const float *)out.p, 2, out.sx / 4 * 2, 0, ll, 1, ll_pitch, 0, (lx + 1) / 2, (ly + 1) / 2, 1, g.stream);
		if (e != hipSuccess)
			return fail("lattice gather failed: %s", hipGetErrorString(e));
	}
	return 0;
}

static int interleaved2d(Wavelet w, bool inverse, bool scale_single, Img src, Img dst, int sox, int soy, int six, int siy,
	int *jp, int decompose_one)
{
	const int j_limit = ceil_log2(decompose_one ? (sox > soy ? sox : soy) : (sox < soy ? sox : soy));
	int J = *jp;
	if (J < 0 || J > j_limit)
		J = j_limit;
	if (!inverse)
		*jp = J;
	if (side_join())
		return 1;
	const bool alias = src.p == dst.p;
	// everything outside the transformed region keeps the caller's values
	const bool sparse = six < sox || siy < soy;
	if (!alias && (J == 0 || sparse) && copy_rect(dst, 0, 0, src, 0, 0, sox, soy))
		return 1;
	if (J == 0 || six < 1 || siy < 1)
		return 0;
	constexpr int kMax = 32;
	IlLevel L[kMax];
	size_t pool = 0;
	for (int j = 0; j < J; j++) {
		L[j].lx = ceil_div_pow2(six, j);
		L[j].ly = ceil_div_pow2(siy, j);
		L[j].pitch = align_up(L[j].lx, 4);
		if (j >= 1)
			pool += (size_t)L[j].pitch * L[j].ly;
	}
	if (J > 1) {
		if (grow(&g.ll[0], &g.ll_bytes[0], pool * 4) || grow(&g.ll[1], &g.ll_bytes[1], pool * 4))
			return 1;
		float *pa = (float *)g.ll[0], *pb = (float *)g.ll[1];
		for (int j = 1; j < J; j++) {
			L[j].a = pa; L[j].b = pb;
			pa += (size_t)L[j].pitch * L[j].ly;
			pb += (size_t)L[j].pitch * L[j].ly;
		}
	}
	auto dense = [&](float *p, const IlLevel &l) { return Img{(char *)p, l.pitch * 4, 4}; };
	auto scatter = [&](const float *from, long from_pitch, char *to, long to_pitch_bytes, long step, const IlLevel &l) -> int {
		// dense level -> lattice of stride `step` (elements) of an image
		hipError_t e = launch_lattice_copy(from, 1, from_pitch, 0, (float *)to, step, to_pitch_bytes / 4 * step, 0, l.lx, l.ly, 1, g.stream);
		if (e != hipSuccess)
			return fail("lattice scatter failed: %s", hipGetErrorString(e));
		return 0;
	};
	// level 0 works on the caller's image; in place it detours through the staging image
	Img stage{nullptr, dst.sx, 4};
	if (alias || inverse) {
		if (grow(&g.stage_img, &g.stage_bytes, (size_t)dst.sx * siy))
			return 1;
		stage.p = (char *)g.stage_img;
	}

	auto pyramid = [&](bool results, int levels) {
		IlPyramid py;
		py.J = levels;
		for (int j = 1; j < levels; j++) {
			py.p[j] = results ? L[j].b : L[j].a;
			py.pitch[j] = L[j].pitch;
		}
		return py;
	};
	// rows 1, 3, 5, ... of the transformed region from one image to another
	auto copy_odd_rows = [&](Img to, Img from) -> int {
		return copy_rect(Img{to.p + to.sx, to.sx * 2, 4}, 0, 0, Img{from.p + from.sx, from.sx * 2, 4}, 0, 0, six, siy / 2);
	};

	if (!inverse) {
		for (int j = 0; j < J; j++) {
			const Img in = j == 0 ? src : dense(L[j].a, L[j]);
			const Img out = j == 0 ? (alias ? stage : dst) : dense(L[j].b, L[j]);
			float *ll = j + 1 < J ? L[j + 1].a : nullptr;
			if (il_level(w, false, scale_single, in, out, L[j].lx, L[j].ly, ll, j + 1 < J ? L[j + 1].pitch : 0))
				return 1;
		}
		if (J == 1)
			return alias ? copy_rect(dst, 0, 0, stage, 0, 0, six, siy) : 0;
		// the even rows receive the samples of the levels >= 1 in ONE pass (in place that pass
		// also brings them back from the staging image; the odd rows are final after level 0)
		if (alias && copy_odd_rows(dst, stage))
			return 1;
		const Img base = alias ? stage : dst;
		hipError_t e = launch_il_compose((const float *)base.p, base.sx / 4, (float *)dst.p, dst.sx / 4, six, siy, pyramid(true, J), g.stream);
		if (e != hipSuccess)
			return fail("interleaved compose failed: %s", hipGetErrorString(e));
		return 0;
	}
	// inverse: the coefficients are read from the source image (never modified before the last
	// sweep has read it, so out of place needs no copy)
	const Img cin = src;
	if (J == 1) {
		if (!alias)
			return il_level(w, true, scale_single, cin, dst, L[0].lx, L[0].ly, nullptr, 0);
		if (il_level(w, true, scale_single, dst, stage, L[0].lx, L[0].ly, nullptr, 0))
			return 1;
		return copy_rect(dst, 0, 0, stage, 0, 0, six, siy);
	}
	hipError_t e = launch_il_decompose((const float *)cin.p, cin.sx / 4, six, siy, pyramid(false, J), g.stream);
	if (e != hipSuccess)
		return fail("interleaved decompose failed: %s", hipGetErrorString(e));
	for (int j = J - 1; j >= 1; j--) {
		if (il_level(w, true, scale_single, dense(L[j].a, L[j]), dense(L[j].b, L[j]), L[j].lx, L[j].ly, nullptr, 0))
			return 1;
		// the reconstructed low-pass band is the even-even lattice of the level above
		if (j >= 2 && scatter(L[j].b, L[j].pitch, (char *)L[j - 1].a, L[j - 1].pitch * 4, 2, L[j]))
			return 1;
	}
	// level 0.  Out of place (fused sweep): odd rows straight from the coefficient image, even
	// rows from a packed copy that carries the reconstructed LL band (one compose pass).  In
	// place the sweep must not read what it overwrites: its whole input is built in the staging
	// image (odd rows copied, even rows composed) and the sweep writes the caller's image.
	const bool split = !alias && !g.force_generic && L[0].lx >= 2 && L[0].ly >= 2;
	if (split) {
		const Img even{stage.p, stage.sx, 4}; // (siy+1)/2 packed rows
		e = launch_il_compose((const float *)cin.p, cin.sx / 4, (float *)even.p, even.sx / 4, six, siy, pyramid(true, 2), g.stream, true);
		if (e != hipSuccess)
			return fail("interleaved compose failed: %s", hipGetErrorString(e));
		return il_level(w, true, scale_single, cin, dst, L[0].lx, L[0].ly, nullptr, 0, &even);
	}
	if (copy_odd_rows(stage, cin))
		return 1;
	e = launch_il_compose((const float *)cin.p, cin.sx / 4, (float *)stage.p, stage.sx / 4, six, siy, pyramid(true, 2), g.stream);
	if (e != hipSuccess)
		return fail("interleaved compose failed: %s", hipGetErrorString(e));
	return il_level(w, true, scale_single, stage, dst, L[0].lx, L[0].ly, nullptr, 0);
}

int dwt_hip_transform2d_interleaved(int wavelet, int inverse, int flavour, const void *src, void *dst, int stride_x, int stride_y,
	int sox, int soy, int six, int siy, int *j, int decompose_one)
{
	if (check_inited())
		return 1;
	if (wavelet != kCdf97S && wavelet != kCdf53S)
		return fail("the interleaved layout takes the float wavelets (CDF 9/7, CDF 5/3), not %d", wavelet);
	if (flavour != 0 && flavour != 1)
		return fail("unknown flavour %d", flavour);
	if (flavour == 1 && inverse)
		return fail("dwt-simple.h has forward transforms only; use flavour 0 for the inverse");
	if (!src || !dst || !j)
		return fail("null pointer argument");
	if (sox <= 0 || soy <= 0 || six < 0 || siy < 0 || six > sox || siy > soy)
		return fail("bad sizes: outer %dx%d inner %dx%d", sox, soy, six, siy);
	g_elems_are_32bit = true;
	// single-sample lines: the 9/7 drivers and fdwt2_* leave them (guards `size > 1`,
	// libdwt.c:12978, dwt-simple.c:2266), the 5/3 _inplace_ drivers scale them (:11041, :11840)
	const bool scale_single = wavelet == kCdf53S && flavour == 0;
	const Wavelet w = wavelet == kCdf97S ? ((g.fma && !inverse) ? kCdf97SFma : kCdf97S) : (flavour == 1 ? kCdf53SNew : kCdf53S);
	const bool dev_src = dwt_hip_is_device_pointer(src), dev_dst = dwt_hip_is_device_pointer(dst);
	if (dev_src != dev_dst)
		return fail("src and dst must both be host or both be device pointers");
	if (dev_dst) {
		if (stride_y != 4 || (stride_x % 4) || stride_x < sox * 4)
			return fail("device images need stride_y == 4 and stride_x a multiple of it >= width*4 (got %d, %d)", stride_x, stride_y);
		return interleaved2d(w, inverse != 0, scale_single, Img{(char *)src, stride_x, 4}, Img{(char *)dst, stride_x, 4}, sox, soy, six, siy, j, decompose_one);
	}
	// host pointers: stage the outer frame through HBM (any byte strides)
	const long pitch = align_up((long)sox * 4, 256);
	if (grow(&g.host_a, &g.host_a_bytes, (size_t)pitch * soy))
		return 1;
	if (host_upload(src, stride_x, stride_y, 4, sox, soy, g.host_a, pitch))
		return 1;
	Img A{(char *)g.host_a, pitch, 4};
	if (interleaved2d(w, inverse != 0, scale_single, A, A, sox, soy, six, siy, j, decompose_one))
		return 1;
	return host_download(dst, stride_x, stride_y, 4, sox, soy, g.host_a, pitch);
}

int dwt_hip_transform2d_batch(int wavelet, int inverse, const void *src, void *dst, size_t batch_stride, int batch,
	int stride_x, int size_x, int size_y, int *j)
{
	if (check_inited())
		return 1;
	if (wavelet < 0 || wavelet > 5 || elem_size((Wavelet)wavelet) != 4)
		return fail("unknown wavelet %d (batches take the 32-bit wavelets)", wavelet);
	g_elems_are_32bit = true;
	if (!src || !dst || !j || batch < 1 || batch > 65535)
		return fail("bad argument (batch must be 1..65535)");
	if (!dwt_hip_is_device_pointer(src) || !dwt_hip_is_device_pointer(dst))
		return fail("batched transforms take device pointers");
	if ((stride_x & 3) || stride_x < size_x * 4 || (batch_stride & 3) || batch_stride < (size_t)stride_x * size_y)
		return fail("bad strides");
	if (batch > 1 && src == dst)
		return fail("in-place batches are not supported; use distinct src and dst");
	const Geom ge{size_x, size_y, size_x, size_y};
	Img s{(char *)src, stride_x}, d{(char *)dst, stride_x};
	if (g.pipeline >= 2 && batch >= 2) {
		// per-image pipelines on internal streams, forked from and joined to the caller's stream
		const int nl = g.pipeline < batch ? g.pipeline : batch;
		if (!g.fork)
			HIP_TRY(hipEventCreateWithFlags(&g.fork, hipEventDisableTiming));
		for (int l = 0; l < nl; l++) {
			if (!g.lanes[l].stream) {
				HIP_TRY(hipStreamCreateWithFlags(&g.lanes[l].stream, hipStreamNonBlocking));
				HIP_TRY(hipEventCreateWithFlags(&g.lanes[l].done, hipEventDisableTiming));
			}
		}
		HIP_TRY(hipEventRecord(g.fork, g.stream));
		for (int l = 0; l < nl; l++)
			HIP_TRY(hipStreamWaitEvent(g.lanes[l].stream, g.fork, 0));
		int rc = 0;
		const int j_in = *j;
		for (int k = 0; k < batch && !rc; k++) {
			Ctx::Lane &lane = g.lanes[k % nl];
			Img sk{s.p + (size_t)k * batch_stride, s.sx}, dk{d.p + (size_t)k * batch_stride, d.sx};
			int jk = j_in;
			swap_lane(lane);
			rc = inverse ? inverse2d((Wavelet)wavelet, sk, dk, ge, jk, 0, 0, 1, 0, 0)
			             : forward2d((Wavelet)wavelet, sk, dk, ge, &jk, 0, 0, 1, 0, 0);
			swap_lane(lane);
			if (!inverse)
				*j = jk;
		}
		for (int l = 0; l < nl; l++) {
			HIP_TRY(hipEventRecord(g.lanes[l].done, g.lanes[l].stream));
			HIP_TRY(hipStreamWaitEvent(g.stream, g.lanes[l].done, 0));
		}
		return rc;
	}
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
	static unsigned *counter = nullptr;
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

// Forward 3-D transform, OUT OF PLACE: the layout and arithmetic of cdf97_3f_op_sep_horizontal_s
// (src/volume-dwt.c:727-785: copy each x line to the destination, then lift x, y, z there), the
// entry the reference's own 3-D perf test drives (volume_perftest_fwd97op_s, src/volume.c).
// Level j reads a dense volume and writes a dense volume, so each level is ONE fused pass
// (k_vol_fwd_fused) where that kernel applies and the two-pass path (xy sweep, z sweep through
// the scratch volume) elsewhere; the even-even-even samples go to the next level densely, and the
// results of the levels >= 1 are scattered into their lattices at the end (deepest first).
int dwt_hip_transform3d_op(const void *src, void *dst, size_t stride_y, size_t stride_z, int nx, int ny, int nz, int levels)
{
	if (check_inited())
		return 1;
	if (!src || !dst || !dwt_hip_is_device_pointer(src) || !dwt_hip_is_device_pointer(dst))
		return fail("dwt_hip_transform3d_op takes device pointers");
	if (src == dst)
		return fail("dwt_hip_transform3d_op is out of place; use dwt_hip_transform3d for in-place volumes");
	if ((stride_y & 3) || (stride_z & 3) || stride_y < (size_t)nx * 4 || stride_z < stride_y * (size_t)ny)
		return fail("bad volume strides");
	constexpr int kMaxLevels = 24;
	if (levels > kMaxLevels)
		return fail("too many levels");
	if (levels >= 1 && (ceil_div_pow2(nx, levels - 1) < 2 || ceil_div_pow2(ny, levels - 1) < 2 || ceil_div_pow2(nz, levels - 1) < 2))
		return fail("volume %dx%dx%d is too small for %d levels", nx, ny, nz, levels);
	const long vsy = (long)stride_y / 4, vsz = (long)stride_z / 4;
	if (levels < 1) {
		// no levels: the reference's copy stage alone
		hipError_t e = launch_lattice_copy((const float *)src, 1, vsy, vsz, (float *)dst, 1, vsy, vsz, nx, ny, nz, g.stream);
		return e == hipSuccess ? 0 : fail("volume copy failed: %s", hipGetErrorString(e));
	}
	struct Lvl { const float *in; float *out; long sy, sz; int lx, ly, lz; } L[kMaxLevels];
	L[0] = {(const float *)src, (float *)dst, vsy, vsz, nx, ny, nz};
	size_t pool = 0;
	for (int j = 1; j < levels; j++) {
		L[j].lx = ceil_div_pow2(nx, j); L[j].ly = ceil_div_pow2(ny, j); L[j].lz = ceil_div_pow2(nz, j);
		L[j].sy = align_up(L[j].lx, 4);
		L[j].sz = L[j].sy * L[j].ly;
		pool += (size_t)L[j].sz * L[j].lz;
	}
	if (levels > 1) {
		if (grow(&g.host_a, &g.host_a_bytes, pool * 4) || grow(&g.host_b, &g.host_b_bytes, pool * 4))
			return 1;
		float *pa = (float *)g.host_a, *pb = (float *)g.host_b;
		for (int j = 1; j < levels; j++) {
			L[j].in = pa; L[j].out = pb;
			pa += (size_t)L[j].sz * L[j].lz;
			pb += (size_t)L[j].sz * L[j].lz;
		}
	}
	for (int j = 0; j < levels; j++)
		if (L[j].lz > 65535 || L[j].ly > 65535)
			return fail("volume too large for the launch grid");
	float *S = nullptr;
	long s_sy = 0, s_sz = 0;
	for (int j = 0; j < levels; j++) {
		const Lvl &b = L[j];
		float *lll = j + 1 < levels ? (float *)L[j + 1].in : nullptr;
		const long lsy = j + 1 < levels ? L[j + 1].sy : 0, lsz = j + 1 < levels ? L[j + 1].sz : 0;
		VolFusedArgs fa{b.in, b.sy, b.sz, b.out, b.sy, b.sz, lll, lsy, lsz, b.lx, b.ly, b.lz};
		const bool can_fuse = fa.in != fa.out && fa.nx >= 2 && fa.ny >= 2 && fa.nz >= 2;
		if (!g.force_generic && ((g.vol.fused == 1 && vol_fused_applies(fa)) || (g.vol.fused >= 2 && can_fuse))) {
			prof_before(j);
			hipError_t e = launch_vol_fwd_fused(fa, g.vol, g.stream);
			prof_after(j);
			if (e != hipSuccess)
				return fail("fused 3-D level launch failed: %s", hipGetErrorString(e));
			continue;
		}
		// two passes through the scratch volume
		if (!S) {
			s_sy = align_up(nx, 4);
			s_sz = s_sy * ny;
			if (grow(&g.stage_img, &g.stage_bytes, (size_t)s_sz * nz * 4))
				return 1;
			S = (float *)g.stage_img;
		}
		FwdLevelArgs a;
		a.in = b.in; a.in_pitch = b.sy; a.in_bstride = b.sz;
		a.out_ll = S; a.ll_pitch = s_sy; a.ll_bstride = s_sz;
		a.out_h = S; a.h_pitch = s_sy; a.h_bstride = s_sz;
		a.W = b.lx; a.H = b.ly; a.batch = b.lz; a.interleaved = 1;
		hipError_t e = launch_fwd_level(kCdf97S, a, g.tune, g.stream);
		if (e != hipSuccess)
			return fail("3-D xy pass launch failed: %s", hipGetErrorString(e));
		e = launch_vol_z(false, S, s_sy, s_sz, b.out, b.sy, b.sz, b.lx, b.ly, b.lz, g.vol, g.stream, lll, lsy, lsz);
		if (e != hipSuccess)
			return fail("3-D z pass launch failed: %s", hipGetErrorString(e));
	}
	for (int j = levels - 1; j >= 1; j--) {
		const Lvl &c = L[j], &par = L[j - 1];
		hipError_t e = launch_lattice_copy(c.out, 1, c.sy, c.sz, par.out, 2, par.sy * 2, par.sz * 2, c.lx, c.ly, c.lz, g.stream);
		if (e != hipSuccess)
			return fail("lattice scatter failed: %s", hipGetErrorString(e));
	}
	return 0;
}

int dwt_hip_transform3d(int inverse, void *vol, size_t stride_y, size_t stride_z, int nx, int ny, int nz, int levels)
{
	if (check_inited())
		return 1;
	if (!vol || !dwt_hip_is_device_pointer(vol))
		return fail("dwt_hip_transform3d takes a device pointer");
	if ((stride_y & 3) || (stride_z & 3) || stride_y < (size_t)nx * 4 || stride_z < stride_y * (size_t)ny)
		return fail("bad volume strides");
	if (levels < 1)
		return 0;
	// every level needs at least 2 samples per axis (the reference asserts >= 5, dwt-simple.c:2172)
	if (ceil_div_pow2(nx, levels - 1) < 2 || ceil_div_pow2(ny, levels - 1) < 2 || ceil_div_pow2(nz, levels - 1) < 2)
		return fail("volume %dx%dx%d is too small for %d levels", nx, ny, nz, levels);
	// scratch: S (pass-to-pass buffer) and, for levels >= 1, dense copies P[j] of the
	// level-j lattice (even-even-even samples of level j-1), all carved from one buffer
	const long s_sy = align_up(nx, 4), s_sz = s_sy * ny;
	if (grow(&g.stage_img, &g.stage_bytes, (size_t)s_sz * nz * 4))
		return 1;
	float *S = (float *)g.stage_img;
	constexpr int kMaxLevels = 24;
	if (levels > kMaxLevels)
		return fail("too many levels");
	struct Lvl { float *p; long sy, sz; int lx, ly, lz; } L[kMaxLevels];
	L[0] = {(float *)vol, (long)stride_y / 4, (long)stride_z / 4, nx, ny, nz};
	size_t p_total = 0;
	for (int j = 1; j < levels; j++) {
		L[j].lx = ceil_div_pow2(nx, j); L[j].ly = ceil_div_pow2(ny, j); L[j].lz = ceil_div_pow2(nz, j);
		L[j].sy = align_up(L[j].lx, 4);
		L[j].sz = L[j].sy * L[j].ly;
		p_total += (size_t)L[j].sz * L[j].lz;
	}
	if (levels > 1) {
		if (grow(&g.host_a, &g.host_a_bytes, p_total * 4))
			return 1;
		float *p = (float *)g.host_a;
		for (int j = 1; j < levels; j++) {
			L[j].p = p;
			p += (size_t)L[j].sz * L[j].lz;
		}
	}
	for (int j = 0; j < levels; j++)
		if (L[j].lz > 65535 || L[j].ly > 65535)
			return fail("volume too large for the launch grid");

	auto one_level = [&](const Lvl &b, const Lvl *next) -> int {
		// x then y fused per slice, then z (src/volume-dwt.c:677-725; inverse :1115-1163)
		hipError_t e;
		if (!inverse) {
			FwdLevelArgs a;
			a.in = b.p; a.in_pitch = b.sy; a.in_bstride = b.sz;
			a.out_ll = S; a.ll_pitch = s_sy; a.ll_bstride = s_sz;
			a.out_h = S; a.h_pitch = s_sy; a.h_bstride = s_sz;
			a.W = b.lx; a.H = b.ly; a.batch = b.lz; a.interleaved = 1;
			e = launch_fwd_level(kCdf97S, a, g.tune, g.stream);
		} else {
			InvLevelArgs a;
			a.in_ll = b.p; a.ll_pitch = 2 * b.sy; a.ll_bstride = b.sz;
			a.in_h = b.p + b.sy; a.h_pitch = 2 * b.sy; a.h_bstride = b.sz;
			a.out = S; a.out_pitch = s_sy; a.out_bstride = s_sz;
			a.W = b.lx; a.H = b.ly; a.batch = b.lz; a.interleaved = 1;
			e = launch_inv_level(kCdf97S, a, g.tune, g.stream);
		}
		if (e != hipSuccess)
			return fail("3-D xy pass launch failed: %s", hipGetErrorString(e));
		// forward: the z pass also writes the next level's input densely (no lattice gather)
		e = launch_vol_z(inverse != 0, S, s_sy, s_sz, b.p, b.sy, b.sz, b.lx, b.ly, b.lz, g.vol, g.stream,
			next ? next->p : nullptr, next ? next->sy : 0, next ? next->sz : 0);
		if (e != hipSuccess)
			return fail("3-D z pass launch failed: %s", hipGetErrorString(e));
		return 0;
	};
	// level j lives on the stride-2 lattice (even-even-even samples) of level j-1
	auto lattice = [&](int j, bool pack) -> int {
		const Lvl &c = L[j], &par = L[j - 1];
		hipError_t e = pack
			? launch_lattice_copy(par.p, 2, par.sy * 2, par.sz * 2, c.p, 1, c.sy, c.sz, c.lx, c.ly, c.lz, g.stream)
			: launch_lattice_copy(c.p, 1, c.sy, c.sz, par.p, 2, par.sy * 2, par.sz * 2, c.lx, c.ly, c.lz, g.stream);
		if (e != hipSuccess)
			return fail("lattice %s failed: %s", pack ? "pack" : "unpack", hipGetErrorString(e));
		return 0;
	};

	if (!inverse) {
		for (int j = 0; j < levels; j++)
			if (one_level(L[j], j + 1 < levels ? &L[j + 1] : nullptr))
				return 1;
		for (int j = levels - 1; j >= 1; j--)
			if (lattice(j, false))
				return 1;
	} else {
		for (int j = 1; j < levels; j++)
			if (lattice(j, true))
				return 1;
		for (int j = levels - 1; j >= 0; j--) {
			if (one_level(L[j], nullptr))
				return 1;
			if (j >= 1 && lattice(j, false))
				return 1;
		}
	}
	return 0;
}

} // extern "C"
#pragma GCC visibility pop
