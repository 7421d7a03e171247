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
	g.stat_allocs++;
	*have = need;
	return 0;
}


bool skip_single(Wavelet w) { return w == kCdf97S; } // only the 9/7 drivers guard on lines > 1

// several rectangles (element coordinates) between two device images in one kernel launch
struct Rect {
	long dx, dy, sx, sy, w, h;
};
static CopyRects make_copy_rects(Img dst, Img src, const Rect *rc, int n, int policy)
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
	return r;
}

int copy_rects_on(hipStream_t st, Img dst, Img src, const Rect *rc, int n, int policy = 3)
{
	hipError_t e = launch_copy_rects(make_copy_rects(dst, src, rc, n, policy), st);
	g.stat_launches++;
	if (e != hipSuccess)
		return fail("rectangle copy launch failed: %s", hipGetErrorString(e));
	return 0;
}

// A rectangle copy that RIDES ALONG with the levels it does not depend on (one image, in place: the staged subbands of
// level 0 going back / aside): its blocks are handed out to those levels' launches as extra workgroups behind their
// tiles (FwdLevelArgs::ride).  The levels below level 0 of one image are latency-bound -- one round of waves, 9-33 us
// each, the memory system mostly idle -- and the copy is bandwidth-bound (402 MB, 62 us as a launch of its own): together
// they take what the larger of the two takes.  Whatever is left when the levels are through goes out as a plain launch.
struct RideCopy {
	CopyRects r;
	int next = 0, total = 0;
	bool on = false;
	// blocks for a level with `own_bytes` of input when `carriers_after` later launches can take some too
	int share(size_t own_bytes, int carriers_after) const
	{
		const int left = total - next;
		const int small = std::min(left, g.ride_mib * 32); // blocks of 32 KiB
		if (carriers_after <= 0)
			return left;
		if (own_bytes <= ((size_t)32 << 20))
			return small;
		return std::max(small, left - carriers_after * g.ride_mib * 32);
	}
	int flush()
	{
		if (!on || next >= total)
			return 0;
		hipError_t e = launch_copy_rects_range(r, next, total, g.stream);
		g.stat_launches++;
		next = total;
		if (e != hipSuccess)
			return fail("rectangle copy launch failed: %s", hipGetErrorString(e));
		return 0;
	}
};

int copy_rect_on(hipStream_t st, Img dst, long dx, long dy, Img src, long sx_, long sy_, long w, long h)
{
	if (w <= 0 || h <= 0)
		return 0;
	HIP_TRY(hipMemcpy2DAsync(dst.p + dy * dst.sx + dx * dst.es, dst.sx, src.p + sy_ * src.sx + sx_ * src.es, src.sx, w * dst.es, h,
		hipMemcpyDeviceToDevice, st));
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
	g.stat_launches++;
	if (e != hipSuccess)
		return fail("line pass launch failed: %s", hipGetErrorString(e));
	if (alias && copy_rect(out, 0, 0, dst, 0, 0, frame_w, frame_h))
		return 1;
	return 0;
}

void prof_before(int level)
{
	g.stat_launches++; // (every fused level launch of the 2-D drivers passes here)
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
	RideCopy ride;
	// the fused levels from j on that can carry copy blocks (one image, a sweep variant with the ride-along kernel)
	auto carriers_from = [&](int j0) {
		int n = 0;
		for (int j = j0; j < J && level_fused_ok(ge, j); j++)
			n += !dbl && sweep_ride_ok(g.tune, ge.Wo(j), ge.Ho(j), batch, false);
		return n;
	};
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
				// a copy-back that follows at once reads the staged subbands out of the 256 MiB Infinity Cache when they were
				// stored temporal (one 8192^2 image: 218.5 -> 205 us); a copy that rides along with the deeper levels is spread
				// over their launches and gains nothing from it (202 against 205 us)
				a.temporal = (g.ride_copy && carriers_from(j + 1) > 0) ? 0 : 1;
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
				apply_tile_choice(tuned_tile_pairs(w, a), &tune, false); // (nothing measured: the launcher's own rule)
			if (ride.on && ride.next < ride.total && !dbl && sweep_ride_ok(tune, Wo, Ho, batch, false)) {
				a.ride = &ride.r;
				a.ride_lo = ride.next;
				a.ride_hi = ride.next + ride.share((size_t)Wo * Ho * es, carriers_from(j + 1));
				ride.next = a.ride_hi;
			}
			a.probe_fuse1 = (DWT_PROBES && j == 0 && !dbl && w == kCdf97S && !a.temporal) ? g.tune.probe_fuse1 : 0;
			prof_before(j);
			hipError_t e = dbl ? launch_fwd_level_d(w, a, tune, g.stream)
			                   : launch_fwd_level((g.fma && w == kCdf97S) ? kCdf97SFma : w, a, tune, g.stream);
			prof_after(j);
			if (e != hipSuccess)
				return fail("forward level %d launch failed: %s", j, hipGetErrorString(e));
			if (detour) {
				// copy the staged subbands to their place: right half, bottom-left, and the LL
				// quadrant too when it was written here (in line: on a side stream beside the deeper
				// levels it measured 8-10 us slower, profiles/archive/r03_entries_summary.md)
				const Rect rc[3] = {{Wd, 0, Wd, 0, Wo - Wd, Ho}, {0, Hd, 0, Hd, Wd, Ho - Hd}, {0, 0, 0, 0, last ? Wd : 0, Hd}};
				ride.r = make_copy_rects(dst, hdst, rc, 3, 3);
				ride.total = copy_rects_plan(&ride.r);
				ride.next = 0;
				ride.on = g.ride_copy && ride.total > 0 && carriers_from(j + 1) > 0;
				if (!ride.on && copy_rects_on(g.stream, dst, hdst, rc, 3))
					return 1;
			}
			ll_in = ll_out;
			cur = dst;
			continue;
		}

		// ---- generic level: exact line semantics, in place on dst ----
		if (ride.flush()) // (the line passes borrow the staging image the copy still reads)
			return 1;
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
	return ride.flush();
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
	// In place (one image), every level fused: the final level would overwrite subbands it still reads, so they are
	// moved aside first -- a copy that depends on none of the deeper levels and rides along with them (RideCopy)
	RideCopy ride;
	if (src.p == dst.p && batch == 1 && J >= 2 && g.ride_copy && !dbl) {
		bool all = true;
		int carriers = 0;
		for (int j = J; j >= 1; j--)
			all = all && fused_ok(j);
		for (int j = J; j >= 2; j--)
			carriers += sweep_ride_ok(g.tune, ge.Wo(j - 1), ge.Ho(j - 1), batch, true);
		if (all && carriers > 0) {
			const int Ws = ge.Wo(1), Hs = ge.Ho(1), Wo = ge.Wo(0), Ho = ge.Ho(0);
			if (grow(&g.stage_img, &g.stage_bytes, (size_t)dst.sx * Ho))
				return 1;
			Img st{(char *)g.stage_img, dst.sx, es};
			const Rect rc[2] = {{Ws, 0, Ws, 0, Wo - Ws, Ho}, {0, Hs, 0, Hs, Ws, Ho - Hs}};
			ride.r = make_copy_rects(st, src, rc, 2, /* temporal both ways: the final level reads the staged subbands */ 0);
			ride.total = copy_rects_plan(&ride.r);
			ride.on = ride.total > 0;
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
					const Rect rc[3] = {{Ws, 0, Ws, 0, Wo - Ws, Ho}, {0, Hs, 0, Hs, Ws, Ho - Hs}, {0, 0, 0, 0, ll_in < 0 ? Ws : 0, Hs}};
					if (ride.on) {
						// (most of it went with the deeper levels' launches; what is left goes now)
						if (ride.flush())
							return 1;
					} else if (copy_rects_on(g.stream, st, cur, rc, 3, /* temporal both ways: the final level reads the staged subbands (233 against 237 us) */ 0))
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
				// (not for an in-place call: its staged subbands want the cache, 221 -> 224 us with both)
				a.temporal_out = g.tune.inv_ll_temporal && src.p != dst.p && (size_t)Wo * Ho * es * batch <= ((size_t)128 << 20);
			}
			SweepTuning tune = g.tune;
			if (tune.inv_pairs <= 0)
				tune.inv_pairs = src.p == dst.p ? 32 : 16; // (in place: 32 pairs, 230.7 against 234.0 us; scripts/r06/inv_call_ab.py)
			if (!dbl && tune.tile_pairs <= 0)
				apply_tile_choice(tuned_tile_pairs(w, a), &tune, true);
			if (ride.on && !last && ride.next < ride.total && sweep_ride_ok(tune, Wo, Ho, batch, true)) {
				int after = 0;
				for (int m = j - 1; m >= 2; m--)
					after += sweep_ride_ok(g.tune, ge.Wo(m - 1), ge.Ho(m - 1), batch, true);
				a.ride = &ride.r;
				a.ride_lo = ride.next;
				a.ride_hi = ride.next + ride.share((size_t)Wo * Ho * es, after);
				ride.next = a.ride_hi;
			}
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

} // namespace dwtb
