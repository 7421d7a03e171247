// dwt_interleaved.hip -- kernels of the interleaved (in-place lifting) layout beyond the sweeps:
// the exact phase-ordered line kernel (the cross-check path and the levels too small for the sweeps),
// and the device-side view helpers (conv_show, compare).
#include "dwt_device.h"

namespace dwt {

// ---- interleaved layout: one phase of the reference's phase-ordered lifting, exact ----
// Same windowed evaluation as k_line_pass, with every step masked to the index range the
// phase gives it.  Mirrored window entries (symmetric extension) carry the index they
// mirror, so they receive the same masked updates as their originals.
template <class W, bool INV>
__global__ __launch_bounds__(256) void k_il_phase(const char *__restrict__ src, char *__restrict__ dst,
	long line_stride, long elem_stride, int n_lines, int N, int lanes_along_lines, IlPhase ph, int k_lo, int k_hi)
{
	using T = typename W::T;
	constexpr int K = W::K, NW = 2 * K + 1;
	const int fast = blockIdx.x * blockDim.x + threadIdx.x;
	const int slow = blockIdx.y;
	const int line = lanes_along_lines ? fast : slow;
	const int k = k_lo + (lanes_along_lines ? slow : fast); // pairs k_lo .. k_hi-1 of every line
	if (line >= n_lines || k >= k_hi)
		return;
	const char *s = src + (long)line * line_stride;
	char *d = dst + (long)line * line_stride;
	// forward: w[0] is the even sample 2k-K; inverse: the odd sample 2k-K+1
	const int first = 2 * k - K + (INV ? 1 : 0);
	T w[NW];
	int idx[NW];
#pragma unroll
	for (int j = 0; j < NW; j++) {
		idx[j] = reflect(first + j, N);
		w[j] = *(const T *)(s + (long)idx[j] * elem_stride);
		if (INV && idx[j] >= ph.sc_lo && idx[j] <= ph.sc_hi)
			w[j] = W::inv_scale(idx[j] & 1, w[j]);
	}
#pragma unroll
	for (int st = 0; st < K; st++) {
#pragma unroll
		for (int j = st + 1; j <= NW - 2 - st; j += 2)
			if (idx[j] >= ph.lo[st] && idx[j] <= ph.hi[st]) {
				// (a line end: both taps are one sample, the reference adds (2c)*x -- dwt_lift.h)
				const bool end = idx[j] == 0 || idx[j] == N - 1;
				w[j] = INV ? inv_step_at<W>(st, end, w[j], w[j - 1], w[j + 1]) : fwd_step_at<W>(st, end, w[j], w[j - 1], w[j + 1]);
			}
	}
	const int c0 = INV ? K - 1 : K; // window position of sample 2k
#pragma unroll
	for (int e = 0; e < 2; e++) {
		const int i = 2 * k + e;
		if (i < N) {
			T v = w[c0 + e];
			if (!INV && i >= ph.sc_lo && i <= ph.sc_hi)
				v = W::fwd_scale(e, v);
			*(T *)(d + (long)i * elem_stride) = v;
		}
	}
}

template <class W>
static hipError_t il_phase_t(bool inverse, const void *src, void *dst, long line_stride, long elem_stride, int n_lines, int N,
	bool lanes_along_lines, const IlPhase &ph, hipStream_t s, int k_lo, int k_hi)
{
	if (n_lines <= 0 || N < 2)
		return hipErrorInvalidValue;
	const int npairs = (N + 1) >> 1;
	if (k_hi < 0 || k_hi > npairs)
		k_hi = npairs;
	if (k_lo < 0)
		k_lo = 0;
	if (k_lo >= k_hi)
		return hipSuccess;
	const int nk = k_hi - k_lo;
	const int fast = lanes_along_lines ? n_lines : nk;
	const int slow = lanes_along_lines ? nk : n_lines;
	const int bs = fast >= 256 ? 256 : 64;
	dim3 grid((fast + bs - 1) / bs, slow);
	if (inverse)
		k_il_phase<W, true><<<grid, bs, 0, s>>>((const char *)src, (char *)dst, line_stride, elem_stride, n_lines, N, lanes_along_lines, ph, k_lo, k_hi);
	else
		k_il_phase<W, false><<<grid, bs, 0, s>>>((const char *)src, (char *)dst, line_stride, elem_stride, n_lines, N, lanes_along_lines, ph, k_lo, k_hi);
	return hipGetLastError();
}

hipError_t launch_il_phase(Wavelet w, bool inverse, const void *src, void *dst, long line_stride, long elem_stride,
	int n_lines, int N, bool lanes_along_lines, const IlPhase &ph, hipStream_t s, int k_lo, int k_hi)
{
	switch (w) {
	case kCdf97S: return il_phase_t<Cdf97S>(inverse, src, dst, line_stride, elem_stride, n_lines, N, lanes_along_lines, ph, s, k_lo, k_hi);
	case kCdf53SNew: return il_phase_t<Cdf53SNew>(inverse, src, dst, line_stride, elem_stride, n_lines, N, lanes_along_lines, ph, s, k_lo, k_hi);
	default: break;
	}
	return hipErrorInvalidValue;
}


// ---- interleaved layout, in-place level: the snapshot of the tiles' foreign samples (IlShell, dwt_kernels.h) ----
// One thread per 16-byte piece; four parts in one launch: the rows around the tile-row boundaries, rows 0..13, the
// columns around the tile-column boundaries, the last columns.  Rows as buffers: any 4-byte aligned image.
struct IlShellGeom {
	int W, H, tile_pairs, ntx, npr; // npr: pieces of a whole row
	long n_rows, n_top, n_cols, n_right;
};

__global__ __launch_bounds__(256) void k_il_shell(const float *__restrict__ img, long pitch, IlShell sh, IlShellGeom g)
{
	long i = (long)blockIdx.x * 256 + threadIdx.x;
	const unsigned wb = (unsigned)g.W * 4;
	int y, col;
	float *drow;
	unsigned doff, dbytes;
	if (i < g.n_rows) {
		const int slot = (int)(i / g.npr), k = slot / 9 + 1;
		y = 2 * k * g.tile_pairs - 5 + slot % 9;
		col = (int)(i % g.npr) * 4;
		drow = (float *)sh.rows + (long)slot * sh.rows_pitch;
		doff = (unsigned)col * 4;
		dbytes = (unsigned)sh.rows_pitch * 4;
	} else if ((i -= g.n_rows) < g.n_top) {
		y = (int)(i / g.npr);
		col = (int)(i % g.npr) * 4;
		drow = (float *)sh.top + (long)y * sh.top_pitch;
		doff = (unsigned)col * 4;
		dbytes = (unsigned)sh.top_pitch * 4;
	} else if ((i -= g.n_top) < g.n_cols) {
		const int npc = 2 * (g.ntx - 1), k = (int)(i % npc);
		y = (int)(i / npc);
		col = 256 * (k / 2 + 1) - 4 + 4 * (k & 1);
		drow = (float *)sh.cols + (long)y * sh.cols_pitch;
		doff = (unsigned)k * 16;
		dbytes = (unsigned)sh.cols_pitch * 4;
	} else if ((i -= g.n_cols) < g.n_right) {
		const int k = (int)(i % (kIlShellRight / 4));
		y = (int)(i / (kIlShellRight / 4));
		col = sh.right_x0 + 4 * k;
		drow = (float *)sh.right + (long)y * sh.right_pitch;
		doff = (unsigned)k * 16;
		dbytes = (unsigned)sh.right_pitch * 4;
	} else {
		return;
	}
	if (y >= g.H)
		return;
	// (a piece that straddles the row's end: zero fill -- its missing columns are never read, reflection brings
	// them from a tile's own side)
	const u4 v = load16_row<true>(row_rsrc(img + (long)y * pitch, wb), (unsigned)col * 4);
	store16_row<false>(row_rsrc(drow, dbytes), doff, v);
}

static bool il_shell_shape(int W, int H, int tile_pairs, IlShellGeom *g, size_t off[5])
{
	if (W < 64 || H < 64 || tile_pairs < 8 || (W % 256 != 0 && W % 256 < 8))
		return false;
	const int Hd = (H + 1) / 2, nty = (Hd + tile_pairs - 1) / tile_pairs, ntx = (W + 255) / 256;
	const long rp = (W + 3) / 4 * 4;
	g->W = W; g->H = H; g->tile_pairs = tile_pairs; g->ntx = ntx; g->npr = (W + 3) / 4;
	g->n_rows = (long)9 * (nty - 1) * g->npr;
	g->n_top = (long)14 * g->npr;
	g->n_cols = (long)H * 2 * (ntx - 1);
	g->n_right = (long)H * (kIlShellRight / 4);
	off[0] = 0;                                                  // rows
	off[1] = off[0] + (size_t)9 * (nty - 1) * rp * 4;           // top
	off[2] = off[1] + (size_t)14 * rp * 4;                       // cols
	off[3] = off[2] + (size_t)H * 8 * (ntx - 1) * 4;             // right
	off[4] = off[3] + (size_t)H * kIlShellRight * 4;             // end
	return true;
}

size_t il_shell_bytes(int W, int H, int tile_pairs)
{
	IlShellGeom g;
	size_t off[5];
	return il_shell_shape(W, H, tile_pairs, &g, off) ? off[4] : 0;
}

hipError_t launch_il_shell(const float *img, long pitch, int W, int H, int tile_pairs, float *scratch, IlShell *sh, hipStream_t s)
{
	IlShellGeom g;
	size_t off[5];
	if (!il_shell_shape(W, H, tile_pairs, &g, off) || !aligned16(scratch))
		return hipErrorInvalidValue;
	const long rp = (W + 3) / 4 * 4;
	char *b = (char *)scratch;
	sh->rows = (const float *)(b + off[0]); sh->rows_pitch = rp;
	sh->top = (const float *)(b + off[1]); sh->top_pitch = rp;
	sh->cols = (const float *)(b + off[2]); sh->cols_pitch = 8L * (g.ntx - 1);
	sh->right = (const float *)(b + off[3]); sh->right_pitch = kIlShellRight;
	sh->right_x0 = (W - 16) & ~3;
	sh->tile_pairs = tile_pairs;
	const long n = g.n_rows + g.n_top + g.n_cols + g.n_right;
	if ((n + 255) / 256 > 0x7fffffffL)
		return hipErrorInvalidValue;
	k_il_shell<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(img, pitch, *sh, g);
	return hipGetLastError();
}

} // namespace dwt

// ---------------------------------------------------------------------------------
// 4. device-side view helpers: conv_show and compare on images
//    that stay in HBM between the forward and the inverse transform
// ---------------------------------------------------------------------------------
namespace dwt {

// dwt_util_conv_show_s (src/libdwt.c:21075-21117): log(1 + |c|*100) / 10 with the log
// taken in double as log_i_s does (:21010); non-finite results become 0.
__global__ __launch_bounds__(256) void k_conv_show_s(const char *__restrict__ src, char *__restrict__ dst, long pitch, int w, int h)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	if (x >= w || y >= h)
		return;
	const float c = *(const float *)(src + (long)y * pitch + (long)x * 4);
	float t = (float)log((double)(1.f + fabsf(c) * 100.f));
	t /= 10.f;
	if (!isfinite(t))
		t = 0.f;
	*(float *)(dst + (long)y * pitch + (long)x * 4) = t;
}

// dwt_util_conv_show_i (src/libdwt.c:21020-21044): |c|
__global__ __launch_bounds__(256) void k_conv_show_i(const char *__restrict__ src, char *__restrict__ dst, long pitch, int w, int h)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	if (x >= w || y >= h)
		return;
	const int c = *(const int *)(src + (long)y * pitch + (long)x * 4);
	*(int *)(dst + (long)y * pitch + (long)x * 4) = c < 0 ? -c : c;
}

// dwt_util_compare_s / _i (src/libdwt.c:1593-1620, 1531-1558): count of differing
// elements (float: |a-b| > 1e-3 or any NaN/Inf; int: a != b) accumulated in *result
template <bool IS_INT>
__global__ __launch_bounds__(256) void k_compare(const char *__restrict__ p1, const char *__restrict__ p2, long pitch, int w, int h, unsigned *result)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	bool differ = false;
	if (x < w && y < h) {
		if (IS_INT) {
			differ = *(const int *)(p1 + (long)y * pitch + (long)x * 4) != *(const int *)(p2 + (long)y * pitch + (long)x * 4);
		} else {
			const float a = *(const float *)(p1 + (long)y * pitch + (long)x * 4);
			const float b = *(const float *)(p2 + (long)y * pitch + (long)x * 4);
			differ = isnan(a) || isinf(a) || isnan(b) || isinf(b) || fabsf(a - b) > 1e-3f;
		}
	}
	const unsigned long long m = __ballot(differ);
	if ((threadIdx.x & 63) == 0 && m)
		atomicAdd(result, (unsigned)__popcll(m));
}

// Up to three device-to-device rectangle copies in ONE launch (the copy-back of an in-place
// level 0: HL, LH|HH and, when level 0 is the last level, LL).  The runtime's own 2-D copy
// (__amd_rocclr_copyBufferRectAligned) moves such rectangles at ~3.3 TB/s; this streams 16 B per
// lane, non-temporal both ways (the bytes are not read again by a kernel), 4 KiB x 8 rows per
// workgroup, any 4-byte alignment.
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy_rects(CopyRects r)
{
	copy_rects_block<NTL, NTS>(r, r.block0 + (int)blockIdx.x);
}

int copy_rects_plan(CopyRects *r)
{
	int n = 0;
	for (int k = 0; k < r->n; k++) {
		if (r->wbytes[k] <= 0 || r->h[k] <= 0)
			continue;
		r->src[n] = r->src[k]; r->dst[n] = r->dst[k]; r->spitch[n] = r->spitch[k]; r->dpitch[n] = r->dpitch[k];
		r->wbytes[n] = r->wbytes[k]; r->h[n] = r->h[k];
		if (r->wbytes[n] % 4 || ((uintptr_t)r->src[n] | (uintptr_t)r->dst[n] | (uintptr_t)r->spitch[n] | (uintptr_t)r->dpitch[n]) % 4)
			return -1;
		n++;
	}
	r->n = n;
	const int seg = 256 * 16;
	int total = 0;
	for (int k = 0; k < n; k++) {
		r->first_block[k] = total;
		total += ((r->wbytes[k] + seg - 1) / seg) * ((r->h[k] + 7) / 8);
	}
	for (int k = n; k < 4; k++)
		r->first_block[k] = total;
	return total;
}

hipError_t launch_copy_rects_range(CopyRects r, int lo, int hi, hipStream_t s)
{
	if (hi <= lo)
		return hipSuccess;
	r.block0 = lo;
	if ((r.policy & 3) == 0)
		k_copy_rects<false, false><<<hi - lo, 256, 0, s>>>(r);
	else
		k_copy_rects<true, true><<<hi - lo, 256, 0, s>>>(r);
	return hipGetLastError();
}

hipError_t launch_copy_rects(CopyRects r, hipStream_t s)
{
	const int total = copy_rects_plan(&r);
	if (total < 0)
		return hipErrorInvalidValue;
	return launch_copy_rects_range(r, 0, total, s);
}

hipError_t launch_conv_show(bool is_int, const void *src, void *dst, long pitch, int w, int h, hipStream_t s)
{
	if (w <= 0 || h <= 0)
		return hipSuccess;
	dim3 grid((w + 255) / 256, h);
	if (is_int)
		k_conv_show_i<<<grid, 256, 0, s>>>((const char *)src, (char *)dst, pitch, w, h);
	else
		k_conv_show_s<<<grid, 256, 0, s>>>((const char *)src, (char *)dst, pitch, w, h);
	return hipGetLastError();
}

hipError_t launch_compare(bool is_int, const void *p1, const void *p2, long pitch, int w, int h, unsigned *result, hipStream_t s)
{
	if (w <= 0 || h <= 0)
		return hipSuccess;
	dim3 grid((w + 255) / 256, h);
	if (is_int)
		k_compare<true><<<grid, 256, 0, s>>>((const char *)p1, (const char *)p2, pitch, w, h, result);
	else
		k_compare<false><<<grid, 256, 0, s>>>((const char *)p1, (const char *)p2, pitch, w, h, result);
	return hipGetLastError();
}

} // namespace dwt
