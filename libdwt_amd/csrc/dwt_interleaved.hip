// dwt_interleaved.hip -- kernels of the interleaved (in-place lifting) layout beyond the sweeps:
// the exact phase-ordered line kernel, the compose / decompose passes over the level lattices,
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
			if (idx[j] >= ph.lo[st] && idx[j] <= ph.hi[st])
				w[j] = INV ? W::inv_step(st, w[j], w[j - 1], w[j + 1]) : W::fwd_step(st, w[j], w[j - 1], w[j + 1]);
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

// ---- interleaved layout: exact border strips of the fused path, one launch per level ----
// The fused sweep finishes rows before columns; the reference's phase order rounds differently in
// the top 8 rows and the last 5 columns of a level only (dwt_backend_il.hip, il_exact_strips).  A
// workgroup takes a tile of such a strip plus a 12-sample margin from the level's INPUT into LDS
// -- 24 samples across the strip (from the image border), 128 along it --, runs the six phase
// passes in the reference's order (rows' prolog, columns' prolog, rows' core, columns' core,
// rows' epilog, columns' epilog) and writes the part no artificial tile edge can have reached
// (4 samples per pass along the strip, 12 in all; 16 across it) over the sweep's result and over
// the dense low-pass copy the next level reads.  A pass is the reference's line kernel restricted
// to the phase: x[i] += c * (x[i-1] + x[i+1]) over the index range the phase owns, step after
// step, TRUE line indices, mirrored neighbours at the image border.  Each thread holds a piece of
// a line in registers -- a whole 24-sample line across the strip, or 8 samples plus 4 either side
// along it -- so a pass costs one barrier (ping-pong LDS images).  LDS image: [across][along]; the
// right strip is held transposed, so both strips run the same code.
template <class W, bool INV, int NE>
static __device__ __forceinline__ void il_phase_piece(typename W::T (&v)[NE], int i0, int N, int o, int e, const IlPhase &ph)
{
	// v[j] <-> true index i0 + j, i0 EVEN (the parity of j is the parity of the index); the tile
	// holds [o, e).  Branch-free: every candidate update is computed and kept or dropped by a
	// select.  An update needs both neighbours inside the tile and inside the piece -- except at
	// the true ends of the line, where the missing neighbour is the mirror image of the other.
	using T = typename W::T;
	constexpr int K = W::K;
	if (INV) {
#pragma unroll
		for (int j = 0; j < NE; j++) {
			const int i = i0 + j;
			const T sc = W::inv_scale(j & 1, v[j]);
			v[j] = (i >= ph.sc_lo && i <= ph.sc_hi) ? sc : v[j];
		}
	}
	const int lo_t = o == 0 ? 0 : o + 1, hi_t = e == N ? N - 1 : e - 2; // both neighbours in the tile
#pragma unroll
	for (int st = 0; st < K; st++) {
		constexpr int dummy = 0;
		(void)dummy;
		const int par = INV ? (st & 1) : !(st & 1);
		const int lo = max(ph.lo[st], lo_t), hi = min(ph.hi[st], hi_t);
#pragma unroll
		for (int j = 0; j < NE; j++) {
			if ((j & 1) != par)
				continue; // compile time
			const int i = i0 + j;
			bool ok = i >= lo && i <= hi;
			T l, r;
			if (j == 0) {
				l = v[1];
				ok = ok && i == 0;
			} else {
				l = (i == 0) ? v[j + 1 < NE ? j + 1 : j] : v[j - 1];
			}
			if (j == NE - 1) {
				r = v[NE - 2];
				ok = ok && i == N - 1;
			} else {
				r = (i == N - 1) ? v[j > 0 ? j - 1 : j] : v[j + 1];
			}
			const T nv = INV ? W::inv_step(st, v[j], l, r) : W::fwd_step(st, v[j], l, r);
			v[j] = ok ? nv : v[j];
		}
	}
	if (!INV) {
#pragma unroll
		for (int j = 0; j < NE; j++) {
			const int i = i0 + j;
			const T sc = W::fwd_scale(j & 1, v[j]);
			v[j] = (i >= ph.sc_lo && i <= ph.sc_hi) ? sc : v[j];
		}
	}
}

template <class W, bool INV>
__global__ __launch_bounds__(512) void k_il_strip(IlStripArgs a)
{
	using T = typename W::T;
	// kept across the strip: 8 rows from the top / the last 8 columns.  The reference's order differs from
	// the sweep's in the last 5 (6 with the parity) columns only; 8 makes the kept region closed under
	// "what a forward level computes from the uncorrected low-pass samples of the level above" (the
	// lazy strips of il_level); the band's artificial edge, 23-24 samples from the border, reaches 4.
	constexpr int kKeep = 104, kMargin = 12, kBand = 24, kKeepTop = 8, kKeepRight = 8, kLong = kKeep + 2 * kMargin;
	constexpr int kOwn = 8, kHalo = 4, kPiece = kOwn + 2 * kHalo;
	// the lazy strips (dwt_backend_il.hip) rely on this: a forward output depends on inputs at most K away, so what the
	// next level computes from not-yet-corrected low-pass samples (rows 0..K-1, the last K columns) stays inside ITS kept
	// region of 2 K rows / columns and is recomputed by its own strips (a race-check tool will flag that read)
	static_assert(kKeepTop >= 2 * W::K && kKeepRight >= 2 * W::K, "kept region must close over the lifting reach");
	__shared__ T buf[2][kBand * kLong];
	// blocks [0, n_top): tiles of the top strip; the rest: tiles of the right strip
	const bool top = (int)blockIdx.x < a.n_top;
	const int tile = top ? blockIdx.x : blockIdx.x - a.n_top;
	// "long" axis: along the strip (x for the top strip, y for the right one); "short": across it
	const int n_long = top ? a.lx : a.ly, n_short = top ? a.ly : a.lx;
	const int l0 = tile * kKeep;
	const int ol = max(0, l0 - kMargin) & ~1, el = min(n_long, l0 + kKeep + kMargin); // tile range, long axis
	// the band starts on an even index (23 or 24 samples for the right strip): piece parity = index parity
	const int os = top ? 0 : max(0, n_short - (kBand - 1)) & ~1, es = top ? min(n_short, kBand) : n_short;
	const int nl = el - ol, ns = es - os;
	auto load = [&](int y, int x) -> T {
		return (a.in_even && !(y & 1)) ? a.in_even[(long)(y >> 1) * a.even_pitch + x] : a.in[(long)y * a.in_pitch + x];
	};
	// tile -> LDS: all of a thread's loads in flight together (kBand * kLong = 6 x 512 elements)
	{
		constexpr int kPer = kBand * kLong / 512;
		T tmp[kPer];
		int where[kPer];
#pragma unroll
		for (int k = 0; k < kPer; k++) {
			const int idx = threadIdx.x + 512 * k;
			// top strip: consecutive lanes along x; right strip: kBand consecutive lanes share an image row
			const int sl = top ? idx / kLong : idx % kBand, ll_ = top ? idx % kLong : idx / kBand;
			const bool ok = sl < ns && ll_ < nl;
			where[k] = ok ? sl * kLong + ll_ : -1;
			const int sh = os + sl, lo = ol + ll_;
			tmp[k] = ok ? (top ? load(sh, lo) : load(lo, sh)) : T(0);
		}
#pragma unroll
		for (int k = 0; k < kPer; k++)
			if (where[k] >= 0)
				buf[0][where[k]] = tmp[k];
	}
	__syncthreads();
	int cur = 0;
#pragma unroll 1
	for (int pass = 0; pass < 6; pass++) {
		const bool rows = !(pass & 1);
		const bool along_long = top ? rows : !rows;
		const IlPhase ph = rows ? a.rph[pass >> 1] : a.cph[pass >> 1];
		const int N = rows ? a.lx : a.ly;
		{
			// a phase that owns no index inside the tile leaves it as it is: no pass, no barrier
			const int o = along_long ? ol : os, e = along_long ? el : es;
			bool touches = ph.sc_lo <= ph.sc_hi && ph.sc_hi >= o && ph.sc_lo < e;
#pragma unroll
			for (int st = 0; st < W::K; st++)
				touches = touches || (ph.lo[st] <= ph.hi[st] && ph.hi[st] >= o && ph.lo[st] < e);
			if (!touches)
				continue;
		}
		const T *src = buf[cur];
		T *dst = buf[cur ^ 1];
		if (along_long) {
			// a thread: 8 samples of one line + 4 either side; pieces of a line on consecutive lanes
			const int npieces = (nl + kOwn - 1) / kOwn; // <= 16
			for (int it = threadIdx.x; it < ns * 16; it += blockDim.x) {
				const int line = it >> 4, pc = it & 15;
				if (pc >= npieces)
					continue;
				const int i0 = ol + pc * kOwn - kHalo; // true index of v[0]
				T v[kPiece];
#pragma unroll
				for (int j = 0; j < kPiece; j++) {
					const int i = i0 + j;
					v[j] = (i >= ol && i < el) ? src[line * kLong + (i - ol)] : T(0);
				}
				il_phase_piece<W, INV, kPiece>(v, i0, N, ol, el, ph);
#pragma unroll
				for (int j = kHalo; j < kHalo + kOwn; j++) {
					const int i = i0 + j;
					if (i < el)
						dst[line * kLong + (i - ol)] = v[j];
				}
			}
		} else {
			// a thread: one whole line across the strip (24 samples, stride kLong)
			for (int line = threadIdx.x; line < nl; line += blockDim.x) {
				T v[kBand];
#pragma unroll
				for (int j = 0; j < kBand; j++)
					v[j] = j < ns ? src[j * kLong + line] : T(0);
				il_phase_piece<W, INV, kBand>(v, os, N, os, es, ph);
#pragma unroll
				for (int j = 0; j < kBand; j++)
					if (j < ns)
						dst[j * kLong + line] = v[j];
			}
		}
		__syncthreads();
		cur ^= 1;
	}
	// kept part: along the strip [l0, l0 + kKeep), across it the 8 rows from the top / the last 6 columns
	const int kl0 = l0, kl1 = min(n_long, l0 + kKeep);
	const int ks0 = top ? 0 : max(0, n_short - kKeepRight), ks1 = top ? min(n_short, kKeepTop) : n_short;
	const int kw = ks1 - ks0, kn = kl1 - kl0;
#pragma unroll
	for (int k = 0; k < 2; k++) { // at most 8 x 104 kept samples
		const int idx = threadIdx.x + 512 * k;
		const int sl = top ? idx / kKeep : idx % kKeepTop, ll_ = top ? idx % kKeep : idx / kKeepTop;
		if (sl < kw && ll_ < kn) {
			const int sh = ks0 + sl, lo = kl0 + ll_;
			const int y = top ? sh : lo, x = top ? lo : sh;
			const T v = buf[cur][(sh - os) * kLong + (lo - ol)];
			a.out[(long)y * a.out_pitch + x] = v;
			if (a.ll && !((x | y) & 1))
				a.ll[(long)(y >> 1) * a.ll_pitch + (x >> 1)] = v;
		}
	}
}

hipError_t launch_il_strip(Wavelet w, bool inverse, IlStripArgs a, hipStream_t s)
{
	if (a.lx < 32 || a.ly < 32)
		return hipErrorInvalidValue;
	a.n_top = (a.lx + 103) / 104;
	const int n = a.n_top + (a.ly + 103) / 104;
	switch (w) {
	case kCdf97S:
		if (inverse)
			k_il_strip<Cdf97S, true><<<n, 512, 0, s>>>(a);
		else
			k_il_strip<Cdf97S, false><<<n, 512, 0, s>>>(a);
		break;
	case kCdf53SNew:
		if (inverse)
			return hipErrorInvalidValue;
		k_il_strip<Cdf53SNew, false><<<n, 512, 0, s>>>(a);
		break;
	default:
		return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// ---- interleaved layout: all levels' lattices in one pass over the even rows ----
// A lattice-1 point (p, q) (image column 2p, row 2q) belongs to level
// j = 1 + min(ctz(p), ctz(q)) capped at J-1; its sample sits at (p >> (j-1), q >> (j-1))
// of that level's dense image.  One thread owns 8 image columns of one even row.
static __device__ __forceinline__ int il_level_of(int p, int q, int J)
{
	const int t = __builtin_ctz((unsigned)(p | q) | (1u << 30)); // ctz(0) -> 30
	const int j = 1 + t;
	return j < J ? j : J - 1;
}

__global__ __launch_bounds__(256) void k_il_compose(const float *__restrict__ base, long base_pitch, float *__restrict__ out,
	long out_pitch, int W, int H, IlPyramid py, int vec_ok, int out_dense, int x_begin)
{
	// grid.x = even rows (may exceed 65535), grid.y = blocks of 2048 columns from x_begin (a multiple of 8)
	const int x0 = x_begin + (blockIdx.y * blockDim.x + threadIdx.x) * 8;
	const int q = blockIdx.x, y = 2 * q;
	if (x0 >= W || y >= H)
		return;
	const float *b = base + (long)y * base_pitch + x0;
	float *o = out + (long)(out_dense ? q : y) * out_pitch + x0;
	const int p0 = x0 >> 1;
	float v[8];
	const bool vec = vec_ok && x0 + 8 <= W;
	if (vec) {
		const u4 t0 = *(const u4 *)b, t1 = *(const u4 *)(b + 4);
#pragma unroll
		for (int e = 0; e < 4; e++) {
			v[e] = from_bits<float>(t0[e]);
			v[4 + e] = from_bits<float>(t1[e]);
		}
		const u4 l1 = *(const u4 *)(py.p[1] + (long)q * py.pitch[1] + p0);
#pragma unroll
		for (int i = 0; i < 4; i++)
			v[2 * i] = from_bits<float>(l1[i]);
	} else {
#pragma unroll
		for (int e = 0; e < 8; e++)
			if (x0 + e < W)
				v[e] = (e & 1) ? b[e] : py.p[1][(long)q * py.pitch[1] + p0 + (e >> 1)];
	}
	if (py.J > 2 && !(q & 1)) {
		// p0 is a multiple of 4: the points p0 and p0+2 lie on deeper lattices
#pragma unroll
		for (int i = 0; i < 4; i += 2)
			if (x0 + 2 * i < W) {
				const int p = p0 + i, j = il_level_of(p, q, py.J);
				v[2 * i] = py.p[j][(long)(q >> (j - 1)) * py.pitch[j] + (p >> (j - 1))];
			}
	}
	if (vec) {
		*(u4 *)o = u4{to_bits(v[0]), to_bits(v[1]), to_bits(v[2]), to_bits(v[3])};
		*(u4 *)(o + 4) = u4{to_bits(v[4]), to_bits(v[5]), to_bits(v[6]), to_bits(v[7])};
	} else {
#pragma unroll
		for (int e = 0; e < 8; e++)
			if (x0 + e < W)
				o[e] = v[e];
	}
}

__global__ __launch_bounds__(256) void k_il_decompose(const float *__restrict__ img, long pitch, int W, int H, IlPyramid py, int vec_ok)
{
	const int x0 = (blockIdx.y * blockDim.x + threadIdx.x) * 8;
	const int q = blockIdx.x, y = 2 * q;
	if (x0 >= W || y >= H)
		return;
	const float *b = img + (long)y * pitch + x0;
	const int p0 = x0 >> 1;
	float v[4];
	const bool vec = vec_ok && x0 + 8 <= W;
	if (vec) {
		const u4 t0 = *(const u4 *)b, t1 = *(const u4 *)(b + 4);
		v[0] = from_bits<float>(t0[0]); v[1] = from_bits<float>(t0[2]);
		v[2] = from_bits<float>(t1[0]); v[3] = from_bits<float>(t1[2]);
		*(u4 *)(py.p[1] + (long)q * py.pitch[1] + p0) = u4{to_bits(v[0]), to_bits(v[1]), to_bits(v[2]), to_bits(v[3])};
	} else {
#pragma unroll
		for (int i = 0; i < 4; i++)
			if (x0 + 2 * i < W) {
				v[i] = b[2 * i];
				py.p[1][(long)q * py.pitch[1] + p0 + i] = v[i];
			}
	}
	// deeper lattices: level j takes the points whose p and q are multiples of 2^(j-1)
	for (int j = 2; j < py.J; j++) {
		const int m = (1 << (j - 1)) - 1;
		if (q & m)
			break;
#pragma unroll
		for (int i = 0; i < 4; i += 2)
			if (!((p0 + i) & m) && x0 + 2 * i < W)
				py.p[j][(long)(q >> (j - 1)) * py.pitch[j] + ((p0 + i) >> (j - 1))] = v[i];
	}
}

static int il_vec_ok(const float *a, long ap, const float *b, long bp, const IlPyramid &py)
{
	return aligned16(a) && aligned16(b) && ap % 4 == 0 && bp % 4 == 0 && py.J > 1 && aligned16(py.p[1]) && py.pitch[1] % 4 == 0;
}

hipError_t launch_il_compose(const float *base, long base_pitch, float *out, long out_pitch, int W, int H, const IlPyramid &py, hipStream_t s,
	bool out_dense, int x_begin)
{
	if (py.J < 2 || py.J > 24 || W < 1 || H < 1 || x_begin < 0 || x_begin >= W || (x_begin & 7) || ((W + 7) / 8 + 255) / 256 > 65535)
		return hipErrorInvalidValue;
	dim3 grid((H + 1) / 2, ((W - x_begin + 7) / 8 + 255) / 256);
	k_il_compose<<<grid, 256, 0, s>>>(base, base_pitch, out, out_pitch, W, H, py, il_vec_ok(base, base_pitch, out, out_pitch, py), out_dense, x_begin);
	return hipGetLastError();
}

hipError_t launch_il_decompose(const float *img, long pitch, int W, int H, const IlPyramid &py, hipStream_t s)
{
	if (py.J < 2 || py.J > 24 || W < 1 || H < 1 || ((W + 7) / 8 + 255) / 256 > 65535)
		return hipErrorInvalidValue;
	dim3 grid((H + 1) / 2, ((W + 7) / 8 + 255) / 256);
	k_il_decompose<<<grid, 256, 0, s>>>(img, pitch, W, H, py, il_vec_ok(img, pitch, img, pitch, py));
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
	int b = blockIdx.x, k = 0;
	while (k + 1 < r.n && b >= r.first_block[k + 1])
		k++;
	b -= r.first_block[k];
	constexpr int kRows = 8;
	constexpr int seg = 256 * 16; // bytes per workgroup row segment
	const int nbx = (r.wbytes[k] + seg - 1) / seg;
	const int bx = b % nbx, by = b / nbx;
	const unsigned x = (unsigned)bx * seg + threadIdx.x * 16;
	const char *s = r.src[k] + (long)by * kRows * r.spitch[k];
	char *d = r.dst[k] + (long)by * kRows * r.dpitch[k];
	const int rows = min(kRows, r.h[k] - by * kRows);
	// rows as buffers: 16 B per lane whatever the alignment, the dwords past a row's end are
	// zero-filled / dropped by the bounds check.  Rows past the rectangle's end load the last row
	// again (never stored): straight-line loads, all eight in flight (with the loads under
	// `if (i < rows)` the compiler built a 236-register kernel).
	u4 v[kRows];
#pragma unroll
	for (int i = 0; i < kRows; i++)
		v[i] = load16_row<NTL>(row_rsrc(s + (long)min(i, rows - 1) * r.spitch[k], (unsigned)r.wbytes[k]), x);
#pragma unroll
	for (int i = 0; i < kRows; i++)
		if (i < rows)
			store16_row<NTS>(row_rsrc(d + (long)i * r.dpitch[k], (unsigned)r.wbytes[k]), x, v[i]);
}

hipError_t launch_copy_rects(CopyRects r, hipStream_t s)
{
	int n = 0;
	for (int k = 0; k < r.n; k++) {
		if (r.wbytes[k] <= 0 || r.h[k] <= 0)
			continue;
		r.src[n] = r.src[k]; r.dst[n] = r.dst[k]; r.spitch[n] = r.spitch[k]; r.dpitch[n] = r.dpitch[k];
		r.wbytes[n] = r.wbytes[k]; r.h[n] = r.h[k];
		if (r.wbytes[n] % 4 || ((uintptr_t)r.src[n] | (uintptr_t)r.dst[n] | (uintptr_t)r.spitch[n] | (uintptr_t)r.dpitch[n]) % 4)
			return hipErrorInvalidValue;
		n++;
	}
	r.n = n;
	if (n == 0)
		return hipSuccess;
	const int seg = 256 * 16;
	int total = 0;
	for (int k = 0; k < n; k++) {
		r.first_block[k] = total;
		total += ((r.wbytes[k] + seg - 1) / seg) * ((r.h[k] + 7) / 8);
	}
	r.first_block[n] = total;
	if ((r.policy & 3) == 0)
		k_copy_rects<false, false><<<total, 256, 0, s>>>(r);
	else
		k_copy_rects<true, true><<<total, 256, 0, s>>>(r);
	return hipGetLastError();
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
