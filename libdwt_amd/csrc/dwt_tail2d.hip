// dwt_tail2d.hip -- the small levels of the 2-D forward drivers: one or two decomposition levels of
// a tile computed entirely in LDS.
//
// A level of a few million samples or less is one round of waves for k_fwd_sweep, and its duration
// is the length of ONE wave's serial chain -- 4 warm-up iterations plus the tile's row pairs, about
// a microsecond each -- whatever the level's size: 13.7 / 9.6 / 8.6 us for the 2048^2 / 1024^2 /
// 512^2 levels of one 8192^2 image (profiles/r02_single_image_timeline.md).  Here a workgroup owns a
// tile of 16 x 16 coefficient pairs of level j+1 (32 x 32 of level j), loads the (85 x 85 for 9/7)
// samples of level j's input they depend on into LDS with all loads in flight at once, and lifts
// rows, then columns, with every thread on its own piece of a line: the chain is a handful of
// 24-sample register windows instead of dozens of sweep iterations, the level-j+1 input never leaves
// the chip, and two launches (and two LL round trips through HBM) become one.  The halo a tile needs
// from its neighbours is recomputed (1.76 x the arithmetic at level j) -- irrelevant at these sizes.
//
// Arithmetic: rows before columns, scale after each direction, the same lift_fwd_regs steps with
// whole-sample symmetric reflection of the TRUE line indices as k_fwd_sweep / k_line_pass, so the
// bits are the reference's (src/libdwt.c:12812-12919 forward driver, :10744-10800 line kernel).
#include "dwt_device.h"

namespace dwt {

namespace {

constexpr int kTailPairs = 16; // pairs of the deepest level per tile side
constexpr int kChunk = 8;      // pairs one thread lifts at a time

// One 1-D pass over a tile held in LDS.  `src` holds the samples [so, so + sn) of `lines` lines of
// true length N (element i of line l at src[i * s_elem + l * s_line]); the pass produces the pairs
// [p0, p0 + np): low-pass to dl, high-pass to dh (pair p of line l at d?[(p - p0) * d_elem + l * d_line]).
template <class W>
static __device__ __forceinline__ void tile_pass(const typename W::T *src, int s_elem, int s_line, int so, int sn, int N, int lines,
	int p0, int np, typename W::T *dl, typename W::T *dh, int d_elem, int d_line)
{
	using T = typename W::T;
	constexpr int K = W::K, NW = 2 * kChunk + 2 * K;
	const int nchunks = (np + kChunk - 1) / kChunk;
	for (int it = threadIdx.x; it < lines * nchunks; it += blockDim.x) {
		const int l = it % lines, pc = p0 + (it / lines) * kChunk; // consecutive lanes: consecutive lines
		const T *s = src + l * s_line;
		T w[NW];
		const int first = 2 * pc - K;
		if (first >= so && first + NW <= so + sn && first + NW <= N) {
			// interior chunk: no reflection, constant offsets from one address
			const T *s0 = s + (first - so) * s_elem;
#pragma unroll
			for (int j = 0; j < NW; j++)
				w[j] = s0[j * s_elem];
		} else {
#pragma unroll
			for (int j = 0; j < NW; j++) {
				// (the padding pairs of a last, partial chunk reach past the stored samples: clamped, discarded)
				const int i = min(max(reflect1(first + j, N) - so, 0), sn - 1);
				w[j] = s[i * s_elem];
			}
		}
		lift_fwd_regs<W, NW>(w);
#pragma unroll
		for (int m = 0; m < kChunk; m++) {
			if (pc + m < p0 + np) {
				dl[(pc + m - p0) * d_elem + l * d_line] = W::fwd_scale(0, w[K + 2 * m]);
				dh[(pc + m - p0) * d_elem + l * d_line] = W::fwd_scale(1, w[K + 2 * m + 1]);
			}
		}
	}
}

struct Range {
	int lo, n; // [lo, lo + n)
};

// pairs [a, a + cnt) of a level with `npairs` pairs: the pairs of the level above they depend on
// (their input samples), clipped to that level's [0, N) -- reflections land inside the clip
template <int K>
static __device__ __forceinline__ Range inputs_of(int a, int cnt, int N)
{
	const int lo = max(0, 2 * a - K), hi = min(N - 1, 2 * (a + cnt - 1) + K);
	return Range{lo, hi - lo + 1};
}

} // namespace

// LEVELS: 1 or 2.  Level j reads the W x H image at `in`; its detail subbands go to `out_h` at their
// Mallat offsets; with LEVELS == 2 level j+1 runs on the LL band in LDS and its details go to
// `out_h` too (top-left quadrant); the deepest LL band goes to `out_ll`.  W, H multiples of 2^LEVELS.
template <class W, int LEVELS>
__global__ __launch_bounds__(256) void k_fwd_tail(FwdLevelArgs a)
{
	using T = typename W::T;
	constexpr int K = W::K;
	constexpr int kP1 = LEVELS == 2 ? 2 * kTailPairs - 1 + 2 * K : kTailPairs; // level-j pairs per tile side (max)
	constexpr int kI = 2 * kP1 - 1 + 2 * K;                                    // level-j input samples per side (max)
	// LDS pitches: ODD, so that lanes on consecutive lines (stride = pitch) fall on distinct banks
	constexpr int kPI = kI | 1, kPP = kP1 | 1, kPB = kTailPairs | 1;
	// R0: input tile, later the four subbands of level j (kP1 x kP1 each); R1 / R2: low / high
	// half after the row pass, later level j+1's row-pass halves and subbands
	__shared__ T R0[kI * kPI > 4 * kP1 * kPP ? kI * kPI : 4 * kP1 * kPP];
	__shared__ T R1[kI * kPP], R2[kI * kPP];

	const int Wd = a.W >> 1, Hd = a.H >> 1;             // level j: pairs per row / column
	const int Wl = LEVELS == 2 ? Wd >> 1 : Wd, Hl = LEVELS == 2 ? Hd >> 1 : Hd; // deepest level's pairs
	const int ntx = (Wl + kTailPairs - 1) / kTailPairs;
	const int ax = (blockIdx.x % ntx) * kTailPairs, ay = (blockIdx.x / ntx) * kTailPairs; // deepest pairs of this tile
	const int cx = min(kTailPairs, Wl - ax), cy = min(kTailPairs, Hl - ay);
	const int img = blockIdx.y;
	const T *in = (const T *)a.in + (long)img * a.in_bstride;
	T *out_h = (T *)a.out_h + (long)img * a.h_bstride;
	T *out_ll = (T *)a.out_ll + (long)img * a.ll_bstride;

	// level j: the pairs to produce (own details; as level j+1's input when LEVELS == 2) and the samples they need
	const Range p1x = LEVELS == 2 ? inputs_of<K>(ax, cx, Wd) : Range{ax, cx};
	const Range p1y = LEVELS == 2 ? inputs_of<K>(ay, cy, Hd) : Range{ay, cy};
	const Range i1x = inputs_of<K>(p1x.lo, p1x.n, a.W), i1y = inputs_of<K>(p1y.lo, p1y.n, a.H);

	// ---- load the input tile (all loads of a thread in flight together) ----
	{
		constexpr int kPer = (kI * kPI + 255) / 256;
		T tmp[kPer];
#pragma unroll
		for (int k = 0; k < kPer; k++) {
			const int idx = threadIdx.x + 256 * k, r = idx / kPI, c = idx % kPI;
			tmp[k] = (r < i1y.n && c < i1x.n) ? in[(long)(i1y.lo + r) * a.in_pitch + i1x.lo + c] : T(0);
		}
#pragma unroll
		for (int k = 0; k < kPer; k++) {
			const int idx = threadIdx.x + 256 * k;
			if (idx < kI * kPI)
				R0[idx] = tmp[k];
		}
	}
	__syncthreads();

	// ---- level j: rows (R0 -> R1 low, R2 high; element [row][pair]) ----
	tile_pass<W>(R0, 1, kPI, i1x.lo, i1x.n, a.W, i1y.n, p1x.lo, p1x.n, R1, R2, 1, kPP);
	__syncthreads();
	// ---- level j: columns.  From the low half: LL, LH; from the high half: HL, HH ([pair_y][pair_x]) ----
	T *LL = R0, *LH = R0 + kP1 * kPP, *HL = R0 + 2 * kP1 * kPP, *HH = R0 + 3 * kP1 * kPP;
	tile_pass<W>(R1, kPP, 1, i1y.lo, i1y.n, a.H, p1x.n, p1y.lo, p1y.n, LL, LH, kPP, 1);
	tile_pass<W>(R2, kPP, 1, i1y.lo, i1y.n, a.H, p1x.n, p1y.lo, p1y.n, HL, HH, kPP, 1);
	__syncthreads();

	// ---- level j: details of the tile's OWN pairs to their Mallat places ----
	{
		const int ox = LEVELS == 2 ? 2 * ax : ax, oy = LEVELS == 2 ? 2 * ay : ay;     // first own pair
		const int nx = LEVELS == 2 ? 2 * cx : cx, ny = LEVELS == 2 ? 2 * cy : cy;
		for (int it = threadIdx.x; it < nx * ny; it += blockDim.x) {
			const int x = it % nx, y = it / nx;
			const int li = (oy + y - p1y.lo) * kPP + (ox + x - p1x.lo);
			out_h[(long)(oy + y) * a.h_pitch + Wd + ox + x] = HL[li];
			out_h[(long)(Hd + oy + y) * a.h_pitch + ox + x] = LH[li];
			out_h[(long)(Hd + oy + y) * a.h_pitch + Wd + ox + x] = HH[li];
			if (LEVELS == 1)
				out_ll[(long)(oy + y) * a.ll_pitch + ox + x] = LL[li];
		}
	}
	if constexpr (LEVELS == 2) {
		// ---- level j+1 on the LL band in LDS (true size Wd x Hd, tile origin p1y.lo / p1x.lo) ----
		const int Wd2 = Wd >> 1, Hd2 = Hd >> 1;
		tile_pass<W>(LL, 1, kPP, p1x.lo, p1x.n, Wd, p1y.n, ax, cx, R1, R2, 1, kPB);
		__syncthreads(); // R1 / R2 complete; every thread has stored its level-j details: their place in R0 is free
		T *LL2 = R0 + kP1 * kPP;
		T *LH2 = LL2 + kTailPairs * kPB, *HL2 = LH2 + kTailPairs * kPB, *HH2 = HL2 + kTailPairs * kPB;
		tile_pass<W>(R1, kPB, 1, p1y.lo, p1y.n, Hd, cx, ay, cy, LL2, LH2, kPB, 1);
		tile_pass<W>(R2, kPB, 1, p1y.lo, p1y.n, Hd, cx, ay, cy, HL2, HH2, kPB, 1);
		__syncthreads();
		for (int it = threadIdx.x; it < cx * cy; it += blockDim.x) {
			const int x = it % cx, y = it / cx;
			const int li = y * kPB + x;
			out_ll[(long)(ay + y) * a.ll_pitch + ax + x] = LL2[li];
			out_h[(long)(ay + y) * a.h_pitch + Wd2 + ax + x] = HL2[li];
			out_h[(long)(Hd2 + ay + y) * a.h_pitch + ax + x] = LH2[li];
			out_h[(long)(Hd2 + ay + y) * a.h_pitch + Wd2 + ax + x] = HH2[li];
		}
	}
}

// true when launch_fwd_tail can take `levels` (1 or 2) levels starting at a W x H level
bool fwd_tail_applies(Wavelet w, int W, int H, int batch, int levels)
{
	if (w != kCdf97S && w != kCdf53I && w != kCdf53S && w != kCdf97I)
		return false;
	const int m = levels == 2 ? 3 : 1;
	// even sizes at every level involved, a few tiles at least, small enough that one round of
	// sweep waves would be the alternative (measured cross-over, single image: 2048^2)
	return (W & m) == 0 && (H & m) == 0 && W >= 64 && H >= 64 && (long)W * H * batch <= (4L << 20) + 1;
}

template <class W>
static hipError_t fwd_tail_t(const FwdLevelArgs &a, int levels, hipStream_t s)
{
	const int Wl = a.W >> levels, Hl = a.H >> levels;
	dim3 grid(((Wl + kTailPairs - 1) / kTailPairs) * ((Hl + kTailPairs - 1) / kTailPairs), a.batch);
	if (levels == 2)
		k_fwd_tail<W, 2><<<grid, 256, 0, s>>>(a);
	else
		k_fwd_tail<W, 1><<<grid, 256, 0, s>>>(a);
	return hipGetLastError();
}

hipError_t launch_fwd_tail(Wavelet w, const FwdLevelArgs &a, int levels, hipStream_t s)
{
	if (levels < 1 || levels > 2 || !fwd_tail_applies(w, a.W, a.H, a.batch, levels))
		return hipErrorInvalidValue;
	switch (w) {
	case kCdf97S: return fwd_tail_t<Cdf97S>(a, levels, s);
	case kCdf53I: return fwd_tail_t<Cdf53I>(a, levels, s);
	case kCdf53S: return fwd_tail_t<Cdf53S>(a, levels, s);
	case kCdf97I: return fwd_tail_t<Cdf97I>(a, levels, s);
	default: break;
	}
	return hipErrorInvalidValue;
}

} // namespace dwt
