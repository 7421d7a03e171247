// dwt_sweep2d_d.hip -- the fused tile sweeps of dwt_sweep2d.hip for the DOUBLE-precision drivers
// (dwt_cdf97_2f_d / _2i_d, src/libdwt.c:12451, :16884; dwt_cdf53_2f_d / _2i_d, :12535, :16962).
//
// Same structure as k_fwd_sweep / k_inv_sweep -- one wave per tile marching down its rows, input
// rows streamed HBM -> LDS by LDS-DMA into a wave-private ring, horizontal lift in registers,
// vertical lift streaming with its state in registers, Mallat de-interleave in registers, 16 B
// per lane stores -- with the byte layout of the float kernels kept: a lane owns 32 B of a row
// (4 doubles where the float sweep has 8 floats; the inverse 16 B = 2 doubles for 4 floats), so
// rows, DMA pieces and stores have the sizes those kernels were tuned for; only the halo is twice
// as many bytes (4 samples = 32 B a side).  Arithmetic order as the reference (rows before
// columns, unfused multiply-add in fp64): bit-identical coefficients.
#include "dwt_device.h"

namespace dwt {

namespace {

static __device__ __forceinline__ double dbl(unsigned lo, unsigned hi)
{
	return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

static __device__ __forceinline__ u4 pack2(double a, double b)
{
	const unsigned long long x = __builtin_bit_cast(unsigned long long, a), y = __builtin_bit_cast(unsigned long long, b);
	return u4{(unsigned)x, (unsigned)(x >> 32), (unsigned)y, (unsigned)(y >> 32)};
}

// six 16 B reads: [a0, a0+16) [a0+16, ..) [a1 ..) [a1+16 ..) [a2 ..) [a2+16 ..)
static __device__ __forceinline__ void lds_read6(unsigned a0, unsigned a1, unsigned a2, u4 (&r)[6])
{
	asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:16\n\t"
	             "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8 offset:16\n\ts_waitcnt lgkmcnt(0)"
		: "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5])
		: "v"(a0), "v"(a1), "v"(a2)
		: "memory");
}

struct SweepGeomD {
	int tile_pairs, ntx, swz, wave_horiz;
};

} // namespace

// ---- forward ---------------------------------------------------------------------------
// LDS row slot (bytes): [main 2048 | left halo 32 | right halo 32]
template <class W, int RING>
static __device__ __forceinline__ void fwd_sweep_d_tile(const FwdLevelArgs &a, const SweepGeomD &g)
{
	using T = double;
	constexpr int K = W::K, CPT = 4, TW = 64 * CPT, RSB = TW * 8 + 64, NARR = CPT + 2 * K;
	constexpr int kAhead = RING / 2 - 1, kDmaPerIter = 2 * 3;
	extern __shared__ __attribute__((aligned(16))) char smem[];

	const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int bid = tile_block_id(g.swz);
	int tx, ty;
	if (g.wave_horiz) {
		const int ntxb = (g.ntx + nwv - 1) / nwv;
		tx = (bid % ntxb) * nwv + wv;
		ty = bid / ntxb;
	} else {
		tx = bid % g.ntx;
		ty = (bid / g.ntx) * nwv + wv;
	}
	const int img = blockIdx.y;
	const int Wd = (a.W + 1) >> 1, Hd = (a.H + 1) >> 1;
	const int A = ty * g.tile_pairs;
	if (A >= Hd || tx >= g.ntx)
		return;
	const int B = min(A + g.tile_pairs, Hd);
	const int c0 = tx * TW;
	const int n_iter = (B - A) + K;
	const int q0 = A - K / 2;

	const T *in = (const T *)a.in + (long)img * a.in_bstride;
	T *out_ll = (T *)a.out_ll + (long)img * a.ll_bstride;
	T *out_h = (T *)a.out_h + (long)img * a.h_bstride;

	char *ring = smem + (size_t)wv * RING * RSB;
	const unsigned ring_off = lds_offset(ring);
	// rows as buffers (see k_fwd_sweep): 16-byte DMAs and stores for every tile, the dwords beyond a
	// row's end zero-filled / dropped by the hardware; the up to four reflected columns right of
	// the image's edge come as 4-byte DMAs (lane l: word l & 1 of column W + (l >> 1))
	const int n_edge = 2 * min(4, c0 + TW - a.W); // lanes of that DMA (<= 0: none)
	// (SelEnds: levels of 64 x 64 and more -- no index is reflected twice, no integer division in the wave's instruction stream)
	constexpr bool kSel = kIsSelEnds<W>;
	const int edge_col = kSel ? a.W - 2 - min(lane >> 1, 3) : reflect(a.W + min(lane >> 1, 3), a.W);
	// halo: lanes 0..7 the four columns left of the tile, 8..15 the four to its right (two words each)
	const int halo_i = lane < 8 ? c0 - 4 + (lane >> 1) : c0 + TW + ((lane >> 1) & 3);
	const int halo_col = kSel ? reflect_near(halo_i, a.W) : reflect(halo_i, a.W);
	const int word = lane & 1;

	int islot = 0, rslot = 0;
	const bool tall = kSel || a.H >= 64;
	// line-end forms (dwt_lift.h; see k_fwd_sweep): the lane's columns c - K .. that are a row's ends, any in this tile,
	// and the test for a row being a column's end
	[[maybe_unused]] unsigned hends = 0;
	[[maybe_unused]] bool h_any = false, h_simple = false;
	constexpr unsigned kCand = (1u << K) | (1u << (K + CPT - 1)); // the two entries that meet a line end when W is a multiple of CPT
	// (SelEnds) the same two entries by selection: the lane's flags and its coefficients for the steps that reach them
	[[maybe_unused]] bool e0 = false, e1 = false;
	[[maybe_unused]] T kh[K];
	if constexpr (kSel) {
		const unsigned m = end_mask_long<NARR>(c0 + lane * CPT - K, a.W);
		e0 = (m >> K) & 1;
		e1 = (m >> (K + CPT - 1)) & 1;
		sel_coefs<W, false, K>(kh, e0, e1);
	} else if constexpr (W::kEndForms) {
		hends = end_mask<NARR>(c0 + lane * CPT - K, a.W);
		h_any = __builtin_amdgcn_ballot_w64(hends != 0) != 0;
		h_simple = __builtin_amdgcn_ballot_w64((hends & ~(kCand | 1u | (1u << (NARR - 1)))) != 0) == 0;
	}
	auto row_is_end = [&](int r) {
		if (tall)
			return r == 0 || r == a.H - 1;
		const int rr = reflect(r, a.H);
		return rr == 0 || rr == a.H - 1;
	};
	auto issue = [&](int it) {
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const int ri = 2 * (q0 + it) - 1 + rr;
			const int r = tall ? reflect1(ri, a.H) : reflect(ri, a.H);
			char *lrow = ring + (size_t)(islot + rr) * RSB;
			const T *grow = in + (long)r * a.in_pitch;
			{
				// (the rows around the tile's upper edge are read again by the tile above at the end of its march:
				// temporal, so that they can wait in the Infinity Cache; every other row is read once.  dwt_sweep2d.hip)
				const row_rsrc_t rs = row_rsrc(grow, (unsigned)a.W * 8);
				if (it < K && A > 0) {
					dma16_row<0>(rs, (unsigned)c0 * 8 + lane * 16, lrow);
					dma16_row<0>(rs, (unsigned)(c0 + 128) * 8 + lane * 16, lrow + 1024);
				} else {
					dma16_row<2>(rs, (unsigned)c0 * 8 + lane * 16, lrow);
					dma16_row<2>(rs, (unsigned)(c0 + 128) * 8 + lane * 16, lrow + 1024);
				}
			}
			if (lane < n_edge)
				dma4<2>((const char *)(grow + edge_col) + 4 * word, lrow + (a.W - c0) * 8);
			if (lane < 16)
				dma4<0>((const char *)(grow + halo_col) + 4 * word, lrow + TW * 8);
		}
		islot = islot + 2 >= RING ? 0 : islot + 2;
	};
	T st[K][CPT];
#pragma unroll
	for (int s = 0; s < K; s++)
#pragma unroll
		for (int v = 0; v < CPT; v++)
			st[s][v] = 0;

	for (int it = 0; it < kAhead && it < n_iter; it++)
		issue(it);

	for (int it = 0; it < n_iter; it++) {
		if (it + kAhead < n_iter) {
			issue(it + kAhead);
			// everything older than the youngest kAhead iterations' DMAs has landed (the element-wise
			// loader issues more per row: the count is then merely conservative)
			DWT_WAIT_VMCNT(kAhead * kDmaPerIter);
		} else {
			DWT_WAIT_VMCNT(0);
		}
		T row[2][CPT];
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const unsigned base = ring_off + (unsigned)(rslot + rr) * RSB;
			const unsigned own = base + lane * CPT * 8;
			const unsigned la = lane == 0 ? base + TW * 8 : own - 32;
			const unsigned ra = lane == 63 ? base + TW * 8 + 32 : own + 32;
			u4 r[6];
			lds_read6(la, own, ra, r);
			T x[NARR];
			// left halo doubles: r[0] = (l0, l1), r[1] = (l2, l3); own r[2], r[3]; right r[4], r[5]
			const T l4[4] = {dbl(r[0][0], r[0][1]), dbl(r[0][2], r[0][3]), dbl(r[1][0], r[1][1]), dbl(r[1][2], r[1][3])};
			const T r4[4] = {dbl(r[4][0], r[4][1]), dbl(r[4][2], r[4][3]), dbl(r[5][0], r[5][1]), dbl(r[5][2], r[5][3])};
#pragma unroll
			for (int e = 0; e < K; e++) {
				x[e] = l4[4 - K + e];
				x[K + CPT + e] = r4[e];
			}
			x[K + 0] = dbl(r[2][0], r[2][1]);
			x[K + 1] = dbl(r[2][2], r[2][3]);
			x[K + 2] = dbl(r[3][0], r[3][1]);
			x[K + 3] = dbl(r[3][2], r[3][3]);
			if constexpr (kSel) {
				lift_regs_sel<W, NARR, false, K, K + CPT - 1>(x, e0, e1, kh);
			} else if (__builtin_expect(!h_any, 1)) {
				lift_fwd_regs<W, NARR>(x, 0u);
			} else if (h_simple) {
				DWT_END_PATH();
				lift_fwd_regs<W, NARR, kCand>(x, hends);
			} else {
				DWT_END_PATH();
				lift_fwd_regs<W, NARR>(x, hends);
			}
#pragma unroll
			for (int v = 0; v < CPT; v++)
				row[rr][v] = W::fwd_scale(v & 1, x[K + v]);
		}
		rslot = rslot + 2 >= RING ? 0 : rslot + 2;

		// step s of this iteration acts on row 2q-1-s: which of them are column ends (wave-uniform; almost never any)
		bool vend[K], v_any = false;
#pragma unroll
		for (int s_ = 0; s_ < K; s_++) {
			vend[s_] = row_is_end(2 * (q0 + it) - 1 - s_);
			v_any = v_any || vend[s_];
		}
		T lo[CPT], hi[CPT];
		// (SelEnds: an end step is the plain step with the coefficient doubled and the state tap dropped)
		[[maybe_unused]] T kv[K];
		if constexpr (kSel) {
#pragma unroll
			for (int s_ = 0; s_ < K; s_++)
				kv[s_] = sel_coef<W, false>(s_, vend[s_]);
		}
		auto vertical = [&](auto ends_tag) {
			constexpr bool ENDS = decltype(ends_tag)::value;
			auto vstep = [&](int s_, T c, T l, T r) {
				if constexpr (kSel && ENDS)
					return sel_step<W, false>(s_, vend[s_], kv[s_], c, l, r);
				else if constexpr (kSel)
					return W::fwd_step(s_, c, l, r);
				else
					return fwd_step_at<W>(s_, ENDS && vend[s_], c, l, r);
			};
#pragma unroll
			for (int v = 0; v < CPT; v++) {
				const T ov = row[0][v], ev = row[1][v];
				if constexpr (K == 4) {
					const T d1n = vstep(0, ov, st[0][v], ev);
					const T s1n = vstep(1, st[0][v], st[1][v], d1n);
					const T d2n = vstep(2, st[1][v], st[2][v], s1n);
					const T s2n = vstep(3, st[2][v], st[3][v], d2n);
					lo[v] = W::fwd_scale(0, s2n);
					hi[v] = W::fwd_scale(1, d2n);
					st[0][v] = ev;
					st[1][v] = d1n;
					st[2][v] = s1n;
					st[3][v] = d2n;
				} else {
					const T d1n = vstep(0, ov, st[0][v], ev);
					const T s1n = vstep(1, st[0][v], st[1][v], d1n);
					lo[v] = W::fwd_scale(0, s1n);
					hi[v] = W::fwd_scale(1, d1n);
					st[0][v] = ev;
					st[1][v] = d1n;
				}
			}
		};
		if (__builtin_expect(v_any, 0)) {
			DWT_END_PATH();
			vertical(std::true_type{});
		}
		else
			vertical(std::false_type{});
		if (it >= K) {
			const int k = A + it - K;
			// each quarter row of the Mallat layout is a buffer of its own
			const unsigned clb = (unsigned)((c0 + lane * CPT) >> 1) * 8;
			const T *top = out_h + (long)k * a.h_pitch, *bot = out_h + (long)(Hd + k) * a.h_pitch;
			const unsigned nlb = (unsigned)Wd * 8, nhb = (unsigned)(a.W >> 1) * 8;
			store16_row<false>(row_rsrc(out_ll + (long)k * a.ll_pitch, nlb), clb, pack2(lo[0], lo[2])); // the next level reads it: temporal
			store16_row<true>(row_rsrc(top + Wd, nhb), clb, pack2(lo[1], lo[3]));
			if (k < (a.H >> 1)) {
				store16_row<true>(row_rsrc(bot, nlb), clb, pack2(hi[0], hi[2]));
				store16_row<true>(row_rsrc(bot + Wd, nhb), clb, pack2(hi[1], hi[3]));
			}
		}
	}
}

// the tile by the instantiation of the policy's line ends the level needs (dwt_lift.h; see fwd_sweep_any_tile)
template <class W, int RING>
__global__ __launch_bounds__(256) void k_fwd_sweep_d(FwdLevelArgs a, SweepGeomD g)
{
	if constexpr (W::kEndForms) {
		if (a.W % 4 == 0 && a.W >= 64 && a.H >= 64)
			fwd_sweep_d_tile<SelEnds<W>, RING>(a, g);
		else
			fwd_sweep_d_tile<W, RING>(a, g);
	} else
		fwd_sweep_d_tile<W, RING>(a, g);
}

template <class W, int RING>
static hipError_t fwd_launch_d(const FwdLevelArgs &a, const SweepGeomD &g, dim3 grid, int waves, hipStream_t s)
{
	const size_t lds = (size_t)waves * RING * (256 * 8 + 64);
	if (hipError_t e = allow_lds((const void *)k_fwd_sweep_d<W, RING>, lds))
		return e;
	k_fwd_sweep_d<W, RING><<<grid, 64 * waves, lds, s>>>(a, g);
	return hipGetLastError();
}

template <class W>
static hipError_t fwd_level_d_t(const FwdLevelArgs &a, const SweepTuning &t, hipStream_t s)
{
	if (a.W < 2 || a.H < 2 || a.batch < 1)
		return hipErrorInvalidValue;
	constexpr int TW = 256;
	SweepGeomD g;
	const int Hd = (a.H + 1) / 2;
	g.ntx = (a.W + TW - 1) / TW;
	// tile heights as the float sweep picks them for the same number of BYTES per row
	int tp = t.tile_pairs > 0 ? t.tile_pairs : 64;
	if (t.tile_pairs <= 0) {
		if ((long)a.W * a.H * a.batch <= (2L << 20))
			tp = 4;
		else
			while (tp > 8 && (long)g.ntx * ((Hd + tp - 1) / tp) * a.batch < 1024)
				tp >>= 1;
	}
	g.tile_pairs = tp;
	g.swz = t.xcd_swizzle;
	const int waves = t.waves >= 1 && t.waves <= 4 ? t.waves : 4;
	const int nty = (Hd + tp - 1) / tp;
	const int ring = (t.ring == 8 || t.ring == 16) ? t.ring : ((g.ntx >= waves && (long)g.ntx * nty * a.batch >= 3072) ? 16 : 8);
	g.wave_horiz = ring == 16;
	dim3 grid;
	if (g.wave_horiz)
		grid = dim3(((g.ntx + waves - 1) / waves) * nty, a.batch);
	else
		grid = dim3(g.ntx * ((nty + waves - 1) / waves), a.batch);
	return ring == 16 ? fwd_launch_d<W, 16>(a, g, grid, waves, s) : fwd_launch_d<W, 8>(a, g, grid, waves, s);
}

hipError_t launch_fwd_level_d(Wavelet w, const FwdLevelArgs &a, const SweepTuning &t, hipStream_t s)
{
	switch (w) {
	case kCdf97D: return fwd_level_d_t<Cdf97D>(a, t, s);
	case kCdf53D: return fwd_level_d_t<Cdf53D>(a, t, s);
	default: break;
	}
	return hipErrorInvalidValue;
}


// ---- inverse ---------------------------------------------------------------------------
// A lane owns 2 output columns (16 B per row and lane, one contiguous store); a tile is 128 columns.
// Source rows are Mallat rows: "L row p" = [LL | HL], "H row p" = [LH | HH].  LDS row slot (bytes):
// [L main 512 | H main 512 | L halo 64 | H halo 64]; a halo block is [4 columns left of the tile |
// 4 columns right of it].  A lane needs the 5 subband columns around its own of each half.
template <class W, int RING>
static __device__ __forceinline__ void inv_sweep_d_tile(const InvLevelArgs &a, const SweepGeomD &g)
{
	using T = double;
	constexpr int K = W::K, CPT = 2, TW = 64 * CPT, M = TW / 2, RSB = 2 * M * 8 + 128, NARR = CPT + 2 * K - 1;
	constexpr int kAhead = RING / 2 - 1, kDmaPerIter = 2 * 2;
	extern __shared__ __attribute__((aligned(16))) char smem[];

	const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int bid = tile_block_id(g.swz);
	int tx, ty;
	if (g.wave_horiz) {
		const int ntxb = (g.ntx + nwv - 1) / nwv;
		tx = (bid % ntxb) * nwv + wv;
		ty = bid / ntxb;
	} else {
		tx = bid % g.ntx;
		ty = (bid / g.ntx) * nwv + wv;
	}
	const int img = blockIdx.y;
	const int Wd = (a.W + 1) >> 1, Hd = (a.H + 1) >> 1;
	const int A = ty * g.tile_pairs;
	if (A >= Hd || tx >= g.ntx)
		return;
	const int B = min(A + g.tile_pairs, Hd);
	const int c0 = tx * TW, cl0 = c0 >> 1;
	const int n_iter = (B - A) + K;
	const int p0 = A - K / 2;

	const T *in_ll = (const T *)a.in_ll + (long)img * a.ll_bstride;
	const T *in_h = (const T *)a.in_h + (long)img * a.h_bstride;
	T *out = (T *)a.out + (long)img * a.out_bstride;

	char *ring = smem + (size_t)wv * RING * RSB;
	const unsigned ring_off = lds_offset(ring);
	const int word = lane & 1;
	// Whole tiles fetch their two subband segments as one plain 16-byte DMA; the tile that holds the
	// image's right edge addresses them as buffers (see k_inv_sweep) and fetches the two L and two
	// H columns right of the edge by reflection: lanes 0..3 the words of L columns nL, nL + 1,
	// lanes 4..7 those of H columns nH, nH + 1.
	const bool edge_tile = c0 + TW > a.W;
	const int nL = Wd, nH = a.W >> 1;
	const int edge_sub = (lane & 4) ? nH + ((lane >> 1) & 1) : nL + ((lane >> 1) & 1);
	constexpr bool kSel = kIsSelEnds<W>; // (levels of 64 x 64 and more: one-bounce reflections, no integer division in the stream)
	const int edge_col = (kSel ? reflect_near(2 * edge_sub + ((lane >> 2) & 1), a.W) : reflect(2 * edge_sub + ((lane >> 2) & 1), a.W)) >> 1;
	// halo: lanes 0..15 the L halo block (8 doubles), 16..31 the H halo block
	const int hd = (lane >> 1) & 7, hs = (lane >> 4) & 1;
	const int hsub = hd < 4 ? cl0 - 4 + hd : cl0 + M + (hd - 4);
	const int halo_col = (kSel ? reflect_near(2 * hsub + hs, a.W) : reflect(2 * hsub + hs, a.W)) >> 1;
	// line-end forms (dwt_lift.h; see k_inv_sweep): the lane's samples c - K + 1 .. that are a row's ends, any in this
	// tile, and the test for a row being a column's end
	[[maybe_unused]] unsigned hends = 0;
	[[maybe_unused]] bool h_any = false, h_simple = false;
	constexpr unsigned kCand = (1u << (K - 1)) | (1u << (K + CPT - 2)); // the two entries that meet a line end when W is a multiple of CPT
	[[maybe_unused]] bool e0 = false, e1 = false;
	[[maybe_unused]] T kh[K];
	if constexpr (kSel) {
		const unsigned m = end_mask_long<NARR>(c0 + lane * CPT - K + 1, a.W);
		e0 = (m >> (K - 1)) & 1;
		e1 = (m >> (K + CPT - 2)) & 1;
		sel_coefs<W, true, K - 1>(kh, e0, e1);
	} else if constexpr (W::kEndForms) {
		hends = end_mask<NARR>(c0 + lane * CPT - K + 1, a.W);
		h_any = __builtin_amdgcn_ballot_w64(hends != 0) != 0;
		h_simple = __builtin_amdgcn_ballot_w64((hends & ~(kCand | 1u | (1u << (NARR - 1)))) != 0) == 0;
	}
	const bool tall = kSel || a.H >= 64;
	auto row_is_end = [&](int r) {
		if (tall)
			return r == 0 || r == a.H - 1;
		const int rr = reflect(r, a.H);
		return rr == 0 || rr == a.H - 1;
	};

	auto issue = [&](int it) {
		const int p = p0 + it;
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const int rs = tall ? reflect1(2 * p + rr, a.H) : reflect(2 * p + rr, a.H);
			const int sub = rs >> 1;
			const T *gl, *gh;
			if (rr == 0) {
				gl = in_ll + (long)sub * a.ll_pitch;
				gh = in_h + (long)sub * a.h_pitch + Wd;
			} else {
				gl = in_h + (long)(Hd + sub) * a.h_pitch;
				gh = gl + Wd;
			}
			char *lrow = ring + (size_t)((2 * it + rr) & (RING - 1)) * RSB;
			if (!edge_tile) {
				// lanes 0..31 fetch the L segment, 32..63 the H segment (16 B = 2 doubles each)
				const T *gsel = lane < 32 ? gl : gh;
				dma16<0>(gsel + cl0 + (lane & 31) * 2, lrow);
			} else {
				if (lane < 32)
					dma16_row<0>(row_rsrc(gl, (unsigned)nL * 8), (unsigned)cl0 * 8 + lane * 16, lrow);
				else
					dma16_row<0>(row_rsrc(gh, (unsigned)nH * 8), (unsigned)cl0 * 8 + (lane & 31) * 16, lrow); // lane's slot lrow + 16 lane = the H half
				if (lane < 4 && nL + (lane >> 1) < cl0 + M)
					dma4<0>((const char *)(gl + edge_col) + 4 * word, lrow + (nL - cl0) * 8);
				if (lane >= 4 && lane < 8 && nH + ((lane >> 1) & 1) < cl0 + M)
					dma4<0>((const char *)(gh + edge_col) + 4 * word, lrow + M * 8 + (nH - cl0) * 8 - 16);
			}
			if (lane < 32)
				dma4<0>((const char *)((hs ? gh : gl) + halo_col) + 4 * word, lrow + 2 * M * 8);
		}
	};

	T st[K][CPT];
#pragma unroll
	for (int s = 0; s < K; s++)
#pragma unroll
		for (int v = 0; v < CPT; v++)
			st[s][v] = 0;

	for (int it = 0; it < kAhead && it < n_iter; it++)
		issue(it);

	// LDS addresses of the 5 subband columns around this lane's own, relative to a row slot
	unsigned offL[5], offH[5];
#pragma unroll
	for (int t = 0; t < 5; t++) {
		const int p = lane + t - 2;
		const bool halo = p < 0 || p >= M;
		offL[t] = p < 0 ? 2 * M * 8 + (4 + p) * 8 : p >= M ? 2 * M * 8 + 32 + (p - M) * 8 : p * 8;
		offH[t] = offL[t] + (halo ? 64 : M * 8);
	}

	for (int it = 0; it < n_iter; it++) {
		if (it + kAhead < n_iter) {
			issue(it + kAhead);
			DWT_WAIT_VMCNT(kAhead * kDmaPerIter);
		} else {
			DWT_WAIT_VMCNT(0);
		}
		const int p = p0 + it;
		T x[2][NARR];
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const unsigned base = ring_off + (unsigned)((2 * it + rr) & (RING - 1)) * RSB;
			u2 lq[5], hq[5];
			asm volatile("ds_read_b64 %0, %10\n\tds_read_b64 %1, %11\n\tds_read_b64 %2, %12\n\tds_read_b64 %3, %13\n\tds_read_b64 %4, %14\n\t"
			             "ds_read_b64 %5, %15\n\tds_read_b64 %6, %16\n\tds_read_b64 %7, %17\n\tds_read_b64 %8, %18\n\tds_read_b64 %9, %19\n\t"
			             "s_waitcnt lgkmcnt(0)"
				: "=&v"(lq[0]), "=&v"(lq[1]), "=&v"(lq[2]), "=&v"(lq[3]), "=&v"(lq[4]),
				  "=&v"(hq[0]), "=&v"(hq[1]), "=&v"(hq[2]), "=&v"(hq[3]), "=&v"(hq[4])
				: "v"(base + offL[0]), "v"(base + offL[1]), "v"(base + offL[2]), "v"(base + offL[3]), "v"(base + offL[4]),
				  "v"(base + offH[0]), "v"(base + offH[1]), "v"(base + offH[2]), "v"(base + offH[3]), "v"(base + offH[4])
				: "memory");
			T l[5], h[5];
#pragma unroll
			for (int t = 0; t < 5; t++) {
				l[t] = dbl(lq[t][0], lq[t][1]);
				h[t] = dbl(hq[t][0], hq[t][1]);
			}
			// x[j] <-> interleaved sample c - K + 1 + j (x[0] odd); sample i: even -> L[i/2], odd -> H[i/2]
#pragma unroll
			for (int j = 0; j < NARR; j++) {
				const int rel = j - K + 1;
				if (rel & 1)
					x[rr][j] = W::inv_scale(1, h[2 + ((rel - 1) >> 1)]);
				else
					x[rr][j] = W::inv_scale(0, l[2 + (rel >> 1)]);
			}
		}
		T val[2][CPT];
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			if constexpr (kSel) {
				lift_regs_sel<W, NARR, true, K - 1, K + CPT - 2>(x[rr], e0, e1, kh);
			} else if (__builtin_expect(!h_any, 1)) {
				lift_inv_regs<W, NARR>(x[rr], 0u);
			} else if (h_simple) {
				DWT_END_PATH();
				lift_inv_regs<W, NARR, kCand>(x[rr], hends);
			} else {
				DWT_END_PATH();
				lift_inv_regs<W, NARR>(x[rr], hends);
			}
#pragma unroll
			for (int v = 0; v < CPT; v++)
				val[rr][v] = W::inv_scale(rr, x[rr][K - 1 + v]);
		}
		// step s of this iteration acts on row 2p-s: which of them are column ends (wave-uniform; almost never any)
		bool vend[K], v_any = false;
#pragma unroll
		for (int s_ = 0; s_ < K; s_++) {
			vend[s_] = row_is_end(2 * p - s_);
			v_any = v_any || vend[s_];
		}
		T odd_row[CPT], even_row[CPT];
		[[maybe_unused]] T kv[K];
		if constexpr (kSel) {
#pragma unroll
			for (int s_ = 0; s_ < K; s_++)
				kv[s_] = sel_coef<W, true>(s_, vend[s_]);
		}
		auto vertical = [&](auto ends_tag) {
			constexpr bool ENDS = decltype(ends_tag)::value;
			auto vstep = [&](int s_, T c, T l, T r) {
				if constexpr (kSel && ENDS)
					return sel_step<W, true>(s_, vend[s_], kv[s_], c, l, r);
				else if constexpr (kSel)
					return W::inv_step(s_, c, l, r);
				else
					return inv_step_at<W>(s_, ENDS && vend[s_], c, l, r);
			};
#pragma unroll
			for (int v = 0; v < CPT; v++) {
				const T s2 = val[0][v], d2 = val[1][v];
				if constexpr (K == 4) {
					const T s1n = vstep(0, s2, st[0][v], d2);
					const T d1n = vstep(1, st[0][v], st[1][v], s1n);
					const T en = vstep(2, st[1][v], st[2][v], d1n);
					const T on = vstep(3, st[2][v], st[3][v], en);
					odd_row[v] = on;
					even_row[v] = en;
					st[0][v] = d2;
					st[1][v] = s1n;
					st[2][v] = d1n;
					st[3][v] = en;
				} else {
					const T en = vstep(0, s2, st[0][v], d2);
					const T on = vstep(1, st[0][v], st[1][v], en);
					odd_row[v] = on;
					even_row[v] = en;
					st[0][v] = d2;
					st[1][v] = en;
				}
			}
		};
		if (__builtin_expect(v_any, 0)) {
			DWT_END_PATH();
			vertical(std::true_type{});
		}
		else
			vertical(std::false_type{});
		const int pe = (K == 4) ? p - 1 : p;
		const int po = (K == 4) ? p - 2 : p - 1;
		const bool ve = pe >= A && pe < B;
		const bool vo = po >= A && po < B && (2 * po + 1 < a.H);
		const unsigned cb = (unsigned)(c0 + lane * CPT) * 8;
		if (vo)
			store16_row<true>(row_rsrc(out + (long)(2 * po + 1) * a.out_pitch, (unsigned)a.W * 8), cb, pack2(odd_row[0], odd_row[1]));
		if (ve)
			store16_row<true>(row_rsrc(out + (long)(2 * pe) * a.out_pitch, (unsigned)a.W * 8), cb, pack2(even_row[0], even_row[1]));
	}
}

template <class W, int RING>
__global__ __launch_bounds__(256) void k_inv_sweep_d(InvLevelArgs a, SweepGeomD g)
{
	if constexpr (W::kEndForms) {
		if (a.W % 2 == 0 && a.W >= 64 && a.H >= 64)
			inv_sweep_d_tile<SelEnds<W>, RING>(a, g);
		else
			inv_sweep_d_tile<W, RING>(a, g);
	} else
		inv_sweep_d_tile<W, RING>(a, g);
}

template <class W, int RING>
static hipError_t inv_launch_d(const InvLevelArgs &a, const SweepGeomD &g, dim3 grid, int waves, hipStream_t s)
{
	const size_t lds = (size_t)waves * RING * (2 * 64 * 8 + 128);
	if (hipError_t e = allow_lds((const void *)k_inv_sweep_d<W, RING>, lds))
		return e;
	k_inv_sweep_d<W, RING><<<grid, 64 * waves, lds, s>>>(a, g);
	return hipGetLastError();
}

template <class W>
static hipError_t inv_level_d_t(const InvLevelArgs &a, const SweepTuning &t, hipStream_t s)
{
	if (a.W < 2 || a.H < 2 || a.batch < 1)
		return hipErrorInvalidValue;
	constexpr int TW = 128;
	SweepGeomD g;
	const int Hd = (a.H + 1) / 2;
	g.ntx = (a.W + TW - 1) / TW;
	int tp = t.tile_pairs > 0 ? t.tile_pairs : 32;
	if (t.tile_pairs <= 0) {
		if ((long)a.W * a.H * a.batch <= (2L << 20))
			tp = 4;
		else
			while (tp > 8 && (long)g.ntx * ((Hd + tp - 1) / tp) * a.batch < 2048)
				tp >>= 1;
	}
	g.tile_pairs = tp;
	g.swz = t.xcd_swizzle;
	const int waves = t.waves >= 1 && t.waves <= 4 ? t.waves : 4;
	const int nty = (Hd + tp - 1) / tp;
	g.wave_horiz = 0;
	dim3 grid;
	if (g.wave_horiz)
		grid = dim3(((g.ntx + waves - 1) / waves) * nty, a.batch);
	else
		grid = dim3(g.ntx * ((nty + waves - 1) / waves), a.batch);
	return t.ring_inv == 16 ? inv_launch_d<W, 16>(a, g, grid, waves, s) : inv_launch_d<W, 8>(a, g, grid, waves, s);
}

hipError_t launch_inv_level_d(Wavelet w, const InvLevelArgs &a, const SweepTuning &t, hipStream_t s)
{
	switch (w) {
	case kCdf97D: return inv_level_d_t<Cdf97D>(a, t, s);
	case kCdf53D: return inv_level_d_t<Cdf53D>(a, t, s);
	default: break;
	}
	return hipErrorInvalidValue;
}

} // namespace dwt
