// dwt_sweep2d.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the 2-D lifting DWT.
//
// Two families:
//
//  1. k_line_pass: a generic out-of-place 1-D pass, one thread per output pair,
//     exact reference semantics for any line length.  Used for sparse frames
//     (size_o != size_i), single-line directions and as a cross-check variant.
//     It restates dwt_cdf97_f_ex_stride_s / _i_ex_stride_s and the 5/3 siblings
//     (src/libdwt.c:10744, 11530, 10950, 11749, 10986, 11785).
//
//  2. k_fwd_sweep / k_inv_sweep: one decomposition level of the 2-D drivers
//     (src/libdwt.c:12812-12919 forward, :17074-17176 inverse, :16334-16383,
//     :18165-18215 for 5/3) fused into a single tile sweep.  Each WAVE owns a tile
//     of 64*CPT columns and marches down its rows:
//       - input rows are streamed HBM -> LDS by asynchronous LDS-DMA
//         (global_load_lds) into a wave-private ring, several rows ahead, counted
//         with s_waitcnt vmcnt(N) -- no barriers, no VGPR staging;
//       - each lane reads its columns plus the 4-sample halo from LDS and lifts
//         them horizontally in registers;
//       - the vertical lifting state (4 partial rows for 9/7, 2 for 5/3) stays in
//         registers for the whole sweep, so every input sample is read from HBM
//         once per level and every coefficient written once;
//       - the Mallat de-interleave is done in registers, each subband row leaving
//         the wave as one contiguous 16 B/lane store.
//     Image borders use whole-sample symmetric reflection applied to the SOURCE
//     address of the DMA, so the arithmetic needs no edge cases.
//
// Arithmetic order follows the reference exactly (rows before columns, etc.) and
// this file is compiled with -ffp-contract=off, so float results are bit-identical
// to libdwt's CPU path; int results are exact.
#include "dwt_sweep2d.h"

namespace dwt {

// ---------------------------------------------------------------------------------
// 1. generic exact line pass
// ---------------------------------------------------------------------------------
template <class W, bool INV>
__global__ __launch_bounds__(256) void k_line_pass(const char *__restrict__ src, char *__restrict__ dst,
	long line_stride, long elem_stride, int n_lines, int N, int hoff, int lanes_along_lines)
{
	using T = typename W::T;
	constexpr int K = W::K;
	const int fast = blockIdx.x * blockDim.x + threadIdx.x;
	const int slow = blockIdx.y;
	const int line = lanes_along_lines ? fast : slow;
	const int k = lanes_along_lines ? slow : fast;
	const int npairs = (N + 1) >> 1;
	if (line >= n_lines || k >= npairs)
		return;
	const char *s = src + (long)line * line_stride;
	char *d = dst + (long)line * line_stride;
	auto ld = [&](int idx) { return *(const T *)(s + (long)idx * elem_stride); };
	auto st = [&](int idx, T v) { *(T *)(d + (long)idx * elem_stride) = v; };
	const bool il = hoff < 0; // interleaved layout on both sides: L_k at 2k, H_k at 2k+1

	if (N == 1) {
		// the float kernels scale a lone sample, the int kernel leaves it
		if (W::kScaleSingle)
			st(0, INV ? W::inv_single(ld(0)) : W::fwd_single(ld(0)));
		return;
	}
	T w[2 * K + 1];
	if (!INV) {
		// w[j] = a[2k-K+j]; w[0] is an even sample
#pragma unroll
		for (int j = 0; j <= 2 * K; j++)
			w[j] = ld(reflect(2 * k - K + j, N));
		lift_fwd_regs<W, 2 * K + 1>(w, W::kEndForms ? end_mask<2 * K + 1>(2 * k - K, N) : 0u);
		st(il ? 2 * k : k, W::fwd_scale(0, w[K]));
		if (2 * k + 1 < N)
			st(il ? 2 * k + 1 : hoff + k, W::fwd_scale(1, w[K + 1]));
	} else {
		// w[j] = a[2k-K+1+j] of the interleaved signal; w[0] is an odd sample
#pragma unroll
		for (int j = 0; j <= 2 * K; j++) {
			const int i = reflect(2 * k - K + 1 + j, N);
			const T raw = il ? ld(i) : (i & 1) ? ld(hoff + (i >> 1)) : ld(i >> 1);
			w[j] = W::inv_scale(i & 1, raw);
		}
		lift_inv_regs<W, 2 * K + 1>(w, W::kEndForms ? end_mask<2 * K + 1>(2 * k - K + 1, N) : 0u);
		st(2 * k, w[K - 1]);
		if (2 * k + 1 < N)
			st(2 * k + 1, w[K]);
	}
}

template <class W>
static hipError_t line_pass_t(bool inverse, const void *src, void *dst, long line_stride, long elem_stride,
	int n_lines, int N, int hoff, bool lanes_along_lines, hipStream_t s)
{
	if (n_lines <= 0 || N <= 0)
		return hipSuccess;
	const int npairs = (N + 1) >> 1;
	const int fast = lanes_along_lines ? n_lines : npairs;
	const int slow = lanes_along_lines ? npairs : n_lines;
	const int bs = fast >= 256 ? 256 : 64;
	dim3 grid((fast + bs - 1) / bs, slow);
	if (inverse)
		k_line_pass<W, true><<<grid, bs, 0, s>>>((const char *)src, (char *)dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines);
	else
		k_line_pass<W, false><<<grid, bs, 0, s>>>((const char *)src, (char *)dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines);
	return hipGetLastError();
}

hipError_t launch_line_pass(Wavelet w, bool inverse, const void *src, void *dst, long line_stride, long elem_stride,
	int n_lines, int N, int hoff, bool lanes_along_lines, hipStream_t s)
{
	switch (w) {
	case kCdf97S: return line_pass_t<Cdf97S>(inverse, src, dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines, s);
	case kCdf53I: return line_pass_t<Cdf53I>(inverse, src, dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines, s);
	case kCdf53S: return line_pass_t<Cdf53S>(inverse, src, dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines, s);
	case kCdf97D: return line_pass_t<Cdf97D>(inverse, src, dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines, s);
	case kCdf53D: return line_pass_t<Cdf53D>(inverse, src, dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines, s);
	case kCdf97I: return line_pass_t<Cdf97I>(inverse, src, dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines, s);
	case kCdf53SNew: return line_pass_t<Cdf53SNew>(inverse, src, dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines, s);
	case kCdf97IIp: return line_pass_t<Cdf97IIp>(inverse, src, dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines, s);
	// the contracted variant exists for the fused sweeps only: line passes of such a call are exact
	case kCdf97SFma: return line_pass_t<Cdf97S>(inverse, src, dst, line_stride, elem_stride, n_lines, N, hoff, lanes_along_lines, s);
	}
	return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------
// 2. fused tile sweeps
// ---------------------------------------------------------------------------------
// ---- forward -------------------------------------------------------------------
// IL: write the result INTERLEAVED in place of the Mallat de-interleave (row 2k = L
// row, row 2k+1 = H row, columns interleaved alike) to `out_h`: the layout of the
// 3-D path (src/volume-dwt.c:677-725), where a batch is the slices of a volume.
// X (interleaved layout, phase-ordered wavelets): the tiles leave out the samples whose rounding depends on the
// reference's phase order -- rows 0..7 and the last 8 columns of the level -- for the strip workgroups of the
// same launch (dwt_il_strip.h).
// The vertical pass of one iteration (no row of it a column end) on two adjacent columns at once, as the halves of packed
// operations: the same steps and rounding as W::fwd_step / fwd_scale (float policies with fk: c + k (l + r)).
// row[0] / row[1]: the odd / even row of the iteration; st: the streaming state; lo / hi: the scaled outputs.
// SEL: the select form of the line ends -- ve[s]: step s acts on a row that is a column's end (wave-uniform), kv[s] its
// coefficient (doubled there); the state tap gives way to -0.0 (dwt_lift.h, SelEnds).
template <class W, int CPT, bool SEL = false, class T>
static __device__ __forceinline__ void vertical_pairs(const T (&row)[2][CPT], T (&st)[W::K][CPT], T (&lo)[CPT], T (&hi)[CPT],
	const bool *ve = nullptr, const T *kv = nullptr)
{
	typedef float f2 __attribute__((ext_vector_type(2)));
	constexpr int K = W::K;
	[[maybe_unused]] const f2 nz = f2{-0.0f, -0.0f};
	auto kk = [&](int s) { return SEL ? kv[s] : W::fk(s); };
	auto tap = [&](int s, f2 v) { return (SEL && ve[s]) ? nz : v; };
	const float zl = W::fwd_scale(0, 1.0f), zh = W::fwd_scale(1, 1.0f); // (the scale factors themselves)
#pragma unroll
	for (int v = 0; v < CPT; v++) {
		if (v & 2)
			continue; // (columns v and v + 2: the stores take lo[0], lo[2], lo[4], lo[6] / lo[1], lo[3], ... as consecutive registers)
		constexpr int P = 2;
		const f2 ov = f2{row[0][v], row[0][v + P]}, ev = f2{row[1][v], row[1][v + P]};
		f2 s_[K], n_[K], lo2, hi2;
#pragma unroll
		for (int i = 0; i < K; i++)
			s_[i] = f2{st[i][v], st[i][v + P]};
		n_[0] = ev;
		n_[1] = ov + kk(0) * (tap(0, s_[0]) + ev); // d1n
		if constexpr (K == 4) {
			n_[2] = s_[0] + kk(1) * (tap(1, s_[1]) + n_[1]); // s1n
			n_[3] = s_[1] + kk(2) * (tap(2, s_[2]) + n_[2]); // d2n
			const f2 s2n = s_[2] + kk(3) * (tap(3, s_[3]) + n_[3]);
			lo2 = s2n * zl;
			hi2 = n_[3] * zh;
		} else {
			const f2 s1n = s_[0] + kk(1) * (tap(1, s_[1]) + n_[1]);
			lo2 = s1n * zl;
			hi2 = n_[1] * zh;
		}
#pragma unroll
		for (int i = 0; i < K; i++) {
			st[i][v] = n_[i][0];
			st[i][v + P] = n_[i][1];
		}
		lo[v] = lo2[0];
		lo[v + P] = lo2[1];
		hi[v] = hi2[0];
		hi[v + P] = hi2[1];
	}
}

template <class W, int CPT, int RING, int NT, bool IL, bool X>
static __device__ __forceinline__ void fwd_sweep_tile(const FwdLevelArgs &a, const SweepGeom &g)
{
	using T = typename W::T;
	constexpr int K = W::K;
	constexpr int kRing = RING;           // ring rows per wave (any even number)
	constexpr int kAhead = kRing / 2 - 1; // sweep iterations of DMA lookahead (2 rows each)
	constexpr int kLdAux = (NT & 2) ? 2 : 0;
	constexpr bool kNtStore = (NT & 1) != 0;
	// the LL band is read again by the next level: bit 2 keeps its stores temporal so it
	// can stay in L2 / Infinity Cache
	[[maybe_unused]] constexpr bool kNtStoreLL = kNtStore && !(NT & 4);
	// bit 3: take the 4-sample neighbour taps from the adjacent lanes' registers with
	// wavefront shifts (DPP) instead of re-reading them from LDS
	constexpr bool kShuffle = (NT & 8) != 0;
	// bit 4: TIMING PROBE of a fused level 0 + 1 (FwdLevelArgs::probe_fuse1): wrong results
	[[maybe_unused]] constexpr bool kProbe = DWT_PROBES && (NT & 16) != 0;
	constexpr int TW = 64 * CPT;
	constexpr int RS = TW + 8; // LDS row slot: [main TW | left halo 4 | right halo 4]
	constexpr int NARR = CPT + 2 * K;
	constexpr int kDmaPerIter = 2 * (CPT / 4 + 1); // fewest DMA instructions an iteration issues
	// (float policies whose step is c + k (l + r) rounded product-then-sum: the horizontal lift takes both rows at once)
	constexpr bool kPairRows = kIsSelEnds<W> && std::is_same<T, float>::value && has_coef_ends<W>::value && !std::is_base_of<Cdf97SFma, W>::value;
	extern __shared__ __attribute__((aligned(16))) char smem[];

	const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
	// wave-uniform on purpose: tile geometry, row indices and row pointers then live in SGPRs
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int bid = tile_block_id(g.swz, X ? g.first : 0, g.tile_blocks);
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
		return; // whole wave leaves; no barriers are used anywhere
	if (a.pair_hi > 0 && (A < a.pair_lo || A >= a.pair_hi))
		return; // this launch computes a band of the level only
	const int B = min(A + g.tile_pairs, Hd);
	// (probe level 2: what a REAL fused level 0 + 1 adds to the geometry -- level 1's warm-up of 4 LL row pairs = 8 more
	// level-0 pairs above every tile, and tiles 496 columns apart whose outer lanes recompute the neighbours' 8 columns)
	// (2: both, 3: the warm-up alone, 4: the tile pitch alone)
	[[maybe_unused]] const int pw = (kProbe && (a.probe_fuse1 == 2 || a.probe_fuse1 == 3)) ? 8 : 0;
	const int c0 = (kProbe && (a.probe_fuse1 == 2 || a.probe_fuse1 == 4)) ? tx * (TW - 16) - 8 : tx * TW;
	const int n_iter = (B - A) + K + pw;
	const int q0 = A - K / 2 - pw;

	const T *in = (const T *)a.in + (long)img * a.in_bstride;
	T *out_ll = (T *)a.out_ll + (long)img * a.ll_bstride;
	T *out_h = (T *)a.out_h + (long)img * a.h_bstride;

	char *ring = smem + (size_t)wv * kRing * RS * 4;
	const unsigned ring_off = lds_offset(ring);

	// Rows are addressed as BUFFERS (a descriptor per row in scalar registers, per-lane byte
	// offsets): the hardware checks every dword against the row's length, zero-fills loads and
	// drops stores beyond it, and 16-byte accesses need only 4-byte alignment (probed:
	// scripts/archive/probes/buf_probe.hip).  So every tile -- overhanging the image or not, rows aligned
	// or not -- takes the same 16-byte path; the up to four reflected columns right of the edge
	// that a valid output can reach come by one 4-byte DMA per row.
	const int n_edge = min(4, c0 + TW - a.W); // columns of this tile's main block right of the edge (<= 0: none)
	// (SelEnds: levels of 64 x 64 and more -- no index is reflected twice, no integer division in the wave's instruction stream)
	const int edge_col = kIsSelEnds<W> ? a.W - 2 - min(lane, 3) : reflect(a.W + min(lane, 3), a.W);
	const int halo_i = lane < 4 ? c0 - 4 + lane : c0 + TW + (lane & 3);
	const int halo_col = kIsSelEnds<W> ? reflect_near(halo_i, a.W) : reflect(halo_i, a.W);

	int islot = 0, rslot = 0; // ring slots of the next rows to fill / to consume
	const bool tall = kIsSelEnds<W> || a.H >= 64; // then a row index leaves [0,H) by less than H: one bounce
	// In place (interleaved layout, a.sh): this tile's own rows and columns come from the image -- its stores trail its
	// loads --, everything else from the snapshot: a neighbour's row as a whole from the row shell, a neighbour's
	// column of an own row from the column shell (offset in that shell's row, or -1: an own column)
	[[maybe_unused]] const bool shl = IL && a.sh.rows != nullptr;
	[[maybe_unused]] int halo_sh = -1;
	if (IL && shl)
		halo_sh = (halo_col >= c0 - 4 && halo_col < c0) ? 8 * (tx - 1) + (halo_col - (c0 - 4))
			: (halo_col >= c0 + TW && halo_col < c0 + TW + 4) ? 8 * tx + 4 + (halo_col - (c0 + TW)) : -1;
	auto issue = [&](int it) {
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const int ri = 2 * (q0 + it) - 1 + rr;
			const int r = tall ? reflect1(ri, a.H) : reflect(ri, a.H);
			char *lrow = ring + (size_t)(islot + rr) * RS * 4;
			const T *grow = in + (long)r * a.in_pitch;
			[[maybe_unused]] const T *hsrc = grow + halo_col; // lanes 0..7: the halo column
			// (in place) an own row read from the image: its last 8 columns -- the border strip waves of the launch may
			// have written theirs already -- come from the snapshot `rgt + column`
			[[maybe_unused]] const T *rgt = nullptr;
			if (IL && shl) {
				if (r < 2 * A)
					grow = (const T *)a.sh.rows + (long)(9 * (ty - 1) + r - (2 * A - 5)) * a.sh.rows_pitch, hsrc = grow + halo_col;
				else if (r >= 2 * B)
					grow = (const T *)a.sh.rows + (long)(9 * ty + r - (2 * B - 5)) * a.sh.rows_pitch, hsrc = grow + halo_col;
				else if (r < kIlKeepTop) // the strips' rows: the snapshot of rows 0..13
					grow = (const T *)a.sh.top + (long)r * a.sh.top_pitch, hsrc = grow + halo_col;
				else {
					rgt = (const T *)a.sh.right + (long)r * a.sh.right_pitch - a.sh.right_x0;
					if (halo_sh >= 0)
						hsrc = (const T *)a.sh.cols + (long)r * a.sh.cols_pitch + halo_sh;
					else if (halo_col >= a.W - kIlKeepRight)
						hsrc = rgt + halo_col;
				}
			}
			[[maybe_unused]] const T *esrc = (IL && rgt) ? rgt + edge_col : grow + edge_col; // lanes < n_edge: a reflected column
			const bool fix_right = IL && rgt && c0 + TW > a.W - kIlKeepRight; // this tile holds the last 8 columns
			const row_rsrc_t rs = row_rsrc(grow, (unsigned)a.W * 4);
			// The 8 rows around a tile's upper edge are read twice: by this tile now, in its warm-up, and by the
			// tile above at the end of its march.  The first read is TEMPORAL whatever the policy, so that the
			// lines can wait in the Infinity Cache for the second one (every other row is read once).  Round 4,
			// 32 x 8192^2, one placement: level 0 6108 -> 6240 GB/s; with the halo columns temporal too 6285.
			if (kLdAux != 0 && it < K && A > 0) {
#pragma unroll
				for (int i = 0; i < CPT / 4; i++)
					dma16_row<0>(rs, (unsigned)(c0 + i * 256 + lane * 4) * 4, lrow + i * 1024);
				if (fix_right && lane < kIlKeepRight) // (lands after the row's own DMA: loads return in order)
					dma4<0>(rgt + (a.W - kIlKeepRight + lane), lrow + (a.W - kIlKeepRight - c0) * 4);
				if (lane < n_edge)
					dma4<0>(esrc, lrow + (a.W - c0) * 4);
				if (lane < 8)
					dma4<0>(IL ? hsrc : grow + halo_col, lrow + TW * 4);
			} else {
#pragma unroll
				for (int i = 0; i < CPT / 4; i++)
					dma16_row<kLdAux>(rs, (unsigned)(c0 + i * 256 + lane * 4) * 4, lrow + i * 1024);
				if (fix_right && lane < kIlKeepRight)
					dma4<0>(rgt + (a.W - kIlKeepRight + lane), lrow + (a.W - kIlKeepRight - c0) * 4);
				if (lane < n_edge)
					dma4<kLdAux>(esrc, lrow + (a.W - c0) * 4);
				// (the halo columns are the x-neighbour tiles' own lines, read by them at about the same time: temporal)
				if (lane < 8)
					dma4<0>(IL ? hsrc : grow + halo_col, lrow + TW * 4);
			}
		}
		islot = islot + 2 >= kRing ? 0 : islot + 2;
	};

	T st[K][CPT];
#pragma unroll
	for (int s = 0; s < K; s++)
#pragma unroll
		for (int v = 0; v < CPT; v++)
			st[s][v] = 0;
	// (probe) level 1's vertical state on this lane's four LL columns, and the odd LL row waiting for its partner
	[[maybe_unused]] T st1[K][4] = {}, p1[4] = {};

	// explicit line-end forms: which of the lane's columns c - K .. c + CPT + K - 1 are a row's ends -- only the tiles
	// that hold column 0 or W - 1 have any (`h_any`, wave-uniform: the interior tiles run the plain lift) --; the rows that
	// are a column's ends are found per iteration (wave-uniform as well)
	[[maybe_unused]] unsigned hends = 0;
	[[maybe_unused]] bool h_any = false, h_simple = false;
	// the two entries of a lane's window that meet a line end when the level's width is a multiple of CPT: column 0 is
	// lane 0's own first column, column W - 1 a lane's own last one (entries 0 and NARR - 1 are never acted on)
	constexpr unsigned kCand = (1u << K) | (1u << (K + CPT - 1));
	if constexpr (W::kEndForms) {
		hends = end_mask<NARR>(c0 + lane * CPT - K, a.W);
		h_any = !a.plain_ends && __builtin_amdgcn_ballot_w64(hends != 0) != 0;
		h_simple = __builtin_amdgcn_ballot_w64((hends & ~(kCand | 1u | (1u << (NARR - 1)))) != 0) == 0;
	}
	// (SelEnds) the same two entries by selection: the lane's flags and its coefficients for the steps that reach them
	[[maybe_unused]] bool e0 = false, e1 = false;
	[[maybe_unused]] T kh[K];
	if constexpr (kIsSelEnds<W>) {
		const unsigned m = end_mask_long<NARR>(c0 + lane * CPT - K, a.W);
		e0 = (m >> K) & 1;
		e1 = (m >> (K + CPT - 1)) & 1;
		sel_coefs<W, false, K>(kh, e0, e1);
	}
	// row r (any r the sweep meets) is an end of its column: r == 0 or r == H - 1 after reflection (one bounce when tall)
	[[maybe_unused]] auto row_is_end = [&](int r) {
		if (tall || kIsSelEnds<W>) // (SelEnds runs on levels of 64 rows or more)
			return r == 0 || r == a.H - 1;
		const int rr = reflect(r, a.H);
		return rr == 0 || rr == a.H - 1;
	};

	for (int it = 0; it < kAhead && it < n_iter; it++)
		issue(it);

	for (int it = 0; it < n_iter; it++) {
		if (it + kAhead < n_iter) {
			issue(it + kAhead);
			// everything older than the youngest kAhead iterations' DMAs has landed
			DWT_WAIT_VMCNT(kAhead * kDmaPerIter);
		} else {
			DWT_WAIT_VMCNT(0);
		}

		// horizontal pass of the two rows (2q-1, 2q)
		T row[2][CPT];
		[[maybe_unused]] T xs[2][kPairRows ? NARR : 1];
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const unsigned base = ring_off + (unsigned)(rslot + rr) * RS * 4;
			const unsigned own = base + lane * CPT * 4;
			const unsigned la = lane == 0 ? base + TW * 4 : own - 16;
			const unsigned ra = lane == 63 ? base + TW * 4 + 16 : own + CPT * 4;
			T x[NARR];
			u4 L4, R4, O0, O1;
			if constexpr (kShuffle) {
				// own columns from LDS; the tile's outer halo (one 16 B block per side) is
				// read by every lane as a broadcast and used by lanes 0 and 63 only
				u4 H4;
				const unsigned ha = base + TW * 4 + (lane == 63 ? 16 : 0);
				if constexpr (CPT == 8) {
					lds_read2o(own, ha, O0, O1, H4);
				} else {
					lds_read2(own, ha, O0, H4);
					O1 = O0;
				}
#pragma unroll
				for (int e = 0; e < 4; e++) {
					const unsigned l = from_left_lane(O1[e]);  // neighbour's last four columns
					const unsigned r = from_right_lane(O0[e]); // neighbour's first four columns
					L4[e] = lane == 0 ? H4[e] : l;
					R4[e] = lane == 63 ? H4[e] : r;
				}
				if constexpr (CPT == 8) {
#pragma unroll
					for (int e = 0; e < 4; e++)
						x[K + 4 + e] = from_bits<T>(O1[e]);
				}
			} else if constexpr (CPT == 8) {
				lds_read4(la, own, ra, L4, O0, O1, R4);
#pragma unroll
				for (int e = 0; e < 4; e++)
					x[K + 4 + e] = from_bits<T>(O1[e]);
			} else {
				lds_read3(la, own, ra, L4, O0, R4);
			}
#pragma unroll
			for (int e = 0; e < K; e++) {
				x[e] = from_bits<T>(L4[4 - K + e]);
				x[K + CPT + e] = from_bits<T>(R4[e]);
			}
#pragma unroll
			for (int e = 0; e < 4; e++)
				x[K + e] = from_bits<T>(O0[e]);
			if constexpr (kPairRows) {
#pragma unroll
				for (int j = 0; j < NARR; j++)
					xs[rr][j] = x[j];
				continue; // (both rows are lifted together below)
			}
			if constexpr (W::kEndForms) {
				if (__builtin_expect(!h_any, 1)) {
					lift_fwd_regs<W, NARR>(x, 0u);
				} else if (h_simple) {
					DWT_END_PATH();
					lift_fwd_regs<W, NARR, kCand>(x, hends);
				} else {
					DWT_END_PATH();
					lift_fwd_regs<W, NARR>(x, hends);
				}
			} else if constexpr (kIsSelEnds<W>) {
				lift_regs_sel<W, NARR, false, K, K + CPT - 1>(x, e0, e1, kh);
			} else {
				lift_fwd_regs<W, NARR>(x, 0u);
			}
#pragma unroll
			for (int v = 0; v < CPT; v++)
				row[rr][v] = W::fwd_scale(v & 1, x[K + v]);
		}

		if constexpr (kPairRows) {
			// both rows at once, as the halves of packed operations: the same steps, the same rounding (the compiler pairs
			// within a row on its own, but not all of it, and the selects at the two candidate entries split its pairs)
			typedef float f2 __attribute__((ext_vector_type(2)));
			f2 x2[NARR];
#pragma unroll
			for (int j = 0; j < NARR; j++)
				x2[j] = f2{xs[0][j], xs[1][j]};
#pragma unroll
			for (int s_ = 0; s_ < K; s_++) {
#pragma unroll
				for (int j = s_ + 1; j <= NARR - 2 - s_; j += 2) {
					if (j == K)
						x2[j] = x2[j] + kh[s_] * ((e0 ? f2{-0.0f, -0.0f} : x2[j - 1]) + x2[j + 1]);
					else if (j == K + CPT - 1)
						x2[j] = x2[j] + kh[s_] * (x2[j - 1] + (e1 ? f2{-0.0f, -0.0f} : x2[j + 1]));
					else
						x2[j] = x2[j] + W::fk(s_) * (x2[j - 1] + x2[j + 1]);
				}
			}
			const float ze = W::fwd_scale(0, 1.0f), zo = W::fwd_scale(1, 1.0f); // (the scale factors themselves)
#pragma unroll
			for (int v = 0; v < CPT; v++) {
				const f2 sc = x2[K + v] * ((v & 1) ? zo : ze);
				row[0][v] = sc[0];
				row[1][v] = sc[1];
			}
		}
		rslot = rslot + 2 >= kRing ? 0 : rslot + 2;

		// vertical pass: streaming lifting, state in registers
		// step s of this iteration acts on row 2q-1-s: which of them are column ends (wave-uniform; almost never any)
		[[maybe_unused]] bool vend[K] = {};
		[[maybe_unused]] bool v_any = false;
		if constexpr (W::kEndForms) {
#pragma unroll
			for (int s_ = 0; s_ < K; s_++) {
				vend[s_] = row_is_end(2 * (q0 + it) - 1 - s_);
				v_any = !a.plain_ends && (v_any || vend[s_]);
			}
		}
		T lo[CPT], hi[CPT];
		// `ENDS`: the iteration meets a column end -- the steps on that row take the end form
		auto vertical = [&](auto ends_tag) {
			constexpr bool ENDS = decltype(ends_tag)::value;
			if constexpr (kPairRows && !ENDS) {
				vertical_pairs<W, CPT>(row, st, lo, hi);
				return;
			}
#pragma unroll
			for (int v = 0; v < CPT; v++) {
				const T ov = row[0][v], ev = row[1][v];
				if constexpr (K == 4) {
					const T d1n = fwd_step_at<W>(0, ENDS && vend[0], ov, st[0][v], ev);
					const T s1n = fwd_step_at<W>(1, ENDS && vend[1], st[0][v], st[1][v], d1n);
					const T d2n = fwd_step_at<W>(2, ENDS && vend[2], st[1][v], st[2][v], s1n);
					const T s2n = fwd_step_at<W>(3, ENDS && vend[3], st[2][v], st[3][v], d2n);
					lo[v] = W::fwd_scale(0, s2n);
					hi[v] = W::fwd_scale(1, d2n);
					st[0][v] = ev;
					st[1][v] = d1n;
					st[2][v] = s1n;
					st[3][v] = d2n;
				} else {
					const T d1n = fwd_step_at<W>(0, ENDS && vend[0], ov, st[0][v], ev);
					const T s1n = fwd_step_at<W>(1, ENDS && vend[1], st[0][v], st[1][v], d1n);
					lo[v] = W::fwd_scale(0, s1n);
					hi[v] = W::fwd_scale(1, d1n);
					st[0][v] = ev;
					st[1][v] = d1n;
				}
			}
		};
		if constexpr (W::kEndForms) {
			if (__builtin_expect(v_any, 0)) {
				DWT_END_PATH();
				vertical(std::true_type{});
			}
			else
				vertical(std::false_type{});
		} else if constexpr (kIsSelEnds<W>) {
			// step s acts on row 2q-1-s; where that row is an end of its column (wave-uniform, the top and bottom tiles'
			// first / last iterations) the step's coefficient is doubled and its state tap -- the same values as the other
			// tap there -- gives way to -0.0.  Deep ring: the iterations that meet no end take the plain body.
			bool ve[K], any = false;
#pragma unroll
			for (int s_ = 0; s_ < K; s_++) {
				ve[s_] = row_is_end(2 * (q0 + it) - 1 - s_);
				any = any || ve[s_];
			}
			auto vertical_sel = [&]() {
				T kv[K];
#pragma unroll
				for (int s_ = 0; s_ < K; s_++)
					kv[s_] = sel_coef<W, false>(s_, ve[s_]);
				if constexpr (kPairRows) {
					vertical_pairs<W, CPT, true>(row, st, lo, hi, ve, kv);
					return;
				}
#pragma unroll
				for (int v = 0; v < CPT; v++) {
					const T ov = row[0][v], ev = row[1][v];
					if constexpr (K == 4) {
						const T d1n = sel_step<W, false>(0, ve[0], kv[0], ov, st[0][v], ev);
						const T s1n = sel_step<W, false>(1, ve[1], kv[1], st[0][v], st[1][v], d1n);
						const T d2n = sel_step<W, false>(2, ve[2], kv[2], st[1][v], st[2][v], s1n);
						const T s2n = sel_step<W, false>(3, ve[3], kv[3], st[2][v], st[3][v], d2n);
						lo[v] = W::fwd_scale(0, s2n);
						hi[v] = W::fwd_scale(1, d2n);
						st[0][v] = ev;
						st[1][v] = d1n;
						st[2][v] = s1n;
						st[3][v] = d2n;
					} else {
						const T d1n = sel_step<W, false>(0, ve[0], kv[0], ov, st[0][v], ev);
						const T s1n = sel_step<W, false>(1, ve[1], kv[1], st[0][v], st[1][v], d1n);
						lo[v] = W::fwd_scale(0, s1n);
						hi[v] = W::fwd_scale(1, d1n);
						st[0][v] = ev;
						st[1][v] = d1n;
					}
				}
			};
			// (shallow ring: the launches of a few rounds of waves, bound by the longest wave -- the top tiles', for whom a
			// second body means instructions fetched cold from HBM, 1 us a launch; there every iteration selects)
			if constexpr (RING == 8 && has_coef_ends<W>::value)
				vertical_sel();
			else if (__builtin_expect(any, 0)) {
				DWT_END_PATH();
				vertical_sel();
			} else
				vertical(std::false_type{});
		} else {
			vertical(std::false_type{});
		}

		if constexpr (IL) {
			if (it >= K && (!X || A + it - K >= kIlKeepTop / 2) && a.out_step > 1) {
				// the level goes straight to the lattice it lives on in a larger image: a dword per sample.  Where a
				// deeper level follows, the lattice points at (even row, even column) are its to fill
				const int k = A + it - K;
				const unsigned sb = (unsigned)a.out_step * 4;
				const unsigned cb = (unsigned)(c0 + lane * CPT) * sb;
				const unsigned row_bytes = (unsigned)((X ? a.W - kIlKeepRight : a.W) - 1) * sb + 4;
				const T *r0 = out_h + (long)(2 * k) * a.h_pitch;
				const row_rsrc_t d0 = row_rsrc(r0, row_bytes);
#pragma unroll
				for (int e = 0; e < CPT; e++)
					if ((e & 1) || !a.il_ll)
						store4_row<false>(d0, cb + e * sb, to_bits(lo[e]));
				if (2 * k + 1 < a.H) {
					const row_rsrc_t d1 = row_rsrc(r0 + a.h_pitch, row_bytes);
#pragma unroll
					for (int e = 0; e < CPT; e++)
						store4_row<false>(d1, cb + e * sb, to_bits(hi[e]));
				}
				if (a.il_ll) {
					const row_rsrc_t dl = row_rsrc(out_ll + (long)k * a.ll_pitch, (unsigned)(X ? (a.W - kIlKeepRight + 1) >> 1 : Wd) * 4);
#pragma unroll
					for (int e = 0; e < CPT; e += 4)
						store8_row<false>(dl, (unsigned)(c0 + lane * CPT) * 2 + e * 2, u2{to_bits(lo[e]), to_bits(lo[e + 2])});
				}
			} else
			if (it >= K && (!X || A + it - K >= kIlKeepTop / 2)) {
				const int k = A + it - K;
				const unsigned cb = (unsigned)(c0 + lane * CPT) * 4; // byte offset in an interleaved row
				const T *r0 = out_h + (long)(2 * k) * a.h_pitch;
				// (X: the rows end 8 columns early -- the buffer drops what lies beyond)
				const unsigned row_bytes = (unsigned)(X ? a.W - kIlKeepRight : a.W) * 4;
				const row_rsrc_t d0 = row_rsrc(r0, row_bytes);
				// multi-level: the compose pass reads this (even) row again soon -- temporal, so that it can stay in the
				// Infinity Cache; the odd rows are final: non-temporal
#pragma unroll
				for (int e = 0; e < CPT; e += 4) {
					const u4 v4{to_bits(lo[e]), to_bits(lo[e + 1]), to_bits(lo[e + 2]), to_bits(lo[e + 3])};
					if (a.il_ll == 2)
						store16_row<false>(d0, cb + e * 4, v4);
					else
						store16_row<kNtStore>(d0, cb + e * 4, v4);
				}
				if (2 * k + 1 < a.H) {
					const row_rsrc_t d1 = row_rsrc(r0 + a.h_pitch, row_bytes);
#pragma unroll
					for (int e = 0; e < CPT; e += 4)
						store16_row<kNtStore>(d1, cb + e * 4, u4{to_bits(hi[e]), to_bits(hi[e + 1]), to_bits(hi[e + 2]), to_bits(hi[e + 3])});
				}
				// multi-level: the next level's input (even row, even column) also goes
				// out densely, so that no level has to gather a strided lattice
				if (a.il_ll) {
					const row_rsrc_t dl = row_rsrc(out_ll + (long)k * a.ll_pitch, (unsigned)(X ? (a.W - kIlKeepRight + 1) >> 1 : Wd) * 4);
#pragma unroll
					for (int e = 0; e < CPT; e += 4)
						store8_row<false>(dl, cb / 2 + e * 2, u2{to_bits(lo[e]), to_bits(lo[e + 2])});
				}
			}
		} else
		if (it >= K + pw) {
			// Mallat rows: [LL (Wd) | HL (W/2)] at row k, [LH | HH] at row Hd + k; each quarter row is a
			// buffer of its own, so the lanes (and dwords) beyond its end are dropped
			const int k = A + it - K - pw;
			const unsigned clb = (unsigned)((c0 + lane * CPT) >> 1) * 4;
			const T *top = out_h + (long)k * a.h_pitch, *bot = out_h + (long)(Hd + k) * a.h_pitch;
			const unsigned nlb = (unsigned)Wd * 4, nhb = (unsigned)(a.W >> 1) * 4;
			const row_rsrc_t dll = row_rsrc(out_ll + (long)k * a.ll_pitch, nlb), dhl = row_rsrc(top + Wd, nhb);
			const bool hrow = k < (a.H >> 1);
			if constexpr (CPT == 8) {
				if constexpr (kProbe && K == 4 && std::is_floating_point<T>::value) {
					// level 1 on the LL row this iteration produced: neighbours' samples by wavefront shifts, horizontal lift,
					// then -- every second row -- the vertical step and the four quarter-row stores of level 1
					T x1[12];
#pragma unroll
					for (int e = 0; e < 4; e++) {
						x1[e] = from_bits<T>(from_left_lane(to_bits(lo[2 * e])));
						x1[4 + e] = lo[2 * e];
						x1[8 + e] = from_bits<T>(from_right_lane(to_bits(lo[2 * e])));
					}
					lift_fwd_regs<W, 12>(x1, 0u);
					T r1[4];
#pragma unroll
					for (int v = 0; v < 4; v++)
						r1[v] = W::fwd_scale(v & 1, x1[4 + v]);
					if (!((it - K - pw) & 1)) {
#pragma unroll
						for (int v = 0; v < 4; v++)
							p1[v] = r1[v];
					} else {
						T lo1[4], hi1[4];
#pragma unroll
						for (int v = 0; v < 4; v++) {
							const T d1n = W::fwd_step(0, p1[v], st1[0][v], r1[v]);
							const T s1n = W::fwd_step(1, st1[0][v], st1[1][v], d1n);
							const T d2n = W::fwd_step(2, st1[1][v], st1[2][v], s1n);
							const T s2n = W::fwd_step(3, st1[2][v], st1[3][v], d2n);
							lo1[v] = W::fwd_scale(0, s2n);
							hi1[v] = W::fwd_scale(1, d2n);
							st1[0][v] = r1[v];
							st1[1][v] = d1n;
							st1[2][v] = s1n;
							st1[3][v] = d2n;
						}
						const int k1 = k >> 1, Wq = (Wd + 1) >> 1, Hq = (Hd + 1) >> 1;
						const unsigned qb = (unsigned)((c0 + lane * CPT) >> 2) * 4, nqb = (unsigned)Wq * 4;
						const T *t1 = out_h + (long)k1 * a.h_pitch, *b1 = out_h + (long)(Hq + k1) * a.h_pitch;
						store8_row<kNtStoreLL>(row_rsrc(out_ll + (long)k1 * a.ll_pitch, nqb), qb, u2{to_bits(lo1[0]), to_bits(lo1[2])});
						store8_row<kNtStore>(row_rsrc(t1 + Wq, nqb), qb, u2{to_bits(lo1[1]), to_bits(lo1[3])});
						store8_row<kNtStore>(row_rsrc(b1, nqb), qb, u2{to_bits(hi1[0]), to_bits(hi1[2])});
						store8_row<kNtStore>(row_rsrc(b1 + Wq, nqb), qb, u2{to_bits(hi1[1]), to_bits(hi1[3])});
					}
				} else
				store16_row<kNtStoreLL>(dll, clb, u4{to_bits(lo[0]), to_bits(lo[2]), to_bits(lo[4]), to_bits(lo[6])});
				store16_row<kNtStore>(dhl, clb, u4{to_bits(lo[1]), to_bits(lo[3]), to_bits(lo[5]), to_bits(lo[7])});
				if (hrow) {
					store16_row<kNtStore>(row_rsrc(bot, nlb), clb, u4{to_bits(hi[0]), to_bits(hi[2]), to_bits(hi[4]), to_bits(hi[6])});
					store16_row<kNtStore>(row_rsrc(bot + Wd, nhb), clb, u4{to_bits(hi[1]), to_bits(hi[3]), to_bits(hi[5]), to_bits(hi[7])});
				}
			} else {
				store8_row<kNtStoreLL>(dll, clb, u2{to_bits(lo[0]), to_bits(lo[2])});
				store8_row<kNtStore>(dhl, clb, u2{to_bits(lo[1]), to_bits(lo[3])});
				if (hrow) {
					store8_row<kNtStore>(row_rsrc(bot, nlb), clb, u2{to_bits(hi[0]), to_bits(hi[2])});
					store8_row<kNtStore>(row_rsrc(bot + Wd, nhb), clb, u2{to_bits(hi[1]), to_bits(hi[3])});
				}
			}
		}
	}
}

// the tile by the instantiation of the policy's line ends the level needs (dwt_lift.h)
template <class W, int CPT, int RING, int NT, bool IL, bool X>
static __device__ __forceinline__ void fwd_sweep_any_tile(const FwdLevelArgs &a, const SweepGeom &g)
{
	if constexpr (W::kEndForms && !(DWT_PROBES && (NT & 16))) {
		if (a.plain_ends)
			fwd_sweep_tile<PlainEnds<W>, CPT, RING, NT, IL, X>(a, g);
		else if (a.W % CPT == 0 && a.W >= 64 && a.H >= 64)
			fwd_sweep_tile<SelEnds<W>, CPT, RING, NT, IL, X>(a, g);
		else // (a width that puts the last column anywhere in a lane's window; short lines, reflected more than once)
			fwd_sweep_tile<W, CPT, RING, NT, IL, X>(a, g);
	} else
		fwd_sweep_tile<W, CPT, RING, NT, IL, X>(a, g);
}

template <class W, int CPT, int RING, int NT, bool IL = false>
__global__ __launch_bounds__(256) void k_fwd_sweep(FwdLevelArgs a, SweepGeom g)
{
	fwd_sweep_any_tile<W, CPT, RING, NT, IL, false>(a, g);
}

// a level with a rectangle copy riding along: the workgroups behind the tiles' copy blocks of `r` (FwdLevelArgs::ride)
template <class W, int CPT, int RING, int NT>
__global__ __launch_bounds__(256) void k_fwd_sweep_r(FwdLevelArgs a, SweepGeom g, CopyRects r)
{
	if ((int)blockIdx.x >= g.tile_blocks) {
		ride_copy_block(r, (int)blockIdx.x - g.tile_blocks);
		return;
	}
	fwd_sweep_any_tile<W, CPT, RING, NT, false, false>(a, g);
}

// one level of a phase-ordered interleaved transform, exact: workgroups [0, g.first) compute the border strips
template <class W, int CPT, int RING, int NT>
__global__ __launch_bounds__(256) void k_fwd_sweep_x(FwdLevelArgs a, SweepGeom g, IlStripArgs strip)
{
	if ((int)blockIdx.x < g.first) {
		il_strip_wave<W, false>(strip, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
		return;
	}
	fwd_sweep_any_tile<W, CPT, RING, NT, true, true>(a, g);
}

// ---- launch wrappers -------------------------------------------------------------
template <class W, int CPT, int RING, int NT, bool IL = false>
static hipError_t fwd_launch(const FwdLevelArgs &a, const SweepGeom &g, dim3 grid, int waves, hipStream_t s)
{
	const size_t lds = (size_t)waves * RING * (64 * CPT + 8) * 4;
	if constexpr (!IL && RING == 8 && (NT == 7 || NT == 3)) {
		// (the variants a level of one image takes; the launcher has checked batch == 1 and four waves)
		if (a.ride && a.ride_hi > a.ride_lo) {
			if (hipError_t e = allow_lds((const void *)k_fwd_sweep_r<W, CPT, RING, NT>, lds))
				return e;
			SweepGeom gr = g;
			gr.tile_blocks = (int)grid.x;
			CopyRects r = *a.ride;
			r.block0 = a.ride_lo;
			k_fwd_sweep_r<W, CPT, RING, NT><<<dim3(grid.x + (unsigned)(a.ride_hi - a.ride_lo), 1), 64 * waves, lds, s>>>(a, gr, r);
			return hipGetLastError();
		}
	}
	if (a.ride && a.ride_hi > a.ride_lo)
		return hipErrorInvalidValue; // the caller asked for a variant that has no ride-along kernel
	if (hipError_t e = allow_lds((const void *)k_fwd_sweep<W, CPT, RING, NT, IL>, lds))
		return e;
	k_fwd_sweep<W, CPT, RING, NT, IL><<<grid, 64 * waves, lds, s>>>(a, g);
	return hipGetLastError();
}

template <class W, int CPT>
static hipError_t fwd_pick(const FwdLevelArgs &a, const SweepGeom &g, dim3 grid, int waves, const SweepTuning &t, hipStream_t s)
{
	// cache policy 6 (non-temporal loads, every store temporal: the staged outputs of an in-place call, read back
	// at once), 7 (non-temporal loads and detail stores, the LL band's stores temporal), 3 (the LL band
	// non-temporal too: launches whose LL bands exceed the Infinity Cache), 15 (= 7 with the neighbour taps
	// by wavefront shifts instead of LDS reads: bit-identical, 0.8 % slower, the cross-check variant).  The
	// other policies and ring depths of rounds 1-3 measured slower and are gone (profiles/archive/r02_experiments.md).
	const int nt = a.temporal ? 6 : (t.nt & 8) ? 15 : (t.nt & 4) ? 7 : 3;
#if DWT_PROBES
	if constexpr (CPT == 8 && std::is_same<W, Cdf97S>::value) {
		if (a.probe_fuse1 && (nt == 3 || nt == 7)) { // the timing probe of a fused level 0 + 1 (wrong results)
			if (t.ring == 16)
				return nt == 3 ? fwd_launch<W, CPT, 16, 19>(a, g, grid, waves, s) : fwd_launch<W, CPT, 16, 23>(a, g, grid, waves, s);
			return nt == 3 ? fwd_launch<W, CPT, 8, 19>(a, g, grid, waves, s) : fwd_launch<W, CPT, 8, 23>(a, g, grid, waves, s);
		}
	}
#endif
	if (t.ring == 16) {
		switch (nt) {
		case 6: return fwd_launch<W, CPT, 16, 6>(a, g, grid, waves, s);
		case 3: return fwd_launch<W, CPT, 16, 3>(a, g, grid, waves, s);
		case 7: return fwd_launch<W, CPT, 16, 7>(a, g, grid, waves, s);
		default: return fwd_launch<W, CPT, 16, 15>(a, g, grid, waves, s);
		}
	}
	switch (nt) {
	case 6: return fwd_launch<W, CPT, 8, 6>(a, g, grid, waves, s);
	case 3: return fwd_launch<W, CPT, 8, 3>(a, g, grid, waves, s);
	case 7: return fwd_launch<W, CPT, 8, 7>(a, g, grid, waves, s);
	default: return fwd_launch<W, CPT, 8, 15>(a, g, grid, waves, s);
	}
}

template <class W, int RING>
static hipError_t fwd_launch_x(const FwdLevelArgs &a, SweepGeom g, dim3 grid, int waves, const IlStripArgs &strip, hipStream_t s)
{
	const size_t lds = (size_t)waves * RING * (64 * 4 + 8) * 4;
	if (hipError_t e = allow_lds((const void *)k_fwd_sweep_x<W, 4, RING, 3>, lds))
		return e;
	g.first = il_strip_blocks(a.W, a.H, waves);
	grid.x += g.first;
	k_fwd_sweep_x<W, 4, RING, 3><<<grid, 64 * waves, lds, s>>>(a, g, strip);
	return hipGetLastError();
}

template <class W>
static hipError_t fwd_level_t(const FwdLevelArgs &a, const SweepTuning &t, hipStream_t s, const IlStripArgs *strip = nullptr)
{
	if (a.W < 2 || a.H < 2 || a.batch < 1)
		return hipErrorInvalidValue;
	const int cpt = a.interleaved ? 4 : pick_cpt(t, a.W, false);
	const int TW = 64 * cpt;
	SweepGeom g;
	g.tile_pairs = pick_tile_pairs(t, a.W, a.H, cpt, a.batch);
	g.ntx = (a.W + TW - 1) / TW;
	if ((a.probe_fuse1 == 2 || a.probe_fuse1 == 4) && !a.interleaved && cpt == 8)
		g.ntx = (a.W + 16 + (TW - 16) - 1) / (TW - 16); // (probe: tiles 496 columns apart)
	g.swz = t.xcd_swizzle;
	g.wave_horiz = 0;
	const int Hd = (a.H + 1) / 2;
	const int waves = t.waves >= 1 && t.waves <= 4 ? t.waves : 4;
	const int nty = (Hd + g.tile_pairs - 1) / g.tile_pairs;
	// Ring depth (measured, scripts/sweep.py): when a launch has several rounds of tiles
	// per CU, 4 waves/CU with a 16-row ring (7 iterations of DMA in flight per wave) and
	// side-by-side waves beat 8 waves/CU with an 8-row ring (5.5 -> 6.0 TB/s at level 0);
	// smaller launches prefer more resident waves.
	SweepTuning tt = t;
	if (tt.ring != 8 && tt.ring != 16)
		// (one image in the interleaved layout: from 512 tiles on -- 8192^2 J=5 250-252 -> 247.5-248 us, the Mallat entry
		// loses 5 us of 148 with that threshold; measured in one process, alternated)
		tt.ring = (g.ntx >= waves && (long)g.ntx * nty * a.batch >= ((a.interleaved && a.batch == 1) ? 512 : 3072)) ? 16 : 8;
	// deep ring: the waves of a workgroup take side-by-side tiles (+ 7 %); shallow: stacked tiles
	g.wave_horiz = tt.ring == 16;
	dim3 grid;
	if (g.wave_horiz)
		grid = dim3(((g.ntx + waves - 1) / waves) * nty, a.batch);
	else
		grid = dim3(g.ntx * ((nty + waves - 1) / waves), a.batch);
	if (a.sh.rows && (!a.interleaved || a.batch != 1 || a.out_step != 1 || g.tile_pairs != a.sh.tile_pairs))
		return hipErrorInvalidValue; // the snapshot of an in-place level was taken for other tiles
	if (a.ride && a.ride_hi > a.ride_lo && (a.interleaved || a.batch != 1 || waves != 4 || tt.ring != 8 || a.temporal || (tt.nt & 8)))
		return hipErrorInvalidValue; // (fwd_ride_ok says when a copy may ride along)
	if (a.interleaved) {
		// 3-D path: float 9/7 only; always 4 columns per lane so that each row leaves the
		// wave as ONE contiguous 16 B/lane store (two strided stores per row cost 40 %)
		if constexpr (std::is_same<W, Cdf97S>::value || std::is_same<W, Cdf53SNew>::value) {
			// exact border strips in the same launch (one image, levels of 64 samples or more either way)
			if (strip) {
				if (a.batch != 1 || a.W < 64 || a.H < 64)
					return hipErrorInvalidValue;
				return tt.ring == 16 ? fwd_launch_x<W, 16>(a, g, grid, waves, *strip, s) : fwd_launch_x<W, 8>(a, g, grid, waves, *strip, s);
			}
		}
		if (strip)
			return hipErrorInvalidValue;
		if constexpr (std::is_base_of<Cdf97S, W>::value || std::is_base_of<Cdf53S, W>::value) {
			if (tt.ring == 16)
				return fwd_launch<W, 4, 16, 3, true>(a, g, grid, waves, s);
			return fwd_launch<W, 4, 8, 3, true>(a, g, grid, waves, s);
		} else {
			return hipErrorInvalidValue;
		}
	}
	if (strip)
		return hipErrorInvalidValue;
	// Cache policy bit 2 keeps the LL band's stores temporal so that the next level finds it in the
	// 256 MiB Infinity Cache.  The LL bands of a large batch do not fit (64 images of 8192^2: 4.3 GB):
	// temporal stores then only displace other lines -- non-temporal like the detail bands (64 images,
	// one process, alternated: level 0 5863-5882 -> 5959-5964 GB/s, level 1 5323-5341 -> 5381-5409, step
	// 8.08-8.10 -> 8.00-8.02 ms).  At 0.5 GB of LL (8 images) the two policies tie (level 0 loses what
	// level 1 gains), below that temporal wins: switch from 1 GiB on.  Option nt_auto = 0 turns it off.
	if (tt.nt_auto && (tt.nt & 12) == 4 && (size_t)a.batch * ((a.W + 1) / 2) * ((a.H + 1) / 2) * sizeof(typename W::T) >= ((size_t)1 << 30))
		tt.nt = 3;
	return cpt == 8 ? fwd_pick<W, 8>(a, g, grid, waves, tt, s) : fwd_pick<W, 4>(a, g, grid, waves, tt, s);
}

// whether launch_fwd_level / launch_inv_level would take a kernel that can carry a copy along for this level
bool sweep_ride_ok(const SweepTuning &t, int W, int H, int batch, bool inverse)
{
	if (batch != 1 || W < 2 || H < 2 || !(t.waves >= 1 && t.waves <= 4 ? t.waves == 4 : true) || (t.nt & 8))
		return false;
	if (inverse)
		return t.ring_inv != 16;
	if (t.ring == 16)
		return false;
	if (t.ring == 8)
		return true;
	// the launcher's own ring choice (fwd_level_t): the deep ring from 3072 tiles on
	const int cpt = pick_cpt(t, W, false);
	const int tp = pick_tile_pairs(t, W, H, cpt, 1);
	const long ntx = (W + 64 * cpt - 1) / (64 * cpt), nty = ((H + 1) / 2 + tp - 1) / tp;
	return !(ntx >= 4 && ntx * nty >= 3072);
}

int il_sweep_tile_pairs(const SweepTuning &t, int W, int H, bool inverse)
{
	return pick_tile_pairs(t, W, H, 4, 1, inverse, true);
}

hipError_t launch_fwd_level(Wavelet w, const FwdLevelArgs &a, const SweepTuning &t, hipStream_t s, const IlStripArgs *strip)
{
	if (strip && w != kCdf97S && w != kCdf53SNew)
		return hipErrorInvalidValue;
	switch (w) {
	case kCdf97S: return fwd_level_t<Cdf97S>(a, t, s, strip);
	case kCdf53I: return fwd_level_t<Cdf53I>(a, t, s);
	case kCdf53S: return fwd_level_t<Cdf53S>(a, t, s);
	case kCdf97I: return fwd_level_t<Cdf97I>(a, t, s);
	case kCdf97SFma: return fwd_level_t<Cdf97SFma>(a, t, s);
	case kCdf53SNew: return a.interleaved ? fwd_level_t<Cdf53SNew>(a, t, s, strip) : hipErrorInvalidValue;
	default: break; // the double-precision drivers run on the line-pass kernels
	}
	return hipErrorInvalidValue;
}

} // namespace dwt
