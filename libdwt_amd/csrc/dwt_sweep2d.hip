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
#include "dwt_device.h"
#include "dwt_il_strip.h"

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
struct SweepGeom {
	int tile_pairs, ntx, swz;
	int wave_horiz; // 1: the waves of a workgroup take horizontally adjacent tiles
	int first = 0;  // k_*_sweep_x: the leading workgroups of the launch that take border strips, not tiles
};

// ---- forward -------------------------------------------------------------------
// IL: write the result INTERLEAVED in place of the Mallat de-interleave (row 2k = L
// row, row 2k+1 = H row, columns interleaved alike) to `out_h`: the layout of the
// 3-D path (src/volume-dwt.c:677-725), where a batch is the slices of a volume.
// X (interleaved layout, phase-ordered wavelets): the tiles leave out the samples whose rounding depends on the
// reference's phase order -- rows 0..7 and the last 8 columns of the level -- for the strip workgroups of the
// same launch (dwt_il_strip.h).
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
	constexpr int TW = 64 * CPT;
	constexpr int RS = TW + 8; // LDS row slot: [main TW | left halo 4 | right halo 4]
	constexpr int NARR = CPT + 2 * K;
	constexpr int kDmaPerIter = 2 * (CPT / 4 + 1); // fewest DMA instructions an iteration issues
	extern __shared__ __attribute__((aligned(16))) char smem[];

	const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
	// wave-uniform on purpose: tile geometry, row indices and row pointers then live in SGPRs
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int bid = tile_block_id(g.swz, X ? g.first : 0);
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
	const int c0 = tx * TW;
	const int n_iter = (B - A) + K;
	const int q0 = A - K / 2;

	const T *in = (const T *)a.in + (long)img * a.in_bstride;
	T *out_ll = (T *)a.out_ll + (long)img * a.ll_bstride;
	T *out_h = (T *)a.out_h + (long)img * a.h_bstride;

	char *ring = smem + (size_t)wv * kRing * RS * 4;
	const unsigned ring_off = lds_offset(ring);

	// Rows are addressed as BUFFERS (a descriptor per row in scalar registers, per-lane byte
	// offsets): the hardware checks every dword against the row's length, zero-fills loads and
	// drops stores beyond it, and 16-byte accesses need only 4-byte alignment (probed:
	// scripts/probes/buf_probe.hip).  So every tile -- overhanging the image or not, rows aligned
	// or not -- takes the same 16-byte path; the up to four reflected columns right of the edge
	// that a valid output can reach come by one 4-byte DMA per row.
	const int n_edge = min(4, c0 + TW - a.W); // columns of this tile's main block right of the edge (<= 0: none)
	const int edge_col = reflect(a.W + min(lane, 3), a.W);
	const int halo_col = reflect(lane < 4 ? c0 - 4 + lane : c0 + TW + (lane & 3), a.W);

	int islot = 0, rslot = 0; // ring slots of the next rows to fill / to consume
	const bool tall = a.H >= 64; // then a row index leaves [0,H) by less than H: one bounce
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

	// explicit line-end forms (int 5/3 only): which of the lane's columns c - K .. c + CPT + K - 1
	// are a row's ends; the rows that are a column's ends are found per iteration (wave-uniform)
	static_assert(!W::kEndForms || K == 2, "end forms are wired into the two-step vertical lift only");
	[[maybe_unused]] unsigned hends = 0;
	if constexpr (W::kEndForms)
		hends = end_mask<NARR>(c0 + lane * CPT - K, a.W);

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
			lift_fwd_regs<W, NARR>(x, hends);
#pragma unroll
			for (int v = 0; v < CPT; v++)
				row[rr][v] = W::fwd_scale(v & 1, x[K + v]);
		}

		rslot = rslot + 2 >= kRing ? 0 : rslot + 2;

		// vertical pass: streaming lifting, state in registers
		[[maybe_unused]] bool vend_o = false, vend_e = false; // rows 2q-1 / 2q-2 are column ends
		if constexpr (W::kEndForms) {
			const int ro = reflect(2 * (q0 + it) - 1, a.H), re = reflect(2 * (q0 + it) - 2, a.H);
			vend_o = ro == 0 || ro == a.H - 1;
			vend_e = re == 0 || re == a.H - 1;
		}
		T lo[CPT], hi[CPT];
#pragma unroll
		for (int v = 0; v < CPT; v++) {
			const T ov = row[0][v], ev = row[1][v];
			if constexpr (K == 4) {
				const T d1n = W::fwd_step(0, ov, st[0][v], ev);
				const T s1n = W::fwd_step(1, st[0][v], st[1][v], d1n);
				const T d2n = W::fwd_step(2, st[1][v], st[2][v], s1n);
				const T s2n = W::fwd_step(3, st[2][v], st[3][v], d2n);
				lo[v] = W::fwd_scale(0, s2n);
				hi[v] = W::fwd_scale(1, d2n);
				st[0][v] = ev;
				st[1][v] = d1n;
				st[2][v] = s1n;
				st[3][v] = d2n;
			} else {
				const T d1n = fwd_step_at<W>(0, vend_o, ov, st[0][v], ev);
				const T s1n = fwd_step_at<W>(1, vend_e, st[0][v], st[1][v], d1n);
				lo[v] = W::fwd_scale(0, s1n);
				hi[v] = W::fwd_scale(1, d1n);
				st[0][v] = ev;
				st[1][v] = d1n;
			}
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
		if (it >= K) {
			// Mallat rows: [LL (Wd) | HL (W/2)] at row k, [LH | HH] at row Hd + k; each quarter row is a
			// buffer of its own, so the lanes (and dwords) beyond its end are dropped
			const int k = A + it - K;
			const unsigned clb = (unsigned)((c0 + lane * CPT) >> 1) * 4;
			const T *top = out_h + (long)k * a.h_pitch, *bot = out_h + (long)(Hd + k) * a.h_pitch;
			const unsigned nlb = (unsigned)Wd * 4, nhb = (unsigned)(a.W >> 1) * 4;
			const row_rsrc_t dll = row_rsrc(out_ll + (long)k * a.ll_pitch, nlb), dhl = row_rsrc(top + Wd, nhb);
			const bool hrow = k < (a.H >> 1);
			if constexpr (CPT == 8) {
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

template <class W, int CPT, int RING, int NT, bool IL = false>
__global__ __launch_bounds__(256) void k_fwd_sweep(FwdLevelArgs a, SweepGeom g)
{
	fwd_sweep_tile<W, CPT, RING, NT, IL, false>(a, g);
}

// one level of a phase-ordered interleaved transform, exact: workgroups [0, g.first) compute the border strips
template <class W, int CPT, int RING, int NT>
__global__ __launch_bounds__(256) void k_fwd_sweep_x(FwdLevelArgs a, SweepGeom g, IlStripArgs strip)
{
	if ((int)blockIdx.x < g.first) {
		il_strip_wave<W, false>(strip, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
		return;
	}
	fwd_sweep_tile<W, CPT, RING, NT, true, true>(a, g);
}

// ---- inverse -------------------------------------------------------------------
// Source rows are Mallat rows: "L row p" = [LL | HL] and "H row p" = [LH | HH].
// LDS row slot (floats): [L main M | H main M | L halo 8 | H halo 8], M = TW/2;
// a halo block is [4 columns left of the tile | 4 columns right of the tile].
// IL: the input is INTERLEAVED (3-D path layout) at `in_h` instead of Mallat subbands.
// SP (interleaved input, CPT 4): the even rows are SPLIT -- their even columns (the low-pass band) come from the dense
// image `in_ll2`, their odd columns from the source row -- and take the Mallat rows' LDS layout and register gather;
// with `in_step` > 1 the source rows are rows of a lattice in a larger image.
template <class W, int CPT, int RING, int NT, bool IL, bool X, bool SP = false>
static __device__ __forceinline__ void inv_sweep_tile(const InvLevelArgs &a, const SweepGeom &g)
{
	static_assert(!SP || (IL && CPT == 4), "split even rows: interleaved input, 4 columns per lane");
	using T = typename W::T;
	constexpr int K = W::K;
	constexpr int kRing = RING;
	constexpr int kAhead = kRing / 2 - 1;
	constexpr int kLdAux = (NT & 2) ? 2 : 0;
	constexpr bool kNtStore = (NT & 1) != 0;
	// the LL band is read again by the next level: bit 2 keeps its stores temporal so it
	// can stay in L2 / Infinity Cache
	[[maybe_unused]] constexpr bool kNtStoreLL = kNtStore && !(NT & 4);
	constexpr int TW = 64 * CPT;
	constexpr int M = TW / 2;
	constexpr int RS = 2 * M + 16;
	// Mallat input, 8 columns per lane: a lane owns TWO groups of 4 columns, 256 columns apart, so that
	// each output row leaves the wave as two contiguous 1 KiB stores (8 adjacent columns per lane
	// made every store instruction write half of each 64-byte line: 136 against 109 us for level 0
	// of one 8192^2 image) while the subband segments it reads are 1 KiB instead of 512 B
	constexpr int G = (!IL && CPT == 8) ? 2 : 1; // column groups per lane
	constexpr int CG = CPT / G;                  // columns per group
	constexpr int NARR = CG + 2 * K - 1;         // interleaved samples c-K+1 .. c+CG+K-1 of a group
	constexpr int HC = CG / 2;                   // subband columns per lane and group
	constexpr int kDmaMain = IL ? CPT / 4 : (CPT == 8 ? 2 : 1);
	// (the fewest an iteration issues: SP even row: L segment, two strided H loads, halo)
	constexpr int kDmaPerIter = SP ? 4 + kDmaMain + 1 : 2 * (kDmaMain + 1);
	extern __shared__ __attribute__((aligned(16))) char smem[];

	const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
	// wave-uniform on purpose: tile geometry, row indices and row pointers then live in SGPRs
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int bid = tile_block_id(g.swz, X ? g.first : 0);
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
	if (a.pair_hi > 0 && (A < a.pair_lo || A >= a.pair_hi))
		return; // this launch computes a band of the level only
	const int B = min(A + g.tile_pairs, Hd);
	const int c0 = tx * TW;
	const int cl0 = c0 >> 1;
	const int n_iter = (B - A) + K;
	const int p0 = A - K / 2;

	const T *in_ll = (const T *)a.in_ll + (long)img * a.ll_bstride;
	const T *in_h = (const T *)a.in_h + (long)img * a.h_bstride;
	T *out = (T *)a.out + (long)img * a.out_bstride;

	char *ring = smem + (size_t)wv * kRing * RS * 4;
	const unsigned ring_off = lds_offset(ring);

	// Whole tiles fetch their subband segments as plain 16-byte DMAs (4-byte alignment is enough);
	// the tile that holds the image's right edge addresses the segments as BUFFERS (bounds-checked
	// per dword: zero fill beyond a segment's end, nothing read past the allocation) and fetches the
	// two L and two H columns right of the edge -- all a valid output can reach -- by reflection.
	const bool edge_tile = c0 + TW > a.W;
	const int nL = Wd, nH = a.W >> 1; // valid columns of an L / H segment
	// lanes 0,1: the reflected L columns nL, nL+1; lanes 2,3: the H columns nH, nH+1
	const int edge_sub = (lane & 2) ? nH + (lane & 1) : nL + (lane & 1);
	const int edge_col = reflect(2 * edge_sub + ((lane >> 1) & 1), a.W) >> 1;
	// halo lanes 0..7 -> L halo, 8..15 -> H halo
	const int hsub = (lane & 7) < 4 ? cl0 - 4 + (lane & 7) : cl0 + M + (lane & 3);
	const int halo_col = reflect(2 * hsub + ((lane >> 3) & 1), a.W) >> 1;
	const bool halo_is_h = (lane >> 3) & 1;

	// pointers to the four subbands' row starts are formed per source row
	// interleaved input: source columns of the element-wise loader and of the halo
	int halo_colI = 0;
	if constexpr (IL) {
		halo_colI = reflect(lane < 4 ? c0 - 4 + lane : c0 + TW + (lane & 3), a.W);
	}
	const int step = IL ? a.in_step : 1;
	const unsigned src_row_bytes = ((unsigned)(a.W - 1) * step + 1) * 4; // a source row up to its last sample
	// in place (a.sh, see the forward sweep): offsets of the lane's halo column in the column shell's row, or -1
	[[maybe_unused]] const bool shl = IL && a.sh.rows != nullptr;
	[[maybe_unused]] int halo_shI = -1, halo_shH = -1;
	if (IL && shl) {
		// (columns further out are fetched by the split rows' halo lanes but feed no output: wherever they come from)
		auto in_shell = [&](int col) {
			return (col >= c0 - 4 && col < c0) ? 8 * (tx - 1) + (col - (c0 - 4)) : (col >= c0 + TW && col < c0 + TW + 4) ? 8 * tx + 4 + (col - (c0 + TW)) : -1;
		};
		halo_shI = in_shell(halo_colI);
		halo_shH = in_shell(2 * halo_col + 1); // (split even rows: the high-pass half's halo, an odd image column)
	}
	auto issue = [&](int it) {
		const int p = p0 + it;
		if constexpr (IL) {
#pragma unroll
			for (int rr = 0; rr < 2; rr++) {
				const int r = reflect(2 * p + rr, a.H);
				char *lrow = ring + (size_t)((2 * it + rr) & (kRing - 1)) * RS * 4;
				// even rows come from in_ll, odd rows from in_h (reflection keeps the parity): the
				// two may be different buffers
				const T *grow = (r & 1) ? in_h + (long)(r >> 1) * a.h_pitch : in_ll + (long)(r >> 1) * a.ll_pitch;
				// own row of an in-place level read from the image: its foreign columns come from `crow`, its last 8
				// columns (the border strip waves may have written theirs already) from `rgt + column`
				[[maybe_unused]] const T *crow = nullptr, *rgt = nullptr;
				if (shl) {
					if (r < 2 * A)
						grow = (const T *)a.sh.rows + (long)(9 * (ty - 1) + r - (2 * A - 5)) * a.sh.rows_pitch;
					else if (r >= 2 * B)
						grow = (const T *)a.sh.rows + (long)(9 * ty + r - (2 * B - 5)) * a.sh.rows_pitch;
					else if (r < kIlKeepTop) // the strips' rows: the snapshot of rows 0..13
						grow = (const T *)a.sh.top + (long)r * a.sh.top_pitch;
					else {
						crow = (const T *)a.sh.cols + (long)r * a.sh.cols_pitch;
						rgt = (const T *)a.sh.right + (long)r * a.sh.right_pitch - a.sh.right_x0;
					}
				}
				const int wr = a.W - kIlKeepRight; // first of the last 8 columns
				const bool fix_right = rgt && c0 + TW > wr;
				if (SP && rr == 0) {
					// even row, split: LDS row as for Mallat rows, [L main M | H main M | L halo 8 | H halo 8]
					const T *gl = (const T *)a.in_ll2 + (long)(r >> 1) * a.ll2_pitch;
					const row_rsrc_t rl = row_rsrc(gl, (unsigned)nL * 4), rh = row_rsrc(grow, src_row_bytes);
					if (lane < 32)
						dma16_row<kLdAux>(rl, (unsigned)(cl0 + lane * 4) * 4, lrow);
#pragma unroll
					for (int i = 0; i < 2; i++)
						dma4_row<kLdAux>(rh, (unsigned)(2 * (cl0 + 64 * i + lane) + 1) * step * 4, lrow + M * 4 + i * 256);
					if (fix_right) {
						// the odd columns among the last 8, from the snapshot (lands after the strided loads above)
						const int fo = wr | 1;
						if (lane < 4 && fo + 2 * lane < a.W)
							dma4<kLdAux>(rgt + fo + 2 * lane, lrow + M * 4 + ((fo - c0) >> 1) * 4);
					}
					if (edge_tile) {
						if (lane < 2 && nL + lane < cl0 + M)
							dma4<kLdAux>(gl + edge_col, lrow + (nL - cl0) * 4);
						if (lane >= 2 && lane < 4 && nH + (lane & 1) < cl0 + M)
							dma4<kLdAux>((rgt ? rgt : grow) + (long)(2 * edge_col + 1) * step, lrow + M * 4 + (nH - cl0) * 4 - 8);
					}
					if (lane < 16) {
						const int hcol = 2 * halo_col + 1;
						dma4<kLdAux>(!halo_is_h ? gl + halo_col : (crow && halo_shH >= 0) ? crow + halo_shH : (rgt && hcol >= wr) ? rgt + hcol : grow + (long)hcol * step,
							lrow + 2 * M * 4);
					}
					continue;
				}
				if (step == 1) {
					const row_rsrc_t rs = row_rsrc(grow, (unsigned)a.W * 4);
#pragma unroll
					for (int i = 0; i < CPT / 4; i++)
						dma16_row<kLdAux>(rs, (unsigned)(c0 + i * 256 + lane * 4) * 4, lrow + i * 1024);
				} else {
					const row_rsrc_t rs = row_rsrc(grow, src_row_bytes);
#pragma unroll
					for (int i = 0; i < CPT; i++)
						dma4_row<kLdAux>(rs, (unsigned)(c0 + 64 * i + lane) * step * 4, lrow + i * 256);
				}
				if (fix_right && lane < kIlKeepRight)
					dma4<kLdAux>(rgt + (wr + lane), lrow + (wr - c0) * 4);
				if (lane < min(4, c0 + TW - a.W))
					dma4<kLdAux>((rgt ? rgt : grow) + (long)reflect(a.W + min(lane, 3), a.W) * step, lrow + (a.W - c0) * 4);
				if (lane < 8)
					dma4<kLdAux>((crow && halo_shI >= 0) ? crow + halo_shI : (rgt && halo_colI >= wr) ? rgt + halo_colI : grow + (long)halo_colI * step, lrow + TW * 4);
			}
			return;
		}
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			// rr = 0: L row p (interleaved row 2p); rr = 1: H row p (row 2p+1)
			const int rs = reflect(2 * p + rr, a.H);
			const int sub = rs >> 1;
			// reflection keeps parity, so an L row stays an L row
			const T *gl, *gh; // [left half | right half] of this Mallat row
			if (rr == 0) {
				gl = in_ll + (long)sub * a.ll_pitch;
				gh = in_h + (long)sub * a.h_pitch + Wd;
			} else {
				gl = in_h + (long)(Hd + sub) * a.h_pitch;
				gh = gl + Wd;
			}
			char *lrow = ring + (size_t)((2 * it + rr) & (kRing - 1)) * RS * 4;
			if (!edge_tile) {
				if constexpr (CPT == 8) {
					dma16<kLdAux>(gl + cl0 + lane * 4, lrow);
					dma16<kLdAux>(gh + cl0 + lane * 4, lrow + M * 4);
				} else {
					// lanes 0..31 fetch the L segment, 32..63 the H segment
					const T *gsel = lane < 32 ? gl : gh;
					dma16<kLdAux>(gsel + cl0 + (lane & 31) * 4, lrow);
				}
			} else {
				const row_rsrc_t rl = row_rsrc(gl, (unsigned)nL * 4), rh = row_rsrc(gh, (unsigned)nH * 4);
				if constexpr (CPT == 8) {
					dma16_row<kLdAux>(rl, (unsigned)(cl0 + lane * 4) * 4, lrow);
					dma16_row<kLdAux>(rh, (unsigned)(cl0 + lane * 4) * 4, lrow + M * 4);
				} else {
					if (lane < 32)
						dma16_row<kLdAux>(rl, (unsigned)(cl0 + lane * 4) * 4, lrow);
					else
						dma16_row<kLdAux>(rh, (unsigned)(cl0 + (lane & 31) * 4) * 4, lrow); // lane's slot is lrow + 16 lane = the H half
				}
				// reflected columns right of the edge that still lie in the tile's main block
				if (lane < 2 && nL + lane < cl0 + M)
					dma4<kLdAux>(gl + edge_col, lrow + (nL - cl0) * 4);
				if (lane >= 2 && lane < 4 && nH + (lane & 1) < cl0 + M)
					dma4<kLdAux>(gh + edge_col, lrow + M * 4 + (nH - cl0) * 4 - 8);
			}
			if (lane < 16)
				dma4<kLdAux>((halo_is_h ? gh : gl) + halo_col, lrow + 2 * M * 4);
		}
	};

	// vertical state: per group CG columns when rows are undone first, NARR when columns are
	// undone first and the horizontal halo must be carried
	constexpr int NVG = W::kInvColsFirst ? NARR : CG;
	T st[K][G][NVG];
#pragma unroll
	for (int s = 0; s < K; s++)
#pragma unroll
		for (int gi = 0; gi < G; gi++)
#pragma unroll
			for (int v = 0; v < NVG; v++)
				st[s][gi][v] = 0;

	// explicit line-end forms (int 5/3 only): the ends of a row among each group's samples
	// c - K + 1 .. c + CG + K - 1; the rows that are a column's ends are found per iteration
	static_assert(!W::kEndForms || (K == 2 && !IL), "end forms are wired into the two-step Mallat sweeps only");
	[[maybe_unused]] unsigned hends[G] = {};
	if constexpr (W::kEndForms) {
#pragma unroll
		for (int gi = 0; gi < G; gi++)
			hends[gi] = end_mask<NARR>(c0 + 64 * CG * gi + lane * CG - K + 1, a.W);
	}

	for (int it = 0; it < kAhead && it < n_iter; it++)
		issue(it);

	for (int it = 0; it < n_iter; it++) {
		if (it + kAhead < n_iter) {
			issue(it + kAhead);
			DWT_WAIT_VMCNT(kAhead * kDmaPerIter);
		} else {
			DWT_WAIT_VMCNT(0);
		}
		const int p = p0 + it;

		// gather the interleaved samples c-K+1 .. c+CG+K-1 of both source rows, per column group
		T x[2][G][NARR];
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const unsigned base = ring_off + (unsigned)((2 * it + rr) & (kRing - 1)) * RS * 4;
			if (IL && !(SP && rr == 0)) { // (compile time once the loop is unrolled)
				// LDS row: [main TW | left halo 4 | right halo 4] of interleaved samples
				const unsigned own = base + lane * CPT * 4;
				const unsigned la = lane == 0 ? base + TW * 4 : own - 16;
				const unsigned ra = lane == 63 ? base + TW * 4 + 16 : own + CPT * 4;
				u4 L4, R4, O0, O1;
				T ownv[CPT];
				if constexpr (CPT == 8) {
					lds_read4(la, own, ra, L4, O0, O1, R4);
#pragma unroll
					for (int e = 0; e < 4; e++)
						ownv[4 + e] = from_bits<T>(O1[e]);
				} else {
					lds_read3(la, own, ra, L4, O0, R4);
				}
#pragma unroll
				for (int e = 0; e < 4; e++)
					ownv[e] = from_bits<T>(O0[e]);
#pragma unroll
				for (int j = 0; j < NARR; j++) {
					const int rel = j - K + 1;
					const T v = rel < 0 ? from_bits<T>(L4[(4 + rel) & 3]) : rel < CPT ? ownv[rel < CPT ? (rel < 0 ? 0 : rel) : 0] : from_bits<T>(R4[(rel - CPT) & 3]);
					x[rr][0][j] = W::inv_scale(rel & 1, v);
				}
				continue;
			}
			const unsigned hbase = base + 2 * M * 4;
#pragma unroll
			for (int gi = 0; gi < G; gi++) {
				// the lane's place among the 64 G groups of 2 subband columns across the tile
				const int vl = 64 * gi + lane;
				// subband values L[cl-2 .. cl+4), H[cl-2 .. cl+4) as l[], h[]
				T l[HC + 4], h[HC + 4];
				const unsigned ownL = base + vl * 8, ownH = base + M * 4 + vl * 8;
				const unsigned laL = vl == 0 ? hbase + 8 : ownL - 8, raL = vl == 64 * G - 1 ? hbase + 16 : ownL + 8;
				const unsigned laH = vl == 0 ? hbase + 40 : ownH - 8, raH = vl == 64 * G - 1 ? hbase + 48 : ownH + 8;
				u2 a0, a1, a2, b0, b1, b2;
				lds_read2x3(laL, ownL, raL, a0, a1, a2);
				lds_read2x3(laH, ownH, raH, b0, b1, b2);
#pragma unroll
				for (int e = 0; e < 2; e++) {
					l[e] = from_bits<T>(a0[e]); l[2 + e] = from_bits<T>(a1[e]); l[4 + e] = from_bits<T>(a2[e]);
					h[e] = from_bits<T>(b0[e]); h[2 + e] = from_bits<T>(b1[e]); h[4 + e] = from_bits<T>(b2[e]);
				}
				// x[j] <-> interleaved sample c-K+1+j (x[0] odd).  Sample i: even -> L[i/2],
				// odd -> H[i/2]; relative to cl: L index (i-c)/2 -> l[2 + ...].
#pragma unroll
				for (int j = 0; j < NARR; j++) {
					const int rel = j - K + 1; // sample index relative to c (c even)
					if (rel & 1)
						x[rr][gi][j] = W::inv_scale(1, h[2 + ((rel - 1) >> 1)]);
					else
						x[rr][gi][j] = W::inv_scale(0, l[2 + (rel >> 1)]);
				}
			}
		}

		T val[2][G][NVG]; // val[0] = L row p, val[1] = H row p as the vertical pass sees them
#pragma unroll
		for (int gi = 0; gi < G; gi++) {
			if constexpr (!W::kInvColsFirst) {
#pragma unroll
				for (int rr = 0; rr < 2; rr++) {
					lift_inv_regs<W, NARR>(x[rr][gi], hends[gi]);
					// after the horizontal inverse the row is plain samples again; the
					// vertical pass descales by ROW parity
#pragma unroll
					for (int v = 0; v < CG; v++)
						val[rr][gi][v] = W::inv_scale(rr, x[rr][gi][K - 1 + v]);
				}
			} else {
#pragma unroll
				for (int rr = 0; rr < 2; rr++)
#pragma unroll
					for (int v = 0; v < NVG; v++)
						val[rr][gi][v] = x[rr][gi][v]; // int 5/3: no scaling anywhere
			}
		}

		// vertical inverse, streaming.  K == 4: at step p the rows 2p-3 (odd) and
		// 2p-2 (even) are final; K == 2: rows 2p-1 and 2p.
		[[maybe_unused]] bool vend_e = false, vend_o = false; // rows 2p / 2p-1 are column ends
		if constexpr (W::kEndForms) {
			const int re = reflect(2 * p, a.H), ro = reflect(2 * p - 1, a.H);
			vend_e = re == 0 || re == a.H - 1;
			vend_o = ro == 0 || ro == a.H - 1;
		}
		T odd_row[G][NVG], even_row[G][NVG];
#pragma unroll
		for (int gi = 0; gi < G; gi++)
#pragma unroll
		for (int v = 0; v < NVG; v++) {
			const T s2 = val[0][gi][v], d2 = val[1][gi][v];
			if constexpr (K == 4) {
				// st: [0] d2[p-1], [1] s1[p-1], [2] d1[p-2], [3] e[p-2]
				const T s1n = W::inv_step(0, s2, st[0][gi][v], d2);               // s1[p]
				const T d1n = W::inv_step(1, st[0][gi][v], st[1][gi][v], s1n);    // d1[p-1]
				const T en = W::inv_step(2, st[1][gi][v], st[2][gi][v], d1n);     // e[p-1]
				const T on = W::inv_step(3, st[2][gi][v], st[3][gi][v], en);      // o[p-2]
				odd_row[gi][v] = on;
				even_row[gi][v] = en;
				st[0][gi][v] = d2;
				st[1][gi][v] = s1n;
				st[2][gi][v] = d1n;
				st[3][gi][v] = en;
			} else {
				// st: [0] d[p-1], [1] e[p-1]
				const T en = inv_step_at<W>(0, vend_e, s2, st[0][gi][v], d2);                // e[p]
				const T on = inv_step_at<W>(1, vend_o, st[0][gi][v], st[1][gi][v], en);      // o[p-1]
				odd_row[gi][v] = on;
				even_row[gi][v] = en;
				st[0][gi][v] = d2;
				st[1][gi][v] = en;
			}
		}
		// output rows and their validity inside this tile
		const int pe = (K == 4) ? p - 1 : p;     // pair index of even_row
		const int po = (K == 4) ? p - 2 : p - 1; // pair index of odd_row
		const bool ve = pe >= A && pe < B && (!X || pe >= kIlKeepTop / 2);
		const bool vo = po >= A && po < B && (2 * po + 1 < a.H) && (!X || po >= kIlKeepTop / 2);
		const unsigned row_bytes = (unsigned)(X ? a.W - kIlKeepRight : a.W) * 4;

		T orow[G][CG], erow[G][CG];
#pragma unroll
		for (int gi = 0; gi < G; gi++) {
			if constexpr (W::kInvColsFirst) {
				lift_inv_regs<W, NARR>(odd_row[gi], hends[gi]);
				lift_inv_regs<W, NARR>(even_row[gi], hends[gi]);
#pragma unroll
				for (int v = 0; v < CG; v++) {
					orow[gi][v] = odd_row[gi][K - 1 + v];
					erow[gi][v] = even_row[gi][K - 1 + v];
				}
			} else {
#pragma unroll
				for (int v = 0; v < CG; v++) {
					orow[gi][v] = odd_row[gi][v];
					erow[gi][v] = even_row[gi][v];
				}
			}
		}

		// output rows as buffers: lanes and dwords beyond the row's end are dropped.  A lane's
		// columns: c0 + lane CG (+ 256 for the second group), 16 bytes each
#pragma unroll
		for (int gi = 0; gi < G; gi++) {
#pragma unroll
			for (int e = 0; e < CG; e += 4) {
				const unsigned cb = (unsigned)(c0 + 64 * CG * gi + lane * CG + e) * 4;
				if (vo)
					store16_row<kNtStore>(row_rsrc(out + (long)(2 * po + 1) * a.out_pitch, row_bytes), cb,
						u4{to_bits(orow[gi][e]), to_bits(orow[gi][e + 1]), to_bits(orow[gi][e + 2]), to_bits(orow[gi][e + 3])});
				if (ve)
					store16_row<kNtStore>(row_rsrc(out + (long)(2 * pe) * a.out_pitch, row_bytes), cb,
						u4{to_bits(erow[gi][e]), to_bits(erow[gi][e + 1]), to_bits(erow[gi][e + 2]), to_bits(erow[gi][e + 3])});
			}
		}
	}
}

template <class W, int CPT, int RING, int NT, bool IL = false>
__global__ __launch_bounds__(256) void k_inv_sweep(InvLevelArgs a, SweepGeom g)
{
	inv_sweep_tile<W, CPT, RING, NT, IL, false>(a, g);
}

// interleaved input, 4 columns per lane; SP: split even rows (InvLevelArgs::in_ll2)
template <class W, int RING, int NT, bool SP>
__global__ __launch_bounds__(256) void k_inv_sweep_il(InvLevelArgs a, SweepGeom g)
{
	inv_sweep_tile<W, 4, RING, NT, true, false, SP>(a, g);
}

template <class W, int RING, int NT, bool SP>
__global__ __launch_bounds__(256) void k_inv_sweep_x(InvLevelArgs a, SweepGeom g, IlStripArgs strip)
{
	if ((int)blockIdx.x < g.first) {
		il_strip_wave<W, true>(strip, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
		return;
	}
	inv_sweep_tile<W, 4, RING, NT, true, true, SP>(a, g);
}

// ---- launch wrappers -------------------------------------------------------------
static int pick_cpt(const SweepTuning &t, int W, bool inverse)
{
	if (t.cpt == 4 || t.cpt == 8)
		return t.cpt;
	// forward: 8 columns/lane gives one 16 B store per subband row; inverse: 4
	// columns/lane gives one contiguous 16 B store per output row.  Narrow levels
	// take the narrower tile so that more waves share the work.
	if (inverse)
		return 4;
	// (below 2048 columns 8/lane leaves fewer than 4 tiles per row: a workgroup of four
	// side-by-side waves would be half idle; measured 4.15 vs 4.85 TB/s on 1024 x 1024^2)
	return W >= 2048 ? 8 : 4;
}

static int pick_tile_pairs(const SweepTuning &t, int W, int H, int cpt, int batch, bool inverse = false)
{
	if (t.tile_pairs > 0)
		return t.tile_pairs;
	// Measured on MI355X (scripts/sweep_levels.sh): big levels are bandwidth bound and
	// want tall tiles (the K-row warm-up re-reads the tile above: 6 % at 64 pairs);
	// levels of a few million samples are latency bound -- a wave's sweep is a serial
	// chain -- and want the shortest tiles so that all CUs work at once.
	const int Hd = (H + 1) / 2;
	// (round 2, single-image sweep of 1024^2 / 512^2 / 256^2: 2 pairs 8.7 / 8.3 / 8.0 us against
	// 10.4 / 10.0 / 9.5 us with 4 pairs -- a launch this small is one round of waves whatever the
	// tile height, and its duration is the length of one wave's serial chain)
	// (the inverse alike: 10.7 against 12.7 us for the 1024^2 and 512^2 levels of a single image)
	if ((long)W * H * batch <= (1L << 20))
		return 2;
	if ((long)W * H * batch <= (4L << 20))
		return 4;
	const long ntx = (W + 64 * cpt - 1) / (64 * cpt);
	// the inverse sweep (256-column tiles, twice the waves) peaks at 32 pairs
	int tp = inverse ? 32 : 64;
	const long want = inverse ? 2048 : 1024; // inverse tiles are half as wide
	while (tp > 8 && ntx * ((Hd + tp - 1) / tp) * batch < want)
		tp >>= 1;
	return tp;
}

template <class W, int CPT, int RING, int NT, bool IL = false>
static hipError_t fwd_launch(const FwdLevelArgs &a, const SweepGeom &g, dim3 grid, int waves, hipStream_t s)
{
	const size_t lds = (size_t)waves * RING * (64 * CPT + 8) * 4;
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
	// other policies and ring depths of rounds 1-3 measured slower and are gone (profiles/r02_experiments.md).
	const int nt = a.temporal ? 6 : (t.nt & 8) ? 15 : (t.nt & 4) ? 7 : 3;
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
		tt.ring = (g.ntx >= waves && (long)g.ntx * nty * a.batch >= 3072) ? 16 : 8;
	// deep ring: the waves of a workgroup take side-by-side tiles (+ 7 %); shallow: stacked tiles
	g.wave_horiz = tt.ring == 16;
	dim3 grid;
	if (g.wave_horiz)
		grid = dim3(((g.ntx + waves - 1) / waves) * nty, a.batch);
	else
		grid = dim3(g.ntx * ((nty + waves - 1) / waves), a.batch);
	if (a.sh.rows && (!a.interleaved || a.batch != 1 || a.out_step != 1 || g.tile_pairs != a.sh.tile_pairs))
		return hipErrorInvalidValue; // the snapshot of an in-place level was taken for other tiles
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

template <class W, int CPT, int RING, int NT, bool IL>
static hipError_t inv_launch(const InvLevelArgs &a, const SweepGeom &g, dim3 grid, int waves, hipStream_t s)
{
	const size_t lds = (size_t)waves * RING * (64 * CPT + 16) * 4;
	if (hipError_t e = allow_lds((const void *)k_inv_sweep<W, CPT, RING, NT, IL>, lds))
		return e;
	k_inv_sweep<W, CPT, RING, NT, IL><<<grid, 64 * waves, lds, s>>>(a, g);
	return hipGetLastError();
}

template <class W, int CPT>
static hipError_t inv_pick(const InvLevelArgs &a, const SweepGeom &g, dim3 grid, int waves, int ring, hipStream_t s)
{
	// non-temporal stores, cacheable loads (the four source segments of a row pair live on L2 hits: non-temporal
	// loads measured 4.37 against 5.35 TB/s); ring 16 is the cross-check variant (5.21)
	if (ring == 16)
		return inv_launch<W, CPT, 16, 1, false>(a, g, grid, waves, s);
	return inv_launch<W, CPT, 8, 1, false>(a, g, grid, waves, s);
}

template <class W, bool SP>
static hipError_t inv_launch_x(const InvLevelArgs &a, SweepGeom g, dim3 grid, int waves, const IlStripArgs &strip, hipStream_t s)
{
	const size_t lds = (size_t)waves * 8 * (64 * 4 + 16) * 4;
	if (hipError_t e = allow_lds((const void *)k_inv_sweep_x<W, 8, 0, SP>, lds))
		return e;
	g.first = il_strip_blocks(a.W, a.H, waves);
	grid.x += g.first;
	k_inv_sweep_x<W, 8, 0, SP><<<grid, 64 * waves, lds, s>>>(a, g, strip);
	return hipGetLastError();
}

template <class W>
static hipError_t inv_launch_il_sp(const InvLevelArgs &a, const SweepGeom &g, dim3 grid, int waves, hipStream_t s)
{
	const size_t lds = (size_t)waves * 8 * (64 * 4 + 16) * 4;
	if (hipError_t e = allow_lds((const void *)k_inv_sweep_il<W, 8, 0, true>, lds))
		return e;
	k_inv_sweep_il<W, 8, 0, true><<<grid, 64 * waves, lds, s>>>(a, g);
	return hipGetLastError();
}

template <class W>
static hipError_t inv_level_t(const InvLevelArgs &a, const SweepTuning &t, hipStream_t s, const IlStripArgs *strip = nullptr)
{
	if (a.W < 2 || a.H < 2 || a.batch < 1)
		return hipErrorInvalidValue;
	const int cpt = a.interleaved ? 4 : pick_cpt(t, a.W, true);
	const int TW = 64 * cpt;
	SweepGeom g;
	g.tile_pairs = pick_tile_pairs(t, a.W, a.H, cpt, a.batch, true);
	g.ntx = (a.W + TW - 1) / TW;
	g.swz = t.xcd_swizzle;
	const int Hd = (a.H + 1) / 2;
	const int waves = t.waves >= 1 && t.waves <= 4 ? t.waves : 4;
	const int nty = (Hd + g.tile_pairs - 1) / g.tile_pairs;
	int ring = t.ring_inv == 16 ? 16 : 8;
	// the waves of a workgroup side by side where the row of tiles has room for them (round 4, tile heights tuned for
	// either layout: 32 images + 0.2 %, 8 images + 1.5 %, one image 168.7 -> 164.5 us)
	g.wave_horiz = g.ntx >= waves;
	dim3 grid;
	if (g.wave_horiz)
		grid = dim3(((g.ntx + waves - 1) / waves) * nty, a.batch);
	else
		grid = dim3(g.ntx * ((nty + waves - 1) / waves), a.batch);
	if (a.sh.rows && (!a.interleaved || a.batch != 1 || a.in_step != 1 || g.tile_pairs != a.sh.tile_pairs))
		return hipErrorInvalidValue; // the snapshot of an in-place level was taken for other tiles
	if (a.interleaved) {
		if (a.in_step < 1 || (a.in_ll2 && (a.ll2_pitch < (a.W + 1) / 2)))
			return hipErrorInvalidValue;
		if constexpr (std::is_same<W, Cdf97S>::value) {
			if (strip) {
				if (a.batch != 1 || a.W < 64 || a.H < 64)
					return hipErrorInvalidValue;
				return a.in_ll2 ? inv_launch_x<W, true>(a, g, grid, waves, *strip, s) : inv_launch_x<W, false>(a, g, grid, waves, *strip, s);
			}
		}
		if (strip)
			return hipErrorInvalidValue;
		if constexpr (std::is_base_of<Cdf97S, W>::value || std::is_base_of<Cdf53S, W>::value) {
			if (a.in_ll2)
				return inv_launch_il_sp<W>(a, g, grid, waves, s);
			return inv_launch<W, 4, 8, 0, true>(a, g, grid, waves, s);
		} else {
			return hipErrorInvalidValue;
		}
	}
	if (strip)
		return hipErrorInvalidValue;
	return cpt == 8 ? inv_pick<W, 8>(a, g, grid, waves, ring, s) : inv_pick<W, 4>(a, g, grid, waves, ring, s);
}

int il_sweep_tile_pairs(const SweepTuning &t, int W, int H, bool inverse)
{
	return pick_tile_pairs(t, W, H, 4, 1, inverse);
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

hipError_t launch_inv_level(Wavelet w, const InvLevelArgs &a, const SweepTuning &t, hipStream_t s, const IlStripArgs *strip)
{
	if (strip && w != kCdf97S)
		return hipErrorInvalidValue;
	switch (w) {
	case kCdf97S: return inv_level_t<Cdf97S>(a, t, s, strip);
	case kCdf53I: return inv_level_t<Cdf53I>(a, t, s);
	case kCdf53S: return inv_level_t<Cdf53S>(a, t, s);
	case kCdf97I: return inv_level_t<Cdf97I>(a, t, s);
	case kCdf97SFma: return inv_level_t<Cdf97SFma>(a, t, s);
	default: break;
	}
	return hipErrorInvalidValue;
}


bool have_fused_inverse(Wavelet) { return true; }

} // namespace dwt
