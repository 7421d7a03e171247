// dwt_sweep2d_inv.hip -- the inverse tile sweep of the 2-D lifting DWT (see dwt_sweep2d.hip for the scheme) and its
// launchers.  A translation unit of its own: the two sweeps compile side by side.
#include "dwt_sweep2d.h"

namespace dwt {

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
	// (SelEnds: levels of 64 x 64 and more -- no index is reflected twice, no integer division in the wave's instruction stream)
	constexpr bool kNear = kIsSelEnds<W>;
	// (float policies whose step is c + k (l + r) rounded product-then-sum: the horizontal lift takes both rows at once)
	constexpr bool kPairRows = kIsSelEnds<W> && std::is_same<T, float>::value && has_coef_ends<W>::value && !std::is_base_of<Cdf97SFma, W>::value;
	const bool tall = kNear || a.H >= 64; // then a row index leaves [0,H) by less than H: one bounce
	const int edge_col = (kNear ? reflect_near(2 * edge_sub + ((lane >> 1) & 1), a.W) : reflect(2 * edge_sub + ((lane >> 1) & 1), a.W)) >> 1;
	// halo lanes 0..7 -> L halo, 8..15 -> H halo
	const int hsub = (lane & 7) < 4 ? cl0 - 4 + (lane & 7) : cl0 + M + (lane & 3);
	const int halo_col = (kNear ? reflect_near(2 * hsub + ((lane >> 3) & 1), a.W) : reflect(2 * hsub + ((lane >> 3) & 1), a.W)) >> 1;
	const bool halo_is_h = (lane >> 3) & 1;

	// pointers to the four subbands' row starts are formed per source row
	// interleaved input: source columns of the element-wise loader and of the halo
	int halo_colI = 0;
	if constexpr (IL) {
		const int hi_ = lane < 4 ? c0 - 4 + lane : c0 + TW + (lane & 3);
		halo_colI = kNear ? reflect_near(hi_, a.W) : reflect(hi_, a.W);
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
				const int r = tall ? reflect1(2 * p + rr, a.H) : reflect(2 * p + rr, a.H);
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
					dma4<kLdAux>((rgt ? rgt : grow) + (long)(kNear ? a.W - 2 - min(lane, 3) : reflect(a.W + min(lane, 3), a.W)) * step, lrow + (a.W - c0) * 4);
				if (lane < 8)
					dma4<kLdAux>((crow && halo_shI >= 0) ? crow + halo_shI : (rgt && halo_colI >= wr) ? rgt + halo_colI : grow + (long)halo_colI * step, lrow + TW * 4);
			}
			return;
		}
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			// rr = 0: L row p (interleaved row 2p); rr = 1: H row p (row 2p+1)
			const int rs = tall ? reflect1(2 * p + rr, a.H) : reflect(2 * p + rr, a.H);
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

	// explicit line-end forms: the ends of a row among each group's samples c - K + 1 .. c + CG + K - 1 -- only the tiles
	// that hold column 0 or W - 1 have any (`h_any`, wave-uniform: the interior tiles run the plain lift) --; the rows that
	// are a column's ends are found per iteration (wave-uniform as well)
	[[maybe_unused]] unsigned hends[G] = {};
	[[maybe_unused]] bool h_any = false, h_simple = false;
	// the two entries of a group's window that meet a line end when the level's width is a multiple of CG: column 0 is the
	// group's own first sample, column W - 1 its own last one (entries 0 and NARR - 1 are never acted on)
	constexpr unsigned kCand = (1u << (K - 1)) | (1u << (K + CG - 2));
	if constexpr (W::kEndForms) {
		unsigned all = 0;
#pragma unroll
		for (int gi = 0; gi < G; gi++) {
			hends[gi] = end_mask<NARR>(c0 + 64 * CG * gi + lane * CG - K + 1, a.W);
			all |= hends[gi];
		}
		h_any = !a.plain_ends && __builtin_amdgcn_ballot_w64(all != 0) != 0;
		h_simple = __builtin_amdgcn_ballot_w64((all & ~(kCand | 1u | (1u << (NARR - 1)))) != 0) == 0;
	}
	// (SelEnds) the same two entries by selection: per group the lane's flags and its coefficients for the steps that reach them
	[[maybe_unused]] bool e0[G] = {}, e1[G] = {};
	[[maybe_unused]] T kh[G][K];
	if constexpr (kIsSelEnds<W>) {
#pragma unroll
		for (int gi = 0; gi < G; gi++) {
			const unsigned m = end_mask_long<NARR>(c0 + 64 * CG * gi + lane * CG - K + 1, a.W);
			e0[gi] = (m >> (K - 1)) & 1;
			e1[gi] = (m >> (K + CG - 2)) & 1;
			sel_coefs<W, true, K - 1>(kh[gi], e0[gi], e1[gi]);
		}
	}
	[[maybe_unused]] auto row_is_end = [&](int r) {
		if (tall || kIsSelEnds<W>) // (SelEnds runs on levels of 64 rows or more)
			return r == 0 || r == a.H - 1;
		const int rr = reflect(r, a.H);
		return rr == 0 || rr == a.H - 1;
	};
	// the horizontal inverse lift of one register row: the end forms only where the tile has a line end
	auto hlift = [&](T (&xr)[NARR], int gi) {
		if constexpr (W::kEndForms) {
			if (__builtin_expect(!h_any, 1)) {
				lift_inv_regs<W, NARR>(xr, 0u);
			} else if (h_simple) {
				DWT_END_PATH();
				lift_inv_regs<W, NARR, kCand>(xr, hends[gi]);
			} else {
				DWT_END_PATH();
				lift_inv_regs<W, NARR>(xr, hends[gi]);
			}
		} else if constexpr (kIsSelEnds<W>) {
			lift_regs_sel<W, NARR, true, K - 1, K + CG - 2>(xr, e0[gi], e1[gi], kh[gi]);
		} else {
			lift_inv_regs<W, NARR>(xr, 0u);
		}
	};

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
					x[rr][0][j] = kPairRows ? v : W::inv_scale(rel & 1, v); // (kPairRows: descaled below, both rows at once)
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
						x[rr][gi][j] = kPairRows ? h[2 + ((rel - 1) >> 1)] : W::inv_scale(1, h[2 + ((rel - 1) >> 1)]);
					else
						x[rr][gi][j] = kPairRows ? l[2 + (rel >> 1)] : W::inv_scale(0, l[2 + (rel >> 1)]);
				}
			}
		}

		T val[2][G][NVG]; // val[0] = L row p, val[1] = H row p as the vertical pass sees them
#pragma unroll
		for (int gi = 0; gi < G; gi++) {
			if constexpr (kPairRows) {
				// both rows of the iteration at once, as the halves of packed operations (the selects at the two candidate
				// entries would otherwise split the pairs the compiler forms within a row): the same steps, the same rounding
				typedef float f2 __attribute__((ext_vector_type(2)));
				f2 x2[NARR];
				const float z0 = W::inv_scale(0, 1.0f), z1 = W::inv_scale(1, 1.0f); // (the descaling factors themselves)
#pragma unroll
				for (int j = 0; j < NARR; j++)
					x2[j] = f2{x[0][gi][j], x[1][gi][j]} * (((j - K + 1) & 1) ? z1 : z0);
#pragma unroll
				for (int s_ = 0; s_ < K; s_++) {
#pragma unroll
					for (int j = s_ + 1; j <= NARR - 2 - s_; j += 2) {
						if (j == K - 1)
							x2[j] = x2[j] + kh[gi][s_] * ((e0[gi] ? f2{-0.0f, -0.0f} : x2[j - 1]) + x2[j + 1]);
						else if (j == K + CG - 2)
							x2[j] = x2[j] + kh[gi][s_] * (x2[j - 1] + (e1[gi] ? f2{-0.0f, -0.0f} : x2[j + 1]));
						else
							x2[j] = x2[j] + W::ik(s_) * (x2[j - 1] + x2[j + 1]);
					}
				}
#pragma unroll
				for (int v = 0; v < CG; v++) {
					const f2 sc = x2[K - 1 + v] * f2{z0, z1}; // (the vertical pass descales by ROW parity)
					val[0][gi][v] = sc[0];
					val[1][gi][v] = sc[1];
				}
			} else if constexpr (!W::kInvColsFirst) {
#pragma unroll
				for (int rr = 0; rr < 2; rr++) {
					hlift(x[rr][gi], gi);
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
		// step s of this iteration acts on row 2p-s: which of them are column ends (wave-uniform; almost never any)
		[[maybe_unused]] bool vend[K] = {};
		[[maybe_unused]] bool v_any = false;
		if constexpr (W::kEndForms) {
#pragma unroll
			for (int s_ = 0; s_ < K; s_++) {
				vend[s_] = row_is_end(2 * p - s_);
				v_any = !a.plain_ends && (v_any || vend[s_]);
			}
		}
		T odd_row[G][NVG], even_row[G][NVG];
		auto vertical = [&](auto ends_tag) {
			constexpr bool ENDS = decltype(ends_tag)::value;
#pragma unroll
			for (int gi = 0; gi < G; gi++)
#pragma unroll
			for (int v = 0; v < NVG; v++) {
				const T s2 = val[0][gi][v], d2 = val[1][gi][v];
				if constexpr (K == 4) {
					// st: [0] d2[p-1], [1] s1[p-1], [2] d1[p-2], [3] e[p-2]
					const T s1n = inv_step_at<W>(0, ENDS && vend[0], s2, st[0][gi][v], d2);               // s1[p]
					const T d1n = inv_step_at<W>(1, ENDS && vend[1], st[0][gi][v], st[1][gi][v], s1n);    // d1[p-1]
					const T en = inv_step_at<W>(2, ENDS && vend[2], st[1][gi][v], st[2][gi][v], d1n);     // e[p-1]
					const T on = inv_step_at<W>(3, ENDS && vend[3], st[2][gi][v], st[3][gi][v], en);      // o[p-2]
					odd_row[gi][v] = on;
					even_row[gi][v] = en;
					st[0][gi][v] = d2;
					st[1][gi][v] = s1n;
					st[2][gi][v] = d1n;
					st[3][gi][v] = en;
				} else {
					// st: [0] d[p-1], [1] e[p-1]
					const T en = inv_step_at<W>(0, ENDS && vend[0], s2, st[0][gi][v], d2);                // e[p]
					const T on = inv_step_at<W>(1, ENDS && vend[1], st[0][gi][v], st[1][gi][v], en);      // o[p-1]
					odd_row[gi][v] = on;
					even_row[gi][v] = en;
					st[0][gi][v] = d2;
					st[1][gi][v] = en;
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
			// step s acts on row 2p-s; where that row is an end of its column (wave-uniform, the top and bottom tiles' first /
			// last iterations) the step's coefficient is doubled and its state tap -- the same values as the other tap
			// there -- gives way to -0.0.  Deep ring: the iterations that meet no end take the plain body.
			bool ve[K], any = false;
#pragma unroll
			for (int s_ = 0; s_ < K; s_++) {
				ve[s_] = row_is_end(2 * p - s_);
				any = any || ve[s_];
			}
			auto vertical_sel = [&]() {
				T kv[K];
#pragma unroll
				for (int s_ = 0; s_ < K; s_++)
					kv[s_] = sel_coef<W, true>(s_, ve[s_]);
#pragma unroll
				for (int gi = 0; gi < G; gi++)
#pragma unroll
				for (int v = 0; v < NVG; v++) {
					const T s2 = val[0][gi][v], d2 = val[1][gi][v];
					if constexpr (K == 4) {
						const T s1n = sel_step<W, true>(0, ve[0], kv[0], s2, st[0][gi][v], d2);
						const T d1n = sel_step<W, true>(1, ve[1], kv[1], st[0][gi][v], st[1][gi][v], s1n);
						const T en = sel_step<W, true>(2, ve[2], kv[2], st[1][gi][v], st[2][gi][v], d1n);
						const T on = sel_step<W, true>(3, ve[3], kv[3], st[2][gi][v], st[3][gi][v], en);
						odd_row[gi][v] = on;
						even_row[gi][v] = en;
						st[0][gi][v] = d2;
						st[1][gi][v] = s1n;
						st[2][gi][v] = d1n;
						st[3][gi][v] = en;
					} else {
						const T en = sel_step<W, true>(0, ve[0], kv[0], s2, st[0][gi][v], d2);
						const T on = sel_step<W, true>(1, ve[1], kv[1], st[0][gi][v], st[1][gi][v], en);
						odd_row[gi][v] = on;
						even_row[gi][v] = en;
						st[0][gi][v] = d2;
						st[1][gi][v] = en;
					}
				}
			};
			// (shallow ring: the launches of a few rounds of waves, bound by the longest wave -- the top tiles', for whom a
			// second body means instructions fetched cold from HBM, 1 us a launch; there every iteration selects)
			if constexpr (RING == 0)
				vertical_sel();
			else if (__builtin_expect(any, 0)) {
				DWT_END_PATH();
				vertical_sel();
			} else
				vertical(std::false_type{});
		} else {
			vertical(std::false_type{});
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
				hlift(odd_row[gi], gi);
				hlift(even_row[gi], gi);
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

// the tile by the instantiation of the policy's line ends the level needs (dwt_lift.h)
template <class W, int CPT, int RING, int NT, bool IL, bool X, bool SP = false>
static __device__ __forceinline__ void inv_sweep_any_tile(const InvLevelArgs &a, const SweepGeom &g)
{
	if constexpr (W::kEndForms) {
		constexpr int CG = (!IL && CPT == 8) ? 4 : CPT; // columns per group, as in the tile
		if (a.plain_ends)
			inv_sweep_tile<PlainEnds<W>, CPT, RING, NT, IL, X, SP>(a, g);
		else if (a.W % CG == 0 && a.W >= 64 && a.H >= 64)
			inv_sweep_tile<SelEnds<W>, CPT, RING, NT, IL, X, SP>(a, g);
		else // (a width that puts the last column anywhere in a group's window; short lines, reflected more than once)
			inv_sweep_tile<W, CPT, RING, NT, IL, X, SP>(a, g);
	} else
		inv_sweep_tile<W, CPT, RING, NT, IL, X, SP>(a, g);
}

template <class W, int CPT, int RING, int NT, bool IL = false>
__global__ __launch_bounds__(256) void k_inv_sweep(InvLevelArgs a, SweepGeom g)
{
	inv_sweep_any_tile<W, CPT, RING, NT, IL, false>(a, g);
}

// interleaved input, 4 columns per lane; SP: split even rows (InvLevelArgs::in_ll2)
// a level with a rectangle copy riding along (InvLevelArgs::ride; see k_fwd_sweep_r)
template <class W, int CPT, int RING, int NT>
__global__ __launch_bounds__(256) void k_inv_sweep_r(InvLevelArgs a, SweepGeom g, CopyRects r)
{
	if ((int)blockIdx.x >= g.tile_blocks) {
		ride_copy_block(r, (int)blockIdx.x - g.tile_blocks);
		return;
	}
	inv_sweep_any_tile<W, CPT, RING, NT, false, false>(a, g);
}

template <class W, int RING, int NT, bool SP>
__global__ __launch_bounds__(256) void k_inv_sweep_il(InvLevelArgs a, SweepGeom g)
{
	inv_sweep_any_tile<W, 4, RING, NT, true, false, SP>(a, g);
}

template <class W, int RING, int NT, bool SP>
__global__ __launch_bounds__(256) void k_inv_sweep_x(InvLevelArgs a, SweepGeom g, IlStripArgs strip)
{
	if ((int)blockIdx.x < g.first) {
		il_strip_wave<W, true>(strip, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
		return;
	}
	inv_sweep_any_tile<W, 4, RING, NT, true, true, SP>(a, g);
}

// ---- launch wrappers -------------------------------------------------------------
template <class W, int CPT, int RING, int NT, bool IL>
static hipError_t inv_launch(const InvLevelArgs &a, const SweepGeom &g, dim3 grid, int waves, hipStream_t s)
{
	const size_t lds = (size_t)waves * RING * (64 * CPT + 16) * 4;
	if constexpr (!IL && RING == 8) {
		if (a.ride && a.ride_hi > a.ride_lo) {
			if (a.batch != 1 || waves != 4)
				return hipErrorInvalidValue;
			if (hipError_t e = allow_lds((const void *)k_inv_sweep_r<W, CPT, RING, NT>, lds))
				return e;
			SweepGeom gr = g;
			gr.tile_blocks = (int)grid.x;
			CopyRects r = *a.ride;
			r.block0 = a.ride_lo;
			k_inv_sweep_r<W, CPT, RING, NT><<<dim3(grid.x + (unsigned)(a.ride_hi - a.ride_lo), 1), 64 * waves, lds, s>>>(a, gr, r);
			return hipGetLastError();
		}
	}
	if (a.ride && a.ride_hi > a.ride_lo)
		return hipErrorInvalidValue;
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
	// the result of a level that is not the last is the next level's low-pass input: temporal stores let it wait in the
	// 256 MiB Infinity Cache (the forward's LL bands alike, cache policy bit 2 there)
	if (a.temporal_out)
		return inv_launch<W, CPT, 8, 0, false>(a, g, grid, waves, s);
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
	g.tile_pairs = pick_tile_pairs(t, a.W, a.H, cpt, a.batch, true, a.interleaved != 0);
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
