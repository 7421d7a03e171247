// dwt_il_strip.h -- interleaved layout: the exact border strips of a fused level, computed by extra
// workgroups of the level's own sweep launch (dwt_sweep2d.hip, k_fwd_sweep_x / k_inv_sweep_x).
//
// The fused sweep finishes rows before columns; the reference's 9/7 in-place drivers and fdwt2_* run
// each line transform's phases over all rows, then all columns, before the next phase
// (src/dwt-simple.c:2266-2350, src/libdwt.c:12970-13480, 17517-17594), which rounds differently in the
// top 8 rows and the last 5 columns of a level only.  Those samples -- rows 0..7, the last 8 columns --
// are left out by the sweep's tiles (their stores are masked) and computed here instead, from the level's
// INPUT, in the reference's order.  A WAVE takes a tile of such a strip: 64 positions along it (40 kept
// plus a 12-sample margin either side), one per lane, and the 14 samples across it (from the image
// border) in each lane's registers.  It runs the six phase passes (rows' prolog, columns' prolog, rows'
// core, columns' core, rows' epilog, columns' epilog) and writes the part no artificial tile edge can
// have reached (4 samples per pass along the strip, 12 in all; 4 across it) to the level's output and
// to the dense low-pass copy the next level reads.  A pass is the reference's line kernel restricted to
// the phase: x[i] += c * (x[i-1] + x[i+1]) over the index range the phase owns, step after step, TRUE
// line indices, mirrored neighbours at the image border.  A pass across the strip works on a lane's own
// registers; a pass along it takes the neighbours from the adjacent lanes by wavefront shifts (DPP): no
// LDS, no barrier -- a strip tile is one serial chain, and the deep levels of a transform last as long
// as that chain does.  The right strip is held transposed (lanes along y), so both strips run the same code.
//
// Strip waves and sweep waves of one launch read the same input and write disjoint samples, so a level
// leaves its launch exact: nothing to correct afterwards, no chain of dependent launches.
#pragma once
#include "dwt_device.h"

namespace dwt {

constexpr int kIlKeepTop = 8, kIlKeepRight = 8; // rows from the top / columns from the right the strips own
constexpr int kIlStripKeep = 40;                // samples along a strip per wave

// workgroups of `waves` waves that the strips of an lx x ly level take
static inline int il_strip_blocks(int lx, int ly, int waves)
{
	const int tiles = (lx + kIlStripKeep - 1) / kIlStripKeep + (ly + kIlStripKeep - 1) / kIlStripKeep;
	return (tiles + waves - 1) / waves;
}

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
			// (a true line end: l and r are one sample, the reference adds (2c)*x -- dwt_lift.h)
			const bool end = i == 0 || i == N - 1;
			const T nv = INV ? inv_step_at<W>(st, end, v[j], l, r) : fwd_step_at<W>(st, end, v[j], l, r);
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

// one phase along the strip: lane <-> true index i of a line, NE lines in the lane's registers; the tile holds [o, e)
template <class W, bool INV, int NE>
static __device__ __forceinline__ void il_phase_lanes(typename W::T (&v)[NE], int i, int N, int o, int e, const IlPhase &ph)
{
	using T = typename W::T;
	constexpr int K = W::K;
	const bool scaled = i >= ph.sc_lo && i <= ph.sc_hi;
	if (INV) {
#pragma unroll
		for (int j = 0; j < NE; j++)
			v[j] = scaled ? W::inv_scale(i & 1, v[j]) : v[j];
	}
	// an update needs both neighbours inside the tile -- except at the true ends of the line, where the
	// missing neighbour is the mirror image of the other
	const int lo_t = o == 0 ? 0 : o + 1, hi_t = e == N ? N - 1 : e - 2;
#pragma unroll
	for (int st = 0; st < K; st++) {
		const int par = INV ? (st & 1) : !(st & 1);
		const bool ok = (i & 1) == par && i >= max(ph.lo[st], lo_t) && i <= min(ph.hi[st], hi_t);
#pragma unroll
		for (int j = 0; j < NE; j++) {
			const T fl = from_bits<T>(from_left_lane(to_bits(v[j]))), fr = from_bits<T>(from_right_lane(to_bits(v[j])));
			const T l = i == 0 ? fr : fl, r = i == N - 1 ? fl : fr;
			const bool end = i == 0 || i == N - 1; // (both taps one sample: the reference's end form)
			const T nv = INV ? inv_step_at<W>(st, end, v[j], l, r) : fwd_step_at<W>(st, end, v[j], l, r);
			v[j] = ok ? nv : v[j];
		}
	}
	if (!INV) {
#pragma unroll
		for (int j = 0; j < NE; j++)
			v[j] = scaled ? W::fwd_scale(i & 1, v[j]) : v[j];
	}
}

// one wave, one tile: `tile` counts the top strip's tiles, then the right strip's; every lane of the wave must be here
template <class W, bool INV>
static __device__ __forceinline__ void il_strip_wave(const IlStripArgs &a, int tile)
{
	using T = typename W::T;
	// kept across the strip: 8 rows from the top / the last 8 columns.  The reference's order differs from
	// the sweep's in rows 0..6 and in the last 5 (6 with the parity) columns only.  Across the strip only ONE pass
	// meets the band's artificial edge -- the core of the lines across it (their prolog / epilog stays within 8
	// samples of the image border, or far away on the other side) --, and that pass spoils 4 samples: a band of
	// 13-14 samples keeps 8 good ones.  Along the strip all three passes can meet a tile's edges: 12 a side.
	constexpr int kKeep = kIlStripKeep, kMargin = 12, kBand = 14;
	static_assert(kKeep + 2 * kMargin == 64 && kKeep % 2 == 0, "a tile is a wave wide and starts on an even index");
	const int lane = threadIdx.x & 63;
	const int n_top = (a.lx + kKeep - 1) / kKeep;
	const bool top = tile < n_top;
	const int t = top ? tile : tile - n_top;
	// "long" axis: along the strip (x for the top strip, y for the right one); "short": across it
	const int n_long = top ? a.lx : a.ly, n_short = top ? a.ly : a.lx;
	const int l0 = t * kKeep;
	if (l0 >= n_long)
		return; // (the last workgroup's spare waves)
	const int ol = max(0, l0 - kMargin), el = min(n_long, l0 + kKeep + kMargin); // tile range along the strip: at most 64, from an even index
	// the band starts on an even index (13 or 14 samples for the right strip): register parity = index parity
	const int os = top ? 0 : max(0, n_short - (kBand - 1)) & ~1, es = top ? min(n_short, kBand) : n_short;
	const int ns = es - os;
	const int il = ol + lane;
	const bool here = il < el;
	T v[kBand];
	{
		const int yl = top ? os : il, xl = top ? il : os; // the lane's first sample
#pragma unroll
		for (int j = 0; j < kBand; j++) {
			const int y = top ? yl + j : yl, x = top ? xl : xl + j;
			const bool ok = here && j < ns;
			// (an in-place level: the tiles may have overwritten the input by now -- the strips read their snapshots)
			const T *p = (a.ll_in && !((x | y) & 1)) ? a.ll_in + (long)(y >> 1) * a.ll_in_pitch + (x >> 1)
				: (top && a.top_in) ? a.top_in + (long)y * a.top_in_pitch + x
				: (!top && a.right_in) ? a.right_in + (long)y * a.right_in_pitch + (x - a.right_x0)
				: a.in + (long)y * a.in_pitch + (long)x * a.in_step;
			v[j] = ok ? *p : T(0);
		}
	}
#pragma unroll 1
	for (int pass = 0; pass < 6; pass++) {
		const bool rows = !(pass & 1);
		const bool along = top ? rows : !rows;
		const IlPhase ph = rows ? a.rph[pass >> 1] : a.cph[pass >> 1];
		const int N = rows ? a.lx : a.ly;
		{
			// a phase that owns no index inside the tile leaves it as it is
			const int o = along ? ol : os, e = along ? el : es;
			bool touches = ph.sc_lo <= ph.sc_hi && ph.sc_hi >= o && ph.sc_lo < e;
#pragma unroll
			for (int st = 0; st < W::K; st++)
				touches = touches || (ph.lo[st] <= ph.hi[st] && ph.hi[st] >= o && ph.lo[st] < e);
			if (!touches)
				continue;
		}
		if (along)
			il_phase_lanes<W, INV, kBand>(v, il, N, ol, el, ph);
		else
			il_phase_piece<W, INV, kBand>(v, os, N, os, es, ph);
	}
	// kept part: along the strip [l0, l0 + kKeep), across it the 8 rows from the top / the last 8 columns
	if (il < l0 || il >= min(n_long, l0 + kKeep))
		return;
	const int ks0 = top ? 0 : max(0, n_short - kIlKeepRight), ks1 = top ? min(n_short, kIlKeepTop) : n_short;
#pragma unroll
	for (int j = 0; j < kBand; j++) {
		const int sh = os + j;
		if (sh >= ks0 && sh < ks1) {
			const int y = top ? sh : il, x = top ? il : sh;
			a.out[(long)y * a.out_pitch + (long)x * a.out_step] = v[j];
			if (a.ll && !((x | y) & 1))
				a.ll[(long)(y >> 1) * a.ll_pitch + (x >> 1)] = v[j];
		}
	}
}

} // namespace dwt
