// dwt_sweep2d.h -- what the forward (dwt_sweep2d.hip) and inverse (dwt_sweep2d_inv.hip) tile sweeps share: the
// tile geometry handed to the kernels and the launchers' rules for columns per lane and tile height.
#pragma once
#include "dwt_device.h"
#include "dwt_il_strip.h"

namespace dwt {

struct SweepGeom {
	int tile_pairs, ntx, swz;
	int wave_horiz; // 1: the waves of a workgroup take horizontally adjacent tiles
	int first = 0;  // k_*_sweep_x: the leading workgroups of the launch that take border strips, not tiles
	int tile_blocks = 0; // k_*_sweep_r: the workgroups [0, tile_blocks) take tiles, the ones behind them copy blocks
};

static inline int pick_cpt(const SweepTuning &t, int W, bool inverse)
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

static inline int pick_tile_pairs(const SweepTuning &t, int W, int H, int cpt, int batch, bool inverse = false, bool interleaved = false)
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
	// the inverse sweep (256-column tiles, twice the waves): 16 pairs.  Round 6, level 0 alone, rotating images, two
	// boxes (scripts/r06/inv_geometry.py): 16 against 32 pairs 103 / 107 us for one 8192^2 image, 754 / 767 us for 8,
	// 5974 / 6096 us for 64, 57 / 60 us at 8192 x 4096, 189 / 197 at 16384 x 8192, 74 / 84 at 7000 x 5000, 408 / 415 for
	// 16 x 4096^2; 8, 12, 20, 24 pairs and 512-column tiles of any height are slower (rounds 3-5 ranked 32 first on 8 images)
	// (the interleaved layout's inverse keeps 32: its in-place levels snapshot 9 rows per boundary between tile rows, and
	// twice the boundaries cost the in-place entry 13 us of 271)
	int tp = inverse ? (interleaved ? 32 : t.inv_pairs >= 2 ? t.inv_pairs : 16) : 64;
	const long want = inverse ? 2048 : 1024; // inverse tiles are half as wide
	while (tp > 8 && ntx * ((Hd + tp - 1) / tp) * batch < want)
		tp >>= 1;
	return tp;
}

} // namespace dwt
