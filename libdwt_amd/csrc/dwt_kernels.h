// dwt_kernels.h -- launch interface between the backend (dwt_backend.hip) and the
// HIP kernels (dwt_sweep2d.hip, dwt_vol3d.hip, dwt_interleaved.hip).  Internal to the shared library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// 1: timing probes are compiled in (`make probes` -> ../libdwt_hip_probes.so; scripts/r06/probe_fuse1.py).  A probe gives
// WRONG RESULTS by design; the shipped libraries are built without them and ignore the options that select them.
#ifndef DWT_PROBES
#define DWT_PROBES 0
#endif

namespace dwt {

enum Wavelet { kCdf97S = 0, kCdf53I = 1, kCdf53S = 2, kCdf97D = 3, kCdf53D = 4, kCdf97I = 5,
	kCdf97SFma = 6 /* internal: float 9/7 with contracted steps, option "fma" */,
	kCdf53SNew = 7 /* internal: float 5/3 of dwt-simple.c (odd scale 1/zeta in float), interleaved layout only */,
	kCdf97IIp = 8 /* internal: fixed-point int 9/7 of the interleaved in-place drivers (rounded terms added) */ };

inline int elem_size(Wavelet w) { return (w == kCdf97D || w == kCdf53D) ? 8 : 4; }

// Tuning knobs of the fused sweep kernels (set through dwt_hip_set_option).
struct SweepTuning {
	int cpt = 0;        // columns per lane: 4 or 8; 0 = choose from the level width
	int tile_pairs = 0; // output row pairs per wave tile; 0 = choose from the level height
	int waves = 4;      // waves per workgroup (each wave owns one tile)
	int xcd_swizzle = 1; // remap workgroups so neighbouring tiles share an XCD's L2
	int ring = 0;        // LDS ring rows per wave (8 or 16); 0 = auto
	int nt = 7;          // forward cache policy: loads and detail stores are non-temporal; bit 2 keeps the LL band's
	                     // stores temporal (the next level reads it), bit 3 takes the neighbour taps by wavefront shifts
	int nt_auto = 1;     // forward: drop bit 2 of `nt` when the launch's LL bands exceed the Infinity Cache
	int ring_inv = 8;    // inverse sweep ring rows (8 or 16)
	int probe_fuse1 = 0; // see FwdLevelArgs (timing probe, wrong results)
	int inv_pairs = 0;   // inverse: tile height (row pairs) of the large levels under the launcher's rule; 0 = 16 (32 for the levels of an in-place call)
	int inv_ll_temporal = 1; // inverse: a level that is not the last stores its result temporal when it fits the Infinity Cache
};

// In-place level of the interleaved layout (input image == output image): a snapshot of what a tile reads of its
// NEIGHBOURS' samples, taken before the level's launch (launch_il_shell).  A tile's own samples are still unwritten
// when it reads them (its stores trail its loads); everything else it reads comes from here.  Tiles of 256 columns x
// `tile_pairs` row pairs.
struct IlShell {
	const float *rows = nullptr;  // 9 rows around every boundary between tile rows: boundary k (rows 2 k tile_pairs - 5 .. + 3)
	long rows_pitch = 0;          //   at slot 9 (k - 1) + i; whole rows
	const float *cols = nullptr;  // per image row the 8 columns around every boundary between tile columns: boundary b
	long cols_pitch = 0;          //   (columns 256 (b + 1) - 4 .. + 3) at 8 b
	const float *top = nullptr;   // rows 0 .. 13, whole (input of the top border strip)
	long top_pitch = 0;
	const float *right = nullptr; // columns right_x0 .. W - 1 of every row (input of the right border strip)
	long right_pitch = 0;
	int right_x0 = 0;
	int tile_pairs = 0;
};
constexpr int kIlShellRight = 20; // floats per row of IlShell::right
// bytes of scratch a shell of a W x H level takes; 0 = this shape cannot run in place (narrow last tile column, short tiles)
size_t il_shell_bytes(int W, int H, int tile_pairs);
// fills the shell from the image (pitch in elements) into `scratch` (il_shell_bytes) and returns its description
hipError_t launch_il_shell(const float *img, long pitch, int W, int H, int tile_pairs, float *scratch, IlShell *sh, hipStream_t s);
// the tile height launch_fwd_level / launch_inv_level give an interleaved single-image level under these settings
int il_sweep_tile_pairs(const SweepTuning &t, int W, int H, bool inverse);
// whether launch_fwd_level / launch_inv_level take a kernel that can carry a copy along (FwdLevelArgs::ride) for this Mallat level
bool sweep_ride_ok(const SweepTuning &t, int W, int H, int batch, bool inverse);

// One decomposition level, forward, dense frame (size_o == size_i, W,H >= 2).
// Reads the W x H region at `in`; writes LL (ceil(W/2) x ceil(H/2)) to `out_ll` and
// the three detail subbands at their Mallat offsets relative to `out_h`:
// HL at (0, Wd), LH at (Hd, 0), HH at (Hd, Wd), Wd = ceil(W/2), Hd = ceil(H/2).
// Pitches are in ELEMENTS.  `batch` images lie `*_bstride` elements apart.
struct CopyRects;
struct FwdLevelArgs {
	const void *in;
	long in_pitch, in_bstride;
	void *out_ll;
	long ll_pitch, ll_bstride;
	void *out_h;
	long h_pitch, h_bstride;
	int W, H, batch;
	int interleaved = 0; // 1: write rows/columns interleaved to out_h (3-D path / in-place lifting layout)
	int plain_ends = 0;  // 1: line ends as reflection gives them, c*(x+x), not the reference's end form (dwt_lift.h): the 3-D
	                     // path's xy sweep, which must give the bits of the fused 3-D level kernels (they keep the reflected form)
	int il_ll = 0;       // interleaved only: also write the LL samples densely to out_ll
	int temporal = 0;    // 1: every store temporal (the outputs are read again at once: staging of an in-place call)
	int pair_lo = 0, pair_hi = 0; // pair_hi > 0: only the tiles that start at a row pair in [pair_lo, pair_hi) -- multiples of 64 --
	                              // run (a level computed band by band while its input is still arriving over PCIe)
	IlShell sh;          // interleaved only, in place (in == out_h, out_step 1): the neighbours' samples come from this snapshot
	// Mallat layout, one image: copy blocks [ride_lo, ride_hi) of a rectangle copy that does not depend on this level run
	// as extra workgroups BEHIND the level's tiles in the same launch (the staged subbands of an in-place call going back
	// while the small, latency-bound levels run: launch_copy_rects_plan)
	const CopyRects *ride = nullptr;
	int ride_lo = 0, ride_hi = 0;
	int probe_fuse1 = 0; // PROBE ONLY (option "probe_fuse1"): level 0 also runs level 1's arithmetic on its LL rows and stores level 1's
	                     // four quarter rows instead of the LL band -- WRONG RESULTS (no halo between tiles), the cost of a fused level 0 + 1
	int out_step = 1;    // interleaved only: elements between neighbouring samples of an output row -- 2^j when the level
	                     // is written straight to the lattice it lives on in a larger image (h_pitch: that lattice's row
	                     // pitch); with il_ll the samples at (even row, even column) are then left to the deeper levels
};

// One reconstruction level, inverse, dense frame.  Reads LL from `in_ll` and the
// detail subbands at their Mallat offsets relative to `in_h`; writes the W x H
// interleaved result to `out`.
struct InvLevelArgs {
	const void *in_ll;
	long ll_pitch, ll_bstride;
	const void *in_h;
	long h_pitch, h_bstride;
	void *out;
	long out_pitch, out_bstride;
	int W, H, batch;
	int interleaved = 0; // 1: interleaved input: even rows at in_ll (row r/2), odd rows at in_h (row r/2)
	int plain_ends = 0;  // see FwdLevelArgs
	int temporal_out = 0; // Mallat: the result is the next level's low-pass input and fits the Infinity Cache: stored temporal
	int pair_lo = 0, pair_hi = 0; // pair_hi > 0: only the tiles that start at a row pair in [pair_lo, pair_hi) run (see FwdLevelArgs)
	const CopyRects *ride = nullptr; // see FwdLevelArgs
	int ride_lo = 0, ride_hi = 0;
	// interleaved only -- a level read straight from the lattice it lives on in a larger image:
	IlShell sh;                   // in place (in_ll == out, in_step 1): the neighbours' samples come from this snapshot
	int in_step = 1;              // elements between neighbouring samples of a source row (2^j on the lattice of level j)
	const void *in_ll2 = nullptr; // dense low-pass band (ceil(W/2) x ceil(H/2), the level below's result): replaces the
	long ll2_pitch = 0;           // samples at (even row, even column) of the source
};

// `strip` (interleaved layout, float 9/7 both ways and fdwt2_cdf53 forward, one image of 64 samples or more either
// way): the launch also computes the level's border strips in the reference's phase order and its tiles leave those
// samples alone (dwt_il_strip.h)
struct IlStripArgs;
hipError_t launch_fwd_level(Wavelet w, const FwdLevelArgs &a, const SweepTuning &t, hipStream_t s, const IlStripArgs *strip = nullptr);

hipError_t launch_inv_level(Wavelet w, const InvLevelArgs &a, const SweepTuning &t, hipStream_t s, const IlStripArgs *strip = nullptr);
// the same two sweeps for the double-precision wavelets (dwt_sweep2d_d.hip); pitches in 8-byte ELEMENTS
hipError_t launch_fwd_level_d(Wavelet w, const FwdLevelArgs &a, const SweepTuning &t, hipStream_t s);
hipError_t launch_inv_level_d(Wavelet w, const InvLevelArgs &a, const SweepTuning &t, hipStream_t s);
// true when launch_inv_level has a fused kernel for this wavelet
bool have_fused_inverse(Wavelet w);

// Generic out-of-place 1-D pass over `n_lines` strided lines of length N (exact
// reference semantics for any N; used for sparse frames, single-line directions and
// as the cross-check variant).  Strides in BYTES.  Forward writes L to dst[0..) and
// H to dst[hoff..); inverse reads L from src[0..), H from src[hoff..).
// `lanes_along_lines`: adjacent lanes take adjacent lines (column passes).
hipError_t launch_line_pass(Wavelet w, bool inverse, const void *src, void *dst, long line_stride, long elem_stride,
	int n_lines, int N, int hoff, bool lanes_along_lines, hipStream_t s);

// z-pass knobs of the 3-D path (measured defaults; options vol_cpt / vol_tile_pairs / vol_nt)
struct VolTuning {
	int cpt = 8;        // columns per lane: 4 or 8 (two groups of 4, 256 columns apart; +4 % at 1024^3)
	int tile_pairs = 0; // slice pairs per wave; 0 = choose from the volume depth
	int nt = -1;        // bit 0 non-temporal stores, bit 1 non-temporal loads; -1 = measured default
	                    // (z pass: 0, fused level: stores non-temporal, +9 %)
	int inplace_fused = 1; // in-place calls: 1 = one fused pass per level in place over a snapshot of the tile halos (forward and inverse),
	                       // 0 = two passes per level (the cross-check)
	int whole = 1;      // whole-tile variant of the fused kernel where the volume allows (0: the general one)
	int direct = 2;     // fused levels >= 1 write into their lattice of the destination: 2 = level 1 merged with level 0's withheld rows where the sizes allow, 1 = strided stores, 0 = dense volume + scatter pass
	int fused = 1;      // out-of-place forward levels: 1 = one fused pass where it pays, 2 = wherever it can run, 0 = two passes
	int swizzle = 1;    // fused level: hand contiguous runs of tiles to one XCD
	int rows = 8;       // fused level: output rows per wave, 8 (measured best) or 6 (two workgroups per CU)
	int ip_waves = 0;   // k_vol_level_ip: waves per workgroup, 4 (tiles of 32 rows, two workgroups per CU) or 8 (64 rows, one); 0 = 8 where the volume has more than 32 rows
};

// z pass of the 3-D path: CDF 9/7 float along the slice axis of an interleaved volume,
// out of place (in != out), x dense; strides in ELEMENTS.  Forward only: when `lll` is set the
// even-x/even-y/even-z samples (the next level's input) are also written densely there.
hipError_t launch_vol_z(bool inverse, const float *in, long in_sy, long in_sz, float *out, long out_sy, long out_sz,
	int nx, int ny, int nz, const VolTuning &vt, hipStream_t s, float *lll = nullptr, long lll_sy = 0, long lll_sz = 0);

// One forward 3-D level in ONE pass, out of place (in != out): x, y and z lifting fused.  A
// workgroup owns 256 x 32 voxel columns and marches along z; see k_vol_fwd_fused.  Applies
// to volumes at least 128 samples wide with about one workgroup per CU (vol_fused_applies);
// overhanging or unaligned tiles are staged column by column.
struct VolFusedArgs {
	const float *in;
	long in_sy, in_sz;
	float *out;
	long out_sy, out_sz;
	float *lll; // optional dense copy of the even-even-even samples (next level's input)
	long lll_sy, lll_sz;
	int nx, ny, nz;
	// multi-level store variants (see k_vol_fwd_fused): 1 = level j >= 1 into its lattice of the
	// destination (out_sx = 2^j, out_sy / out_sz the destination's strides times 2^j); 2 = level 0
	// withholding its even-y even-z rows (odd-x samples to `side`, laid out like `lll`); 3 = level 1
	// writing those rows whole (out_sy / out_sz the destination's strides times 2; `side` read)
	int mode = 0;
	long out_sx = 1;
	int temporal_shared = 0; // mode 3: the rows the next level's merge pass reads again are stored temporal
	float *side = nullptr;
	long side_sy = 0, side_sz = 0;
};
bool vol_fused_applies(const VolFusedArgs &a);
hipError_t launch_vol_fwd_fused(const VolFusedArgs &a, const VolTuning &vt, hipStream_t s);

// One 3-D level in ONE pass and IN PLACE (in == out, same strides), forward or inverse
// (dwt_vol3d_ip.hip): a snapshot of the SHELL -- the rows, columns and slices a tile of the fused
// kernel reads but does not own, about a quarter of the volume -- is taken into `scratch`
// (vol_level_ip_scratch bytes), then the fused kernel reads tile interiors from the volume and halos
// from the shell.  mode 0 (dense rows) or, forward only, 2 (see VolFusedArgs); `lll` as above.
struct VolShell {
	float *rs; // rows: [nz][7 (tile rows - 1)][nx]
	long rs_sy, rs_sz;
	float *cs; // columns: [nz][ny][8 (tile columns - 1)]
	long cs_sy, cs_sz;
	float *zs; // slices: [9 (marches - 1) + 5][ny][nx]
	long zs_sy, zs_sz;
	int tile_pairs_z, nzt; // slice pairs per march, marches along z
};
bool vol_level_ip_can(const VolFusedArgs &a);     // the kernel can run (any size from 2 x 2 x 8)
bool vol_level_ip_applies(const VolFusedArgs &a); // ... and pays
size_t vol_level_ip_scratch(const VolFusedArgs &a, const VolTuning &vt);
hipError_t launch_vol_level_ip(bool inverse, const VolFusedArgs &a, float *scratch, const VolTuning &vt, hipStream_t s);
// the same kernel OUT OF PLACE (in != out, dense source, no shell).  Forward: mode 0 / 2 and `lll` as for
// launch_vol_fwd_fused (tiles of 64 rows where the volume has them).  Inverse: mode 0 dense result, mode 1
// result into the stride-out_sx lattice of `out` (a level >= 1 of a multi-level inverse).
hipError_t launch_vol_level_op(bool inverse, const VolFusedArgs &a, const VolTuning &vt, hipStream_t s);

// Strided 3-D copy (lattice pack/unpack for the levels >= 1 of the 3-D path);
// strides in ELEMENTS, including the x strides.
hipError_t launch_lattice_copy(const float *src, long s_sx, long s_sy, long s_sz, float *dst, long d_sx, long d_sy, long d_sz,
	int nx, int ny, int nz, hipStream_t s);
// the same for a dense source and a destination whose rows hold dst_nx samples: strides 2 and 4 as
// a read-modify-write of whole 16-byte pieces where they fit the rows
hipError_t launch_lattice_scatter(const float *src, long s_sy, long s_sz, float *dst, long d_sx, long d_sy, long d_sz,
	int nx, int ny, int nz, int dst_nx, hipStream_t s);

// One PHASE of the reference's phase-ordered in-place lifting (src/dwt-simple.c:2266-2350,
// src/libdwt.c:17517-17594): lifting step s updates the coefficients lo[s]..hi[s] of its
// parity only, the scaling touches sc_lo..sc_hi only; everything else is copied.  Out of
// place, one thread per coefficient pair, interleaved layout on both sides.
struct IlPhase {
	int lo[4], hi[4];
	int sc_lo, sc_hi;
};
// k_lo / k_hi: only the coefficient pairs k_lo .. k_hi-1 of every line are computed and written
// (k_hi < 0: to the end of the line) -- the exact border strips of the fused interleaved path.
hipError_t launch_il_phase(Wavelet w, bool inverse, const void *src, void *dst, long line_stride, long elem_stride,
	int n_lines, int N, bool lanes_along_lines, const IlPhase &ph, hipStream_t s, int k_lo = 0, int k_hi = -1);

// The two exact border strips of a fused interleaved level (k_il_strip, one launch): the level's input (odd rows at
// `in`, even rows there too or packed at `in_even`), the sweep's output `out` it corrects, the
// optional dense low-pass copy `ll`; pitches in ELEMENTS; rph / cph: the prolog, core and epilog
// ranges of the row (N = lx) and column (N = ly) transforms.
struct IlStripArgs {
	const float *in;
	long in_pitch;
	int in_step;         // elements between neighbouring samples of an input row (a level on its lattice in a larger image)
	const float *ll_in;  // or null: dense low-pass band that replaces the input samples at (even row, even column)
	long ll_in_pitch;
	const float *top_in;   // or null (in-place level): rows 0 .. 13 of the input, IlShell::top
	long top_in_pitch;
	const float *right_in; // or null (in-place level): columns right_x0 .. of every input row, IlShell::right
	long right_in_pitch;
	int right_x0;
	float *out;
	long out_pitch;
	int out_step; // elements between neighbouring samples of an output row
	float *ll;
	long ll_pitch;
	int lx, ly;
	IlPhase rph[3], cph[3];
};


// Up to three device-to-device rectangle copies in one launch; widths in BYTES (multiples of 4).
struct CopyRects {
	const char *src[3];
	char *dst[3];
	long spitch[3], dpitch[3];
	int wbytes[3], h[3];
	int first_block[4];
	int n;
	int policy = 3; // 3: non-temporal loads and stores (data moved once); 0: temporal both ways (the copy is read again at once)
	int block0 = 0; // a launch's workgroup b copies block block0 + b
};
hipError_t launch_copy_rects(CopyRects r, hipStream_t s);
// the same copy cut into blocks of 8 rows x 4 KiB: fills first_block / n, returns the number of blocks (< 0: bad
// arguments); launch_copy_rects_range runs blocks [lo, hi) as a launch of their own
int copy_rects_plan(CopyRects *r);
hipError_t launch_copy_rects_range(CopyRects r, int lo, int hi, hipStream_t s);

// the strided gather / scatter (dwt_util_memcpy_stride_s / _i, src/system.c:102-164) on the device: w x h elements of
// `es` bytes between a dense image (row pitch `pitch`) and one whose element (y, x) lies at y*sx + x*sy; all in BYTES
hipError_t launch_strided_pack(void *dense, long pitch, const void *strided, long sx, long sy, int es, int w, int h, hipStream_t st);
hipError_t launch_strided_unpack(void *strided, long sx, long sy, const void *dense, long pitch, int es, int w, int h, hipStream_t st);

// device-side view helpers: pitch in BYTES, 4-byte elements
hipError_t launch_conv_show(bool is_int, const void *src, void *dst, long pitch, int w, int h, hipStream_t s);
hipError_t launch_compare(bool is_int, const void *p1, const void *p2, long pitch, int w, int h, unsigned *result, hipStream_t s);

} // namespace dwt
