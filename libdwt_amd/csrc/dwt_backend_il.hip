// dwt_backend_il.hip -- the interleaved (in-place lifting) layout: level chain on dense images,
// compose / decompose of the lattices, the exact phase-ordered path, and the C-ABI entry.
#include "dwt_backend.h"

using namespace dwtb;

#pragma GCC visibility push(default)
extern "C" {


// ---------------------------------------------------------------------------------
// Interleaved (in-place lifting) layout: libdwt.h dwt_cdf97_2f_inplace_s (src/libdwt.c:12926),
// dwt_cdf97_2i_inplace_s (:17474), dwt_cdf53_2f_inplace_s (:16553), dwt_cdf53_2i_inplace_s
// (:17886) and dwt-simple.h fdwt2_cdf97_* / fdwt2_cdf53_* (src/dwt-simple.c:2224, :2356).
// Level j transforms the stride-2^j lattice of the image in place.  On the device every
// level runs on a DENSE image instead: the forward sweep of level j writes its low-pass
// samples a second time, densely, as the input of level j+1, and the results of the levels
// >= 1 are scattered into the lattice afterwards (deepest last); the inverse gathers the
// lattices first.  Rows are finished before columns at every level; the 9/7 entries of the
// reference interleave the two in phases (prolog / core / epilog), which changes fp32
// rounding in the 8-sample border bands only (tests/test_oracle_interleaved.py).
// ---------------------------------------------------------------------------------
struct IlLevel {
	float *a = nullptr, *b = nullptr; // dense input / output of the level (levels >= 1)
	long pitch = 0;                   // elements
	int lx = 0, ly = 0;
};

// The reference's 9/7 in-place drivers and fdwt2_* cut every line transform into phases --
// SHORT (whole line, lines shorter than `min_phased`), PROLOG, CORE, EPILOG -- and run each
// phase over all rows, then all columns, before the next (src/dwt-simple.c:2266-2350,
// src/libdwt.c:12970-13480, 17517-17594).  This is that order, phase by phase, for
// dwt_util_set_accel(1): bit-identical to the reference, eight passes per level instead of one.
// Index ranges per lifting step: prolog src/dwt-simple.c:580-611, core :981-1029, epilog
// :1469-1528, short :424-510; inverse src/libdwt.c:9591-9668, 7661-7740, 9929-10010.
static void il_phase_ranges(int N, int K, bool inverse, int phase, IlPhase *ph)
{
	// phase: 0 short, 1 prolog, 2 core, 3 epilog
	for (int s = 0; s < 4; s++) {
		ph->lo[s] = 1;
		ph->hi[s] = 0;
	}
	if (phase == 0) {
		for (int s = 0; s < K; s++) {
			ph->lo[s] = 0;
			ph->hi[s] = N - 1;
		}
		ph->sc_lo = 0;
		ph->sc_hi = N - 1;
	} else if (!inverse) {
		const int M = (((N - 1) & ~1) - K) / 2; // core pairs, counted from index 1
		for (int s = 0; s < K; s++) {
			if (phase == 1) { ph->lo[s] = 0; ph->hi[s] = K - 1 - s; }
			else if (phase == 2) { ph->lo[s] = K + 1 - s; ph->hi[s] = K - 1 - s + 2 * M; }
			else { ph->lo[s] = K + 1 - s + 2 * M; ph->hi[s] = N - 1; }
		}
		if (phase == 1) { ph->sc_lo = 0; ph->sc_hi = 0; }
		else if (phase == 2) { ph->sc_lo = 1; ph->sc_hi = 2 * M; }
		else { ph->sc_lo = 2 * M + 1; ph->sc_hi = N - 1; }
	} else {
		const int M = ((N & ~1) - K) / 2; // core pairs, counted from index 0
		for (int s = 0; s < K; s++) {
			if (phase == 1) { ph->lo[s] = 0; ph->hi[s] = K - 2 - s; }
			else if (phase == 2) { ph->lo[s] = K - s; ph->hi[s] = K - 2 - s + 2 * M; }
			else { ph->lo[s] = K - s + 2 * M; ph->hi[s] = N - 1; }
		}
		if (phase == 1) { ph->sc_lo = 0; ph->sc_hi = K - 1; }
		else if (phase == 2) { ph->sc_lo = K; ph->sc_hi = K - 1 + 2 * M; }
		else { ph->sc_lo = K + 2 * M; ph->sc_hi = N - 1; }
	}
}

static bool il_is_phased(Wavelet w) { return w == kCdf97S || w == kCdf97SFma || w == kCdf53SNew; }

static int il_level_phased(Wavelet w, bool inverse, Img in, Img out, int lx, int ly)
{
	if (in.sx != out.sx)
		return fail("interleaved phased level: pitches differ");
	const int K = w == kCdf53SNew ? 2 : 4;
	const int min_phased = w == kCdf53SNew ? 3 : (inverse ? 4 : 5);
	struct Pass { bool rows; int phase; };
	Pass seq[8];
	int n = 0;
	for (int phase = 0; phase < 4; phase++) {
		if (lx > 1 && (phase == 0) == (lx < min_phased))
			seq[n++] = {true, phase};
		if (ly > 1 && (phase == 0) == (ly < min_phased))
			seq[n++] = {false, phase};
	}
	if (n == 0)
		return copy_rect(out, 0, 0, in, 0, 0, lx, ly);
	if (grow(&g.host_b, &g.host_b_bytes, (size_t)in.sx * ly))
		return 1;
	Img tmp{(char *)g.host_b, in.sx, 4};
	// ping-pong so that the last pass writes `out`
	Img cur = in;
	for (int i = 0; i < n; i++) {
		const Img nxt = ((n - 1 - i) % 2 == 0) ? out : tmp;
		IlPhase ph;
		const int N = seq[i].rows ? lx : ly;
		il_phase_ranges(N, K, inverse, seq[i].phase, &ph);
		hipError_t e = launch_il_phase(w == kCdf97SFma ? kCdf97S : w, inverse, cur.p, nxt.p, seq[i].rows ? cur.sx : 4, seq[i].rows ? 4 : cur.sx,
			seq[i].rows ? ly : lx, N, !seq[i].rows, ph, g.stream);
		if (e != hipSuccess)
			return fail("interleaved phase launch failed: %s", hipGetErrorString(e));
		cur = nxt;
	}
	return 0;
}

// Exact border strips for the fused sweep.  The fused level finishes the rows before the
// columns; the reference's phase order (rows' prolog, columns' prolog, rows' core, columns'
// core, rows' epilog, columns' epilog) rounds differently only where a column phase runs
// BEFORE a row phase that touches the same coefficients: the rows the columns' prolog reaches
// (0 .. 7, forward and inverse) and the columns the rows' epilog updates (the last 5).  Both
// strips are computed from the level's input in the reference's order by extra workgroups of the
// sweep's own launch (dwt_il_strip.h), whose tiles leave those samples alone: every level leaves
// its launch exact, the dense low-pass copy the next level reads included.
//   in_step / ll2 / out_step: see il_level
static IlStripArgs il_strip_args(Wavelet w, bool inverse, Img in, int in_step, const float *ll2, long ll2_pitch, Img out, int out_step,
	int lx, int ly, float *ll, long ll_pitch, const IlShell *sh)
{
	const int K = w == kCdf53SNew ? 2 : 4;
	IlStripArgs a;
	a.in = (const float *)in.p;
	a.in_pitch = in.sx / 4;
	a.in_step = in_step;
	a.ll_in = ll2;
	a.ll_in_pitch = ll2_pitch;
	a.top_in = sh ? sh->top : nullptr;
	a.top_in_pitch = sh ? sh->top_pitch : 0;
	a.right_in = sh ? sh->right : nullptr;
	a.right_in_pitch = sh ? sh->right_pitch : 0;
	a.right_x0 = sh ? sh->right_x0 : 0;
	a.out = (float *)out.p;
	a.out_pitch = out.sx / 4;
	a.out_step = out_step;
	a.ll = inverse ? nullptr : ll;
	a.ll_pitch = ll_pitch;
	a.lx = lx;
	a.ly = ly;
	for (int phase = 1; phase <= 3; phase++) {
		il_phase_ranges(lx, K, inverse, phase, &a.rph[phase - 1]);
		il_phase_ranges(ly, K, inverse, phase, &a.cph[phase - 1]);
	}
	return a;
}

// one level on dense images with a common pitch: rows completely, then columns
// whether a level takes the fused sweep.  Phase-ordered wavelets: the sweep plus exact border strips needs room for
// the strips; smaller levels take the exact phase passes for the whole level
static bool il_fusable(Wavelet w, bool scale_single, int lx, int ly, int dirs)
{
	const bool phased_small = il_is_phased(w) && w != kCdf97SFma && !scale_single && (lx < 64 || ly < 64);
	return dirs == 3 && !g.force_generic && !phased_small && lx >= 2 && ly >= 2;
}

// Fused sweep only: a level may be read (inverse) or written (forward) where it lives -- `in` / `out` = the rows of its
// lattice in a larger image (sx = that lattice's row pitch), `lat_step` elements between neighbouring samples of a row.
// Inverse: the samples at (even row, even column) come from the dense low-pass band `ll2` (the level below's result)
// instead; forward: with `ll` (a deeper level follows) those lattice points are left for the deeper levels to fill.
// in == out (fused sweep, lat_step 1, shapes il_shell_bytes accepts): the level runs IN PLACE over a snapshot of what its
// tiles read of their neighbours (IlShell), built here in the staging image's memory.
static int il_level(Wavelet w, bool inverse, bool scale_single, Img in, Img out, int lx, int ly, float *ll, long ll_pitch,
	int dirs = 3, int lat_step = 1, const float *ll2 = nullptr, long ll2_pitch = 0)
{
	const int in_step = inverse ? lat_step : 1, out_step = inverse ? 1 : lat_step;
	const bool in_place = in.p == out.p;
	// dirs: bit 0 rows, bit 1 columns (fdwt2h1_* / fdwt2v1_* lift one direction only: line passes)
	const bool phased_small = il_is_phased(w) && w != kCdf97SFma && !scale_single && (lx < 64 || ly < 64);
	const bool fused = il_fusable(w, scale_single, lx, ly, dirs) && (((uintptr_t)in.p | (uintptr_t)out.p) % 4 == 0);
	if ((lat_step != 1 || ll2) && !fused)
		return fail("internal: only the fused sweeps work on a lattice");
	// (option il_exact_borders = 0: no strips -- the borders keep the sweep's rows-then-columns rounding, a few ulp off
	// the reference's phase order there, far inside the 1e-5 relative tolerance; like "fma" an opt-in, never the default)
	const bool strips = fused && il_is_phased(w) && w != kCdf97SFma && !scale_single && g.il_exact_borders;
	if (in_place && !(fused && lat_step == 1))
		return fail("internal: only the fused sweeps run a level in place");
	if (fused) {
		hipError_t e;
		IlShell sh;
		SweepTuning tune = g.tune;
		if (in_place) {
			tune.tile_pairs = il_sweep_tile_pairs(g.tune, lx, ly, inverse);
			const size_t need = il_shell_bytes(lx, ly, tune.tile_pairs);
			if (!need)
				return fail("internal: this level cannot run in place");
			if (grow(&g.stage_img, &g.stage_bytes, need))
				return 1;
			e = launch_il_shell((const float *)in.p, in.sx / 4, lx, ly, tune.tile_pairs, (float *)g.stage_img, &sh, g.stream);
			if (e != hipSuccess)
				return fail("interleaved in-place snapshot failed: %s", hipGetErrorString(e));
		}
		IlStripArgs sa;
		if (strips)
			sa = il_strip_args(w, inverse, in, in_step, ll2, ll2_pitch, out, out_step, lx, ly, ll, ll_pitch, in_place ? &sh : nullptr);
		if (!inverse) {
			FwdLevelArgs a;
			a.in = in.p; a.in_pitch = in.sx / 4; a.in_bstride = 0;
			a.out_ll = ll; a.ll_pitch = ll_pitch; a.ll_bstride = 0;
			a.out_h = out.p; a.h_pitch = out.sx / 4; a.h_bstride = 0;
			a.W = lx; a.H = ly; a.batch = 1; a.interleaved = 1; a.il_ll = ll != nullptr ? (g.il_temporal ? 2 : 1) : 0;
			a.out_step = out_step;
			if (in_place)
				a.sh = sh;
			e = launch_fwd_level(w, a, tune, g.stream, strips ? &sa : nullptr);
		} else {
			InvLevelArgs a;
			a.in_ll = in.p; a.ll_pitch = in.sx / 4 * 2; a.ll_bstride = 0;
			a.in_h = in.p + in.sx; a.h_pitch = in.sx / 4 * 2; a.h_bstride = 0;
			a.out = out.p; a.out_pitch = out.sx / 4; a.out_bstride = 0;
			a.W = lx; a.H = ly; a.batch = 1; a.interleaved = 1;
			a.in_step = in_step; a.in_ll2 = ll2; a.ll2_pitch = ll2_pitch;
			if (in_place)
				a.sh = sh;
			e = launch_inv_level(w == kCdf53SNew ? kCdf53S : w, a, tune, g.stream, strips ? &sa : nullptr);
		}
		if (e != hipSuccess)
			return fail("interleaved sweep launch failed: %s", hipGetErrorString(e));
		return 0;
	}
	// generic.  The phase-ordered entries reproduce the reference's order exactly when the
	// generic path was asked for (accel 1); tiny levels of the fused path and the 5/3 _inplace_
	// pair (rows, then columns in the reference too) take two exact line passes.
	if (dirs == 3 && (g.force_generic || phased_small) && il_is_phased(w) && !scale_single) {
		if (il_level_phased(w, inverse, in, out, lx, ly))
			return 1;
		if (ll && !inverse) {
			hipError_t e = launch_lattice_copy((const float *)out.p, 2, out.sx / 4 * 2, 0, ll, 1, ll_pitch, 0, (lx + 1) / 2, (ly + 1) / 2, 1, g.stream);
			if (e != hipSuccess)
				return fail("lattice gather failed: %s", hipGetErrorString(e));
		}
		return 0;
	}
	if (in.sx != out.sx)
		return fail("interleaved generic level: pitches differ");
	if (grow(&g.host_b, &g.host_b_bytes, (size_t)in.sx * ly))
		return 1;
	Img tmp{(char *)g.host_b, in.sx, 4};
	auto pass = [&](bool rows, Img from, Img to) -> int {
		const int N = rows ? lx : ly, lines = rows ? ly : lx;
		if ((N == 1 && !scale_single) || !(dirs & (rows ? 1 : 2)))
			return copy_rect(to, 0, 0, from, 0, 0, lx, ly);
		hipError_t e = launch_line_pass(w, inverse, from.p, to.p, rows ? from.sx : 4, rows ? 4 : from.sx, lines, N, -1, !rows, g.stream);
		if (e != hipSuccess)
			return fail("interleaved line pass launch failed: %s", hipGetErrorString(e));
		return 0;
	};
	if (pass(true, in, tmp) || pass(false, tmp, out))
		return 1;
	if (ll && !inverse) {
		hipError_t e = launch_lattice_copy((const float *)out.p, 2, out.sx / 4 * 2, 0, ll, 1, ll_pitch, 0, (lx + 1) / 2, (ly + 1) / 2, 1, g.stream);
		if (e != hipSuccess)
			return fail("lattice gather failed: %s", hipGetErrorString(e));
	}
	return 0;
}

static int interleaved2d(Wavelet w, bool inverse, bool scale_single, Img src, Img dst, int sox, int soy, int six, int siy,
	int *jp, int decompose_one, int dirs = 3)
{
	const int j_limit = ceil_log2(decompose_one ? (sox > soy ? sox : soy) : (sox < soy ? sox : soy));
	int J = *jp;
	if (J < 0 || J > j_limit)
		J = j_limit;
	if (!inverse)
		*jp = J;
	const bool alias = src.p == dst.p;
	// in place the forward result is built in the staging image and copied back at once: temporal stores leave the even
	// rows in the Infinity Cache for that copy (8192^2 J=5, one process, alternated: 342 -> 335 us); out of place nothing
	// reads them again: non-temporal (245-247 -> 243 us)
	g.il_temporal = alias;
	// everything outside the transformed region keeps the caller's values
	const bool sparse = six < sox || siy < soy;
	if (!alias && (J == 0 || sparse) && copy_rect(dst, 0, 0, src, 0, 0, sox, soy))
		return 1;
	if (J == 0 || six < 1 || siy < 1)
		return 0;
	constexpr int kMax = 32;
	IlLevel L[kMax];
	size_t pool = 0;
	for (int j = 0; j < J; j++) {
		L[j].lx = ceil_div_pow2(six, j);
		L[j].ly = ceil_div_pow2(siy, j);
		L[j].pitch = align_up(L[j].lx, 4);
		if (j >= 1)
			pool += (size_t)L[j].pitch * L[j].ly;
	}
	if (J > 1) {
		if (g.ll_external)
			return fail("the interleaved entries keep their level pyramid in the library's own scratch: hand it back first (dwt_hip_set_workspace(NULL, 0, NULL, 0))");
		if (grow(&g.ll[0], &g.ll_bytes[0], pool * 4) || grow(&g.ll[1], &g.ll_bytes[1], pool * 4))
			return 1;
		float *pa = (float *)g.ll[0], *pb = (float *)g.ll[1];
		for (int j = 1; j < J; j++) {
			L[j].a = pa; L[j].b = pb;
			pa += (size_t)L[j].pitch * L[j].ly;
			pb += (size_t)L[j].pitch * L[j].ly;
		}
	}
	auto dense = [&](float *p, const IlLevel &l) { return Img{(char *)p, l.pitch * 4, 4}; };
	auto scatter = [&](const float *from, long from_pitch, char *to, long to_pitch_bytes, long step, const IlLevel &l) -> int {
		// dense level -> lattice of stride `step` (elements) of an image
		hipError_t e = launch_lattice_copy(from, 1, from_pitch, 0, (float *)to, step, to_pitch_bytes / 4 * step, 0, l.lx, l.ly, 1, g.stream);
		if (e != hipSuccess)
			return fail("lattice scatter failed: %s", hipGetErrorString(e));
		return 0;
	};
	// Level 0 works on the caller's image.  In place (round 4) it runs over a snapshot of the tiles' foreign samples
	// (il_level, IlShell: 8192^2 14 % of the image instead of a detour of the whole image through a staging copy);
	// shapes the snapshot does not cover, and the generic path, detour through the staging image.
	const bool ptrs_ok = ((uintptr_t)src.p | (uintptr_t)dst.p) % 4 == 0;
	const bool ip0 = alias && g.il_inplace_shell && ptrs_ok && il_fusable(w, scale_single, L[0].lx, L[0].ly, dirs) &&
		il_shell_bytes(L[0].lx, L[0].ly, il_sweep_tile_pairs(g.tune, L[0].lx, L[0].ly, inverse)) != 0;
	Img stage{nullptr, dst.sx, 4};
	if ((alias && !ip0) || inverse) {
		if (grow(&g.stage_img, &g.stage_bytes, (size_t)dst.sx * siy))
			return 1;
		stage.p = (char *)g.stage_img;
	}

	if (!inverse) {
		// Every level runs on a dense image (the level above hands its low-pass samples over densely) and writes its
		// result where it lives: a fused level straight to its lattice in the destination (round 4: no compose pass),
		// other levels through a dense image and a scatter.  Where a deeper level follows, the lattice points at (even
		// row, even column) are rewritten by it later in the stream.  In place without the snapshot the whole result is
		// built in the staging image -- the sweep of level 0 must not write what other tiles still read -- and copied back.
		const bool detour = alias && !ip0;
		const Img res = detour ? stage : dst;
		const bool aligned = ((uintptr_t)src.p | (uintptr_t)res.p) % 4 == 0;
		for (int j = 0; j < J; j++) {
			const Img in = j == 0 ? src : dense(L[j].a, L[j]);
			float *ll = j + 1 < J ? L[j + 1].a : nullptr;
			const long ll_pitch = j + 1 < J ? L[j + 1].pitch : 0;
			if (j == 0 || (aligned && il_fusable(w, scale_single, L[j].lx, L[j].ly, dirs))) {
				if (il_level(w, false, scale_single, in, Img{res.p, res.sx << j, 4}, L[j].lx, L[j].ly, ll, ll_pitch, dirs, 1 << j))
					return 1;
				continue;
			}
			if (il_level(w, false, scale_single, in, dense(L[j].b, L[j]), L[j].lx, L[j].ly, ll, ll_pitch, dirs) ||
				scatter(L[j].b, L[j].pitch, res.p, res.sx, 1L << j, L[j]))
				return 1;
		}
		return detour ? copy_rect(dst, 0, 0, stage, 0, 0, six, siy) : 0;
	}
	// inverse: the coefficients are read from the source image, never modified before the last sweep has read it.
	// A level that takes the fused sweep reads its lattice where it lives in the image, the samples at (even row, even
	// column) from the dense result of the level below: no gather of the lattices, no copies between the levels, no
	// compose pass (round 4; 8192^2 J=5: 309 -> 2xx us).  Other levels (tiny ones of the phase-ordered wavelets, the
	// generic path) get their input gathered into a dense image first.
	const Img cin = src;
	if (J == 1) {
		if (!alias || ip0)
			return il_level(w, true, scale_single, cin, dst, L[0].lx, L[0].ly, nullptr, 0);
		if (il_level(w, true, scale_single, dst, stage, L[0].lx, L[0].ly, nullptr, 0))
			return 1;
		return copy_rect(dst, 0, 0, stage, 0, 0, six, siy);
	}
	const bool aligned = ((uintptr_t)cin.p | (uintptr_t)dst.p) % 4 == 0;
	hipError_t e;
	for (int j = J - 1; j >= 1; j--) {
		const float *ll2 = j + 1 < J ? L[j + 1].b : nullptr;
		const long ll2_pitch = j + 1 < J ? L[j + 1].pitch : 0;
		if (aligned && il_fusable(w, scale_single, L[j].lx, L[j].ly, 3)) {
			if (il_level(w, true, scale_single, Img{cin.p, cin.sx << j, 4}, dense(L[j].b, L[j]), L[j].lx, L[j].ly, nullptr, 0, 3, 1 << j, ll2, ll2_pitch))
				return 1;
			continue;
		}
		e = launch_lattice_copy((const float *)cin.p, 1L << j, (cin.sx / 4) << j, 0, L[j].a, 1, L[j].pitch, 0, L[j].lx, L[j].ly, 1, g.stream);
		if (e != hipSuccess)
			return fail("lattice gather failed: %s", hipGetErrorString(e));
		// the reconstructed low-pass band of the level below is the even-even lattice of this one
		if (ll2 && scatter(ll2, ll2_pitch, (char *)L[j].a, L[j].pitch * 4, 2, L[j + 1]))
			return 1;
		if (il_level(w, true, scale_single, dense(L[j].a, L[j]), dense(L[j].b, L[j]), L[j].lx, L[j].ly, nullptr, 0))
			return 1;
	}
	// level 0.  In place the sweep must not write what other tiles still read: it runs over the snapshot, or writes
	// the staging image
	if (aligned && il_fusable(w, scale_single, L[0].lx, L[0].ly, 3)) {
		const bool detour = alias && !ip0;
		if (il_level(w, true, scale_single, cin, detour ? stage : dst, L[0].lx, L[0].ly, nullptr, 0, 3, 1, L[1].b, L[1].pitch))
			return 1;
		return detour ? copy_rect(dst, 0, 0, stage, 0, 0, six, siy) : 0;
	}
	// generic path: the whole input is built in the staging image (the coefficients, the level below's result on their
	// even-even lattice)
	if (copy_rect(stage, 0, 0, cin, 0, 0, six, siy) || scatter(L[1].b, L[1].pitch, stage.p, stage.sx, 2, L[1]))
		return 1;
	return il_level(w, true, scale_single, stage, dst, L[0].lx, L[0].ly, nullptr, 0);
}

// dwt_cdf97_2f_inplace_i / dwt_cdf97_2i_inplace_i (src/libdwt.c:17424, :17308): fixed-point int
// 9/7, interleaved, exactly as the reference runs them -- the strides are NOT scaled per level,
// so level j re-transforms the dense top-left ceil(size/2^j) block of the interleaved image (the
// reference marks the pair "tested only with j=1", :17423).  Exact line passes; forward rows then
// columns, inverse columns then rows.
static int inplace_int2d(bool inverse, Img src, Img dst, int sox, int soy, int six, int siy, int *jp, int decompose_one)
{
	const int j_limit = ceil_log2(decompose_one ? (sox > soy ? sox : soy) : (sox < soy ? sox : soy));
	int J = *jp;
	if (J < 0 || J > j_limit)
		J = j_limit;
	if (!inverse)
		*jp = J;
	if (src.p != dst.p && copy_rect(dst, 0, 0, src, 0, 0, sox, soy))
		return 1;
	for (int step = 0; step < J; step++) {
		const int j = inverse ? J - 1 - step : step;
		const int nx = ceil_div_pow2(six, j), ny = ceil_div_pow2(siy, j);
		for (int pass = 0; pass < 2; pass++) {
			const bool rows = inverse ? pass == 1 : pass == 0;
			const int N = rows ? nx : ny, lines = rows ? ny : nx;
			if (N < 2 || lines < 1)
				continue; // the line kernels leave shorter lines alone (:17365, :17246)
			if (generic_pass(kCdf97IIp, inverse, rows, dst, dst, nx, ny, lines, N, -1))
				return 1;
		}
	}
	return 0;
}

int dwt_hip_transform2d_interleaved(int wavelet, int inverse, int flavour, const void *src, void *dst, int stride_x, int stride_y,
	int sox, int soy, int six, int siy, int *j, int decompose_one)
{
	if (check_inited())
		return 1;
	const bool fixed = wavelet == kCdf97I && flavour == 0;
	if (wavelet != kCdf97S && wavelet != kCdf53S && !fixed)
		return fail("the interleaved layout takes float CDF 9/7, float CDF 5/3 and (libdwt.h entries) fixed-point int CDF 9/7, not %d", wavelet);
	if (flavour < 0 || flavour > 3)
		return fail("unknown flavour %d", flavour);
	if (flavour >= 1 && inverse)
		return fail("dwt-simple.h has forward transforms only; use flavour 0 for the inverse");
	if (flavour >= 2 && wavelet != kCdf97S)
		return fail("the one-direction entries exist for float CDF 9/7 only");
	// flavours 2 / 3: fdwt2h1_cdf97_vertical_s / fdwt2v1_cdf97_vertical_s (src/dwt-simple.c:1747, 1837) lift
	// the rows / the columns of every level only
	const int dirs = flavour == 2 ? 1 : flavour == 3 ? 2 : 3;
	if (!src || !dst || !j)
		return fail("null pointer argument");
	if (sox <= 0 || soy <= 0 || six < 0 || siy < 0 || six > sox || siy > soy)
		return fail("bad sizes: outer %dx%d inner %dx%d", sox, soy, six, siy);
	g_elems_are_32bit = true;
	// single-sample lines: the 9/7 drivers and fdwt2_* leave them (guards `size > 1`,
	// libdwt.c:12978, dwt-simple.c:2266), the 5/3 _inplace_ drivers scale them (:11041, :11840)
	const bool scale_single = wavelet == kCdf53S && flavour == 0;
	const Wavelet w = wavelet == kCdf97S ? ((g.fma && !inverse && dirs == 3) ? kCdf97SFma : kCdf97S) : (flavour == 1 ? kCdf53SNew : kCdf53S);
	const bool dev_src = dwt_hip_is_device_pointer(src), dev_dst = dwt_hip_is_device_pointer(dst);
	if (dev_src != dev_dst)
		return fail("src and dst must both be host or both be device pointers");
	if (dev_dst && stride_y == 4 && stride_x % 4 == 0 && stride_x >= sox * 4 && (uintptr_t)src % 4 == 0 && (uintptr_t)dst % 4 == 0) {
		if (fixed)
			return inplace_int2d(inverse != 0, Img{(char *)src, stride_x, 4}, Img{(char *)dst, stride_x, 4}, sox, soy, six, siy, j, decompose_one);
		return interleaved2d(w, inverse != 0, scale_single, Img{(char *)src, stride_x, 4}, Img{(char *)dst, stride_x, 4}, sox, soy, six, siy, j, decompose_one, dirs);
	}
	if (dev_dst && (stride_y < 4 || (long)stride_x < (long)(sox - 1) * stride_y + 4))
		return fail("device image: stride_y %d must be >= 4 and stride_x %d >= (width-1)*stride_y + 4", stride_y, stride_x);
	// host pointers (any byte strides): the outer frame is staged through HBM; device images whose elements are not
	// adjacent or not aligned: packed, transformed and spread back on the device (dwt_strided.hip)
	const long pitch = align_up((long)sox * 4, 256);
	if (grow(&g.host_a, &g.host_a_bytes, (size_t)pitch * soy))
		return 1;
	if (dev_dst) {
		if (hipError_t e = launch_strided_pack(g.host_a, pitch, src, stride_x, stride_y, 4, sox, soy, g.stream))
			return fail("strided pack launch failed: %s", hipGetErrorString(e));
	} else if (host_upload(src, stride_x, stride_y, 4, sox, soy, g.host_a, pitch))
		return 1;
	Img A{(char *)g.host_a, pitch, 4};
	if (fixed ? inplace_int2d(inverse != 0, A, A, sox, soy, six, siy, j, decompose_one)
	          : interleaved2d(w, inverse != 0, scale_single, A, A, sox, soy, six, siy, j, decompose_one, dirs))
		return 1;
	if (dev_dst) {
		if (hipError_t e = launch_strided_unpack(dst, stride_x, stride_y, g.host_a, pitch, 4, sox, soy, g.stream))
			return fail("strided unpack launch failed: %s", hipGetErrorString(e));
		return 0;
	}
	return host_download(dst, stride_x, stride_y, 4, sox, soy, g.host_a, pitch);
}

} // extern "C"
#pragma GCC visibility pop
