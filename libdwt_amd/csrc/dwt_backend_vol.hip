// dwt_backend_vol.hip -- the 3-D entries: out of place (fused one-pass levels) and in place
// (two passes per level), multi-level over dense per-level volumes.
#include "dwt_backend.h"

using namespace dwtb;

constexpr int kMaxVolLevels = 24;

// The levels of a forward out-of-place call: src (strides ssy / ssz, in samples) -> dst (vsy / vsz).
// Arguments are validated by the callers.
static int vol_forward_op(const float *src, long ssy, long ssz, float *dst, long vsy, long vsz, int nx, int ny, int nz, int levels)
{
	if (levels < 1) {
		// no levels: the reference's copy stage alone
		hipError_t e = launch_lattice_copy(src, 1, ssy, ssz, dst, 1, vsy, vsz, nx, ny, nz, g.stream);
		return e == hipSuccess ? 0 : fail("volume copy failed: %s", hipGetErrorString(e));
	}
	struct Lvl { const float *in; float *out; long sy, sz; int lx, ly, lz; } L[kMaxVolLevels]; // sy, sz: of `out` (and of `in` for levels >= 1)
	L[0] = {src, dst, vsy, vsz, nx, ny, nz};
	size_t pool = 0;
	for (int j = 1; j < levels; j++) {
		L[j].lx = ceil_div_pow2(nx, j); L[j].ly = ceil_div_pow2(ny, j); L[j].lz = ceil_div_pow2(nz, j);
		L[j].sy = align_up(L[j].lx, 4);
		L[j].sz = L[j].sy * L[j].ly;
		pool += (size_t)L[j].sz * L[j].lz;
	}
	if (levels > 1) {
		if (grow(&g.host_a, &g.host_a_bytes, pool * 4) || grow(&g.host_b, &g.host_b_bytes, pool * 4))
			return 1;
		float *pa = (float *)g.host_a, *pb = (float *)g.host_b;
		for (int j = 1; j < levels; j++) {
			L[j].in = pa; L[j].out = pb;
			pa += (size_t)L[j].sz * L[j].lz;
			pb += (size_t)L[j].sz * L[j].lz;
		}
	}
	for (int j = 0; j < levels; j++)
		if (L[j].lz > 65535 || L[j].ly > 65535)
			return fail("volume too large for the launch grid");
	float *S = nullptr;
	long s_sy = 0, s_sz = 0;
	bool in_place[kMaxVolLevels] = {}; // level wrote its lattice of dst itself
	// in place (src == dst, the caller has checked that the one-pass in-place level applies): level 0
	// runs over a snapshot of its tile halos (dwt_vol3d_ip.hip); the deeper levels read dense scratch
	// volumes and write into their lattices of the volume as in an out-of-place call
	const bool in_place_call = src == dst;
	auto fuses = [&](int j) {
		VolFusedArgs t{L[j].in, j ? L[j].sy : ssy, j ? L[j].sz : ssz, L[j].out, L[j].sy, L[j].sz, nullptr, 0, 0, L[j].lx, L[j].ly, L[j].lz};
		if (j == 0 && in_place_call)
			return true;
		const bool can = t.in != t.out && t.nx >= 2 && t.ny >= 2 && t.nz >= 2;
		return !g.force_generic && ((g.vol.fused == 1 && vol_fused_applies(t)) || (g.vol.fused >= 2 && can));
	};
	// levels 0 and 1 as a pair (an even x size: level 1's merged rows are exactly twice its own)
	const bool merged = levels >= 2 && g.vol.direct >= 2 && g.vol.whole && g.vol.rows != 6 && fuses(0) && fuses(1) && nx % 2 == 0;
	for (int j = 0; j < levels; j++) {
		const Lvl &b = L[j];
		float *lll = j + 1 < levels ? (float *)L[j + 1].in : nullptr;
		const long lsy = j + 1 < levels ? L[j + 1].sy : 0, lsz = j + 1 < levels ? L[j + 1].sz : 0;
		VolFusedArgs fa{b.in, j ? b.sy : ssy, j ? b.sz : ssz, b.out, b.sy, b.sz, lll, lsy, lsz, b.lx, b.ly, b.lz};
		if (fuses(j)) {
			if (j <= 1 && merged) {
				// the rows with even y and even z are written once, by level 1 (level 0 parks their
				// odd-x samples in level 1's otherwise unused dense result)
				fa.mode = j == 0 ? 2 : 3;
				fa.side = L[1].out; fa.side_sy = L[1].sy; fa.side_sz = L[1].sz;
				if (j == 1) {
					fa.out = dst;
					fa.out_sy = vsy * 2; fa.out_sz = vsz * 2;
					fa.temporal_shared = levels > 2; // level 2's merge pass reads those rows at once (1024^3, 3 levels: 2.040 -> 2.024 ms)
					in_place[1] = true;
				}
			} else if (j >= 1 && g.vol.direct && g.vol.rows != 6 && (j == 1 || in_place[j - 1])) {
				// straight into the level's lattice of the destination: no dense result, no scatter pass
				fa.mode = 1;
				fa.out = dst;
				fa.out_sx = 1L << j;
				fa.out_sy = vsy << j;
				fa.out_sz = vsz << j;
				in_place[j] = true;
			}
			prof_before(j);
			hipError_t e;
			if (j == 0 && in_place_call) {
				if (grow(&g.vol_out, &g.vol_out_bytes, vol_level_ip_scratch(fa, g.vol)))
					return 1;
				e = launch_vol_level_ip(false, fa, (float *)g.vol_out, g.vol, g.stream);
			} else if ((fa.mode == 0 || fa.mode == 2) && g.vol.whole && g.vol.rows != 6 && vol_level_ip_can(fa)) {
				// dense / withholding levels: the 64-row tiles of dwt_vol3d_ip.hip's kernel, out of place
				e = launch_vol_level_op(false, fa, g.vol, g.stream);
			} else {
				e = launch_vol_fwd_fused(fa, g.vol, g.stream);
			}
			prof_after(j);
			if (e != hipSuccess)
				return fail("fused 3-D level launch failed: %s", hipGetErrorString(e));
			continue;
		}
		// two passes through the scratch volume
		if (!S) {
			// sized for the first level that needs it (the deeper ones are smaller)
			s_sy = align_up(b.lx, 4);
			s_sz = s_sy * b.ly;
			if (grow(&g.stage_img, &g.stage_bytes, (size_t)s_sz * b.lz * 4))
				return 1;
			S = (float *)g.stage_img;
		}
		FwdLevelArgs a;
		a.in = b.in; a.in_pitch = j ? b.sy : ssy; a.in_bstride = j ? b.sz : ssz;
		a.out_ll = S; a.ll_pitch = s_sy; a.ll_bstride = s_sz;
		a.out_h = S; a.h_pitch = s_sy; a.h_bstride = s_sz;
		a.W = b.lx; a.H = b.ly; a.batch = b.lz; a.interleaved = 1; a.plain_ends = 1;
		hipError_t e = launch_fwd_level(kCdf97S, a, g.tune, g.stream);
		if (e != hipSuccess)
			return fail("3-D xy pass launch failed: %s", hipGetErrorString(e));
		e = launch_vol_z(false, S, s_sy, s_sz, b.out, b.sy, b.sz, b.lx, b.ly, b.lz, g.vol, g.stream, lll, lsy, lsz);
		if (e != hipSuccess)
			return fail("3-D z pass launch failed: %s", hipGetErrorString(e));
	}
	// the other levels >= 1 hold dense results: into their lattices of dst, shallow first (a level's
	// even-even-even samples are the next level's, which overwrites them)
	for (int j = 1; j < levels; j++) {
		if (in_place[j])
			continue;
		const Lvl &c = L[j];
		hipError_t e = launch_lattice_scatter(c.out, c.sy, c.sz, dst, 1L << j, vsy << j, vsz << j, c.lx, c.ly, c.lz, nx, g.stream);
		if (e != hipSuccess)
			return fail("lattice scatter failed: %s", hipGetErrorString(e));
	}
	return 0;
}

#pragma GCC visibility push(default)
extern "C" {


// Forward 3-D transform, OUT OF PLACE: the layout and arithmetic of cdf97_3f_op_sep_horizontal_s
// (src/volume-dwt.c:727-785: copy each x line to the destination, then lift x, y, z there), the
// entry the reference's own 3-D perf test drives (volume_perftest_fwd97op_s, src/volume.c).
// Level j reads a dense volume and writes a dense volume, so each level is ONE fused pass
// (k_vol_fwd_fused) where that kernel applies and the two-pass path (xy sweep, z sweep through
// the scratch volume) elsewhere; the even-even-even samples go to the next level densely.  A fused
// level >= 1 writes its result straight into its lattice of the destination (stride 2^j in x, y
// and z); the two-pass levels keep a dense result that is scattered there at the end.
int dwt_hip_transform3d_op(const void *src, void *dst, size_t stride_y, size_t stride_z, int nx, int ny, int nz, int levels)
{
	if (check_inited())
		return 1;
	if (!src || !dst || !dwt_hip_is_device_pointer(src) || !dwt_hip_is_device_pointer(dst))
		return fail("dwt_hip_transform3d_op takes device pointers");
	if (src == dst)
		return fail("dwt_hip_transform3d_op is out of place; use dwt_hip_transform3d for in-place volumes");
	if ((stride_y & 3) || (stride_z & 3) || stride_y < (size_t)nx * 4 || stride_z < stride_y * (size_t)ny)
		return fail("bad volume strides");
	if (levels > kMaxVolLevels)
		return fail("too many levels");
	if (levels >= 1 && (ceil_div_pow2(nx, levels - 1) < 2 || ceil_div_pow2(ny, levels - 1) < 2 || ceil_div_pow2(nz, levels - 1) < 2))
		return fail("volume %dx%dx%d is too small for %d levels", nx, ny, nz, levels);
	const long vsy = (long)stride_y / 4, vsz = (long)stride_z / 4;
	return vol_forward_op((const float *)src, vsy, vsz, (float *)dst, vsy, vsz, nx, ny, nz, levels);
}

int dwt_hip_transform3d(int inverse, void *vol, size_t stride_y, size_t stride_z, int nx, int ny, int nz, int levels)
{
	if (check_inited())
		return 1;
	if (!vol || !dwt_hip_is_device_pointer(vol))
		return fail("dwt_hip_transform3d takes a device pointer");
	if ((stride_y & 3) || (stride_z & 3) || stride_y < (size_t)nx * 4 || stride_z < stride_y * (size_t)ny)
		return fail("bad volume strides");
	if (levels < 1)
		return 0;
	// every level needs at least 2 samples per axis (the reference asserts >= 5, dwt-simple.c:2172)
	if (ceil_div_pow2(nx, levels - 1) < 2 || ceil_div_pow2(ny, levels - 1) < 2 || ceil_div_pow2(nz, levels - 1) < 2)
		return fail("volume %dx%dx%d is too small for %d levels", nx, ny, nz, levels);
	// One fused pass per level IN PLACE, over a snapshot of the tile halos (dwt_vol3d_ip.hip; round 3):
	// ~10.5 B per voxel instead of 16.  Forward: level 0 in place, the deeper levels as in an
	// out-of-place call (dense scratch inputs, results into their lattices of the volume).
	const long vsy0 = (long)stride_y / 4, vsz0 = (long)stride_z / 4;
	auto ip_level = [&](const float *p, long sy, long sz, int lx, int ly, int lz) {
		VolFusedArgs t{p, sy, sz, (float *)p, sy, sz, nullptr, 0, 0, lx, ly, lz};
		return g.vol.inplace_fused == 1 && !g.force_generic && g.vol.fused >= 1 &&
			(g.vol.fused >= 2 ? vol_level_ip_can(t) : vol_level_ip_applies(t));
	};
	if (!inverse && levels <= kMaxVolLevels && ip_level((const float *)vol, vsy0, vsz0, nx, ny, nz))
		return vol_forward_op((const float *)vol, vsy0, vsz0, (float *)vol, vsy0, vsz0, nx, ny, nz, levels);
	// scratch: S (pass-to-pass buffer) and, for levels >= 1, dense copies P[j] of the
	// level-j lattice (even-even-even samples of level j-1), all carved from one buffer
	const long s_sy = align_up(nx, 4), s_sz = s_sy * ny;
	float *S = nullptr; // sized below for the largest level that takes the two passes (none does when every level runs in one)
	constexpr int kMaxLevels = 24;
	if (levels > kMaxLevels)
		return fail("too many levels");
	struct Lvl { float *p; long sy, sz; int lx, ly, lz; } L[kMaxLevels];
	L[0] = {(float *)vol, (long)stride_y / 4, (long)stride_z / 4, nx, ny, nz};
	size_t p_total = 0;
	for (int j = 1; j < levels; j++) {
		L[j].lx = ceil_div_pow2(nx, j); L[j].ly = ceil_div_pow2(ny, j); L[j].lz = ceil_div_pow2(nz, j);
		L[j].sy = align_up(L[j].lx, 4);
		L[j].sz = L[j].sy * L[j].ly;
		p_total += (size_t)L[j].sz * L[j].lz;
	}
	if (levels > 1) {
		if (grow(&g.host_a, &g.host_a_bytes, p_total * 4))
			return 1;
		float *p = (float *)g.host_a;
		for (int j = 1; j < levels; j++) {
			L[j].p = p;
			p += (size_t)L[j].sz * L[j].lz;
		}
	}
	for (int j = 0; j < levels; j++)
		if (L[j].lz > 65535 || L[j].ly > 65535)
			return fail("volume too large for the launch grid");
	for (int j = 0; j < levels; j++)
		if (!inverse || !ip_level(L[j].p, L[j].sy, L[j].sz, L[j].lx, L[j].ly, L[j].lz)) {
			// S keeps level 0's strides for every level: slice z of level j at z * s_sz
			if (grow(&g.stage_img, &g.stage_bytes, (size_t)s_sz * L[j].lz * 4))
				return 1;
			S = (float *)g.stage_img;
			break;
		}

	if (inverse) {
		// the shell of the one-pass levels, sized ONCE for the largest of them (the loop below runs from the
		// smallest level up: growing it level by level meant a stream sync + free + malloc per level)
		size_t shell = 0;
		for (int j = 0; j < levels; j++)
			if (ip_level(L[j].p, L[j].sy, L[j].sz, L[j].lx, L[j].ly, L[j].lz)) {
				VolFusedArgs fa{L[j].p, L[j].sy, L[j].sz, L[j].p, L[j].sy, L[j].sz, nullptr, 0, 0, L[j].lx, L[j].ly, L[j].lz};
				shell = std::max(shell, vol_level_ip_scratch(fa, g.vol));
			}
		if (shell && grow(&g.vol_out, &g.vol_out_bytes, shell))
			return 1;
	}

	auto one_level = [&](const Lvl &b, const Lvl *next) -> int {
		// x then y fused per slice, then z (src/volume-dwt.c:677-725; inverse :1115-1163)
		hipError_t e;
		if (inverse && ip_level(b.p, b.sy, b.sz, b.lx, b.ly, b.lz)) {
			// one pass, in place (the volume itself for level 0, the dense copy of its lattice above)
			VolFusedArgs fa{b.p, b.sy, b.sz, b.p, b.sy, b.sz, nullptr, 0, 0, b.lx, b.ly, b.lz};
			if (grow(&g.vol_out, &g.vol_out_bytes, vol_level_ip_scratch(fa, g.vol)))
				return 1;
			e = launch_vol_level_ip(true, fa, (float *)g.vol_out, g.vol, g.stream);
			if (e != hipSuccess)
				return fail("in-place fused 3-D level launch failed: %s", hipGetErrorString(e));
			return 0;
		}
		if (!inverse) {
			FwdLevelArgs a;
			a.in = b.p; a.in_pitch = b.sy; a.in_bstride = b.sz;
			a.out_ll = S; a.ll_pitch = s_sy; a.ll_bstride = s_sz;
			a.out_h = S; a.h_pitch = s_sy; a.h_bstride = s_sz;
			a.W = b.lx; a.H = b.ly; a.batch = b.lz; a.interleaved = 1; a.plain_ends = 1;
			e = launch_fwd_level(kCdf97S, a, g.tune, g.stream);
		} else {
			InvLevelArgs a;
			a.in_ll = b.p; a.ll_pitch = 2 * b.sy; a.ll_bstride = b.sz;
			a.in_h = b.p + b.sy; a.h_pitch = 2 * b.sy; a.h_bstride = b.sz;
			a.out = S; a.out_pitch = s_sy; a.out_bstride = s_sz;
			a.W = b.lx; a.H = b.ly; a.batch = b.lz; a.interleaved = 1; a.plain_ends = 1;
			e = launch_inv_level(kCdf97S, a, g.tune, g.stream);
		}
		if (e != hipSuccess)
			return fail("3-D xy pass launch failed: %s", hipGetErrorString(e));
		// forward: the z pass also writes the next level's input densely (no lattice gather)
		e = launch_vol_z(inverse != 0, S, s_sy, s_sz, b.p, b.sy, b.sz, b.lx, b.ly, b.lz, g.vol, g.stream,
			next ? next->p : nullptr, next ? next->sy : 0, next ? next->sz : 0);
		if (e != hipSuccess)
			return fail("3-D z pass launch failed: %s", hipGetErrorString(e));
		return 0;
	};
	// level j lives on the stride-2 lattice (even-even-even samples) of level j-1
	auto lattice = [&](int j, bool pack) -> int {
		const Lvl &c = L[j], &par = L[j - 1];
		hipError_t e = pack
			? launch_lattice_copy(par.p, 2, par.sy * 2, par.sz * 2, c.p, 1, c.sy, c.sz, c.lx, c.ly, c.lz, g.stream)
			: launch_lattice_scatter(c.p, c.sy, c.sz, par.p, 2, par.sy * 2, par.sz * 2, c.lx, c.ly, c.lz, par.lx, g.stream);
		if (e != hipSuccess)
			return fail("lattice %s failed: %s", pack ? "pack" : "unpack", hipGetErrorString(e));
		return 0;
	};

	if (!inverse) {
		for (int j = 0; j < levels; j++)
			if (one_level(L[j], j + 1 < levels ? &L[j + 1] : nullptr))
				return 1;
		for (int j = levels - 1; j >= 1; j--)
			if (lattice(j, false))
				return 1;
	} else {
		for (int j = 1; j < levels; j++)
			if (lattice(j, true))
				return 1;
		for (int j = levels - 1; j >= 0; j--) {
			if (j >= 1 && g.vol.direct && ip_level(L[j].p, L[j].sy, L[j].sz, L[j].lx, L[j].ly, L[j].lz)) {
				// a level >= 1 in ONE pass, out of place: from its dense copy straight into the lattice of the
				// level above (no shell: the source stays intact; no dense result, no scatter pass)
				const Lvl &c = L[j], &par = L[j - 1];
				VolFusedArgs fa{c.p, c.sy, c.sz, par.p, par.sy * 2, par.sz * 2, nullptr, 0, 0, c.lx, c.ly, c.lz};
				fa.mode = 1;
				fa.out_sx = 2;
				hipError_t e = launch_vol_level_op(true, fa, g.vol, g.stream);
				if (e != hipSuccess)
					return fail("fused 3-D inverse level launch failed: %s", hipGetErrorString(e));
				continue;
			}
			if (one_level(L[j], nullptr))
				return 1;
			if (j >= 1 && lattice(j, false))
				return 1;
		}
	}
	return 0;
}


// ---- struct volume_t level (include/volume.h, include/volume-dwt.h): host or device volumes ----

// whole-volume transfer between a HOST volume (any byte strides: libdwt's "optimal" strides are odd
// byte counts) and a dense DEVICE volume of 4-byte samples.  Host rows the DMA engines like (64-byte
// multiples, 16-byte aligned) go as one 2-D copy, or one per slice when the slices are padded; any
// other layout goes slice by slice through host_upload / host_download (CPU repacking into a pinned
// buffer, pipelined with the transfer).  Padding is never touched.
static int vol_xfer(bool to_device, void *dev, size_t d_sy, size_t d_sz, void *host, size_t h_sy, size_t h_sz, int nx, int ny, int nz)
{
	const size_t row = (size_t)nx * 4;
	const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
	const bool fast = h_sy % 4 == 0 && h_sz % 4 == 0 && (uintptr_t)host % 4 == 0; // (see host_pitch_is_fast)
	if (!fast)
		return host_volume_xfer(to_device, dev, d_sy, d_sz, host, h_sy, h_sz, nx, ny, nz);
	void *dst = to_device ? dev : host;
	const void *src = to_device ? host : dev;
	const size_t dst_sy = to_device ? d_sy : h_sy, dst_sz = to_device ? d_sz : h_sz, src_sy = to_device ? h_sy : d_sy, src_sz = to_device ? h_sz : d_sz;
	if (d_sy == h_sy && d_sz == h_sz && h_sy == row && h_sz == row * ny) {
		HIP_TRY(hipMemcpyAsync(dst, src, row * ny * nz, kind, g.stream));
	} else if (d_sz == d_sy * ny && h_sz == h_sy * ny) {
		HIP_TRY(hipMemcpy2DAsync(dst, dst_sy, src, src_sy, row, (size_t)ny * nz, kind, g.stream));
	} else {
		for (int z = 0; z < nz; z++)
			HIP_TRY(hipMemcpy2DAsync((char *)dst + z * dst_sz, dst_sy, (const char *)src + z * src_sz, src_sy, row, ny, kind, g.stream));
	}
	HIP_TRY(hipStreamSynchronize(g.stream));
	return 0;
}

static int vol_check(const void *p, size_t sy, size_t sz, int nx, int ny, int nz, const char *who)
{
	if (!p || nx < 1 || ny < 1 || nz < 1)
		return fail("%s: empty volume", who);
	if (sy < (size_t)nx * 4 || sz < sy * (size_t)ny)
		return fail("%s: bad volume strides", who);
	if (dwt_hip_is_device_pointer(p) && ((sy & 3) || (sz & 3)))
		return fail("%s: device volumes need strides that are multiples of 4 bytes", who);
	return 0;
}

// one direction only, in place on a device volume (the reference's VOL_SEP_HORIZONTAL_X / _Y / _Z
// measurements, src/volume-dwt.c:788-981): exact line passes through a staging volume
static int vol_one_direction(float *vol, long sy, long sz, int nx, int ny, int nz, int dir)
{
	// the line pass shares its strides between source and destination: the staging volume mirrors the caller's
	if (grow(&g.stage_img, &g.stage_bytes, (size_t)sz * nz * 4))
		return 1;
	float *T = (float *)g.stage_img;
	hipError_t e = hipSuccess;
	if (dir == 4) {
		// the z lines of row y: adjacent lanes take adjacent x
		for (int y = 0; y < ny && e == hipSuccess; y++)
			e = launch_line_pass(kCdf97S, false, vol + (long)y * sy, T + (long)y * sy, 4, sz * 4, nx, nz, -1, true, g.stream);
	} else {
		for (int z = 0; z < nz && e == hipSuccess; z++)
			e = dir == 1 ? launch_line_pass(kCdf97S, false, vol + (long)z * sz, T + (long)z * sz, sy * 4, 4, ny, nx, -1, false, g.stream)
			             : launch_line_pass(kCdf97S, false, vol + (long)z * sz, T + (long)z * sz, 4, sy * 4, nx, ny, -1, true, g.stream);
	}
	if (e != hipSuccess)
		return fail("3-D line pass launch failed: %s", hipGetErrorString(e));
	e = launch_lattice_copy(T, 1, sy, sz, vol, 1, sy, sz, nx, ny, nz, g.stream);
	if (e != hipSuccess)
		return fail("3-D copy back failed: %s", hipGetErrorString(e));
	return 0;
}

// Forward, out of place, on the fields of two struct volume_t (cdf97_3f_op_sep_horizontal_s,
// src/volume-dwt.c:727-785, and the single-direction variants :788-981).  Host or device pointers,
// each with its own strides.  dirs: 7 = the transform; 1 = x only (copy, then x lines); 2 / 4 = y / z
// only, IN PLACE on dst as the reference does (src is not read).
int dwt_hip_volume_fwd_op(const void *src, size_t s_sy, size_t s_sz, void *dst, size_t d_sy, size_t d_sz, int nx, int ny, int nz, int dirs)
{
	if (check_inited())
		return 1;
	if (dirs != 7 && dirs != 1 && dirs != 2 && dirs != 4)
		return fail("dwt_hip_volume_fwd_op: dirs must be 7, 1, 2 or 4");
	const bool reads_src = dirs == 7 || dirs == 1;
	if ((reads_src && vol_check(src, s_sy, s_sz, nx, ny, nz, __func__)) || vol_check(dst, d_sy, d_sz, nx, ny, nz, __func__))
		return 1;
	if (reads_src && src == dst)
		return fail("dwt_hip_volume_fwd_op is out of place");
	const bool s_dev = !reads_src || dwt_hip_is_device_pointer(src), d_dev = dwt_hip_is_device_pointer(dst);
	const long t_sy = align_up(nx, 4), t_sz = t_sy * ny;
	const float *S = (const float *)src;
	float *D = (float *)dst;
	long S_sy = (long)s_sy / 4, S_sz = (long)s_sz / 4, D_sy = (long)d_sy / 4, D_sz = (long)d_sz / 4;
	if (!s_dev) {
		if (grow(&g.vol_host[0], &g.vol_host_bytes[0], (size_t)t_sz * nz * 4) ||
			vol_xfer(true, g.vol_host[0], t_sy * 4, t_sz * 4, (void *)src, s_sy, s_sz, nx, ny, nz))
			return 1;
		S = (const float *)g.vol_host[0];
		S_sy = t_sy; S_sz = t_sz;
	}
	if (!d_dev) {
		if (grow(&g.vol_host[1], &g.vol_host_bytes[1], (size_t)t_sz * nz * 4))
			return 1;
		D = (float *)g.vol_host[1];
		D_sy = t_sy; D_sz = t_sz;
		if (!reads_src && vol_xfer(true, D, t_sy * 4, t_sz * 4, dst, d_sy, d_sz, nx, ny, nz))
			return 1;
	}
	int rc;
	if (dirs == 7) {
		rc = vol_forward_op(S, S_sy, S_sz, D, D_sy, D_sz, nx, ny, nz, 1);
	} else if (dirs == 1) {
		// copy every x line to the destination, lift it there (src/volume-dwt.c:800-813)
		hipError_t e = launch_lattice_copy(S, 1, S_sy, S_sz, D, 1, D_sy, D_sz, nx, ny, nz, g.stream);
		rc = e == hipSuccess ? vol_one_direction(D, D_sy, D_sz, nx, ny, nz, 1) : fail("volume copy failed: %s", hipGetErrorString(e));
	} else {
		rc = vol_one_direction(D, D_sy, D_sz, nx, ny, nz, dirs);
	}
	if (rc)
		return 1;
	if (!d_dev)
		return vol_xfer(false, D, t_sy * 4, t_sz * 4, dst, d_sy, d_sz, nx, ny, nz);
	return 0;
}

// One level in place on the fields of a struct volume_t (cdf97_3f_ip_sep_horizontal_s /
// cdf97_3i_ip_sep_horizontal_s, src/volume-dwt.c:677, :1115); host volumes are staged through HBM.
int dwt_hip_volume_ip(int inverse, void *data, size_t sy, size_t sz, int nx, int ny, int nz)
{
	if (check_inited() || vol_check(data, sy, sz, nx, ny, nz, __func__))
		return 1;
	if (dwt_hip_is_device_pointer(data))
		return dwt_hip_transform3d(inverse, data, sy, sz, nx, ny, nz, 1);
	const long t_sy = align_up(nx, 4), t_sz = t_sy * ny;
	if (grow(&g.vol_host[0], &g.vol_host_bytes[0], (size_t)t_sz * nz * 4) ||
		vol_xfer(true, g.vol_host[0], t_sy * 4, t_sz * 4, data, sy, sz, nx, ny, nz) ||
		dwt_hip_transform3d(inverse, g.vol_host[0], t_sy * 4, t_sz * 4, nx, ny, nz, 1))
		return 1;
	return vol_xfer(false, g.vol_host[0], t_sy * 4, t_sz * 4, data, sy, sz, nx, ny, nz);
}

} // extern "C"
#pragma GCC visibility pop
