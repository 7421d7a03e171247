// dwt_vol3d_ip.hip -- one 3-D level in ONE pass and IN PLACE, forward and inverse
// (cdf97_3f_ip_sep_horizontal_s / cdf97_3i_ip_sep_horizontal_s, src/volume-dwt.c:677-725,
// :1115-1163: x lines, then y, then z, both directions; interleaved layout, so every coefficient
// is written where its input sample was).
//
// The fused level of dwt_vol3d.hip cannot run in place as it is: a tile (256 columns x 64 or 32 rows,
// marching along z) reads its neighbours' rows, columns and -- at the ends of its march -- slices,
// which those neighbours overwrite at a time of their own.  Everything ELSE a tile reads is its
// own and still unwritten when it is read (its stores trail its reads by four slices).  So the
// level takes a snapshot of exactly the foreign part first -- the SHELL: 7 of every 64 (32) rows, 8
// of every 256 columns, 9 slices around every march boundary and the 5 the reflection at the far
// end re-reads; a seventh of the volume, bandwidth-bound copies -- and the fused kernel then reads
// the interior of its tile from the volume and its halo from the shell, and writes the volume:
// 10.8 B per voxel (measured, 1024^3) instead of the 16 of two passes through a scratch volume.
//
// The same kernel runs OUT OF PLACE without a shell (launch_vol_level_op: the source stays intact):
// level 0 of the out-of-place forward calls in its 64-row tiles, and the levels >= 1 of a multi-level
// inverse from the dense copy of their lattice straight into the lattice of the level above.
//
// Arithmetic, operand order and reflection are those of k_vol_fwd_fused / k_inv_sweep / k_vol_z,
// hence the reference's bits.
#include "dwt_device.h"

namespace dwt {

// ---------------------------------------------------------------------------------
// the shell
// ---------------------------------------------------------------------------------
// rows: boundary b (between the tile rows b and b+1) holds the rows 32 (b+1) - HL .. + 6, HL = rows
// a tile reads above itself (4 forward, 3 inverse)
__global__ __launch_bounds__(256) void k_shell_rows(const float *__restrict__ in, long in_sy, long in_sz, VolShell sh, int nx, int ny, int hl, int ty_rows)
{
	const int k = blockIdx.y, b = k / 7, o = k % 7, z = blockIdx.z;
	const int r = ty_rows * (b + 1) - hl + o;
	if (r >= ny)
		return;
	const unsigned x = (blockIdx.x * 256 + threadIdx.x) * 16;
	const row_rsrc_t s = row_rsrc(in + (long)z * in_sz + (long)r * in_sy, (unsigned)nx * 4);
	const row_rsrc_t d = row_rsrc(sh.rs + (long)z * sh.rs_sz + (long)k * sh.rs_sy, (unsigned)nx * 4);
	// read once here (its owner reads the volume's row again much later), written for one reader: both
	// non-temporal (1024^3: 330 -> 285 us)
	store16_row<true>(d, x, load16_row<true>(s, x));
}

// columns: boundary b (between the tile columns b and b+1) holds the columns 256 (b+1) - 4 .. + 3
// (clamped to the row): one thread per 16-byte piece
__global__ __launch_bounds__(256) void k_shell_cols(const float *__restrict__ in, long in_sy, long in_sz, VolShell sh, int nx, int ny, int npc)
{
	const int i = blockIdx.x * 256 + threadIdx.x; // (row, piece)
	const int y = i / npc, k = i % npc, z = blockIdx.y;
	if (y >= ny)
		return;
	const int col = 256 * (k / 2 + 1) - 4 + 4 * (k & 1);
	const float *row = in + (long)z * in_sz + (long)y * in_sy;
	u4 v = load16_row<true>(row_rsrc(row, (unsigned)nx * 4), (unsigned)col * 4);
	// (a piece that straddles the row's end: its missing columns are never read -- reflection
	// brings them from the tile's own side)
	store16_row<false>(row_rsrc(sh.cs + (long)z * sh.cs_sz + (long)y * sh.cs_sy, (unsigned)npc * 16), (unsigned)k * 16, v);
}

// slices: 9 around every march boundary (2B-5 .. 2B+3), then the 5 before the last one (nz-6 .. nz-2)
__global__ __launch_bounds__(256) void k_shell_slices(const float *__restrict__ in, long in_sy, long in_sz, VolShell sh, int nx, int ny, int nz)
{
	const int slot = blockIdx.z, nb = 9 * (sh.nzt - 1);
	const int z = slot < nb ? 2 * (slot / 9 + 1) * sh.tile_pairs_z - 5 + slot % 9 : nz - 6 + (slot - nb);
	if (z < 0 || z >= nz)
		return;
	const int y = blockIdx.y;
	const unsigned x = (blockIdx.x * 256 + threadIdx.x) * 16;
	const row_rsrc_t s = row_rsrc(in + (long)z * in_sz + (long)y * in_sy, (unsigned)nx * 4);
	const row_rsrc_t d = row_rsrc(sh.zs + (long)slot * sh.zs_sz + (long)y * sh.zs_sy, (unsigned)nx * 4);
	store16_row<true>(d, x, load16_row<true>(s, x));
}

// ---------------------------------------------------------------------------------
// the level
// ---------------------------------------------------------------------------------
// Structure of k_vol_fwd_fused (its lean variant: every row a buffer, no per-lane bounds code): a
// workgroup of 4 waves owns 256 x 32 voxel columns; per slice each wave LDS-DMAs 10 of the tile's 39
// rows, lifts them along x in registers (two rows at a time, packed fp32), parks them in a shared LDS
// slab; after a barrier lifts the 15 slab rows around its 8 output rows along y, and feeds the 8 x 4
// samples per lane into the streaming z recurrence whose state stays in registers for the march.
// INV runs the inverse steps in the same x, y, z order (the reference's inverse keeps the axis order,
// src/volume-dwt.c:1115-1163); an inverse output row needs 3 rows above and 4 below where the
// forward needs 4 and 3, so its window of 39 rows starts one row later.
// MODE 4: every row of the level stored densely; MODE 2 (forward, level 0 of a multi-level call): the
// rows with even y in the even slices are withheld -- level 1 writes them whole -- and their odd-x
// samples parked in `side` (see k_vol_fwd_fused).
// NW: waves per workgroup = 4 (tile of 32 rows, two workgroups per CU) or 8 (tile of 64 rows, ONE
// workgroup of 512 threads per CU: half the halo rows -- 7 per 64 -- at the price of one barrier domain).
template <bool INV, int MODE, int NT, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void k_vol_level_ip(VolFusedArgs a, VolShell sh, int ntx, int nty, int swz)
{
	using W = Cdf97S;
	constexpr int K = 4, CPT = 4, TW = 256, RS = TW + 8, RW = 8, TY = NW * RW, NR = TY + 2 * K - 1, RPW = 10; // 39 = 10+10+10+9 rows, 71 = 7 x 9 + 8 (one slot idle)
	static_assert((NR + NW - 1) / NW <= RPW, "rows per wave");
	constexpr int NV = RW + 2 * K - 1; // slab rows a wave's vertical lift reads
	constexpr int HL = INV ? K - 1 : K; // halo rows above the tile
	constexpr int kLdAux = (NT & 2) ? 2 : 0;
	constexpr bool kNtStore = (NT & 1) != 0;
	static_assert(MODE == 4 || (MODE == 2 && !INV) || (MODE == 1 && INV), "store variants");
	// no row shell: an OUT-OF-PLACE call (a.in != a.out) -- the source stays intact, every row, column and
	// slice a tile reads comes from it (inverse levels >= 1 of a multi-level call: dense source, result
	// into the lattice of the level above, MODE 1)
	const bool ip = sh.rs != nullptr;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int bid = tile_block_id(swz);
	const int tx = bid % ntx, ty = (bid / ntx) % nty, tz = bid / (ntx * nty);
	const int c0 = tx * TW, c = c0 + lane * CPT, y0 = ty * TY;
	const int Zd = (a.nz + 1) >> 1;
	const int A = tz * sh.tile_pairs_z;
	if (A >= Zd)
		return; // the whole workgroup leaves together
	const int B = min(A + sh.tile_pairs_z, Zd);
	const int n_iter = (B - A) + K, q0 = A - K / 2;
	const int n_slices = 2 * n_iter;
	const int vfirst = 2 * q0 - (INV ? 0 : 1); // virtual slice of step 0
	const int own_end = min(2 * B, a.nz);

	char *ring = smem + (size_t)wv * RPW * RS * 4;
	// (4 waves: the last wave's idle tenth slot overlaps the slab's first row, so that two workgroups fit a CU)
	constexpr int kRingRows = NW == 4 ? NR : NW * RPW;
	char *slab = smem + (size_t)kRingRows * RS * 4;
	const unsigned ring_off = lds_offset(ring), slab_off = lds_offset(slab);

	// Where a staged row comes from does not change along the march: lane i of the wave keeps the
	// facts of the wave's i-th row (tile row wv + NW i) and the loop fetches them with v_readlane --
	// bits 0..15 the (reflected) row, 16 "this tile's own row", 17 "in the row shell", 18.. its row there.
	int rowinfo;
	{
		const int r = reflect(y0 - HL + min(wv + NW * lane, NR - 1), a.ny);
		const bool own = r >= y0 && r < y0 + TY;
		// a neighbour's row: boundary b = (r + HL) / TY - 1, offset (r + HL) % TY < 7.  (Rows outside
		// that set are reached only by reflections that feed no valid output.)
		const int u = r + HL, b = u / TY - 1, o = u % TY;
		const bool inrs = !own && o < 7 && b >= 0 && b < nty - 1;
		// (the shell row index takes bits 18..31: decoded as UNSIGNED, good for 7 * 2340 rows of tiles)
		rowinfo = (int)((unsigned)r | (ip && own ? 1u << 16 : 0u) | (ip && inrs ? (1u << 17) | ((unsigned)(7 * b + o) << 18) : 0u));
	}
	// Likewise the single columns a row needs beside its 16-byte pieces: lanes 8..15 fetch the tile's
	// halo columns (c0-4 .. c0-1, c0+256 .. c0+259), lanes 0..3 the reflected columns right of the
	// volume's edge inside an overhanging tile.  A column of this tile comes from the row itself, a
	// neighbour's column from the row itself too unless the row is the VOLUME's (then the neighbour may
	// have overwritten it: column shell).  colrow: offset in the row; colsh: offset in the column shell's
	// row, or -1.
	int colrow, colsh;
	{
		const int col = lane < 8 ? a.nx + (lane & 3) : lane < 12 ? c0 - 4 + (lane & 3) : c0 + TW + (lane & 3);
		const int rc = reflect(col, a.nx);
		colrow = rc;
		colsh = rc < c0 ? 8 * (tx - 1) + max(rc - (c0 - 4), 0) : rc >= c0 + TW ? 8 * tx + 4 + min(rc - (c0 + TW), 3) : -1;
	}
	const int over = min(4, c0 + TW - a.nx); // reflected columns inside the tile (<= 0: none)

	auto issue = [&](int t) {
		const int v = vfirst + t;
		// live: a slice of this march's own range, or (first march) its mirror image before the
		// volume's first slice -- unwritten so far: the tile's own part comes from the volume, the rest
		// from the row and column shells.  Anything else was (or is being) overwritten by another march
		// and comes from the slice shell as a whole.
		const bool live = !ip || (v < own_end && (v >= 2 * A || tz == 0));
		const int s = ip ? (v < 0 ? -v : v) : reflect(v, a.nz);
		int slot = 0;
		if (!live)
			slot = v >= a.nz ? 9 * (sh.nzt - 1) + (a.nz + 4 - v) : v < 2 * A ? 9 * (tz - 1) + (v - (2 * A - 5)) : 9 * tz + (v - (2 * B - 5));
		const float *sl = live ? a.in + (long)s * a.in_sz : sh.zs + (long)slot * sh.zs_sz;
		const long sy = live ? a.in_sy : sh.zs_sy;
		const float *rsl = sh.rs + (long)s * sh.rs_sz;
		const float *csl = sh.cs + (long)s * sh.cs_sz;
#pragma unroll 1
		for (int i = 0; i < RPW; i++) {
			if (wv + NW * i < NR) {
				const int info = __builtin_amdgcn_readlane(rowinfo, i);
				const int r = info & 0xffff;
				const float *grow = sl + (long)r * sy;
				if (live && (info & (1 << 17)))
					grow = rsl + (long)((unsigned)info >> 18) * sh.rs_sy;
				// the volume's own row: its foreign columns come from the column shell
				const bool shell_cols = live && (info & (1 << 16));
				const float *crow = csl + (long)r * sh.cs_sy;
				char *lrow = ring + (size_t)i * RS * 4;
				// lane i's 16 B land at lrow + 16 i; beyond the row's end the bounds check fills zeros, and
				// the four reflected columns next to the edge -- all a valid output can reach -- come one by one
				const float *cp = shell_cols && colsh >= 0 ? crow + colsh : grow + colrow;
				if (shell_cols) {
					// the volume's row: read by this tile only
					dma16_row<kLdAux>(row_rsrc(grow, (unsigned)a.nx * 4), (unsigned)c * 4, lrow);
					if (lane >= 8 && lane < 16)
						dma4<kLdAux>(cp, lrow + TW * 4 - 32); // lane 8's dword lands behind the row's 256 columns
					if (lane < over)
						dma4<kLdAux>(cp, lrow + (a.nx - c0) * 4);
				} else {
					// a shell row: the x-neighbour tiles read the lines next to this tile's 1 KiB at about the same
					// time and this tile needs 16 bytes of each -- cacheable, so that the second one hits in L2
					dma16_row<0>(row_rsrc(grow, (unsigned)a.nx * 4), (unsigned)c * 4, lrow);
					if (lane >= 8 && lane < 16)
						dma4<0>(cp, lrow + TW * 4 - 32);
					if (lane < over)
						dma4<0>(cp, lrow + (a.nx - c0) * 4);
				}
			}
		}
	};

	float st[K][RW][CPT], ra[RW][CPT];
#pragma unroll
	for (int s = 0; s < K; s++)
#pragma unroll
		for (int r = 0; r < RW; r++)
#pragma unroll
			for (int e = 0; e < CPT; e++)
				st[s][r][e] = 0.f;

	issue(0);
	for (int t = 0; t < n_slices; t++) {
		DWT_WAIT_VMCNT(0); // this slice's rows have landed (and the previous stores are out)
		// x: this wave's rows, two at a time as the halves of packed fp32 operations, the LDS reads of
		// the next pair in flight while this pair is lifted; parked in the shared slab
		{
			typedef float f2 __attribute__((ext_vector_type(2)));
			static_assert(RPW % 2 == 0, "rows per wave are lifted in pairs");
			const unsigned own0 = ring_off + lane * CPT * 4;
			const unsigned la0 = lane == 0 ? ring_off + TW * 4 : own0 - 16;
			const unsigned ra0 = lane == 63 ? ring_off + TW * 4 + 16 : own0 + CPT * 4;
			u4 L[2][2], O[2][2], R[2][2]; // [buffer][row of the pair]
			lds_issue3(la0, own0, ra0, L[0][0], O[0][0], R[0][0]);
			lds_issue3(la0 + RS * 4, own0 + RS * 4, ra0 + RS * 4, L[0][1], O[0][1], R[0][1]);
#pragma unroll
			for (int ip = 0; ip < RPW / 2; ip++) {
				const int b = ip & 1;
				if (ip + 1 < RPW / 2) {
					const unsigned d = (unsigned)(2 * ip + 2) * RS * 4;
					lds_issue3(la0 + d, own0 + d, ra0 + d, L[b ^ 1][0], O[b ^ 1][0], R[b ^ 1][0]);
					lds_issue3(la0 + d + RS * 4, own0 + d + RS * 4, ra0 + d + RS * 4, L[b ^ 1][1], O[b ^ 1][1], R[b ^ 1][1]);
					lds_arrived3<6>(L[b][0], O[b][0], R[b][0]);
					lds_arrived3<6>(L[b][1], O[b][1], R[b][1]);
				} else {
					lds_arrived3<0>(L[b][0], O[b][0], R[b][0]);
					lds_arrived3<0>(L[b][1], O[b][1], R[b][1]);
				}
				f2 x[CPT + 2 * K]; // columns c-4 .. c+7: element 0 is an even sample
#pragma unroll
				for (int e = 0; e < K; e++) {
					x[e] = f2{from_bits<float>(L[b][0][e]), from_bits<float>(L[b][1][e])};
					x[K + e] = f2{from_bits<float>(O[b][0][e]), from_bits<float>(O[b][1][e])};
					x[K + CPT + e] = f2{from_bits<float>(R[b][0][e]), from_bits<float>(R[b][1][e])};
				}
				f2 y0_, y1_, y2_, y3_;
				if constexpr (!INV) {
					// lift_fwd_regs on both rows at once: x[j] += c_s * (x[j-1] + x[j+1]), product and sums rounded separately
#pragma unroll
					for (int st_ = 0; st_ < K; st_++)
#pragma unroll
						for (int j = st_ + 1; j <= CPT + 2 * K - 2 - st_; j += 2)
							x[j] = x[j] + W::fc(st_) * (x[j - 1] + x[j + 1]);
					y0_ = x[K] * W::zeta(), y1_ = x[K + 1] * (1.0f / W::zeta()), y2_ = x[K + 2] * W::zeta(), y3_ = x[K + 3] * (1.0f / W::zeta());
				} else {
					// descale (even * 1/zeta, odd * zeta), then the inverse steps: step s acts on the samples of
					// parity s & 1, which with an even element 0 are the entries j of that parity
#pragma unroll
					for (int j = 1; j < CPT + 2 * K; j++)
						x[j] = (j & 1) ? x[j] * W::zeta() : x[j] * (1.0f / W::zeta());
#pragma unroll
					for (int st_ = 0; st_ < K; st_++)
#pragma unroll
						for (int j = st_ + 2; j <= CPT + 2 * K - 2 - st_; j += 2)
							x[j] = x[j] + W::ic(st_) * (x[j - 1] + x[j + 1]);
					y0_ = x[K], y1_ = x[K + 1], y2_ = x[K + 2], y3_ = x[K + 3];
				}
#pragma unroll
				for (int q = 0; q < 2; q++) {
					const int i = 2 * ip + q;
					if (wv + NW * i < NR) // (a wave may stage fewer rows than it has slots: the idle slot's lift is harmless)
						lds_write4(slab_off + (unsigned)(wv + NW * i) * TW * 4 + lane * 16,
							u4{to_bits(y0_[q]), to_bits(y1_[q]), to_bits(y2_[q]), to_bits(y3_[q])});
				}
			}
		}
		// the staging rows are consumed: the next slice's DMA flies during the rest of the iteration
		if (t + 1 < n_slices)
			issue(t + 1);
		wg_barrier_lds(); // the slab is complete

		// y: slab rows RW wv .. RW wv + NV - 1 give this wave's RW output rows
		u4 v[NV];
		{
			const unsigned vb = slab_off + (unsigned)(RW * wv) * TW * 4 + lane * 16;
#pragma unroll
			for (int j = 0; j < NV; j++)
				asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v[j]) : "v"(vb), "n"(j * TW * 4) : "memory");
			// the barrier: every wave has read the slab, the next slice may overwrite it
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
			for (int j = 0; j < NV; j++)
				asm volatile("" : "+v"(v[j])); // uses of v[j] stay below the wait
		}
		float cur[RW][CPT];
#pragma unroll
		for (int e = 0; e < CPT; e++) {
			float col[NV]; // rows RW wv - HL ..: element 0 is an even row (forward) / an odd row (inverse)
#pragma unroll
			for (int j = 0; j < NV; j++)
				col[j] = from_bits<float>(v[j][e]);
			if constexpr (!INV) {
				lift_fwd_regs<W, NV>(col);
#pragma unroll
				for (int r = 0; r < RW; r++)
					cur[r][e] = W::fwd_scale(r & 1, col[HL + r]);
			} else {
#pragma unroll
				for (int j = 0; j < NV; j++)
					col[j] = W::inv_scale((j + 1) & 1, col[j]);
				lift_inv_regs<W, NV>(col);
#pragma unroll
				for (int r = 0; r < RW; r++)
					cur[r][e] = col[HL + r];
			}
		}

		// z: slices arrive in pairs; the first one waits in registers for its partner
		if (!(t & 1)) {
#pragma unroll
			for (int r = 0; r < RW; r++)
#pragma unroll
				for (int e = 0; e < CPT; e++)
					ra[r][e] = cur[r][e];
			continue;
		}
		const int it = t >> 1;
#pragma unroll
		for (int r = 0; r < RW; r++) {
			float o0[CPT], o1[CPT];
#pragma unroll
			for (int e = 0; e < CPT; e++) {
				if constexpr (!INV) {
					// ra = slice 2q-1 (odd), cur = slice 2q (even); out: pair k = q-2
					const float d1n = W::fwd_step(0, ra[r][e], st[0][r][e], cur[r][e]);
					const float s1n = W::fwd_step(1, st[0][r][e], st[1][r][e], d1n);
					const float d2n = W::fwd_step(2, st[1][r][e], st[2][r][e], s1n);
					const float s2n = W::fwd_step(3, st[2][r][e], st[3][r][e], d2n);
					o0[e] = W::fwd_scale(0, s2n);
					o1[e] = W::fwd_scale(1, d2n);
					st[0][r][e] = cur[r][e];
					st[1][r][e] = d1n;
					st[2][r][e] = s1n;
					st[3][r][e] = d2n;
				} else {
					// ra = slice 2q (even), cur = slice 2q+1 (odd); out: slices 2q-3 (o0) and 2q-2 (o1)
					const float s2 = W::inv_scale(0, ra[r][e]), d2 = W::inv_scale(1, cur[r][e]);
					const float s1n = W::inv_step(0, s2, st[0][r][e], d2);
					const float d1n = W::inv_step(1, st[0][r][e], st[1][r][e], s1n);
					const float en = W::inv_step(2, st[1][r][e], st[2][r][e], d1n);
					const float on = W::inv_step(3, st[2][r][e], st[3][r][e], en);
					o0[e] = on;
					o1[e] = en;
					st[0][r][e] = d2;
					st[1][r][e] = s1n;
					st[2][r][e] = d1n;
					st[3][r][e] = en;
				}
			}
			const int y = y0 + RW * wv + r;
			if (y >= a.ny)
				continue;
			const unsigned nb = (unsigned)a.nx * 4, cb = (unsigned)c * 4;
			const u4 p0 = u4{to_bits(o0[0]), to_bits(o0[1]), to_bits(o0[2]), to_bits(o0[3])};
			const u4 p1 = u4{to_bits(o1[0]), to_bits(o1[1]), to_bits(o1[2]), to_bits(o1[3])};
			if constexpr (!INV) {
				const int k = A + it - K;
				if (it < K)
					continue;
				float *row0 = a.out + (long)(2 * k) * a.out_sz + (long)y * a.out_sy;
				const bool hz = 2 * k + 1 < a.nz;
				const bool to_lll = a.lll && !(r & 1);
				if (MODE == 2 && !(r & 1)) // rows with even y in the even slice are level 1's to write
					store8_row<kNtStore>(row_rsrc(a.side + (long)k * a.side_sz + (long)(y >> 1) * a.side_sy, (unsigned)(a.nx >> 1) * 4), cb / 2, u2{to_bits(o0[1]), to_bits(o0[3])});
				else
					store16_row<kNtStore>(row_rsrc(row0, nb), cb, p0);
				if (hz)
					store16_row<kNtStore>(row_rsrc(row0 + a.out_sz, nb), cb, p1);
				if (to_lll)
					store8_row<false>(row_rsrc(a.lll + (long)k * a.lll_sz + (long)(y >> 1) * a.lll_sy, (unsigned)((a.nx + 1) >> 1) * 4), cb / 2, u2{to_bits(o0[0]), to_bits(o0[2])});
			} else {
				const int q = q0 + it, po = q - 2, pe = q - 1;
				const bool st_o = po >= A && po < B && 2 * po + 1 < a.nz, st_e = pe >= A && pe < B;
				float *row_o = a.out + (long)(2 * po + 1) * a.out_sz + (long)y * a.out_sy, *row_e = a.out + (long)(2 * pe) * a.out_sz + (long)y * a.out_sy;
				if constexpr (MODE == 1) {
					// into the lattice of the level above: sample x of this level at x * out_sx of the row, one
					// dword store per sample, the row's last lattice sample bounds the buffer
					const unsigned nbl = ((unsigned)(a.nx - 1) * (unsigned)a.out_sx + 1) * 4, sx4 = (unsigned)a.out_sx * 4;
					const row_rsrc_t d_o = row_rsrc(row_o, nbl), d_e = row_rsrc(row_e, nbl);
#pragma unroll
					for (int e = 0; e < CPT; e++) {
						if (st_o)
							__builtin_amdgcn_raw_buffer_store_b32(to_bits(o0[e]), d_o, (unsigned)c * sx4, e * sx4, 0);
						if (st_e)
							__builtin_amdgcn_raw_buffer_store_b32(to_bits(o1[e]), d_e, (unsigned)c * sx4, e * sx4, 0);
					}
				} else {
					if (st_o)
						store16_row<kNtStore>(row_rsrc(row_o, nb), cb, p0);
					if (st_e)
						store16_row<kNtStore>(row_rsrc(row_e, nb), cb, p1);
				}
			}
		}
	}
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
static int vol_ip_march(const VolFusedArgs &a, const VolTuning &vt, int nw = 4)
{
	// The model of launch_vol_fwd_fused -- rounds of workgroups over the chip's slots (4 waves: two
	// workgroups per CU, 512 slots, a workgroup that has its CU to itself marches faster; 8 waves: one per
	// CU, 256) times (slice pairs + 4 warm-up pairs) -- over BALANCED marches: the depth is cut into nzt
	// equal parts (896^3: 4 x 112 pairs instead of 128 + 128 + 128 + 64).
	const int Zd = (a.nz + 1) / 2, ntx = (a.nx + 255) / 256, nty = (a.ny + 8 * nw - 1) / (8 * nw);
	const double slots = nw == 4 ? 512 : 256;
	int tp = Zd;
	double best = -1;
	for (int nzt = 1; nzt <= 64; nzt++) {
		const int cand = (Zd + nzt - 1) / nzt;
		if (cand < 8 && nzt > 1)
			break;
		const double n = (double)ntx * nty * ((Zd + cand - 1) / cand);
		const double rounds = n <= slots ? ((nw == 4 && n <= slots / 2) ? 0.6 : 1.0) : n / slots + 0.35;
		const double cost = rounds * (cand + 4);
		if (best < 0 || cost < best) {
			best = cost;
			tp = cand;
		}
	}
	if (tp > 256)
		tp = 256; // (bounds the slice shell's distance from the data it mirrors; deeper volumes take more marches)
	if (vt.tile_pairs >= 4)
		tp = vt.tile_pairs;
	return tp < 4 ? 4 : tp;
}

// Waves per workgroup: tiles of 64 rows (8 waves, one workgroup per CU) halve the halo rows and win at
// every size measured (1024^3 in place: forward 2.30 -> 1.96 ms, inverse 2.38 -> 2.16; 512^3 0.40 -> 0.36;
// scripts/archive/r03/r03_vol_ip_waves.py) -- unless the volume has a single tile row of 32 anyway.
static int vol_ip_waves(const VolFusedArgs &a, const VolTuning &vt)
{
	if (vt.ip_waves == 4 || vt.ip_waves == 8)
		return vt.ip_waves;
	return a.ny > 32 ? 8 : 4;
}

static size_t vol_shell_layout(const VolFusedArgs &a, int tp, float *base, VolShell *sh, int nw = 4)
{
	const int ntx = (a.nx + 255) / 256, nty = (a.ny + 8 * nw - 1) / (8 * nw), Zd = (a.nz + 1) / 2;
	const int nzt = (Zd + tp - 1) / tp;
	VolShell s{};
	s.tile_pairs_z = tp;
	s.nzt = nzt;
	const long row = (a.nx + 3) / 4 * 4;
	size_t off = 0;
	s.rs = base + off;
	s.rs_sy = row;
	s.rs_sz = row * 7 * (nty - 1);
	off += (size_t)s.rs_sz * a.nz;
	s.cs = base + off;
	s.cs_sy = 8 * (ntx - 1);
	s.cs_sz = s.cs_sy * a.ny;
	off += (size_t)s.cs_sz * a.nz;
	off = (off + 3) / 4 * 4;
	s.zs = base + off;
	s.zs_sy = row;
	s.zs_sz = row * a.ny;
	off += (size_t)s.zs_sz * (9 * (nzt - 1) + 5);
	if (sh)
		*sh = s;
	return off;
}

bool vol_level_ip_can(const VolFusedArgs &a)
{
	return a.nx >= 2 && a.ny >= 2 && a.nz >= 8 && a.ny <= 65535 && a.nz <= 65535;
}

bool vol_level_ip_applies(const VolFusedArgs &a)
{
	// about one workgroup per CU at 32 slice pairs per march, tiles mostly used.  Measured (one level,
	// forward / inverse, ms; scripts/archive/r03/r03_vol_ip_sizes.py): 448^3 one pass 0.33 / 0.36 against two passes
	// 0.28 / 0.29 (which run partly out of the Infinity Cache); 512^3 0.38 / 0.41 against 0.41 / 0.42;
	// 640^3 0.80 / 0.87 against 0.90 / 0.86; 768^3 1.05 / 1.13 against 1.50 / 1.46
	if (!vol_level_ip_can(a) || a.nx < 128)
		return false;
	const long tiles = (long)((a.nx + 255) / 256) * ((a.ny + 31) / 32);
	return tiles * (((a.nz + 1) / 2 + 31) / 32) >= 256;
}

size_t vol_level_ip_scratch(const VolFusedArgs &a, const VolTuning &vt)
{
	const int nw = vol_ip_waves(a, vt);
	return vol_shell_layout(a, vol_ip_march(a, vt, nw), nullptr, nullptr, nw) * 4;
}

template <bool INV, int MODE, int NT, int NW = 4>
static hipError_t vol_ip_go(const VolFusedArgs &a, const VolShell &sh, int ntx, int nty, int swz, hipStream_t s)
{
	constexpr int NR = 8 * NW + 7;
	const size_t lds = (size_t)(NW == 4 ? NR : NW * 10) * (256 + 8) * 4 + (size_t)NR * 256 * 4;
	if (hipError_t e = allow_lds((const void *)k_vol_level_ip<INV, MODE, NT, NW>, lds))
		return e;
	k_vol_level_ip<INV, MODE, NT, NW><<<dim3(ntx * nty * sh.nzt), 64 * NW, lds, s>>>(a, sh, ntx, nty, swz);
	return hipGetLastError();
}

hipError_t launch_vol_level_ip(bool inverse, const VolFusedArgs &a, float *scratch, const VolTuning &vt, hipStream_t s)
{
	if (!vol_level_ip_can(a) || a.in != a.out || a.in_sy != a.out_sy || a.in_sz != a.out_sz || !scratch)
		return hipErrorInvalidValue;
	if (a.mode != 0 && !(a.mode == 2 && !inverse && a.side))
		return hipErrorInvalidValue;
	const int nw = vol_ip_waves(a, vt);
	VolShell sh;
	vol_shell_layout(a, vol_ip_march(a, vt, nw), scratch, &sh, nw);
	const int ntx = (a.nx + 255) / 256, nty = (a.ny + 8 * nw - 1) / (8 * nw);
	if ((long)ntx * nty * sh.nzt > 0x7fffffffL)
		return hipErrorInvalidValue;
	const int hl = inverse ? 3 : 4;
	const int nbx = ((a.nx + 3) / 4 + 255) / 256;
	if (nty > 1)
		k_shell_rows<<<dim3(nbx, 7 * (nty - 1), a.nz), 256, 0, s>>>(a.in, a.in_sy, a.in_sz, sh, a.nx, a.ny, hl, 8 * nw);
	if (ntx > 1) {
		const int npc = 2 * (ntx - 1);
		k_shell_cols<<<dim3((unsigned)(((long)a.ny * npc + 255) / 256), a.nz), 256, 0, s>>>(a.in, a.in_sy, a.in_sz, sh, a.nx, a.ny, npc);
	}
	k_shell_slices<<<dim3(nbx, a.ny, 9 * (sh.nzt - 1) + 5), 256, 0, s>>>(a.in, a.in_sy, a.in_sz, sh, a.nx, a.ny, a.nz);
	if (hipError_t e = hipGetLastError())
		return e;
	// cache policy: a tile's own rows are loaded by no other tile (the halo comes from the shell): non-temporal
	// like the stores (1024^3: inverse 2.51 -> 2.47 ms, forward unchanged); shell rows stay cacheable
	const int want = vt.nt < 0 ? 3 : (vt.nt & 3);
	const bool nt_loads = (want & 2) != 0;
#define DWT_IP_GO(INV_, MODE_) \
	do { \
		if (nw == 8) \
			return nt_loads ? vol_ip_go<INV_, MODE_, 3, 8>(a, sh, ntx, nty, vt.swizzle, s) : vol_ip_go<INV_, MODE_, 1, 8>(a, sh, ntx, nty, vt.swizzle, s); \
		return nt_loads ? vol_ip_go<INV_, MODE_, 3>(a, sh, ntx, nty, vt.swizzle, s) : vol_ip_go<INV_, MODE_, 1>(a, sh, ntx, nty, vt.swizzle, s); \
	} while (0)
	if (inverse)
		DWT_IP_GO(true, 4);
	if (a.mode == 2)
		DWT_IP_GO(false, 2);
	DWT_IP_GO(false, 4);
#undef DWT_IP_GO
}

// One level in one pass, OUT OF PLACE, with this file's kernel: the source `a.in` stays intact, so no
// shell is needed (every row, column and slice a tile reads comes from the source; cacheable loads:
// neighbouring tiles share their halo lines).  Forward: a.mode 0 (dense result) or 2 (level 0 of a
// multi-level call withholding the rows level 1 writes whole, see k_vol_fwd_fused), `lll` as there;
// tiles of 64 rows (8 waves) where the volume has more than 32 rows.  Inverse: a.mode 0, or 1 = result
// into the stride-out_sx lattice of `a.out` (out_sy / out_sz the destination's strides times out_sx):
// a level >= 1 of a multi-level inverse writing straight into the level above instead of a dense
// result plus a scatter pass.
hipError_t launch_vol_level_op(bool inverse, const VolFusedArgs &a, const VolTuning &vt, hipStream_t s)
{
	if (!vol_level_ip_can(a) || a.in == a.out)
		return hipErrorInvalidValue;
	if (inverse ? ((a.mode != 0 && a.mode != 1) || (a.mode == 0 && a.out_sx != 1)) : (a.mode != 0 && !(a.mode == 2 && a.side)))
		return hipErrorInvalidValue;
	const int nw = vol_ip_waves(a, vt);
	VolShell sh{};
	sh.tile_pairs_z = vol_ip_march(a, vt, nw);
	sh.nzt = ((a.nz + 1) / 2 + sh.tile_pairs_z - 1) / sh.tile_pairs_z;
	const int ntx = (a.nx + 255) / 256, nty = (a.ny + 8 * nw - 1) / (8 * nw);
	if ((long)ntx * nty * sh.nzt > 0x7fffffffL)
		return hipErrorInvalidValue;
#define DWT_OP_GO(INV_, MODE_) return nw == 8 ? vol_ip_go<INV_, MODE_, 1, 8>(a, sh, ntx, nty, vt.swizzle, s) : vol_ip_go<INV_, MODE_, 1, 4>(a, sh, ntx, nty, vt.swizzle, s)
	if (inverse) {
		if (a.mode == 1)
			DWT_OP_GO(true, 1);
		DWT_OP_GO(true, 4);
	}
	if (a.mode == 2)
		DWT_OP_GO(false, 2);
	DWT_OP_GO(false, 4);
#undef DWT_OP_GO
}

} // namespace dwt
