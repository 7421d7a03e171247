// dwt_vol3d.hip -- the 3-D path (BASELINE config 5): the z pass of the two-pass scheme, the fused
// one-pass level (x, y and z lifting in one kernel) and the strided lattice copy.
#include "dwt_device.h"

namespace dwt {

// ---------------------------------------------------------------------------------
// 3. z pass of the 3-D path and the lattice copy
// ---------------------------------------------------------------------------------
// One wave owns 256 contiguous x columns of one row y and marches along z with the
// lifting state in registers (the z neighbours of a sample are whole slices apart, but
// each access is a contiguous 1 KiB row segment).  Same streaming recurrences as the
// vertical pass of the 2-D sweeps; out of place, because the symmetric extension at
// the far end re-reads slices the sweep has already produced.
template <bool INV, int CPT, int NT>
__global__ __launch_bounds__(256) void k_vol_z(const float *__restrict__ in, long in_sy, long in_sz,
	float *__restrict__ out, long out_sy, long out_sz, int nx, int ny, int nz, int tile_pairs,
	float *__restrict__ lll, long lll_sy, long lll_sz)
{
	using W = Cdf97S;
	constexpr int K = 4, NV = CPT / 4;
	const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
	// wave-uniform on purpose: tile geometry, row indices and row pointers then live in SGPRs
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// a lane owns NV groups of 4 columns, 256 columns apart: every load/store instruction
	// of the wave is one contiguous 1 KiB segment
	const int c = (blockIdx.x * nwv + wv) * 64 * CPT + lane * 4;
	const int y = blockIdx.y;
	const int Zd = (nz + 1) >> 1;
	const int A = blockIdx.z * tile_pairs;
	if (A >= Zd || (blockIdx.x * nwv + wv) * 64 * CPT >= nx)
		return;
	const int B = min(A + tile_pairs, Zd);
	const int n_iter = (B - A) + K;
	const int q0 = A - K / 2;
	// rows as buffers (per-dword hardware bounds check, 4-byte alignment suffices): the segments
	// that overhang the volume load zeros and drop their stores, no per-lane branches
	const float *src = in + (long)y * in_sy;
	float *dst = out + (long)y * out_sy;
	const unsigned cb = (unsigned)c * 4, nb_row = (unsigned)nx * 4;

	auto load = [&](int slice, float (&v)[CPT]) {
		const row_rsrc_t rs = row_rsrc(src + (long)reflect(slice, nz) * in_sz, nb_row);
#pragma unroll
		for (int g = 0; g < NV; g++) {
			const u4 t = load16_row<(NT & 2) != 0>(rs, cb + 1024 * g);
#pragma unroll
			for (int e = 0; e < 4; e++)
				v[4 * g + e] = from_bits<float>(t[e]);
		}
	};
	auto store = [&](int slice, const float (&v)[CPT]) {
		const row_rsrc_t rd = row_rsrc(dst + (long)slice * out_sz, nb_row);
#pragma unroll
		for (int g = 0; g < NV; g++)
			store16_row<(NT & 1) != 0>(rd, cb + 1024 * g, u4{to_bits(v[4 * g]), to_bits(v[4 * g + 1]), to_bits(v[4 * g + 2]), to_bits(v[4 * g + 3])});
	};

	float st[K][CPT];
#pragma unroll
	for (int s = 0; s < K; s++)
#pragma unroll
		for (int e = 0; e < CPT; e++)
			st[s][e] = 0.f;

	float na[CPT], nb[CPT];
	load(2 * q0 - (INV ? 0 : 1), na);
	load(2 * q0 + (INV ? 1 : 0), nb);
	for (int it = 0; it < n_iter; it++) {
		const int q = q0 + it;
		float ra[CPT], rb[CPT];
#pragma unroll
		for (int e = 0; e < CPT; e++) {
			ra[e] = na[e];
			rb[e] = nb[e];
		}
		if (it + 1 < n_iter) { // software prefetch of the next pair of slices
			load(2 * (q + 1) - (INV ? 0 : 1), na);
			load(2 * (q + 1) + (INV ? 1 : 0), nb);
		}
		float o0[CPT], o1[CPT];
#pragma unroll
		for (int e = 0; e < CPT; e++) {
			if constexpr (!INV) {
				// ra = slice 2q-1 (odd), rb = slice 2q (even)
				const float d1n = W::fwd_step(0, ra[e], st[0][e], rb[e]);
				const float s1n = W::fwd_step(1, st[0][e], st[1][e], d1n);
				const float d2n = W::fwd_step(2, st[1][e], st[2][e], s1n);
				const float s2n = W::fwd_step(3, st[2][e], st[3][e], d2n);
				o0[e] = W::fwd_scale(0, s2n);
				o1[e] = W::fwd_scale(1, d2n);
				st[0][e] = rb[e];
				st[1][e] = d1n;
				st[2][e] = s1n;
				st[3][e] = d2n;
			} else {
				// ra = slice 2q (even, s2'), rb = slice 2q+1 (odd, d2')
				const float s2 = W::inv_scale(0, ra[e]), d2 = W::inv_scale(1, rb[e]);
				const float s1n = W::inv_step(0, s2, st[0][e], d2);
				const float d1n = W::inv_step(1, st[0][e], st[1][e], s1n);
				const float en = W::inv_step(2, st[1][e], st[2][e], d1n);
				const float on = W::inv_step(3, st[2][e], st[3][e], en);
				o0[e] = on; // slice 2q-3
				o1[e] = en; // slice 2q-2
				st[0][e] = d2;
				st[1][e] = s1n;
				st[2][e] = d1n;
				st[3][e] = en;
			}
		}
		if constexpr (!INV) {
			if (it >= K) {
				const int k = A + it - K;
				store(2 * k, o0);
				if (2 * k + 1 < nz)
					store(2 * k + 1, o1);
				// forward multi-level: the next level's input (even x, even y, even z = LLL)
				// also goes out densely, so that no lattice gather is needed
				if (lll && !(y & 1)) {
					const row_rsrc_t rl = row_rsrc(lll + (long)k * lll_sz + (long)(y >> 1) * lll_sy, (unsigned)((nx + 1) >> 1) * 4);
#pragma unroll
					for (int g = 0; g < NV; g++)
						store8_row<false>(rl, cb / 2 + 512 * g, u2{to_bits(o0[4 * g]), to_bits(o0[4 * g + 2])});
				}
			}
		} else {
			const int pe = q - 1, po = q - 2;
			if (po >= A && po < B && 2 * po + 1 < nz)
				store(2 * po + 1, o0);
			if (pe >= A && pe < B)
				store(2 * pe, o1);
		}
	}
}

template <bool INV, int CPT>
static void vol_z_nt(int nt, dim3 grid, int threads, hipStream_t s, const float *in, long in_sy, long in_sz, float *out,
	long out_sy, long out_sz, int nx, int ny, int nz, int tp, float *lll, long lll_sy, long lll_sz)
{
	switch ((nt < 0 ? 0 : nt) & 3) {
	case 0: k_vol_z<INV, CPT, 0><<<grid, threads, 0, s>>>(in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, lll, lll_sy, lll_sz); break;
	case 1: k_vol_z<INV, CPT, 1><<<grid, threads, 0, s>>>(in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, lll, lll_sy, lll_sz); break;
	case 2: k_vol_z<INV, CPT, 2><<<grid, threads, 0, s>>>(in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, lll, lll_sy, lll_sz); break;
	default: k_vol_z<INV, CPT, 3><<<grid, threads, 0, s>>>(in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, lll, lll_sy, lll_sz); break;
	}
}

hipError_t launch_vol_z(bool inverse, const float *in, long in_sy, long in_sz, float *out, long out_sy, long out_sz,
	int nx, int ny, int nz, const VolTuning &vt, hipStream_t s, float *lll, long lll_sy, long lll_sz)
{
	if (nx < 1 || ny < 1 || nz < 2 || ny > 65535 || (lll && (inverse || lll_sy % 2 || lll_sz % 2 || ((uintptr_t)lll & 7))))
		return hipErrorInvalidValue;
	const int Zd = (nz + 1) / 2;
	const int cpt = (vt.cpt == 8 && nx >= 512) ? 8 : 4;
	const int ntx = (nx + 64 * cpt - 1) / (64 * cpt);
	// long z lines: split them so that at least ~2048 waves exist
	int tp = 64;
	while (tp > 8 && (long)ntx * ny * ((Zd + tp - 1) / tp) < 2048)
		tp >>= 1;
	if (vt.tile_pairs >= 4)
		tp = vt.tile_pairs;
	const int nzt = (Zd + tp - 1) / tp;
	if (nzt > 65535)
		return hipErrorInvalidValue;
	const int waves = ntx >= 4 ? 4 : ntx;
	dim3 grid((ntx + waves - 1) / waves, ny, nzt);
	if (inverse) {
		if (cpt == 8)
			vol_z_nt<true, 8>(vt.nt, grid, 64 * waves, s, in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, lll, lll_sy, lll_sz);
		else
			vol_z_nt<true, 4>(vt.nt, grid, 64 * waves, s, in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, lll, lll_sy, lll_sz);
	} else {
		if (cpt == 8)
			vol_z_nt<false, 8>(vt.nt, grid, 64 * waves, s, in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, lll, lll_sy, lll_sz);
		else
			vol_z_nt<false, 4>(vt.nt, grid, 64 * waves, s, in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, lll, lll_sy, lll_sz);
	}
	return hipGetLastError();
}

// ---- 3-D, one pass: x, y and z lifting of a level fused (forward, out of place) ----
// "Slab-tiled z pass": a workgroup (4 waves) owns 256 x 32 voxel columns and marches along
// z.  Per slice: each wave DMAs 10 of the tile's 40 input rows (32 + 4 halo rows each side,
// row and column reflection in the source address) into its own LDS ring, one slice ahead;
// lifts them horizontally in registers; parks the x-lifted rows in a workgroup-shared LDS
// slab; after a barrier reads the 16 rows around its 8 output rows back, lifts them
// vertically in registers; and feeds the 8 x 4 samples per lane into the streaming z
// recurrence whose state (4 partial slices x 32 columns) stays in registers for the whole
// march.  The intermediate volume of the two-pass path never exists: 8 B per voxel (+ 25 %
// halo rows, + z warm-up) instead of 16.  Same arithmetic and operand order as
// k_fwd_sweep / k_vol_z, hence the same bits.
// RW: output rows per wave (the tile has 4 RW rows).  8 rows hold 128 + 32 + 32 + 60 live values
// per lane in the vertical / z phase: 304 registers (48 of them accumulator registers), ONE wave
// per SIMD, one workgroup per CU.  6 rows fit 256 registers and 64 KiB of LDS -- two workgroups per
// CU -- but lift 31/24 instead of 39/32 input rows per output row: measured slower (1024^3: 2.43
// against 2.12-2.21 ms), because the level is bound by VALU issue (a wave64 fp32 instruction holds
// the SIMD for 4 cycles; 557 M of them per launch = 0.95 ms per SIMD), not by latency.  Built,
// measured and taken out again in round 2: the z state's deeper planes in LDS and the vertical
// window in two halves (no accumulator-register moves, 249 registers): 2.24-2.28 ms, no gain.
// MODE (multi-level calls): 0 = dense result; 1 = a level >= 1 storing into its stride-2^j lattice
// of the destination; 2 and 3 = levels 0 and 1 of a call whose rows with even y and even z are
// written ONCE: level 0 (2) withholds them and parks their odd-x samples in `side`, level 1 (3)
// writes them whole, its own samples interleaved with the parked ones (which it brings in by
// LDS-DMA one iteration ahead: no registers, no exposed latency); 4 = 0 with every row addressed
// as a buffer (per-dword hardware bounds check, 4-byte alignment suffices): without the
// column-by-column staging and the bounded stores of the general kernel the vertical / z phase
// needs no accumulator registers; 2 and 3 are built the same way.  0 remains as a cross-check.
template <int NT, int RW, int MODE>
__global__ __launch_bounds__(256, RW == 8 ? 1 : 2) void k_vol_fwd_fused(VolFusedArgs a, int tile_pairs_z, int vec_ok, int ntx, int nty, int swz)
{
	using W = Cdf97S;
	// the RW output rows of a wave need x-lifted rows -4 .. RW+2 around its first row: a tile of
	// 32 rows needs 39 input rows; wave w stages rows w, w+4, ... (10, 10, 10, 9 of them)
	constexpr int K = 4, CPT = 4, TW = 256, RS = TW + 8, TY = 4 * RW, NR = TY + 2 * K - 1, RPW = (NR + 3) / 4;
	constexpr int NV = RW + 2 * K - 1; // slab rows a wave's vertical lift reads
	constexpr int kLdAux = (NT & 2) ? 2 : 0;
	// bit 2: the halo columns (lines the x-neighbour tile streams as its own) with the default
	// cache policy, so that whichever of the two comes second can hit in L2
	constexpr int kHaloAux = (NT & 4) ? 0 : kLdAux;
	constexpr bool kNtStore = (NT & 1) != 0;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// workgroup -> tile: x tiles fastest, then y tiles, then z tiles; with the XCD swizzle an
	// XCD (workgroups id % 8) owns a contiguous run of tiles, so the halo rows and columns that
	// neighbouring tiles share are hits in that XCD's L2
	const int bid = tile_block_id(swz);
	const int c0 = (bid % ntx) * TW, c = c0 + lane * CPT;
	const int y0 = ((bid / ntx) % nty) * TY;
	const int Zd = (a.nz + 1) >> 1;
	const int A = (bid / (ntx * nty)) * tile_pairs_z;
	if (A >= Zd)
		return; // the whole workgroup leaves together
	const int B = min(A + tile_pairs_z, Zd);
	const int n_iter = (B - A) + K, q0 = A - K / 2;
	const int n_slices = 2 * n_iter;

	// LDS: [wave-private staging rows, one slice: NR x RS floats] [shared slab: NR x TW floats];
	// 81 KiB, so that two workgroups share a CU and one computes while the other waits
	char *ring = smem + (size_t)wv * RPW * RS * 4;
	char *slab = smem + (size_t)NR * RS * 4;
	const unsigned ring_off = lds_offset(ring), slab_off = lds_offset(slab);
	// MODE 3: the parked rows of this wave's 2 RW output rows, 1 KiB each
	char *parked = slab + (size_t)NR * TW * 4 + (size_t)wv * 2 * RW * TW * 4;
	const unsigned parked_off = lds_offset(parked);
	// tiles that overhang the volume (or unaligned volumes) are staged column by column
	const bool full = MODE >= 1 || (vec_ok && c0 + TW <= a.nx);
	// interior tiles fetch each 4-column halo as ONE aligned 16 B piece (two lanes) instead of four
	// 4 B ones; tiles at the volume's x borders reflect column by column
	const bool halo16 = full && c0 >= 4 && c0 + TW + 4 <= a.nx;
	// (the reflected source columns of overhanging and border tiles are recomputed where they are
	// used: kept in registers across the march they cost 40 accumulator registers)

	auto issue = [&](int t) {
		const float *sl = a.in + (long)reflect(2 * q0 - 1 + t, a.nz) * a.in_sz;
#pragma unroll
		for (int i = 0; i < RPW; i++) {
			if (wv + 4 * i < NR) {
				const int r = reflect(y0 - K + wv + 4 * i, a.ny);
				const float *grow = sl + (long)r * a.in_sy;
				char *lrow = ring + (size_t)i * RS * 4;
				if (full) {
					// the DMA places lane i's 16 B at lrow + 16 i; in a tile that overhangs the volume
					// (MODE >= 1) the dwords beyond the row's end are zero-filled by the bounds check and the four
					// reflected columns next to the edge -- all a valid output can reach -- come one by one
					if constexpr (MODE >= 1)
						dma16_row<kLdAux>(row_rsrc(grow, (unsigned)a.nx * 4), (unsigned)c * 4, lrow);
					else
						dma16<kLdAux>(grow + c, lrow);
					if (MODE >= 1 && lane < min(4, c0 + TW - a.nx))
						dma4<kLdAux>(grow + reflect(a.nx + lane, a.nx), lrow + (a.nx - c0) * 4);
				} else {
#pragma unroll
					for (int e = 0; e < CPT; e++)
						dma4<kLdAux>(grow + reflect(c0 + e * 64 + lane, a.nx), lrow + e * 256);
				}
				if (halo16) {
					if (lane < 2)
						dma16<kHaloAux>(grow + (lane == 0 ? c0 - 4 : c0 + TW), lrow + TW * 4);
				} else if (lane < 8) {
					dma4<kHaloAux>(grow + reflect(lane < 4 ? c0 - 4 + lane : c0 + TW + (lane & 3), a.nx), lrow + TW * 4);
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
		// horizontal lift of this wave's rows, parked in the shared slab.  TWO rows at a time as the
		// two halves of packed fp32 operations (v_pk_add_f32 / v_pk_mul_f32: the level is bound by
		// VALU issue, and a row pair shares every instruction of the lift), software-pipelined: the
		// LDS reads of the next pair are in flight while this pair is lifted (counted lgkmcnt: LDS
		// operations of a wave complete in order).
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
				f2 x[CPT + 2 * K];
#pragma unroll
				for (int e = 0; e < K; e++) {
					x[e] = f2{from_bits<float>(L[b][0][e]), from_bits<float>(L[b][1][e])};
					x[K + e] = f2{from_bits<float>(O[b][0][e]), from_bits<float>(O[b][1][e])};
					x[K + CPT + e] = f2{from_bits<float>(R[b][0][e]), from_bits<float>(R[b][1][e])};
				}
				// lift_fwd_regs on both rows at once: x[j] += c_s * (x[j-1] + x[j+1]), product and sums rounded separately
#pragma unroll
				for (int st_ = 0; st_ < K; st_++)
#pragma unroll
					for (int j = st_ + 1; j <= CPT + 2 * K - 2 - st_; j += 2)
						x[j] = x[j] + W::fc(st_) * (x[j - 1] + x[j + 1]);
				const f2 e0 = x[K] * W::zeta(), o0 = x[K + 1] * (1.0f / W::zeta()), e1 = x[K + 2] * W::zeta(), o1 = x[K + 3] * (1.0f / W::zeta());
#pragma unroll
				for (int q = 0; q < 2; q++) {
					const int i = 2 * ip + q;
					if (wv + 4 * i < NR) // (the last wave stages one row fewer: its read of that slot is harmless)
						lds_write4(slab_off + (unsigned)(wv + 4 * i) * TW * 4 + lane * 16,
							u4{to_bits(e0[q]), to_bits(o0[q]), to_bits(e1[q]), to_bits(o1[q])});
				}
			}
		}
		// the staging rows are consumed: the next slice's DMA flies during the rest of the iteration
		if (t + 1 < n_slices)
			issue(t + 1);
		if constexpr (MODE == 3) {
			// the next iteration stores slices 2k, 2k+1: fetch the parked halves of those rows
			const int it1 = (t + 1) >> 1, k1 = A + it1 - K;
			if (!(t & 1) && it1 >= K) {
#pragma unroll
				for (int r = 0; r < RW; r++) {
					const int y = y0 + RW * wv + r;
					if (y < a.ny) {
						const float *sp = a.side + (long)(2 * k1) * a.side_sz + (long)y * a.side_sy;
						dma16_row<0>(row_rsrc(sp, (unsigned)a.nx * 4), (unsigned)c * 4, parked + (size_t)(2 * r) * TW * 4);
						if (2 * k1 + 1 < a.nz)
							dma16_row<0>(row_rsrc(sp + a.side_sz, (unsigned)a.nx * 4), (unsigned)c * 4, parked + (size_t)(2 * r + 1) * TW * 4);
					}
				}
			}
		}
		wg_barrier_lds(); // the slab is complete

		// vertical lift: slab rows RW wv .. RW wv + NV - 1 give this wave's RW output rows
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
			float col[NV];
#pragma unroll
			for (int j = 0; j < NV; j++)
				col[j] = from_bits<float>(v[j][e]);
			lift_fwd_regs<W, NV>(col);
#pragma unroll
			for (int r = 0; r < RW; r++)
				cur[r][e] = W::fwd_scale(r & 1, col[K + r]);
		}

		// z: slices arrive as (2q-1, 2q); the odd one waits in registers for its partner
		if (!(t & 1)) {
#pragma unroll
			for (int r = 0; r < RW; r++)
#pragma unroll
				for (int e = 0; e < CPT; e++)
					ra[r][e] = cur[r][e];
			continue;
		}
		const int it = t >> 1;
		const int k = A + it - K;
#pragma unroll
		for (int r = 0; r < RW; r++) {
			float o0[CPT], o1[CPT];
#pragma unroll
			for (int e = 0; e < CPT; e++) {
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
			}
			const int y = y0 + RW * wv + r;
			if (it >= K && y < a.ny) {
				float *p = a.out + (long)(2 * k) * a.out_sz + (long)y * a.out_sy + (MODE == 1 ? c * a.out_sx : MODE == 3 ? 2L * c : (long)c);
				const bool hz = 2 * k + 1 < a.nz;
				float *pl = a.lll && !(r & 1) ? a.lll + (long)k * a.lll_sz + (long)(y >> 1) * a.lll_sy + (c >> 1) : nullptr;
				// MODE >= 1: rows as buffers -- the lanes beyond the edge of a tile that overhangs the
				// volume are dropped by the hardware's bounds check, no per-lane branches
				[[maybe_unused]] const float *row0 = a.out + (long)(2 * k) * a.out_sz + (long)y * a.out_sy;
				[[maybe_unused]] const bool to_lll = a.lll && !(r & 1);
				if constexpr (MODE == 3) {
					// own samples at even x, level 0's parked ones at odd x: a row of 2 nx samples
					u4 s0, s1;
					lds_read2(parked_off + (unsigned)(2 * r) * TW * 4 + lane * 16, parked_off + (unsigned)(2 * r + 1) * TW * 4 + lane * 16, s0, s1);
					const row_rsrc_t d0 = row_rsrc(row0, (unsigned)a.nx * 8);
					if (to_lll && a.temporal_shared) {
						// a further level follows: its merge pass reads this row (even y, even slice of this level) again
						// at once -- temporal, so that it can stay in the Infinity Cache
						store16_row<false>(d0, (unsigned)c * 8, u4{to_bits(o0[0]), s0[0], to_bits(o0[1]), s0[1]});
						store16_row<false>(d0, (unsigned)c * 8 + 16, u4{to_bits(o0[2]), s0[2], to_bits(o0[3]), s0[3]});
					} else {
						store16_row<kNtStore>(d0, (unsigned)c * 8, u4{to_bits(o0[0]), s0[0], to_bits(o0[1]), s0[1]});
						store16_row<kNtStore>(d0, (unsigned)c * 8 + 16, u4{to_bits(o0[2]), s0[2], to_bits(o0[3]), s0[3]});
					}
					if (hz) {
						const row_rsrc_t d1 = row_rsrc(row0 + a.out_sz, (unsigned)a.nx * 8);
						store16_row<kNtStore>(d1, (unsigned)c * 8, u4{to_bits(o1[0]), s1[0], to_bits(o1[1]), s1[1]});
						store16_row<kNtStore>(d1, (unsigned)c * 8 + 16, u4{to_bits(o1[2]), s1[2], to_bits(o1[3]), s1[3]});
					}
					if (to_lll)
						store8_row<false>(row_rsrc(a.lll + (long)k * a.lll_sz + (long)(y >> 1) * a.lll_sy, (unsigned)((a.nx + 1) >> 1) * 4), (unsigned)c * 2, u2{to_bits(o0[0]), to_bits(o0[2])});
				} else if constexpr (MODE == 2) {
					// rows with even y in the even slice are level 1's to write
					if (r & 1)
						store16_row<kNtStore>(row_rsrc(row0, (unsigned)a.nx * 4), (unsigned)c * 4, u4{to_bits(o0[0]), to_bits(o0[1]), to_bits(o0[2]), to_bits(o0[3])});
					else
						store8_row<kNtStore>(row_rsrc(a.side + (long)k * a.side_sz + (long)(y >> 1) * a.side_sy, (unsigned)(a.nx >> 1) * 4), (unsigned)c * 2, u2{to_bits(o0[1]), to_bits(o0[3])});
					if (hz)
						store16_row<kNtStore>(row_rsrc(row0 + a.out_sz, (unsigned)a.nx * 4), (unsigned)c * 4, u4{to_bits(o1[0]), to_bits(o1[1]), to_bits(o1[2]), to_bits(o1[3])});
					if (to_lll)
						store8_row<false>(row_rsrc(a.lll + (long)k * a.lll_sz + (long)(y >> 1) * a.lll_sy, (unsigned)((a.nx + 1) >> 1) * 4), (unsigned)c * 2, u2{to_bits(o0[0]), to_bits(o0[2])});
				} else if constexpr (MODE == 1) {
					// a level >= 1 writing into its lattice of the destination volume: sample x of this
					// level sits at x * out_sx of the destination row; one dword store per sample, the
					// row's last lattice sample bounds the buffer
					const unsigned nb = ((unsigned)(a.nx - 1) * (unsigned)a.out_sx + 1) * 4, sx4 = (unsigned)a.out_sx * 4;
					const row_rsrc_t d0 = row_rsrc(row0, nb), d1 = row_rsrc(row0 + a.out_sz, nb);
#pragma unroll
					for (int e = 0; e < CPT; e++) {
						__builtin_amdgcn_raw_buffer_store_b32(to_bits(o0[e]), d0, (unsigned)c * sx4, e * sx4, 0);
						if (hz)
							__builtin_amdgcn_raw_buffer_store_b32(to_bits(o1[e]), d1, (unsigned)c * sx4, e * sx4, 0);
					}
					if (to_lll)
						store8_row<false>(row_rsrc(a.lll + (long)k * a.lll_sz + (long)(y >> 1) * a.lll_sy, (unsigned)((a.nx + 1) >> 1) * 4), (unsigned)c * 2, u2{to_bits(o0[0]), to_bits(o0[2])});
				} else if (MODE == 4 || full) {
					if constexpr (MODE == 4) {
						store16_row<kNtStore>(row_rsrc(row0, (unsigned)a.nx * 4), (unsigned)c * 4, u4{to_bits(o0[0]), to_bits(o0[1]), to_bits(o0[2]), to_bits(o0[3])});
						if (hz)
							store16_row<kNtStore>(row_rsrc(row0 + a.out_sz, (unsigned)a.nx * 4), (unsigned)c * 4, u4{to_bits(o1[0]), to_bits(o1[1]), to_bits(o1[2]), to_bits(o1[3])});
						if (to_lll)
							store8_row<false>(row_rsrc(a.lll + (long)k * a.lll_sz + (long)(y >> 1) * a.lll_sy, (unsigned)((a.nx + 1) >> 1) * 4), (unsigned)c * 2, u2{to_bits(o0[0]), to_bits(o0[2])});
					} else {
						store_vec<kNtStore>((u4 *)p, u4{to_bits(o0[0]), to_bits(o0[1]), to_bits(o0[2]), to_bits(o0[3])});
						if (hz)
							store_vec<kNtStore>((u4 *)(p + a.out_sz), u4{to_bits(o1[0]), to_bits(o1[1]), to_bits(o1[2]), to_bits(o1[3])});
						if (pl)
							*(u2 *)pl = u2{to_bits(o0[0]), to_bits(o0[2])};
					}
				} else {
#pragma unroll
					for (int e = 0; e < CPT; e++)
						if (c + e < a.nx) {
							p[e] = o0[e];
							if (hz)
								p[a.out_sz + e] = o1[e];
							if (pl && !(e & 1))
								pl[e >> 1] = o0[e];
						}
				}
			}
		}
	}
}

bool vol_fused_applies(const VolFusedArgs &a)
{
	// A workgroup's march along z is a serial chain: the fused level pays off once the
	// volume has about one workgroup per CU at 32 slice pairs per march (512^3: 0.31 ms fused
	// against 0.43 in two passes; 256^3: 0.11 against 0.07).  Narrow volumes would leave most
	// of a 256-column tile idle.
	if (a.in == a.out || a.nx < 128 || a.ny < 2 || a.nz < 2)
		return false;
	const long tiles = (long)((a.nx + 255) / 256) * ((a.ny + 31) / 32);
	return tiles * (((a.nz + 1) / 2 + 31) / 32) >= 192;
}

static bool vol_fused_vec_ok(const VolFusedArgs &a)
{
	return aligned16(a.in) && a.in_sy % 4 == 0 && a.in_sz % 4 == 0 &&
		(a.mode == 1 || (aligned16(a.out) && a.out_sy % 4 == 0 && a.out_sz % 4 == 0)) &&
		(!a.lll || (((uintptr_t)a.lll & 7) == 0 && a.lll_sy % 2 == 0 && a.lll_sz % 2 == 0));
}

template <int NT, int RW, int MODE>
static hipError_t vol_fused_launch(const VolFusedArgs &a, int tp, int ntx, int nty, int nzt, int swz, hipStream_t s)
{
	constexpr int NR = 4 * RW + 7;
	const size_t lds = (size_t)NR * (256 + 8) * 4 + (size_t)NR * 256 * 4 + (MODE == 3 ? (size_t)4 * 2 * RW * 256 * 4 : 0);
	if (hipError_t e = allow_lds((const void *)k_vol_fwd_fused<NT, RW, MODE>, lds))
		return e;
	k_vol_fwd_fused<NT, RW, MODE><<<dim3(ntx * nty * nzt), 256, lds, s>>>(a, tp, vol_fused_vec_ok(a), ntx, nty, swz);
	return hipGetLastError();
}

hipError_t launch_vol_fwd_fused(const VolFusedArgs &a, const VolTuning &vt, hipStream_t s)
{
	if (a.in == a.out || a.nx < 2 || a.ny < 2 || a.nz < 2)
		return hipErrorInvalidValue;
	const int rw = vt.rows == 6 ? 6 : 8;
	const int Zd = (a.nz + 1) / 2;
	const int ntx = (a.nx + 255) / 256, nty = (a.ny + 4 * rw - 1) / (4 * rw);
	int mode = a.mode;
	if (mode < 0 || mode > 3)
		return hipErrorInvalidValue;
	if (mode == 0 && vt.whole)
		mode = 4;
	// Length of a workgroup's march along z.  The chip has `slots` places for workgroups (two per CU
	// for the lean variants, one for the merged level 1 with its parked rows and for the general
	// kernel); every march pays a 4-pair warm-up; a launch that fits the slots is one round, a
	// larger one runs N / slots rounds plus a ragged tail; and a workgroup that has its CU to itself
	// marches faster.  Pick the length with the least (rounds) x (pairs + 4).  Measured (one level,
	// ms): 512^3 16 pairs 0.25 / 32 pairs 0.29 / 64 pairs 0.48; 640^3 0.53 / 0.61; 768^3 64 pairs
	// 0.80 / 32 pairs 0.83 / 128 pairs 0.94; 896^3 32 pairs 1.39 / 128 pairs 1.45 / 64 pairs 1.50;
	// 1024^3 128 pairs 1.66 / 64 pairs 1.67 / 32 pairs 1.71 / 16 pairs 1.83.
	const double slots = (mode == 2 || mode == 4) ? 512 : 256;
	int tp = 128;
	{
		double best = -1;
		for (int cand = 128; cand >= 16; cand >>= 1) {
			const double n = (double)ntx * nty * ((Zd + cand - 1) / cand);
			const double rounds = n <= slots ? (n <= slots / 2 ? 0.6 : 1.0) : n / slots + 0.35;
			const double cost = rounds * (cand + 4);
			if (best < 0 || cost < best) {
				best = cost;
				tp = cand;
			}
		}
	}
	if (vt.tile_pairs >= 4)
		tp = vt.tile_pairs;
	const int nzt = (Zd + tp - 1) / tp;
	if ((long)ntx * nty * nzt > 0x7fffffffL)
		return hipErrorInvalidValue;
	const int swz = vt.swizzle;
	// cache policy (template NT: bit 0 = non-temporal stores, bit 1 = non-temporal loads, bit 2 = halo
	// columns exempt from bit 1).  Default 1: the stores stream past the caches, the loads do not --
	// the 7 halo rows and 8 halo columns of a tile are its neighbours' own rows and columns, read at
	// about the same time by workgroups of the same XCD, and hit in its L2 (1024^3: 1.65 ms against
	// 1.77 with non-temporal loads, 1.69 with only the halo columns exempt).  Every variant has 1 and
	// 3; the whole-tile one also 0, 2 and 7 for measurements.
	const int want = vt.nt < 0 ? 1 : (vt.nt & 7);
	const bool nt_loads = want == 3 || want == 2 || want == 7;
	if (mode >= 1 && mode <= 3 && rw != 8) // the multi-level store variants exist for the default row count
		return hipErrorInvalidValue;
	if ((mode == 2 || mode == 3) && !a.side)
		return hipErrorInvalidValue;
#define DWT_VOL_GO(NT_, RW_, MODE_) return vol_fused_launch<NT_, RW_, MODE_>(a, tp, ntx, nty, nzt, swz, s)
	if (mode == 4 && rw == 8) {
		switch (want) {
		case 0: DWT_VOL_GO(0, 8, 4);
		case 2: DWT_VOL_GO(2, 8, 4);
		case 3: DWT_VOL_GO(3, 8, 4);
		case 7: DWT_VOL_GO(7, 8, 4);
		default: DWT_VOL_GO(1, 8, 4);
		}
	}
	if (rw == 6) {
		if (mode == 4) {
			if (nt_loads) DWT_VOL_GO(3, 6, 4);
			DWT_VOL_GO(1, 6, 4);
		}
		if (nt_loads) DWT_VOL_GO(3, 6, 0);
		DWT_VOL_GO(1, 6, 0);
	}
	switch (mode * 2 + (nt_loads ? 1 : 0)) {
	case 0: DWT_VOL_GO(1, 8, 0);
	case 1: DWT_VOL_GO(3, 8, 0);
	case 2: DWT_VOL_GO(1, 8, 1);
	case 3: DWT_VOL_GO(3, 8, 1);
	case 4: DWT_VOL_GO(1, 8, 2);
	case 5: DWT_VOL_GO(3, 8, 2);
	case 6: DWT_VOL_GO(1, 8, 3);
	default: DWT_VOL_GO(3, 8, 3);
	}
#undef DWT_VOL_GO
}

__global__ __launch_bounds__(256) void k_lattice_copy(const float *__restrict__ src, long s_sx, long s_sy, long s_sz,
	float *__restrict__ dst, long d_sx, long d_sy, long d_sz, int nx, int ny, int nxb)
{
	// grid.x = column blocks x rows (rows can exceed the 65535 limit of grid.y), grid.y = slices
	const int x = (blockIdx.x % nxb) * blockDim.x + threadIdx.x;
	const int y = blockIdx.x / nxb, z = blockIdx.y;
	if (x < nx && y < ny)
		dst[(long)z * d_sz + (long)y * d_sy + (long)x * d_sx] = src[(long)z * s_sz + (long)y * s_sy + (long)x * s_sx];
}

// Scatter of a dense volume into a stride-2 or stride-4 lattice as an explicit read-modify-write of
// whole 16-byte pieces of the destination (a lattice store of 4 bytes is a masked write of a line
// that left the caches long ago: measured 181 us for 256^3 into a stride-4 lattice against the
// round trip of full pieces here).  One thread per piece: SX = 4 replaces its first sample, SX = 2
// its first and third.
template <int SX>
__global__ __launch_bounds__(256) void k_lattice_merge(const float *__restrict__ src, long s_sy, long s_sz, float *__restrict__ dst, long d_sy, long d_sz,
	int npieces, int nx, int ny, int nxb)
{
	constexpr int R = 8; // rows per thread: all loads first, then the stores
	const int i = (blockIdx.x % nxb) * blockDim.x + threadIdx.x; // piece of the destination row
	const int y0 = (blockIdx.x / nxb) * R, z = blockIdx.y;
	if (i >= npieces)
		return;
	u4 *p = (u4 *)(dst + (long)z * d_sz + (long)y0 * d_sy) + i;
	const float *q = src + (long)z * s_sz + (long)y0 * s_sy + i * (4 / SX);
	const bool second = SX == 2 && i * 2 + 1 < nx;
	u4 v[R];
	float a[R], b[R];
	// rows past the end load row y0 again (never stored): straight-line loads, eight in flight
#pragma unroll
	for (int r = 0; r < R; r++) {
		const long ro = y0 + r < ny ? r : 0;
		v[r] = p[ro * (d_sy / 4)];
		a[r] = q[ro * s_sy];
		b[r] = q[ro * s_sy + (second ? 1 : 0)];
	}
#pragma unroll
	for (int r = 0; r < R; r++)
		if (y0 + r < ny) {
			v[r][0] = to_bits(a[r]);
			if (second)
				v[r][2] = to_bits(b[r]);
			p[r * (d_sy / 4)] = v[r];
		}
}

// Gather of a stride-2 lattice into a dense volume (the pack of a multi-level inverse): one thread
// per four output samples -- two 16-byte loads of the source row (rows as buffers: the tail of an odd
// row is zero-filled), the even samples of them as one 16-byte store.
__global__ __launch_bounds__(256) void k_lattice_pack2(const float *__restrict__ src, long s_sy, long s_sz, float *__restrict__ dst, long d_sy, long d_sz,
	int nx, int src_nx, int ny, int nxb)
{
	const int x4 = ((blockIdx.x % nxb) * 256 + threadIdx.x) * 4; // first output sample of this thread
	const int y = blockIdx.x / nxb, z = blockIdx.y;
	if (x4 >= nx || y >= ny)
		return;
	const row_rsrc_t rs = row_rsrc(src + (long)z * s_sz + (long)y * s_sy, (unsigned)src_nx * 4);
	const u4 a = load16_row<true>(rs, (unsigned)x4 * 8), b = load16_row<true>(rs, (unsigned)x4 * 8 + 16);
	store16_row<false>(row_rsrc(dst + (long)z * d_sz + (long)y * d_sy, (unsigned)nx * 4), (unsigned)x4 * 4, u4{a[0], a[2], b[0], b[2]});
}

hipError_t launch_lattice_copy(const float *src, long s_sx, long s_sy, long s_sz, float *dst, long d_sx, long d_sy, long d_sz,
	int nx, int ny, int nz, hipStream_t s)
{
	const int nxb = (nx + 255) / 256;
	if (nx < 1 || ny < 1 || nz < 1 || nz > 65535 || (long)nxb * ny > 0x7fffffffL)
		return hipErrorInvalidValue;
	if (s_sx == 2 && d_sx == 1 && nx >= 64) {
		// the source row holds the samples 0, 2, ..., 2 (nx - 1): 2 nx - 1 of them are addressable for sure
		const int nxb4 = ((nx + 3) / 4 + 255) / 256;
		k_lattice_pack2<<<dim3(nxb4 * ny, nz), 256, 0, s>>>(src, s_sy, s_sz, dst, d_sy, d_sz, nx, 2 * nx - 1, ny, nxb4);
		return hipGetLastError();
	}
	dim3 grid(nxb * ny, nz);
	k_lattice_copy<<<grid, 256, 0, s>>>(src, s_sx, s_sy, s_sz, dst, d_sx, d_sy, d_sz, nx, ny, nxb);
	return hipGetLastError();
}

// Dense volume -> lattice of a destination whose rows hold dst_nx samples: whole-piece merge where
// every piece lies inside its row, the sample-wise copy elsewhere.
hipError_t launch_lattice_scatter(const float *src, long s_sy, long s_sz, float *dst, long d_sx, long d_sy, long d_sz,
	int nx, int ny, int nz, int dst_nx, hipStream_t s)
{
	if (nx < 1 || ny < 1 || nz < 1 || nz > 65535)
		return hipErrorInvalidValue;
	const int npieces = d_sx == 4 ? nx : (nx + 1) / 2;
	if ((d_sx == 2 || d_sx == 4) && aligned16(dst) && d_sy % 4 == 0 && d_sz % 4 == 0 && (long)npieces * 4 <= dst_nx) {
		const int nxb = (npieces + 255) / 256;
		if ((long)nxb * ny > 0x7fffffffL)
			return hipErrorInvalidValue;
		dim3 grid(nxb * ((ny + 7) / 8), nz);
		if (d_sx == 4)
			k_lattice_merge<4><<<grid, 256, 0, s>>>(src, s_sy, s_sz, dst, d_sy, d_sz, npieces, nx, ny, nxb);
		else
			k_lattice_merge<2><<<grid, 256, 0, s>>>(src, s_sy, s_sz, dst, d_sy, d_sz, npieces, nx, ny, nxb);
		return hipGetLastError();
	}
	return launch_lattice_copy(src, 1, s_sy, s_sz, dst, d_sx, d_sy, d_sz, nx, ny, nz, s);
}

} // namespace dwt
