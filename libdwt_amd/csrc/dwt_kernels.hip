// dwt_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the lifting DWT.
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
#include "dwt_kernels.h"
#include "dwt_lift.h"

#include <stdint.h>
#include <type_traits>

namespace dwt {

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <class T> static __device__ __forceinline__ T from_bits(unsigned u) { return __builtin_bit_cast(T, u); }
template <class T> static __device__ __forceinline__ unsigned to_bits(T v) { return __builtin_bit_cast(unsigned, v); }

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
		lift_fwd_regs<W, 2 * K + 1>(w);
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
		lift_inv_regs<W, 2 * K + 1>(w);
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
	case kCdf97SFma: break; // the contracted variant exists for the fused sweeps only
	}
	return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------
// 2. fused tile sweeps
// ---------------------------------------------------------------------------------
#define DWT_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// AUX selects the cache policy of the LDS-DMA: 0 = default, 2 = non-temporal (the
// image is read once per level; nt keeps it from displacing reusable lines).
template <int AUX = 0>
static __device__ __forceinline__ void dma16(const void *g, void *l)
{
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
		(__attribute__((address_space(3))) void *)l, 16, 0, AUX);
}

template <int AUX = 0>
static __device__ __forceinline__ void dma4(const void *g, void *l)
{
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
		(__attribute__((address_space(3))) void *)l, 4, 0, AUX);
}

template <bool NT, class V>
static __device__ __forceinline__ void store_vec(V *p, V v)
{
	if constexpr (NT)
		__builtin_nontemporal_store(v, p);
	else
		*p = v;
}

// LDS reads go through inline asm: hipcc (ROCm 7.2) otherwise drains every
// outstanding LDS-DMA with vmcnt(0) before any ds_read, which would serialise the
// prefetch ring.  The wait for the data is inside the statement, so the outputs
// cannot be consumed early.
static __device__ __forceinline__ void lds_read3(unsigned a0, unsigned a1, unsigned a2, u4 &r0, u4 &r1, u4 &r2)
{
	asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
		: "=&v"(r0), "=&v"(r1), "=&v"(r2)
		: "v"(a0), "v"(a1), "v"(a2)
		: "memory");
}

static __device__ __forceinline__ void lds_read4(unsigned a0, unsigned a1, unsigned a2, u4 &r0, u4 &r1, u4 &r2, u4 &r3)
{
	asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %5 offset:16\n\tds_read_b128 %3, %6\n\ts_waitcnt lgkmcnt(0)"
		: "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
		: "v"(a0), "v"(a1), "v"(a2)
		: "memory");
}

static __device__ __forceinline__ void lds_read2x3(unsigned a0, unsigned a1, unsigned a2, u2 &r0, u2 &r1, u2 &r2)
{
	asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %4\n\tds_read_b64 %2, %5\n\ts_waitcnt lgkmcnt(0)"
		: "=&v"(r0), "=&v"(r1), "=&v"(r2)
		: "v"(a0), "v"(a1), "v"(a2)
		: "memory");
}

static __device__ __forceinline__ void lds_read1(unsigned a0, u4 &r0)
{
	asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0) : "v"(a0) : "memory");
}

static __device__ __forceinline__ void lds_read2o(unsigned a0, unsigned a1, u4 &r0, u4 &r1, u4 &r2)
{
	asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %4\n\ts_waitcnt lgkmcnt(0)"
		: "=&v"(r0), "=&v"(r1), "=&v"(r2)
		: "v"(a0), "v"(a1)
		: "memory");
}

static __device__ __forceinline__ void lds_read2(unsigned a0, unsigned a1, u4 &r0, u4 &r1)
{
	asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
		: "=&v"(r0), "=&v"(r1)
		: "v"(a0), "v"(a1)
		: "memory");
}

// Wavefront shifts by one lane (DPP, no LDS traffic): lane t receives lane t-1 / t+1;
// the wave's first / last lane keeps its own value (replaced by the caller).
static __device__ __forceinline__ unsigned from_left_lane(unsigned v)
{
	return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

static __device__ __forceinline__ unsigned from_right_lane(unsigned v)
{
	return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

static __device__ __forceinline__ unsigned lds_offset(const void *p)
{
	return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const void *)p);
}


// Workgroup -> tile mapping shared by both sweeps.  Each XCD has its own L2 and
// workgroups are dealt round-robin over the 8 XCDs, so with `swz` consecutive
// tiles are handed to the same XCD (neighbouring tiles share halo lines).
static __device__ __forceinline__ int tile_block_id(int swz)
{
	int b = blockIdx.x;
	const int nb = gridDim.x;
	if (swz && (nb & 7) == 0)
		b = (b & 7) * (nb >> 3) + (b >> 3);
	return b;
}

struct SweepGeom {
	int tile_pairs, ntx, swz, in_vec_ok, out_vec_ok;
	int ll_vec_ok = 0; // interleaved layout with a dense LL copy: that copy takes 8 B stores
	int wave_horiz; // 1: the waves of a workgroup take horizontally adjacent tiles
};

// ---- forward -------------------------------------------------------------------
// IL: write the result INTERLEAVED in place of the Mallat de-interleave (row 2k = L
// row, row 2k+1 = H row, columns interleaved alike) to `out_h`: the layout of the
// 3-D path (src/volume-dwt.c:677-725), where a batch is the slices of a volume.
template <class W, int CPT, int RING, int NT, bool IL = false>
__global__ __launch_bounds__(256) void k_fwd_sweep(FwdLevelArgs a, SweepGeom g)
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
	const int bid = tile_block_id(g.swz);
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
	const int B = min(A + g.tile_pairs, Hd);
	const int c0 = tx * TW;
	const int n_iter = (B - A) + K;
	const int q0 = A - K / 2;

	const T *in = (const T *)a.in + (long)img * a.in_bstride;
	T *out_ll = (T *)a.out_ll + (long)img * a.ll_bstride;
	T *out_h = (T *)a.out_h + (long)img * a.h_bstride;

	char *ring = smem + (size_t)wv * kRing * RS * 4;
	const unsigned ring_off = lds_offset(ring);

	const bool full = (c0 + TW <= a.W);
	const bool main16 = full && g.in_vec_ok;
	// source columns for the element-wise loader (partial / unaligned tiles)
	int colmap[CPT];
#pragma unroll
	for (int i = 0; i < CPT; i++)
		colmap[i] = reflect(c0 + i * 64 + lane, a.W);
	const int halo_col = reflect(lane < 4 ? c0 - 4 + lane : c0 + TW + (lane & 3), a.W);

	int islot = 0, rslot = 0; // ring slots of the next rows to fill / to consume
	const bool tall = a.H >= 64; // then a row index leaves [0,H) by less than H: one bounce
	auto issue = [&](int it) {
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const int ri = 2 * (q0 + it) - 1 + rr;
			const int r = tall ? reflect1(ri, a.H) : reflect(ri, a.H);
			char *lrow = ring + (size_t)(islot + rr) * RS * 4;
			const T *grow = in + (long)r * a.in_pitch;
			if (main16) {
#pragma unroll
				for (int i = 0; i < CPT / 4; i++)
					dma16<kLdAux>(grow + c0 + i * 256 + lane * 4, lrow + i * 1024);
			} else {
#pragma unroll
				for (int i = 0; i < CPT; i++)
					dma4<kLdAux>(grow + colmap[i], lrow + i * 256);
			}
			if (lane < 8)
				dma4<kLdAux>(grow + halo_col, lrow + TW * 4);
		}
		islot = islot + 2 >= kRing ? 0 : islot + 2;
	};

	T st[K][CPT];
#pragma unroll
	for (int s = 0; s < K; s++)
#pragma unroll
		for (int v = 0; v < CPT; v++)
			st[s][v] = 0;

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
			lift_fwd_regs<W, NARR>(x);
#pragma unroll
			for (int v = 0; v < CPT; v++)
				row[rr][v] = W::fwd_scale(v & 1, x[K + v]);
		}

		rslot = rslot + 2 >= kRing ? 0 : rslot + 2;

		// vertical pass: streaming lifting, state in registers
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
				const T d1n = W::fwd_step(0, ov, st[0][v], ev);
				const T s1n = W::fwd_step(1, st[0][v], st[1][v], d1n);
				lo[v] = W::fwd_scale(0, s1n);
				hi[v] = W::fwd_scale(1, d1n);
				st[0][v] = ev;
				st[1][v] = d1n;
			}
		}

		if constexpr (IL) {
			if (it >= K) {
				const int k = A + it - K;
				const int c = c0 + lane * CPT;
				T *r0 = out_h + (long)(2 * k) * a.h_pitch + c;
				T *r1 = r0 + a.h_pitch;
				const bool hrow = 2 * k + 1 < a.H;
				if (full && g.out_vec_ok) {
#pragma unroll
					for (int e = 0; e < CPT; e += 4) {
						store_vec<kNtStore>((u4 *)(r0 + e), u4{to_bits(lo[e]), to_bits(lo[e + 1]), to_bits(lo[e + 2]), to_bits(lo[e + 3])});
						if (hrow)
							store_vec<kNtStore>((u4 *)(r1 + e), u4{to_bits(hi[e]), to_bits(hi[e + 1]), to_bits(hi[e + 2]), to_bits(hi[e + 3])});
					}
				} else {
#pragma unroll
					for (int e = 0; e < CPT; e++)
						if (c + e < a.W) {
							r0[e] = lo[e];
							if (hrow)
								r1[e] = hi[e];
						}
				}
				// multi-level: the next level's input (even row, even column) also goes
				// out densely, so that no level has to gather a strided lattice
				if (a.il_ll) {
					T *ll = out_ll + (long)k * a.ll_pitch + (c >> 1);
					if (full && g.ll_vec_ok) {
#pragma unroll
						for (int e = 0; e < CPT; e += 4)
							store_vec<false>((u2 *)(ll + (e >> 1)), u2{to_bits(lo[e]), to_bits(lo[e + 2])});
					} else {
#pragma unroll
						for (int e = 0; e < CPT; e += 2)
							if (c + e < a.W)
								ll[e >> 1] = lo[e];
					}
				}
			}
		} else
		if (it >= K) {
			const int k = A + it - K;
			const int cl = (c0 + lane * CPT) >> 1;
			T *ll = out_ll + (long)k * a.ll_pitch + cl;
			T *hl = out_h + (long)k * a.h_pitch + Wd + cl;
			T *lh = out_h + (long)(Hd + k) * a.h_pitch + cl;
			T *hh = lh + Wd;
			const bool hrow = k < (a.H >> 1);
			if (full && g.out_vec_ok) {
				if constexpr (CPT == 8) {
					store_vec<kNtStoreLL>((u4 *)ll, u4{to_bits(lo[0]), to_bits(lo[2]), to_bits(lo[4]), to_bits(lo[6])});
					store_vec<kNtStore>((u4 *)hl, u4{to_bits(lo[1]), to_bits(lo[3]), to_bits(lo[5]), to_bits(lo[7])});
					if (hrow) {
						store_vec<kNtStore>((u4 *)lh, u4{to_bits(hi[0]), to_bits(hi[2]), to_bits(hi[4]), to_bits(hi[6])});
						store_vec<kNtStore>((u4 *)hh, u4{to_bits(hi[1]), to_bits(hi[3]), to_bits(hi[5]), to_bits(hi[7])});
					}
				} else {
					store_vec<kNtStoreLL>((u2 *)ll, u2{to_bits(lo[0]), to_bits(lo[2])});
					store_vec<kNtStore>((u2 *)hl, u2{to_bits(lo[1]), to_bits(lo[3])});
					if (hrow) {
						store_vec<kNtStore>((u2 *)lh, u2{to_bits(hi[0]), to_bits(hi[2])});
						store_vec<kNtStore>((u2 *)hh, u2{to_bits(hi[1]), to_bits(hi[3])});
					}
				}
			} else {
				const int nl = Wd, nh = a.W >> 1;
#pragma unroll
				for (int v = 0; v < CPT; v += 2) {
					const int ci = cl + (v >> 1);
					if (ci < nl) {
						ll[v >> 1] = lo[v];
						if (hrow)
							lh[v >> 1] = hi[v];
					}
					if (ci < nh) {
						hl[v >> 1] = lo[v + 1];
						if (hrow)
							hh[v >> 1] = hi[v + 1];
					}
				}
			}
		}
	}
}

// ---- inverse -------------------------------------------------------------------
// Source rows are Mallat rows: "L row p" = [LL | HL] and "H row p" = [LH | HH].
// LDS row slot (floats): [L main M | H main M | L halo 8 | H halo 8], M = TW/2;
// a halo block is [4 columns left of the tile | 4 columns right of the tile].
// IL: the input is INTERLEAVED (3-D path layout) at `in_h` instead of Mallat subbands.
template <class W, int CPT, int RING, int NT, bool IL = false>
__global__ __launch_bounds__(256) void k_inv_sweep(InvLevelArgs a, SweepGeom g)
{
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
	constexpr int HC = CPT / 2;          // subband columns per lane
	constexpr int RS = 2 * M + 16;
	constexpr int NARR = CPT + 2 * K - 1; // interleaved samples c-K+1 .. c+CPT+K-1
	constexpr int kDmaMain = IL ? CPT / 4 : (CPT == 8 ? 2 : 1);
	constexpr int kDmaPerIter = 2 * (kDmaMain + 1);
	extern __shared__ __attribute__((aligned(16))) char smem[];

	const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
	// wave-uniform on purpose: tile geometry, row indices and row pointers then live in SGPRs
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int bid = tile_block_id(g.swz);
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

	const bool full = (c0 + TW <= a.W);
	const bool main16 = full && g.in_vec_ok;
	// element-wise loader: element e of a subband segment <-> subband column cl0+e,
	// reflected through the interleaved index so that parity is preserved
	int colmapL[CPT / 2], colmapH[CPT / 2];
#pragma unroll
	for (int i = 0; i < CPT / 2; i++) {
		const int e = cl0 + i * 64 + lane;
		colmapL[i] = reflect(2 * e, a.W) >> 1;
		colmapH[i] = reflect(2 * e + 1, a.W) >> 1;
	}
	// halo lanes 0..7 -> L halo, 8..15 -> H halo
	const int hsub = (lane & 7) < 4 ? cl0 - 4 + (lane & 7) : cl0 + M + (lane & 3);
	const int halo_col = reflect(2 * hsub + ((lane >> 3) & 1), a.W) >> 1;
	const bool halo_is_h = (lane >> 3) & 1;

	// pointers to the four subbands' row starts are formed per source row
	// interleaved input: source columns of the element-wise loader and of the halo
	int colmapI[IL ? CPT : 1];
	int halo_colI = 0;
	if constexpr (IL) {
#pragma unroll
		for (int i = 0; i < CPT; i++)
			colmapI[i] = reflect(c0 + i * 64 + lane, a.W);
		halo_colI = reflect(lane < 4 ? c0 - 4 + lane : c0 + TW + (lane & 3), a.W);
	}
	auto issue = [&](int it) {
		const int p = p0 + it;
		if constexpr (IL) {
#pragma unroll
			for (int rr = 0; rr < 2; rr++) {
				const int r = reflect(2 * p + rr, a.H);
				char *lrow = ring + (size_t)((2 * it + rr) & (kRing - 1)) * RS * 4;
				// even rows come from in_ll, odd rows from in_h (reflection keeps the parity): the
				// two may be different buffers (multi-level inverse: composed even rows)
				const T *grow = (r & 1) ? in_h + (long)(r >> 1) * a.h_pitch : in_ll + (long)(r >> 1) * a.ll_pitch;
				if (main16) {
#pragma unroll
					for (int i = 0; i < CPT / 4; i++)
						dma16<kLdAux>(grow + c0 + i * 256 + lane * 4, lrow + i * 1024);
				} else {
#pragma unroll
					for (int i = 0; i < CPT; i++)
						dma4<kLdAux>(grow + colmapI[i], lrow + i * 256);
				}
				if (lane < 8)
					dma4<kLdAux>(grow + halo_colI, lrow + TW * 4);
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
			if (main16) {
				if constexpr (CPT == 8) {
					dma16<kLdAux>(gl + cl0 + lane * 4, lrow);
					dma16<kLdAux>(gh + cl0 + lane * 4, lrow + M * 4);
				} else {
					// lanes 0..31 fetch the L segment, 32..63 the H segment
					const T *gsel = lane < 32 ? gl : gh;
					dma16<kLdAux>(gsel + cl0 + (lane & 31) * 4, lrow);
				}
			} else {
#pragma unroll
				for (int i = 0; i < CPT / 2; i++)
					dma4<kLdAux>(gl + colmapL[i], lrow + i * 256);
#pragma unroll
				for (int i = 0; i < CPT / 2; i++)
					dma4<kLdAux>(gh + colmapH[i], lrow + M * 4 + i * 256);
			}
			if (lane < 16)
				dma4<kLdAux>((halo_is_h ? gh : gl) + halo_col, lrow + 2 * M * 4);
		}
	};

	// vertical state: NV columns per lane (CPT when rows are undone first, NARR
	// when columns are undone first and the horizontal halo must be carried)
	constexpr int NV = W::kInvColsFirst ? NARR : CPT;
	T st[K][NV];
#pragma unroll
	for (int s = 0; s < K; s++)
#pragma unroll
		for (int v = 0; v < NV; v++)
			st[s][v] = 0;

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

		// gather the interleaved samples c-K+1 .. c+CPT+K-1 of both source rows
		T x[2][NARR];
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const unsigned base = ring_off + (unsigned)((2 * it + rr) & (kRing - 1)) * RS * 4;
			if constexpr (IL) {
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
					x[rr][j] = W::inv_scale(rel & 1, v);
				}
				continue;
			}
			const unsigned hbase = base + 2 * M * 4;
			// subband values L[cl-2 .. cl+HC+2), H[cl-2 .. cl+HC+2) as l[], h[]
			T l[HC + 4], h[HC + 4];
			if constexpr (CPT == 8) {
				const unsigned ownL = base + lane * 16, ownH = base + M * 4 + lane * 16;
				const unsigned laL = lane == 0 ? hbase : ownL - 16, raL = lane == 63 ? hbase + 16 : ownL + 16;
				const unsigned laH = lane == 0 ? hbase + 32 : ownH - 16, raH = lane == 63 ? hbase + 48 : ownH + 16;
				u4 a0, a1, a2, b0, b1, b2;
				lds_read3(laL, ownL, raL, a0, a1, a2);
				lds_read3(laH, ownH, raH, b0, b1, b2);
				l[0] = from_bits<T>(a0[2]); l[1] = from_bits<T>(a0[3]);
				h[0] = from_bits<T>(b0[2]); h[1] = from_bits<T>(b0[3]);
#pragma unroll
				for (int e = 0; e < 4; e++) {
					l[2 + e] = from_bits<T>(a1[e]);
					h[2 + e] = from_bits<T>(b1[e]);
				}
				l[6] = from_bits<T>(a2[0]); l[7] = from_bits<T>(a2[1]);
				h[6] = from_bits<T>(b2[0]); h[7] = from_bits<T>(b2[1]);
			} else {
				const unsigned ownL = base + lane * 8, ownH = base + M * 4 + lane * 8;
				const unsigned laL = lane == 0 ? hbase + 8 : ownL - 8, raL = lane == 63 ? hbase + 16 : ownL + 8;
				const unsigned laH = lane == 0 ? hbase + 40 : ownH - 8, raH = lane == 63 ? hbase + 48 : ownH + 8;
				u2 a0, a1, a2, b0, b1, b2;
				lds_read2x3(laL, ownL, raL, a0, a1, a2);
				lds_read2x3(laH, ownH, raH, b0, b1, b2);
#pragma unroll
				for (int e = 0; e < 2; e++) {
					l[e] = from_bits<T>(a0[e]); l[2 + e] = from_bits<T>(a1[e]); l[4 + e] = from_bits<T>(a2[e]);
					h[e] = from_bits<T>(b0[e]); h[2 + e] = from_bits<T>(b1[e]); h[4 + e] = from_bits<T>(b2[e]);
				}
			}
			// x[j] <-> interleaved sample c-K+1+j (x[0] odd).  Sample i: even -> L[i/2],
			// odd -> H[i/2]; relative to cl: L index (i-c)/2 -> l[2 + ...].
#pragma unroll
			for (int j = 0; j < NARR; j++) {
				const int rel = j - K + 1; // sample index relative to c (c even)
				if (rel & 1)
					x[rr][j] = W::inv_scale(1, h[2 + ((rel - 1) >> 1)]);
				else
					x[rr][j] = W::inv_scale(0, l[2 + (rel >> 1)]);
			}
		}

		T val[2][NV]; // val[0] = L row p, val[1] = H row p as the vertical pass sees them
		if constexpr (!W::kInvColsFirst) {
#pragma unroll
			for (int rr = 0; rr < 2; rr++) {
				lift_inv_regs<W, NARR>(x[rr]);
				// after the horizontal inverse the row is plain samples again; the
				// vertical pass descales by ROW parity
#pragma unroll
				for (int v = 0; v < CPT; v++)
					val[rr][v] = W::inv_scale(rr, x[rr][K - 1 + v]);
			}
		} else {
#pragma unroll
			for (int rr = 0; rr < 2; rr++)
#pragma unroll
				for (int v = 0; v < NV; v++)
					val[rr][v] = x[rr][v]; // int 5/3: no scaling anywhere
		}

		// vertical inverse, streaming.  K == 4: at step p the rows 2p-3 (odd) and
		// 2p-2 (even) are final; K == 2: rows 2p-1 and 2p.
		T odd_row[NV], even_row[NV];
#pragma unroll
		for (int v = 0; v < NV; v++) {
			const T s2 = val[0][v], d2 = val[1][v];
			if constexpr (K == 4) {
				// st: [0] d2[p-1], [1] s1[p-1], [2] d1[p-2], [3] e[p-2]
				const T s1n = W::inv_step(0, s2, st[0][v], d2);            // s1[p]
				const T d1n = W::inv_step(1, st[0][v], st[1][v], s1n);     // d1[p-1]
				const T en = W::inv_step(2, st[1][v], st[2][v], d1n);      // e[p-1]
				const T on = W::inv_step(3, st[2][v], st[3][v], en);       // o[p-2]
				odd_row[v] = on;
				even_row[v] = en;
				st[0][v] = d2;
				st[1][v] = s1n;
				st[2][v] = d1n;
				st[3][v] = en;
			} else {
				// st: [0] d[p-1], [1] e[p-1]
				const T en = W::inv_step(0, s2, st[0][v], d2);             // e[p]
				const T on = W::inv_step(1, st[0][v], st[1][v], en);       // o[p-1]
				odd_row[v] = on;
				even_row[v] = en;
				st[0][v] = d2;
				st[1][v] = en;
			}
		}
		// output rows and their validity inside this tile
		const int pe = (K == 4) ? p - 1 : p;     // pair index of even_row
		const int po = (K == 4) ? p - 2 : p - 1; // pair index of odd_row
		const bool ve = pe >= A && pe < B;
		const bool vo = po >= A && po < B && (2 * po + 1 < a.H);

		T orow[CPT], erow[CPT];
		if constexpr (W::kInvColsFirst) {
			lift_inv_regs<W, NARR>(odd_row);
			lift_inv_regs<W, NARR>(even_row);
#pragma unroll
			for (int v = 0; v < CPT; v++) {
				orow[v] = odd_row[K - 1 + v];
				erow[v] = even_row[K - 1 + v];
			}
		} else {
#pragma unroll
			for (int v = 0; v < CPT; v++) {
				orow[v] = odd_row[v];
				erow[v] = even_row[v];
			}
		}

		const int c = c0 + lane * CPT;
		if (full && g.out_vec_ok) {
			if (vo) {
				T *o = out + (long)(2 * po + 1) * a.out_pitch + c;
#pragma unroll
				for (int e = 0; e < CPT; e += 4)
					store_vec<kNtStore>((u4 *)(o + e), u4{to_bits(orow[e]), to_bits(orow[e + 1]), to_bits(orow[e + 2]), to_bits(orow[e + 3])});
			}
			if (ve) {
				T *o = out + (long)(2 * pe) * a.out_pitch + c;
#pragma unroll
				for (int e = 0; e < CPT; e += 4)
					store_vec<kNtStore>((u4 *)(o + e), u4{to_bits(erow[e]), to_bits(erow[e + 1]), to_bits(erow[e + 2]), to_bits(erow[e + 3])});
			}
		} else {
			if (vo) {
				T *o = out + (long)(2 * po + 1) * a.out_pitch + c;
#pragma unroll
				for (int e = 0; e < CPT; e++)
					if (c + e < a.W)
						o[e] = orow[e];
			}
			if (ve) {
				T *o = out + (long)(2 * pe) * a.out_pitch + c;
#pragma unroll
				for (int e = 0; e < CPT; e++)
					if (c + e < a.W)
						o[e] = erow[e];
			}
		}
	}
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

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// Dynamic LDS above 64 KiB per workgroup has to be granted per kernel (gfx950 has
// 160 KiB per CU).
static hipError_t allow_lds(const void *kernel, size_t bytes)
{
	if (bytes <= 48 * 1024)
		return hipSuccess;
	return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
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
	// bits 0-2: cache policy (all eight built for A/B runs); bit 3 (wavefront shuffles
	// for the neighbour taps) is built on top of the default policy 7 and of policy 0
	int nt = t.nt & 15;
	if ((nt & 8) && nt != 15 && nt != 8)
		nt = 15;
#define DWT_FWD_CASE(R, N) case N: return fwd_launch<W, CPT, R, N>(a, g, grid, waves, s)
	if (nt == 7 && (t.ring == 10 || t.ring == 12 || t.ring == 14)) {
		if (t.ring == 10)
			return fwd_launch<W, CPT, 10, 7>(a, g, grid, waves, s);
		if (t.ring == 12)
			return fwd_launch<W, CPT, 12, 7>(a, g, grid, waves, s);
		return fwd_launch<W, CPT, 14, 7>(a, g, grid, waves, s);
	}
	if (t.ring == 16) {
		switch (nt) {
			DWT_FWD_CASE(16, 0); DWT_FWD_CASE(16, 1); DWT_FWD_CASE(16, 2); DWT_FWD_CASE(16, 3);
			DWT_FWD_CASE(16, 4); DWT_FWD_CASE(16, 5); DWT_FWD_CASE(16, 6); DWT_FWD_CASE(16, 7);
			DWT_FWD_CASE(16, 8); DWT_FWD_CASE(16, 15);
		}
	}
	switch (nt) {
		DWT_FWD_CASE(8, 0); DWT_FWD_CASE(8, 1); DWT_FWD_CASE(8, 2); DWT_FWD_CASE(8, 3);
		DWT_FWD_CASE(8, 4); DWT_FWD_CASE(8, 5); DWT_FWD_CASE(8, 6); DWT_FWD_CASE(8, 7);
		DWT_FWD_CASE(8, 8); DWT_FWD_CASE(8, 15);
	}
#undef DWT_FWD_CASE
	return hipErrorInvalidValue;
}

template <class W>
static hipError_t fwd_level_t(const FwdLevelArgs &a, const SweepTuning &t, hipStream_t s)
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
	const int Wd = (a.W + 1) / 2, Hd = (a.H + 1) / 2;
	g.in_vec_ok = aligned16(a.in) && (a.in_pitch % 4 == 0) && (a.in_bstride % 4 == 0);
	const int ov = cpt / 2; // elements per vector store
	g.out_vec_ok = ((uintptr_t)a.out_ll % (4 * ov) == 0) && ((uintptr_t)a.out_h % (4 * ov) == 0) &&
		(a.ll_pitch % ov == 0) && (a.h_pitch % ov == 0) && (a.ll_bstride % ov == 0) && (a.h_bstride % ov == 0) &&
		(Wd % ov == 0);
	const int waves = t.waves >= 1 && t.waves <= 4 ? t.waves : 4;
	const int nty = (Hd + g.tile_pairs - 1) / g.tile_pairs;
	// Ring depth (measured, scripts/sweep.py): when a launch has several rounds of tiles
	// per CU, 4 waves/CU with a 16-row ring (7 iterations of DMA in flight per wave) and
	// side-by-side waves beat 8 waves/CU with an 8-row ring (5.5 -> 6.0 TB/s at level 0);
	// smaller launches prefer more resident waves.
	SweepTuning tt = t;
	if (tt.ring != 8 && tt.ring != 16 && tt.ring != 10 && tt.ring != 12 && tt.ring != 14)
		tt.ring = (g.ntx >= waves && (long)g.ntx * nty * a.batch >= 3072) ? 16 : 8;
	if (tt.wave_horiz < 0)
		tt.wave_horiz = tt.ring >= 10;
	g.wave_horiz = tt.wave_horiz;
	dim3 grid;
	if (g.wave_horiz)
		grid = dim3(((g.ntx + waves - 1) / waves) * nty, a.batch);
	else
		grid = dim3(g.ntx * ((nty + waves - 1) / waves), a.batch);
	if (a.interleaved) {
		// 3-D path: float 9/7 only; always 4 columns per lane so that each row leaves the
		// wave as ONE contiguous 16 B/lane store (two strided stores per row cost 40 %)
		if constexpr (std::is_base_of<Cdf97S, W>::value || std::is_base_of<Cdf53S, W>::value) {
			g.out_vec_ok = aligned16(a.out_h) && (a.h_pitch % 4 == 0) && (a.h_bstride % 4 == 0);
			g.ll_vec_ok = a.il_ll && ((uintptr_t)a.out_ll % 8 == 0) && (a.ll_pitch % 2 == 0) && (a.ll_bstride % 2 == 0);
			if (tt.ring == 16)
				return fwd_launch<W, 4, 16, 3, true>(a, g, grid, waves, s);
			return fwd_launch<W, 4, 8, 3, true>(a, g, grid, waves, s);
		} else {
			return hipErrorInvalidValue;
		}
	}
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
static hipError_t inv_pick(const InvLevelArgs &a, const SweepGeom &g, dim3 grid, int waves, int ring, int nt, hipStream_t s)
{
#define DWT_INV_CASE(R, N) case N: return inv_launch<W, CPT, R, N, false>(a, g, grid, waves, s)
	if (ring == 16) {
		switch (nt & 3) { DWT_INV_CASE(16, 0); DWT_INV_CASE(16, 1); DWT_INV_CASE(16, 2); DWT_INV_CASE(16, 3); }
	}
	switch (nt & 3) { DWT_INV_CASE(8, 0); DWT_INV_CASE(8, 1); DWT_INV_CASE(8, 2); DWT_INV_CASE(8, 3); }
#undef DWT_INV_CASE
	return hipErrorInvalidValue;
}

template <class W>
static hipError_t inv_level_t(const InvLevelArgs &a, const SweepTuning &t, hipStream_t s)
{
	if (a.W < 2 || a.H < 2 || a.batch < 1)
		return hipErrorInvalidValue;
	const int cpt = a.interleaved ? 4 : pick_cpt(t, a.W, true);
	const int TW = 64 * cpt;
	SweepGeom g;
	g.tile_pairs = pick_tile_pairs(t, a.W, a.H, cpt, a.batch, true);
	g.ntx = (a.W + TW - 1) / TW;
	g.swz = t.xcd_swizzle;
	const int Wd = (a.W + 1) / 2, Hd = (a.H + 1) / 2;
	g.in_vec_ok = aligned16(a.in_ll) && aligned16(a.in_h) && (a.ll_pitch % 4 == 0) && (a.h_pitch % 4 == 0) &&
		(a.ll_bstride % 4 == 0) && (a.h_bstride % 4 == 0) && (Wd % 4 == 0);
	g.out_vec_ok = aligned16(a.out) && (a.out_pitch % 4 == 0) && (a.out_bstride % 4 == 0);
	const int waves = t.waves >= 1 && t.waves <= 4 ? t.waves : 4;
	const int nty = (Hd + g.tile_pairs - 1) / g.tile_pairs;
	int ring = t.ring_inv == 16 ? 16 : 8;
	g.wave_horiz = t.wave_horiz_inv > 0;
	dim3 grid;
	if (g.wave_horiz)
		grid = dim3(((g.ntx + waves - 1) / waves) * nty, a.batch);
	else
		grid = dim3(g.ntx * ((nty + waves - 1) / waves), a.batch);
	if (a.interleaved) {
		if constexpr (std::is_base_of<Cdf97S, W>::value || std::is_base_of<Cdf53S, W>::value) {
			g.in_vec_ok = aligned16(a.in_h) && (a.h_pitch % 4 == 0) && (a.h_bstride % 4 == 0) &&
				aligned16(a.in_ll) && (a.ll_pitch % 4 == 0) && (a.ll_bstride % 4 == 0);
			return cpt == 8 ? inv_launch<W, 8, 8, 0, true>(a, g, grid, waves, s) : inv_launch<W, 4, 8, 0, true>(a, g, grid, waves, s);
		} else {
			return hipErrorInvalidValue;
		}
	}
	return cpt == 8 ? inv_pick<W, 8>(a, g, grid, waves, ring, t.nt_inv, s) : inv_pick<W, 4>(a, g, grid, waves, ring, t.nt_inv, s);
}

// ---- forward, two levels in one sweep ---------------------------------------------
// Levels j and j+1 of the forward driver fused: the LL band between them never
// reaches HBM (-2 of the 10.66 B per sample of a 5-level transform).  A wave runs the
// level-j sweep exactly as k_fwd_sweep does on 512 columns, but keeps every LL row it
// produces in a wave-private LDS ring; whenever an even LL row arrives it runs one
// iteration of a second, level-(j+1) sweep on the ring (4 LL columns per lane, the
// 4-column neighbour taps read back from the ring).  Lanes 0 and 63 only feed their
// neighbours: a tile advances by 62 lanes = 496 columns (1984 B = 31 x 64 B), and a
// tile's level-j sweep starts K LL rows early / ends K-2 late for the inner sweep's
// warm-up.  Image borders: the outer sweep reflects the source address as before; at
// the left/top the LL rows/columns it produces from reflected input ARE the inner
// level's symmetric extension, at the right/bottom (even sizes) they are not, so the
// inner sweep reflects its ring index there explicitly.
struct Fwd2Geom {
	int tile_pairs1, ntx, swz, in_vec_ok, out_vec_ok;
};

static __device__ __forceinline__ void lds_write4(unsigned addr, u4 v)
{
	asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

static __device__ __forceinline__ unsigned lds_read_dword(unsigned addr)
{
	unsigned r;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
	return r;
}

template <class W, int RING, int NT>
__global__ __launch_bounds__(256) void k_fwd2_sweep(Fwd2LevelArgs a, Fwd2Geom g)
{
	using T = typename W::T;
	constexpr int K = W::K;
	constexpr int CPT = 8, TW = 512, RS = TW + 8, NARR = CPT + 2 * K;
	constexpr int kAhead = RING / 2 - 1;
	constexpr int kDmaPerIter = 2 * (CPT / 4 + 1);
	constexpr int kLdAux = (NT & 2) ? 2 : 0;
	constexpr bool kNtStore = (NT & 1) != 0;
	constexpr int kLLRing = 8;             // LL rows kept (the bottom reflection reaches 4 back)
	constexpr int kStride = 62 * CPT;      // columns a tile advances by
	constexpr int kWaveLds = RING * RS * 4 + kLLRing * 1024;
	extern __shared__ __attribute__((aligned(16))) char smem[];

	const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
	// wave-uniform on purpose: tile geometry, row indices and row pointers then live in SGPRs
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int bid = tile_block_id(g.swz);
	const int ntxb = (g.ntx + nwv - 1) / nwv;
	const int tx = (bid % ntxb) * nwv + wv;
	const int ty = bid / ntxb;
	const int img = blockIdx.y;
	const int Wd0 = (a.W + 1) >> 1, Hd0 = (a.H + 1) >> 1; // level j+1 input = LL of level j
	const int Wd1 = (Wd0 + 1) >> 1, Hd1 = (Hd0 + 1) >> 1;
	const int A1 = ty * g.tile_pairs1;
	if (A1 >= Hd1 || tx >= g.ntx)
		return;
	const int B1 = min(A1 + g.tile_pairs1, Hd1);
	const int c0 = kStride * tx - CPT;  // lane 0 is a halo lane
	const int cl = c0 / 2;              // LL column of lane 0's first value (c0 is a multiple of 8)
	const int kstart = 2 * A1 - K;
	const int kend = min(2 * B1 + K - 2, Hd0 - 1);
	const int n_iter = (kend + 1 - kstart) + K;
	const int q0 = kstart - K / 2;

	const T *in = (const T *)a.in + (long)img * a.in_bstride;
	T *out_ll2 = (T *)a.out_ll2 + (long)img * a.ll2_bstride;
	T *out_h = (T *)a.out_h + (long)img * a.h_bstride;

	char *ring = smem + (size_t)wv * kWaveLds;
	const unsigned ring_off = lds_offset(ring);
	const unsigned ll_off = ring_off + RING * RS * 4;

	const bool main16 = g.in_vec_ok && c0 >= 0 && c0 + TW <= a.W;
	int colmap[CPT];
#pragma unroll
	for (int i = 0; i < CPT; i++)
		colmap[i] = reflect(c0 + i * 64 + lane, a.W);
	const int halo_col = reflect(lane < 4 ? c0 - 4 + lane : c0 + TW + (lane & 3), a.W);

	// ring slots advance by two rows per iteration; kept as running counters (RING need
	// not be a power of two); rows reflect with a single bounce (H >= 64 here)
	int islot = 0, rslot = 0;
	auto issue = [&](int it) {
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const int r = reflect1(2 * (q0 + it) - 1 + rr, a.H);
			char *lrow = ring + (size_t)(islot + rr) * RS * 4;
			const T *grow = in + (long)r * a.in_pitch;
			if (main16) {
#pragma unroll
				for (int i = 0; i < CPT / 4; i++)
					dma16<kLdAux>(grow + c0 + i * 256 + lane * 4, lrow + i * 1024);
			} else {
#pragma unroll
				for (int i = 0; i < CPT; i++)
					dma4<kLdAux>(grow + colmap[i], lrow + i * 256);
			}
			if (lane < 8)
				dma4<kLdAux>(grow + halo_col, lrow + TW * 4);
		}
		islot = islot + 2 >= RING ? 0 : islot + 2;
	};

	T st0[K][CPT], st1[K][4];
#pragma unroll
	for (int s = 0; s < K; s++) {
#pragma unroll
		for (int v = 0; v < CPT; v++)
			st0[s][v] = 0;
#pragma unroll
		for (int v = 0; v < 4; v++)
			st1[s][v] = 0;
	}

	const bool lane_valid = lane >= 1 && lane <= 62;
	const int ci = cl + 4 * lane;      // first LL / detail column of this lane at level j
	const int c2 = cl / 2 + 2 * lane;  // first column of this lane at level j+1 (cl is even)
	const bool ll_edge = cl + 256 + K > Wd0; // the inner taps may cross the right border

	// one iteration of the inner (level j+1) sweep: consumes LL rows 2*q1-1 and 2*q1
	auto inner = [&](int q1) {
		T row1[2][4];
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			int r = 2 * q1 - 1 + rr;
			if (r >= Hd0)
				r = 2 * (Hd0 - 1) - r; // bottom border: whole-sample reflection of the LL row index
			const unsigned rowb = ll_off + (unsigned)(r & (kLLRing - 1)) * 1024;
			T x[12];
			if (!ll_edge) {
				// positions 4*lane-4 .. 4*lane+7 of the ring row (halo lanes read clamped garbage)
				const unsigned own = rowb + lane * 16;
				const unsigned la = lane == 0 ? own : own - 16;
				const unsigned ra = lane == 63 ? own : own + 16;
				u4 L4, O4, R4;
				lds_read3(la, own, ra, L4, O4, R4);
#pragma unroll
				for (int e = 0; e < 4; e++) {
					x[e] = from_bits<T>(L4[e]);
					x[4 + e] = from_bits<T>(O4[e]);
					x[8 + e] = from_bits<T>(R4[e]);
				}
			} else {
#pragma unroll
				for (int j = 0; j < 12; j++) {
					int i = ci - 4 + j; // LL column
					if (i >= Wd0)
						i = 2 * (Wd0 - 1) - i; // right border
					int p = i - cl;
					p = p < 0 ? 0 : (p > 255 ? 255 : p);
					x[j] = from_bits<T>(lds_read_dword(rowb + p * 4));
				}
			}
			lift_fwd_regs<W, 12>(x);
#pragma unroll
			for (int v = 0; v < 4; v++)
				row1[rr][v] = W::fwd_scale(v & 1, x[4 + v]);
		}
		T lo1[4], hi1[4];
#pragma unroll
		for (int v = 0; v < 4; v++) {
			const T ov = row1[0][v], ev = row1[1][v];
			if constexpr (K == 4) {
				const T d1n = W::fwd_step(0, ov, st1[0][v], ev);
				const T s1n = W::fwd_step(1, st1[0][v], st1[1][v], d1n);
				const T d2n = W::fwd_step(2, st1[1][v], st1[2][v], s1n);
				const T s2n = W::fwd_step(3, st1[2][v], st1[3][v], d2n);
				lo1[v] = W::fwd_scale(0, s2n);
				hi1[v] = W::fwd_scale(1, d2n);
				st1[0][v] = ev;
				st1[1][v] = d1n;
				st1[2][v] = s1n;
				st1[3][v] = d2n;
			} else {
				const T d1n = W::fwd_step(0, ov, st1[0][v], ev);
				const T s1n = W::fwd_step(1, st1[0][v], st1[1][v], d1n);
				lo1[v] = W::fwd_scale(0, s1n);
				hi1[v] = W::fwd_scale(1, d1n);
				st1[0][v] = ev;
				st1[1][v] = d1n;
			}
		}
		const int k1 = q1 - K / 2;
		if (k1 >= A1 && k1 < B1 && lane_valid && c2 < Wd1) {
			T *ll = out_ll2 + (long)k1 * a.ll2_pitch + c2;
			T *hl = out_h + (long)k1 * a.h_pitch + Wd1 + c2;
			T *lh = out_h + (long)(Hd1 + k1) * a.h_pitch + c2;
			T *hh = lh + Wd1;
			const bool hrow = k1 < (Hd0 >> 1);
			const int nh = Wd0 >> 1;
			if (g.out_vec_ok && c2 + 2 <= nh) {
				*(u2 *)ll = u2{to_bits(lo1[0]), to_bits(lo1[2])};
				store_vec<kNtStore>((u2 *)hl, u2{to_bits(lo1[1]), to_bits(lo1[3])});
				if (hrow) {
					store_vec<kNtStore>((u2 *)lh, u2{to_bits(hi1[0]), to_bits(hi1[2])});
					store_vec<kNtStore>((u2 *)hh, u2{to_bits(hi1[1]), to_bits(hi1[3])});
				}
			} else {
#pragma unroll
				for (int e = 0; e < 2; e++) {
					if (c2 + e < Wd1) {
						ll[e] = lo1[2 * e];
						if (hrow)
							lh[e] = hi1[2 * e];
					}
					if (c2 + e < nh) {
						hl[e] = lo1[2 * e + 1];
						if (hrow)
							hh[e] = hi1[2 * e + 1];
					}
				}
			}
		}
	};

	for (int it = 0; it < kAhead && it < n_iter; it++)
		issue(it);
	int next_q1 = A1 - K / 2;

	for (int it = 0; it < n_iter; it++) {
		if (it + kAhead < n_iter) {
			issue(it + kAhead);
			DWT_WAIT_VMCNT(kAhead * kDmaPerIter);
		} else {
			DWT_WAIT_VMCNT(0);
		}
		T row[2][CPT];
#pragma unroll
		for (int rr = 0; rr < 2; rr++) {
			const unsigned base = ring_off + (unsigned)(rslot + rr) * RS * 4;
			const unsigned own = base + lane * CPT * 4;
			const unsigned la = lane == 0 ? base + TW * 4 : own - 16;
			const unsigned ra = lane == 63 ? base + TW * 4 + 16 : own + CPT * 4;
			T x[NARR];
			u4 L4, R4, O0, O1;
			lds_read4(la, own, ra, L4, O0, O1, R4);
#pragma unroll
			for (int e = 0; e < K; e++) {
				x[e] = from_bits<T>(L4[4 - K + e]);
				x[K + CPT + e] = from_bits<T>(R4[e]);
			}
#pragma unroll
			for (int e = 0; e < 4; e++) {
				x[K + e] = from_bits<T>(O0[e]);
				x[K + 4 + e] = from_bits<T>(O1[e]);
			}
			lift_fwd_regs<W, NARR>(x);
#pragma unroll
			for (int v = 0; v < CPT; v++)
				row[rr][v] = W::fwd_scale(v & 1, x[K + v]);
		}
		rslot = rslot + 2 >= RING ? 0 : rslot + 2;
		T lo[CPT], hi[CPT];
#pragma unroll
		for (int v = 0; v < CPT; v++) {
			const T ov = row[0][v], ev = row[1][v];
			if constexpr (K == 4) {
				const T d1n = W::fwd_step(0, ov, st0[0][v], ev);
				const T s1n = W::fwd_step(1, st0[0][v], st0[1][v], d1n);
				const T d2n = W::fwd_step(2, st0[1][v], st0[2][v], s1n);
				const T s2n = W::fwd_step(3, st0[2][v], st0[3][v], d2n);
				lo[v] = W::fwd_scale(0, s2n);
				hi[v] = W::fwd_scale(1, d2n);
				st0[0][v] = ev;
				st0[1][v] = d1n;
				st0[2][v] = s1n;
				st0[3][v] = d2n;
			} else {
				const T d1n = W::fwd_step(0, ov, st0[0][v], ev);
				const T s1n = W::fwd_step(1, st0[0][v], st0[1][v], d1n);
				lo[v] = W::fwd_scale(0, s1n);
				hi[v] = W::fwd_scale(1, d1n);
				st0[0][v] = ev;
				st0[1][v] = d1n;
			}
		}
		if (it >= K) {
			const int k0 = kstart + it - K;
			// detail subbands of the outer level: this tile's own rows and lanes only
			if (k0 >= 2 * A1 && k0 < 2 * B1 && lane_valid && ci < Wd0) {
				T *hl = out_h + (long)k0 * a.h_pitch + Wd0 + ci;
				T *lh = out_h + (long)(Hd0 + k0) * a.h_pitch + ci;
				T *hh = lh + Wd0;
				const bool hrow = k0 < (a.H >> 1);
				const int nh = a.W >> 1;
				if (g.out_vec_ok && ci + 4 <= nh) {
					store_vec<kNtStore>((u4 *)hl, u4{to_bits(lo[1]), to_bits(lo[3]), to_bits(lo[5]), to_bits(lo[7])});
					if (hrow) {
						store_vec<kNtStore>((u4 *)lh, u4{to_bits(hi[0]), to_bits(hi[2]), to_bits(hi[4]), to_bits(hi[6])});
						store_vec<kNtStore>((u4 *)hh, u4{to_bits(hi[1]), to_bits(hi[3]), to_bits(hi[5]), to_bits(hi[7])});
					}
				} else {
#pragma unroll
					for (int e = 0; e < 4; e++) {
						if (ci + e < Wd0 && hrow)
							lh[e] = hi[2 * e];
						if (ci + e < nh) {
							hl[e] = lo[2 * e + 1];
							if (hrow)
								hh[e] = hi[2 * e + 1];
						}
					}
				}
			}
			// the LL row goes to the ring (every lane: the halo lanes feed their neighbours)
			lds_write4(ll_off + (unsigned)(k0 & (kLLRing - 1)) * 1024 + lane * 16,
				u4{to_bits(lo[0]), to_bits(lo[2]), to_bits(lo[4]), to_bits(lo[6])});
			if ((k0 & 1) == 0) {
				inner(k0 >> 1);
				next_q1 = (k0 >> 1) + 1;
			}
		}
	}
	// bottom tiles: the remaining inner iterations take reflected rows from the ring
	for (int q1 = next_q1; q1 <= B1 + K / 2 - 1; q1++)
		inner(q1);
}

template <class W, int RING, int NT>
static hipError_t fwd2_launch(const Fwd2LevelArgs &a, const Fwd2Geom &g, dim3 grid, int waves, hipStream_t s)
{
	const size_t lds = (size_t)waves * (RING * (512 + 8) * 4 + 8 * 1024);
	if (hipError_t e = allow_lds((const void *)k_fwd2_sweep<W, RING, NT>, lds))
		return e;
	k_fwd2_sweep<W, RING, NT><<<grid, 64 * waves, lds, s>>>(a, g);
	return hipGetLastError();
}

// tile height (in level j+1 output pairs) or 0 when the fused sweep does not apply
int fwd2_tile_pairs(const Fwd2LevelArgs &a, const SweepTuning &t)
{
	if (t.fuse2 <= 0)
		return 0;
	// even sizes down to level j+1's output, room for the 496-column tiles, vector stores
	if ((a.W & 7) || (a.H & 3) || a.W < 1024 || a.H < 64)
		return 0;
	if (!aligned16(a.in) || (a.in_pitch & 3) || (a.in_bstride & 3) || !aligned16(a.out_h) || (a.h_pitch & 3) ||
		(a.h_bstride & 3) || ((uintptr_t)a.out_ll2 & 7) || (a.ll2_pitch & 1) || (a.ll2_bstride & 1))
		return 0;
	if (t.fuse2 > 1)
		return t.fuse2; // explicit tile height
	const long ntx = (a.W + 495) / 496;
	const int Hd1 = a.H / 4;
	for (int tp = 64; tp >= 16; tp >>= 1)
		if (ntx * ((Hd1 + tp - 1) / tp) * a.batch >= 2048)
			return tp;
	return 0; // too few tiles to fill the chip: the separate levels do better
}

template <class W>
static hipError_t fwd2_level_t(const Fwd2LevelArgs &a, const SweepTuning &t, hipStream_t s)
{
	Fwd2Geom g;
	g.tile_pairs1 = fwd2_tile_pairs(a, t);
	if (g.tile_pairs1 <= 0)
		return hipErrorInvalidValue;
	g.ntx = (a.W + 495) / 496;
	g.swz = t.xcd_swizzle;
	g.in_vec_ok = 1;
	g.out_vec_ok = 1;
	const int waves = t.waves >= 1 && t.waves <= 4 ? t.waves : 4;
	const int Hd1 = (a.H / 2 + 1) / 2;
	const int nty = (Hd1 + g.tile_pairs1 - 1) / g.tile_pairs1;
	dim3 grid(((g.ntx + waves - 1) / waves) * nty, a.batch);
	if (t.ring == 8)
		return fwd2_launch<W, 8, 3>(a, g, grid, waves, s);
	return fwd2_launch<W, 14, 3>(a, g, grid, waves, s);
}

hipError_t launch_fwd2_level(Wavelet w, const Fwd2LevelArgs &a, const SweepTuning &t, hipStream_t s)
{
	switch (w) {
	case kCdf97S: return fwd2_level_t<Cdf97S>(a, t, s);
	case kCdf53I: return fwd2_level_t<Cdf53I>(a, t, s);
	case kCdf53S: return fwd2_level_t<Cdf53S>(a, t, s);
	case kCdf97I: return fwd2_level_t<Cdf97I>(a, t, s);
	case kCdf97SFma: return fwd2_level_t<Cdf97SFma>(a, t, s);
	default: break;
	}
	return hipErrorInvalidValue;
}

hipError_t launch_fwd_level(Wavelet w, const FwdLevelArgs &a, const SweepTuning &t, hipStream_t s)
{
	switch (w) {
	case kCdf97S: return fwd_level_t<Cdf97S>(a, t, s);
	case kCdf53I: return fwd_level_t<Cdf53I>(a, t, s);
	case kCdf53S: return fwd_level_t<Cdf53S>(a, t, s);
	case kCdf97I: return fwd_level_t<Cdf97I>(a, t, s);
	case kCdf97SFma: return fwd_level_t<Cdf97SFma>(a, t, s);
	case kCdf53SNew: return a.interleaved ? fwd_level_t<Cdf53SNew>(a, t, s) : hipErrorInvalidValue;
	default: break; // the double-precision drivers run on the line-pass kernels
	}
	return hipErrorInvalidValue;
}

hipError_t launch_inv_level(Wavelet w, const InvLevelArgs &a, const SweepTuning &t, hipStream_t s)
{
	switch (w) {
	case kCdf97S: return inv_level_t<Cdf97S>(a, t, s);
	case kCdf53I: return inv_level_t<Cdf53I>(a, t, s);
	case kCdf53S: return inv_level_t<Cdf53S>(a, t, s);
	case kCdf97I: return inv_level_t<Cdf97I>(a, t, s);
	case kCdf97SFma: return inv_level_t<Cdf97SFma>(a, t, s);
	default: break;
	}
	return hipErrorInvalidValue;
}

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
	float *__restrict__ out, long out_sy, long out_sz, int nx, int ny, int nz, int tile_pairs, int vec_ok,
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
	const bool vec = vec_ok && (c + 256 * (NV - 1) + 4 <= nx);
	const float *src = in + (long)y * in_sy + c;
	float *dst = out + (long)y * out_sy + c;

	auto load = [&](int slice, float (&v)[CPT]) {
		const float *p = src + (long)reflect(slice, nz) * in_sz;
		if (vec) {
#pragma unroll
			for (int g = 0; g < NV; g++) {
				const u4 t = (NT & 2) ? __builtin_nontemporal_load((const u4 *)(p + 256 * g)) : *(const u4 *)(p + 256 * g);
#pragma unroll
				for (int e = 0; e < 4; e++)
					v[4 * g + e] = from_bits<float>(t[e]);
			}
		} else {
#pragma unroll
			for (int e = 0; e < CPT; e++) {
				const int x = c + 256 * (e >> 2) + (e & 3);
				v[e] = (x < nx) ? p[256 * (e >> 2) + (e & 3)] : 0.f;
			}
		}
	};
	auto store = [&](int slice, const float (&v)[CPT]) {
		float *p = dst + (long)slice * out_sz;
		if (vec) {
#pragma unroll
			for (int g = 0; g < NV; g++) {
				const u4 t = u4{to_bits(v[4 * g]), to_bits(v[4 * g + 1]), to_bits(v[4 * g + 2]), to_bits(v[4 * g + 3])};
				if (NT & 1)
					__builtin_nontemporal_store(t, (u4 *)(p + 256 * g));
				else
					*(u4 *)(p + 256 * g) = t;
			}
		} else {
#pragma unroll
			for (int e = 0; e < CPT; e++)
				if (c + 256 * (e >> 2) + (e & 3) < nx)
					p[256 * (e >> 2) + (e & 3)] = v[e];
		}
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
					float *p = lll + (long)k * lll_sz + (long)(y >> 1) * lll_sy + (c >> 1);
#pragma unroll
					for (int g = 0; g < NV; g++) {
						if (vec) {
							*(u2 *)(p + 128 * g) = u2{to_bits(o0[4 * g]), to_bits(o0[4 * g + 2])};
						} else {
							if (c + 256 * g < nx)
								p[128 * g] = o0[4 * g];
							if (c + 256 * g + 2 < nx)
								p[128 * g + 1] = o0[4 * g + 2];
						}
					}
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
	long out_sy, long out_sz, int nx, int ny, int nz, int tp, int vec_ok, float *lll, long lll_sy, long lll_sz)
{
	switch ((nt < 0 ? 0 : nt) & 3) {
	case 0: k_vol_z<INV, CPT, 0><<<grid, threads, 0, s>>>(in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, vec_ok, lll, lll_sy, lll_sz); break;
	case 1: k_vol_z<INV, CPT, 1><<<grid, threads, 0, s>>>(in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, vec_ok, lll, lll_sy, lll_sz); break;
	case 2: k_vol_z<INV, CPT, 2><<<grid, threads, 0, s>>>(in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, vec_ok, lll, lll_sy, lll_sz); break;
	default: k_vol_z<INV, CPT, 3><<<grid, threads, 0, s>>>(in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, vec_ok, lll, lll_sy, lll_sz); break;
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
	const int vec_ok = aligned16(in) && aligned16(out) && in_sy % 4 == 0 && in_sz % 4 == 0 && out_sy % 4 == 0 && out_sz % 4 == 0;
	if (inverse) {
		if (cpt == 8)
			vol_z_nt<true, 8>(vt.nt, grid, 64 * waves, s, in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, vec_ok, lll, lll_sy, lll_sz);
		else
			vol_z_nt<true, 4>(vt.nt, grid, 64 * waves, s, in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, vec_ok, lll, lll_sy, lll_sz);
	} else {
		if (cpt == 8)
			vol_z_nt<false, 8>(vt.nt, grid, 64 * waves, s, in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, vec_ok, lll, lll_sy, lll_sz);
		else
			vol_z_nt<false, 4>(vt.nt, grid, 64 * waves, s, in, in_sy, in_sz, out, out_sy, out_sz, nx, ny, nz, tp, vec_ok, lll, lll_sy, lll_sz);
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
static __device__ __forceinline__ void wg_barrier_lds()
{
	// LDS traffic of this wave done, then the barrier; outstanding LDS-DMA keeps flying
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NT>
__global__ __launch_bounds__(256) void k_vol_fwd_fused(VolFusedArgs a, int tile_pairs_z, int vec_ok, int ntx, int nty, int swz)
{
	using W = Cdf97S;
	// the 8 output rows of a wave need x-lifted rows -4 .. +10 around its first row: the tile's
	// 32 rows need 39 input rows; wave w stages rows w, w+4, ... (10, 10, 10, 9 of them)
	constexpr int K = 4, CPT = 4, TW = 256, RS = TW + 8, TY = 32, NR = TY + 2 * K - 1, RPW = (NR + 3) / 4;
	constexpr int kLdAux = (NT & 2) ? 2 : 0;
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
	const int halo_col = reflect(lane < 4 ? c0 - 4 + lane : c0 + TW + (lane & 3), a.nx);
	// tiles that overhang the volume (or unaligned volumes) are staged column by column
	const bool full = vec_ok && c0 + TW <= a.nx;
	int colmap[CPT];
#pragma unroll
	for (int i = 0; i < CPT; i++)
		colmap[i] = reflect(c0 + i * 64 + lane, a.nx);

	auto issue = [&](int t) {
		const float *sl = a.in + (long)reflect(2 * q0 - 1 + t, a.nz) * a.in_sz;
#pragma unroll
		for (int i = 0; i < RPW; i++) {
			if (wv + 4 * i < NR) {
				const int r = reflect(y0 - K + wv + 4 * i, a.ny);
				const float *grow = sl + (long)r * a.in_sy;
				char *lrow = ring + (size_t)i * RS * 4;
				if (full) {
					dma16<kLdAux>(grow + c, lrow); // the DMA places lane i's 16 B at lrow + 16 i
				} else {
#pragma unroll
					for (int e = 0; e < CPT; e++)
						dma4<kLdAux>(grow + colmap[e], lrow + e * 256);
				}
				if (lane < 8)
					dma4<kLdAux>(grow + halo_col, lrow + TW * 4);
			}
		}
	};

	float st[K][8][CPT], ra[8][CPT];
#pragma unroll
	for (int s = 0; s < K; s++)
#pragma unroll
		for (int r = 0; r < 8; r++)
#pragma unroll
			for (int e = 0; e < CPT; e++)
				st[s][r][e] = 0.f;

	issue(0);
	for (int t = 0; t < n_slices; t++) {
		DWT_WAIT_VMCNT(0); // this slice's rows have landed (and the previous stores are out)
		// horizontal lift of this wave's rows, parked in the shared slab
#pragma unroll
		for (int i = 0; i < RPW; i++) {
			if (wv + 4 * i < NR) {
				const unsigned base = ring_off + (unsigned)i * RS * 4;
				const unsigned own = base + lane * CPT * 4;
				const unsigned la = lane == 0 ? base + TW * 4 : own - 16;
				const unsigned ra_ = lane == 63 ? base + TW * 4 + 16 : own + CPT * 4;
				u4 L4, O0, R4;
				lds_read3(la, own, ra_, L4, O0, R4);
				float x[CPT + 2 * K];
#pragma unroll
				for (int e = 0; e < K; e++) {
					x[e] = from_bits<float>(L4[e]);
					x[K + e] = from_bits<float>(O0[e]);
					x[K + CPT + e] = from_bits<float>(R4[e]);
				}
				lift_fwd_regs<W, CPT + 2 * K>(x);
				const u4 o = u4{to_bits(W::fwd_scale(0, x[K])), to_bits(W::fwd_scale(1, x[K + 1])),
					to_bits(W::fwd_scale(0, x[K + 2])), to_bits(W::fwd_scale(1, x[K + 3]))};
				lds_write4(slab_off + (unsigned)(wv + 4 * i) * TW * 4 + lane * 16, o);
			}
		}
		// the staging rows are consumed: the next slice's DMA flies during the rest of the iteration
		if (t + 1 < n_slices)
			issue(t + 1);
		wg_barrier_lds(); // the slab is complete

		// vertical lift: slab rows 8 wv .. 8 wv + 14 give this wave's 8 output rows
		u4 v[15];
		{
			const unsigned vb = slab_off + (unsigned)(8 * wv) * TW * 4 + lane * 16;
			asm volatile(
				"ds_read_b128 %0, %15\n\tds_read_b128 %1, %15 offset:1024\n\tds_read_b128 %2, %15 offset:2048\n\tds_read_b128 %3, %15 offset:3072\n\t"
				"ds_read_b128 %4, %15 offset:4096\n\tds_read_b128 %5, %15 offset:5120\n\tds_read_b128 %6, %15 offset:6144\n\tds_read_b128 %7, %15 offset:7168\n\t"
				"ds_read_b128 %8, %15 offset:8192\n\tds_read_b128 %9, %15 offset:9216\n\tds_read_b128 %10, %15 offset:10240\n\tds_read_b128 %11, %15 offset:11264\n\t"
				"ds_read_b128 %12, %15 offset:12288\n\tds_read_b128 %13, %15 offset:13312\n\tds_read_b128 %14, %15 offset:14336\n\t"
				"s_waitcnt lgkmcnt(0)\n\ts_barrier"
				: "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
				  "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14])
				: "v"(vb)
				: "memory"); // the barrier: every wave has read the slab, the next slice may overwrite it
		}
		float cur[8][CPT];
#pragma unroll
		for (int e = 0; e < CPT; e++) {
			float col[15];
#pragma unroll
			for (int j = 0; j < 15; j++)
				col[j] = from_bits<float>(v[j][e]);
			lift_fwd_regs<W, 15>(col);
#pragma unroll
			for (int r = 0; r < 8; r++)
				cur[r][e] = W::fwd_scale(r & 1, col[K + r]);
		}

		// z: slices arrive as (2q-1, 2q); the odd one waits in registers for its partner
		if (!(t & 1)) {
#pragma unroll
			for (int r = 0; r < 8; r++)
#pragma unroll
				for (int e = 0; e < CPT; e++)
					ra[r][e] = cur[r][e];
			continue;
		}
		const int it = t >> 1;
		const int k = A + it - K;
#pragma unroll
		for (int r = 0; r < 8; r++) {
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
			const int y = y0 + 8 * wv + r;
			if (it >= K && y < a.ny) {
				float *p = a.out + (long)(2 * k) * a.out_sz + (long)y * a.out_sy + c;
				const bool hz = 2 * k + 1 < a.nz;
				float *pl = a.lll && !(r & 1) ? a.lll + (long)k * a.lll_sz + (long)(y >> 1) * a.lll_sy + (c >> 1) : nullptr;
				if (full) {
					store_vec<kNtStore>((u4 *)p, u4{to_bits(o0[0]), to_bits(o0[1]), to_bits(o0[2]), to_bits(o0[3])});
					if (hz)
						store_vec<kNtStore>((u4 *)(p + a.out_sz), u4{to_bits(o1[0]), to_bits(o1[1]), to_bits(o1[2]), to_bits(o1[3])});
					if (pl)
						*(u2 *)pl = u2{to_bits(o0[0]), to_bits(o0[2])};
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
	return aligned16(a.in) && aligned16(a.out) && a.in_sy % 4 == 0 && a.in_sz % 4 == 0 && a.out_sy % 4 == 0 && a.out_sz % 4 == 0 &&
		(!a.lll || (((uintptr_t)a.lll & 7) == 0 && a.lll_sy % 2 == 0 && a.lll_sz % 2 == 0));
}

hipError_t launch_vol_fwd_fused(const VolFusedArgs &a, const VolTuning &vt, hipStream_t s)
{
	if (a.in == a.out || a.nx < 2 || a.ny < 2 || a.nz < 2)
		return hipErrorInvalidValue;
	const int Zd = (a.nz + 1) / 2;
	const int ntx = (a.nx + 255) / 256, nty = (a.ny + 31) / 32;
	// z lines are split until the 512 workgroup slots (two per CU) are filled, but not below 32
	// slice pairs per march: the 8-slice warm-up is 12 % there (512^3: 32 pairs 0.31 ms, 16
	// pairs 0.37, 64 pairs -- half the CUs idle -- 0.55)
	int tp = 128;
	while (tp > 32 && (long)ntx * nty * ((Zd + tp - 1) / tp) < 512)
		tp >>= 1;
	if (vt.tile_pairs >= 4)
		tp = vt.tile_pairs;
	const int nzt = (Zd + tp - 1) / tp;
	if ((long)ntx * nty * nzt > 0x7fffffffL)
		return hipErrorInvalidValue;
	const size_t lds = (size_t)39 * (256 + 8) * 4 + (size_t)39 * 256 * 4;
	dim3 grid(ntx * nty * nzt);
	const int swz = vt.swizzle;
	if (vt.nt < 0 || (vt.nt & 1)) {
		if (hipError_t e = allow_lds((const void *)k_vol_fwd_fused<3>, lds))
			return e;
		k_vol_fwd_fused<3><<<grid, 256, lds, s>>>(a, tp, vol_fused_vec_ok(a), ntx, nty, swz);
	} else {
		if (hipError_t e = allow_lds((const void *)k_vol_fwd_fused<2>, lds))
			return e;
		k_vol_fwd_fused<2><<<grid, 256, lds, s>>>(a, tp, vol_fused_vec_ok(a), ntx, nty, swz);
	}
	return hipGetLastError();
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

hipError_t launch_lattice_copy(const float *src, long s_sx, long s_sy, long s_sz, float *dst, long d_sx, long d_sy, long d_sz,
	int nx, int ny, int nz, hipStream_t s)
{
	const int nxb = (nx + 255) / 256;
	if (nx < 1 || ny < 1 || nz < 1 || nz > 65535 || (long)nxb * ny > 0x7fffffffL)
		return hipErrorInvalidValue;
	dim3 grid(nxb * ny, nz);
	k_lattice_copy<<<grid, 256, 0, s>>>(src, s_sx, s_sy, s_sz, dst, d_sx, d_sy, d_sz, nx, ny, nxb);
	return hipGetLastError();
}

// ---- interleaved layout: one phase of the reference's phase-ordered lifting, exact ----
// Same windowed evaluation as k_line_pass, with every step masked to the index range the
// phase gives it.  Mirrored window entries (symmetric extension) carry the index they
// mirror, so they receive the same masked updates as their originals.
template <class W, bool INV>
__global__ __launch_bounds__(256) void k_il_phase(const char *__restrict__ src, char *__restrict__ dst,
	long line_stride, long elem_stride, int n_lines, int N, int lanes_along_lines, IlPhase ph)
{
	using T = typename W::T;
	constexpr int K = W::K, NW = 2 * K + 1;
	const int fast = blockIdx.x * blockDim.x + threadIdx.x;
	const int slow = blockIdx.y;
	const int line = lanes_along_lines ? fast : slow;
	const int k = lanes_along_lines ? slow : fast;
	if (line >= n_lines || k >= ((N + 1) >> 1))
		return;
	const char *s = src + (long)line * line_stride;
	char *d = dst + (long)line * line_stride;
	// forward: w[0] is the even sample 2k-K; inverse: the odd sample 2k-K+1
	const int first = 2 * k - K + (INV ? 1 : 0);
	T w[NW];
	int idx[NW];
#pragma unroll
	for (int j = 0; j < NW; j++) {
		idx[j] = reflect(first + j, N);
		w[j] = *(const T *)(s + (long)idx[j] * elem_stride);
		if (INV && idx[j] >= ph.sc_lo && idx[j] <= ph.sc_hi)
			w[j] = W::inv_scale(idx[j] & 1, w[j]);
	}
#pragma unroll
	for (int st = 0; st < K; st++) {
#pragma unroll
		for (int j = st + 1; j <= NW - 2 - st; j += 2)
			if (idx[j] >= ph.lo[st] && idx[j] <= ph.hi[st])
				w[j] = INV ? W::inv_step(st, w[j], w[j - 1], w[j + 1]) : W::fwd_step(st, w[j], w[j - 1], w[j + 1]);
	}
	const int c0 = INV ? K - 1 : K; // window position of sample 2k
#pragma unroll
	for (int e = 0; e < 2; e++) {
		const int i = 2 * k + e;
		if (i < N) {
			T v = w[c0 + e];
			if (!INV && i >= ph.sc_lo && i <= ph.sc_hi)
				v = W::fwd_scale(e, v);
			*(T *)(d + (long)i * elem_stride) = v;
		}
	}
}

template <class W>
static hipError_t il_phase_t(bool inverse, const void *src, void *dst, long line_stride, long elem_stride, int n_lines, int N,
	bool lanes_along_lines, const IlPhase &ph, hipStream_t s)
{
	if (n_lines <= 0 || N < 2)
		return hipErrorInvalidValue;
	const int npairs = (N + 1) >> 1;
	const int fast = lanes_along_lines ? n_lines : npairs;
	const int slow = lanes_along_lines ? npairs : n_lines;
	const int bs = fast >= 256 ? 256 : 64;
	dim3 grid((fast + bs - 1) / bs, slow);
	if (inverse)
		k_il_phase<W, true><<<grid, bs, 0, s>>>((const char *)src, (char *)dst, line_stride, elem_stride, n_lines, N, lanes_along_lines, ph);
	else
		k_il_phase<W, false><<<grid, bs, 0, s>>>((const char *)src, (char *)dst, line_stride, elem_stride, n_lines, N, lanes_along_lines, ph);
	return hipGetLastError();
}

hipError_t launch_il_phase(Wavelet w, bool inverse, const void *src, void *dst, long line_stride, long elem_stride,
	int n_lines, int N, bool lanes_along_lines, const IlPhase &ph, hipStream_t s)
{
	switch (w) {
	case kCdf97S: return il_phase_t<Cdf97S>(inverse, src, dst, line_stride, elem_stride, n_lines, N, lanes_along_lines, ph, s);
	case kCdf53SNew: return il_phase_t<Cdf53SNew>(inverse, src, dst, line_stride, elem_stride, n_lines, N, lanes_along_lines, ph, s);
	default: break;
	}
	return hipErrorInvalidValue;
}

// ---- interleaved layout: all levels' lattices in one pass over the even rows ----
// A lattice-1 point (p, q) (image column 2p, row 2q) belongs to level
// j = 1 + min(ctz(p), ctz(q)) capped at J-1; its sample sits at (p >> (j-1), q >> (j-1))
// of that level's dense image.  One thread owns 8 image columns of one even row.
static __device__ __forceinline__ int il_level_of(int p, int q, int J)
{
	const int t = __builtin_ctz((unsigned)(p | q) | (1u << 30)); // ctz(0) -> 30
	const int j = 1 + t;
	return j < J ? j : J - 1;
}

__global__ __launch_bounds__(256) void k_il_compose(const float *__restrict__ base, long base_pitch, float *__restrict__ out,
	long out_pitch, int W, int H, IlPyramid py, int vec_ok, int out_dense)
{
	// grid.x = even rows (may exceed 65535), grid.y = blocks of 2048 columns
	const int x0 = (blockIdx.y * blockDim.x + threadIdx.x) * 8;
	const int q = blockIdx.x, y = 2 * q;
	if (x0 >= W || y >= H)
		return;
	const float *b = base + (long)y * base_pitch + x0;
	float *o = out + (long)(out_dense ? q : y) * out_pitch + x0;
	const int p0 = x0 >> 1;
	float v[8];
	const bool vec = vec_ok && x0 + 8 <= W;
	if (vec) {
		const u4 t0 = *(const u4 *)b, t1 = *(const u4 *)(b + 4);
#pragma unroll
		for (int e = 0; e < 4; e++) {
			v[e] = from_bits<float>(t0[e]);
			v[4 + e] = from_bits<float>(t1[e]);
		}
		const u4 l1 = *(const u4 *)(py.p[1] + (long)q * py.pitch[1] + p0);
#pragma unroll
		for (int i = 0; i < 4; i++)
			v[2 * i] = from_bits<float>(l1[i]);
	} else {
#pragma unroll
		for (int e = 0; e < 8; e++)
			if (x0 + e < W)
				v[e] = (e & 1) ? b[e] : py.p[1][(long)q * py.pitch[1] + p0 + (e >> 1)];
	}
	if (py.J > 2 && !(q & 1)) {
		// p0 is a multiple of 4: the points p0 and p0+2 lie on deeper lattices
#pragma unroll
		for (int i = 0; i < 4; i += 2)
			if (x0 + 2 * i < W) {
				const int p = p0 + i, j = il_level_of(p, q, py.J);
				v[2 * i] = py.p[j][(long)(q >> (j - 1)) * py.pitch[j] + (p >> (j - 1))];
			}
	}
	if (vec) {
		*(u4 *)o = u4{to_bits(v[0]), to_bits(v[1]), to_bits(v[2]), to_bits(v[3])};
		*(u4 *)(o + 4) = u4{to_bits(v[4]), to_bits(v[5]), to_bits(v[6]), to_bits(v[7])};
	} else {
#pragma unroll
		for (int e = 0; e < 8; e++)
			if (x0 + e < W)
				o[e] = v[e];
	}
}

__global__ __launch_bounds__(256) void k_il_decompose(const float *__restrict__ img, long pitch, int W, int H, IlPyramid py, int vec_ok)
{
	const int x0 = (blockIdx.y * blockDim.x + threadIdx.x) * 8;
	const int q = blockIdx.x, y = 2 * q;
	if (x0 >= W || y >= H)
		return;
	const float *b = img + (long)y * pitch + x0;
	const int p0 = x0 >> 1;
	float v[4];
	const bool vec = vec_ok && x0 + 8 <= W;
	if (vec) {
		const u4 t0 = *(const u4 *)b, t1 = *(const u4 *)(b + 4);
		v[0] = from_bits<float>(t0[0]); v[1] = from_bits<float>(t0[2]);
		v[2] = from_bits<float>(t1[0]); v[3] = from_bits<float>(t1[2]);
		*(u4 *)(py.p[1] + (long)q * py.pitch[1] + p0) = u4{to_bits(v[0]), to_bits(v[1]), to_bits(v[2]), to_bits(v[3])};
	} else {
#pragma unroll
		for (int i = 0; i < 4; i++)
			if (x0 + 2 * i < W) {
				v[i] = b[2 * i];
				py.p[1][(long)q * py.pitch[1] + p0 + i] = v[i];
			}
	}
	// deeper lattices: level j takes the points whose p and q are multiples of 2^(j-1)
	for (int j = 2; j < py.J; j++) {
		const int m = (1 << (j - 1)) - 1;
		if (q & m)
			break;
#pragma unroll
		for (int i = 0; i < 4; i += 2)
			if (!((p0 + i) & m) && x0 + 2 * i < W)
				py.p[j][(long)(q >> (j - 1)) * py.pitch[j] + ((p0 + i) >> (j - 1))] = v[i];
	}
}

static int il_vec_ok(const float *a, long ap, const float *b, long bp, const IlPyramid &py)
{
	return aligned16(a) && aligned16(b) && ap % 4 == 0 && bp % 4 == 0 && py.J > 1 && aligned16(py.p[1]) && py.pitch[1] % 4 == 0;
}

hipError_t launch_il_compose(const float *base, long base_pitch, float *out, long out_pitch, int W, int H, const IlPyramid &py, hipStream_t s,
	bool out_dense)
{
	if (py.J < 2 || py.J > 24 || W < 1 || H < 1 || ((W + 7) / 8 + 255) / 256 > 65535)
		return hipErrorInvalidValue;
	dim3 grid((H + 1) / 2, ((W + 7) / 8 + 255) / 256);
	k_il_compose<<<grid, 256, 0, s>>>(base, base_pitch, out, out_pitch, W, H, py, il_vec_ok(base, base_pitch, out, out_pitch, py), out_dense);
	return hipGetLastError();
}

hipError_t launch_il_decompose(const float *img, long pitch, int W, int H, const IlPyramid &py, hipStream_t s)
{
	if (py.J < 2 || py.J > 24 || W < 1 || H < 1 || ((W + 7) / 8 + 255) / 256 > 65535)
		return hipErrorInvalidValue;
	dim3 grid((H + 1) / 2, ((W + 7) / 8 + 255) / 256);
	k_il_decompose<<<grid, 256, 0, s>>>(img, pitch, W, H, py, il_vec_ok(img, pitch, img, pitch, py));
	return hipGetLastError();
}

bool have_fused_inverse(Wavelet) { return true; }

} // namespace dwt

// ---------------------------------------------------------------------------------
// 4. device-side view helpers: conv_show and compare on images
//    that stay in HBM between the forward and the inverse transform
// ---------------------------------------------------------------------------------
namespace dwt {

// dwt_util_conv_show_s (src/libdwt.c:21075-21117): log(1 + |c|*100) / 10 with the log
// taken in double as log_i_s does (:21010); non-finite results become 0.
__global__ __launch_bounds__(256) void k_conv_show_s(const char *__restrict__ src, char *__restrict__ dst, long pitch, int w, int h)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	if (x >= w || y >= h)
		return;
	const float c = *(const float *)(src + (long)y * pitch + (long)x * 4);
	float t = (float)log((double)(1.f + fabsf(c) * 100.f));
	t /= 10.f;
	if (!isfinite(t))
		t = 0.f;
	*(float *)(dst + (long)y * pitch + (long)x * 4) = t;
}

// dwt_util_conv_show_i (src/libdwt.c:21020-21044): |c|
__global__ __launch_bounds__(256) void k_conv_show_i(const char *__restrict__ src, char *__restrict__ dst, long pitch, int w, int h)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	if (x >= w || y >= h)
		return;
	const int c = *(const int *)(src + (long)y * pitch + (long)x * 4);
	*(int *)(dst + (long)y * pitch + (long)x * 4) = c < 0 ? -c : c;
}

// dwt_util_compare_s / _i (src/libdwt.c:1593-1620, 1531-1558): count of differing
// elements (float: |a-b| > 1e-3 or any NaN/Inf; int: a != b) accumulated in *result
template <bool IS_INT>
__global__ __launch_bounds__(256) void k_compare(const char *__restrict__ p1, const char *__restrict__ p2, long pitch, int w, int h, unsigned *result)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	bool differ = false;
	if (x < w && y < h) {
		if (IS_INT) {
			differ = *(const int *)(p1 + (long)y * pitch + (long)x * 4) != *(const int *)(p2 + (long)y * pitch + (long)x * 4);
		} else {
			const float a = *(const float *)(p1 + (long)y * pitch + (long)x * 4);
			const float b = *(const float *)(p2 + (long)y * pitch + (long)x * 4);
			differ = isnan(a) || isinf(a) || isnan(b) || isinf(b) || fabsf(a - b) > 1e-3f;
		}
	}
	const unsigned long long m = __ballot(differ);
	if ((threadIdx.x & 63) == 0 && m)
		atomicAdd(result, (unsigned)__popcll(m));
}

hipError_t launch_conv_show(bool is_int, const void *src, void *dst, long pitch, int w, int h, hipStream_t s)
{
	if (w <= 0 || h <= 0)
		return hipSuccess;
	dim3 grid((w + 255) / 256, h);
	if (is_int)
		k_conv_show_i<<<grid, 256, 0, s>>>((const char *)src, (char *)dst, pitch, w, h);
	else
		k_conv_show_s<<<grid, 256, 0, s>>>((const char *)src, (char *)dst, pitch, w, h);
	return hipGetLastError();
}

hipError_t launch_compare(bool is_int, const void *p1, const void *p2, long pitch, int w, int h, unsigned *result, hipStream_t s)
{
	if (w <= 0 || h <= 0)
		return hipSuccess;
	dim3 grid((w + 255) / 256, h);
	if (is_int)
		k_compare<true><<<grid, 256, 0, s>>>((const char *)p1, (const char *)p2, pitch, w, h, result);
	else
		k_compare<false><<<grid, 256, 0, s>>>((const char *)p1, (const char *)p2, pitch, w, h, result);
	return hipGetLastError();
}

} // namespace dwt
