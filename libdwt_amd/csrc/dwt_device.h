// dwt_device.h -- device-side helpers shared by the kernel files: bit casts, LDS-DMA, LDS reads
// through inline asm, wavefront shifts, workgroup -> tile mapping, launch helpers.
#pragma once
#include "dwt_kernels.h"
#include "dwt_lift.h"

#include <stdint.h>
#include <type_traits>

namespace dwt {

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <class T> static __device__ __forceinline__ T from_bits(unsigned u) { return __builtin_bit_cast(T, u); }
template <class T> static __device__ __forceinline__ unsigned to_bits(T v) { return __builtin_bit_cast(unsigned, v); }

// First statement of a line-end path (dwt_lift.h): an asm the optimiser may not speculate, so that the path stays behind a
// REAL (scalar) branch -- if-conversion otherwise folds the rarely needed end forms into selects that every wave executes
// (measured on the 3-D level kernels: + 25 %)
#define DWT_END_PATH() asm volatile("; line-end forms" ::: "memory")

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

// Row-wise buffer addressing: a descriptor per row (wave-uniform base, length in bytes) and a
// per-lane byte offset.  The hardware drops stores and zero-fills loads whose offset is past the
// row's end, so a tile that overhangs the image needs no per-lane branches.
typedef __amdgpu_buffer_rsrc_t row_rsrc_t;
static __device__ __forceinline__ row_rsrc_t row_rsrc(const void *base, unsigned bytes)
{
	return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}

template <int AUX = 0>
static __device__ __forceinline__ void dma16_row(row_rsrc_t r, unsigned byte_off, void *l)
{
	__builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l, 16, byte_off, 0, 0, AUX);
}

template <int AUX = 0>
static __device__ __forceinline__ void dma4_row(row_rsrc_t r, unsigned byte_off, void *l)
{
	__builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l, 4, byte_off, 0, 0, AUX);
}

template <bool NT>
static __device__ __forceinline__ u4 load16_row(row_rsrc_t r, unsigned byte_off)
{
	return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, NT ? 2 : 0);
}

template <bool NT>
static __device__ __forceinline__ void store16_row(row_rsrc_t r, unsigned byte_off, u4 v)
{
	__builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, NT ? 2 : 0);
}

template <bool NT>
static __device__ __forceinline__ void store8_row(row_rsrc_t r, unsigned byte_off, u2 v)
{
	__builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, NT ? 2 : 0);
}

template <bool NT>
static __device__ __forceinline__ void store4_row(row_rsrc_t r, unsigned byte_off, unsigned v)
{
	__builtin_amdgcn_raw_buffer_store_b32(v, r, byte_off, 0, NT ? 2 : 0);
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

// Split form for software pipelines: issue three reads now, consume them after lds_arrived3<N>
// ("all but the N newest LDS operations of this wave are done"; the operands pass through the
// wait statement so that no use can be scheduled above it).
static __device__ __forceinline__ void lds_issue3(unsigned a0, unsigned a1, unsigned a2, u4 &r0, u4 &r1, u4 &r2)
{
	asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5"
		: "=&v"(r0), "=&v"(r1), "=&v"(r2)
		: "v"(a0), "v"(a1), "v"(a2)
		: "memory");
}

template <int N>
static __device__ __forceinline__ void lds_arrived3(u4 &r0, u4 &r1, u4 &r2)
{
	asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(r0), "+v"(r1), "+v"(r2) : "n"(N) : "memory");
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
static __device__ __forceinline__ int tile_block_id(int swz, int first = 0, int tile_blocks = 0)
{
	// (`first`: leading workgroups of the launch that do not take tiles; `tile_blocks` > 0: trailing ones neither)
	int b = blockIdx.x - first;
	const int nb = tile_blocks > 0 ? tile_blocks : gridDim.x - first;
	if (swz && (nb & 7) == 0)
		b = (b & 7) * (nb >> 3) + (b >> 3);
	return b;
}

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

static __device__ __forceinline__ void lds_write_dword(unsigned addr, unsigned v)
{
	asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

static __device__ __forceinline__ void wg_barrier_lds()
{
	// LDS traffic of this wave done, then the barrier; outstanding LDS-DMA keeps flying
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One block of a rectangle copy (CopyRects): 8 rows x 4 KiB per workgroup of 256 threads, 16 B per lane, rows as
// buffers (any 4-byte alignment; the dwords past a row's end are zero-filled / dropped by the bounds check).  Rows past
// the rectangle's end load the last row again (never stored): straight-line loads, all eight in flight (with the loads
// under `if (i < rows)` the compiler built a 236-register kernel).
template <bool NTL, bool NTS>
static __device__ __forceinline__ void copy_rects_block(const CopyRects &r, int b)
{
	int k = 0;
	while (k + 1 < r.n && b >= r.first_block[k + 1])
		k++;
	if (b >= r.first_block[r.n])
		return;
	b -= r.first_block[k];
	constexpr int kRows = 8;
	constexpr int seg = 256 * 16; // bytes per workgroup row segment
	const int nbx = (r.wbytes[k] + seg - 1) / seg;
	const int bx = b % nbx, by = b / nbx;
	const unsigned x = (unsigned)bx * seg + threadIdx.x * 16;
	const char *s = r.src[k] + (long)by * kRows * r.spitch[k];
	char *d = r.dst[k] + (long)by * kRows * r.dpitch[k];
	const int rows = min(kRows, r.h[k] - by * kRows);
	u4 v[kRows];
#pragma unroll
	for (int i = 0; i < kRows; i++)
		v[i] = load16_row<NTL>(row_rsrc(s + (long)min(i, rows - 1) * r.spitch[k], (unsigned)r.wbytes[k]), x);
#pragma unroll
	for (int i = 0; i < kRows; i++)
		if (i < rows)
			store16_row<NTS>(row_rsrc(d + (long)i * r.dpitch[k], (unsigned)r.wbytes[k]), x, v[i]);
}

// the trailing workgroups of a sweep launch that carry a copy along (FwdLevelArgs::ride)
static __device__ __forceinline__ void ride_copy_block(const CopyRects &r, int b)
{
	if ((r.policy & 3) == 0)
		copy_rects_block<false, false>(r, r.block0 + b);
	else
		copy_rects_block<true, true>(r, r.block0 + b);
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// Dynamic LDS above 64 KiB per workgroup has to be granted per kernel (gfx950 has
// 160 KiB per CU).
static inline hipError_t allow_lds(const void *kernel, size_t bytes)
{
	if (bytes <= 48 * 1024)
		return hipSuccess;
	return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

} // namespace dwt
