// dwt_strided.hip -- the strided gather / scatter of the path ON THE DEVICE.
//
// libdwt's transforms take any element stride (`stride_y`): every line goes through
// dwt_util_memcpy_stride_s / _i (src/system.c:102-164) into a dense temp line and back, and the
// reference's OpenCV wrapper relies on it to transform one channel of an interleaved multi-channel
// image (ptr = data + elemSize1*channel, stride_y = elemSize: src/cvdwt.cpp:98-135).  The fused
// sweeps want dense rows, so a device image whose elements are not adjacent is packed into a dense
// image by k_strided_pack, transformed there, and spread back by k_strided_unpack -- which writes
// the image's own elements only: the other channels and the pitch padding are never touched.
//
// One thread per element, x fastest: a wave reads 64 elements `stride_y` bytes apart (three
// channels of floats: one 768-byte run) and writes 256 contiguous bytes, or the reverse.  Elements
// that are not naturally aligned (odd byte strides or base) move byte by byte.
#include "dwt_kernels.h"

namespace dwt {

template <int ES, bool ALIGNED>
static __device__ __forceinline__ void move_elem(char *d, const char *s)
{
	if constexpr (ALIGNED) {
		if constexpr (ES == 4)
			*(unsigned *)d = *(const unsigned *)s;
		else
			*(unsigned long long *)d = *(const unsigned long long *)s;
	} else {
#pragma unroll
		for (int b = 0; b < ES; b++)
			d[b] = s[b];
	}
}

// dense[y][x] = strided[y*sx + x*sy]   (PACK)   or the reverse (!PACK)
template <int ES, bool ALIGNED, bool PACK>
__global__ __launch_bounds__(256) void k_strided_move(char *__restrict__ dense, long pitch, char *__restrict__ strided, long sx, long sy, int w, int h)
{
	const int x = blockIdx.x * 256 + threadIdx.x;
	if (x >= w)
		return;
	// a few rows per workgroup: the launch stays small for tall images
	for (int y = blockIdx.y; y < h; y += gridDim.y) {
		char *d = dense + (long)y * pitch + (long)x * ES;
		char *s = strided + (long)y * sx + (long)x * sy;
		if constexpr (PACK)
			move_elem<ES, ALIGNED>(d, s);
		else
			move_elem<ES, ALIGNED>(s, d);
	}
}

template <int ES, bool PACK>
static hipError_t strided_move_t(void *dense, long pitch, void *strided, long sx, long sy, int w, int h, hipStream_t st)
{
	if (w <= 0 || h <= 0)
		return hipSuccess;
	const bool aligned = ((uintptr_t)strided | (uintptr_t)sx | (uintptr_t)sy) % ES == 0;
	dim3 grid((w + 255) / 256, h < 16384 ? h : 16384);
	if (aligned)
		k_strided_move<ES, true, PACK><<<grid, 256, 0, st>>>((char *)dense, pitch, (char *)strided, sx, sy, w, h);
	else
		k_strided_move<ES, false, PACK><<<grid, 256, 0, st>>>((char *)dense, pitch, (char *)strided, sx, sy, w, h);
	return hipGetLastError();
}

hipError_t launch_strided_pack(void *dense, long pitch, const void *strided, long sx, long sy, int es, int w, int h, hipStream_t st)
{
	return es == 8 ? strided_move_t<8, true>(dense, pitch, (void *)strided, sx, sy, w, h, st)
	               : strided_move_t<4, true>(dense, pitch, (void *)strided, sx, sy, w, h, st);
}

hipError_t launch_strided_unpack(void *strided, long sx, long sy, const void *dense, long pitch, int es, int w, int h, hipStream_t st)
{
	return es == 8 ? strided_move_t<8, false>((void *)dense, pitch, strided, sx, sy, w, h, st)
	               : strided_move_t<4, false>((void *)dense, pitch, strided, sx, sy, w, h, st);
}

} // namespace dwt
