/*
 * dwt-simple.h -- the 2-D entry points of libdwt's "new API" (src/dwt-simple.h) on the
 * MI355X backend: forward transforms in the interleaved (in-place lifting) layout.
 *
 * Drop-in for the declarations at src/dwt-simple.h:68 (fdwt2_cdf97_horizontal_s), :116
 * (_vertical_), :161 (_diagonal_), :289 (fdwt2_cdf53_horizontal_s), :390 (_vertical_) and
 * :413 (_diagonal_).  In the reference the three names of a wavelet are three CPU loop
 * schedules ("vectorisations") that give identical bits; here they are one device path.
 * Results equal the reference's within fp32 rounding (<= 1e-5 relative): the reference
 * interleaves row and column work in phases, the device finishes the rows first, which
 * changes the rounding in the 8-sample border bands only; after dwt_util_set_accel(1) the
 * device follows the reference's phase order and the results are bit-identical.
 *
 * `ptr` may be a host pointer (staged through HBM) or a device pointer (stride_y == 4).
 * Element (y, x) lives at ptr + y*stride_x + x*stride_y (bytes).  `*j_max_ptr` < 0 or
 * beyond the limit asks for the full depth and receives the level count.  The inverse is
 * libdwt.h's dwt_cdf97_2i_inplace_s / dwt_cdf53_2i_inplace_s, as in
 * examples/simple-newapi/simple.c.
 */
#ifndef DWT_SIMPLE_H
#define DWT_SIMPLE_H

#ifdef __cplusplus
extern "C" {
#endif

void fdwt2_cdf97_horizontal_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one);
void fdwt2_cdf97_vertical_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one);
void fdwt2_cdf97_diagonal_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one);
void fdwt2_cdf53_horizontal_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one);
void fdwt2_cdf53_vertical_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one);
void fdwt2_cdf53_diagonal_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one);

/* src/dwt-simple.h:127, 138: the rows (h1) or the columns (v1) of every level only; bit-identical */
void fdwt2h1_cdf97_vertical_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one);
void fdwt2v1_cdf97_vertical_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one);

/* The complete 1-D transforms of the header (src/dwt-simple.h:78-100; src/dwt-simple.c:2059,
 * 2118, 2166, 2195): one strided line, interleaved in place; `stride` is the element pitch in
 * bytes.  They run through the same device path as a one-row image (bit-identical); kept for
 * source compatibility -- a single line is PCIe-latency bound, not a GPU workload.  The partial
 * building blocks of the header (fdwt_cdf97_horizontal_s etc., which are cores without their
 * prolog / epilog) and the EAW entries are not provided. */
void fdwt1_cdf97_horizontal_s(void *ptr, int size, int stride, int *j_max_ptr);
void fdwt1_single_cdf97_horizontal_s(void *ptr, int size, int stride);
void fdwt1_single_cdf97_horizontal_min5_s(void *ptr, int size, int stride);
void fdwt1_single_cdf97_vertical_min5_s(void *ptr, int size, int stride);

#ifdef __cplusplus
}
#endif

#endif
