/*
 * libdwt.h -- drop-in declarations for the hot-path subset of libdwt's C API, served
 * by the MI355X backend (libdwt_amd/libdwt_hip.so).
 *
 * Written fresh: only the prototypes (names, argument order and meaning, in/out
 * conventions, error behaviour) follow the reference so that a program written
 * against xbarin02/libdwt -- e.g. its examples/simple/simple.c and
 * examples/simple-int/simple.c -- compiles and links unchanged.  Each block cites
 * the reference declaration it replaces (paths relative to the reference tree).
 *
 * Image addressing everywhere: element (y,x) lives at
 *     (char *)ptr + y*stride_x + x*stride_y
 * i.e. stride_x is the ROW pitch and stride_y the ELEMENT pitch, both in bytes
 * (src/inline.h:180-189).  "size_o_big" is the outer allocated frame, "size_i_big"
 * the inner image nested at its origin.
 *
 * `ptr` may be host memory (any strides; staged through HBM) or device memory
 * (see libdwt_hip.h).  Failures are logged and abort(), as dwt_util_error does in
 * the reference (src/libdwt.c:20410-20421); nothing falls back to a CPU path.
 */
#ifndef LIBDWT_H
#define LIBDWT_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- 2-D multi-level transforms, Mallat layout ---------------------------------- */

/* Forward float CDF 9/7, in place.  *j_max_ptr: requested levels in, levels done out;
 * negative or too large means "as many as possible".  src/libdwt.h:562-573. */
void dwt_cdf97_2f_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);

/* Inverse float CDF 9/7, in place.  j_max < 0 undoes a full decomposition.
 * src/libdwt.h:867-878. */
void dwt_cdf97_2i_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);

/* Out-of-place variants: first level reads src, everything lands in dst.
 * src/libdwt.h:667-679, 962-974. */
void dwt_cdf97_2f_s2(const void *src, void *dst, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf97_2i_s2(const void *src, void *dst, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);

/* Reversible int32 CDF 5/3.  src/libdwt.h:686, 981. */
void dwt_cdf53_2f_i(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf53_2i_i(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);

/* Float CDF 5/3.  src/libdwt.h:722, 1053. */
void dwt_cdf53_2f_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf53_2i_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);

/* Fixed-point int32 CDF 9/7.  src/libdwt.h:704, 999. */
void dwt_cdf97_2f_i(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf97_2i_i(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);

/* Double precision CDF 9/7 and 5/3 (elements of 8 bytes).  src/libdwt.h:526, 831, 544, 849. */
void dwt_cdf97_2f_d(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf97_2i_d(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);
void dwt_cdf53_2f_d(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf53_2i_d(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);

/* Interleaved (in-place lifting) layout: no de-interleave, level j works on the
 * stride-2^j lattice of the image.  src/libdwt.h:586, 889, 599, 944.  The 9/7 pair runs its
 * rows and columns in interleaved phases in the reference (src/libdwt.c:12970-13480,
 * 17517-17594), which rounds differently from rows-then-columns in the top 8 rows and the last
 * 5 columns of a level.  Bit-identical to the reference BY DEFAULT: the fused sweep's two border
 * strips are recomputed in the reference's phase order (k_il_strip); dwt_util_set_accel(1) runs
 * that order pass by pass over the whole image (the cross-check).  The 5/3 pair is bit-identical
 * either way. */
void dwt_cdf97_2f_inplace_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf97_2i_inplace_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);
void dwt_cdf53_2f_inplace_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf53_2i_inplace_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);
/* level count only, no transform (src/libdwt.h:779, src/libdwt.c:16780) */
void dwt_cdf53_2f_dummy_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one);
/* fixed-point int32 CDF 9/7, interleaved in place (src/libdwt.h:1035, 1017).  As in the reference
 * the strides are not scaled per level: above one level the dense top-left block is transformed
 * again (src/libdwt.c:17423 "tested only with j=1"); bit-identical, forward + inverse restore
 * the image exactly for any level count. */
void dwt_cdf97_2f_inplace_i(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf97_2i_inplace_i(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);
/* the reference's other CPU schedules of the same forward transform (identical bits):
 * src/libdwt.h:612, 625, 649 */
void dwt_cdf97_2f_inplace_sep_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf97_2f_inplace_sep_sdl_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void dwt_cdf97_2f_inplace_sdl_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);

/* ---- lifecycle and backend knobs (src/libdwt.h:1667-1745, 1974-1986) -------------- */
void dwt_util_init(void);   /* brings the device up (the reference loads BCE firmware here) */
void dwt_util_finish(void); /* releases device workspace */
void dwt_util_abort(void);

/* Acceleration selector.  The reference's values 0..16 choose among CPU loop
 * schedules that all give the same coefficients (src/libdwt.h:1703-1720); here 0
 * (default) runs the fused tile-sweep kernels and 1 the exact line-pass kernels;
 * any other value is accepted and treated as 0. */
void dwt_util_set_accel(int accel_type);
int dwt_util_get_accel(void);
/* Accepted for source compatibility; the device schedules its own waves. */
void dwt_util_set_num_threads(int num_threads);
int dwt_util_get_num_threads(void);
int dwt_util_get_max_threads(void);
void dwt_util_set_num_workers(int num_workers);
int dwt_util_get_num_workers(void);

/* ---- image helpers used by the examples (host memory) ---------------------------- */
/* src/libdwt.h:2231, 2242 (next prime >= min_stride on x86_64; eight layouts) */
int dwt_util_get_opt_stride(int min_stride);
int dwt_util_get_stride(int min_stride, int opt);
/* src/libdwt.h:2847 */
size_t dwt_util_image_size(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y);
/* src/libdwt.c:1437, 1482 */
void dwt_util_alloc_image(void **pptr, int stride_x, int stride_y, int size_o_big_x, int size_o_big_y);
void dwt_util_free_image(void **pptr);
/* synthetic test patterns, src/libdwt.c:1338, 1270 */
void dwt_util_test_image_fill_s(void *ptr, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y, int rand);
void dwt_util_test_image_fill_i(void *ptr, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y, int rand);
/* the other synthetic patterns, selected by `type` (src/libdwt.h:1438, 1421; src/libdwt.c:1201-1244) */
void dwt_util_test_image_fill2_s(void *ptr, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y, int rand, int type);
void dwt_util_test_image_fill2_i(void *ptr, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y, int rand, int type);
/* src/libdwt.c:21154, 21235 */
void dwt_util_copy_s(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y);
void dwt_util_copy_i(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y);
/* returns 0 when equal: float within 1e-3 absolute and finite, int exactly
 * (src/libdwt.c:1593-1620, 1531-1558) */
int dwt_util_compare_s(void *ptr1, void *ptr2, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y);
int dwt_util_compare_i(void *ptr1, void *ptr2, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y);
/* log-magnitude view of a transform, src/libdwt.c:21075, 21020 */
void dwt_util_conv_show_s(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y);
void dwt_util_conv_show_i(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y);
/* double-precision twins (examples/simple-double): pattern with 0-based x, y
 * (src/libdwt.c:1112-1125), copy, compare within 1e-6 absolute, view, PGM writer */
void dwt_util_test_image_fill_d(void *ptr, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y, int rand);
void dwt_util_copy_d(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y);
int dwt_util_compare_d(void *ptr1, void *ptr2, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y);
void dwt_util_conv_show_d(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y);
int dwt_util_save_to_pgm_d(const char *filename, double max_value, const void *ptr, int stride_x, int stride_y,
	int size_i_big_x, int size_i_big_y);
/* ASCII PGM writers, src/libdwt.h:1755, 1783; return 0 on success */
int dwt_util_save_to_pgm_s(const char *filename, float max_value, const void *ptr, int stride_x, int stride_y,
	int size_i_big_x, int size_i_big_y);
int dwt_util_save_to_pgm_i(const char *filename, int max_value, const void *ptr, int stride_x, int stride_y,
	int size_i_big_x, int size_i_big_y);
/* src/libdwt.h:1799 (src/libdwt.c:19727-19792): log(1 + |x|) of every sample (dwt_util_conv_show_s), scaled
 * to the largest of them, as an ASCII PGM; returns 0 */
int dwt_util_save_log_to_pgm_s(const char *path, const void *ptr, int stride_x, int stride_y, int size_x, int size_y);
/* ASCII PGM readers: allocate the image with the optimal stride (src/libdwt.h:1894, 1926);
 * 0 on success, 1 open, 2 header, 3 depth, 4 data, 5 sample out of range */
int dwt_util_load_from_pgm_s(const char *filename, float max_value, void **pptr, int *pstride_x, int *pstride_y,
	int *psize_x, int *psize_y);
int dwt_util_load_from_pgm_i(const char *filename, int max_value, void **pptr, int *pstride_x, int *pstride_y,
	int *psize_x, int *psize_y);
/* text matrices ("MAT": one row per line, comma separated), src/libdwt.h:1829, 1909, 1951 */
int dwt_util_save_to_mat_s(const char *path, const void *ptr, int size_x, int size_y, int stride_x, int stride_y);
int dwt_util_load_from_mat_s(const char *path, void **ptr, int *size_x, int *size_y, int *stride_x, int *stride_y);
int dwt_util_load_from_mat_i(const char *path, void **ptr, int *size_x, int *size_y, int *stride_x, int *stride_y);

/* ---- subband addressing (src/libdwt.h:2276-2330, src/libdwt.c:20731-20950) -------- */
enum dwt_subbands { DWT_LL, DWT_HL, DWT_LH, DWT_HH };
/* Address and size of a subband after j_max levels.  Pure address arithmetic: valid
 * for host and for device images. */
void dwt_util_subband(void *ptr, int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, enum dwt_subbands band,
	void **dst_ptr, int *dst_size_x, int *dst_size_y);
void dwt_util_subband_s(void *ptr, int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, enum dwt_subbands band,
	void **dst_ptr, int *dst_size_x, int *dst_size_y);
void dwt_util_subband_i(void *ptr, int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, enum dwt_subbands band,
	void **dst_ptr, int *dst_size_x, int *dst_size_y);
float *dwt_util_addr_coeff_s(void *ptr, int y, int x, int stride_x, int stride_y); /* src/libdwt.c:1064 */
int *dwt_util_addr_coeff_i(void *ptr, int y, int x, int stride_x, int stride_y);

/* ---- measurement and self-test helpers (src/libdwt.h:2618-2760; src/libdwt.c:21262,
 * 21391, 22296, 22559, 23788, 23877, 24163, 24203) ---------------------------------- */
enum dwt_array { DWT_ARR_SIMPLE, DWT_ARR_SPARSE, DWT_ARR_PACKED };
int dwt_util_pow2_ceil_log2(int x);
void dwt_util_get_sizes_s(enum dwt_array array_type, int size_x, int size_y, int opt_stride,
	int *stride_x, int *stride_y, int *size_o_big_x, int *size_o_big_y, int *size_i_big_x, int *size_i_big_y);
void dwt_util_get_sizes_i(enum dwt_array array_type, int size_x, int size_y, int opt_stride,
	int *stride_x, int *stride_y, int *size_o_big_x, int *size_o_big_y, int *size_i_big_x, int *size_i_big_y);
/* M transforms per loop, minimum over N loops, seconds per transform; host images (the
 * call is timed as a drop-in user sees it, staging included) */
void dwt_util_perf_cdf97_2_s(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs);
void dwt_util_perf_cdf53_2_i(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs);
/* round-trip self-tests: 0 = success */
int dwt_util_test_cdf97_2_s(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding);
int dwt_util_test_cdf97_2_s2(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding);
int dwt_util_test2_cdf97_2_s(enum dwt_array array_type, int size_x, int size_y, int opt_stride, int j_max, int decompose_one);
int dwt_util_test2_cdf97_2_s2(enum dwt_array array_type, int size_x, int size_y, int opt_stride, int j_max, int decompose_one);
/* the same self-tests for the double and the fixed-point int 9/7 drivers (src/libdwt.h:2688,
 * 2703, 2727, 2739; examples/test/test.c:61-73) */
void dwt_util_get_sizes_d(enum dwt_array array_type, int size_x, int size_y, int opt_stride,
	int *stride_x, int *stride_y, int *size_o_big_x, int *size_o_big_y, int *size_i_big_x, int *size_i_big_y);
int dwt_util_test_cdf97_2_d(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding);
int dwt_util_test_cdf97_2_i(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding);
int dwt_util_test2_cdf97_2_d(enum dwt_array array_type, int size_x, int size_y, int opt_stride, int j_max, int decompose_one);
int dwt_util_test2_cdf97_2_i(enum dwt_array array_type, int size_x, int size_y, int opt_stride, int j_max, int decompose_one);

/* ---- timers (src/libdwt.h:1589-1658) --------------------------------------------- */
enum dwt_timer_types {
	DWT_TIME_CLOCK_GETTIME,
	DWT_TIME_CLOCK_GETTIME_REALTIME,
	DWT_TIME_CLOCK_GETTIME_MONOTONIC,
	DWT_TIME_CLOCK_GETTIME_MONOTONIC_RAW,
	DWT_TIME_CLOCK_GETTIME_PROCESS_CPUTIME_ID,
	DWT_TIME_CLOCK_GETTIME_THREAD_CPUTIME_ID,
	DWT_TIME_CLOCK,
	DWT_TIME_TIMES,
	DWT_TIME_GETRUSAGE,
	DWT_TIME_GETRUSAGE_SELF,
	DWT_TIME_GETRUSAGE_CHILDREN,
	DWT_TIME_GETRUSAGE_THREAD,
	DWT_TIME_GETTIMEOFDAY,
	DWT_TIME_IOCTL_RTC,
	DWT_TIME_AUTOSELECT
};
typedef int64_t dwt_clock_t;
int dwt_util_clock_available(int type);
int dwt_util_clock_autoselect(void);
dwt_clock_t dwt_util_get_frequency(int type);
dwt_clock_t dwt_util_get_clock(int type);

/* ---- logging and identification (src/libdwt.h:2154-2224, 1577-1582) --------------- */
enum dwt_util_loglevel { LOG_NONE = 0, LOG_DBG, LOG_INFO, LOG_WARN, LOG_ERR, LOG_TEST };
int dwt_util_log(enum dwt_util_loglevel level, const char *format, ...);
void dwt_util_error(const char *format, ...);
const char *dwt_util_version(void);
const char *dwt_util_arch(void);
const char *dwt_util_node(void);
const char *dwt_util_appname(void);

/* size sweep writing "pixels<TAB>seconds per pixel" rows (src/libdwt.c:22559) */
void dwt_util_measure_perf_cdf97_2_s(enum dwt_array array_type, int min_x, int max_x, int opt_stride,
	int j_max, int decompose_one, int zero_padding, int M, int N, int clock_type,
	FILE *fwd_plot_data, FILE *inv_plot_data);
/* the same protocols for the interleaved-layout entries (src/libdwt.h:2520, 2537, 2554, 2576,
 * 2744, 2764, 2784, 2804) */
void dwt_util_perf_cdf97_2_inplace_s(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs);
void dwt_util_measure_perf_cdf97_2_inplace_s(enum dwt_array array_type, int min_x, int max_x, int opt_stride,
	int j_max, int decompose_one, int zero_padding, int M, int N, int clock_type,
	FILE *fwd_plot_data, FILE *inv_plot_data);
void dwt_util_perf_cdf97_2_inplace_sep_s(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs);
void dwt_util_measure_perf_cdf97_2_inplace_sep_s(enum dwt_array array_type, int min_x, int max_x, int opt_stride,
	int j_max, int decompose_one, int zero_padding, int M, int N, int clock_type,
	FILE *fwd_plot_data, FILE *inv_plot_data);
void dwt_util_perf_cdf97_2_inplace_sep_sdl_s(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs);
void dwt_util_measure_perf_cdf97_2_inplace_sep_sdl_s(enum dwt_array array_type, int min_x, int max_x, int opt_stride,
	int j_max, int decompose_one, int zero_padding, int M, int N, int clock_type,
	FILE *fwd_plot_data, FILE *inv_plot_data);
void dwt_util_perf_cdf97_2_inplace_sdl_s(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs);
void dwt_util_measure_perf_cdf97_2_inplace_sdl_s(enum dwt_array array_type, int min_x, int max_x, int opt_stride,
	int j_max, int decompose_one, int zero_padding, int M, int N, int clock_type,
	FILE *fwd_plot_data, FILE *inv_plot_data);

#ifdef __cplusplus
}
#endif
#endif
