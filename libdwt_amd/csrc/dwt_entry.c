/*
 * dwt_entry.c -- libdwt's 2-D entry points and lifecycle hooks as thin C wrappers
 * over the device backend (include/libdwt_hip.h).  Host code in C; all compute is in
 * the HIP kernels.  There is no CPU path here: a call that cannot run on the device
 * logs the reason and aborts, as dwt_util_error does in the reference
 * (src/libdwt.c:20410-20421).
 */
#include "../../include/libdwt.h"
#include "../../include/libdwt_hip.h"
#include "../../include/dwt-simple.h"

#include <stdlib.h>

static int g_accel = 0;
static int g_threads = 1;
static int g_workers = 1;

static void run(int wavelet, int inverse, const void *src, void *dst, int stride_x, int stride_y,
	int sox, int soy, int six, int siy, int *j, int decompose_one, int zero_padding, const char *who)
{
	if (dwt_hip_transform2d(wavelet, inverse, src, dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, zero_padding))
		dwt_util_error("%s: %s\n", who, dwt_hip_last_error());
}

/* src/libdwt.c:12776 */
void dwt_cdf97_2f_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF97_S, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding, __func__);
}

/* src/libdwt.c:17040 */
void dwt_cdf97_2i_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF97_S, 1, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, &j_max, decompose_one, zero_padding, __func__);
}

/* src/libdwt.c:12619 */
void dwt_cdf97_2f_s2(const void *src, void *dst, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF97_S, 0, src, dst, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding, __func__);
}

/* src/libdwt.c:17985 */
void dwt_cdf97_2i_s2(const void *src, void *dst, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF97_S, 1, src, dst, stride_x, stride_y, sox, soy, six, siy, &j_max, decompose_one, zero_padding, __func__);
}

/* src/libdwt.c:16304 */
void dwt_cdf53_2f_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF53_I, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding, __func__);
}

/* src/libdwt.c:18142 */
void dwt_cdf53_2i_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF53_I, 1, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, &j_max, decompose_one, zero_padding, __func__);
}

/* src/libdwt.c:16470 */
void dwt_cdf53_2f_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF53_S, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding, __func__);
}

/* src/libdwt.c:18296 */
void dwt_cdf53_2i_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF53_S, 1, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, &j_max, decompose_one, zero_padding, __func__);
}

/* fixed-point int32 CDF 9/7: src/libdwt.c:16387, 18219 */
void dwt_cdf97_2f_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF97_I, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding, __func__);
}

void dwt_cdf97_2i_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF97_I, 1, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, &j_max, decompose_one, zero_padding, __func__);
}

/* double precision: src/libdwt.c:12451, 16884, 12535, 16962 (line-pass kernels) */
void dwt_cdf97_2f_d(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF97_D, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding, __func__);
}

void dwt_cdf97_2i_d(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF97_D, 1, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, &j_max, decompose_one, zero_padding, __func__);
}

void dwt_cdf53_2f_d(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF53_D, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding, __func__);
}

void dwt_cdf53_2i_d(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	run(DWT_HIP_CDF53_D, 1, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, &j_max, decompose_one, zero_padding, __func__);
}

/* interleaved (in-place lifting) layout: src/libdwt.c:12926, 17474, 16553, 17886 */
static void run_il(int wavelet, int inverse, int flavour, void *ptr, int stride_x, int stride_y,
	int sox, int soy, int six, int siy, int *j, int decompose_one, const char *who)
{
	if (dwt_hip_transform2d_interleaved(wavelet, inverse, flavour, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j, decompose_one))
		dwt_util_error("%s: %s\n", who, dwt_hip_last_error());
}

void dwt_cdf97_2f_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	(void)zero_padding; /* unused by the reference too (its padding code is commented out) */
	run_il(DWT_HIP_CDF97_S, 0, 0, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, __func__);
}

void dwt_cdf97_2i_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	run_il(DWT_HIP_CDF97_S, 1, 0, ptr, stride_x, stride_y, sox, soy, six, siy, &j_max, decompose_one, __func__);
}

void dwt_cdf53_2f_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	run_il(DWT_HIP_CDF53_S, 0, 0, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, __func__);
}

void dwt_cdf53_2i_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	run_il(DWT_HIP_CDF53_S, 1, 0, ptr, stride_x, stride_y, sox, soy, six, siy, &j_max, decompose_one, __func__);
}

/* fixed-point int 9/7, interleaved in place: src/libdwt.c:17424, :17308 */
void dwt_cdf97_2f_inplace_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	run_il(DWT_HIP_CDF97_I, 0, 0, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, __func__);
}

void dwt_cdf97_2i_inplace_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	run_il(DWT_HIP_CDF97_I, 1, 0, ptr, stride_x, stride_y, sox, soy, six, siy, &j_max, decompose_one, __func__);
}

/* dwt-simple.h: src/dwt-simple.c:2224 / :1615 / :3034 and :2356 / :1927 / :3166 -- three
 * CPU schedules per wavelet with identical results, one device path here */
#define DWT_NEWAPI(name, wavelet)                                                                      \
	void name(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one) \
	{                                                                                                  \
		run_il(wavelet, 0, 1, ptr, stride_x, stride_y, size_x, size_y, size_x, size_y, j_max_ptr, decompose_one, __func__); \
	}
DWT_NEWAPI(fdwt2_cdf97_horizontal_s, DWT_HIP_CDF97_S)
DWT_NEWAPI(fdwt2_cdf97_vertical_s, DWT_HIP_CDF97_S)
DWT_NEWAPI(fdwt2_cdf97_diagonal_s, DWT_HIP_CDF97_S)
DWT_NEWAPI(fdwt2_cdf53_horizontal_s, DWT_HIP_CDF53_S)
DWT_NEWAPI(fdwt2_cdf53_vertical_s, DWT_HIP_CDF53_S)
DWT_NEWAPI(fdwt2_cdf53_diagonal_s, DWT_HIP_CDF53_S)
/* one direction only, every level (src/dwt-simple.c:1747, :1837) */
void fdwt2h1_cdf97_vertical_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one)
{
	run_il(DWT_HIP_CDF97_S, 0, 2, ptr, stride_x, stride_y, size_x, size_y, size_x, size_y, j_max_ptr, decompose_one, __func__);
}

void fdwt2v1_cdf97_vertical_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one)
{
	run_il(DWT_HIP_CDF97_S, 0, 3, ptr, stride_x, stride_y, size_x, size_y, size_x, size_y, j_max_ptr, decompose_one, __func__);
}
#undef DWT_NEWAPI

/* 1-D: a line is a one-row image (src/dwt-simple.c:2059, 2118, 2166, 2195) */
void fdwt1_cdf97_horizontal_s(void *ptr, int size, int stride, int *j_max_ptr)
{
	run_il(DWT_HIP_CDF97_S, 0, 1, ptr, stride * size, stride, size, 1, size, 1, j_max_ptr, 1, __func__);
}

void fdwt1_single_cdf97_horizontal_s(void *ptr, int size, int stride)
{
	int j = 1;
	run_il(DWT_HIP_CDF97_S, 0, 1, ptr, stride * size, stride, size, 1, size, 1, &j, 1, __func__);
}

void fdwt1_single_cdf97_horizontal_min5_s(void *ptr, int size, int stride)
{
	fdwt1_single_cdf97_horizontal_s(ptr, size, stride);
}

void fdwt1_single_cdf97_vertical_min5_s(void *ptr, int size, int stride)
{
	fdwt1_single_cdf97_horizontal_s(ptr, size, stride);
}

/* src/libdwt.c:19158: platform bring-up.  The reference loads accelerator firmware
 * on ASVP and does nothing on x86; here the device context is created. */
void dwt_util_init(void)
{
	if (dwt_hip_init())
		dwt_util_error("dwt_util_init: %s\n", dwt_hip_last_error());
}

/* src/libdwt.c:19186 */
void dwt_util_finish(void)
{
	dwt_hip_finish();
}

/* src/libdwt.c:19200 */
void dwt_util_abort(void)
{
	abort();
}

/* src/libdwt.c:19946 / libdwt.h:1703-1720 */
void dwt_util_set_accel(int accel_type)
{
	g_accel = accel_type;
	dwt_hip_set_option("generic", accel_type == 1);
}

int dwt_util_get_accel(void)
{
	return g_accel;
}

/* src/libdwt.c:19116-19156: OpenMP threads / SIMD-BCE workers of the CPU schedules.
 * Stored and reported back so that callers' bookkeeping keeps working. */
void dwt_util_set_num_threads(int num_threads)
{
	if (num_threads > 0)
		g_threads = num_threads;
}

int dwt_util_get_num_threads(void)
{
	return g_threads;
}

int dwt_util_get_max_threads(void)
{
	return 1;
}

void dwt_util_set_num_workers(int num_workers)
{
	if (num_workers > 0)
		g_workers = num_workers;
}

int dwt_util_get_num_workers(void)
{
	return g_workers;
}
