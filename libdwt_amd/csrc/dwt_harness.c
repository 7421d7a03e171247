/*
 * dwt_harness.c -- libdwt's measurement / self-test / subband-addressing helpers for the
 * 2-D path, host code in C over the same entry points a user
 * program calls, so examples/simple-perf, examples/perf-plot and examples/subbands run
 * against the MI355X backend.  Behaviour restated from the cited reference lines.
 *
 * The perf helpers keep the reference's protocol (M transforms per loop, min over N
 * loops, src/libdwt.c:21444-21476) on HOST images, i.e. they time the drop-in call
 * including the H2D/D2H staging.  dwt_hip_perf_cdf97_2_s is the same protocol with the
 * images resident in HBM.
 */
#include "../../include/libdwt.h"
#include "../../include/libdwt_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

/* smallest power of two >= x (src/inline.h:401-419, exported at src/libdwt.c:786) */
int dwt_util_pow2_ceil_log2(int x)
{
	int p = 1;
	while (p < x)
		p <<= 1;
	return p;
}

static void get_sizes(int elem, enum dwt_array array_type, int size_x, int size_y, int opt_stride,
	int *stride_x, int *stride_y, int *sox, int *soy, int *six, int *siy)
{
	*stride_y = elem;
	*stride_x = dwt_util_get_stride(elem * dwt_util_pow2_ceil_log2(size_x), opt_stride);
	*sox = *six = size_x;
	*soy = *siy = size_y;
	if (array_type == DWT_ARR_SPARSE || array_type == DWT_ARR_SIMPLE) {
		*sox = dwt_util_pow2_ceil_log2(size_x);
		*soy = dwt_util_pow2_ceil_log2(size_y);
	}
}

/* src/libdwt.c:22296-22337 */
void dwt_util_get_sizes_s(enum dwt_array array_type, int size_x, int size_y, int opt_stride,
	int *stride_x, int *stride_y, int *size_o_big_x, int *size_o_big_y, int *size_i_big_x, int *size_i_big_y)
{
	get_sizes(sizeof(float), array_type, size_x, size_y, opt_stride, stride_x, stride_y,
		size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y);
}

void dwt_util_get_sizes_i(enum dwt_array array_type, int size_x, int size_y, int opt_stride,
	int *stride_x, int *stride_y, int *size_o_big_x, int *size_o_big_y, int *size_i_big_x, int *size_i_big_y)
{
	get_sizes(sizeof(int), array_type, size_x, size_y, opt_stride, stride_x, stride_y,
		size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y);
}

void dwt_util_get_sizes_d(enum dwt_array array_type, int size_x, int size_y, int opt_stride,
	int *stride_x, int *stride_y, int *size_o_big_x, int *size_o_big_y, int *size_i_big_x, int *size_i_big_y)
{
	get_sizes(sizeof(double), array_type, size_x, size_y, opt_stride, stride_x, stride_y,
		size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y);
}

typedef void (*fwd_fn)(void *, int, int, int, int, int, int, int *, int, int);
typedef void (*inv_fn)(void *, int, int, int, int, int, int, int, int, int);

/* the perf protocol shared by the float 9/7 and the int 5/3 helper */
static void perf_2d(fwd_fn fwd, inv_fn inv, int is_int, int device, int stride_x, int stride_y,
	int sox, int soy, int six, int siy, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs)
{
	void **ptr = malloc(sizeof(void *) * (size_t)M);
	int *j = malloc(sizeof(int) * (size_t)M);
	void *host = NULL;
	const size_t bytes = dwt_util_image_size(stride_x, stride_y, sox, soy);
	if (!ptr || !j)
		dwt_util_error("%s: out of memory\n", __func__);
	dwt_util_alloc_image(&host, stride_x, stride_y, sox, soy);
	if (is_int)
		dwt_util_test_image_fill_i(host, stride_x, stride_y, six, siy, 0);
	else
		dwt_util_test_image_fill_s(host, stride_x, stride_y, six, siy, 0);
	for (int m = 0; m < M; m++) {
		j[m] = j_max;
		if (device) {
			ptr[m] = dwt_hip_malloc(bytes);
			if (!ptr[m] || dwt_hip_memcpy_h2d(ptr[m], host, bytes))
				dwt_util_error("%s: %s\n", __func__, dwt_hip_last_error());
		} else {
			dwt_util_alloc_image(&ptr[m], stride_x, stride_y, sox, soy);
			dwt_util_copy_s(host, ptr[m], stride_x, stride_y, six, siy);
		}
	}
	*fwd_secs = INFINITY;
	*inv_secs = INFINITY;
	for (int n = 0; n < N; n++) {
		dwt_hip_sync();
		const dwt_clock_t f0 = dwt_util_get_clock(clock_type);
		for (int m = 0; m < M; m++)
			fwd(ptr[m], stride_x, stride_y, sox, soy, six, siy, &j[m], decompose_one, zero_padding);
		dwt_hip_sync();
		const dwt_clock_t f1 = dwt_util_get_clock(clock_type);
		const float fs = (float)(f1 - f0) / M / dwt_util_get_frequency(clock_type);
		if (fs < *fwd_secs)
			*fwd_secs = fs;
		const dwt_clock_t i0 = dwt_util_get_clock(clock_type);
		for (int m = 0; m < M; m++)
			inv(ptr[m], stride_x, stride_y, sox, soy, six, siy, j[m], decompose_one, zero_padding);
		dwt_hip_sync();
		const dwt_clock_t i1 = dwt_util_get_clock(clock_type);
		const float is = (float)(i1 - i0) / M / dwt_util_get_frequency(clock_type);
		if (is < *inv_secs)
			*inv_secs = is;
	}
	for (int m = 0; m < M; m++) {
		if (device)
			dwt_hip_free(ptr[m]);
		else
			dwt_util_free_image(&ptr[m]);
	}
	dwt_util_free_image(&host);
	free(ptr);
	free(j);
}

/* src/libdwt.c:21391-21517 */
void dwt_util_perf_cdf97_2_s(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs)
{
	perf_2d(dwt_cdf97_2f_s, dwt_cdf97_2i_s, 0, 0, stride_x, stride_y, size_o_big_x, size_o_big_y,
		size_i_big_x, size_i_big_y, j_max, decompose_one, zero_padding, M, N, clock_type, fwd_secs, inv_secs);
}

/* src/libdwt.c:21262-21389 */
void dwt_util_perf_cdf53_2_i(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs)
{
	perf_2d(dwt_cdf53_2f_i, dwt_cdf53_2i_i, 1, 0, stride_x, stride_y, size_o_big_x, size_o_big_y,
		size_i_big_x, size_i_big_y, j_max, decompose_one, zero_padding, M, N, clock_type, fwd_secs, inv_secs);
}

/* same protocol, images resident in HBM (stride_y == 4, stride_x % 4 == 0) */
void dwt_hip_perf_cdf97_2_s(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs)
{
	perf_2d(dwt_cdf97_2f_s, dwt_cdf97_2i_s, 0, 1, stride_x, stride_y, size_o_big_x, size_o_big_y,
		size_i_big_x, size_i_big_y, j_max, decompose_one, zero_padding, M, N, clock_type, fwd_secs, inv_secs);
}

typedef void (*perf_fn)(int, int, int, int, int, int, int, int, int, int, int, int, float *, float *);

/* src/libdwt.c:22559-22645: size sweep x = min_x, ceil(x*1.13), ... writing
 * "pixels <TAB> seconds per pixel" (MEASURE_PER_PIXEL is defined at :8) */
static void measure_perf(perf_fn perf, enum dwt_array array_type, int min_x, int max_x, int opt_stride,
	int j_max, int decompose_one, int zero_padding, int M, int N, int clock_type,
	FILE *fwd_plot_data, FILE *inv_plot_data)
{
	const float growth_factor = 1.13f; /* g_growth_factor_s, :22385 */
	for (int x = min_x; x <= max_x; x = (int)ceilf(x * growth_factor)) {
		const int y = x;
		int stride_x, stride_y, sox, soy, six, siy;
		dwt_util_get_sizes_s(array_type, x, y, opt_stride, &stride_x, &stride_y, &sox, &soy, &six, &siy);
		dwt_util_log(LOG_DBG, "performance test for [%ix%i] in [%ix%i] with strides (%i, %i)...\n",
			six, siy, sox, soy, stride_x, stride_y);
		float fwd_secs, inv_secs;
		perf(stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, zero_padding,
			M, N, clock_type, &fwd_secs, &inv_secs);
		const int denominator = x * y;
		fprintf(fwd_plot_data, "%i\t%.10f\n", x * y, fwd_secs / denominator);
		fprintf(inv_plot_data, "%i\t%.10f\n", x * y, inv_secs / denominator);
	}
}

void dwt_util_measure_perf_cdf97_2_s(enum dwt_array array_type, int min_x, int max_x, int opt_stride,
	int j_max, int decompose_one, int zero_padding, int M, int N, int clock_type,
	FILE *fwd_plot_data, FILE *inv_plot_data)
{
	measure_perf(dwt_util_perf_cdf97_2_s, array_type, min_x, max_x, opt_stride, j_max, decompose_one, zero_padding,
		M, N, clock_type, fwd_plot_data, inv_plot_data);
}

/* Interleaved (in-place lifting) layout: the reference's four CPU schedules of the forward
 * transform (src/libdwt.c:12926 plain, :13485 _sep_, :13641 _sep_sdl_, :14847 _sdl_) give
 * identical bits; one device path serves them.  Perf helpers: src/libdwt.c:21519-21920,
 * size sweeps :22647-22990. */
#define DWT_INPLACE_VARIANT(suffix)                                                                              \
	void dwt_cdf97_2f_inplace_##suffix(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,  \
		int *j_max_ptr, int decompose_one, int zero_padding)                                                      \
	{                                                                                                            \
		dwt_cdf97_2f_inplace_s(ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding); \
	}
DWT_INPLACE_VARIANT(sep_s)
DWT_INPLACE_VARIANT(sep_sdl_s)
DWT_INPLACE_VARIANT(sdl_s)
#undef DWT_INPLACE_VARIANT

#define DWT_INPLACE_PERF(suffix)                                                                                 \
	void dwt_util_perf_cdf97_2_inplace_##suffix(int stride_x, int stride_y, int sox, int soy, int six, int siy,   \
		int j_max, int decompose_one, int zero_padding, int M, int N, int clock_type, float *fwd_secs, float *inv_secs) \
	{                                                                                                            \
		perf_2d(dwt_cdf97_2f_inplace_s, dwt_cdf97_2i_inplace_s, 0, 0, stride_x, stride_y, sox, soy, six, siy,       \
			j_max, decompose_one, zero_padding, M, N, clock_type, fwd_secs, inv_secs);                               \
	}                                                                                                            \
	void dwt_util_measure_perf_cdf97_2_inplace_##suffix(enum dwt_array array_type, int min_x, int max_x, int opt_stride, \
		int j_max, int decompose_one, int zero_padding, int M, int N, int clock_type, FILE *fwd_plot_data, FILE *inv_plot_data) \
	{                                                                                                            \
		measure_perf(dwt_util_perf_cdf97_2_inplace_##suffix, array_type, min_x, max_x, opt_stride, j_max,          \
			decompose_one, zero_padding, M, N, clock_type, fwd_plot_data, inv_plot_data);                            \
	}
DWT_INPLACE_PERF(s)
DWT_INPLACE_PERF(sep_s)
DWT_INPLACE_PERF(sep_sdl_s)
DWT_INPLACE_PERF(sdl_s)
#undef DWT_INPLACE_PERF

/* src/libdwt.c:23788-23875: fill, forward, inverse, compare; 0 = success */
int dwt_util_test_cdf97_2_s(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding)
{
	int j = j_max;
	void *data, *copy;
	dwt_util_alloc_image(&data, stride_x, stride_y, size_o_big_x, size_o_big_y);
	dwt_util_alloc_image(&copy, stride_x, stride_y, size_o_big_x, size_o_big_y);
	dwt_util_test_image_fill_s(data, stride_x, stride_y, size_i_big_x, size_i_big_y, 0);
	dwt_util_copy_s(data, copy, stride_x, stride_y, size_i_big_x, size_i_big_y);
	dwt_cdf97_2f_s(data, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y, &j, decompose_one, zero_padding);
	dwt_cdf97_2i_s(data, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y, j, decompose_one, zero_padding);
	const int ret = dwt_util_compare_s(data, copy, stride_x, stride_y, size_i_big_x, size_i_big_y) ? 1 : 0;
	dwt_util_free_image(&data);
	dwt_util_free_image(&copy);
	return ret;
}

/* src/libdwt.c:23877-23963: the same through the out-of-place entries */
int dwt_util_test_cdf97_2_s2(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding)
{
	int j = j_max;
	void *data1, *data2, *data3, *copy;
	dwt_util_alloc_image(&data1, stride_x, stride_y, size_o_big_x, size_o_big_y);
	dwt_util_alloc_image(&data2, stride_x, stride_y, size_o_big_x, size_o_big_y);
	dwt_util_alloc_image(&data3, stride_x, stride_y, size_o_big_x, size_o_big_y);
	dwt_util_alloc_image(&copy, stride_x, stride_y, size_o_big_x, size_o_big_y);
	dwt_util_test_image_fill_s(data1, stride_x, stride_y, size_i_big_x, size_i_big_y, 0);
	dwt_util_copy_s(data1, copy, stride_x, stride_y, size_i_big_x, size_i_big_y);
	dwt_cdf97_2f_s2(data1, data2, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y, &j, decompose_one, zero_padding);
	dwt_cdf97_2i_s2(data2, data3, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y, j, decompose_one, zero_padding);
	const int ret = dwt_util_compare_s(data3, copy, stride_x, stride_y, size_i_big_x, size_i_big_y) ? 1 : 0;
	dwt_util_free_image(&data1);
	dwt_util_free_image(&data2);
	dwt_util_free_image(&data3);
	dwt_util_free_image(&copy);
	return ret;
}

/* src/libdwt.c:24163-24201 */
/* the same self-test for the double and the fixed-point int 9/7 drivers
 * (src/libdwt.c:23877-23960 double, :23962-24045 int; examples/test/test.c:61-73) */
int dwt_util_test_cdf97_2_d(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding)
{
	int j = j_max;
	void *data, *copy;
	dwt_util_alloc_image(&data, stride_x, stride_y, size_o_big_x, size_o_big_y);
	dwt_util_alloc_image(&copy, stride_x, stride_y, size_o_big_x, size_o_big_y);
	dwt_util_test_image_fill_d(data, stride_x, stride_y, size_i_big_x, size_i_big_y, 0);
	dwt_util_copy_d(data, copy, stride_x, stride_y, size_i_big_x, size_i_big_y);
	dwt_cdf97_2f_d(data, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y, &j, decompose_one, zero_padding);
	dwt_cdf97_2i_d(data, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y, j, decompose_one, zero_padding);
	const int ret = dwt_util_compare_d(data, copy, stride_x, stride_y, size_i_big_x, size_i_big_y) ? 1 : 0;
	dwt_util_free_image(&data);
	dwt_util_free_image(&copy);
	return ret;
}

int dwt_util_test_cdf97_2_i(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding)
{
	int j = j_max;
	void *data, *copy;
	dwt_util_alloc_image(&data, stride_x, stride_y, size_o_big_x, size_o_big_y);
	dwt_util_alloc_image(&copy, stride_x, stride_y, size_o_big_x, size_o_big_y);
	dwt_util_test_image_fill_i(data, stride_x, stride_y, size_i_big_x, size_i_big_y, 0);
	dwt_util_copy_i(data, copy, stride_x, stride_y, size_i_big_x, size_i_big_y);
	dwt_cdf97_2f_i(data, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y, &j, decompose_one, zero_padding);
	dwt_cdf97_2i_i(data, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y, j, decompose_one, zero_padding);
	const int ret = dwt_util_compare_i(data, copy, stride_x, stride_y, size_i_big_x, size_i_big_y) ? 1 : 0;
	dwt_util_free_image(&data);
	dwt_util_free_image(&copy);
	return ret;
}

int dwt_util_test2_cdf97_2_d(enum dwt_array array_type, int size_x, int size_y, int opt_stride, int j_max, int decompose_one)
{
	int stride_x, stride_y, sox, soy, six, siy;
	dwt_util_get_sizes_d(array_type, size_x, size_y, opt_stride, &stride_x, &stride_y, &sox, &soy, &six, &siy);
	return dwt_util_test_cdf97_2_d(stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, 0);
}

int dwt_util_test2_cdf97_2_i(enum dwt_array array_type, int size_x, int size_y, int opt_stride, int j_max, int decompose_one)
{
	int stride_x, stride_y, sox, soy, six, siy;
	dwt_util_get_sizes_i(array_type, size_x, size_y, opt_stride, &stride_x, &stride_y, &sox, &soy, &six, &siy);
	return dwt_util_test_cdf97_2_i(stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, 0);
}

int dwt_util_test2_cdf97_2_s(enum dwt_array array_type, int size_x, int size_y, int opt_stride, int j_max, int decompose_one)
{
	int stride_x, stride_y, sox, soy, six, siy;
	dwt_util_get_sizes_s(array_type, size_x, size_y, opt_stride, &stride_x, &stride_y, &sox, &soy, &six, &siy);
	return dwt_util_test_cdf97_2_s(stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, 0);
}

/* src/libdwt.c:24203-24241 */
int dwt_util_test2_cdf97_2_s2(enum dwt_array array_type, int size_x, int size_y, int opt_stride, int j_max, int decompose_one)
{
	int stride_x, stride_y, sox, soy, six, siy;
	dwt_util_get_sizes_s(array_type, size_x, size_y, opt_stride, &stride_x, &stride_y, &sox, &soy, &six, &siy);
	return dwt_util_test_cdf97_2_s2(stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, 0);
}

/* src/libdwt.c:20731-20790: address and size of subband `band` after j_max levels.
 * Pure address arithmetic: works for host and for device images alike. */
void dwt_util_subband(void *ptr, int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, enum dwt_subbands band,
	void **dst_ptr, int *dst_size_x, int *dst_size_y)
{
	int inner_h_x = 0, inner_h_y = 0;
	int inner_l_x = size_i_big_x, inner_l_y = size_i_big_y;
	int outer_x = size_o_big_x, outer_y = size_o_big_y;
	for (int j = 1; j <= j_max; j++) {
		inner_h_x = inner_l_x / 2;
		inner_h_y = inner_l_y / 2;
		inner_l_x = (inner_l_x + 1) / 2;
		inner_l_y = (inner_l_y + 1) / 2;
		outer_x = (outer_x + 1) / 2;
		outer_y = (outer_y + 1) / 2;
	}
	char *base = ptr;
	switch (band) {
	case DWT_LL:
		*dst_ptr = base;
		*dst_size_x = inner_l_x;
		*dst_size_y = inner_l_y;
		break;
	case DWT_HL:
		*dst_ptr = base + (long)outer_x * stride_y;
		*dst_size_x = inner_h_x;
		*dst_size_y = inner_l_y;
		break;
	case DWT_LH:
		*dst_ptr = base + (long)outer_y * stride_x;
		*dst_size_x = inner_l_x;
		*dst_size_y = inner_h_y;
		break;
	case DWT_HH:
		*dst_ptr = base + (long)outer_y * stride_x + (long)outer_x * stride_y;
		*dst_size_x = inner_h_x;
		*dst_size_y = inner_h_y;
		break;
	}
}

/* src/libdwt.c:20892, 20950 */
void dwt_util_subband_s(void *ptr, int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, enum dwt_subbands band,
	void **dst_ptr, int *dst_size_x, int *dst_size_y)
{
	dwt_util_subband(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
		j_max, band, dst_ptr, dst_size_x, dst_size_y);
}

void dwt_util_subband_i(void *ptr, int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, enum dwt_subbands band,
	void **dst_ptr, int *dst_size_x, int *dst_size_y)
{
	dwt_util_subband(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
		j_max, band, dst_ptr, dst_size_x, dst_size_y);
}

/* src/libdwt.c:1064-1073 */
float *dwt_util_addr_coeff_s(void *ptr, int y, int x, int stride_x, int stride_y)
{
	return (float *)((char *)ptr + (long)y * stride_x + (long)x * stride_y);
}

int *dwt_util_addr_coeff_i(void *ptr, int y, int x, int stride_x, int stride_y)
{
	return (int *)((char *)ptr + (long)y * stride_x + (long)x * stride_y);
}

/* src/libdwt.c:16780-16799: computes the level count a transform would use and nothing else */
void dwt_cdf53_2f_dummy_s(void *ptr, int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int *j_max_ptr, int decompose_one)
{
	(void)ptr; (void)stride_x; (void)stride_y; (void)size_i_big_x; (void)size_i_big_y;
	const int lo = size_o_big_x < size_o_big_y ? size_o_big_x : size_o_big_y;
	const int hi = size_o_big_x > size_o_big_y ? size_o_big_x : size_o_big_y;
	int j_limit = 0;
	while (j_limit < 31 && (1 << j_limit) < (decompose_one ? hi : lo))
		j_limit++;
	if (*j_max_ptr < 0 || *j_max_ptr > j_limit)
		*j_max_ptr = j_limit;
}
