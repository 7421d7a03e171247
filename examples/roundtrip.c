/*
 * roundtrip.c -- the flow of libdwt's examples/simple (512x512 float, prime row pitch,
 * full decomposition, inverse, compare) plus a device-resident 8192x8192 5-level
 * transform through the same entry points.  Own code written against include/libdwt.h.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/roundtrip.c -o roundtrip \
 *       -Llibdwt_amd -l:libdwt_hip.so -Wl,-rpath,$PWD/libdwt_amd -lm
 */
#include "libdwt.h"
#include "libdwt_hip.h"

#include <stdlib.h>

int main(void)
{
	dwt_util_init();
	dwt_util_log(LOG_INFO, "library: %s on %s\n", dwt_util_version(), dwt_hip_device_name());

	/* host image, drop-in call */
	const int x = 512, y = 512;
	const int stride_y = sizeof(float);
	const int stride_x = dwt_util_get_opt_stride(stride_y * x);
	void *a, *b;
	dwt_util_alloc_image(&a, stride_x, stride_y, x, y);
	dwt_util_alloc_image(&b, stride_x, stride_y, x, y);
	dwt_util_test_image_fill_s(a, stride_x, stride_y, x, y, 0);
	dwt_util_copy_s(a, b, stride_x, stride_y, x, y);
	int j = -1;
	const int clk = dwt_util_clock_autoselect();
	dwt_clock_t t0 = dwt_util_get_clock(clk);
	dwt_cdf97_2f_s(a, stride_x, stride_y, x, y, x, y, &j, 0, 0);
	dwt_clock_t t1 = dwt_util_get_clock(clk);
	dwt_util_log(LOG_INFO, "host image %dx%d pitch %d: %d levels in %f s\n", x, y, stride_x, j,
		(double)(t1 - t0) / dwt_util_get_frequency(clk));
	dwt_cdf97_2i_s(a, stride_x, stride_y, x, y, x, y, j, 0, 0);
	const int bad_host = dwt_util_compare_s(a, b, stride_x, stride_y, x, y);
	dwt_util_log(LOG_INFO, bad_host ? "host round trip: images differ\n" : "host round trip: success\n");

	/* device-resident image */
	const int n = 8192;
	const size_t bytes = (size_t)n * n * sizeof(float);
	float *h = malloc(bytes), *r = malloc(bytes);
	dwt_util_test_image_fill_s(h, n * 4, 4, n, n, 0);
	void *src = dwt_hip_malloc(bytes), *dst = dwt_hip_malloc(bytes);
	if (!h || !r || !src || !dst)
		dwt_util_error("allocation failed: %s\n", dwt_hip_last_error());
	dwt_hip_memcpy_h2d(src, h, bytes);
	j = 5;
	dwt_cdf97_2f_s2(src, dst, n * 4, 4, n, n, n, n, &j, 0, 0); /* warm-up */
	dwt_hip_sync();
	t0 = dwt_util_get_clock(clk);
	for (int i = 0; i < 10; i++)
		dwt_cdf97_2f_s2(src, dst, n * 4, 4, n, n, n, n, &j, 0, 0);
	dwt_hip_sync();
	t1 = dwt_util_get_clock(clk);
	const double s = (double)(t1 - t0) / dwt_util_get_frequency(clk) / 10;
	dwt_util_log(LOG_INFO, "device image %dx%d, %d levels: %.1f us per transform, %.1f Gsamples/s\n", n, n, j,
		s * 1e6, (double)n * n / s / 1e9);
	dwt_cdf97_2i_s2(dst, src, n * 4, 4, n, n, n, n, j, 0, 0);
	dwt_hip_memcpy_d2h(r, src, bytes);
	const int bad_dev = dwt_util_compare_s(r, h, n * 4, 4, n, n);
	dwt_util_log(LOG_INFO, bad_dev ? "device round trip: images differ\n" : "device round trip: success\n");

	dwt_hip_free(src);
	dwt_hip_free(dst);
	free(h);
	free(r);
	dwt_util_free_image(&a);
	dwt_util_free_image(&b);
	dwt_util_finish();
	return bad_host || bad_dev;
}
