/*
 * volume3d.c -- BASELINE config 5 from C: a 512^3 float volume resident in HBM, 3-level
 * forward CDF 9/7 out of place (one fused x+y+z pass per level), in-place inverse, and the
 * interleaved-layout 2-D entries of libdwt.h / dwt-simple.h on a host image.
 * Own code written against the headers under include/.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/volume3d.c -o volume3d \
 *       -Llibdwt_amd -l:libdwt_hip.so -Wl,-rpath,$PWD/libdwt_amd -lm
 */
#include "libdwt.h"
#include "libdwt_hip.h"
#include "dwt-simple.h"

#include <math.h>
#include <stdlib.h>

int main(void)
{
	dwt_util_init();
	const int n = 512, levels = 3;
	const size_t bytes = (size_t)n * n * n * sizeof(float);
	float *h = malloc(bytes), *r = malloc(bytes);
	void *src = dwt_hip_malloc(bytes), *dst = dwt_hip_malloc(bytes);
	if (!h || !r || !src || !dst)
		dwt_util_error("allocation failed: %s\n", dwt_hip_last_error());
	for (size_t i = 0; i < (size_t)n * n * n; i++)
		h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
	dwt_hip_memcpy_h2d(src, h, bytes);

	const int clk = dwt_util_clock_autoselect();
	if (dwt_hip_transform3d_op(src, dst, (size_t)n * 4, (size_t)n * n * 4, n, n, n, levels)) /* warm-up */
		dwt_util_error("%s\n", dwt_hip_last_error());
	dwt_hip_sync();
	dwt_clock_t t0 = dwt_util_get_clock(clk);
	for (int i = 0; i < 5; i++)
		dwt_hip_transform3d_op(src, dst, (size_t)n * 4, (size_t)n * n * 4, n, n, n, levels);
	dwt_hip_sync();
	dwt_clock_t t1 = dwt_util_get_clock(clk);
	const double s = (double)(t1 - t0) / dwt_util_get_frequency(clk) / 5;
	dwt_util_log(LOG_INFO, "volume %d^3, %d levels, out of place: %.1f us per transform, %.1f Gvoxels/s\n", n, levels,
		s * 1e6, (double)n * n * n / s / 1e9);
	if (dwt_hip_transform3d(1, dst, (size_t)n * 4, (size_t)n * n * 4, n, n, n, levels))
		dwt_util_error("%s\n", dwt_hip_last_error());
	dwt_hip_memcpy_d2h(r, dst, bytes);
	double worst = 0;
	for (size_t i = 0; i < (size_t)n * n * n; i++)
		worst = fmax(worst, fabs((double)r[i] - h[i]));
	dwt_util_log(LOG_INFO, worst < 1e-3 ? "volume round trip: success (max error %g)\n" : "volume round trip: differs (max error %g)\n", worst);

	/* interleaved-layout 2-D entries on a host image with libdwt's prime row pitch */
	const int x = 513, y = 300;
	const int stride_x = dwt_util_get_opt_stride(4 * x);
	void *a, *b;
	dwt_util_alloc_image(&a, stride_x, 4, x, y);
	dwt_util_alloc_image(&b, stride_x, 4, x, y);
	dwt_util_test_image_fill_s(a, stride_x, 4, x, y, 0);
	dwt_util_copy_s(a, b, stride_x, 4, x, y);
	int j = -1;
	fdwt2_cdf97_diagonal_s(a, x, y, stride_x, 4, &j, 0);
	dwt_cdf97_2i_inplace_s(a, stride_x, 4, x, y, x, y, j, 0, 0);
	const int bad = dwt_util_compare_s(a, b, stride_x, 4, x, y);
	dwt_util_log(LOG_INFO, bad ? "interleaved round trip: differs\n" : "interleaved round trip: success (%d levels)\n", j);

	dwt_util_free_image(&a);
	dwt_util_free_image(&b);
	dwt_hip_free(src);
	dwt_hip_free(dst);
	free(h);
	free(r);
	dwt_util_finish();
	return (worst < 1e-3 && !bad) ? 0 : 1;
}
