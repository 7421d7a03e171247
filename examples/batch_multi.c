/*
 * batch_multi.c -- a batch of images sharded over all the GPUs of the node from ONE process, through the
 * C-ABI only (include/libdwt_hip.h): dwt_hip_transform2d_batch_sharded.  The batch lives on device 0; image
 * b is transformed on device b*G/B; the result is compared with the single-GPU batched call.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/batch_multi.c -o batch_multi \
 *       -Llibdwt_amd -l:libdwt_hip.so -Wl,-rpath,$PWD/libdwt_amd -lm
 *   ./batch_multi [images] [size] [levels] [slots]      (slots > GPUs: several contexts per GPU)
 */
#include "libdwt.h"
#include "libdwt_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
	const int B = argc > 1 ? atoi(argv[1]) : 16, n = argc > 2 ? atoi(argv[2]) : 2048, J = argc > 3 ? atoi(argv[3]) : 5;
	if (dwt_hip_set_device(0)) {
		fprintf(stderr, "%s\n", dwt_hip_last_error());
		return 1;
	}
	const int gpus = dwt_hip_device_count();
	const int G = argc > 4 ? atoi(argv[4]) : gpus;
	int devices[64];
	for (int k = 0; k < G && k < 64; k++)
		devices[k] = k % gpus;
	const size_t img = (size_t)n * n * sizeof(float), total = img * B;
	float *host = malloc(total), *one = malloc(total), *many = malloc(total);
	for (size_t i = 0; i < total / sizeof(float); i++)
		host[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
	void *src = dwt_hip_malloc(total), *dst = dwt_hip_malloc(total);
	if (!host || !one || !many || !src || !dst || dwt_hip_memcpy_h2d(src, host, total)) {
		fprintf(stderr, "allocation failed: %s\n", dwt_hip_last_error());
		return 1;
	}
	const int clk = dwt_util_clock_autoselect();
	int j = J;
	if (dwt_hip_transform2d_batch(DWT_HIP_CDF97_S, 0, src, dst, img, B, n * 4, n, n, &j) || dwt_hip_memcpy_d2h(one, dst, total)) {
		fprintf(stderr, "%s\n", dwt_hip_last_error());
		return 1;
	}
	double best = 1e30;
	for (int rep = 0; rep < 3; rep++) { /* the first pass creates the slots' threads, contexts and staging */
		j = J;
		const dwt_clock_t t0 = dwt_util_get_clock(clk);
		if (dwt_hip_transform2d_batch_sharded(DWT_HIP_CDF97_S, 0, src, dst, img, B, n * 4, n, n, &j, devices, G)) {
			fprintf(stderr, "%s\n", dwt_hip_last_error());
			return 1;
		}
		const double s = (double)(dwt_util_get_clock(clk) - t0) / dwt_util_get_frequency(clk);
		if (rep && s < best)
			best = s;
	}
	if (dwt_hip_memcpy_d2h(many, dst, total))
		return 1;
	const int same = !memcmp(one, many, total);
	printf("%d images %dx%d, %d levels, %d slot(s) on %d GPU(s): %.3f ms per batch incl. the split = %.2f Gsamples/s; bits %s the single-GPU call\n",
		B, n, n, j, G, gpus, best * 1e3, (double)B * n * n / best / 1e9, same ? "equal" : "DIFFER FROM");
	dwt_hip_free(src);
	dwt_hip_free(dst);
	free(host);
	free(one);
	free(many);
	dwt_util_finish();
	return same ? 0 : 2;
}
