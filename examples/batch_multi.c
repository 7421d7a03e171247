/*
 * batch_multi.c -- a batch of images sharded over all the GPUs of the node from ONE process, through the
 * C-ABI only (include/libdwt_hip.h).  Image b belongs to slot b*G/B (dwt_hip_shard_bounds).  Two modes:
 *
 *   (default)    dwt_hip_transform2d_batch_sharded: the whole batch lives on device 0; every other slot pulls its
 *                shard over xGMI, transforms it and pushes the coefficients back (root-egress bound);
 *   --resident   dwt_hip_transform2d_batch_multi: every shard is allocated and filled ON ITS OWN DEVICE and
 *                transformed where it lies -- nothing crosses xGMI (the >= 7x case);
 *                dwt_hip_tune_batch_multi first (explicit measurement: tile heights, scratch placement, per slot).
 *
 * Either way the result is compared with the single-GPU batched call bit for bit.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/batch_multi.c -o batch_multi \
 *       -Llibdwt_amd -l:libdwt_hip.so -Wl,-rpath,$PWD/libdwt_amd -lm
 *   ./batch_multi [--resident] [images] [size] [levels] [slots]      (slots > GPUs: several contexts per GPU)
 */
#include "libdwt.h"
#include "libdwt_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, ": %s\n", dwt_hip_last_error()); return 1; } while (0)

int main(int argc, char **argv)
{
	int resident = 0;
	if (argc > 1 && !strcmp(argv[1], "--resident")) {
		resident = 1;
		argv++;
		argc--;
	}
	const int B = argc > 1 ? atoi(argv[1]) : 16, n = argc > 2 ? atoi(argv[2]) : 2048, J = argc > 3 ? atoi(argv[3]) : 5;
	if (dwt_hip_set_device(0))
		DIE("no device");
	const int gpus = dwt_hip_device_count();
	int G = argc > 4 ? atoi(argv[4]) : gpus;
	if (G < 1 || G > 64 || B < 1)
		return 1;
	if (G > B)
		G = B;
	int devices[64];
	for (int k = 0; k < G; k++)
		devices[k] = k % gpus;
	const size_t img = (size_t)n * n * sizeof(float), total = img * B;
	float *host = malloc(total), *one = malloc(total), *many = malloc(total);
	if (!host || !one || !many)
		return 1;
	for (size_t i = 0; i < total / sizeof(float); i++)
		host[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
	void *src = dwt_hip_malloc(total), *dst = dwt_hip_malloc(total);
	if (!src || !dst || dwt_hip_memcpy_h2d(src, host, total))
		DIE("allocation failed");
	const int clk = dwt_util_clock_autoselect();
	int j = J;
	/* the yardstick: the whole batch in one call on device 0 */
	if (dwt_hip_transform2d_batch(DWT_HIP_CDF97_S, 0, src, dst, img, B, n * 4, n, n, &j) || dwt_hip_memcpy_d2h(one, dst, total))
		DIE("single-GPU call");
	double best = 1e30;
	if (!resident) {
		for (int rep = 0; rep < 3; rep++) { /* the first pass creates the slots' threads, contexts and staging */
			j = J;
			const dwt_clock_t t0 = dwt_util_get_clock(clk);
			if (dwt_hip_transform2d_batch_sharded(DWT_HIP_CDF97_S, 0, src, dst, img, B, n * 4, n, n, &j, devices, G))
				DIE("sharded call");
			const double s = (double)(dwt_util_get_clock(clk) - t0) / dwt_util_get_frequency(clk);
			if (rep && s < best)
				best = s;
		}
		if (dwt_hip_memcpy_d2h(many, dst, total))
			DIE("download");
	} else {
		/* every shard allocated and filled by a context bound to ITS device; the main thread hops between them */
		const void *srcs[64];
		void *dsts[64];
		int counts[64], first[64];
		for (int k = 0; k < G; k++) {
			dwt_hip_shard_bounds(B, G, k, &first[k], &counts[k]);
			if (dwt_hip_set_device(devices[k]))
				DIE("device %d", devices[k]);
			void *s = dwt_hip_malloc(img * counts[k]);
			dsts[k] = dwt_hip_malloc(img * counts[k]);
			if (!s || !dsts[k] || dwt_hip_memcpy_h2d(s, (char *)host + img * first[k], img * counts[k]))
				DIE("shard %d", k);
			srcs[k] = s;
		}
		if (dwt_hip_set_device(0))
			DIE("device 0");
		/* explicit measurement, once, for shards that stay resident (same bits with and without) */
		if (dwt_hip_tune_batch_multi(DWT_HIP_CDF97_S, 0, srcs, dsts, counts, devices, G, img, n * 4, n, n, J))
			DIE("tune");
		for (int rep = 0; rep < 4; rep++) {
			j = J;
			const dwt_clock_t t0 = dwt_util_get_clock(clk);
			if (dwt_hip_transform2d_batch_multi(DWT_HIP_CDF97_S, 0, srcs, dsts, counts, devices, G, img, n * 4, n, n, &j))
				DIE("resident call");
			const double s = (double)(dwt_util_get_clock(clk) - t0) / dwt_util_get_frequency(clk);
			if (rep && s < best)
				best = s;
		}
		for (int k = 0; k < G; k++) {
			if (dwt_hip_set_device(devices[k]) || dwt_hip_memcpy_d2h((char *)many + img * first[k], dsts[k], img * counts[k]))
				DIE("download of shard %d", k);
			dwt_hip_free((void *)srcs[k]);
			dwt_hip_free(dsts[k]);
		}
		if (dwt_hip_set_device(0))
			DIE("device 0");
	}
	const int same = !memcmp(one, many, total);
	printf("%d images %dx%d, %d levels, %d slot(s) on %d GPU(s), %s: %.3f ms per batch = %.2f Gsamples/s; bits %s the single-GPU call\n",
		B, n, n, j, G, gpus, resident ? "shards resident per device" : "batch on device 0, split by peer copies", best * 1e3,
		(double)B * n * n / best / 1e9, same ? "equal" : "DIFFER FROM");
	dwt_hip_free(src);
	dwt_hip_free(dst);
	free(host);
	free(one);
	free(many);
	dwt_util_finish();
	return same ? 0 : 2;
}
