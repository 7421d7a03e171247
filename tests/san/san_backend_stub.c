/*
 * san_backend_stub.c -- TEST DOUBLE of the device backend for the host-only sanitizer build (tests/san/Makefile).
 *
 * The product's host C files (dwt_entry.c, dwt_util.c, dwt_harness.c, dwt_io.c, dwt_volume.c) call the backend
 * through the dwt_hip_* C-ABI.  GPU AddressSanitizer does not exist on this pool, so the host code is exercised
 * under ASan / UBSan with THIS file in the backend's place: "device" memory is host memory and a transform is
 * the oracle's (oracle/dwt_oracle.c -- test infrastructure, linked here only).  It is never part of the product.
 */
#include "../../include/libdwt_hip.h"
#include "../../oracle/dwt_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_err[256] = "";
static int g_inited = 0;

int dwt_hip_init(void) { g_inited = 1; return 0; }
void dwt_hip_finish(void) { g_inited = 0; }
const char *dwt_hip_last_error(void) { return g_err; }
void *dwt_hip_malloc(size_t n) { return malloc(n ? n : 1); }
void dwt_hip_free(void *p) { free(p); }
void *dwt_hip_malloc_host(size_t n) { return malloc(n ? n : 1); }
void dwt_hip_free_host(void *p) { free(p); }
int dwt_hip_memcpy_h2d(void *d, const void *h, size_t n) { memcpy(d, h, n); return 0; }
int dwt_hip_memcpy_d2h(void *h, const void *d, size_t n) { memcpy(h, d, n); return 0; }
int dwt_hip_is_device_pointer(const void *p) { (void)p; return 0; }
int dwt_hip_set_option(const char *name, int value) { (void)name; (void)value; return 0; }
void dwt_hip_sync(void) {}

static void copy_frame(void *dst, const void *src, int stride_x, int stride_y, int w, int h, int es)
{
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++)
			memcpy((char *)dst + (long)y * stride_x + (long)x * stride_y, (const char *)src + (long)y * stride_x + (long)x * stride_y, (size_t)es);
}

int dwt_hip_transform2d(int wavelet, int inverse, const void *src, void *dst, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j, int decompose_one, int zero_padding)
{
	if (!g_inited) {
		snprintf(g_err, sizeof g_err, "stub backend: not initialised");
		return 1;
	}
	if (wavelet == DWT_HIP_CDF97_S && src != dst) {
		if (inverse)
			oracle_cdf97_2i_s2(src, dst, stride_x, stride_y, sox, soy, six, siy, *j, decompose_one, zero_padding);
		else
			oracle_cdf97_2f_s2(src, dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, zero_padding);
		return 0;
	}
	if (src != dst)
		copy_frame(dst, src, stride_x, stride_y, sox, soy, (wavelet == DWT_HIP_CDF97_D || wavelet == DWT_HIP_CDF53_D) ? 8 : 4);
	switch (wavelet) {
	case DWT_HIP_CDF97_S: if (inverse) oracle_cdf97_2i_s(dst, stride_x, stride_y, sox, soy, six, siy, *j, decompose_one, zero_padding); else oracle_cdf97_2f_s(dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, zero_padding); break;
	case DWT_HIP_CDF53_I: if (inverse) oracle_cdf53_2i_i(dst, stride_x, stride_y, sox, soy, six, siy, *j, decompose_one, zero_padding); else oracle_cdf53_2f_i(dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, zero_padding); break;
	case DWT_HIP_CDF53_S: if (inverse) oracle_cdf53_2i_s(dst, stride_x, stride_y, sox, soy, six, siy, *j, decompose_one, zero_padding); else oracle_cdf53_2f_s(dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, zero_padding); break;
	case DWT_HIP_CDF97_D: if (inverse) oracle_cdf97_2i_d(dst, stride_x, stride_y, sox, soy, six, siy, *j, decompose_one, zero_padding); else oracle_cdf97_2f_d(dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, zero_padding); break;
	case DWT_HIP_CDF53_D: if (inverse) oracle_cdf53_2i_d(dst, stride_x, stride_y, sox, soy, six, siy, *j, decompose_one, zero_padding); else oracle_cdf53_2f_d(dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, zero_padding); break;
	case DWT_HIP_CDF97_I: if (inverse) oracle_cdf97_2i_i(dst, stride_x, stride_y, sox, soy, six, siy, *j, decompose_one, zero_padding); else oracle_cdf97_2f_i(dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, zero_padding); break;
	default:
		snprintf(g_err, sizeof g_err, "stub backend: unknown wavelet %d", wavelet);
		return 1;
	}
	return 0;
}

int dwt_hip_transform2d_interleaved(int wavelet, int inverse, int flavour, const void *src, void *dst, int stride_x, int stride_y,
	int sox, int soy, int six, int siy, int *j, int decompose_one)
{
	if (src != dst)
		copy_frame(dst, src, stride_x, stride_y, sox, soy, 4);
	if (flavour == 1 && !inverse) {
		if (wavelet == DWT_HIP_CDF97_S)
			oracle_fdwt2_cdf97_s(dst, sox, soy, stride_x, stride_y, j, decompose_one);
		else
			oracle_fdwt2_cdf53_s(dst, sox, soy, stride_x, stride_y, j, decompose_one);
		return 0;
	}
	if (flavour != 0) {
		snprintf(g_err, sizeof g_err, "stub backend: flavour %d not provided", flavour);
		return 1;
	}
	if (wavelet == DWT_HIP_CDF97_S) {
		if (inverse) oracle_cdf97_2i_inplace_s(dst, stride_x, stride_y, sox, soy, six, siy, *j, decompose_one, 0);
		else oracle_cdf97_2f_inplace_s(dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, 0);
	} else if (wavelet == DWT_HIP_CDF53_S) {
		if (inverse) oracle_cdf53_2i_inplace_s(dst, stride_x, stride_y, sox, soy, six, siy, *j, decompose_one, 0);
		else oracle_cdf53_2f_inplace_s(dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, 0);
	} else {
		if (inverse) oracle_cdf97_2i_inplace_i(dst, stride_x, stride_y, sox, soy, six, siy, *j, decompose_one, 0);
		else oracle_cdf97_2f_inplace_i(dst, stride_x, stride_y, sox, soy, six, siy, j, decompose_one, 0);
	}
	return 0;
}

int dwt_hip_volume_fwd_op(const void *src, size_t ssy, size_t ssz, void *dst, size_t dsy, size_t dsz, int nx, int ny, int nz, int dirs)
{
	if (dirs != 7) {
		snprintf(g_err, sizeof g_err, "stub backend: single-direction schedules not provided");
		return 1;
	}
	for (int z = 0; z < nz; z++)
		for (int y = 0; y < ny; y++)
			memcpy((char *)dst + z * dsz + y * dsy, (const char *)src + z * ssz + y * ssy, (size_t)nx * 4);
	oracle_cdf97_3f_s(dst, 4, (long)dsy, (long)dsz, nx, ny, nz);
	return 0;
}

int dwt_hip_volume_ip(int inverse, void *data, size_t sy, size_t sz, int nx, int ny, int nz)
{
	if (inverse)
		oracle_cdf97_3i_s(data, 4, (long)sy, (long)sz, nx, ny, nz);
	else
		oracle_cdf97_3f_s(data, 4, (long)sy, (long)sz, nx, ny, nz);
	return 0;
}
