/*
 * dwt_util.c -- the host-side helpers of libdwt's API that callers of the 2-D path
 * use around the transforms (examples/simple/simple.c:12-98): image allocation,
 * strides, synthetic test images, comparison, viewing, PGM output, timers, logging.
 * Plain C, host memory only; restated from the behaviour of the cited reference
 * lines, not from their text.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include "../../include/libdwt.h"

#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>
#include <sys/times.h>
#include <time.h>
#include <unistd.h>

static inline char *px(const void *ptr, int y, int x, int stride_x, int stride_y)
{
	return (char *)ptr + (long)y * stride_x + (long)x * stride_y; /* src/inline.h:180-189 */
}

static inline float ld_s(const void *p)
{
	float v;
	memcpy(&v, p, sizeof v);
	return v;
}

static inline int ld_i(const void *p)
{
	int v;
	memcpy(&v, p, sizeof v);
	return v;
}

/* ---- logging (src/libdwt.c:20334-20421) ---- */
static const char *const k_prefix[] = {"", "DEBUG: ", "INFO: ", "WARNING: ", "ERROR: ", "TEST: "};

static int vlog(enum dwt_util_loglevel level, const char *format, va_list ap)
{
	int n = 0;
	flockfile(stderr);
	n += fputs(k_prefix[level <= LOG_TEST ? level : LOG_NONE], stderr) >= 0;
	n += vfprintf(stderr, format, ap);
	fflush(stderr);
	funlockfile(stderr);
	return n;
}

int dwt_util_log(enum dwt_util_loglevel level, const char *format, ...)
{
	va_list ap;
	va_start(ap, format);
	const int n = vlog(level, format, ap);
	va_end(ap);
	return n;
}

void dwt_util_error(const char *format, ...)
{
	va_list ap;
	va_start(ap, format);
	vlog(LOG_ERR, format, ap);
	va_end(ap);
	dwt_util_abort();
}

/* ---- identification (src/libdwt.c:19220-19330) ---- */
const char *dwt_util_version(void)
{
	return "libdwt_amd 0.1 (MI355X/gfx950 backend; API of libdwt 2015-02-18-dev)";
}

const char *dwt_util_arch(void)
{
	return "x86_64+gfx950";
}

const char *dwt_util_node(void)
{
	static char name[256];
	if (gethostname(name, sizeof(name) - 1))
		strcpy(name, "unknown");
	return name;
}

const char *dwt_util_appname(void)
{
	static char name[4096];
	const ssize_t n = readlink("/proc/self/exe", name, sizeof(name) - 1);
	if (n <= 0)
		return "unknown";
	name[n] = 0;
	const char *slash = strrchr(name, '/');
	return slash ? slash + 1 : name;
}

/* ---- strides and allocation ---- */
/* The reference's primality test (src/libdwt.c:20534-20562) accepts N when the
 * multiplicative order of 2 modulo N divides N-1, i.e. the base-2 Fermat test. */
static int fermat2_prime(int n)
{
	if (n == 2)
		return 1;
	if (n < 2 || !(n & 1))
		return 0;
	unsigned long long r = 1, b = 2;
	for (int e = n - 1; e; e >>= 1) {
		if (e & 1)
			r = r * b % (unsigned)n;
		b = b * b % (unsigned)n;
	}
	return r == 1;
}

static int next_prime(int n) /* src/libdwt.c:20573-20584 */
{
	if (n <= 2)
		return 2;
	n |= 1;
	while (!fermat2_prime(n))
		n += 2;
	return n;
}

static int align_to(int v, int a)
{
	return (v + a - 1) / a * a;
}

static int ceil_log2_i(int x)
{
	int n = 0;
	while (n < 31 && (1 << n) < x)
		n++;
	return n;
}

int dwt_util_get_opt_stride(int min_stride) /* src/libdwt.c:20641-20669, x86_64 branch */
{
	return next_prime(min_stride);
}

int dwt_util_get_stride(int min_stride, int opt) /* src/libdwt.c:20688-20729 */
{
	switch (opt) {
	case 0: return min_stride;
	case 1: return next_prime(min_stride);
	case 2: return next_prime(align_to(min_stride, 64) >> 6) << 6;
	case 3: return ((align_to(min_stride, 4096) >> 6) + 1) << 6;
	case 4: return align_to(min_stride, 64);
	case 5: return align_to(min_stride, 4096) + (1 << ceil_log2_i(min_stride));
	case 6: return min_stride | 1;
	case 7: return ((align_to(min_stride, 64) >> 6) | 1) << 6;
	default:
		dwt_util_log(LOG_DBG, "%s: invalid stride choice (%i)\n", __func__, opt);
		return min_stride;
	}
}

size_t dwt_util_image_size(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y)
{
	(void)stride_y;
	(void)size_o_big_x;
	return (size_t)stride_x * size_o_big_y;
}

void dwt_util_alloc_image(void **pptr, int stride_x, int stride_y, int size_o_big_x, int size_o_big_y)
{
	(void)stride_y;
	(void)size_o_big_x;
	/* 16-byte aligned rows*pitch bytes, as src/libdwt.c:1452 */
	void *p = NULL;
	size_t bytes = (size_t)stride_x * (size_t)size_o_big_y;
	if (bytes == 0)
		bytes = 16;
	if (posix_memalign(&p, 16, bytes)) {
		dwt_util_log(LOG_ERR, "Unable to allocate memory.\n");
		dwt_util_abort();
	}
	*pptr = p;
}

void dwt_util_free_image(void **pptr)
{
	free(*pptr);
	*pptr = NULL;
}

/* ---- synthetic inputs (src/libdwt.c:1201-1244, 1142-1167) ---- */
void dwt_util_test_image_fill_s(void *ptr, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y, int rand)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			int xx = (x + 1) >> rand;
			const int yy = y + 1;
			const float v = 2 * xx * yy / (float)(xx * xx + yy * yy + 1);
			memcpy(px(ptr, y, x, stride_x, stride_y), &v, sizeof v);
		}
}

void dwt_util_test_image_fill_i(void *ptr, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y, int rand)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			const int xx = x >> rand;
			const int v = 255 * (2 * xx * y) / (xx * xx + y * y + 1);
			memcpy(px(ptr, y, x, stride_x, stride_y), &v, sizeof v);
		}
}

/* the other synthetic patterns (src/libdwt.c:1201-1244 float, :1142-1167 int): type 0 as above,
 * 1 = type 0 modulated by |sin(x/10)| |cos(xy/5)| (float only), 2 = (x ^ y) & 0xff (float: / 32,
 * 1-based x, y), 3 = the 2x2 parity pattern (float only) */
void dwt_util_test_image_fill2_s(void *ptr, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y, int rand, int type)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			int xx = x + 1;
			const int yy = y + 1;
			float v;
			switch (type) {
			case 0:
				xx >>= rand;
				v = 2 * xx * yy / (float)(xx * xx + yy * yy + 1);
				break;
			case 1:
				xx >>= rand;
				v = 2 * xx * yy / (float)(xx * xx + yy * yy + 1) * fabsf(sinf(xx / 10.f)) * fabsf(cosf(yy * xx / 5.f));
				break;
			case 2:
				v = (float)((xx ^ yy) & 0xff) / 32;
				break;
			case 3:
				v = ((((xx & 1) << 1) | (yy & 1)) + 1) / 4.f;
				break;
			default:
				dwt_util_log(LOG_ERR, "Unknown test image type.\n");
				dwt_util_abort();
				return;
			}
			memcpy(px(ptr, y, x, stride_x, stride_y), &v, sizeof v);
		}
}

void dwt_util_test_image_fill2_i(void *ptr, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y, int rand, int type)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			int v;
			if (type == 0) {
				const int xx = x >> rand;
				v = 255 * (2 * xx * y) / (xx * xx + y * y + 1);
			} else if (type == 2) {
				v = (x ^ y) & 0xff;
			} else {
				dwt_util_log(LOG_ERR, "Unknown test image type.\n");
				dwt_util_abort();
				return;
			}
			memcpy(px(ptr, y, x, stride_x, stride_y), &v, sizeof v);
		}
}

void dwt_util_copy_s(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++)
			memcpy(px(dst, y, x, stride_x, stride_y), px(src, y, x, stride_x, stride_y), 4);
}

void dwt_util_copy_i(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y)
{
	dwt_util_copy_s(src, dst, stride_x, stride_y, size_i_big_x, size_i_big_y);
}

/* double-precision twins used by examples/simple-double and the self-test
 * (src/libdwt.c:1112-1125 + 1246-1266 pattern with 0-based x, y; copy :1426; compare 1e-6
 * absolute; view; PGM writer) */
void dwt_util_test_image_fill_d(void *ptr, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y, int rand)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			const int xx = x >> rand;
			const double v = 2 * xx * y / (double)(xx * xx + y * y + 1);
			memcpy(px(ptr, y, x, stride_x, stride_y), &v, sizeof v);
		}
}

void dwt_util_copy_d(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++)
			memcpy(px(dst, y, x, stride_x, stride_y), px(src, y, x, stride_x, stride_y), 8);
}

int dwt_util_compare_d(void *ptr1, void *ptr2, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			double a, b;
			memcpy(&a, px(ptr1, y, x, stride_x, stride_y), 8);
			memcpy(&b, px(ptr2, y, x, stride_x, stride_y), 8);
			if (!isfinite(a) || !isfinite(b) || fabs(a - b) > 1e-6)
				return 1;
		}
	return 0;
}

void dwt_util_conv_show_d(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			double c;
			memcpy(&c, px(src, y, x, stride_x, stride_y), 8);
			const double t = log(1. + fabs(c) * 100.) / 10.;
			memcpy(px(dst, y, x, stride_x, stride_y), &t, 8);
		}
}

int dwt_util_save_to_pgm_d(const char *filename, double max_value, const void *ptr, int stride_x, int stride_y,
	int size_i_big_x, int size_i_big_y)
{
	FILE *f = fopen(filename, "w");
	if (!f)
		return 1;
	fprintf(f, "P2\n%i %i\n%i\n", size_i_big_x, size_i_big_y, 255);
	int incidents = 0;
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			double p;
			memcpy(&p, px(ptr, y, x, stride_x, stride_y), 8);
			int val = (int)(255 * p / max_value);
			if (p - 1e-6 > max_value && !incidents++)
				dwt_util_log(LOG_WARN, "%s: Maximum pixel intensity exceeded (%f > %f) at (y=%i, x=%i). Such an incident will be reported only once.\n", __func__, p, max_value, y, x);
			if (p > max_value)
				val = 255;
			if (p + 1e-6 < 0.0 && !incidents++)
				dwt_util_log(LOG_WARN, "%s: Minimum pixel intensity exceeded (%f < %f) at (y=%i, x=%i). Such an incident will be reported only once.\n", __func__, p, 0.0f, y, x);
			if (p < 0.0)
				val = 0;
			if (fprintf(f, "%i\n", val) < 0) {
				dwt_util_log(LOG_WARN, "%s: error writing into file.\n", __func__);
				fclose(f);
				return 1;
			}
		}
	fclose(f);
	if (incidents)
		dwt_util_log(LOG_WARN, "%s: %i errors ocurred while saving a file.\n", __func__, incidents);
	return 0;
}

/* ---- comparison (src/libdwt.c:1593-1620, 1531-1558) ---- */
int dwt_util_compare_s(void *ptr1, void *ptr2, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y)
{
	const float eps = 1e-3f;
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			const float a = ld_s(px(ptr1, y, x, stride_x, stride_y));
			const float b = ld_s(px(ptr2, y, x, stride_x, stride_y));
			if (isnan(a) || isinf(a) || isnan(b) || isinf(b))
				return 1;
			if (fabsf(a - b) > eps)
				return 1;
		}
	return 0;
}

int dwt_util_compare_i(void *ptr1, void *ptr2, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++)
			if (ld_i(px(ptr1, y, x, stride_x, stride_y)) != ld_i(px(ptr2, y, x, stride_x, stride_y)))
				return 1;
	return 0;
}

/* ---- viewing (src/libdwt.c:21075-21117, 21020-21044) ---- */
void dwt_util_conv_show_s(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y)
{
	int reported = 0;
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			const float c = ld_s(px(src, y, x, stride_x, stride_y));
			float t = (float)log(1.f + fabsf(c) * 100.f); /* log_i_s, :21010 */
			t /= 10.f;
			if (!isfinite(t)) {
				if (!reported++)
					dwt_util_log(LOG_ERR, "either NaN or INFINITY; this error will be reported only once\n");
				t = 0.f;
			}
			memcpy(px(dst, y, x, stride_x, stride_y), &t, sizeof t);
		}
}

void dwt_util_conv_show_i(const void *src, void *dst, int stride_x, int stride_y, int size_i_big_x, int size_i_big_y)
{
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			const int v = abs(ld_i(px(src, y, x, stride_x, stride_y)));
			memcpy(px(dst, y, x, stride_x, stride_y), &v, sizeof v);
		}
}

/* src/libdwt.c:19727-19792: temp = conv_show(input); scale = the largest sample of temp (dwt_util_find_min_max_s,
 * :25426); ASCII PGM of temp against that scale */
int dwt_util_save_log_to_pgm_s(const char *path, const void *ptr, int stride_x, int stride_y, int size_x, int size_y)
{
	void *temp = NULL;
	dwt_util_alloc_image(&temp, stride_x, stride_y, size_x, size_y);
	dwt_util_conv_show_s(ptr, temp, stride_x, stride_y, size_x, size_y);
	float maxv = ld_s(px(temp, 0, 0, stride_x, stride_y));
	for (int y = 0; y < size_y; y++)
		for (int x = 0; x < size_x; x++) {
			const float v = ld_s(px(temp, y, x, stride_x, stride_y));
			if (v > maxv)
				maxv = v;
		}
	dwt_util_save_to_pgm_s(path, maxv, temp, stride_x, stride_y, size_x, size_y);
	dwt_util_free_image(&temp);
	return 0;
}

/* ---- ASCII PGM (src/libdwt.c:19794-19872 float, :19728-19792 int) ---- */
int dwt_util_save_to_pgm_s(const char *filename, float max_value, const void *ptr, int stride_x, int stride_y,
	int size_i_big_x, int size_i_big_y)
{
	FILE *f = fopen(filename, "w");
	if (!f)
		return 1;
	fprintf(f, "P2\n%i %i\n%i\n", size_i_big_x, size_i_big_y, 255);
	int incidents = 0;
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			const float p = ld_s(px(ptr, y, x, stride_x, stride_y));
			int val = (int)(255 * p / max_value);
			if (p != p) {
				if (!incidents++)
					dwt_util_log(LOG_WARN, "%s: NaN value at (y=%i, x=%i). Such an incident will be reported only once.\n", __func__, y, x);
				val = 0;
			}
			if (p - 1e-3f > max_value && !incidents++)
				dwt_util_log(LOG_WARN, "%s: Maximum pixel intensity exceeded (%f > %f) at (y=%i, x=%i). Such an incident will be reported only once.\n", __func__, p, max_value, y, x);
			if (p + 1e-3f < 0.0f && !incidents++)
				dwt_util_log(LOG_WARN, "%s: Minimum pixel intensity exceeded (%f < %f) at (y=%i, x=%i). Such an incident will be reported only once.\n", __func__, p, 0.0f, y, x);
			if (p > max_value)
				val = 255;
			if (p < 0.0f)
				val = 0;
			if (fprintf(f, "%i\n", val) < 0) {
				dwt_util_log(LOG_WARN, "%s: error writing into file.\n", __func__);
				fclose(f);
				return 1;
			}
		}
	fclose(f);
	if (incidents)
		dwt_util_log(LOG_WARN, "%s: %i errors ocurred while saving a file.\n", __func__, incidents);
	return 0;
}

int dwt_util_save_to_pgm_i(const char *filename, int max_value, const void *ptr, int stride_x, int stride_y,
	int size_i_big_x, int size_i_big_y)
{
	FILE *f = fopen(filename, "w");
	if (!f)
		return 1;
	fprintf(f, "P2\n%i %i\n%i\n", size_i_big_x, size_i_big_y, 255);
	int incidents = 0;
	for (int y = 0; y < size_i_big_y; y++)
		for (int x = 0; x < size_i_big_x; x++) {
			const int p = ld_i(px(ptr, y, x, stride_x, stride_y));
			int val = max_value ? (int)(255LL * p / max_value) : 0;
			if (p > max_value) {
				if (!incidents++)
					dwt_util_log(LOG_WARN, "%s: Maximum pixel intensity exceeded (%i > %i) at (y=%i, x=%i). Such an incident will be reported only once.\n", __func__, p, max_value, y, x);
				val = 255;
			}
			if (p < 0) {
				if (!incidents++)
					dwt_util_log(LOG_WARN, "%s: Minimum pixel intensity exceeded (%i < %i) at (y=%i, x=%i). Such an incident will be reported only once.\n", __func__, p, 0, y, x);
				val = 0;
			}
			if (fprintf(f, "%i\n", val) < 0) {
				dwt_util_log(LOG_WARN, "%s: error writing into file.\n", __func__);
				fclose(f);
				return 1;
			}
		}
	fclose(f);
	if (incidents)
		dwt_util_log(LOG_WARN, "%s: %i errors ocurred while saving a file.\n", __func__, incidents);
	return 0;
}

/* ---- timers (src/libdwt.c:18534-18957) ---- */
static clockid_t clock_id_of(int type)
{
	switch (type) {
	case DWT_TIME_CLOCK_GETTIME:
	case DWT_TIME_CLOCK_GETTIME_REALTIME: return CLOCK_REALTIME; /* autoselect = CLOCK_REALTIME, :18711 */
	case DWT_TIME_CLOCK_GETTIME_MONOTONIC: return CLOCK_MONOTONIC;
	case DWT_TIME_CLOCK_GETTIME_MONOTONIC_RAW: return CLOCK_MONOTONIC_RAW;
	case DWT_TIME_CLOCK_GETTIME_PROCESS_CPUTIME_ID: return CLOCK_PROCESS_CPUTIME_ID;
	case DWT_TIME_CLOCK_GETTIME_THREAD_CPUTIME_ID: return CLOCK_THREAD_CPUTIME_ID;
	default: return (clockid_t)-1;
	}
}

int dwt_util_clock_available(int type)
{
	return (type >= DWT_TIME_CLOCK_GETTIME && type <= DWT_TIME_GETTIMEOFDAY) || type == DWT_TIME_AUTOSELECT ? 0 : -1;
}

int dwt_util_clock_autoselect(void)
{
	return DWT_TIME_CLOCK_GETTIME;
}

dwt_clock_t dwt_util_get_frequency(int type)
{
	if (type == DWT_TIME_AUTOSELECT)
		type = dwt_util_clock_autoselect();
	if (clock_id_of(type) != (clockid_t)-1)
		return 1000000000;
	switch (type) {
	case DWT_TIME_CLOCK: return CLOCKS_PER_SEC;
	case DWT_TIME_TIMES: return sysconf(_SC_CLK_TCK);
	case DWT_TIME_GETRUSAGE:
	case DWT_TIME_GETRUSAGE_SELF:
	case DWT_TIME_GETRUSAGE_CHILDREN:
	case DWT_TIME_GETRUSAGE_THREAD:
	case DWT_TIME_GETTIMEOFDAY: return 1000000;
	default: abort();
	}
}

dwt_clock_t dwt_util_get_clock(int type)
{
	if (type == DWT_TIME_AUTOSELECT)
		type = dwt_util_clock_autoselect();
	const clockid_t id = clock_id_of(type);
	if (id != (clockid_t)-1) {
		struct timespec ts;
		if (clock_gettime(id, &ts))
			abort();
		return (dwt_clock_t)ts.tv_sec * 1000000000 + ts.tv_nsec;
	}
	switch (type) {
	case DWT_TIME_CLOCK: return clock();
	case DWT_TIME_TIMES: {
		struct tms t;
		times(&t);
		return t.tms_utime;
	}
	case DWT_TIME_GETRUSAGE:
	case DWT_TIME_GETRUSAGE_SELF:
	case DWT_TIME_GETRUSAGE_CHILDREN:
	case DWT_TIME_GETRUSAGE_THREAD: {
		struct rusage ru;
		const int who = type == DWT_TIME_GETRUSAGE_CHILDREN ? RUSAGE_CHILDREN
			: type == DWT_TIME_GETRUSAGE_THREAD ? RUSAGE_THREAD : RUSAGE_SELF;
		if (getrusage(who, &ru))
			abort();
		return (dwt_clock_t)ru.ru_utime.tv_sec * 1000000 + ru.ru_utime.tv_usec;
	}
	case DWT_TIME_GETTIMEOFDAY: {
		struct timeval tv;
		gettimeofday(&tv, NULL);
		return (dwt_clock_t)tv.tv_sec * 1000000 + tv.tv_usec;
	}
	default: abort();
	}
}
