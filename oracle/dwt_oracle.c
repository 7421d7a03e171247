/*
 * dwt_oracle.c -- CPU restatement of libdwt's lifting DWT hot path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY (see dwt_oracle.h).  Parity status: PINNED against the
 * reference compiled from its own sources (oracle/_ref/libdwt_ref.so) and the
 * committed fixtures under tests/golden/.
 *
 * Build WITHOUT floating-point contraction (-ffp-contract=off): the reference is
 * built by gcc without -mfma (arch.mk:15,38-39), so every `x += c*(a+b)` rounds
 * the sum, the product and the accumulation separately.
 *
 * The reference realises the lifting through prolog / main / epilog / short
 * variants and 17 loop schedules (src/libdwt.c:10551-10742); they all compute the
 * same thing, which is restated here once per wavelet as whole-line sweeps with
 * whole-sample symmetric reflection at both ends.
 *
 * Line ends of every float and double kernel (Mallat, interleaved and 3-D paths): the reference writes
 * the reflected pair of taps as `2*c*x` (src/libdwt.c:9545-9552, 9873-9907, 10228-10360;
 * 10994-11017, 2024-2083; src/dwt-simple.c:596-603), i.e. (2c)*x: one rounding, like c*(x+x), and the same bits
 * -- EXCEPT when x+x overflows (|x| > FLT_MAX/2) while (2c)*x does not.  The restatement
 * follows the reference ((2c)*x).  oracle_set_end_form(0) switches every such end to the
 * reflected form c*(x+x): that is what the HIP kernels evaluate (reflection is applied to
 * their load addresses), so the GPU tests on near-overflow data can separate this one
 * documented difference from everything else.
 */
#include "dwt_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants: src/inline.h:309-315 and :331-335 (stored as float there) ---- */
static const float C97_P1 = 1.58613434342059;
static const float C97_U1 = -0.0529801185729;
static const float C97_P2 = -0.8829110755309;
static const float C97_U2 = 0.4435068520439;
static const float C97_S1 = 1.1496043988602;
static const float C97_S2 = 1 / 1.1496043988602; /* double division, then rounded: inline.h:315 */
static const float C53_P1 = 0.5;
static const float C53_U1 = 0.25;
static const float C53_S1 = 1.41421356237309504880;
static const float C53_S2 = 0.70710678118654752440;
/* double constants: src/inline.h:317-323, 337-341 */
static const double D97_P1 = 1.58613434342059;
static const double D97_U1 = -0.0529801185729;
static const double D97_P2 = -0.8829110755309;
static const double D97_U2 = 0.4435068520439;
static const double D97_S1 = 1.1496043988602;
static const double D97_S2 = 1 / 1.1496043988602;
static const double D53_P1 = 0.5;
static const double D53_U1 = 0.25;
static const double D53_S1 = 1.41421356237309504880;
static const double D53_S2 = 0.70710678118654752440;

/* ---- integer helpers: src/inline.h:443-461 ---- */
int oracle_ceil_div_pow2(int i, int j)
{
	return (i + (1 << j) - 1) >> j;
}

int oracle_ceil_log2(int x)
{
	/* smallest n with (1<<n) >= x; 0 for x <= 1 */
	int n = 0;
	while (n < 31 && (1 << n) < x)
		n++;
	return n;
}

static int g_threads = 0;

int oracle_max_threads(void)
{
#ifdef _OPENMP
	return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
	return 1;
#endif
}

void oracle_set_threads(int n)
{
	g_threads = n;
}

/* whole-sample symmetric reflection of index i into [0,N-1], N >= 2 */
static inline int refl(int i, int N)
{
	const int period = 2 * (N - 1);
	i %= period;
	if (i < 0)
		i += period;
	return i < N ? i : period - i;
}

/* 1: line ends as the reference writes them, (2c)*x; 0: as reflection gives them, c*(x+x) */
static int g_end_form = 1;

void oracle_set_end_form(int faithful)
{
	g_end_form = faithful != 0;
}

/* the doubled tap of a line end, c applied to the pair (x, x): as the reference writes it, 2*c*x, or
 * as reflection gives it, c*(x+x) */
#define END2(c, x) (g_end_form ? 2 * (c) * (x) : (c) * ((x) + (x)))

/* one lifting sweep over samples of given parity: a[i] += c*(a[i-1]+a[i+1]); at the two line
 * ends both taps are one sample x and the reference adds 2*c*x (:9545, :9552, :9873, :9879) */
static inline void sweep_s(float *a, int N, int parity, float c)
{
	for (int i = parity; i < N; i += 2) {
		if (g_end_form && (i == 0 || i == N - 1)) {
			a[i] += 2 * c * a[i == 0 ? 1 : N - 2];
			continue;
		}
		const float l = a[refl(i - 1, N)];
		const float r = a[refl(i + 1, N)];
		a[i] += c * (l + r);
	}
}

/* ---- 1-D CDF 9/7 float, forward (interleaved result: even = L, odd = H) ----
 * src/libdwt.c:10744-10800 calls accel_lift_op4s_s(tmp,1,N,-p1,u1,-p2,u2,s1,+1):
 * predict1, update1, predict2, update2 (:2331-2341), then even*=zeta, odd*=1/zeta
 * (:2344-2353); ends use 2*c*neighbour (:9545,:9873): equal to c*(x+x) unless x+x overflows. */
void oracle_line_cdf97_f_s(float *a, int N)
{
	if (N < 2) {
		if (N == 1)
			a[0] = a[0] * C97_S1; /* :10757-10762 */
		return;
	}
	const float alpha = -C97_P1, beta = C97_U1, gamma = -C97_P2, delta = C97_U2;
	const float zeta = C97_S1;
	const float inv_zeta = 1 / zeta; /* float division, as `1/zeta` at :2327 */
	sweep_s(a, N, 1, alpha);
	sweep_s(a, N, 0, beta);
	sweep_s(a, N, 1, gamma);
	sweep_s(a, N, 0, delta);
	for (int i = 0; i < N; i += 2)
		a[i] *= zeta;
	for (int i = 1; i < N; i += 2)
		a[i] *= inv_zeta;
}

/* ---- 1-D CDF 9/7 float, inverse (input interleaved) ----
 * src/libdwt.c:11530-11571: accel_lift_op4s_s(tmp,0,N,-u2,p2,-u1,p1,s1,-1);
 * descale first (even*=1/zeta, odd*=zeta, :2292-2300) then four sweeps
 * even(-u2), odd(p2), even(-u1), odd(p1) (:2303-2315). */
void oracle_line_cdf97_i_s(float *a, int N)
{
	if (N < 2) {
		if (N == 1)
			a[0] = a[0] * C97_S2; /* :11546 */
		return;
	}
	const float zeta = C97_S1;
	const float inv_zeta = 1 / zeta;
	for (int i = 0; i < N; i += 2)
		a[i] *= inv_zeta;
	for (int i = 1; i < N; i += 2)
		a[i] *= zeta;
	sweep_s(a, N, 0, -C97_U2);
	sweep_s(a, N, 1, C97_P2);
	sweep_s(a, N, 0, -C97_U1);
	sweep_s(a, N, 1, C97_P1);
}

/* ---- 1-D CDF 5/3 int32: src/libdwt.c:10950-10984 (statement order kept) ---- */
void oracle_line_cdf53_f_i(int *a, int N)
{
	if (N < 2)
		return;
	for (int i = 1; i < N - 2 + (N & 1); i += 2)
		a[i] -= (a[i - 1] + a[i + 1]) >> 1;
	if (N & 1)
		a[N - 1] += (a[N - 2] + 1) >> 1;
	else
		a[N - 1] -= a[N - 2];
	a[0] += (a[1] + 1) >> 1;
	for (int i = 2; i < N - (N & 1); i += 2)
		a[i] += ((a[i - 1] + a[i + 1]) + 2) >> 2;
}

/* src/libdwt.c:11749-11783 */
void oracle_line_cdf53_i_i(int *a, int N)
{
	if (N < 2)
		return;
	for (int i = 2; i < N - (N & 1); i += 2)
		a[i] -= ((a[i - 1] + a[i + 1]) + 2) >> 2;
	a[0] -= (a[1] + 1) >> 1;
	if (N & 1)
		a[N - 1] -= (a[N - 2] + 1) >> 1;
	else
		a[N - 1] += a[N - 2];
	for (int i = 1; i < N - 2 + (N & 1); i += 2)
		a[i] += (a[i - 1] + a[i + 1]) >> 1;
}

/* ---- 1-D CDF 9/7 int32 (fixed point), src/libdwt.c:10901-10948 / 11699-11746 ---- */
void oracle_line_cdf97_f_i(int *a, int N)
{
	if (N < 2)
		return;
	for (int i = 1; i < N - 2 + (N & 1); i += 2)
		a[i] -= (+203 * (a[i - 1] + a[i + 1]) - (1 << 6)) >> 7;
	if (N & 1)
		a[N - 1] += (-217 * (a[N - 2] + a[N - 2]) + (1 << 11)) >> 12;
	else
		a[N - 1] -= (+203 * (a[N - 2] + a[N - 2]) - (1 << 6)) >> 7;
	a[0] += (-217 * (a[1] + a[1]) + (1 << 11)) >> 12;
	for (int i = 2; i < N - (N & 1); i += 2)
		a[i] += (-217 * (a[i - 1] + a[i + 1]) + (1 << 11)) >> 12;
	for (int i = 1; i < N - 2 + (N & 1); i += 2)
		a[i] -= (-113 * (a[i - 1] + a[i + 1]) - (1 << 6)) >> 7;
	if (N & 1)
		a[N - 1] += (1817 * (a[N - 2] + a[N - 2]) + (1 << 11)) >> 12;
	else
		a[N - 1] -= (-113 * (a[N - 2] + a[N - 2]) - (1 << 6)) >> 7;
	a[0] += (1817 * (a[1] + a[1]) + (1 << 11)) >> 12;
	for (int i = 2; i < N - (N & 1); i += 2)
		a[i] += (1817 * (a[i - 1] + a[i + 1]) + (1 << 11)) >> 12;
}

void oracle_line_cdf97_i_i(int *a, int N)
{
	if (N < 2)
		return;
	for (int i = 2; i < N - (N & 1); i += 2)
		a[i] -= (1817 * (a[i - 1] + a[i + 1]) + (1 << 11)) >> 12;
	a[0] -= (1817 * (a[1] + a[1]) + (1 << 11)) >> 12;
	if (N & 1)
		a[N - 1] -= (1817 * (a[N - 2] + a[N - 2]) + (1 << 11)) >> 12;
	else
		a[N - 1] += (-113 * (a[N - 2] + a[N - 2]) - (1 << 6)) >> 7;
	for (int i = 1; i < N - 2 + (N & 1); i += 2)
		a[i] += (-113 * (a[i - 1] + a[i + 1]) - (1 << 6)) >> 7;
	for (int i = 2; i < N - (N & 1); i += 2)
		a[i] -= (-217 * (a[i - 1] + a[i + 1]) + (1 << 11)) >> 12;
	a[0] -= (-217 * (a[1] + a[1]) + (1 << 11)) >> 12;
	if (N & 1)
		a[N - 1] -= (-217 * (a[N - 2] + a[N - 2]) + (1 << 11)) >> 12;
	else
		a[N - 1] += (+203 * (a[N - 2] + a[N - 2]) - (1 << 6)) >> 7;
	for (int i = 1; i < N - 2 + (N & 1); i += 2)
		a[i] += (+203 * (a[i - 1] + a[i + 1]) - (1 << 6)) >> 7;
}

/* ---- 1-D CDF 5/3 float: src/libdwt.c:10986-11030 ---- */
void oracle_line_cdf53_f_s(float *a, int N)
{
	if (N < 2) {
		if (N == 1)
			a[0] = a[0] * C53_S1;
		return;
	}
	for (int i = 1; i < N - 2 + (N & 1); i += 2)
		a[i] -= C53_P1 * (a[i - 1] + a[i + 1]);
	if (N & 1)
		a[N - 1] += END2(C53_U1, a[N - 2]);
	else
		a[N - 1] -= END2(C53_P1, a[N - 2]);
	a[0] += END2(C53_U1, a[1]);
	for (int i = 2; i < N - (N & 1); i += 2)
		a[i] += C53_U1 * (a[i - 1] + a[i + 1]);
	for (int i = 0; i < N; i += 2)
		a[i] = a[i] * C53_S1;
	for (int i = 1; i < N; i += 2)
		a[i] = a[i] * C53_S2;
}

/* src/libdwt.c:11785-11829 */
void oracle_line_cdf53_i_s(float *a, int N)
{
	if (N < 2) {
		if (N == 1)
			a[0] = a[0] * C53_S2;
		return;
	}
	for (int i = 0; i < N; i += 2)
		a[i] = a[i] * C53_S2;
	for (int i = 1; i < N; i += 2)
		a[i] = a[i] * C53_S1;
	for (int i = 2; i < N - (N & 1); i += 2)
		a[i] -= C53_U1 * (a[i - 1] + a[i + 1]);
	a[0] -= END2(C53_U1, a[1]);
	if (N & 1)
		a[N - 1] -= END2(C53_U1, a[N - 2]);
	else
		a[N - 1] += END2(C53_P1, a[N - 2]);
	for (int i = 1; i < N - 2 + (N & 1); i += 2)
		a[i] += C53_P1 * (a[i - 1] + a[i + 1]);
}

/* ---- double precision line kernels, statement order of the reference kept ----
 * generic in the four constants: CDF 9/7 runs (p1,u1) then (p2,u2); CDF 5/3 only (p1,u1)
 * src/libdwt.c:2024-2083 (9/7 fwd), :11423-11482 (9/7 inv), :2085-2130, :11484-11530 (5/3) */
static void pu_fwd_d(double *a, int N, double p, double u)
{
	for (int i = 1; i < N - 2 + (N & 1); i += 2)
		a[i] -= p * (a[i - 1] + a[i + 1]);
	if (N & 1)
		a[N - 1] += END2(u, a[N - 2]);
	else
		a[N - 1] -= END2(p, a[N - 2]);
	a[0] += END2(u, a[1]);
	for (int i = 2; i < N - (N & 1); i += 2)
		a[i] += u * (a[i - 1] + a[i + 1]);
}

static void pu_inv_d(double *a, int N, double p, double u)
{
	for (int i = 2; i < N - (N & 1); i += 2)
		a[i] -= u * (a[i - 1] + a[i + 1]);
	a[0] -= END2(u, a[1]);
	if (N & 1)
		a[N - 1] -= END2(u, a[N - 2]);
	else
		a[N - 1] += END2(p, a[N - 2]);
	for (int i = 1; i < N - 2 + (N & 1); i += 2)
		a[i] += p * (a[i - 1] + a[i + 1]);
}

void oracle_line_cdf97_f_d(double *a, int N)
{
	if (N < 2) {
		if (N == 1)
			a[0] = a[0] * D97_S1;
		return;
	}
	pu_fwd_d(a, N, D97_P1, D97_U1);
	pu_fwd_d(a, N, D97_P2, D97_U2);
	for (int i = 0; i < N; i += 2)
		a[i] = a[i] * D97_S1;
	for (int i = 1; i < N; i += 2)
		a[i] = a[i] * D97_S2;
}

void oracle_line_cdf97_i_d(double *a, int N)
{
	if (N < 2) {
		if (N == 1)
			a[0] = a[0] * D97_S2;
		return;
	}
	for (int i = 0; i < N; i += 2)
		a[i] = a[i] * D97_S2;
	for (int i = 1; i < N; i += 2)
		a[i] = a[i] * D97_S1;
	pu_inv_d(a, N, D97_P2, D97_U2);
	pu_inv_d(a, N, D97_P1, D97_U1);
}

void oracle_line_cdf53_f_d(double *a, int N)
{
	if (N < 2) {
		if (N == 1)
			a[0] = a[0] * D53_S1;
		return;
	}
	pu_fwd_d(a, N, D53_P1, D53_U1);
	for (int i = 0; i < N; i += 2)
		a[i] = a[i] * D53_S1;
	for (int i = 1; i < N; i += 2)
		a[i] = a[i] * D53_S2;
}

void oracle_line_cdf53_i_d(double *a, int N)
{
	if (N < 2) {
		if (N == 1)
			a[0] = a[0] * D53_S2;
		return;
	}
	for (int i = 0; i < N; i += 2)
		a[i] = a[i] * D53_S2;
	for (int i = 1; i < N; i += 2)
		a[i] = a[i] * D53_S1;
	pu_inv_d(a, N, D53_P1, D53_U1);
}

/* ---- strided line drivers (gather -> lift -> Mallat scatter) ----
 * The gather/scatter is dwt_util_memcpy_stride_{s,i} (src/system.c:102-164). */
typedef void (*line_fn)(void *a, int N);

enum wavelet { W97S, W53I, W53S, W97D, W53D, W97I };

static int elem_size(enum wavelet w) { return (w == W97D || w == W53D) ? 8 : 4; }

static void lift_fwd(enum wavelet w, void *tmp, int N)
{
	switch (w) {
	case W97S: oracle_line_cdf97_f_s((float *)tmp, N); break;
	case W53I: oracle_line_cdf53_f_i((int *)tmp, N); break;
	case W53S: oracle_line_cdf53_f_s((float *)tmp, N); break;
	case W97D: oracle_line_cdf97_f_d((double *)tmp, N); break;
	case W53D: oracle_line_cdf53_f_d((double *)tmp, N); break;
	case W97I: oracle_line_cdf97_f_i((int *)tmp, N); break;
	}
}

static void lift_inv(enum wavelet w, void *tmp, int N)
{
	switch (w) {
	case W97S: oracle_line_cdf97_i_s((float *)tmp, N); break;
	case W53I: oracle_line_cdf53_i_i((int *)tmp, N); break;
	case W53S: oracle_line_cdf53_i_s((float *)tmp, N); break;
	case W97D: oracle_line_cdf97_i_d((double *)tmp, N); break;
	case W53D: oracle_line_cdf53_i_d((double *)tmp, N); break;
	case W97I: oracle_line_cdf97_i_i((int *)tmp, N); break;
	}
}

/* element access: 4- or 8-byte elements at arbitrary byte strides (possibly unaligned) */
static inline unsigned ld32(const char *p)
{
	unsigned v;
	memcpy(&v, p, 4);
	return v;
}

static inline void st32(char *p, unsigned v)
{
	memcpy(p, &v, 4);
}

static inline void zero_elem(char *p, int es)
{
	memset(p, 0, (size_t)es);
}

/* forward line: src -> (dst_l, dst_h); *_ex_stride_* at :10744, :10950, :10986, :2024 */
static void fwd_line(enum wavelet w, const char *src, char *dst_l, char *dst_h,
	unsigned *tmp32, int N, long stride)
{
	const int es = elem_size(w);
	char *tmp = (char *)tmp32;
	if (N < 2) {
		/* float/double kernels scale the lone sample; the int kernel leaves it (:10961) */
		if (N == 1 && w != W53I && w != W97I) {
			memcpy(tmp, src, (size_t)es);
			lift_fwd(w, tmp, 1);
			memcpy(dst_l, tmp, (size_t)es);
		}
		return;
	}
	for (int i = 0; i < N; i++)
		memcpy(tmp + (size_t)i * es, src + i * stride, (size_t)es);
	lift_fwd(w, tmp, N);
	const int nl = (N + 1) >> 1, nh = N >> 1;
	for (int i = 0; i < nl; i++)
		memcpy(dst_l + i * stride, tmp + (size_t)(2 * i) * es, (size_t)es);
	for (int i = 0; i < nh; i++)
		memcpy(dst_h + i * stride, tmp + (size_t)(2 * i + 1) * es, (size_t)es);
}

/* inverse line: (src_l, src_h) -> dst; :11530, :11749, :11785, :11423 */
static void inv_line(enum wavelet w, const char *src_l, const char *src_h, char *dst,
	unsigned *tmp32, int N, long stride)
{
	const int es = elem_size(w);
	char *tmp = (char *)tmp32;
	if (N < 2) {
		if (N == 1 && w != W53I && w != W97I) {
			memcpy(tmp, src_l, (size_t)es);
			lift_inv(w, tmp, 1);
			memcpy(dst, tmp, (size_t)es);
		}
		return;
	}
	const int nl = (N + 1) >> 1, nh = N >> 1;
	for (int i = 0; i < nl; i++)
		memcpy(tmp + (size_t)(2 * i) * es, src_l + i * stride, (size_t)es);
	for (int i = 0; i < nh; i++)
		memcpy(tmp + (size_t)(2 * i + 1) * es, src_h + i * stride, (size_t)es);
	lift_inv(w, tmp, N);
	for (int i = 0; i < N; i++)
		memcpy(dst + i * stride, tmp + (size_t)i * es, (size_t)es);
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

static unsigned *alloc_tmp(int n, int threads)
{
	unsigned *t = (unsigned *)malloc((size_t)threads * (size_t)(n + 8) * 2 * sizeof(unsigned));
	if (!t) {
		fprintf(stderr, "oracle: out of memory\n");
		abort();
	}
	return t;
}

static inline int thread_id(void)
{
#ifdef _OPENMP
	return omp_get_thread_num();
#else
	return 0;
#endif
}

/* dwt_zero_padding_f_stride_* (src/libdwt.c:12079-12131): zero [ceil(N/2),N_dst_L)
 * after dst_l and [floor(N/2),N_dst_H) after dst_h */
static void zero_pad_f(char *dst_l, char *dst_h, int N, int n_dst_l, int n_dst_h, long stride, int es)
{
	if (n_dst_l || n_dst_h) {
		for (int i = (N + 1) >> 1; i < n_dst_l; i++)
			zero_elem(dst_l + i * stride, es);
		for (int i = N >> 1; i < n_dst_h; i++)
			zero_elem(dst_h + i * stride, es);
	}
}

/* dwt_zero_padding_i_stride_* (src/libdwt.c:12161-12215) */
static void zero_pad_i(char *dst, int N, int n_dst, long stride, int es)
{
	for (int i = N; i < n_dst; i++)
		zero_elem(dst + i * stride, es);
}

/* Generic forward driver.  `skip_single` models the `lines_x > 1` / `lines_y > 1`
 * guards that only the CDF 9/7 float drivers have (:12837, :12867); the 5/3
 * drivers run their line kernels unconditionally (:16343-16359, :16507-16523).
 * src/dst follow the _s2 convention (:12709,:12742): a pass reads `src`, writes
 * `dst`, and afterwards src = dst. */
static void fwd_2d(enum wavelet w, int skip_single, const void *src0, void *dst,
	int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	const int so_min = imin(sox, soy), so_max = imax(sox, soy);
	const int threads = oracle_max_threads();
	unsigned *tmp_all = alloc_tmp(so_max, threads);
	const int j_limit = oracle_ceil_log2(decompose_one ? so_max : so_min);
	const char *src = (const char *)src0;
	char *d = (char *)dst;

	if (*j_max_ptr < 0 || *j_max_ptr > j_limit)
		*j_max_ptr = j_limit;

	for (int j = 0; j < *j_max_ptr; j++) {
		const int so_src_x = oracle_ceil_div_pow2(sox, j), so_src_y = oracle_ceil_div_pow2(soy, j);
		const int so_dst_x = oracle_ceil_div_pow2(sox, j + 1), so_dst_y = oracle_ceil_div_pow2(soy, j + 1);
		const int si_src_x = oracle_ceil_div_pow2(six, j), si_src_y = oracle_ceil_div_pow2(siy, j);

		if (!skip_single || so_src_x > 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
			for (int y = 0; y < so_src_y; y++) {
				unsigned *tmp = tmp_all + (size_t)thread_id() * (so_max + 8) * 2;
				fwd_line(w, src + (long)y * stride_x, d + (long)y * stride_x,
					d + (long)y * stride_x + (long)so_dst_x * stride_y,
					tmp, si_src_x, stride_y);
			}
			src = d;
		}
		if (!skip_single || so_src_y > 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
			for (int x = 0; x < so_src_x; x++) {
				unsigned *tmp = tmp_all + (size_t)thread_id() * (so_max + 8) * 2;
				fwd_line(w, src + (long)x * stride_y, d + (long)x * stride_y,
					d + (long)so_dst_y * stride_x + (long)x * stride_y,
					tmp, si_src_y, stride_x);
			}
			src = d;
		}
		if (zero_padding) {
			for (int y = 0; y < so_src_y; y++)
				zero_pad_f(d + (long)y * stride_x, d + (long)y * stride_x + (long)so_dst_x * stride_y,
					si_src_x, so_dst_x, so_src_x - so_dst_x, stride_y, elem_size(w));
			for (int x = 0; x < so_src_x; x++)
				zero_pad_f(d + (long)x * stride_y, d + (long)so_dst_y * stride_x + (long)x * stride_y,
					si_src_y, so_dst_y, so_src_y - so_dst_y, stride_x, elem_size(w));
		}
	}
	free(tmp_all);
}

/* Generic inverse driver.  `cols_first`: the int 5/3 inverse undoes columns before
 * rows (:18178-18195); the float drivers run rows then columns (:17098-17154,
 * :18333-18349). */
static void inv_2d(enum wavelet w, int skip_single, int cols_first, void *ptr,
	int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	const int so_min = imin(sox, soy), so_max = imax(sox, soy);
	const int threads = oracle_max_threads();
	unsigned *tmp_all = alloc_tmp(so_max, threads);
	char *p = (char *)ptr;
	int j = oracle_ceil_log2(decompose_one ? so_max : so_min);

	if (j_max >= 0 && j_max < j)
		j = j_max;

	for (; j > 0; j--) {
		const int so_src_x = oracle_ceil_div_pow2(sox, j), so_src_y = oracle_ceil_div_pow2(soy, j);
		const int so_dst_x = oracle_ceil_div_pow2(sox, j - 1), so_dst_y = oracle_ceil_div_pow2(soy, j - 1);
		const int si_dst_x = oracle_ceil_div_pow2(six, j - 1), si_dst_y = oracle_ceil_div_pow2(siy, j - 1);

		for (int pass = 0; pass < 2; pass++) {
			const int do_rows = cols_first ? (pass == 1) : (pass == 0);
			if (do_rows) {
				if (!skip_single || so_dst_x > 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
					for (int y = 0; y < so_dst_y; y++) {
						unsigned *tmp = tmp_all + (size_t)thread_id() * (so_max + 8) * 2;
						inv_line(w, p + (long)y * stride_x,
							p + (long)y * stride_x + (long)so_src_x * stride_y,
							p + (long)y * stride_x, tmp, si_dst_x, stride_y);
					}
				}
			} else {
				if (!skip_single || so_dst_y > 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
					for (int x = 0; x < so_dst_x; x++) {
						unsigned *tmp = tmp_all + (size_t)thread_id() * (so_max + 8) * 2;
						inv_line(w, p + (long)x * stride_y,
							p + (long)so_src_y * stride_x + (long)x * stride_y,
							p + (long)x * stride_y, tmp, si_dst_y, stride_x);
					}
				}
			}
		}
		if (zero_padding) {
			for (int y = 0; y < so_dst_y; y++)
				zero_pad_i(p + (long)y * stride_x, si_dst_x, so_dst_x, stride_y, elem_size(w));
			for (int x = 0; x < so_dst_x; x++)
				zero_pad_i(p + (long)x * stride_y, si_dst_y, so_dst_y, stride_x, elem_size(w));
		}
	}
	free(tmp_all);
}

void oracle_cdf97_2f_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	fwd_2d(W97S, 1, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding);
}

void oracle_cdf97_2f_s2(const void *src, void *dst, int stride_x, int stride_y, int sox, int soy,
	int six, int siy, int *j_max_ptr, int decompose_one, int zero_padding)
{
	fwd_2d(W97S, 1, src, dst, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding);
}

void oracle_cdf97_2i_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	inv_2d(W97S, 1, 0, ptr, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, zero_padding);
}

void oracle_cdf97_2i_s2(const void *src, void *dst, int stride_x, int stride_y, int sox, int soy,
	int six, int siy, int j_max, int decompose_one, int zero_padding)
{
	/* :18001-18008: copy the inner size_i region of src into dst, then in place */
	for (int y = 0; y < siy; y++)
		for (int x = 0; x < six; x++)
			st32((char *)dst + (long)y * stride_x + (long)x * stride_y,
				ld32((const char *)src + (long)y * stride_x + (long)x * stride_y));
	inv_2d(W97S, 1, 0, dst, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, zero_padding);
}

void oracle_cdf53_2f_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	fwd_2d(W53I, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding);
}

void oracle_cdf53_2i_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	inv_2d(W53I, 0, 1, ptr, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, zero_padding);
}

void oracle_cdf53_2f_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	fwd_2d(W53S, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding);
}

void oracle_cdf53_2i_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	inv_2d(W53S, 0, 0, ptr, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, zero_padding);
}

/* int32 CDF 9/7: src/libdwt.c:16387-16468 (rows, columns), :18219-18294 (columns, rows) */
void oracle_cdf97_2f_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	fwd_2d(W97I, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding);
}

void oracle_cdf97_2i_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	inv_2d(W97I, 0, 1, ptr, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, zero_padding);
}

/* double precision drivers: src/libdwt.c:12451, 16884 (9/7), :12535, :16962 (5/3); rows
 * then columns both ways, no single-line guards */
void oracle_cdf97_2f_d(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	fwd_2d(W97D, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding);
}

void oracle_cdf97_2i_d(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	inv_2d(W97D, 0, 0, ptr, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, zero_padding);
}

void oracle_cdf53_2f_d(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	fwd_2d(W53D, 0, ptr, ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, zero_padding);
}

void oracle_cdf53_2i_d(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	inv_2d(W53D, 0, 0, ptr, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, zero_padding);
}

/* ---- 3-D single level, interleaved in place: x lines, then y, then z ----
 * src/volume-dwt.c:677-725 (forward) applies fdwt1_single_cdf97_horizontal_min5_s
 * (src/dwt-simple.c:2166-2193 = prolog :580 + main :981 + epilog :1469), which is
 * the same lifting as the 2-D path without the de-interleave.  The inverse
 * (src/volume-dwt.c:1115-1163) runs the axes in the same x, y, z order. */
static void line3_s(char *base, long stride, int N, float *tmp, int inverse)
{
	for (int i = 0; i < N; i++)
		memcpy(&tmp[i], base + i * stride, 4);
	if (inverse)
		oracle_line_cdf97_i_s(tmp, N);
	else
		oracle_line_cdf97_f_s(tmp, N);
	for (int i = 0; i < N; i++)
		memcpy(base + i * stride, &tmp[i], 4);
}

static void vol3(void *ptr, long sx, long sy, long sz, int nx, int ny, int nz, int inverse)
{
	const int threads = oracle_max_threads();
	const int nmax = imax(nx, imax(ny, nz));
	float *tmp_all = (float *)alloc_tmp(nmax, threads);
	char *p = (char *)ptr;

#pragma omp parallel for schedule(static) num_threads(threads)
	for (int z = 0; z < nz; z++)
		for (int y = 0; y < ny; y++)
			line3_s(p + y * sy + z * sz, sx, nx, tmp_all + (size_t)thread_id() * (nmax + 8), inverse);
#pragma omp parallel for schedule(static) num_threads(threads)
	for (int z = 0; z < nz; z++)
		for (int x = 0; x < nx; x++)
			line3_s(p + x * sx + z * sz, sy, ny, tmp_all + (size_t)thread_id() * (nmax + 8), inverse);
#pragma omp parallel for schedule(static) num_threads(threads)
	for (int y = 0; y < ny; y++)
		for (int x = 0; x < nx; x++)
			line3_s(p + x * sx + y * sy, sz, nz, tmp_all + (size_t)thread_id() * (nmax + 8), inverse);
	free(tmp_all);
}

void oracle_cdf97_3f_s(void *ptr, long sx, long sy, long sz, int nx, int ny, int nz)
{
	vol3(ptr, sx, sy, sz, nx, ny, nz, 0);
}

void oracle_cdf97_3i_s(void *ptr, long sx, long sy, long sz, int nx, int ny, int nz)
{
	vol3(ptr, sx, sy, sz, nx, ny, nz, 1);
}

/* ---- 2-D transforms in the INTERLEAVED (in-place lifting) layout ----
 * No de-interleave: level j works on the stride-2^j lattice of the image, even lattice
 * index = low-pass, odd = high-pass (src/dwt-simple.c:2224-2354 fdwt2_cdf97_horizontal_s,
 * :2356-2489 fdwt2_cdf53_horizontal_s; src/libdwt.c:12926 dwt_cdf97_2f_inplace_s,
 * :17474 dwt_cdf97_2i_inplace_s, :16553 dwt_cdf53_2f_inplace_s, :17886
 * dwt_cdf53_2i_inplace_s).
 *
 * The 9/7 drivers and fdwt2_cdf53 do not finish the rows before they start the columns:
 * a line transform is cut into phases -- SHORT (whole line, lines too short for the
 * rest), PROLOG (first few coefficients), CORE (pairs), EPILOG (the remainder) -- and
 * each phase runs over all rows, then over all columns, before the next phase starts
 * (dwt-simple.c:2266-2350, libdwt.c:17517-17594).  Mathematically that is the separable
 * transform; in fp32 the rounding in the border bands depends on this order, so it is
 * restated: a phase is, per lifting step, the index range whose coefficients the step
 * updates (from dwt-simple.c:580-611 prolog, :981-1029 core, :1469-1528 epilog, :424-510
 * short; libdwt.c:9591-9668, 7661-7740, 9929-10010, 10375-10540 for the inverse).
 * The 5/3 `_inplace_` drivers are plain rows-then-columns (libdwt.c:16595-16604,
 * 17917-17926) over whole-line kernels (libdwt.c:11032-11067, 11831-11866). */
struct il_kind {
	int nsteps;      /* lifting steps: 4 (9/7) or 2 (5/3) */
	float c[4];      /* step coefficients in application order */
	float k_even, k_odd; /* scale factors: applied after the steps (forward) or before (inverse) */
	int inverse;     /* forward: step 0 updates odd indices; inverse: step 0 updates even ones */
	int min_phased;  /* shortest line the prolog/core/epilog split handles */
};

enum il_phase { IL_SHORT, IL_PROLOG, IL_CORE, IL_EPILOG };

static inline float *il_at(char *line, long stride, int i) { return (float *)(line + (long)i * stride); }

/* one lifting step restricted to target indices lo..hi of the step's parity */
static void il_step(char *line, long stride, int N, int parity, float c, int lo, int hi)
{
	if (lo < 0)
		lo = 0;
	if (hi > N - 1)
		hi = N - 1;
	if ((lo & 1) != parity)
		lo++;
	for (int t = lo; t <= hi; t += 2) {
		float *x = il_at(line, stride, t);
		if (t == 0 && g_end_form)
			*x += 2 * c * *il_at(line, stride, 1);
		else if (t == N - 1 && g_end_form)
			*x += 2 * c * *il_at(line, stride, N - 2);
		else if (t == 0 || t == N - 1) {
			const float m = *il_at(line, stride, t == 0 ? 1 : N - 2);
			*x += c * (m + m);
		} else
			*x += c * (*il_at(line, stride, t - 1) + *il_at(line, stride, t + 1));
	}
}

static void il_scale(char *line, long stride, int N, const struct il_kind *k, int lo, int hi)
{
	if (lo < 0)
		lo = 0;
	if (hi > N - 1)
		hi = N - 1;
	for (int t = lo; t <= hi; t++)
		*il_at(line, stride, t) *= (t & 1) ? k->k_odd : k->k_even;
}

static void il_line_phase(char *line, long stride, int N, const struct il_kind *k, enum il_phase ph)
{
	const int K = k->nsteps;
	int slo[4], shi[4], sc_lo, sc_hi;
	if (ph == IL_SHORT) {
		for (int s = 0; s < K; s++) { slo[s] = 0; shi[s] = N - 1; }
		sc_lo = 0; sc_hi = N - 1;
	} else if (!k->inverse) {
		const int M = (((N - 1) & ~1) - K) / 2; /* core pairs, counted from index 1 */
		for (int s = 0; s < K; s++) {
			if (ph == IL_PROLOG) { slo[s] = 0; shi[s] = K - 1 - s; }
			else if (ph == IL_CORE) { slo[s] = K + 1 - s; shi[s] = K - 1 - s + 2 * M; }
			else { slo[s] = K + 1 - s + 2 * M; shi[s] = N - 1; }
		}
		if (ph == IL_PROLOG) { sc_lo = 0; sc_hi = 0; }
		else if (ph == IL_CORE) { sc_lo = 1; sc_hi = 2 * M; }
		else { sc_lo = 2 * M + 1; sc_hi = N - 1; }
	} else {
		const int M = ((N & ~1) - K) / 2; /* core pairs, counted from index 0 */
		for (int s = 0; s < K; s++) {
			if (ph == IL_PROLOG) { slo[s] = 0; shi[s] = K - 2 - s; }
			else if (ph == IL_CORE) { slo[s] = K - s; shi[s] = K - 2 - s + 2 * M; }
			else { slo[s] = K - s + 2 * M; shi[s] = N - 1; }
		}
		if (ph == IL_PROLOG) { sc_lo = 0; sc_hi = K - 1; }
		else if (ph == IL_CORE) { sc_lo = K; sc_hi = K - 1 + 2 * M; }
		else { sc_lo = K + 2 * M; sc_hi = N - 1; }
	}
	if (k->inverse)
		il_scale(line, stride, N, k, sc_lo, sc_hi);
	for (int s = 0; s < K; s++)
		il_step(line, stride, N, k->inverse ? (s & 1) : !(s & 1), k->c[s], slo[s], shi[s]);
	if (!k->inverse)
		il_scale(line, stride, N, k, sc_lo, sc_hi);
}

/* one level on the lattice: phases outermost, rows before columns inside each phase */
static int il_dirs = 3; /* bit 0: rows, bit 1: columns (fdwt2h1_* / fdwt2v1_* transform one direction only) */

static void il_level_phased(char *ptr, long sx, long sy, int nx, int ny, const struct il_kind *k)
{
	for (int ph = IL_SHORT; ph <= IL_EPILOG; ph++) {
		if ((il_dirs & 1) && nx > 1 && (ph == IL_SHORT) == (nx < k->min_phased))
			for (int y = 0; y < ny; y++)
				il_line_phase(ptr + (long)y * sx, sy, nx, k, (enum il_phase)ph);
		if ((il_dirs & 2) && ny > 1 && (ph == IL_SHORT) == (ny < k->min_phased))
			for (int x = 0; x < nx; x++)
				il_line_phase(ptr + (long)x * sy, sx, ny, k, (enum il_phase)ph);
	}
}

/* one level, rows completely, then columns completely; single-sample lines are scaled */
static void il_level_separable(char *ptr, long sx, long sy, int nx, int ny, const struct il_kind *k, float single)
{
	for (int pass = 0; pass < 2; pass++) {
		const int n = pass ? ny : nx, lines = pass ? nx : ny;
		const long ls = pass ? sy : sx, es = pass ? sx : sy;
		for (int l = 0; l < lines; l++) {
			char *line = ptr + (long)l * ls;
			if (n == 1)
				*il_at(line, es, 0) *= single;
			else if (n > 1)
				il_line_phase(line, es, n, k, IL_SHORT);
		}
	}
}

static const struct il_kind IL_97_F = {4, {-C97_P1, C97_U1, -C97_P2, C97_U2}, C97_S1, 1 / C97_S1, 0, 5};
static const struct il_kind IL_97_I = {4, {-C97_U2, C97_P2, -C97_U1, C97_P1}, 1 / C97_S1, C97_S1, 1, 4};
static const struct il_kind IL_53_F_NEW = {2, {-C53_P1, C53_U1}, C53_S1, 1 / C53_S1, 0, 3};
static const struct il_kind IL_53_F = {2, {-C53_P1, C53_U1}, C53_S1, C53_S2, 0, 0};
static const struct il_kind IL_53_I = {2, {-C53_U1, C53_P1}, C53_S2, C53_S1, 1, 0};

static int il_levels(int sox, int soy, int j_max, int decompose_one)
{
	const int j_limit = oracle_ceil_log2(decompose_one ? imax(sox, soy) : imin(sox, soy));
	return (j_max < 0 || j_max > j_limit) ? j_limit : j_max;
}

static void il_forward(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy, int *j_max_ptr,
	int decompose_one, const struct il_kind *k, int phased)
{
	*j_max_ptr = il_levels(sox, soy, *j_max_ptr, decompose_one);
	for (int j = 0; j < *j_max_ptr; j++) {
		const int nx = oracle_ceil_div_pow2(six, j), ny = oracle_ceil_div_pow2(siy, j);
		if (phased)
			il_level_phased((char *)ptr, (long)stride_x << j, (long)stride_y << j, nx, ny, k);
		else
			il_level_separable((char *)ptr, (long)stride_x << j, (long)stride_y << j, nx, ny, k, C53_S1);
	}
}

static void il_inverse(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy, int j_max,
	int decompose_one, const struct il_kind *k, int phased)
{
	for (int j = il_levels(sox, soy, j_max, decompose_one); j > 0; j--) {
		const int nx = oracle_ceil_div_pow2(six, j - 1), ny = oracle_ceil_div_pow2(siy, j - 1);
		if (phased)
			il_level_phased((char *)ptr, (long)stride_x << (j - 1), (long)stride_y << (j - 1), nx, ny, k);
		else
			il_level_separable((char *)ptr, (long)stride_x << (j - 1), (long)stride_y << (j - 1), nx, ny, k, C53_S2);
	}
}

/* src/libdwt.c:12926 (and, with size_o == size_i, src/dwt-simple.c:2224, :1615, :3034:
 * the horizontal / vertical / diagonal schedules give identical bits) */
void oracle_cdf97_2f_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	il_forward(ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, &IL_97_F, 1);
}

/* src/libdwt.c:17474 */
void oracle_cdf97_2i_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	il_inverse(ptr, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, &IL_97_I, 1);
}

/* src/libdwt.c:16553 */
void oracle_cdf53_2f_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	il_forward(ptr, stride_x, stride_y, sox, soy, six, siy, j_max_ptr, decompose_one, &IL_53_F, 0);
}

/* src/libdwt.c:17886 */
void oracle_cdf53_2i_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	il_inverse(ptr, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, &IL_53_I, 0);
}

/* ---- fixed-point int 9/7, interleaved in place: src/libdwt.c:17424 (forward), :17308 (inverse) ----
 * The line kernels (:17356-17422, :17237-17306) add the rounded terms instead of subtracting
 * them as the Mallat int kernels do ((-203*s + 64) >> 7 is not -((203*s - 64) >> 7)), and the
 * drivers do NOT scale the strides by 2^j: level j re-transforms the dense top-left
 * ceil(size/2^j) block of the already interleaved image (the reference marks them "tested only
 * with j=1", :17423).  Restated as they are.  Forward: rows, then columns; inverse: columns,
 * then rows. */
static inline int *ii_at(char *line, long stride, int i) { return (int *)(line + (long)i * stride); }

static void ii_step(char *line, long stride, int N, int parity, int mul, int add, int shift, int sign)
{
	for (int t = parity; t < N; t += 2) {
		const int l = *ii_at(line, stride, t == 0 ? 1 : t - 1);
		const int r = *ii_at(line, stride, t == N - 1 ? N - 2 : t + 1);
		*ii_at(line, stride, t) += sign * ((mul * (l + r) + add) >> shift);
	}
}

static void ii_line(char *line, long stride, int N, int inverse)
{
	if (N < 2)
		return;
	if (!inverse) {
		ii_step(line, stride, N, 1, -203, 1 << 6, 7, +1);
		ii_step(line, stride, N, 0, -217, 1 << 11, 12, +1);
		ii_step(line, stride, N, 1, +113, 1 << 6, 7, +1);
		ii_step(line, stride, N, 0, 1817, 1 << 11, 12, +1);
	} else {
		ii_step(line, stride, N, 0, 1817, 1 << 11, 12, -1);
		ii_step(line, stride, N, 1, +113, 1 << 6, 7, -1);
		ii_step(line, stride, N, 0, -217, 1 << 11, 12, -1);
		ii_step(line, stride, N, 1, -203, 1 << 6, 7, -1);
	}
}

void oracle_cdf97_2f_inplace_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	*j_max_ptr = il_levels(sox, soy, *j_max_ptr, decompose_one);
	for (int j = 0; j < *j_max_ptr; j++) {
		const int nx = oracle_ceil_div_pow2(six, j), ny = oracle_ceil_div_pow2(siy, j);
		for (int y = 0; y < ny; y++)
			ii_line((char *)ptr + (long)y * stride_x, stride_y, nx, 0);
		for (int x = 0; x < nx; x++)
			ii_line((char *)ptr + (long)x * stride_y, stride_x, ny, 0);
	}
}

void oracle_cdf97_2i_inplace_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding)
{
	(void)zero_padding;
	for (int j = il_levels(sox, soy, j_max, decompose_one); j > 0; j--) {
		const int nx = oracle_ceil_div_pow2(six, j - 1), ny = oracle_ceil_div_pow2(siy, j - 1);
		for (int x = 0; x < nx; x++)
			ii_line((char *)ptr + (long)x * stride_y, stride_x, ny, 1);
		for (int y = 0; y < ny; y++)
			ii_line((char *)ptr + (long)y * stride_x, stride_y, nx, 1);
	}
}

/* src/dwt-simple.c:2224 (fdwt2_cdf97_{horizontal,vertical,diagonal}_s) */
void oracle_fdwt2_cdf97_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one)
{
	il_forward(ptr, stride_x, stride_y, size_x, size_y, size_x, size_y, j_max_ptr, decompose_one, &IL_97_F, 1);
}

/* src/dwt-simple.c:1747 (fdwt2h1_cdf97_vertical_s: the rows of every level only) and :1837
 * (fdwt2v1_cdf97_vertical_s: the columns only) */
void oracle_fdwt2h1_cdf97_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one)
{
	il_dirs = 1;
	il_forward(ptr, stride_x, stride_y, size_x, size_y, size_x, size_y, j_max_ptr, decompose_one, &IL_97_F, 1);
	il_dirs = 3;
}

void oracle_fdwt2v1_cdf97_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one)
{
	il_dirs = 2;
	il_forward(ptr, stride_x, stride_y, size_x, size_y, size_x, size_y, j_max_ptr, decompose_one, &IL_97_F, 1);
	il_dirs = 3;
}

/* src/dwt-simple.c:2356 (fdwt2_cdf53_{horizontal,vertical,diagonal}_s): phased like the 9/7
 * one, odd coefficients scaled by 1/zeta computed in float, single-sample lines untouched */
void oracle_fdwt2_cdf53_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one)
{
	il_forward(ptr, stride_x, stride_y, size_x, size_y, size_x, size_y, j_max_ptr, decompose_one, &IL_53_F_NEW, 1);
}

/* ---- libdwt's synthetic test patterns ---- */
/* float, type 0: x,y made 1-based, x >>= rand, 2xy/(float)(x^2+y^2+1)
 * (src/libdwt.c:1209-1217, filled by :1338) */
void oracle_test_image_fill_s(void *ptr, int stride_x, int stride_y, int size_x, int size_y, int rnd)
{
	for (int y = 0; y < size_y; y++)
		for (int x = 0; x < size_x; x++) {
			int xx = x + 1, yy = y + 1;
			xx >>= rnd;
			const float v = 2 * xx * yy / (float)(xx * xx + yy * yy + 1);
			memcpy((char *)ptr + (long)y * stride_x + (long)x * stride_y, &v, 4);
		}
}

/* int, type 0: 0-based, 255*(2xy)/(x^2+y^2+1) in integer arithmetic
 * (src/libdwt.c:1152-1155, filled by :1270) */
void oracle_test_image_fill_i(void *ptr, int stride_x, int stride_y, int size_x, int size_y, int rnd)
{
	for (int y = 0; y < size_y; y++)
		for (int x = 0; x < size_x; x++) {
			int xx = x;
			xx >>= rnd;
			const int v = 255 * (2 * xx * y) / (xx * xx + y * y + 1);
			memcpy((char *)ptr + (long)y * stride_x + (long)x * stride_y, &v, 4);
		}
}
