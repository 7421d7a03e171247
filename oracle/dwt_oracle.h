/*
 * dwt_oracle.h -- CPU restatement of libdwt's 2-D lifting DWT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load
 * or call it, and there only as the checker.  The product path (libdwt_amd/) never
 * links or calls this file.
 *
 * Parity status: PINNED.  The reference ships no golden vectors,
 * so the restatement is pinned against outputs of the reference itself compiled
 * from its own sources where they lie (oracle/Makefile -> oracle/_ref/libdwt_ref.so)
 * and against the committed fixtures in tests/golden/ generated from that build by
 * oracle/gen_golden.py.  tests/test_oracle_vs_ref.py checks bit equality.
 *
 * Every function cites the reference lines (relative to /root/reference/) whose
 * behaviour it restates.  All symbols carry the oracle_ prefix so this library can
 * sit in one process with the product library and with libdwt_ref.so.
 */
#ifndef DWT_ORACLE_H
#define DWT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* 2-D multi-level, in place, Mallat layout.  Image element (y,x) lives at
 * (char*)ptr + y*stride_x + x*stride_y   (src/inline.h:180-189). */

/* src/libdwt.c:12776-12924 */
void oracle_cdf97_2f_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
/* src/libdwt.c:17040-17180 */
void oracle_cdf97_2i_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);
/* src/libdwt.c:12619-12774 */
void oracle_cdf97_2f_s2(const void *src, void *dst, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
/* src/libdwt.c:17985-18140 */
void oracle_cdf97_2i_s2(const void *src, void *dst, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);
/* src/libdwt.c:16304-16385 */
void oracle_cdf53_2f_i(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
/* src/libdwt.c:18142-18217 */
void oracle_cdf53_2i_i(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);
/* src/libdwt.c:16470-16546 */
void oracle_cdf53_2f_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
/* src/libdwt.c:18296-18372 */
void oracle_cdf53_2i_s(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);

/* int32 CDF 9/7 (fixed point): src/libdwt.c:16387, 18219; line kernels :10901, :11699 */
void oracle_cdf97_2f_i(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void oracle_cdf97_2i_i(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);
void oracle_line_cdf97_f_i(int *a, int N);
void oracle_line_cdf97_i_i(int *a, int N);

/* double precision: src/libdwt.c:12451, 16884, 12535, 16962 */
void oracle_cdf97_2f_d(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void oracle_cdf97_2i_d(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);
void oracle_cdf53_2f_d(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j_max_ptr, int decompose_one, int zero_padding);
void oracle_cdf53_2i_d(void *ptr, int stride_x, int stride_y,
	int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int j_max, int decompose_one, int zero_padding);
void oracle_line_cdf97_f_d(double *a, int N);  /* src/libdwt.c:2024-2083 */
void oracle_line_cdf97_i_d(double *a, int N);  /* src/libdwt.c:11423-11482 */
void oracle_line_cdf53_f_d(double *a, int N);  /* src/libdwt.c:2085-2130 */
void oracle_line_cdf53_i_d(double *a, int N);  /* src/libdwt.c:11484-11530 */

/* 1-D line kernels on a dense temporary (the arithmetic of the path). */
void oracle_line_cdf97_f_s(float *a, int N);   /* src/libdwt.c:10744-10800 + 10551 */
void oracle_line_cdf97_i_s(float *a, int N);   /* src/libdwt.c:11530-11571 */
void oracle_line_cdf53_f_i(int *a, int N);     /* src/libdwt.c:10950-10984 */
void oracle_line_cdf53_i_i(int *a, int N);     /* src/libdwt.c:11749-11783 */
void oracle_line_cdf53_f_s(float *a, int N);   /* src/libdwt.c:10986-11030 */
void oracle_line_cdf53_i_s(float *a, int N);   /* src/libdwt.c:11785-11829 */

/* 3-D single level, interleaved in-place layout (src/volume-dwt.c:677-785,
 * src/dwt-simple.c:2166-2193); strides follow volume_t naming: stride_x = element,
 * stride_y = row, stride_z = slice (src/volume.h:14-24). */
void oracle_cdf97_3f_s(void *ptr, long stride_x, long stride_y, long stride_z,
	int size_x, int size_y, int size_z);
void oracle_cdf97_3i_s(void *ptr, long stride_x, long stride_y, long stride_z,
	int size_x, int size_y, int size_z);

/* interleaved (in-place lifting) layout: src/libdwt.c:12926, 17474, 16553, 17886;
 * src/dwt-simple.c:2224 (= :1615, :3034), :2356 (= :1927, :3166) */
void oracle_cdf97_2f_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding);
void oracle_cdf97_2i_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding);
void oracle_cdf53_2f_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding);
void oracle_cdf53_2i_inplace_s(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding);
void oracle_cdf97_2f_inplace_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int *j_max_ptr, int decompose_one, int zero_padding); /* src/libdwt.c:17424 */
void oracle_cdf97_2i_inplace_i(void *ptr, int stride_x, int stride_y, int sox, int soy, int six, int siy,
	int j_max, int decompose_one, int zero_padding);      /* src/libdwt.c:17308 */
void oracle_fdwt2_cdf97_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one);
void oracle_fdwt2h1_cdf97_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one); /* src/dwt-simple.c:1747 */
void oracle_fdwt2v1_cdf97_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one); /* src/dwt-simple.c:1837 */
void oracle_fdwt2_cdf53_s(void *ptr, int size_x, int size_y, int stride_x, int stride_y, int *j_max_ptr, int decompose_one);

/* libdwt's synthetic inputs (src/libdwt.c:1201-1244, 1142-1167, 1338, 1270). */
void oracle_test_image_fill_s(void *ptr, int stride_x, int stride_y, int size_x, int size_y, int rnd);
void oracle_test_image_fill_i(void *ptr, int stride_x, int stride_y, int size_x, int size_y, int rnd);

/* helpers restated from src/inline.h:443-461 */
int oracle_ceil_log2(int x);
int oracle_ceil_div_pow2(int i, int j);

/* number of OpenMP threads the 2-D drivers will use (1 if built without OpenMP) */
int oracle_max_threads(void);
void oracle_set_threads(int n);
/* float / double line ends: 1 (default) = the reference's own (2c)*x, 0 = the reflected c*(x+x) the
 * HIP kernels evaluate -- the two differ only where x+x overflows (dwt_oracle.c header) */
void oracle_set_end_form(int faithful);

#ifdef __cplusplus
}
#endif
#endif
