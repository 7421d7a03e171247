/*
 * volume-dwt.h -- libdwt's single-level 3-D CDF 9/7 float transforms on struct volume_t, drop-in
 * subset of the reference's src/volume-dwt.h (written fresh; same names and argument meaning).
 * Interleaved in-place layout (even index = low-pass, odd = high-pass along every axis), x lines,
 * then y, then z, forward and inverse alike (src/volume-dwt.c:677-785, :1115-1163).
 *
 * Every function takes host or device volumes (volume.h).  The reference ships ten loop schedules
 * of the out-of-place forward transform (its `enum volume_approach` 0..9: separable, slice-wise,
 * 2x2x2 / 4x4x2 / 4x4x4 cores, ...).  They are the same transform; here all of them run the ONE
 * fused x+y+z kernel and return the bits of the separable schedules cdf97_3f_op_sep_horizontal_s /
 * _vertical_s (the two have the same bits; the reference's core schedules 2..9 differ from them by at
 * most 1.07e-6 absolute = 4e-7 of the largest coefficient, measured on the reference itself:
 * tests/test_oracle.py::test_distance_between_the_references_own_3d_schedules).  A failure (no
 * usable device, bad strides) is logged and abort()s, as dwt_util_error does in the reference.
 */
#ifndef VOLUME_DWT_H
#define VOLUME_DWT_H

#include "volume.h"

#ifdef __cplusplus
extern "C" {
#endif

/* src/volume-dwt.h:21 (src/volume-dwt.c:677-725): forward, in place */
void cdf97_3f_ip_sep_horizontal_s(struct volume_t *volume);
/* src/volume-dwt.h:38 (src/volume-dwt.c:727-785): forward, out of place; strides may differ */
void cdf97_3f_op_sep_horizontal_s(struct volume_t *volume_src, struct volume_t *volume_dst);
/* src/volume-dwt.h:55-191: the other schedules of the same transform */
void cdf97_3f_op_sep_vertical_s(struct volume_t *volume_src, struct volume_t *volume_dst);
void cdf97_3f_op_slices_vert4x4_s(struct volume_t *volume_src, struct volume_t *volume_dst);
void cdf97_3f_op_baseline_vert2x2x2_s(struct volume_t *volume_src, struct volume_t *volume_dst);
void cdf97_3f_op_HORIZ_vert2x2x2_s(struct volume_t *volume_src, struct volume_t *volume_dst);
void cdf97_3f_op_cube_vert4x4x2_s(struct volume_t *volume_src, struct volume_t *volume_dst);
void cdf97_3f_op_HORIZ_vert4x4x2_s(struct volume_t *volume_src, struct volume_t *volume_dst);
void cdf97_3f_op_baseline_diag2x2x2_s(struct volume_t *volume_src, struct volume_t *volume_dst);
void cdf97_3f_op_HORIZ_diag2x2x2_s(struct volume_t *volume_src, struct volume_t *volume_dst);
void cdf97_3f_op_HORIZ_vert4x4x4_s(struct volume_t *volume_src, struct volume_t *volume_dst);
/* src/volume-dwt.h:208 (src/volume-dwt.c:1115-1163): inverse, in place */
void cdf97_3i_ip_sep_horizontal_s(struct volume_t *volume);

/* src/volume-dwt.h:210-225 */
enum volume_approach {
	VOL_SEP_HORIZONTAL = 0,
	VOL_SEP_VERTICAL = 1,
	VOL_SLICES_VERT4X4 = 2,
	VOL_BASELINE_VERT2X2X2 = 3,
	VOL_HORIZ_VERT2X2X2 = 4,
	VOL_BASELINE_VERT4X4X2 = 5,
	VOL_HORIZ_VERT4X4X2 = 6,
	VOL_HORIZ_VERT4X4X4 = 7,
	VOL_BASELINE_DIAG2X2X2 = 8,
	VOL_HORIZ_DIAG2X2X2 = 9,
	VOL_SEP_HORIZONTAL_X = 10, /* x lines only: copy, then lift (src/volume-dwt.c:788) */
	VOL_SEP_HORIZONTAL_Y = 11, /* y lines only, in place on the destination (:852) */
	VOL_SEP_HORIZONTAL_Z = 12, /* z lines only, in place on the destination (:918) */
	VOL_LAST
};

/* src/volume-dwt.h:227 (src/volume-dwt.c:2787-2808) */
void cdf97_3f_op_wrapper_s(struct volume_t *volume_src, struct volume_t *volume_dst, enum volume_approach approach);

/* src/volume-dwt.h:234-241 (src/volume-dwt.c:2810-2881): N times { volume_fill_s, forward out of
 * place (timed), inverse in place, compare }; *secs = best seconds per voxel; returns the number of
 * failed comparisons.  Host volumes as in the reference: the time includes the PCIe transfers.
 * *faults is 0 (no page-fault counting here). */
int volume_perftest_fwd97op_s(int size, int opt_stride, enum volume_approach approach, int N, double *secs, long unsigned *faults);
/* the same protocol with both volumes resident in HBM (not in the reference) */
int volume_perftest_fwd97op_device_s(int size, int opt_stride, enum volume_approach approach, int N, double *secs);
/* src/volume-dwt.h:246-253 (src/volume-dwt.c:2898-2960): sizes size_min, grow(size) ... < size_max;
 * writes data/perftest/time-stride=S-approach=A.txt and faults-...txt like the reference */
int volume_measure_fwd97op_s(int size_min, int size_max, int size_step, int N, int opt_stride, enum volume_approach approach);

#ifdef __cplusplus
}
#endif
#endif
