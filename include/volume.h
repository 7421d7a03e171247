/*
 * volume.h -- the 3-D container of libdwt's volume path, drop-in subset (written fresh; same
 * names, field order and argument meaning as the reference's src/volume.h:14-24, :29-100).
 *
 * A volume's samples live at data + x*stride_x + y*stride_y + z*stride_z (bytes).  `data` may
 * be host memory (what volume_alloc_realiably* return: every transform stages it through HBM)
 * or device memory (volume_alloc_device / dwt_hip_malloc: nothing crosses PCIe; the helpers
 * below that touch samples on the CPU -- fill, copy, compare, save -- stage such volumes).
 * float samples only (stride_x == sizeof(float)), as in the reference's 3-D transforms.
 */
#ifndef VOLUME_H
#define VOLUME_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/volume.h:14-24 */
struct volume_t {
	int size_x; /* columns */
	int size_y; /* rows */
	int size_z; /* slices */

	size_t stride_x; /* sizeof(sample) */
	size_t stride_y; /* bytes from row to row */
	size_t stride_z; /* bytes from slice to slice */

	void *data;
};

/* src/volume.c:10-31: strides from dwt_util_get_stride(.., opt_stride); host memory; aborts on failure */
struct volume_t *volume_alloc_realiably(size_t pix_size, int size_x, int size_y, int size_z, int opt_stride);
/* src/volume.c:193-219: the same in page-locked memory (here: pinned for the GPU's DMA engines) */
struct volume_t *volume_alloc_realiably_locked(size_t pix_size, int size_x, int size_y, int size_z, int opt_stride);
/* not in the reference: the same layout resident in HBM (freed by volume_free like the others) */
struct volume_t *volume_alloc_device(size_t pix_size, int size_x, int size_y, int size_z, int opt_stride);
/* src/volume.c:33-39 */
void volume_free(struct volume_t *volume);

/* src/volume.c:41-66: slice z = libdwt's 2-D test pattern type 0 with rand = z & 11 folded at 5 */
void volume_fill_s(struct volume_t *volume);
/* src/volume.c:68-97: sizes must agree, strides may differ; returns 0 */
int volume_copy_s(struct volume_t *volume_dst, struct volume_t *volume_src);
/* src/volume.c:99-132: 0 equal (|a-b| <= 1e-3, no NaN / Inf), non-zero otherwise */
int volume_compare_s(struct volume_t *volume_l, struct volume_t *volume_r);
/* src/volume.c:134-163: one PGM per slice, `path` is a printf format taking the slice number */
void volume_save_to_pgm_s(struct volume_t *volume, const char *path);
/* src/volume.h:70 (src/volume.c:165-193): the same in a logarithmic scale (dwt_util_save_log_to_pgm_s per slice) */
void volume_save_log_to_pgm_s(struct volume_t *volume, const char *path);
/* src/volume.c:221-225: nothing to flush on the device path; kept for source compatibility */
void volume_invalidate_cache(struct volume_t *volume);

/* src/volume.h:41-48, :81-88 (address arithmetic only: valid for host and device volumes) */
static inline void *volume_get_slice(struct volume_t *volume, int pos_z)
{
	return (char *)volume->data + (size_t)pos_z * volume->stride_z;
}

static inline void *volume_get_pix(struct volume_t *volume, int x, int y, int z)
{
	return (char *)volume->data + (size_t)x * volume->stride_x + (size_t)y * volume->stride_y + (size_t)z * volume->stride_z;
}

#ifdef __cplusplus
}
#endif
#endif
