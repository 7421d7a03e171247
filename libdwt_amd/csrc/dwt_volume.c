/*
 * dwt_volume.c -- the host side of libdwt's 3-D path: struct volume_t housekeeping
 * (src/volume.c) and the typed transform entries, schedule dispatcher and perf test of
 * src/volume-dwt.c, as thin C over the device backend (include/libdwt_hip.h).  No CPU transform
 * here: a call that cannot run on the device logs the reason and aborts (dwt_util_error).
 */
#include "../../include/libdwt.h"
#include "../../include/libdwt_hip.h"
#include "../../include/volume-dwt.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* what kind of memory a volume's data is: decided at allocation, needed by volume_free */
enum vol_mem { VOL_MEM_HOST = 0, VOL_MEM_PINNED = 1, VOL_MEM_DEVICE = 2 };
struct vol_rec {
	struct volume_t vol; /* first member: the pointer handed out */
	enum vol_mem mem;
	unsigned magic;
};
#define VOL_MAGIC 0x766f6c33u

static struct volume_t *vol_alloc(size_t pix_size, int size_x, int size_y, int size_z, int opt_stride, enum vol_mem mem)
{
	if (pix_size != sizeof(float) || size_x < 1 || size_y < 1 || size_z < 1)
		dwt_util_error("volume_alloc: float volumes with positive sizes only\n");
	struct vol_rec *r = calloc(1, sizeof *r);
	if (!r)
		dwt_util_error("volume_alloc: out of memory\n");
	r->mem = mem;
	r->magic = VOL_MAGIC;
	struct volume_t *v = &r->vol;
	v->size_x = size_x;
	v->size_y = size_y;
	v->size_z = size_z;
	/* src/volume.c:18-20.  (dwt_util_get_stride works in int like the reference's: rows of less than 2 GiB) */
	v->stride_x = pix_size;
	v->stride_y = (size_t)dwt_util_get_stride((int)(v->stride_x * (size_t)size_x), opt_stride);
	if (mem == VOL_MEM_DEVICE) /* libdwt's "optimal" strides are odd byte counts (CPU cache aliasing); the device wants whole samples */
		v->stride_y = (v->stride_y + 3) & ~(size_t)3;
	const size_t slice = v->stride_y * (size_t)size_y;
	v->stride_z = slice < 0x7fffffffu ? (size_t)dwt_util_get_stride((int)slice, opt_stride) : slice;
	if (mem == VOL_MEM_DEVICE)
		v->stride_z = (v->stride_z + 3) & ~(size_t)3;
	const size_t total = v->stride_z * (size_t)size_z;
	if (mem == VOL_MEM_DEVICE || mem == VOL_MEM_PINNED) {
		dwt_util_init();
		v->data = mem == VOL_MEM_DEVICE ? dwt_hip_malloc(total) : dwt_hip_malloc_host(total);
	} else if (posix_memalign(&v->data, 64, total)) {
		v->data = NULL;
	}
	if (!v->data)
		dwt_util_error("volume_alloc: cannot allocate %zu bytes (%s)\n", total, mem == VOL_MEM_HOST ? "host" : dwt_hip_last_error());
	return v;
}

struct volume_t *volume_alloc_realiably(size_t pix_size, int size_x, int size_y, int size_z, int opt_stride)
{
	return vol_alloc(pix_size, size_x, size_y, size_z, opt_stride, VOL_MEM_HOST);
}

struct volume_t *volume_alloc_realiably_locked(size_t pix_size, int size_x, int size_y, int size_z, int opt_stride)
{
	return vol_alloc(pix_size, size_x, size_y, size_z, opt_stride, VOL_MEM_PINNED);
}

struct volume_t *volume_alloc_device(size_t pix_size, int size_x, int size_y, int size_z, int opt_stride)
{
	return vol_alloc(pix_size, size_x, size_y, size_z, opt_stride, VOL_MEM_DEVICE);
}

void volume_free(struct volume_t *volume)
{
	if (!volume)
		return;
	struct vol_rec *r = (struct vol_rec *)volume;
	if (r->magic != VOL_MAGIC)
		dwt_util_error("volume_free: not a volume of volume_alloc_*\n");
	if (r->mem == VOL_MEM_DEVICE)
		dwt_hip_free(volume->data);
	else if (r->mem == VOL_MEM_PINNED)
		dwt_hip_free_host(volume->data);
	else
		free(volume->data);
	r->magic = 0;
	free(r);
}

/* a host image of the volume's bytes for the helpers that touch samples on the CPU */
static void *host_view(struct volume_t *v, int load)
{
	if (!dwt_hip_is_device_pointer(v->data))
		return v->data;
	const size_t total = v->stride_z * (size_t)v->size_z;
	void *h = malloc(total);
	if (!h)
		dwt_util_error("volume: out of host memory (%zu bytes)\n", total);
	if (load && dwt_hip_memcpy_d2h(h, v->data, total))
		dwt_util_error("volume: %s\n", dwt_hip_last_error());
	return h;
}

static void host_view_done(struct volume_t *v, void *h, int store)
{
	if (h == v->data)
		return;
	if (store && dwt_hip_memcpy_h2d(v->data, h, v->stride_z * (size_t)v->size_z))
		dwt_util_error("volume: %s\n", dwt_hip_last_error());
	free(h);
}

/* src/volume.c:41-66 */
void volume_fill_s(struct volume_t *volume)
{
	char *h = host_view(volume, 1);
	for (int z = 0; z < volume->size_z; z++) {
		int rnd = z & 11;
		if (rnd > 11 / 2)
			rnd = 11 - rnd;
		dwt_util_test_image_fill2_s(h + (size_t)z * volume->stride_z, (int)volume->stride_y, (int)volume->stride_x,
			volume->size_x, volume->size_y, rnd, 0);
	}
	host_view_done(volume, h, 1);
}

static int same_sizes(const struct volume_t *a, const struct volume_t *b)
{
	return a->size_x == b->size_x && a->size_y == b->size_y && a->size_z == b->size_z;
}

/* src/volume.c:68-97 */
int volume_copy_s(struct volume_t *volume_dst, struct volume_t *volume_src)
{
	if (!same_sizes(volume_dst, volume_src))
		dwt_util_error("volume_copy_s: sizes differ\n");
	char *s = host_view(volume_src, 1), *d = host_view(volume_dst, 1);
	for (int z = 0; z < volume_src->size_z; z++)
		for (int y = 0; y < volume_src->size_y; y++) {
			const char *ps = s + (size_t)z * volume_src->stride_z + (size_t)y * volume_src->stride_y;
			char *pd = d + (size_t)z * volume_dst->stride_z + (size_t)y * volume_dst->stride_y;
			for (int x = 0; x < volume_src->size_x; x++)
				memcpy(pd + (size_t)x * volume_dst->stride_x, ps + (size_t)x * volume_src->stride_x, sizeof(float));
		}
	host_view_done(volume_src, s, 0);
	host_view_done(volume_dst, d, 1);
	return 0;
}

/* src/volume.c:99-132 with dwt_util_compare2_s (src/libdwt.c:1622-1672): the first differing slice ends it */
int volume_compare_s(struct volume_t *volume_l, struct volume_t *volume_r)
{
	if (!same_sizes(volume_l, volume_r))
		dwt_util_error("volume_compare_s: sizes differ\n");
	char *l = host_view(volume_l, 1), *r = host_view(volume_r, 1);
	int code = 0;
	for (int z = 0; z < volume_r->size_z && !code; z++)
		for (int y = 0; y < volume_r->size_y; y++) {
			const char *pl = l + (size_t)z * volume_l->stride_z + (size_t)y * volume_l->stride_y;
			const char *pr = r + (size_t)z * volume_r->stride_z + (size_t)y * volume_r->stride_y;
			for (int x = 0; x < volume_r->size_x; x++) {
				float a, b;
				memcpy(&a, pl + (size_t)x * volume_l->stride_x, sizeof a);
				memcpy(&b, pr + (size_t)x * volume_r->stride_x, sizeof b);
				if (isnan(a) || isinf(a) || isnan(b) || isinf(b) || fabsf(a - b) > 1.e-3f)
					code = 1;
			}
		}
	host_view_done(volume_l, l, 0);
	host_view_done(volume_r, r, 0);
	return code;
}

/* src/volume.c:134-163 */
void volume_save_to_pgm_s(struct volume_t *volume, const char *path)
{
	char *h = host_view(volume, 1);
	for (int z = 0; z < volume->size_z; z++) {
		char file_name[4096];
		snprintf(file_name, sizeof file_name, path, z);
		dwt_util_save_to_pgm_s(file_name, 1.f, h + (size_t)z * volume->stride_z, (int)volume->stride_y, (int)volume->stride_x,
			volume->size_x, volume->size_y);
	}
	host_view_done(volume, h, 0);
}

/* src/volume.c:165-193 */
void volume_save_log_to_pgm_s(struct volume_t *volume, const char *path)
{
	char *h = host_view(volume, 1);
	for (int z = 0; z < volume->size_z; z++) {
		char file_name[4096];
		snprintf(file_name, sizeof file_name, path, z);
		dwt_util_save_log_to_pgm_s(file_name, h + (size_t)z * volume->stride_z, (int)volume->stride_y, (int)volume->stride_x,
			volume->size_x, volume->size_y);
	}
	host_view_done(volume, h, 0);
}

void volume_invalidate_cache(struct volume_t *volume)
{
	(void)volume; /* src/volume.c:221-225 flushes the CPU's caches before a timed run; HBM has no such state to reset */
}

/* ---- transforms ---- */

static void check_pix(const struct volume_t *v, const char *who)
{
	if (!v || !v->data)
		dwt_util_error("%s: no volume\n", who);
	if (v->stride_x != sizeof(float))
		dwt_util_error("%s: float volumes with dense rows only (stride_x = %zu)\n", who, v->stride_x);
}

static void forward_op(struct volume_t *src, struct volume_t *dst, int dirs, const char *who)
{
	check_pix(dst, who);
	if (dirs == 7 || dirs == 1) {
		check_pix(src, who);
		if (!same_sizes(src, dst))
			dwt_util_error("%s: sizes differ\n", who);
	}
	if (dwt_hip_volume_fwd_op(src ? src->data : NULL, src ? src->stride_y : 0, src ? src->stride_z : 0, dst->data, dst->stride_y,
			dst->stride_z, dst->size_x, dst->size_y, dst->size_z, dirs))
		dwt_util_error("%s: %s\n", who, dwt_hip_last_error());
}

void cdf97_3f_ip_sep_horizontal_s(struct volume_t *volume)
{
	check_pix(volume, __func__);
	if (dwt_hip_volume_ip(0, volume->data, volume->stride_y, volume->stride_z, volume->size_x, volume->size_y, volume->size_z))
		dwt_util_error("%s: %s\n", __func__, dwt_hip_last_error());
}

void cdf97_3i_ip_sep_horizontal_s(struct volume_t *volume)
{
	check_pix(volume, __func__);
	if (dwt_hip_volume_ip(1, volume->data, volume->stride_y, volume->stride_z, volume->size_x, volume->size_y, volume->size_z))
		dwt_util_error("%s: %s\n", __func__, dwt_hip_last_error());
}

/* the ten schedules of the one transform (src/volume-dwt.c:727, :983, :1165 ...): one kernel here */
#define SAME_TRANSFORM(name) \
	void name(struct volume_t *volume_src, struct volume_t *volume_dst) { forward_op(volume_src, volume_dst, 7, #name); }
SAME_TRANSFORM(cdf97_3f_op_sep_horizontal_s)
SAME_TRANSFORM(cdf97_3f_op_sep_vertical_s)
SAME_TRANSFORM(cdf97_3f_op_slices_vert4x4_s)
SAME_TRANSFORM(cdf97_3f_op_baseline_vert2x2x2_s)
SAME_TRANSFORM(cdf97_3f_op_HORIZ_vert2x2x2_s)
SAME_TRANSFORM(cdf97_3f_op_cube_vert4x4x2_s)
SAME_TRANSFORM(cdf97_3f_op_HORIZ_vert4x4x2_s)
SAME_TRANSFORM(cdf97_3f_op_baseline_diag2x2x2_s)
SAME_TRANSFORM(cdf97_3f_op_HORIZ_diag2x2x2_s)
SAME_TRANSFORM(cdf97_3f_op_HORIZ_vert4x4x4_s)

/* src/volume-dwt.c:2787-2808 */
void cdf97_3f_op_wrapper_s(struct volume_t *volume_src, struct volume_t *volume_dst, enum volume_approach approach)
{
	if ((int)approach < 0 || approach >= VOL_LAST)
		dwt_util_error("cdf97_3f_op_wrapper_s: unknown approach %d\n", (int)approach);
	const int dirs = approach == VOL_SEP_HORIZONTAL_X ? 1 : approach == VOL_SEP_HORIZONTAL_Y ? 2 : approach == VOL_SEP_HORIZONTAL_Z ? 4 : 7;
	forward_op(volume_src, volume_dst, dirs, __func__);
}

/* src/volume-dwt.c:2810-2881 */
static int perftest(int size, int opt_stride, enum volume_approach approach, int N, double *secs, int device)
{
	const double voxels = (double)size * size * size;
	const int clock_type = dwt_util_clock_autoselect();
	int return_code = 0;
	*secs = INFINITY;
	struct volume_t *data1 = device ? volume_alloc_device(sizeof(float), size, size, size, opt_stride)
	                                : volume_alloc_realiably_locked(sizeof(float), size, size, size, opt_stride);
	struct volume_t *data2 = device ? volume_alloc_device(sizeof(float), size, size, size, opt_stride)
	                                : volume_alloc_realiably_locked(sizeof(float), size, size, size, opt_stride);
	/* the pattern is the same every round: a device run keeps a host master with the device volume's
	 * layout and uploads it */
	const size_t total = data1->stride_z * (size_t)size;
	struct volume_t master_rec = *data1, *master = NULL;
	if (device) {
		master_rec.data = malloc(total);
		if (!master_rec.data)
			dwt_util_error("volume_perftest: out of host memory\n");
		master = &master_rec;
		volume_fill_s(master);
	}
	for (int n = 0; n < N; n++) {
		if (master) {
			if (dwt_hip_memcpy_h2d(data1->data, master->data, total))
				dwt_util_error("volume_perftest: %s\n", dwt_hip_last_error());
		} else {
			volume_fill_s(data1);
		}
		dwt_hip_sync();
		const dwt_clock_t start = dwt_util_get_clock(clock_type);
		cdf97_3f_op_wrapper_s(data1, data2, approach);
		dwt_hip_sync();
		const dwt_clock_t stop = dwt_util_get_clock(clock_type);
		const double per_voxel = (double)(stop - start) / (double)dwt_util_get_frequency(clock_type) / voxels;
		if (per_voxel < *secs)
			*secs = per_voxel;
		cdf97_3i_ip_sep_horizontal_s(data2);
		return_code += volume_compare_s(master ? master : data1, data2);
	}
	volume_free(data1);
	volume_free(data2);
	if (master)
		free(master->data);
	return return_code;
}

int volume_perftest_fwd97op_s(int size, int opt_stride, enum volume_approach approach, int N, double *secs, long unsigned *faults)
{
	if (faults)
		*faults = 0;
	return perftest(size, opt_stride, approach, N, secs, 0);
}

int volume_perftest_fwd97op_device_s(int size, int opt_stride, enum volume_approach approach, int N, double *secs)
{
	return perftest(size, opt_stride, approach, N, secs, 1);
}

/* src/volume-dwt.c:2883-2896 with g_growth_factor_s = 1.13f (src/libdwt.c:22385) */
static int size_grow(int size, int align)
{
	size = (int)(size * 1.13f);
	size += 1;
	size += align - 1;
	size &= ~(align - 1);
	return size;
}

/* src/volume-dwt.c:2898-2960 */
int volume_measure_fwd97op_s(int size_min, int size_max, int size_step, int N, int opt_stride, enum volume_approach approach)
{
	char path[4096];
	snprintf(path, sizeof path, "data/perftest/time-stride=%i-approach=%i.txt", opt_stride, (int)approach);
	FILE *file_time = fopen(path, "w");
	if (!file_time)
		dwt_util_error("unable to open file: %s\n", path);
	snprintf(path, sizeof path, "data/perftest/faults-stride=%i-approach=%i.txt", opt_stride, (int)approach);
	FILE *file_faults = fopen(path, "w");
	if (!file_faults)
		dwt_util_error("unable to open file: %s\n", path);
	fprintf(file_time, "# voxels secs/pel\n");
	fprintf(file_faults, "# voxels page_faults\n");
	int total_errors = 0;
	if (size_step < 1 || (size_step & (size_step - 1)))
		dwt_util_error("volume_measure_fwd97op_s: size_step must be a power of two\n");
	for (int size = size_min; size < size_max; size = size_grow(size, size_step)) {
		double secs;
		long unsigned faults;
		const int errors = volume_perftest_fwd97op_s(size, opt_stride, approach, N, &secs, &faults);
		const int voxels = size * size * size;
		dwt_util_log(LOG_INFO, "perftest: size=%4i opt_stride=%i approach=%2i (N=%2i): time=%f [nsecs/pel]; errors=%i; faults=%lu\n",
			size, opt_stride, (int)approach, N, secs * 1e9, errors, faults);
		fprintf(file_time, "%i\t%.20f\n", voxels, secs);
		fprintf(file_faults, "%i\t%lu\n", voxels, faults);
		total_errors += errors;
	}
	fclose(file_time);
	fclose(file_faults);
	return total_errors;
}
