/*
 * volume_perftest.c -- the reference's 3-D perf test protocol (volume_perftest_fwd97op_s,
 * src/volume-dwt.c:2810-2881: fill, forward out of place [timed], inverse in place, compare) driven
 * from C through include/volume-dwt.h, exactly as a program written against the reference's
 * volume-dwt.h would: first with host volumes (staged through HBM, the time includes PCIe), then
 * with both volumes resident in HBM, then the schedule dispatcher and the typed entries by hand.
 * Own code written against the headers under include/.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/volume_perftest.c -o volume_perftest \
 *       -Llibdwt_amd -l:libdwt_hip.so -Wl,-rpath,$PWD/libdwt_amd -lm
 */
#include "libdwt.h"
#include "volume-dwt.h"

#include <stdlib.h>

int main(int argc, char *argv[])
{
	const int size = argc > 1 ? atoi(argv[1]) : 256;
	dwt_util_init();
	int errors = 0;
	double secs;
	long unsigned faults;

	int e = volume_perftest_fwd97op_s(size, 1, VOL_SEP_HORIZONTAL, 3, &secs, &faults);
	dwt_util_log(LOG_INFO, "perftest: size=%4i host volumes:   time=%f [nsecs/pel]; errors=%i\n", size, secs * 1e9, e);
	errors += e;
	e = volume_perftest_fwd97op_device_s(size, 1, VOL_SEP_HORIZONTAL, 5, &secs);
	dwt_util_log(LOG_INFO, "perftest: size=%4i device volumes: time=%f [nsecs/pel] = %.1f Gvoxels/s; errors=%i\n", size, secs * 1e9,
		1e-9 / secs, e);
	errors += e;
	e = volume_perftest_fwd97op_s(64, 0, VOL_HORIZ_VERT4X4X4, 1, &secs, &faults); /* another schedule, same transform */
	errors += e;

	/* by hand: out of place into a volume with other strides, in place, back */
	struct volume_t *a = volume_alloc_realiably(sizeof(float), 70, 33, 21, 1);
	struct volume_t *b = volume_alloc_realiably_locked(sizeof(float), 70, 33, 21, 0);
	struct volume_t *c = volume_alloc_device(sizeof(float), 70, 33, 21, 2);
	volume_fill_s(a);
	cdf97_3f_op_sep_horizontal_s(a, b);   /* host -> host */
	cdf97_3f_op_wrapper_s(a, c, VOL_SEP_VERTICAL); /* host -> device */
	errors += volume_compare_s(b, c);
	cdf97_3i_ip_sep_horizontal_s(c);
	errors += volume_compare_s(a, c);
	volume_copy_s(c, a);
	cdf97_3f_ip_sep_horizontal_s(c);
	errors += volume_compare_s(b, c);
	volume_free(a);
	volume_free(b);
	volume_free(c);

	dwt_util_log(LOG_INFO, errors ? "volume perftest: %d errors\n" : "volume perftest: success (%d errors)\n", errors);
	dwt_util_finish();
	return errors != 0;
}
