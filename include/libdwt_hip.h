/*
 * libdwt_hip.h -- C-ABI of the MI355X (gfx950) backend behind libdwt's 2-D entry
 * points.  Plain C: pointers, ints and sizes only.
 *
 * The functions declared in include/libdwt.h (dwt_cdf97_2f_s & co.) are thin C
 * wrappers over dwt_hip_transform2d(); this header is what a foreign-function
 * binding (cgo, JNI, ctypes ...) or the reference's own sources would bind when they
 * want the device path directly, keep images resident in HBM, run batches, pick a
 * stream or read kernel timings.  See INTEGRATION.md.
 *
 * Pointer rule for every transform entry: `src`/`dst`/`ptr` may be ordinary host
 * memory (any byte strides; staged through HBM, result copied back) or device
 * memory (hipMalloc / dwt_hip_malloc / a torch tensor's data_ptr; transformed in HBM, nothing
 * crosses PCIe).  Device images with adjacent, aligned elements (stride_y == element size, stride_x and
 * the pointer multiples of it) are what the fused sweeps read and write directly.  Any other byte strides
 * -- one channel of an interleaved multi-channel matrix as src/cvdwt.cpp:98-135 passes it (ptr = data +
 * elemSize1*channel, stride_y = elemSize), odd pitches -- are packed into a dense image, transformed and
 * spread back ON THE DEVICE, the device-side dwt_util_memcpy_stride_s / _i (src/system.c:102-164): only
 * the image's own elements are written; rows must not overlap (stride_x >= (width-1)*stride_y + element
 * size).  The batch and 3-D entries take dense rows only.  Every wavelet runs on fused tile sweeps (float, int32 and, since round 2,
 * double); the exact line-pass kernels serve sparse frames, single-line directions and accel 1.
 *
 * Threading: one context PER HOST THREAD (device binding, stream, workspace, options), so
 * calls from different threads never share scratch memory -- unlike the reference, whose 2-D
 * drivers mutate process globals (src/libdwt.c:12839-12862).  One process drives several
 * GPUs with one thread per device: each thread calls dwt_hip_set_device(d) first.  Options
 * (dwt_hip_set_option, dwt_util_set_accel) and dwt_hip_set_stream are per thread as well.
 * Calls are asynchronous for device pointers (stream-ordered on the stream given to
 * dwt_hip_set_stream) and synchronous for host pointers.
 *
 * Error rule: every int function returns 0 on success and non-zero on failure with
 * a message retrievable by dwt_hip_last_error().  There is NO CPU fallback: without
 * a usable gfx950 device every transform entry fails.
 */
#ifndef LIBDWT_HIP_H
#define LIBDWT_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* wavelet/type selectors; the 2-D drivers they replace are cited in libdwt.h */
enum dwt_hip_wavelet {
	DWT_HIP_CDF97_S = 0, /* float CDF 9/7: dwt_cdf97_2f_s / dwt_cdf97_2i_s (src/libdwt.c:12776, 17040) */
	DWT_HIP_CDF53_I = 1, /* int32 CDF 5/3: dwt_cdf53_2f_i / dwt_cdf53_2i_i (src/libdwt.c:16304, 18142) */
	DWT_HIP_CDF53_S = 2, /* float CDF 5/3: dwt_cdf53_2f_s / dwt_cdf53_2i_s (src/libdwt.c:16470, 18296) */
	DWT_HIP_CDF97_D = 3, /* double CDF 9/7: dwt_cdf97_2f_d / dwt_cdf97_2i_d (src/libdwt.c:12451, 16884) */
	DWT_HIP_CDF53_D = 4, /* double CDF 5/3: dwt_cdf53_2f_d / dwt_cdf53_2i_d (src/libdwt.c:12535, 16962) */
	DWT_HIP_CDF97_I = 5  /* int32 fixed-point CDF 9/7: dwt_cdf97_2f_i / dwt_cdf97_2i_i (src/libdwt.c:16387, 18219) */
};

/* Lifecycle.  dwt_hip_init picks the device from DWT_HIP_DEVICE, else LOCAL_RANK,
 * else 0; it is idempotent.  Replaces the BCE firmware load of dwt_util_init
 * (src/libdwt.c:19158-19181). */
int dwt_hip_init(void);
void dwt_hip_finish(void);
/* Bind the CALLING THREAD's context to a device (0 .. dwt_hip_device_count()-1); a thread that
 * never calls it uses DWT_HIP_DEVICE / LOCAL_RANK / 0.  Rebinding frees the thread's workspace on
 * the old device.  dwt_hip_get_device: the bound device, -1 before the first use. */
int dwt_hip_set_device(int device);
int dwt_hip_get_device(void);
int dwt_hip_device_count(void);
const char *dwt_hip_device_name(void);
const char *dwt_hip_last_error(void);

/* Run on this hipStream_t (NULL = the default stream).  The context's scratch is shared by its streams: a change of
 * stream makes the new one wait (event) for everything the context queued on the old one, so alternating two streams
 * on one thread is safe -- the chains are serialised where they share scratch.  Independent concurrent chains belong
 * to different host threads (one context each).  Streams under capture are not ordered. */
void dwt_hip_set_stream(void *hip_stream);
/* The running-LL scratch of the 2-D Mallat drivers (two bands: the level-1 band, ceil(W/2) x ceil(H/2)
 * elements per image, and the level-2 band) in memory the CALLER owns and places -- the library then
 * neither grows nor frees it (a call that needs more fails).  Two NULLs hand the scratch back to the
 * library.  Per thread, like the rest of the context. */
int dwt_hip_set_workspace(void *band0, size_t bytes0, void *band1, size_t bytes1);
void dwt_hip_sync(void);

/* Placement.  The rate of a forward level depends on where in PHYSICAL memory its three streams lie
 * relative to each other (source rows, detail subbands, running LL band: DESIGN.md s5,
 * profiles/r04_placement.md).  The reference hands its callers a placement-aware allocator for the same
 * kind of reason -- dwt_util_get_opt_stride / dwt_util_get_stride, src/libdwt.c:20641-20707 -- and so does
 * this library:
 *   - the library's own LL scratch: dwt_hip_tune (below) on a forward call (batch or single image, distinct
 *     source and destination, two levels or more) that needs "place_min_mib" (option, default 1024) MiB or more
 *     of it tries up to "place_tries" (option, default 4; 1 = off) allocations, times the call itself on each
 *     and keeps the fastest.  dwt_hip_placement_report returns what the last search measured (ms per
 *     candidate, return value = number of candidates, 0 = no search ran).  A transform call itself never
 *     searches (unless DWT_HIP_TUNE=1): it allocates what it needs once and afterwards nothing.
 *   - dwt_hip_alloc_batch: source and destination of a resident batch of `n_images` dense size_x x size_y
 *     images (pitch size_x elements, images size_x * size_y elements apart) together with the scratch, placed
 *     by measurement: most of the card's free memory is mapped as one arena, the destination is tried at
 *     every 4 GiB step of it (one level against the source), the scratch at every step for the best
 *     destinations (the `levels`-level forward transform of the whole batch; < 0: full depth), the best
 *     arrangement is kept and the rest of the arena returned.  Seconds, once, for a batch that stays
 *     resident; dwt_hip_alloc_batch_report says what was measured.  Free both with dwt_hip_free. */
int dwt_hip_alloc_batch(int wavelet, int n_images, int size_x, int size_y, int levels, void **src, void **dst);
/* the same for the two dense volumes of an out-of-place 3-D call (dwt_hip_transform3d_op) of `levels` levels */
int dwt_hip_alloc_volumes(int size_x, int size_y, int size_z, int levels, void **src, void **dst);
int dwt_hip_placement_report(double *ms, int n);
/* "" when the last dwt_hip_alloc_batch / _volumes of this thread ran its search, else why it allocated plainly */
const char *dwt_hip_alloc_batch_note(void);
/* Buffers of dwt_hip_alloc_batch / _volumes are mapped through the virtual-memory API with access for their owner
 * alone (hipDeviceEnablePeerAccess does not cover such ranges).  dwt_hip_transform2d_batch_sharded grants its
 * slots' devices by itself; dwt_hip_grant_access does it for a caller's own peer copies (`dev_ptr` may point
 * anywhere into the buffer) and, for plain allocations, enables peer access from each device named.
 * 0 = every device named can reach the buffer. */
int dwt_hip_grant_access(void *dev_ptr, const int *devices, int n_devices);
/* MEASUREMENT IS EXPLICIT (round 5).  A transform call never measures anything: it allocates its scratch plainly,
 * launches every level once and uses the launcher's tile rule -- unless dwt_hip_tune has run for its shape on
 * the calling thread's context.  dwt_hip_tune runs the `levels`-level transform on THE CALLER'S OWN BUFFERS
 * (`dst` receives the transform of `src`, as after a call; src != dst, device pointers, batch_stride may be 0
 * for one image) a few times: the scratch placement search (forward, two levels or more, "place_min_mib" MiB of
 * scratch or more: up to "place_tries" allocations behind growing spacers, each timed with the call itself,
 * the fastest kept) and the tile-height tuner (every level whose input is 512 MiB or more -- a level that fits the
 * 256 MiB Infinity Cache cannot be measured by repeating it: 64 / 32 / 16 row pairs forward, 32 / 16 / 8 inverse).  Synchronous, one at a time per device; results are kept by the calling thread's
 * context per (wavelet, direction, width, height, batch) until dwt_hip_finish.  Same bits with and without.
 * The reference's analogue is explicit too: dwt_util_get_opt_stride, src/libdwt.c:20641-20707.
 * Programs that only know libdwt.h: DWT_HIP_TUNE=1 in the environment (option "tune_in_call") lets the first
 * large call of a shape measure by itself, as rounds 3-4 did. */
int dwt_hip_tune(int wavelet, int inverse, const void *src, void *dst, size_t batch_stride, int batch,
	int stride_x, int size_x, int size_y, int levels);
void dwt_hip_alloc_batch_report(int *chunks, int *dst_tried, int *ll_tried, int *dst_at, int *ll_at, double *ms4, double *seconds);

/* Tuning / variant selection (mirrors dwt_util_set_accel, src/libdwt.c:19946).  Every setting gives the
 * same bits; what is left after round 4's pruning is what the tests use as cross-checks or a caller may need.
 * 2-D: "generic" (1 = force the exact line-pass kernels), "cpt" (0 = auto / 4 / 8 columns per lane),
 * "tile_pairs" (0 = auto), "waves" (1..4 per workgroup), "xcd_swizzle" (0/1), "ring" (0 = auto / 8 / 16 LDS
 * rows per wave, forward) and "ring_inv" (8 / 16), "nt" (7 = default cache policy, 3 = the LL band's stores
 * non-temporal too, 15 = 7 with the neighbour taps by wavefront shifts instead of LDS reads), "nt_auto"
 * (1 = policy 3 by itself when a launch's LL bands exceed 1 GiB), "fma" (1 = contracted lifting steps:
 * NOT the reference's rounding, within 1e-5), "fused_d" (0 = double precision through the exact line passes),
 * "ride_copy" (1 = in-place Mallat calls on one image: the copy of level 0's staged subbands rides along with the deeper
 * levels' launches as extra workgroups; 0 = a launch of its own, the cross-check) and "ride_mib" (MiB of it per small level),
 * "host_pipeline" (1 = host-pointer calls on images of 64 MiB and more run band by band under their own
 * PCIe transfers, the caller's memory pinned in place for the call; 0 = upload, transform, download),
 * "il_inplace_shell" (1 = in-place calls of the interleaved entries run level 0 in place over a snapshot of the tile
 * halos; 0 = through a staging copy of the image, the cross-check),
 * "il_exact_borders" (0 = no border strips at all: the top 8 rows / last 5 columns of a level keep the sweep's
 * rows-then-columns rounding -- NOT the reference's bits there, a few ulp, far inside 1e-5; opt-in like "fma"),
 * "tune_tiles" (1 = levels of 512 MiB and more use the tile height dwt_hip_tune measured for their shape;
 * 0 = always the launcher's rule), "tune_in_call" (1 = the first large call of a shape measures by itself;
 * default: DWT_HIP_TUNE), "place_tries" / "place_min_mib" (placement search, below), "place_max_gib" (cap of the
 * arena dwt_hip_alloc_batch / _volumes map for their search; 0 = free memory - 8 GiB).
 * Read-only: "stat_launches" / "stat_allocs" (kernel launches / device allocations of this context's 2-D drivers
 * so far), "tile_cache_size", "place_last_tries", "place_last_best".
 * 3-D: "vol_fused" (1 = one-pass levels where they pay, 2 = wherever they can run, 0 = two passes),
 * "vol_whole" (0 = the general kernel variant as a cross-check), "vol_direct" (levels >= 1 into their lattice
 * of the destination: 2 = rows shared by levels 0 and 1 written once, 1 = sample-wise stores, 0 = dense
 * results + scatter passes), "vol_nt", "vol_rows" (8 / 6), "vol_tile_pairs", "vol_swizzle",
 * "vol_ip_waves" (0 = auto / 4 / 8 waves per workgroup of the one-pass levels: tiles of 32 or 64 rows),
 * "vol_inplace_fused" (in-place calls: 1 = one fused pass per level in place over a snapshot of the tile
 * halos, forward and inverse; 0 = two passes per level).
 * Environment (diagnostics only, read once): DWT_HIP_PLACE_VERBOSE (the placement search prints its timings),
 * DWT_HIP_TUNE_VERBOSE (the tile tuner prints every candidate's time),
 * DWT_HIP_PIPE_VERBOSE (a pipelined host-pointer call prints when its upload / download streams end),
 * DWT_HIP_PIPE_BAND (row pairs per band of such a call, a multiple of 64; default 256). */
int dwt_hip_set_option(const char *name, int value);
int dwt_hip_get_option(const char *name);

/* Multi-level 2-D transform, Mallat layout, all arguments as in libdwt's drivers
 * (src/libdwt.h:562-573, 867-878, 667-679, 962-974).  src == dst selects the
 * in-place entries, src != dst the `_s2` out-of-place entries.  `*j` is in/out for
 * forward (clamped as the reference does) and in for inverse. */
int dwt_hip_transform2d(int wavelet, int inverse, const void *src, void *dst,
	int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int *j, int decompose_one, int zero_padding);

/* Batch of independent equally sized dense images resident in HBM, `batch_stride`
 * bytes apart; one launch per level covers the whole batch. */
int dwt_hip_transform2d_batch(int wavelet, int inverse, const void *src, void *dst,
	size_t batch_stride, int batch, int stride_x, int size_x, int size_y, int *j);

/* The same batch sharded over several GPUs of ONE process (images are independent: no
 * collective in the transform).  `src` / `dst` lie in the memory of the calling thread's device, which must
 * be devices[0]; image b belongs to slot b * n_slots / batch (n_slots = min(n_devices, batch)), i.e. slot k
 * owns the images [ceil(k * batch / n_slots), ceil((k + 1) * batch / n_slots)) -- dwt_hip_shard_bounds.
 * Slot 0's shard is transformed where it lies; every other slot is a persistent host thread with its own
 * context on devices[k] (a device may be named more than once) that pulls its shard across
 * (hipMemcpyPeerAsync over xGMI) in up to four pieces, transforms each with dwt_hip_transform2d_batch and
 * pushes the result back while the next piece arrives -- all slots at the same time.  Synchronous: returns
 * when `dst` is complete.  Bytes of `dst` outside the frames keep their values.  Root-egress bound: the
 * whole batch leaves and re-enters one device. */
int dwt_hip_transform2d_batch_sharded(int wavelet, int inverse, const void *src, void *dst,
	size_t batch_stride, int batch, int stride_x, int size_x, int size_y, int *j, const int *devices, int n_devices);
/* slot outside [0, n_slots) or batch < 0: *count = 0 */
void dwt_hip_shard_bounds(int batch, int n_slots, int slot, int *first, int *count);

/* The batch split for shards that are RESIDENT where they are transformed ("the >= 7x scaling
 * claim is measured on per-GPU-resident data"): shard k -- counts[k] images at srcs[k] / dsts[k],
 * `batch_stride` bytes apart -- lies in the memory of devices[k] (allocated there by a thread bound to it with
 * dwt_hip_set_device; a device may be named more than once; counts[k] == 0 skips a shard).  All shards are
 * transformed at the same time, each by a persistent host thread with a context of its own on its device (the
 * calling thread takes the first shard on its own device); nothing crosses xGMI.  Synchronous; every shard's device is
 * drained (hipDeviceSynchronize) before its shard is read, so producers on any stream of that device come first.  `*j` as in
 * dwt_hip_transform2d_batch.  dwt_hip_tune_batch_multi runs dwt_hip_tune in every slot instead (once, before
 * the first transform of shards that stay resident): the slots' contexts keep what it measures. */
int dwt_hip_transform2d_batch_multi(int wavelet, int inverse, const void *const *srcs, void *const *dsts, const int *counts,
	const int *devices, int n_shards, size_t batch_stride, int stride_x, int size_x, int size_y, int *j);
int dwt_hip_tune_batch_multi(int wavelet, int inverse, const void *const *srcs, void *const *dsts, const int *counts,
	const int *devices, int n_shards, size_t batch_stride, int stride_x, int size_x, int size_y, int levels);

/* 2-D transforms in the INTERLEAVED (in-place lifting) layout: no de-interleave, level j
 * works on the stride-2^j lattice of the image (even lattice index = low-pass).
 * `wavelet` is DWT_HIP_CDF97_S or DWT_HIP_CDF53_S (and, flavour 0 only, DWT_HIP_CDF97_I for
 * dwt_cdf97_2f_inplace_i / dwt_cdf97_2i_inplace_i, src/libdwt.c:17424, 17308).  `flavour` 0 = libdwt.h's
 * dwt_cdf97_2f_inplace_s / dwt_cdf97_2i_inplace_s / dwt_cdf53_2f_inplace_s /
 * dwt_cdf53_2i_inplace_s (src/libdwt.c:12926, 17474, 16553, 17886); flavour 1 =
 * dwt-simple.h's forward fdwt2_cdf97_* / fdwt2_cdf53_* (src/dwt-simple.c:2224, 2356); flavours
 * 2 / 3 = fdwt2h1_cdf97_vertical_s / fdwt2v1_cdf97_vertical_s (rows only / columns only, :1747, :1837).
 * Host or device pointers, in place (src == dst) or out of place. */
int dwt_hip_transform2d_interleaved(int wavelet, int inverse, int flavour, const void *src, void *dst,
	int stride_x, int stride_y, int size_o_big_x, int size_o_big_y, int size_i_big_x, int size_i_big_y,
	int *j, int decompose_one);

/* Single-level 3-D CDF 9/7 float over the interleaved in-place layout of
 * cdf97_3f_ip_sep_horizontal_s / cdf97_3i_ip_sep_horizontal_s
 * (src/volume-dwt.c:677, 1115); `levels` > 1 re-applies it on the LLL lattice
 * (strides doubled) as SURVEY.md s8 a11 describes.  Device pointer, dense x.  Levels of 512^3 and
 * more (256 tiles of 256 x 64 voxel columns or more) run in ONE pass in place (tile halos read from a
 * snapshot, ~10.8 B per voxel), smaller ones in two passes through a scratch volume. */
int dwt_hip_transform3d(int inverse, void *vol, size_t stride_y, size_t stride_z,
	int size_x, int size_y, int size_z, int levels);

/* The same forward transform OUT OF PLACE (src != dst, both device pointers, same strides):
 * cdf97_3f_op_sep_horizontal_s (src/volume-dwt.c:727-785), the entry the reference's 3-D
 * perf test drives.  Each level is one fused x+y+z pass where that pays (volumes of about
 * 448^3 and more, at least 128 samples wide; any size and 4-byte alignment), two passes otherwise. */
int dwt_hip_transform3d_op(const void *src, void *dst, size_t stride_y, size_t stride_z,
	int size_x, int size_y, int size_z, int levels);

/* The same two transforms on the FIELDS of the reference's struct volume_t (include/volume.h; the
 * typed wrappers cdf97_3f_op_sep_horizontal_s & co. of include/volume-dwt.h sit on these): one level,
 * host or device pointers (host volumes are staged through HBM), source and destination with
 * their own row / slice strides in bytes, samples dense along x.  dirs: 7 = x, y and z; 1 = x lines
 * only (copy, then lift); 2 / 4 = y / z lines only, in place on dst (src unused) -- the reference's
 * VOL_SEP_HORIZONTAL_X / _Y / _Z measurements (src/volume-dwt.c:788, :852, :918). */
int dwt_hip_volume_fwd_op(const void *src, size_t src_stride_y, size_t src_stride_z, void *dst, size_t dst_stride_y,
	size_t dst_stride_z, int size_x, int size_y, int size_z, int dirs);
int dwt_hip_volume_ip(int inverse, void *data, size_t stride_y, size_t stride_z, int size_x, int size_y, int size_z);

/* Device-side twins of dwt_util_conv_show_{s,i} (src/libdwt.c:21075, 21020) and
 * dwt_util_compare_{s,i} (:1593, :1531) for images that stay in HBM between a forward and
 * an inverse transform.  compare returns 0 equal / 1 differ (float: 1e-3 absolute, NaN or
 * Inf => differ) / -1 error. */
int dwt_hip_conv_show(int is_int, const void *src, void *dst, int stride_x, int stride_y, int size_x, int size_y);
int dwt_hip_compare(int is_int, const void *ptr1, const void *ptr2, int stride_x, int stride_y, int size_x, int size_y);

/* Device memory helpers so that C callers need no HIP headers. */
void *dwt_hip_malloc(size_t bytes);
void dwt_hip_free(void *dev_ptr);
/* page-locked host memory (what volume_alloc_realiably_locked hands out): DMA without a bounce buffer */
void *dwt_hip_malloc_host(size_t bytes);
void dwt_hip_free_host(void *host_ptr);
int dwt_hip_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes);
int dwt_hip_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes);
int dwt_hip_is_device_pointer(const void *p);

/* dwt_util_perf_cdf97_2_s's protocol (src/libdwt.c:21444-21476) with the M images
 * resident in HBM: seconds per transform, minimum over N loops. */
void dwt_hip_perf_cdf97_2_s(int stride_x, int stride_y, int size_o_big_x, int size_o_big_y,
	int size_i_big_x, int size_i_big_y, int j_max, int decompose_one, int zero_padding,
	int M, int N, int clock_type, float *fwd_secs, float *inv_secs);

/* Kernel timing with HIP events on the stream the kernels run on.  While enabled,
 * every launch of the level-0 sweep kernel (the dominant kernel) is bracketed by
 * an event pair; dwt_hip_prof_read synchronises and returns the summed duration
 * and the number of launches since the last reset. */
void dwt_hip_prof_enable(int on); /* 1: level-0 kernel only; 2: every level's kernel */
int dwt_hip_prof_read(double *level0_ms_sum, int *launches);
/* mode 2: per-level sums (index = level whose input/output is the larger frame) */
int dwt_hip_prof_read_levels(double *ms_sum, int *launches, int n);

#ifdef __cplusplus
}
#endif
#endif
