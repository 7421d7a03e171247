// dwt_backend.h -- what the backend's translation units share: the device context, workspace and
// staging helpers, the level-pass helpers.  Internal to the shared library (hidden visibility).
#pragma once
#include "../../include/libdwt_hip.h"
#include "dwt_kernels.h"

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <utility>
#include <vector>

namespace dwtb {
using namespace dwt;

struct Ctx {
	bool inited = false;
	int device = 0;
	int want_device = -1; // dwt_hip_set_device: the device this thread's context binds to (-1: environment / 0)
	hipStream_t stream = nullptr;
	char devname[256] = {0};
	// workspace
	void *stage_img = nullptr; // frame-sized staging image (in-place detour, generic passes)
	size_t stage_bytes = 0;
	void *ll[2] = {nullptr, nullptr}; // LL ping-pong
	size_t ll_bytes[2] = {0, 0};
	bool ll_external = false; // the caller owns the LL scratch (dwt_hip_set_workspace): never grown, never freed
	void *host_a = nullptr, *host_b = nullptr; // device images for host-pointer calls
	size_t host_a_bytes = 0, host_b_bytes = 0;
	void *vol_out = nullptr; // dense result volume of an in-place 3-D forward call (fused levels, then copied back)
	size_t vol_out_bytes = 0;
	void *vol_host[2] = {nullptr, nullptr}; // device staging of host volumes (struct volume_t entries)
	size_t vol_host_bytes[2] = {0, 0};
	hipEvent_t dl_ev[8] = {}; // strip events of host_download, created once
	hipEvent_t switch_ev = nullptr; // dwt_hip_set_stream: orders a newly set stream behind the old one's work
	// host-pointer calls on large images: level 0 band by band while the image is still crossing PCIe (host_forward_pipelined)
	hipStream_t up = nullptr, down = nullptr;
	hipEvent_t pipe_ev[3][16] = {};
	int host_pipeline = 1; // 0: upload, transform, download one after the other
	void *pin = nullptr; // pinned host staging for host-pointer calls with awkward strides
	size_t pin_bytes = 0;
	// options
	SweepTuning tune;
	VolTuning vol;
	int force_generic = 0;
	int fma = 0; // opt-in: contract the float 9/7 lifting steps (not bit-identical to libdwt)
	int il_temporal = 0; // set per interleaved call: the forward sweep of level 0 stores its even rows temporal (in place: the copy back reads them)
	int il_inplace_shell = 1; // interleaved in-place calls: level 0 over a snapshot of the tile halos (0: through a staging image, the cross-check)
	int il_exact_borders = 1; // interleaved 9/7: 0 = skip the exact border strips (opt-in: not bit-identical in the top 8 rows / last 5 columns of a level)
	// placement of the LL scratch (DESIGN s5): on the first forward call that needs `place_min_mib` or more of
	// scratch, up to `place_tries` allocations of it -- each behind a spacer that moves it into other
	// physical memory -- are timed with the call's own first two levels and the fastest kept
	// Measurement never happens inside an ordinary transform call (round 5): dwt_hip_tune runs the tile-height
	// tuner and the scratch placement search on the caller's buffers, once, and the context remembers the
	// results; a transform call looks them up, allocates plainly and launches each level once.  DWT_HIP_TUNE=1 in
	// the environment (option "tune_in_call") restores the implicit behaviour for programs that only know libdwt.h.
	int tune_tiles = 1; // use / measure tile heights of large levels (tuned_tile_pairs); 0: the launcher's rule
	std::map<unsigned long long, int> tile_cache;
	int place_tries = 4;  // candidates of the scratch placement search (< 2: no search)
	int place_min_mib = 1024;
	int place_max_gib = 0; // cap of the placement arena of dwt_hip_alloc_batch / _volumes (0: free memory - 8 GiB)
	int tune_in_call = -1; // -1: read DWT_HIP_TUNE on first use
	bool tuning = false;        // inside dwt_hip_tune: measurements allowed
	bool placing = false;       // inside a timed trial: no nested search
	long stat_launches = 0, stat_allocs = 0; // kernel launches / device allocations made by this context's 2-D drivers (tests)
	double place_ms[8] = {0};   // what the last search measured, per candidate
	int place_n = 0, place_best = -1;
	int fused_d = 1; // double-precision wavelets through the fused sweeps (0: exact line passes only)
	int ride_copy = 1; // in-place calls on one image: the staged subbands' copy rides along with the deeper levels' launches (0: a launch of its own)
	int ride_mib = 32; // ... MiB of it per small level (8192^2: 8 / 16 / 24 / 32 / 48 MiB: forward 202 / 202 / 199 / 199 / 199 us, inverse 224 / 227 / 225 / 224 / 229)
	// profiling
	int prof_on = 0;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
	std::vector<int> prof_tag; // level index of each recorded pair
	size_t prof_used = 0;
	double prof_ms = 0;
	int prof_launches = 0;
	double prof_level_ms[16] = {0};
	int prof_level_n[16] = {0};
};

// One context PER HOST THREAD: its own device binding, stream, workspace and options.  Calls from
// different threads therefore never share scratch buffers (the library is reentrant across
// threads), and one process drives several GPUs with one thread per device: each thread calls
// dwt_hip_set_device(d) first ("single process, 8 devices, one host thread per
// device").  Options set through dwt_hip_set_option / dwt_util_set_accel are per thread too.
extern thread_local Ctx g;
extern thread_local char g_err[512];
extern thread_local bool g_elems_are_32bit; // set per call: the fused sweeps exist for 4-byte elements only

int fail(const char *fmt, ...);

#define HIP_TRY(expr)                                                                          \
	do {                                                                                       \
		hipError_t e_ = (expr);                                                                \
		if (e_ != hipSuccess)                                                                  \
			return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

inline int ceil_div_pow2(int i, int j) { return (i + (1 << j) - 1) >> j; } // src/inline.h:455-461
inline int ceil_log2(int x)                                               // src/inline.h:443-448
{
	int n = 0;
	while (n < 31 && (1 << n) < x)
		n++;
	return n;
}
inline long align_up(long v, long a) { return (v + a - 1) / a * a; }

// A device image: element (y,x) at p + y*sx + x*es (dense elements of es = 4 or 8 bytes).
struct Img {
	char *p;
	long sx;    // row pitch in bytes
	int es = 4; // element size in bytes
};

struct Geom {
	int sox, soy, six, siy;
	int Wo(int j) const { return ceil_div_pow2(sox, j); }
	int Ho(int j) const { return ceil_div_pow2(soy, j); }
	int Wi(int j) const { return ceil_div_pow2(six, j); }
	int Hi(int j) const { return ceil_div_pow2(siy, j); }
	bool dense() const { return sox == six && soy == siy; }
};

int grow(void **p, size_t *have, size_t need);
void dev_free(void *p); // hipFree, or the release of a buffer mapped by dwt_placement.hip
int grant_range(const void *p, int owner, const int *devices, int n_devices); // a placed (VMM) buffer made reachable for these devices; plain allocations: no-op
int copy_rect_on(hipStream_t st, Img dst, long dx, long dy, Img src, long sx_, long sy_, long w, long h);
int copy_rect(Img dst, long dx, long dy, Img src, long sx_, long sy_, long w, long h);
int zero_rect(Img img, long x, long y, long w, long h);
// host images <-> dense device images (any byte strides; awkward pitches go through a pinned buffer)
int host_upload(const void *hp, int stride_x, int stride_y, int es, int w, int h, void *dp, long pitch);
int host_download(void *hp, int stride_x, int stride_y, int es, int w, int h, const void *dp, long pitch);
int host_volume_xfer(bool to_device, void *dev, size_t d_sy, size_t d_sz, void *host, size_t h_sy, size_t h_sz, int nx, int ny, int nz);
// one exact out-of-place 1-D pass over the lines of a frame (in == out is staged)
int generic_pass(Wavelet w, bool inverse, bool rows, Img in, Img out, int frame_w, int frame_h, int n_lines, int N, int hoff);
// placement (dwt_backend.hip / dwt_placement.hip)
size_t ll_band_bytes(const Geom &ge, int k, int batch, int es); // bytes of LL scratch band k (0: level-1 band, 1: level-2 band)
int timed_forward(Wavelet w, Img s, Img d, const Geom &ge, int levels, int batch, long sb, long db, double *ms); // ms of the 2nd of two calls
int place_ll_scratch(Wavelet w, Img s, Img d, const Geom &ge, int levels, int batch, long sb, long db);
int tune2d(Wavelet w, bool inverse, Img s, Img d, const Geom &ge, int levels, int batch, long sb, long db);
bool stream_is_capturing();
bool may_measure(); // inside dwt_hip_tune, or DWT_HIP_TUNE=1 / option "tune_in_call"
int tuned_tile_pairs(Wavelet w, const FwdLevelArgs &a); // the measured choice for this level (packed; 0: none) ...
int tuned_tile_pairs(Wavelet w, const InvLevelArgs &a);
void apply_tile_choice(int choice, SweepTuning *t, bool inverse); // ... applied to the launch's tuning
int forward2d(Wavelet w, Img src, Img dst, const Geom &ge, int *jp, int decompose_one, int zero_padding, int batch, long src_bstride, long dst_bstride);
int inverse2d(Wavelet w, Img src, Img dst, const Geom &ge, int j_max, int decompose_one, int zero_padding, int batch, long src_bstride, long dst_bstride);
bool level_fused_ok(const Geom &ge, int j); // level j runs on the fused sweeps (dense frame, both sides >= 2)
// host-pointer calls on large images, band by band under their own PCIe transfers (dwt_host_xfer.hip):
// 0 done, 1 error, -1 not applicable (the caller takes the plain path)
int host_forward_pipelined(Wavelet w, const void *src, void *dst, int stride_x, int W, int H, int *jp, int decompose_one);
int host_inverse_pipelined(Wavelet w, const void *src, void *dst, int stride_x, int W, int H, int j_max, int decompose_one);
int prof_drain();
void prof_before(int level = 0);
void prof_after(int level = 0);
int check_inited();

} // namespace dwtb
