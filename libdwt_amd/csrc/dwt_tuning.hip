// dwt_tuning.hip -- measurement that belongs to no transform call: tile heights of large levels and the
// placement of the library's own LL scratch, run by dwt_hip_tune (explicit) and cached per context.
#include "dwt_backend.h"

namespace dwtb {

// ---- tile height of a large forward level: measured, once per shape ------------------------------------
// The launcher's rule (64 row pairs per tile unless that leaves too few tiles) is within 1-2 % of the best
// height for level 0 of most calls, but the best height of a level depends on more than its tile count --
// level 1 of 32 images wants 32 pairs (735 against 765 us), level 0 of 8 images wants 64 (761 against 778),
// both have 8192 tiles of 64 pairs; level 3 of 64 images wants 16 (113 against 141 us).  So a level whose
// input is 512 MiB or more is timed ONCE per (wavelet, width, height, batch) with 64, 32 and 16 pairs -- the level
// is idempotent while its input stands, which it does until the next level runs -- and the fastest height is
// remembered by the calling thread's context.  Same bits with every height (tests: tile variants).  Measured
// only inside dwt_hip_tune (or with DWT_HIP_TUNE=1): an ordinary transform call looks the height up and
// falls back to the launcher's rule.  Option "tune_tiles" = 0 turns both off; a forced "tile_pairs" wins.
bool may_measure()
{
	if (g.tune_in_call < 0) {
		const char *e = getenv("DWT_HIP_TUNE");
		g.tune_in_call = (e && atoi(e) > 0) ? 1 : 0;
	}
	return g.tuning || g.tune_in_call > 0;
}

// A measured choice: tile height, and (forward) columns per lane / ring depth where the candidate names them, packed
// into one int: bits 0-15 row pairs, 16-23 columns per lane (0: the launcher's), 24-31 ring rows (0: the launcher's).
struct TileCand {
	int pairs, cpt, ring;
};
static int pack_choice(const TileCand &c) { return c.pairs | (c.cpt << 16) | (c.ring << 24); }

void apply_tile_choice(int choice, SweepTuning *t, bool inverse)
{
	if (choice <= 0)
		return;
	t->tile_pairs = choice & 0xffff;
	if ((choice >> 16) & 0xff)
		t->cpt = (choice >> 16) & 0xff;
	if ((choice >> 24) & 0xff)
		(inverse ? t->ring_inv : t->ring) = (choice >> 24) & 0xff;
}

// One measurement at a time per device: several contexts on one GPU (the slots of dwt_multi.hip, a caller's own threads)
// would time each other's noise and stack their placement spacers up to an out-of-memory.  Recursive: dwt_hip_tune holds
// it around the placement search and the tuned transform, whose levels take it again -- and so does a transform call
// that measures by itself (DWT_HIP_TUNE=1 / option tune_in_call, which the slots of a sharded call inherit).
static std::recursive_mutex &measure_mutex()
{
	static std::recursive_mutex per_device[64];
	return per_device[g.device & 63];
}

static int tune_tile_pairs(unsigned long long key, bool inverse, std::initializer_list<TileCand> cands, const std::function<hipError_t(const SweepTuning &)> &launch)
{
	auto it = g.tile_cache.find(key);
	if (it != g.tile_cache.end())
		return it->second;
	if (!may_measure() || g.placing || stream_is_capturing())
		return 0; // the launcher's rule; decided by dwt_hip_tune
	std::lock_guard<std::recursive_mutex> turn(measure_mutex());
	hipEvent_t e0, e1;
	if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
		return 0;
	int best = 0;
	float best_ms = 0;
	static const bool verbose = getenv("DWT_HIP_TUNE_VERBOSE") != nullptr;
	for (const TileCand &c : cands) {
		if ((c.cpt && g.tune.cpt) || (c.ring && (inverse ? g.tune.ring_inv != 8 : g.tune.ring != 0)))
			continue; // a width / ring depth set by hand stands
		SweepTuning t = g.tune;
		apply_tile_choice(pack_choice(c), &t, inverse);
		// three launches, the faster of the last two (the first one warms the caches it can)
		float ms = 0;
		bool ok = true;
		for (int r = 0; r < 3 && ok; r++) {
			float one = 0;
			hipEventRecord(e0, g.stream);
			ok = launch(t) == hipSuccess;
			hipEventRecord(e1, g.stream);
			g.stat_launches++;
			ok = ok && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&one, e0, e1) == hipSuccess;
			if (r == 1 || (r == 2 && one < ms))
				ms = one;
		}
		if (verbose)
			fprintf(stderr, "tune %s W %d: %d pairs, cpt %d, ring %d: %.1f us%s\n", inverse ? "inv" : "fwd", (int)((key >> 38) & 0xfffff), c.pairs, c.cpt, c.ring,
				ms * 1e3, ok ? "" : " (failed)");
		if (ok && (!best || ms < best_ms)) {
			best = pack_choice(c);
			best_ms = ms;
		}
	}
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	(void)hipGetLastError();
	g.tile_cache[key] = best;
	return best;
}

static unsigned long long tile_key(Wavelet w, bool inverse, int W, int H, int batch)
{
	return ((unsigned long long)w << 59) ^ ((unsigned long long)inverse << 58) ^ ((unsigned long long)W << 38) ^ ((unsigned long long)H << 18) ^ (unsigned long long)batch;
}

// Which levels are measured: those whose input does not fit the 256 MiB Infinity Cache.  The tuner launches a level
// several times on the SAME input; a level of 64 ... 256 MiB (one 8192^2 image, its level 1) then runs out of that cache
// and the ranking it gives does not hold for the call on fresh data -- round 5: one image tuned this way ran its inverse
// at 172.8-180.8 us against 170.1-174.4 with the launcher's rule (scripts/r05/tuned_single.py).
static bool tunable(int W, int H, int batch, int interleaved)
{
	return g.tune_tiles && !interleaved && (size_t)W * H * batch * sizeof(float) >= ((size_t)512 << 20) && W >= 1024 && H >= 256;
}

// the packed choice (apply_tile_choice) for a large forward level; 0: the launcher's own rule
int tuned_tile_pairs(Wavelet w, const FwdLevelArgs &a)
{
	if (!tunable(a.W, a.H, a.batch, a.interleaved))
		return 0;
	const Wavelet wk = (g.fma && w == kCdf97S) ? kCdf97SFma : w;
	// (round 5 also tried 128 pairs and the 256-column tile with the deep ring: never the fastest on a batch, and on one
	// image -- 100.5 against 105.1 us in a variant scan -- inside the noise of where the image lies; scripts/r05/tuned_single.py)
	return tune_tile_pairs(tile_key(w, false, a.W, a.H, a.batch), false, {{64, 0, 0}, {32, 0, 0}, {16, 0, 0}},
		[&](const SweepTuning &t) { return launch_fwd_level(wk, a, t, g.stream); });
}

// the inverse levels alike (32 images of 8192^2: 16 pairs 520 against 507-510 Gsamples/s with the rule's 32)
int tuned_tile_pairs(Wavelet w, const InvLevelArgs &a)
{
	if (!tunable(a.W, a.H, a.batch, a.interleaved))
		return 0;
	const Wavelet wk = (g.fma && w == kCdf97S) ? kCdf97SFma : w;
	// (round 6: the 512-column tile joins the candidates; the launcher's rule is 16 pairs)
	return tune_tile_pairs(tile_key(w, true, a.W, a.H, a.batch), true, {{16, 0, 0}, {32, 0, 0}, {8, 0, 0}, {16, 8, 0}, {32, 8, 0}},
		[&](const SweepTuning &t) { return launch_inv_level(wk, a, t, g.stream); });
}
// ---- placement of the LL scratch ------------------------------------------------------------------
// The rate of a forward level depends on where in PHYSICAL memory its three streams lie relative to each
// other -- source rows, detail subbands, running LL band (profiles/r04_placement.md: coarse regions of
// three classes; +13 % when the two write streams are in different ones) -- and nothing finer than that
// matters.  The caller owns source and destination; the LL scratch is the library's.  So the first
// forward call that needs a large scratch tries a few allocations of it, each behind a spacer that pushes
// it into other physical memory, times the call itself on each (it writes exactly what the call will
// write: idempotent for distinct source and destination), and keeps the fastest.  Once per
// size: later calls find the scratch in place, allocate nothing and never synchronise.
int timed_forward(Wavelet w, Img s, Img d, const Geom &ge, int levels, int batch, long sb, long db, double *ms)
{
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	int rc = 0;
	g.placing = true;
	for (int r = 0; r < 2 && !rc; r++) {
		int j = levels;
		hipEventRecord(e0, g.stream);
		rc = forward2d(w, s, d, ge, &j, 0, 0, batch, sb, db);
		hipEventRecord(e1, g.stream);
	}
	g.placing = false;
	float t = 0;
	if (!rc && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess))
		rc = fail("timing a placement trial failed: %s", hipGetErrorString(hipGetLastError()));
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	*ms = t;
	return rc;
}

// spacer in front of candidate k of a placement search
static size_t place_jump(int k)
{
	return k <= 0 ? 0 : ((size_t)14 << 30) << (k > 3 ? 2 : k - 1);
}

bool stream_is_capturing()
{
	hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing(g.stream, &st) != hipSuccess) {
		(void)hipGetLastError();
		return true; // unknown: do nothing that synchronises
	}
	return st != hipStreamCaptureStatusNone;
}

int place_ll_scratch(Wavelet w, Img s, Img d, const Geom &ge, int levels, int batch, long sb, long db)
{
	const int es = elem_size(w);
	const size_t need[2] = {ll_band_bytes(ge, 0, batch, es), ll_band_bytes(ge, 1, batch, es)};
	g.place_n = 0;
	g.place_best = -1;
	if (!may_measure() || g.placing || g.ll_external || g.place_tries < 2 || s.p == d.p || (g.ll_bytes[0] >= need[0] && g.ll_bytes[1] >= need[1]) ||
		need[0] + need[1] < ((size_t)g.place_min_mib << 20) || !ge.dense() || ge.Wo(2) < 2 || ge.Ho(2) < 2 || g.force_generic ||
		stream_is_capturing())
		return 0;
	std::lock_guard<std::recursive_mutex> turn(measure_mutex());
	struct Cand {
		void *ll[2], *spacer;
		double ms;
		bool ok;
	};
	std::vector<Cand> cands;
	int rc = 0;
	bool own_released = false;
	for (int k = 0; k < g.place_tries && k < 8 && !rc; k++) {
		Cand c{{nullptr, nullptr}, nullptr, 0, false};
		// the spacers stay allocated during the search, so the jumps add up: candidates 14, 44, 104 ... GiB
		// further on (the classes come in 16 GiB granules, runs of one class can be 64 GiB long)
		const size_t jump = place_jump(k);
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need[0] + need[1] + (k ? jump : 0) + ((size_t)2 << 30))
			break;
		if (k && hipMalloc(&c.spacer, jump) != hipSuccess) {
			(void)hipGetLastError();
			break;
		}
		if (hipMalloc(&c.ll[0], need[0]) != hipSuccess || hipMalloc(&c.ll[1], need[1]) != hipSuccess) {
			(void)hipGetLastError();
			for (void *p : {c.ll[0], c.ll[1], c.spacer})
				if (p)
					hipFree(p);
			break;
		}
		g.stat_allocs += 2 + (k ? 1 : 0);
		// the context works on this candidate for the trial; the too-small scratch of earlier calls goes first
		if (!own_released) {
			if (g.ll[0] || g.ll[1])
				(void)hipStreamSynchronize(g.stream);
			for (int b = 0; b < 2; b++)
				if (g.ll[b])
					dev_free(g.ll[b]);
			own_released = true;
		}
		for (int b = 0; b < 2; b++) {
			g.ll[b] = c.ll[b];
			g.ll_bytes[b] = need[b];
		}
		rc = timed_forward(w, s, d, ge, levels, batch, sb, db, &c.ms);
		c.ok = rc == 0; // (a trial that failed has no time: never the best)
		cands.push_back(c);
	}
	if (cands.empty())
		return 0; // nothing allocated here and the context's scratch is as it was: the call allocates plainly
	int best = 0;
	for (size_t k = 0; k < cands.size(); k++) {
		if (cands[k].ok && (!cands[best].ok || cands[k].ms < cands[best].ms))
			best = (int)k;
		g.place_ms[k] = cands[k].ok ? cands[k].ms : -1;
	}
	g.place_n = (int)cands.size();
	g.place_best = best;
	(void)hipStreamSynchronize(g.stream); // (no early return from here on: every candidate is released or kept)
	for (size_t k = 0; k < cands.size(); k++) {
		if (cands[k].spacer)
			hipFree(cands[k].spacer);
		if ((int)k != best) {
			hipFree(cands[k].ll[0]);
			hipFree(cands[k].ll[1]);
		}
	}
	for (int b = 0; b < 2; b++) {
		g.ll[b] = cands[best].ll[b];
		g.ll_bytes[b] = need[b];
	}
	return rc;
}

// the body of dwt_hip_tune: placement search of the scratch (forward, large batches), then ONE transform with the
// tile tuner switched on -- every large level measures its tile heights on the way and the context keeps them
int tune2d(Wavelet w, bool inverse, Img s, Img d, const Geom &ge, int levels, int batch, long sb, long db)
{
	std::lock_guard<std::recursive_mutex> turn(measure_mutex());
	struct Guard {
		Guard() { g.tuning = true; }
		~Guard() { g.tuning = false; }
	} guard;
	if (stream_is_capturing())
		return fail("dwt_hip_tune measures and synchronises: not under a stream capture");
	if (!inverse && (levels < 0 || levels >= 2) && place_ll_scratch(w, s, d, ge, levels, batch, sb, db))
		return 1;
	int j = levels;
	const int rc = inverse ? inverse2d(w, s, d, ge, levels, 0, 0, batch, sb, db) : forward2d(w, s, d, ge, &j, 0, 0, batch, sb, db);
	if (!rc)
		HIP_TRY(hipStreamSynchronize(g.stream));
	return rc;
}

} // namespace dwtb

using namespace dwtb;

#pragma GCC visibility push(default)
extern "C" {

int dwt_hip_tune(int wavelet, int inverse, const void *src, void *dst, size_t batch_stride, int batch, int stride_x, int size_x, int size_y, int levels)
{
	if (check_inited())
		return 1;
	if (wavelet < 0 || wavelet > 5)
		return fail("unknown wavelet %d", wavelet);
	const int es = elem_size((Wavelet)wavelet);
	g_elems_are_32bit = es == 4;
	if (!src || !dst || batch < 1 || batch > 65535)
		return fail("dwt_hip_tune: bad argument (batch must be 1..65535)");
	if (!dwt_hip_is_device_pointer(src) || !dwt_hip_is_device_pointer(dst))
		return fail("dwt_hip_tune takes the device buffers the transforms will run on");
	if (batch == 1 && batch_stride == 0)
		batch_stride = (size_t)stride_x * size_y;
	if ((stride_x % es) || stride_x < size_x * es || (batch_stride % es) || batch_stride < (size_t)stride_x * size_y)
		return fail("bad strides");
	if (src == dst)
		return 0; // (the in-place entries stage level 0: nothing of theirs is measured)
	const Geom ge{size_x, size_y, size_x, size_y};
	return tune2d((Wavelet)wavelet, inverse != 0, Img{(char *)src, stride_x, es}, Img{(char *)dst, stride_x, es}, ge, levels, batch, (long)batch_stride,
		(long)batch_stride);
}

} // extern "C"
#pragma GCC visibility pop
