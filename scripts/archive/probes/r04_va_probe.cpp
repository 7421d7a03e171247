// Do virtual addresses take part in the rate of a sweep?  The SAME physical chunks (hipMemCreate, 1 GiB each), in the same
// order, mapped at different virtual addresses: 1 GiB aligned, the same range shifted by 2 MiB / 512 MiB, another 1 GiB
// aligned range; each time the batched forward transform of 16 images 8192^2 (source, destination and the two LL bands
// inside the mapped range) is timed.  Build: hipcc -O2 -I include scripts/archive/probes/r04_va_probe.cpp -L libdwt_amd -ldwt_hip
//   -Wl,-rpath,$PWD/libdwt_amd -o gpurun_out/va_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "libdwt_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main()
{
	const size_t GiB = (size_t)1 << 30, MiB = (size_t)1 << 20;
	const int n = 8192, nb = 16, J = 5, n_chunks = 12;
	if (dwt_hip_init()) return 1;
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = 0;
	std::vector<hipMemGenericAllocationHandle_t> h(n_chunks);
	for (auto &x : h) CK(hipMemCreate(&x, GiB, &prop, 0));
	hipMemAccessDesc acc = {};
	acc.location = prop.location;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto run = [&](char *base, const char *what) {
		for (int i = 0; i < n_chunks; i++) CK(hipMemMap(base + i * GiB, GiB, 0, h[i], 0));
		CK(hipMemSetAccess(base, n_chunks * GiB, &acc, 1));
		char *src = base, *dst = base + 4 * GiB, *w0 = base + 8 * GiB, *w1 = base + 10 * GiB;
		const size_t b0 = (size_t)nb * (n / 2) * (n / 2) * 4 + 4096, b1 = (size_t)nb * (n / 4) * (n / 4) * 4 + 4096;
		if (dwt_hip_set_workspace(w0, b0, w1, b1)) { fprintf(stderr, "workspace: %s\n", dwt_hip_last_error()); exit(1); }
		int j = J;
		float best = 1e9, worst = 0;
		for (int rep = 0; rep < 14; rep++) {
			CK(hipEventRecord(e0, 0));
			if (dwt_hip_transform2d_batch(0, 0, src, dst, (size_t)n * n * 4, nb, n * 4, n, n, &j)) { fprintf(stderr, "transform: %s\n", dwt_hip_last_error()); exit(1); }
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (rep >= 4) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
		}
		printf("%-44s base %p: %.4f .. %.4f ms per 16-image call\n", what, (void *)base, best, worst);
		fflush(stdout);
		dwt_hip_set_workspace(nullptr, 0, nullptr, 0);
		CK(hipDeviceSynchronize());
		CK(hipMemUnmap(base, n_chunks * GiB));
	};
	dwt_hip_set_option("place_tries", 1);
	dwt_hip_set_option("tune_tiles", 0);
	void *r1 = nullptr, *r2 = nullptr;
	CK(hipMemAddressReserve(&r1, (n_chunks + 2) * GiB, GiB, nullptr, 0));
	CK(hipMemAddressReserve(&r2, (n_chunks + 2) * GiB, GiB, nullptr, 0));
	// the three buffers in ranges of their own: only the DISTANCES between their virtual addresses change
	{
		void *big = nullptr;
		CK(hipMemAddressReserve(&big, 200 * GiB, 2 * MiB, nullptr, 0));
		auto run3 = [&](size_t d_dst, size_t d_ws, const char *what) {
			char *src = (char *)big, *dst = (char *)big + d_dst, *ws = (char *)big + d_ws;
			for (int i = 0; i < 4; i++) CK(hipMemMap(src + i * GiB, GiB, 0, h[i], 0));
			for (int i = 0; i < 4; i++) CK(hipMemMap(dst + i * GiB, GiB, 0, h[4 + i], 0));
			for (int i = 0; i < 4; i++) CK(hipMemMap(ws + i * GiB, GiB, 0, h[8 + i], 0));
			CK(hipMemSetAccess(src, 4 * GiB, &acc, 1)); CK(hipMemSetAccess(dst, 4 * GiB, &acc, 1)); CK(hipMemSetAccess(ws, 4 * GiB, &acc, 1));
			const size_t b0 = (size_t)nb * (n / 2) * (n / 2) * 4 + 4096, b1 = (size_t)nb * (n / 4) * (n / 4) * 4 + 4096;
			if (dwt_hip_set_workspace(ws, b0, ws + 2 * GiB, b1)) { fprintf(stderr, "workspace: %s\n", dwt_hip_last_error()); exit(1); }
			int j = J;
			float best = 1e9, worst = 0;
			for (int rep = 0; rep < 14; rep++) {
				CK(hipEventRecord(e0, 0));
				if (dwt_hip_transform2d_batch(0, 0, src, dst, (size_t)n * n * 4, nb, n * 4, n, n, &j)) { fprintf(stderr, "transform: %s\n", dwt_hip_last_error()); exit(1); }
				CK(hipEventRecord(e1, 0));
				CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				if (rep >= 4) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
			}
			printf("%-44s dst at +%6.1f GiB, scratch at +%6.1f GiB: %.4f .. %.4f ms\n", what, d_dst / (double)GiB, d_ws / (double)GiB, best, worst);
			fflush(stdout);
			dwt_hip_set_workspace(nullptr, 0, nullptr, 0);
			CK(hipDeviceSynchronize());
			CK(hipMemUnmap(src, 4 * GiB)); CK(hipMemUnmap(dst, 4 * GiB)); CK(hipMemUnmap(ws, 4 * GiB));
		};
		for (int round = 0; round < 2; round++) {
			run3(4 * GiB, 8 * GiB, "back to back");
			run3(4 * GiB + 2 * MiB, 8 * GiB + 6 * MiB, "2 / 6 MiB further");
			run3(16 * GiB, 32 * GiB, "16 GiB steps");
			run3(64 * GiB, 128 * GiB, "64 GiB steps");
			run3(37 * GiB + 74 * MiB, 111 * GiB + 38 * MiB, "odd distances");
			run3(128 * GiB, 4 * GiB, "destination far, scratch next to the source");
		}
		CK(hipMemAddressFree(big, 200 * GiB));
	}
	for (int round = 0; round < 2; round++) {
		run((char *)r1, "range 1, 1 GiB aligned");
		run((char *)r1 + 2 * MiB, "range 1 + 2 MiB");
		run((char *)r1 + 512 * MiB, "range 1 + 512 MiB");
		run((char *)r1 + GiB, "range 1 + 1 GiB");
		run((char *)r2, "range 2, 1 GiB aligned");
		run((char *)r2 + 2 * MiB, "range 2 + 2 MiB");
	}
	return 0;
}
