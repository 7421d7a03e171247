// Micro-benchmark: bandwidth of rectangle copies inside an 8192 x 8192 float image (pitch 32 KiB)
// versus the rectangle's width/position and the staging pitch.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
struct R { const char *src; char *dst; long sp, dp; int wbytes, h; };
template <int ROWS>
__global__ __launch_bounds__(256) void k_copy(R r, int interleave_rows)
{
	const int nbx = (r.wbytes + 4095) / 4096;
	const int bx = blockIdx.x % nbx; int by = blockIdx.x / nbx;
	const long x = (long)bx * 4096 + threadIdx.x * 16;
	if (x >= r.wbytes) return;
	const char *s = r.src + (long)by * ROWS * r.sp + x;
	char *d = r.dst + (long)by * ROWS * r.dp + x;
	u4 v[ROWS];
#pragma unroll
	for (int i = 0; i < ROWS; i++) v[i] = __builtin_nontemporal_load((const u4 *)(s + (long)i * r.sp));
#pragma unroll
	for (int i = 0; i < ROWS; i++) __builtin_nontemporal_store(v[i], (u4 *)(d + (long)i * r.dp));
}
static float run(R r, hipStream_t st)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	const int nb = ((r.wbytes + 4095) / 4096) * (r.h / 8);
	for (int i = 0; i < 3; i++) k_copy<8><<<nb, 256, 0, st>>>(r, 0);
	hipEventRecord(a, st);
	for (int i = 0; i < 10; i++) k_copy<8><<<nb, 256, 0, st>>>(r, 0);
	hipEventRecord(b, st); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main()
{
	const long N = 8192, P = N * 4;
	char *a, *b, *c;
	hipMalloc(&a, P * N * 2); hipMalloc(&b, P * N * 2); hipMalloc(&c, (P + 4096) * N);
	hipMemset(a, 1, P * N * 2); hipMemset(b, 2, P * N * 2);
	hipStream_t st; hipStreamCreate(&st);
	struct { const char *name; R r; } cases[] = {
		{"full width 8192 rows x 32 KiB", {a, b, P, P, (int)P, (int)N}},
		{"right half 8192 rows x 16 KiB (pitch 32 KiB both)", {a + P / 2, b + P / 2, P, P, (int)(P / 2), (int)N}},
		{"left half 8192 rows x 16 KiB (pitch 32 KiB both)", {a, b, P, P, (int)(P / 2), (int)N}},
		{"right half, src dense (pitch 16 KiB) -> dst pitch 32 KiB", {a, b + P / 2, P / 2, P, (int)(P / 2), (int)N}},
		{"right half, src pitch 32 KiB + 4 KiB -> dst pitch 32 KiB", {c + P / 2, b + P / 2, P + 4096, P, (int)(P / 2), (int)N}},
		{"quarter [8-16 KiB) 8192 rows", {a + P / 4, b + P / 4, P, P, (int)(P / 4), (int)N}},
		{"full width, src pitch 32K+4K", {c, b, P + 4096, P, (int)P, (int)N}},
	};
	for (auto &cs : cases) {
		float ms = run(cs.r, st);
		double bytes = 2.0 * cs.r.wbytes * cs.r.h;
		printf("%-62s %8.1f us  %7.1f GB/s (read+write)\n", cs.name, ms * 1e3, bytes / ms / 1e6);
	}
	return 0;
}
