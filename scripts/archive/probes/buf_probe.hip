// buf_probe.hip -- what the hardware's buffer bounds check and unaligned 16-byte accesses do on
// gfx950: hipcc --offload-arch=gfx950 -O2 scripts/archive/probes/buf_probe.hip -o scripts/archive/probes/buf_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// each lane stores / loads 16 B at byte offset off0 + 16 lane into a buffer of `records` bytes
__global__ void k_store(unsigned *out, unsigned records, unsigned off0)
{
	rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)records, 0x00020000);
	const unsigned l = threadIdx.x;
	__builtin_amdgcn_raw_buffer_store_b128(u4{l * 4 + 1, l * 4 + 2, l * 4 + 3, l * 4 + 4}, r, off0 + l * 16, 0, 0);
}
__global__ void k_load_lds(const unsigned *in, unsigned records, unsigned off0, unsigned *dump)
{
	__shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
	for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xdeadbeef;
	__syncthreads();
	rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)records, 0x00020000);
	__builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, off0 + threadIdx.x * 16, 0, 0, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (int i = threadIdx.x; i < 256; i += 64) dump[i] = lds[i];
}
__global__ void k_global_lds(const unsigned *in, unsigned *dump)
{
	__shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
	for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xdeadbeef;
	__syncthreads();
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + threadIdx.x * 4),
		(__attribute__((address_space(3))) void *)lds, 16, 0, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (int i = threadIdx.x; i < 256; i += 64) dump[i] = lds[i];
}
__global__ void k_global_store(unsigned *out)
{
	const unsigned l = threadIdx.x;
	*(u4 *)(out + l * 4) = u4{l * 4 + 1, l * 4 + 2, l * 4 + 3, l * 4 + 4};
}

int main()
{
	unsigned *d, *dump;
	CHECK(hipMalloc(&d, 8192)); CHECK(hipMalloc(&dump, 1024));
	std::vector<unsigned> h(2048), hd(256);
	// 1. store, buffer of 1000 bytes (250 dwords): lane 62 covers dwords 248..251 -- half inside
	CHECK(hipMemset(d, 0, 8192));
	k_store<<<1, 64>>>(d, 1000, 0); CHECK(hipDeviceSynchronize());
	CHECK(hipMemcpy(h.data(), d, 8192, hipMemcpyDeviceToHost));
	printf("store, records=1000 B: dwords 246..253 = "); for (int i = 246; i < 254; i++) printf("%u ", h[i]); printf("\n");
	// 2. buffer_load..lds with the same range
	for (int i = 0; i < 2048; i++) h[i] = 1000 + i;
	CHECK(hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice));
	k_load_lds<<<1, 64>>>(d, 1000, 0, dump); CHECK(hipDeviceSynchronize());
	CHECK(hipMemcpy(hd.data(), dump, 1024, hipMemcpyDeviceToHost));
	printf("buffer_load lds, records=1000 B: dwords 246..253 = "); for (int i = 246; i < 254; i++) printf("%u ", hd[i]); printf("\n");
	// 3. unaligned (4-byte aligned) sources: buffer_load lds with offset 4, global_load_lds from in+1
	k_load_lds<<<1, 64>>>(d, 4096, 4, dump); CHECK(hipDeviceSynchronize());
	CHECK(hipMemcpy(hd.data(), dump, 1024, hipMemcpyDeviceToHost));
	printf("buffer_load lds, byte offset 4: dwords 0..5 = "); for (int i = 0; i < 6; i++) printf("%u ", hd[i]); printf(" (expect 1001..)\n");
	k_global_lds<<<1, 64>>>(d + 1, dump); CHECK(hipDeviceSynchronize());
	CHECK(hipMemcpy(hd.data(), dump, 1024, hipMemcpyDeviceToHost));
	printf("global_load_lds from base+4 B: dwords 0..5 = "); for (int i = 0; i < 6; i++) printf("%u ", hd[i]); printf(" (expect 1001..)\n");
	// 4. unaligned 16-byte stores
	CHECK(hipMemset(d, 0, 8192));
	k_global_store<<<1, 64>>>(d + 1); CHECK(hipDeviceSynchronize());
	CHECK(hipMemcpy(h.data(), d, 8192, hipMemcpyDeviceToHost));
	printf("global_store_dwordx4 to base+4 B: dwords 0..5 = "); for (int i = 0; i < 6; i++) printf("%u ", h[i]); printf(" (expect 0 1 2 3 4 5)\n");
	CHECK(hipMemset(d, 0, 8192));
	k_store<<<1, 64>>>(d, 4096, 4); CHECK(hipDeviceSynchronize());
	CHECK(hipMemcpy(h.data(), d, 8192, hipMemcpyDeviceToHost));
	printf("buffer_store_dwordx4 at byte offset 4: dwords 0..5 = "); for (int i = 0; i < 6; i++) printf("%u ", h[i]); printf(" (expect 0 1 2 3 4 5)\n");
	return 0;
}
