// Where does global_load_lds_dwordx4 put a lane's 16 bytes when only some lanes are active?  (gfx950)
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/lds_dma.cpp -o /tmp/lds_dma ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef const __attribute__((address_space(1))) void* GPtr;
typedef __attribute__((address_space(3))) void* LPtr;
__global__ void k(const uint4* src, uint4* dst, unsigned long long mask)
{
	__shared__ uint4 buf[192];
	const int lane = threadIdx.x;
	for (int i = lane; i < 192; i += 64) buf[i] = make_uint4(0xdeadu, 0xdeadu, 0xdeadu, 0xdeadu);
	__syncthreads();
	if ((mask >> lane) & 1ull) __builtin_amdgcn_global_load_lds((GPtr)(src + 100 + 3 * lane), (LPtr)(buf + 8), 16, 0, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (int i = lane; i < 192; i += 64) dst[i] = buf[i];
}
int main()
{
	std::vector<uint4> h(1024);
	for (int i = 0; i < 1024; ++i) h[i] = make_uint4(i, i, i, i);
	uint4 *s, *d;
	hipMalloc(&s, 1024 * 16); hipMalloc(&d, 192 * 16);
	hipMemcpy(s, h.data(), 1024 * 16, hipMemcpyHostToDevice);
	const unsigned long long masks[3] = { ~0ull, 0xffffffffull, 0xf0f0f0f0f0f0f0f0ull };
	for (unsigned long long m : masks)
	{
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, m);
		std::vector<uint4> o(192);
		hipMemcpy(o.data(), d, 192 * 16, hipMemcpyDeviceToHost);
		printf("mask %016llx:", m);
		for (int i = 0; i < 80; ++i) { if (o[i].x == 0xdeadu) printf(" ."); else printf(" %u", (o[i].x - 100) / 3); }
		printf("\n");
	}
	return 0;
}
