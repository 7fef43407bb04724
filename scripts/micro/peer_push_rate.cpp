// What does it cost to copy one buffer into seven others (stand-ins for seven peers) from one kernel, as a function of the store flavour
// and of the grid?  (The peer-store transport's push kernels, csrc/mv_peer.hip: measured 0.4 TB/s there -- is that the stores or the loop?)
// hipcc -O3 --offload-arch=gfx950 peer_push_rate.cpp -o peer_push_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int FLAVOUR> __device__ __forceinline__ void store16(void* p, uint4 v)
{
	u32x4_t w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
	if (FLAVOUR == 0) *reinterpret_cast<uint4*>(p) = v;
	else if (FLAVOUR == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(w) : "memory");
	else if (FLAVOUR == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(w) : "memory");
	else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
}
struct Dst { uint8_t* p[7]; };
template <int FLAVOUR, int BATCH> __global__ __launch_bounds__(256) void push(const uint8_t* src, Dst d, size_t units, int peers)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
	for (size_t i = t; i < units; i += BATCH * n)
	{
		uint4 v[BATCH];
#pragma unroll
		for (int j = 0; j < BATCH; ++j) if (i + j * n < units) v[j] = *reinterpret_cast<const uint4*>(src + ((i + j * n) << 4));
#pragma unroll
		for (int j = 0; j < BATCH; ++j) if (i + j * n < units) for (int p = 0; p < peers; ++p) store16<FLAVOUR>(d.p[p] + ((i + j * n) << 4), v[j]);
	}
}
template <int F, int B> float run(const uint8_t* src, Dst d, size_t bytes, int peers, int grid)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((push<F, B>), dim3(grid), dim3(256), 0, 0, src, d, bytes >> 4, peers);
	hipEventRecord(a);
	for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((push<F, B>), dim3(grid), dim3(256), 0, 0, src, d, bytes >> 4, peers);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms = 0; hipEventElapsedTime(&ms, a, b);
	return ms / 20 * 1000;
}
int main()
{
	const size_t cap = 16u << 20;
	uint8_t* src; hipMalloc(&src, cap); hipMemset(src, 1, cap);
	Dst d; for (int p = 0; p < 7; ++p) { hipMalloc(&d.p[p], cap); hipMemset(d.p[p], 0, cap); }
	const char* names[4] = { "plain", "sc0 sc1", "nt", "sc1" };
	for (size_t bytes : { (size_t)590 << 10, (size_t)1 << 20, (size_t)1812 << 10 })
		for (int grid : { 64, 128, 512, 2048 })
		{
			printf("%5zu KB x 7, grid %4d:", bytes >> 10, grid);
			printf("  %s %5.1f us", names[0], run<0, 4>(src, d, bytes, 7, grid));
			printf("  %s %5.1f us", names[1], run<1, 4>(src, d, bytes, 7, grid));
			printf("  %s %5.1f us", names[2], run<2, 4>(src, d, bytes, 7, grid));
			printf("  %s %5.1f us", names[3], run<3, 4>(src, d, bytes, 7, grid));
			printf("  | batch 1: %s %5.1f us  one peer: %s %5.1f us\n", names[1], run<1, 1>(src, d, bytes, 7, grid), names[1], run<1, 4>(src, d, bytes, 1, grid));
		}
	return 0;
}
