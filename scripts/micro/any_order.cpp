// Does hipExtAnyOrderLaunch (an AQL dispatch without the barrier bit) let consecutive kernels of ONE stream overlap on this runtime / GPU?
// hip_ext.h says the flag "is not supported on AMD GFX9xx boards"; this measures it.  One long single-workgroup kernel N times on one stream:
// in order (N x T), with the flag on all but the first of every group of G (N / G x T if honoured).
// hipcc -O2 --offload-arch=gfx950 any_order.cpp -o any_order.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned* out, unsigned tag)
{
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
	if (threadIdx.x == 0) out[blockIdx.x] = tag;
}
__global__ void chainCheck(const unsigned* in, unsigned* out, unsigned expect)	// ordering check: reads what the kernels before it wrote
{
	if (threadIdx.x == 0) out[0] += in[0] == expect ? 0u : 1u;
}
int main()
{
	unsigned* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	const unsigned long long ticks = 20000;	// 100 MHz wall clock: 200 us
	auto run = [&](int n, int group, unsigned blocks)
	{
		hipStreamSynchronize(s);
		const auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < n; ++i)
		{
			const int flags = (group > 1 && i % group) ? hipExtAnyOrderLaunch : 0;
			hipExtLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s, nullptr, nullptr, flags, ticks, d + 16, (unsigned)i);
		}
		hipStreamSynchronize(s);
		return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
	};
	run(4, 1, 1);
	for (unsigned blocks : { 1u, 256u })
		for (int group : { 1, 2, 4, 8 })
			std::printf("blocks %3u: 8 kernels of 200 us on one stream, groups of %d without the barrier bit: %7.1f us\n", blocks, group, run(8, group, blocks));
	// does a normal launch behind an any-order one wait for BOTH (the barrier bit waits for every earlier packet)?
	unsigned* flag; hipMalloc(&flag, 64); hipMemset(flag, 0, 64);
	for (int rep = 0; rep < 50; ++rep)
	{
		hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, (unsigned long long)2000, d, 1000u + rep);
		hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, (unsigned long long)10000, d + 1, 2000u + rep);
		hipExtLaunchKernelGGL(chainCheck, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, d + 1, flag, 2000u + rep);
	}
	hipStreamSynchronize(s);
	unsigned bad = 0; hipMemcpy(&bad, flag, 4, hipMemcpyDeviceToHost);
	std::printf("a normal launch behind {normal, any-order}: saw the any-order kernel unfinished %u times of 50\n", bad);
	return 0;
}
