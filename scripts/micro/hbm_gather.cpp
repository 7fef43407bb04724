// Microbenchmark: what HBM delivers for (a) a streaming read, (b) random 128-byte lines, (c) random 512-byte bricks (4 consecutive
// lines), (d) random 2-KB runs -- the access shapes of the view march.  hipcc --offload-arch=gfx950 -O3 hbm_gather.cpp -o hbm_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void stream(const uint4* __restrict__ p, size_t n, uint32_t* out)
{
	uint32_t acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) *out = acc;
}

// every group of `lanesPerRun` consecutive lanes reads one run of runBytes (16 B per lane), runs placed pseudo-randomly
__global__ void gather(const uint4* __restrict__ p, size_t nRuns, uint32_t runBytes, uint32_t iters, uint32_t* out)
{
	const uint32_t lanesPerRun = runBytes / 16u;
	uint32_t acc = 0;
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t nThreads = (size_t)gridDim.x * blockDim.x;
	for (uint32_t it = 0; it < iters; ++it)
	{
		const size_t g = (tid + (size_t)it * nThreads) / lanesPerRun;
		uint64_t h = g * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
		const size_t run = h % nRuns;
		const uint4 v = p[run * lanesPerRun + (tid % lanesPerRun)];
		acc ^= v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u) *out = acc;
}

// like the march: each lane reads 8 bytes out of a random 128-byte line (the rest of the line is fetched for nothing unless a neighbour uses it)
__global__ void gatherSparse(const uint2* __restrict__ p, size_t nLines, uint32_t iters, uint32_t* out)
{
	uint32_t acc = 0;
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t nThreads = (size_t)gridDim.x * blockDim.x;
	for (uint32_t it = 0; it < iters; ++it)
	{
		uint64_t h = (tid + (size_t)it * nThreads) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
		const uint2 v = p[(h % nLines) * 16u + (tid & 15u)];
		acc ^= v.x ^ v.y;
	}
	if (acc == 0x12345678u) *out = acc;
}

int main()
{
	const size_t bytes = (size_t)2 << 30;
	uint4* d; uint32_t* out;
	hipMalloc(&d, bytes); hipMalloc(&out, 4); hipMemset(d, 1, bytes);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	auto time = [&](auto&& launch) { launch(); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms / 5; };
	{
		const float ms = time([&] { hipLaunchKernelGGL(stream, dim3(256 * 32), dim3(256), 0, 0, d, bytes / 16, out); });
		printf("stream read            %7.1f GB in %.3f ms -> %.2f TB/s\n", bytes / 1e9, ms, bytes / ms / 1e9);
	}
	const uint32_t iters = 16;
	const dim3 grid(256 * 64), block(256);
	const size_t threads = (size_t)grid.x * block.x;
	for (uint32_t runBytes : { 128u, 256u, 512u, 2048u, 16384u })
	{
		const float ms = time([&] { hipLaunchKernelGGL(gather, grid, block, 0, 0, d, bytes / runBytes, runBytes, iters, out); });
		const double moved = (double)threads * iters * 16;
		printf("random runs of %5u B   %7.2f GB in %.3f ms -> %.2f TB/s\n", runBytes, moved / 1e9, ms, moved / ms / 1e9);
	}
	{
		const float ms = time([&] { hipLaunchKernelGGL(gatherSparse, grid, block, 0, 0, (const uint2*)d, bytes / 128, iters, out); });
		const double lines = (double)threads * iters;
		printf("8 B of a random line   %7.2f G lines in %.3f ms -> %.2f TB/s of lines, %.2f G lines/s\n", lines / 1e9, ms, lines * 128 / ms / 1e9, lines / ms / 1e6);
	}
	return 0;
}
