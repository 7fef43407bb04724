// Exhaustive check, over all 2^32 float bit patterns, of the division-free form of GetStep's factorEv (RayMarch.hlsli:186):
//   min(1/256 / |d|, 2)   ==   min(RN-corrected rcp(|d|) * 2^-8, 2)
// The numerator is a power of two, so the IEEE quotient is the correctly rounded reciprocal scaled exactly; v_rcp_f32 is within 1 ulp
// and one fused Newton step r1 = fma(fma(-x, r0, 1), r0, r0) rounds it correctly -- this program proves (or refutes) that for every input.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -o rcp_exact.bin rcp_exact.cpp   (the library's flags)
// tests/test_gpu_configs.py::test_step_factor_equals_the_ieee_quotient_for_every_input builds and runs it
#define MV_STEP_RCP 1
#include "../../multivolumes_amd/csrc/mv_device.hpp"
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float viaDivision(float d) { return __builtin_fminf(1.0f / 256.0f / __builtin_fabsf(d), 2.0f); }
// the library's own function (mv_device.hpp, MV_STEP_RCP = 1: v_rcp_f32, one fused Newton step, max(e, -1) for the NaN cases, x 2^-8)
__device__ __forceinline__ float viaRcp(float d) { return mv::StepFactorEv(d); }
__global__ void check(unsigned long long* out)
{
	const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;	// 2^20 threads, 2^12 patterns each
	unsigned long long bad = 0, badFinite = 0; uint32_t first = 0xFFFFFFFFu;	// bad: any; badFinite: finite inputs; first: smallest |d| pattern
	for (uint32_t k = 0; k < 4096u; ++k)
	{
		const uint32_t bits = k * (1u << 20) + lane;
		const float d = __uint_as_float(bits);
		const uint32_t a = __float_as_uint(viaDivision(d)), b = __float_as_uint(viaRcp(d));
		if (a != b)
		{
			++bad;
			const uint32_t ex = (bits >> 23) & 255u;
			if (ex != 255u) { ++badFinite; if ((bits & 0x7FFFFFFFu) < first) first = bits & 0x7FFFFFFFu; }
		}
	}
	if (bad) { atomicAdd(out, bad); atomicAdd(out + 1, badFinite); atomicMin(out + 2, (unsigned long long)first); }
}
int main()
{
	unsigned long long* d; unsigned long long h[3] = { 0, 0, ~0ull };
	hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
	check<<<4096, 256>>>(d);
	if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
	hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	float smallest = 0.0f; const uint32_t sb = (uint32_t)h[2]; memcpy(&smallest, &sb, 4);
	printf("rcp_exact: %llu of 4294967296 inputs differ: %llu infinities / NaNs, %llu finite, the smallest of them |d| = %g (bits 0x%08x = 2^118 is %g)\n",
		h[0], h[0] - h[1], h[1], h[1] ? smallest : 0.0f, sb, 3.3230699894622813e35);
	return (h[0] != h[1] || (h[1] && smallest < 3.3230699894622813e35f)) ? 1 : 0;	// fails if a non-finite input or any |d| < 2^118 differs
}
