// How many HIP streams of one process really run side by side on this runtime (hardware queues), and whether GPU_MAX_HW_QUEUES changes it:
// one long single-workgroup kernel per stream on S streams; S concurrent streams finish in one kernel's time.
// hipcc -O2 --offload-arch=gfx950 hw_queues.cpp -o hw_queues;  ./hw_queues;  GPU_MAX_HW_QUEUES=8 ./hw_queues
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long long cycles, int* out)
{
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < cycles) {}
	if (out && threadIdx.x == 9999) *out = 1;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const char* e = getenv("GPU_MAX_HW_QUEUES");
	printf("GPU_MAX_HW_QUEUES=%s\n", e ? e : "(unset)");
	hipStream_t s[12];
	for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
	int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);	// kHz
	const long long cycles = (long long)rate * 1000 / 5000;	// 200 us
	for (int S : { 1, 2, 3, 4, 5, 6, 8, 12 })
	{
		for (int w = 0; w < 3; ++w) { for (int i = 0; i < S; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], cycles, nullptr); hipDeviceSynchronize(); }
		const double t0 = now();
		for (int r = 0; r < 5; ++r) { for (int i = 0; i < S; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], cycles, nullptr); hipDeviceSynchronize(); }
		printf("%2d streams x one 200 us kernel: %.0f us per round\n", S, (now() - t0) / 5);
	}
	return 0;
}
