// Host-side cost of the HIP calls a frame is made of (enqueue only), on this runtime: hipcc -O2 host_call_cost.cpp -o host_call_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(int* p) { if (p && threadIdx.x == 1024) *p = 1; }
struct Big { char b[640]; };
__global__ void tinyBig(Big s, int* p) { if (p && threadIdx.x == 1024) *p = s.b[0]; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	hipStream_t s[4]; for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
	hipEvent_t e[8]; for (auto& x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
	int* d; hipMalloc(&d, 1 << 20); void* h; hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
	const int n = 2000; Big big{};
	auto run = [&](const char* name, auto f)
	{
		hipDeviceSynchronize();
		for (int i = 0; i < 200; ++i) f(i);
		hipDeviceSynchronize();
		const double t0 = now();
		for (int i = 0; i < n; ++i) f(i);
		const double t1 = now();
		hipDeviceSynchronize();
		printf("%-52s %6.2f us per call\n", name, (t1 - t0) / n);
	};
	run("kernel launch, 1 stream, 8-byte args", [&](int) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], d); });
	run("kernel launch, 1 stream, 648-byte args", [&](int) { hipLaunchKernelGGL(tinyBig, dim3(1), dim3(64), 0, s[0], big, d); });
	run("kernel launch, round-robin over 4 streams", [&](int i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[i & 3], d); });
	run("hipEventRecord", [&](int i) { hipEventRecord(e[i & 7], s[0]); });
	run("hipStreamWaitEvent (event of another stream, done)", [&](int i) { hipStreamWaitEvent(s[1], e[i & 7], 0); });
	run("launch + record + wait on another stream + launch", [&](int i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], d); hipEventRecord(e[i & 7], s[0]); hipStreamWaitEvent(s[1], e[i & 7], 0); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[1], d); });
	run("hipMemcpyAsync H2D 4 KiB pinned", [&](int) { hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, s[0]); });
	run("hipMemcpyAsync D2H 512 B pinned", [&](int) { hipMemcpyAsync(h, d, 512, hipMemcpyDeviceToHost, s[0]); });
	run("hipMemcpyAsync D2D 64 KiB", [&](int) { hipMemcpyAsync(d, d + 65536, 65536, hipMemcpyDeviceToDevice, s[0]); });
	run("hipGetLastError", [&](int) { (void)hipGetLastError(); });

	// ---- a frame's chain as a captured graph: 10 dependent kernels with 648-byte arguments on one stream ----
	hipGraph_t graph; hipGraphExec_t exec;
	hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal);
	for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(tinyBig, dim3(1), dim3(64), 0, s[0], big, d);
	if (hipStreamEndCapture(s[0], &graph) != hipSuccess || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { printf("graph capture failed\n"); return 1; }
	run("10 launches (648-byte args), 1 stream", [&](int) { for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(tinyBig, dim3(1), dim3(64), 0, s[0], big, d); });
	run("hipGraphLaunch of the same 10 kernels", [&](int) { hipGraphLaunch(exec, s[0]); });
	run("hipGraphLaunch, round-robin over 3 streams", [&](int i) { hipGraphLaunch(exec, s[i % 3]); });
	{
		size_t nn = 0; hipGraphGetNodes(graph, nullptr, &nn);
		std::vector<hipGraphNode_t> nodes(nn); hipGraphGetNodes(graph, nodes.data(), &nn);
		void* args[2] = { &big, &d };
		hipKernelNodeParams kp{}; kp.func = (void*)tinyBig; kp.gridDim = dim3(1); kp.blockDim = dim3(64); kp.kernelParams = args;
		run("10 x hipGraphExecKernelNodeSetParams + hipGraphLaunch", [&](int) { for (size_t k = 0; k < nn; ++k) hipGraphExecKernelNodeSetParams(exec, nodes[k], &kp); hipGraphLaunch(exec, s[0]); });
	}
	// GPU-side: how long the 10-kernel chain takes from enqueue to completion, direct and as a graph
	auto chain = [&](const char* name, auto f)
	{
		for (int i = 0; i < 50; ++i) { f(); hipStreamSynchronize(s[0]); }
		const double t0 = now();
		for (int i = 0; i < 500; ++i) { f(); hipStreamSynchronize(s[0]); }
		printf("%-52s %6.2f us per chain (enqueue + run + sync)\n", name, (now() - t0) / 500);
	};
	chain("10 dependent tiny kernels, direct", [&]() { for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(tinyBig, dim3(1), dim3(64), 0, s[0], big, d); });
	chain("10 dependent tiny kernels, graph", [&]() { hipGraphLaunch(exec, s[0]); });
	chain("1 tiny kernel, direct", [&]() { hipLaunchKernelGGL(tinyBig, dim3(1), dim3(64), 0, s[0], big, d); });
	// throughput of back-to-back dependent tiny kernels on one stream (the floor a chain of small kernels pays per kernel)
	{
		hipDeviceSynchronize();
		const double t0 = now();
		for (int i = 0; i < 4000; ++i) hipLaunchKernelGGL(tinyBig, dim3(1), dim3(64), 0, s[0], big, d);
		hipStreamSynchronize(s[0]);
		printf("%-52s %6.2f us per kernel\n", "4000 dependent tiny kernels, one stream, total", (now() - t0) / 4000);
		const double t1 = now();
		for (int i = 0; i < 400; ++i) hipGraphLaunch(exec, s[0]);
		hipStreamSynchronize(s[0]);
		printf("%-52s %6.2f us per kernel\n", "400 graphs of 10 tiny kernels, one stream, total", (now() - t1) / 4000);
	}
	return 0;
}
