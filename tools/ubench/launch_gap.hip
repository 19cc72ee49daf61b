// microbenchmark: GPU-side cost per dependent kernel launch, plain stream vs. a captured hipGraph.
// MI355X: tiny kernels 2.9 us (stream, host-rate bound) vs 1.5 us (graph); ~390 us kernels: 3.7 vs 3.3 us per boundary.
// => a hipGraph over the 75 launches of a denoise step would save ~30 us of 7.3 ms (0.4 %): not built.
// build: hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
// ~40 us of dependent FMAs per thread: long enough that the host is far ahead of the GPU
__global__ void busy(float* p, int iters) { float v = p[threadIdx.x]; for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.001f); if (v == 12345.f) p[0] = v; }
int main() {
    float* d; hipMalloc(&d, 1 << 22);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    for (int grid : {1, 256, 2048}) {
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, grid * 256);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, grid * 256);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("stream  grid %5d: %.2f us per kernel\n", grid, ms * 1e3 / N);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, grid * 256);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("graph   grid %5d: %.2f us per kernel\n", grid, ms * 1e3 / N);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    {
        const int M = 400, iters = 20000;
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, s, d, iters);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, s, d, iters * 50);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float one; hipEventElapsedTime(&one, e0, e1);          // 50 kernels' worth of work in ONE launch
        hipEventRecord(e0, s);
        for (int i = 0; i < M; ++i) hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, s, d, iters);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("long kernels, stream: %.2f us each, pure work %.2f us -> gap %.2f us\n", ms * 1e3 / M, one * 1e3 / 50, ms * 1e3 / M - one * 1e3 / 50);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < M; ++i) hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, s, d, iters);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("long kernels, graph : %.2f us each -> gap %.2f us\n", ms * 1e3 / M, ms * 1e3 / M - one * 1e3 / 50);
    }
    return 0;
}
