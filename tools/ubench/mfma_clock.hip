// microbenchmark: register-resident v_mfma_f32_32x32x16_bf16 rate AND the shader clock it runs at, for different operand data.
// Every wave keeps 8 independent accumulators and issues `iters` x 8 MFMAs on fragments loaded once from `src`; wave 0 of every
// workgroup brackets its loop with s_memtime (shader-clock ticks) and s_memrealtime (100 MHz ticks): clock = d(memtime) / d(realtime).
// Output per data kind: PFLOP/s from the HIP-event time, mean / min / max effective clock over the workgroups, and the matrix-pipe
// utilisation AT THAT CLOCK (MFMA cycles issued / elapsed shader cycles; a 32x32x16 bf16 MFMA is 8 passes = 32 cycles of its SIMD's pipe).
// build: hipcc --offload-arch=gfx950 -O3 mfma_clock.hip -o mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256, 2) void k(const bf16x8* __restrict__ src, float* __restrict__ out, uint64_t* __restrict__ ticks, int iters) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = src[(blockIdx.x * 256 + threadIdx.x) * 6 + i];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = src[(blockIdx.x * 256 + threadIdx.x) * 6 + 4 + i];
    const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int maxgrid = 512;
    const size_t nsrc = (size_t)maxgrid * 256 * 6 * 8;
    std::vector<uint16_t> h(nsrc);
    uint16_t* dsrc; float* out; uint64_t* ticks;
    hipMalloc(&dsrc, nsrc * 2); hipMalloc(&out, maxgrid * 256 * 4); hipMalloc(&ticks, maxgrid * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"zeros", "ones", "N(0,1) bf16", "N(0,1) small (x 1/64)"};
    for (int kind = 0; kind < 4; ++kind) {
        srand(1);
        for (size_t i = 0; i < nsrc; ++i) {
            float v = 0.f;
            if (kind == 1) v = 1.f;
            if (kind >= 2) {
                const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
                v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * (kind == 3 ? 1.f / 64 : 1.f);
            }
            h[i] = f2bf(v);
        }
        hipMemcpy(dsrc, h.data(), nsrc * 2, hipMemcpyHostToDevice);
        for (int grid : {256, 512}) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, (const bf16x8*)dsrc, out, ticks, iters);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, (const bf16x8*)dsrc, out, ticks, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint64_t> t(2 * grid);
            hipMemcpy(t.data(), ticks, grid * 16, hipMemcpyDeviceToHost);
            double cmean = 0, cmin = 1e9, cmax = 0, util = 0;
            for (int g = 0; g < grid; ++g) {
                const double mhz = (double)t[2 * g] / (double)t[2 * g + 1] * 100.0;
                cmean += mhz; cmin = mhz < cmin ? mhz : cmin; cmax = mhz > cmax ? mhz : cmax;
                util += (double)iters * 8 * 32 * (grid / 256) / (double)t[2 * g];      // waves per SIMD x MFMA cycles / elapsed cycles
            }
            cmean /= grid; util /= grid;
            const double flops = (double)grid * 4 * iters * 8 * 32768.0;
            printf("%-24s %d wave(s)/SIMD: %7.1f us  %.3f PFLOP/s  clock mean %4.0f MHz (min %4.0f, max %4.0f)  matrix pipe busy %.2f of the elapsed shader cycles\n",
                   names[kind], grid / 256, ms * 1e3, flops / (ms * 1e-3) / 1e15, cmean, cmin, cmax, util);
        }
    }
    return 0;
}
