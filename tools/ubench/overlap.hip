// microbenchmark: does a VALU phase of one wave overlap the MFMA phase of the other wave resident on its SIMD?
// build: hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap.  MI355X result (scalar v_fma_f32 filler): MFMA-only 479 us,
// VALU-only 1204 us, both 1332-1358 us (sum 1683, max 1204): 70 % of the shorter phase hides, in or out of phase at
// start (free-running waves drift apart by themselves).  With v_pk_fma_f32 as the filler almost nothing overlapped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// each wave: `iters` x [ mf MFMA groups of 32 | va VALU groups of 32 pk_fma ]; phase_shift lets odd workgroups start with VALU
__global__ __launch_bounds__(256, 2) void k4(float* out, int iters, int mf, int va, int shift_odd) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x + 2 * e)); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.5f + i + 0.25f * threadIdx.x;
    const float c1 = 1.0001f, c2 = 0.001f;
    const bool odd = __builtin_amdgcn_s_getreg(6148) & 1;     // HW_ID.wave_id[0]: one of the two waves resident on this SIMD
    for (int it = 0; it < iters; ++it) {
        const bool valu_first = shift_odd && odd;
        for (int ph = 0; ph < 2; ++ph) {
            const bool do_valu = (ph == 0) == valu_first;
            if (!do_valu) {
                for (int g = 0; g < mf; ++g) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                }
            } else {
                for (int g = 0; g < va; ++g) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, int grid, int block, int iters, int mf, int va, int shift) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k4, dim3(grid), dim3(block), 0, 0, out, iters, mf, va, shift);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k4, dim3(grid), dim3(block), 0, 0, out, iters, mf, va, shift);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s grid %4d x %3d  iters %3d mf %2d va %2d : %8.1f us\n", name, grid, block, iters, mf, va, ms * 1e3);
    };
    // per CU the same total work in all cases: 8 waves x iters x (mf MFMA-groups + va VALU-groups)
    run("empty", 512, 256, 50, 0, 0, 0);
    run("VALU only va=8", 512, 256, 50, 0, 8, 0);
    run("VALU only va=32", 512, 256, 50, 0, 32, 0);
    run("VALU only va=128", 512, 256, 50, 0, 128, 0);
    run("MFMA only mf=8", 512, 256, 50, 8, 0, 0);
    run("MFMA only mf=16", 512, 256, 50, 16, 0, 0);
    run("both in phase  mf=8 va=128", 512, 256, 50, 8, 128, 0);
    run("both out phase mf=8 va=128", 512, 256, 50, 8, 128, 1);
    run("both in phase  mf=8 va=64", 512, 256, 50, 8, 64, 0);
    run("both out phase mf=8 va=64", 512, 256, 50, 8, 64, 1);
    return 0;
}
