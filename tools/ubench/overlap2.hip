// microbenchmark: which VALU instruction kinds run beside MFMAs on gfx950?  Two set-ups per filler kind:
//   phase:  two waves per SIMD, each alternating a long MFMA phase and a long VALU phase (free running)   -- cross-wave overlap
//   inter:  ONE wave per SIMD whose instruction stream is [1 MFMA, n VALU] repeated (all independent)        -- in-wave co-issue
// reported: MFMA-only, VALU-only and combined times; hidden = (mfma + valu - both) / min(mfma, valu).
// build: hipcc --offload-arch=gfx950 -O3 overlap2.hip -o overlap2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int K> __device__ __forceinline__ void filler(float (&v)[16], int i, float c1, float c2, unsigned ones) {
    if constexpr (K == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
    else if constexpr (K == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double*>(&v[(2 * i) & 15])) : "v"(*reinterpret_cast<double*>(&v[14])), "v"(*reinterpret_cast<double*>(&v[12])));
    else if constexpr (K == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    else if constexpr (K == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
    else if constexpr (K == 4) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(ones));
    else if constexpr (K == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
    else if constexpr (K == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
    else if constexpr (K == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[(2 * i) & 15])) : "v"(*reinterpret_cast<double*>(&v[14])));
    else if constexpr (K == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c2));
    else if constexpr (K == 9) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 1) & 15]));
    else if constexpr (K == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(c1));
    else if constexpr (K == 11) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(ones));
}

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int K>
__global__ __launch_bounds__(256, 2) void phase_k(float* out, int iters, int mf, int va) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x + 2 * e)); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.5f + i + 0.25f * threadIdx.x;
    const float c1 = 1.0001f, c2 = 0.001f; const unsigned ones = 0x3F803F80u;
    for (int it = 0; it < iters; ++it) {
        for (int g = 0; g < mf; ++g) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) MFMA(acc[i]);
        }
        for (int g = 0; g < va; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) filler<K>(v, i, c1, c2, ones);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one stream: per step `nm` MFMAs (0 or 1) then NV fillers
template <int K, int NV>
__global__ __launch_bounds__(256, 2) void inter_k(float* out, int iters, int with_mfma, int with_valu) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x + 2 * e)); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.5f + i + 0.25f * threadIdx.x;
    const float c1 = 1.0001f, c2 = 0.001f; const unsigned ones = 0x3F803F80u;
    if (with_mfma && with_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                MFMA(acc[q & 3]);
#pragma unroll
                for (int i = 0; i < NV; ++i) filler<K>(v, (q * NV + i) & 15, c1, c2, ones);
            }
        }
    } else if (with_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) MFMA(acc[q & 3]);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int i = 0; i < NV; ++i) filler<K>(v, (q * NV + i) & 15, c1, c2, ones);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

float* out; hipEvent_t e0, e1;
template <typename F> float timeit(F f) {
    f(); f();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}

template <int K> void run(const char* name) {
    // phase: grid 512 x 256 threads = 2 workgroups per CU = 2 waves per SIMD; per wave iters x (mf x 32 MFMAs | va x 64 fillers)
    const int it = 40, mf = 4, va = 12;
    float m = timeit([&] { hipLaunchKernelGGL(phase_k<K>, dim3(512), dim3(256), 0, 0, out, it, mf, 0); });
    float v = timeit([&] { hipLaunchKernelGGL(phase_k<K>, dim3(512), dim3(256), 0, 0, out, it, 0, va); });
    float bo = timeit([&] { hipLaunchKernelGGL(phase_k<K>, dim3(512), dim3(256), 0, 0, out, it, mf, va); });
    // inter: 1 wave per SIMD (grid 256) and 2 waves per SIMD (grid 512), [1 MFMA, 6 fillers]
    float r[2][3];
    for (int w = 0; w < 2; ++w) {
        const int grid = w ? 512 : 256;
        r[w][0] = timeit([&] { hipLaunchKernelGGL((inter_k<K, 6>), dim3(grid), dim3(256), 0, 0, out, 400, 1, 0); });
        r[w][1] = timeit([&] { hipLaunchKernelGGL((inter_k<K, 6>), dim3(grid), dim3(256), 0, 0, out, 400, 0, 1); });
        r[w][2] = timeit([&] { hipLaunchKernelGGL((inter_k<K, 6>), dim3(grid), dim3(256), 0, 0, out, 400, 1, 1); });
    }
    auto hid = [](float a, float b, float c) { return (a + b - c) / (a < b ? a : b); };
    printf("%-20s phase(2w): mfma %6.1f valu %6.1f both %6.1f hidden %4.2f | inter 1w: %6.1f %6.1f %6.1f hidden %4.2f | inter 2w: %6.1f %6.1f %6.1f hidden %4.2f\n",
           name, m, v, bo, hid(m, v, bo), r[0][0], r[0][1], r[0][2], hid(r[0][0], r[0][1], r[0][2]), r[1][0], r[1][1], r[1][2], hid(r[1][0], r[1][1], r[1][2]));
}

int main() {
    hipMalloc(&out, 1024 * 512 * 4);
    hipEventCreate(&e0); hipEventCreate(&e1);
    run<0>("v_fma_f32");
    run<1>("v_pk_fma_f32");
    run<2>("v_exp_f32");
    run<3>("v_cvt_pk_bf16_f32");
    run<4>("v_dot2c_f32_bf16");
    run<5>("v_max3_f32");
    run<6>("v_mul_f32");
    run<7>("v_pk_mul_f32");
    run<8>("v_add_f32");
    run<9>("v_permlane32_swap");
    run<10>("v_mov_b32");
    run<11>("v_xor_b32");
    return 0;
}
