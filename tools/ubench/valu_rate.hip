// microbenchmark: issue cost (cycles per wave64 instruction) of the VALU operations the fused depthwise + GELU epilogue is
// made of, alone and with 1 / 2 waves per SIMD: v_fma_f32, v_pk_fma_f32, v_dot2c_f32_bf16, v_rcp_f32, v_exp_f32,
// v_cvt_pk_bf16_f32, v_pk_mul_f32.  16 independent chains per wave, s_memtime around 64 x 16 instructions.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

template <int OP>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int reps) {
    float v[16], w[16];
    for (int i = 0; i < 16; ++i) { v[i] = 0.5f + 0.01f * (threadIdx.x + i); w[i] = 1.0f + 0.001f * i; }
    const float c1 = 1.0001f, c2 = 0.001f;
    unsigned p1 = 0x3f803f80u + threadIdx.x, p2 = 0x3c003c00u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
                if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double*>(&v[i & ~1])) : "v"(*reinterpret_cast<const double*>(&w[0])), "v"(*reinterpret_cast<const double*>(&w[2])));
                if constexpr (OP == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[i]) : "v"(p1), "v"(p2));
                if constexpr (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                if constexpr (OP == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                if constexpr (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[i]) : "v"(w[i]), "v"(c1));
                if constexpr (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[i & ~1])) : "v"(*reinterpret_cast<const double*>(&w[0])));
                if constexpr (OP == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
                if constexpr (OP == 8) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[i]) : "v"(p1), "v"(p2));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char* name, unsigned long long* dout, float* sink) {
    const int reps = 64;
    for (int block : {256, 512, 1024}) {          // 1, 2, 4 waves per SIMD (one workgroup per CU)
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(block), 0, 0, dout, sink, reps);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, dout, sizeof(unsigned long long) * (block / 64), hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < block / 64; ++i) avg += (double)h[i];
        avg /= block / 64;
        const double per = avg / (reps * 64.0);
        printf("%-20s %d wave(s)/SIMD: %6.2f cycles per instruction per wave  -> %5.2f cycles of SIMD time each\n", name,
               block / 256, per, per / (block / 256));
    }
}

int main() {
    unsigned long long* dout; float* sink;
    hipMalloc(&dout, 4096 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    run<0>("v_fma_f32", dout, sink);
    run<7>("v_mul_f32", dout, sink);
    run<1>("v_pk_fma_f32", dout, sink);
    run<6>("v_pk_mul_f32", dout, sink);
    run<2>("v_dot2c_f32_bf16", dout, sink);
    run<3>("v_rcp_f32", dout, sink);
    run<4>("v_exp_f32", dout, sink);
    run<5>("v_cvt_pk_bf16_f32", dout, sink);
    run<8>("v_bfi_b32", dout, sink);
    return 0;
}
