// tr_probe.hip -- what ds_read_b64_tr_b16 returns, and what it costs on the layouts tld_train_attn.hip reads.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/tr_probe.hip -o tools/ubench/_build/tr_probe && tools/ubench/_build/tr_probe
// Part 1 fills LDS with element indices, gives lane l the address 8 l and prints, for each lane, the four indices it received: the
//   documented mapping is lane i of a 16-lane group <- element (i & 15) + 16 j of the group's 64 elements.
// Part 2 times 4096 dependent-free reads per wave (8 waves) on a [256][64] bf16 image at pitch 128 / 144 bytes (tr_acc pattern) and on a
//   [64][256] image at pitch 512 / 528 (tr_seq pattern), against ds_read_b128 fragments of the pitch-144 image.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void map_kernel(short* out) {
    __shared__ __attribute__((aligned(16))) short s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = (short)i;
    __syncthreads();
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

template <int MODE>
__global__ __launch_bounds__(512) void time_kernel(long long* cycles, int* sink, int pitch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 40960 / 4; i += 512) reinterpret_cast<int*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, s = lane & 15, hi = lane >> 5, g1 = (lane >> 4) & 1;
    int acc = 0;
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            if (MODE == 0) {           // tr_acc on a token-major image: rows (u % 16) * 16 + 4 hi + (s >> 2), dims 32 (u / 16 % 2) + 16 g1 + 4 (s & 3)
                const char* p = smem + (((u & 15) * 16 + 4 * hi + (s >> 2)) * pitch) + (32 * ((u >> 4) & 1) + 16 * g1 + 4 * (s & 3)) * 2;
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
                acc += v[0] + v[3];
            } else if (MODE == 1) {    // tr_seq on a dim-major image: rows 16 (u % 4) + 8 hi + (s >> 2), keys 32 (u / 4 % 8) + 16 g1 + 4 (s & 3)
                const char* p = smem + ((16 * (u & 3) + 8 * hi + (s >> 2)) * pitch) + (32 * ((u >> 2) & 7) + 16 * g1 + 4 * (s & 3)) * 2;
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
                acc += v[0] + v[3];
            } else if (MODE == 3) {    // tr_seq on the GEMM's K-step image [64 k-rows][512 B], 64-byte blocks XOR-swizzled with (row & 3) | (row bit 3) << 2
                const int row = 16 * (u & 3) + 8 * hi + (s >> 2), sw = (s >> 2) | (hi << 2);
                const char* p = smem + row * 512 + (((((u >> 2) & 7)) ^ sw) << 6) + 32 * g1 + 8 * (s & 3);
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
                acc += v[0] + v[3];
            } else {                   // ds_read_b128 fragment: row 32 (u % 8) + l31, chunk 2 (u / 8 % 4) + hi
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (32 * (u & 7) + (lane & 31)) * pitch + (2 * ((u >> 3) & 3) + hi) * 16);
                acc += v[0] + v[3];
            }
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
    short* out; hipMalloc(&out, 512);
    hipLaunchKernelGGL(map_kernel, dim3(1), dim3(64), 0, 0, out);
    std::vector<short> h(256);
    hipMemcpy(h.data(), out, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) bad += h[l * 4 + j] != (l & 15) + 16 * j + (l >> 4) * 64;
    printf("mapping: lane l elem j <- element (l & 15) + 16 j + 64 (l >> 4): %s\n", bad ? "NO" : "yes");
    if (bad) for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    long long* cyc; int* sink; hipMalloc(&cyc, 8); hipMalloc(&sink, 512 * 4);
    auto run = [&](auto kern, int pitch, const char* name) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
        long long c = 0;
        for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(1), dim3(512), 40960, 0, cyc, sink, pitch); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
        printf("%-34s pitch %3d: %6.2f clk (s_memtime ticks) per wave-read, 8 waves\n", name, pitch, (double)c / 4096.0);
    };
    run(time_kernel<0>, 128, "tr_acc, token-major [256][64]");
    run(time_kernel<0>, 144, "tr_acc, token-major [256][64]");
    run(time_kernel<0>, 136, "tr_acc, token-major [256][64]");
    run(time_kernel<1>, 512, "tr_seq, dim-major [64][256]");
    run(time_kernel<1>, 528, "tr_seq, dim-major [64][256]");
    run(time_kernel<1>, 520, "tr_seq, dim-major [64][256]");
    run(time_kernel<3>, 512, "tr_seq, [64][256], block-swizzled");
    run(time_kernel<2>, 144, "ds_read_b128 fragment");
    run(time_kernel<2>, 128, "ds_read_b128 fragment");
    return 0;
}
