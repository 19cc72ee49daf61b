// gemm_2wg.hip -- EXPERIMENT: C[M,N] = A[M,K] . W[N,K]^T with TWO independent 4-wave workgroups per CU on 256 x 128 tiles.
//
// Why: in the shipped kernels everything that is not the K loop -- epilogue stores (issue-bound at ~15-25 B/clk/CU, profiles/
// r03_ring_tile_trace.txt), the LayerNorm folds, the depthwise + GELU epilogue -- runs with the matrix pipe idle, because the CU's one
// 8-wave workgroup is in the same phase everywhere.  Two unsynchronised workgroups per CU (the 256-token attention kernel's recipe) let one
// workgroup's epilogue overlap the other's MFMAs.  Cost: 256 x 128 tiles move 1.5 x the operand bytes per flop of 256 x 256 ones and a
// workgroup alone on a SIMD has one wave to hide its own latencies.
//
//   4 waves as 2 (M) x 2 (N), wave tile 128 x 64 (the shipped per-wave shape: 4 x 2 tiles of 32 x 32, 128 accumulator registers).
//   A (256 x 64 per K-tile, 32 KiB): two LDS stages by global_load_lds (8 pieces per wave), same row image + XOR swizzle as the shipped kernel.
//   W (128 x 64 per K-tile): never through LDS -- a lane's MFMA fragment (row, 8 consecutive k) is 16 contiguous bytes of W, loaded
//      straight into registers one K-tile ahead (8 x 16 B per lane and K-tile; the two waves of a column pair read the same rows: L1 hits).
//   One s_barrier per K-tile.  LDS 64 KiB stages + 16 KiB epilogue scratch = 80 KiB -> two workgroups per CU.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gemm_2wg tools/ubench/gemm_2wg.hip && ./gemm_2wg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int A_STAGE = BM * BK * 2;                 // 32 KiB
constexpr int SCR_OFF = 2 * A_STAGE;
constexpr int LDS_BYTES = SCR_OFF + 4 * 4096;        // 80 KiB

__device__ __forceinline__ bf16x8 read_frag(const char* st, int row, int kchunk) {
    return *reinterpret_cast<const bf16x8*>(st + row * 128 + ((kchunk ^ ((row >> 1) & 7)) << 4));
}

// epi_work: extra VALU work per output element in the epilogue (emulates a fused elementwise epilogue: 0 = plain stores)
__global__ __launch_bounds__(256, 2) void gemm2wg_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, bf16* __restrict__ C,
                                                          int M, int N, int K, int store, int epi_work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int ntn = N / BN, ntiles = (M / BM) * ntn;
    const int nk = K / BK;

    // persistent: 2 workgroups per CU; XCD-contiguous tile order as in the shipped kernel
    const int nblocks = gridDim.x;
    const int bid = blockIdx.x, xcd = bid & 7, lidx = bid >> 3;
    const int per_xcd_blocks = (nblocks + 7 - xcd) / 8;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcount = q8 + (xcd < r8 ? 1 : 0);
    const int my_tiles = lidx < xcount ? (xcount - lidx + per_xcd_blocks - 1) / per_xcd_blocks : 0;

    unsigned voffA[8];
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = xbase + lidx + it * per_xcd_blocks;
        const int tm = tile / ntn;
        const int m0 = tm * BM, n0 = (tile - tm * ntn) * BN;
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = (wid * 8 + q) * 8 + (ln >> 3);
                voffA[q] = (unsigned)(m0 + r) * (unsigned)(K * 2) + (unsigned)(((ln & 7) ^ ((r >> 1) & 7)) * 16);
            }
        }
        auto stage_a = [&](int kt, int slot) {
            const char* base = reinterpret_cast<const char*>(A) + (size_t)kt * (BK * 2);
            asm volatile("" : "+s"(base));
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                unsigned o = voffA[q];
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_global_load_lds((gptr_t)(base + o), (lptr_t)(smem + slot * A_STAGE + (wid * 8 + q) * 1024), 16, 0, 0);
            }
        };
        // W fragments of one K-tile: [qb][ks], lane (row l31, half hi): 16 B at W[(n0 + wc * 64 + qb * 32 + l31) * K + kt * 64 + ks * 16 + hi * 8]:
        // a uniform base (SGPR pair) + ONE 32-bit lane offset + the k-slice as an immediate
        unsigned wl = (unsigned)l31 * (unsigned)(K * 2) + (unsigned)hi * 16u;
        asm volatile("" : "+v"(wl));
        auto load_b = [&](int kt, bf16x8 (&fb)[2][4]) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const char* base = reinterpret_cast<const char*>(W) + ((size_t)(n0 + wc * 64 + qb * 32) * K + (size_t)kt * BK) * 2;
                asm volatile("" : "+s"(base));
                unsigned o = wl;
                asm volatile("" : "+v"(o));
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fb[qb][ks] = *reinterpret_cast<const bf16x8*>(base + o + ks * 32);
            }
        };
        // A fragment read addresses: one register per k-slice (row wr * 128 + l31, chunk (2 ks + hi) ^ swizzle); row block / stage as immediates
        unsigned ra[4];
        {
            int l31v = l31, hiv = hi;
            asm volatile("" : "+v"(l31v), "+v"(hiv));
            const int sw = (l31v >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                ra[ks] = (unsigned)((wr * 128 + l31v) * 128 + (((ks * 2 + hiv) ^ sw) << 4));
                asm volatile("" : "+v"(ra[ks]));
            }
        }

        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        bf16x8 fb0[2][4], fb1[2][4], fa[2][4];
        __builtin_amdgcn_s_barrier();                    // the previous tile's epilogue scratch / last stage are free
        stage_a(0, 0);
        load_b(0, fb0);
        auto body = [&](int kt, bf16x8 (&fb)[2][4], bf16x8 (&fbn)[2][4]) {
            const int so = (kt & 1) * A_STAGE;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own A pieces and W fragments of K-tile kt
            __builtin_amdgcn_s_barrier();                         // everybody's; everybody is done reading the other stage
            if (kt + 1 < nk) { stage_a(kt + 1, (kt + 1) & 1); load_b(kt + 1, fbn); }
#pragma unroll
            for (int qa = 0; qa < 2; ++qa) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) fa[ii][ks] = *reinterpret_cast<const bf16x8*>(smem + ra[ks] + so + (qa * 64 + ii * 32) * 128);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
                            acc[qa * 2 + ii][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[qb][ks], fa[ii][ks], acc[qa * 2 + ii][qb], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
        };
        for (int kt = 0; kt < nk; kt += 2) {             // (nk even: the two fragment sets swap roles without copies)
            body(kt, fb0, fb1);
            body(kt + 1, fb1, fb0);
        }
        if (store) {
            char* ws = smem + SCR_OFF + wid * 4096;
            const int row0 = m0 + wr * 128, col0 = n0 + wc * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int cl = j2 * 32 + 8 * rq + 4 * hi;
                        bf16x4 pk;
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            float v = acc[i][j2][rq * 4 + e2];
                            for (int wk = 0; wk < epi_work; ++wk) v = fmaf(v, 1.0001f, 0.0001f * v);   // emulated elementwise epilogue work
                            pk[e2] = (bf16)v;
                        }
                        *reinterpret_cast<bf16x4*>(ws + l31 * 128 + ((((cl >> 3) ^ (l31 & 7)) << 4) | ((cl & 7) << 1))) = pk;
                    }
#pragma unroll
                for (int itr = 0; itr < 4; ++itr) {
                    const int idx = itr * 64 + lane;
                    const int rl = idx >> 3, ch = idx & 7;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(ws + rl * 128 + ((ch ^ (rl & 7)) << 4));
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(C + (size_t)(row0 + i * 32 + rl) * N + col0 + ch * 8));
                }
            }
        } else if (acc[0][0][0] == 12345.678f || acc[3][1][5] == 12345.678f || acc[1][1][3] == 777.f || acc[2][0][9] == 777.f) {
            C[0] = (bf16)1.f;
        }
    }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    struct Shape { const char* name; int M, N, K; } shapes[] = {{"512 ", 512, 512, 512}, {"4k  ", 4096, 4096, 4096}, {"qkv ", 32768, 2304, 768},
                                                                  {"up  ", 32768, 3072, 768}, {"down", 32768, 768, 3072}};
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm2wg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(gemm2wg_kernel));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm2wg_kernel, 256, LDS_BYTES);
    printf("gemm2wg_kernel: %d registers per thread, %zu bytes scratch, %d B LDS, %d workgroups per CU\n", fa.numRegs, (size_t)fa.localSizeBytes, LDS_BYTES, occ);
    int ncu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
    for (auto& s : shapes) {
        const size_t na = (size_t)s.M * s.K, nw = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
        std::vector<uint16_t> ha(na), hw(nw);
        uint32_t st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : ha) v = f2bf(rnd());
        for (auto& v : hw) v = f2bf(rnd());
        bf16 *dA, *dW, *dC;
        hipMalloc(&dA, na * 2); hipMalloc(&dW, nw * 2); hipMalloc(&dC, nc * 2);
        hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice);
        hipMemcpy(dW, hw.data(), nw * 2, hipMemcpyHostToDevice);
        const int ntiles = (s.M / BM) * (s.N / BN);
        const int grid = ntiles < 2 * ncu ? ntiles : 2 * ncu;
        auto launch = [&](int store, int work) { hipLaunchKernelGGL(gemm2wg_kernel, dim3(grid), dim3(256), LDS_BYTES, 0, dA, dW, dC, s.M, s.N, s.K, store, work); };
        std::vector<uint16_t> hc(nc), hc0;
        double worst = 0.0;
        int mism = 0;
        for (int run = 0; run < 3; ++run) {
            hipMemset(dC, 0xff, nc * 2);
            launch(1, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", s.name, hipGetErrorString(hipGetLastError())); return 1; }
            hipMemcpy(hc.data(), dC, nc * 2, hipMemcpyDeviceToHost);
            if (run == 0) {
                hc0 = hc;
                for (int t = 0; t < 4000; ++t) {
                    const size_t m = ((size_t)t * 7919 + 13) % s.M, n = ((size_t)t * 104729 + 7) % s.N;
                    double ref = 0.0;
                    for (int k = 0; k < s.K; ++k) ref += (double)bf2f(ha[m * s.K + k]) * (double)bf2f(hw[n * s.K + k]);
                    worst = fmax(worst, fabs((double)bf2f(hc[m * s.N + n]) - ref) / (fabs(ref) + 0.02 * sqrt((double)s.K)));
                }
            } else if (memcmp(hc.data(), hc0.data(), nc * 2) != 0) ++mism;
        }
        printf("%s M=%d N=%d K=%d: max rel err %.2e; %d of 2 repeat runs differ\n", s.name, s.M, s.N, s.K, worst, mism);
        if (s.M < 1024) { hipFree(dA); hipFree(dW); hipFree(dC); continue; }
        struct { const char* label; int store, work; } modes[] = {{"stores            ", 1, 0}, {"no stores         ", 0, 0}, {"stores + 8 VALU/el", 1, 4}, {"stores + 32 VALU/el", 1, 16}};
        for (auto& md : modes) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) launch(md.store, md.work);
            hipEventRecord(e0, 0);
            const int iters = 20;
            for (int i = 0; i < iters; ++i) launch(md.store, md.work);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / iters, tf = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
            printf("  %s %s: %8.1f us  %7.1f TFLOP/s (%.1f%% of 2500)\n", s.name, md.label, us, tf, tf / 25.0);
        }
        hipFree(dA); hipFree(dW); hipFree(dC);
    }
    return 0;
}
