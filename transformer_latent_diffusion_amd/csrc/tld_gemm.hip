// tld_gemm.hip -- bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T  (fp32 accumulate)
//
// Replaces the ATen mm/addmm/1x1-conv call sites of the reference's decoder block
// (tld/transformer_blocks.py:54,58 qkv_linear; :95 and :104 the two 1x1 convs of MLPSepConv).
//
// Structure: 128x128 block tile, BK = 64, 4 waves (2x2), each wave a 64x64 sub-tile built from
// 2x2 v_mfma_f32_32x32x16_bf16 tiles.  Both operands are K-contiguous, so A and W tiles are staged
// by direct global->LDS DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction = 8 rows x 128 B).
// The DMA destination is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE
// address and again on the ds_read_b128 side (same involution on both):
//     physical 16-B chunk = logical chunk ^ ((row >> 1) & 7)        within a 128-B tile row
// which makes every ds_read_b128 lane group touch 16 distinct 16-B slots of the 256-B bank row.
// Two LDS stages; one barrier per K-step; the DMA of step t+1 is issued before the MFMAs of step t.
#include "tld_common.h"

namespace tld {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileBytes = BM * BK * 2;            // 16 KiB per operand per stage
constexpr int kStageBytes = 2 * kTileBytes;        // A + W
constexpr int kLdsBytes = 2 * kStageBytes;         // two stages = 64 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One wave stages 32 rows (4 x 8-row DMA pieces) of a [128 x 64] bf16 tile.
__device__ __forceinline__ void stage_rows(const bf16* __restrict__ g, int ld, int row0, int row_max,
                                           int k0, char* lds_tile, int wid, int lane) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = wid * 32 + it * 8 + (lane >> 3);
        const int cphys = lane & 7;
        const int clog = cphys ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < row_max ? gr : row_max - 1;        // clamp: rows past the edge are never stored
        const bf16* src = g + (size_t)gr * ld + k0 + clog * 8;
        char* dst = lds_tile + (wid * 32 + it * 8) * (BK * 2);   // wave-uniform; lane i lands at +16*i
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row, int kchunk) {
    const int off = row * (BK * 2) + ((kchunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + off);
}

// XCD-aware bijective remap: the dispatcher places block b on XCD b % 8; give each XCD a contiguous
// run of tiles (n fastest inside an m-panel) so A panels and the W matrix stay in that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 1, wc = wid & 1;

    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (tile / ntn) * BM;
    const int n0 = (tile % ntn) * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    stage_rows(p.A, p.lda, m0, p.M, 0, smem, wid, lane);
    stage_rows(p.W, p.ldw, n0, p.N, 0, smem + kTileBytes, wid, lane);

    for (int t = 0; t < nk; ++t) {
        __syncthreads();                                  // tile t landed; stage (t+1)&1 is free
        char* cur = smem + (t & 1) * kStageBytes;
        if (t + 1 < nk) {
            char* nxt = smem + ((t + 1) & 1) * kStageBytes;
            stage_rows(p.A, p.lda, m0, p.M, (t + 1) * BK, nxt, wid, lane);
            stage_rows(p.W, p.ldw, n0, p.N, (t + 1) * BK, nxt + kTileBytes, wid, lane);
        }
        const char* At = cur;
        const char* Wt = cur + kTileBytes;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int kc = ks * 2 + (lane >> 5);
            bf16x8 a0 = read_frag(At, wr * 64 + (lane & 31), kc);
            bf16x8 a1 = read_frag(At, wr * 64 + 32 + (lane & 31), kc);
            bf16x8 b0 = read_frag(Wt, wc * 64 + (lane & 31), kc);
            bf16x8 b1 = read_frag(Wt, wc * 64 + 32 + (lane & 31), kc);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // ---- epilogue.  C layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + (lane & 31);
            const int rbase = m0 + wr * 64 + i * 32 + 4 * (lane >> 5);
            if (col >= p.N) continue;
            if constexpr (EPI == EPI_F32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) p.c_f32[(size_t)row * p.ldc + col] = acc[i][j][r];
                }
            } else if constexpr (EPI == EPI_BIAS_BF16) {
                const float bv = p.bias[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) p.out_bf16[(size_t)row * p.ldo + col] = (bf16)(acc[i][j][r] + bv);
                }
            } else if constexpr (EPI == EPI_BIAS_RESID) {
                const float bv = p.bias[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) {
                        resid_t* px = p.resid + (size_t)row * p.ldr + col;
                        *px = (resid_t)((float)*px + acc[i][j][r] + bv);
                    }
                }
            } else if constexpr (EPI == EPI_QKV) {
                const int twod = 2 * p.d;
                if (col < twod) {                              // q | k : row-major [M, 2d]
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row < p.M) p.out_bf16[(size_t)row * p.ldo + col] = (bf16)acc[i][j][r];
                    }
                } else {                                       // v : transposed [b, h, c, tok]
                    const int c = col - twod;                  // h * 64 + cc  -> row of the [B, d, ntok] view
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int row = rbase + 8 * rq;        // 4 consecutive tokens row..row+3
                        if (row < p.M) {
                            const int b = row / p.ntok, tk = row - b * p.ntok;
                            bf16x4 pk;
                            pk[0] = (bf16)acc[i][j][rq * 4 + 0];
                            pk[1] = (bf16)acc[i][j][rq * 4 + 1];
                            pk[2] = (bf16)acc[i][j][rq * 4 + 2];
                            pk[3] = (bf16)acc[i][j][rq * 4 + 3];
                            *reinterpret_cast<bf16x4*>(p.vt + ((size_t)b * p.d + c) * p.ntok + tk) = pk;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace

void launch_gemm(const GemmParams& p, int epilogue, hipStream_t s) {
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    dim3 grid(ntm * ntn), block(256);
    switch (epilogue) {
        case EPI_F32: hipLaunchKernelGGL(gemm_bf16_kernel<EPI_F32>, grid, block, kLdsBytes, s, p); break;
        case EPI_QKV: hipLaunchKernelGGL(gemm_bf16_kernel<EPI_QKV>, grid, block, kLdsBytes, s, p); break;
        case EPI_BIAS_BF16: hipLaunchKernelGGL(gemm_bf16_kernel<EPI_BIAS_BF16>, grid, block, kLdsBytes, s, p); break;
        case EPI_BIAS_RESID: hipLaunchKernelGGL(gemm_bf16_kernel<EPI_BIAS_RESID>, grid, block, kLdsBytes, s, p); break;
        default: break;
    }
}

}  // namespace tld
