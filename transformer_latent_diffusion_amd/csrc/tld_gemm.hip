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
#include <cstdlib>
#include <cstring>
#include <type_traits>

#ifndef TLD_GLDS_AUX
#define TLD_GLDS_AUX 0      // cache-policy bits of the tile DMA (experiment knob: 2 = nt)
#endif

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


// =================================================================================================
// 256-row tile kernel: BM = 256, BN in {256, 128}, BK = 64, 8 waves, two LDS stages.
//
// Why a bigger tile: with K = 768 a forward's GEMMs are short-K / long-M.  The 128x128 kernel above
// prefetches one K-step = ~600 MFMA cycles per SIMD ahead, less than the DMA's issue->landed latency
// under load, so every barrier waits on memory.  A 256x256 tile holds 4x the MFMA work per K-step
// (32 MFMAs per wave, two waves per SIMD = ~2000 cycles), so the same one-step-ahead DMA has landed by
// the time it is needed, and L2->LDS traffic per flop halves (128 flop/B).  Staging keeps FULL 128-B
// lines per row (BK = 64): a BK = 32 ring with 64-B row segments measured no faster than the small
// kernel -- half-line DMA pieces cost the vector memory path a full line each.
// LDS tile image as in the 128x128 kernel: 128-B rows, chunk ^= (row >> 1) & 7, source-side swizzle.
// Waves: BN=256 -> 2(M) x 4(N), wave tile 128 x 64 (4 x 2 MFMA tiles); BN=128 -> 4 x 2, 64 x 64;
// BN=192 -> 4 x 2, 64 x 96 (residual-add epilogue only: N = 768 then fills 256 CUs in exactly 2 rounds).
template <int BN>
struct G256 {
    static constexpr int BM = 256, BK = 64;
    static constexpr int STAGES = (BN == 128) ? 3 : 2;          // (256+128)*128 B = 48 KiB per stage -> 3 fit
    static constexpr int WN = (BN == 256) ? 4 : 2, WMc = 8 / WN;   // waves along N / M
    static constexpr int WROWS = BM / WMc;                      // rows per wave: 128 or 64
    static constexpr int WCOLS = BN / WN;                       // cols per wave: 64 (BN 256/128) or 96 (BN 192)
    static constexpr int TM = WROWS / 32, TN = WCOLS / 32;      // 32x32 MFMA tiles per wave
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int A_PIECES = BM / 8 / 8, B_PIECES = BN / 8 / 8;     // 1-KiB DMA pieces per wave per tile
    static constexpr int LOADS_PER_TILE = A_PIECES + B_PIECES;
};

// one 1-KiB DMA piece (8 rows x 128 B) of a [rows x 64] bf16 tile; piece index is wave-uniform
__device__ __forceinline__ void stage64_piece(const bf16* __restrict__ g, int ld, int row0, int row_max, int k0,
                                              char* lds_tile, int piece, int lane) {
    const int r = piece * 8 + (lane >> 3);
    const int cphys = lane & 7;
    const int clog = cphys ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < row_max ? gr : row_max - 1;
    const bf16* src = g + (size_t)gr * ld + k0 + clog * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + piece * 1024), 16, 0, TLD_GLDS_AUX);
}

template <int PIECES>
__device__ __forceinline__ void stage64(const bf16* __restrict__ g, int ld, int row0, int row_max, int k0,
                                        char* lds_tile, int wid, int lane) {
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
        const int piece = wid * PIECES + it;
        const int r = piece * 8 + (lane >> 3);
        const int cphys = lane & 7;
        const int clog = cphys ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < row_max ? gr : row_max - 1;
        const bf16* src = g + (size_t)gr * ld + k0 + clog * 8;
        char* dst = lds_tile + piece * 1024;                    // wave-uniform; lane i lands at +16*i
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, TLD_GLDS_AUX);
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N < 0, "unsupported vmcnt");
}

// Epilogue design (the first version stored 2 B per lane straight from the accumulators and was
// store-ISSUE bound: ~2 B/cycle/CU, more time than the whole K loop).  Now every output leaves through
// a per-wave LDS transpose so that global stores are 16 B per lane on whole 128/256-B row segments:
//   * row-major outputs (q|k, MLP-up, residual add) use SWAPPED MFMA operands, D^T = W_tile . A_tile^T,
//     which makes a lane hold 4 consecutive COLUMNS of one row (one 8/16-B LDS write per register group);
//   * the V^T output uses the natural order (a lane holds 4 consecutive TOKENS of one feature).
// Each wave owns LDS_BYTES/8 of the (now idle) staging memory; LDS operations of one wave complete in
// order, so the write->read->write sequence needs no barrier beyond the one that ends the K loop.
template <int BN, int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p) {
    using G = G256<BN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / G::WN, wn = wid % G::WN;
    const int l31 = lane & 31, hi = lane >> 5;

    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + G::BM - 1) / G::BM;
    // tile order inside an XCD's contiguous run: super-rows of 8 m-panels, n-tiles outer, m-panels inner,
    // so any 32 co-resident workgroups of an XCD (one per CU) touch ~8 A panels + ~4 W panels instead of
    // ~3 + all n-tiles: fewer distinct K-slices competing for the 4 MiB L2 at any moment.
    const int tile = xcd_remap(blockIdx.x, ntm * ntn);
    int tm_idx, tn_idx;
    {
        constexpr int SR = 8;
        const int per_sr = SR * ntn;
        const int sr = tile / per_sr, rem = tile - sr * per_sr;
        const int rows_here = (ntm - sr * SR) < SR ? (ntm - sr * SR) : SR;   // last super-row may be short
        tn_idx = rem / rows_here;
        tm_idx = sr * SR + (rem - tn_idx * rows_here);
    }
    const int m0 = tm_idx * G::BM;
    const int n0 = tn_idx * BN;

    // operand order: swapped => lane owns 4 consecutive columns of a row; natural => 4 consecutive rows of a column
    bool swapped = (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_RESID);
    if constexpr (EPI == EPI_QKV) swapped = n0 < 2 * p.d;

    f32x16 acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / G::BK;
    auto issue = [&](int t) {
        char* st = smem + (t % G::STAGES) * G::STAGE_BYTES;
        stage64<G::A_PIECES>(p.A, p.lda, m0, p.M, t * G::BK, st, wid, lane);
        stage64<G::B_PIECES>(p.W, p.ldw, n0, p.N, t * G::BK, st + G::A_BYTES, wid, lane);
    };
    auto load_frags = [&](const char* st, int ks, bf16x8 (&a)[G::TM], bf16x8 (&b)[G::TN]) {
        const int kc = ks * 2 + hi;
#pragma unroll
        for (int i = 0; i < G::TM; ++i) a[i] = read_frag(st, wm * G::WROWS + i * 32 + l31, kc);
#pragma unroll
        for (int j = 0; j < G::TN; ++j) b[j] = read_frag(st + G::A_BYTES, wn * G::WCOLS + j * 32 + l31, kc);
    };

    // prologue: STAGES-1 tiles in flight
    issue(0);
    if constexpr (G::STAGES == 3) { if (nk > 1) issue(1); }
    bf16x8 a0[G::TM], b0[G::TN], a1[G::TM], b1[G::TN];
    auto kloop = [&](auto swp) {
        constexpr bool SW = decltype(swp)::value;
        auto mma = [&](const bf16x8 (&a)[G::TM], const bf16x8 (&b)[G::TN]) {
#pragma unroll
            for (int i = 0; i < G::TM; ++i)
#pragma unroll
                for (int j = 0; j < G::TN; ++j) {
                    if constexpr (SW) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
        };
        for (int t = 0; t < nk; ++t) {
            // this wave's pieces of tile t have landed (with 3 stages tile t+1 may stay in flight: counted vmcnt)
            if (G::STAGES == 3 && t + 1 < nk) wait_vmcnt<G::LOADS_PER_TILE>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // ... everybody's; and the stage of tile t-1 is no longer read
            const char* st = smem + (t % G::STAGES) * G::STAGE_BYTES;
            load_frags(st, 0, a0, b0);
            if (t + G::STAGES - 1 < nk && p.dbg_same_tile != 2) issue(t + G::STAGES - 1);   // DMA address math overlaps the first fragment reads
            load_frags(st, 1, a1, b1);
            mma(a0, b0);
            load_frags(st, 2, a0, b0);
            mma(a1, b1);
            load_frags(st, 3, a1, b1);
            mma(a0, b0);
            mma(a1, b1);
        }
    };
    if constexpr (EPI == EPI_QKV) {
        if (swapped) kloop(std::true_type{}); else kloop(std::false_type{});
    } else if constexpr (EPI == EPI_F32) {
        kloop(std::false_type{});
    } else {
        kloop(std::true_type{});
    }

    const int row0 = m0 + wm * G::WROWS;          // first global row of this wave's sub-tile
    const int col0 = n0 + wn * G::WCOLS;          // first global column
    static_assert(G::WCOLS == 64 || EPI == EPI_BIAS_RESID, "96-column wave tiles: residual epilogue only");

    if constexpr (EPI == EPI_F32) {
        // debug / test path: direct stores, natural layout (col = lane & 31, row = (r&3) + 8(r>>2) + 4 hi)
#pragma unroll
        for (int i = 0; i < G::TM; ++i)
#pragma unroll
            for (int j = 0; j < G::TN; ++j) {
                const int col = col0 + j * 32 + l31;
                if (col >= p.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + i * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) p.c_f32[(size_t)row * p.ldc + col] = acc[i][j][r];
                }
            }
        return;
    } else {
        __builtin_amdgcn_s_barrier();              // all waves are done reading the staging buffers
        char* ws = smem + wid * (G::LDS_BYTES / 8);

        if constexpr (EPI == EPI_BIAS_RESID) {
            // x[row, col] += acc + bias[col]; one 32-row MFMA tile-row per pass through LDS (fp32, padded pitch)
            constexpr int P = G::WCOLS * 4 + 16;
            constexpr int CH = G::WCOLS / 4;               // 16-B chunks per row
#pragma unroll
            for (int i = 0; i < G::TM; ++i) {
#pragma unroll
                for (int j = 0; j < G::TN; ++j)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int cl = j * 32 + 8 * rq + 4 * hi;
                        const int cg = col0 + cl < p.N ? col0 + cl : 0;
                        const float4 bv = *reinterpret_cast<const float4*>(p.bias + cg);
                        float4 v;
                        v.x = acc[i][j][rq * 4 + 0] + bv.x; v.y = acc[i][j][rq * 4 + 1] + bv.y;
                        v.z = acc[i][j][rq * 4 + 2] + bv.z; v.w = acc[i][j][rq * 4 + 3] + bv.w;
                        *reinterpret_cast<float4*>(ws + l31 * P + cl * 4) = v;
                    }
#pragma unroll
                for (int it = 0; it < 32 * CH / 64; ++it) {
                    const int idx = it * 64 + lane;
                    const int rl = idx / CH, ch = idx - rl * CH;
                    const float4 v = *reinterpret_cast<const float4*>(ws + rl * P + ch * 16);
                    const int row = row0 + i * 32 + rl, col = col0 + ch * 4;
                    if (row < p.M && col < p.N) {
                        float4* px = reinterpret_cast<float4*>(p.resid + (size_t)row * p.ldr + col);
                        float4 o = *px;
                        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                        *px = o;
                    }
                }
            }
        } else if constexpr (G::WCOLS == 64) {
            // bf16 outputs
            const bool to_vt = (EPI == EPI_QKV) && !swapped;
            if (!to_vt) {
                // row-major [rows][64] bf16, 128-B pitch (exactly fills the wave's LDS share), 16-B chunk index
                // XOR-swizzled with (row & 7); swapped layout: row = i*32 + l31, cols 4-consecutive
                constexpr int P = 64 * 2;
#pragma unroll
                for (int i = 0; i < G::TM; ++i)
#pragma unroll
                    for (int j = 0; j < G::TN; ++j)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const int cl = j * 32 + 8 * rq + 4 * hi;
                            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                            if constexpr (EPI == EPI_BIAS_BF16) {
                                const int cg = col0 + cl < p.N ? col0 + cl : 0;
                                bv = *reinterpret_cast<const float4*>(p.bias + cg);
                            }
                            bf16x4 pk;
                            pk[0] = (bf16)(acc[i][j][rq * 4 + 0] + bv.x);
                            pk[1] = (bf16)(acc[i][j][rq * 4 + 1] + bv.y);
                            pk[2] = (bf16)(acc[i][j][rq * 4 + 2] + bv.z);
                            pk[3] = (bf16)(acc[i][j][rq * 4 + 3] + bv.w);
                            const int rl = i * 32 + l31;
                            *reinterpret_cast<bf16x4*>(ws + rl * P + ((((cl >> 3) ^ (rl & 7)) << 4) | ((cl & 7) << 1))) = pk;
                        }
#pragma unroll
                for (int it = 0; it < G::WROWS / 8; ++it) {
                    const int idx = it * 64 + lane;
                    const int rl = idx >> 3, ch = idx & 7;
                    const uint4 v = *reinterpret_cast<const uint4*>(ws + rl * P + ((ch ^ (rl & 7)) << 4));
                    const int row = row0 + rl, col = col0 + ch * 8;
                    if (row < p.M && col < p.N)
                        *reinterpret_cast<uint4*>(p.out_bf16 + (size_t)row * p.ldo + col) = v;
                }
            } else {
                // V^T: [64 features][64 tokens] bf16 per pass (pitch 144 B); natural layout:
                // feature = j*32 + l31, tokens 4-consecutive: i*32 + 8 rq + 4 hi
                constexpr int P = 64 * 2 + 16;
                const int cbase = col0 - 2 * p.d;              // feature index (h*64 + c) of local column 0
#pragma unroll
                for (int half = 0; half < G::TM / 2; ++half) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < G::TN; ++j)
#pragma unroll
                            for (int rq = 0; rq < 4; ++rq) {
                                const int i = half * 2 + ii;
                                bf16x4 pk;
                                pk[0] = (bf16)acc[i][j][rq * 4 + 0]; pk[1] = (bf16)acc[i][j][rq * 4 + 1];
                                pk[2] = (bf16)acc[i][j][rq * 4 + 2]; pk[3] = (bf16)acc[i][j][rq * 4 + 3];
                                *reinterpret_cast<bf16x4*>(ws + (j * 32 + l31) * P + (ii * 32 + 8 * rq + 4 * hi) * 2) = pk;
                            }
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int idx = it * 64 + lane;
                        const int f = idx >> 3, ch = idx & 7;                   // feature row, 8-token chunk
                        const uint4 v = *reinterpret_cast<const uint4*>(ws + f * P + ch * 16);
                        const int row = row0 + half * 64 + ch * 8;              // global token row of this chunk
                        if (row < p.M && col0 + f < p.N) {
                            const int b = row / p.ntok, tk = row - b * p.ntok;
                            *reinterpret_cast<uint4*>(p.vt + ((size_t)b * p.d + cbase + f) * p.ntok + tk) = v;
                        }
                    }
                }
            }
        }
    }
}

template <int BN>
void launch256(const GemmParams& p, int epilogue, hipStream_t s) {
    using G = G256<BN>;
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + G::BM - 1) / G::BM;
    dim3 grid(ntm * ntn), block(512);
#define TLD_L256(E)                                                                                   \
    do {                                                                                              \
        static bool once = false;                                                                     \
        if (!once) {                                                                                  \
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<BN, E>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);            \
            once = true;                                                                              \
        }                                                                                             \
        hipLaunchKernelGGL((gemm256_kernel<BN, E>), grid, block, G::LDS_BYTES, s, p);                 \
    } while (0)
    if constexpr (BN == 192) {
        TLD_L256(EPI_BIAS_RESID);
    } else {
        switch (epilogue) {
            case EPI_F32: TLD_L256(EPI_F32); break;
            case EPI_QKV: TLD_L256(EPI_QKV); break;
            case EPI_BIAS_BF16: TLD_L256(EPI_BIAS_BF16); break;
            case EPI_BIAS_RESID: TLD_L256(EPI_BIAS_RESID); break;
            default: break;
        }
    }
#undef TLD_L256
}


// =================================================================================================
// Persistent variant of the 256-row kernel: one workgroup per CU walks a static list of tiles and the
// two-stage LDS ring runs CONTINUOUSLY across tile boundaries -- the DMA of the next tile's first K-step is
// issued during the last K-step of the current tile and lands while the epilogue runs.  That removes, per
// tile, the exposed first-load latency and the workgroup turnover gap (PMC: wave-resident cycles were only
// ~88 % of the kernel's wall time with one short-lived workgroup per tile).
// The epilogue stages one 32-row MFMA tile-row at a time through <= 4.6 KB of LDS per wave, carved out of
// the stage that was consumed last (the other stage is receiving the prefetch).
template <int BN>
struct G256P : G256<BN> {
    static constexpr int STAGES = 2;
    static constexpr int STAGE_BYTES = G256<BN>::A_BYTES + G256<BN>::B_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr int SCRATCH = 4608;             // per-wave epilogue scratch (8 x 4608 <= one stage)
};

template <int BN, int EPI>
__global__ __launch_bounds__(512) void gemm256p_kernel(GemmParams p, int nblocks) {
    using G = G256P<BN>;
    static_assert(8 * G::SCRATCH <= G::STAGE_BYTES, "epilogue scratch must fit in one stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / G::WN, wn = wid % G::WN;
    const int l31 = lane & 31, hi = lane >> 5;

    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + G::BM - 1) / G::BM;
    const int ntiles = ntm * ntn;
    // static schedule: XCD x (= block id % 8, where the dispatcher puts this block) owns a contiguous run of the
    // tile order; its workgroups take that run round-robin
    const int bid = blockIdx.x;
    const int xcd = bid & 7, lidx = bid >> 3;
    const int per_xcd_blocks = (nblocks + 7 - xcd) / 8;               // blocks with this xcd id
    const int q = ntiles >> 3, rr = ntiles & 7;
    const int xbase = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    const int xcount = q + (xcd < rr ? 1 : 0);
    const int my_tiles = lidx < xcount ? (xcount - lidx + per_xcd_blocks - 1) / per_xcd_blocks : 0;
    auto tile_coords = [&](int i, int& m0, int& n0) {
        const int tile = xbase + lidx + i * per_xcd_blocks;
        constexpr int SR = 8;
        const int per_sr = SR * ntn;
        const int sr = tile / per_sr, rem = tile - sr * per_sr;
        const int rows_here = (ntm - sr * SR) < SR ? (ntm - sr * SR) : SR;
        const int tn_idx = rem / rows_here;
        m0 = (sr * SR + (rem - tn_idx * rows_here)) * G::BM;
        n0 = tn_idx * BN;
    };
    if (my_tiles == 0) return;

    const int nk = p.K / G::BK;
    auto issue = [&](int m0, int n0, int k, int g) {
        char* st = smem + (g & 1) * G::STAGE_BYTES;
        stage64<G::A_PIECES>(p.A, p.lda, m0, p.M, k * G::BK, st, wid, lane);
        stage64<G::B_PIECES>(p.W, p.ldw, n0, p.N, k * G::BK, st + G::A_BYTES, wid, lane);
    };
    auto load_frags = [&](const char* st, int ks, bf16x8 (&a)[G::TM], bf16x8 (&b)[G::TN]) {
        const int kc = ks * 2 + hi;
#pragma unroll
        for (int i = 0; i < G::TM; ++i) a[i] = read_frag(st, wm * G::WROWS + i * 32 + l31, kc);
#pragma unroll
        for (int j = 0; j < G::TN; ++j) b[j] = read_frag(st + G::A_BYTES, wn * G::WCOLS + j * 32 + l31, kc);
    };

    int m0, n0;
    tile_coords(0, m0, n0);
    issue(m0, n0, 0, 0);
    int g = 0;                                        // global K-step counter (ring position)
    for (int it = 0; it < my_tiles; ++it) {
        int m0n = 0, n0n = 0;
        const bool has_next = it + 1 < my_tiles;
        if (has_next) tile_coords(it + 1, m0n, n0n);
        bool swapped = (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_RESID);
        if constexpr (EPI == EPI_QKV) swapped = n0 < 2 * p.d;

        f32x16 acc[G::TM][G::TN];
#pragma unroll
        for (int i = 0; i < G::TM; ++i)
#pragma unroll
            for (int j = 0; j < G::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        bf16x8 a0[G::TM], b0[G::TN], a1[G::TM], b1[G::TN];
        auto kloop = [&](auto swp) {
            constexpr bool SW = decltype(swp)::value;
            auto mma = [&](const bf16x8 (&a)[G::TM], const bf16x8 (&b)[G::TN]) {
#pragma unroll
                for (int i = 0; i < G::TM; ++i)
#pragma unroll
                    for (int j = 0; j < G::TN; ++j) {
                        if constexpr (SW) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    }
            };
            // The DMA pieces of the next K-step are spread over the four MFMA groups of this step (a burst of
            // 8 pieces right after the barrier kept both waves of a SIMD out of the MFMA pipe for ~800 cycles),
            // and the last MFMA group of a step is executed AFTER the next step's barrier, so the pipe has
            // register-resident work while the first fragments of the new tile are read from LDS.
            constexpr int NP = G::A_PIECES + G::B_PIECES;           // pieces per wave per K-step
            for (int k = 0; k < nk; ++k, ++g) {
                wait_vmcnt<0>();                   // own DMA pieces of step g landed (and earlier epilogue stores)
                __builtin_amdgcn_s_barrier();      // everybody's; stage (g+1)&1 (operands or epilogue scratch) is idle
                const char* st = smem + (g & 1) * G::STAGE_BYTES;
                char* nst = smem + ((g + 1) & 1) * G::STAGE_BYTES;
                const bool more = (k + 1 < nk) || has_next;
                const int pm0 = (k + 1 < nk) ? m0 : m0n, pn0 = (k + 1 < nk) ? n0 : n0n;
                const int pk = (k + 1 < nk) ? (k + 1) * G::BK : 0;
                auto pieces = [&](int lo, int hi_) {
                    if (!more) return;
#pragma unroll
                    for (int q2 = 0; q2 < NP; ++q2) {
                        if (q2 < lo || q2 >= hi_) continue;
                        if (q2 < G::A_PIECES) stage64_piece(p.A, p.lda, pm0, p.M, pk, nst, wid * G::A_PIECES + q2, lane);
                        else stage64_piece(p.W, p.ldw, pn0, p.N, pk, nst + G::A_BYTES, wid * G::B_PIECES + (q2 - G::A_PIECES), lane);
                    }
                };
                load_frags(st, 0, a0, b0);
                pieces(0, (NP + 3) / 4);
                if (k > 0) mma(a1, b1);            // deferred: k-slice 3 of the previous step (fragments already in registers)
                load_frags(st, 1, a1, b1);
                pieces((NP + 3) / 4, (NP + 1) / 2);
                mma(a0, b0);
                load_frags(st, 2, a0, b0);
                pieces((NP + 1) / 2, (3 * NP + 3) / 4);
                mma(a1, b1);
                load_frags(st, 3, a1, b1);
                pieces((3 * NP + 3) / 4, NP);
                mma(a0, b0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // k-slice 3 fragments are in registers before the stage is released
            }
            mma(a1, b1);                           // k-slice 3 of the tile's last step
        };
        if constexpr (EPI == EPI_QKV) {
            if (swapped) kloop(std::true_type{}); else kloop(std::false_type{});
        } else if constexpr (EPI == EPI_F32) {
            kloop(std::false_type{});
        } else {
            kloop(std::true_type{});
        }

        const int row0 = m0 + wm * G::WROWS;
        const int col0 = n0 + wn * G::WCOLS;
        if constexpr (EPI == EPI_F32) {
#pragma unroll
            for (int i = 0; i < G::TM; ++i)
#pragma unroll
                for (int j = 0; j < G::TN; ++j) {
                    const int col = col0 + j * 32 + l31;
                    if (col >= p.N) continue;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + i * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                        if (row < p.M) p.c_f32[(size_t)row * p.ldc + col] = acc[i][j][r];
                    }
                }
        } else {
            __builtin_amdgcn_s_barrier();          // all waves finished reading the last stage: reuse it as scratch
            char* ws = smem + ((g - 1) & 1) * G::STAGE_BYTES + wid * G::SCRATCH;
            if constexpr (EPI == EPI_BIAS_RESID) {
                constexpr int P = 32 * 4 + 16;      // one 32x32 fp32 tile, padded pitch
#pragma unroll
                for (int i = 0; i < G::TM; ++i)
#pragma unroll
                    for (int j = 0; j < G::TN; ++j) {
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const int cl = 8 * rq + 4 * hi;
                            const int cg = col0 + j * 32 + cl < p.N ? col0 + j * 32 + cl : 0;
                            const float4 bv = *reinterpret_cast<const float4*>(p.bias + cg);
                            float4 v;
                            v.x = acc[i][j][rq * 4 + 0] + bv.x; v.y = acc[i][j][rq * 4 + 1] + bv.y;
                            v.z = acc[i][j][rq * 4 + 2] + bv.z; v.w = acc[i][j][rq * 4 + 3] + bv.w;
                            *reinterpret_cast<float4*>(ws + l31 * P + cl * 4) = v;
                        }
#pragma unroll
                        for (int itr = 0; itr < 4; ++itr) {
                            const int idx = itr * 64 + lane;
                            const int rl = idx >> 3, ch = idx & 7;
                            const float4 v = *reinterpret_cast<const float4*>(ws + rl * P + ch * 16);
                            const int row = row0 + i * 32 + rl, col = col0 + j * 32 + ch * 4;
                            if (row < p.M && col < p.N) {
                                float4* px = reinterpret_cast<float4*>(p.resid + (size_t)row * p.ldr + col);
                                float4 o = *px;
                                o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                                *px = o;
                            }
                        }
                    }
            } else if constexpr (G::WCOLS == 64) {
                const bool to_vt = (EPI == EPI_QKV) && !swapped;
                if (!to_vt) {
                    // one 32-row x 64-col bf16 slab per pass: 128-B pitch, 16-B chunks XOR-swizzled with (row & 7)
                    constexpr int P = 128;
#pragma unroll
                    for (int i = 0; i < G::TM; ++i) {
#pragma unroll
                        for (int j = 0; j < G::TN; ++j)
#pragma unroll
                            for (int rq = 0; rq < 4; ++rq) {
                                const int cl = j * 32 + 8 * rq + 4 * hi;
                                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                                if constexpr (EPI == EPI_BIAS_BF16) {
                                    const int cg = col0 + cl < p.N ? col0 + cl : 0;
                                    bv = *reinterpret_cast<const float4*>(p.bias + cg);
                                }
                                bf16x4 pk;
                                pk[0] = (bf16)(acc[i][j][rq * 4 + 0] + bv.x);
                                pk[1] = (bf16)(acc[i][j][rq * 4 + 1] + bv.y);
                                pk[2] = (bf16)(acc[i][j][rq * 4 + 2] + bv.z);
                                pk[3] = (bf16)(acc[i][j][rq * 4 + 3] + bv.w);
                                *reinterpret_cast<bf16x4*>(ws + l31 * P + ((((cl >> 3) ^ (l31 & 7)) << 4) | ((cl & 7) << 1))) = pk;
                            }
#pragma unroll
                        for (int itr = 0; itr < 4; ++itr) {
                            const int idx = itr * 64 + lane;
                            const int rl = idx >> 3, ch = idx & 7;
                            const uint4 v = *reinterpret_cast<const uint4*>(ws + rl * P + ((ch ^ (rl & 7)) << 4));
                            const int row = row0 + i * 32 + rl, col = col0 + ch * 8;
                            if (row < p.M && col < p.N)
                                *reinterpret_cast<uint4*>(p.out_bf16 + (size_t)row * p.ldo + col) = v;
                        }
                    }
                } else {
                    // V^T: per pass [32 features (one tn)][64 tokens (two tm)] bf16, pitch 144 B
                    constexpr int P = 64 * 2 + 16;
                    const int cbase = col0 - 2 * p.d;
#pragma unroll
                    for (int half = 0; half < G::TM / 2; ++half)
#pragma unroll
                        for (int j = 0; j < G::TN; ++j) {
#pragma unroll
                            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                                for (int rq = 0; rq < 4; ++rq) {
                                    const int i = half * 2 + ii;
                                    bf16x4 pk;
                                    pk[0] = (bf16)acc[i][j][rq * 4 + 0]; pk[1] = (bf16)acc[i][j][rq * 4 + 1];
                                    pk[2] = (bf16)acc[i][j][rq * 4 + 2]; pk[3] = (bf16)acc[i][j][rq * 4 + 3];
                                    *reinterpret_cast<bf16x4*>(ws + l31 * P + (ii * 32 + 8 * rq + 4 * hi) * 2) = pk;
                                }
#pragma unroll
                            for (int itr = 0; itr < 4; ++itr) {
                                const int idx = itr * 64 + lane;
                                const int f = idx >> 3, ch = idx & 7;
                                const uint4 v = *reinterpret_cast<const uint4*>(ws + f * P + ch * 16);
                                const int row = row0 + half * 64 + ch * 8;
                                const int fg = j * 32 + f;
                                if (row < p.M && col0 + fg < p.N) {
                                    const int b = row / p.ntok, tk = row - b * p.ntok;
                                    *reinterpret_cast<uint4*>(p.vt + ((size_t)b * p.d + cbase + fg) * p.ntok + tk) = v;
                                }
                            }
                        }
                }
            }
        }
        m0 = m0n; n0 = n0n;
    }
}

template <int BN>
void launch256p(const GemmParams& p, int epilogue, hipStream_t s) {
    using G = G256P<BN>;
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + G::BM - 1) / G::BM;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipGetDevice(&dev);
        hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int nblocks = ntm * ntn < ncu ? ntm * ntn : ncu;
    dim3 grid(nblocks), block(512);
#define TLD_L256P(E)                                                                                  \
    do {                                                                                              \
        static bool once = false;                                                                     \
        if (!once) {                                                                                  \
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<BN, E>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);            \
            once = true;                                                                              \
        }                                                                                             \
        hipLaunchKernelGGL((gemm256p_kernel<BN, E>), grid, block, G::LDS_BYTES, s, p, nblocks);       \
    } while (0)
    if constexpr (BN == 192) {
        TLD_L256P(EPI_BIAS_RESID);
    } else {
        switch (epilogue) {
            case EPI_F32: TLD_L256P(EPI_F32); break;
            case EPI_QKV: TLD_L256P(EPI_QKV); break;
            case EPI_BIAS_BF16: TLD_L256P(EPI_BIAS_BF16); break;
            case EPI_BIAS_RESID: TLD_L256P(EPI_BIAS_RESID); break;
            default: break;
        }
    }
#undef TLD_L256P
}

}  // namespace

static int gemm_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("TLD_GEMM");          // "128": 2-stage 128x128 kernel; default: 256-row ring kernel
        v = (e && !strcmp(e, "128")) ? 128 : (e && !strcmp(e, "np")) ? 256 : 257;   // 257: persistent 256-row kernel
    }
    return v;
}

void launch_gemm(const GemmParams& p, int epilogue, hipStream_t s) {
    if (gemm_variant() >= 256 && p.K % 64 == 0) {
        // BN = 256 unless that leaves the last round of workgroups mostly empty on 256 CUs; then prefer the
        // widest tile whose workgroup count is a whole number of rounds (192 for the residual epilogue), else 128.
        const long ntm = (p.M + 255) / 256;
        const long blocks256 = ntm * ((p.N + 255) / 256);
        const bool narrow = (p.N % 256 != 0) || (blocks256 % 256 != 0 && blocks256 < 3 * 256);
        static const char* force = getenv("TLD_GEMM_BN");
        int bn = narrow ? 128 : 256;
        if (narrow && epilogue == EPI_BIAS_RESID && p.N % 192 == 0 && (ntm * (p.N / 192)) % 256 == 0) bn = 192;
        if (force) bn = atoi(force);
        if (bn == 192 && (epilogue != EPI_BIAS_RESID || p.N % 192)) bn = 128;
        if (gemm_variant() == 257) {
            if (bn == 192) launch256p<192>(p, epilogue, s);
            else if (bn == 128) launch256p<128>(p, epilogue, s);
            else launch256p<256>(p, epilogue, s);
            return;
        }
        if (bn == 192) launch256<192>(p, epilogue, s);
        else if (bn == 128) launch256<128>(p, epilogue, s);
        else launch256<256>(p, epilogue, s);
        return;
    }
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
