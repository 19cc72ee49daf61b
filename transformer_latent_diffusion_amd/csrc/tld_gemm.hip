// tld_gemm.hip -- bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T  (fp32 accumulate)
//
// Replaces the ATen mm/addmm/1x1-conv call sites of the reference's decoder block
// (tld/transformer_blocks.py:54,58 qkv_linear; :95 and :104 the two 1x1 convs of MLPSepConv).
//
// One persistent workgroup per CU (8 waves) walks a static list of 256-row tiles; see the comment above
// gemm256p_kernel for the pipeline and DESIGN.md 4.1 for the measurements that shaped it.
#include "tld_common.h"
#include "tld_attn_core.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

// -DTLD_DBG_EPI builds honour GemmParams::dbg_epi in the fused depthwise epilogue (cost attribution):
//   1 = no global stores, 2 = identity instead of GELU, 4 = skip the conv phase, 8 = skip the LDS image write too,
//   16 = (residual-add epilogue) no read of the residual either
#ifdef TLD_DBG_EPI
#define TLD_EPI_BIT(b) ((p.dbg_epi & (b)) != 0)
#else
#define TLD_EPI_BIT(b) false
#endif

#ifndef TLD_KLOOP_RING
#define TLD_KLOOP_RING 1      // counted-vmcnt half-tile ring K loop for 256 x 256 tiles (see kloop_ring); 0 = never instantiate it (the two-stage staggered loop everywhere)
#endif

namespace tld {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Fragment read of a [rows][64] bf16 LDS tile image (128-B rows).  The image is written by DMA, whose
// destination is lane-linear, so the bank-conflict swizzle lives on the per-lane SOURCE address of the DMA and
// is undone here:  physical 16-B chunk = logical chunk ^ ((row >> 1) & 7).  The 16 rows of every
// ds_read_b128 lane group then cover all 16 slots of the 256-B bank row.
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row, int kchunk) {
    const int off = row * 128 + ((kchunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + off);
}

// Tile geometry: BM = 256 rows (one 16x16-token image), BN in {256, 192, 128}, BK = 64 (FULL 128-B lines per
// row per K-step: a BK = 32 ring with 64-B row segments measured no faster than a 128x128 kernel), 8 waves,
// two LDS stages.  Waves: BN=256 -> 2(M) x 4(N), wave tile 128 x 64 (4 x 2 MFMA tiles of 32x32);
// BN=128 -> 4 x 2, 64 x 64;  BN=192 -> 4 x 2, 64 x 96 (residual-add epilogue only: N = 768 then fills
// 256 CUs in exactly 2 rounds);  BN=384 -> 2 x 4, 128 x 96 (residual-add epilogue only, 192 accumulator registers
// per lane, single-buffered fragments, all 160 KB of LDS: N = 768 at M = 32 K is ONE round of 256 workgroups).
template <int BN>
struct G256 {
    static constexpr int BM = 256, BK = 64;
    static constexpr int WN = (BN >= 256) ? 4 : 2, WMc = 8 / WN;   // waves along N / M
    static constexpr int WROWS = BM / WMc;                      // rows per wave: 128 or 64
    static constexpr int WCOLS = BN / WN;                       // cols per wave: 64 (BN 256/128) or 96 (BN 192)
    static constexpr int TM = WROWS / 32, TN = WCOLS / 32;      // 32x32 MFMA tiles per wave
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    static constexpr int A_PIECES = BM / 8 / 8, B_PIECES = BN / 8 / 8;     // 1-KiB DMA pieces per wave per K-step
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N < 0, "unsupported vmcnt");
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
    // EPI_QKV_LN: behind the two stages, the tile's raw row partial sums (256 x 64 B, DMA) and the reduced (mean, rstd)
    static constexpr int LN_RAW = LDS_BYTES, LN_ST = LN_RAW + 256 * 8 * kLnSlots, LN_CB = LN_ST + 256 * 8, QKVLN_LDS = LN_CB + 2048;   // + c1 | b1 of the tile's columns
    // fused depthwise epilogue (EPI_UP_DWCONV2): image of TOKEN PAIRS, one dword = (token 2p, token 2p+1)
    // of one channel, [128 pairs][256 channels] with a 1-KiB pitch (no padding needed: every access of a wave is a
    // contiguous run), an all-zero pair-row for the rows above / below the image, then the (mean, rstd) pairs
    static constexpr int IMG2_PITCH = 1024, IMG2_BYTES = 128 * IMG2_PITCH;
    static constexpr int ZROW_OFF = IMG2_BYTES, ZROW_BYTES = 16 * IMG2_PITCH;          // (16 pair-rows: one row of the 32-wide grid; the 16-wide form reads 8)
    static constexpr int ROWSTAT2_OFF = ZROW_OFF + ZROW_BYTES;
    static constexpr int CB2_OFF = ROWSTAT2_OFF + 256 * 8;          // c1 | bias of the tile's 256 columns (fp32)
    static constexpr int UPDW2_LDS = CB2_OFF + 2048;
    static constexpr int PLAINRS_OFF = LDS_BYTES;    // EPI_BIAS_BF16 with folded LayerNorm-3: the tile's 256 (mean, rstd) pairs
    static constexpr int PLAINCB_OFF = PLAINRS_OFF + 256 * 8;       // ... and c1 | bias of the tile's columns (fp32; bf16 operands, no conv)
    static constexpr int PLAINLN_LDS = PLAINCB_OFF + 2048;
    static constexpr int SCRATCH = 4608;             // per-wave epilogue scratch (8 x 4608 <= one stage)
    // EPI_QKV_ATTN (BN = 192): stage 0 keeps receiving the next tile's first K-step while the epilogue runs; the head's K, V^T and Q
    // images live behind it (stage 1, which a tile of an even number of K-steps consumes last, the LayerNorm side tables -- dead once the
    // accumulators are normalised -- and the rest of the CU's 160 KiB)
    static constexpr int AT_K = STAGE_BYTES, AT_V = AT_K + 256 * 128, AT_Q = AT_V + 64 * kAttnVPitch, ATTN_LDS = AT_Q + 256 * 128;
};

// F8 = true: both operands are OCP e4m3 bytes with MX block scales (one E8M0 byte per 32 K-elements), multiplied by
// v_mfma_scale_f32_32x32x64_f8f6f4 at twice the bf16 rate.  A 128-byte tile row is then 128 K-elements, so the DMA, the
// LDS image, its swizzle and every epilogue are those of the bf16 kernel; what changes is the fragment (32 bytes per
// lane: two 16-B reads), the MFMA, and a 1-KiB strip of scale bytes per operand and K-step that travels with the tile
// (scale layout [K/128][rows][4], see GemmParams).  Used for BASELINE config C4 (QKV / MLP GEMMs in fp8).
// CONV = true: implicit 3x3 convolution over a channels-last image (GemmParams::conv): only the A-side DMA addressing
// differs -- the per-lane row offsets are recomputed whenever the K loop moves to the next of the 9 taps.
// TN = true (launch_gemm_tn): both operands are given with the contraction index as the ROW (GemmParams::tn_ktotal) -- the weight gradients
// dW = dY^T X of the training step read dY and X as the forward / backward left them, no transposed copies.  A K-step's operand image is
// then [64 k-rows][256 columns] (512-byte rows, two k-rows per 1-KiB DMA piece) and the MFMA fragments -- 8 consecutive k of one column
// per lane -- come out of it with the transposing LDS read ds_read_b64_tr_b16 (two per fragment).  One wave-read touches 8 k-rows x 64
// bytes at the same column offset, so the 64-byte blocks of a row are XOR-swizzled with (row & 3) | (row bit 3) << 2 on the DMA's source side.
template <int BN, int EPI, bool F8 = false, bool CONV = false, bool RING = false, bool TN = false>
__global__ __launch_bounds__(512) void gemm256p_kernel(GemmParams p, int nblocks) {
    using G = G256P<BN>;
    static_assert(!RING || BN == 256, "the half-tile ring exists for 256 x 256 tiles (bf16, and since round 6 MX-fp8)");
    static_assert(!TN || (BN == 256 && EPI == EPI_F32 && !F8 && !CONV && !RING), "transposed operands: fp32 output, 256 x 256 tiles, two-stage loop");
    using frag_t = std::conditional_t<F8, i32x8, bf16x8>;
    constexpr int ESZ = F8 ? 1 : 2;                                     // operand bytes per element
    constexpr int SC_OFF = G::LDS_BYTES;                                // F8: [stage][A scales 1 KiB | W scales 1 KiB] behind the stages
    static_assert(!F8 || (EPI == EPI_F32 || EPI == EPI_QKV || EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_RESID), "fp8 epilogues");
    static_assert(!CONV || (!F8 && (EPI == EPI_F32 || EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_RESID)), "conv mode");
    constexpr bool PLAIN_CB = EPI == EPI_BIAS_BF16 && BN != 384 && !F8 && !CONV;       // bias / c1 of the tile's columns staged in LDS (see aux_dma)
    constexpr bool IS_QKV = EPI == EPI_QKV || EPI == EPI_QKV_LN, LN = EPI == EPI_QKV_LN || EPI == EPI_QKV_ATTN;
    constexpr bool UPDW = EPI == EPI_UP_DWCONV2 || EPI == EPI_UP_DWCONV32;       // fused depthwise epilogues (16 x 16 / 32 x 32 token grids)
    static_assert(EPI != EPI_QKV_ATTN || (BN == 192 && !F8 && !CONV && !RING), "the fused attention epilogue is written for 256 x 192 tiles");
    static_assert(EPI != EPI_QKV_ATTN || G::ATTN_LDS <= 160 * 1024, "LDS");
    static_assert(8 * G::SCRATCH <= G::STAGE_BYTES, "epilogue scratch must fit in one stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / G::WN, wn = wid % G::WN;
    const int l31 = lane & 31, hi = lane >> 5;

    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + G::BM - 1) / G::BM;
    // split-K (GemmParams::ksplit; fp32 slices, two-stage loop only): the tile list is `ksplit` copies of the (m, n) grid, copy s multiplying
    // the K range [s K, (s + 1) K) of both operands into output slice s
    constexpr bool KSPLIT_OK = EPI == EPI_F32 && !F8 && !CONV && !RING && !TN;
    const int nsplit = (KSPLIT_OK && p.ksplit > 1) ? p.ksplit : 1;
    const int ntiles = ntm * ntn * nsplit;
    int kb_dec = 0;                                                    // K byte offset of the tile tile_coords() decoded last
    // static schedule: XCD x (= block id % 8, where the dispatcher puts this block) owns a contiguous run of the
    // tile order; its workgroups take that run round-robin
    const int bid = blockIdx.x;
    const int xcd = bid & 7, lidx = bid >> 3;
    const int per_xcd_blocks = (nblocks + 7 - xcd) / 8;               // blocks with this xcd id
    // default: the tile order is row-major (n fastest) and XCD x owns a contiguous 1/8 of it.
    // xcd_ngroups = G > 1: the XCDs form an (8/G) x G grid; XCD (xm, xn) owns tile-rows block xm and the tile-column
    // group xn, walked row-major inside.  Its W working set shrinks to ntn/G tiles (resident in the 4 MB L2 across
    // rounds) at the price of every A tile being fetched by G XCDs.
    const int G2 = p.xcd_ngroups > 1 ? p.xcd_ngroups : 1;
    const int xn = xcd % G2, xm = xcd / G2, XM = 8 / G2;
    const int gcols = ntn / G2;                                        // tile-columns per group (ntn % G2 == 0)
    const int r0 = (int)((long)ntm * xm / XM), r1 = (int)((long)ntm * (xm + 1) / XM);
    const int q = ntiles >> 3, rr = ntiles & 7;
    const int xbase = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    const int xcount = G2 > 1 ? (r1 - r0) * gcols : q + (xcd < rr ? 1 : 0);
    int my_tiles = lidx < xcount ? (xcount - lidx + per_xcd_blocks - 1) / per_xcd_blocks : 0;
    // Half-tile tail (RING): when the XCD's last round has R left-over tiles for P workgroups and 2 R <= P, each left-over tile is computed
    // by TWO workgroups as their last item -- workgroup j the rows {wm 128 + [0, 64)} (quadrant row qa = 0 of every wave), workgroup R + j
    // the rows {wm 128 + 64 + [0, 64)} -- so every workgroup's time is q tiles + about 0.6 instead of q + 1 for some and q for the rest
    // (QKV at C1: 1152 tiles = 4.5 rounds).  No data moves between the two; each output element is accumulated exactly as in the whole
    // tile, so results stay bitwise independent of how the tiles were cut.  The half-tile's K loop keeps every phase, barrier and DMA of the
    // full one and only skips the MFMAs and A-fragment reads of the other quadrant row.
    constexpr bool HT_OK = RING && !CONV && (EPI == EPI_F32 || EPI == EPI_QKV || EPI == EPI_QKV_LN || EPI == EPI_BIAS_BF16);
    const int ht_q = xcount / per_xcd_blocks, ht_R = xcount - ht_q * per_xcd_blocks;
    const bool ht = HT_OK && p.half_tail && G2 == 1 && ht_R > 0 && 2 * ht_R <= per_xcd_blocks;
    const int ht_qa = lidx >= ht_R ? 1 : 0;
    if (ht) my_tiles = ht_q + (lidx < 2 * ht_R ? 1 : 0);
    auto item_tile = [&](int i) {                   // position of work item i in the XCD's run of tiles
        if (ht && i == ht_q) return ht_q * per_xcd_blocks + (lidx < ht_R ? lidx : lidx - ht_R);
        return lidx + i * per_xcd_blocks;
    };
    auto tile_coords = [&](int i, int& m0, int& n0) {
        // Row-major over tiles (n fastest), and XCD x owns a contiguous run of it: the ~32 workgroups of an XCD then
        // work on a few tile-rows x ALL tile-columns at a time, so an A tile is fetched into the XCD's L2 once and
        // serves every column while it is hot, and the W tiles are re-used by every round.  (The earlier order --
        // super-rows of 8 tile-rows, m fastest -- revisited each A super-row once per group of 4 columns, a full
        // round apart: PMC L2-miss traffic 381 MB vs 204 MB algorithmic on the QKV GEMM; this order: 134 -> 111 us.)
        const int t = item_tile(i);
        if (G2 > 1) {
            const int tm = t / gcols;
            m0 = (r0 + tm) * G::BM;
            n0 = (xn * gcols + (t - tm * gcols)) * BN;
        } else {
            int tile = xbase + t;
            if constexpr (KSPLIT_OK) {
                if (nsplit > 1) {
                    const int sp = tile / (ntm * ntn);
                    tile -= sp * (ntm * ntn);
                    kb_dec = sp * p.K * 2;
                }
            }
            const int tm = tile / ntn;
            m0 = tm * G::BM;
            n0 = (tile - tm * ntn) * BN;
        }
    };
    if (my_tiles == 0) return;

    const int nk = CONV ? 9 * (p.cv_cin >> 6) : p.K * ESZ / (G::BK * 2);   // 128-byte K-steps
    // DMA addressing: a uniform 64-bit base (operand + K offset, SGPRs) plus one 32-bit byte offset per piece and
    // lane (row clamp, row pitch and the source-side swizzle), recomputed once per tile -- per K-step and piece the
    // only VALU work is the load itself.  (Operands are < 4 GiB: checked at launch.)
    unsigned voffA[G::A_PIECES], voffB[G::B_PIECES];
    // CONV: output pixel (y << 16 | x) of every A row this lane brings in, and the first pixel of its sample in the source image
    unsigned cpix[G::A_PIECES], cbase[G::A_PIECES];
    size_t tnA = 0, tnW = 0;                                  // TN: byte offset of k-row 0 of the tile the DMA is working on (its split), per operand
    auto conv_tap = [&](int tap) {                            // A-row offsets of one of the nine taps (uniform tap)
        if constexpr (CONV) {
            const int ky = tap / 3;
            const int dy = ky - 1, dx = tap - ky * 3 - 1;
            const int ws_ = p.cv_w >> p.cv_up;
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int q2 = 0; q2 < G::A_PIECES; ++q2) {
                const int r = (wid * G::A_PIECES + q2) * 8 + (ln >> 3);
                const unsigned c16 = (unsigned)(((ln & 7) ^ ((r >> 1) & 7)) * 16);
                const int yy = (int)(cpix[q2] >> 16) + dy, xx = (int)(cpix[q2] & 0xffffu) + dx;
                const bool inb = (unsigned)yy < (unsigned)p.cv_h && (unsigned)xx < (unsigned)p.cv_w;
                const unsigned sp = cbase[q2] + (unsigned)((yy >> p.cv_up) * ws_ + (xx >> p.cv_up));
                unsigned v = inb ? p.cv_data_off + sp * (unsigned)(p.cv_cin * 2) + c16 : c16;     // outside: the zero page
                asm volatile("" : "+v"(v));
                voffA[q2] = v;
            }
        }
    };
    auto set_offsets = [&](int tm0, int tn0) {
        // (the empty asm statements keep the compiler from hoisting the lane-only sub-expressions out of the tile
        // loop as 64-bit loop invariants -- it then spilled them inside the K loop)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if constexpr (TN) {
            int split = 0, acol0 = tm0;
            if (p.w_batch_rows) { split = tm0 / p.w_batch_rows; acol0 = tm0 - split * p.w_batch_rows; }
            tnA = (size_t)split * p.K * p.lda * 2;
            tnW = (size_t)split * p.K * p.ldw * 2;
#pragma unroll
            for (int q2 = 0; q2 < G::A_PIECES; ++q2) {
                const int row = (wid * G::A_PIECES + q2) * 2 + (ln >> 5), c = ln & 31;
                const int csrc = ((((c >> 2) ^ ((row & 3) | (((row >> 3) & 1) << 2))) << 2) | (c & 3));
                unsigned v = __umul24((unsigned)row, (unsigned)(p.lda * 2)) + (unsigned)(acol0 * 2 + csrc * 16);
                asm volatile("" : "+v"(v));
                voffA[q2] = v;
                unsigned w2 = __umul24((unsigned)row, (unsigned)(p.ldw * 2)) + (unsigned)(tn0 * 2 + csrc * 16);
                asm volatile("" : "+v"(w2));
                voffB[q2] = w2;
            }
            return;
        }
        if constexpr (CONV) {
            const unsigned hw = (unsigned)(p.cv_h * p.cv_w);
            const unsigned shw = (unsigned)((p.cv_h >> p.cv_up) * (p.cv_w >> p.cv_up));
#pragma unroll
            for (int q2 = 0; q2 < G::A_PIECES; ++q2) {
                const int r = (wid * G::A_PIECES + q2) * 8 + (ln >> 3);
                int gr = tm0 + r;
                gr = gr < p.M ? gr : p.M - 1;
                const unsigned b = (unsigned)gr / hw, rem = (unsigned)gr - b * hw;
                const unsigned y = rem / (unsigned)p.cv_w, x = rem - y * (unsigned)p.cv_w;
                cpix[q2] = (y << 16) | x;
                cbase[q2] = b * shw;
            }
            conv_tap(0);
        } else {
#pragma unroll
        for (int q2 = 0; q2 < G::A_PIECES; ++q2) {
            const int r = (wid * G::A_PIECES + q2) * 8 + (ln >> 3);
            const int clog = (ln & 7) ^ ((r >> 1) & 7);
            int gr = tm0 + r;
            gr = gr < p.M ? gr : p.M - 1;
            unsigned v = __umul24((unsigned)gr, (unsigned)(p.lda * ESZ)) + (unsigned)(clog * 16);   // rows, pitch < 2^24
            asm volatile("" : "+v"(v));
            voffA[q2] = v;
        }
        }
        unsigned wgrp = 0;                                   // block-diagonal batching: this tile-row's own W matrix (uniform)
        if constexpr (!F8 && !CONV && (EPI == EPI_F32 || EPI == EPI_BIAS_BF16)) {
            if (p.w_batch_rows) wgrp = (unsigned)(tm0 / p.w_batch_rows) * p.w_batch_stride_bytes;
        }
#pragma unroll
        for (int q2 = 0; q2 < G::B_PIECES; ++q2) {
            const int r = (wid * G::B_PIECES + q2) * 8 + (ln >> 3);
            const int clog = (ln & 7) ^ ((r >> 1) & 7);
            int gr = tn0 + r;
            gr = gr < p.N ? gr : p.N - 1;
            unsigned v = __umul24((unsigned)gr, (unsigned)(p.ldw * ESZ)) + (unsigned)(clog * 16) + wgrp;
            asm volatile("" : "+v"(v));
            voffB[q2] = v;
        }
    };
    int conv_akb = 0;                                         // CONV: byte offset of the K-step's 64-channel block inside a pixel (A side)
    auto dma_piece = [&](int q2, int kbyte, char* st) {       // q2 < A_PIECES: A piece, else W piece; kbyte uniform
        if constexpr (TN) {                                   // kbyte / 128 = K-step: 64 k-rows further down both operands
            const bool isA = q2 < G::A_PIECES;
            const char* base = isA ? reinterpret_cast<const char*>(p.A) + tnA + (size_t)(kbyte >> 7) * 64 * p.lda * 2
                                   : reinterpret_cast<const char*>(p.W) + tnW + (size_t)(kbyte >> 7) * 64 * p.ldw * 2;
            const unsigned vo = isA ? voffA[q2] : voffB[q2 - G::A_PIECES];
            __builtin_amdgcn_global_load_lds((gptr_t)(base + vo), (lptr_t)(st + (isA ? 0 : G::A_BYTES) + (wid * G::A_PIECES + (isA ? q2 : q2 - G::A_PIECES)) * 1024), 16, 0, 0);
            return;
        }
        if (q2 < G::A_PIECES) {
            const char* base = reinterpret_cast<const char*>(p.A) + (CONV ? conv_akb : kbyte);
            __builtin_amdgcn_global_load_lds((gptr_t)(base + voffA[q2]), (lptr_t)(st + (wid * G::A_PIECES + q2) * 1024), 16, 0, 0);
        } else {
            const char* base = reinterpret_cast<const char*>(p.W) + kbyte;
            __builtin_amdgcn_global_load_lds((gptr_t)(base + voffB[q2 - G::A_PIECES]),
                                             (lptr_t)(st + G::A_BYTES + (wid * G::B_PIECES + (q2 - G::A_PIECES)) * 1024), 16, 0, 0);
        }
    };
    // F8: the block scales of one K-step -- 4 bytes per row, [K/128][rows][4] in memory, so a tile's strip is one
    // contiguous KiB (16 B = 4 rows per lane) -- wave 0 brings A's, wave 1 W's
    auto dma_scales = [&](int tm0, int tn0, int kstep, int stage) {
        if constexpr (F8) {
            if (wid < 2) {
                const int rows = wid == 0 ? p.M : p.N;
                int r = (wid == 0 ? tm0 : tn0) + lane * 4;
                r = r + 3 < rows ? r : rows - 4;                       // (rows % 4 == 0: lanes of valid rows never clamp)
                const uint8_t* src = (wid == 0 ? p.a_scale : p.w_scale) + ((size_t)kstep * rows + r) * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + SC_OFF + stage * 2048 + wid * 1024), 16, 0, 0);
            }
        }
    };
    auto issue = [&](int m0, int n0, int g) {                 // K-step 0 of a tile, all pieces at once
        char* st = smem + (g & 1) * G::STAGE_BYTES;
        set_offsets(m0, n0);
#pragma unroll
        for (int q2 = 0; q2 < G::A_PIECES + G::B_PIECES; ++q2) dma_piece(q2, KSPLIT_OK ? kb_dec : 0, st);     // (called right after tile_coords of that tile)
        dma_scales(m0, n0, 0, g & 1);
    };
    // ks: 16-element k-slice (bf16: 4 per K-step) / 64-element k-slice (fp8: 2 per K-step)
    // TN: per-lane byte offsets of the (tile i / j, k-slice 0) fragment inside a stage: k-row 8 hi + (s >> 2), 64-byte block (column block ^ row swizzle)
    unsigned tnoA[TN ? G::TM : 1], tnoB[TN ? G::TN : 1];
    if constexpr (TN) {
        const int s16 = lane & 15, g1 = (lane >> 4) & 1, rowl = 8 * hi + (s16 >> 2), sw = (s16 >> 2) | (hi << 2);
#pragma unroll
        for (int i = 0; i < G::TM; ++i) tnoA[i] = (unsigned)(rowl * 512 + (((wm * G::TM + i) ^ sw) << 6) + 32 * g1 + 8 * (s16 & 3));
#pragma unroll
        for (int j = 0; j < G::TN; ++j) tnoB[j] = (unsigned)(G::A_BYTES + rowl * 512 + (((wn * G::TN + j) ^ sw) << 6) + 32 * g1 + 8 * (s16 & 3));
    }
    auto load_frags = [&](const char* st, int ks, frag_t (&a)[G::TM], frag_t (&b)[G::TN]) {
        if constexpr (TN) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            auto rd = [&](unsigned off) {
                const char* q0 = st + off + ks * 8192;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q0));
                const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q0 + 2048));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                return __builtin_bit_cast(bf16x8, v);
            };
#pragma unroll
            for (int i = 0; i < G::TM; ++i) a[i] = rd(tnoA[i]);
#pragma unroll
            for (int j = 0; j < G::TN; ++j) b[j] = rd(tnoB[j]);
        } else if constexpr (!F8) {
            const int kc = ks * 2 + hi;
#pragma unroll
            for (int i = 0; i < G::TM; ++i) a[i] = read_frag(st, wm * G::WROWS + i * 32 + l31, kc);
#pragma unroll
            for (int j = 0; j < G::TN; ++j) b[j] = read_frag(st + G::A_BYTES, wn * G::WCOLS + j * 32 + l31, kc);
        } else {
            // Operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (probed on gfx950, tools/ubench/mx_probe.hip): lane
            // (row l31, half hi) holds K-elements 16 hi .. 16 hi + 15 in its lower four registers and 32 + 16 hi .. + 15 in
            // its upper four; the scale byte of lane (r, 0) applies to k 0-31 of row r, that of lane (r, 1) to k 32-63.
            // In 16-B chunks of the 64-element k-slice ks: chunks hi and 2 + hi.
            const int c = ks * 4 + hi;
            auto rd = [&](const char* base, int row) {
                const int sw = (row >> 1) & 7;
                const u32x4 lo = *reinterpret_cast<const u32x4*>(base + row * 128 + ((c ^ sw) << 4));
                const u32x4 hi4 = *reinterpret_cast<const u32x4*>(base + row * 128 + (((c + 2) ^ sw) << 4));
                return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
            };
#pragma unroll
            for (int i = 0; i < G::TM; ++i) a[i] = rd(st, wm * G::WROWS + i * 32 + l31);
#pragma unroll
            for (int j = 0; j < G::TN; ++j) b[j] = rd(st + G::A_BYTES, wn * G::WCOLS + j * 32 + l31);
        }
    };

    // ---- half-tile ring (RING): see kloop_ring below.  A half-tile is 128 operand rows x 64 K (16 KiB, same row image and swizzle as
    // a stage's), brought in by two 1-KiB pieces per wave.  Half-tile A_h holds the rows {wm * 128 + h * 64 + [0, 64)} of the tile and
    // B_h the W rows {wn * 64 + h * 32 + [0, 32)}: quadrant (qa, qb) of every wave's 128 x 64 output reads exactly A_qa and B_qb.
    constexpr int HT = 16384;
    unsigned rvA[RING ? 4 : 1], rvB[RING ? 4 : 1];            // [half * 2 + piece]: per-lane source byte offsets
    // RING + CONV: output pixel (y << 16 | x) and first source pixel of the sample for the four A rows this lane brings in per K-tile; the
    // A offsets are rebuilt from them whenever the stream moves on to the next of the nine taps (ring_conv_tap)
    unsigned rcpix[RING && CONV ? 4 : 1], rcbase[RING && CONV ? 4 : 1];
    int sc_m0 = 0, sc_n0 = 0;                                 // RING + F8: tile whose half-tiles are being staged (block-scale strips travel with B0)
    auto ring_conv_tap = [&](int tap) {
        if constexpr (RING && CONV) {
            const int ky = tap / 3;
            const int dy = ky - 1, dx = tap - ky * 3 - 1;
            const int ws_ = p.cv_w >> p.cv_up;
            unsigned full = ~0u;
            asm volatile("" : "+s"(full));
            int ln = (int)__builtin_amdgcn_mbcnt_hi(full, __builtin_amdgcn_mbcnt_lo(full, 0u));
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int r = (wid * 2 + q2) * 8 + (ln >> 3);
                    const unsigned c16 = (unsigned)(((ln & 7) ^ ((r >> 1) & 7)) * 16);
                    const int yy = (int)(rcpix[h * 2 + q2] >> 16) + dy, xx = (int)(rcpix[h * 2 + q2] & 0xffffu) + dx;
                    const bool inb = (unsigned)yy < (unsigned)p.cv_h && (unsigned)xx < (unsigned)p.cv_w;
                    const unsigned sp = rcbase[h * 2 + q2] + (unsigned)((yy >> p.cv_up) * ws_ + (xx >> p.cv_up));
                    unsigned v = inb ? p.cv_data_off + sp * (unsigned)(p.cv_cin * 2) + c16 : c16;     // outside: the zero page
                    asm volatile("" : "+v"(v));
                    rvA[h * 2 + q2] = v;
                }
        }
    };
    auto ring_offsets = [&](int tm0, int tn0) {
        if constexpr (RING && CONV) {
            unsigned full = ~0u;
            asm volatile("" : "+s"(full));
            int ln = (int)__builtin_amdgcn_mbcnt_hi(full, __builtin_amdgcn_mbcnt_lo(full, 0u));
            asm volatile("" : "+v"(ln));
            const unsigned hw = (unsigned)(p.cv_h * p.cv_w);
            const unsigned shw = (unsigned)((p.cv_h >> p.cv_up) * (p.cv_w >> p.cv_up));
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int r = (wid * 2 + q2) * 8 + (ln >> 3);
                    const unsigned c16 = (unsigned)(((ln & 7) ^ ((r >> 1) & 7)) * 16);
                    int ga = tm0 + (r >> 6) * 128 + h * 64 + (r & 63);
                    ga = ga < p.M ? ga : p.M - 1;
                    const unsigned bb = (unsigned)ga / hw, rem = (unsigned)ga - bb * hw;
                    const unsigned y = rem / (unsigned)p.cv_w, x = rem - y * (unsigned)p.cv_w;
                    rcpix[h * 2 + q2] = (y << 16) | x;
                    rcbase[h * 2 + q2] = bb * shw;
                    int gb = tn0 + (r >> 5) * 64 + h * 32 + (r & 31);
                    gb = gb < p.N ? gb : p.N - 1;
                    unsigned vb = __umul24((unsigned)gb, (unsigned)(p.ldw * 2)) + c16;
                    asm volatile("" : "+v"(vb));
                    rvB[h * 2 + q2] = vb;
                }
            ring_conv_tap(0);
        } else
        if constexpr (RING) {
            sc_m0 = tm0; sc_n0 = tn0;
            int ln = lane;
            asm volatile("" : "+v"(ln));
            unsigned wgrp = 0;
            if constexpr (!F8 && (EPI == EPI_F32 || EPI == EPI_BIAS_BF16)) {
                if (p.w_batch_rows) wgrp = (unsigned)(tm0 / p.w_batch_rows) * p.w_batch_stride_bytes;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int r = (wid * 2 + q2) * 8 + (ln >> 3);                      // row of the half-tile image
                    const unsigned c16 = (unsigned)(((ln & 7) ^ ((r >> 1) & 7)) * 16);
                    int ga = tm0 + (r >> 6) * 128 + h * 64 + (r & 63);
                    ga = ga < p.M ? ga : p.M - 1;
                    unsigned va = __umul24((unsigned)ga, (unsigned)(p.lda * ESZ)) + c16;
                    int gb = tn0 + (r >> 5) * 64 + h * 32 + (r & 31);
                    gb = gb < p.N ? gb : p.N - 1;
                    unsigned vb = __umul24((unsigned)gb, (unsigned)(p.ldw * ESZ)) + c16 + wgrp;
                    asm volatile("" : "+v"(va), "+v"(vb));
                    rvA[h * 2 + q2] = va;
                    rvB[h * 2 + q2] = vb;
                }
        }
    };
    // stream element i of K-tile tk: 0 = B0, 1 = A0, 2 = B1, 3 = A1 (the order in which a K-tile's half-tiles are first read)
    auto ring_stage = [&](auto ic, int tk, int slot) {
        if constexpr (RING) {
            constexpr int i = decltype(ic)::value;
            constexpr bool isA = (i & 1) != 0;
            constexpr int half = i >> 1;
            size_t koff = (size_t)tk * (G::BK * 2);
            if constexpr (CONV && isA) {        // K-tile tk = (tap, 64-channel block): the A side addresses a 128-byte channel slice of a shifted pixel
                const int ckpt = p.cv_cin >> 6, tap = tk / ckpt, cb = tk - tap * ckpt;
                if (i == 1 && cb == 0 && tk > 0) ring_conv_tap(tap);      // A0 is the first A element of a K-tile: new tap, new offsets (tap 0: ring_offsets)
                koff = (size_t)cb * (G::BK * 2);
            }
            const char* base = reinterpret_cast<const char*>(isA ? (const void*)p.A : (const void*)p.W) + koff;
            asm volatile("" : "+s"(base));         // one SGPR pair + a 32-bit lane offset per piece (no 64-bit vector address arithmetic)
            char* dst = smem + slot * HT + wid * 2048;
            unsigned o0 = isA ? rvA[half * 2] : rvB[half * 2], o1 = isA ? rvA[half * 2 + 1] : rvB[half * 2 + 1];
            asm volatile("" : "+v"(o0), "+v"(o1));   // (keeps the zero-extension next to the load: saddr + 32-bit voffset form)
            __builtin_amdgcn_global_load_lds((gptr_t)(base + o0), (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(base + o1), (lptr_t)(dst + 1024), 16, 0, 0);
            if constexpr (F8 && i == 0) {
                // MX block scales of K-tile tk (4 bytes per row and K-tile, [K/128][rows][4] in memory: a tile's strip is one contiguous KiB per operand) travel
                // with the K-tile's FIRST stream element, one quarter strip (64 rows = 16 lanes x 16 B) per wave -- waves 0-3 the A strip, 4-7 the W strip -- so that
                // every wave issues the same number of VMEM operations per K-tile (the counted waits below: three half-tiles + this piece in flight = vmcnt(7))
                const bool sa = wid < 4;
                const int rows = sa ? p.M : p.N;
                int r = (sa ? sc_m0 : sc_n0) + (wid & 3) * 64 + (lane & 15) * 4;
                r = r + 3 < rows ? r : rows - 4;                       // (rows % 4 == 0: lanes of valid rows never clamp)
                const uint8_t* src = (sa ? p.a_scale : p.w_scale) + ((size_t)tk * rows + r) * 4;
                char* sdst = smem + SC_OFF + (slot >> 2) * 2048 + (sa ? 0 : 1024) + (wid & 3) * 256;
                if (lane < 16) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)sdst, 16, 0, 0);
            }
        }
    };

    int m0, n0;
    tile_coords(0, m0, n0);
    int kb0 = kb_dec, kbn = 0;                        // split-K: K byte offset of the current / next tile (0 without splits)
    if constexpr (!RING) issue(m0, n0, 0);
    int g = 0;                                        // global K-step counter (ring position)
    int ctap = 0, ccb = 0;                            // CONV: (tap, 64-channel block) of the K-step whose DMA was issued last
    const int kpt = CONV ? (p.cv_cin >> 6) : 1;       // K-steps per tap
    for (int it = 0; it < my_tiles; ++it) {
        int m0n = 0, n0n = 0;
        const bool has_next = it + 1 < my_tiles;
        if (has_next) { tile_coords(it + 1, m0n, n0n); kbn = kb_dec; }
        // MFMA operand order: swapped (D^T = W A^T: a lane owns a token row and 4 consecutive columns per register quad)
        // everywhere except the fp32 debug epilogue.  The V^T tiles of the QKV GEMM are swapped too and transposed on
        // their way through the epilogue scratch with 2-byte LDS writes: one K-loop instantiation instead of two took
        // the kernel from 245 to 213 VGPRs and 111 -> 110 us.
        constexpr bool swapped = EPI != EPI_F32 && !UPDW;   // (the pair image wants lane = channel)
        bool v_tile = false;
        if constexpr (IS_QKV) v_tile = n0 >= 2 * p.d;

        int nkt = nk;                                 // K-steps of this tile
        if constexpr (TN) {
            const int split = p.w_batch_rows ? m0 / p.w_batch_rows : 0;
            const int rows = p.tn_ktotal - split * p.K;
            nkt = (rows < p.K ? rows : p.K) >> 6;
        }
        const bool hmode = ht && it == ht_q;          // this item is one row half (ht_qa) of a left-over tile
        f32x16 acc[G::TM][G::TN];
#pragma unroll
        for (int i = 0; i < G::TM; ++i)
#pragma unroll
            for (int j = 0; j < G::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        frag_t a0[G::TM], b0[G::TN];
        int sca[G::TM], scb[G::TN];                 // F8: this K-step's scale dwords of the lane's rows (already shifted by 8 hi)
        // per-tile side tables of the folded LayerNorms (row partial sums / (mean, rstd) pairs, column constants): DMA into LDS behind
        // the operand ring; issued in K-step 1 of the two-stage loop and at the tile start of the ring loop (every wave is then past
        // the previous tile's epilogue, which read them)
        auto aux_dma = [&]() {
            if constexpr (LN) {
                static_assert(kLnSlots == 8, "one partial-sum row is 64 B");
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int piece = wid * 2 + q2;
                    int row = m0 + piece * 16 + (lane >> 2);
                    row = row < p.M ? row : p.M - 1;
                    const char* src = reinterpret_cast<const char*>(p.ln_stats) + (size_t)row * 64 + (lane & 3) * 16;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + G::LN_RAW + piece * 1024), 16, 0, 0);
                }
                if (wid < 2) {                                    // wave 0: c1, wave 1: b1 of the tile's columns (4 per lane)
                    int col = n0 + lane * 4;
                    col = col < p.N ? col : 0;
                    const float* src = (wid == 0 ? p.ln_c1 : p.ln_b1) + col;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + G::LN_CB + wid * 1024), 16, 0, 0);
                }
            }
            if constexpr (EPI == EPI_BIAS_BF16) {      // folded LayerNorm-3 of the plain up-projection (grids other than 16 x 16)
                if (p.row_stats && wid < 2) {
                    int r0s = m0 + wid * 128 + lane * 2;                     // 2 rows (16 B) per lane, clamped at the matrix end
                    r0s = r0s + 1 < p.M ? r0s : (p.M >= 2 ? p.M - 2 : 0);
                    const char* src = reinterpret_cast<const char*>(p.row_stats + r0s);
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + G::PLAINRS_OFF + wid * 1024), 16, 0, 0);
                }
                // the tile's bias (and LayerNorm-3 column sums) by DMA as well: as global loads inside the store loop they queued behind the previous
                // 32-row slab's stores (vmcnt retires in order), i.e. every slab waited for the stores of the one before
                if constexpr (PLAIN_CB) {
                    if (wid == 3 || (wid == 2 && p.row_stats)) {
                        int col = n0 + lane * 4;
                        col = col < p.N ? col : 0;
                        const float* src = (wid == 2 ? p.ln_c1 : p.bias) + col;
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + G::PLAINCB_OFF + (wid - 2) * 1024), 16, 0, 0);
                    }
                }
            }
            if constexpr (UPDW) {
                constexpr int RS_OFF = G::ROWSTAT2_OFF;
                if (p.row_stats && wid < 2) {
                    const char* src = reinterpret_cast<const char*>(p.row_stats + m0) + wid * 1024 + lane * 16;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + RS_OFF + wid * 1024), 16, 0, 0);
                }
                if (wid == 3 || (wid == 2 && p.row_stats)) {      // waves 2 / 3: c1 / bias of the tile's columns
                    const float* src = (wid == 2 ? p.ln_c1 : p.bias) + n0 + lane * 4;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + G::CB2_OFF + (wid - 2) * 1024), 16, 0, 0);
                }
            }
        };
        auto kloop = [&](auto swp) {
            constexpr bool SW = decltype(swp)::value;
            auto mma = [&](const frag_t (&a)[G::TM], const frag_t (&b)[G::TN], auto slice) {
                constexpr int S2 = decltype(slice)::value * 2;      // F8: scale byte of k-slice s is byte 2 s (+ hi, shifted in)
#pragma unroll
                for (int i = 0; i < G::TM; ++i)
#pragma unroll
                    for (int j = 0; j < G::TN; ++j) {
                        if constexpr (F8) {
                            if constexpr (SW) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b[j], a[i], acc[i][j], 0, 0, S2, scb[j], S2, sca[i]);
                            else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 0, 0, S2, sca[i], S2, scb[j]);
                        } else {
                            if constexpr (SW) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                        }
                    }
            };
            // The DMA pieces of the next K-step are spread over the four MFMA groups of this step (a burst of
            // 8 pieces right after the barrier kept both waves of a SIMD out of the MFMA pipe for ~800 cycles),
            // and the last MFMA group of a step is executed AFTER the next step's barrier, so the pipe has
            // register-resident work while the first fragments of the new tile are read from LDS.
            constexpr int NP = G::A_PIECES + G::B_PIECES;           // pieces per wave per K-step
            {
            // Staggered form (the 8-phase idea of the gfx950 GEMM template): the two waves of every SIMD are w and w + 4;
            // waves 4-7 run ONE barrier interval behind waves 0-3, so in every interval one wave of each SIMD executes
            // its 8 (12) MFMAs of a k-slice from registers at raised priority while its partner reads the next slice's
            // fragments from LDS and issues the tile DMA -- the matrix pipe never waits for an LDS read and the two
            // waves never compete for it.  Intervals alternate R (fragment reads + tile DMA) and M (MFMAs).
            //   hazards: fragments are waited for (lgkmcnt 0) BEFORE the barrier closing an R interval, so a stage is
            //   free for DMA as soon as that barrier is passed; every wave waits for its own DMA pieces (vmcnt 0) before
            //   the last barrier of a K-step, which both groups pass before anyone reads the next stage.
            const int grp = wid >> 2;
            wait_vmcnt<0>();                       // first K-step of the tile landed (own pieces) ...  (skipping this wait for tiles whose first K-step was waited for in the previous tile's last step -- it then only covers that tile's output stores -- measured neutral: kept)
            __builtin_amdgcn_s_barrier();          // ... everybody's; also: the previous epilogue's scratch reads are done
            if (grp) __builtin_amdgcn_s_barrier(); // stagger in
            for (int k = 0; k < nkt; ++k, ++g) {
                if (k == (nkt > 1 ? 1 : 0)) aux_dma();        // (K = 64: the tile's only K-step; every wave is past the previous epilogue since the barrier above)
                const char* st = smem + (g & 1) * G::STAGE_BYTES;
                char* nst = smem + ((g + 1) & 1) * G::STAGE_BYTES;
                const bool more = (k + 1 < nkt) || (has_next && !UPDW);
                const int pkb = (k + 1 < nkt) ? (KSPLIT_OK ? kb0 : 0) + (k + 1) * G::BK * 2 : (KSPLIT_OK ? kbn : 0);
                if constexpr (CONV) {
                    if (k + 1 < nkt) {
                        if (++ccb == kpt) { ccb = 0; ++ctap; conv_tap(ctap); }
                    } else { ccb = 0; ctap = 0; }                   // (set_offsets below starts the next tile at tap 0)
                    conv_akb = ccb * (G::BK * 2);
                }
                if (k + 1 == nkt && more) set_offsets(m0n, n0n);      // this step's DMA targets the next tile
                // one k-slice per interval: 8 intervals (barriers) per bf16 K-step with 6 / 8 / 12 MFMAs each (two k-slices per interval -- 4
                // barriers, two fragment sets -- measured slower); the tile DMA of the next K-step rides in the first two R intervals
                // (one: -1.7 %, three: +-0.3 %); fragments are waited for AFTER the barrier (the latency overlaps the barrier wait) except in the
                // step's last interval, whose barrier frees the stage for DMA
                constexpr int NI = F8 ? 2 : 4;
                constexpr int NDMA = (NI == 4) ? 2 : 1;                       // intervals that carry tile DMA
#pragma unroll
                for (int h = 0; h < NI; ++h) {
                    // ---- R interval
                    load_frags(st, h, a0, b0);
                    if constexpr (F8) {
                        if (h == 0) {       // this K-step's block scales: 4 bytes per row; half hi uses bytes hi and 2 + hi
                            const char* sc = smem + SC_OFF + (g & 1) * 2048;
#pragma unroll
                            for (int i = 0; i < G::TM; ++i) sca[i] = *reinterpret_cast<const int*>(sc + (wm * G::WROWS + i * 32 + l31) * 4) >> (8 * hi);
#pragma unroll
                            for (int j = 0; j < G::TN; ++j) scb[j] = *reinterpret_cast<const int*>(sc + 1024 + (wn * G::WCOLS + j * 32 + l31) * 4) >> (8 * hi);
                        }
                    }
                    // tile DMA of the next K-step: early in the step, so that every piece has >= 2 intervals to land before
                    // the vmcnt(0) of the step's last interval
                    if (h < NDMA && more) {
                        constexpr int PER = (NP + NDMA - 1) / NDMA;
#pragma unroll
                        for (int q2 = 0; q2 < NP; ++q2) {
                            if (q2 < h * PER || q2 >= (h + 1) * PER) continue;
                            dma_piece(q2, pkb, nst);
                        }
                        if (h == 0) {
                            if (k + 1 < nkt) dma_scales(m0, n0, k + 1, (g + 1) & 1);
                            else dma_scales(m0n, n0n, 0, (g + 1) & 1);
                        }
                    }
                    if (h == NI - 1) wait_vmcnt<0>();
                    if (h == NI - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- M interval
                    __builtin_amdgcn_s_setprio(1);
                    if constexpr (F8) {
                        if (h == 0) mma(a0, b0, std::integral_constant<int, 0>{}); else mma(a0, b0, std::integral_constant<int, 1>{});
                    } else {
                        mma(a0, b0, std::integral_constant<int, 0>{});
                    }
                    __builtin_amdgcn_s_setprio(0);
                    if (h == NI - 1) wait_vmcnt<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!grp) __builtin_amdgcn_s_barrier();        // stagger out: both groups are past their last MFMA
            }
        };
        // ---- Half-tile ring K loop (RING; the counted-vmcnt 8-phase structure of the gfx950 GEMM template, on 32x32x16 MFMAs).
        //   LDS: 8 half-tile slots = 2 K-tiles x {B0, A0, B1, A1}.  An iteration is 8 phases = 2 K-tiles; phase p:
        //     R half: ds_read the quadrant's operand subtile | stage ONE half-tile of the stream (2 pieces per wave)
        //     s_barrier, lgkmcnt(0), 8 MFMAs (one C quadrant x K = 64) at raised priority, s_barrier.
        //   Quadrants per K-tile: (0,0) (0,1) (1,1) (1,0); reads: B0 + A0 | B1 | A1 | none (B0 stays in registers).  Waves 4-7 run one
        //   barrier behind waves 0-3, so the two waves of a SIMD alternate between the R and the M half.
        //   The half-tile stream runs 6 phases ahead of its first read (phase P stages element P + 6).  s_waitcnt vmcnt(6) in phases 4
        //   and 8 only: three half-tiles stay in flight, never 0 inside a tile's main loop; a half-tile is first read in the phase AFTER
        //   the wait that retired it.  WAR: a slot is restaged >= 2 phases after its last read, except B0 (1 phase), whose reads are
        //   issued first and retired by lgkmcnt(8) before the reading phase's first barrier.
        //   Tile boundary: the next tile's K-tile 0 is staged in phases 2-5 of the last iteration (slots 0-3) and lands under the
        //   epilogue, whose scratch is slots 4-7; its K-tile 1 elements B0 A0 B1 are staged right after the epilogue.  The fused
        //   depthwise epilogues own the whole ring (the image), so their tiles start cold.
        auto kloop_ring = [&](auto swp) {
            if constexpr (RING) {
            constexpr bool SW = decltype(swp)::value;
            constexpr bool NOXT = UPDW;
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
            const int grp = wid >> 2;
            const int nj = nk >> 1;
            const bool xt = has_next && !NOXT;
            // MX-fp8 (round 6): the same ring -- a 128-byte row is 128 K-elements, so half-tiles, slots, swizzle and the stream are unchanged; a k-slice of 64 elements is
            // TWO of the 16-byte pieces the bf16 loop reads per (row, k-slice) (pieces 2 s and 2 s + 1 = the operand's lower / upper four registers), a phase is 4
            // v_mfma_scale_f32_32x32x64_f8f6f4 instead of 8 bf16 MFMAs (same 256 matrix cycles), and one quarter strip of block scales per wave rides with every
            // K-tile's B0 (ring_stage): one more VMEM operation in flight at the counted waits.
            constexpr int RW = F8 ? 7 : 6;
            if (it == 0 || NOXT) {
                aux_dma();
                ring_offsets(m0, n0);
                ring_stage(I0{}, 0, 0); ring_stage(I1{}, 0, 1); ring_stage(I2{}, 0, 2); ring_stage(I3{}, 0, 3);
                ring_stage(I0{}, 1, 4); ring_stage(I1{}, 1, 5); ring_stage(I2{}, 1, 6);
                wait_vmcnt<RW>();                  // K-tile 0 (and the side tables) landed: own pieces ...
                __builtin_amdgcn_s_barrier();      // ... everybody's
            } else {
                __builtin_amdgcn_s_barrier();      // every wave is done with its epilogue scratch (slots 4-7)
                aux_dma();
                ring_stage(I0{}, 1, 4); ring_stage(I1{}, 1, 5); ring_stage(I2{}, 1, 6);
            }
            using piece_t = std::conditional_t<F8, u32x4, bf16x8>;      // one ds_read_b128 of a fragment row
            piece_t fa[2][4], fb0[4], fb1[4];
            int rsa[2][2], rsb[2];                                       // F8: this K-tile's raw scale dwords of the lane's rows (A: [quadrant row][32-row tile], B: [quadrant column])
            // fragment read addresses: one register per (operand, k-slice) -- byte offset of (row, chunk (2 ks + hi) ^ swizzle) inside a
            // half-tile image; the slot and the A row-block are immediates, and the registers flip between the two K-tiles' slot groups
            // (bit 16) as the loop goes: no per-read address arithmetic
            unsigned ra[4], rb[4];
            {
                int l31v = l31, hiv = hi;
                asm volatile("" : "+v"(l31v), "+v"(hiv));
                const int sw = (l31v >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    ra[ks] = (unsigned)((wm * 64 + l31v) * 128 + (((ks * 2 + hiv) ^ sw) << 4));
                    rb[ks] = (unsigned)((wn * 32 + l31v) * 128 + (((ks * 2 + hiv) ^ sw) << 4));
                    asm volatile("" : "+v"(ra[ks]), "+v"(rb[ks]));
                }
            }
            auto rd = [&](unsigned a, int imm) { return *reinterpret_cast<const piece_t*>(smem + a + imm); };
            if (grp) __builtin_amdgcn_s_barrier(); // stagger in
            auto iter = [&](int j, auto lastc) {
                constexpr bool last = decltype(lastc)::value;
                auto phase = [&](auto pc) {
                    constexpr int ph = decltype(pc)::value;          // 1 .. 8
                    constexpr int q = (ph - 1) & 3;
                    // ---- R half   (slots of this K-tile: B0 A0 B1 A1 at 0, HT, 2 HT, 3 HT from the current slot group)
                    if constexpr (q == 0) {
                        if constexpr (F8) {        // issued BEFORE the B0 reads: retired with them (lgkmcnt(8) below), so the strip is free when B0's slot is
                            const char* sc = smem + SC_OFF + ((ph - 1) >> 2) * 2048;
#pragma unroll
                            for (int qa2 = 0; qa2 < 2; ++qa2)
#pragma unroll
                                for (int ii = 0; ii < 2; ++ii) rsa[qa2][ii] = *reinterpret_cast<const int*>(sc + (wm * 128 + qa2 * 64 + ii * 32 + l31) * 4);
#pragma unroll
                            for (int qb2 = 0; qb2 < 2; ++qb2) rsb[qb2] = *reinterpret_cast<const int*>(sc + 1024 + (wn * 64 + qb2 * 32 + l31) * 4);
                        }
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) fb0[ks] = rd(rb[ks], 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (!(hmode && ht_qa == 1)) {
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) fa[ii][ks] = rd(ra[ks], HT + ii * 4096);
                        }
                    } else if constexpr (q == 1) {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) fb1[ks] = rd(rb[ks], 2 * HT);
                    } else if constexpr (q == 2) {
                        if (!(hmode && ht_qa == 0)) {
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) fa[ii][ks] = rd(ra[ks], 3 * HT + ii * 4096);
                        }
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(rb[ks]));     // B reads of this K-tile are issued: other slot group
                    } else {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(ra[ks]));
                    }
                    {
                        constexpr int i = (ph + 6) & 3, slot = (ph + 6) & 7, dk = (ph + 6) >> 2;
                        using IC = std::integral_constant<int, i>;
                        if constexpr (ph == 1 || !last) ring_stage(IC{}, 2 * j + dk, slot);
                        else if constexpr (ph <= 5) {            // last iteration: the next tile's K-tile 0
                            if (xt) {
                                if constexpr (ph == 2) ring_offsets(m0n, n0n);
                                ring_stage(IC{}, 0, slot);
                            }
                        }
                    }
                    if constexpr (ph == 4) {
                        if (last && !xt) wait_vmcnt<0>(); else wait_vmcnt<RW>();
                    }
                    if constexpr (ph == 8 && !last) wait_vmcnt<RW>();
                    if constexpr (q == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // B0 reads retired: its slot is restaged next phase
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- M half: one C quadrant x K = 64
                    constexpr int qa = (q >= 2) ? 1 : 0, qb = (q == 1 || q == 2) ? 1 : 0;
                    if (!(hmode && ht_qa != qa)) {        // (half-tile item: the other quadrant row's MFMAs are not this workgroup's)
                    if constexpr (F8) {
                        // half hi of a row uses scale bytes hi (k 0-31 of slice 0 ... see load_frags) and 2 + hi: shift once, op_sel picks byte 0 / 2 per k-slice
                        const int sb = (qb ? rsb[1] : rsb[0]) >> (8 * hi);
                        const int sa0 = rsa[qa][0] >> (8 * hi), sa1 = rsa[qa][1] >> (8 * hi);
                        auto cat = [](const piece_t& lo, const piece_t& hi4) {
                            return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
                        };
                        __builtin_amdgcn_s_setprio(1);
#pragma unroll
                        for (int sl = 0; sl < 2; ++sl) {
                            const i32x8 bo = qb ? cat(fb1[2 * sl], fb1[2 * sl + 1]) : cat(fb0[2 * sl], fb0[2 * sl + 1]);
#pragma unroll
                            for (int ii = 0; ii < 2; ++ii) {
                                const i32x8 ao = cat(fa[ii][2 * sl], fa[ii][2 * sl + 1]);
                                const int sa = ii ? sa1 : sa0;
                                if (sl == 0) {
                                    if constexpr (SW) acc[qa * 2 + ii][qb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bo, ao, acc[qa * 2 + ii][qb], 0, 0, 0, sb, 0, sa);
                                    else acc[qa * 2 + ii][qb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ao, bo, acc[qa * 2 + ii][qb], 0, 0, 0, sa, 0, sb);
                                } else {
                                    if constexpr (SW) acc[qa * 2 + ii][qb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bo, ao, acc[qa * 2 + ii][qb], 0, 0, 2, sb, 2, sa);
                                    else acc[qa * 2 + ii][qb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ao, bo, acc[qa * 2 + ii][qb], 0, 0, 2, sa, 2, sb);
                                }
                            }
                        }
                        __builtin_amdgcn_s_setprio(0);
                    } else {
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            const piece_t& bf = qb ? fb1[ks] : fb0[ks];
                            if constexpr (SW) acc[qa * 2 + ii][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, fa[ii][ks], acc[qa * 2 + ii][qb], 0, 0, 0);
                            else acc[qa * 2 + ii][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ii][ks], bf, acc[qa * 2 + ii][qb], 0, 0, 0);
                        }
                    __builtin_amdgcn_s_setprio(0);
                    }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                };
                phase(std::integral_constant<int, 1>{}); phase(std::integral_constant<int, 2>{});
                phase(std::integral_constant<int, 3>{}); phase(std::integral_constant<int, 4>{});
                phase(std::integral_constant<int, 5>{}); phase(std::integral_constant<int, 6>{});
                phase(std::integral_constant<int, 7>{}); phase(std::integral_constant<int, 8>{});
            };
            for (int j = 0; j + 1 < nj; ++j) iter(j, std::false_type{});
            iter(nj - 1, std::true_type{});
            if (xt) wait_vmcnt<0>();                   // the next tile's K-tile 0 (own pieces), before this tile's stores are issued
            if (!grp) __builtin_amdgcn_s_barrier();    // stagger out: both groups are past their last MFMA
            g += nk;
            }
        };
        if constexpr (RING) {
            if constexpr (swapped) kloop_ring(std::true_type{}); else kloop_ring(std::false_type{});
        } else {
            if constexpr (swapped) kloop(std::true_type{}); else kloop(std::false_type{});
        }

        const int row0 = m0 + wm * G::WROWS;
        const int col0 = n0 + wn * G::WCOLS;
        // CONV epilogues: quad partials red[WMc][BN / 4] (sum, sum of squares over each wave's rows) -> the tile's contribution to
        // the consumer's GroupNorm, partial[(sample, 256-pixel chunk)][group]: the layout of the separate statistics kernel
        // (tld_vae.hip).  Needs one whole tile per chunk (gn_hw % 256 == 0, M % 256 == 0), N % BN == 0 and gn_cpg % 4 == 0.
        auto gn_combine = [&](const float2* red) {
            if constexpr (CONV) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const int tid = wid * 64 + lane;
                const int qpg = p.gn_cpg >> 2;                       // quads per group
                if (tid < BN / p.gn_cpg) {
                    float a = 0.f, a2 = 0.f;
                    for (int w2 = 0; w2 < G::WMc; ++w2)
                        for (int k2 = 0; k2 < qpg; ++k2) { const float2 t = red[w2 * (BN / 4) + tid * qpg + k2]; a += t.x; a2 += t.y; }
                    const int bb = m0 / p.gn_hw;
                    const int chunk = (m0 - bb * p.gn_hw) >> 8, nchunk = p.gn_hw >> 8;
                    p.gn_partial[((size_t)bb * nchunk + chunk) * p.gn_groups + n0 / p.gn_cpg + tid] = make_float2(a, a2);
                }
            }
        };
        if constexpr (EPI == EPI_F32) {
#pragma unroll
            for (int i = 0; i < G::TM; ++i)
#pragma unroll
                for (int j = 0; j < G::TN; ++j) {
                    const int col = col0 + j * 32 + l31;
                    if (col >= p.N || (hmode && (i >> 1) != ht_qa)) continue;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + i * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                        if (row < p.M) p.c_f32[(KSPLIT_OK && nsplit > 1 ? (size_t)(kb0 / (p.K * 2)) * p.M * p.ldc : (size_t)0) + (size_t)row * p.ldc + col] = acc[i][j][r];
                    }
                }
        } else {
            __builtin_amdgcn_s_barrier();          // all waves finished reading the last stage: reuse it as scratch
            char* ws = RING ? smem + 4 * HT + wid * G::SCRATCH : smem + ((g - 1) & 1) * G::STAGE_BYTES + wid * G::SCRATCH;   // (ring: slots 4-7 are free by now)
            if constexpr (UPDW) {
                // Fused depthwise 3x3 + GELU epilogue (tld/transformer_blocks.py:96-103).  The tile's 256 rows are the 16 x 16 tokens of ONE
                // sample, so the conv of the MLP is tile-local: the pre-conv hidden tensor never travels to HBM (a 200 MB write + read per
                // layer) and the separate kernel disappears.  (A first form -- token-major fp32 window, packed FMAs -- was retired in round 4:
                // within 1 % of this one)  The K loop runs in the NATURAL MFMA order, so a
                // lane owns a channel and four consecutive tokens (= four consecutive image columns of one image row)
                // per register quad: two v_cvt_pk_bf16_f32 give the dwords (col 2q, col 2q+1) of that channel, and the
                // image in LDS is [token pair][channel] dwords.  The conv then needs no unpacking at all: every tap
                // pair is one v_dot2c_f32_bf16 (fp32 accumulate) against a packed bf16 weight pair,
                //   out(col 2q)   = (0,w0).P[q-1] + (w1,w2).P[q]          out(col 2q+1) = (w0,w1).P[q] + (w2,0).P[q+1]
                // per image row of the 3x3 window: 6 instructions per output instead of 9 FMAs + 3 unpacks.  The depthwise
                // weights are rounded to bf16 for this (like every other weight on the MFMA path); bias, accumulation and
                // GELU stay fp32.  A thread owns a channel quad and TWO ADJACENT image rows (4 window rows in registers,
                // 16 independent accumulation chains per step); rows outside the image are read from an all-zero
                // pair-row in LDS, columns outside the image are zero registers: no border code in the loop.
                static_assert(BN == 256, "fused depthwise epilogue is written for 256-column tiles");
                char* H = smem;
                {   // zero pair-row (never overwritten by the stages: it lies behind them)
                    if (it == 0) {
                        *reinterpret_cast<u32x4*>(smem + G::ZROW_OFF + threadIdx.x * 16) = u32x4{0u, 0u, 0u, 0u};
                        *reinterpret_cast<u32x4*>(smem + G::ZROW_OFF + 8192 + threadIdx.x * 16) = u32x4{0u, 0u, 0u, 0u};
                    }
                }
                const bool ln3 = p.row_stats != nullptr;
                float cst[G::TN], bst[G::TN];
#pragma unroll
                for (int j = 0; j < G::TN; ++j) {
                    const int cl = wn * 64 + j * 32 + l31;
                    bst[j] = *reinterpret_cast<const float*>(smem + G::CB2_OFF + 1024 + cl * 4);
                    cst[j] = ln3 ? *reinterpret_cast<const float*>(smem + G::CB2_OFF + cl * 4) : 0.f;
                }
                if (!TLD_EPI_BIT(8))
#pragma unroll
                for (int i = 0; i < G::TM; ++i) {
                    // (mean, rstd) of this lane's 16 tokens of the MFMA row-tile: all eight 16-B reads first, one wait
                    float4 sv[8];
                    if (ln3) {
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const int tok0 = wm * G::WROWS + i * 32 + 8 * rq + 4 * hi;
                            sv[2 * rq] = *reinterpret_cast<const float4*>(smem + G::ROWSTAT2_OFF + tok0 * 8);
                            sv[2 * rq + 1] = *reinterpret_cast<const float4*>(smem + G::ROWSTAT2_OFF + tok0 * 8 + 16);
                        }
                    } else {
#pragma unroll
                        for (int q8 = 0; q8 < 8; ++q8) sv[q8] = make_float4(0.f, 1.f, 0.f, 1.f);
                    }
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int tok0 = wm * G::WROWS + i * 32 + 8 * rq + 4 * hi;          // 4 consecutive tokens
                        const float4 s01 = sv[2 * rq], s23 = sv[2 * rq + 1];
                        const float rs0 = s01.y, rs1 = s01.w, rs2 = s23.y, rs3 = s23.w;
                        const float nm0 = -s01.y * s01.x, nm1 = -s01.w * s01.z, nm2 = -s23.y * s23.x, nm3 = -s23.w * s23.z;
#pragma unroll
                        for (int j = 0; j < G::TN; ++j) {
                            bf16x2 lo, hi2;
                            lo[0] = (bf16)fmaf(rs0, acc[i][j][rq * 4 + 0], fmaf(nm0, cst[j], bst[j]));
                            lo[1] = (bf16)fmaf(rs1, acc[i][j][rq * 4 + 1], fmaf(nm1, cst[j], bst[j]));
                            hi2[0] = (bf16)fmaf(rs2, acc[i][j][rq * 4 + 2], fmaf(nm2, cst[j], bst[j]));
                            hi2[1] = (bf16)fmaf(rs3, acc[i][j][rq * 4 + 3], fmaf(nm3, cst[j], bst[j]));
                            char* dst = H + (tok0 >> 1) * G::IMG2_PITCH + (wn * 64 + j * 32 + l31) * 4;
                            *reinterpret_cast<bf16x2*>(dst) = lo;
                            *reinterpret_cast<bf16x2*>(dst + G::IMG2_PITCH) = hi2;
                        }
                    }
                }
                __builtin_amdgcn_s_barrier();
                if constexpr (EPI == EPI_UP_DWCONV32) {
                    // ---- 32 x 32 token grid (512 px): the tile is image rows 8 t .. 8 t + 7 of one sample (t = tile row & 3).  Same image format -- token
                    // pair (row 32 + col) >> 1, 16 pair-columns per row -- and the same tap arithmetic as the 16 x 16 form below.  Output rows 1 .. 6 need
                    // only tile rows; rows 0 and 7 need the neighbouring tile's hidden rows unless they are the image's own top / bottom row (zero pad).
                    // Work items: (row pair r0 in {1, 3, 5}) x (column half) = 6 of the 8 thread groups; groups 6, 7 do the image-border row of the first /
                    // last tile of a sample.  The tile's rows 0, 1, 6, 7 go to p.dw_seam as they stand in the image; launch_dwconv_seam finishes the
                    // two rows at every seam from them.
                    const int cq = threadIdx.x & 63;
                    const int c0 = n0 + cq * 4;
                    const int w2 = threadIdx.x >> 6;
                    const int trow = (m0 >> 8) & 3;                       // which quarter of the sample
                    {   // seam rows: pair-rows [0, 32) and [96, 128) of the image, 16 bytes (4 channels) per thread and pair-row
                        uint32_t* sb = p.dw_seam + ((size_t)(m0 >> 8) * (p.N >> 8) + (n0 >> 8)) * (64 * 256);
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int pr = w2 * 8 + k;                                         // 0 .. 63
                            const int src = pr < 32 ? pr : pr + 64;
                            if (pr < 32 ? trow == 0 : trow == 3) continue;                     // (no seam above the sample's first tile / below its last)
                            *reinterpret_cast<u32x4*>(sb + (size_t)pr * 256 + cq * 4) = *reinterpret_cast<const u32x4*>(H + src * G::IMG2_PITCH + cq * 16);
                        }
                    }
                    // work split: wave group w2 owns pair-columns 2 w2 and 2 w2 + 1 (token columns 4 w2 .. 4 w2 + 3) of EVERY row pair -- (1, 2), (3, 4), (5, 6), and in
                    // the first / last tile of a sample also (0, 1) / (6, 7), of which only the image-border row is stored.  (A first version gave six of the
                    // eight groups a (row pair, column half) each and left two idle on interior tiles: 235 us per launch against 165 + 94 for the two kernels.)
                    {
                        u32x4 WA[3], WB[3], WC[3], WD[3];
#pragma unroll
                        for (int du = 0; du < 3; ++du) {
                            WA[du] = *reinterpret_cast<const u32x4*>(p.dw_wpk + (size_t)(du * 4 + 0) * p.N + c0);
                            WB[du] = *reinterpret_cast<const u32x4*>(p.dw_wpk + (size_t)(du * 4 + 1) * p.N + c0);
                            WC[du] = *reinterpret_cast<const u32x4*>(p.dw_wpk + (size_t)(du * 4 + 2) * p.N + c0);
                            WD[du] = *reinterpret_cast<const u32x4*>(p.dw_wpk + (size_t)(du * 4 + 3) * p.N + c0);
                        }
                        const float4 bsv = *reinterpret_cast<const float4*>(p.dw_b + c0);
                        const f32x4 bs = {bsv.x, bsv.y, bsv.z, bsv.w};
                        auto dot2 = [](unsigned a, unsigned b, float c) {
                            return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
                        };
                        const int npairs = (trow == 0 || trow == 3) ? 4 : 3;
#pragma unroll 1
                        for (int pi = 0; pi < npairs; ++pi) {
                            const int r0 = pi < 3 ? 1 + 2 * pi : (trow == 0 ? 0 : 6);
                            const bool st0 = pi < 3 || trow == 0, st1 = pi < 3 || trow == 3;
                            const char* rb[4];                                   // window rows r0 - 1 .. r0 + 2 (16 pair-columns each)
#pragma unroll
                            for (int k4 = 0; k4 < 4; ++k4) {
                                const int rr = r0 - 1 + k4;
                                rb[k4] = ((rr < 0 || rr > 7) ? smem + G::ZROW_OFF : H + rr * 16 * G::IMG2_PITCH) + cq * 16;
                            }
                            auto ld = [&](int q, u32x4 (&c)[4]) {
#pragma unroll
                                for (int k4 = 0; k4 < 4; ++k4) c[k4] = *reinterpret_cast<const u32x4*>(rb[k4] + q * G::IMG2_PITCH);
                            };
                            auto zero = [&](u32x4 (&c)[4]) {
#pragma unroll
                                for (int k4 = 0; k4 < 4; ++k4) c[k4] = u32x4{0u, 0u, 0u, 0u};
                            };
                            bf16* dst0 = p.out_bf16 + ((size_t)m0 + (size_t)r0 * 32) * p.ldo + c0;
                            auto emit = [&](const u32x4 (&L)[4], const u32x4 (&Mc)[4], const u32x4 (&R)[4], int q) {       // q: pair-column inside the row
#pragma unroll
                                for (int rr = 0; rr < 2; ++rr) {
                                    f32x4 ae = bs, ao = bs;
#pragma unroll
                                    for (int du = 0; du < 3; ++du)
#pragma unroll
                                        for (int ch = 0; ch < 4; ++ch) {
                                            ae[ch] = dot2(L[rr + du][ch], WA[du][ch], ae[ch]);
                                            ao[ch] = dot2(Mc[rr + du][ch], WC[du][ch], ao[ch]);
                                            ae[ch] = dot2(Mc[rr + du][ch], WB[du][ch], ae[ch]);
                                            ao[ch] = dot2(R[rr + du][ch], WD[du][ch], ao[ch]);
                                        }
                                    f32x2 e0 = {ae[0], ae[1]}, e1 = {ae[2], ae[3]}, o0 = {ao[0], ao[1]}, o1 = {ao[2], ao[3]};
                                    e0 = gelu_erf_fast2_half(e0); e1 = gelu_erf_fast2_half(e1);
                                    o0 = gelu_erf_fast2_half(o0); o1 = gelu_erf_fast2_half(o1);
                                    bf16x4 oe, oo;
                                    oe[0] = (bf16)e0[0]; oe[1] = (bf16)e0[1]; oe[2] = (bf16)e1[0]; oe[3] = (bf16)e1[1];
                                    oo[0] = (bf16)o0[0]; oo[1] = (bf16)o0[1]; oo[2] = (bf16)o1[0]; oo[3] = (bf16)o1[1];
                                    if (rr == 0 ? st0 : st1) {
                                        TLD_STORE(reinterpret_cast<bf16x4*>(dst0 + ((size_t)rr * 32 + 2 * q) * p.ldo), oe);
                                        TLD_STORE(reinterpret_cast<bf16x4*>(dst0 + ((size_t)rr * 32 + 2 * q + 1) * p.ldo), oo);
                                    }
                                }
                            };
                            u32x4 cL[4], cA[4], cB[4], cR[4];
                            const int q0 = 2 * w2;
                            if (w2) ld(q0 - 1, cL); else zero(cL);
                            ld(q0, cA); ld(q0 + 1, cB);
                            if (w2 < 7) ld(q0 + 2, cR); else zero(cR);
                            emit(cL, cA, cB, q0);
                            emit(cA, cB, cR, q0 + 1);
                        }
                    }
                } else if (!TLD_EPI_BIT(4)) {
                    // (round 4, both measured and dropped: these 13 loads hoisted above the image write -- 20 spilled registers, 188 -> 198 us -- and the
                    // same constants staged in LDS by DMA with the side tables -- neutral, 185.5 vs 185.4 us: their latency is not what this phase waits for)
                    const int cq = threadIdx.x & 63;                     // channel quad
                    const int c0 = n0 + cq * 4;
                    u32x4 WA[3], WB[3], WC[3], WD[3];                    // packed bf16 weight pairs, 4 channels each
#pragma unroll
                    for (int du = 0; du < 3; ++du) {
                        WA[du] = *reinterpret_cast<const u32x4*>(p.dw_wpk + (size_t)(du * 4 + 0) * p.N + c0);
                        WB[du] = *reinterpret_cast<const u32x4*>(p.dw_wpk + (size_t)(du * 4 + 1) * p.N + c0);
                        WC[du] = *reinterpret_cast<const u32x4*>(p.dw_wpk + (size_t)(du * 4 + 2) * p.N + c0);
                        WD[du] = *reinterpret_cast<const u32x4*>(p.dw_wpk + (size_t)(du * 4 + 3) * p.N + c0);
                    }
                    const float4 bsv = *reinterpret_cast<const float4*>(p.dw_b + c0);
                    const f32x4 bs = {bsv.x, bsv.y, bsv.z, bsv.w};
                    const int w2 = threadIdx.x >> 6;                     // output image rows 2 w2 and 2 w2 + 1
                    const char* rb[4];                                   // window rows 2 w2 - 1 .. 2 w2 + 2 (8 pair-columns each)
                    rb[0] = (w2 == 0 ? smem + G::ZROW_OFF : H + (2 * w2 - 1) * 8 * G::IMG2_PITCH) + cq * 16;
                    rb[1] = H + (2 * w2) * 8 * G::IMG2_PITCH + cq * 16;
                    rb[2] = H + (2 * w2 + 1) * 8 * G::IMG2_PITCH + cq * 16;
                    rb[3] = (w2 == 7 ? smem + G::ZROW_OFF : H + (2 * w2 + 2) * 8 * G::IMG2_PITCH) + cq * 16;
                    auto ld = [&](int q, u32x4 (&c)[4]) {
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) c[k4] = *reinterpret_cast<const u32x4*>(rb[k4] + q * G::IMG2_PITCH);
                    };
                    auto zero = [&](u32x4 (&c)[4]) {
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) c[k4] = u32x4{0u, 0u, 0u, 0u};
                    };
                    auto dot2 = [](unsigned a, unsigned b, float c) {
                        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
                    };
                    bf16* dst0 = p.out_bf16 + ((size_t)m0 + (size_t)w2 * 32) * p.ldo + c0;
                    auto emit = [&](const u32x4 (&L)[4], const u32x4 (&Mc)[4], const u32x4 (&R)[4], int q) {
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr) {
                            f32x4 ae = bs, ao = bs;
#pragma unroll
                            for (int du = 0; du < 3; ++du)
#pragma unroll
                                for (int ch = 0; ch < 4; ++ch) {
                                    ae[ch] = dot2(L[rr + du][ch], WA[du][ch], ae[ch]);
                                    ao[ch] = dot2(Mc[rr + du][ch], WC[du][ch], ao[ch]);
                                    ae[ch] = dot2(Mc[rr + du][ch], WB[du][ch], ae[ch]);
                                    ao[ch] = dot2(R[rr + du][ch], WD[du][ch], ao[ch]);
                                }
                            f32x2 e0 = {ae[0], ae[1]}, e1 = {ae[2], ae[3]}, o0 = {ao[0], ao[1]}, o1 = {ao[2], ao[3]};
                            if (!TLD_EPI_BIT(2)) {
                                e0 = gelu_erf_fast2_half(e0); e1 = gelu_erf_fast2_half(e1);
                                o0 = gelu_erf_fast2_half(o0); o1 = gelu_erf_fast2_half(o1);
                            }
                            bf16x4 oe, oo;
                            oe[0] = (bf16)e0[0]; oe[1] = (bf16)e0[1]; oe[2] = (bf16)e1[0]; oe[3] = (bf16)e1[1];
                            oo[0] = (bf16)o0[0]; oo[1] = (bf16)o0[1]; oo[2] = (bf16)o1[0]; oo[3] = (bf16)o1[1];
                            if (!TLD_EPI_BIT(1)) {
                                TLD_STORE(reinterpret_cast<bf16x4*>(dst0 + ((size_t)rr * 16 + 2 * q) * p.ldo), oe);
                                TLD_STORE(reinterpret_cast<bf16x4*>(dst0 + ((size_t)rr * 16 + 2 * q + 1) * p.ldo), oo);
                            }
                        }
                    };
                    u32x4 c0v[4], c1v[4], c2v[4];
                    zero(c0v);
                    ld(0, c1v);
#pragma unroll 1
                    for (int q = 0; q < 6; q += 3) {
                        ld(q + 1, c2v); emit(c0v, c1v, c2v, q);
                        ld(q + 2, c0v); emit(c1v, c2v, c0v, q + 1);
                        ld(q + 3, c1v); emit(c2v, c0v, c1v, q + 2);
                    }
                    ld(7, c2v); emit(c0v, c1v, c2v, 6);
                    zero(c0v); emit(c1v, c2v, c0v, 7);
                }
                // the ring restarts for the next tile: its first K-step could not be prefetched (LDS was the image)
                __builtin_amdgcn_s_barrier();
                if constexpr (!RING) { if (has_next) issue(m0n, n0n, g); }   // (ring: the next tile starts cold in kloop_ring)
            } else if constexpr (EPI == EPI_BIAS_RESID) {
                constexpr int P = 32 * 4 + 16;      // one 32x32 fp32 tile, padded pitch
                float gsum[CONV ? G::TN : 1], gsq[CONV ? G::TN : 1];     // CONV: GroupNorm partials of this lane's column quad (lane & 7) per 32-column block
                if constexpr (CONV) {
#pragma unroll
                    for (int j = 0; j < G::TN; ++j) { gsum[j] = 0.f; gsq[j] = 0.f; }
                }
                // the residual values of sub-tile (i, j + 1) are fetched while (i, j) goes through the LDS transpose: each of the TM x TN sub-tiles used
                // to expose one global-load latency (every workgroup of the one-round down projection at the same moment)
                // (one buffer of four 8-byte values: each is reloaded for the next sub-tile right after it has been consumed -- a second buffer
                // pushed the 384-wide kernel, 192 accumulator registers, into spills inside its K loop)
                resid4_t rnx[4];
                auto rfetch1 = [&](int i2, int j2, int itr) {
                    const int idx = itr * 64 + lane;
                    const int row = row0 + i2 * 32 + (idx >> 3), col = col0 + j2 * 32 + (idx & 7) * 4;
                    if (row < p.M && col < p.N && !TLD_EPI_BIT(16)) rnx[itr] = rs_raw4(p.resid + (size_t)row * p.ldr + col);
                    else rnx[itr] = resid4_t{};
                };
#pragma unroll
                for (int itr = 0; itr < 4; ++itr) rfetch1(0, 0, itr);
#pragma unroll
                for (int i = 0; i < G::TM; ++i) {
                    float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};   // row partials of the new residual
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
                                resid_t* px = p.resid + (size_t)row * p.ldr + col;
                                float4 o = rs_widen4(rnx[itr]);
                                o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                                if (!TLD_EPI_BIT(1)) rs_store4(px, o);
                                if constexpr (CONV) {
                                    const float r0 = rs_round(o.x), r1 = rs_round(o.y), r2 = rs_round(o.z), r3 = rs_round(o.w);
                                    gsum[j] += (r0 + r1) + (r2 + r3);
                                    gsq[j] = fmaf(r0, r0, fmaf(r1, r1, fmaf(r2, r2, fmaf(r3, r3, gsq[j]))));
                                }
                                if constexpr (G::WCOLS == 96) {          // (LayerNorm-1 fold: 96-column groups only)
                                    const float r0 = rs_round(o.x), r1 = rs_round(o.y), r2 = rs_round(o.z), r3 = rs_round(o.w);
                                    ps[itr] += (r0 + r1) + (r2 + r3);
                                    pq[itr] = fmaf(r0, r0, fmaf(r1, r1, fmaf(r2, r2, fmaf(r3, r3, pq[itr]))));
                                }
                            }
                            if (j + 1 < G::TN) rfetch1(i, j + 1, itr); else if (i + 1 < G::TM) rfetch1(i + 1, 0, itr);
                        }
                    }
                    if constexpr (G::WCOLS == 96) {
                        if (p.stats_out) {
                            // a row's 96 columns of this wave sit in 8 adjacent lanes: three DPP steps, lane (ch == 0) writes
                            const int slot = (n0 + wn * G::WCOLS) / 96;
#pragma unroll
                            for (int itr = 0; itr < 4; ++itr) {
                                float a = ps[itr], q2 = pq[itr];
                                a = dpp_add<0xB1>(a); q2 = dpp_add<0xB1>(q2);
                                a = dpp_add<0x4E>(a); q2 = dpp_add<0x4E>(q2);
                                a = dpp_add<0x141>(a); q2 = dpp_add<0x141>(q2);
                                const int row = row0 + i * 32 + itr * 8 + (lane >> 3);
                                if ((lane & 7) == 0 && row < p.M && slot < kLnSlots) p.stats_out[(size_t)row * kLnSlots + slot] = make_float2(a, q2);
                            }
                        }
                    }
                }
                if constexpr (CONV) {
                    if (p.gn_partial) {
                        // lanes with the same (lane & 7) hold the same column quad for different rows: three xor steps (8, 16, 32)
                        float2* red = reinterpret_cast<float2*>(smem + ((g - 1) & 1) * G::STAGE_BYTES + 8 * G::SCRATCH);       // [WMc][BN / 4]
#pragma unroll
                        for (int j = 0; j < G::TN; ++j) {
                            float a = gsum[j], a2 = gsq[j];
#pragma unroll
                            for (int o2 = 8; o2 < 64; o2 <<= 1) { a += __shfl_xor(a, o2, 64); a2 += __shfl_xor(a2, o2, 64); }
                            if (lane < 8) red[wm * (BN / 4) + wn * (G::WCOLS / 4) + j * 8 + lane] = make_float2(a, a2);
                        }
                        gn_combine(red);
                    }
                }
            } else if constexpr (EPI == EPI_QKV_ATTN) {
                // ---- fused self-attention of (sample m0 / 256, head n0 / 192).  The accumulators hold x W'^T for the head's q | k | v columns
                // (swapped order: lane = token row, registers = 4 consecutive columns per quad).  LayerNorm-1 is applied in place exactly as
                // in EPI_QKV_LN, then q, k (row images, the GEMM's 128-byte rows + XOR swizzle) and v^T go to LDS and every wave runs the
                // head's attention for 32 query rows (tld_attn_core.h).  Only att[256 x 64] is stored.
                int e_lane = lane, e_l31 = l31, e_hi = hi;
                asm volatile("" : "+v"(e_lane), "+v"(e_l31), "+v"(e_hi));
                {
                    const int tid = wid * 64 + e_lane;
                    const int s0 = (tid & 1) * 4;
                    const float4* raw = reinterpret_cast<const float4*>(smem + G::LN_RAW + (tid >> 1) * 64 + s0 * 8);
                    float su = 0.f, sq = 0.f;
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
                        if (s0 + 2 * q2 < p.ln_slots) {
                            const float4 v = raw[q2];
                            su += v.x + v.z; sq += v.y + v.w;
                        }
                    su = dpp_add<0xB1>(su); sq = dpp_add<0xB1>(sq);
                    const float inv_k = 1.0f / (float)p.K;
                    const float mu = su * inv_k;
                    const float var = fmaxf(fmaf(sq, inv_k, -mu * mu), 0.f);
                    if (!(tid & 1))
                        reinterpret_cast<float2*>(smem + G::LN_ST)[tid >> 1] = make_float2(mu, __builtin_amdgcn_rsqf(var + kLnEps));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                {
                    float rs[G::TM], nm[G::TM];
#pragma unroll
                    for (int i = 0; i < G::TM; ++i) {
                        const float2 st = reinterpret_cast<const float2*>(smem + G::LN_ST)[wm * G::WROWS + i * 32 + e_l31];
                        rs[i] = st.y; nm[i] = -st.y * st.x;
                    }
#pragma unroll
                    for (int j = 0; j < G::TN; ++j)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const int cl = wn * G::WCOLS + j * 32 + 8 * rq + 4 * e_hi;         // column inside the tile
                            const float4 c4 = *reinterpret_cast<const float4*>(smem + G::LN_CB + cl * 4);
                            const float4 b4 = *reinterpret_cast<const float4*>(smem + G::LN_CB + 1024 + cl * 4);
#pragma unroll
                            for (int i = 0; i < G::TM; ++i) {
                                acc[i][j][rq * 4 + 0] = fmaf(rs[i], acc[i][j][rq * 4 + 0], fmaf(nm[i], c4.x, b4.x));
                                acc[i][j][rq * 4 + 1] = fmaf(rs[i], acc[i][j][rq * 4 + 1], fmaf(nm[i], c4.y, b4.y));
                                acc[i][j][rq * 4 + 2] = fmaf(rs[i], acc[i][j][rq * 4 + 2], fmaf(nm[i], c4.z, b4.z));
                                acc[i][j][rq * 4 + 3] = fmaf(rs[i], acc[i][j][rq * 4 + 3], fmaf(nm[i], c4.w, b4.w));
                            }
                        }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();              // every wave is done with the LayerNorm tables: the images overlay them
                char* Kimg = smem + G::AT_K;
                char* Vimg = smem + G::AT_V;
                char* Qimg = smem + G::AT_Q;
                // tile columns (host-side row order of the packed weight, tld_engine.hip): wave column wn = feature half, its three 32-column blocks j = q, k, v of
                // features 32 wn .. 32 wn + 31 -- every wave writes a third of its values to each image
#pragma unroll
                for (int j = 0; j < G::TN; ++j) {
#pragma unroll
                    for (int i = 0; i < G::TM; ++i) {
                        const int row = wm * G::WROWS + i * 32 + e_l31;
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            bf16x4 pk;
                            pk[0] = (bf16)acc[i][j][rq * 4 + 0]; pk[1] = (bf16)acc[i][j][rq * 4 + 1];
                            pk[2] = (bf16)acc[i][j][rq * 4 + 2]; pk[3] = (bf16)acc[i][j][rq * 4 + 3];
                            const int f = wn * 32 + 8 * rq + 4 * e_hi;                 // feature inside q, k or v
                            if (j < 2) {
                                char* img = j == 0 ? Qimg : Kimg;
                                *reinterpret_cast<bf16x4*>(img + row * 128 + ((((f >> 3) ^ ((row >> 1) & 7)) << 4) | ((f & 7) << 1))) = pk;
                            } else {
#pragma unroll
                                for (int e2 = 0; e2 < 4; ++e2)
                                    *reinterpret_cast<bf16*>(Vimg + (f + e2) * kAttnVPitch + row * 2) = pk[e2];
                            }
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                {
                    const int q0 = wid * 32;                     // this wave's query rows
                    bf16x8 qf[4];
                    {
                        const int row = q0 + e_l31;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
                            qf[ks] = *reinterpret_cast<const bf16x8*>(Qimg + row * 128 + (((ks * 2 + e_hi) ^ ((row >> 1) & 7)) << 4));
                    }
                    f32x16 o[2];
#ifdef TLD_QA_DBG      // cost attribution build (wrong results): 1 = no attention arithmetic at all (K loop + images + stores only)
                    float inv = 1.0f;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[ct][r] = (float)qf[ct][r & 7];
#else
                    const float inv = attn256_wave(Kimg, Vimg, qf, o, e_l31, e_hi);
#endif
                    // whole 128-byte rows through a per-wave transpose patch (16 rows x 144 B), as attn1_kernel stores them; the patch overlays the
                    // wave's OWN query rows of the Q image (read by nobody else, and this wave's reads are long retired)
                    char* T = Qimg + wid * 4096;
                    bf16* dst0 = p.out_bf16 + (size_t)(m0 + q0) * p.ldo + (n0 / BN) * 64;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        if ((e_l31 >> 4) == half) {
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                                for (int rq = 0; rq < 4; ++rq) {
                                    bf16x4 pk;
#pragma unroll
                                    for (int e2 = 0; e2 < 4; ++e2) pk[e2] = (bf16)(o[ct][rq * 4 + e2] * inv);
                                    *reinterpret_cast<bf16x4*>(T + (e_l31 & 15) * 144 + ct * 64 + rq * 16 + e_hi * 8) = pk;
                                }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
                        for (int it2 = 0; it2 < 2; ++it2) {
                            const int r = it2 * 8 + (e_lane >> 3), c16 = e_lane & 7;
                            const u32x4 w = *reinterpret_cast<const u32x4*>(T + r * 144 + c16 * 16);
                            __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(dst0 + (size_t)(half * 16 + r) * p.ldo + c16 * 8));
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                }
                // (the next tile's K loop opens with a barrier: every wave is out of the images before stage 1 is written again)
            } else if constexpr (G::WCOLS == 64) {
                // lane-derived values re-materialised per tile: otherwise every address expression of this epilogue is
                // hoisted out of the tile loop and lives (or spills) across the K loops
                int e_lane = lane, e_l31 = l31, e_hi = hi;
                asm volatile("" : "+v"(e_lane), "+v"(e_l31), "+v"(e_hi));
                if constexpr (LN) {
                    // (mean, rstd) of the tile's rows from the partial sums: two threads per row, four slots each
                    {
                        const int tid = wid * 64 + e_lane;
                        const int s0 = (tid & 1) * 4;
                        const float4* raw = reinterpret_cast<const float4*>(smem + G::LN_RAW + (tid >> 1) * 64 + s0 * 8);
                        float su = 0.f, sq = 0.f;
#pragma unroll
                        for (int q2 = 0; q2 < 2; ++q2)
                            if (s0 + 2 * q2 < p.ln_slots) {
                                const float4 v = raw[q2];
                                su += v.x + v.z; sq += v.y + v.w;
                            }
                        su = dpp_add<0xB1>(su); sq = dpp_add<0xB1>(sq);          // the row's other half sits in lane ^ 1
                        const float inv_k = 1.0f / (float)p.K;
                        const float mu = su * inv_k;
                        const float var = fmaxf(fmaf(sq, inv_k, -mu * mu), 0.f);
                        if (!(tid & 1))
                            reinterpret_cast<float2*>(smem + G::LN_ST)[tid >> 1] = make_float2(mu, __builtin_amdgcn_rsqf(var + kLnEps));
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    // y = rstd (acc - mean c1) + b1 in place: lane = token row (all tiles are swapped), column constants
                    // fetched once per 4 columns
                    float rs[G::TM], nm[G::TM];
#pragma unroll
                    for (int i = 0; i < G::TM; ++i) {
                        const float2 st = reinterpret_cast<const float2*>(smem + G::LN_ST)[wm * G::WROWS + i * 32 + e_l31];
                        rs[i] = st.y; nm[i] = -st.y * st.x;
                    }
#pragma unroll
                    for (int j = 0; j < G::TN; ++j)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const int cl = wn * G::WCOLS + j * 32 + 8 * rq + 4 * e_hi;         // column inside the tile
                            const float4 c4 = *reinterpret_cast<const float4*>(smem + G::LN_CB + cl * 4);
                            const float4 b4 = *reinterpret_cast<const float4*>(smem + G::LN_CB + 1024 + cl * 4);
#pragma unroll
                            for (int i = 0; i < G::TM; ++i) {
                                acc[i][j][rq * 4 + 0] = fmaf(rs[i], acc[i][j][rq * 4 + 0], fmaf(nm[i], c4.x, b4.x));
                                acc[i][j][rq * 4 + 1] = fmaf(rs[i], acc[i][j][rq * 4 + 1], fmaf(nm[i], c4.y, b4.y));
                                acc[i][j][rq * 4 + 2] = fmaf(rs[i], acc[i][j][rq * 4 + 2], fmaf(nm[i], c4.z, b4.z));
                                acc[i][j][rq * 4 + 3] = fmaf(rs[i], acc[i][j][rq * 4 + 3], fmaf(nm[i], c4.w, b4.w));
                            }
                        }
                }
                const bool to_vt = IS_QKV && v_tile;
                if (!to_vt) {
                    // one 32-row x 64-col bf16 slab per pass: 128-B pitch, 16-B chunks XOR-swizzled with (row & 7)
                    constexpr int P = 128;
                    // CONV: GroupNorm statistics of the tile for the consumer (GemmParams::gn_partial): per lane, (sum, sum of squares)
                    // of the STORED values of its rows, one pair per 4-column quad (j, rq)
                    // (packed fp32 pairs; taken on the fp32 values BEFORE their bf16 rounding -- the rounding error is zero-mean, so a group's
                    // mean / variance over >= 1024 values move by far less than one bf16 ulp, and the four conversions back per quad go away)
                    f32x2 gsum[CONV ? G::TN * 4 : 1], gsq[CONV ? G::TN * 4 : 1];
                    if constexpr (CONV) {
#pragma unroll
                        for (int q4 = 0; q4 < G::TN * 4; ++q4) { gsum[q4] = f32x2{0.f, 0.f}; gsq[q4] = f32x2{0.f, 0.f}; }
                    }
#pragma unroll
                    for (int i = 0; i < G::TM; ++i) {
                        if (hmode && (i >> 1) != ht_qa) continue;       // half-tile item: the other workgroup's rows
#pragma unroll
                        for (int j = 0; j < G::TN; ++j)
#pragma unroll
                            for (int rq = 0; rq < 4; ++rq) {
                                const int cl = j * 32 + 8 * rq + 4 * e_hi;
                                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                                float rs_ = 1.0f;
                                if constexpr (EPI == EPI_BIAS_BF16) {
                                    const int cg = col0 + cl < p.N ? col0 + cl : 0;
                                    const int ct = wn * G::WCOLS + cl;                  // column inside the tile
                                    if constexpr (PLAIN_CB) bv = *reinterpret_cast<const float4*>(smem + G::PLAINCB_OFF + 1024 + ct * 4);
                                    else bv = *reinterpret_cast<const float4*>(p.bias + cg);
                                    if (p.row_stats) {      // LayerNorm-3 folded in: rstd_m (acc - mean_m c1[n]) + b1[n], lane = token row
                                        const float2 st2 = *reinterpret_cast<const float2*>(smem + G::PLAINRS_OFF + (wm * G::WROWS + i * 32 + e_l31) * 8);
                                        const float4 c4 = PLAIN_CB ? *reinterpret_cast<const float4*>(smem + G::PLAINCB_OFF + ct * 4) : *reinterpret_cast<const float4*>(p.ln_c1 + cg);
                                        rs_ = st2.y;
                                        const float nm_ = -st2.y * st2.x;
                                        bv.x = fmaf(nm_, c4.x, bv.x); bv.y = fmaf(nm_, c4.y, bv.y);
                                        bv.z = fmaf(nm_, c4.z, bv.z); bv.w = fmaf(nm_, c4.w, bv.w);
                                    }
                                }
                                const f32x2 lo2 = {fmaf(rs_, acc[i][j][rq * 4 + 0], bv.x), fmaf(rs_, acc[i][j][rq * 4 + 1], bv.y)};
                                const f32x2 hi2 = {fmaf(rs_, acc[i][j][rq * 4 + 2], bv.z), fmaf(rs_, acc[i][j][rq * 4 + 3], bv.w)};
                                bf16x4 pk;
                                pk[0] = (bf16)lo2[0]; pk[1] = (bf16)lo2[1]; pk[2] = (bf16)hi2[0]; pk[3] = (bf16)hi2[1];
                                *reinterpret_cast<bf16x4*>(ws + e_l31 * P + ((((cl >> 3) ^ (e_l31 & 7)) << 4) | ((cl & 7) << 1))) = pk;
                                if constexpr (CONV) {
                                    gsum[j * 4 + rq] += lo2; gsum[j * 4 + rq] += hi2;
                                    gsq[j * 4 + rq] = __builtin_elementwise_fma(lo2, lo2, gsq[j * 4 + rq]);
                                    gsq[j * 4 + rq] = __builtin_elementwise_fma(hi2, hi2, gsq[j * 4 + rq]);
                                }
                            }
#pragma unroll
                        for (int itr = 0; itr < 4; ++itr) {
                            const int idx = itr * 64 + e_lane;
                            const int rl = idx >> 3, ch = idx & 7;
                            const u32x4 v = *reinterpret_cast<const u32x4*>(ws + rl * P + ((ch ^ (rl & 7)) << 4));
                            const int row = row0 + i * 32 + rl, col = col0 + ch * 8;
                            if (row < p.M && col < p.N) {
                                u32x4* dst = reinterpret_cast<u32x4*>(p.out_bf16 + (size_t)row * p.ldo + col);
                                if constexpr (IS_QKV) *dst = v;              // attention re-reads q|k from L2
                                else TLD_STORE(dst, v);
                            }
                        }
                    }
                    if constexpr (CONV) {
                        if (p.gn_partial) {
                            // the 32 lanes of a half-wave are the rows of quad (j, rq, hi): one half_sum each, lane 0 / 32 files it
                            float2* red = reinterpret_cast<float2*>(smem + ((g - 1) & 1) * G::STAGE_BYTES + 8 * G::SCRATCH);   // [WMc][BN / 4]
#pragma unroll
                            for (int q4 = 0; q4 < G::TN * 4; ++q4) {
                                const float a = half_sum(gsum[q4][0] + gsum[q4][1]), a2 = half_sum(gsq[q4][0] + gsq[q4][1]);
                                if (e_l31 == 0) red[wm * (BN / 4) + wn * (G::WCOLS / 4) + (q4 >> 2) * 8 + (q4 & 3) * 2 + e_hi] = make_float2(a, a2);
                            }
                            gn_combine(red);
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
                            if (hmode && half != ht_qa) continue;
#pragma unroll
                            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                                for (int rq = 0; rq < 4; ++rq) {
                                    const int i = half * 2 + ii;
                                    // swapped accumulators: lane = token, registers = features -> 2-byte writes
#pragma unroll
                                    for (int e2 = 0; e2 < 4; ++e2)
                                        *reinterpret_cast<bf16*>(ws + (8 * rq + 4 * e_hi + e2) * P + (ii * 32 + e_l31) * 2) = (bf16)acc[i][j][rq * 4 + e2];
                                }
#pragma unroll
                            for (int itr = 0; itr < 4; ++itr) {
                                const int idx = itr * 64 + e_lane;
                                const int f = idx >> 3, ch = idx & 7;
                                const u32x4 v = *reinterpret_cast<const u32x4*>(ws + f * P + ch * 16);
                                const int row = row0 + half * 64 + ch * 8;
                                const int fg = j * 32 + f;
                                if (row < p.M && col0 + fg < p.N) {
                                    const int b = row / p.ntok, tk = row - b * p.ntok;
                                    *reinterpret_cast<u32x4*>(p.vt + ((size_t)b * p.d + cbase + fg) * p.ntok + tk) = v;
                                }
                            }
                        }
                }
            }
        }
        m0 = m0n; n0 = n0n; kb0 = kbn;
    }
}

template <int BN>
void launch256p(const GemmParams& p, int epilogue, hipStream_t s) {
    using G = G256P<BN>;
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + G::BM - 1) / G::BM;
    const int ncu = device_cu_count();
    const int nsplit = (epilogue == EPI_F32 && !p.f8 && !p.conv && p.ksplit > 1) ? p.ksplit : 1;
    const int nblocks = ntm * ntn * nsplit < ncu ? ntm * ntn * nsplit : ncu;
    dim3 grid(nblocks), block(512);
    // Up-projection (12 tile-columns at d = 768): two column groups x four tile-row blocks over the 8 XCDs keep each
    // XCD's W working set at 2.4 MB (L2-resident across rounds; it was re-fetched every round, PMC fetch 296 MB vs
    // 55 MB algorithmic) while A is fetched by two XCDs instead of one: 235 -> 228 us.  Measured and rejected:
    // four groups for the up-projection (neutral), two / four groups for the down-projection (A is the 201 MB
    // operand there: neutral / 175 -> 195 us).  Large launches only: every XCD cell needs workgroups of its own.
    // half-tile ring K loop (256 x 256 tiles, bf16, no conv): an iteration is two K-tiles, so K must be a multiple of 128
    const bool use_ring = nsplit > 1 ? false : p.conv ? (9 * (p.cv_cin >> 6)) % 2 == 0 : p.f8 ? (p.K >= 256 && p.K % 256 == 0)
                                                                                        : (p.K >= 128 && p.K % 128 == 0);      // (an even number of 128-byte K-tiles; conv: K = 9 cv_cin)
    (void)use_ring;
    static const bool f8_ring = !(getenv("TLD_F8_RING") && atoi(getenv("TLD_F8_RING")) == 0);        // A/B hook: 0 = the two-stage loop under the fp8 GEMMs (round 2 - 5)
    GemmParams pg = p;
    static const bool half_tail = !(getenv("TLD_GEMM_HALFTAIL") && atoi(getenv("TLD_GEMM_HALFTAIL")) == 0);     // test hook: tests/test_gpu_parity.py holds the row-split tail bitwise equal to the unsplit run
    pg.half_tail = half_tail ? 1 : 0;
    if ((epilogue == EPI_UP_DWCONV2 || epilogue == EPI_UP_DWCONV32 || epilogue == EPI_BIAS_BF16) && ntn % 2 == 0 && ntm >= 8 && nblocks == ncu && ncu % 8 == 0)
        pg.xcd_ngroups = 2;
    // (EPI_QKV_ATTN, round 4: its PMC traffic is 2.6 x algorithmic -- an XCD's round of 32 items wants 2.7 samples' A tiles + all 12 heads' weights,
    // 4.6 MB against 4 MB of L2 -- but two or four head groups over the XCDs left the kernel at 131.7 us: like the other GEMMs it is not fetch-bound)
#define TLD_L256P_(E, F8) TLD_L256P__(E, F8, false)
#define TLD_L256P_LAUNCH(E, F8, CV, RG)                                                               \
    do {                                                                                              \
        static PerDeviceOnce once;                                                                    \
        once.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<BN, E, F8, CV, RG>), \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds); });      \
        hipLaunchKernelGGL((gemm256p_kernel<BN, E, F8, CV, RG>), grid, block, lds, s, pg, nblocks);   \
    } while (0)
#define TLD_L256P__(E, F8, CV)                                                                        \
    do {                                                                                              \
        constexpr int lds = (F8) ? G::LDS_BYTES + 4096                                                \
                            : (((E) == EPI_UP_DWCONV2 || (E) == EPI_UP_DWCONV32) ? G::UPDW2_LDS                                   \
                            : ((E) == EPI_QKV_ATTN ? G::ATTN_LDS                                      \
                            : ((E) == EPI_QKV_LN ? G::QKVLN_LDS                                        \
                            : ((E) == EPI_BIAS_BF16 && BN != 384 ? G::PLAINLN_LDS : G::LDS_BYTES))));  /* (384-wide: 160 KB of stages, no LayerNorm-3 fold) */ \
        /* (fp8 + residual-add epilogue on the ring: 172 spilled registers, also with the lane-derived values re-materialised -- that one stays on the two-stage loop) */ \
        constexpr bool ring_ok = TLD_KLOOP_RING && BN == 256 && !((F8) && ((CV) || (E) == EPI_BIAS_RESID)); \
        if constexpr (ring_ok) {                                                                      \
            if (use_ring && (!(F8) || f8_ring)) TLD_L256P_LAUNCH(E, F8, CV, true); else TLD_L256P_LAUNCH(E, F8, CV, false); \
        } else {                                                                                      \
            TLD_L256P_LAUNCH(E, F8, CV, false);                                                       \
        }                                                                                             \
    } while (0)
#define TLD_L256P(E) TLD_L256P_(E, false)
    if (p.conv) {           // implicit 3x3 convolution (VAE decoder): 256- or 128-wide tiles, three epilogues
        if constexpr (BN == 256 || BN == 128) {
            switch (epilogue) {
                case EPI_F32: TLD_L256P__(EPI_F32, false, true); break;
                case EPI_BIAS_BF16: TLD_L256P__(EPI_BIAS_BF16, false, true); break;
                case EPI_BIAS_RESID: TLD_L256P__(EPI_BIAS_RESID, false, true); break;
                default: break;
            }
        }
    } else if (p.f8) {      // MX-fp8 operands: 256- or 128-wide tiles with four epilogues, 192-wide for the residual add
        if constexpr (BN == 256 || BN == 128) {
            switch (epilogue) {
                case EPI_F32: TLD_L256P_(EPI_F32, true); break;
                case EPI_QKV: TLD_L256P_(EPI_QKV, true); break;
                case EPI_BIAS_BF16: TLD_L256P_(EPI_BIAS_BF16, true); break;
                case EPI_BIAS_RESID: TLD_L256P_(EPI_BIAS_RESID, true); break;
                default: break;
            }
        } else if constexpr (BN == 192) {
            if (epilogue == EPI_BIAS_RESID) TLD_L256P_(EPI_BIAS_RESID, true);
        }
    } else if constexpr (BN == 384) {       // residual add (the down projection) and, for the training step's N = 768 linears, bias -> bf16
        if (epilogue == EPI_BIAS_BF16) TLD_L256P(EPI_BIAS_BF16); else TLD_L256P(EPI_BIAS_RESID);
    } else if constexpr (BN == 192) {
        if (epilogue == EPI_QKV_ATTN) TLD_L256P(EPI_QKV_ATTN); else TLD_L256P(EPI_BIAS_RESID);
    } else {
        switch (epilogue) {
            case EPI_F32: TLD_L256P(EPI_F32); break;
            case EPI_QKV: TLD_L256P(EPI_QKV); break;
            case EPI_QKV_LN: TLD_L256P(EPI_QKV_LN); break;
            case EPI_BIAS_BF16: TLD_L256P(EPI_BIAS_BF16); break;
            case EPI_BIAS_RESID: TLD_L256P(EPI_BIAS_RESID); break;
            case EPI_UP_DWCONV2: if constexpr (BN == 256) { TLD_L256P(EPI_UP_DWCONV2); } break;
            case EPI_UP_DWCONV32: if constexpr (BN == 256) { TLD_L256P(EPI_UP_DWCONV32); } break;
            default: break;
        }
    }
#undef TLD_L256P__
#undef TLD_L256P_LAUNCH
#undef TLD_L256P_
#undef TLD_L256P
}

}  // namespace

namespace {
// BN = 256 unless that leaves the last round of workgroups mostly empty on 256 CUs; then prefer the widest
// tile whose workgroup count is a whole number of rounds (192 for the residual epilogue), else 128.
int choose_bn(long M, long N, int epilogue, long K) {
    const long ntm = (M + 255) / 256;
    const long blocks256 = ntm * ((N + 255) / 256);
    // (a last round that is at least 85 % full counts as whole: 252 tiles on 256 CUs are not a reason to halve the tile)
    const long rounds256 = (blocks256 + 255) / 256;
    bool narrow = (N % 256 != 0) || (blocks256 * 100 < rounds256 * 256 * 85 && blocks256 < 3 * 256);
    // Round 6: a partly filled last round is cheap where the half-tile tail applies (gemm256p_kernel, HT_OK epilogues on the ring K loop): when the R
    // left-over tiles of an XCD's 32 workgroups satisfy 2 R <= 32, two workgroups share each of them and the launch costs q + ~0.6 tile times instead of q + 1.
    // 256-wide tiles then beat twice as many 128-wide ones (C3's block-0 QKV projection, 16 384 rows x 2304: 576 tiles = 2.25 rounds; 84 us as 1152 tiles of 128).
    // Results do not depend on the tile width (every output element sees the same K order), so the choice may follow the batch size.
    if (narrow && N % 256 == 0 && K >= 128 && K % 128 == 0 && (epilogue == EPI_F32 || epilogue == EPI_QKV || epilogue == EPI_QKV_LN || epilogue == EPI_BIAS_BF16) && blocks256 >= 256) {
        const long per_xcd = blocks256 / 8, R = per_xcd % 32;
        if (blocks256 % 8 == 0 && R > 0 && 2 * R <= 32) narrow = false;
    }
    int bn = narrow ? 128 : 256;
    if (narrow && epilogue == EPI_BIAS_RESID && N % 192 == 0 && (ntm * (N / 192)) % 256 == 0) bn = 192;
    // down projection at the bench size: 256 x 384 tiles make N = 768 ONE round of 256 workgroups (176 -> 155 us);
    // any other batch size of that width uses 192-wide tiles, never 128 / 256: both 192 and 384 give 96-column wave
    // tiles, which is what the LayerNorm-1 partial sums are defined on (results must not depend on the batch size)
    if (epilogue == EPI_BIAS_RESID && N % 192 == 0) bn = 192;
    if (epilogue == EPI_BIAS_RESID && N % 384 == 0 && (ntm * (N / 384)) % 256 == 0) bn = 384;
    if (epilogue == EPI_UP_DWCONV2 || epilogue == EPI_UP_DWCONV32) bn = 256;          // caller guarantees N % 256 == 0 and one 16x16 image per 256 rows
    if (epilogue == EPI_QKV_ATTN) return 192;       // one tile = one (sample, head): caller guarantees N = heads x 192, ntok == 256
    if (bn == 192 && (epilogue != EPI_BIAS_RESID || N % 192)) bn = 128;
    // N = 768 with the plain bias epilogue (the training step's five per layer): one round of 256 x 384 tiles at the training batch instead of
    // three rounds of 256 x 128 (0.71 -> 1.0 PFLOP/s)
    if (epilogue == EPI_BIAS_BF16 && N % 384 == 0 && N < 1536 && (ntm * (N / 384)) % 256 == 0) bn = 384;      // (callers with a LayerNorm-3 fold have N = 4 d >= 1536)
    if (bn == 384 && ((epilogue != EPI_BIAS_RESID && epilogue != EPI_BIAS_BF16) || N % 384)) bn = 128;
    if (bn != 256 && bn != 192 && bn != 384) bn = 128;
    return bn;
}
}  // namespace

int gemm_resid_stat_slots(int N) {
    return (N % 192 == 0 && N / 96 <= kLnSlots) ? N / 96 : 0;
}

void launch_gemm(const GemmParams& p_in, int epilogue, hipStream_t s) {
#ifdef TLD_DBG_EPI
    static const int dbg_epi_env = getenv("TLD_EPI_DBG") ? atoi(getenv("TLD_EPI_DBG")) : 0;
    GemmParams p = p_in;
    if (!p.dbg_epi) p.dbg_epi = dbg_epi_env;
#else
    const GemmParams& p = p_in;
#endif
    if (epilogue == EPI_UP_DWCONV2) {
        // small batches (one to five images per call): one 256 x 128 tile per workgroup on twice as many CUs finishes sooner than one 256 x 256 tile (17.7 vs 28.6 us at one
        // image); results are bitwise those of the 8-wave kernel (tld_updw.hip), so the choice may follow the batch size.  TLD_UPDW_SMALL=0: A/B and test hook.
        static const bool small_on = !(getenv("TLD_UPDW_SMALL") && atoi(getenv("TLD_UPDW_SMALL")) == 0);
        if (small_on && updw_pp_supported(p) && (long)(p.M / 256) * (p.N / 128) <= device_cu_count()) { launch_updw_pp(p, s); return; }
    }
    if (epilogue == EPI_F32 && p.ksplit > 1 && !p.f8 && !p.conv) {
        // the low-latency classes' split-K down projection: one 256 x 128 x (K / splits) item per 4-wave workgroup while the launch has at most two items per CU
        // (tld_updw.hip; the slices are bitwise those of the 8-wave two-stage kernel below).  TLD_SPLITK_SMALL=0: A/B and test hook.
        static const bool sk_on = !(getenv("TLD_SPLITK_SMALL") && atoi(getenv("TLD_SPLITK_SMALL")) == 0);
        // (two such workgroups fit a CU and, K loops only, share it well: eight images in class 1 -- 384 items -- 58.3 -> 48.8 ms per generate; the fused up-projection's epilogue does
        // not: 288 - 480 tiles on two workgroups per CU measured 1 - 2 ms slower than the 8-wave kernel, hence one tile per CU there)
        if (sk_on && splitk_pp_supported(p) && (long)(p.M / 256) * (p.N / 128) * p.ksplit <= 2L * device_cu_count()) { launch_splitk_pp(p, s); return; }
    }
    if (epilogue == EPI_BIAS_RESID && !p.f8 && !p.conv) {
        // the default class's down projection at small batch: 64 x 192 or 128 x 192 tiles on 4-wave workgroups while the launch has at most one tile per CU (tld_updw.hip; bitwise the
        // 8-wave 192- / 384-wide kernels below).  TLD_DOWN_SMALL=0: A/B and test hook.
        static const bool dn_on = !(getenv("TLD_DOWN_SMALL") && atoi(getenv("TLD_DOWN_SMALL")) == 0);
        if (dn_on && down_pp_supported(p) && down_pp_fits(p)) { launch_down_pp(p, s); return; }
    }
    int bn = choose_bn(p.M, p.N, epilogue, p.K);
    if (p.conv) {           // 256-wide tiles when the width allows and they fill the chip, else 128
        const long ntm = (p.M + 255) / 256;
        bn = (p.N % 256 == 0 && ntm * (p.N / 256) >= 192) ? 256 : 128;
    }
    if (p.f8) {             // the fp8 kernel is instantiated for 256 / 128 (all epilogues) and 192 (residual add: N = 768 in whole rounds)
        const long ntm = (p.M + 255) / 256;
        // (round 6: where 256-wide tiles fill whole rounds they beat the 192-wide ones -- the down projection at C4, 65 536 rows: 768 tiles in 3 rounds against 1024 in 4,
        // 207.7 -> 174.8 us same-box; results do not depend on the tile width, the LayerNorm-1 partial sums are not taken in fp8 mode)
        if (epilogue == EPI_BIAS_RESID && p.N % 256 == 0 && (ntm * (p.N / 256)) % 256 == 0) bn = 256;
        else if (epilogue == EPI_BIAS_RESID && p.N % 192 == 0 && ((ntm * (p.N / 192)) % 256 == 0 || p.N % 256 != 0)) bn = 192;
        else bn = (p.N % 256 == 0) ? 256 : 128;
    }
    if (epilogue == EPI_F32 && p.ksplit > 1 && !p.f8 && !p.conv) bn = 128;     // split-K: the narrow tile has no ring instantiation to fall into and gives the most work items
    // (A column-split QKV launch -- 8 tile-columns of 256 as 4 whole rounds + the 9th as 128-wide tiles -- was
    // measured: 109.8 + 27.8 us vs 134 us for the single 4.5-round launch; a one-round launch pays ~12 us of
    // ramp/drain, so the half-empty fifth round is the cheaper tail.)
    if (bn == 384) launch256p<384>(p, epilogue, s);
    else if (bn == 192) launch256p<192>(p, epilogue, s);
    else if (bn == 128) launch256p<128>(p, epilogue, s);
    else launch256p<256>(p, epilogue, s);
}

void launch_gemm_tn(const GemmParams& p, hipStream_t s) {
    using G = G256P<256>;
    const int ntiles = (p.M / 256) * (p.N / 256);
    const int ncu = device_cu_count();
    const int nblocks = ntiles < ncu ? ntiles : ncu;
    static PerDeviceOnce once;
    once.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<256, EPI_F32, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::LDS_BYTES); });
    hipLaunchKernelGGL((gemm256p_kernel<256, EPI_F32, false, false, false, true>), dim3(nblocks), dim3(512), G::LDS_BYTES, s, p, nblocks);
}

}  // namespace tld
