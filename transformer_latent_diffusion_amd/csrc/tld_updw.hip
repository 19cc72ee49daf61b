// tld_updw.hip -- the MLP up-projection with the depthwise 3x3 + GELU epilogue (EPI_UP_DWCONV2, 16 x 16 token grid) on 256 x 128 tiles, 4 waves per workgroup:
// the SMALL-BATCH form of gemm256p_kernel<256, EPI_UP_DWCONV2, ring> (tld_gemm.hip), taken by launch_gemm while the launch has no more 256 x 128 tiles than the chip
// has CUs (one to five images per generate call: the serving edge, tld/app.py:48-65).
//
// Replaces tld/transformer_blocks.py:95-103 (Conv2d 1x1 -> depthwise Conv2d 3x3 -> GELU of MLPSepConv) like the 8-wave kernel does.  At those sizes a launch is ONE tile
// per workgroup and its time is the latency of that tile: twice as many workgroups of half the width, each with a K loop that needs no partner wave on its SIMD,
// finish a tile in 17.7 us against 28.6 (one image, MI355X).
//   * results: every output element is accumulated over K in the order of the 8-wave kernel (K-tile by K-tile, k-slice by k-slice) and goes through the same
//     epilogue expressions: BITWISE equal to it (tests/test_gpu_dwconv.py), so the tile shape may follow the batch size.
//   * LDS per workgroup (79 KiB): a 1.5 K-tile operand ring -- three 16-KiB A half-tile slots and three 8-KiB W half-tile slots -- which the epilogue's token-pair
//     image [128 pairs][128 channels] dwords (64 KiB) overlays; behind it the zero pair-row and the LayerNorm-3 side tables.
//   * K loop: four phases per K-tile as in kloop_ring (quadrants (0,0) (0,1) (1,1) (1,0) of the wave's 128 x 64 output, 8 MFMAs each), one wave per SIMD: the
//     fragments of phase p + 1 are read DURING the MFMAs of phase p into a second register set, ds_reads and tile DMA are interleaved between the MFMAs, ONE s_barrier
//     per phase, counted vmcnt waits (14 / 8 / 16 / 10 pieces may stay in flight at the four phase ends), never 0 inside a tile's main loop.  61 % matrix-pipe duty alone.
//   * history: written in round 6 to put TWO such workgroups on every CU at the bench size, so that one's epilogue would run under the other's K loop -- bitwise equal and
//     7 % slower there (the younger workgroup's VALU stream starves beside the older one's MFMAs: profiles/r06_updw_two_workgroups_experiment.txt, where the timestamp
//     instrumentation of that experiment is kept: tools/ubench/updw_two_workgroups_experiment.hip.txt).
#include "tld_common.h"
#include <cstdlib>
#include <type_traits>

namespace tld {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct PP {
    static constexpr int A_SLOT = 16384, B_SLOT = 8192;                  // half-tiles: 128 rows x 64 K (A), 64 rows x 64 K (W); 128-byte rows, swizzled as in tld_gemm.hip
    static constexpr int A_RING = 0, B_RING = 3 * A_SLOT;
    static constexpr int RING_BYTES = 3 * (A_SLOT + B_SLOT);             // 72 KiB
    static constexpr int PITCH = 512, IMG_BYTES = 128 * PITCH;            // token-pair image: [128 pairs][128 channels] dwords
    static constexpr int ZROW = RING_BYTES;                              // 8 all-zero pair-rows (one image row), never overwritten by the ring
    static constexpr int RS = ZROW + 8 * PITCH;                          // (mean, rstd) of the tile's 256 rows
    static constexpr int CB = RS + 256 * 8;                              // c1[128] | bias[128] of the tile's columns (fp32)
    static constexpr int LDS = CB + 1024;                                // 80 896 B: two workgroups per CU
};
static_assert(PP::IMG_BYTES <= PP::RING_BYTES, "the image overlays the ring");
static_assert(2 * PP::LDS <= 160 * 1024, "two workgroups per CU");

template <int N> __device__ __forceinline__ void pp_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N < 0, "unsupported vmcnt");
}

// MODE 0: EPI_UP_DWCONV2 (the fused depthwise epilogue).  MODE 1: EPI_F32 with GemmParams::ksplit -- the low-latency classes' split-K down projection: the tile list is
// `ksplit` copies of the (m, n) grid, copy s multiplying the K range [s K, (s + 1) K) of both operands into fp32 slice s (c_f32 + s M ldc); same K order per output element as
// gemm256p_kernel<128, EPI_F32>'s two-stage loop, so the slices are bitwise the same.
template <int MODE>
__global__ __launch_bounds__(256, 2) void updw_pp_kernel(GemmParams p, int nblocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;                   // 2 x 2 waves, wave tile 128 tokens x 64 channels

    // ---- static schedule (gemm256p_kernel's): XCD x (= block id % 8) owns a contiguous run of the row-major tile order -- or, with xcd_ngroups = G, the cell
    // (tile-row block, tile-column group) of an (8 / G) x G grid -- and its workgroups take that run round-robin
    const int ntn = p.N >> 7, ntm = p.M >> 8;
    const int nsplit = (MODE == 1 && p.ksplit > 1) ? p.ksplit : 1;
    const int ntiles = ntm * ntn * nsplit;
    const int bid = blockIdx.x, xcd = bid & 7, lidx = bid >> 3;
    const int per_xcd_blocks = (nblocks + 7 - xcd) / 8;
    const int G2 = p.xcd_ngroups > 1 ? p.xcd_ngroups : 1;
    const int xn = xcd % G2, xm = xcd / G2, XM = 8 / G2, gcols = ntn / G2;
    const int r0 = (int)((long)ntm * xm / XM), r1 = (int)((long)ntm * (xm + 1) / XM);
    const int q8 = ntiles >> 3, rr8 = ntiles & 7;
    const int xbase = xcd < rr8 ? xcd * (q8 + 1) : rr8 * (q8 + 1) + (xcd - rr8) * q8;
    const int xcount = G2 > 1 ? (r1 - r0) * gcols : q8 + (xcd < rr8 ? 1 : 0);
    const int my_tiles = lidx < xcount ? (xcount - lidx + per_xcd_blocks - 1) / per_xcd_blocks : 0;
    if (my_tiles == 0) return;
    int sp_dec = 0;                                          // MODE 1: K-split of the tile tile_coords() decoded last
    auto tile_coords = [&](int i, int& m0, int& n0) {
        const int t = lidx + i * per_xcd_blocks;
        if (G2 > 1) {
            const int tm = t / gcols;
            m0 = (r0 + tm) << 8;
            n0 = (xn * gcols + (t - tm * gcols)) << 7;
        } else {
            int tile = xbase + t;
            if constexpr (MODE == 1) {
                const int sp = tile / (ntm * ntn);
                tile -= sp * (ntm * ntn);
                sp_dec = sp;
            }
            const int tm = tile / ntn;
            m0 = tm << 8;
            n0 = (tile - tm * ntn) << 7;
        }
    };

    // zero pair-row (behind the ring: written once)
    if constexpr (MODE == 0) *reinterpret_cast<u32x4*>(smem + PP::ZROW + tid * 16) = u32x4{0u, 0u, 0u, 0u};

    const int nk = p.K >> 6;                                 // 64-element K-tiles (even, >= 4: checked by the launcher)
    const unsigned lda2 = (unsigned)p.lda * 2u, ldw2 = (unsigned)p.ldw * 2u;

    // fragment read offsets inside a half-tile image: row (wm 64 | wn 32) + l31, 16-byte chunk (2 ks + hi) ^ swizzle; ks enters as an XOR of ks << 5
    unsigned ra0, rb0;
    {
        int l31v = lane & 31, hiv = lane >> 5;
        asm volatile("" : "+v"(l31v), "+v"(hiv));
        const int sw = (l31v >> 1) & 7;
        ra0 = (unsigned)((wm * 64 + l31v) * 128 + ((hiv ^ sw) << 4));
        rb0 = (unsigned)((wn * 32 + l31v) * 128 + ((hiv ^ sw) << 4));
    }

    for (int it = 0; it < my_tiles; ++it) {
        int m0, n0;
        tile_coords(it, m0, n0);
        const size_t kb = (size_t)sp_dec * (size_t)p.K * 2;       // MODE 1: K byte offset of this tile's split (0 otherwise)

        // ---- DMA source offsets.  A half-tile h, piece q2 of this wave: image rows r = (wid 4 + q2) 8 + (lane >> 3) = tile rows (r >> 6) 128 + h 64 + (r & 63);
        // only the parity of q2 reaches the swizzle, everything else is a scalar offset: two registers per operand
        unsigned vA[2], vB[2];
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int r = (wid * 4 + o) * 8 + (ln >> 3);
                const unsigned c16 = (unsigned)(((ln & 7) ^ ((r >> 1) & 7)) * 16);
                const int ga = m0 + (r >> 6) * 128 + (r & 63);
                unsigned va = __umul24((unsigned)ga, lda2) + c16;
                const int rb_ = (wid * 2 + o) * 8 + (ln >> 3);
                const unsigned c16b = (unsigned)(((ln & 7) ^ ((rb_ >> 1) & 7)) * 16);
                const int gb = n0 + (rb_ >> 5) * 64 + (rb_ & 31);
                unsigned vb = __umul24((unsigned)gb, ldw2) + c16b;
                asm volatile("" : "+v"(va), "+v"(vb));
                vA[o] = va; vB[o] = vb;
            }
        }
        // stage half-tile h of K-tile t into ring slot `slot`: pieces q2 = 0 .. 3 (A) / 0 .. 1 (W) of this wave; [lo, hi) selects a sub-range (interleaving)
        auto stageA = [&](int h, int t, int slot, int lo, int hi_) {
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                if (q2 < lo || q2 >= hi_) continue;
                const char* base = reinterpret_cast<const char*>(p.A) + kb + (size_t)t * 128 + (size_t)(h * 64 + (q2 >> 1) * 16) * lda2;
                asm volatile("" : "+s"(base));
                unsigned o = vA[q2 & 1];
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_global_load_lds((gptr_t)(base + o), (lptr_t)(smem + PP::A_RING + slot * PP::A_SLOT + (wid * 4 + q2) * 1024), 16, 0, 0);
            }
        };
        auto stageB = [&](int h, int t, int slot) {
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const char* base = reinterpret_cast<const char*>(p.W) + kb + (size_t)t * 128 + (size_t)(h * 32) * ldw2;
                asm volatile("" : "+s"(base));
                unsigned o = vB[q2];
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_global_load_lds((gptr_t)(base + o), (lptr_t)(smem + PP::B_RING + slot * PP::B_SLOT + (wid * 2 + q2) * 1024), 16, 0, 0);
            }
        };
        auto slot3 = [](int k) { return k - (k / 3) * 3; };

        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // ---- prologue: side tables, K-tile 0 and the first half of K-tile 1 (all six slots)
        if constexpr (MODE == 0) {
        if (p.row_stats && wid < 2) {
            const char* src = reinterpret_cast<const char*>(p.row_stats + m0) + wid * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + PP::RS + wid * 1024), 16, 0, 0);
        }
        if ((wid == 3 || (wid == 2 && p.row_stats)) && lane < 32) {      // wave 2: c1, wave 3: bias of the tile's 128 columns
            const float* src = (wid == 2 ? p.ln_c1 : p.bias) + n0 + lane * 4;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + PP::CB + (wid - 2) * 512), 16, 0, 0);
        }
        }
        stageB(0, 0, 0); stageA(0, 0, 0, 0, 4); stageB(1, 0, 1); stageA(1, 0, 1, 0, 4); stageB(0, 1, 2); stageA(0, 1, 2, 0, 4);
        pp_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();

        bf16x8 pa0[2][4], pa1[2][4], fb0[4], fb1[4];
        auto readA = [&](bf16x8 (&dst)[2][4], int slot, int ii_lo, int ii_hi) {
            const unsigned base = (unsigned)(PP::A_RING + slot * PP::A_SLOT);
            unsigned a = ra0;
            asm volatile("" : "+v"(a));                 // (recomputed per read group: hoisted out of the K loop these addresses cost 12 registers per operand)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                if (ii < ii_lo || ii >= ii_hi) continue;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dst[ii][ks] = *reinterpret_cast<const bf16x8*>(smem + ((a ^ (unsigned)(ks << 5)) + base) + ii * 4096);
            }
        };
        auto readB = [&](bf16x8 (&dst)[4], int slot, int lo, int hi_) {
            const unsigned base = (unsigned)(PP::B_RING + slot * PP::B_SLOT);
            unsigned b = rb0;
            asm volatile("" : "+v"(b));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < lo || ks >= hi_) continue;
                dst[ks] = *reinterpret_cast<const bf16x8*>(smem + ((b ^ (unsigned)(ks << 5)) + base));
            }
        };
        readA(pa0, 0, 0, 2);
        readB(fb0, 0, 0, 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stageA(1, 1, 0, 0, 4);                          // A1 of K-tile 1 takes the slot A0 of K-tile 0 was just read from

        // one phase = one C quadrant x K = 64: chunk c = the two MFMAs of k-slice c (ii = 0, 1) followed by side work `side(c)` (fragment reads for the NEXT phase,
        // tile DMA), pinned in this order
        auto phase = [&](auto qac, auto qbc, const bf16x8 (&a)[2][4], const bf16x8 (&b)[4], auto&& side) {
            constexpr int qa = decltype(qac)::value, qb = decltype(qbc)::value;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
                    acc[qa * 2 + ii][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ii][ks], b[ks], acc[qa * 2 + ii][qb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                side(ks);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        // K-tile t.  PAR = t & 1 decides which register set holds this K-tile's B0 (bx) and which its B1 (by); TAIL: 0 = main loop, 1 = second-to-last K-tile
        // (stages only B1 of the last one), 2 = last (stages and prefetches nothing)
        auto ktile = [&](int t, bf16x8 (&bx)[4], bf16x8 (&by)[4], auto tailc) {
            constexpr int TAIL = decltype(tailc)::value;
            const int k2 = 2 * t;
            const int s0 = slot3(k2), s1 = slot3(k2 + 1), s2 = slot3(k2 + 2);
            // (t, 0): quadrant (0, 0) = A0 x B0.  read B1(t); stage B1(t + 1) -> the slot B0(t) left
            phase(C0{}, C0{}, pa0, bx, [&](int c) {
                if (c == 0) readB(by, s1, 0, 4);
                if (c == 1 && TAIL < 2) stageB(1, t + 1, s0);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (TAIL < 2) pp_wait_vmcnt<14>();
            __builtin_amdgcn_s_barrier();
            // (t, 1): quadrant (0, 1) = A0 x B1.  read A1(t); stage B0(t + 2) -> the slot B1(t) left
            phase(C0{}, C1{}, pa0, by, [&](int c) {
                if (c == 0) readA(pa1, s1, 0, 1);
                if (c == 1) readA(pa1, s1, 1, 2);
                if (c == 2 && TAIL == 0) stageB(0, t + 2, s1);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (TAIL == 0) pp_wait_vmcnt<8>(); else if constexpr (TAIL == 1) pp_wait_vmcnt<6>();
            __builtin_amdgcn_s_barrier();
            // (t, 2): quadrant (1, 1) = A1 x B1.  read A0(t + 1); stage A0(t + 2) -> the slot A1(t) left
            phase(C1{}, C1{}, pa1, by, [&](int c) {
                if (TAIL < 2) {
                    if (c == 0) readA(pa0, s2, 0, 1);
                    if (c == 1) readA(pa0, s2, 1, 2);
                }
                if (TAIL == 0) {
                    if (c == 2) stageA(0, t + 2, s1, 0, 2);
                    if (c == 3) stageA(0, t + 2, s1, 2, 4);
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (TAIL == 0) pp_wait_vmcnt<16>(); else if constexpr (TAIL == 1) pp_wait_vmcnt<10>();
            __builtin_amdgcn_s_barrier();
            // (t, 3): quadrant (1, 0) = A1 x B0.  read B0(t + 1) into the B1 registers (the next K-tile's bx); stage A1(t + 2) -> the slot A0(t + 1) left
            phase(C1{}, C0{}, pa1, bx, [&](int c) {
                if (TAIL < 2 && c == 0) readB(by, s2, 0, 4);
                if (TAIL == 0) {
                    if (c == 1) stageA(1, t + 2, s2, 0, 2);
                    if (c == 2) stageA(1, t + 2, s2, 2, 4);
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (TAIL == 0) pp_wait_vmcnt<10>(); else if constexpr (TAIL == 1) pp_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        };
        for (int t = 0; t < nk - 2; t += 2) {
            ktile(t, fb0, fb1, C0{});
            ktile(t + 1, fb1, fb0, C0{});
        }
        ktile(nk - 2, fb0, fb1, C1{});
        ktile(nk - 1, fb1, fb0, std::integral_constant<int, 2>{});

        if constexpr (MODE == 1) {
            // ---- fp32 slice of this split (natural MFMA order: a lane owns a column, a register quad four consecutive rows)
            int tide = tid;
            asm volatile("" : "+v"(tide));
            const int l31 = tide & 31, hi = (tide >> 5) & 1;
            float* cs = p.c_f32 + (size_t)sp_dec * p.M * p.ldc;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * 128 + i * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                        cs[(size_t)row * p.ldc + col] = acc[i][j][r];
                    }
                }
        } else {
        // ---- epilogue (the arithmetic of gemm256p_kernel's EPI_UP_DWCONV2 branch on a 128-channel image): bf16(rstd (acc - mean c1) + bias) as token-pair dwords into
        // LDS, then the depthwise 3x3 + GELU from the image: a thread owns a channel quad and two adjacent image rows
        char* H = smem;
        const bool ln3 = p.row_stats != nullptr;
        int tide = tid;
        asm volatile("" : "+v"(tide));                   // (lane-derived values of the epilogue are rebuilt per tile, not carried through the K loop)
        const int l31 = tide & 31, hi = (tide >> 5) & 1;
        float cst[2], bst[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cl = wn * 64 + j * 32 + l31;
            bst[j] = *reinterpret_cast<const float*>(smem + PP::CB + 512 + cl * 4);
            cst[j] = ln3 ? *reinterpret_cast<const float*>(smem + PP::CB + cl * 4) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 sv[8];
            if (ln3) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int tok0 = wm * 128 + i * 32 + 8 * rq + 4 * hi;
                    sv[2 * rq] = *reinterpret_cast<const float4*>(smem + PP::RS + tok0 * 8);
                    sv[2 * rq + 1] = *reinterpret_cast<const float4*>(smem + PP::RS + tok0 * 8 + 16);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) sv[q] = make_float4(0.f, 1.f, 0.f, 1.f);
            }
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int tok0 = wm * 128 + i * 32 + 8 * rq + 4 * hi;          // 4 consecutive tokens
                const float4 s01 = sv[2 * rq], s23 = sv[2 * rq + 1];
                const float rs0 = s01.y, rs1 = s01.w, rs2 = s23.y, rs3 = s23.w;
                const float nm0 = -s01.y * s01.x, nm1 = -s01.w * s01.z, nm2 = -s23.y * s23.x, nm3 = -s23.w * s23.z;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bf16x2 lo, hi2;
                    lo[0] = (bf16)fmaf(rs0, acc[i][j][rq * 4 + 0], fmaf(nm0, cst[j], bst[j]));
                    lo[1] = (bf16)fmaf(rs1, acc[i][j][rq * 4 + 1], fmaf(nm1, cst[j], bst[j]));
                    hi2[0] = (bf16)fmaf(rs2, acc[i][j][rq * 4 + 2], fmaf(nm2, cst[j], bst[j]));
                    hi2[1] = (bf16)fmaf(rs3, acc[i][j][rq * 4 + 3], fmaf(nm3, cst[j], bst[j]));
                    char* dst = H + (tok0 >> 1) * PP::PITCH + (wn * 64 + j * 32 + l31) * 4;
                    *reinterpret_cast<bf16x2*>(dst) = lo;
                    *reinterpret_cast<bf16x2*>(dst + PP::PITCH) = hi2;
                }
            }
        }
        __builtin_amdgcn_s_barrier();
        {
            const int cq = tide & 31;                            // channel quad of the tile's 128 channels
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
            const int w2 = tide >> 5;                            // output image rows 2 w2 and 2 w2 + 1
            const char* rb[4];                                   // window rows 2 w2 - 1 .. 2 w2 + 2 (8 pair-columns each)
            rb[0] = (w2 == 0 ? smem + PP::ZROW : H + (2 * w2 - 1) * 8 * PP::PITCH) + cq * 16;
            rb[1] = H + (2 * w2) * 8 * PP::PITCH + cq * 16;
            rb[2] = H + (2 * w2 + 1) * 8 * PP::PITCH + cq * 16;
            rb[3] = (w2 == 7 ? smem + PP::ZROW : H + (2 * w2 + 2) * 8 * PP::PITCH) + cq * 16;
            auto ld = [&](int q, u32x4 (&c)[4]) {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) c[k4] = *reinterpret_cast<const u32x4*>(rb[k4] + q * PP::PITCH);
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
                    e0 = gelu_erf_fast2_half(e0); e1 = gelu_erf_fast2_half(e1);
                    o0 = gelu_erf_fast2_half(o0); o1 = gelu_erf_fast2_half(o1);
                    bf16x4 oe, oo;
                    oe[0] = (bf16)e0[0]; oe[1] = (bf16)e0[1]; oe[2] = (bf16)e1[0]; oe[3] = (bf16)e1[1];
                    oo[0] = (bf16)o0[0]; oo[1] = (bf16)o0[1]; oo[2] = (bf16)o1[0]; oo[3] = (bf16)o1[1];
                    TLD_STORE(reinterpret_cast<bf16x4*>(dst0 + ((size_t)rr * 16 + 2 * q) * p.ldo), oe);
                    TLD_STORE(reinterpret_cast<bf16x4*>(dst0 + ((size_t)rr * 16 + 2 * q + 1) * p.ldo), oo);
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
        }
        // every wave is done with the image before the next tile's operands overwrite it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}


// =================================================================================================
// EPI_BIAS_RESID (the MLP down projection, x += hid Wdown^T + b, + the LayerNorm-1 partial sums of the next block: tld/transformer_blocks.py:104,138) for SMALL launches of the
// DEFAULT class: 128 x 192 tiles, 4 waves as 2 (M) x 2 (N) with the 64 x 96 wave tiles -- and, line by line, the per-wave epilogue -- of gemm256p_kernel<192, EPI_BIAS_RESID>.
// At one image that kernel has 8 work items of 48 K-steps each (69 us per launch, half of a default-class step); here the same product is 16 items whose K loop needs no
// partner wave: a 3-K-tile LDS ring (40 KiB per K-tile: A 128 rows + W 192 rows of 128 bytes), two phases per K-tile (k-slices 0-1 | 2-3: 12 MFMAs each), the fragments of the
// next phase read during the MFMAs of the current one, one s_barrier per phase, K-tile t + 2 staged during K-tile t (vmcnt(4) at the one wait per K-tile).
// Every output element is accumulated over K in the 8-wave kernel's order and finished by the same expressions: bitwise equal to it (tests/test_gpu_configs.py).
struct DP {
    static constexpr int A_BYTES = 128 * 128, B_BYTES = 192 * 128, KT_BYTES = A_BYTES + B_BYTES;      // one K-tile: 40 KiB
    static constexpr int LDS = 3 * KT_BYTES;                                                            // 120 KiB
    static constexpr int SCRATCH = 4608;                                                                // per-wave epilogue scratch (one 32 x 32 fp32 tile, 144-byte pitch)
};

// TM = 32-row MFMA tiles per wave: 2 = 128 x 192 tiles (64 x 96 wave tiles), 1 = 64 x 192 tiles (32 x 96): twice the work items of half the K-tile time while they still fit one per CU
template <int TM>
__global__ __launch_bounds__(256, 2) void down_pp_kernel(GemmParams p, int nblocks) {
    constexpr int BM = 64 * TM, NA = 2 * TM;                 // tile rows; A pieces (8 rows each) per wave and K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int ntn = p.N / 192, ntm = p.M / BM, ntiles = ntm * ntn;
    const int nk = p.K >> 6;                                 // 64-element K-tiles (>= 3)
    const unsigned lda2 = (unsigned)p.lda * 2u, ldw2 = (unsigned)p.ldw * 2u;
    unsigned ra0, rb0;
    {
        int l31v = lane & 31, hiv = lane >> 5;
        asm volatile("" : "+v"(l31v), "+v"(hiv));
        const int sw = (l31v >> 1) & 7;
        ra0 = (unsigned)((wm * 32 * TM + l31v) * 128 + ((hiv ^ sw) << 4));
        rb0 = (unsigned)(DP::A_BYTES + (wn * 96 + l31v) * 128 + ((hiv ^ sw) << 4));
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += nblocks) {
        const int tm = tile / ntn;
        const int m0 = tm * BM, n0 = (tile - tm * ntn) * 192;
        // DMA source offsets: piece q of this wave = image rows (wid NA + q) 8 + (lane >> 3) of A (NA pieces), (wid 6 + q) 8 + (lane >> 3) of W (6 pieces); only the parity of q
        // reaches the swizzle, the rest is a scalar row offset
        unsigned vA[2], vB[2];
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int r = (wid * NA + o) * 8 + (ln >> 3);
                unsigned va = __umul24((unsigned)(m0 + r), lda2) + (unsigned)(((ln & 7) ^ ((r >> 1) & 7)) * 16);
                const int rb_ = (wid * 6 + o) * 8 + (ln >> 3);
                unsigned vb = __umul24((unsigned)(n0 + rb_), ldw2) + (unsigned)(((ln & 7) ^ ((rb_ >> 1) & 7)) * 16);
                asm volatile("" : "+v"(va), "+v"(vb));
                vA[o] = va; vB[o] = vb;
            }
        }
        auto stageA = [&](int t, int slot, int lo, int hi_) {
#pragma unroll
            for (int q2 = 0; q2 < NA; ++q2) {
                if (q2 < lo || q2 >= hi_) continue;
                const char* base = reinterpret_cast<const char*>(p.A) + (size_t)t * 128 + (size_t)((q2 >> 1) * 16) * lda2;
                asm volatile("" : "+s"(base));
                unsigned o = vA[q2 & 1];
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_global_load_lds((gptr_t)(base + o), (lptr_t)(smem + slot * DP::KT_BYTES + (wid * NA + q2) * 1024), 16, 0, 0);
            }
        };
        auto stageB = [&](int t, int slot, int lo, int hi_) {
#pragma unroll
            for (int q2 = 0; q2 < 6; ++q2) {
                if (q2 < lo || q2 >= hi_) continue;
                const char* base = reinterpret_cast<const char*>(p.W) + (size_t)t * 128 + (size_t)((q2 >> 1) * 16) * ldw2;
                asm volatile("" : "+s"(base));
                unsigned o = vB[q2 & 1];
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_global_load_lds((gptr_t)(base + o), (lptr_t)(smem + slot * DP::KT_BYTES + DP::A_BYTES + (wid * 6 + q2) * 1024), 16, 0, 0);
            }
        };
        f32x16 acc[TM][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // fragments of one phase: k-slices 2 h, 2 h + 1 of a K-tile -- A [row tile][k-slice of the pair], W [column tile][k-slice of the pair]
        bf16x8 fa[2][TM][2], fb[2][3][2];                      // [buffer][tile][k-slice of the pair]
        auto read_frags = [&](auto bufc, int slot, int h) {
            constexpr int bf = decltype(bufc)::value;
            const unsigned base = (unsigned)(slot * DP::KT_BYTES);
            unsigned a = ra0, b = rb0;
            asm volatile("" : "+v"(a), "+v"(b));
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned kx = (unsigned)((2 * h + kk) << 5);
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[bf][i][kk] = *reinterpret_cast<const bf16x8*>(smem + ((a ^ kx) + base) + i * 4096);
#pragma unroll
                for (int j = 0; j < 3; ++j) fb[bf][j][kk] = *reinterpret_cast<const bf16x8*>(smem + ((b ^ kx) + base) + j * 4096);
            }
        };
        // swapped operand order (a lane owns a token row, a register quad four consecutive columns), as the 8-wave kernel; side(g) = the fragment reads / DMA pieces issued
        // after MFMA group g (3 MFMAs), pinned in this order so that their issue time hides under the matrix pipe
        auto mma = [&](auto bufc, auto&& side) {
            constexpr int bf = decltype(bufc)::value;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[bf][j][kk], fa[bf][i][kk], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    side(kk * TM + i);
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>;
        auto slot3 = [](int k) { return k - (k / 3) * 3; };

        // ---- prologue: K-tiles 0 and 1
        stageA(0, 0, 0, NA); stageB(0, 0, 0, 6); stageA(1, 1, 0, NA); stageB(1, 1, 0, 6);
        pp_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        read_frags(B0{}, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int t = 0; t < nk; ++t) {
            const int s0 = slot3(t), s1 = slot3(t + 1), s2 = slot3(t + 2);
            const bool st = t + 2 < nk;
            // phase 0: k-slices 0, 1 of K-tile t from buffer 0; read k-slices 2, 3 into buffer 1; stage A of K-tile t + 2 (its slot held K-tile t - 1, whose reads ended a phase ago)
            mma(B0{}, [&](int g) {                          // (2 TM groups of 3 MFMAs per phase)
                if (g == 0) read_frags(B1{}, s0, 1);
                if (st && g == 1) stageA(t + 2, s2, 0, 2);
                if (st && TM == 2 && g == 2) stageA(t + 2, s2, 2, 4);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (st) pp_wait_vmcnt<NA>(); else pp_wait_vmcnt<0>();          // K-tile t + 1 has landed (own pieces) ...
            __builtin_amdgcn_s_barrier();                                  // ... everybody's
            // phase 1: k-slices 2, 3 from buffer 1; read k-slices 0, 1 of K-tile t + 1 into buffer 0; stage W of K-tile t + 2
            mma(B1{}, [&](int g) {
                if (g == 0 && t + 1 < nk) read_frags(B0{}, s1, 0);
                if constexpr (TM == 2) {
                    if (st && g == 1) stageB(t + 2, s2, 0, 2);
                    if (st && g == 2) stageB(t + 2, s2, 2, 4);
                    if (st && g == 3) stageB(t + 2, s2, 4, 6);
                } else {
                    if (st && g == 0) stageB(t + 2, s2, 0, 3);
                    if (st && g == 1) stageB(t + 2, s2, 3, 6);
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }

        // ---- epilogue: gemm256p_kernel's EPI_BIAS_RESID branch for a 64 x 96 wave tile (G::TM = 2, G::TN = 3, G::WCOLS = 96)
        int lanev = lane;
        asm volatile("" : "+v"(lanev));
        const int l31 = lanev & 31, hi = lanev >> 5;
        char* ws = smem + wid * DP::SCRATCH;
        const int row0 = m0 + wm * 32 * TM, col0 = n0 + wn * 96;
        constexpr int P = 32 * 4 + 16;
        resid4_t rnx[4];
        auto rfetch1 = [&](int i2, int j2, int itr) {
            const int idx = itr * 64 + lanev;
            const int row = row0 + i2 * 32 + (idx >> 3), col = col0 + j2 * 32 + (idx & 7) * 4;
            rnx[itr] = rs_raw4(p.resid + (size_t)row * p.ldr + col);
        };
#pragma unroll
        for (int itr = 0; itr < 4; ++itr) rfetch1(0, 0, itr);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int cl = 8 * rq + 4 * hi;
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + col0 + j * 32 + cl);
                    float4 v;
                    v.x = acc[i][j][rq * 4 + 0] + bv.x; v.y = acc[i][j][rq * 4 + 1] + bv.y;
                    v.z = acc[i][j][rq * 4 + 2] + bv.z; v.w = acc[i][j][rq * 4 + 3] + bv.w;
                    *reinterpret_cast<float4*>(ws + l31 * P + cl * 4) = v;
                }
#pragma unroll
                for (int itr = 0; itr < 4; ++itr) {
                    const int idx = itr * 64 + lanev;
                    const int rl = idx >> 3, ch = idx & 7;
                    const float4 v = *reinterpret_cast<const float4*>(ws + rl * P + ch * 16);
                    const int row = row0 + i * 32 + rl, col = col0 + j * 32 + ch * 4;
                    resid_t* px = p.resid + (size_t)row * p.ldr + col;
                    float4 o = rs_widen4(rnx[itr]);
                    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                    rs_store4(px, o);
                    const float r0 = rs_round(o.x), r1 = rs_round(o.y), r2 = rs_round(o.z), r3 = rs_round(o.w);
                    ps[itr] += (r0 + r1) + (r2 + r3);
                    pq[itr] = fmaf(r0, r0, fmaf(r1, r1, fmaf(r2, r2, fmaf(r3, r3, pq[itr]))));
                    if (j + 1 < 3) rfetch1(i, j + 1, itr); else if (i + 1 < TM) rfetch1(i + 1, 0, itr);
                }
            }
            if (p.stats_out) {
                const int slot = col0 / 96;
#pragma unroll
                for (int itr = 0; itr < 4; ++itr) {
                    float a = ps[itr], q2 = pq[itr];
                    a = dpp_add<0xB1>(a); q2 = dpp_add<0xB1>(q2);
                    a = dpp_add<0x4E>(a); q2 = dpp_add<0x4E>(q2);
                    a = dpp_add<0x141>(a); q2 = dpp_add<0x141>(q2);
                    const int row = row0 + i * 32 + itr * 8 + (lanev >> 3);
                    if ((lanev & 7) == 0 && slot < kLnSlots) p.stats_out[(size_t)row * kLnSlots + slot] = make_float2(a, q2);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // every wave is done with its scratch before the next tile's operands arrive
    }
}

}  // namespace

bool updw_pp_supported(const GemmParams& p) {
    return !p.f8 && !p.conv && !p.w_batch_rows && p.M % 256 == 0 && p.N % 128 == 0 && p.K % 128 == 0 && p.K >= 256 && p.ldo % 4 == 0 &&
           (size_t)p.M * p.lda * 2 < ((size_t)1 << 32) && (size_t)p.N * p.ldw * 2 < ((size_t)1 << 32) && (unsigned)p.lda * 2u < (1u << 24) && (unsigned)p.ldw * 2u < (1u << 24);
}

void launch_updw_pp(const GemmParams& p, hipStream_t s) {
    static PerDeviceOnce once;
    once.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(updw_pp_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, PP::LDS); });
    const int ntn = p.N / 128, ntm = p.M / 256;
    const int ncu = device_cu_count();
    const int ntiles = ntm * ntn;
    const int nblocks = ntiles < 2 * ncu ? ntiles : 2 * ncu;           // (two workgroups fit a CU; launch_gemm only comes here with ntiles <= ncu)
    GemmParams pg = p;
    pg.xcd_ngroups = (ntn % 2 == 0 && ntm >= 8 && nblocks == 2 * ncu && ncu % 8 == 0) ? 2 : 0;
    hipLaunchKernelGGL(updw_pp_kernel<0>, dim3(nblocks), dim3(256), PP::LDS, s, pg, nblocks);
}

// EPI_F32 with ksplit > 1 (the low-latency classes' down projection) on the same 4-wave K loop: K = the length of ONE split (a multiple of 128, >= 256)
bool splitk_pp_supported(const GemmParams& p) {
    return p.ksplit > 1 && p.c_f32 != nullptr && updw_pp_supported(p);
}

void launch_splitk_pp(const GemmParams& p, hipStream_t s) {
    static PerDeviceOnce once;
    once.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(updw_pp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, PP::RING_BYTES); });
    const int ntiles = (p.N / 128) * (p.M / 256) * p.ksplit;
    const int ncu = device_cu_count();
    const int nblocks = ntiles < 2 * ncu ? ntiles : 2 * ncu;
    GemmParams pg = p;
    pg.xcd_ngroups = 0;
    hipLaunchKernelGGL(updw_pp_kernel<1>, dim3(nblocks), dim3(256), PP::RING_BYTES, s, pg, nblocks);
}

// EPI_BIAS_RESID of the default class (bf16 operands, no conv) on 64 x 192 or 128 x 192 tiles: M % 64 == 0, N % 192 == 0, K % 64 == 0, K >= 192
bool down_pp_supported(const GemmParams& p) {
    return !p.f8 && !p.conv && !p.w_batch_rows && p.ksplit <= 1 && p.M % 64 == 0 && p.N % 192 == 0 && p.K % 64 == 0 && p.K >= 192 && p.ldr % 4 == 0 && p.bias && p.resid &&
           (size_t)p.M * p.lda * 2 < ((size_t)1 << 32) && (size_t)p.N * p.ldw * 2 < ((size_t)1 << 32) && (unsigned)p.lda * 2u < (1u << 24) && (unsigned)p.ldw * 2u < (1u << 24);
}

// largest launch this form takes: one 128 x 192 tile per CU
bool down_pp_fits(const GemmParams& p) {
    const long ncu = device_cu_count();
    return (long)(p.M / 64) * (p.N / 192) <= ncu || (p.M % 128 == 0 && (long)(p.M / 128) * (p.N / 192) <= ncu);
}

void launch_down_pp(const GemmParams& p, hipStream_t s) {
    static PerDeviceOnce once;
    once.run([&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(down_pp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, DP::LDS);
        hipFuncSetAttribute(reinterpret_cast<const void*>(down_pp_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, DP::LDS);
    });
    const int ncu = device_cu_count();
    const int n64 = (p.M / 64) * (p.N / 192);
    if (n64 <= ncu) hipLaunchKernelGGL(down_pp_kernel<1>, dim3(n64), dim3(256), DP::LDS, s, p, n64);       // 64-row tiles while they fit one per CU
    else {
        const int n128 = (p.M / 128) * (p.N / 192);
        hipLaunchKernelGGL(down_pp_kernel<2>, dim3(n128 < ncu ? n128 : ncu), dim3(256), DP::LDS, s, p, n128 < ncu ? n128 : ncu);
    }
}

}  // namespace tld
