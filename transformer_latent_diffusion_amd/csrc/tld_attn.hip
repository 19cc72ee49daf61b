// tld_attn.hip -- non-causal multi-head self-attention over latent tokens, head_dim 64, gfx950.
//
// Replaces MHAttention.forward / F.scaled_dot_product_attention for the self-attention call
// (tld/transformer_blocks.py:31-48 via :57-59): softmax(q k^T / sqrt(64)) v, no mask, no dropout,
// heads are contiguous 64-wide column groups ("bs n (h d) -> bs h n d", :35).  The merged-heads
// output is written as bf16 [M, d]; the residual add happens in the following row kernel.
//
// Input layout: q | k row-major [M, 2d] as the QKV GEMM's epilogue writes them (a head's K is 128-byte pieces 3 KB apart; a
// head-major [sample][q, k][head][token][64] layout was measured and changes nothing: profiles/r02_attention_experiments.txt),
// V^T [sample][head * 64 + c][token].
//
// Work split: one workgroup per (sample, head, block of 32*NW query rows); each wave owns 32 query
// rows.  Keys/values are processed in chunks of 32*KT keys staged once per workgroup in LDS:
//   K chunk  [keys][64]  bf16, 128-B rows, DMA'd with global_load_lds using the same source-side
//            XOR swizzle as the GEMM tiles (conflict-free ds_read_b128 of MFMA A-fragments);
//   V^T chunk [64][keys] bf16 (the QKV GEMM epilogue already stores V transposed per head), row
//            stride padded by 8 B so the 32 lanes of a ds_read_b64 hit 32 distinct bank pairs.
// Scores are computed TRANSPOSED, S^T = K . Q^T (A = K fragment, B = Q fragment), so that a lane
// holds, for ITS query column, 16 keys of every 32-key tile: row max / row sum are lane-local plus
// one exchange with lane^32, and exp(S^T) converts in place into the B operand of the second MFMA,
// O^T = V^T . P^T.  The k-slot <-> key assignment of that operand is a permutation of the tile's
// keys (lanes < 32: keys {0-3, 8-11}, lanes >= 32: {4-7, 12-15} of each 16-key step); the V^T
// fragment is gathered with the same permutation, so no cross-lane shuffle of P is needed.
// With more than one chunk (ntok > 256) the usual online-softmax rescale is applied per chunk.
#include "tld_common.h"
#include <cstdlib>

namespace tld {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;


// -DTLD_ATTN_DBG=1 / 2: cost-attribution builds of the 256-token kernel (1: loads, score MFMAs, row max and stores only; 2: loads
// and stores only; 3: everything but the stores) -- wrong results by construction, for tools/ab_r2u.sh
#ifndef TLD_ATTN_DBG
#define TLD_ATTN_DBG 0
#endif
#ifndef TLD_ATTN_ST16
#define TLD_ATTN_ST16 2       // 256-token kernel's output stores: 2 = whole 128-byte rows via an LDS transpose, 1 = 16-byte stores after a lane-pair exchange, 0 = 8-byte stores
#endif

constexpr float kScaleLog2e = 0.125f * 1.44269504088896340736f;   // (1/sqrt(64)) * log2(e)

// MASK (any token count that is a multiple of 8 -- grids the reference accepts and the shape-specialised kernels do not take, e.g. 12 x 12 or
// 20 x 20): the last key chunk and the last query block are partial; keys past ntok score -inf (their K rows are the sample's last row, their
// V^T values zero), query rows past ntok are computed on the last row and not stored.
template <int KT, int NW, bool MASK = false>   // 32-key tiles per chunk; NW waves (32 query rows each) per workgroup
__global__ __launch_bounds__(NW * 64) void attn_kernel(const bf16* __restrict__ qk,
                                                       const bf16* __restrict__ vt,
                                                       bf16* __restrict__ att, int ntok, int d, int nbuf) {
    constexpr int KC = KT * 32;                 // keys per chunk
    constexpr int VSTRIDE = KC * 2 + 8;         // bytes per V^T row in LDS (padded)
    constexpr int KBYTES = KC * 128, VBYTES = 64 * VSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // nbuf = 2 (more than one chunk): K / V^T chunks are double-buffered -- the K chunk of step c+1 travels by DMA and the
    // V^T chunk through registers while step c computes.  The kernel holds ~210 VGPRs (128 of them scores), i.e. ONE
    // workgroup per CU, so with a single buffer every chunk's staging latency was exposed (two barriers and ~1.5 us of
    // waiting per 256 keys).  Layout: [K0 | K1 | V0 | V1].
    char* Kbase = smem;
    char* Vbase = smem + nbuf * KBYTES;

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int hi = lane >> 5, l31 = lane & 31;
    const int twod = 2 * d;

    const size_t row_base = (size_t)b * ntok;
    const int q0 = qblk * (NW * 32) + wid * 32;

    // Q fragments (B operand of S^T = K Q^T): lane holds q-row l31, features ks*16 + hi*8 .. +8
    bf16x8 qf[4];
    {
        const int qrow = MASK ? (q0 + l31 < ntok ? q0 + l31 : ntok - 1) : q0 + l31;
        const bf16* qp = qk + (row_base + qrow) * twod + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    f32x16 o[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nchunks = MASK ? (ntok + KC - 1) / KC : ntok / KC;
    constexpr int KP = KC / 8 / NW;              // 8-row K DMA pieces per wave
    constexpr int PIECES = 64 * (KC / 8);
    constexpr int PER_THREAD = PIECES / (NW * 64);
    // K chunk: KC rows x 128 B, source-side XOR swizzle as in the GEMM tiles
    auto stage_k = [&](int ch, char* Ks) {
        const bf16* kbase = qk + (row_base + (size_t)ch * KC) * twod + d + h * 64;
#pragma unroll
        for (int it = 0; it < KP; ++it) {
            const int r = (wid * KP + it) * 8 + (lane >> 3);
            const int clog = (lane & 7) ^ ((r >> 1) & 7);
            const int rs = MASK ? (ch * KC + r < ntok ? r : ntok - 1 - ch * KC) : r;        // (the image position stays r)
            __builtin_amdgcn_global_load_lds((gptr_t)(kbase + (ptrdiff_t)rs * twod + clog * 8), (lptr_t)(Ks + (wid * KP + it) * 1024), 16, 0, 0);
        }
    };
    u32x4 vreg[PER_THREAD];
    auto load_v = [&](int ch) {
        const bf16* vbase = vt + ((size_t)b * d + h * 64) * ntok + (size_t)ch * KC;
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int pidx = it * (NW * 64) + threadIdx.x;
            const int c = pidx / (KC / 8), kc8 = pidx % (KC / 8);
            if (MASK && ch * KC + kc8 * 8 >= ntok) vreg[it] = u32x4{0u, 0u, 0u, 0u};     // (ntok % 8 == 0: an 8-key piece is whole or absent)
            else vreg[it] = *reinterpret_cast<const u32x4*>(vbase + (size_t)c * ntok + kc8 * 8);
        }
    };
    stage_k(0, Kbase);
    load_v(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = nbuf == 2 ? (ch & 1) : 0;
        char* Ks = Kbase + cur * KBYTES;
        char* Vs = Vbase + cur * VBYTES;
        // K chunk ch landed (own pieces: vmcnt; everybody's: barrier).  The barrier also says: every wave is past the PV
        // MFMAs of chunk ch-1, so the OTHER K buffer (read by chunk ch-1's scores) may be overwritten now.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nbuf == 2 && ch + 1 < nchunks) stage_k(ch + 1, Kbase + (cur ^ 1) * KBYTES);

        // ---- S^T = K Q^T : KT tiles of [32 keys x 32 queries]
        f32x16 st[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int row = t * 32 + l31;
                const int kc = ks * 2 + hi;
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(
                    Ks + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
                st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[t], 0, 0, 0);
            }
        }

        if (MASK && (ch + 1) * KC > ntok) {            // partial last chunk: keys past the end never win the max and weigh exp2(-huge) = 0
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (ch * KC + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= ntok) st[t][r] = -3.0e38f;
        }
        // ---- softmax statistics for this lane's query (shared with lane ^ 32)
        float mx = st[0][0];
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * kScaleLog2e);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // 0 on the first chunk
        m_run = m_new;
        l_run *= alpha;
        if (ch > 0) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        }

        // ---- V^T chunk ch: registers (loaded during the previous chunk) -> LDS (padded pitch); its buffer was last read by
        // the PV MFMAs of chunk ch-2 (ch-1 with a single buffer): over for every wave since the barrier above
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int pidx = it * (NW * 64) + threadIdx.x;
            const int c = pidx / (KC / 8), kc8 = pidx % (KC / 8);
            uint2* dst = reinterpret_cast<uint2*>(Vs + c * VSTRIDE + kc8 * 16);
            dst[0] = make_uint2(vreg[it][0], vreg[it][1]);
            dst[1] = make_uint2(vreg[it][2], vreg[it][3]);
        }
        // (raw barrier: __syncthreads() would also drain the K DMA of the next chunk, which is in flight here)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 1 < nchunks) {
            if (nbuf == 1) stage_k(ch + 1, Kbase);     // single buffer: every wave is past its score MFMAs (barrier above)
            load_v(ch + 1);                           // lands under the PV MFMAs below and the next chunk's scores
        }

        // ---- P = exp2(s' - m), O^T += V^T P^T
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                bf16x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(st[t][hf * 8 + e] * kScaleLog2e - m_new);
                    l_run += pv;
                    pf[e] = (bf16)pv;
                }
                const int kb = t * 32 + hf * 16 + hi * 4;          // first key of this lane's k-slots
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const char* vp = Vs + (ct * 32 + l31) * VSTRIDE + kb * 2;
                    const uint2 v0 = *reinterpret_cast<const uint2*>(vp);        // keys kb .. kb+3
                    const uint2 v1 = *reinterpret_cast<const uint2*>(vp + 16);   // keys kb+8 .. kb+11
                    union { uint4 u; bf16x8 v; } cvt;
                    cvt.u = make_uint4(v0.x, v0.y, v1.x, v1.y);
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt.v, pf, o[ct], 0, 0, 0);
                }
            }
        }
    }

    // ---- normalise and store O^T: lane's query row, 4 consecutive features per register group
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // 16-byte stores: lane pairs (l, l ^ 32) trade their odd / even feature quads (see attn1_kernel for the measurements)
    bf16* op = att + (row_base + q0 + l31) * d + h * 64 + 8 * hi;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            union { bf16x4 v; unsigned u[2]; } x, y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x.v[e] = (bf16)(o[ct][(2 * pr) * 4 + e] * inv);
                y.v[e] = (bf16)(o[ct][(2 * pr + 1) * 4 + e] * inv);
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(x.u[0], y.u[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(x.u[1], y.u[1], false, false);
            u32x4 w;
            w[0] = s0[0]; w[1] = s1[0]; w[2] = s0[1]; w[3] = s1[1];
            if (!MASK || q0 + l31 < ntok) *reinterpret_cast<u32x4*>(op + ct * 32 + pr * 16) = w;
        }
}

// Single-chunk specialisation (ntok == 32*KT <= 256, the 256 px case): 4-wave workgroups that walk QT query
// tiles per wave, so K / V^T are still staged once per (sample, head) but TWO workgroups share a CU (2 x 66 KB
// LDS, 8 waves).  The two are not synchronised with each other, so one's softmax (VALU/exp bound: ~2x the MFMA
// time of a tile) overlaps the other's MFMAs and staging -- with one 8-wave workgroup per CU every wave was in
// the same phase.
// PIPE: the LDS fragment reads of both MFMA phases are software-pipelined one step ahead (scores: the four K fragments of
// the next (tile pair, k half); P V: the two V^T fragments of the next 16-key step).  In the plain form the compiler issues
// every fragment read right in front of its MFMA (ISA: ds_read ... ~10 VALU ... s_waitcnt ... v_mfma), so each of the 32
// steps of a query tile exposes an LDS round trip under 8 contending waves.
template <int KT, int NW, int QT, bool PIPE>
__global__ __launch_bounds__(NW * 64, 2) void attn1_kernel(const bf16* __restrict__ qk, const bf16* __restrict__ vt,
                                                        bf16* __restrict__ att, int ntok, int d) {
    constexpr int KC = KT * 32;
    constexpr int VSTRIDE = KC * 2 + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + KC * 128;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    const int hi = lane >> 5, l31 = lane & 31;
    const int twod = 2 * d;
    const size_t row_base = (size_t)b * ntok;

    // Q fragments of the wave's first query tile: issued before the K/V staging waits, and inside the loop the
    // next tile's are issued before the current tile is processed (the load used to sit in front of its MFMAs)
    bf16x8 qnext[4];
    auto load_q = [&](int qt, bf16x8 (&q)[4]) {
        const bf16* qp = qk + (row_base + (wid * QT + qt) * 32 + l31) * twod + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) q[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    };
    load_q(0, qnext);
    {   // K: DMA, source-side swizzle
        const bf16* kbase = qk + row_base * twod + d + h * 64;
        constexpr int KP = KC / 8 / NW;
#pragma unroll
        for (int it = 0; it < KP; ++it) {
            const int r = (wid * KP + it) * 8 + (lane >> 3);
            const int clog = (lane & 7) ^ ((r >> 1) & 7);
            __builtin_amdgcn_global_load_lds((gptr_t)(kbase + (size_t)r * twod + clog * 8),
                                             (lptr_t)(Ks + (wid * KP + it) * 1024), 16, 0, 0);
        }
    }
    __syncthreads();
    constexpr int PIECES = 64 * (KC / 8);
    constexpr int PER_THREAD = PIECES / (NW * 64);
    u32x4 vreg[PER_THREAD];
    {
        const bf16* vbase = vt + ((size_t)b * d + h * 64) * ntok;
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int pidx = it * (NW * 64) + threadIdx.x;
            const int c = pidx / (KC / 8), kc8 = pidx % (KC / 8);
            vreg[it] = *reinterpret_cast<const u32x4*>(vbase + (size_t)c * ntok + kc8 * 8);
        }
    }
#pragma unroll 1
    for (int qt = 0; qt < QT; ++qt) {
        const int q0 = (wid * QT + qt) * 32;
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = qnext[ks];
        if (qt + 1 < QT) load_q(qt + 1, qnext);
        f32x16 st[KT];
        if constexpr (TLD_ATTN_DBG == 2) {          // attribution build: no score MFMAs
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[t][r] = (float)qf[t & 3][r & 7];
        } else if constexpr (PIPE) {
            static_assert(KT % 2 == 0, "tile pairs");
            auto kfrag = [&](int t, int ks) {
                const int row = t * 32 + l31;
                const int kc = ks * 2 + hi;
                return *reinterpret_cast<const bf16x8*>(Ks + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
            };
            // step s = (tile pair s >> 1, k half s & 1): fragments K[2p][2h], K[2p][2h+1], K[2p+1][2h], K[2p+1][2h+1]
            bf16x8 fa[4], fb[4];
            auto fetch = [&](int s2, bf16x8 (&f)[4]) {
                const int p2 = s2 >> 1, h2 = s2 & 1;
                f[0] = kfrag(2 * p2, 2 * h2); f[1] = kfrag(2 * p2 + 1, 2 * h2);
                f[2] = kfrag(2 * p2, 2 * h2 + 1); f[3] = kfrag(2 * p2 + 1, 2 * h2 + 1);
            };
            auto fire = [&](int s2, const bf16x8 (&f)[4]) {
                const int p2 = s2 >> 1, h2 = s2 & 1;
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                st[2 * p2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], qf[2 * h2], h2 ? st[2 * p2] : zero, 0, 0, 0);
                st[2 * p2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], qf[2 * h2], h2 ? st[2 * p2 + 1] : zero, 0, 0, 0);
                st[2 * p2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2], qf[2 * h2 + 1], st[2 * p2], 0, 0, 0);
                st[2 * p2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[3], qf[2 * h2 + 1], st[2 * p2 + 1], 0, 0, 0);
            };
            fetch(0, fa);
#pragma unroll
            for (int s2 = 0; s2 < KT; s2 += 2) {
                fetch(s2 + 1, fb);
                __builtin_amdgcn_sched_barrier(0);
                fire(s2, fa);
                __builtin_amdgcn_sched_barrier(0);
                if (s2 + 2 < KT) fetch(s2 + 2, fa);
                __builtin_amdgcn_sched_barrier(0);
                fire(s2 + 1, fb);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int row = t * 32 + l31;
                const int kc = ks * 2 + hi;
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
                st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[t], 0, 0, 0);
            }
        }
        }
        float mx = st[0][0];
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = mx * kScaleLog2e;
        if (qt == 0) {
#pragma unroll
            for (int it = 0; it < PER_THREAD; ++it) {
                const int pidx = it * (NW * 64) + threadIdx.x;
                const int c = pidx / (KC / 8), kc8 = pidx % (KC / 8);
                uint2* dst = reinterpret_cast<uint2*>(Vs + c * VSTRIDE + kc8 * 16);
                dst[0] = make_uint2(vreg[it][0], vreg[it][1]);
                dst[1] = make_uint2(vreg[it][2], vreg[it][3]);
            }
            __syncthreads();
        }
        f32x16 o[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
        float l_run = 0.f;
        if constexpr (TLD_ATTN_DBG != 0) {          // attribution build: no exponentials / P V MFMAs
            l_run = 0.5f + m_new * 0.f;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] = st[ct][r] + st[2 + ct][r] + st[4 + ct][r] + st[6 + ct][r] + (float)*reinterpret_cast<const bf16*>(Vs + (ct * 32 + l31) * VSTRIDE + r * 2);
        } else if constexpr (PIPE) {
            auto vfrag = [&](int s2, int ct) {                 // step s2 = 2 t + hf: keys s2 * 16 + hi * 4 .. (+3, +8 .. +11)
                const char* vp = Vs + (ct * 32 + l31) * VSTRIDE + (s2 * 16 + hi * 4) * 2;
                const uint2 v0 = *reinterpret_cast<const uint2*>(vp);
                const uint2 v1 = *reinterpret_cast<const uint2*>(vp + 16);
                union { uint4 u; bf16x8 v; } cvt;
                cvt.u = make_uint4(v0.x, v0.y, v1.x, v1.y);
                return cvt.v;
            };
            auto probs = [&](int s2) {
                bf16x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(st[s2 >> 1][(s2 & 1) * 8 + e] * kScaleLog2e - m_new);
                    l_run += pv;
                    pf[e] = (bf16)pv;
                }
                return pf;
            };
            bf16x8 va[2], vb[2];
            va[0] = vfrag(0, 0); va[1] = vfrag(0, 1);
#pragma unroll
            for (int s2 = 0; s2 < 2 * KT; s2 += 2) {
                vb[0] = vfrag(s2 + 1, 0); vb[1] = vfrag(s2 + 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                {
                    const bf16x8 pf = probs(s2);
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[0], pf, o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[1], pf, o[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s2 + 2 < 2 * KT) { va[0] = vfrag(s2 + 2, 0); va[1] = vfrag(s2 + 2, 1); }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const bf16x8 pf = probs(s2 + 1);
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[0], pf, o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[1], pf, o[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                bf16x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(st[t][hf * 8 + e] * kScaleLog2e - m_new);
                    l_run += pv;
                    pf[e] = (bf16)pv;
                }
                const int kb = t * 32 + hf * 16 + hi * 4;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const char* vp = Vs + (ct * 32 + l31) * VSTRIDE + kb * 2;
                    const uint2 v0 = *reinterpret_cast<const uint2*>(vp);
                    const uint2 v1 = *reinterpret_cast<const uint2*>(vp + 16);
                    union { uint4 u; bf16x8 v; } cvt;
                    cvt.u = make_uint4(v0.x, v0.y, v1.x, v1.y);
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt.v, pf, o[ct], 0, 0, 0);
                }
            }
        }
        }
        const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
#if TLD_ATTN_DBG == 3
        if (inv > 1e30f)    // attribution build: (practically) no stores
#endif
        {
#if TLD_ATTN_ST16 == 2
            // Whole-row stores.  The MFMA leaves a lane with 4-feature pieces of ITS query row (8 bytes each; 16 bytes per row
            // and store instruction).  This kernel is HBM-bound (200 MB per launch at C1: 26 us without the stores, 48.6 us
            // with the direct 8-byte ones, profiles/r02_attention_experiments.txt), and the store shape is worth 10 % of it:
            // 16-byte pieces after a lane-pair exchange 44.6 us, whole 128-byte rows 45.2 us, whole rows nontemporal 43.6 us.
            // The tile is transposed through a 2.3-KB per-wave LDS patch, 16 rows at a time (pitch 144 B: conflict-free
            // 8-byte writes, 16-byte aligned reads).  Wave-local: LDS operations of one wave execute in order, no barrier.
            char* T = smem + KC * 128 + 64 * VSTRIDE + wid * (16 * 144);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if ((l31 >> 4) == half) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            bf16x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pk[e] = (bf16)(o[ct][rq * 4 + e] * inv);
                            *reinterpret_cast<bf16x4*>(T + (l31 & 15) * 144 + ct * 64 + rq * 16 + hi * 8) = pk;
                        }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int r = it * 8 + (lane >> 3), c16 = lane & 7;
                    const u32x4 w = *reinterpret_cast<const u32x4*>(T + r * 144 + c16 * 16);
                    // (nontemporal: 45.2 -> 43.6 us; the consumer, cross_row, is not slowed by it)
                    __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(att + (row_base + q0 + half * 16 + r) * d + h * 64 + c16 * 8));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#elif TLD_ATTN_ST16
            // 16-byte stores: a lane owns feature quads {8 rq + 4 hi ..+3}; lane pairs (l, l ^ 32) trade the odd / even quads with
            // v_permlane32_swap so that each ends up with 8 consecutive features (32 contiguous bytes per row and instruction
            // instead of 16)
            bf16* op = att + (row_base + q0 + l31) * d + h * 64 + 8 * hi;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    union { bf16x4 v; unsigned u[2]; } x, y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x.v[e] = (bf16)(o[ct][(2 * pr) * 4 + e] * inv);
                        y.v[e] = (bf16)(o[ct][(2 * pr + 1) * 4 + e] * inv);
                    }
                    // v_permlane32_swap a, b: a.upper <-> b.lower.  (x, y) -> lower lanes: own x | partner's x; upper: partner's y | own y
                    const auto s0 = __builtin_amdgcn_permlane32_swap(x.u[0], y.u[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(x.u[1], y.u[1], false, false);
                    u32x4 w;
                    w[0] = s0[0]; w[1] = s1[0]; w[2] = s0[1]; w[3] = s1[1];
                    *reinterpret_cast<u32x4*>(op + ct * 32 + pr * 16) = w;
                }
#else
            bf16* op = att + (row_base + q0 + l31) * d + h * 64 + 4 * hi;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    bf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (bf16)(o[ct][rq * 4 + e] * inv);
                    *reinterpret_cast<bf16x4*>(op + ct * 32 + rq * 8) = pk;
                }
#endif
        }
    }
}

// =====================================================================================================================
// attn2_kernel (round 3): the multi-chunk kernel for 512 / 1024 px grids (ntok >= 512), replacing attn_kernel<8, 8> there.
// What bounded attn_kernel (151 us at C3 = 0.68 PFLOP/s, 455 us at C4): per 32-key x 32-query tile a wave spends 256 cycles of MFMA,
// ~260 cycles of VALU (4.5 instructions per score at 2 waves/SIMD, profiles/r02_valu_issue_rates.txt) and 8 KB of LDS fragment reads
// (= the CU's 128 B/clk for four SIMDs), and all 8 waves of the only resident workgroup walk the phases in lockstep.  Changes:
//   * a wave owns TWO query tiles at once (64 queries): every K / V^T fragment read from LDS feeds two MFMAs -- LDS bytes per flop halve;
//   * 4-wave workgroups (256 queries), two per CU and not synchronised with each other (74 KB of LDS each);
//   * row max with v_max3; scale and (stale) max in one v_fma per score;
//   * lazy rescaling: the running max is only advanced when a tile exceeds it by more than 2^kLazy (P stays <= 2^kLazy, exact in the
//     ratio o / l); after the first tiles of a row the 32-multiply rescale of O practically never runs;
//   * K rows land in LDS in the order [0-3, 8-11, 4-7, 12-15] of every 16 (free with the DMA's per-lane source address), which makes the
//     keys of a lane's P fragment CONTIGUOUS: the V^T fragment is one ds_read_b128 from a plain [64][128 keys] tile, so V^T can travel
//     by global_load_lds as well (16-byte chunk XOR swizzle by row & 15 instead of the padded pitch) -- no staging through registers;
//   * 128-key chunks, K and V^T double-buffered with separate barriers (K of chunk c+1 is needed one tile EARLIER than its V^T, because
//     the scores of tile i+1 are issued while tile i's softmax runs; V^T of chunk c is still in use then), counted vmcnt as in the GEMM.
// Per tile i the wave issues S(i+1) (8 MFMAs), exp / pack of tile i, P V (8 MFMAs), row max of tile i+1.
constexpr float kLazy = 6.0f;
// -DTLD_A2_DBG=n: cost-attribution builds (wrong results by construction; tools/r3_attn2_attr.sh): 1 no exponentials, 2 no softmax VALU work
// on the scores at all, 3 = 2 and no row max, 5 = 3 and no LDS fragment reads (MFMAs, DMA and barriers only)
#ifndef TLD_A2_DBG
#define TLD_A2_DBG 0
#endif
#ifndef TLD_A2_CLK
#define TLD_A2_CLK 0
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N> __device__ __forceinline__ void a2_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Round 6: XCD-local block order.  The grid used to be (query block, head, sample) with the query block fastest, so the ntok / 256 workgroups that share one
//     (sample, head)'s K / V^T were dealt round-robin over the 8 XCDs and EVERY XCD's L2 fetched every pair's K / V^T: 1.74 GB per launch at C4 against
//     0.40 GB algorithmic, 483 MB against 201 at C3 (gpurun_out/r06_base PMC passes; after: profiles/r06_pmc_traffic_c4_*.json / _c3.json).  Time is unchanged -- the kernel is VALU-issue-bound, not fetch-bound
//     (profiles/r06_attn2_prescaled_experiment.txt) -- the traffic is what the order is for.  Now the grid is 1-D: workgroup L runs on
//     XCD L & 7 (where the dispatcher puts it) and takes query block (L >> 3) % gx of pair ((L >> 3) / gx) * 8 + (L & 7) -- the gx workgroups of a pair are
//     consecutive workgroups of ONE XCD, resident at the same time, and walk the key chunks together.
__global__ __launch_bounds__(256, 2) void attn2_kernel(const bf16* __restrict__ qk, const bf16* __restrict__ vt,
                                                       bf16* __restrict__ att, int ntok, int d, int heads, int npairs) {
    constexpr int KC = 128, KB = KC * 128, VB = 64 * KC * 2, TP = 16 * 144;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gx = ntok >> 8;
    const int jb = (int)blockIdx.x >> 3;
    const int pair = (jb / gx) * 8 + ((int)blockIdx.x & 7);
    if (pair >= npairs) return;                       // (the grid is padded to 8 pairs per round)
    const int b = pair / heads, h = pair - b * heads;
    const int hi = lane >> 5, l31 = lane & 31;
    const int twod = 2 * d;
    const size_t row_base = (size_t)b * ntok;
    const int q0 = (jb - (jb / gx) * gx) * 256 + wid * 64;
    const int nchunks = ntok / KC;
#if TLD_A2_CLK
    const uint64_t clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    bf16x8 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const bf16* qp = qk + (row_base + q0 + qt * 32 + l31) * twod + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }
    const bf16* kbase0 = qk + row_base * twod + d + h * 64;
    const bf16* vbase0 = vt + ((size_t)b * d + h * 64) * ntok;
    // DMA addressing as in the GEMM: a uniform 64-bit base (SGPR pair, advanced per chunk) + one 32-bit per-lane byte offset per piece, rebuilt from the
    // lane id where it is used (8 pieces per 128-key chunk: a few VALU instructions each) instead of per-lane 64-bit pointers kept across the chunk loop:
    // 243 -> 221 VGPRs
    auto stage_k = [&](int ch) {            // LDS row r of the chunk <- key pi(r): quads 1 and 2 of every 16 rows swapped
        const char* kb = reinterpret_cast<const char*>(kbase0) + (size_t)ch * KC * twod * 2;
        asm volatile("" : "+s"(kb));
        char* dst = smem + (ch & 1) * KB;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int piece = wid * 4 + it;
            const int r = piece * 8 + (ln >> 3);
            const int key = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
            const int clog = (ln & 7) ^ ((r >> 1) & 7);
            unsigned o = (unsigned)(key * twod + clog * 8) * 2u;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gptr_t)(kb + o), (lptr_t)(dst + piece * 1024), 16, 0, 0);
        }
    };
    auto stage_v = [&](int ch) {            // [64 features][128 keys], 16-byte chunk c of row f at chunk c ^ (f & 15)
        const char* vb = reinterpret_cast<const char*>(vbase0) + (size_t)ch * KC * 2;
        asm volatile("" : "+s"(vb));
        char* dst = smem + 2 * KB + (ch & 1) * VB;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int piece = wid * 4 + it;
            const int f = piece * 4 + (ln >> 4);
            const int c = (ln & 15) ^ (f & 15);
            unsigned o = (unsigned)(f * ntok + c * 8) * 2u;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + o), (lptr_t)(dst + piece * 1024), 16, 0, 0);
        }
    };
    // lane parts of the fragment addresses: K row l31, chunk (2 ks + hi) ^ ((l31 >> 1) & 7);  V^T row l31 (+32 ct), chunk (2 s + hi) ^ (l31 & 15)
    const int k_lane = l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4);
    const int v_lane = l31 * 256 + ((hi ^ (l31 & 15)) << 4);

    f32x16 st[2][2];            // [tile parity][query tile]
    f32x16 o[2][2];             // [query tile][feature half]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][ct][r] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    auto scores = [&](int kofs, f32x16 (&s)[2]) {
        bf16x8 kf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (TLD_A2_DBG == 5) kf[ks] = qf[1][ks];
            else kf[ks] = *reinterpret_cast<const bf16x8*>(smem + ((kofs + k_lane) ^ (ks << 5)));
        }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
                s[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[qt][ks], ks ? s[qt] : zero, 0, 0, 0);
    };
    // row max of a fresh score tile; advances the running max (and rescales O, l) only past the lazy threshold
    auto advance = [&](const f32x16 (&s)[2]) {
        if constexpr (TLD_A2_DBG >= 3) return;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = fmaxf(fmaxf(s[qt][0], s[qt][1]), s[qt][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[qt][r]), s[qt][r + 1]);
            mx = fmaxf(mx, s[qt][15]);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * kScaleLog2e;
            if (__builtin_amdgcn_ballot_w64(mx > m_run[qt] + kLazy) != 0) {
                const float m_new = fmaxf(m_run[qt], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
                m_run[qt] = m_new;
                l_run[qt] *= alpha;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qt][ct][r] *= alpha;
            }
        }
    };
    // P fragment (8 keys x this lane's query) of score registers [hf * 8, +8): exp2(s * scale - m) -> packed bf16; row sum in fp32.
    // Scalar v_fma / v_add on purpose: v_pk_fma_f32, v_pk_mul_f32 and v_dot2c_f32_bf16 do NOT run beside MFMAs (they serialise with the
    // matrix pipe and cost extra, tools/ubench/overlap2.hip: "hidden" -0.16 .. -0.31), v_fma / v_exp / v_cvt_pk / v_max3 hide 50-85 %.
    auto probs = [&](const f32x16& s, int hf, float negm, float& l) {
        union { bf16x8 v; unsigned u[4]; } pf;
        if constexpr (TLD_A2_DBG >= 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pf.u[e] = __float_as_uint(s[hf * 8 + 2 * e]);
            l = 1.f;
            return pf.v;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float tt = __builtin_fmaf(s[hf * 8 + e], kScaleLog2e, negm);
            const float pv = TLD_A2_DBG == 1 ? tt : __builtin_amdgcn_exp2f(tt);
            l += pv;
            pf.v[e] = (bf16)pv;
        }
        return pf.v;
    };

    stage_k(0);
    stage_v(0);
    stage_k(1);
    a2_wait_vmcnt<8>();                     // Q, K(0) landed (own pieces) ...
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();           // ... everybody's
    __builtin_amdgcn_sched_barrier(0);
    scores(0, st[0]);
    advance(st[0]);

#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        const int vofs = 2 * KB + (c & 1) * VB + v_lane;
        const bool more = c + 1 < nchunks;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t == 3 && more) {           // K(c+1) landed; every wave has read the last K fragments of chunk c: its buffer is free
                a2_wait_vmcnt<4>();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (c + 2 < nchunks) stage_k(c + 2);
            }
            if (t == 0) {                   // V^T(c) landed; every wave is past the P V of chunk c-1: that buffer is free
                if (more) a2_wait_vmcnt<4>(); else a2_wait_vmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (more) stage_v(c + 1);
            }
            f32x16 (&sc)[2] = st[t & 1];
            f32x16 (&sn)[2] = st[(t + 1) & 1];
            // S(i+1): next tile's scores (nothing to prefetch after the very last tile).  Order inside the body is the compiler's: it
            // interleaves these MFMAs with tile i's exponentials.  (Pinning "all fragment reads first, then VALU, then MFMAs" with
            // sched_barriers measured 3 % slower, profiles/r03_attn2_attribution.txt.)
            const int kofs = ((t == 3 ? c + 1 : c) & 1) * KB + ((t + 1) & 3) * 4096;
            if (t < 3 || more) scores(kofs, sn);
            // tile i: P and P V
            bf16x8 pf[2][2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) pf[hf][qt] = probs(sc[qt], hf, -m_run[qt], l_run[qt]);
                bf16x8 vf[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    if constexpr (TLD_A2_DBG == 5) vf[ct] = qf[0][ct + 2 * hf];
                    else vf[ct] = *reinterpret_cast<const bf16x8*>(smem + ((vofs + ct * 8192) ^ ((t * 2 + hf) << 5)));
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                        o[qt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ct], pf[hf][qt], o[qt][ct], 0, 0, 0);
            }
            if (t < 3 || more) advance(sn);
        }
    }

    // ---- normalise, transpose through the per-wave LDS patch, whole-row stores (as attn1_kernel)
    char* T = smem + 2 * KB + 2 * VB + wid * TP;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float lt = l_run[qt] + __shfl_xor(l_run[qt], 32, 64);
        const float inv = 1.0f / lt;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((l31 >> 4) == half) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        bf16x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = (bf16)(o[qt][ct][rq * 4 + e] * inv);
                        *reinterpret_cast<bf16x4*>(T + (l31 & 15) * 144 + ct * 64 + rq * 16 + hi * 8) = pk;
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = it * 8 + (lane >> 3), c16 = lane & 7;
                const u32x4 w = *reinterpret_cast<const u32x4*>(T + r * 144 + c16 * 16);
                __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(att + (row_base + q0 + qt * 32 + half * 16 + r) * d + h * 64 + c16 * 8));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
#if TLD_A2_CLK      // clock probe build: workgroup (0,0,0) leaves (shader-clock ticks, 100 MHz ticks) of its lifetime in att[0..15]
    __builtin_amdgcn_s_waitcnt(0);
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        uint64_t* dst = reinterpret_cast<uint64_t*>(att);
        dst[0] = __builtin_readcyclecounter() - clk0;
        dst[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
}

template <int KT, int NW, int QT, bool PIPE>
void launch_attn1(const bf16* qk, const bf16* vt, bf16* att, int batch, int ntok, int heads, hipStream_t s) {
    constexpr int KC = KT * 32;
    const int lds = KC * 128 + 64 * (KC * 2 + 8) + (TLD_ATTN_ST16 == 2 ? NW * 16 * 144 : 0);
    static PerDeviceOnce attr_set;
    attr_set.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(attn1_kernel<KT, NW, QT, PIPE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    hipLaunchKernelGGL((attn1_kernel<KT, NW, QT, PIPE>), dim3(1, heads, batch), dim3(NW * 64), lds, s, qk, vt, att, ntok,
                       heads * 64);
}

template <int KT, int NW>
void launch_kt(const bf16* qk, const bf16* vt, bf16* att, int batch, int ntok, int heads, hipStream_t s) {
    constexpr int KC = KT * 32;
    const int nbuf = (ntok > KC && 2 * (KC * 128 + 64 * (KC * 2 + 8)) <= 160 * 1024) ? 2 : 1;       // (the engine's shapes are all single-chunk here: ntok == KC)
    const int lds = nbuf * (KC * 128 + 64 * (KC * 2 + 8));
    static PerDeviceMax attr_lds;
    attr_lds.run(lds, [&] { hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<KT, NW>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    dim3 grid(ntok / (NW * 32), heads, batch), block(NW * 64);
    hipLaunchKernelGGL((attn_kernel<KT, NW>), grid, block, lds, s, qk, vt, att, ntok, heads * 64, nbuf);
}

// any token count (multiple of 8): 128-key chunks, 4-wave workgroups of 128 queries, partial last chunk / block masked
void launch_masked(const bf16* qk, const bf16* vt, bf16* att, int batch, int ntok, int heads, hipStream_t s) {
    constexpr int KT = 4, NW = 4, KC = KT * 32;
    const int nbuf = ntok > KC ? 2 : 1;
    const int lds = nbuf * (KC * 128 + 64 * (KC * 2 + 8));
    static PerDeviceMax attr_lds;
    attr_lds.run(lds, [&] { hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<KT, NW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    dim3 grid((ntok + NW * 32 - 1) / (NW * 32), heads, batch), block(NW * 64);
    hipLaunchKernelGGL((attn_kernel<KT, NW, true>), grid, block, lds, s, qk, vt, att, ntok, heads * 64, nbuf);
}

void launch_attn2(const bf16* qk, const bf16* vt, bf16* att, int batch, int ntok, int heads, hipStream_t s) {
    constexpr int lds = 2 * 128 * 128 + 2 * 64 * 256 + 4 * 16 * 144;
    static PerDeviceOnce attr_set;
    attr_set.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(attn2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    const int npairs = batch * heads;
    const int nblocks = ((npairs + 7) / 8) * 8 * (ntok / 256);         // pairs padded to whole rounds of the 8 XCDs (the surplus workgroups return at once)
    hipLaunchKernelGGL(attn2_kernel, dim3(nblocks), dim3(256), lds, s, qk, vt, att, ntok, heads * 64, heads, npairs);
}

}  // namespace

void launch_attention(const bf16* qk, const bf16* vt, bf16* att, int batch, int ntok, int heads,
                      hipStream_t s) {
    // 256 tokens: two unsynchronised 4-wave workgroups per CU, each staging its (sample, head)'s K / V^T once (attn1_kernel; the single
    // 8-wave workgroup it replaced took 60 us against 44 at C1).  512+ tokens: the chunked two-query-tile kernel (attn2_kernel).
    // (At the C1 shape the inference engine no longer comes here: its self-attention runs in the QKV GEMM's epilogue, EPI_QKV_ATTN.)
    if (ntok == 256) launch_attn1<8, 4, 2, true>(qk, vt, att, batch, ntok, heads, s);
    else if (ntok % 256 == 0) launch_attn2(qk, vt, att, batch, ntok, heads, s);
    else if (ntok == 128) launch_kt<4, 4>(qk, vt, att, batch, ntok, heads, s);
    else if (ntok == 64) launch_kt<2, 2>(qk, vt, att, batch, ntok, heads, s);
    else if (ntok == 32) launch_kt<1, 1>(qk, vt, att, batch, ntok, heads, s);
    else launch_masked(qk, vt, att, batch, ntok, heads, s);       // any other multiple of 8 (tld_engine_create checks)
}

}  // namespace tld
