// tld_attn.hip -- non-causal multi-head self-attention over latent tokens, head_dim 64, gfx950.
//
// Replaces MHAttention.forward / F.scaled_dot_product_attention for the self-attention call
// (tld/transformer_blocks.py:31-48 via :57-59): softmax(q k^T / sqrt(64)) v, no mask, no dropout,
// heads are contiguous 64-wide column groups ("bs n (h d) -> bs h n d", :35).  The merged-heads
// output is written as bf16 [M, d]; the residual add happens in the following row kernel.
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


constexpr float kScaleLog2e = 0.125f * 1.44269504088896340736f;   // (1/sqrt(64)) * log2(e)

template <int KT, int NW>   // 32-key tiles per chunk; NW waves (32 query rows each) per workgroup
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
        const bf16* qp = qk + (row_base + q0 + l31) * twod + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    f32x16 o[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nchunks = ntok / KC;
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
            __builtin_amdgcn_global_load_lds((gptr_t)(kbase + (size_t)r * twod + clog * 8), (lptr_t)(Ks + (wid * KP + it) * 1024), 16, 0, 0);
        }
    };
    u32x4 vreg[PER_THREAD];
    auto load_v = [&](int ch) {
        const bf16* vbase = vt + ((size_t)b * d + h * 64) * ntok + (size_t)ch * KC;
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int pidx = it * (NW * 64) + threadIdx.x;
            const int c = pidx / (KC / 8), kc8 = pidx % (KC / 8);
            vreg[it] = *reinterpret_cast<const u32x4*>(vbase + (size_t)c * ntok + kc8 * 8);
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
    bf16* op = att + (row_base + q0 + l31) * d + h * 64 + 4 * hi;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (bf16)(o[ct][rq * 4 + e] * inv);
            *reinterpret_cast<bf16x4*>(op + ct * 32 + rq * 8) = pk;
        }
    }
}

// Single-chunk specialisation (ntok == 32*KT <= 256, the 256 px case): 4-wave workgroups that walk QT query
// tiles per wave, so K / V^T are still staged once per (sample, head) but TWO workgroups share a CU (2 x 66 KB
// LDS, 8 waves).  The two are not synchronised with each other, so one's softmax (VALU/exp bound: ~2x the MFMA
// time of a tile) overlaps the other's MFMAs and staging -- with one 8-wave workgroup per CU every wave was in
// the same phase.
template <int KT, int NW, int QT>
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
        const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
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
    }
}

template <int KT, int NW, int QT>
void launch_attn1(const bf16* qk, const bf16* vt, bf16* att, int batch, int ntok, int heads, hipStream_t s) {
    constexpr int KC = KT * 32;
    const int lds = KC * 128 + 64 * (KC * 2 + 8);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn1_kernel<KT, NW, QT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((attn1_kernel<KT, NW, QT>), dim3(1, heads, batch), dim3(NW * 64), lds, s, qk, vt, att, ntok,
                       heads * 64);
}

template <int KT, int NW>
void launch_kt(const bf16* qk, const bf16* vt, bf16* att, int batch, int ntok, int heads, hipStream_t s) {
    constexpr int KC = KT * 32;
    static const bool dbuf_on = !(getenv("TLD_ATTN_DBUF") && atoi(getenv("TLD_ATTN_DBUF")) == 0);     // A/B knob
    const int nbuf = (ntok > KC && dbuf_on && 2 * (KC * 128 + 64 * (KC * 2 + 8)) <= 160 * 1024) ? 2 : 1;
    const int lds = nbuf * (KC * 128 + 64 * (KC * 2 + 8));
    static int attr_lds = 0;
    if (attr_lds < lds) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<KT, NW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_lds = lds;
    }
    dim3 grid(ntok / (NW * 32), heads, batch), block(NW * 64);
    hipLaunchKernelGGL((attn_kernel<KT, NW>), grid, block, lds, s, qk, vt, att, ntok, heads * 64, nbuf);
}

}  // namespace

void launch_attention(const bf16* qk, const bf16* vt, bf16* att, int batch, int ntok, int heads,
                      hipStream_t s) {
    // 256-key chunks staged once per workgroup of 8 waves (256 query rows).  Measured alternative: 4-wave
    // workgroups, two per CU (staging overlapped with compute) -- 88 us vs 60 us per layer at C1, the K/V
    // chunk is then staged twice per head and the extra L2->LDS traffic costs more than the overlap buys.
    static const bool one_wg = getenv("TLD_ATTN_8W") && atoi(getenv("TLD_ATTN_8W")) != 0;     // A/B knob: single 8-wave workgroup per CU
    if (ntok == 256 && !one_wg) launch_attn1<8, 4, 2>(qk, vt, att, batch, ntok, heads, s);
    else if (ntok % 256 == 0) launch_kt<8, 8>(qk, vt, att, batch, ntok, heads, s);
    else if (ntok == 128) launch_kt<4, 4>(qk, vt, att, batch, ntok, heads, s);
    else if (ntok == 64) launch_kt<2, 2>(qk, vt, att, batch, ntok, heads, s);
    else if (ntok == 32) launch_kt<1, 1>(qk, vt, att, batch, ntok, heads, s);
    // other token counts are rejected in tld_engine_create
}

}  // namespace tld
