// tld_train_attn.hip -- backward of the denoiser's self-attention for the training step.
//
// Forward (tld/transformer_blocks.py:24-48): per (sample, head), O = softmax(Q K^T / 8) V over N tokens, head_dim 64, no mask.
// Backward, given dO:   P = softmax(Q K^T / 8);  dV = P^T dO;  dP = dO V^T;  dS = P o (dP - delta),  delta_q = dO_q . O_q;
//                       dQ = dS K / 8;  dK = dS^T Q / 8.
// A workgroup of NW waves works on blocks of BT = 32 NW tokens: Q, K, dO of a block live in LDS as row-major [BT][64] bf16 (144-byte
// pitch) and V as the forward left it, V^T [64][BT].  Seven BT x BT x 64-class products on v_mfma_f32_32x32x16_bf16, in two passes:
//   pass 1 (a wave owns 32 queries):  S^T = K Q^T and dP^T = V dO^T -- a lane owns a QUERY column, so the softmax statistics and delta are
//           lane-local (+ one lane^32 exchange) -- then dS^T, and dQ^T += K^T dS^T with dS^T going straight from the accumulators into
//           the MFMA's B operand: its 8 values per lane are the keys {(r & 3) + 8 (r >> 2) + 4 hi}, r = 8 s .. 8 s + 7, of the 32-key tile.
//   pass 2 (a wave owns 32 keys):     S = Q K^T and dP = dO V^T (a lane owns a KEY column), P = exp2(S c - L_q) with the row statistics
//           L_q of pass 1, then dV^T += dO^T P and dK^T += Q^T dS the same way.
// Every operand that a product needs TRANSPOSED (K^T, Q^T, dO^T in the accumulators' key / query order; V rows out of V^T) is read
// from the one row-major image with gfx950's transposing LDS read (ds_read_b64_tr_b16: lane i of a 16-lane group receives
// img[k0 + j][m0 + i], j = 0..3, from the 4 x 16 block whose rows the group's lanes address) -- nothing is transposed through memory,
// and no image is kept twice.
//   N = BT (64, 128, 256 tokens): one workgroup per (sample, head) runs both passes; L and delta stay in LDS.
//   N = NB x 256:  kernel 1, one workgroup per (sample, head, query block): a first sweep over the key blocks for the row statistics,
//           a second for dQ; L and delta go to a scratch vector.  Kernel 2, one workgroup per (sample, head, key block), sweeps the
//           query blocks for dK and dV.
// Outputs are row-major bf16 [M, 3 d] (dq | dk | dv), the layout the weight- and input-gradient GEMMs consume.
#include "tld_common.h"

// attribution builds (tools/attn_bwd_bench.py): -DTLD_AB_DBG=<bits>: 1 no staging loads, 2 no pass 1, 4 no pass 2, 8 no output stores
#ifndef TLD_AB_DBG
#define TLD_AB_DBG 0
#endif

namespace tld {

namespace {

constexpr int kPitch = 144;             // bytes per LDS row of a token-major image: 64 bf16 + 16 B pad
constexpr float kC2 = 0.125f * 1.44269504088896340736f;       // 1/8 scale in log2 units
enum { ATTN_FUSED = 0, ATTN_DQ = 1, ATTN_DKV = 2 };

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int NW> struct Lay {
    static constexpr int BT = 32 * NW;                 // tokens per block
    static constexpr int PV = BT * 2 + 16;             // pitch of the V^T image
    static constexpr int MAT = BT * kPitch;
    static constexpr int Q = 0, K = MAT, G = 2 * MAT, V = 3 * MAT, L = V + 64 * PV, D = L + BT * 4, BYTES = D + BT * 4;
};

__device__ __forceinline__ bf16x8 frag(const char* base, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(base + row * kPitch + chunk * 16);
}
// lane i of every 16-lane group: img[k0 + j][m0 + i], j = 0..3 (img row-major bf16, `pitch` bytes per row, m0 a multiple of 4)
__device__ __forceinline__ s16x4 tr4(const char* img, int pitch, int k0, int m0, int lane) {
    const int s = lane & 15;
    const char* p = img + (k0 + (s >> 2)) * pitch + (m0 + 4 * (s & 3)) * 2;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 join(s16x4 a, s16x4 b) {
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// MFMA operand (lane = column col0 + l31 of the image): rows row0 + 8 hi + {0 .. 7}
__device__ __forceinline__ bf16x8 tr_seq(const char* img, int pitch, int row0, int col0, int lane) {
    const int m0 = col0 + 16 * ((lane >> 4) & 1), r = row0 + 8 * (lane >> 5);
    return join(tr4(img, pitch, r, m0, lane), tr4(img, pitch, r + 4, m0, lane));
}
// the same with rows row0 + 4 hi + {0 .. 3, 8 .. 11}: the accumulator row order of 16 rows of a 32 x 32 tile
__device__ __forceinline__ bf16x8 tr_acc(const char* img, int pitch, int row0, int col0, int lane) {
    const int m0 = col0 + 16 * ((lane >> 4) & 1), r = row0 + 4 * (lane >> 5);
    return join(tr4(img, pitch, r, m0, lane), tr4(img, pitch, r + 8, m0, lane));
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)a[8 * s + e];
    return v;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
// a 32-token x 64-dim tile held transposed (registers = dims (r & 3) + 8 (r >> 2) + 4 hi of each 32-dim half, lanes = tokens) -> row-major bf16
__device__ __forceinline__ void store_tile(bf16* dst /* row of token l31, first dim */, const f32x16 (&t)[2], float scale, int hi) {
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (bf16)(t[dh][rq * 4 + e] * scale);
            *reinterpret_cast<bf16x4*>(dst + dh * 32 + 8 * rq + 4 * hi) = v;
        }
}

template <int NW, int MODE, typename TG>      // TG: dL/dO as fp32 (the test hook) or bf16 (the training step: the LayerNorm-2 backward leaves a bf16 copy of the residual gradient)
__global__ __launch_bounds__(64 * NW) void attn_bwd_kernel(const bf16* __restrict__ qk, const bf16* __restrict__ vt, const bf16* __restrict__ o,
                                                          const TG* __restrict__ g, bf16* __restrict__ dqkv, float* __restrict__ stats, int H, int N) {
    using Ly = Lay<NW>;
    constexpr int BT = Ly::BT, PV = Ly::PV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem + Ly::Q; char* sK = smem + Ly::K; char* sG = smem + Ly::G; char* sV = smem + Ly::V;
    float* sL = reinterpret_cast<float*>(smem + Ly::L);
    float* sD = reinterpret_cast<float*>(smem + Ly::D);
    const int NB = N / BT, d = H * 64;
    int bid = blockIdx.x;
    const int blk = bid % NB; bid /= NB;
    const int h = bid % H, b = bid / H;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const size_t row0 = (size_t)b * N;
    float* Lg = stats + ((size_t)(b * H + h) * 2) * N;        // [L | delta] of this (sample, head): only the multi-block kernels touch it
    float* Dg = Lg + N;

    // ---- staging: two threads per token row, four 16-byte chunks each
    const int srow = tid >> 1, sc0 = (tid & 1) * 4;
    auto stage_rows = [&](char* img, const bf16* src, int ld, int tb) {
        const u32x4* p = reinterpret_cast<const u32x4*>(src + (row0 + (size_t)tb * BT + srow) * ld) + sc0;
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(img + srow * kPitch + (sc0 + c) * 16) = p[c];
    };
    auto stage_g = [&](int qb, bool with_delta) {          // dO -> bf16 image; delta = dO . O per row
        const size_t row = row0 + (size_t)qb * BT + srow;
        const u32x4* po = reinterpret_cast<const u32x4*>(o + row * d + h * 64) + sc0;
        float delta = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 gb;
            float gv[8];
            if constexpr (sizeof(TG) == 4) {
                const float4* pg = reinterpret_cast<const float4*>(g + row * d + h * 64) + sc0 * 2;
                const float4 g0 = pg[2 * c], g1 = pg[2 * c + 1];
                const float t[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) { gv[e] = t[e]; gb[e] = (bf16)t[e]; }
            } else {
                gb = __builtin_bit_cast(bf16x8, (reinterpret_cast<const u32x4*>(g + row * d + h * 64) + sc0)[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[e] = (float)gb[e];
            }
            if (with_delta) {
                const bf16x8 ob = __builtin_bit_cast(bf16x8, po[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) delta = fmaf(gv[e], (float)ob[e], delta);
            }
            *reinterpret_cast<bf16x8*>(sG + srow * kPitch + (sc0 + c) * 16) = gb;
        }
        if (with_delta) {
            delta += __shfl_xor(delta, 1, 64);
            if ((tid & 1) == 0) {
                sD[srow] = delta;
                if (MODE == ATTN_DQ) Dg[qb * BT + srow] = delta;
            }
        }
    };
    auto stage_v = [&](int kb) {                           // V^T rows (dims) x BT keys
        constexpr int CPR = BT / 8;                        // 16-byte chunks per row
        const bf16* base = vt + ((size_t)(b * H + h) * 64) * N + (size_t)kb * BT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + j * 2 * BT, r = idx / CPR, c = idx % CPR;
            *reinterpret_cast<u32x4*>(sV + r * PV + c * 16) = *reinterpret_cast<const u32x4*>(base + (size_t)r * N + c * 8);
        }
    };
    const bf16* qsrc = qk + h * 64;
    const bf16* ksrc = qk + d + h * 64;

    // ================================ pass 1: queries [32 wid, 32 wid + 32) of block `blk` ================================
    if (MODE != ATTN_DKV) {
        if (!(TLD_AB_DBG & 1)) {
            stage_rows(sQ, qsrc, 2 * d, blk);
            stage_g(blk, true);
            if (MODE == ATTN_FUSED) { stage_rows(sK, ksrc, 2 * d, 0); stage_v(0); }
        }
        __syncthreads();
        if (!(TLD_AB_DBG & 2)) {
        const int q0 = 32 * wid;
        bf16x8 qf[4], gf[4];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) { qf[kc] = frag(sQ, q0 + l31, kc * 2 + hi); gf[kc] = frag(sG, q0 + l31, kc * 2 + hi); }
        const float dlt = sD[q0 + l31];
        f32x16 st[NW];
        auto scores = [&]() {
#pragma unroll
            for (int kt = 0; kt < NW; ++kt) {
                st[kt] = zero16();
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
                    st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sK, kt * 32 + l31, kc * 2 + hi), qf[kc], st[kt], 0, 0, 0);
            }
        };
        f32x16 dq[2] = {zero16(), zero16()};
        auto dq_accum = [&]() {                             // st holds P^T of the staged key block
#pragma unroll
            for (int kt = 0; kt < NW; ++kt) {
                f32x16 dp = zero16();
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_seq(sV, PV, 16 * kc, kt * 32, lane), gf[kc], dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = st[kt][r] * (dp[r] - dlt);            // dS^T
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 bfrag = pack8(dp, s);
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh)
                        dq[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_acc(sK, kPitch, kt * 32 + 16 * s, dh * 32, lane), bfrag, dq[dh], 0, 0, 0);
                }
            }
        };
        if (MODE == ATTN_FUSED) {
            scores();
            float m = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < NW; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, st[kt][r]);
            m = fmaxf(m, __shfl_xor(m, 32, 64)) * kC2;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NW; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { st[kt][r] = __builtin_amdgcn_exp2f(st[kt][r] * kC2 - m); sum += st[kt][r]; }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int kt = 0; kt < NW; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kt][r] *= inv;
            if (hi == 0) sL[q0 + l31] = m + __builtin_amdgcn_logf(sum);        // v_log_f32 is log2
            dq_accum();
        } else {
            // sweep 1: row statistics over all key blocks
            float m = -3.0e38f, sum = 0.f;
            for (int kb = 0; kb < NB; ++kb) {
                if (kb) __syncthreads();
                stage_rows(sK, ksrc, 2 * d, kb);
                __syncthreads();
                scores();
                float mb = -3.0e38f;
#pragma unroll
                for (int kt = 0; kt < NW; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mb = fmaxf(mb, st[kt][r]);
                mb = fmaxf(mb, __shfl_xor(mb, 32, 64)) * kC2;
                const float mn = fmaxf(m, mb);
                float part = 0.f;
#pragma unroll
                for (int kt = 0; kt < NW; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part += __builtin_amdgcn_exp2f(st[kt][r] * kC2 - mn);
                sum = sum * __builtin_amdgcn_exp2f(m - mn) + part;
                m = mn;
            }
            sum += __shfl_xor(sum, 32, 64);
            const float Lq = m + __builtin_amdgcn_logf(sum);
            if (hi == 0) Lg[blk * BT + q0 + l31] = Lq;
            // sweep 2: dQ
            for (int kb = 0; kb < NB; ++kb) {
                __syncthreads();
                stage_rows(sK, ksrc, 2 * d, kb);
                stage_v(kb);
                __syncthreads();
                scores();
#pragma unroll
                for (int kt = 0; kt < NW; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kt][r] = __builtin_amdgcn_exp2f(st[kt][r] * kC2 - Lq);
                dq_accum();
            }
        }
        if (!(TLD_AB_DBG & 8)) store_tile(dqkv + (row0 + (size_t)blk * BT + q0 + l31) * 3 * d + h * 64, dq, 0.125f, hi);
        }
    }
    if (MODE == ATTN_DQ) return;
    if (TLD_AB_DBG & 4) return;

    // ================================ pass 2: keys [32 wid, 32 wid + 32) of block `blk` ================================
    if (MODE == ATTN_DKV) { stage_rows(sK, ksrc, 2 * d, blk); stage_v(blk); }
    __syncthreads();            // FUSED: every query's L is in LDS
    {
        const int k0 = 32 * wid;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) { kf[kc] = frag(sK, k0 + l31, kc * 2 + hi); vf[kc] = tr_seq(sV, PV, 16 * kc, k0, lane); }
        f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
        for (int qb = 0; qb < NB; ++qb) {
            if (MODE == ATTN_DKV) {
                if (qb) __syncthreads();
                stage_rows(sQ, qsrc, 2 * d, qb);
                stage_g(qb, false);
                if (tid < BT) { sL[tid] = Lg[qb * BT + tid]; sD[tid] = Dg[qb * BT + tid]; }
                __syncthreads();
            }
#pragma unroll 2
            for (int qt = 0; qt < NW; ++qt) {
                f32x16 s = zero16(), dp = zero16();
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sQ, qt * 32 + l31, kc * 2 + hi), kf[kc], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sG, qt * 32 + l31, kc * 2 + hi), vf[kc], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float p = __builtin_amdgcn_exp2f(s[r] * kC2 - sL[qi]);
                    s[r] = p;                                   // P
                    dp[r] = p * (dp[r] - sD[qi]);               // dS
                }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const bf16x8 pf = pack8(s, sl), dsf = pack8(dp, sl);
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh) {
                        dv[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_acc(sG, kPitch, qt * 32 + 16 * sl, dh * 32, lane), pf, dv[dh], 0, 0, 0);
                        dk[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_acc(sQ, kPitch, qt * 32 + 16 * sl, dh * 32, lane), dsf, dk[dh], 0, 0, 0);
                    }
                }
            }
        }
        bf16* dst = dqkv + (row0 + (size_t)blk * BT + k0 + l31) * 3 * d + h * 64;
        if (!(TLD_AB_DBG & 8)) {
            store_tile(dst + d, dk, 0.125f, hi);
            store_tile(dst + 2 * d, dv, 1.0f, hi);
        }
    }
}

template <int NW, int MODE, typename TG>
void launch_one(const bf16* qk, const bf16* vt, const bf16* o, const TG* g, bf16* dqkv, float* stats, int batch, int ntok, int heads, hipStream_t s) {
    static PerDeviceOnce once;
    once.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<NW, MODE, TG>), hipFuncAttributeMaxDynamicSharedMemorySize, Lay<NW>::BYTES); });
    hipLaunchKernelGGL((attn_bwd_kernel<NW, MODE, TG>), dim3(batch * heads * (ntok / (32 * NW))), dim3(64 * NW), Lay<NW>::BYTES, s, qk, vt, o, g, dqkv, stats, heads,
                       ntok);
}

template <typename TG>
int launch_bwd(const bf16* qk, const bf16* vt, const bf16* o, const TG* g, bf16* dqkv, float* stats, int batch, int ntok, int heads, hipStream_t s) {
    if (ntok == 256) launch_one<8, ATTN_FUSED>(qk, vt, o, g, dqkv, stats, batch, ntok, heads, s);
    else if (ntok == 128) launch_one<4, ATTN_FUSED>(qk, vt, o, g, dqkv, stats, batch, ntok, heads, s);
    else if (ntok == 64) launch_one<2, ATTN_FUSED>(qk, vt, o, g, dqkv, stats, batch, ntok, heads, s);
    else if (ntok > 256 && ntok % 256 == 0 && stats) {
        launch_one<8, ATTN_DQ>(qk, vt, o, g, dqkv, stats, batch, ntok, heads, s);
        launch_one<8, ATTN_DKV>(qk, vt, o, g, dqkv, stats, batch, ntok, heads, s);
    } else return 1;
    return 0;
}

}  // namespace

// qk [M, 2 d] (q | k) and vt [B, H, 64, N]: the forward's saved operands;  o [M, d]: the forward's output;  g [M, d]: dL/dO (fp32 or bf16);
// dqkv [M, 3 d] bf16 out.  N = 64, 128 or a multiple of 256; `stats` = 2 B H N floats of scratch, touched only when N > 256.
int launch_attention_bwd(const bf16* qk, const bf16* vt, const bf16* o, const float* g, bf16* dqkv, float* stats, int batch, int ntok, int heads, hipStream_t s) {
    return launch_bwd<float>(qk, vt, o, g, dqkv, stats, batch, ntok, heads, s);
}
int launch_attention_bwd(const bf16* qk, const bf16* vt, const bf16* o, const bf16* g, bf16* dqkv, float* stats, int batch, int ntok, int heads, hipStream_t s) {
    return launch_bwd<bf16>(qk, vt, o, g, dqkv, stats, batch, ntok, heads, s);
}

}  // namespace tld
