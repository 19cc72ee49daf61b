// tld_train_attn.hip -- backward of the denoiser's self-attention for the training step.
//
// Forward (tld/transformer_blocks.py:24-48): per (sample, head), O = softmax(Q K^T / 8) V over N = 256 tokens, head_dim 64, no mask.
// Backward, given dO:   P = softmax(Q K^T / 8);  dV = P^T dO;  dP = dO V^T;  dS = P o (dP - delta),  delta_q = dO_q . O_q;
//                       dQ = dS K / 8;  dK = dS^T Q / 8.
// One 4-wave workgroup per (sample, head); Q, K, V, dO of the head live in LDS as row-major [256][64] bf16 (144-byte pitch).
// Seven 256 x 256 x 64-class products on v_mfma_f32_32x32x16_bf16, in two passes so that nothing is ever transposed through memory:
//   pass 1 (a wave owns 64 queries):  S^T = K Q^T and dP^T = V dO^T  -- a lane owns a QUERY column, so the softmax statistics and delta are
//           lane-local (+ one lane^32 exchange) -- then dS^T, and dQ^T += K^T dS^T with dS^T going straight from the accumulators into
//           the MFMA's B operand: its 8 values per lane are the keys {(r & 3) + 8 (r >> 2) + 4 hi}, r = 8 s .. 8 s + 7, of the 32-key tile,
//           and the A operand (K^T) is gathered from the row-major K with the same key order, so the product is exact.
//   pass 2 (a wave owns 64 keys):     S = Q K^T and dP = dO V^T (a lane owns a KEY column), P = exp2(S c - L_q) with the row statistics
//           L_q left in LDS by pass 1, then dV^T += dO^T P and dK^T += Q^T dS the same way (A operands gathered from dO / Q).
// Outputs are row-major bf16 [M, 3 d] (dq | dk | dv), the layout the weight- and input-gradient GEMMs consume.
#include "tld_common.h"

namespace tld {

namespace {

constexpr int kN = 256;                 // tokens
constexpr int kPitch = 144;             // bytes per LDS row: 64 bf16 + 16 B pad
constexpr int kMat = kN * kPitch;       // one operand image
constexpr int kLdsBytes = 4 * kMat + 2 * kN * 4;

__device__ __forceinline__ bf16x8 frag(const char* base, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(base + row * kPitch + chunk * 16);
}
// 8 elements of column `col` at the rows  row0 + (r & 3) + 8 (r >> 2) + 4 hi,  r = 8 s + e  (the accumulator row order of a 32 x 32 tile)
__device__ __forceinline__ bf16x8 gfrag(const char* base, int col, int row0, int hi, int s) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = 8 * s + e;
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        v[e] = *reinterpret_cast<const bf16*>(base + row * kPitch + col * 2);
    }
    return v;
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)a[8 * s + e];
    return v;
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(const bf16* __restrict__ qk, const bf16* __restrict__ vt, const bf16* __restrict__ o,
                                                       const float* __restrict__ g, bf16* __restrict__ dqkv, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem; char* sK = smem + kMat; char* sV = smem + 2 * kMat; char* sG = smem + 3 * kMat;
    float* sL = reinterpret_cast<float*>(smem + 4 * kMat);
    float* sD = sL + kN;
    const int d = H * 64;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, hi = lane >> 5;

    // ---- stage: thread t owns token t.  q, k rows from qk [M, 2 d]; v column-wise from V^T [B, H, 64, N]; dO (fp32) and O rows.
    {
        const size_t row = (size_t)b * kN + tid;
        const u32x4* pq = reinterpret_cast<const u32x4*>(qk + row * 2 * d + h * 64);
        const u32x4* pk = reinterpret_cast<const u32x4*>(qk + row * 2 * d + d + h * 64);
        const u32x4* po = reinterpret_cast<const u32x4*>(o + row * d + h * 64);
        const float4* pg = reinterpret_cast<const float4*>(g + row * d + h * 64);
        float delta = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            *reinterpret_cast<u32x4*>(sQ + tid * kPitch + c * 16) = pq[c];
            *reinterpret_cast<u32x4*>(sK + tid * kPitch + c * 16) = pk[c];
            const u32x4 ov = po[c];
            const bf16x8 ob = __builtin_bit_cast(bf16x8, ov);
            const float4 g0 = pg[2 * c], g1 = pg[2 * c + 1];
            const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            bf16x8 gb;
#pragma unroll
            for (int e = 0; e < 8; ++e) { gb[e] = (bf16)gv[e]; delta = fmaf(gv[e], (float)ob[e], delta); }
            *reinterpret_cast<bf16x8*>(sG + tid * kPitch + c * 16) = gb;
        }
        sD[tid] = delta;
        const bf16* pv = vt + ((size_t)b * H + h) * 64 * kN + tid;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) *reinterpret_cast<bf16*>(sV + tid * kPitch + i * 2) = pv[(size_t)i * kN];
    }
    __syncthreads();

    const float c2 = 0.125f * 1.44269504088896340736f;       // 1/8 scale in log2 units
    // ---- pass 1: queries [64 wid, 64 wid + 64)
    for (int qt = 2 * wid; qt < 2 * wid + 2; ++qt) {
        f32x16 st[8];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sK, kt * 32 + l31, kc * 2 + hi), frag(sQ, qt * 32 + l31, kc * 2 + hi), st[kt], 0, 0, 0);
        }
        float m = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kt][r] *= c2; m = fmaxf(m, st[kt][r]); }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kt][r] = exp2f(st[kt][r] - m); sum += st[kt][r]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (hi == 0) sL[qt * 32 + l31] = m + log2f(sum);
        const float dq_ = sD[qt * 32 + l31];
        f32x16 dq[2];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dh][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            f32x16 dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sV, kt * 32 + l31, kc * 2 + hi), frag(sG, qt * 32 + l31, kc * 2 + hi), dp, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = st[kt][r] * inv * (dp[r] - dq_);            // dS^T
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 bfrag = pack8(dp, s);
#pragma unroll
                for (int dh = 0; dh < 2; ++dh)
                    dq[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfrag(sK, dh * 32 + l31, kt * 32, hi, s), bfrag, dq[dh], 0, 0, 0);
            }
        }
        // dQ^T tile: rows (registers) = dims, columns (lanes) = queries -> row-major dq, 4 consecutive dims per store
        bf16* dst = dqkv + ((size_t)b * kN + qt * 32 + l31) * 3 * d + h * 64;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (bf16)(dq[dh][rq * 4 + e] * 0.125f);
                *reinterpret_cast<bf16x4*>(dst + dh * 32 + 8 * rq + 4 * hi) = v;
            }
    }
    __syncthreads();            // every query's L is in LDS

    // ---- pass 2: keys [64 wid, 64 wid + 64)
    for (int kt = 2 * wid; kt < 2 * wid + 2; ++kt) {
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dh][r] = 0.f; dv[dh][r] = 0.f; }
        for (int qt = 0; qt < 8; ++qt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sQ, qt * 32 + l31, kc * 2 + hi), frag(sK, kt * 32 + l31, kc * 2 + hi), s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sG, qt * 32 + l31, kc * 2 + hi), frag(sV, kt * 32 + l31, kc * 2 + hi), dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float p = exp2f(s[r] * c2 - sL[qi]);
                s[r] = p;                                   // P
                dp[r] = p * (dp[r] - sD[qi]);               // dS
            }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const bf16x8 pf = pack8(s, sl), dsf = pack8(dp, sl);
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    dv[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfrag(sG, dh * 32 + l31, qt * 32, hi, sl), pf, dv[dh], 0, 0, 0);
                    dk[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfrag(sQ, dh * 32 + l31, qt * 32, hi, sl), dsf, dk[dh], 0, 0, 0);
                }
            }
        }
        bf16* dst = dqkv + ((size_t)b * kN + kt * 32 + l31) * 3 * d + h * 64;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                bf16x4 vk, vv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vk[e] = (bf16)(dk[dh][rq * 4 + e] * 0.125f); vv[e] = (bf16)dv[dh][rq * 4 + e]; }
                *reinterpret_cast<bf16x4*>(dst + d + dh * 32 + 8 * rq + 4 * hi) = vk;
                *reinterpret_cast<bf16x4*>(dst + 2 * d + dh * 32 + 8 * rq + 4 * hi) = vv;
            }
    }
}

}  // namespace

// qk [M, 2 d] (q | k) and vt [B, H, 64, N]: the forward's saved operands;  o [M, d]: the forward's output;  g [M, d] fp32: dL/dO;
// dqkv [M, 3 d] bf16 out.  N must be 256.
int launch_attention_bwd(const bf16* qk, const bf16* vt, const bf16* o, const float* g, bf16* dqkv, int batch, int ntok, int heads, hipStream_t s) {
    if (ntok != kN) return 1;
    static PerDeviceOnce once;
    if (once.first()) hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(batch * heads), dim3(256), kLdsBytes, s, qk, vt, o, g, dqkv, heads);
    return 0;
}

}  // namespace tld
