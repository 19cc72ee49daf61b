// tld_attn_core.h -- one wave's share of a 256-key self-attention head with q, k, v^T already resident in LDS.
//
// softmax(q k^T / sqrt(64)) v for 32 query rows against 256 keys, head_dim 64 (tld/transformer_blocks.py:37-44: non-causal, no mask,
// no dropout in eval).  Used by the fused QKV -> attention epilogue of the GEMM (tld_gemm.hip, EPI_QKV_ATTN), where the operands come
// straight from the accumulators of the head's 256 x 192 (q_h | k_h | v_h) tile and never travel through HBM.  Same arithmetic as
// attn1_kernel (tld_attn.hip): scores TRANSPOSED (S^T = K Q^T, a lane owns one query column), row max / row sum lane-local plus one
// lane^32 exchange, exp2 output converted in place into the B operand of O^T = V^T P^T, V^T gathered with the P operand's key order.
//
// LDS images (written by the caller):
//   K   [256 keys][64 features] bf16, 128-byte rows, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7)
//   Q   [256 queries][64]       bf16, same image
//   V^T [64 features][256 keys] bf16, row pitch 520 bytes (512 + 8: the 32 lanes of an 8-byte read hit 32 distinct bank pairs)
#pragma once
#include "tld_common.h"

namespace tld {

constexpr int kAttnVPitch = 256 * 2 + 8;
constexpr float kAttnScaleLog2e = 0.125f * 1.44269504088896340736f;   // (1 / sqrt(64)) * log2(e)

// o[ct][r]: O^T accumulators (feature tile ct, lane = query l31); returns 1 / (softmax denominator) of the lane's query
__device__ __forceinline__ float attn256_wave(const char* Ks, const char* Vs, const bf16x8 (&qf)[4], f32x16 (&o)[2], int l31, int hi) {
    constexpr int KT = 8;
    f32x16 st[KT];
    {
        auto kfrag = [&](int t, int ks) {
            const int row = t * 32 + l31;
            const int kc = ks * 2 + hi;
            return *reinterpret_cast<const bf16x8*>(Ks + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
        };
        // step s = (tile pair s >> 1, k half s & 1): fragments K[2p][2h], K[2p+1][2h], K[2p][2h+1], K[2p+1][2h+1]; reads one step ahead
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
    }
    float mx = st[0][0];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = mx * kAttnScaleLog2e;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float l_run = 0.f;
    auto vfrag = [&](int s2, int ct) {                 // step s2 = 2 t + hf: keys s2 * 16 + hi * 4 .. (+3, +8 .. +11)
        const char* vp = Vs + (ct * 32 + l31) * kAttnVPitch + (s2 * 16 + hi * 4) * 2;
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
            const float pv = __builtin_amdgcn_exp2f(st[s2 >> 1][(s2 & 1) * 8 + e] * kAttnScaleLog2e - m_new);
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
    return 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
}

}  // namespace tld
