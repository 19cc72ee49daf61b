// tld_train_kernels.h -- device kernels of the training step (tld/train.py:118-175): the row / elementwise /
// reduction pieces of the forward-with-saved-activations and of the backward pass.  The dense contractions (projections and their
// dX / dW products) run on the MFMA GEMM of tld_gemm.hip; self-attention backward is tld_train_attn.hip.  Included by tld_train.hip only.
//
// Conventions: a wave owns a token row where a row reduction is needed (lane l holds features l + 64 j); fp32 arithmetic throughout,
// bf16 only as the storage type of the large saved activations.  Parameter-gradient reductions over the batch are two-stage
// (fixed-size partials, then a sum in a fixed order), so a step is bit-reproducible.
#pragma once
#include "tld_common.h"

namespace tld {
namespace train {

constexpr int kMaxJ = 16;          // features per lane: d <= 1024
constexpr float kEps = 1e-5f;

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return (float)*p; }
__device__ __forceinline__ f32x4 ldf4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ldf4(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// exact-erf GELU and its derivative (nn.GELU default; tld/denoiser.py:108, tld/transformer_blocks.py:103)
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// GELU(x) and GELU'(x) together, erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 the results are stored in): the
// polynomial's exp(-z^2), z = x / sqrt(2), is the Gaussian of the derivative's density term, so the pair costs one v_exp and one v_rcp
// (erff + expf made the separate GELU'-multiply kernel of the backward VALU-bound at 91 us per block).
__device__ __forceinline__ void gelu_pair(float x, float& g, float& gp) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float E = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);          // exp(-x^2 / 2)
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float half_erfc = 0.5f * poly * E;                                           // 0.5 erfc(|z|)
    const float phi = x >= 0.f ? 1.0f - half_erfc : half_erfc;                        // Phi(x)
    g = x * phi;
    gp = fmaf(x * 0.39894228040143267794f, E, phi);
}

// ---- small fp32 products (conditioning path, kv projection of the cross-attention, out-projection pieces) ------------------------------
// C[i, j] (+)= bias[j] + sum_k A(i, k) B(j, k)   with   A(i, k) = A[i sai + k sak],  B(j, k) = B[j sbj + k sbk]:  one kernel serves
//   forward   out[r, n] = b[n] + sum_k in[r, k] W[n, k]            (A = in, B = W)
//   dx[r, k'] = sum_n dy[r, n] W[n, k']                            (A = dy, B(k', n) = W[n K + k'])
//   dW[n, k'] = sum_r dy[r, n] x[r, k']                            (A(n, r) = dy[r ldy + n], B(k', r) = x[r ldx + k'])
// 32 x 32 output tile per 256-thread workgroup (2 x 2 per thread), 32-deep K-steps through LDS; exact fp32 FMA chains in k order.
__global__ __launch_bounds__(256) void tiled_f32_kernel(const float* __restrict__ A, long sai, long sak, const float* __restrict__ B, long sbj, long sbk,
                                                        const float* __restrict__ bias, float* __restrict__ C, int ldc, int I, int J, int K,
                                                        float* __restrict__ pre, int gelu, int accumulate, long za = 0, long zb = 0, long zc = 0) {
    // blockIdx.z: a batch of equally shaped products (the 12 blocks' kv projections of the conditioning tokens in one launch: each is 0.6 GFLOP and
    // latency-bound on its own, 36 us per launch), operand / output z at element offsets z za, z zb, z zc
    A += (long)blockIdx.z * za; B += (long)blockIdx.z * zb; C += (long)blockIdx.z * zc;
    __shared__ float sa[32][33], sb[32][33];
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    // the next K-step's operands are fetched into registers before this step's FMAs (round 3: every step used to expose a global-memory
    // round trip between two barriers -- 91 us for products of 0.6 GFLOP); same FMA chains in k order
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = threadIdx.x + u * 256;
            // the faster-varying index of a load follows the operand's unit stride so that both layouts read coalesced
            int ii, kk;
            if (sak == 1) { kk = t & 31; ii = t >> 5; } else { ii = t & 31; kk = t >> 5; }
            ra[u] = (i0 + ii < I && k0 + kk < K) ? A[(long)(i0 + ii) * sai + (long)(k0 + kk) * sak] : 0.f;
            int jj, k2;
            if (sbk == 1) { k2 = t & 31; jj = t >> 5; } else { jj = t & 31; k2 = t >> 5; }
            rb[u] = (j0 + jj < J && k0 + k2 < K) ? B[(long)(j0 + jj) * sbj + (long)(k0 + k2) * sbk] : 0.f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = threadIdx.x + u * 256;
            int ii, kk;
            if (sak == 1) { kk = t & 31; ii = t >> 5; } else { ii = t & 31; kk = t >> 5; }
            sa[kk][ii] = ra[u];
            int jj, k2;
            if (sbk == 1) { k2 = t & 31; jj = t >> 5; } else { jj = t & 31; k2 = t >> 5; }
            sb[k2][jj] = rb[u];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        stash();
        __syncthreads();
        if (k0 + 32 < K) fetch(k0 + 32);
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float a0 = sa[kk][ty * 2], a1 = sa[kk][ty * 2 + 1], b0 = sb[kk][tx * 2], b1 = sb[kk][tx * 2 + 1];
            acc[0][0] = fmaf(a0, b0, acc[0][0]); acc[0][1] = fmaf(a0, b1, acc[0][1]);
            acc[1][0] = fmaf(a1, b0, acc[1][0]); acc[1][1] = fmaf(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int i = i0 + ty * 2 + a, j = j0 + tx * 2 + b;
            if (i >= I || j >= J) continue;
            float v = acc[a][b] + (bias ? bias[j] : 0.f);
            float* o = C + (size_t)i * ldc + j;
            if (pre) pre[(size_t)i * ldc + j] = v;
            if (gelu) v = gelu_exact(v);
            *o = accumulate ? *o + v : v;
        }
}
// db[n] (+)= sum_r dy[r, n]
__global__ void small_colsum(const float* __restrict__ dy, int ldy, float* __restrict__ db, int R, int N, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += dy[(size_t)r * ldy + n];
    db[n] = accumulate ? db[n] + s : s;
}
__global__ void mul_gelu_grad(float* __restrict__ g, const float* __restrict__ pre, int n) {     // g *= GELU'(pre)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] *= gelu_grad(pre[i]);
}

// ---- LayerNorm (eps 1e-5, biased variance, affine; tld/transformer_blocks.py:131-133, tld/denoiser.py:42,44,113) -----------------
// forward: one wave per row;  out = (x - mean) rstd gamma + beta  [+ add[row % add_rows]]  as bf16 and / or fp32;  stats = (mean, rstd)
template <typename TX>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     bf16* __restrict__ out_bf, float* __restrict__ out_f, float2* __restrict__ stats,
                                                     const float* __restrict__ add, int add_rows, int M, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int J = d >> 6;
    float v[kMaxJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) { v[j] = ldf(x + (size_t)row * d + lane + 64 * j); s += v[j]; }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) { const float c = v[j] - mean; q = fmaf(c, c, q); }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + kEps);
    if (lane == 0 && stats) stats[row] = make_float2(mean, rstd);
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) {
        const int c = lane + 64 * j;
        float o = (v[j] - mean) * rstd * gamma[c] + beta[c];
        if (add) o += add[(size_t)(row % add_rows) * d + c];
        if (out_bf) out_bf[(size_t)row * d + c] = (bf16)o;
        if (out_f) out_f[(size_t)row * d + c] = o;
    }
}
// d = 256 NQ forms of the two row kernels of a block's forward (lane l owns features {4 l .. 4 l + 3} + 256 j: 8-byte accesses; the 2-byte
// forms ran at 3.6 TB/s): x_out = x_in + delta (bf16-rounded, as the reference's bf16 residual stream would be) and, when a_out is given,
// a_out = LN(x_out) with its (mean, rstd);  delta == nullptr: plain LayerNorm of x_in (x_out is not written).
template <int NQ>
__global__ __launch_bounds__(256) void resid_add_ln_q4_kernel(const bf16* __restrict__ x_in, const bf16* __restrict__ delta, bf16* __restrict__ x_out,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ a_out,
                                                              float2* __restrict__ stats, int M) {
    constexpr int d = NQ * 256;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    f32x4 v[NQ];
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const size_t o = (size_t)row * d + j * 256 + 4 * lane;
        if (delta) {
            const f32x4 t = ldf4(x_in + o) + ldf4(delta + o);
            bf16x4 r;
            r[0] = (bf16)t[0]; r[1] = (bf16)t[1]; r[2] = (bf16)t[2]; r[3] = (bf16)t[3];
            *reinterpret_cast<bf16x4*>(x_out + o) = r;
            v[j] = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
        } else {
            v[j] = ldf4(x_in + o);
        }
        s4 += v[j];
    }
    if (!a_out) return;
    const float mean = wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) / (float)d;
    f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NQ; ++j) { const f32x4 c = v[j] - mean; q4 = __builtin_elementwise_fma(c, c, q4); }
    const float rstd = rsqrtf(wave_sum((q4[0] + q4[1]) + (q4[2] + q4[3])) / (float)d + kEps);
    if (lane == 0) stats[row] = make_float2(mean, rstd);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int c = j * 256 + 4 * lane;
        const f32x4 o = (v[j] - mean) * rstd * ldf4(gamma + c) + ldf4(beta + c);
        bf16x4 r;
        r[0] = (bf16)o[0]; r[1] = (bf16)o[1]; r[2] = (bf16)o[2]; r[3] = (bf16)o[3];
        *reinterpret_cast<bf16x4*>(a_out + (size_t)row * d + c) = r;
    }
}
// backward: dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma;  written (accumulate = 0) or added (1) into dx (fp32);
// per-workgroup partial sums of dgamma = sum dy xhat and dbeta = sum dy over the workgroup's rows -> part[blockIdx][2][d]
template <typename TDY, typename TX>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x, const float2* __restrict__ stats,
                                                     const float* __restrict__ gamma, float* __restrict__ dx, int accumulate,
                                                     float* __restrict__ part, int rows_per_block, int M, int d, bf16* __restrict__ dx_bf16 = nullptr) {
    __shared__ float red[4][2][1024];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int J = d >> 6;
    float dg[kMaxJ], db[kMaxJ];
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) { dg[j] = 0.f; db[j] = 0.f; }
    const int r0 = blockIdx.x * rows_per_block;
    for (int rr = wid; rr < rows_per_block; rr += 4) {
        const int row = r0 + rr;
        if (row >= M) break;
        const float2 st = stats[row];
        float g[kMaxJ], xh[kMaxJ];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxJ; ++j) if (j < J) {
            const int c = lane + 64 * j;
            const float dyv = ldf(dy + (size_t)row * d + c);
            xh[j] = (ldf(x + (size_t)row * d + c) - st.x) * st.y;
            g[j] = dyv * gamma[c];
            s1 += g[j]; s2 = fmaf(g[j], xh[j], s2);
            dg[j] = fmaf(dyv, xh[j], dg[j]); db[j] += dyv;
        }
        const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int j = 0; j < kMaxJ; ++j) if (j < J) {
            const float o = st.y * (g[j] - m1 - xh[j] * m2);
            float* px = dx + (size_t)row * d + lane + 64 * j;
            const float v = accumulate ? *px + o : o;
            *px = v;
            if (dx_bf16) dx_bf16[(size_t)row * d + lane + 64 * j] = (bf16)v;      // the next block's GEMM operand (was a separate cast pass)
        }
    }
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) { red[wid][0][lane + 64 * j] = dg[j]; red[wid][1][lane + 64 * j] = db[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        part[((size_t)blockIdx.x * 2 + 0) * d + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
        part[((size_t)blockIdx.x * 2 + 1) * d + c] = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
    }
}
// d % 256 == 0 form of the same kernel: lane l owns features {4 l .. 4 l + 3} + 256 j, i.e. 8- / 16-byte accesses instead of 2 / 4
// (115 -> see profiles: the 2-byte form ran at 2.3 TB/s).  Same arithmetic per element; the row means are summed in a different lane order.
template <typename TDY, typename TX, int NQ>
__global__ __launch_bounds__(256) void ln_bwd_q4_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x, const float2* __restrict__ stats,
                                                        const float* __restrict__ gamma, float* __restrict__ dx, int accumulate,
                                                        float* __restrict__ part, int rows_per_block, int M, bf16* __restrict__ dx_bf16 = nullptr) {
    constexpr int d = NQ * 256;
    __shared__ float red[4][2][d];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 dg[NQ], db[NQ], gm[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) { dg[j] = f32x4{0.f, 0.f, 0.f, 0.f}; db[j] = dg[j]; gm[j] = ldf4(gamma + j * 256 + 4 * lane); }
    const int r0 = blockIdx.x * rows_per_block;
    for (int rr = wid; rr < rows_per_block; rr += 4) {
        const int row = r0 + rr;
        if (row >= M) break;
        const float2 st = stats[row];
        f32x4 g[NQ], xh[NQ];
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const size_t o = (size_t)row * d + j * 256 + 4 * lane;
            const f32x4 dyv = ldf4(dy + o);
            xh[j] = (ldf4(x + o) - st.x) * st.y;
            g[j] = dyv * gm[j];
            s1 += g[j]; s2 = __builtin_elementwise_fma(g[j], xh[j], s2);
            dg[j] = __builtin_elementwise_fma(dyv, xh[j], dg[j]); db[j] += dyv;
        }
        const float m1 = wave_sum((s1[0] + s1[1]) + (s1[2] + s1[3])) / (float)d, m2 = wave_sum((s2[0] + s2[1]) + (s2[2] + s2[3])) / (float)d;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            f32x4 o = (g[j] - m1 - xh[j] * m2) * st.y;
            f32x4* px = reinterpret_cast<f32x4*>(dx + (size_t)row * d + j * 256 + 4 * lane);
            if (accumulate) o += *px;
            *px = o;
            if (dx_bf16) {
                bf16x4 ob;
                ob[0] = (bf16)o[0]; ob[1] = (bf16)o[1]; ob[2] = (bf16)o[2]; ob[3] = (bf16)o[3];
                *reinterpret_cast<bf16x4*>(dx_bf16 + (size_t)row * d + j * 256 + 4 * lane) = ob;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        *reinterpret_cast<f32x4*>(&red[wid][0][j * 256 + 4 * lane]) = dg[j];
        *reinterpret_cast<f32x4*>(&red[wid][1][j * 256 + 4 * lane]) = db[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        part[((size_t)blockIdx.x * 2 + 0) * d + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
        part[((size_t)blockIdx.x * 2 + 1) * d + c] = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
    }
}
// out[c] (+)= sum_i part[i * stride + c]   (fixed order: bit-reproducible).  16 columns x 64 part-lanes per workgroup: lane l sums the parts
// l, l + 64, ... and an LDS tree finishes -- the outputs are few (a weight's size), the parts up to thousands.
__global__ __launch_bounds__(1024) void reduce_partials(const float* __restrict__ part, int nparts, size_t stride, float* __restrict__ out, int n, int accumulate) {
    // 16 columns x 64 part-lanes per workgroup (round 3; was 64 x 16): the outputs are a weight's size (a few thousand), so column blocks of
    // 64 left 24 workgroups walking 1024 parts in 64 dependent steps each -- 30 us per call, 118 calls per step
    __shared__ float red[64][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    float a = 0.f;
    if (c < n) for (int i = ty; i < nparts; i += 64) a += part[(size_t)i * stride + c];
    red[ty][tx] = a;
    __syncthreads();
    if (ty < 4) {                                  // four part-lane groups of 16, then the last four: fixed order
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[ty * 16 + l][tx];
        red[ty * 16][tx] = t;
    }
    __syncthreads();
    if (ty == 0 && c < n) {
        const float t = (red[0][tx] + red[16][tx]) + (red[32][tx] + red[48][tx]);
        out[c] = accumulate ? out[c] + t : t;
    }
}
// out[c] = sum_{i < nparts} part[i * stride + c] for FEW parts and MANY outputs (the split-K slices of a weight gradient): a thread per four
// outputs, slices added in order.  n % 4 == 0, stride % 4 == 0.
__global__ void sum_slices(const float* __restrict__ part, int nparts, size_t stride, float* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = reinterpret_cast<const f32x4*>(part)[i];
    for (int k = 1; k < nparts; ++k) a += reinterpret_cast<const f32x4*>(part + (size_t)k * stride)[i];
    reinterpret_cast<f32x4*>(out)[i] = a;
}
// [C][9] -> [9][C]  (tap-major copy of a depthwise weight)
__global__ void dw_tapmajor_kernel(const float* __restrict__ w, float* __restrict__ out, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 9) return;
    const int c = i / 9, k = i - c * 9;
    out[(size_t)k * C + c] = w[i];
}
// column sums of a [M, C] matrix over row chunks: part[chunk][C]
template <typename T>
__global__ void colsum_partial(const T* __restrict__ a, int M, int C, int rows_per_chunk, float* __restrict__ part) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += ldf(a + (size_t)r * C + c);
    part[(size_t)blockIdx.y * C + c] = s;
}
// C % 4 == 0 form: a thread owns four adjacent columns (8- / 16-byte loads; the 2-byte form above ran at 2 TB/s on the MLP bias gradients)
template <typename T>
__global__ void colsum4_partial(const T* __restrict__ a, int M, int C, int rows_per_chunk, float* __restrict__ part) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) s += ldf4(a + (size_t)r * C + c);
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.y * C + c) = s;
}

// LayerNorm backward over NARROW rows (width <= 64: the patch LayerNorm, tld/denoiser.py:42): one thread per row;
// dgamma / dbeta partials per 256-row workgroup -> part[blockIdx][2][width]
__global__ __launch_bounds__(256) void ln_small_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float2* __restrict__ stats,
                                                           const float* __restrict__ gamma, float* __restrict__ dx, float* __restrict__ part, int M, int width) {
    __shared__ float red[256][33];
    const int row = blockIdx.x * 256 + threadIdx.x;
    // pass over the columns in groups of 16 so that the shared tile stays small: (dy xhat, dy) pairs of 16 columns
    float xh[64], g[64];
    float m1 = 0.f, m2 = 0.f;
    float2 st = make_float2(0.f, 0.f);
    if (row < M) {
        st = stats[row];
        for (int c = 0; c < width; ++c) {
            xh[c] = (x[(size_t)row * width + c] - st.x) * st.y;
            g[c] = dy[(size_t)row * width + c] * gamma[c];
            m1 += g[c]; m2 = fmaf(g[c], xh[c], m2);
        }
        m1 /= (float)width; m2 /= (float)width;
        for (int c = 0; c < width; ++c) dx[(size_t)row * width + c] = st.y * (g[c] - m1 - xh[c] * m2);
    }
    for (int c0 = 0; c0 < width; c0 += 16) {
        const int nc = min(16, width - c0);
        for (int c = 0; c < nc; ++c) {
            const float dyv = row < M ? dy[(size_t)row * width + c0 + c] : 0.f;
            red[threadIdx.x][c] = row < M ? dyv * xh[c0 + c] : 0.f;
            red[threadIdx.x][16 + c] = dyv;
        }
        __syncthreads();
        if (threadIdx.x < 2 * nc) {
            const int which = threadIdx.x / nc, c = threadIdx.x % nc;
            float a = 0.f;
            for (int r = 0; r < 256; ++r) a += red[r][which * 16 + c];
            part[((size_t)blockIdx.x * 2 + which) * width + c0 + c] = a;
        }
        __syncthreads();
    }
}
__global__ void transpose_f32_small(const float* __restrict__ in /*[R, C]*/, float* __restrict__ out /*[C, R]*/, int R, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * C) return;
    const int r = i / C, c = i - r * C;
    out[(size_t)c * R + r] = in[i];
}

// ---- residual stream --------------------------------------------------------------------------------------------------------
// x_out = bf16(x_in + delta)  (the stored stream, as at inference);  optionally the next LayerNorm of the STORED row in the same pass
__global__ __launch_bounds__(256) void resid_add_ln_kernel(const bf16* __restrict__ x_in, const bf16* __restrict__ delta, bf16* __restrict__ x_out,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ a_out,
                                                           float2* __restrict__ stats, int M, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int J = d >> 6;
    float v[kMaxJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) {
        const size_t o = (size_t)row * d + lane + 64 * j;
        const bf16 r = (bf16)((float)x_in[o] + (float)delta[o]);
        x_out[o] = r;
        v[j] = (float)r; s += v[j];
    }
    if (!a_out) return;
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) { const float c = v[j] - mean; q = fmaf(c, c, q); }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + kEps);
    if (lane == 0) stats[row] = make_float2(mean, rstd);
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) {
        const int c = lane + 64 * j;
        a_out[(size_t)row * d + c] = (bf16)((v[j] - mean) * rstd * gamma[c] + beta[c]);
    }
}

// ---- layout helpers -----------------------------------------------------------------------------------------------------------
// [R, C] -> [C, R] (bf16 out), 32 x 32 tiles through LDS;  TIN = bf16 or float (rounded once)
template <typename TIN>
__global__ __launch_bounds__(256) void transpose_to_bf16(const TIN* __restrict__ in, int ldi, bf16* __restrict__ out, int ldo, int R, int C, int splits = 1,
                                                         bf16* __restrict__ copy = nullptr /* optional [R, C] bf16 copy of the input (the weight refresh: one pass for both operand forms) */) {
    // splits > 1 (split-K operand of a weight-gradient GEMM): the R rows are cut into `splits` equal runs and the output is the stack
    // [split][C][R / splits] -- ldo = R / splits then
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        const float v = (r < R && c < C) ? ldf(in + (size_t)r * ldi + c) : 0.f;
        tile[i][tx] = v;
        if (copy && r < R && c < C) copy[(size_t)r * C + c] = (bf16)v;
    }
    __syncthreads();
    const int rs = R / splits;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) {
            const int sp = r / rs;
            out[((size_t)sp * C + c) * ldo + (r - sp * rs)] = (bf16)tile[tx][i];
        }
    }
}
// bf16 [R, C] -> bf16 [C, R] in 64 x 64 tiles, 16-byte global accesses both ways (R, C multiples of 64; split-K stacking as above)
__global__ __launch_bounds__(256) void transpose_bf16_64(const bf16* __restrict__ in, int ldi, bf16* __restrict__ out, int ldo, int R, int C, int splits) {
    // the runs are ldo rows long; splits * ldo may exceed R (runs padded to a multiple of 128 so that any split count fits): the grid covers
    // the padded rows and the blocks past R write zeros
    __shared__ bf16 tile[64][72];                      // 144-byte pitch: 16-byte aligned rows
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = it * 256 + threadIdx.x;        // 64 rows x 8 chunks
        const int r = idx >> 3, ch = idx & 7;
        *reinterpret_cast<uint4*>(&tile[r][ch * 8]) = r0 < R ? *reinterpret_cast<const uint4*>(in + (size_t)(r0 + r) * ldi + c0 + ch * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const int rs = ldo;
    const int sp = r0 / rs;
    (void)splits;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = it * 256 + threadIdx.x;        // 64 output rows (input columns) x 8 chunks of 8 input rows
        const int c = idx >> 3, ch = idx & 7;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[ch * 8 + e][c];
        *reinterpret_cast<bf16x8*>(out + ((size_t)sp * C + c0 + c) * ldo + (r0 - sp * rs) + ch * 8) = v;
    }
}

// ---- patch embedding (tld/denoiser.py:34-45,75-77) ------------------------------------------------------------------------------
// forward, one wave per token: conv(k = s = patch) -> p [pd] -> LN(pd) -> Linear(pd, d) -> e [d] -> LN(d) -> + pos -> x0 (bf16)
struct EmbedTrain {
    const float* x;            // [B, C, S, S] fp32
    const float *conv_w, *conv_b, *ln1_w, *ln1_b, *lin_w /*[d, pd]*/, *lin_b, *ln2_w, *ln2_b, *pos /*[N, d]*/;
    float* p;                  // [M, pd]  conv output
    float* pn;                 // [M, pd]  after LN(pd)
    float2* st1;               // [M]
    float* e;                  // [M, d]   Linear output
    float2* st2;               // [M]
    bf16* x0;                  // [M, d]
    int B, C, S, patch, grid, pd, d;
};
__global__ __launch_bounds__(256) void embed_fwd_kernel(EmbedTrain q) {
    const int N = q.grid * q.grid, M = q.B * N;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int b = row / N, t = row - b * N, gy = t / q.grid, gx = t - gy * q.grid;
    const int pp = q.patch * q.patch, taps = q.C * pp;
    // lane f < pd: feature f of the patch convolution
    float pv = 0.f;
    if (lane < q.pd) {
        pv = q.conv_b[lane];
        for (int k = 0; k < taps; ++k) {
            const int c = k / pp, r = k - c * pp, p1 = r / q.patch, p2 = r - p1 * q.patch;
            pv = fmaf(q.conv_w[lane * taps + k], q.x[(((size_t)b * q.C + c) * q.S + gy * q.patch + p1) * q.S + gx * q.patch + p2], pv);
        }
        q.p[(size_t)row * q.pd + lane] = pv;
    }
    const float m1 = wave_sum(lane < q.pd ? pv : 0.f) / (float)q.pd;
    const float c1 = lane < q.pd ? pv - m1 : 0.f;
    const float r1 = rsqrtf(wave_sum(c1 * c1) / (float)q.pd + kEps);
    const float pn = lane < q.pd ? c1 * r1 * q.ln1_w[lane] + q.ln1_b[lane] : 0.f;
    if (lane < q.pd) q.pn[(size_t)row * q.pd + lane] = pn;
    if (lane == 0) q.st1[row] = make_float2(m1, r1);
    const int J = q.d >> 6;
    float ev[kMaxJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) {
        const int c = lane + 64 * j;
        float a = q.lin_b[c];
        for (int f = 0; f < q.pd; ++f) a = fmaf(__shfl(pn, f, 64), q.lin_w[c * q.pd + f], a);
        ev[j] = a; s += a;
        q.e[(size_t)row * q.d + c] = a;
    }
    const float m2 = wave_sum(s) / (float)q.d;
    float v2 = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) { const float c = ev[j] - m2; v2 = fmaf(c, c, v2); }
    const float r2 = rsqrtf(wave_sum(v2) / (float)q.d + kEps);
    if (lane == 0) q.st2[row] = make_float2(m2, r2);
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) {
        const int c = lane + 64 * j;
        q.x0[(size_t)row * q.d + c] = (bf16)((ev[j] - m2) * r2 * q.ln2_w[c] + q.ln2_b[c] + q.pos[(size_t)t * q.d + c]);
    }
}
// The same with the Linear's weight staged transposed in LDS ([pd][d] fp32, <= 64 KB) and a lane owning 4 CONSECUTIVE features per 256-feature
// block: 16-byte LDS reads / global stores instead of 64-byte-strided weight loads per token (763 -> ~60 us at the training shape).  Waves walk the
// token rows with a grid stride.  Same arithmetic order per output as embed_fwd_kernel (bias first, features f ascending).
template <int JB>      // 256-feature blocks: d <= 256 JB
__global__ __launch_bounds__(256) void embed_fwd_lds_kernel(EmbedTrain q) {
    extern __shared__ __attribute__((aligned(16))) float lw[];     // [pd][d]
    const int N = q.grid * q.grid, M = q.B * N, d = q.d, pd = q.pd;
    for (int i = threadIdx.x; i < pd * d; i += 256) { const int c = i / pd, f = i - c * pd; lw[f * d + c] = q.lin_w[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pp = q.patch * q.patch, taps = q.C * pp;
    float4 lb[JB], g2[JB], b2[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int c0 = 4 * lane + 256 * j;
        if (c0 < d) { lb[j] = *reinterpret_cast<const float4*>(q.lin_b + c0); g2[j] = *reinterpret_cast<const float4*>(q.ln2_w + c0); b2[j] = *reinterpret_cast<const float4*>(q.ln2_b + c0); }
        else lb[j] = g2[j] = b2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float cb = lane < pd ? q.conv_b[lane] : 0.f, l1w = lane < pd ? q.ln1_w[lane] : 0.f, l1b = lane < pd ? q.ln1_b[lane] : 0.f;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const int b = row / N, t = row - b * N, gy = t / q.grid, gx = t - gy * q.grid;
        float pv = 0.f;
        if (lane < pd) {
            pv = cb;
            for (int k = 0; k < taps; ++k) {
                const int c = k / pp, r = k - c * pp, p1 = r / q.patch, p2 = r - p1 * q.patch;
                pv = fmaf(q.conv_w[lane * taps + k], q.x[(((size_t)b * q.C + c) * q.S + gy * q.patch + p1) * q.S + gx * q.patch + p2], pv);
            }
            q.p[(size_t)row * pd + lane] = pv;
        }
        const float m1 = wave_sum(lane < pd ? pv : 0.f) / (float)pd;
        const float c1 = lane < pd ? pv - m1 : 0.f;
        const float r1 = rsqrtf(wave_sum(c1 * c1) / (float)pd + kEps);
        const float pn = lane < pd ? c1 * r1 * l1w + l1b : 0.f;
        if (lane < pd) q.pn[(size_t)row * pd + lane] = pn;
        if (lane == 0) q.st1[row] = make_float2(m1, r1);
        float4 ev[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j) ev[j] = lb[j];
        for (int f = 0; f < pd; ++f) {
            const float pf = __shfl(pn, f, 64);
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const int c0 = 4 * lane + 256 * j;
                if (c0 < d) {
                    const float4 w = *reinterpret_cast<const float4*>(lw + f * d + c0);
                    ev[j].x = fmaf(pf, w.x, ev[j].x); ev[j].y = fmaf(pf, w.y, ev[j].y); ev[j].z = fmaf(pf, w.z, ev[j].z); ev[j].w = fmaf(pf, w.w, ev[j].w);
                }
            }
        }
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int c0 = 4 * lane + 256 * j;
            if (c0 < d) { *reinterpret_cast<float4*>(q.e + (size_t)row * d + c0) = ev[j]; sm += (ev[j].x + ev[j].y) + (ev[j].z + ev[j].w); }
        }
        const float m2 = wave_sum(sm) / (float)d;
        float v2 = 0.f;
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int c0 = 4 * lane + 256 * j;
            if (c0 < d) {
                const float a = ev[j].x - m2, bq = ev[j].y - m2, c = ev[j].z - m2, dd = ev[j].w - m2;
                v2 = fmaf(a, a, v2); v2 = fmaf(bq, bq, v2); v2 = fmaf(c, c, v2); v2 = fmaf(dd, dd, v2);
            }
        }
        const float r2 = rsqrtf(wave_sum(v2) / (float)d + kEps);
        if (lane == 0) q.st2[row] = make_float2(m2, r2);
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int c0 = 4 * lane + 256 * j;
            if (c0 < d) {
                const float4 ps = *reinterpret_cast<const float4*>(q.pos + (size_t)t * d + c0);
                bf16x4 o;
                o[0] = (bf16)((ev[j].x - m2) * r2 * g2[j].x + b2[j].x + ps.x);
                o[1] = (bf16)((ev[j].y - m2) * r2 * g2[j].y + b2[j].y + ps.y);
                o[2] = (bf16)((ev[j].z - m2) * r2 * g2[j].z + b2[j].z + ps.z);
                o[3] = (bf16)((ev[j].w - m2) * r2 * g2[j].w + b2[j].w + ps.w);
                *reinterpret_cast<bf16x4*>(q.x0 + (size_t)row * d + c0) = o;
            }
        }
    }
}
// dpos[t, c] = sum_b g[b N + t, c]
__global__ void pos_grad_kernel(const float* __restrict__ g, float* __restrict__ dpos, int B, int N, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * d) return;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += g[(size_t)b * N * d + i];
    dpos[i] = a;
}
// gather the patch pixels of every token as a matrix [M, C p p] (operand of the conv-weight gradient)
__global__ void patches_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C, int S, int patch, int grid) {
    const int pp = patch * patch, taps = C * pp, N = grid * grid;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * N * taps) return;
    const int k = (int)(i % taps);
    const size_t m = i / taps;
    const int b = (int)(m / N), t = (int)(m % N), gy = t / grid, gx = t - gy * grid;
    const int c = k / pp, r = k - c * pp, p1 = r / patch, p2 = r - p1 * patch;
    out[i] = x[(((size_t)b * C + c) * S + gy * patch + p1) * S + gx * patch + p2];
}

// ---- cross-attention over the two conditioning tokens (tld/transformer_blocks.py:62-72) -----------------------------------------------
// One workgroup per (sample, head), one thread per token.  kv [B, 2, 2d] fp32 = (k | v) of tokens (noise, label).
// forward:  p0 = softmax([q.k0, q.k1] / 8)[0];  out = p0 v0 + (1 - p0) v1
// (round 3: eight lanes per token, 16-byte accesses -- a thread per token with 2-byte loads / stores ran at 157 / 177 us per launch)
__global__ __launch_bounds__(256) void cross_fwd_kernel(const bf16* __restrict__ q, const float* __restrict__ kv, bf16* __restrict__ out,
                                                        float* __restrict__ p0_out, int N, int d) {
    const int H = d >> 6, b = blockIdx.x / H, h = blockIdx.x % H;
    const int sub = threadIdx.x & 7, tl = threadIdx.x >> 3;      // feature octet of the head, token slot (32 tokens per pass)
    float k0[8], k1[8], v0[8], v1[8];
    {
        const float* kb0 = kv + ((size_t)b * 2 + 0) * 2 * d + h * 64 + sub * 8;
        const float* kb1 = kv + ((size_t)b * 2 + 1) * 2 * d + h * 64 + sub * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { k0[e] = kb0[e]; k1[e] = kb1[e]; v0[e] = kb0[d + e]; v1[e] = kb1[d + e]; }
    }
    for (int t = tl; t < N; t += 32) {
        const size_t row = (size_t)b * N + t;
        const bf16x8 q8 = *reinterpret_cast<const bf16x8*>(q + row * d + h * 64 + sub * 8);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float qi = (float)q8[e]; s0 = fmaf(qi, k0[e], s0); s1 = fmaf(qi, k1[e], s1); }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) { s0 += __shfl_xor(s0, m, 64); s1 += __shfl_xor(s1, m, 64); }
        const float p0 = 1.0f / (1.0f + __expf((s1 - s0) * 0.125f));
        if (sub == 0) p0_out[row * H + h] = p0;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(p0 * v0[e] + (1.0f - p0) * v1[e]);
        *reinterpret_cast<bf16x8*>(out + row * d + h * 64 + sub * 8) = o;
    }
}
// backward:  g = dL/dout (fp32 [M, d]).  dq -> bf16 [M, d];  dkv[b, t, :] (fp32, written: one workgroup owns a (sample, head) slice)
// Round 4: the token reductions (dK of token 0 = -dK of token 1, dV0, dV1) are accumulated in the same pass that produces dq -- every thread keeps
// the partial sums of its feature octet over its token slot, 32 slots are then added in a fixed order through LDS -- instead of a second pass that
// re-read g and q with 4- / 2-byte loads (67 us per launch).
__global__ __launch_bounds__(256) void cross_bwd_kernel(const float* __restrict__ g, const bf16* __restrict__ q, const float* __restrict__ kv,
                                                        const float* __restrict__ p0_in, bf16* __restrict__ dq, float* __restrict__ dkv, int N, int d) {
    const int H = d >> 6, b = blockIdx.x / H, h = blockIdx.x % H;
    __shared__ float red[32][3][64];              // [token slot][dk0 | dv0 | dv1][feature]
    const int sub = threadIdx.x & 7, tl = threadIdx.x >> 3;
    float kd[8], v0[8], v1[8];
    {
        const float* kb0 = kv + ((size_t)b * 2 + 0) * 2 * d + h * 64 + sub * 8;
        const float* kb1 = kv + ((size_t)b * 2 + 1) * 2 * d + h * 64 + sub * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { kd[e] = kb0[e] - kb1[e]; v0[e] = kb0[d + e]; v1[e] = kb1[d + e]; }
    }
    float ak[8], a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ak[e] = 0.f; a0[e] = 0.f; a1[e] = 0.f; }
    for (int t = tl; t < N; t += 32) {
        const size_t row = (size_t)b * N + t;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(g + row * d + h * 64 + sub * 8);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(g + row * d + h * 64 + sub * 8 + 4);
        const bf16x8 q8 = *reinterpret_cast<const bf16x8*>(q + row * d + h * 64 + sub * 8);
        const float gv[8] = {ga[0], ga[1], ga[2], ga[3], gb[0], gb[1], gb[2], gb[3]};
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { d0 = fmaf(gv[e], v0[e], d0); d1 = fmaf(gv[e], v1[e], d1); }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) { d0 += __shfl_xor(d0, m, 64); d1 += __shfl_xor(d1, m, 64); }
        const float p0 = p0_in[row * H + h];
        const float s0 = p0 * (1.0f - p0) * (d0 - d1) * 0.125f;            // dL/ds0 = -dL/ds1, with the 1/8 score scale
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = (bf16)(s0 * kd[e]);
            ak[e] = fmaf(s0, (float)q8[e], ak[e]);
            a0[e] = fmaf(p0, gv[e], a0[e]);
            a1[e] = fmaf(1.0f - p0, gv[e], a1[e]);
        }
        *reinterpret_cast<bf16x8*>(dq + row * d + h * 64 + sub * 8) = o;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tl][0][sub * 8 + e] = ak[e]; red[tl][1][sub * 8 + e] = a0[e]; red[tl][2][sub * 8 + e] = a1[e]; }
    __syncthreads();
    if (threadIdx.x < 192) {
        const int i = threadIdx.x & 63, k = threadIdx.x >> 6;
        float t = 0.f;
#pragma unroll 8
        for (int sl = 0; sl < 32; ++sl) t += red[sl][k][i];
        float* o0 = dkv + ((size_t)b * 2 + 0) * 2 * d;
        float* o1 = dkv + ((size_t)b * 2 + 1) * 2 * d;
        if (k == 0) { o0[h * 64 + i] = t; o1[h * 64 + i] = -t; }
        else if (k == 1) o0[d + h * 64 + i] = t;
        else o1[d + h * 64 + i] = t;
    }
}

// ---- depthwise 3x3 (zero "same" padding, cross-correlation) + GELU (tld/transformer_blocks.py:96-103) --------------------------------
// channels-last image [B, G, G, C] == token-major [M, C].  w: TAP-MAJOR copy [9][C] fp32 of the reference's [C, 1, 3, 3] (refreshed with the
// bf16 GEMM operands), so that a thread's 8 channels are two 16-byte loads per tap.
// forward (flip = 0): out = b + sum_taps w[c][ky][kx] in[y + ky - 1][x + kx - 1];  also gelu_out = GELU(out) when given.
// input gradient (flip = 1, no bias): din[y][x] = sum_taps w[c][ky][kx] dout[y - ky + 1][x - kx + 1]
// One workgroup per (sample, 64-channel slab, band of `R` image rows): the band's tokens plus one halo row on either side are staged in
// LDS with 16-byte copies (the whole 16 x 16 image, 32 KB, at the training shape; 18 rows of 32 / 64 tokens at the fine-tuning
// sizes), then a thread (channel quad, image row) slides a rotating 3 x 3 fp32 register window along the row (the inference path's
// dwconv_gelu_kernel recipe): every input value is read from HBM once (halo rows twice).
inline int dwconv_band_rows(int G) { return G < 16 ? G : 16; }
inline size_t dwconv_lds_bytes(int G) { const int R = dwconv_band_rows(G); return (size_t)(R + 2 < G ? R + 2 : G) * G * 128; }
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, bf16* __restrict__ out,
                                                     bf16* __restrict__ gelu_out, int B, int G, int C, int flip, int R) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];   // [rows ylo .. yhi][G tokens][64 ch] bf16
    const int nslab = C >> 6, nband = (G + R - 1) / R;
    int bid = blockIdx.x;
    const int band = bid % nband; bid /= nband;
    const int b = bid / nslab, cc = bid - b * nslab;
    const int ntok = G * G;
    const int y0 = band * R, y1 = y0 + R < G ? y0 + R : G;          // rows computed here
    const int ylo = y0 > 0 ? y0 - 1 : 0, yhi = y1 < G ? y1 + 1 : G; // rows staged
    char* smem = smem_ - (size_t)ylo * G * 128;                     // indexed with absolute token numbers below
    const bf16* src = in + (size_t)b * ntok * C + cc * 64;
    {   // global -> LDS DMA, 8 tokens x 128 B per instruction, every piece of a wave in flight at once (a loop of load + ds_write pairs exposed
        // one memory round trip per iteration: 113 us per launch at the training shape)
        typedef const __attribute__((address_space(1))) void* gp_t;
        typedef __attribute__((address_space(3))) void* lp_t;
        const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int t0 = ylo * G, pieces = ((yhi - ylo) * G + 7) >> 3;
        for (int pc = wid; pc < pieces; pc += 4) {
            int t = t0 + pc * 8 + (lane >> 3);
            t = t < yhi * G ? t : yhi * G - 1;
            __builtin_amdgcn_global_load_lds((gp_t)(src + (size_t)t * C + (lane & 7) * 8), (lp_t)(smem + (size_t)(t0 + pc * 8) * 128), 16, 0, 0);
        }
    }
    const int cq = threadIdx.x & 15;
    const int c0 = cc * 64 + cq * 4;
    float4 wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const float4*>(w + (size_t)(flip ? 8 - k : k) * C + c0);    // flipped taps = reversed order
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && !flip) bs = *reinterpret_cast<const float4*>(bias + c0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = y0 + (threadIdx.x >> 4); i < y1; i += 16) {
        const bool up_ok = i > 0, dn_ok = i + 1 < G;
        auto load_col = [&](int j, float4 (&col)[3]) {
            const bool jok = j >= 0 && j < G;
#pragma unroll
            for (int du = 0; du < 3; ++du) {
                const bool ok = jok && (du == 1 || (du == 0 ? up_ok : dn_ok));
                if (ok) {
                    const bf16x4 v = *reinterpret_cast<const bf16x4*>(smem + ((i + du - 1) * G + j) * 128 + cq * 8);
                    col[du] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                } else {
                    col[du] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        const size_t obase = ((size_t)b * ntok + (size_t)i * G) * C + c0;
        auto emit = [&](const float4 (&L)[3], const float4 (&Mc)[3], const float4 (&R)[3], int j) {
            float4 a = bs;
#pragma unroll
            for (int du = 0; du < 3; ++du) {
                const float4 w0 = wt[du * 3 + 0], w1 = wt[du * 3 + 1], w2 = wt[du * 3 + 2];
                a.x = fmaf(w2.x, R[du].x, fmaf(w1.x, Mc[du].x, fmaf(w0.x, L[du].x, a.x)));
                a.y = fmaf(w2.y, R[du].y, fmaf(w1.y, Mc[du].y, fmaf(w0.y, L[du].y, a.y)));
                a.z = fmaf(w2.z, R[du].z, fmaf(w1.z, Mc[du].z, fmaf(w0.z, L[du].z, a.z)));
                a.w = fmaf(w2.w, R[du].w, fmaf(w1.w, Mc[du].w, fmaf(w0.w, L[du].w, a.w)));
            }
            bf16x4 o;
            if (gelu_out) {                                   // forward: `out` receives GELU'(pre-activation), all the backward needs of it (round 4; it was the
                const float av[4] = {a.x, a.y, a.z, a.w};     // pre-activation itself, differentiated by a separate VALU-bound pass), `gelu_out` GELU(.)
                bf16x4 g;
#pragma unroll
                for (int e = 0; e < 4; ++e) { float gv, gd; gelu_pair(av[e], gv, gd); g[e] = (bf16)gv; o[e] = (bf16)gd; }
                *reinterpret_cast<bf16x4*>(gelu_out + obase + (size_t)j * C) = g;
            } else {
                o[0] = (bf16)a.x; o[1] = (bf16)a.y; o[2] = (bf16)a.z; o[3] = (bf16)a.w;
            }
            *reinterpret_cast<bf16x4*>(out + obase + (size_t)j * C) = o;
        };
        float4 c0v[3], c1v[3], c2v[3];
        load_col(-1, c0v);
        load_col(0, c1v);
        int j = 0;
        for (; j + 3 <= G; j += 3) {
            load_col(j + 1, c2v); emit(c0v, c1v, c2v, j);
            load_col(j + 2, c0v); emit(c1v, c2v, c0v, j + 1);
            load_col(j + 3, c1v); emit(c2v, c0v, c1v, j + 2);
        }
        if (j < G) { load_col(j + 1, c2v); emit(c0v, c1v, c2v, j); ++j; }
        if (j < G) { load_col(j + 1, c0v); emit(c1v, c2v, c0v, j); }
    }
}
// dhc = dg * gp, gp = GELU'(pre-activation) as the forward stored it (grids the fused backward below does not take)
__global__ void gelu_bwd_kernel(const bf16* __restrict__ dg, const bf16* __restrict__ gp, bf16* __restrict__ out, size_t n8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const bf16x8 g = reinterpret_cast<const bf16x8*>(dg)[i], h = reinterpret_cast<const bf16x8*>(gp)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)g[e] * (float)h[e]);
    reinterpret_cast<bf16x8*>(out)[i] = o;
}
// weight / bias gradient partials per (sample, image row): part[b G + y][10][C]: taps 0..8, then the bias;  thread per channel, a 3 x 3
// register window of `in` slides along x (3 new loads per position instead of 9)
__global__ void dwconv_wgrad_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ in, float* __restrict__ part, int G, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const size_t by = blockIdx.y;                 // b * G + y
    const int y = (int)(by % G);
    const size_t b = by / G;
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    auto ld = [&](int yy, int xx) -> float {
        return ((unsigned)yy < (unsigned)G && (unsigned)xx < (unsigned)G) ? (float)in[((b * G + yy) * G + xx) * C + c] : 0.f;
    };
    float w0[3], w1[3], w2[3];                     // columns x - 1, x, x + 1 of rows y - 1 .. y + 1
#pragma unroll
    for (int r = 0; r < 3; ++r) { w0[r] = 0.f; w1[r] = ld(y + r - 1, 0); }
    for (int x = 0; x < G; ++x) {
#pragma unroll
        for (int r = 0; r < 3; ++r) w2[r] = ld(y + r - 1, x + 1);
        const float gv = (float)dout[((b * G + y) * G + x) * C + c];
        acc[9] += gv;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            acc[r * 3 + 0] = fmaf(gv, w0[r], acc[r * 3 + 0]);
            acc[r * 3 + 1] = fmaf(gv, w1[r], acc[r * 3 + 1]);
            acc[r * 3 + 2] = fmaf(gv, w2[r], acc[r * 3 + 2]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) { w0[r] = w1[r]; w1[r] = w2[r]; }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) part[(by * 10 + k) * C + c] = acc[k];
}
// The whole backward of GELU + depthwise 3x3 for grids up to 16 x 16 in ONE pass (round 4; it was three: GELU' multiply, weight-gradient partials,
// input gradient -- 1.4 GB of traffic per block, now 0.8): one workgroup = one sample x 64 channels; the forward input `in` comes into LDS by DMA, the
// image of dhc = dg * gp (gp = GELU'(pre-activation) from the forward) is formed on the way in.  A thread owns a channel quad and one image row and
// slides two 3 x 3 register windows along x: `in`'s for its 9 weight-gradient taps + bias, dhc's for the input gradient
// din[y][x] = sum_taps w[c][ky][kx] dhc[y - ky + 1][x - kx + 1] (tap-major weights read in reverse).  part[b][10][C]: per sample, taps 0..8 then the bias; the 16 rows are added in a fixed order through LDS.
__global__ __launch_bounds__(256) void dwconv_bwd_img_kernel(const bf16* __restrict__ dg, const bf16* __restrict__ gp, const bf16* __restrict__ in,
                                                             const float* __restrict__ w, bf16* __restrict__ din, float* __restrict__ part, int G, int C) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    typedef const __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    const int N = G * G, nchunk = C >> 6;
    const int b = blockIdx.x / nchunk, cc = blockIdx.x - b * nchunk;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* Iin = wsm;                               // [N][64 ch] bf16
    char* Idy = wsm + (size_t)N * 128;
    const int pieces = (N + 7) >> 3;               // 8 tokens x 128 B per DMA instruction
    for (int pc = wid; pc < pieces; pc += 4) {
        int t = pc * 8 + (lane >> 3);
        t = t < N ? t : N - 1;
        const bf16* sp = in + ((size_t)b * N + t) * C + cc * 64 + (lane & 7) * 8;
        __builtin_amdgcn_global_load_lds((gp_t)sp, (lp_t)(Iin + pc * 1024), 16, 0, 0);
    }
    for (int base = threadIdx.x; base < N * 8; base += 4 * 256) {           // four 16-byte pieces of each operand in flight per thread
        bf16x8 a[4], m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256;
            const int t = idx < N * 8 ? idx >> 3 : N - 1, q = idx & 7;
            const size_t o = ((size_t)b * N + t) * C + cc * 64 + q * 8;
            a[u] = *reinterpret_cast<const bf16x8*>(dg + o); m[u] = *reinterpret_cast<const bf16x8*>(gp + o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256;
            if (idx >= N * 8) continue;
            bf16x8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = (bf16)((float)a[u][e] * (float)m[u][e]);
            *reinterpret_cast<bf16x8*>(Idy + (idx >> 3) * 128 + (idx & 7) * 16) = r;
        }
    }
    const int cq = threadIdx.x & 15, y = threadIdx.x >> 4;      // 16 channel quads x up to 16 image rows
    const int c0 = cc * 64 + cq * 4;
    f32x2 wt[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(w + (size_t)(8 - k) * C + c0);      // flipped taps = reversed order
        wt[k][0] = f32x2{t.x, t.y}; wt[k][1] = f32x2{t.z, t.w};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x2 acc[11][2];                              // 9 taps, the depthwise bias, and the column sum of din (= the up-projection's bias gradient)
#pragma unroll
    for (int k = 0; k < 11; ++k) { acc[k][0] = f32x2{0.f, 0.f}; acc[k][1] = f32x2{0.f, 0.f}; }
    if (y < G) {
        auto ld = [&](const char* img, int yy, int xx, f32x2 (&v)[2]) {
            if ((unsigned)yy < (unsigned)G && (unsigned)xx < (unsigned)G) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(img + (size_t)(yy * G + xx) * 128 + cq * 8);
                v[0] = f32x2{(float)t[0], (float)t[1]}; v[1] = f32x2{(float)t[2], (float)t[3]};
            } else { v[0] = f32x2{0.f, 0.f}; v[1] = f32x2{0.f, 0.f}; }
        };
        f32x2 w0[3][2], w1[3][2], w2[3][2];        // `in`: columns x - 1, x, x + 1 of rows y - 1 .. y + 1
        f32x2 d0[3][2], d1[3][2], d2[3][2];        // dhc: the same window
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            w0[r][0] = f32x2{0.f, 0.f}; w0[r][1] = f32x2{0.f, 0.f}; ld(Iin, y + r - 1, 0, w1[r]);
            d0[r][0] = f32x2{0.f, 0.f}; d0[r][1] = f32x2{0.f, 0.f}; ld(Idy, y + r - 1, 0, d1[r]);
        }
        bf16* orow = din + ((size_t)b * N + (size_t)y * G) * C + c0;
        for (int x = 0; x < G; ++x) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { ld(Iin, y + r - 1, x + 1, w2[r]); ld(Idy, y + r - 1, x + 1, d2[r]); }
            f32x2 o2[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const f32x2 gv = d1[1][h2];                    // dhc[y][x]
                acc[9][h2] += gv;
                f32x2 o = f32x2{0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    acc[r * 3 + 0][h2] = __builtin_elementwise_fma(gv, w0[r][h2], acc[r * 3 + 0][h2]);
                    acc[r * 3 + 1][h2] = __builtin_elementwise_fma(gv, w1[r][h2], acc[r * 3 + 1][h2]);
                    acc[r * 3 + 2][h2] = __builtin_elementwise_fma(gv, w2[r][h2], acc[r * 3 + 2][h2]);
                    o = __builtin_elementwise_fma(wt[r * 3 + 0][h2], d0[r][h2], o);
                    o = __builtin_elementwise_fma(wt[r * 3 + 1][h2], d1[r][h2], o);
                    o = __builtin_elementwise_fma(wt[r * 3 + 2][h2], d2[r][h2], o);
                }
                o2[h2] = o;
                acc[10][h2] += o;
            }
            bf16x4 ob;
            ob[0] = (bf16)o2[0][0]; ob[1] = (bf16)o2[0][1]; ob[2] = (bf16)o2[1][0]; ob[3] = (bf16)o2[1][1];
            *reinterpret_cast<bf16x4*>(orow + (size_t)x * C) = ob;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) { w0[r][h2] = w1[r][h2]; w1[r][h2] = w2[r][h2]; d0[r][h2] = d1[r][h2]; d1[r][h2] = d2[r][h2]; }
        }
    }
    __syncthreads();                               // images consumed: their LDS becomes the [16 rows][11][64 ch] reduction buffer (44 KB <= 2 N 128 B for N >= 176)
    float* red = reinterpret_cast<float*>(wsm);
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        float* dst = red + ((size_t)y * 11 + k) * 64 + cq * 4;
        *reinterpret_cast<f32x4*>(dst) = f32x4{acc[k][0][0], acc[k][0][1], acc[k][1][0], acc[k][1][1]};
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 704; i += 256) {                 // (slot k, channel c) of this chunk; rows added in order
        float t = 0.f;
        for (int r = 0; r < 16; ++r) t += red[(size_t)r * 704 + i];
        const int k = i >> 6, c = i & 63;
        part[((size_t)b * 11 + k) * C + cc * 64 + c] = t;
    }
}
// dw [C, 9] / db [C] (/ db2 [C]: slot 10, when nslot == 11) from part[nparts][nslot][C]  (64 (slot, channel) outputs x 16 part-lanes per workgroup, fixed order)
__global__ __launch_bounds__(1024) void dwconv_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int nparts, int C,
                                                            int nslot = 10, float* __restrict__ db2 = nullptr) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;                 // over nslot * C, channel fastest
    float a = 0.f;
    if (i < nslot * C) for (int p = ty; p < nparts; p += 16) a += part[(size_t)p * nslot * C + i];
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && i < nslot * C) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[l][tx];
        const int c = i % C, k = i / C;
        if (k < 9) dw[c * 9 + k] = t; else if (k == 9) db[c] = t; else db2[c] = t;
    }
}

// ---- output projection + unpatchify + loss (tld/denoiser.py:47-52,72,82; tld/train.py:167) -----------------------------------------------
// forward, one wave per token: o[f] = b[f] + x . W[f];  pred[b, c, gy p + p1, gx p + p2] = o[c p p + p1 p + p2];  sq-error partial per row
__global__ __launch_bounds__(256) void tail_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                       const float* __restrict__ target, float* __restrict__ pred, float* __restrict__ dout /*[M, pd]*/,
                                                       float* __restrict__ row_loss, int B, int C, int S, int patch, int grid, int pd, int d, float inv_numel) {
    const int N = grid * grid, M = B * N;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int b = row / N, t = row - b * N, gy = t / grid, gx = t - gy * grid;
    const int J = d >> 6, pp = patch * patch;
    float xv[kMaxJ];
#pragma unroll
    for (int j = 0; j < kMaxJ; ++j) if (j < J) xv[j] = (float)x[(size_t)row * d + lane + 64 * j];
    float sq = 0.f;
    for (int f = 0; f < pd; ++f) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxJ; ++j) if (j < J) a = fmaf(xv[j], W[(size_t)f * d + lane + 64 * j], a);
        a = wave_sum(a) + bias[f];
        const int c = f / pp, r = f - c * pp, p1 = r / patch, p2 = r - p1 * patch;
        const size_t pi = (((size_t)b * C + c) * S + gy * patch + p1) * S + gx * patch + p2;
        const float diff = a - target[pi];
        if (lane == 0) {
            pred[pi] = a;
            dout[(size_t)row * pd + f] = 2.0f * diff * inv_numel;             // d mean((pred - target)^2) / d pred
        }
        sq = fmaf(diff, diff, sq);
    }
    if (lane == 0) row_loss[row] = sq;
}
// loss = inv_numel * sum_rows row_loss  (one workgroup, fixed order)
__global__ void loss_reduce_kernel(const float* __restrict__ row_loss, int M, float inv_numel, float* __restrict__ loss) {
    __shared__ float red[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < M; i += 256) a += row_loss[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) *loss = red[0] * inv_numel;
}
// backward into the stream: gx[row, c] = sum_f dout[row, f] W[f, c]   (fp32, written)
__global__ void tail_dx_kernel(const float* __restrict__ dout, const float* __restrict__ W, float* __restrict__ gx, bf16* __restrict__ gx_bf16, int M, int pd, int d) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * d) return;
    const int c = (int)(i % d);
    const size_t row = i / d;
    float a = 0.f;
    for (int f = 0; f < pd; ++f) a = fmaf(dout[row * pd + f], W[(size_t)f * d + c], a);
    gx[i] = a;
    gx_bf16[i] = (bf16)a;
}
// four consecutive columns per thread (d % 4 == 0): 16-byte weight loads and stores, the row's dout values as wave-uniform 16-byte loads
template <int PD>
__global__ __launch_bounds__(256) void tail_dx4_kernel(const float* __restrict__ dout, const float* __restrict__ W, float* __restrict__ gx, bf16* __restrict__ gx_bf16,
                                                       int M, int d) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = d >> 2;
    if (i >= (size_t)M * q) return;
    const int c = (int)(i % q) * 4;
    const size_t row = i / q;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f4 = 0; f4 < PD / 4; ++f4) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dout + row * PD + 4 * f4);
#pragma unroll
        for (int e = 0; e < 4; ++e) a = __builtin_elementwise_fma(f32x4{dv[e], dv[e], dv[e], dv[e]}, ldf4(W + (size_t)(4 * f4 + e) * d + c), a);
    }
    *reinterpret_cast<f32x4*>(gx + row * d + c) = a;
    bf16x4 o;
    o[0] = (bf16)a[0]; o[1] = (bf16)a[1]; o[2] = (bf16)a[2]; o[3] = (bf16)a[3];
    *reinterpret_cast<bf16x4*>(gx_bf16 + row * d + c) = o;
}
// generic "tall" weight gradient with a SMALL output (out-projection, patch Linear, patch conv):
// part[chunk][n][k] = sum_{rows of chunk} dy[r, n] x[r, k];  n < Nn <= 64, thread per (n, k);  dy fp32, x bf16 or fp32
template <typename TX>
__global__ void tall_dw_partial(const float* __restrict__ dy, int Nn, const TX* __restrict__ x, int K, int M, int rows_per_chunk, float* __restrict__ part) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nn * K) return;
    const int n = i / K, k = i - n * K;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a = fmaf(dy[(size_t)r * Nn + n], ldf(x + (size_t)r * K + k), a);
    part[(size_t)blockIdx.y * Nn * K + i] = a;
}
// The same partials with a thread per x-column computing ALL NN outputs of its column: x is read once instead of NN times through L2 (the form
// above took 232 / 140 us for the out-projection / patch-embedding weights at the training shape), dy rows are wave-uniform 16-byte loads.
template <typename TX, int NN>
__global__ __launch_bounds__(256) void tall_dw_cols_partial(const float* __restrict__ dy, const TX* __restrict__ x, int K, int M, int rows_per_chunk,
                                                            float* __restrict__ part) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float a[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) a[n] = 0.f;
#pragma unroll 4
    for (int r = r0; r < r1; ++r) {
        const float xv = ldf(x + (size_t)r * K + k);
        const float4* dr = reinterpret_cast<const float4*>(dy + (size_t)r * NN);
#pragma unroll
        for (int n4 = 0; n4 < NN / 4; ++n4) {
            const float4 d4 = dr[n4];
            a[4 * n4 + 0] = fmaf(d4.x, xv, a[4 * n4 + 0]); a[4 * n4 + 1] = fmaf(d4.y, xv, a[4 * n4 + 1]);
            a[4 * n4 + 2] = fmaf(d4.z, xv, a[4 * n4 + 2]); a[4 * n4 + 3] = fmaf(d4.w, xv, a[4 * n4 + 3]);
        }
    }
#pragma unroll
    for (int n = 0; n < NN; ++n) part[(size_t)blockIdx.y * NN * K + (size_t)n * K + k] = a[n];
}

// ---- sinusoidal embedding (tld/transformer_blocks.py:7-21) -------------------------------------------------------------------------
__global__ void sinusoid_kernel(const float* __restrict__ sigma, const float* __restrict__ angular, float* __restrict__ out, int B, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, k = i - b * half;
    const float a = sigma[b] * angular[k];
    out[(size_t)b * 2 * half + k] = sinf(a);
    out[(size_t)b * 2 * half + half + k] = cosf(a);
}

// ---- optimizer: Adam (torch.optim.Adam defaults: no weight decay, no amsgrad) + EMA (tld/train.py:55-58,169) --------------------------
// grad_scale multiplies the gradient first (1 / world_size after a sum all-reduce).  bias corrections passed as 1 - beta^t.
__global__ void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, float* __restrict__ ema,
                                size_t n, float lr, float b1, float b2, float eps, float bc1, float bc2, float alpha, float grad_scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;            // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    const float pi = p[i] - (lr / bc1) * (mi / denom);
    p[i] = pi;
    if (ema) ema[i] = ema[i] * alpha + pi * (1.0f - alpha);   // ema.mul_(alpha).add_(param, alpha = 1 - alpha)
}

}  // namespace train
}  // namespace tld
