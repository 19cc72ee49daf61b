// tld_rows.hip -- the HBM-bound row kernels of the denoiser (wavefront reductions, coalesced rows).
//
//   embed_kernel        patchify conv + LN(pd) + Linear(pd->d) + LN(d) + pos-embed   tld/denoiser.py:34-45,75-77
//   layernorm_bf16      LayerNorm rows -> bf16 GEMM operand                           tld/transformer_blocks.py:131,136
//   cross_row_kernel    SA residual add + whole cross-attention sub-block + LN3       tld/transformer_blocks.py:136-138, 62-72
//   tail_kernel         out_proj Linear(d->pd) + unpatchify                           tld/denoiser.py:47-52,72,82
//   update_kernel       CFG combine + DPM-Solver++(2M)/DDIM update + latent shifts    tld/diffusion.py:66-89,122-125
//   dwconv_gelu_kernel  depthwise 3x3 + bias + exact GELU, channels-last              tld/transformer_blocks.py:96-103
//
// Row layout: a wave owns one token row of d features; lane l holds features {2l, 2l+1} + 128*j
// (float2 per access, 512 B per wave-instruction).  d = 128 NJ, or 128 NJ - 64 (template flag HALF: embed_dim is any
// multiple of the head width 64, transformer_blocks.py:126-128): the last group then holds 64 features in lanes 0-31
// and lanes 32-63 carry zeros through every sum and skip their loads / stores (`live`).
#include "tld_common.h"
#include <cstdlib>

namespace tld {

namespace {


// ------------------------------------------------------------------------------------------------
// Workgroup = 32 token rows (8 per wave); the Linear(pd -> d) weight table [pd][d] fp32 (49 KB at d = 768)
// is staged in LDS once per workgroup -- read per row from L1/L2 it made the kernel L1-bandwidth bound.
template <int NJ, bool HALF>
__global__ __launch_bounds__(256) void embed_kernel(EmbedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int d = NJ * 128 - (HALF ? 64 : 0);
    float* wt = reinterpret_cast<float*>(smem);                  // [pd][d]
    float* cwt = wt + p.pd * d;                                  // [C p p][pd]  conv weight, transposed (lane = output)
    for (int i = threadIdx.x; i < p.pd * d / 4; i += 256)
        reinterpret_cast<float4*>(wt)[i] = reinterpret_cast<const float4*>(p.lin_wt)[i];
    for (int i = threadIdx.x; i < p.pd * p.C * p.p * p.p; i += 256) {
        const int o = i / (p.C * p.p * p.p), k = i - o * (p.C * p.p * p.p);
        cwt[k * p.pd + o] = p.conv_w[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool live_last = !HALF || lane < 32;                   // does this lane hold features of the last 128-group?
    auto live = [&](int j) { return j + 1 < NJ || live_last; };
    const int pp = p.p * p.p, cpp = p.C * pp;
    const int total = p.batch * p.ntok;
    // per-lane constants
    const float cb = lane < p.pd ? p.conv_b[lane] : 0.f;
    const float* cw = cwt + (lane < p.pd ? lane : 0);            // + i * pd
    const float g1 = lane < p.pd ? p.ln1_w[lane] : 0.f, b1 = lane < p.pd ? p.ln1_b[lane] : 0.f;
    const int ic = lane / pp, iuv = lane - ic * pp, iu = iuv / p.p, iv = iuv - iu * p.p;   // lane = (c,u,v) for loads
    // 8 rows per wave, two workgroups per CU at the bench size: with 16 rows per wave and one workgroup per CU the
    // kernel was bound by the latency of each row's input load (90 us for 16 K rows); the next row's load is issued early
    constexpr int ROWS_PER_WAVE = 8;
    auto load_in = [&](int row) {
        float xv = 0.f;
        if (row < total && lane < cpp) {
            const int b = row / p.ntok, t = row - b * p.ntok;
            const int ti = t / p.grid, tj = t - ti * p.grid;
            xv = p.x[(((size_t)(b % p.src_batch) * p.C + ic) * p.S + (ti * p.p + iu)) * p.S + (tj * p.p + iv)];
        }
        return xv;
    };
    const int row_first = (blockIdx.x * 4 + wid) * ROWS_PER_WAVE;
    float xnext = load_in(row_first);
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
        const int row = row_first + rr;
        if (row >= total) return;
        const int t = row % p.ntok;
        const float xin = xnext;
        if (rr + 1 < ROWS_PER_WAVE) xnext = load_in(row + 1);
        // (both small loops are unrolled by hand -- `#pragma unroll` refuses loops around v_readlane -- so that the
        // LDS reads of four taps are in flight together; they were one exposed LDS latency per tap)
        float pv = cb;
        {
            int i = 0;
            for (; i + 4 <= cpp; i += 4) {
                const float c0 = cw[(i + 0) * p.pd], c1 = cw[(i + 1) * p.pd], c2 = cw[(i + 2) * p.pd], c3 = cw[(i + 3) * p.pd];
                pv = fmaf(c0, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xin), i + 0)), pv);
                pv = fmaf(c1, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xin), i + 1)), pv);
                pv = fmaf(c2, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xin), i + 2)), pv);
                pv = fmaf(c3, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xin), i + 3)), pv);
            }
            for (; i < cpp; ++i)
                pv = fmaf(cw[i * p.pd], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xin), i)), pv);
        }
        if (lane >= p.pd) pv = 0.f;
        const float inv_pd = 1.0f / (float)p.pd;
        const float mean1 = wave_sum(pv) * inv_pd;
        const float dv = lane < p.pd ? pv - mean1 : 0.f;
        const float rstd1 = 1.0f / sqrtf(wave_sum(dv * dv) * inv_pd + kLnEps);
        const float pn = dv * rstd1 * g1 + b1;                   // 0 for lanes >= pd

        float2 e[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) e[j] = live(j) ? *reinterpret_cast<const float2*>(p.lin_b + j * 128 + 2 * lane) : make_float2(0.f, 0.f);
        auto lin_tap = [&](int o, const float2 (&w)[NJ]) {
            const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pn), o));
#pragma unroll
            for (int j = 0; j < NJ; ++j) { e[j].x = fmaf(a, w[j].x, e[j].x); e[j].y = fmaf(a, w[j].y, e[j].y); }
        };
        auto lin_load = [&](int o, float2 (&w)[NJ]) {
            const float* wrow = wt + o * d + 2 * lane;
#pragma unroll
            for (int j = 0; j < NJ; ++j) w[j] = live(j) ? *reinterpret_cast<const float2*>(wrow + j * 128) : make_float2(0.f, 0.f);
        };
        {
            int o = 0;
            for (; o + 2 <= p.pd; o += 2) {
                float2 w0[NJ], w1[NJ];
                lin_load(o, w0); lin_load(o + 1, w1);
                lin_tap(o, w0); lin_tap(o + 1, w1);
            }
            if (o < p.pd) { float2 w0[NJ]; lin_load(o, w0); lin_tap(o, w0); }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) s += e[j].x + e[j].y;
        const float mean2 = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (live(j)) { e[j].x -= mean2; e[j].y -= mean2; }
            q += e[j].x * e[j].x + e[j].y * e[j].y;
        }
        const float rstd2 = 1.0f / sqrtf(wave_sum(q) / (float)d + kLnEps);
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (!live(j)) continue;
            const int n = j * 128 + 2 * lane;
            const float2 g = *reinterpret_cast<const float2*>(p.ln2_w + n);
            const float2 bb = *reinterpret_cast<const float2*>(p.ln2_b + n);
            const float2 pe = *reinterpret_cast<const float2*>(p.pos + (size_t)t * d + n);
            float2 o;
            o.x = e[j].x * rstd2 * g.x + bb.x + pe.x;
            o.y = e[j].y * rstd2 * g.y + bb.y + pe.y;
            rs_store2(p.tok + (size_t)row * d + n, o);
            const float r0 = rs_round(o.x), r1 = rs_round(o.y);
            ssum += r0 + r1;
            ssq = fmaf(r0, r0, fmaf(r1, r1, ssq));
        }
        if (p.stats_out) {       // LayerNorm-1 statistics of block 0, consumed by its QKV GEMM epilogue (slots come in pairs)
            ssum = wave_sum(ssum); ssq = wave_sum(ssq);
            if (lane == 0) *reinterpret_cast<float4*>(p.stats_out + (size_t)row * kLnSlots) = make_float4(ssum, ssq, 0.f, 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <int NJ, bool HALF>
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(const resid_t* __restrict__ x,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ b,
                                                             bf16* __restrict__ out, int M, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bool live_last = !HALF || lane < 32;
    auto live = [&](int j) { return j + 1 < NJ || live_last; };
    float2 v[NJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        v[j] = live(j) ? rs_load2(x + (size_t)row * d + j * 128 + 2 * lane) : make_float2(0.f, 0.f);
        s += v[j].x + v[j].y;
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (live(j)) { v[j].x -= mean; v[j].y -= mean; }
        q += v[j].x * v[j].x + v[j].y * v[j].y;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + kLnEps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (!live(j)) continue;
        const int n = j * 128 + 2 * lane;
        const float2 gg = *reinterpret_cast<const float2*>(g + n);
        const float2 bb = *reinterpret_cast<const float2*>(b + n);
        bf16x2 o;
        o[0] = (bf16)(v[j].x * rstd * gg.x + bb.x);
        o[1] = (bf16)(v[j].y * rstd * gg.y + bb.y);
        *reinterpret_cast<bf16x2*>(out + (size_t)row * d + n) = o;
    }
}

// LayerNorm arithmetic of the 4-features-per-lane kernels, shared by the bf16 writer and the MX-fp8 writer: the two must produce the
// SAME bf16 values bit for bit (tests/test_gpu_fp8.py: quantising producers == separate passes).  Under HIP's -ffp-contract=fast the
// backend decides per context which multiply-adds fuse, so identical source in two kernels is not identical arithmetic (seen: one token
// row in ~5000 differing between the two at a LayerNorm-1); contraction is pinned here and the fused operations are spelled out.
template <int NQ>
__device__ __forceinline__ void ln_q4_stats(float4 (&v)[NQ], int d, float& rstd) {
#pragma clang fp contract(off)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
        q += __builtin_fmaf(v[j].x, v[j].x, v[j].y * v[j].y) + __builtin_fmaf(v[j].z, v[j].z, v[j].w * v[j].w);
    }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + kLnEps);
}
__device__ __forceinline__ float ln_q4_affine(float v, float rstd, float g, float b) {
#pragma clang fp contract(off)
    return __builtin_fmaf(v * rstd, g, b);
}

// d % 256 == 0 form: lane l holds features {4l .. 4l+3} + 256 j, i.e. 8-byte loads of the bf16 residual and 8-byte
// stores (the 2-feature layout above moves 4 bytes per lane and instruction)
template <int NQ>
__global__ __launch_bounds__(256) void layernorm_bf16_q4_kernel(const resid_t* __restrict__ x,
                                                                const float* __restrict__ g,
                                                                const float* __restrict__ b,
                                                                bf16* __restrict__ out, int M, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float4 v[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) v[j] = rs_load4(x + (size_t)row * d + j * 256 + 4 * lane);
    float rstd;
    ln_q4_stats<NQ>(v, d, rstd);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int n = j * 256 + 4 * lane;
        const float4 gg = *reinterpret_cast<const float4*>(g + n);
        const float4 bb = *reinterpret_cast<const float4*>(b + n);
        bf16x4 o;
        o[0] = (bf16)ln_q4_affine(v[j].x, rstd, gg.x, bb.x);
        o[1] = (bf16)ln_q4_affine(v[j].y, rstd, gg.y, bb.y);
        o[2] = (bf16)ln_q4_affine(v[j].z, rstd, gg.z, bb.z);
        o[3] = (bf16)ln_q4_affine(v[j].w, rstd, gg.w, bb.w);
        *reinterpret_cast<bf16x4*>(out + (size_t)row * d + n) = o;
    }
}

// fp8 GEMM mode: the same LayerNorm rows written straight in the MX-fp8 operand format (no bf16 copy, no separate
// quantisation pass).  A 32-feature block is 8 adjacent lanes of one 256-feature chunk.
template <int NQ>
__global__ __launch_bounds__(256) void layernorm_mx8_kernel(const resid_t* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, uint8_t* __restrict__ out8,
                                                            uint8_t* __restrict__ scale8, int M, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float4 v[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) v[j] = rs_load4(x + (size_t)row * d + j * 256 + 4 * lane);
    float rstd;
    ln_q4_stats<NQ>(v, d, rstd);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int n = j * 256 + 4 * lane;
        const float4 gg = *reinterpret_cast<const float4*>(g + n);
        const float4 bb = *reinterpret_cast<const float4*>(b + n);
        // (rounded through bf16 first: the operand the bf16 path would have produced is what gets quantised)
        const float y0 = (float)(bf16)ln_q4_affine(v[j].x, rstd, gg.x, bb.x), y1 = (float)(bf16)ln_q4_affine(v[j].y, rstd, gg.y, bb.y);
        const float y2 = (float)(bf16)ln_q4_affine(v[j].z, rstd, gg.z, bb.z), y3 = (float)(bf16)ln_q4_affine(v[j].w, rstd, gg.w, bb.w);
        int e8;
        const unsigned w = mx8_pack4(y0, y1, y2, y3, &e8);
        *reinterpret_cast<unsigned*>(out8 + (size_t)row * d + n) = w;
        if ((lane & 7) == 0) scale8[mx8_scale_index(n, (size_t)row, (size_t)M)] = (uint8_t)e8;
    }
}

// ------------------------------------------------------------------------------------------------
// Cross-attention has only two keys per sample (the noise token and the label token), so
//   softmax([q.k_n, q.k_l] / 8) = [1 - s, s],  s = sigmoid((q.k_l - q.k_n) / 8),
// and q.k_t / 8 = LN2(x) . (Wq_h^T k_t[h] / 8): the query projection folds into one d-vector per
// (token row, head), prepared once on the conditioning path (wq table, LN2 gamma folded in, LN2 beta
// contribution in bwq).  The sub-block therefore needs no GEMM: per row it is 12 dot products of
// the centred row against LDS-resident vectors, a sigmoid per head, and a blend of the two value rows.
template <int NJ, bool HALF>
__global__ __launch_bounds__(256, 3) void cross_row_kernel(CrossRowParams p, int chunks_per_sample) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int d = NJ * 128 - (HALF ? 64 : 0), H = d / 64;
    constexpr float inv_d = 1.0f / (float)d;
    float* wd = reinterpret_cast<float*>(smem);          // [H][d]  wq_label - wq_noise (gamma folded)
    float* vn = wd + H * d;                              // [d]     value row of the noise token
    float* vdiff = vn + d;                               // [d]     v_label - v_noise
    float* bw = vdiff + d;                               // [H]     beta contribution to the logit diff

    // A workgroup owns one contiguous chunk of ROW PAIRS of one sample (balanced split, so the grid can be sized
    // to exactly one resident round: 3 workgroups per CU); its four waves take the pairs round-robin.
    const int b = blockIdx.x / chunks_per_sample;
    const int ck = blockIdx.x - b * chunks_per_sample;
    const int pairs = p.ntok >> 1;
    const int pp0 = (int)((long)ck * pairs / chunks_per_sample);
    const int pp1 = (int)((long)(ck + 1) * pairs / chunks_per_sample);
    const int tn = p.noise_row[b], tl = p.label_row[b];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool upper = lane >= 32;
    const bool live_last = !HALF || !upper;
    auto live = [&](int j) { return j + 1 < NJ || live_last; };
    // input rows: normally the same rows; with CFG layer-0 sharing the cond and uncond samples read the one copy
    const resid_t* xin = p.x_in ? p.x_in : p.x;
    const size_t obase = (size_t)b * p.ntok;
    const size_t ibase = p.x_in ? (size_t)(b % p.src_batch) * p.ntok : obase;

    // Two rows per wave at a time (independent reduction chains fill the DPP wait states), with the next
    // pair's HBM loads issued before the current pair is processed (the first pair's before the table fill).
    float2 xr[2][NJ];
    bf16x2 ar[2][NJ];
    auto fetch = [&](size_t row, float2 (&xv)[NJ], bf16x2 (&av)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = j * 128 + 2 * lane;
            if (live(j)) {
                xv[j] = rs_load2(xin + row * d + n);
                av[j] = *reinterpret_cast<const bf16x2*>(p.att + row * d + n);
            } else {
                xv[j] = make_float2(0.f, 0.f);
                av[j] = bf16x2{(bf16)0.f, (bf16)0.f};
            }
        }
    };
    int pr = pp0 + wid;
    if (pr < pp1) {
        fetch(ibase + 2 * (size_t)pr, xr[0], ar[0]);
        fetch(ibase + 2 * (size_t)pr + 1, xr[1], ar[1]);
    }

    {
        const float4* wl = reinterpret_cast<const float4*>(p.wq + (size_t)tl * H * d);
        const float4* wn = reinterpret_cast<const float4*>(p.wq + (size_t)tn * H * d);
        float4* dst = reinterpret_cast<float4*>(wd);
        for (int i = threadIdx.x; i < H * d / 4; i += 256) {
            const float4 a = wl[i], c = wn[i];
            dst[i] = make_float4(a.x - c.x, a.y - c.y, a.z - c.z, a.w - c.w);
        }
    }
    for (int i = threadIdx.x; i < d; i += 256) {
        const float a = p.v[(size_t)tn * p.v_ld + i];
        vn[i] = a;
        vdiff[i] = p.v[(size_t)tl * p.v_ld + i] - a;
    }
    if (threadIdx.x < H) bw[threadIdx.x] = p.bwq[(size_t)tl * H + threadIdx.x] - p.bwq[(size_t)tn * H + threadIdx.x];
    __syncthreads();
    // lane's features of group j belong to head 2j + (lane >> 5)
    float bwl[NJ];
#pragma unroll
    for (int hh = 0; hh < NJ; ++hh) bwl[hh] = live(hh) ? bw[2 * hh + (upper ? 1 : 0)] : 0.f;

    for (; pr < pp1; pr += 4) {
        f32x2 v[2][NJ];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                v[u][j][0] = xr[u][j].x + (float)ar[u][j][0];        // x = SA(LN1 x) + x
                v[u][j][1] = xr[u][j].y + (float)ar[u][j][1];
            }
        const size_t row = obase + 2 * (size_t)pr;
        if (pr + 4 < pp1) {
            fetch(ibase + 2 * (size_t)(pr + 4), xr[0], ar[0]);
            fetch(ibase + 2 * (size_t)(pr + 4) + 1, xr[1], ar[1]);
        }
        if (p.sa_out) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (live(j)) *reinterpret_cast<f32x2*>(p.sa_out + (row + u) * d + j * 128 + 2 * lane) = v[u][j];
        }
        float mean[2], rstd[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x2 s2 = v[u][0];
#pragma unroll
            for (int j = 1; j < NJ; ++j) s2 += v[u][j];
            mean[u] = wave_sum(s2[0] + s2[1]) * inv_d;
        }
        f32x2 c[2][NJ];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x2 q2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                c[u][j] = live(j) ? v[u][j] - mean[u] : f32x2{0.f, 0.f};
                q2 = __builtin_elementwise_fma(c[u][j], c[u][j], q2);
            }
            rstd[u] = __builtin_amdgcn_rsqf(fmaf(wave_sum(q2[0] + q2[1]), inv_d, kLnEps));
        }

        // per-head logit difference -> sigmoid weight of the label token.  Packed-fp32 FMAs over the lane's
        // feature pairs; the cross-lane sums of a head PAIR share one v_permlane32_swap (lanes 0-31 finish
        // head 2hh, lanes 32-63 head 2hh+1 -- exactly the head whose features each half holds).
        float plab[2][NJ];
#pragma unroll
        for (int hh = 0; hh < NJ; ++hh) {
            f32x2 acc[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
            const float* w0 = wd + (2 * hh) * d + 2 * lane;
            const float* w1 = w0 + d;
            const bool h1 = !HALF || hh + 1 < NJ;              // head 2 hh + 1 exists
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x2 a = live(j) ? *reinterpret_cast<const f32x2*>(w0 + j * 128) : f32x2{0.f, 0.f};
                const f32x2 e = (h1 && live(j)) ? *reinterpret_cast<const f32x2*>(w1 + j * 128) : f32x2{0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[u][0] = __builtin_elementwise_fma(c[u][j], a, acc[u][0]);
                    acc[u][1] = __builtin_elementwise_fma(c[u][j], e, acc[u][1]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float p0 = acc[u][0][0] + acc[u][0][1], p1 = acc[u][1][0] + acc[u][1][1];
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
                const float dl = half_sum(__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * rstd[u] + bwl[hh];
                plab[u][hh] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-dl * 1.44269504088896340736f));
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the next head pair's 2 x NJ LDS reads from being hoisted (VGPRs)
        }
        // x += p_noise v_n + p_label v_l ; then LN3
        float mean3[2], rstd3[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x2 s2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (!live(j)) continue;
                const int n = j * 128 + 2 * lane;
                const f32x2 a = *reinterpret_cast<const f32x2*>(vn + n);
                const f32x2 dd = *reinterpret_cast<const f32x2*>(vdiff + n);
                const f32x2 pl = {plab[u][j], plab[u][j]};
                v[u][j] += __builtin_elementwise_fma(pl, dd, a);
                rs_store2(p.x + (row + u) * d + n, make_float2(v[u][j][0], v[u][j][1]));
                s2 += v[u][j];
            }
            mean3[u] = wave_sum(s2[0] + s2[1]) * inv_d;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x2 q2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (!live(j)) continue;
                v[u][j] -= mean3[u];
                q2 = __builtin_elementwise_fma(v[u][j], v[u][j], q2);
            }
            rstd3[u] = __builtin_amdgcn_rsqf(fmaf(wave_sum(q2[0] + q2[1]), inv_d, kLnEps));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (!live(j)) continue;
            const int n = j * 128 + 2 * lane;
            const f32x2 gg = *reinterpret_cast<const f32x2*>(p.ln3_w + n);
            const f32x2 bb = *reinterpret_cast<const f32x2*>(p.ln3_b + n);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x2 r = __builtin_elementwise_fma(v[u][j] * rstd3[u], gg, bb);
                bf16x2 o;
                o[0] = (bf16)r[0];
                o[1] = (bf16)r[1];
                *reinterpret_cast<bf16x2*>(p.xn3 + (row + u) * d + n) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// d % 256 == 0 (the 100 M model): lane l holds features {4l .. 4l+3} + 256 j -- 8-byte loads and stores of the bf16 streams -- and the
// per-head logit dot products run on the MATRIX pipe (round 4; up to 16 heads).
// The 4 H d fp32 MACs per row pair of the round-3 VALU kernel (cross_row_q4_kernel, lane = 4 features + 256 j; retired) were 11 of its 46 us at C1 (profiles/r03_cross_row_attribution.txt) with the
// matrix cores idle.  Here a 512-thread workgroup walks 16-row groups of one sample: every wave brings in one row pair (the next pair's
// loads in flight), forms x1 = x + att and its LayerNorm-2 statistics as before, and parks the CENTRED rows in an LDS tile [16][d] as a
// SPLIT bf16 pair (hi = bf16(c), lo = bf16(c - hi): 16 significant bits; centring first keeps rows with a large common offset exact, g9).
// After one barrier the tile's logits against the H folded query-difference vectors are v_mfma_f32_16x16x32_bf16 products
// hi.hi + lo.hi + hi.lo (fp32 accumulate: the logits keep fp32-grade accuracy, which plain bf16 operands did not: forward rel-rms
// 6.5e-3 instead of 6.3e-3 on g5) with the K range split over the eight waves: wave w multiplies features [d w / 8, d (w + 1) / 8), whose
// slices of the query-difference vectors (split the same way) it keeps in REGISTERS for the whole launch -- no table in LDS, no table fill.
// D[head][row] partials land with a lane = (row, head quad) and cross the waves through 8 KB of LDS (second barrier); every wave then sums
// the eight partials of its own two rows, applies rstd and the beta term, takes the sigmoids and finishes its rows as before.
// LDS images: 16-byte chunk c of row r at chunk c ^ r: conflict-free for the row writes (ds_write_b64) and for the MFMA operand reads (a
// ds_read_b128 lane group mixes two k-quarters of eight rows each; chunk = 4 kb + kq keeps their slots apart).
template <int NQ>
__global__ __launch_bounds__(512) void cross_row_mfma_kernel(CrossRowParams p, int groups_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int d = NQ * 256, H = NQ * 4, PITCH = d * 2;
    constexpr float inv_d = 1.0f / (float)d;
    char* tileH = smem;                                              // [16][PITCH] bf16 hi halves of the group's centred rows
    char* tileL = tileH + 16 * PITCH;                                // [16][PITCH] lo halves
    float* part = reinterpret_cast<float*>(tileL + 16 * PITCH);      // [8 waves][16 rows][16 heads] partial logits
    float* vn = part + 8 * 256;                                      // [d] value row of the noise token
    float* vdiff = vn + d;                                           // [d] v_label - v_noise

    const int wgs_per_sample = (p.ntok >> 4) / groups_per_wg;
    const int b = blockIdx.x / wgs_per_sample;
    const int g0 = (blockIdx.x - b * wgs_per_sample) * groups_per_wg;   // first 16-row group of this workgroup inside the sample
    const int tn = p.noise_row[b], tl = p.label_row[b];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const resid_t* xin = p.x_in ? p.x_in : p.x;
    const size_t obase = (size_t)b * p.ntok;
    const size_t ibase = p.x_in ? (size_t)(b % p.src_batch) * p.ntok : obase;

    auto pair_row = [&](int g) { return 16 * (g0 + g) + 2 * wid; };   // this wave's row pair of group g (row inside the sample)
    // the next row pair's loads are in flight while the current one is processed (a second pair in flight measured 1 us SLOWER: 40.8 vs 39.7; round 4:
    // the pairs of the next TWO groups by global -> LDS DMA into per-wave slots, own-vmcnt waits, 151 registers, bitwise the same result: 42.3 vs 37.4 us --
    // bytes in flight are not what the kernel waits for, the per-group chain of reductions and its two barriers is)
    resid4_t xr[2][NQ];
    bf16x4 ar[2][NQ];
    auto fetch = [&](size_t row, resid4_t (&xv)[NQ], bf16x4 (&av)[NQ]) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int n = j * 256 + 4 * lane;
            xv[j] = rs_raw4(xin + row * d + n);
            av[j] = *reinterpret_cast<const bf16x4*>(p.att + row * d + n);
        }
    };
    fetch(ibase + pair_row(0), xr[0], ar[0]);
    fetch(ibase + pair_row(0) + 1, xr[1], ar[1]);

    const int tok = lane & 15, kq = lane >> 4;                       // MFMA roles: row of an operand / output column, and k-quarter / head quad
    // A operand (rows = heads), this wave's K slice, split hi / lo: K-block kb = wid NQ + s2 covers features 32 kb .. 32 kb + 31, of which this
    // lane holds 8 kq .. 8 kq + 7.  (Operand rows H .. 15 repeat the last head: their outputs are never read.)
    bf16x8 wh[NQ], wlo[NQ];
    {
        const int hrow = tok < H ? tok : H - 1;
        const float* wlp = p.wq + ((size_t)tl * H + hrow) * d;
        const float* wnp = p.wq + ((size_t)tn * H + hrow) * d;
#pragma unroll
        for (int s2 = 0; s2 < NQ; ++s2) {
            const int f = 32 * (wid * NQ + s2) + 8 * kq;
            const float4 a0 = *reinterpret_cast<const float4*>(wlp + f), a1 = *reinterpret_cast<const float4*>(wlp + f + 4);
            const float4 c0 = *reinterpret_cast<const float4*>(wnp + f), c1 = *reinterpret_cast<const float4*>(wnp + f + 4);
            const float df[8] = {a0.x - c0.x, a0.y - c0.y, a0.z - c0.z, a0.w - c0.w, a1.x - c1.x, a1.y - c1.y, a1.z - c1.z, a1.w - c1.w};
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) {
                const bf16 hi = (bf16)df[e2];
                wh[s2][e2] = hi;
                wlo[s2][e2] = (bf16)(df[e2] - (float)hi);
            }
        }
    }
    for (int i = threadIdx.x; i < d; i += 512) {
        const float a = p.v[(size_t)tn * p.v_ld + i];
        vn[i] = a;
        vdiff[i] = p.v[(size_t)tl * p.v_ld + i] - a;
    }
    // lane's features of group j belong to head 4 j + (lane >> 4): beta contribution to that head's logit difference
    float bwl[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) bwl[j] = p.bwq[(size_t)tl * H + 4 * j + (lane >> 4)] - p.bwq[(size_t)tn * H + 4 * j + (lane >> 4)];
    const bool fold3 = p.ln3_stats != nullptr;

    for (int g = 0; g < groups_per_wg; ++g) {
        // ---- phase A: the wave's row pair -> x1, LayerNorm-2 statistics, centred split-bf16 rows into the tile
        f32x4 v[2][NQ];                                              // x1 = x + att, fp32
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const float4 xw = rs_widen4(xr[u][j]);
                v[u][j][0] = xw.x + (float)ar[u][j][0];              // x = SA(LN1 x) + x
                v[u][j][1] = xw.y + (float)ar[u][j][1];
                v[u][j][2] = xw.z + (float)ar[u][j][2];
                v[u][j][3] = xw.w + (float)ar[u][j][3];
            }
        if (g + 1 < groups_per_wg) {
            fetch(ibase + pair_row(g + 1), xr[0], ar[0]);
            fetch(ibase + pair_row(g + 1) + 1, xr[1], ar[1]);
        }
        const size_t row = obase + pair_row(g);
        if (p.sa_out) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NQ; ++j)
                    *reinterpret_cast<f32x4*>(p.sa_out + (row + u) * d + j * 256 + 4 * lane) = v[u][j];
        }
        float rstd[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 s4 = v[u][0];
#pragma unroll
            for (int j = 1; j < NQ; ++j) s4 += v[u][j];
            const float mean = wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * inv_d;
            f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
            const int gr = 2 * wid + u;                              // row inside the group's tile
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const f32x4 c = v[u][j] - mean;
                q4 = __builtin_elementwise_fma(c, c, q4);
                bf16x4 oh, ol;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) { oh[e2] = (bf16)c[e2]; ol[e2] = (bf16)(c[e2] - (float)oh[e2]); }
                const int off = gr * PITCH + ((((32 * j + (lane >> 1)) ^ gr) << 4) | ((lane & 1) << 3));
                *reinterpret_cast<bf16x4*>(tileH + off) = oh;
                *reinterpret_cast<bf16x4*>(tileL + off) = ol;
            }
            rstd[u] = __builtin_amdgcn_rsqf(fmaf(wave_sum((q4[0] + q4[1]) + (q4[2] + q4[3])), inv_d, kLnEps));
        }
        __syncthreads();                                             // the group's 16 rows are in LDS (and nobody still reads last group's partials)
        // ---- phase B: this wave's K slice of the tile's logits on the matrix pipe, D[head 4 kq + r][row tok] partial
        {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < NQ; ++s2) {
                const int off = tok * PITCH + (((4 * (wid * NQ + s2) + kq) ^ tok) << 4);
                const bf16x8 ch = *reinterpret_cast<const bf16x8*>(tileH + off);
                const bf16x8 cl = *reinterpret_cast<const bf16x8*>(tileL + off);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s2], ch, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s2], cl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[s2], ch, acc, 0, 0, 0);
            }
            *reinterpret_cast<f32x4*>(part + (wid * 16 + tok) * 16 + 4 * kq) = acc;
        }
        __syncthreads();                                             // all eight K slices are in; the tile is free for the next group's rows
        // ---- phase C: logits of the wave's own two rows -> sigmoid weights; blend the two value rows, store, LayerNorm-3
        float plab[2][NQ];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const float* pp = part + (2 * wid + u) * 16 + 4 * j + (lane >> 4);
                float t = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < 8; ++w2) t += pp[w2 * 256];
                const float dl = fmaf(t, rstd[u], bwl[j]);
                plab[u][j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-dl * 1.44269504088896340736f));
                // (round 3 kept these one (row, head) at a time to save 40 registers for a 128-register build; at one workgroup per CU the 48 partial reads go together: 38.1 -> 37.6 us)
            }
        float mean3[2], rstd3[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int n = j * 256 + 4 * lane;
                const f32x4 a = *reinterpret_cast<const f32x4*>(vn + n);
                const f32x4 dd = *reinterpret_cast<const f32x4*>(vdiff + n);
                const f32x4 pl = {plab[u][j], plab[u][j], plab[u][j], plab[u][j]};
                v[u][j] += __builtin_elementwise_fma(pl, dd, a);
                rs_store4(p.x + (row + u) * d + n, make_float4(v[u][j][0], v[u][j][1], v[u][j][2], v[u][j][3]));
                if (fold3) {        // LN3 is applied by the consumer GEMM to the STORED (rounded) row: take its statistics
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) v[u][j][e2] = rs_round(v[u][j][e2]);
                }
                s4 += v[u][j];
            }
            mean3[u] = wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * inv_d;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                v[u][j] -= mean3[u];
                q4 = __builtin_elementwise_fma(v[u][j], v[u][j], q4);
            }
            rstd3[u] = __builtin_amdgcn_rsqf(fmaf(wave_sum((q4[0] + q4[1]) + (q4[2] + q4[3])), inv_d, kLnEps));
        }
        if (fold3) {
            if (lane == 0) {
                p.ln3_stats[row] = make_float2(mean3[0], rstd3[0]);
                p.ln3_stats[row + 1] = make_float2(mean3[1], rstd3[1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int n = j * 256 + 4 * lane;
                const f32x4 gg = *reinterpret_cast<const f32x4*>(p.ln3_w + n);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln3_b + n);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x4 r = __builtin_elementwise_fma(v[u][j] * rstd3[u], gg, bb);
                    bf16x4 o;
                    o[0] = (bf16)r[0]; o[1] = (bf16)r[1]; o[2] = (bf16)r[2]; o[3] = (bf16)r[3];
                    if (p.xn3_f8) {      // fp8 GEMM mode: MX-quantise the bf16-rounded row in place of the bf16 store
                        int e8;
                        const unsigned w = mx8_pack4((float)o[0], (float)o[1], (float)o[2], (float)o[3], &e8);
                        *reinterpret_cast<unsigned*>(p.xn3_f8 + (row + u) * d + n) = w;
                        if ((lane & 7) == 0) p.xn3_s8[mx8_scale_index(n, row + u, (size_t)p.batch * p.ntok)] = (uint8_t)e8;
                    } else {
                        *reinterpret_cast<bf16x4*>(p.xn3 + (row + u) * d + n) = o;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out_proj + unpatchify.  16 tokens per wave, weights [pd][d] fp32 resident in LDS.
template <int NJ, bool HALF>
__global__ __launch_bounds__(256) void tail_kernel(TailParams p, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int d = NJ * 128 - (HALF ? 64 : 0);
    float* w = reinterpret_cast<float*>(smem);           // [pd][d]
    for (int i = threadIdx.x; i < p.pd * d / 4; i += 256)
        reinterpret_cast<float4*>(w)[i] = reinterpret_cast<const float4*>(p.w)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool live_last = !HALF || lane < 32;
    auto live = [&](int j) { return j + 1 < NJ || live_last; };
    const int rows_per_wave = rows_per_block / 4;            // even
    const int total = p.batch * p.ntok;                      // even (ntok is)
    const int row_first = blockIdx.x * rows_per_block + wid * rows_per_wave;
    // two rows at a time (eight independent dot-product / reduction chains), the next pair's loads issued first
    float2 nx[2][NJ];
    auto fetch = [&](int row, float2 (&v)[NJ]) {
        const int r = row < total ? row : total - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = live(j) ? rs_load2(p.tok + (size_t)r * d + j * 128 + 2 * lane) : make_float2(0.f, 0.f);
    };
    fetch(row_first, nx[0]); fetch(row_first + 1, nx[1]);
    const float bias = p.b[lane < p.pd ? lane : 0];
    for (int rr = 0; rr < rows_per_wave; rr += 2) {
        const int row = row_first + rr;
        if (row >= total) return;
        float2 v[2][NJ];
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[u2][j] = nx[u2][j];
        if (rr + 2 < rows_per_wave) { fetch(row + 2, nx[0]); fetch(row + 3, nx[1]); }
        float mine[2] = {0.f, 0.f};
        for (int o4 = 0; o4 < p.pd; o4 += 4) {
            float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int o = o4 + u < p.pd ? o4 + u : p.pd - 1;
                const float* wrow = w + o * d + 2 * lane;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float2 ww = live(j) ? *reinterpret_cast<const float2*>(wrow + j * 128) : make_float2(0.f, 0.f);
#pragma unroll
                    for (int u2 = 0; u2 < 2; ++u2)
                        part[u2][u] = fmaf(v[u2][j].x, ww.x, fmaf(v[u2][j].y, ww.y, part[u2][u]));
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int u2 = 0; u2 < 2; ++u2) {
                    const float tot = wave_sum(part[u2][u]);
                    if (lane == o4 + u) mine[u2] = tot + bias;
                }
        }
        if (lane < p.pd) {
            // feature f = (c, u, v) of token (ti, tj) -> out[b, c, ti*p+u, tj*p+v]
            const int c = lane / (p.p * p.p), uv = lane - c * p.p * p.p;
            const int u = uv / p.p, vv = uv - u * p.p;
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {
                const int r = row + u2;
                if (r < total) {
                    const int b = r / p.ntok, t = r - b * p.ntok;
                    const int ti = t / p.grid, tj = t - ti * p.grid;
                    p.out[(((size_t)b * p.C + c) * p.S + (ti * p.p + u)) * p.S + (tj * p.p + vv)] = mine[u2];
                }
            }
        }
    }
}

#ifdef TLD_RESID_BF16
// ------------------------------------------------------------------------------------------------
// Patch embedding on the matrix pipe (round 4; patch_dim == 16 -- four latent channels in 2 x 2 patches, every published model -- and
// d % 256 == 0).  embed_kernel above is one latency chain per row (readlane-broadcast taps, four wave-wide reductions): 52 us per forward
// for 16 K rows.  Here a 4-wave workgroup takes 32 token rows; every lane (token, half) builds its token's conv outputs / LayerNorm(16)
// values 8 half .. 8 half + 7 in registers -- exactly the K slice the B operand of v_mfma_f32_32x32x16_bf16 wants from it -- as a split
// bf16 pair, and Linear(16 -> d) is three MFMAs (hi.hi + hi.lo + lo.hi: fp32-grade) per 32-feature tile against the weight image in LDS;
// wave w owns feature tiles [w d / 128, (w + 1) d / 128), whose accumulators stay in registers through the two-pass LayerNorm(d) (row sums
// cross the waves through LDS).  Output rows leave as whole 128-byte segments through a per-wave transpose patch.  tld/denoiser.py:34-45,75-77.
template <int TPW>      // 32-feature tiles per wave: d = 128 TPW, TPW even
__global__ __launch_bounds__(256) void embed_mfma_kernel(EmbedParams p, const bf16* __restrict__ wl_hl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int d = TPW * 128, PD = 16, PP = 144;
    // (round 6: the Linear weight's split-bf16 rows go from L2 straight into the MFMA operand registers -- lane (l31, hi) of tile t wants the 16 bytes
    // [plane][feature 32 t + l31][8 hi .. 8 hi + 7], and a wave's 64 lanes read one contiguous KiB -- instead of being copied into a 48 KiB LDS image by every
    // 32-row workgroup first: the fill was as many bytes as the workgroup's output)
    float* cw = reinterpret_cast<float*>(smem);                   // [16][16] conv weight (row = output)
    float* red = cw + PD * PD;                                    // [3][4 waves][32 tokens] row partials (sum; centred squares; rounded sum) + [4][32] rounded squares
    float* vec = red + 4 * 4 * 32;                                // [3][d]: Linear bias, LayerNorm(d) weight and bias (read per feature quad by every token)
    char* patch = reinterpret_cast<char*>(vec + 3 * d);           // [4 waves][32 tokens][PP] output transpose
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int total = p.batch * p.ntok;
    const int grow = blockIdx.x * 32 + l31;
    const int row = grow < total ? grow : total - 1;
    const int b = row / p.ntok, tk = row - b * p.ntok;
    const int ti = tk / p.grid, tj = tk - ti * p.grid;
    // the token's 16 patch inputs, k = (c, u, v)
    float xin[PD];
    {
        const int pp = p.p * p.p;
        const float* xb = p.x + (size_t)(b % p.src_batch) * p.C * p.S * p.S;
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            const int c = k / pp, uv = k - c * pp, u = uv / p.p, v = uv - u * p.p;
            xin[k] = xb[((size_t)c * p.S + (ti * p.p + u)) * p.S + (tj * p.p + v)];
        }
    }
    bf16x8 whf[TPW], wlf[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int f = 32 * (wid * TPW + t) + l31;
        whf[t] = *reinterpret_cast<const bf16x8*>(wl_hl + (size_t)f * PD + hi * 8);
        wlf[t] = *reinterpret_cast<const bf16x8*>(wl_hl + ((size_t)d + f) * PD + hi * 8);
    }
    if (threadIdx.x < PD * PD) cw[threadIdx.x] = p.conv_w[threadIdx.x];
    for (int i = threadIdx.x; i < 3 * d / 4; i += 256) {
        const int which = i / (d / 4), j = i - which * (d / 4);
        const float* src = which == 0 ? p.lin_b : (which == 1 ? p.ln2_w : p.ln2_b);
        reinterpret_cast<float4*>(vec)[i] = reinterpret_cast<const float4*>(src)[j];
    }
    __syncthreads();
    // conv outputs o = 8 hi + e, then LayerNorm(16) over the token's 16 outputs (8 here, 8 in lane ^ 32)
    float pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int o = 8 * hi + e;
        const float4* wr = reinterpret_cast<const float4*>(cw + o * PD);
        float a = p.conv_b[o];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const float4 w4 = wr[k4];
            a = fmaf(w4.x, xin[4 * k4], a); a = fmaf(w4.y, xin[4 * k4 + 1], a); a = fmaf(w4.z, xin[4 * k4 + 2], a); a = fmaf(w4.w, xin[4 * k4 + 3], a);
        }
        pv[e] = a;
    }
    auto pair_sum = [&](float v) {          // + the partner lane (same token, other half)
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    };
    float s8 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s8 += pv[e];
    const float mean1 = pair_sum(s8) * (1.0f / PD);
    float q8 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { pv[e] -= mean1; q8 = fmaf(pv[e], pv[e], q8); }
    const float rstd1 = 1.0f / sqrtf(pair_sum(q8) * (1.0f / PD) + kLnEps);
    bf16x8 bh, bl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int o = 8 * hi + e;
        const float pn = pv[e] * rstd1 * p.ln1_w[o] + p.ln1_b[o];
        bh[e] = (bf16)pn;
        bl[e] = (bf16)(pn - (float)bh[e]);
    }
    // Linear(16 -> d): this wave's TPW feature tiles, D[feature][token]
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const bf16x8 wh = whf[t], wlo = wlf[t];
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bh, z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bl, z, 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, bh, z, 0, 0, 0);
    }
    // + bias; LayerNorm(d), two passes over the registers; the row's other features live in the partner lane and in the other three waves
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = *reinterpret_cast<const float4*>(vec + 32 * (wid * TPW + t) + 8 * q + 4 * hi);
            acc[t][4 * q + 0] += b4.x; acc[t][4 * q + 1] += b4.y; acc[t][4 * q + 2] += b4.z; acc[t][4 * q + 3] += b4.w;
            s += (acc[t][4 * q + 0] + acc[t][4 * q + 1]) + (acc[t][4 * q + 2] + acc[t][4 * q + 3]);
        }
    s = pair_sum(s);
    if (!hi) red[wid * 32 + l31] = s;
    __syncthreads();
    const float mean2 = ((red[l31] + red[32 + l31]) + (red[64 + l31] + red[96 + l31])) * (1.0f / d);
    float qq = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] -= mean2; qq = fmaf(acc[t][r], acc[t][r], qq); }
    qq = pair_sum(qq);
    if (!hi) red[128 + wid * 32 + l31] = qq;
    __syncthreads();
    const float rstd2 = 1.0f / sqrtf(((red[128 + l31] + red[160 + l31]) + (red[192 + l31] + red[224 + l31])) * (1.0f / d) + kLnEps);
    // affine + position table -> bf16, two tiles (64 features = 128 B per token) at a time through the wave's transpose patch
    char* T = patch + wid * 32 * PP;
    float ssum = 0.f, ssq = 0.f;
    // (the token's position-table values -- the only per-token global loads left in this loop -- are fetched one tile PAIR ahead: as loads
    // inside the loop they exposed one L2 round trip per feature quad, 24 per lane, and were most of the kernel's first version: 42.6 us)
    const float* prow = p.pos + (size_t)tk * d + 32 * wid * TPW + 4 * hi;
    float4 pnx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pnx[i] = *reinterpret_cast<const float4*>(prow + 32 * (i >> 2) + 8 * (i & 3));
#pragma unroll
    for (int tp = 0; tp < TPW / 2; ++tp) {
        float4 pc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) pc[i] = pnx[i];
        if (tp + 1 < TPW / 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) pnx[i] = *reinterpret_cast<const float4*>(prow + 64 * (tp + 1) + 32 * (i >> 2) + 8 * (i & 3));
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * tp + tt;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 32 * (wid * TPW + t) + 8 * q + 4 * hi;
                const float4 g4 = *reinterpret_cast<const float4*>(vec + d + n);
                const float4 c4 = *reinterpret_cast<const float4*>(vec + 2 * d + n);
                const float4 pe = pc[tt * 4 + q];
                bf16x4 o;
                o[0] = (bf16)(acc[t][4 * q + 0] * rstd2 * g4.x + c4.x + pe.x);
                o[1] = (bf16)(acc[t][4 * q + 1] * rstd2 * g4.y + c4.y + pe.y);
                o[2] = (bf16)(acc[t][4 * q + 2] * rstd2 * g4.z + c4.z + pe.z);
                o[3] = (bf16)(acc[t][4 * q + 3] * rstd2 * g4.w + c4.w + pe.w);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float r = (float)o[e]; ssum += r; ssq = fmaf(r, r, ssq); }
                *reinterpret_cast<bf16x4*>(T + l31 * PP + (tt * 32 + 8 * q + 4 * hi) * 2) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = it * 8 + (lane >> 3), ch = lane & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(T + r * PP + ch * 16);
            const int gr = blockIdx.x * 32 + r;
            if (gr < total) *reinterpret_cast<u32x4*>(p.tok + (size_t)gr * d + 32 * (wid * TPW + 2 * tp) + ch * 8) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (p.stats_out) {      // LayerNorm-1 statistics of block 0 (sum, sum of squares of the ROUNDED row), consumed by its QKV GEMM epilogue
        ssum = pair_sum(ssum); ssq = pair_sum(ssq);
        if (!hi) { red[256 + wid * 32 + l31] = ssum; red[384 + wid * 32 + l31] = ssq; }
        __syncthreads();
        if (wid == 0 && !hi && grow < total) {
            const float a = (red[256 + l31] + red[288 + l31]) + (red[320 + l31] + red[352 + l31]);
            const float a2 = (red[384 + l31] + red[416 + l31]) + (red[448 + l31] + red[480 + l31]);
            *reinterpret_cast<float4*>(p.stats_out + (size_t)grow * kLnSlots) = make_float4(a, a2, 0.f, 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out_proj + unpatchify on the matrix pipe (round 4; bf16 residual stream, pd d <= 40960).  The VALU kernel above spends a wave-wide
// reduction per (row, output feature) -- 16 per row at the 100 M model -- and was a 43 us latency chain per forward for 2 MFLOP per row.
// Here D[feature][token] = W[pd x d] . x[16 tokens x d]^T per wave as d / 32 v_mfma_f32_16x16x32_bf16 steps per 16-feature tile: the token
// rows are ALREADY the operand format (bf16, feature-contiguous: lane (token, k-quarter) loads its 16 bytes straight from the residual
// stream), the weights sit in LDS as a split bf16 pair (hi + lo: fp32-grade products, same trick as cross_row's logits), accumulation fp32.
// A lane ends with 4 consecutive output features of its token: bias, then the pixel-shuffle store (tld/denoiser.py:47-52,72,82).
template <int NT>       // 16-feature tiles covering patch_dim (1 .. 4)
__global__ __launch_bounds__(256) void tail_mfma_kernel(TailParams p, const bf16* __restrict__ whl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = p.d, PITCH = d * 2, pdp = NT * 16;
    char* wh = smem;                                  // [pdp][PITCH] hi halves (rows >= pd: zeros), chunk c of row r at chunk c ^ (r & 15)
    char* wl = smem + pdp * PITCH;                    // lo halves
    const int chunks = d >> 3;
    for (int i = threadIdx.x; i < 2 * pdp * chunks; i += 256) {
        const int plane = i / (pdp * chunks), rem = i - plane * pdp * chunks;
        const int r = rem / chunks, c = rem - r * chunks;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < p.pd) v = *reinterpret_cast<const u32x4*>(whl + ((size_t)plane * p.pd + r) * d + c * 8);
        *reinterpret_cast<u32x4*>((plane ? wl : wh) + r * PITCH + ((c ^ (r & 15)) << 4)) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tok = lane & 15, kq = lane >> 4;
    const int total = p.batch * p.ntok;
    const int row0 = (blockIdx.x * 4 + wid) * 16;
    if (row0 >= total) return;
    const int row = row0 + tok < total ? row0 + tok : total - 1;
    const bf16* xrow = p.tok + (size_t)row * d + kq * 8;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkb = d >> 5;
    // token fragments eight K-blocks ahead of their use (the loop is one dependent MFMA chain per tile: what it must hide is the loads)
    constexpr int PF = 8;
    bf16x8 xf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xrow + (i < nkb ? i : 0) * 32);
    for (int kb0 = 0; kb0 < nkb; kb0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int kb = kb0 + i;
            if (kb >= nkb) break;
            const bf16x8 xb = xf[i];
            if (kb + PF < nkb) xf[i] = *reinterpret_cast<const bf16x8*>(xrow + (kb + PF) * 32);
            const int chunk = 4 * kb + kq;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int r = t * 16 + tok;
                const int off = r * PITCH + ((chunk ^ (r & 15)) << 4);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wh + off), xb, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wl + off), xb, acc[t], 0, 0, 0);
            }
        }
    }
    if (row0 + tok >= total) return;
    const int b = row / p.ntok, tk = row - b * p.ntok;
    const int ti = tk / p.grid, tj = tk - ti * p.grid;
    const int pp = p.p * p.p;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = t * 16 + 4 * kq + r;              // feature f = (c, u, v) of token (ti, tj) -> out[b, c, ti*p+u, tj*p+v]
            if (f < p.pd) {
                const int c = f / pp, uv = f - c * pp;
                const int u = uv / p.p, vv = uv - u * p.p;
                p.out[(((size_t)b * p.C + c) * p.S + (ti * p.p + u)) * p.S + (tj * p.p + vv)] = acc[t][r] + p.b[f];
            }
        }
}
#endif

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void update_kernel(UpdateParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = p.batch * p.img;
    if (i >= n) return;
    const float cond = p.x0_2b[i], unc = p.x0_2b[n + i];
    float x0 = p.g * cond + (1.0f - p.g) * unc;                   // diffusion.py:124-125
    if (p.final_step) {
        const int ch = (i % p.img) / p.chan_stride;
        if (ch == 3) x0 += p.sharp;                               // diffusion.py:88
        if (ch == 0) x0 += p.bright;                              // diffusion.py:89
        p.x0_out[i] = x0;
        return;
    }
    p.x0_out[i] = x0;
    if (p.trace_x0) p.trace_x0[i] = x0;
    const float D = p.c1 * x0 - p.c2 * p.x0_prev[i];              // diffusion.py:76 (c1=1,c2=0: :72/:79)
    const float xt = (p.a * D + p.b * p.x_t[i]) / p.c;            // diffusion.py:81
    p.x_t[i] = xt;
    p.x0_prev[i] = x0;
    if (p.trace_xt) p.trace_xt[i] = xt;
}

// ------------------------------------------------------------------------------------------------
// Depthwise 3x3 + bias + exact GELU.  One workgroup = one sample x 64 channels: the whole g x g image
// slab (g*g tokens x 128 B) is pulled into LDS once with 16-B coalesced loads, so HBM/L2 see every input
// exactly once (the register-window version re-read each row three times through L2).  Thread (row i,
// channel quad) then slides a 3x3 fp32 window along j: 3 ds_read_b64 + 36 FMAs + 4 GELUs + one 8-B
// store per position; weights stay in registers.  GELU uses the Abramowitz-Stegun erf (1.5e-7).
constexpr int DW_CB = 64;                       // channels per workgroup
__global__ __launch_bounds__(256) void dwconv_gelu_kernel(const bf16* __restrict__ in, bf16* __restrict__ out,
                                                          const float* __restrict__ w9c,
                                                          const float* __restrict__ bias, int batch,
                                                          int g, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [g*g tokens][64 ch] bf16
    const int nchunk = C / DW_CB;
    const int b = blockIdx.x / nchunk, cc = blockIdx.x - b * nchunk;
    const int ntok = g * g;
    const bf16* src = in + (size_t)b * ntok * C + cc * DW_CB;
    for (int idx = threadIdx.x; idx < ntok * 8; idx += 256) {       // 8 x 16-B pieces per token
        const int t = idx >> 3, q = idx & 7;
        *reinterpret_cast<uint4*>(smem + t * 128 + q * 16) =
            *reinterpret_cast<const uint4*>(src + (size_t)t * C + q * 8);
    }
    const int cq = threadIdx.x & 15;             // channel quad within the 64-channel slab
    const int c0 = cc * DW_CB + cq * 4;
    float4 w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const float4*>(w9c + (size_t)k * C + c0);
    const float4 bs = *reinterpret_cast<const float4*>(bias + c0);
    __syncthreads();

    for (int i = threadIdx.x >> 4; i < g; i += 16) {
        const bool up_ok = i > 0, dn_ok = i + 1 < g;
        auto load_col = [&](int j, float4 (&col)[3]) {
            const bool jok = j >= 0 && j < g;
#pragma unroll
            for (int du = 0; du < 3; ++du) {
                const bool ok = jok && (du == 1 || (du == 0 ? up_ok : dn_ok));
                if (ok) {
                    const bf16x4 v = *reinterpret_cast<const bf16x4*>(smem + ((i + du - 1) * g + j) * 128 + cq * 8);
                    col[du] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                } else {
                    col[du] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        bf16* dst = out + ((size_t)b * ntok + (size_t)i * g) * C + c0;
        // one output position from the three window columns (L = j-1, M = j, R = j+1); three independent
        // accumulation chains (one per image row) keep the packed-FMA pipeline free of dependency stalls
        auto emit = [&](const float4 (&L)[3], const float4 (&Mc)[3], const float4 (&R)[3], int j) {
            float4 part[3];
#pragma unroll
            for (int du = 0; du < 3; ++du) {
                const float4 w0 = w[du * 3 + 0], w1 = w[du * 3 + 1], w2 = w[du * 3 + 2];
                part[du].x = fmaf(w2.x, R[du].x, fmaf(w1.x, Mc[du].x, w0.x * L[du].x));
                part[du].y = fmaf(w2.y, R[du].y, fmaf(w1.y, Mc[du].y, w0.y * L[du].y));
                part[du].z = fmaf(w2.z, R[du].z, fmaf(w1.z, Mc[du].z, w0.z * L[du].z));
                part[du].w = fmaf(w2.w, R[du].w, fmaf(w1.w, Mc[du].w, w0.w * L[du].w));
            }
            const float ax = (part[0].x + part[1].x) + (part[2].x + bs.x);
            const float ay = (part[0].y + part[1].y) + (part[2].y + bs.y);
            const float az = (part[0].z + part[1].z) + (part[2].z + bs.z);
            const float aw = (part[0].w + part[1].w) + (part[2].w + bs.w);
            bf16x4 o;
            o[0] = (bf16)TLD_DW_GELU(ax); o[1] = (bf16)TLD_DW_GELU(ay);
            o[2] = (bf16)TLD_DW_GELU(az); o[3] = (bf16)TLD_DW_GELU(aw);
            *reinterpret_cast<bf16x4*>(dst + (size_t)j * C) = o;
        };
        // the window rotates through three named column buffers, so no register copies are needed
        float4 c0v[3], c1v[3], c2v[3];
        load_col(-1, c0v);
        load_col(0, c1v);
        int j = 0;
        for (; j + 3 <= g; j += 3) {
            load_col(j + 1, c2v); emit(c0v, c1v, c2v, j);
            load_col(j + 2, c0v); emit(c1v, c2v, c0v, j + 1);
            load_col(j + 3, c1v); emit(c2v, c0v, c1v, j + 2);
        }
        if (j < g) { load_col(j + 1, c2v); emit(c0v, c1v, c2v, j); ++j; }
        if (j < g) { load_col(j + 1, c0v); emit(c1v, c2v, c0v, j); }
    }
}

// Same computation for grids larger than 16x16 (512 / 1024 px latents): one workgroup = one sample x 64 channels x one
// 16x16 spatial tile.  The tile plus a one-token halo (18 x 18 tokens x 128 B) is brought into LDS by direct
// global->LDS DMA (41 pieces of 8 tokens, all in flight at once; the register-staged fill it replaces waited for
// one load per iteration and the kernel ran at 2.7 TB/s).  Halo tokens outside the image are DMA'd from a clamped
// (valid) address and never contribute: rows above / below the image meet zeroed copies of the top / bottom taps,
// columns left / right of it are replaced by zero registers -- no zero fill, no predicated code in the window loop.
template <bool F8OUT>
__global__ __launch_bounds__(256) void dwconv_gelu_tiled_kernel(const bf16* __restrict__ in, bf16* __restrict__ out,
                                                                const float* __restrict__ w9c,
                                                                const float* __restrict__ bias, int batch,
                                                                int g, int C, uint8_t* __restrict__ out8,
                                                                uint8_t* __restrict__ scale8) {
    constexpr int T = 16, TP = T + 2;
    constexpr int PIECES = (TP * TP + 7) / 8;              // 1-KiB DMA pieces (8 tokens x 128 B)
    __shared__ __attribute__((aligned(16))) char tile[PIECES * 1024];
    const int nchunk = C / DW_CB;
    const int tiles = (g + T - 1) / T;
    int bid = blockIdx.x;
    const int cc = bid % nchunk; bid /= nchunk;
    const int tx = bid % tiles; bid /= tiles;
    const int ty = bid % tiles;
    const int b = bid / tiles;
    const int i0 = ty * T - 1, j0 = tx * T - 1;
    const bf16* src = in + (size_t)b * g * g * C + cc * DW_CB;
    {
        typedef const __attribute__((address_space(1))) void* gp_t;
        typedef __attribute__((address_space(3))) void* lp_t;
        const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        for (int pc = wid; pc < PIECES; pc += 4) {
            int t = pc * 8 + (lane >> 3);
            t = t < TP * TP ? t : TP * TP - 1;
            const int li = t / TP, lj = t - li * TP;
            int gi = i0 + li, gj = j0 + lj;
            gi = gi < 0 ? 0 : (gi >= g ? g - 1 : gi);
            gj = gj < 0 ? 0 : (gj >= g ? g - 1 : gj);
            const bf16* sp = src + ((size_t)gi * g + gj) * C + (lane & 7) * 8;
            __builtin_amdgcn_global_load_lds((gp_t)sp, (lp_t)(tile + pc * 1024), 16, 0, 0);
        }
    }
    const int cq = threadIdx.x & 15;
    const int c0 = cc * DW_CB + cq * 4;
    // packed-fp32 arithmetic: channel pairs {c0, c0+1} and {c0+2, c0+3} ride in f32x2 registers.  w9c / bias are the
    // HALVED tables (the conv delivers y = x / 2 and the GELU is evaluated in its half-argument form, see tld_common.h)
    f32x2 w[9][2], bs[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(w9c + (size_t)k * C + c0);
        w[k][0] = f32x2{t.x, t.y}; w[k][1] = f32x2{t.z, t.w};
    }
    {
        const float4 t = *reinterpret_cast<const float4*>(bias + c0);
        bs[0] = f32x2{t.x, t.y}; bs[1] = f32x2{t.z, t.w};
    }
    const int li = threadIdx.x >> 4;                       // output row inside the tile (0..15)
    const int gi = ty * T + li;
    {   // image rows -1 and g: zero the taps that would meet them (the halo row holds a clamped copy)
        const float mu = gi > 0 ? 1.0f : 0.0f, md = gi + 1 < g ? 1.0f : 0.0f;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) { w[t3][h2] *= mu; w[6 + t3][h2] *= md; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // own DMA pieces landed ...
    __syncthreads();                                       // ... everybody's
    if (gi >= g) return;
    auto col = [&](int lj, f32x2 (&c)[3][2]) {             // lj: halo-tile column index 0..17
#pragma unroll
        for (int du = 0; du < 3; ++du) {
            const bf16x4 v = *reinterpret_cast<const bf16x4*>(tile + ((li + du) * TP + lj) * 128 + cq * 8);
            c[du][0] = f32x2{(float)v[0], (float)v[1]};
            c[du][1] = f32x2{(float)v[2], (float)v[3]};
        }
    };
    bf16* dst = out + ((size_t)b * g * g + (size_t)gi * g + (size_t)tx * T) * C + c0;
    const int ncols = g - tx * T < T ? g - tx * T : T;     // (a multiple of the tile width for every supported grid)
    auto emit = [&](const f32x2 (&L)[3][2], const f32x2 (&Mc)[3][2], const f32x2 (&R)[3][2], int lj) {
        f32x2 a[2] = {bs[0], bs[1]};
#pragma unroll
        for (int du = 0; du < 3; ++du)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                a[h2] = __builtin_elementwise_fma(w[du * 3 + 0][h2], L[du][h2], a[h2]);
                a[h2] = __builtin_elementwise_fma(w[du * 3 + 1][h2], Mc[du][h2], a[h2]);
                a[h2] = __builtin_elementwise_fma(w[du * 3 + 2][h2], R[du][h2], a[h2]);
            }
        a[0] = gelu_erf_fast2_half(a[0]); a[1] = gelu_erf_fast2_half(a[1]);
        bf16x4 o;
        o[0] = (bf16)a[0][0]; o[1] = (bf16)a[0][1]; o[2] = (bf16)a[1][0]; o[3] = (bf16)a[1][1];
        if constexpr (F8OUT) {      // fp8 GEMM mode: MX-quantise the bf16-rounded quad (8 adjacent channel quads = one 32-block)
            int e8;
            const unsigned w = mx8_pack4((float)o[0], (float)o[1], (float)o[2], (float)o[3], &e8);
            const size_t row = (size_t)b * g * g + (size_t)gi * g + (size_t)tx * T + lj;
            if (lj < ncols) {
                *reinterpret_cast<unsigned*>(out8 + row * C + c0) = w;
                if ((cq & 7) == 0) scale8[mx8_scale_index(c0, row, (size_t)batch * g * g)] = (uint8_t)e8;
            }
        } else {
            if (lj < ncols) *reinterpret_cast<bf16x4*>(dst + (size_t)lj * C) = o;
        }
    };
    auto zero = [&](f32x2 (&c)[3][2]) {
#pragma unroll
        for (int du = 0; du < 3; ++du) { c[du][0] = f32x2{0.f, 0.f}; c[du][1] = f32x2{0.f, 0.f}; }
    };
    // the window rotates through three named column buffers (no register copies); 16 = 5 x 3 + 1 positions
    f32x2 c0v[3][2], c1v[3][2], c2v[3][2];
    if (tx == 0) zero(c0v); else col(0, c0v);              // image column -1
    col(1, c1v);
    int lj = 0;
    for (; lj + 3 <= T; lj += 3) {
        col(lj + 2, c2v); emit(c0v, c1v, c2v, lj);
        col(lj + 3, c0v); emit(c1v, c2v, c0v, lj + 1);
        col(lj + 4, c1v); emit(c2v, c0v, c1v, lj + 2);
    }
    if (tx * T + T >= g) zero(c2v); else col(lj + 2, c2v);  // image column g
    emit(c0v, c1v, c2v, lj);                                // lj == 15
}

}  // namespace

// d = 64 k (k = 1 .. 16): NJ = ceil(d / 128) feature groups per lane, HALF when the last one holds 64 features
#define TLD_DISPATCH_NJ_(nj, HF, ...)                                                               \
    switch (nj) {                                                                                    \
        case 1: { constexpr int NJ = 1; constexpr bool HALF = HF; __VA_ARGS__; } break;                    \
        case 2: { constexpr int NJ = 2; constexpr bool HALF = HF; __VA_ARGS__; } break;                    \
        case 3: { constexpr int NJ = 3; constexpr bool HALF = HF; __VA_ARGS__; } break;                    \
        case 4: { constexpr int NJ = 4; constexpr bool HALF = HF; __VA_ARGS__; } break;                    \
        case 5: { constexpr int NJ = 5; constexpr bool HALF = HF; __VA_ARGS__; } break;                    \
        case 6: { constexpr int NJ = 6; constexpr bool HALF = HF; __VA_ARGS__; } break;                    \
        case 7: { constexpr int NJ = 7; constexpr bool HALF = HF; __VA_ARGS__; } break;                    \
        case 8: { constexpr int NJ = 8; constexpr bool HALF = HF; __VA_ARGS__; } break;                    \
        default: break;                                                                              \
    }
#define TLD_DISPATCH_D(dd, ...)                                                                      \
    do {                                                                                             \
        if ((dd) % 128) TLD_DISPATCH_NJ_(((dd) + 127) / 128, true, __VA_ARGS__) else TLD_DISPATCH_NJ_((dd) / 128, false, __VA_ARGS__)  \
    } while (0)

// dynamic LDS above the 64-KiB default needs an opt-in per kernel and device (embed / out-proj / cross-attention tables at d = 1024)
#define TLD_LDS_OPT_IN(KERNEL, bytes)                                                                \
    do {                                                                                             \
        static PerDeviceMax optin;                                                                   \
        if ((bytes) > 64 * 1024)                                                                     \
            optin.run((bytes), [&] { hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); }); \
    } while (0)

void launch_embed(const EmbedParams& p, hipStream_t s) {
#ifdef TLD_RESID_BF16
    if (p.lin_w_hl && p.pd == 16 && p.C * p.p * p.p == 16 && p.d % 256 == 0 && p.d <= 1024) {       // matrix-pipe form
        const int rows = p.batch * p.ntok;
        const int lds = 16 * 16 * 4 + 4 * 4 * 32 * 4 + 3 * p.d * 4 + 4 * 32 * 144;      // conv weight, row partials, bias / LayerNorm vectors, transpose patches
        dim3 grid((unsigned)((rows + 31) / 32));
#define TLD_EM(TPW) do { TLD_LDS_OPT_IN((embed_mfma_kernel<TPW>), lds); hipLaunchKernelGGL((embed_mfma_kernel<TPW>), grid, dim3(256), lds, s, p, p.lin_w_hl); } while (0)
        if (p.d == 256) TLD_EM(2); else if (p.d == 512) TLD_EM(4); else if (p.d == 768) TLD_EM(6); else TLD_EM(8);
#undef TLD_EM
        return;
    }
#endif
    const int rows = p.batch * p.ntok;
    const int lds = (p.pd * p.d + p.pd * p.C * p.p * p.p) * (int)sizeof(float);
    TLD_DISPATCH_D(p.d, { TLD_LDS_OPT_IN((embed_kernel<NJ, HALF>), lds); hipLaunchKernelGGL((embed_kernel<NJ, HALF>), dim3((rows + 31) / 32), dim3(256), lds, s, p); });
}

void launch_layernorm_bf16(const resid_t* x, const float* g, const float* b, bf16* out, int M, int d,
                           hipStream_t s) {
    if (d == 768) { hipLaunchKernelGGL(layernorm_bf16_q4_kernel<3>, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, out, M, d); return; }
    if (d == 512) { hipLaunchKernelGGL(layernorm_bf16_q4_kernel<2>, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, out, M, d); return; }
    if (d == 256) { hipLaunchKernelGGL(layernorm_bf16_q4_kernel<1>, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, out, M, d); return; }
    if (d == 1024) { hipLaunchKernelGGL(layernorm_bf16_q4_kernel<4>, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, out, M, d); return; }
    TLD_DISPATCH_D(d, hipLaunchKernelGGL((layernorm_bf16_kernel<NJ, HALF>), dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, out, M, d));
}

bool layernorm_mx8_supported(int d) { return d == 256 || d == 512 || d == 768; }      // (the engine's embed_dim limit is 896)

void launch_layernorm_mx8(const resid_t* x, const float* g, const float* b, uint8_t* out8, uint8_t* scale8, int M, int d,
                          hipStream_t s) {
    const dim3 gr((M + 3) / 4), bl(256);
    if (d == 768) hipLaunchKernelGGL(layernorm_mx8_kernel<3>, gr, bl, 0, s, x, g, b, out8, scale8, M, d);
    else if (d == 512) hipLaunchKernelGGL(layernorm_mx8_kernel<2>, gr, bl, 0, s, x, g, b, out8, scale8, M, d);
    else if (d == 256) hipLaunchKernelGGL(layernorm_mx8_kernel<1>, gr, bl, 0, s, x, g, b, out8, scale8, M, d);
    else if (d == 1024) hipLaunchKernelGGL(layernorm_mx8_kernel<4>, gr, bl, 0, s, x, g, b, out8, scale8, M, d);
}

// the LN3-statistics output (CrossRowParams::ln3_stats) exists in the matrix-pipe kernel only
bool cross_row_supports_ln3_stats(int d) { return d == 768 || d == 512 || d == 256; }

void launch_cross_row(const CrossRowParams& p, hipStream_t s) {
    if (p.d % 256 == 0 && p.d <= 1024 && p.ntok % 16 == 0) {
        // 512-thread workgroups over 16-row groups of one sample, ONE per CU (the kernel holds ~170 registers at d = 768: the split-bf16 slices of
        // the query-difference vectors and the prefetched row pair live in registers; two workgroups per CU at 128 registers spilled and ran
        // 43.7 us against 38.0).  Groups per workgroup: the largest power of two (<= 16) that divides a sample's groups and still leaves >= 1
        // workgroup per CU (at C1: 128 samples x 16 groups / 8 = 256 workgroups = one resident round)
        const int gps = p.ntok / 16;
        const long want = device_cu_count();
        int gpw = 1;
        while (gpw < 16 && gps % (gpw * 2) == 0 && (long)p.batch * (gps / (gpw * 2)) >= want) gpw *= 2;
        const int ldsm = 2 * 16 * p.d * 2 + 8 * 256 * 4 + 2 * p.d * 4;     // split-bf16 tile, partial logits, the two value rows
        dim3 gridm((unsigned)(p.batch * (gps / gpw)));
#define TLD_CRM(NQ) do { TLD_LDS_OPT_IN((cross_row_mfma_kernel<NQ>), ldsm); hipLaunchKernelGGL((cross_row_mfma_kernel<NQ>), gridm, dim3(512), ldsm, s, p, gpw); } while (0)
        if (p.d == 256) TLD_CRM(1); else if (p.d == 512) TLD_CRM(2); else if (p.d == 768) TLD_CRM(3); else TLD_CRM(4);
#undef TLD_CRM
        return;
    }
    // other widths (any multiple of 64): the 2-features-per-lane VALU kernel.  ~43 KB of LDS per workgroup -> 3 workgroups per CU.  Split each
    // sample's row pairs into the number of chunks that makes the grid ONE full resident round (768 workgroups on 256 CUs) when the batch
    // allows, else k rounds of <= ~48 rows per workgroup; a partial extra round costs as much as a full one.
    const int slots = 3 * device_cu_count();
    const long rows = (long)p.batch * p.ntok;
    const long k = (rows + (long)slots * 48 - 1) / ((long)slots * 48);
    long cps = slots * k / p.batch;
    const long max_cps = p.ntok / 8 > 0 ? p.ntok / 8 : 1;          // at least one row pair per wave
    if (cps > max_cps) cps = max_cps;
    if (cps < 1) cps = 1;
    const int lds = (p.heads * p.d + 2 * p.d + p.heads) * (int)sizeof(float);
    dim3 grid((unsigned)(p.batch * cps));
    TLD_DISPATCH_D(p.d, { TLD_LDS_OPT_IN((cross_row_kernel<NJ, HALF>), lds); hipLaunchKernelGGL((cross_row_kernel<NJ, HALF>), grid, dim3(256), lds, s, p, (int)cps); });
}

// ------------------------------------------------------------------------------------------------
// Low-latency class (round 5): the down projection of a SMALL batch as split-K -- `nsplit` fp32 slices from the persistent GEMM -- finished here:
//   x[row] = bf16(x[row] + bias + slice_0[row] + ... + slice_{nsplit-1}[row])     (fixed order)   tld/transformer_blocks.py:104,138
// plus the LayerNorm-1 partial sums (sum, sum of squares of the ROUNDED row per 96-column group) the next block's fused QKV kernel reads --
// what EPI_BIAS_RESID does in its own epilogue.  One wave per row, lane l owns columns [CPL l, CPL (l + 1)): a group is 96 / CPL adjacent lanes.
template <int CPL>
__global__ __launch_bounds__(256) void splitk_resid_kernel(const float* __restrict__ parts, int nsplit, size_t slice_stride, const float* __restrict__ bias,
                                                           resid_t* __restrict__ x, float2* __restrict__ stats_out, int M) {
    constexpr int d = CPL * 64, GL = 96 / CPL;                  // lanes per statistics group (8 at d = 768, 16 at d = 384)
    static_assert(96 % CPL == 0 && (GL & (GL - 1)) == 0 && CPL % 2 == 0, "a 96-column group must be a power-of-two run of lanes");
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c += 2) {
        const float2 b2 = *reinterpret_cast<const float2*>(bias + CPL * lane + c);
        v[c] = b2.x; v[c + 1] = b2.y;
    }
    for (int sp = 0; sp < nsplit; ++sp) {
        const float* src = parts + sp * slice_stride + (size_t)row * d + CPL * lane;
#pragma unroll
        for (int c = 0; c < CPL; c += 2) {
            const float2 t = *reinterpret_cast<const float2*>(src + c);
            v[c] += t.x; v[c + 1] += t.y;
        }
    }
    resid_t* px = x + (size_t)row * d + CPL * lane;
    float su = 0.f, sq = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; c += 2) {
        const float2 xv = rs_load2(px + c);
        const float o0 = xv.x + v[c], o1 = xv.y + v[c + 1];
        rs_store2(px + c, make_float2(o0, o1));
        const float r0 = rs_round(o0), r1 = rs_round(o1);
        su += r0 + r1;
        sq = fmaf(r0, r0, fmaf(r1, r1, sq));
    }
    if (stats_out) {
#pragma unroll
        for (int o2 = 1; o2 < GL; o2 <<= 1) { su += __shfl_xor(su, o2, 64); sq += __shfl_xor(sq, o2, 64); }
        const int slot = lane / GL;
        if ((lane & (GL - 1)) == 0 && slot < kLnSlots) stats_out[(size_t)row * kLnSlots + slot] = make_float2(su, sq);
    }
}

bool splitk_resid_supported(int d) { return d == 768 || d == 384; }

void launch_splitk_resid(const float* parts, int nsplit, size_t slice_stride, const float* bias, resid_t* x, float2* stats_out, int M, int d, hipStream_t s) {
    const dim3 grid((unsigned)((M + 3) / 4)), block(256);
    if (d == 768) hipLaunchKernelGGL(splitk_resid_kernel<12>, grid, block, 0, s, parts, nsplit, slice_stride, bias, x, stats_out, M);
    else if (d == 384) hipLaunchKernelGGL(splitk_resid_kernel<6>, grid, block, 0, s, parts, nsplit, slice_stride, bias, x, stats_out, M);
}

void launch_tail(const TailParams& p, hipStream_t s) {
#ifdef TLD_RESID_BF16
    if (p.w_hl && p.d % 128 == 0 && (long)p.pd * p.d <= 40960) {      // matrix-pipe form: split-bf16 weights [2][pd][d] in <= 160 KiB of LDS (the XOR swizzle of its
                                                                      // weight image permutes 16-byte chunks within aligned groups of 16: rows of d / 8 = 16 k chunks)
        const int nt = (p.pd + 15) / 16, rows = p.batch * p.ntok;
        const int lds = 2 * nt * 16 * p.d * 2;
        dim3 grid((unsigned)((rows + 63) / 64));
#define TLD_TM(NT) do { TLD_LDS_OPT_IN((tail_mfma_kernel<NT>), lds); hipLaunchKernelGGL((tail_mfma_kernel<NT>), grid, dim3(256), lds, s, p, p.w_hl); } while (0)
        if (nt == 1) TLD_TM(1); else if (nt == 2) TLD_TM(2); else if (nt == 3) TLD_TM(3); else TLD_TM(4);
#undef TLD_TM
        return;
    }
#endif
    const int rpb = 64;
    const int rows = p.batch * p.ntok;
    const int lds = p.pd * p.d * (int)sizeof(float);
    TLD_DISPATCH_D(p.d, { TLD_LDS_OPT_IN((tail_kernel<NJ, HALF>), lds); hipLaunchKernelGGL((tail_kernel<NJ, HALF>), dim3((rows + rpb - 1) / rpb), dim3(256), lds, s, p, rpb); });
}

void launch_update(const UpdateParams& p, hipStream_t s) {
    const int n = p.batch * p.img;
    hipLaunchKernelGGL(update_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p);
}

// Streaming form of the same computation for grids that are a multiple of 32 wide (512 / 1024 px latents; round 3): one workgroup =
// one sample x 64 channels x a 32-column strip, walking DOWN the image.  The tiled kernel above loads an 18 x 18 halo tile, waits, computes,
// stores -- 1.27 x the input bytes and three phases per workgroup that only other workgroups can overlap (107 us = 3.8 TB/s at C3).  Here
//   * a fifth wave does nothing but DMA: image row r (32 tokens x 128 B = four 1-KiB pieces, + one halo token per inner strip edge) goes
//     into slot r % NR of an LDS ring, 4 - 5 rows ahead of its first use, with a counted vmcnt of its own (the four computing waves'
//     vmcnt sees only their stores, which nobody waits for);
//   * the computing waves keep the 3-row window in REGISTERS (a thread owns 2 adjacent columns x 4 channels): every input row is read from
//     LDS once and from memory once (1.0 x, + 1/32 per inner strip edge), one barrier per row;
//   * image borders: row -1 / row g are zero registers, columns -1 / g zeroed LDS cells the DMA never writes.
// Same halved taps, bias and half-argument GELU as the tiled kernel, same fp8 output option.
template <bool F8OUT>
__global__ __launch_bounds__(320, 4) void dwconv_gelu_stream_kernel(const bf16* __restrict__ in, bf16* __restrict__ out,
                                                                 const float* __restrict__ w9c, const float* __restrict__ bias, int batch,
                                                                 int g, int C, uint8_t* __restrict__ out8, uint8_t* __restrict__ scale8) {
    constexpr int SW = 32, NR = 7, PF = NR - 1;
    constexpr int RPITCH = (SW + 2) * 128;                  // one ring row: tokens j0 - 1 .. j0 + 32
    __shared__ __attribute__((aligned(16))) char ring[NR * RPITCH];
    typedef const __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    const int nchunk = C / DW_CB, strips = g / SW;
    int bid = blockIdx.x;
    const int cc = bid % nchunk; bid /= nchunk;
    const int sx = bid % strips;
    const int b = bid / strips;
    const int j0 = sx * SW;
    const bool edge_l = sx > 0, edge_r = sx + 1 < strips;   // inner strip edges: a halo token comes from the neighbouring strip
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bf16* src = in + (size_t)b * g * g * C + cc * DW_CB;

    // image columns -1 / g: zero cells (never written by the DMA)
    if (threadIdx.x < NR * 32) {
        const int r = threadIdx.x >> 5, w = threadIdx.x & 31;
        if (!edge_l) *reinterpret_cast<unsigned*>(ring + r * RPITCH + w * 4) = 0u;
        if (!edge_r) *reinterpret_cast<unsigned*>(ring + r * RPITCH + (SW + 1) * 128 + w * 4) = 0u;
    }
    auto stage_row = [&](int r) {       // producer wave only
        char* dst = ring + (r % NR) * RPITCH;
        const bf16* rp = src + ((size_t)r * g + j0) * C + (lane & 7) * 8;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc)
            __builtin_amdgcn_global_load_lds((gp_t)(rp + (size_t)(pc * 8 + (lane >> 3)) * C), (lp_t)(dst + 128 + pc * 1024), 16, 0, 0);
        if (lane < 8) {                 // (both halo loads are issued by every row of a strip with that edge: the count per row is uniform)
            if (edge_l) __builtin_amdgcn_global_load_lds((gp_t)(rp - C), (lp_t)dst, 16, 0, 0);
            if (edge_r) __builtin_amdgcn_global_load_lds((gp_t)(rp + (size_t)SW * C), (lp_t)(dst + (SW + 1) * 128), 16, 0, 0);
        }
    };
    const int ndma = 4 + (edge_l ? 1 : 0) + (edge_r ? 1 : 0);
    // before barrier B_y rows <= y + 5 have been issued and rows <= y + 1 must have landed: at most PF - 2 = 4 rows may still be in flight
    auto wait_rows = [&]() {
        if (ndma == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (PF - 2)) : "memory");
        else if (ndma == 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (PF - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (PF - 2)) : "memory");
    };
    __syncthreads();                    // zero cells written before any row is read

    if (wid == 4) {
        // ---- producer: rows 0 .. PF up front, then one row per barrier
        for (int r = 0; r <= PF && r < g; ++r) stage_row(r);
        for (int y = 0; y < g; ++y) {
            if (y + PF + 1 >= g) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tail: fewer rows in flight than the counted wait assumes
            else wait_rows();
            __builtin_amdgcn_s_barrier();                    // B_y: rows <= y + 1 are in LDS; slot of row y is free
            if (y >= 1 && y + PF < g) stage_row(y + PF);
        }
        return;
    }
    // ---- four computing waves: thread = channel quad cq x column pair cg
    const int cq = threadIdx.x & 15, cg = threadIdx.x >> 4;
    const int c0 = cc * DW_CB + cq * 4;
    f32x2 w[9][2], bs[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(w9c + (size_t)k * C + c0);
        w[k][0] = f32x2{t.x, t.y}; w[k][1] = f32x2{t.z, t.w};
    }
    {
        const float4 t = *reinterpret_cast<const float4*>(bias + c0);
        bs[0] = f32x2{t.x, t.y}; bs[1] = f32x2{t.z, t.w};
    }
    // window rows: [column 2 cg - 1 .. 2 cg + 2][channel pair]
    f32x2 R0[4][2], R1[4][2], R2[4][2];
    auto zero_row = [&](f32x2 (&R)[4][2]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { R[k][0] = f32x2{0.f, 0.f}; R[k][1] = f32x2{0.f, 0.f}; }
    };
    auto read_row = [&](int r, f32x2 (&R)[4][2]) {
        const char* rp = ring + (r % NR) * RPITCH + (2 * cg) * 128 + cq * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bf16x4 v = *reinterpret_cast<const bf16x4*>(rp + k * 128);
            R[k][0] = f32x2{(float)v[0], (float)v[1]};
            R[k][1] = f32x2{(float)v[2], (float)v[3]};
        }
    };
    // output addresses as running pointers, one add per image row (round 6: `row * C + c0` and the scale index rebuilt per output column were ~20 integer
    // instructions -- 64-bit multiplies among them -- of the ~90 per column of this VALU-bound loop)
    // (32-bit byte offsets from the uniform bases: the engine keeps max_batch x tokens x 4 d x 2 B below 4 GiB, tld_engine_create)
    const size_t orow0 = (size_t)b * g * g + j0 + 2 * cg;
    unsigned ooff = (unsigned)((orow0 * C + c0) * (F8OUT ? 1 : 2));                       // this lane's output column 2 cg of image row y, bytes
    unsigned soff = F8OUT ? (unsigned)mx8_scale_index(c0, orow0, (size_t)batch * g * g) : 0u;
    const unsigned ostride = (unsigned)g * (unsigned)C * (F8OUT ? 1u : 2u);               // bytes per image row
    auto emit = [&](const f32x2 (&U)[4][2], const f32x2 (&M)[4][2], const f32x2 (&D)[4][2]) {
#pragma unroll
        for (int oc = 0; oc < 2; ++oc) {                     // output column 2 cg + oc: window columns oc .. oc + 2
            f32x2 a[2] = {bs[0], bs[1]};
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    a[h2] = __builtin_elementwise_fma(w[0 + dx][h2], U[oc + dx][h2], a[h2]);
                    a[h2] = __builtin_elementwise_fma(w[3 + dx][h2], M[oc + dx][h2], a[h2]);
                    a[h2] = __builtin_elementwise_fma(w[6 + dx][h2], D[oc + dx][h2], a[h2]);
                }
            a[0] = gelu_erf_fast2_half(a[0]); a[1] = gelu_erf_fast2_half(a[1]);
            bf16x4 o;
            o[0] = (bf16)a[0][0]; o[1] = (bf16)a[0][1]; o[2] = (bf16)a[1][0]; o[3] = (bf16)a[1][1];
            if constexpr (F8OUT) {
                int e8;
                const unsigned pk = mx8_pack4((float)o[0], (float)o[1], (float)o[2], (float)o[3], &e8);
                *reinterpret_cast<unsigned*>(out8 + ooff + (unsigned)(oc * C)) = pk;
                if ((cq & 7) == 0) scale8[soff + oc * 4] = (uint8_t)e8;
            } else {
                *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(out) + ooff + (unsigned)(oc * C * 2)) = o;
            }
        }
        ooff += ostride;
        if constexpr (F8OUT) soff += (unsigned)g * 4u;
    };
    zero_row(R0);                                            // image row -1
    __builtin_amdgcn_s_barrier();                            // B_0: rows 0, 1 landed
    read_row(0, R1);
    // three named window rows rotate (no register copies): y, y + 1, y + 2 per trip
    auto step = [&](f32x2 (&U)[4][2], f32x2 (&M)[4][2], f32x2 (&D)[4][2], int y) {
        if (y > 0) __builtin_amdgcn_s_barrier();             // B_y
        if (y + 1 < g) read_row(y + 1, D); else zero_row(D);
        emit(U, M, D);
    };
    int y = 0;
    for (; y + 3 <= g; y += 3) {
        step(R0, R1, R2, y);
        step(R1, R2, R0, y + 1);
        step(R2, R0, R1, y + 2);
    }
    if (y < g) { step(R0, R1, R2, y); ++y; }
    if (y < g) { step(R1, R2, R0, y); }
}

// ---- seam rows of the depthwise conv fused into the up-projection at 32 x 32 tokens (GemmParams::dw_grid = 32, tld_gemm.hip) ----------------------
// A 256-row GEMM tile is 8 rows of the image; its epilogue finishes rows 1 .. 6 (and the image's own border rows) and leaves its hidden rows 0, 1, 6, 7
// in `seam` as token-pair dwords: [tile][column tile][64 pair-rows: rows 0, 1, 6, 7 x 16 pair-columns][256 channels].  Here: the output rows on either
// side of each of a sample's three tile seams -- window rows (6, 7) of the upper tile and (0, 1) of the lower one -- with the epilogue's own tap
// arithmetic (halved taps as packed bf16 pairs, v_dot2 accumulation in fp32, GELU of twice the argument).  One workgroup per (sample, seam, column
// tile): 64 channel quads x 4 column quarters.
__global__ __launch_bounds__(256) void dwconv_seam_kernel(const uint32_t* __restrict__ seam, const uint32_t* __restrict__ wpk, const float* __restrict__ bias_half,
                                                          bf16* __restrict__ out, int ldo, int channels) {
    const int ntn = channels >> 8;
    int bid = blockIdx.x;
    const int n = bid % ntn; bid /= ntn;
    const int sm = bid % 3, b = bid / 3;
    const int cq = threadIdx.x & 63, qq = threadIdx.x >> 6;
    const int c0 = n * 256 + cq * 4;
    const uint32_t* lo = seam + ((size_t)(b * 4 + sm) * ntn + n) * (64 * 256) + cq * 4;        // upper tile: its rows 6, 7 are pair-rows 32 .. 63
    const uint32_t* hi = seam + ((size_t)(b * 4 + sm + 1) * ntn + n) * (64 * 256) + cq * 4;    // lower tile: its rows 0, 1 are pair-rows 0 .. 31
    u32x4 WA[3], WB[3], WC[3], WD[3];
#pragma unroll
    for (int du = 0; du < 3; ++du) {
        WA[du] = *reinterpret_cast<const u32x4*>(wpk + (size_t)(du * 4 + 0) * channels + c0);
        WB[du] = *reinterpret_cast<const u32x4*>(wpk + (size_t)(du * 4 + 1) * channels + c0);
        WC[du] = *reinterpret_cast<const u32x4*>(wpk + (size_t)(du * 4 + 2) * channels + c0);
        WD[du] = *reinterpret_cast<const u32x4*>(wpk + (size_t)(du * 4 + 3) * channels + c0);
    }
    const float4 bsv = *reinterpret_cast<const float4*>(bias_half + c0);
    const f32x4 bs = {bsv.x, bsv.y, bsv.z, bsv.w};
    // the six pair-columns 4 qq - 1 .. 4 qq + 4 of the four window rows (zero outside the image)
    u32x4 w[6][4];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int q = 4 * qq - 1 + k;
        const bool ok = q >= 0 && q < 16;
        const int qc = ok ? q : 0;
        const u32x4 z = {0u, 0u, 0u, 0u};
        w[k][0] = ok ? *reinterpret_cast<const u32x4*>(lo + (size_t)(32 + qc) * 256) : z;
        w[k][1] = ok ? *reinterpret_cast<const u32x4*>(lo + (size_t)(48 + qc) * 256) : z;
        w[k][2] = ok ? *reinterpret_cast<const u32x4*>(hi + (size_t)qc * 256) : z;
        w[k][3] = ok ? *reinterpret_cast<const u32x4*>(hi + (size_t)(16 + qc) * 256) : z;
    }
    auto dot2 = [](unsigned a, unsigned b2, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b2), c, false);
    };
    bf16* dst0 = out + ((size_t)b * 1024 + (size_t)(8 * sm + 7) * 32) * ldo + c0;             // image row 8 sm + 7; the next row is 32 tokens on
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32x4 (&L)[4] = w[k]; const u32x4 (&Mc)[4] = w[k + 1]; const u32x4 (&R)[4] = w[k + 2];
        const int q = 4 * qq + k;
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
            *reinterpret_cast<bf16x4*>(dst0 + ((size_t)rr * 32 + 2 * q) * ldo) = oe;
            *reinterpret_cast<bf16x4*>(dst0 + ((size_t)rr * 32 + 2 * q + 1) * ldo) = oo;
        }
    }
}

void launch_dwconv_seam(const uint32_t* seam, const uint32_t* dw_wpk, const float* dw_b_half, bf16* out, int ldo, int batch, int channels, hipStream_t s) {
    hipLaunchKernelGGL(dwconv_seam_kernel, dim3((unsigned)(batch * 3 * (channels >> 8))), dim3(256), 0, s, seam, dw_wpk, dw_b_half, out, ldo, channels);
}

void launch_dwconv_gelu(const bf16* in, bf16* out, const float* w9c, const float* bias, const float* w9c_half,
                        const float* bias_half, int batch, int grid, int channels, hipStream_t s, uint8_t* out8,
                        uint8_t* scale8) {
    if (grid > 16 && grid % 32 == 0) {       // row-streaming variant (ring of image rows in LDS, a DMA wave); halved tables
        const dim3 gr((unsigned)(batch * (grid / 32) * (channels / DW_CB)));
        if (out8) hipLaunchKernelGGL(dwconv_gelu_stream_kernel<true>, gr, dim3(320), 0, s, in, out, w9c_half, bias_half, batch, grid, channels, out8, scale8);
        else hipLaunchKernelGGL(dwconv_gelu_stream_kernel<false>, gr, dim3(320), 0, s, in, out, w9c_half, bias_half, batch, grid, channels, out8, scale8);
        return;
    }
    if (grid > 16) {        // spatially tiled variant (halo in LDS); takes the halved tables
        const int tiles = (grid + 15) / 16;
        const dim3 gr((unsigned)(batch * tiles * tiles * (channels / DW_CB)));
        if (out8) hipLaunchKernelGGL(dwconv_gelu_tiled_kernel<true>, gr, dim3(256), 0, s, in, out, w9c_half, bias_half, batch, grid,
                                     channels, out8, scale8);
        else hipLaunchKernelGGL(dwconv_gelu_tiled_kernel<false>, gr, dim3(256), 0, s, in, out, w9c_half, bias_half, batch, grid,
                                channels, out8, scale8);
        return;
    }
    const int lds = grid * grid * 128;
    hipLaunchKernelGGL(dwconv_gelu_kernel, dim3((unsigned)(batch * (channels / DW_CB))), dim3(256), lds, s, in, out,
                       w9c, bias, batch, grid, channels);
}

}  // namespace tld
