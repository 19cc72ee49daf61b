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
// (float2 per access, 512 B per wave-instruction), so d must be a multiple of 128.
#include "tld_common.h"

namespace tld {

namespace {


// ------------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(256) void embed_kernel(EmbedParams p) {
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));   // token row
    if (row >= p.batch * p.ntok) return;
    const int b = row / p.ntok, t = row - b * p.ntok;
    const int ti = t / p.grid, tj = t - ti * p.grid;
    const int pp = p.p * p.p, cpp = p.C * pp;

    // the patch's C*p*p input values, one per lane (lane = (c, u, v))
    float xin = 0.f;
    if (lane < cpp) {
        const int c = lane / pp, uv = lane - c * pp, u = uv / p.p, v = uv - u * p.p;
        xin = p.x[(((size_t)(b % p.src_batch) * p.C + c) * p.S + (ti * p.p + u)) * p.S + (tj * p.p + v)];
    }
    // patchify conv: lane o < pd computes feature o
    float pv = lane < p.pd ? p.conv_b[lane] : 0.f;
    const float* cw = p.conv_w + (lane < p.pd ? lane : 0) * cpp;
    for (int i = 0; i < cpp; ++i) {
        const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xin), i));
        pv = fmaf(cw[i], xv, pv);
    }
    if (lane >= p.pd) pv = 0.f;
    // LN over pd
    const float inv_pd = 1.0f / (float)p.pd;
    const float mean1 = wave_sum(pv) * inv_pd;
    const float dv = lane < p.pd ? pv - mean1 : 0.f;
    const float rstd1 = 1.0f / sqrtf(wave_sum(dv * dv) * inv_pd + kLnEps);
    float pn = 0.f;
    if (lane < p.pd) pn = dv * rstd1 * p.ln1_w[lane] + p.ln1_b[lane];

    // Linear pd -> d: lane owns features {2l, 2l+1} + 128 j
    float2 e[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) e[j] = *reinterpret_cast<const float2*>(p.lin_b + j * 128 + 2 * lane);
#pragma unroll 4
    for (int o = 0; o < p.pd; ++o) {
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pn), o));
        const float* wrow = p.lin_wt + (size_t)o * p.d + 2 * lane;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float2 w = *reinterpret_cast<const float2*>(wrow + j * 128);
            e[j].x = fmaf(a, w.x, e[j].x); e[j].y = fmaf(a, w.y, e[j].y);
        }
    }
    // LN over d
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s += e[j].x + e[j].y;
    const float mean2 = wave_sum(s) / (float)p.d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { e[j].x -= mean2; e[j].y -= mean2; q += e[j].x * e[j].x + e[j].y * e[j].y; }
    const float rstd2 = 1.0f / sqrtf(wave_sum(q) / (float)p.d + kLnEps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = j * 128 + 2 * lane;
        const float2 g = *reinterpret_cast<const float2*>(p.ln2_w + n);
        const float2 bb = *reinterpret_cast<const float2*>(p.ln2_b + n);
        const float2 pe = *reinterpret_cast<const float2*>(p.pos + (size_t)t * p.d + n);
        float2 o;
        o.x = e[j].x * rstd2 * g.x + bb.x + pe.x;
        o.y = e[j].y * rstd2 * g.y + bb.y + pe.y;
        *reinterpret_cast<float2*>(p.tok + (size_t)row * p.d + n) = o;
    }
}

// ------------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(const resid_t* __restrict__ x,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ b,
                                                             bf16* __restrict__ out, int M, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float2 v[NJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        v[j] = *reinterpret_cast<const float2*>(x + (size_t)row * d + j * 128 + 2 * lane);
        s += v[j].x + v[j].y;
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { v[j].x -= mean; v[j].y -= mean; q += v[j].x * v[j].x + v[j].y * v[j].y; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + kLnEps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = j * 128 + 2 * lane;
        const float2 gg = *reinterpret_cast<const float2*>(g + n);
        const float2 bb = *reinterpret_cast<const float2*>(b + n);
        bf16x2 o;
        o[0] = (bf16)(v[j].x * rstd * gg.x + bb.x);
        o[1] = (bf16)(v[j].y * rstd * gg.y + bb.y);
        *reinterpret_cast<bf16x2*>(out + (size_t)row * d + n) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Cross-attention has only two keys per sample (the noise token and the label token), so
//   softmax([q.k_n, q.k_l] / 8) = [1 - s, s],  s = sigmoid((q.k_l - q.k_n) / 8),
// and q.k_t / 8 = LN2(x) . (Wq_h^T k_t[h] / 8): the query projection folds into one d-vector per
// (token row, head), prepared once on the conditioning path (wq table, LN2 gamma folded in, LN2 beta
// contribution in bwq).  The sub-block therefore needs no GEMM: per row it is 12 dot products of
// the centred row against LDS-resident vectors, a sigmoid per head, and a blend of the two value rows.
template <int NJ>
__global__ __launch_bounds__(256) void cross_row_kernel(CrossRowParams p, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int d = NJ * 128, H = NJ * 2;
    float* wd = reinterpret_cast<float*>(smem);          // [H][d]  wq_label - wq_noise (gamma folded)
    float* vn = wd + H * d;                              // [d]     value row of the noise token
    float* vdiff = vn + d;                               // [d]     v_label - v_noise
    float* bw = vdiff + d;                               // [H]     beta contribution to the logit diff

    const int blocks_per_sample = p.ntok / rows_per_block;
    const int b = blockIdx.x / blocks_per_sample;
    const int r0 = (blockIdx.x - b * blocks_per_sample) * rows_per_block;
    const int tn = p.noise_row[b], tl = p.label_row[b];

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

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool upper = lane >= 32;
    const int rows_per_wave = rows_per_block / 4;
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const size_t row = (size_t)b * p.ntok + r0 + wid * rows_per_wave + rr;
        float2 v[NJ];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = j * 128 + 2 * lane;
            const float2 xv = *reinterpret_cast<const float2*>(p.x + row * d + n);
            const bf16x2 av = *reinterpret_cast<const bf16x2*>(p.att + row * d + n);
            v[j].x = xv.x + (float)av[0];                   // x = SA(LN1 x) + x
            v[j].y = xv.y + (float)av[1];
            s += v[j].x + v[j].y;
            if (p.sa_out) *reinterpret_cast<float2*>(p.sa_out + row * d + n) = v[j];
        }
        const float mean = wave_sum(s) / (float)d;
        float2 c[NJ];
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            c[j].x = v[j].x - mean; c[j].y = v[j].y - mean;
            q += c[j].x * c[j].x + c[j].y * c[j].y;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + kLnEps);

        // per-head logit difference -> sigmoid weight of the label token.
        // lane's features of group j belong to head 2j + (lane >> 5)
        float plab[NJ];
#pragma unroll
        for (int hh = 0; hh < NJ; ++hh) {
            float part0 = 0.f, part1 = 0.f;
            const float* w0 = wd + (2 * hh) * d + 2 * lane;
            const float* w1 = w0 + d;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float2 a = *reinterpret_cast<const float2*>(w0 + j * 128);
                const float2 e = *reinterpret_cast<const float2*>(w1 + j * 128);
                part0 = fmaf(c[j].x, a.x, fmaf(c[j].y, a.y, part0));
                part1 = fmaf(c[j].x, e.x, fmaf(c[j].y, e.y, part1));
            }
            const float d0 = wave_sum(part0) * rstd + bw[2 * hh];
            const float d1 = wave_sum(part1) * rstd + bw[2 * hh + 1];
            const float dl = upper ? d1 : d0;
            plab[hh] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-dl * 1.44269504088896340736f));
        }
        // x += p_noise v_n + p_label v_l ; then LN3
        float s3 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = j * 128 + 2 * lane;
            const float2 a = *reinterpret_cast<const float2*>(vn + n);
            const float2 dd = *reinterpret_cast<const float2*>(vdiff + n);
            v[j].x += fmaf(plab[j], dd.x, a.x);
            v[j].y += fmaf(plab[j], dd.y, a.y);
            *reinterpret_cast<float2*>(p.x + row * d + n) = v[j];
            s3 += v[j].x + v[j].y;
        }
        const float mean3 = wave_sum(s3) / (float)d;
        float q3 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            v[j].x -= mean3; v[j].y -= mean3;
            q3 += v[j].x * v[j].x + v[j].y * v[j].y;
        }
        const float rstd3 = 1.0f / sqrtf(wave_sum(q3) / (float)d + kLnEps);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = j * 128 + 2 * lane;
            const float2 gg = *reinterpret_cast<const float2*>(p.ln3_w + n);
            const float2 bb = *reinterpret_cast<const float2*>(p.ln3_b + n);
            bf16x2 o;
            o[0] = (bf16)(v[j].x * rstd3 * gg.x + bb.x);
            o[1] = (bf16)(v[j].y * rstd3 * gg.y + bb.y);
            *reinterpret_cast<bf16x2*>(p.xn3 + row * d + n) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out_proj + unpatchify.  16 tokens per wave, weights [pd][d] fp32 resident in LDS.
template <int NJ>
__global__ __launch_bounds__(256) void tail_kernel(TailParams p, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int d = NJ * 128;
    float* w = reinterpret_cast<float*>(smem);           // [pd][d]
    for (int i = threadIdx.x; i < p.pd * d / 4; i += 256)
        reinterpret_cast<float4*>(w)[i] = reinterpret_cast<const float4*>(p.w)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int rows_per_wave = rows_per_block / 4;
    const int total = p.batch * p.ntok;
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int row = blockIdx.x * rows_per_block + wid * rows_per_wave + rr;
        if (row >= total) return;
        float2 v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = *reinterpret_cast<const float2*>(p.tok + (size_t)row * d + j * 128 + 2 * lane);
        float mine = 0.f;
#pragma unroll 4
        for (int o = 0; o < p.pd; ++o) {
            float part = 0.f;
            const float* wrow = w + o * d + 2 * lane;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float2 ww = *reinterpret_cast<const float2*>(wrow + j * 128);
                part = fmaf(v[j].x, ww.x, fmaf(v[j].y, ww.y, part));
            }
            const float tot = wave_sum(part);
            if (lane == o) mine = tot + p.b[o];
        }
        if (lane < p.pd) {
            // feature f = (c, u, v) of token (ti, tj) -> out[b, c, ti*p+u, tj*p+v]
            const int b = row / p.ntok, t = row - b * p.ntok;
            const int ti = t / p.grid, tj = t - ti * p.grid;
            const int c = lane / (p.p * p.p), uv = lane - c * p.p * p.p;
            const int u = uv / p.p, vv = uv - u * p.p;
            p.out[(((size_t)b * p.C + c) * p.S + (ti * p.p + u)) * p.S + (tj * p.p + vv)] = mine;
        }
    }
}

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
// One thread owns 8 channels of one image row (b, i) and slides a 3x3 window along j: 3 new 16-B
// loads and one 16-B store per output, weights (72 + 8 floats) stay in registers.
__global__ __launch_bounds__(256) void dwconv_gelu_kernel(const bf16* __restrict__ in, bf16* __restrict__ out,
                                                          const float* __restrict__ w9c,
                                                          const float* __restrict__ bias, int batch,
                                                          int g, int C) {
    const int c8n = C / 8;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)batch * g * c8n;
    if (tid >= total) return;
    const int c8 = (int)(tid % c8n);
    const int bi = (int)(tid / c8n);
    const int b = bi / g, i = bi - b * g;
    const int c0 = c8 * 8;

    float w[9][8], bs[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float4 lo = *reinterpret_cast<const float4*>(w9c + (size_t)k * C + c0);
        const float4 hi = *reinterpret_cast<const float4*>(w9c + (size_t)k * C + c0 + 4);
        w[k][0] = lo.x; w[k][1] = lo.y; w[k][2] = lo.z; w[k][3] = lo.w;
        w[k][4] = hi.x; w[k][5] = hi.y; w[k][6] = hi.z; w[k][7] = hi.w;
    }
    {
        const float4 lo = *reinterpret_cast<const float4*>(bias + c0);
        const float4 hi = *reinterpret_cast<const float4*>(bias + c0 + 4);
        bs[0] = lo.x; bs[1] = lo.y; bs[2] = lo.z; bs[3] = lo.w;
        bs[4] = hi.x; bs[5] = hi.y; bs[6] = hi.z; bs[7] = hi.w;
    }
    const bf16* base = in + ((size_t)b * g * g) * C + c0;
    auto load_col = [&](int j, bf16x8 (&col)[3]) {
#pragma unroll
        for (int du = 0; du < 3; ++du) {
            const int ii = i + du - 1;
            if (ii >= 0 && ii < g && j >= 0 && j < g)
                col[du] = *reinterpret_cast<const bf16x8*>(base + ((size_t)ii * g + j) * C);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) col[du][e] = (bf16)0.f;
        }
    };
    bf16x8 win[3][3];     // win[dv][du]: column j-1+dv, row i-1+du
    load_col(-1, win[0]);
    load_col(0, win[1]);
    for (int j = 0; j < g; ++j) {
        load_col(j + 1, win[2]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float acc = bs[e];
#pragma unroll
            for (int du = 0; du < 3; ++du)
#pragma unroll
                for (int dv = 0; dv < 3; ++dv) acc += w[du * 3 + dv][e] * (float)win[dv][du][e];
            o[e] = (bf16)gelu_erf_fast(acc);
        }
        *reinterpret_cast<bf16x8*>(out + (((size_t)b * g + i) * g + j) * C + c0) = o;
#pragma unroll
        for (int du = 0; du < 3; ++du) { win[0][du] = win[1][du]; win[1][du] = win[2][du]; }
    }
}

}  // namespace

#define TLD_DISPATCH_NJ(nj, CALL)                                                                  \
    switch (nj) {                                                                                    \
        case 1: { constexpr int NJ = 1; CALL; } break;                                              \
        case 2: { constexpr int NJ = 2; CALL; } break;                                              \
        case 3: { constexpr int NJ = 3; CALL; } break;                                              \
        case 4: { constexpr int NJ = 4; CALL; } break;                                              \
        case 5: { constexpr int NJ = 5; CALL; } break;                                              \
        case 6: { constexpr int NJ = 6; CALL; } break;                                              \
        case 7: { constexpr int NJ = 7; CALL; } break;                                              \
        case 8: { constexpr int NJ = 8; CALL; } break;                                              \
        default: break;                                                                              \
    }

void launch_embed(const EmbedParams& p, hipStream_t s) {
    const int rows = p.batch * p.ntok;
    TLD_DISPATCH_NJ(p.d / 128, hipLaunchKernelGGL(embed_kernel<NJ>, dim3((rows + 3) / 4), dim3(256), 0, s, p));
}

void launch_layernorm_bf16(const resid_t* x, const float* g, const float* b, bf16* out, int M, int d,
                           hipStream_t s) {
    TLD_DISPATCH_NJ(d / 128, hipLaunchKernelGGL(layernorm_bf16_kernel<NJ>, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, out, M, d));
}

void launch_cross_row(const CrossRowParams& p, hipStream_t s) {
    const int rpb = (p.ntok % 64 == 0) ? 64 : 32;
    const int lds = (p.heads * p.d + 2 * p.d + p.heads) * (int)sizeof(float);
    dim3 grid(p.batch * (p.ntok / rpb));
    TLD_DISPATCH_NJ(p.d / 128, hipLaunchKernelGGL(cross_row_kernel<NJ>, grid, dim3(256), lds, s, p, rpb));
}

void launch_tail(const TailParams& p, hipStream_t s) {
    const int rpb = 64;
    const int rows = p.batch * p.ntok;
    const int lds = p.pd * p.d * (int)sizeof(float);
    TLD_DISPATCH_NJ(p.d / 128, hipLaunchKernelGGL(tail_kernel<NJ>, dim3((rows + rpb - 1) / rpb), dim3(256), lds, s, p, rpb));
}

void launch_update(const UpdateParams& p, hipStream_t s) {
    const int n = p.batch * p.img;
    hipLaunchKernelGGL(update_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p);
}

void launch_dwconv_gelu(const bf16* in, bf16* out, const float* w9c, const float* bias, int batch, int grid,
                        int channels, hipStream_t s) {
    const size_t total = (size_t)batch * grid * (channels / 8);
    hipLaunchKernelGGL(dwconv_gelu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out,
                       w9c, bias, batch, grid, channels);
}

}  // namespace tld
