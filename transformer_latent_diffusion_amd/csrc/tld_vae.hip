// tld_vae.hip -- VAE decode of the final latents on gfx950.
//
// Replaces `self.vae.decode(latents)[0]` of the reference's sampler (tld/diffusion.py:91), where `vae` is diffusers'
// AutoencoderKL "madebyollin/sdxl-vae-fp16-fix" (tld/configs.py:39-43) -- a third-party model that is not part of the
// reference checkout.  What is restated here is AutoencoderKL.decode of diffusers 0.2x (models/autoencoders/vae.py
// Decoder, models/resnet.py ResnetBlock2D, models/attention_processor.py Attention, models/upsampling.py Upsample2D):
//   post_quant_conv 1x1 -> conv_in 3x3 -> mid block (resnet, single-head attention over h*w tokens, resnet)
//   -> up blocks (layers_per_block + 1 resnets, nearest-2x + 3x3 conv except in the last) -> GroupNorm, SiLU, conv_out 3x3.
//
// Device layout: activations are bf16 channels-last [B, H, W, C]; every activation buffer starts with a 2-KiB zero
// page that the implicit-GEMM convolution reads for taps outside the image (GemmParams::conv, tld_gemm.hip).  The 3x3
// convolutions (99 % of the FLOPs) are that persistent MFMA GEMM with K = 9 C_in and tap-dependent DMA row addresses; the
// nearest-neighbour upsampling is folded into the same addressing (the 4x larger image is never written); 1x1 shortcuts,
// the attention projections and both attention matmuls (all samples per launch, block-diagonal W) are the plain GEMM; GroupNorm statistics are a deterministic
// two-stage fp32 / fp64 reduction and its affine + SiLU one elementwise pass.
#include "../../include/tld_hip.h"
#include "tld_common.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace tld;

namespace {

constexpr unsigned kHdr = 2048;       // zero page in front of every activation buffer (bytes): one pixel of up to 1024 channels
constexpr float kGnEps = 1e-6f;       // AutoencoderKL: every GroupNorm is built with eps = 1e-6

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    set_last_error(buf);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(TLD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
        else if (prev < 0) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// ---- kernels -----------------------------------------------------------------------------------------------------------

// post_quant_conv (1x1, optional) + conv_in (3x3, zero pad) on the fp32 NCHW latent: one workgroup per output pixel,
// one thread per output channel.  K = 9 * zc = 36: not a GEMM worth the name.
//   wt [9 * zc][C0] (tap-major, channel, then output channel: coalesced over threads)
__global__ void vae_conv_in_kernel(const float* __restrict__ z, int zc, int h, int w, const float* __restrict__ pq_w,
                                   const float* __restrict__ pq_b, const float* __restrict__ wt, const float* __restrict__ bias,
                                   bf16* __restrict__ out, int C0) {
    __shared__ float v[9 * 16];
    const int pix = blockIdx.x;
    const int hw = h * w;
    const int b = pix / hw, rem = pix - b * hw, y = rem / w, x = rem - y * w;
    const int t = threadIdx.x;
    if (t < 9 * zc) {
        const int tap = t / zc, c = t - tap * zc;
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        float a = 0.f;
        if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
            const float* zp = z + (size_t)b * zc * hw + (size_t)yy * w + xx;
            if (pq_w) {
                a = pq_b[c];
                for (int c2 = 0; c2 < zc; ++c2) a = fmaf(pq_w[c * zc + c2], zp[(size_t)c2 * hw], a);
            } else {
                a = zp[(size_t)c * hw];
            }
        }
        v[t] = a;
    }
    __syncthreads();
    if (t < C0) {
        float acc = bias[t];
        for (int k = 0; k < 9 * zc; ++k) acc = fmaf(v[k], wt[(size_t)k * C0 + t], acc);
        out[(size_t)pix * C0 + t] = (bf16)acc;
    }
}

// GroupNorm statistics, stage 1: one workgroup per (pixel chunk, sample) sums x and x^2 per group over its pixels.
// A thread owns 8 consecutive channels (one 16-byte load per pixel); C / 8 threads cover a pixel, 256 / (C / 8) pixels
// per iteration.  Fixed reduction order: results do not depend on scheduling.
__global__ __launch_bounds__(256) void vae_gn_stats_kernel(const bf16* __restrict__ x, int HW, int C, int G, int ppb,
                                                           float2* __restrict__ partial) {
    extern __shared__ float2 red[];                // [256 / TPP][C] then [C]
    const int chunk = blockIdx.x, nchunk = gridDim.x, b = blockIdx.y;
    const int TPP = C >> 3, PPI = 256 / TPP;
    const int tp = threadIdx.x % TPP, slot = threadIdx.x / TPP;
    const int p0 = chunk * ppb, p1 = min(p0 + ppb, HW);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    const bf16* xb = x + (size_t)b * HW * C + tp * 8;
    for (int pix = p0 + slot; pix < p1; pix += PPI) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(xb + (size_t)pix * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; q[e] = fmaf(f, f, q[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[slot * C + tp * 8 + e] = make_float2(s[e], q[e]);
    __syncthreads();
    float2* ch = red + PPI * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, a2 = 0.f;
        for (int sl = 0; sl < PPI; ++sl) { const float2 t = red[sl * C + c]; a += t.x; a2 += t.y; }
        ch[c] = make_float2(a, a2);
    }
    __syncthreads();
    if (threadIdx.x < G) {
        const int cpg = C / G;
        float a = 0.f, a2 = 0.f;
        for (int c = 0; c < cpg; ++c) { const float2 t = ch[threadIdx.x * cpg + c]; a += t.x; a2 += t.y; }
        partial[((size_t)b * nchunk + chunk) * G + threadIdx.x] = make_float2(a, a2);
    }
}
// stage 2: (mean, rstd) per (sample, group); the chunk partials are combined in fp64 (E[x^2] - mean^2 loses nothing there)
// (eight threads per group, each summing every eighth chunk; the order is fixed, so the result is reproducible)
__global__ void vae_gn_finalize_kernel(const float2* __restrict__ partial, int nchunk, int G, double inv_n, float eps,
                                       float2* __restrict__ stats) {
    const int b = blockIdx.x, g = threadIdx.x >> 3, part = threadIdx.x & 7;
    if (g >= G) return;
    double a = 0.0, a2 = 0.0;
    for (int c = part; c < nchunk; c += 8) { const float2 t = partial[((size_t)b * nchunk + c) * G + g]; a += (double)t.x; a2 += (double)t.y; }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o); a2 += __shfl_xor(a2, o); }
    if (part) return;
    const double mean = a * inv_n;
    double var = a2 * inv_n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    stats[(size_t)b * G + g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
}
// affine + optional SiLU: y = act((x - mean) rstd gamma + beta), bf16 in / out, same thread mapping as the statistics
template <bool SILU>
__global__ __launch_bounds__(256) void vae_gn_apply_kernel(const bf16* __restrict__ x, const float2* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16* __restrict__ out, int HW, int C, int G, int ppb) {
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int TPP = C >> 3, PPI = 256 / TPP;
    const int tp = threadIdx.x % TPP, slot = threadIdx.x / TPP;
    const int cpg = C / G;
    float a[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = tp * 8 + e;
        const float2 st = stats[(size_t)b * G + c / cpg];
        a[e] = st.y * gamma[c];
        sh[e] = fmaf(-st.x, a[e], beta[c]);
    }
    const int p0 = chunk * ppb, p1 = min(p0 + ppb, HW);
    const size_t base = (size_t)b * HW * C + tp * 8;
    for (int pix = p0 + slot; pix < p1; pix += PPI) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + base + (size_t)pix * C);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = fmaf((float)v[e], a[e], sh[e]);
            if (SILU) f = f * __builtin_amdgcn_rcpf(1.0f + __expf(-f));
            o[e] = (bf16)f;
        }
        *reinterpret_cast<bf16x8*>(out + base + (size_t)pix * C) = o;
    }
}

// attention probabilities of one sample: P[i, :] = softmax(scale * S[i, :]), fp32 in, bf16 out; one workgroup per row
__global__ __launch_bounds__(256) void vae_softmax_rows_kernel(const float* __restrict__ S, bf16* __restrict__ P, int n, float scale) {
    __shared__ float sh[8];
    const float* row = S + (size_t)blockIdx.x * n;
    bf16* prow = P + (size_t)blockIdx.x * n;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < n; j += 256) m = fmaxf(m, row[j]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) sh[wid] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    float sum = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) sum += __expf((row[j] - m) * scale);
    sum = wave_sum(sum);
    if (lane == 0) sh[4 + wid] = sum;
    __syncthreads();
    const float inv = 1.0f / ((sh[4] + sh[5]) + (sh[6] + sh[7]));
    for (int j = threadIdx.x; j < n; j += 256) prow[j] = (bf16)(__expf((row[j] - m) * scale) * inv);
}

// out[c][r] = in[r][c]  (bf16; 32 x 32 tiles): V of one sample [tokens, C] (row pitch ld_in) -> V^T [C, tokens]
//   (blockIdx.z = sample: in / out advance by in_bstride / out_bstride elements)
__global__ void vae_transpose_kernel(const bf16* __restrict__ in, int ld_in, bf16* __restrict__ out, int ld_out, int rows, int cols,
                                     size_t in_bstride, size_t out_bstride) {
    __shared__ bf16 tile[32][33];
    in += blockIdx.z * in_bstride; out += blockIdx.z * out_bstride;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : (bf16)0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (c < cols && r < rows) out[(size_t)c * ld_out + r] = tile[threadIdx.x][i];
    }
}

// conv_out tail: fp32 [B*HW][OC] (the GEMM's fp32 epilogue) + bias -> fp32 NCHW [B, OC, H, W]
__global__ void vae_out_kernel(const float* __restrict__ in, const float* __restrict__ bias, float* __restrict__ out, int HW, int OC, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over B * HW pixels
    if (i >= total) return;
    const long b = i / HW, pix = i - b * HW;
    for (int c = 0; c < OC; ++c) out[((size_t)b * OC + c) * HW + pix] = in[(size_t)i * OC + c] + bias[c];
}

__global__ void vae_cast_in_kernel(const void* __restrict__ src, int dtype, float* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (dtype == TLD_DTYPE_F32) dst[i] = reinterpret_cast<const float*>(src)[i];
    else if (dtype == TLD_DTYPE_BF16) dst[i] = (float)reinterpret_cast<const bf16*>(src)[i];
    else dst[i] = (float)reinterpret_cast<const _Float16*>(src)[i];
}

// ---- host structures ---------------------------------------------------------------------------------------------------

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };

struct ConvW { bf16* w = nullptr; float* b = nullptr; int cin = 0, cout = 0, k = 0; };     // [cout][k*k*cin] (tap-major), fp32 bias
struct GnW { float* g = nullptr; float* b = nullptr; int c = 0; };
struct Resnet { GnW n1, n2; ConvW c1, c2, sc; bool has_sc = false; int cin = 0, cout = 0; };
struct UpBlock { std::vector<Resnet> res; bool has_up = false; ConvW up; int cout = 0; };

enum VClass { VC_CONV = 0, VC_GEMM, VC_GN, VC_OTHER, VC_COUNT };

struct Stage { std::string name; bf16* dev = nullptr; int B = 0, C = 0, H = 0, W = 0; };

}  // namespace

struct tld_vae {
    tld_vae_config cfg{};
    int G = 32, zc = 4, oc = 3, nb = 0, hl = 0;
    std::vector<int> boc;                 // block_out_channels (encoder order)
    bool finalized = false;
    std::map<std::string, HostTensor> host;
    std::vector<void*> allocs;
    int64_t weight_bytes = 0;

    // weights
    float *pq_w = nullptr, *pq_b = nullptr, *cin_wt = nullptr, *cin_b = nullptr, *cout_b = nullptr, *zero_bias = nullptr;
    int C0 = 0;
    Resnet mid0, mid1;
    GnW attn_gn;
    ConvW attn_qkv, attn_out;             // [3C][C] (q | k | v rows) and [C][C]
    std::vector<UpBlock> ups;
    GnW norm_out;
    ConvW conv_out;

    // workspace: four activation buffers (each with a zero page in front) + small ones
    char* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t buf_elems = 0;                 // bf16 elements of data per buffer
    float* io_z = nullptr;
    float2 *gn_partial = nullptr, *gn_stats = nullptr;
    int gn_max_chunks = 0;
    // mid-block attention scratch for att_nb samples at a time: scores fp32 [att_nb][HW][HW], probabilities bf16 (same shape),
    // V^T bf16 [att_nb][C][HW].  One batched launch per step (GemmParams::w_batch_rows): a sample alone is 16 tiles at 256 px.
    int att_nb = 1;
    float* scores = nullptr; bf16* probs = nullptr; bf16* vt = nullptr;
    float* out_f32 = nullptr;

    // GroupNorm statistics fused into the producing convolution's epilogue: true while gn_partial describes the tensor the
    // next group_norm() normalises (set by conv3x3, consumed / invalidated by group_norm and by anything else that writes x)
    bool have_partial = false;
    bool fuse_stats = true;               // GroupNorm statistics in the producing conv epilogue wherever the shapes allow
    bool debug = false;
    std::vector<Stage> stages;
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[VC_COUNT];
    size_t ev_used[VC_COUNT] = {0, 0, 0, 0};

    bf16* data(int i) const { return reinterpret_cast<bf16*>(buf[i] + kHdr); }
};

namespace {

template <typename T>
int dev_alloc(tld_vae* v, T** out, size_t count, bool weight = false) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
    v->allocs.push_back(p);
    if (weight) v->weight_bytes += (int64_t)(count * sizeof(T));
    *out = reinterpret_cast<T*>(p);
    return TLD_OK;
}

int upload_f32(tld_vae* v, const std::vector<float>& h, float** out) {
    if (int rc = dev_alloc(v, out, h.size(), true)) return rc;
    HIP_TRY(hipMemcpy(*out, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return TLD_OK;
}
int upload_bf16(tld_vae* v, const std::vector<float>& h, bf16** out) {
    std::vector<uint16_t> t(h.size());
    for (size_t i = 0; i < h.size(); ++i) t[i] = f32_to_bf16_rne(h[i]);
    if (int rc = dev_alloc(v, out, h.size(), true)) return rc;
    HIP_TRY(hipMemcpy(*out, t.data(), t.size() * 2, hipMemcpyHostToDevice));
    return TLD_OK;
}

const HostTensor* find(const tld_vae* v, const std::string& key) {
    auto it = v->host.find(key);
    return it == v->host.end() ? nullptr : &it->second;
}

int need(const tld_vae* v, const std::string& key, std::initializer_list<int64_t> shape, const HostTensor** out) {
    const HostTensor* t = find(v, key);
    if (!t) return fail(TLD_ERR_STATE, "missing state_dict entry '%s'", key.c_str());
    int64_t n = 1, m = 1;
    for (int64_t s : shape) n *= s;
    for (int64_t s : t->shape) m *= s;
    // trailing singleton dimensions may differ (Linear [C, C] vs 1x1 conv [C, C, 1, 1] in older checkpoints)
    std::vector<int64_t> a(shape), b2(t->shape);
    while (!a.empty() && a.back() == 1) a.pop_back();
    while (!b2.empty() && b2.back() == 1) b2.pop_back();
    if (n != m || a != b2) {
        std::string got;
        for (int64_t s : t->shape) got += std::to_string(s) + ",";
        std::string want;
        for (int64_t s : shape) want += std::to_string(s) + ",";
        return fail(TLD_ERR_SHAPE, "'%s' has shape [%s], expected [%s]", key.c_str(), got.c_str(), want.c_str());
    }
    *out = t;
    return TLD_OK;
}

// conv weight [cout][cin][k][k] -> bf16 [cout][k*k][cin]
int pack_conv(tld_vae* v, const std::string& prefix, int cin, int cout, int k, ConvW* cw) {
    const HostTensor *w = nullptr, *b = nullptr;
    if (int rc = need(v, prefix + ".weight", {cout, cin, k, k}, &w)) return rc;
    if (int rc = need(v, prefix + ".bias", {cout}, &b)) return rc;
    std::vector<float> t((size_t)cout * k * k * cin);
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
            for (int tap = 0; tap < k * k; ++tap)
                t[((size_t)o * k * k + tap) * cin + c] = w->data[((size_t)o * cin + c) * k * k + tap];
    cw->cin = cin; cw->cout = cout; cw->k = k;
    if (int rc = upload_bf16(v, t, &cw->w)) return rc;
    return upload_f32(v, b->data, &cw->b);
}
int pack_gn(tld_vae* v, const std::string& prefix, int c, GnW* g) {
    const HostTensor *w = nullptr, *b = nullptr;
    if (int rc = need(v, prefix + ".weight", {c}, &w)) return rc;
    if (int rc = need(v, prefix + ".bias", {c}, &b)) return rc;
    g->c = c;
    if (int rc = upload_f32(v, w->data, &g->g)) return rc;
    return upload_f32(v, b->data, &g->b);
}
int pack_resnet(tld_vae* v, const std::string& prefix, int cin, int cout, Resnet* r) {
    r->cin = cin; r->cout = cout; r->has_sc = cin != cout;
    if (int rc = pack_gn(v, prefix + ".norm1", cin, &r->n1)) return rc;
    if (int rc = pack_conv(v, prefix + ".conv1", cin, cout, 3, &r->c1)) return rc;
    if (int rc = pack_gn(v, prefix + ".norm2", cout, &r->n2)) return rc;
    if (int rc = pack_conv(v, prefix + ".conv2", cout, cout, 3, &r->c2)) return rc;
    if (r->has_sc) return pack_conv(v, prefix + ".conv_shortcut", cin, cout, 1, &r->sc);
    return TLD_OK;
}

struct Timer {
    tld_vae* v; int kc; hipStream_t s; bool on = false; size_t idx = 0;
    Timer(tld_vae* v_, int kc_, hipStream_t s_) : v(v_), kc(kc_), s(s_) {
        if (!v->profile) return;
        if (v->ev_used[kc] == v->ev[kc].size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            v->ev[kc].push_back({a, b});
        }
        idx = v->ev_used[kc]++;
        on = true;
        (void)hipEventRecord(v->ev[kc][idx].first, s);
    }
    ~Timer() { if (on) (void)hipEventRecord(v->ev[kc][idx].second, s); }
};

bool chan_ok(int c) { return c == 64 || c == 128 || c == 256 || c == 512 || c == 1024; }

// ---- op sequencing ---------------------------------------------------------------------------------------------------

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TLD_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return TLD_OK;
}

// GroupNorm (+ SiLU) of the [B, HW, C] image in buffer `src` into buffer `dst`
int group_norm(tld_vae* v, int src, int dst, const GnW& gn, int B, int HW, int C, bool silu, hipStream_t s) {
    Timer t(v, VC_GN, s);
    const int ppb = 256;
    const int nchunk = (HW + ppb - 1) / ppb;
    const int TPP = C / 8, PPI = 256 / TPP;
    const size_t lds = (size_t)(PPI * C + C) * sizeof(float2);
    if (!v->have_partial)
        hipLaunchKernelGGL(vae_gn_stats_kernel, dim3(nchunk, B), dim3(256), lds, s, v->data(src), HW, C, v->G, ppb, v->gn_partial);
    v->have_partial = false;
    hipLaunchKernelGGL(vae_gn_finalize_kernel, dim3(B), dim3(v->G * 8), 0, s, v->gn_partial, nchunk, v->G,
                       1.0 / ((double)HW * (C / v->G)), kGnEps, v->gn_stats);
    if (silu) hipLaunchKernelGGL(vae_gn_apply_kernel<true>, dim3(nchunk, B), dim3(256), 0, s, v->data(src), v->gn_stats, gn.g, gn.b, v->data(dst), HW, C, v->G, ppb);
    else hipLaunchKernelGGL(vae_gn_apply_kernel<false>, dim3(nchunk, B), dim3(256), 0, s, v->data(src), v->gn_stats, gn.g, gn.b, v->data(dst), HW, C, v->G, ppb);
    return check_launch("group_norm");
}

// 3x3 convolution of buffer `src` ([B, H >> up, W >> up, cin]) into an [B, H, W, cout] image
//   epi EPI_BIAS_BF16: written to buffer dst;  EPI_BIAS_RESID: added to buffer dst in place;  EPI_F32: fp32 [M][cout] to c_f32
int conv3x3(tld_vae* v, int src, int dst, const ConvW& cw, int B, int H, int W, int up, int epi, float* c_f32, hipStream_t s) {
    Timer t(v, VC_CONV, s);
    GemmParams p{};
    p.A = reinterpret_cast<const bf16*>(v->buf[src]);
    p.conv = 1; p.cv_h = H; p.cv_w = W; p.cv_up = up; p.cv_cin = cw.cin; p.cv_data_off = kHdr;
    p.lda = cw.cin;
    p.W = cw.w; p.ldw = 9 * cw.cin;
    p.M = B * H * W; p.N = cw.cout; p.K = 9 * cw.cin;
    p.bias = cw.b;
    if (epi == EPI_BIAS_BF16) { p.out_bf16 = v->data(dst); p.ldo = cw.cout; }
    else if (epi == EPI_BIAS_RESID) { p.resid = reinterpret_cast<resid_t*>(v->data(dst)); p.ldr = cw.cout; }
    else { p.c_f32 = c_f32; p.ldc = cw.cout; }
    // the consumer of a bf16 conv output is always a GroupNorm: leave its per-chunk statistics behind when the shapes allow
    // (a 256-row tile = one 256-pixel chunk of one sample; whole column tiles; groups made of whole 4-column quads)
    const int cpg = cw.cout / v->G;
    v->have_partial = false;
    if (v->fuse_stats && epi != EPI_F32 && (H * W) % 256 == 0 && cw.cout % 128 == 0 && cpg % 4 == 0 && cw.cout % cpg == 0) {
        p.gn_partial = v->gn_partial; p.gn_groups = v->G; p.gn_cpg = cpg; p.gn_hw = H * W;
        v->have_partial = true;
    }
    launch_gemm(p, epi, s);
    return check_launch("conv3x3");
}

// plain GEMM C[M,N] = A[M,K] W[N,K]^T with one of the three epilogues
int gemm(tld_vae* v, const bf16* A, int lda, const bf16* Wt, int ldw, int M, int N, int K, int epi, const float* bias, bf16* out, int ldo,
         float* c_f32, hipStream_t s, int w_batch_rows = 0, size_t w_batch_stride_bytes = 0) {
    Timer t(v, VC_GEMM, s);
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = Wt; p.ldw = ldw; p.M = M; p.N = N; p.K = K; p.bias = bias;
    p.w_batch_rows = w_batch_rows; p.w_batch_stride_bytes = (unsigned)w_batch_stride_bytes;
    if (epi == EPI_BIAS_BF16) { p.out_bf16 = out; p.ldo = ldo; }
    else if (epi == EPI_BIAS_RESID) { p.resid = reinterpret_cast<resid_t*>(out); p.ldr = ldo; }
    else { p.c_f32 = c_f32; p.ldc = ldo; }
    launch_gemm(p, epi, s);
    return check_launch("gemm");
}

int snapshot(tld_vae* v, const char* name, int src, int B, int H, int W, int C, hipStream_t s) {
    if (!v->debug) return TLD_OK;
    Stage st;
    st.name = name; st.B = B; st.C = C; st.H = H; st.W = W;
    const size_t n = (size_t)B * H * W * C;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&st.dev), n * 2));
    HIP_TRY(hipMemcpyAsync(st.dev, v->data(src), n * 2, hipMemcpyDeviceToDevice, s));
    v->stages.push_back(st);
    return TLD_OK;
}

void clear_stages(tld_vae* v) {
    for (auto& st : v->stages) (void)hipFree(st.dev);
    v->stages.clear();
}

// ResnetBlock2D (diffusers models/resnet.py; temb is None in the VAE, output_scale_factor 1):
//   h = conv1(silu(norm1(x)));  h = conv2(silu(norm2(h)));  return shortcut(x) + h
// x lives in buffer *xi; on return *xi names the buffer holding the result.  bufs: x, t (normalised), h, s (shortcut)
int resnet(tld_vae* v, const Resnet& r, int* xi, int B, int H, int W, hipStream_t s) {
    int idx[4], n = 0;
    for (int i = 0; i < 4; ++i) if (i != *xi) idx[n++] = i;
    const int x = *xi, t = idx[0], h = idx[1], sc = idx[2];
    const int HW = H * W;
    if (int rc = group_norm(v, x, t, r.n1, B, HW, r.cin, true, s)) return rc;
    if (int rc = conv3x3(v, t, h, r.c1, B, H, W, 0, EPI_BIAS_BF16, nullptr, s)) return rc;
    if (int rc = group_norm(v, h, t, r.n2, B, HW, r.cout, true, s)) return rc;
    if (r.has_sc) {
        if (int rc = gemm(v, v->data(x), r.cin, r.sc.w, r.cin, B * HW, r.cout, r.cin, EPI_BIAS_BF16, r.sc.b, v->data(sc), r.cout, nullptr, s)) return rc;
        if (int rc = conv3x3(v, t, sc, r.c2, B, H, W, 0, EPI_BIAS_RESID, nullptr, s)) return rc;
        *xi = sc;
    } else {
        if (int rc = conv3x3(v, t, x, r.c2, B, H, W, 0, EPI_BIAS_RESID, nullptr, s)) return rc;
    }
    return TLD_OK;
}

// Attention block of the mid block (diffusers Attention with heads = 1, dim_head = C, residual_connection, bias):
//   t = group_norm(x) as [B, HW, C] tokens;  q, k, v = linear(t);  o = softmax(q k^T / sqrt(C)) v;  x += to_out(o)
int attention(tld_vae* v, int* xi, int B, int H, int W, int C, hipStream_t s) {
    int idx[4], n = 0;
    for (int i = 0; i < 4; ++i) if (i != *xi) idx[n++] = i;
    const int x = *xi, t = idx[0], qkv = idx[1];
    const int HW = H * W, M = B * HW;
    if (int rc = group_norm(v, x, t, v->attn_gn, B, HW, C, false, s)) return rc;
    if (int rc = gemm(v, v->data(t), C, v->attn_qkv.w, C, M, 3 * C, C, EPI_BIAS_BF16, v->attn_qkv.b, v->data(qkv), 3 * C, nullptr, s)) return rc;
    const float scale = 1.0f / sqrtf((float)C);
    // groups of `step` samples per launch: all of att_nb when a 256-row tile cannot straddle two samples, else one by one
    const int step = HW % 256 == 0 ? v->att_nb : 1;
    for (int b0 = 0; b0 < B; b0 += step) {
        const int n = std::min(step, B - b0);
        const bf16* q = v->data(qkv) + (size_t)b0 * HW * 3 * C;
        // scores_b = Q_b K_b^T: Q of the group as one tall [n * HW, C] matrix, K_b picked per tile-row
        if (int rc = gemm(v, q, 3 * C, q + C, 3 * C, n * HW, HW, C, EPI_F32, nullptr, nullptr, HW, v->scores, s, n > 1 ? HW : 0, (size_t)HW * 3 * C * 2)) return rc;
        {
            Timer tm(v, VC_OTHER, s);
            hipLaunchKernelGGL(vae_softmax_rows_kernel, dim3(n * HW), dim3(256), 0, s, v->scores, v->probs, HW, scale);
            hipLaunchKernelGGL(vae_transpose_kernel, dim3((C + 31) / 32, (HW + 31) / 32, n), dim3(32, 8), 0, s, q + 2 * C, 3 * C, v->vt, HW, HW, C,
                               (size_t)HW * 3 * C, (size_t)C * HW);
        }
        // O_b = P_b V_b  (into the tokens of buffer t, which the projections no longer need)
        if (int rc = gemm(v, v->probs, HW, v->vt, HW, n * HW, C, HW, EPI_BIAS_BF16, v->zero_bias, v->data(t) + (size_t)b0 * HW * C, C, nullptr, s,
                          n > 1 ? HW : 0, (size_t)C * HW * 2)) return rc;
    }
    if (int rc = gemm(v, v->data(t), C, v->attn_out.w, C, M, C, C, EPI_BIAS_RESID, v->attn_out.b, v->data(x), C, nullptr, s)) return rc;
    v->have_partial = false;               // x changed: the statistics a convolution left behind are stale
    return check_launch("attention");
}

size_t max_act_elems(const tld_vae* v) {                    // per sample, over all stages of the decoder
    size_t mx = 0;
    int H = v->hl;
    int c = v->boc[v->nb - 1];
    mx = std::max(mx, (size_t)H * H * c * 3);               // attention q|k|v
    for (int i = 0; i < v->nb; ++i) {
        const int cout = v->boc[v->nb - 1 - i];
        mx = std::max(mx, (size_t)H * H * std::max(c, cout));
        c = cout;
        if (i != v->nb - 1) { H *= 2; mx = std::max(mx, (size_t)H * H * c); }
    }
    return mx;
}

}  // namespace

// ---- C ABI -------------------------------------------------------------------------------------------------------------

extern "C" {

int tld_vae_create(const tld_vae_config* cfg, tld_vae** out) {
    if (!cfg || !out) return fail(TLD_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->n_blocks < 1 || cfg->n_blocks > 4) return fail(TLD_ERR_INVALID, "n_blocks=%d: 1..4 supported", cfg->n_blocks);
    for (int i = 0; i < cfg->n_blocks; ++i)
        if (!chan_ok(cfg->block_out_channels[i]))
            return fail(TLD_ERR_INVALID, "block_out_channels[%d]=%d: must be one of 64, 128, 256, 512, 1024", i, cfg->block_out_channels[i]);
    if (cfg->latent_channels < 1 || cfg->latent_channels > 16) return fail(TLD_ERR_INVALID, "latent_channels=%d: 1..16 supported", cfg->latent_channels);
    if (cfg->out_channels < 1 || cfg->out_channels > 8) return fail(TLD_ERR_INVALID, "out_channels=%d: 1..8 supported", cfg->out_channels);
    if (cfg->norm_num_groups < 1 || cfg->norm_num_groups > 64) return fail(TLD_ERR_INVALID, "norm_num_groups=%d: 1..64 supported", cfg->norm_num_groups);
    for (int i = 0; i < cfg->n_blocks; ++i)
        if (cfg->block_out_channels[i] % cfg->norm_num_groups)
            return fail(TLD_ERR_INVALID, "block_out_channels[%d]=%d is not a multiple of norm_num_groups=%d", i, cfg->block_out_channels[i], cfg->norm_num_groups);
    if (cfg->layers_per_block < 1 || cfg->layers_per_block > 8) return fail(TLD_ERR_INVALID, "layers_per_block=%d: 1..8 supported", cfg->layers_per_block);
    if (cfg->latent_size < 4 || cfg->latent_size > 256) return fail(TLD_ERR_INVALID, "latent_size=%d: 4..256 supported", cfg->latent_size);
    if (cfg->mid_block_attention && cfg->latent_size % 8)
        return fail(TLD_ERR_INVALID, "latent_size=%d: the mid-block attention needs a multiple of 8 (h*w tokens in 64-wide K-steps)", cfg->latent_size);
    if (cfg->max_batch < 1) return fail(TLD_ERR_INVALID, "max_batch must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TLD_ERR_HIP, "no HIP device available (the VAE decoder has no CPU path)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(TLD_ERR_INVALID, "device_id=%d out of range (%d devices)", cfg->device_id, ndev);
    DeviceGuard guard(cfg->device_id);

    tld_vae* v = new tld_vae();
    v->cfg = *cfg;
    v->G = cfg->norm_num_groups; v->zc = cfg->latent_channels; v->oc = cfg->out_channels; v->nb = cfg->n_blocks; v->hl = cfg->latent_size;
    v->boc.assign(cfg->block_out_channels, cfg->block_out_channels + cfg->n_blocks);
    v->C0 = v->boc[v->nb - 1];
    v->fuse_stats = true;
    v->buf_elems = max_act_elems(v) * (size_t)cfg->max_batch;
    const size_t bytes = v->buf_elems * 2 + kHdr;
    if (bytes >= (1ull << 32)) {
        const size_t per = max_act_elems(v) * 2;
        delete v;
        return fail(TLD_ERR_INVALID, "max_batch=%d: an activation buffer (%zu bytes per sample) must stay below 4 GiB (32-bit DMA offsets); "
                    "decode in chunks of at most %zu", cfg->max_batch, per, (size_t)(((1ull << 32) - kHdr - 1) / per));
    }
    auto bail = [&](int rc) { tld_vae_destroy(v); return rc; };
    for (int i = 0; i < 4; ++i) {
        if (int rc = dev_alloc(v, &v->buf[i], bytes)) return bail(rc);
        if (hipMemset(v->buf[i], 0, kHdr) != hipSuccess) return bail(fail(TLD_ERR_HIP, "hipMemset failed"));
    }
    const int Hout = v->hl << (v->nb - 1);
    const size_t lat = (size_t)cfg->max_batch * v->zc * v->hl * v->hl;
    if (int rc = dev_alloc(v, &v->io_z, lat)) return bail(rc);
    v->gn_max_chunks = (Hout * Hout + 255) / 256;
    if (int rc = dev_alloc(v, &v->gn_partial, (size_t)cfg->max_batch * v->gn_max_chunks * v->G)) return bail(rc);
    if (int rc = dev_alloc(v, &v->gn_stats, (size_t)cfg->max_batch * v->G)) return bail(rc);
    if (cfg->mid_block_attention) {
        const size_t hw = (size_t)v->hl * v->hl;
        const size_t fit = ((size_t)256 << 20) / (hw * hw * 4);             // samples whose fp32 scores fit in 256 MB
        v->att_nb = (int)std::max<size_t>(1, std::min<size_t>(fit, (size_t)cfg->max_batch));
        if (int rc = dev_alloc(v, &v->scores, (size_t)v->att_nb * hw * hw)) return bail(rc);
        if (int rc = dev_alloc(v, &v->probs, (size_t)v->att_nb * hw * hw)) return bail(rc);
        if (int rc = dev_alloc(v, &v->vt, (size_t)v->att_nb * hw * v->C0)) return bail(rc);
    }
    if (int rc = dev_alloc(v, &v->out_f32, (size_t)cfg->max_batch * Hout * Hout * v->oc)) return bail(rc);
    *out = v;
    return TLD_OK;
}

int tld_vae_load_tensor(tld_vae* v, const char* key, const void* host_ptr, const int64_t* shape, int32_t ndim, int32_t dtype) {
    if (!v || !key || (!host_ptr && ndim > 0) || ndim < 0 || ndim > 8) return fail(TLD_ERR_INVALID, "bad argument");
    if (v->finalized) return fail(TLD_ERR_STATE, "weights already finalized");
    std::string k(key);
    if (k.rfind("encoder.", 0) == 0 || k.rfind("quant_conv.", 0) == 0) return TLD_OK;      // the encoder half is not used by decode
    if (k.rfind("decoder.", 0) != 0 && k.rfind("post_quant_conv.", 0) != 0) return fail(TLD_ERR_KEY, "unknown state_dict key '%s'", key);
    if (dtype != TLD_DTYPE_F32) return fail(TLD_ERR_INVALID, "'%s': host tensors must be fp32", key);
    // pre-0.19 diffusers spelling of the attention block
    static const char* const ren[][2] = {{".query.", ".to_q."}, {".key.", ".to_k."}, {".value.", ".to_v."}, {".proj_attn.", ".to_out.0."}};
    for (auto& r : ren) {
        const size_t pos = k.find(r[0]);
        if (pos != std::string::npos && k.find(".attentions.") != std::string::npos) k.replace(pos, strlen(r[0]), r[1]);
    }
    HostTensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { if (shape[i] < 0) return fail(TLD_ERR_SHAPE, "'%s': negative dimension", key); t.shape.push_back(shape[i]); n *= shape[i]; }
    t.data.assign(reinterpret_cast<const float*>(host_ptr), reinterpret_cast<const float*>(host_ptr) + n);
    v->host[k] = std::move(t);
    return TLD_OK;
}

int tld_vae_finalize_weights(tld_vae* v) {
    if (!v) return fail(TLD_ERR_INVALID, "null vae");
    if (v->finalized) return fail(TLD_ERR_STATE, "weights already finalized");
    DeviceGuard guard(v->cfg.device_id);
    const int zc = v->zc, C0 = v->C0;
    const HostTensor* t = nullptr;
    if (v->cfg.use_post_quant_conv) {
        if (int rc = need(v, "post_quant_conv.weight", {zc, zc, 1, 1}, &t)) return rc;
        if (int rc = upload_f32(v, t->data, &v->pq_w)) return rc;
        if (int rc = need(v, "post_quant_conv.bias", {zc}, &t)) return rc;
        if (int rc = upload_f32(v, t->data, &v->pq_b)) return rc;
    }
    {   // conv_in: [C0][zc][3][3] -> [tap][zc][C0]
        if (int rc = need(v, "decoder.conv_in.weight", {C0, zc, 3, 3}, &t)) return rc;
        std::vector<float> wt((size_t)9 * zc * C0);
        for (int o = 0; o < C0; ++o)
            for (int c = 0; c < zc; ++c)
                for (int tap = 0; tap < 9; ++tap) wt[((size_t)tap * zc + c) * C0 + o] = t->data[((size_t)o * zc + c) * 9 + tap];
        if (int rc = upload_f32(v, wt, &v->cin_wt)) return rc;
        if (int rc = need(v, "decoder.conv_in.bias", {C0}, &t)) return rc;
        if (int rc = upload_f32(v, t->data, &v->cin_b)) return rc;
    }
    if (int rc = pack_resnet(v, "decoder.mid_block.resnets.0", C0, C0, &v->mid0)) return rc;
    if (int rc = pack_resnet(v, "decoder.mid_block.resnets.1", C0, C0, &v->mid1)) return rc;
    if (v->cfg.mid_block_attention) {
        const std::string a = "decoder.mid_block.attentions.0";
        if (int rc = pack_gn(v, a + ".group_norm", C0, &v->attn_gn)) return rc;
        std::vector<float> w((size_t)3 * C0 * C0), b((size_t)3 * C0);
        const char* names[3] = {".to_q", ".to_k", ".to_v"};
        for (int i = 0; i < 3; ++i) {
            if (int rc = need(v, a + names[i] + ".weight", {C0, C0}, &t)) return rc;
            memcpy(w.data() + (size_t)i * C0 * C0, t->data.data(), (size_t)C0 * C0 * 4);
            if (int rc = need(v, a + names[i] + ".bias", {C0}, &t)) return rc;
            memcpy(b.data() + (size_t)i * C0, t->data.data(), (size_t)C0 * 4);
        }
        v->attn_qkv.cin = C0; v->attn_qkv.cout = 3 * C0; v->attn_qkv.k = 1;
        if (int rc = upload_bf16(v, w, &v->attn_qkv.w)) return rc;
        if (int rc = upload_f32(v, b, &v->attn_qkv.b)) return rc;
        if (int rc = need(v, a + ".to_out.0.weight", {C0, C0}, &t)) return rc;
        v->attn_out.cin = C0; v->attn_out.cout = C0; v->attn_out.k = 1;
        if (int rc = upload_bf16(v, t->data, &v->attn_out.w)) return rc;
        if (int rc = need(v, a + ".to_out.0.bias", {C0}, &t)) return rc;
        if (int rc = upload_f32(v, t->data, &v->attn_out.b)) return rc;
    }
    {
        std::vector<float> z(1024, 0.f);
        if (int rc = upload_f32(v, z, &v->zero_bias)) return rc;
    }
    v->ups.resize(v->nb);
    int c = C0;
    for (int i = 0; i < v->nb; ++i) {
        UpBlock& ub = v->ups[i];
        ub.cout = v->boc[v->nb - 1 - i];
        ub.res.resize(v->cfg.layers_per_block + 1);
        for (int j = 0; j <= v->cfg.layers_per_block; ++j) {
            const std::string pre = "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
            if (int rc = pack_resnet(v, pre, j == 0 ? c : ub.cout, ub.cout, &ub.res[j])) return rc;
        }
        c = ub.cout;
        ub.has_up = i != v->nb - 1;
        if (ub.has_up)
            if (int rc = pack_conv(v, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", c, c, 3, &ub.up)) return rc;
    }
    if (int rc = pack_gn(v, "decoder.conv_norm_out", c, &v->norm_out)) return rc;
    {   // conv_out: bf16 [oc][9][c]; its bias is added by the fp32 tail kernel
        if (int rc = pack_conv(v, "decoder.conv_out", c, v->oc, 3, &v->conv_out)) return rc;
        v->cout_b = v->conv_out.b;
    }
    v->host.clear();
    HIP_TRY(hipDeviceSynchronize());
    v->finalized = true;
    return TLD_OK;
}

int tld_vae_decode(tld_vae* v, const void* z, float* out, int32_t batch, int32_t io_dtype, void* hip_stream) {
    if (!v || !z || !out) return fail(TLD_ERR_INVALID, "null argument");
    if (!v->finalized) return fail(TLD_ERR_STATE, "weights not finalized");
    if (batch < 1 || batch > v->cfg.max_batch) return fail(TLD_ERR_INVALID, "batch=%d outside 1..max_batch=%d", batch, v->cfg.max_batch);
    if (io_dtype != TLD_DTYPE_F32 && io_dtype != TLD_DTYPE_BF16 && io_dtype != TLD_DTYPE_F16) return fail(TLD_ERR_INVALID, "io_dtype=%d", io_dtype);
    DeviceGuard guard(v->cfg.device_id);
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    const int B = batch;
    if (v->debug) clear_stages(v);
    for (int k = 0; k < VC_COUNT; ++k) if (!v->profile) v->ev_used[k] = 0;

    int H = v->hl, W = v->hl;
    v->have_partial = false;
    const float* zf = reinterpret_cast<const float*>(z);
    {
        Timer t(v, VC_OTHER, s);
        if (io_dtype != TLD_DTYPE_F32) {
            const long n = (long)B * v->zc * H * W;
            hipLaunchKernelGGL(vae_cast_in_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, io_dtype, v->io_z, n);
            zf = v->io_z;
        }
        const int threads = ((std::max(v->C0, 9 * v->zc) + 63) / 64) * 64;
        hipLaunchKernelGGL(vae_conv_in_kernel, dim3(B * H * W), dim3(threads), 0, s, zf, v->zc, H, W, v->pq_w, v->pq_b, v->cin_wt, v->cin_b, v->data(0), v->C0);
        if (int rc = check_launch("conv_in")) return rc;
    }
    int x = 0;
    int C = v->C0;
    if (int rc = snapshot(v, "conv_in", x, B, H, W, C, s)) return rc;
    if (int rc = resnet(v, v->mid0, &x, B, H, W, s)) return rc;
    if (int rc = snapshot(v, "mid.res0", x, B, H, W, C, s)) return rc;
    if (v->cfg.mid_block_attention) {
        if (int rc = attention(v, &x, B, H, W, C, s)) return rc;
        if (int rc = snapshot(v, "mid.attn", x, B, H, W, C, s)) return rc;
    }
    if (int rc = resnet(v, v->mid1, &x, B, H, W, s)) return rc;
    if (int rc = snapshot(v, "mid.res1", x, B, H, W, C, s)) return rc;
    for (int i = 0; i < v->nb; ++i) {
        const UpBlock& ub = v->ups[i];
        for (size_t j = 0; j < ub.res.size(); ++j) {
            if (int rc = resnet(v, ub.res[j], &x, B, H, W, s)) return rc;
            C = ub.cout;
            const std::string nm = "up" + std::to_string(i) + ".res" + std::to_string(j);
            if (int rc = snapshot(v, nm.c_str(), x, B, H, W, C, s)) return rc;
        }
        if (ub.has_up) {                      // Upsample2D: nearest 2x, then conv 3x3 -- one implicit GEMM over the small image
            const int dst = (x + 1) & 3;
            H *= 2; W *= 2;
            if (int rc = conv3x3(v, x, dst, ub.up, B, H, W, 1, EPI_BIAS_BF16, nullptr, s)) return rc;
            x = dst;
            const std::string nm = "up" + std::to_string(i) + ".upsample";
            if (int rc = snapshot(v, nm.c_str(), x, B, H, W, C, s)) return rc;
        }
    }
    const int t = (x + 1) & 3;
    if (int rc = group_norm(v, x, t, v->norm_out, B, H * W, C, true, s)) return rc;
    if (int rc = snapshot(v, "norm_out", t, B, H, W, C, s)) return rc;
    if (int rc = conv3x3(v, t, -1, v->conv_out, B, H, W, 0, EPI_F32, v->out_f32, s)) return rc;
    {
        Timer tm(v, VC_OTHER, s);
        const long total = (long)B * H * W;
        hipLaunchKernelGGL(vae_out_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, v->out_f32, v->cout_b, out, H * W, v->oc, total);
        if (int rc = check_launch("conv_out tail")) return rc;
    }
    return TLD_OK;
}

int tld_vae_set_debug(tld_vae* v, int32_t enable) {
    if (!v) return fail(TLD_ERR_INVALID, "null vae");
    DeviceGuard guard(v->cfg.device_id);
    v->debug = enable != 0;
    if (!v->debug) clear_stages(v);
    return TLD_OK;
}

int tld_vae_read_stage(tld_vae* v, const char* name, float* host_out, int64_t numel, int64_t* shape4) {
    if (!v || !name || !host_out) return fail(TLD_ERR_INVALID, "null argument");
    DeviceGuard guard(v->cfg.device_id);
    for (const Stage& st : v->stages) {
        if (st.name != name) continue;
        const size_t n = (size_t)st.B * st.C * st.H * st.W;
        if (shape4) { shape4[0] = st.B; shape4[1] = st.C; shape4[2] = st.H; shape4[3] = st.W; }
        if ((int64_t)n != numel) return fail(TLD_ERR_SHAPE, "stage '%s' has %zu elements, caller passed %lld", name, n, (long long)numel);
        HIP_TRY(hipDeviceSynchronize());
        std::vector<uint16_t> tmp(n);
        HIP_TRY(hipMemcpy(tmp.data(), st.dev, n * 2, hipMemcpyDeviceToHost));
        const size_t HW = (size_t)st.H * st.W;
        for (int b = 0; b < st.B; ++b)
            for (size_t p = 0; p < HW; ++p)
                for (int c = 0; c < st.C; ++c)
                    host_out[((size_t)b * st.C + c) * HW + p] = bf16_to_f32(tmp[((size_t)b * HW + p) * st.C + c]);
        return TLD_OK;
    }
    return fail(TLD_ERR_KEY, "no captured stage named '%s' (set_debug before decode?)", name);
}

int tld_vae_set_profile(tld_vae* v, int32_t enable) {
    if (!v) return fail(TLD_ERR_INVALID, "null vae");
    v->profile = enable != 0;
    for (int k = 0; k < VC_COUNT; ++k) v->ev_used[k] = 0;
    return TLD_OK;
}

int tld_vae_get_profile(tld_vae* v, int32_t kclass, double* total_ms, int64_t* launches) {
    if (!v || kclass < 0 || kclass >= VC_COUNT || !total_ms || !launches) return fail(TLD_ERR_INVALID, "bad argument");
    DeviceGuard guard(v->cfg.device_id);
    HIP_TRY(hipDeviceSynchronize());
    double tot = 0.0;
    for (size_t i = 0; i < v->ev_used[kclass]; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, v->ev[kclass][i].first, v->ev[kclass][i].second));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int64_t)v->ev_used[kclass];
    return TLD_OK;
}

int64_t tld_vae_weight_bytes(const tld_vae* v) { return v ? v->weight_bytes : 0; }

int tld_vae_destroy(tld_vae* v) {
    if (!v) return TLD_OK;
    DeviceGuard guard(v->cfg.device_id);
    clear_stages(v);
    for (int k = 0; k < VC_COUNT; ++k)
        for (auto& e : v->ev[k]) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (void* p : v->allocs) (void)hipFree(p);
    delete v;
    return TLD_OK;
}

// Test hook: the implicit-GEMM 3x3 convolution alone.  in: bf16 channels-last [B, H >> up, W >> up, cin] (device);
// w: bf16 [cout][3][3][cin] (device); out: fp32 [B*H*W][cout] (device).  Allocates a scratch copy with the zero page.
int tld_debug_conv3x3(const void* in_bf16, const void* w_bf16, float* out_f32, int32_t B, int32_t H, int32_t W, int32_t cin,
                      int32_t cout, int32_t up, void* hip_stream) {
    if (!in_bf16 || !w_bf16 || !out_f32) return fail(TLD_ERR_INVALID, "null argument");
    if (cin % 64 || cin < 64) return fail(TLD_ERR_INVALID, "cin=%d must be a multiple of 64", cin);
    if (up != 0 && up != 1) return fail(TLD_ERR_INVALID, "up must be 0 or 1");
    PtrDeviceGuard guard(in_bf16);
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    const size_t n = (size_t)B * (H >> up) * (W >> up) * cin * 2;
    if (n + kHdr >= (1ull << 32)) return fail(TLD_ERR_INVALID, "operands must be smaller than 4 GiB");
    char* buf = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&buf), n + kHdr));
    HIP_TRY(hipMemsetAsync(buf, 0, kHdr, s));
    HIP_TRY(hipMemcpyAsync(buf + kHdr, in_bf16, n, hipMemcpyDeviceToDevice, s));
    GemmParams p{};
    p.A = reinterpret_cast<const bf16*>(buf);
    p.conv = 1; p.cv_h = H; p.cv_w = W; p.cv_up = up; p.cv_cin = cin; p.cv_data_off = kHdr; p.lda = cin;
    p.W = reinterpret_cast<const bf16*>(w_bf16); p.ldw = 9 * cin;
    p.M = B * H * W; p.N = cout; p.K = 9 * cin;
    p.c_f32 = out_f32; p.ldc = cout;
    launch_gemm(p, EPI_F32, s);
    const hipError_t e = hipGetLastError();
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipFree(buf));
    if (e != hipSuccess) return fail(TLD_ERR_HIP, "conv3x3 launch failed: %s", hipGetErrorString(e));
    return TLD_OK;
}

}  // extern "C"
