// tld_clip.hip -- CLIP text tower on gfx950: the front edge of the pipeline.
//
// Replaces `model.encode_text(text_tokens)` of the reference's encode_text (tld/diffusion.py:136-140), where `model` is
// OpenAI CLIP "ViT-L/14" loaded by `clip.load` (tld/diffusion.py:160, tld/configs.py:46-48) -- a third-party package that is
// not part of the reference checkout.  What is restated is CLIP.encode_text of openai/CLIP (clip/model.py):
//   x = token_embedding[text] + positional_embedding
//   12 x ResidualAttentionBlock: x += MHA(ln_1(x), causal mask);  x += c_proj(QuickGELU(c_fc(ln_2(x))))
//   x = ln_final(x);  out = x[arange(B), text.argmax(-1)] @ text_projection
// Tokenisation (a Python BPE over a vocabulary file) stays on the host; this engine takes token ids.
//
// Layout: the residual stream is fp32 [B*ctx, width]; LayerNorm outputs and the projections' operands are bf16; the four
// projections of a block are the persistent MFMA GEMM of tld_gemm.hip (bias -> bf16, or fp32 out for the two that end in the
// residual add); the add, its bias and the next LayerNorm are one row kernel; causal attention over <= 128 tokens with
// head_dim 64 is a one-wave-per-(sample, head) fp32 kernel (77 x 77 scores: no tile to speak of).  One prompt batch is
// ~13 GFLOP per prompt: this path is about removing the host round trip, not about its own speed.
#include "../../include/tld_hip.h"
#include "tld_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace tld;

namespace {

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

// x[t] = token_embedding[text[t]] + positional_embedding[t % ctx]      (clip/model.py encode_text, first two lines)
__global__ void clip_embed_kernel(const int* __restrict__ tokens, const float* __restrict__ tok_emb, const float* __restrict__ pos,
                                  float* __restrict__ x, int T, int W, int ctx, int vocab) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * W) return;
    const int t = (int)(i / W), c = (int)(i - (long)t * W);
    int id = tokens[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    x[i] = tok_emb[(size_t)id * W + c] + pos[(size_t)(t % ctx) * W + c];
}

// One wave per row:  if add: x[row] += add[row] + bias (the residual add of the block that just finished);  out = bf16(LayerNorm(x[row])).
// fp32 statistics, two passes over the row held in registers (W <= 64 * 16).
__global__ __launch_bounds__(256) void clip_add_ln_kernel(float* __restrict__ x, const float* __restrict__ add, const float* __restrict__ bias,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          bf16* __restrict__ out, int T, int W) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    float v[16];
    const int n = W / 64;                          // elements per lane (W % 64 == 0, n <= 16)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < n) {
            const int c = j * 64 + lane;
            float f = x[(size_t)row * W + c];
            if (add) { f += add[(size_t)row * W + c] + bias[c]; x[(size_t)row * W + c] = f; }
            v[j] = f; s += f;
        }
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j < n) { const float c = v[j] - mean; q = fmaf(c, c, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + kLnEps);
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < n) { const int c = j * 64 + lane; out[(size_t)row * W + c] = (bf16)((v[j] - mean) * rstd * gamma[c] + beta[c]); }
}

// The same for the B pooled rows only (row = b * ctx + eot[b]): residual add of the last block, ln_final, fp32 out [B, W]
__global__ __launch_bounds__(64) void clip_final_ln_kernel(const float* __restrict__ x, const float* __restrict__ add, const float* __restrict__ bias,
                                                           const int* __restrict__ eot, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ out, int W, int ctx) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int e = eot[b];
    e = e < 0 ? 0 : (e >= ctx ? ctx - 1 : e);
    const size_t row = (size_t)b * ctx + e;
    float v[16];
    const int n = W / 64;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < n) { const int c = j * 64 + lane; v[j] = x[row * W + c] + add[row * W + c] + bias[c]; s += v[j]; }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j < n) { const float c = v[j] - mean; q = fmaf(c, c, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + kLnEps);
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < n) { const int c = j * 64 + lane; out[(size_t)b * W + c] = (v[j] - mean) * rstd * gamma[c] + beta[c]; }
}

// Causal multi-head attention, head_dim 64, ctx <= 128 tokens: one wave per (head, sample).  q | k | v are the bf16 rows of the
// in_proj GEMM ([T, 3W]: q at column h*64, k at W + h*64, v at 2W + h*64 -- nn.MultiheadAttention's packed in_proj order).
//   scores s_ij = q_i . k_j / 8 for j <= i (attn_mask: -inf above the diagonal), softmax over j, o_i = sum_j p_ij v_j.
// Lane j owns keys j and j + 64 in the score phase and output feature j in the value phase.
__global__ __launch_bounds__(64) void clip_attn_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, int ctx, int W) {
    extern __shared__ float sm[];
    float* Qs = sm;                        // [ctx][64]
    float* Ks = Qs + ctx * 64;             // [ctx][65]  (pitch 65: lane j reads row j)
    float* Vs = Ks + ctx * 65;             // [ctx][64]
    float* P = Vs + ctx * 64;              // [128]
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const bf16* base = qkv + (size_t)b * ctx * 3 * W + h * 64;
    for (int i = 0; i < ctx; ++i) {
        const bf16* r = base + (size_t)i * 3 * W;
        Qs[i * 64 + lane] = (float)r[lane];
        Ks[i * 65 + lane] = (float)r[W + lane];
        Vs[i * 64 + lane] = (float)r[2 * W + lane];
    }
    __syncthreads();
    const int j0 = lane, j1 = lane + 64;
    for (int i = 0; i < ctx; ++i) {
        float s0 = -INFINITY, s1 = -INFINITY;
        if (j0 <= i) {
            float a = 0.f;
            for (int k = 0; k < 64; ++k) a = fmaf(Qs[i * 64 + k], Ks[j0 * 65 + k], a);
            s0 = a * 0.125f;
        }
        if (j1 <= i) {                     // (j1 <= i < ctx)
            float a = 0.f;
            for (int k = 0; k < 64; ++k) a = fmaf(Qs[i * 64 + k], Ks[j1 * 65 + k], a);
            s1 = a * 0.125f;
        }
        float m = fmaxf(s0, s1);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const float e0 = j0 <= i ? expf(s0 - m) : 0.f, e1 = j1 <= i ? expf(s1 - m) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        P[j0] = e0 * inv; P[j1] = e1 * inv;
        __syncthreads();
        float o = 0.f;
        for (int j = 0; j <= i; ++j) o = fmaf(P[j], Vs[j * 64 + lane], o);
        out[((size_t)b * ctx + i) * W + h * 64 + lane] = (bf16)o;
        __syncthreads();
    }
}

// QuickGELU in place: x * sigmoid(1.702 x)      (clip/model.py QuickGELU)
__global__ void clip_quickgelu_kernel(bf16* __restrict__ f, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    bf16x8 v = *reinterpret_cast<bf16x8*>(f + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (float)v[e];
        v[e] = (bf16)(x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)));
    }
    *reinterpret_cast<bf16x8*>(f + i * 8) = v;
}

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };

struct Block {
    float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
    bf16 *in_w = nullptr, *out_w = nullptr, *fc_w = nullptr, *proj_w = nullptr;
    float *in_b = nullptr, *out_b = nullptr, *fc_b = nullptr, *proj_b = nullptr;
};

}  // namespace

struct tld_clip {
    tld_clip_config cfg{};
    int W = 0, L = 0, H = 0, ctx = 0, E = 0, V = 0;
    bool finalized = false;
    std::map<std::string, HostTensor> host;
    std::vector<void*> allocs;
    int64_t weight_bytes = 0;
    float *tok_emb = nullptr, *pos = nullptr, *lnf_g = nullptr, *lnf_b = nullptr, *proj_t = nullptr;     // proj_t: text_projection^T [E][W]
    std::vector<Block> blocks;
    // workspace (max_batch * ctx rows)
    float *x = nullptr, *tmp = nullptr, *pooled = nullptr;
    bf16 *h = nullptr, *qkv = nullptr, *att = nullptr, *f = nullptr;
};

namespace {

template <typename T>
int dev_alloc(tld_clip* c, T** out, size_t count, bool weight = false) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
    c->allocs.push_back(p);
    if (weight) c->weight_bytes += (int64_t)(count * sizeof(T));
    *out = reinterpret_cast<T*>(p);
    return TLD_OK;
}
int upload_f32(tld_clip* c, const std::vector<float>& hv, float** out) {
    if (int rc = dev_alloc(c, out, hv.size(), true)) return rc;
    HIP_TRY(hipMemcpy(*out, hv.data(), hv.size() * sizeof(float), hipMemcpyHostToDevice));
    return TLD_OK;
}
int upload_bf16(tld_clip* c, const std::vector<float>& hv, bf16** out) {
    std::vector<uint16_t> t(hv.size());
    for (size_t i = 0; i < hv.size(); ++i) t[i] = f32_to_bf16_rne(hv[i]);
    if (int rc = dev_alloc(c, out, hv.size(), true)) return rc;
    HIP_TRY(hipMemcpy(*out, t.data(), t.size() * 2, hipMemcpyHostToDevice));
    return TLD_OK;
}
int need(const tld_clip* c, const std::string& key, const std::vector<int64_t>& shape, const HostTensor** out) {
    auto it = c->host.find(key);
    if (it == c->host.end()) return fail(TLD_ERR_STATE, "missing state_dict entry '%s'", key.c_str());
    if (it->second.shape != shape) {
        std::string got, want;
        for (int64_t s : it->second.shape) got += std::to_string(s) + ",";
        for (int64_t s : shape) want += std::to_string(s) + ",";
        return fail(TLD_ERR_SHAPE, "'%s' has shape [%s], expected [%s]", key.c_str(), got.c_str(), want.c_str());
    }
    *out = &it->second;
    return TLD_OK;
}
int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TLD_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return TLD_OK;
}
void gemm(const bf16* A, int lda, const bf16* Wt, int M, int N, int K, int epi, const float* bias, bf16* out_bf16, float* out_f32, hipStream_t s) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = Wt; p.ldw = K; p.M = M; p.N = N; p.K = K; p.bias = bias;
    if (epi == EPI_BIAS_BF16) { p.out_bf16 = out_bf16; p.ldo = N; }
    else { p.c_f32 = out_f32; p.ldc = N; }
    launch_gemm(p, epi, s);
}

}  // namespace

extern "C" {

int tld_clip_create(const tld_clip_config* cfg, tld_clip** out) {
    if (!cfg || !out) return fail(TLD_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->width < 64 || cfg->width > 1024 || cfg->width % 64) return fail(TLD_ERR_INVALID, "width=%d: a multiple of 64 in 64..1024", cfg->width);
    if (cfg->heads * 64 != cfg->width) return fail(TLD_ERR_INVALID, "heads=%d: head_dim must be 64 (width / heads)", cfg->heads);
    if (cfg->layers < 1 || cfg->layers > 64) return fail(TLD_ERR_INVALID, "layers=%d: 1..64", cfg->layers);
    if (cfg->context_length < 1 || cfg->context_length > 128) return fail(TLD_ERR_INVALID, "context_length=%d: 1..128", cfg->context_length);
    if (cfg->embed_dim < 1 || cfg->vocab_size < 1 || cfg->max_batch < 1) return fail(TLD_ERR_INVALID, "embed_dim / vocab_size / max_batch must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TLD_ERR_HIP, "no HIP device available (the text tower has no CPU path)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(TLD_ERR_INVALID, "device_id=%d out of range (%d devices)", cfg->device_id, ndev);
    DeviceGuard guard(cfg->device_id);
    tld_clip* c = new tld_clip();
    c->cfg = *cfg;
    c->W = cfg->width; c->L = cfg->layers; c->H = cfg->heads; c->ctx = cfg->context_length; c->E = cfg->embed_dim; c->V = cfg->vocab_size;
    const size_t T = (size_t)cfg->max_batch * c->ctx, W = c->W;
    auto bail = [&](int rc) { tld_clip_destroy(c); return rc; };
    if (int rc = dev_alloc(c, &c->x, T * W)) return bail(rc);
    if (int rc = dev_alloc(c, &c->tmp, T * W)) return bail(rc);
    if (int rc = dev_alloc(c, &c->pooled, (size_t)cfg->max_batch * W)) return bail(rc);
    if (int rc = dev_alloc(c, &c->h, T * W)) return bail(rc);
    if (int rc = dev_alloc(c, &c->qkv, T * 3 * W)) return bail(rc);
    if (int rc = dev_alloc(c, &c->att, T * W)) return bail(rc);
    if (int rc = dev_alloc(c, &c->f, T * 4 * W)) return bail(rc);
    *out = c;
    return TLD_OK;
}

int tld_clip_load_tensor(tld_clip* c, const char* key, const void* host_ptr, const int64_t* shape, int32_t ndim, int32_t dtype) {
    if (!c || !key || (!host_ptr && ndim > 0) || ndim < 0 || ndim > 8) return fail(TLD_ERR_INVALID, "bad argument");
    if (c->finalized) return fail(TLD_ERR_STATE, "weights already finalized");
    std::string k(key);
    if (k.rfind("visual.", 0) == 0 || k == "logit_scale" || k == "input_resolution" || k == "context_length" || k == "vocab_size")
        return TLD_OK;                                      // the image tower and the archive's metadata are not used by encode_text
    const bool known = k == "token_embedding.weight" || k == "positional_embedding" || k == "text_projection" ||
                       k.rfind("ln_final.", 0) == 0 || k.rfind("transformer.resblocks.", 0) == 0;
    if (!known) return fail(TLD_ERR_KEY, "unknown state_dict key '%s'", key);
    if (dtype != TLD_DTYPE_F32) return fail(TLD_ERR_INVALID, "'%s': host tensors must be fp32", key);
    HostTensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { if (shape[i] < 0) return fail(TLD_ERR_SHAPE, "'%s': negative dimension", key); t.shape.push_back(shape[i]); n *= shape[i]; }
    t.data.assign(reinterpret_cast<const float*>(host_ptr), reinterpret_cast<const float*>(host_ptr) + n);
    c->host[k] = std::move(t);
    return TLD_OK;
}

int tld_clip_finalize_weights(tld_clip* c) {
    if (!c) return fail(TLD_ERR_INVALID, "null handle");
    if (c->finalized) return fail(TLD_ERR_STATE, "weights already finalized");
    DeviceGuard guard(c->cfg.device_id);
    const int W = c->W, E = c->E;
    const HostTensor* t = nullptr;
    if (int rc = need(c, "token_embedding.weight", {c->V, W}, &t)) return rc;
    if (int rc = upload_f32(c, t->data, &c->tok_emb)) return rc;
    if (int rc = need(c, "positional_embedding", {c->ctx, W}, &t)) return rc;
    if (int rc = upload_f32(c, t->data, &c->pos)) return rc;
    if (int rc = need(c, "ln_final.weight", {W}, &t)) return rc;
    if (int rc = upload_f32(c, t->data, &c->lnf_g)) return rc;
    if (int rc = need(c, "ln_final.bias", {W}, &t)) return rc;
    if (int rc = upload_f32(c, t->data, &c->lnf_b)) return rc;
    if (int rc = need(c, "text_projection", {W, E}, &t)) return rc;
    {
        std::vector<float> pt((size_t)E * W);
        for (int k = 0; k < W; ++k) for (int n = 0; n < E; ++n) pt[(size_t)n * W + k] = t->data[(size_t)k * E + n];
        if (int rc = upload_f32(c, pt, &c->proj_t)) return rc;
    }
    c->blocks.resize(c->L);
    for (int l = 0; l < c->L; ++l) {
        Block& b = c->blocks[l];
        const std::string p = "transformer.resblocks." + std::to_string(l) + ".";
        struct Item { const char* key; std::vector<int64_t> shape; float** f32; bf16** b16; };
        const std::vector<Item> items = {
            {"ln_1.weight", {W}, &b.ln1_g, nullptr}, {"ln_1.bias", {W}, &b.ln1_b, nullptr},
            {"attn.in_proj_weight", {3 * W, W}, nullptr, &b.in_w}, {"attn.in_proj_bias", {3 * W}, &b.in_b, nullptr},
            {"attn.out_proj.weight", {W, W}, nullptr, &b.out_w}, {"attn.out_proj.bias", {W}, &b.out_b, nullptr},
            {"ln_2.weight", {W}, &b.ln2_g, nullptr}, {"ln_2.bias", {W}, &b.ln2_b, nullptr},
            {"mlp.c_fc.weight", {4 * W, W}, nullptr, &b.fc_w}, {"mlp.c_fc.bias", {4 * W}, &b.fc_b, nullptr},
            {"mlp.c_proj.weight", {W, 4 * W}, nullptr, &b.proj_w}, {"mlp.c_proj.bias", {W}, &b.proj_b, nullptr},
        };
        for (auto& it : items) {
            if (int rc = need(c, p + it.key, it.shape, &t)) return rc;
            if (it.f32) { if (int rc = upload_f32(c, t->data, it.f32)) return rc; }
            else { if (int rc = upload_bf16(c, t->data, it.b16)) return rc; }
        }
    }
    c->host.clear();
    HIP_TRY(hipDeviceSynchronize());
    c->finalized = true;
    return TLD_OK;
}

int tld_clip_encode_text(tld_clip* c, const int32_t* tokens, const int32_t* eot_index, float* out, int32_t batch, void* hip_stream) {
    if (!c || !tokens || !eot_index || !out) return fail(TLD_ERR_INVALID, "null argument");
    if (!c->finalized) return fail(TLD_ERR_STATE, "weights not finalized");
    if (batch < 1 || batch > c->cfg.max_batch) return fail(TLD_ERR_INVALID, "batch=%d outside 1..max_batch=%d", batch, c->cfg.max_batch);
    DeviceGuard guard(c->cfg.device_id);
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    const int W = c->W, ctx = c->ctx, T = batch * ctx;
    {
        const long n = (long)T * W;
        hipLaunchKernelGGL(clip_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tokens, c->tok_emb, c->pos, c->x, T, W, ctx, c->V);
    }
    const dim3 rows((T + 3) / 4);
    const size_t attn_lds = (size_t)(ctx * 64 * 2 + ctx * 65 + 128) * sizeof(float);
    static PerDeviceOnce attr_set;
    attr_set.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(clip_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * (64 * 2 + 65) * 4 + 512); });
    hipLaunchKernelGGL(clip_add_ln_kernel, rows, dim3(256), 0, s, c->x, (const float*)nullptr, (const float*)nullptr, c->blocks[0].ln1_g, c->blocks[0].ln1_b, c->h, T, W);
    for (int l = 0; l < c->L; ++l) {
        const Block& b = c->blocks[l];
        gemm(c->h, W, b.in_w, T, 3 * W, W, EPI_BIAS_BF16, b.in_b, c->qkv, nullptr, s);
        hipLaunchKernelGGL(clip_attn_kernel, dim3(c->H, batch), dim3(64), attn_lds, s, c->qkv, c->att, ctx, W);
        gemm(c->att, W, b.out_w, T, W, W, EPI_F32, nullptr, nullptr, c->tmp, s);
        hipLaunchKernelGGL(clip_add_ln_kernel, rows, dim3(256), 0, s, c->x, c->tmp, b.out_b, b.ln2_g, b.ln2_b, c->h, T, W);
        gemm(c->h, W, b.fc_w, T, 4 * W, W, EPI_BIAS_BF16, b.fc_b, c->f, nullptr, s);
        {
            const long n8 = (long)T * 4 * W / 8;
            hipLaunchKernelGGL(clip_quickgelu_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, c->f, n8);
        }
        gemm(c->f, 4 * W, b.proj_w, T, W, 4 * W, EPI_F32, nullptr, nullptr, c->tmp, s);
        if (l + 1 < c->L)
            hipLaunchKernelGGL(clip_add_ln_kernel, rows, dim3(256), 0, s, c->x, c->tmp, b.proj_b, c->blocks[l + 1].ln1_g, c->blocks[l + 1].ln1_b, c->h, T, W);
        else
            hipLaunchKernelGGL(clip_final_ln_kernel, dim3(batch), dim3(64), 0, s, c->x, c->tmp, b.proj_b, eot_index, c->lnf_g, c->lnf_b, c->pooled, W, ctx);
    }
    launch_linear_f32(c->pooled, W, c->proj_t, nullptr, out, c->E, batch, W, c->E, 0, s);
    return check_launch("encode_text");
}

int tld_clip_read_buffer(tld_clip* c, const char* name, float* host_out, int64_t numel) {
    if (!c || !name || !host_out) return fail(TLD_ERR_INVALID, "null argument");
    DeviceGuard guard(c->cfg.device_id);
    HIP_TRY(hipDeviceSynchronize());
    const std::string n(name);
    {   // bound the read by the buffer's size (workspace is sized for max_batch * context_length rows)
        const int64_t TW = (int64_t)c->cfg.max_batch * c->ctx * c->W;
        const int64_t cap = n == "pooled" ? (int64_t)c->cfg.max_batch * c->W : n == "qkv" ? 3 * TW : n == "f" ? 4 * TW : TW;
        if (numel <= 0 || numel > cap) return fail(TLD_ERR_INVALID, "numel %lld outside (0, %lld] for buffer '%s'", (long long)numel, (long long)cap, name);
    }
    const float* f32 = n == "x" ? c->x : n == "tmp" ? c->tmp : n == "pooled" ? c->pooled : nullptr;
    const bf16* b16 = n == "h" ? c->h : n == "qkv" ? c->qkv : n == "att" ? c->att : n == "f" ? c->f : nullptr;
    if (f32) { HIP_TRY(hipMemcpy(host_out, f32, (size_t)numel * 4, hipMemcpyDeviceToHost)); return TLD_OK; }
    if (b16) {
        std::vector<uint16_t> t((size_t)numel);
        HIP_TRY(hipMemcpy(t.data(), b16, (size_t)numel * 2, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < numel; ++i) { const uint32_t u = (uint32_t)t[i] << 16; memcpy(host_out + i, &u, 4); }
        return TLD_OK;
    }
    return fail(TLD_ERR_KEY, "no buffer named '%s' (x, tmp, pooled, h, qkv, att, f)", name);
}

int64_t tld_clip_weight_bytes(const tld_clip* c) { return c ? c->weight_bytes : 0; }

int tld_clip_destroy(tld_clip* c) {
    if (!c) return TLD_OK;
    DeviceGuard guard(c->cfg.device_id);
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
    return TLD_OK;
}

}  // extern "C"
