// tld_train.hip -- the training step of the denoiser on gfx950: C ABI tld_train_*.
//
// Replaces, for one optimisation step of tld/train.py:162-173,
//     pred = model(x_noisy, noise_level.view(-1, 1), label); loss = MSELoss(pred, x); loss.backward()      -> tld_train_forward_backward
//     optimizer.step() (torch.optim.Adam) + update_ema(ema_model, model, alpha)  (tld/train.py:55-58)       -> tld_train_adam_ema
// The gradient all-reduce of accelerate's DDP wrapper stays with the caller (torch.distributed over RCCL on the flat gradient buffer,
// transformer_latent_diffusion_amd/train.py): parameters, gradients, Adam moments and the EMA copy are FLAT fp32 device buffers
// owned by the caller in the canonical order of Denoiser.named_parameters() (tld_train_param_layout).
//
// Arithmetic: fp32 master parameters; the big projections (QKV, cross-attention Q, MLP up / down) and their dX / dW products are
// bf16-operand, fp32-accumulate MFMA GEMMs (tld_gemm.hip); activations that are saved for the backward are bf16; LayerNorm, softmax,
// GELU, the conditioning path, the patch embedding, reductions and the optimizer are fp32.  The training forward is the plain module
// graph (none of the inference path's algebraic folds), so every saved tensor is the one autograd would save.
#include "../../include/tld_hip.h"
#include "tld_train_kernels.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace tld;
using namespace tld::train;

namespace tld {
int launch_attention_bwd(const bf16* qk, const bf16* vt, const bf16* o, const float* g, bf16* dqkv, float* stats, int batch, int ntok, int heads, hipStream_t s);
int launch_attention_bwd(const bf16* qk, const bf16* vt, const bf16* o, const bf16* g, bf16* dqkv, float* stats, int batch, int ntok, int heads, hipStream_t s);
}

namespace {

int tfail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    tld::set_last_error(buf);
    return code;
}
#define HIP_TRY(expr)                                                                                                            \
    do {                                                                                                                         \
        hipError_t _e = (expr);                                                                                                  \
        if (_e != hipSuccess) return tfail(TLD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct DevGuard {
    int prev = -1; bool switched = false;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
};

struct Tensor { std::string key; int64_t off, numel; };

struct LayerP {          // offsets into the flat parameter / gradient vectors
    int64_t qkv, kv, q, up_w, up_b, dw_w, dw_b, down_w, down_b, n1w, n1b, n2w, n2b, n3w, n3b;
};
struct LayerB {          // engine-owned per-layer buffers
    bf16 *wqkv, *wqkv_t, *wq, *wq_t, *wup, *wup_t, *wdown, *wdown_t;          // bf16 GEMM operands and their transposes
    bf16 *x1, *x2, *x3;                                                      // residual stream at the input of the three sub-blocks
    float2 *st1, *st2, *st3;
    bf16 *a1, *a2, *a3, *qk, *vt, *att, *qc, *cr, *h, *hc, *gl, *o;
    float *kvc, *p0;
    float* dww_t;                                                            // depthwise taps, tap-major [9][hid] (refreshed with the operands)
};

inline dim3 g1(size_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

}  // namespace

struct tld_train {
    tld_config cfg{};
    int d = 0, L = 0, H = 0, N = 0, G = 0, pd = 0, hid = 0, S = 0, C = 0, ne = 0, text = 0, B = 0;
    std::vector<Tensor> layout;
    int64_t nparam = 0;
    // offsets of the non-layer tensors
    int64_t ff1w, ff1b, ff3w, ff3b, cvw, cvb, l1w, l1b, liw, lib, l2w, l2b, pos, outw, outb, nw, nb, lbw, lbb;
    std::vector<LayerP> lp;
    std::vector<LayerB> lb;
    float *params = nullptr, *grads = nullptr;
    std::vector<void*> allocs;
    float* angular = nullptr;            // sinusoid buffer (not a parameter; tld/transformer_blocks.py:11-15)
    float* zero_bias = nullptr;
    // conditioning path
    float *sinb, *h1, *g1v, *ycat, *y, *dy, *dycat, *dg1;
    float *kvc_all, *dkv_all, *dy_parts;     // per block, contiguous: (k | v) of the conditioning tokens [L][2B, 2d], its gradient, and each block's share of dL/dy [L][2B, d]
    float2* yst;
    // embedding
    float *p16, *p16n, *e, *patches, *de, *dpn, *dp16;
    float2 *est1, *est2;
    bf16* xfin;
    // tail / loss
    float *dout, *row_loss, *io;
    // backward scratch
    float* gx;                           // dL/d(residual stream) fp32 [M, d]
    bf16 *gxb, *T1, *T2, *dbig, *dsmall, *dsmall2;
    float* scr;                          // small fp32 scratch ([pd, d])
    float* attn_stats;                   // attention backward at > 256 tokens: [B, H, 2, N] (log-sum-exp | delta)
    float* splitk;                       // split-K partials of the weight-gradient GEMMs [8][max(hid, 3d)][d]
    float* part;                         // reduction partials
    size_t part_floats = 0;
    size_t tr_rows = 0, splitk_floats = 0;      // capacity of the transposed split-K operands (rows) and of the slice buffer
    bool weights_fresh = false;
    bool tn_wgrad = true;                // weight gradients without transposed copies (TLD_TRAIN_TN_WGRAD=0: the transposing form, a test hook)
};

namespace {

void add_t(tld_train* e, const std::string& k, int64_t n, int64_t* off) {
    *off = e->nparam;
    e->layout.push_back({k, e->nparam, n});
    e->nparam += n;
}

template <typename T>
int dalloc(tld_train* e, T** p, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, n * sizeof(T)) != hipSuccess) return tfail(TLD_ERR_HIP, "hipMalloc of %zu bytes failed", n * sizeof(T));
    e->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return 0;
}
#define DALLOC(ptr, n) do { int _r = dalloc(e, &(ptr), (size_t)(n)); if (_r) return _r; } while (0)

// C[Mr, Nc] = A[Mr, K] . W[Nc, K]^T on the engine's MFMA GEMM
// dW[Nout, Kin] = dY^T X (dY [M, Nout], X [M, Kin], bf16 row-major) with both operands as they are -- the contraction index is the row, the MFMA
// fragments come through the transposing LDS read (launch_gemm_tn): no transposed copies (8 transposes per block were 8.4 % of the step).
// The output is small and the contraction long, so the rows are cut into `sk` runs of a multiple of 64 rows (split-K into fp32 slices
// [sk][Nout][Kin], summed in a fixed order: bit-reproducible); sk = the count whose last round of 256 x 256 tiles is fullest.
// Returns false (nothing launched) for shapes the kernel does not take.  (256 x 384 tiles -- 768-byte W rows with a rotating block swizzle, 240 tiles of
// 10 splits instead of 252 of 7 -- measured 30.80 vs 30.88 ms per step, with 17 spilled registers: not kept.)
bool wgrad_tn(const bf16* dY, int Nout, const bf16* X, int Kin, int M, float* dW, float* slices, size_t slice_floats, hipStream_t s) {
    if (Nout % 256 || Kin % 256 || M % 64 || M <= 0) return false;
    const int ncu = device_cu_count();
    const long per = (long)(Nout / 256) * (Kin / 256);
    double best = 0.0; int bsk = 1, bms = M;
    for (int c = 1; c <= 32; ++c) {
        const int mp = ((M + c - 1) / c + 63) / 64 * 64;
        if ((c > 1 && mp < 1024) || (long)(c - 1) * mp >= M) continue;                            // runs too short / last run empty
        if (c > 1 && (size_t)c * Nout * Kin > slice_floats) continue;                            // slice workspace
        const long tiles = per * c, rounds = (tiles + ncu - 1) / ncu;
        const double eff = (double)tiles / (double)(rounds * ncu) * ((double)M / ((double)c * mp));
        if (eff > best + 1e-9) { best = eff; bsk = c; bms = mp; }
    }
    GemmParams g{};
    g.A = dY; g.lda = Nout; g.W = X; g.ldw = Kin; g.M = bsk * Nout; g.N = Kin; g.K = bms; g.tn_ktotal = M; g.ldc = Kin;
    g.c_f32 = bsk == 1 ? dW : slices;
    g.w_batch_rows = bsk == 1 ? 0 : Nout;
    launch_gemm_tn(g, s);
    if (bsk > 1) hipLaunchKernelGGL(sum_slices, dim3((unsigned)(((size_t)Nout * Kin / 4 + 255) / 256)), dim3(256), 0, s, slices, bsk, (size_t)Nout * Kin, dW, (size_t)Nout * Kin / 4);
    return true;
}

void gemm_f32(const bf16* A, int lda, const bf16* W, int ldw, float* C, int Mr, int Nc, int K, hipStream_t s) {
    GemmParams g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = Mr; g.N = Nc; g.K = K; g.c_f32 = C; g.ldc = Nc;
    launch_gemm(g, EPI_F32, s);
}
void gemm_bf16(const bf16* A, int lda, const bf16* W, int ldw, const float* bias, bf16* out, int Mr, int Nc, int K, hipStream_t s) {
    GemmParams g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = Mr; g.N = Nc; g.K = K; g.out_bf16 = out; g.ldo = Nc; g.bias = bias;
    launch_gemm(g, EPI_BIAS_BF16, s);
}

}  // namespace

extern "C" {

int tld_train_create(const tld_config* cfg, tld_train** out) {
    if (!cfg || !out) return tfail(TLD_ERR_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tfail(TLD_ERR_HIP, "no HIP device: the training engine has no CPU path");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return tfail(TLD_ERR_INVALID, "device_id %d out of range", cfg->device_id);
    if (cfg->embed_dim % 64 || cfg->embed_dim > 1024 || cfg->embed_dim <= 0) return tfail(TLD_ERR_INVALID, "embed_dim must be a multiple of 64 (the head width), <= 1024");
    if (cfg->patch_size <= 0 || cfg->image_size % cfg->patch_size) return tfail(TLD_ERR_INVALID, "image_size must be a multiple of patch_size");
    const int G = cfg->image_size / cfg->patch_size;
    // token counts the attention kernels are built for (forward: launch_attention; backward: tld_train_attn.hip); the reference trains at
    // 256 tokens and fine-tunes at 1024 / 4096 (README.md:23)
    if (!(G * G == 64 || (G * G) % 256 == 0) || G > 64)
        return tfail(TLD_ERR_INVALID, "the training step supports 64 tokens or a multiple of 256 up to 4096 (image_size / patch_size = 8, 16, 32, 64); got %d", G * G);
    if (cfg->n_channels * cfg->patch_size * cfg->patch_size > 64) return tfail(TLD_ERR_INVALID, "patch_dim must be <= 64");
    if (cfg->noise_embed_dims % 2 || cfg->max_batch <= 0 || cfg->n_layers <= 0) return tfail(TLD_ERR_INVALID, "bad configuration");
    DevGuard dg(cfg->device_id);
    tld_train* e = new tld_train();
    e->cfg = *cfg;
    e->d = cfg->embed_dim; e->L = cfg->n_layers; e->H = e->d / 64; e->G = G; e->N = G * G; e->S = cfg->image_size; e->C = cfg->n_channels;
    e->pd = e->C * cfg->patch_size * cfg->patch_size; e->hid = e->d * cfg->mlp_multiplier; e->ne = cfg->noise_embed_dims; e->text = cfg->text_emb_size;
    e->B = cfg->max_batch;
    e->tn_wgrad = !(getenv("TLD_TRAIN_TN_WGRAD") && atoi(getenv("TLD_TRAIN_TN_WGRAD")) == 0);
    const int d = e->d, hid = e->hid, pd = e->pd;
    // canonical order: Denoiser.named_parameters() of the reference (tld/denoiser.py:85-114; pinned by tests/test_train_host.py)
    add_t(e, "fourier_feats.1.weight", (int64_t)d * e->ne, &e->ff1w); add_t(e, "fourier_feats.1.bias", d, &e->ff1b);
    add_t(e, "fourier_feats.3.weight", (int64_t)d * d, &e->ff3w); add_t(e, "fourier_feats.3.bias", d, &e->ff3b);
    const std::string blk = "denoiser_trans_block.";
    add_t(e, blk + "patchify_and_embed.0.weight", (int64_t)pd * pd, &e->cvw); add_t(e, blk + "patchify_and_embed.0.bias", pd, &e->cvb);
    add_t(e, blk + "patchify_and_embed.2.weight", pd, &e->l1w); add_t(e, blk + "patchify_and_embed.2.bias", pd, &e->l1b);
    add_t(e, blk + "patchify_and_embed.3.weight", (int64_t)d * pd, &e->liw); add_t(e, blk + "patchify_and_embed.3.bias", d, &e->lib);
    add_t(e, blk + "patchify_and_embed.4.weight", d, &e->l2w); add_t(e, blk + "patchify_and_embed.4.bias", d, &e->l2b);
    add_t(e, blk + "pos_embed.weight", (int64_t)e->N * d, &e->pos);
    e->lp.resize(e->L);
    for (int i = 0; i < e->L; ++i) {
        const std::string p = blk + "decoder_blocks." + std::to_string(i) + ".";
        LayerP& q = e->lp[i];
        add_t(e, p + "self_attention.qkv_linear.weight", (int64_t)3 * d * d, &q.qkv);
        add_t(e, p + "cross_attention.kv_linear.weight", (int64_t)2 * d * d, &q.kv);
        add_t(e, p + "cross_attention.q_linear.weight", (int64_t)d * d, &q.q);
        add_t(e, p + "mlp.mlp.0.weight", (int64_t)hid * d, &q.up_w); add_t(e, p + "mlp.mlp.0.bias", hid, &q.up_b);
        add_t(e, p + "mlp.mlp.1.weight", (int64_t)hid * 9, &q.dw_w); add_t(e, p + "mlp.mlp.1.bias", hid, &q.dw_b);
        add_t(e, p + "mlp.mlp.3.weight", (int64_t)d * hid, &q.down_w); add_t(e, p + "mlp.mlp.3.bias", d, &q.down_b);
        add_t(e, p + "norm1.weight", d, &q.n1w); add_t(e, p + "norm1.bias", d, &q.n1b);
        add_t(e, p + "norm2.weight", d, &q.n2w); add_t(e, p + "norm2.bias", d, &q.n2b);
        add_t(e, p + "norm3.weight", d, &q.n3w); add_t(e, p + "norm3.bias", d, &q.n3b);
    }
    add_t(e, blk + "out_proj.0.weight", (int64_t)pd * d, &e->outw); add_t(e, blk + "out_proj.0.bias", pd, &e->outb);
    add_t(e, "norm.weight", d, &e->nw); add_t(e, "norm.bias", d, &e->nb);
    add_t(e, "label_proj.weight", (int64_t)d * e->text, &e->lbw); add_t(e, "label_proj.bias", d, &e->lbb);

    const size_t B = e->B, M = B * e->N;
    if (M * (size_t)hid * 2 >= ((size_t)1 << 32)) { delete e; return tfail(TLD_ERR_INVALID, "max_batch too large: the MLP hidden activation must stay below 4 GiB"); }
    auto cleanup = [&](int rc) { for (void* p : e->allocs) hipFree(p); delete e; return rc; };
    auto alloc_all = [&]() -> int {
        DALLOC(e->angular, e->ne / 2); DALLOC(e->zero_bias, 3 * hid > 4096 ? 3 * hid : 4096);
        DALLOC(e->sinb, B * e->ne); DALLOC(e->h1, B * d); DALLOC(e->g1v, B * d); DALLOC(e->ycat, B * 2 * d); DALLOC(e->y, B * 2 * d);
        DALLOC(e->dy, B * 2 * d); DALLOC(e->dycat, B * 2 * d); DALLOC(e->dg1, B * d); DALLOC(e->dkv_all, (size_t)e->L * B * 2 * 2 * d); DALLOC(e->kvc_all, (size_t)e->L * B * 2 * 2 * d);
        DALLOC(e->dy_parts, (size_t)e->L * B * 2 * d); DALLOC(e->yst, B * 2);
        DALLOC(e->p16, M * pd); DALLOC(e->p16n, M * pd); DALLOC(e->e, M * d); DALLOC(e->patches, M * pd); DALLOC(e->de, M * d);
        DALLOC(e->dpn, M * pd); DALLOC(e->dp16, M * pd); DALLOC(e->est1, M); DALLOC(e->est2, M); DALLOC(e->xfin, M * d);
        DALLOC(e->dout, M * pd); DALLOC(e->row_loss, M); DALLOC(e->io, 4);
        const size_t wide = hid > 3 * d ? hid : 3 * d;
        DALLOC(e->gx, M * d); DALLOC(e->gxb, M * d); e->tr_rows = M + 4096; DALLOC(e->T1, e->tr_rows * wide); DALLOC(e->T2, e->tr_rows * wide); DALLOC(e->dbig, M * hid);
        DALLOC(e->dsmall, M * 3 * d); DALLOC(e->dsmall2, M * d); DALLOC(e->attn_stats, 2 * M * e->H); DALLOC(e->scr, (size_t)pd * d + 64);
        const size_t nchunk = (M + 255) / 256;
        size_t need = nchunk * 2 * (size_t)wide;                              // LN / colsum partials
        if (((M + 31) / 32) * 2 * (size_t)d > need) need = ((M + 31) / 32) * 2 * (size_t)d;   // LayerNorm-backward partials (32-row workgroups)
        if (B * e->G * 10 * (size_t)hid > need) need = B * e->G * 10 * (size_t)hid;          // depthwise weight-gradient partials (per sample and image row)
        if (((M + 63) / 64) * (size_t)wide > need) need = ((M + 63) / 64) * (size_t)wide;     // column-sum partials (64-row chunks)
        e->splitk_floats = 8 * (size_t)wide * d;
        DALLOC(e->splitk, e->splitk_floats);
        if (((M + 63) / 64) * (size_t)pd * d > need) need = ((M + 63) / 64) * (size_t)pd * d;    // tall weight-gradient partials (64-row chunks)
        e->part_floats = need;
        DALLOC(e->part, need);
        e->lb.resize(e->L);
        for (int i = 0; i < e->L; ++i) {
            LayerB& q = e->lb[i];
            DALLOC(q.wqkv, 3 * d * d); DALLOC(q.wqkv_t, 3 * d * d); DALLOC(q.wq, d * d); DALLOC(q.wq_t, d * d);
            DALLOC(q.wup, hid * d); DALLOC(q.wup_t, hid * d); DALLOC(q.wdown, hid * d); DALLOC(q.wdown_t, hid * d);
            DALLOC(q.x1, M * d); DALLOC(q.x2, M * d); DALLOC(q.x3, M * d); DALLOC(q.st1, M); DALLOC(q.st2, M); DALLOC(q.st3, M);
            DALLOC(q.a1, M * d); DALLOC(q.a2, M * d); DALLOC(q.a3, M * d); DALLOC(q.qk, M * 2 * d); DALLOC(q.vt, M * d); DALLOC(q.att, M * d);
            DALLOC(q.qc, M * d); DALLOC(q.cr, M * d); DALLOC(q.h, M * hid); DALLOC(q.hc, M * hid); DALLOC(q.gl, M * hid); DALLOC(q.o, M * d);
            q.kvc = e->kvc_all + (size_t)i * B * 2 * 2 * d; DALLOC(q.p0, M * e->H); DALLOC(q.dww_t, 9 * hid);
        }
        return 0;
    };
    if (int rc = alloc_all()) return cleanup(rc);
    if (hipMemset(e->zero_bias, 0, (size_t)(3 * hid > 4096 ? 3 * hid : 4096) * 4) != hipSuccess) return cleanup(tfail(TLD_ERR_HIP, "hipMemset failed"));
    // angular_speeds = 2 pi exp(linspace(log 1, log 1000, ne / 2))   (tld/transformer_blocks.py:11-15; float32 arithmetic as torch does it)
    {
        const int half = e->ne / 2;
        std::vector<float> a(half);
        const float lo = logf(1.0f), hiv = logf(1000.0f);
        const float step = half > 1 ? (hiv - lo) / (float)(half - 1) : 0.f;
        for (int k = 0; k < half; ++k) {
            // torch.linspace fills the upper half from the end (hi - step * (n - 1 - k)) for symmetry
            const float v = k < half / 2 ? lo + step * (float)k : hiv - step * (float)(half - 1 - k);
            a[k] = 2.0f * 3.14159265358979323846f * expf(v);
        }
        if (hipMemcpy(e->angular, a.data(), half * 4, hipMemcpyHostToDevice) != hipSuccess) return cleanup(tfail(TLD_ERR_HIP, "hipMemcpy failed"));
    }
    *out = e;
    return TLD_OK;
}

int64_t tld_train_param_count(const tld_train* e) { return e ? e->nparam : 0; }
int32_t tld_train_tensor_count(const tld_train* e) { return e ? (int32_t)e->layout.size() : 0; }

int tld_train_param_layout(const tld_train* e, int32_t index, char* key_out, int32_t key_cap, int64_t* offset, int64_t* numel) {
    if (!e || index < 0 || index >= (int32_t)e->layout.size() || !key_out || key_cap <= 0) return tfail(TLD_ERR_INVALID, "bad argument");
    const Tensor& t = e->layout[index];
    snprintf(key_out, (size_t)key_cap, "%s", t.key.c_str());
    if (offset) *offset = t.off;
    if (numel) *numel = t.numel;
    return TLD_OK;
}

/* The sinusoid buffer "fourier_feats.0.angular_speeds" is a registered buffer of the reference, not a parameter; a checkpoint's
 * values can be installed here (host fp32 [noise_embed_dims / 2]); the default is the constructor's formula. */
int tld_train_set_angular_speeds(tld_train* e, const float* host, int32_t n) {
    if (!e || !host || n != e->ne / 2) return tfail(TLD_ERR_SHAPE, "angular_speeds must have %d entries", e ? e->ne / 2 : 0);
    DevGuard dg(e->cfg.device_id);
    HIP_TRY(hipMemcpy(e->angular, host, (size_t)n * 4, hipMemcpyHostToDevice));
    return TLD_OK;
}

int tld_train_bind(tld_train* e, float* params, float* grads) {
    if (!e || !params || !grads) return tfail(TLD_ERR_INVALID, "null argument");
    e->params = params; e->grads = grads; e->weights_fresh = false;
    return TLD_OK;
}

int tld_train_refresh_weights(tld_train* e, void* hip_stream) {
    if (!e || !e->params) return tfail(TLD_ERR_STATE, "tld_train_bind first");
    DevGuard dg(e->cfg.device_id);
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    const int d = e->d, hid = e->hid;
    auto both = [&](int64_t off, int R, int Cc, bf16* w, bf16* wt) {       // [R, C] fp32 -> bf16 copy and bf16 transpose [C, R]
        hipLaunchKernelGGL((transpose_to_bf16<float>), dim3((Cc + 31) / 32, (R + 31) / 32), dim3(256), 0, s, e->params + off, Cc, wt, R, R, Cc, 1, w);
    };
    for (int i = 0; i < e->L; ++i) {
        both(e->lp[i].qkv, 3 * d, d, e->lb[i].wqkv, e->lb[i].wqkv_t);
        both(e->lp[i].q, d, d, e->lb[i].wq, e->lb[i].wq_t);
        both(e->lp[i].up_w, hid, d, e->lb[i].wup, e->lb[i].wup_t);
        both(e->lp[i].down_w, d, hid, e->lb[i].wdown, e->lb[i].wdown_t);
        hipLaunchKernelGGL(dw_tapmajor_kernel, g1((size_t)hid * 9), dim3(256), 0, s, e->params + e->lp[i].dw_w, e->lb[i].dww_t, hid);
    }
    HIP_TRY(hipGetLastError());
    e->weights_fresh = true;
    return TLD_OK;
}

int tld_train_forward_backward(tld_train* e, const float* x_noisy, const float* noise_level, const float* label, const float* target,
                               int32_t batch, float* loss_out, float* pred_out, void* hip_stream) {
    return tld_train_forward_backward_cb(e, x_noisy, noise_level, label, target, batch, loss_out, pred_out, hip_stream, nullptr, nullptr);
}

int tld_train_forward_backward_cb(tld_train* e, const float* x_noisy, const float* noise_level, const float* label, const float* target,
                                  int32_t batch, float* loss_out, float* pred_out, void* hip_stream, tld_grad_ready_fn grad_ready, void* user) {
    if (!e || !x_noisy || !noise_level || !label || !target || !loss_out || !pred_out) return tfail(TLD_ERR_INVALID, "null argument");
    if (!e->params) return tfail(TLD_ERR_STATE, "tld_train_bind first");
    if (batch <= 0 || batch > e->B) return tfail(TLD_ERR_INVALID, "batch %d outside [1, max_batch = %d]", batch, e->B);
    DevGuard dg(e->cfg.device_id);
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    if (!e->weights_fresh) { if (int rc = tld_train_refresh_weights(e, hip_stream)) return rc; }
    const int d = e->d, hid = e->hid, pd = e->pd, N = e->N, H = e->H, B = batch, M = B * N, G = e->G;
    float* P = e->params; float* Gd = e->grads;
    const dim3 blk(256);
    const int nchunk = (M + 255) / 256;
    const int dw_rows = dwconv_band_rows(G);
    const dim3 dw_grid(B * (hid / 64) * ((G + dw_rows - 1) / dw_rows));
    const size_t dw_lds = dwconv_lds_bytes(G);
    {
        static PerDeviceOnce once;
        once.run([&] { hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    }
    const float inv_numel = 1.0f / (float)((size_t)B * e->C * e->S * e->S);
    // the three small fp32 products on the tiled kernel (see tld_train_kernels.h)
    auto lin_fwd = [&](const float* in, int ldi, const float* W, const float* bias, float* out, int ldo, int R, int Nn, int K, float* pre, int gelu) {
        hipLaunchKernelGGL(tiled_f32_kernel, dim3((Nn + 31) / 32, (R + 31) / 32), dim3(256), 0, s, in, (long)ldi, 1L, W, (long)K, 1L, bias, out, ldo, R, Nn, K, pre, gelu, 0);
    };
    auto lin_dx = [&](const float* dyp, int ldy, const float* W, float* dx, int ldx, int R, int Nn, int K, int acc) {      // dx[r, k] (+)= sum_n dy[r, n] W[n, k]
        hipLaunchKernelGGL(tiled_f32_kernel, dim3((K + 31) / 32, (R + 31) / 32), dim3(256), 0, s, dyp, (long)ldy, 1L, W, 1L, (long)K, (const float*)nullptr, dx, ldx, R, K, Nn,
                           (float*)nullptr, 0, acc);
    };
    auto lin_dw = [&](const float* dyp, int ldy, const float* x, int ldx, float* dW, float* db, int R, int Nn, int K) {     // dW[n, k] = sum_r dy[r, n] x[r, k]; db[n]
        hipLaunchKernelGGL(tiled_f32_kernel, dim3((K + 31) / 32, (Nn + 31) / 32), dim3(256), 0, s, dyp, 1L, (long)ldy, x, 1L, (long)ldx, (const float*)nullptr, dW, K, Nn, K, R,
                           (float*)nullptr, 0, 0);
        if (db) hipLaunchKernelGGL(small_colsum, g1(Nn), dim3(256), 0, s, dyp, ldy, db, R, Nn, 0);
    };

    // ================================================ forward ================================================
    // conditioning (tld/denoiser.py:105-122): sinusoid -> Linear -> GELU -> Linear | label_proj -> stack -> LayerNorm
    hipLaunchKernelGGL(sinusoid_kernel, g1((size_t)B * e->ne / 2), blk, 0, s, noise_level, e->angular, e->sinb, B, e->ne / 2);
    lin_fwd(e->sinb, e->ne, P + e->ff1w, P + e->ff1b, e->g1v, d, B, d, e->ne, e->h1, 1);
    lin_fwd(e->g1v, d, P + e->ff3w, P + e->ff3b, e->ycat, 2 * d, B, d, d, nullptr, 0);
    lin_fwd(label, e->text, P + e->lbw, P + e->lbb, e->ycat + d, 2 * d, B, d, e->text, nullptr, 0);
    hipLaunchKernelGGL((ln_fwd_kernel<float>), dim3((2 * B + 3) / 4), blk, 0, s, e->ycat, P + e->nw, P + e->nb, (bf16*)nullptr, e->y, e->yst,
                       (const float*)nullptr, 1, 2 * B, d);
    // patch embedding (tld/denoiser.py:34-45,75-77)
    {
        EmbedTrain q{};
        q.x = x_noisy; q.conv_w = P + e->cvw; q.conv_b = P + e->cvb; q.ln1_w = P + e->l1w; q.ln1_b = P + e->l1b; q.lin_w = P + e->liw; q.lin_b = P + e->lib;
        q.ln2_w = P + e->l2w; q.ln2_b = P + e->l2b; q.pos = P + e->pos; q.p = e->p16; q.pn = e->p16n; q.st1 = e->est1; q.e = e->e; q.st2 = e->est2;
        q.x0 = e->lb[0].x1; q.B = B; q.C = e->C; q.S = e->S; q.patch = e->cfg.patch_size; q.grid = G; q.pd = pd; q.d = d;
        const size_t lwb = (size_t)pd * d * 4;
        if (lwb <= 65536) {
            const int nwg = (M + 3) / 4 < 4 * device_cu_count() ? (M + 3) / 4 : 4 * device_cu_count();
            if (d <= 256) hipLaunchKernelGGL((embed_fwd_lds_kernel<1>), dim3(nwg), blk, lwb, s, q);
            else if (d <= 512) hipLaunchKernelGGL((embed_fwd_lds_kernel<2>), dim3(nwg), blk, lwb, s, q);
            else if (d <= 768) hipLaunchKernelGGL((embed_fwd_lds_kernel<3>), dim3(nwg), blk, lwb, s, q);
            else hipLaunchKernelGGL((embed_fwd_lds_kernel<4>), dim3(nwg), blk, lwb, s, q);
        } else {
            hipLaunchKernelGGL(embed_fwd_kernel, dim3((M + 3) / 4), blk, 0, s, q);
        }
    }
    // x_out = x_in + delta [; a_out = LN(x_out), stats]   (delta == nullptr: LayerNorm of x_in alone)
    auto resid_ln = [&](const bf16* x_in, const bf16* delta, bf16* x_out, const float* gamma, const float* beta, bf16* a_out, float2* st) {
        const dim3 grid((M + 3) / 4);
        if (d == 768) hipLaunchKernelGGL((resid_add_ln_q4_kernel<3>), grid, blk, 0, s, x_in, delta, x_out, gamma, beta, a_out, st, M);
        else if (d == 512) hipLaunchKernelGGL((resid_add_ln_q4_kernel<2>), grid, blk, 0, s, x_in, delta, x_out, gamma, beta, a_out, st, M);
        else if (d == 256) hipLaunchKernelGGL((resid_add_ln_q4_kernel<1>), grid, blk, 0, s, x_in, delta, x_out, gamma, beta, a_out, st, M);
        else if (d == 1024) hipLaunchKernelGGL((resid_add_ln_q4_kernel<4>), grid, blk, 0, s, x_in, delta, x_out, gamma, beta, a_out, st, M);
        else if (delta) hipLaunchKernelGGL(resid_add_ln_kernel, grid, blk, 0, s, x_in, delta, x_out, gamma, beta, a_out, st, M, d);
        else hipLaunchKernelGGL((ln_fwd_kernel<bf16>), grid, blk, 0, s, x_in, gamma, beta, a_out, (float*)nullptr, st, (const float*)nullptr, 1, M, d);
    };
    // (k | v) of the two conditioning tokens for every block in one launch (tld/transformer_blocks.py:66-68): the blocks' parameters are laid out
    // identically, so block i's kv_linear.weight sits i block-strides after block 0's
    const long blk_stride = e->L > 1 ? (long)(e->lp[1].kv - e->lp[0].kv) : 0;
    const long kv_stride = (long)e->B * 2 * 2 * d;
    hipLaunchKernelGGL(tiled_f32_kernel, dim3((2 * d + 31) / 32, (2 * B + 31) / 32, e->L), dim3(256), 0, s, e->y, (long)d, 1L, P + e->lp[0].kv, (long)d, 1L,
                       (const float*)nullptr, e->kvc_all, 2 * d, 2 * B, 2 * d, d, (float*)nullptr, 0, 0, 0L, blk_stride, kv_stride);
    for (int i = 0; i < e->L; ++i) {
        LayerB& b = e->lb[i]; const LayerP& p = e->lp[i];
        // x = x + SA(LN1 x)   (tld/transformer_blocks.py:51-59,136)
        if (i == 0) resid_ln(b.x1, (const bf16*)nullptr, (bf16*)nullptr, P + p.n1w, P + p.n1b, b.a1, b.st1);      // (blocks > 0: LN1 rides on the previous block's last residual add)
        {
            GemmParams g{};
            g.A = b.a1; g.lda = d; g.W = b.wqkv; g.ldw = d; g.M = M; g.N = 3 * d; g.K = d; g.out_bf16 = b.qk; g.ldo = 2 * d; g.vt = b.vt; g.ntok = N; g.d = d;
            launch_gemm(g, EPI_QKV, s);
        }
        launch_attention(b.qk, b.vt, b.att, B, N, H, s);
        resid_ln(b.x1, b.att, b.x2, P + p.n2w, P + p.n2b, b.a2, b.st2);
        // x = x + CA(LN2 x, y)   (:62-72,137)
        gemm_bf16(b.a2, d, b.wq, d, e->zero_bias, b.qc, M, d, d, s);
        hipLaunchKernelGGL(cross_fwd_kernel, dim3(B * H), blk, 0, s, b.qc, b.kvc, b.cr, b.p0, N, d);
        resid_ln(b.x2, b.cr, b.x3, P + p.n3w, P + p.n3b, b.a3, b.st3);
        // x = x + MLPSepConv(LN3 x)   (:89-113,138)
        gemm_bf16(b.a3, d, b.wup, d, P + p.up_b, b.h, M, hid, d, s);
        hipLaunchKernelGGL(dwconv_kernel, dw_grid, blk, dw_lds, s, b.h, b.dww_t, P + p.dw_b, b.hc, b.gl, B, G, hid, 0, dw_rows);
        gemm_bf16(b.gl, hid, b.wdown, hid, P + p.down_b, b.o, M, d, hid, s);
        bf16* xnext = i + 1 < e->L ? e->lb[i + 1].x1 : e->xfin;
        if (i + 1 < e->L) resid_ln(b.x3, b.o, xnext, P + e->lp[i + 1].n1w, P + e->lp[i + 1].n1b, e->lb[i + 1].a1, e->lb[i + 1].st1);      // + the next block's LN1
        else resid_ln(b.x3, b.o, xnext, (const float*)nullptr, (const float*)nullptr, (bf16*)nullptr, (float2*)nullptr);
    }
    // out_proj + unpatchify + MSE (tld/denoiser.py:47-52,72,82; tld/train.py:167)
    hipLaunchKernelGGL(tail_fwd_kernel, dim3((M + 3) / 4), blk, 0, s, e->xfin, P + e->outw, P + e->outb, target, pred_out, e->dout, e->row_loss, B, e->C, e->S,
                       e->cfg.patch_size, G, pd, d, inv_numel);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), blk, 0, s, e->row_loss, M, inv_numel, loss_out);

    // ================================================ backward ===============================================
    auto reduce = [&](int nparts, size_t stride, size_t part_off, float* dst, int n, int acc) {
        hipLaunchKernelGGL(reduce_partials, dim3((n + 15) / 16), dim3(1024), 0, s, e->part + part_off, nparts, stride, dst, n, acc);
    };
    // dW[Nn, K] = dy^T x over the M rows (dy fp32 [M, Nn], Nn = patch_dim or smaller; x [M, K]) -> dst (fixed-order sum of per-chunk partials)
    auto tall_dw = [&](const float* dyp, int Nn, auto xp, int K, float* dst) {
        using TX = std::remove_cv_t<std::remove_pointer_t<decltype(xp)>>;
        const int nc64 = (M + 63) / 64;
        if ((Nn == 16 || Nn == 32 || Nn == 64) && (size_t)nc64 * Nn * K <= e->part_floats) {
            const dim3 grid((K + 255) / 256, nc64);
            if (Nn == 16) hipLaunchKernelGGL((tall_dw_cols_partial<TX, 16>), grid, blk, 0, s, dyp, xp, K, M, 64, e->part);
            else if (Nn == 32) hipLaunchKernelGGL((tall_dw_cols_partial<TX, 32>), grid, blk, 0, s, dyp, xp, K, M, 64, e->part);
            else hipLaunchKernelGGL((tall_dw_cols_partial<TX, 64>), grid, blk, 0, s, dyp, xp, K, M, 64, e->part);
            reduce(nc64, (size_t)Nn * K, 0, dst, Nn * K, 0);
        } else {
            hipLaunchKernelGGL((tall_dw_partial<TX>), dim3((Nn * K + 255) / 256, nchunk), blk, 0, s, dyp, Nn, xp, K, M, 256, e->part);
            reduce(nchunk, (size_t)Nn * K, 0, dst, Nn * K, 0);
        }
    };
    auto ln_bwd_rows = [&](auto dyp, auto xp, const float2* st, const float* gamma, float* dx, int acc, float* dgamma, float* dbeta, int rows, int width,
                           bf16* dxb = nullptr) {
        using TDY = std::remove_cv_t<std::remove_pointer_t<decltype(dyp)>>;
        using TX = std::remove_cv_t<std::remove_pointer_t<decltype(xp)>>;
        const int nb = (rows + 31) / 32;                                  // 32 rows per workgroup: >= 1024 workgroups at the training batch
        if (width == 768) hipLaunchKernelGGL((ln_bwd_q4_kernel<TDY, TX, 3>), dim3(nb), blk, 0, s, dyp, xp, st, gamma, dx, acc, e->part, 32, rows, dxb);
        else if (width == 512) hipLaunchKernelGGL((ln_bwd_q4_kernel<TDY, TX, 2>), dim3(nb), blk, 0, s, dyp, xp, st, gamma, dx, acc, e->part, 32, rows, dxb);
        else if (width == 256) hipLaunchKernelGGL((ln_bwd_q4_kernel<TDY, TX, 1>), dim3(nb), blk, 0, s, dyp, xp, st, gamma, dx, acc, e->part, 32, rows, dxb);
        else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX>), dim3(nb), blk, 0, s, dyp, xp, st, gamma, dx, acc, e->part, 32, rows, width, dxb);
        if (dbeta == dgamma + width) reduce(nb, 2 * (size_t)width, 0, dgamma, 2 * width, 0);       // (weight, bias) are neighbours in the flat vector: one launch
        else {
            reduce(nb, 2 * (size_t)width, 0, dgamma, width, 0);
            reduce(nb, 2 * (size_t)width, width, dbeta, width, 0);
        }
    };
    auto colsum = [&](auto ap, int rows, int cols, float* dst) {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(ap)>>;
        if (cols % 4 == 0 && rows >= 4096) {        // four columns per thread, 64-row chunks (the chunk count feeds the 64 part-lanes of the reduction)
            const int nb = (rows + 63) / 64;
            hipLaunchKernelGGL((colsum4_partial<T>), dim3((cols / 4 + 255) / 256, nb), blk, 0, s, ap, rows, cols, 64, e->part);
            reduce(nb, (size_t)cols, 0, dst, cols, 0);
            return;
        }
        const int nb = (rows + 255) / 256;
        hipLaunchKernelGGL((colsum_partial<T>), dim3((cols + 255) / 256, nb), blk, 0, s, ap, rows, cols, 256, e->part);
        reduce(nb, (size_t)cols, 0, dst, cols, 0);
    };
    auto transpose = [&](auto inp, int rows, int cols, bf16* outp) {        // [rows, cols] -> [cols, rows]
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(inp)>>;
        hipLaunchKernelGGL((transpose_to_bf16<T>), dim3((cols + 31) / 32, (rows + 31) / 32), blk, 0, s, inp, cols, outp, rows, rows, cols);
    };
    // dW[Nout, Kin] = dY^T X with dY [M, Nout], X [M, Kin] (bf16): both operands transposed so that the contraction (M) is contiguous.
    // The output is small and the contraction long, so the rows are cut into `sk` runs (split-K): the transposes write the stacked
    // operands [split][Nout | Kin][M / sk], ONE GEMM launch multiplies every split with its own W block (GemmParams::w_batch_rows)
    // into fp32 partials [split][Nout][Kin], and a fixed-order sum finishes (bit-reproducible).
    auto weight_grad = [&](const bf16* dY, int Nout, const bf16* X, int Kin, float* dW) {
        int sk = 1;
        if (Nout % 256 == 0) while (sk < 8 && (M / (sk * 2)) % 128 == 0 && M / (sk * 2) >= 1024) sk *= 2;
        int ms = M / sk;
        // Round 3: any split count, runs padded to a multiple of 128 rows (zero rows contribute nothing).  With 256 x 256 tiles the launch has
        // sk (Nout / 256) (Kin / 256) workgroups; pick the sk whose last round is fullest (e.g. 7 x 12 x 3 = 252 of 256 CUs for the
        // up-projection weight instead of 8 x 12 x 6 = 576 128-wide tiles = 2.25 rounds), discounted by the padding.
        if (Nout % 256 == 0 && Kin % 256 == 0 && M % 64 == 0 && M >= 4096) {
            const int ncu = device_cu_count();
            const long per = (long)(Nout / 256) * (Kin / 256);
            double best = 0.0; int bsk = 0, bms = 0;
            for (int c = 2; c <= 32; ++c) {
                const int mp = ((M + c - 1) / c + 127) / 128 * 128;
                if (mp < 1024 || (long)(c - 1) * mp >= M) continue;                                     // runs too short / last run empty
                if ((size_t)c * Nout * Kin > e->splitk_floats || (size_t)c * mp > e->tr_rows) continue;   // workspace
                const long tiles = per * c, rounds = (tiles + ncu - 1) / ncu;
                const double eff = (double)tiles / (double)(rounds * ncu) * ((double)M / ((double)c * mp));
                if (eff > best + 1e-9) { best = eff; bsk = c; bms = mp; }
            }
            if (bsk) { sk = bsk; ms = bms; }
        }
        if (e->tn_wgrad && wgrad_tn(dY, Nout, X, Kin, M, dW, e->splitk, e->splitk_floats, s)) return;
        auto tr = [&](const bf16* src, int cols, bf16* dst) {
            if (M % 64 == 0 && cols % 64 == 0 && ms % 64 == 0)
                hipLaunchKernelGGL(transpose_bf16_64, dim3(cols / 64, sk * ms / 64), blk, 0, s, src, cols, dst, ms, M, cols, sk);
            else
                hipLaunchKernelGGL((transpose_to_bf16<bf16>), dim3((cols + 31) / 32, (M + 31) / 32), blk, 0, s, src, cols, dst, ms, M, cols, sk);
        };
        tr(dY, Nout, e->T1);
        tr(X, Kin, e->T2);
        if (sk == 1) { gemm_f32(e->T1, M, e->T2, M, dW, Nout, Kin, M, s); return; }      // (ms == M here)
        GemmParams g{};
        g.A = e->T1; g.lda = ms; g.W = e->T2; g.ldw = ms; g.M = sk * Nout; g.N = Kin; g.K = ms; g.c_f32 = e->splitk; g.ldc = Kin;
        g.w_batch_rows = Nout; g.w_batch_stride_bytes = (unsigned)((size_t)Kin * ms * 2);
        launch_gemm(g, EPI_F32, s);
        if ((Nout * Kin) % 4 == 0) hipLaunchKernelGGL(sum_slices, dim3((unsigned)(((size_t)Nout * Kin / 4 + 255) / 256)), blk, 0, s, e->splitk, sk, (size_t)Nout * Kin, dW, (size_t)Nout * Kin / 4);
        else hipLaunchKernelGGL(reduce_partials, dim3((Nout * Kin + 15) / 16), dim3(1024), 0, s, e->splitk, sk, (size_t)Nout * Kin, dW, Nout * Kin, 0);
    };

    // out_proj: gx = dout Wout;  dWout = dout^T x_final;  dbout
    // gx and its bf16 copy (the top block's GEMM operand)
    if (pd == 16) hipLaunchKernelGGL((tail_dx4_kernel<16>), g1((size_t)M * d / 4), blk, 0, s, e->dout, P + e->outw, e->gx, e->gxb, M, d);
    else if (pd == 32) hipLaunchKernelGGL((tail_dx4_kernel<32>), g1((size_t)M * d / 4), blk, 0, s, e->dout, P + e->outw, e->gx, e->gxb, M, d);
    else if (pd == 64) hipLaunchKernelGGL((tail_dx4_kernel<64>), g1((size_t)M * d / 4), blk, 0, s, e->dout, P + e->outw, e->gx, e->gxb, M, d);
    else hipLaunchKernelGGL(tail_dx_kernel, g1((size_t)M * d), blk, 0, s, e->dout, P + e->outw, e->gx, e->gxb, M, pd, d);
    tall_dw(e->dout, pd, (const bf16*)e->xfin, d, Gd + e->outw);
    colsum(e->dout, M, pd, Gd + e->outb);

    for (int i = e->L - 1; i >= 0; --i) {
        LayerB& b = e->lb[i]; const LayerP& p = e->lp[i];
        // ---- MLP: o = g Wdown^T + b;  g = GELU(hc);  hc = dwconv(h);  h = a3 Wup^T + b;  a3 = LN3(x3)
        // (e->gxb = bf16(e->gx): written by whoever completed gx -- tail_dx_kernel for the top block, the LayerNorm-1 backward of the block above otherwise)
        colsum(e->gxb, M, d, Gd + p.down_b);
        weight_grad(e->gxb, d, b.gl, hid, Gd + p.down_w);
        gemm_bf16(e->gxb, d, b.wdown_t, d, e->zero_bias, e->dbig, M, hid, d, s);                         // dg = go Wdown
        const bool dw_fused = G <= 16 && N >= 176 && hid % 64 == 0;
        if (dw_fused) {      // GELU' multiply, depthwise weight-gradient partials and input gradient in one pass (both images of a (sample, 64-channel chunk) in LDS)
            hipLaunchKernelGGL(dwconv_bwd_img_kernel, dim3(B * (hid / 64)), blk, (size_t)2 * N * 128, s, e->dbig, b.hc, b.h, b.dww_t, b.gl, e->part, G, hid);   // dh -> b.gl (its forward value is consumed)
            hipLaunchKernelGGL(dwconv_wgrad_reduce, dim3((hid * 11 + 63) / 64), dim3(1024), 0, s, e->part, Gd + p.dw_w, Gd + p.dw_b, B, hid, 11, Gd + p.up_b);   // + the up-projection's bias gradient
        } else {
            hipLaunchKernelGGL(gelu_bwd_kernel, g1((size_t)M * hid / 8), blk, 0, s, e->dbig, b.hc, e->dbig, (size_t)M * hid / 8);      // dhc (in place); b.hc = GELU'(pre-activation)
            hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3((hid + 255) / 256, B * G), blk, 0, s, e->dbig, b.h, e->part, G, hid);
            hipLaunchKernelGGL(dwconv_wgrad_reduce, dim3((hid * 10 + 63) / 64), dim3(1024), 0, s, e->part, Gd + p.dw_w, Gd + p.dw_b, B * G, hid);
            hipLaunchKernelGGL(dwconv_kernel, dw_grid, blk, dw_lds, s, e->dbig, b.dww_t, (const float*)nullptr, b.gl, (bf16*)nullptr, B, G, hid, 1, dw_rows);   // dh -> b.gl
        }
        if (!dw_fused) colsum(b.gl, M, hid, Gd + p.up_b);
        weight_grad(b.gl, hid, b.a3, d, Gd + p.up_w);
        gemm_bf16(b.gl, hid, b.wup_t, hid, e->zero_bias, e->dsmall2, M, d, hid, s);                     // da3 = dh Wup
        ln_bwd_rows(e->dsmall2, b.x3, b.st3, P + p.n3w, e->gx, 1, Gd + p.n3w, Gd + p.n3b, M, d);
        // ---- cross-attention: cr = CA(qc, kv);  qc = a2 Wq^T;  kv = y Wkv^T;  a2 = LN2(x2)
        float* dkv = e->dkv_all + (size_t)i * kv_stride;
        hipLaunchKernelGGL(cross_bwd_kernel, dim3(B * H), blk, 0, s, e->gx, b.qc, b.kvc, b.p0, e->dsmall2, dkv, N, d);   // dqc -> dsmall2
        lin_dw(dkv, 2 * d, e->y, d, Gd + p.kv, nullptr, 2 * B, 2 * d, d);       // (per block: its gradient range must be complete at the grad_ready call below)
        weight_grad(e->dsmall2, d, b.a2, d, Gd + p.q);
        gemm_bf16(e->dsmall2, d, b.wq_t, d, e->zero_bias, e->dsmall, M, d, d, s);                       // da2 = dqc Wq
        ln_bwd_rows(e->dsmall, b.x2, b.st2, P + p.n2w, e->gx, 1, Gd + p.n2w, Gd + p.n2b, M, d);
        // ---- self-attention: att = SDPA(q, k, v);  qkv = a1 Wqkv^T;  a1 = LN1(x1)
        if (launch_attention_bwd(b.qk, b.vt, b.att, e->gx, e->dsmall, e->attn_stats, B, N, H, s)) return tfail(TLD_ERR_INVALID, "attention backward: unsupported token count %d", N);
        // (dO as a bf16 copy from the LayerNorm-2 backward: 192 -> 189 us, but delta = dO . O from rounded dO moves the worst g15 gradient from 1.86e-2 to 1.92e-2 of a 2e-2 bound: not taken)
        weight_grad(e->dsmall, 3 * d, b.a1, d, Gd + p.qkv);
        gemm_bf16(e->dsmall, 3 * d, b.wqkv_t, 3 * d, e->zero_bias, e->dsmall2, M, d, 3 * d, s);         // da1 = dqkv Wqkv
        ln_bwd_rows(e->dsmall2, b.x1, b.st1, P + p.n1w, e->gx, 1, Gd + p.n1w, Gd + p.n1b, M, d, i > 0 ? e->gxb : nullptr);
        // every gradient of this block is enqueued (its 15 tensors are one contiguous range of the flat vector): the data-parallel
        // reduction of that slice can start now, under the backward of the blocks below
        if (grad_ready) grad_ready(user, p.qkv, (p.n3b + d) - p.qkv);
    }
    // dL/dy = sum over blocks of dkv_i Wkv_i: one batched launch into per-block parts, then a fixed-order sum
    hipLaunchKernelGGL(tiled_f32_kernel, dim3((d + 31) / 32, (2 * B + 31) / 32, e->L), dim3(256), 0, s, e->dkv_all, (long)(2 * d), 1L, P + e->lp[0].kv, 1L, (long)d,
                       (const float*)nullptr, e->dy_parts, d, 2 * B, d, 2 * d, (float*)nullptr, 0, 0, kv_stride, blk_stride, (long)e->B * 2 * d);
    hipLaunchKernelGGL(reduce_partials, dim3((2 * B * d + 15) / 16), dim3(1024), 0, s, e->dy_parts, e->L, (size_t)e->B * 2 * d, e->dy, 2 * B * d, 0);
    // ---- patch embedding: x0 = LN2(e) + pos;  e = pn Wlin^T + b;  pn = LN1(p);  p = conv(x)     (tld/denoiser.py:34-45,75-77)
    hipLaunchKernelGGL(pos_grad_kernel, g1((size_t)N * d), blk, 0, s, e->gx, Gd + e->pos, B, N, d);
    ln_bwd_rows(e->gx, e->e, e->est2, P + e->l2w, e->de, 0, Gd + e->l2w, Gd + e->l2b, M, d);
    colsum(e->de, M, d, Gd + e->lib);
    lin_dx(e->de, d, P + e->liw, e->dpn, pd, M, d, pd, 0);               // dpn = de Wlin
    tall_dw(e->p16n, pd, (const float*)e->de, d, e->scr);                                                                              // dWlin^T [pd, d]
    hipLaunchKernelGGL(transpose_f32_small, g1((size_t)pd * d), blk, 0, s, e->scr, Gd + e->liw, pd, d);
    hipLaunchKernelGGL(ln_small_bwd_kernel, dim3(nchunk), blk, 0, s, e->dpn, e->p16, e->est1, P + e->l1w, e->dp16, e->part, M, pd);
    reduce(nchunk, 2 * (size_t)pd, 0, Gd + e->l1w, pd, 0);
    reduce(nchunk, 2 * (size_t)pd, pd, Gd + e->l1b, pd, 0);
    hipLaunchKernelGGL(patches_kernel, g1((size_t)M * pd), blk, 0, s, x_noisy, e->patches, B, e->C, e->S, e->cfg.patch_size, G);
    tall_dw(e->dp16, pd, (const float*)e->patches, pd, Gd + e->cvw);
    colsum(e->dp16, M, pd, Gd + e->cvb);
    // ---- conditioning: y = LN(stack[nz, lb]);  nz = W3 GELU(W1 sin + b1) + b3;  lb = label_proj(label)     (tld/denoiser.py:105-122)
    ln_bwd_rows(e->dy, e->ycat, e->yst, P + e->nw, e->dycat, 0, Gd + e->nw, Gd + e->nb, 2 * B, d);
    lin_dw(e->dycat + d, 2 * d, label, e->text, Gd + e->lbw, Gd + e->lbb, B, d, e->text);
    lin_dw(e->dycat, 2 * d, e->g1v, d, Gd + e->ff3w, Gd + e->ff3b, B, d, d);
    lin_dx(e->dycat, 2 * d, P + e->ff3w, e->dg1, d, B, d, d, 0);
    hipLaunchKernelGGL(mul_gelu_grad, g1((size_t)B * d), blk, 0, s, e->dg1, e->h1, B * d);
    lin_dw(e->dg1, d, e->sinb, e->ne, Gd + e->ff1w, Gd + e->ff1b, B, d, e->ne);
    if (grad_ready) {       // the ranges around the blocks: conditioning MLP / patch embedding / position table; out_proj, norm, label_proj
        grad_ready(user, 0, e->lp[0].qkv);
        grad_ready(user, e->outw, e->nparam - e->outw);
    }
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

/* optimizer.step() of torch.optim.Adam(model.parameters(), lr) (tld/train.py:87,169) fused with update_ema (tld/train.py:55-58,172) over
 * flat fp32 device vectors; step counts from 1.  ema may be NULL (ranks other than the main process hold no EMA copy, :110-112).
 * grad_scale multiplies the gradient first (1 / world_size after a SUM all-reduce). */
int tld_train_adam_ema(tld_train* e, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema, int64_t numel, float lr,
                       float beta1, float beta2, float eps, int32_t step, float ema_alpha, float grad_scale, void* hip_stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || numel <= 0 || step <= 0) return tfail(TLD_ERR_INVALID, "bad argument");
    DevGuard dg(e ? e->cfg.device_id : 0);
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_ema_kernel, g1((size_t)numel), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, ema, (size_t)numel, lr, beta1, beta2, eps, bc1, bc2,
                       ema_alpha, grad_scale);
    HIP_TRY(hipGetLastError());
    if (e && params == e->params) e->weights_fresh = false;          // the bf16 operand copies are stale now
    return TLD_OK;
}

/* Test hook: backward of softmax(Q K^T / 8) V for `batch` samples x `heads` heads over `ntok` tokens (64, 128 or a multiple of 256).
 * qk [M, 2 d] bf16 (q | k), vt [B, H, 64, ntok] bf16, o [M, d] bf16 (the forward output), g [M, d] fp32 (dL/dO); dqkv [M, 3 d] bf16 out
 * (dq | dk | dv); scratch: 2 * batch * heads * ntok floats (used when ntok > 256).  Device pointers. */
int tld_debug_attention_bwd(const void* qk, const void* vt, const void* o, const float* g, void* dqkv, float* scratch, int32_t batch, int32_t ntok,
                            int32_t heads, void* hip_stream) {
    if (!qk || !vt || !o || !g || !dqkv || batch <= 0 || heads <= 0 || ntok <= 0) return tfail(TLD_ERR_INVALID, "bad argument");
    PtrDeviceGuard guard(qk);
    if (launch_attention_bwd(reinterpret_cast<const bf16*>(qk), reinterpret_cast<const bf16*>(vt), reinterpret_cast<const bf16*>(o), g,
                             reinterpret_cast<bf16*>(dqkv), scratch, batch, ntok, heads, reinterpret_cast<hipStream_t>(hip_stream)))
        return tfail(TLD_ERR_INVALID, "attention backward: unsupported token count %d (or no scratch)", ntok);
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

/* Test hook: the weight-gradient product dW[n_out, k_in] = dY^T X of the training step (dY [rows, n_out], X [rows, k_in] bf16 row-major, fp32 out)
 * on the transposed-operand GEMM; `slices`: fp32 workspace of slice_floats elements for the split-K partial sums.  Device pointers. */
int tld_debug_wgrad(const void* dy, const void* x, float* dw, float* slices, int64_t slice_floats, int32_t rows, int32_t n_out, int32_t k_in, void* hip_stream) {
    if (!dy || !x || !dw || !slices || rows <= 0) return tfail(TLD_ERR_INVALID, "bad argument");
    PtrDeviceGuard guard(dy);
    if (!wgrad_tn(reinterpret_cast<const bf16*>(dy), n_out, reinterpret_cast<const bf16*>(x), k_in, rows, dw, slices, (size_t)slice_floats,
                  reinterpret_cast<hipStream_t>(hip_stream)))
        return tfail(TLD_ERR_SHAPE, "n_out and k_in must be multiples of 256, rows a multiple of 64");
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

int tld_train_destroy(tld_train* e) {
    if (!e) return TLD_OK;
    DevGuard dg(e->cfg.device_id);
    for (void* p : e->allocs) hipFree(p);
    delete e;
    return TLD_OK;
}

}  // extern "C"
