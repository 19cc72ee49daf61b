// tld_engine.hip -- host side of libtld_hip.so: weight packing, workspace, kernel sequencing, C ABI.
//
// One engine = one device + one model.  The forward is a fixed sequence of hand-written kernels
// (7 per decoder block) enqueued on the caller's stream; the sampler prepares every conditioning
// table once (all timesteps, all prompts) and then runs embed -> blocks -> tail -> update per step
// with no host round trip.  See DESIGN.md for the data layout and per-kernel roofline notes.
#include "../../include/tld_hip.h"
#include "tld_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

using namespace tld;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace
namespace tld {
void set_last_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
}  // namespace tld
namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(TLD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// Every ABI entry point runs with the engine's device current and puts the caller's device back on exit: the
// library never changes the calling thread's current HIP device (a model on cuda:1 used from a thread whose
// current device is cuda:0 would otherwise silently redirect the caller's later allocations and launches).
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

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

struct Layer {
    bf16 *qkv_w = nullptr, *up_w = nullptr, *down_w = nullptr;
    bf16 *qkv_wf = nullptr;                               // bf16(gamma1 (.) Wqkv): LayerNorm-1 folded into the QKV GEMM
    float *qkv_c1 = nullptr, *qkv_b1 = nullptr;           // [3d] column sums of qkv_wf; beta1 . Wqkv^T
    bf16 *qkv_wp = nullptr;                               // qkv_wf with its rows permuted to [head][q_h | k_h | v_h] (fused QKV -> attention kernel)
    float *qkv_c1p = nullptr, *qkv_b1p = nullptr;         // the same permutation of qkv_c1 / qkv_b1
    float *up_b = nullptr, *dw_w9c = nullptr, *dw_b = nullptr, *down_b = nullptr;
    float *dw_w9c_half = nullptr, *dw_b_half = nullptr;   // 0.5 x (exact): operands of the fused up-projection epilogue
    uint32_t* dw_wpk = nullptr;                           // the halved taps as packed bf16 pairs [3][4][hid] (EPI_UP_DWCONV2)
    // MX-fp8 GEMM mode (tld_engine_set_gemm_dtype): e4m3 weights + E8M0 block scales [K/128][N][4]
    uint8_t *qkv_w8 = nullptr, *qkv_s8 = nullptr, *up_w8 = nullptr, *up_s8 = nullptr, *down_w8 = nullptr, *down_s8 = nullptr;
    bf16 *up_wf = nullptr;                                // bf16(gamma3 (.) Wup): LayerNorm-3 folded into the up-projection
    float *up_c1 = nullptr, *up_b1 = nullptr;             // [hid] column sums of up_wf; up_b + beta3 . Wup^T
    float *n1_w = nullptr, *n1_b = nullptr, *n2_w = nullptr, *n2_b = nullptr, *n3_w = nullptr, *n3_b = nullptr;
    float *kv_w = nullptr, *q_w = nullptr;   // fp32, conditioning path
};

enum KClass { KC_GEMM_QKV = 0, KC_GEMM_UP, KC_GEMM_DOWN, KC_ATTN, KC_CROSS, KC_DWCONV, KC_LN, KC_EMBED,
              KC_TAIL, KC_UPDATE, KC_COND, KC_COUNT };

}  // namespace

struct tld_engine {
    tld_config cfg{};
    int d = 0, L = 0, H = 0, ntok = 0, grid = 0, pd = 0, hid = 0, img = 0, ne = 0, text = 0;
    bool finalized = false;
    bool fuse_dwconv = true;            // TLD_FUSE_DWCONV=0 selects the two-kernel path (A/B testing)
    bool fp8 = false;                   // QKV / MLP GEMMs on MX-fp8 operands (BASELINE config C4); set before finalize
    bool fp8_fused = true;              // TLD_FP8_FUSED=0: separate quantisation passes instead of quantising producers (A/B testing)
    uint8_t *a8 = nullptr, *as8 = nullptr;   // fp8 mode: quantised A operand [M, hid] and its block scales [hid/128][M][4]
    std::map<std::string, HostTensor> host;
    std::vector<void*> allocs;
    int64_t weight_bytes = 0;

    // fp32 parameters
    float *angular = nullptr, *ff1_w = nullptr, *ff1_b = nullptr, *ff3_w = nullptr, *ff3_b = nullptr;
    float *label_w = nullptr, *label_b = nullptr, *norm_w = nullptr, *norm_b = nullptr;
    float *conv_w = nullptr, *conv_b = nullptr, *pln1_w = nullptr, *pln1_b = nullptr, *plin_wt = nullptr,
          *plin_b = nullptr, *pln2_w = nullptr, *pln2_b = nullptr, *pos = nullptr, *out_w = nullptr,
          *out_b = nullptr;
    bf16* out_w_hl = nullptr;           // out_proj weight as a split bf16 pair [2][pd][d] (tail_mfma_kernel)
    bf16* plin_w_hl = nullptr;          // patch-embedding Linear weight as a split bf16 pair [2][d][pd] (embed_mfma_kernel)
    std::vector<Layer> layers;
    const float **tab_kv_w = nullptr, **tab_q_w = nullptr, **tab_n2_w = nullptr, **tab_n2_b = nullptr;   // [L] device tables

    // activations (sized for cfg.max_batch)
    resid_t* x = nullptr;
    resid_t* x_half = nullptr;         // patch embedding of the un-doubled batch (CFG layer-0 sharing)
    bool share_l0 = true;              // TLD_SHARE_L0=0 disables (A/B testing)
    bool fold_ln1 = true;              // TLD_FOLD_LN1=0: separate LayerNorm-1 kernel (A/B testing)
    bool low_latency = false;          // tld_engine_set_low_latency: capacity class for small batches (round 5) -- the down projection runs as split-K (see run_body)
    float* splitk = nullptr;           // [ll_split][max rows][d] fp32 slices of it
    int ll_split = 0;                  // K-splits of the low-latency down projection (4 or 8, from the engine's capacity: lowlat_split)
    bool fuse_qkv_attn = true;         // 256-token grids with the LayerNorm-1 fold: QKV GEMM + self-attention as ONE kernel per (sample, head) (TLD_FUSE_QKV_ATTN=0: two kernels)
    float2* ln_stats = nullptr;        // [M][kLnSlots] row partial sums of the residual stream (embed / down GEMM -> QKV GEMM)
    bool fold_ln3 = true;              // TLD_FOLD_LN3=0: cross_row writes LN3(x) and the up-projection reads it (A/B testing)
    float2* row_stats = nullptr;       // [M] (mean, rstd) of the residual rows, cross_row -> up-projection epilogue
    bf16 *xn = nullptr, *qk = nullptr, *vt = nullptr, *att = nullptr, *hid1 = nullptr, *hid2 = nullptr;
    uint32_t* seam = nullptr;            // 32 x 32 grids: the seam rows of the hidden tensor between the fused up-projection and launch_dwconv_seam
    float *io_x = nullptr, *io_sigma = nullptr, *io_label = nullptr, *io_out = nullptr;
    float *xt = nullptr, *x0_prev = nullptr, *x0_cfg = nullptr;
    int* rows_dev = nullptr;           // noise_row / label_row tables
    int64_t rows_cap = 0;
    void* stage_host = nullptr;        // pinned staging for tld_sample's sigma / row tables
    size_t stage_cap = 0;
    hipEvent_t stage_ev = nullptr;     // recorded after the last copy out of stage_host

    // conditioning tables (sized for cond_cap token rows)
    int cond_cap = 0;
    float *c_sigma = nullptr, *c_sin = nullptr, *c_h1 = nullptr, *c_pre = nullptr, *c_y = nullptr,
          *c_label = nullptr;
    float *c_kv = nullptr;             // [L][T][2d]
    float *c_wq = nullptr;             // [L][T][H][d]
    float *c_bwq = nullptr;            // [L][T][H]

    // debug stage capture
    bool debug = false;
    int dbg_batch = 0, dbg_T = 0;
    std::map<std::string, float*> stages;

    // per-class event profiling
    // (event pairs come from a per-class pool that set_profile / profile_reserve fill OUTSIDE any timed region;
    // a launch only records into the next free pair)
    uint32_t prof_mask = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[KC_COUNT];
    size_t prof_used[KC_COUNT] = {};
};

namespace {

template <typename T>
int dev_alloc(tld_engine* e, T** p, size_t count) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, count * sizeof(T) + 256));
    e->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return TLD_OK;
}

int upload_f32(tld_engine* e, const char* key, float** dst, int64_t expect) {
    auto it = e->host.find(key);
    if (it == e->host.end()) return fail(TLD_ERR_STATE, "state_dict entry missing: %s", key);
    if ((int64_t)it->second.data.size() != expect)
        return fail(TLD_ERR_SHAPE, "%s: expected %lld elements, got %lld", key, (long long)expect,
                    (long long)it->second.data.size());
    if (int rc = dev_alloc(e, dst, (size_t)expect)) return rc;
    HIP_TRY(hipMemcpy(*dst, it->second.data.data(), expect * sizeof(float), hipMemcpyHostToDevice));
    e->weight_bytes += expect * sizeof(float);
    return TLD_OK;
}

int upload_bf16(tld_engine* e, const char* key, bf16** dst, int64_t expect) {
    auto it = e->host.find(key);
    if (it == e->host.end()) return fail(TLD_ERR_STATE, "state_dict entry missing: %s", key);
    if ((int64_t)it->second.data.size() != expect)
        return fail(TLD_ERR_SHAPE, "%s: expected %lld elements, got %lld", key, (long long)expect,
                    (long long)it->second.data.size());
    std::vector<uint16_t> tmp((size_t)expect);
    const float* src = it->second.data.data();
    for (int64_t i = 0; i < expect; ++i) tmp[(size_t)i] = f32_to_bf16_rne(src[i]);
    if (int rc = dev_alloc(e, dst, (size_t)expect)) return rc;
    HIP_TRY(hipMemcpy(*dst, tmp.data(), expect * 2, hipMemcpyHostToDevice));
    e->weight_bytes += expect * 2;
    return TLD_OK;
}

int ensure_cond_capacity(tld_engine* e, int T) {
    if (T <= e->cond_cap) return TLD_OK;
    // grow-only; old tables are kept in e->allocs and freed at destroy (growth happens at most a few times)
    const int cap = ((T + 63) / 64) * 64;
    const size_t d = e->d, L = e->L, H = e->H;
    if (int rc = dev_alloc(e, &e->c_sigma, (size_t)cap)) return rc;
    if (int rc = dev_alloc(e, &e->c_sin, (size_t)cap * e->ne)) return rc;
    if (int rc = dev_alloc(e, &e->c_h1, (size_t)cap * d)) return rc;
    if (int rc = dev_alloc(e, &e->c_pre, (size_t)cap * d)) return rc;
    if (int rc = dev_alloc(e, &e->c_y, (size_t)cap * d)) return rc;
    if (int rc = dev_alloc(e, &e->c_label, (size_t)cap * e->text)) return rc;
    if (int rc = dev_alloc(e, &e->c_kv, L * cap * 2 * d)) return rc;
    if (int rc = dev_alloc(e, &e->c_wq, L * cap * H * d)) return rc;
    if (int rc = dev_alloc(e, &e->c_bwq, L * cap * H)) return rc;
    e->cond_cap = cap;
    return TLD_OK;
}

int ensure_rows_capacity(tld_engine* e, int64_t n) {
    if (n <= e->rows_cap) return TLD_OK;
    if (int rc = dev_alloc(e, &e->rows_dev, (size_t)n)) return rc;
    e->rows_cap = n;
    return TLD_OK;
}

// Pinned host staging owned by the engine: tld_sample fills it and returns without waiting for the copies; the
// next call waits on stage_ev (long complete by then) before overwriting.
int stage_acquire(tld_engine* e, size_t bytes) {
    if (!e->stage_ev) HIP_TRY(hipEventCreateWithFlags(&e->stage_ev, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(e->stage_ev));
    if (bytes <= e->stage_cap) return TLD_OK;
    if (e->stage_host) { HIP_TRY(hipHostFree(e->stage_host)); e->stage_host = nullptr; e->stage_cap = 0; }
    const size_t cap = (bytes + 4095) & ~(size_t)4095;
    HIP_TRY(hipHostMalloc(&e->stage_host, cap, hipHostMallocDefault));
    e->stage_cap = cap;
    return TLD_OK;
}

struct ProfScope {
    tld_engine* e; int cls; hipStream_t s; hipEvent_t b = nullptr; bool on;
    ProfScope(tld_engine* e_, int cls_, hipStream_t s_) : e(e_), cls(cls_), s(s_) {
        on = (e->prof_mask >> cls) & 1u;
        if (!on) return;
        auto& pool = e->prof_ev[cls];
        if (e->prof_used[cls] == pool.size()) {          // pool exhausted (no tld_engine_profile_reserve): grow here
            hipEvent_t a2 = nullptr, b2 = nullptr;
            hipEventCreate(&a2); hipEventCreate(&b2);
            pool.emplace_back(a2, b2);
        }
        const auto& pr = pool[e->prof_used[cls]++];
        b = pr.second;
        hipEventRecord(pr.first, s);
    }
    ~ProfScope() { if (on) hipEventRecord(b, s); }
};

// block 0's MLP hidden tensors (bf16 [M, hid]; debug only, own buffers: they are mlp_multiplier times a residual-stream stage): "blk0_hid" = after the
// depthwise conv + GELU on whichever path ran, "blk0_hid_pre" = the up-projection's output where it exists in HBM (the two-kernel path)
// K-splits of the low-latency down projection: class 1 = four (engines of at most 4096 token rows), class 2 = eight (at most 1024 token rows: one or two images per CFG call, the
// one-prompt-per-call pattern).  Round 6: eight splits take a one-image generate from 36.4 to 33.4 ms, but cost 6-8 ms at four or five images (twice the work items of 6 K-steps each
// no longer fit one round of workgroups, and the finishing kernel reads twice the slices) -- hence two classes, each chosen by the CALLER for the engine and each with its own fp32
// summation order: inside a class results are bit-identical across batch sizes.
constexpr int kLowLatMaxRows2 = 1024;
constexpr int kLowLatMaxRows = 4096;        // capacity of the low-latency class (token rows = max_batch x tokens): beyond it the tiles fill the chip by themselves

int capture_hidden(tld_engine* e, const char* name, const bf16* src, size_t count, hipStream_t s) {
    if (!e->debug) return TLD_OK;
    float*& buf = e->stages[name];
    if (!buf) { if (int rc = dev_alloc(e, &buf, (size_t)e->cfg.max_batch * e->ntok * e->hid)) return rc; }
    launch_cast_to_f32(src, TLD_DTYPE_BF16, buf, (int64_t)count, s);
    return TLD_OK;
}

int capture(tld_engine* e, const char* name, const resid_t* src, size_t count, hipStream_t s) {
    if (!e->debug) return TLD_OK;
    float*& buf = e->stages[name];
    if (!buf) {
        const size_t cap = (size_t)std::max(e->cfg.max_batch * e->ntok, 4 * e->cfg.max_batch + 1024) * e->d;
        if (int rc = dev_alloc(e, &buf, cap)) return rc;
    }
    launch_cast_to_f32(src, sizeof(resid_t) == 2 ? TLD_DTYPE_BF16 : TLD_DTYPE_F32, buf, (int64_t)count, s);
    return TLD_OK;
}

// Conditioning tables for T token rows whose pre-LN vectors sit in c_pre[0..T): y = LN(pre), then per
// layer K|V = y Wkv^T and the folded query vectors (denoiser.py:121-122, transformer_blocks.py:65-71).
int cond_tables(tld_engine* e, int T, hipStream_t s) {
    const int d = e->d;
    ProfScope ps(e, KC_COND, s);
    launch_layernorm_f32(e->c_pre, e->norm_w, e->norm_b, e->c_y, T, d, s);
    // all layers in two launches (blockIdx.z = layer; per-layer weights through pointer tables): these are small
    // fp32 problems (T ~ 100 token rows), 24 dependent launches of them were 2 ms per generate
    launch_linear_f32_layers(e->c_y, d, e->tab_kv_w, e->c_kv, (size_t)e->cond_cap * 2 * d, 2 * d, T, d, 2 * d, e->L, s);
    launch_wq_layers(e->c_kv, (size_t)e->cond_cap * 2 * d, 2 * d, e->tab_q_w, e->tab_n2_w, e->tab_n2_b, e->c_wq,
                     (size_t)e->cond_cap * e->H * d, e->c_bwq, (size_t)e->cond_cap * e->H, T, e->H, d, e->L, s);
    return TLD_OK;
}

// noise rows: sigma[0..Tn) in c_sigma -> c_pre[0..Tn)   (denoiser.py:105-110,117)
void cond_noise_rows(tld_engine* e, int Tn, hipStream_t s) {
    launch_sinusoid(e->c_sigma, e->angular, e->c_sin, Tn, e->ne / 2, s);
    launch_linear_f32(e->c_sin, e->ne, e->ff1_w, e->ff1_b, e->c_h1, e->d, Tn, e->ne, e->d, 1, s);
    launch_linear_f32(e->c_h1, e->d, e->ff3_w, e->ff3_b, e->c_pre, e->d, Tn, e->d, e->d, 0, s);
}

// label rows: c_label[0..Tl) -> c_pre[row0 .. row0+Tl)   (denoiser.py:114,119)
void cond_label_rows(tld_engine* e, int row0, int Tl, hipStream_t s) {
    launch_linear_f32(e->c_label, e->text, e->label_w, e->label_b, e->c_pre + (size_t)row0 * e->d, e->d, Tl,
                      e->text, e->d, 0, s);
}

// embed -> L decoder blocks -> tail, for `batch` model samples whose latents are x_src[b % src_batch].
// share_l0: the model batch is [x_src ; x_src] (CFG doubling), so everything before the first cross-attention --
// patch embedding, LN1, the QKV GEMM and self-attention of block 0 -- is identical for both halves and is
// computed for src_batch samples only; block 0's row kernel then fans it out to the 2*src_batch streams.
int run_body(tld_engine* e, const float* x_src, int src_batch, int batch, const int* noise_row,
             const int* label_row, float* out, hipStream_t s, bool share_l0 = false) {
    const int d = e->d, M = batch * e->ntok;
    share_l0 = share_l0 && e->share_l0 && batch == 2 * src_batch && !e->debug;
    const int b0 = share_l0 ? src_batch : batch;            // samples processed up to block 0's attention
    resid_t* xe = share_l0 ? e->x_half : e->x;
    const bool fold1 = e->fold_ln1;                          // LayerNorm-1 applied inside the QKV GEMM's epilogue
    const int ln_slots = gemm_resid_stat_slots(d);
    {
        ProfScope ps(e, KC_EMBED, s);
        EmbedParams ep{};
        ep.x = x_src; ep.conv_w = e->conv_w; ep.conv_b = e->conv_b; ep.ln1_w = e->pln1_w; ep.ln1_b = e->pln1_b;
        ep.lin_wt = e->plin_wt; ep.lin_w_hl = e->plin_w_hl; ep.lin_b = e->plin_b; ep.ln2_w = e->pln2_w; ep.ln2_b = e->pln2_b; ep.pos = e->pos;
        ep.tok = xe; ep.stats_out = fold1 ? e->ln_stats : nullptr; ep.batch = b0; ep.src_batch = src_batch; ep.C = e->cfg.n_channels;
        ep.S = e->cfg.image_size; ep.p = e->cfg.patch_size; ep.grid = e->grid; ep.pd = e->pd; ep.d = d;
        ep.ntok = e->ntok;
        launch_embed(ep, s);
    }
    if (int rc = capture(e, "tokens0", e->x, (size_t)M * d, s)) return rc;
    for (int l = 0; l < e->L; ++l) {
        const Layer& Ly = e->layers[l];
        const bool half = share_l0 && l == 0;
        const int bl = half ? b0 : batch, Ml = bl * e->ntok;
        const bool fuse8 = e->fp8 && e->fp8_fused;      // fp8 mode: producers write the MX-fp8 operand themselves
        if (!fold1) {   // xn = LN1(x)
            ProfScope ps(e, KC_LN, s);
            if (fuse8 && layernorm_mx8_supported(d)) launch_layernorm_mx8(half ? xe : e->x, Ly.n1_w, Ly.n1_b, e->a8, e->as8, Ml, d, s);
            else launch_layernorm_bf16(half ? xe : e->x, Ly.n1_w, Ly.n1_b, e->xn, Ml, d, s);
        }
        // 256-token grids with the LayerNorm-1 fold: the QKV projection and the whole self-attention of a (sample, head) are one 256 x 192
        // GEMM tile + epilogue; q | k, v^T never reach HBM and `att` is written directly.  (the fp8 mode keeps the two kernels)
        const bool fused_qa = e->fuse_qkv_attn && fold1 && !e->fp8;
        if (fused_qa) {
            ProfScope ps(e, KC_GEMM_QKV, s);
            GemmParams g{};
            g.A = half ? xe : e->x; g.lda = d; g.W = Ly.qkv_wp; g.ldw = d; g.M = Ml; g.N = 3 * d; g.K = d;
            g.out_bf16 = e->att; g.ldo = d; g.ntok = e->ntok; g.d = d;
            g.ln_stats = e->ln_stats; g.ln_slots = l == 0 ? 2 : ln_slots; g.ln_c1 = Ly.qkv_c1p; g.ln_b1 = Ly.qkv_b1p;
            launch_gemm(g, EPI_QKV_ATTN, s);
        } else {   // q|k, v^T = LN1(x) Wqkv^T
            ProfScope ps(e, KC_GEMM_QKV, s);
            GemmParams g{};
            g.A = e->xn; g.lda = d; g.W = Ly.qkv_w; g.ldw = d; g.M = Ml; g.N = 3 * d; g.K = d;
            g.out_bf16 = e->qk; g.ldo = 2 * d; g.vt = e->vt; g.ntok = e->ntok; g.d = d;
            if (e->fp8) {   // MX-fp8: quantise LN1(x) (one pass over [M, d]; fused into the LayerNorm kernel when possible), then the e4m3 GEMM
                if (!(fuse8 && layernorm_mx8_supported(d))) launch_quant_mx8(e->xn, e->a8, e->as8, Ml, d, s);
                g.f8 = 1; g.A = reinterpret_cast<const bf16*>(e->a8); g.W = reinterpret_cast<const bf16*>(Ly.qkv_w8);
                g.a_scale = e->as8; g.w_scale = Ly.qkv_s8;
            }
#ifdef TLD_RESID_BF16
            if (fold1) {    // raw residual rows x gamma-scaled weights; partial sums from embed (block 0) / the down projection
                g.A = half ? xe : e->x; g.W = Ly.qkv_wf; g.ln_stats = e->ln_stats; g.ln_slots = l == 0 ? 2 : ln_slots;
                g.ln_c1 = Ly.qkv_c1; g.ln_b1 = Ly.qkv_b1;
            }
#endif
            launch_gemm(g, fold1 ? EPI_QKV_LN : EPI_QKV, s);
        }
        if (!fused_qa) {
            ProfScope ps(e, KC_ATTN, s);
            launch_attention(e->qk, e->vt, e->att, bl, e->ntok, e->H, s);
        }
        if (l == 0 && e->debug && !e->stages["blk0_sa"]) {
            float* buf = nullptr;
            if (int rc = dev_alloc(e, &buf, (size_t)e->cfg.max_batch * e->ntok * d)) return rc;
            e->stages["blk0_sa"] = buf;
        }
        // 256 px (16x16 tokens): one GEMM tile row-block is one image, so the depthwise conv + GELU run inside the
        // up-projection's epilogue and the pre-conv hidden never reaches HBM.  Other grids: separate kernels.
        // 512 px (32x32 tokens, round 4): a tile row-block is 8 image rows; the epilogue finishes the interior rows and a thin second kernel the two rows at
        // every tile seam from the hidden rows the epilogue leaves for it (half of the hidden tensor instead of a write + read of all of it).
        const bool fuse_dw = e->fuse_dwconv && (e->grid == 16 || (e->grid == 32 && e->seam)) && e->hid % 256 == 0;
        const bool fold3 = e->fold_ln3;                 // LN3 applied in the up-projection's epilogue: cross_row writes row statistics, not xn
        {   // x += att; x += CA(LN2 x, y); xn = LN3(x) (or its row statistics)
            ProfScope ps(e, KC_CROSS, s);
            CrossRowParams cp{};
            cp.x = e->x; cp.att = e->att;
            cp.x_in = half ? xe : nullptr; cp.src_batch = src_batch;
            cp.wq = e->c_wq + (size_t)l * e->cond_cap * e->H * d;
            cp.bwq = e->c_bwq + (size_t)l * e->cond_cap * e->H;
            cp.v = e->c_kv + (size_t)l * e->cond_cap * 2 * d + d;     // V half of each [2d] row
            cp.v_ld = 2 * d;
            cp.noise_row = noise_row; cp.label_row = label_row;
            cp.ln2_w = Ly.n2_w; cp.ln2_b = Ly.n2_b; cp.ln3_w = Ly.n3_w; cp.ln3_b = Ly.n3_b;
            cp.xn3 = fold3 ? nullptr : e->xn; cp.ln3_stats = fold3 ? e->row_stats : nullptr; cp.batch = batch; cp.ntok = e->ntok; cp.d = d; cp.heads = e->H;
            cp.sa_out = (l == 0 && e->debug) ? e->stages["blk0_sa"] : nullptr;
            const bool cross8 = fuse8 && cross_row_supports_ln3_stats(d);      // (= the 4-features-per-lane kernel is in use)
            if (cross8) { cp.xn3_f8 = e->a8; cp.xn3_s8 = e->as8; }
            launch_cross_row(cp, s);
        }
        if (l == 0) if (int rc = capture(e, "blk0_ca", e->x, (size_t)M * d, s)) return rc;
        if (fuse_dw) {
            ProfScope ps(e, KC_GEMM_UP, s);
            GemmParams g{};
            g.A = e->xn; g.lda = d; g.W = Ly.up_w; g.ldw = d; g.M = M; g.N = e->hid; g.K = d;
            g.out_bf16 = e->hid2; g.ldo = e->hid; g.bias = Ly.up_b; g.dw_b = Ly.dw_b_half;
            g.dw_wpk = Ly.dw_wpk;
#ifdef TLD_RESID_BF16
            if (fold3) {    // LN3 inside the epilogue: raw residual rows x gamma-scaled weights, statistics from cross_row
                g.A = e->x; g.W = Ly.up_wf; g.bias = Ly.up_b1; g.ln_c1 = Ly.up_c1; g.row_stats = e->row_stats;
            }
#endif
            g.dw_seam = e->seam;
            launch_gemm(g, e->grid == 32 ? EPI_UP_DWCONV32 : EPI_UP_DWCONV2, s);
            if (e->grid == 32) {
                ProfScope ps2(e, KC_DWCONV, s);
                launch_dwconv_seam(e->seam, Ly.dw_wpk, Ly.dw_b_half, e->hid2, e->hid, batch, e->hid, s);
            }
        } else {
            {   // hid1 = xn Wup^T + b
                ProfScope ps(e, KC_GEMM_UP, s);
                GemmParams g{};
                g.A = e->xn; g.lda = d; g.W = Ly.up_w; g.ldw = d; g.M = M; g.N = e->hid; g.K = d;
                g.out_bf16 = e->hid1; g.ldo = e->hid; g.bias = Ly.up_b;
                if (e->fp8) {
                    if (!(fuse8 && cross_row_supports_ln3_stats(d))) launch_quant_mx8(e->xn, e->a8, e->as8, M, d, s);
                    g.f8 = 1; g.A = reinterpret_cast<const bf16*>(e->a8); g.W = reinterpret_cast<const bf16*>(Ly.up_w8);
                    g.a_scale = e->as8; g.w_scale = Ly.up_s8;
                }
#ifdef TLD_RESID_BF16
                if (fold3) {
                    g.A = e->x; g.W = Ly.up_wf; g.bias = Ly.up_b1; g.ln_c1 = Ly.up_c1; g.row_stats = e->row_stats;
                }
#endif
                launch_gemm(g, EPI_BIAS_BF16, s);
            }
            if (l == 0 && !e->fp8) if (int rc = capture_hidden(e, "blk0_hid_pre", e->hid1, (size_t)M * e->hid, s)) return rc;
            {
                ProfScope ps(e, KC_DWCONV, s);
                const bool dw8 = fuse8 && e->grid > 16;       // the tiled kernel writes the fp8 operand of the down projection itself
                launch_dwconv_gelu(e->hid1, e->hid2, Ly.dw_w9c, Ly.dw_b, Ly.dw_w9c_half, Ly.dw_b_half, batch, e->grid, e->hid, s,
                                   dw8 ? e->a8 : nullptr, dw8 ? e->as8 : nullptr);
            }
        }
        if (l == 0) if (int rc = capture_hidden(e, "blk0_hid", e->hid2, (size_t)M * e->hid, s)) return rc;
        if (e->low_latency && !e->fp8) {
            // Low-latency class: a handful of samples are a handful of 256-row tiles, and the down projection's 48 K-steps per tile (K = 4 d) were
            // half of a layer's time however empty the chip was.  Four K-splits quadruple the work items (fp32 slices, summed in a fixed order by
            // the finishing kernel together with bias, residual add and the LayerNorm-1 partial sums).  Results differ from the default class in the
            // fp32 summation order of that one product -- which is why the class is a property of the ENGINE (chosen by the caller, checked against
            // its capacity), never of the batch a call happens to carry: inside a class results stay bit-identical across batch sizes.
            ProfScope ps(e, KC_GEMM_DOWN, s);
            GemmParams g{};
            g.A = e->hid2; g.lda = e->hid; g.W = Ly.down_w; g.ldw = e->hid; g.M = M; g.N = d; g.K = e->hid / e->ll_split; g.ksplit = e->ll_split;
            g.c_f32 = e->splitk; g.ldc = d;
            launch_gemm(g, EPI_F32, s);
            launch_splitk_resid(e->splitk, e->ll_split, (size_t)M * d, Ly.down_b, e->x, (fold1 && l + 1 < e->L) ? e->ln_stats : nullptr, M, d, s);
        } else
        {   // x += hid2 Wdown^T + b
            ProfScope ps(e, KC_GEMM_DOWN, s);
            GemmParams g{};
            g.A = e->hid2; g.lda = e->hid; g.W = Ly.down_w; g.ldw = e->hid; g.M = M; g.N = d; g.K = e->hid;
            g.bias = Ly.down_b; g.resid = e->x; g.ldr = d;
            g.stats_out = (fold1 && l + 1 < e->L) ? e->ln_stats : nullptr;
            if (e->fp8) {
                if (!(fuse8 && e->grid > 16)) launch_quant_mx8(e->hid2, e->a8, e->as8, M, e->hid, s);
                g.f8 = 1; g.A = reinterpret_cast<const bf16*>(e->a8); g.W = reinterpret_cast<const bf16*>(Ly.down_w8);
                g.a_scale = e->as8; g.w_scale = Ly.down_s8;
            }
            launch_gemm(g, EPI_BIAS_RESID, s);
        }
        if (l == 0) if (int rc = capture(e, "blk0_mlp", e->x, (size_t)M * d, s)) return rc;
    }
    if (int rc = capture(e, "tokens_final", e->x, (size_t)M * d, s)) return rc;
    {
        ProfScope ps(e, KC_TAIL, s);
        TailParams tp{};
        tp.tok = e->x; tp.w = e->out_w; tp.w_hl = e->out_w_hl; tp.b = e->out_b; tp.out = out; tp.batch = batch;
        tp.C = e->cfg.n_channels; tp.S = e->cfg.image_size; tp.p = e->cfg.patch_size; tp.grid = e->grid;
        tp.pd = e->pd; tp.d = d; tp.ntok = e->ntok;
        launch_tail(tp, s);
    }
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

const char* tld_last_error(void) { return g_err; }


int tld_engine_create(const tld_config* c, tld_engine** out) {
    if (!c || !out) return fail(TLD_ERR_INVALID, "null argument");
    *out = nullptr;
    if (c->embed_dim <= 0 || c->embed_dim % 64 != 0 || c->embed_dim > 1024)
        return fail(TLD_ERR_INVALID, "embed_dim=%d unsupported: must be a multiple of the head width 64 (n_heads = embed_dim // 64, "
                    "tld/transformer_blocks.py:126-128) and <= 1024 (the row kernels hold a token row in 8 x 128-feature register groups)", c->embed_dim);
    if (c->patch_size <= 0 || c->image_size % c->patch_size != 0)
        return fail(TLD_ERR_INVALID, "image_size=%d must be divisible by patch_size=%d", c->image_size, c->patch_size);
    const int grid = c->image_size / c->patch_size, ntok = grid * grid;
    // round 4: any grid whose side is a multiple of 4 (token count a multiple of 16: the 16-row groups of the cross-attention row kernel, 16-byte V^T
    // rows); 64 / 128 / k 256 tokens have shape-specialised attention kernels, everything else takes the masked chunked one
    if (ntok % 16 != 0)
        return fail(TLD_ERR_INVALID, "token count %d unsupported: image_size / patch_size must be a multiple of 4", ntok);
    const int pd = c->n_channels * c->patch_size * c->patch_size;
    if (pd > 64) return fail(TLD_ERR_INVALID, "patch_dim=%d > 64 unsupported", pd);
    if (c->noise_embed_dims % 2 || c->noise_embed_dims <= 0) return fail(TLD_ERR_INVALID, "noise_embed_dims must be even");
    if (c->max_batch <= 0 || c->n_layers <= 0 || c->mlp_multiplier <= 0 || c->text_emb_size <= 0)
        return fail(TLD_ERR_INVALID, "non-positive size in config");
    if ((c->mlp_multiplier * c->embed_dim) % 64) return fail(TLD_ERR_INVALID, "hidden width must be a multiple of 64");
    {   // the GEMM tile DMA addresses operands with 32-bit byte offsets
        const int64_t g = c->image_size / c->patch_size;
        const int64_t hid_bytes = (int64_t)c->max_batch * g * g * c->mlp_multiplier * c->embed_dim * 2;
        if (hid_bytes >= (int64_t)1 << 32)
            return fail(TLD_ERR_INVALID, "max_batch=%d: the hidden activation (%lld bytes) must stay below 4 GiB; "
                        "run larger batches as several calls", c->max_batch, (long long)hid_bytes);
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (c->device_id < 0 || c->device_id >= ndev) return fail(TLD_ERR_INVALID, "device_id %d out of range (%d devices)", c->device_id, ndev);
    {   // the row kernels keep per-workgroup tables in dynamic LDS (opted in above the 64-KiB default; a CU has 160 KiB)
        const long d = c->embed_dim, H = d / 64;
        const long embed_lds = (pd * d + (long)pd * pd) * 4, tail_lds = pd * d * 4, cross_lds = (H * d + 2 * d + H) * 4;
        const long lim = 160 * 1024;
        if (embed_lds > lim || tail_lds > lim || cross_lds > lim)
            return fail(TLD_ERR_INVALID, "embed_dim=%d with patch_dim=%d needs %ld / %ld / %ld bytes of LDS in the embed / "
                        "out-proj / cross-attention row kernels (limit %ld each): reduce patch_size*patch_size*n_channels "
                        "or embed_dim", c->embed_dim, pd, embed_lds, tail_lds, cross_lds, lim);
    }
    DeviceGuard dg(c->device_id);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(TLD_ERR_INVALID, "device %d is %s; this engine is built for gfx950 only", c->device_id, prop.gcnArchName);
    tld_engine* e = new tld_engine();
    e->cfg = *c;
    e->d = c->embed_dim; e->L = c->n_layers; e->H = c->embed_dim / 64; e->grid = grid; e->ntok = ntok;
    e->pd = pd; e->hid = c->mlp_multiplier * c->embed_dim; e->img = c->n_channels * c->image_size * c->image_size;
    e->ne = c->noise_embed_dims; e->text = c->text_emb_size;
    e->layers.resize(e->L);
    if (const char* fd = getenv("TLD_FUSE_DWCONV")) e->fuse_dwconv = atoi(fd) != 0;
    if (const char* sl = getenv("TLD_SHARE_L0")) e->share_l0 = atoi(sl) != 0;
    if (const char* f8 = getenv("TLD_FP8_FUSED")) e->fp8_fused = atoi(f8) != 0;
    if (const char* fl = getenv("TLD_FOLD_LN3")) e->fold_ln3 = atoi(fl) != 0;
    if (const char* fl = getenv("TLD_FOLD_LN1")) e->fold_ln1 = atoi(fl) != 0;
#ifndef TLD_RESID_BF16
    e->fold_ln3 = false;               // the folds feed the bf16 residual stream straight to the MFMA
    e->fold_ln1 = false;
#endif
    // LayerNorm-1 fold: needs the down projection's 96-column partial sums (d % 192 == 0, at most 8 groups) and a QKV
    // width the 256-wide kernel takes for every batch size
    e->fold_ln1 = e->fold_ln1 && gemm_resid_stat_slots(e->d) > 0 && (3 * e->d) % 256 == 0;
    // needs the statistics-writing row kernel; both up-projection epilogues (fused depthwise at 16 x 16 tokens, plain
    // bias + bf16 otherwise) apply the fold
    e->fold_ln3 = e->fold_ln3 && e->hid % 256 == 0 && cross_row_supports_ln3_stats(e->d);
    // fused QKV -> attention: one 256 x 192 tile per (sample, head) needs 256-token samples, the LayerNorm-1 fold (the kernel reads the raw
    // residual stream) and an even number of 64-wide K-steps (its LDS map relies on which stage a tile consumes last)
    if (const char* fq = getenv("TLD_FUSE_QKV_ATTN")) e->fuse_qkv_attn = atoi(fq) != 0;
    e->fuse_qkv_attn = e->fuse_qkv_attn && e->fold_ln1 && ntok == 256 && e->d % 128 == 0;
    *out = e;
    return TLD_OK;
}

int tld_engine_load_tensor(tld_engine* e, const char* key, const void* host_ptr, const int64_t* shape,
                           int32_t ndim, int32_t dtype) {
    if (!e || !key || (!host_ptr && ndim > 0)) return fail(TLD_ERR_INVALID, "null argument");
    if (e->finalized) return fail(TLD_ERR_STATE, "weights already finalized");
    if (strstr(key, "precomputed_pos_enc")) return TLD_OK;          // arange buffer (denoiser.py:55)
    if (dtype != TLD_DTYPE_F32) return fail(TLD_ERR_INVALID, "%s: only fp32 state_dict tensors are accepted", key);
    int64_t n = 1;
    HostTensor t;
    for (int i = 0; i < ndim; ++i) { n *= shape[i]; t.shape.push_back(shape[i]); }
    t.data.assign(static_cast<const float*>(host_ptr), static_cast<const float*>(host_ptr) + n);
    e->host[key] = std::move(t);
    return TLD_OK;
}

int tld_engine_finalize_weights(tld_engine* e) {
    if (!e) return fail(TLD_ERR_INVALID, "null engine");
    if (e->finalized) return TLD_OK;
    DeviceGuard dg(e->cfg.device_id);
    const int64_t d = e->d, ne = e->ne, pd = e->pd, hid = e->hid, N = e->ntok, text = e->text;
    const int64_t cpp = pd;   // C*p*p
#define UP32(key, field, n) if (int rc = upload_f32(e, key, &e->field, (n))) return rc;
    UP32("fourier_feats.0.angular_speeds", angular, ne / 2)
    UP32("fourier_feats.1.weight", ff1_w, d * ne) UP32("fourier_feats.1.bias", ff1_b, d)
    UP32("fourier_feats.3.weight", ff3_w, d * d) UP32("fourier_feats.3.bias", ff3_b, d)
    UP32("label_proj.weight", label_w, d * text) UP32("label_proj.bias", label_b, d)
    UP32("norm.weight", norm_w, d) UP32("norm.bias", norm_b, d)
    UP32("denoiser_trans_block.patchify_and_embed.0.weight", conv_w, pd * cpp)
    UP32("denoiser_trans_block.patchify_and_embed.0.bias", conv_b, pd)
    UP32("denoiser_trans_block.patchify_and_embed.2.weight", pln1_w, pd)
    UP32("denoiser_trans_block.patchify_and_embed.2.bias", pln1_b, pd)
    UP32("denoiser_trans_block.patchify_and_embed.3.bias", plin_b, d)
    UP32("denoiser_trans_block.patchify_and_embed.4.weight", pln2_w, d)
    UP32("denoiser_trans_block.patchify_and_embed.4.bias", pln2_b, d)
    UP32("denoiser_trans_block.pos_embed.weight", pos, N * d)
    UP32("denoiser_trans_block.out_proj.0.weight", out_w, pd * d)
    UP32("denoiser_trans_block.out_proj.0.bias", out_b, pd)
#undef UP32
    {   // out_proj weight as hi + lo bf16 halves for the matrix-pipe tail kernel (hi = bf16(w), lo = bf16(w - hi): 16 significant bits)
        const std::vector<float>& W = e->host["denoiser_trans_block.out_proj.0.weight"].data;
        std::vector<uint16_t> hl((size_t)(2 * pd * d));
        for (int64_t i = 0; i < pd * d; ++i) {
            const uint16_t hi = f32_to_bf16_rne(W[(size_t)i]);
            uint32_t u = (uint32_t)hi << 16; float hf; memcpy(&hf, &u, 4);
            hl[(size_t)i] = hi; hl[(size_t)(pd * d + i)] = f32_to_bf16_rne(W[(size_t)i] - hf);
        }
        if (int rc = dev_alloc(e, &e->out_w_hl, hl.size())) return rc;
        HIP_TRY(hipMemcpy(e->out_w_hl, hl.data(), hl.size() * 2, hipMemcpyHostToDevice));
        e->weight_bytes += (int64_t)hl.size() * 2;
    }
    {   // the same split for the patch-embedding Linear weight [d, pd] (embed_mfma_kernel)
        const std::vector<float>& W = e->host["denoiser_trans_block.patchify_and_embed.3.weight"].data;
        if ((int64_t)W.size() == d * pd) {
            std::vector<uint16_t> hl((size_t)(2 * pd * d));
            for (int64_t i = 0; i < pd * d; ++i) {
                const uint16_t hi = f32_to_bf16_rne(W[(size_t)i]);
                uint32_t u = (uint32_t)hi << 16; float hf; memcpy(&hf, &u, 4);
                hl[(size_t)i] = hi; hl[(size_t)(pd * d + i)] = f32_to_bf16_rne(W[(size_t)i] - hf);
            }
            if (int rc = dev_alloc(e, &e->plin_w_hl, hl.size())) return rc;
            HIP_TRY(hipMemcpy(e->plin_w_hl, hl.data(), hl.size() * 2, hipMemcpyHostToDevice));
            e->weight_bytes += (int64_t)hl.size() * 2;
        }
    }
    {   // Linear(pd -> d) weight [d, pd] -> transposed [pd, d] for coalesced per-feature reads
        const char* key = "denoiser_trans_block.patchify_and_embed.3.weight";
        auto it = e->host.find(key);
        if (it == e->host.end()) return fail(TLD_ERR_STATE, "state_dict entry missing: %s", key);
        if ((int64_t)it->second.data.size() != d * pd) return fail(TLD_ERR_SHAPE, "%s: bad size", key);
        std::vector<float> tr((size_t)(d * pd));
        for (int64_t n = 0; n < d; ++n)
            for (int64_t o = 0; o < pd; ++o) tr[(size_t)(o * d + n)] = it->second.data[(size_t)(n * pd + o)];
        if (int rc = dev_alloc(e, &e->plin_wt, (size_t)(d * pd))) return rc;
        HIP_TRY(hipMemcpy(e->plin_wt, tr.data(), tr.size() * sizeof(float), hipMemcpyHostToDevice));
        e->weight_bytes += (int64_t)tr.size() * 4;
    }
    char key[256];
    for (int l = 0; l < e->L; ++l) {
        Layer& Ly = e->layers[l];
#define LK(suffix) (snprintf(key, sizeof(key), "denoiser_trans_block.decoder_blocks.%d.%s", l, suffix), key)
        // bf16 GEMM operands: only the copies this engine's paths read are made resident (the LayerNorm folds use gamma-scaled
        // copies built below, the fp8 mode its own e4m3 ones); presence and size of the three matrices are checked either way
        {
            const struct { const char* suffix; int64_t n; } need[3] = {{"self_attention.qkv_linear.weight", 3 * d * d},
                                                                       {"mlp.mlp.0.weight", hid * d}, {"mlp.mlp.3.weight", d * hid}};
            for (const auto& nd : need) {
                auto it = e->host.find(LK(nd.suffix));
                if (it == e->host.end()) return fail(TLD_ERR_STATE, "state_dict entry missing: %s", key);
                if ((int64_t)it->second.data.size() != nd.n)
                    return fail(TLD_ERR_SHAPE, "%s: expected %lld elements, got %lld", key, (long long)nd.n, (long long)it->second.data.size());
            }
            for (const char* sfx : {"norm1.weight", "norm1.bias", "norm3.weight", "norm3.bias", "mlp.mlp.0.bias", "mlp.mlp.1.bias"})
                if (e->host.find(LK(sfx)) == e->host.end()) return fail(TLD_ERR_STATE, "state_dict entry missing: %s", key);
        }
        if (!e->fold_ln1 && !e->fp8)
            if (int rc = upload_bf16(e, LK("self_attention.qkv_linear.weight"), &Ly.qkv_w, 3 * d * d)) return rc;
        if (int rc = upload_f32(e, LK("cross_attention.kv_linear.weight"), &Ly.kv_w, 2 * d * d)) return rc;
        if (int rc = upload_f32(e, LK("cross_attention.q_linear.weight"), &Ly.q_w, d * d)) return rc;
        if (!e->fold_ln3 && !e->fp8)
            if (int rc = upload_bf16(e, LK("mlp.mlp.0.weight"), &Ly.up_w, hid * d)) return rc;
        if (int rc = upload_f32(e, LK("mlp.mlp.0.bias"), &Ly.up_b, hid)) return rc;
        if (int rc = upload_f32(e, LK("mlp.mlp.1.bias"), &Ly.dw_b, hid)) return rc;
        if (!e->fp8)
            if (int rc = upload_bf16(e, LK("mlp.mlp.3.weight"), &Ly.down_w, d * hid)) return rc;
        if (int rc = upload_f32(e, LK("mlp.mlp.3.bias"), &Ly.down_b, d)) return rc;
        if (int rc = upload_f32(e, LK("norm1.weight"), &Ly.n1_w, d)) return rc;
        if (int rc = upload_f32(e, LK("norm1.bias"), &Ly.n1_b, d)) return rc;
        if (e->fold_ln1) {
            const std::vector<float>& W = e->host[LK("self_attention.qkv_linear.weight")].data;   // [3d][d]
            const std::vector<float>& g1 = e->host[LK("norm1.weight")].data;
            const std::vector<float>& be = e->host[LK("norm1.bias")].data;
            std::vector<uint16_t> wf((size_t)(3 * d * d));
            std::vector<float> c1((size_t)(3 * d)), b1((size_t)(3 * d));
            for (int64_t n = 0; n < 3 * d; ++n) {
                double sc = 0.0, sb = 0.0;
                for (int64_t k2 = 0; k2 < d; ++k2) {
                    const float w = W[(size_t)(n * d + k2)];
                    const uint16_t q = f32_to_bf16_rne(g1[(size_t)k2] * w);
                    wf[(size_t)(n * d + k2)] = q;
                    uint32_t u = (uint32_t)q << 16; float qf; memcpy(&qf, &u, 4);
                    sc += qf; sb += (double)be[(size_t)k2] * w;
                }
                c1[(size_t)n] = (float)sc; b1[(size_t)n] = (float)sb;
            }
            if (int rc = dev_alloc(e, &Ly.qkv_wf, wf.size())) return rc;
            if (int rc = dev_alloc(e, &Ly.qkv_c1, c1.size())) return rc;
            if (int rc = dev_alloc(e, &Ly.qkv_b1, b1.size())) return rc;
            HIP_TRY(hipMemcpy(Ly.qkv_wf, wf.data(), wf.size() * 2, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(Ly.qkv_c1, c1.data(), c1.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(Ly.qkv_b1, b1.data(), b1.size() * 4, hipMemcpyHostToDevice));
            e->weight_bytes += (int64_t)wf.size() * 2 + (int64_t)c1.size() * 8;
            if (e->fuse_qkv_attn) {     // rows [q; k; v] x [head][64]  ->  [head][feature half][q | k | v][32]: tile-column h of the fused kernel is head h, and each of
                // its two 96-column wave columns holds 32 features of q, of k AND of v -- so the slow part of the image write (V^T: 2-byte scattered LDS writes) is
                // shared by all eight waves instead of falling on the four that held v (round 5; the values and their accumulation order do not change)
                std::vector<uint16_t> wp(wf.size());
                std::vector<float> c1p(c1.size()), b1p(b1.size());
                for (int64_t h = 0; h < e->H; ++h)
                    for (int part = 0; part < 3; ++part)
                        for (int64_t c = 0; c < 64; ++c) {
                            const int64_t src = part * d + h * 64 + c, dst = h * 192 + (c >> 5) * 96 + part * 32 + (c & 31);
                            memcpy(&wp[(size_t)(dst * d)], &wf[(size_t)(src * d)], (size_t)d * 2);
                            c1p[(size_t)dst] = c1[(size_t)src]; b1p[(size_t)dst] = b1[(size_t)src];
                        }
                if (int rc = dev_alloc(e, &Ly.qkv_wp, wp.size())) return rc;
                if (int rc = dev_alloc(e, &Ly.qkv_c1p, c1p.size() + 64)) return rc;      // (+64: the side-table DMA of the last head reads a 256-column window)
                if (int rc = dev_alloc(e, &Ly.qkv_b1p, b1p.size() + 64)) return rc;
                HIP_TRY(hipMemcpy(Ly.qkv_wp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(Ly.qkv_c1p, c1p.data(), c1p.size() * 4, hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(Ly.qkv_b1p, b1p.data(), b1p.size() * 4, hipMemcpyHostToDevice));
                e->weight_bytes += (int64_t)wp.size() * 2 + (int64_t)c1p.size() * 8;
            }
        }
        if (e->fp8) {
            struct Q { const char* key; int64_t rows, K; uint8_t** w; uint8_t** sc; };
            const Q qs[3] = {{"self_attention.qkv_linear.weight", 3 * d, d, &Ly.qkv_w8, &Ly.qkv_s8},
                             {"mlp.mlp.0.weight", hid, d, &Ly.up_w8, &Ly.up_s8},
                             {"mlp.mlp.3.weight", d, hid, &Ly.down_w8, &Ly.down_s8}};
            for (const Q& q : qs) {
                const std::vector<float>& W = e->host[LK(q.key)].data;
                std::vector<uint8_t> w8((size_t)(q.rows * q.K)), s8((size_t)(q.rows * q.K / 32));
                quant_mx8_host(W.data(), (int)q.rows, (int)q.K, w8.data(), s8.data());
                if (int rc = dev_alloc(e, q.w, w8.size())) return rc;
                if (int rc = dev_alloc(e, q.sc, s8.size())) return rc;
                HIP_TRY(hipMemcpy(*q.w, w8.data(), w8.size(), hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(*q.sc, s8.data(), s8.size(), hipMemcpyHostToDevice));
                e->weight_bytes += (int64_t)(w8.size() + s8.size());
            }
        }
        if (int rc = upload_f32(e, LK("norm2.weight"), &Ly.n2_w, d)) return rc;
        if (int rc = upload_f32(e, LK("norm2.bias"), &Ly.n2_b, d)) return rc;
        if (int rc = upload_f32(e, LK("norm3.weight"), &Ly.n3_w, d)) return rc;
        if (int rc = upload_f32(e, LK("norm3.bias"), &Ly.n3_b, d)) return rc;
        if (e->fold_ln3) {
            const std::vector<float>& W = e->host[LK("mlp.mlp.0.weight")].data;      // [hid][d]  (1x1 conv = Linear)
            const std::vector<float>& ub = e->host[LK("mlp.mlp.0.bias")].data;
            const std::vector<float>& g3 = e->host[LK("norm3.weight")].data;
            const std::vector<float>& b3 = e->host[LK("norm3.bias")].data;
            std::vector<uint16_t> wf((size_t)(hid * d));
            std::vector<float> c1((size_t)hid), b1((size_t)hid);
            for (int64_t n = 0; n < hid; ++n) {
                double sc = 0.0, sb = 0.0;
                for (int64_t k2 = 0; k2 < d; ++k2) {
                    const float w = W[(size_t)(n * d + k2)];
                    const uint16_t q = f32_to_bf16_rne(g3[(size_t)k2] * w);
                    wf[(size_t)(n * d + k2)] = q;
                    uint32_t u = (uint32_t)q << 16; float qf; memcpy(&qf, &u, 4);
                    sc += qf; sb += (double)b3[(size_t)k2] * w;
                }
                c1[(size_t)n] = (float)sc; b1[(size_t)n] = (float)(sb + ub[(size_t)n]);
            }
            if (int rc = dev_alloc(e, &Ly.up_wf, wf.size())) return rc;
            if (int rc = dev_alloc(e, &Ly.up_c1, c1.size())) return rc;
            if (int rc = dev_alloc(e, &Ly.up_b1, b1.size())) return rc;
            HIP_TRY(hipMemcpy(Ly.up_wf, wf.data(), wf.size() * 2, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(Ly.up_c1, c1.data(), c1.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(Ly.up_b1, b1.data(), b1.size() * 4, hipMemcpyHostToDevice));
            e->weight_bytes += (int64_t)wf.size() * 2 + (int64_t)c1.size() * 8;
        }
        {   // depthwise weight [hid,1,3,3] -> [9][hid]
            auto it = e->host.find(LK("mlp.mlp.1.weight"));
            if (it == e->host.end()) return fail(TLD_ERR_STATE, "state_dict entry missing: %s", key);
            if ((int64_t)it->second.data.size() != hid * 9) return fail(TLD_ERR_SHAPE, "%s: bad size", key);
            std::vector<float> tr((size_t)(hid * 9));
            for (int64_t c = 0; c < hid; ++c)
                for (int k = 0; k < 9; ++k) tr[(size_t)(k * hid + c)] = it->second.data[(size_t)(c * 9 + k)];
            if (int rc = dev_alloc(e, &Ly.dw_w9c, (size_t)(hid * 9))) return rc;
            HIP_TRY(hipMemcpy(Ly.dw_w9c, tr.data(), tr.size() * sizeof(float), hipMemcpyHostToDevice));
            e->weight_bytes += (int64_t)tr.size() * 4;
            // halved copies for the fused epilogue (its GELU is written for x / 2; scaling by 0.5 is exact)
            std::vector<float> hb(e->host[LK("mlp.mlp.1.bias")].data);
            for (float& v : tr) v *= 0.5f;
            for (float& v : hb) v *= 0.5f;
            if (int rc = dev_alloc(e, &Ly.dw_w9c_half, tr.size())) return rc;
            if (int rc = dev_alloc(e, &Ly.dw_b_half, hb.size())) return rc;
            HIP_TRY(hipMemcpy(Ly.dw_w9c_half, tr.data(), tr.size() * sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(Ly.dw_b_half, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
            e->weight_bytes += (int64_t)(tr.size() + hb.size()) * 4;
            // packed bf16 tap pairs for the v_dot2c form: per window row du and channel c, with (w0, w1, w2) the halved
            // taps of that row:  kind 0 = (lo 0, hi w0), 1 = (w1, w2), 2 = (w0, w1), 3 = (w2, 0)
            std::vector<uint32_t> pk((size_t)(12 * hid));
            for (int du = 0; du < 3; ++du)
                for (int64_t c = 0; c < hid; ++c) {
                    const uint32_t w0 = f32_to_bf16_rne(tr[(size_t)((du * 3 + 0) * hid + c)]);
                    const uint32_t w1 = f32_to_bf16_rne(tr[(size_t)((du * 3 + 1) * hid + c)]);
                    const uint32_t w2 = f32_to_bf16_rne(tr[(size_t)((du * 3 + 2) * hid + c)]);
                    pk[(size_t)((du * 4 + 0) * hid + c)] = w0 << 16;
                    pk[(size_t)((du * 4 + 1) * hid + c)] = w1 | (w2 << 16);
                    pk[(size_t)((du * 4 + 2) * hid + c)] = w0 | (w1 << 16);
                    pk[(size_t)((du * 4 + 3) * hid + c)] = w2;
                }
            if (int rc = dev_alloc(e, &Ly.dw_wpk, pk.size())) return rc;
            HIP_TRY(hipMemcpy(Ly.dw_wpk, pk.data(), pk.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            e->weight_bytes += (int64_t)pk.size() * 4;
        }
#undef LK
    }
    e->host.clear();
    {   // per-layer pointer tables for the layer-batched conditioning launches
        const size_t L = (size_t)e->L;
        std::vector<const float*> hk(L), hq(L), hg(L), hb(L);
        for (size_t l = 0; l < L; ++l) { hk[l] = e->layers[l].kv_w; hq[l] = e->layers[l].q_w; hg[l] = e->layers[l].n2_w; hb[l] = e->layers[l].n2_b; }
        const float** tabs = nullptr;
        if (int rc = dev_alloc(e, &tabs, 4 * L)) return rc;
        e->tab_kv_w = tabs; e->tab_q_w = tabs + L; e->tab_n2_w = tabs + 2 * L; e->tab_n2_b = tabs + 3 * L;
        HIP_TRY(hipMemcpy(e->tab_kv_w, hk.data(), L * sizeof(float*), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->tab_q_w, hq.data(), L * sizeof(float*), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->tab_n2_w, hg.data(), L * sizeof(float*), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->tab_n2_b, hb.data(), L * sizeof(float*), hipMemcpyHostToDevice));
    }

    const size_t B2 = (size_t)e->cfg.max_batch, M = B2 * e->ntok;
    if (int rc = dev_alloc(e, &e->x, M * d)) return rc;
    if (int rc = dev_alloc(e, &e->x_half, (M + 1) / 2 * d)) return rc;
    if (int rc = dev_alloc(e, &e->xn, M * d)) return rc;
    if (int rc = dev_alloc(e, &e->row_stats, M + 256)) return rc;
    if (int rc = dev_alloc(e, &e->ln_stats, (M + 256) * kLnSlots)) return rc;
    if (int rc = dev_alloc(e, &e->qk, M * 2 * d)) return rc;
    if (int rc = dev_alloc(e, &e->vt, M * d)) return rc;
    if (int rc = dev_alloc(e, &e->att, M * d)) return rc;
    if (int rc = dev_alloc(e, &e->hid1, M * hid)) return rc;
    if (e->grid == 32 && e->fuse_dwconv && hid % 256 == 0) { if (int rc = dev_alloc(e, &e->seam, M * hid / 4)) return rc; }
    if (int rc = dev_alloc(e, &e->hid2, M * hid)) return rc;
    if (e->fp8) {
        if (int rc = dev_alloc(e, &e->a8, M * hid)) return rc;
        if (int rc = dev_alloc(e, &e->as8, M * hid / 32 + 1024)) return rc;
    }
    if (int rc = dev_alloc(e, &e->io_x, B2 * e->img)) return rc;
    if (int rc = dev_alloc(e, &e->io_out, B2 * e->img)) return rc;
    if (int rc = dev_alloc(e, &e->io_sigma, B2)) return rc;
    if (int rc = dev_alloc(e, &e->io_label, B2 * e->text)) return rc;
    if (int rc = dev_alloc(e, &e->xt, B2 * e->img)) return rc;
    if (int rc = dev_alloc(e, &e->x0_prev, B2 * e->img)) return rc;
    if (int rc = dev_alloc(e, &e->x0_cfg, B2 * e->img)) return rc;
    if (int rc = ensure_cond_capacity(e, 2 * (int)B2)) return rc;
    if (int rc = ensure_rows_capacity(e, 64 * (int64_t)B2)) return rc;
    e->finalized = true;
    return TLD_OK;
}

int tld_engine_set_debug(tld_engine* e, int32_t enable) {
    if (!e) return fail(TLD_ERR_INVALID, "null engine");
    e->debug = enable != 0;
    return TLD_OK;
}

int tld_engine_read_stage(tld_engine* e, const char* name, float* host_out, int64_t numel) {
    if (!e || !name || !host_out) return fail(TLD_ERR_INVALID, "null argument");
    DeviceGuard dg(e->cfg.device_id);
    HIP_TRY(hipDeviceSynchronize());
    const float* src = nullptr;
    if (!strcmp(name, "cond_y")) src = e->c_y;
    else {
        auto it = e->stages.find(name);
        if (it != e->stages.end()) src = it->second;
    }
    if (!src) return fail(TLD_ERR_KEY, "no captured stage named %s (was debug enabled before the forward?)", name);
    HIP_TRY(hipMemcpy(host_out, src, numel * sizeof(float), hipMemcpyDeviceToHost));
    return TLD_OK;
}

int tld_denoiser_forward(tld_engine* e, const void* x, const void* noise, const void* label, void* out,
                         int32_t batch, int32_t io_dtype, void* hip_stream) {
    if (!e || !x || !noise || !label || !out) return fail(TLD_ERR_INVALID, "null argument");
    if (!e->finalized) return fail(TLD_ERR_STATE, "weights not finalized");
    if (batch <= 0 || batch > e->cfg.max_batch) return fail(TLD_ERR_INVALID, "batch %d outside (0, max_batch=%d]", batch, e->cfg.max_batch);
    if (io_dtype < 0 || io_dtype > 2) return fail(TLD_ERR_INVALID, "bad io_dtype %d", io_dtype);
    DeviceGuard dg(e->cfg.device_id);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const float* xin = static_cast<const float*>(x);
    float* o = static_cast<float*>(out);
    if (io_dtype != TLD_DTYPE_F32) {
        launch_cast_to_f32(x, io_dtype, e->io_x, (int64_t)batch * e->img, s);
        xin = e->io_x; o = e->io_out;
    }
    // conditioning token rows: [0,batch) noise tokens, [batch, 2*batch) label tokens
    launch_cast_to_f32(noise, io_dtype, e->c_sigma, batch, s);
    launch_cast_to_f32(label, io_dtype, e->c_label, (int64_t)batch * e->text, s);
    {
        ProfScope ps(e, KC_COND, s);
        cond_noise_rows(e, batch, s);
        cond_label_rows(e, batch, batch, s);
    }
    if (int rc = cond_tables(e, 2 * batch, s)) return rc;
    launch_iota(e->rows_dev, 2 * batch, 0, s);
    e->dbg_batch = batch; e->dbg_T = 2 * batch;
    if (int rc = run_body(e, xin, batch, batch, e->rows_dev, e->rows_dev + batch, o, s)) return rc;
    if (io_dtype != TLD_DTYPE_F32) launch_cast_from_f32(e->io_out, out, io_dtype, (int64_t)batch * e->img, s);
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

int tld_sample(tld_engine* e, const void* x_T, const void* labels, const float* coeffs, int32_t n_levels,
               float class_guidance, float sharp_f, float bright_f, void* out_latent, int32_t batch,
               void* trace_x0, void* trace_xt, void* hip_stream) {
    if (!e || !x_T || !labels || !coeffs || !out_latent) return fail(TLD_ERR_INVALID, "null argument");
    if (!e->finalized) return fail(TLD_ERR_STATE, "weights not finalized");
    if (batch <= 0 || 2 * batch > e->cfg.max_batch)
        return fail(TLD_ERR_INVALID, "sampler batch %d needs max_batch >= %d (have %d)", batch, 2 * batch, e->cfg.max_batch);
    if (n_levels < 2) return fail(TLD_ERR_INVALID, "need at least two noise levels");
    DeviceGuard dg(e->cfg.device_id);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const int B = batch, B2 = 2 * batch, T = n_levels + B + 1;
    if (int rc = ensure_cond_capacity(e, T)) return rc;
    if (int rc = ensure_rows_capacity(e, (int64_t)(n_levels + 1) * B2)) return rc;

    // ---- conditioning tables for the whole trajectory, once
    const size_t n_rows = (size_t)(n_levels + 1) * B2;
    if (int rc = stage_acquire(e, (size_t)n_levels * sizeof(float) + n_rows * sizeof(int))) return rc;
    float* sig = static_cast<float*>(e->stage_host);
    int* rows = reinterpret_cast<int*>(sig + n_levels);
    for (int i = 0; i < n_levels; ++i) sig[i] = coeffs[(size_t)i * 6 + 0];
    HIP_TRY(hipMemcpyAsync(e->c_sigma, sig, (size_t)n_levels * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->c_label, labels, (size_t)B * e->text * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemsetAsync(e->c_label + (size_t)B * e->text, 0, e->text * sizeof(float), s));   // uncond = zeros (diffusion.py:61)
    {
        ProfScope ps(e, KC_COND, s);
        cond_noise_rows(e, n_levels, s);
        cond_label_rows(e, n_levels, B + 1, s);
    }
    if (int rc = cond_tables(e, T, s)) return rc;
    // row tables: [step][B2] noise rows, then one [B2] label-row table
    for (int i = 0; i < n_levels; ++i)
        for (int b = 0; b < B2; ++b) rows[(size_t)i * B2 + b] = i;
    for (int b = 0; b < B2; ++b) rows[(size_t)n_levels * B2 + b] = n_levels + (b < B ? b : B);
    HIP_TRY(hipMemcpyAsync(e->rows_dev, rows, n_rows * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(e->stage_ev, s));     // no stream sync: the staging buffer is the engine's, guarded by this event
    const int* label_row = e->rows_dev + (size_t)n_levels * B2;

    const size_t tot = (size_t)B * e->img;
    HIP_TRY(hipMemcpyAsync(e->xt, x_T, tot * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemsetAsync(e->x0_prev, 0, tot * sizeof(float), s));
    e->dbg_batch = B2; e->dbg_T = T;

    for (int i = 0; i < n_levels; ++i) {
        const bool final_step = (i == n_levels - 1);
        // pred_image: model(cat[x_t, x_t], sigma_i, [labels; 0])   (diffusion.py:94-101)
        if (int rc = run_body(e, e->xt, B, B2, e->rows_dev + (size_t)i * B2, label_row, e->io_out, s, true)) return rc;
        ProfScope ps(e, KC_UPDATE, s);
        UpdateParams up{};
        const float* c = coeffs + (size_t)i * 6;
        up.x0_2b = e->io_out; up.x_t = e->xt; up.x0_prev = e->x0_prev;
        up.x0_out = final_step ? static_cast<float*>(out_latent) : e->x0_cfg;
        up.trace_x0 = (!final_step && trace_x0) ? static_cast<float*>(trace_x0) + (size_t)i * tot : nullptr;
        up.trace_xt = (!final_step && trace_xt) ? static_cast<float*>(trace_xt) + (size_t)i * tot : nullptr;
        up.g = class_guidance; up.a = c[1]; up.b = c[2]; up.c = c[3]; up.c1 = c[4]; up.c2 = c[5];
        up.sharp = sharp_f; up.bright = bright_f; up.final_step = final_step ? 1 : 0;
        up.batch = B; up.img = e->img; up.chan_stride = e->cfg.image_size * e->cfg.image_size; up.C = e->cfg.n_channels;
        launch_update(up, s);
    }
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

int tld_debug_gemm_bf16(const void* a, const void* w, float* c, int32_t M, int32_t N, int32_t K, void* hip_stream) {
    if (!a || !w || !c) return fail(TLD_ERR_INVALID, "null argument");
    if (K % 64 || K <= 0 || M <= 0 || N <= 0) return fail(TLD_ERR_INVALID, "need K %% 64 == 0 and positive sizes");
    if ((int64_t)M * K * 2 >= (int64_t)1 << 32 || (int64_t)N * K * 2 >= (int64_t)1 << 32)
        return fail(TLD_ERR_INVALID, "operands must be smaller than 4 GiB");
    GemmParams g{};
    g.A = static_cast<const bf16*>(a); g.lda = K; g.W = static_cast<const bf16*>(w); g.ldw = K;
    g.M = M; g.N = N; g.K = K; g.c_f32 = c; g.ldc = N;
    launch_gemm(g, EPI_F32, static_cast<hipStream_t>(hip_stream));
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

int tld_debug_gemm_splitk(const void* a, const void* w, float* c_slices, int32_t M, int32_t N, int32_t K, int32_t ksplit, void* hip_stream) {
    if (!a || !w || !c_slices) return fail(TLD_ERR_INVALID, "null argument");
    if (ksplit <= 0 || K <= 0 || K % (64 * ksplit) || M <= 0 || N <= 0) return fail(TLD_ERR_INVALID, "need K %% (64 ksplit) == 0 and positive sizes");
    if ((int64_t)M * K * 2 >= (int64_t)1 << 32 || (int64_t)N * K * 2 >= (int64_t)1 << 32) return fail(TLD_ERR_INVALID, "operands must be smaller than 4 GiB");
    GemmParams g{};
    g.A = static_cast<const bf16*>(a); g.lda = K; g.W = static_cast<const bf16*>(w); g.ldw = K;
    g.M = M; g.N = N; g.K = K / ksplit; g.ksplit = ksplit; g.c_f32 = c_slices; g.ldc = N;
    launch_gemm(g, EPI_F32, static_cast<hipStream_t>(hip_stream));
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

int tld_debug_gemm_bench(int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t ntok, int32_t iters,
                         double* avg_ms) {
    if (!avg_ms || M <= 0 || N <= 0 || K <= 0 || K % 64 || iters <= 0) return fail(TLD_ERR_INVALID, "bad argument");
    if ((int64_t)M * K * 2 >= (int64_t)1 << 32 || (int64_t)N * K * 2 >= (int64_t)1 << 32)
        return fail(TLD_ERR_INVALID, "operands must be smaller than 4 GiB");
    if (epilogue == EPI_QKV && (N % 3 || ntok <= 0 || M % ntok)) return fail(TLD_ERR_INVALID, "QKV epilogue needs N = 3d, M %% ntok == 0");
    bf16 *A = nullptr, *W = nullptr, *out = nullptr, *vt = nullptr;
    float *bias = nullptr, *res = nullptr;   // res doubles as fp32 C and as the residual buffer (sized for fp32)
    HIP_TRY(hipMalloc(&A, (size_t)M * K * 2)); HIP_TRY(hipMalloc(&W, (size_t)N * K * 2));
    HIP_TRY(hipMalloc(&out, (size_t)M * N * 2)); HIP_TRY(hipMalloc(&vt, (size_t)M * N * 2));
    HIP_TRY(hipMalloc(&bias, (size_t)N * 4)); HIP_TRY(hipMalloc(&res, (size_t)M * N * 4));
    const float zs = getenv("TLD_GEMM_ZERO") ? 0.0f : 1.0f;     // DVFS experiment: all-zero operands
    launch_fill_bf16(A, (int64_t)M * K, 1u, 1.0f * zs, nullptr);
    const float wsc = getenv("TLD_GEMM_WSCALE") ? (float)atof(getenv("TLD_GEMM_WSCALE")) : 0.05f;     // 1.0: both operands uniform in [-1, 1) (the GEMM template's benchmark data)
    launch_fill_bf16(W, (int64_t)N * K, 2u, wsc * zs, nullptr);
    HIP_TRY(hipMemset(bias, 0, (size_t)N * 4)); HIP_TRY(hipMemset(res, 0, (size_t)M * N * 4));
    GemmParams g{};
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K;
    g.c_f32 = res; g.ldc = N; g.out_bf16 = out; g.ldo = epilogue == EPI_QKV ? 2 * (N / 3) : N; g.vt = vt;
    g.ntok = ntok > 0 ? ntok : 1; g.d = N / 3; g.bias = bias; g.resid = reinterpret_cast<resid_t*>(res); g.ldr = N;
    g.dbg_epi = getenv("TLD_EPI_DBG") ? atoi(getenv("TLD_EPI_DBG")) : 0;
    float *dww = nullptr;
    if (epilogue == EPI_UP_DWCONV2) {
        if (M % 256 || N % 256) return fail(TLD_ERR_INVALID, "fused depthwise epilogue needs M, N multiples of 256");
        HIP_TRY(hipMalloc(&dww, (size_t)N * 22 * 4));
        std::vector<float> hw((size_t)N * 22);
        for (size_t i = 0; i < (size_t)N * 10; ++i) hw[i] = 0.05f + 0.01f * (float)(i % 7);
        uint32_t* pk = reinterpret_cast<uint32_t*>(hw.data() + (size_t)N * 10);
        for (size_t i = 0; i < (size_t)N * 12; ++i) pk[i] = 0x3d803d00u + (uint32_t)(i % 5) * 0x00010001u;     // small bf16 pairs
        HIP_TRY(hipMemcpy(dww, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        g.dw_b = dww + (size_t)N * 9; g.dw_wpk = reinterpret_cast<const uint32_t*>(dww + (size_t)N * 10);
    }
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
    auto run = [&]() { launch_gemm(g, epilogue, nullptr); };
    for (int i = 0; i < 3; ++i) run();
    HIP_TRY(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) run();
    HIP_TRY(hipEventRecord(b, nullptr));
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    *avg_ms = ms / iters;
    hipEventDestroy(a); hipEventDestroy(b);
    hipFree(A); hipFree(W); hipFree(out); hipFree(vt); hipFree(bias); hipFree(res); hipFree(dww);
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

int tld_engine_set_low_latency(tld_engine* e, int32_t on) {
    if (!e) return fail(TLD_ERR_INVALID, "null engine");
    DeviceGuard dg(e->cfg.device_id);
    if (!on) { e->low_latency = false; return TLD_OK; }
    if (on != 1 && on != 2) return fail(TLD_ERR_INVALID, "low-latency class %d: 0 = off, 1 = four K-splits (engines up to %d token rows), 2 = eight (up to %d)", on, kLowLatMaxRows, kLowLatMaxRows2);
    if (e->fp8)
        return fail(TLD_ERR_INVALID, "low-latency class: bf16 GEMM operands only -- this engine runs MX-fp8 GEMMs (tld_engine_set_gemm_dtype), whose down projection has no split-K form");
    const int64_t rows = (int64_t)e->cfg.max_batch * e->ntok;
    const int max_rows = on == 2 ? kLowLatMaxRows2 : kLowLatMaxRows;
    if (rows > max_rows)
        return fail(TLD_ERR_INVALID, "low-latency class %d: engine capacity %lld token rows (max_batch %d x %d tokens) exceeds %d -- at that size %s",
                    on, (long long)rows, e->cfg.max_batch, e->ntok, max_rows, on == 2 ? "class 1 (four K-splits) is the faster one" : "the default tiles fill the chip");
    const int split = on == 2 ? 8 : 4;
    if (!splitk_resid_supported(e->d) || e->hid % (64 * split) != 0)
        return fail(TLD_ERR_INVALID, "low-latency class: needs embed_dim 384 or 768 (got %d) and a hidden width that splits into %d multiples of 64 (got %d)", e->d, split, e->hid);
    if (e->splitk && e->ll_split < split) e->splitk = nullptr;      // (a class-1 buffer is too small for eight slices: allocate anew; the old one is released with the engine)
    if (!e->splitk) { if (int rc = dev_alloc(e, &e->splitk, (size_t)split * rows * e->d)) return rc; }
    e->ll_split = split;
    e->low_latency = true;
    return TLD_OK;
}

int tld_engine_set_gemm_dtype(tld_engine* e, int32_t dtype) {
    if (!e) return fail(TLD_ERR_INVALID, "null engine");
    if (e->finalized) return fail(TLD_ERR_STATE, "the GEMM operand type must be chosen before tld_engine_finalize_weights");
    if (dtype != 0 && dtype != 1) return fail(TLD_ERR_INVALID, "gemm dtype %d: 0 = bf16, 1 = MX-fp8 (e4m3 + E8M0 block scales)", dtype);
    if (dtype == 1) {
        if (e->low_latency) return fail(TLD_ERR_INVALID, "fp8 GEMMs and the low-latency class exclude each other (the split-K down projection is bf16 only): clear tld_engine_set_low_latency first");
#ifndef TLD_RESID_BF16
        return fail(TLD_ERR_INVALID, "the fp8 GEMM mode exists in the bf16-residual build only");
#endif
        if (e->d % 128 || e->hid % 128) return fail(TLD_ERR_INVALID, "fp8 GEMMs need embed_dim and hidden width multiples of 128");
        e->fp8 = true;
        // activations are quantised from the separately written LayerNorm / GELU outputs: no folds, no fused depthwise epilogue
        e->fold_ln1 = false; e->fold_ln3 = false; e->fuse_dwconv = false;
    } else {
        e->fp8 = false;
    }
    return TLD_OK;
}

int tld_debug_quant_mx8_host(const float* w, int32_t rows, int32_t K, void* out_e4m3, void* out_scale) {
    if (!w || !out_e4m3 || !out_scale || rows <= 0 || K <= 0 || K % 128) return fail(TLD_ERR_INVALID, "bad argument (K %% 128 == 0)");
    quant_mx8_host(w, rows, K, static_cast<uint8_t*>(out_e4m3), static_cast<uint8_t*>(out_scale));
    return TLD_OK;
}

int tld_debug_quant_mx8(const void* in_bf16, void* out_e4m3, void* out_scale, int32_t M, int32_t K, void* hip_stream) {
    if (!in_bf16 || !out_e4m3 || !out_scale || M <= 0 || K <= 0 || K % 128) return fail(TLD_ERR_INVALID, "bad argument (K %% 128 == 0)");
    PtrDeviceGuard guard(in_bf16);
    launch_quant_mx8(static_cast<const bf16*>(in_bf16), static_cast<uint8_t*>(out_e4m3), static_cast<uint8_t*>(out_scale), M, K,
                     static_cast<hipStream_t>(hip_stream));
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

int tld_debug_gemm_mx8(const void* a_e4m3, const void* a_scale, const void* w_e4m3, const void* w_scale, float* c, int32_t M,
                       int32_t N, int32_t K, void* hip_stream) {
    if (!a_e4m3 || !a_scale || !w_e4m3 || !w_scale || !c) return fail(TLD_ERR_INVALID, "null argument");
    if (K % 128 || K <= 0 || M <= 0 || N <= 0 || M % 4 || N % 4) return fail(TLD_ERR_INVALID, "need K %% 128 == 0, M %% 4 == 0, N %% 4 == 0");
    PtrDeviceGuard guard(a_e4m3);
    GemmParams g{};
    g.f8 = 1;
    g.A = static_cast<const bf16*>(a_e4m3); g.lda = K; g.W = static_cast<const bf16*>(w_e4m3); g.ldw = K;
    g.a_scale = static_cast<const uint8_t*>(a_scale); g.w_scale = static_cast<const uint8_t*>(w_scale);
    g.M = M; g.N = N; g.K = K; g.c_f32 = c; g.ldc = N;
    launch_gemm(g, EPI_F32, static_cast<hipStream_t>(hip_stream));
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

int tld_engine_set_profile(tld_engine* e, uint32_t class_mask) {
    if (!e) return fail(TLD_ERR_INVALID, "null engine");
    for (int k = 0; k < KC_COUNT; ++k) e->prof_used[k] = 0;       // recorded pairs are forgotten, the pool is kept
    e->prof_mask = class_mask;
    return TLD_OK;
}

int tld_engine_profile_reserve(tld_engine* e, int32_t kclass, int64_t launches) {
    if (!e) return fail(TLD_ERR_INVALID, "null engine");
    if (kclass < 0 || kclass >= KC_COUNT || launches < 0) return fail(TLD_ERR_INVALID, "bad kernel class %d / count", kclass);
    DeviceGuard dg(e->cfg.device_id);
    auto& pool = e->prof_ev[kclass];
    while ((int64_t)pool.size() < launches) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
        pool.emplace_back(a, b);
    }
    return TLD_OK;
}

int tld_engine_get_profile(tld_engine* e, int32_t kclass, double* total_ms, int64_t* launches) {
    if (!e || !total_ms || !launches) return fail(TLD_ERR_INVALID, "null argument");
    if (kclass < 0 || kclass >= KC_COUNT) return fail(TLD_ERR_INVALID, "bad kernel class %d", kclass);
    DeviceGuard dg(e->cfg.device_id);
    HIP_TRY(hipDeviceSynchronize());
    double tot = 0.0;
    for (size_t i = 0; i < e->prof_used[kclass]; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->prof_ev[kclass][i].first, e->prof_ev[kclass][i].second));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int64_t)e->prof_used[kclass];
    return TLD_OK;
}

int64_t tld_engine_weight_bytes(const tld_engine* e) { return e ? e->weight_bytes : 0; }

int tld_debug_dwconv_gelu(const void* in, const float* weight, const float* bias, void* out, int32_t batch, int32_t grid, int32_t channels,
                          void* hip_stream) {
    if (!in || !weight || !bias || !out || batch <= 0 || grid <= 0 || channels <= 0 || channels % 64) return fail(TLD_ERR_INVALID, "bad argument");
    if (grid > 16 && grid % 16) return fail(TLD_ERR_INVALID, "grids wider than 16 tokens must be a multiple of 16 (the engine's token counts are multiples of 256)");
    PtrDeviceGuard guard(in);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const size_t nc = (size_t)channels;
    std::vector<float> tab(nc * 20);                       // [9][C] taps | [C] bias | the same halved
    for (size_t c = 0; c < nc; ++c) {
        for (int k = 0; k < 9; ++k) { tab[k * nc + c] = weight[c * 9 + k]; tab[nc * 10 + k * nc + c] = 0.5f * weight[c * 9 + k]; }
        tab[9 * nc + c] = bias[c]; tab[nc * 19 + c] = 0.5f * bias[c];
    }
    float* d = nullptr;
    HIP_TRY(hipMalloc(&d, tab.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(d, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    launch_dwconv_gelu(static_cast<const bf16*>(in), static_cast<bf16*>(out), d, d + 9 * nc, d + 10 * nc, d + 19 * nc, batch, grid, channels, s,
                       nullptr, nullptr);
    HIP_TRY(hipStreamSynchronize(s));
    (void)hipFree(d);
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

int tld_debug_attention_fwd(const void* qk, const void* vt, void* att, int32_t batch, int32_t ntok, int32_t heads, int32_t iters,
                            float* ms_per_launch, void* hip_stream) {
    if (!qk || !vt || !att || batch <= 0 || heads <= 0 || iters <= 0) return fail(TLD_ERR_INVALID, "bad argument");
    if (ntok <= 0 || ntok % 8 != 0)
        return fail(TLD_ERR_INVALID, "attention supports token counts that are multiples of 8 (got %d)", ntok);
    PtrDeviceGuard guard(qk);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        launch_attention(static_cast<const bf16*>(qk), static_cast<const bf16*>(vt), static_cast<bf16*>(att), batch, ntok, heads, s);
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (ms_per_launch) *ms_per_launch = ms / (float)iters;
    HIP_TRY(hipGetLastError());
    return TLD_OK;
}

int tld_engine_destroy(tld_engine* e) {
    if (!e) return TLD_OK;
    DeviceGuard dg(e->cfg.device_id);
    (void)hipDeviceSynchronize();
    for (int k = 0; k < KC_COUNT; ++k)
        for (auto& ev : e->prof_ev[k]) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->stage_host) (void)hipHostFree(e->stage_host);
    if (e->stage_ev) (void)hipEventDestroy(e->stage_ev);
    delete e;
    return TLD_OK;
}

}  // extern "C"
