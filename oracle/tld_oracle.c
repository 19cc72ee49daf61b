/*
 * tld_oracle.c -- CPU restatement (fp32, plain C + OpenMP) of the reference's denoising hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the shipped engine (libtld_hip.so) never links or
 * calls anything in this directory.
 *
 * What it restates (all citations relative to /root/reference):
 *   SinusoidalEmbedding.forward            tld/transformer_blocks.py:17-21
 *   MHAttention / SelfAttention            tld/transformer_blocks.py:31-48, 57-59
 *   CrossAttention                         tld/transformer_blocks.py:69-72
 *   MLPSepConv                             tld/transformer_blocks.py:92-113
 *   DecoderBlock.forward                   tld/transformer_blocks.py:135-139
 *   DenoiserTransBlock (patchify/pos/out)  tld/denoiser.py:34-52, 74-82
 *   Denoiser.forward                       tld/denoiser.py:116-126
 *   DiffusionGenerator.generate/pred_image tld/diffusion.py:29-103, 122-125
 *
 * Parity status: PINNED -- checked in tests/test_oracle_golden.py against golden vectors captured by
 * importing the reference in the build container (oracle/gen_golden.py -> tests/golden/ (npz files)).
 * The reference's own tests hold no numerical assertions (tests/test_diffuser.py:45,93 are shape /
 * type checks only), so captured outputs are the only pin available.
 *
 * All arithmetic is float32 with sequential-k accumulation; the sampler's scalar coefficients
 * are float64 on the host exactly as the reference computes them (diffusion.py:54-57), then the
 * tensor update happens in float32 (model_dtype = float32 path of the reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define TLD_O_EXPORT __attribute__((visibility("default")))

typedef struct {
    float *qkv_w;            /* [3d, d]   rows ordered q;k;v  (transformer_blocks.py:58) */
    float *kv_w;             /* [2d, d]   rows ordered k;v    (transformer_blocks.py:71) */
    float *q_w;              /* [d, d] */
    float *up_w, *up_b;      /* [4d, d], [4d]     1x1 conv == linear (transformer_blocks.py:95) */
    float *dw_w, *dw_b;      /* [4d, 3, 3], [4d]  depthwise 3x3     (transformer_blocks.py:96-102) */
    float *down_w, *down_b;  /* [d, 4d], [d]      (transformer_blocks.py:104) */
    float *n1_w, *n1_b, *n2_w, *n2_b, *n3_w, *n3_b;
} tld_o_layer;

typedef struct tld_o_model {
    int image_size, noise_embed_dims, patch_size, embed_dim, n_layers, text_emb_size, n_channels,
        mlp_multiplier;
    int seq_len, patch_dim, hidden, n_heads, head_dim, grid; /* derived */
    float *angular;                 /* [ne/2] */
    float *ff1_w, *ff1_b;           /* [d, ne], [d] */
    float *ff3_w, *ff3_b;           /* [d, d], [d] */
    float *label_w, *label_b;       /* [d, text], [d] */
    float *norm_w, *norm_b;         /* [d] */
    float *pconv_w, *pconv_b;       /* [pd, C, p, p], [pd] */
    float *pln1_w, *pln1_b;         /* [pd] */
    float *plin_w, *plin_b;         /* [d, pd], [d] */
    float *pln2_w, *pln2_b;         /* [d] */
    float *pos;                     /* [N, d] */
    float *out_w, *out_b;           /* [pd, d], [pd] */
    tld_o_layer *layers;
} tld_o_model;

static float *falloc(size_t n) {
    float *p = (float *)calloc(n ? n : 1, sizeof(float));
    if (!p) { fprintf(stderr, "tld_oracle: out of memory (%zu floats)\n", n); abort(); }
    return p;
}

TLD_O_EXPORT tld_o_model *tld_o_create(int image_size, int noise_embed_dims, int patch_size,
                                       int embed_dim, int n_layers, int text_emb_size,
                                       int n_channels, int mlp_multiplier) {
    tld_o_model *m = (tld_o_model *)calloc(1, sizeof(*m));
    m->image_size = image_size; m->noise_embed_dims = noise_embed_dims; m->patch_size = patch_size;
    m->embed_dim = embed_dim; m->n_layers = n_layers; m->text_emb_size = text_emb_size;
    m->n_channels = n_channels; m->mlp_multiplier = mlp_multiplier;
    /* denoiser.py:31-32 */
    m->seq_len = (int)(((double)image_size / patch_size) * ((double)image_size / patch_size));
    m->grid = image_size / patch_size;
    m->patch_dim = n_channels * patch_size * patch_size;
    m->hidden = mlp_multiplier * embed_dim;
    m->n_heads = embed_dim / 64;                 /* transformer_blocks.py:126,128 */
    if (m->n_heads < 1) m->n_heads = 1;
    m->head_dim = embed_dim / m->n_heads;
    int d = embed_dim, ne = noise_embed_dims, pd = m->patch_dim, hid = m->hidden;
    m->angular = falloc(ne / 2);
    m->ff1_w = falloc((size_t)d * ne); m->ff1_b = falloc(d);
    m->ff3_w = falloc((size_t)d * d);  m->ff3_b = falloc(d);
    m->label_w = falloc((size_t)d * text_emb_size); m->label_b = falloc(d);
    m->norm_w = falloc(d); m->norm_b = falloc(d);
    m->pconv_w = falloc((size_t)pd * pd); m->pconv_b = falloc(pd);
    m->pln1_w = falloc(pd); m->pln1_b = falloc(pd);
    m->plin_w = falloc((size_t)d * pd); m->plin_b = falloc(d);
    m->pln2_w = falloc(d); m->pln2_b = falloc(d);
    m->pos = falloc((size_t)m->seq_len * d);
    m->out_w = falloc((size_t)pd * d); m->out_b = falloc(pd);
    m->layers = (tld_o_layer *)calloc(n_layers, sizeof(tld_o_layer));
    for (int l = 0; l < n_layers; ++l) {
        tld_o_layer *L = &m->layers[l];
        L->qkv_w = falloc((size_t)3 * d * d); L->kv_w = falloc((size_t)2 * d * d);
        L->q_w = falloc((size_t)d * d);
        L->up_w = falloc((size_t)hid * d); L->up_b = falloc(hid);
        L->dw_w = falloc((size_t)hid * 9); L->dw_b = falloc(hid);
        L->down_w = falloc((size_t)d * hid); L->down_b = falloc(d);
        L->n1_w = falloc(d); L->n1_b = falloc(d); L->n2_w = falloc(d); L->n2_b = falloc(d);
        L->n3_w = falloc(d); L->n3_b = falloc(d);
    }
    return m;
}

TLD_O_EXPORT void tld_o_destroy(tld_o_model *m) {
    if (!m) return;
    free(m->angular); free(m->ff1_w); free(m->ff1_b); free(m->ff3_w); free(m->ff3_b);
    free(m->label_w); free(m->label_b); free(m->norm_w); free(m->norm_b);
    free(m->pconv_w); free(m->pconv_b); free(m->pln1_w); free(m->pln1_b);
    free(m->plin_w); free(m->plin_b); free(m->pln2_w); free(m->pln2_b); free(m->pos);
    free(m->out_w); free(m->out_b);
    for (int l = 0; l < m->n_layers; ++l) {
        tld_o_layer *L = &m->layers[l];
        free(L->qkv_w); free(L->kv_w); free(L->q_w); free(L->up_w); free(L->up_b); free(L->dw_w);
        free(L->dw_b); free(L->down_w); free(L->down_b); free(L->n1_w); free(L->n1_b);
        free(L->n2_w); free(L->n2_b); free(L->n3_w); free(L->n3_b);
    }
    free(m->layers); free(m);
}

/* state_dict key -> destination buffer.  Returns expected numel, or -1 if the key is unknown,
 * 0 for buffers that are accepted and ignored (precomputed_pos_enc is arange). */
static long resolve(tld_o_model *m, const char *key, float **dst) {
    int d = m->embed_dim, ne = m->noise_embed_dims, pd = m->patch_dim, hid = m->hidden;
    *dst = NULL;
#define K(name, ptr, n) if (!strcmp(key, name)) { *dst = (ptr); return (long)(n); }
    K("fourier_feats.0.angular_speeds", m->angular, ne / 2)
    K("fourier_feats.1.weight", m->ff1_w, (long)d * ne)
    K("fourier_feats.1.bias", m->ff1_b, d)
    K("fourier_feats.3.weight", m->ff3_w, (long)d * d)
    K("fourier_feats.3.bias", m->ff3_b, d)
    K("label_proj.weight", m->label_w, (long)d * m->text_emb_size)
    K("label_proj.bias", m->label_b, d)
    K("norm.weight", m->norm_w, d)
    K("norm.bias", m->norm_b, d)
    const char *pre = "denoiser_trans_block.";
    size_t pl = strlen(pre);
    if (strncmp(key, pre, pl)) return -1;
    key += pl;
    if (!strcmp(key, "precomputed_pos_enc")) return 0;
    K("patchify_and_embed.0.weight", m->pconv_w, (long)pd * pd)
    K("patchify_and_embed.0.bias", m->pconv_b, pd)
    K("patchify_and_embed.2.weight", m->pln1_w, pd)
    K("patchify_and_embed.2.bias", m->pln1_b, pd)
    K("patchify_and_embed.3.weight", m->plin_w, (long)d * pd)
    K("patchify_and_embed.3.bias", m->plin_b, d)
    K("patchify_and_embed.4.weight", m->pln2_w, d)
    K("patchify_and_embed.4.bias", m->pln2_b, d)
    K("pos_embed.weight", m->pos, (long)m->seq_len * d)
    K("out_proj.0.weight", m->out_w, (long)pd * d)
    K("out_proj.0.bias", m->out_b, pd)
    int li = -1, off = 0;
    if (sscanf(key, "decoder_blocks.%d.%n", &li, &off) != 1 || li < 0 || li >= m->n_layers || !off)
        return -1;
    key += off;
    tld_o_layer *L = &m->layers[li];
    K("self_attention.qkv_linear.weight", L->qkv_w, (long)3 * d * d)
    K("cross_attention.kv_linear.weight", L->kv_w, (long)2 * d * d)
    K("cross_attention.q_linear.weight", L->q_w, (long)d * d)
    K("mlp.mlp.0.weight", L->up_w, (long)hid * d)
    K("mlp.mlp.0.bias", L->up_b, hid)
    K("mlp.mlp.1.weight", L->dw_w, (long)hid * 9)
    K("mlp.mlp.1.bias", L->dw_b, hid)
    K("mlp.mlp.3.weight", L->down_w, (long)d * hid)
    K("mlp.mlp.3.bias", L->down_b, d)
    K("norm1.weight", L->n1_w, d) K("norm1.bias", L->n1_b, d)
    K("norm2.weight", L->n2_w, d) K("norm2.bias", L->n2_b, d)
    K("norm3.weight", L->n3_w, d) K("norm3.bias", L->n3_b, d)
#undef K
    return -1;
}

/* copy one fp32 state_dict tensor in; 0 ok, 1 unknown key, 2 size mismatch */
TLD_O_EXPORT int tld_o_set_tensor(tld_o_model *m, const char *key, const float *src, long numel) {
    float *dst;
    long n = resolve(m, key, &dst);
    if (n < 0) return 1;
    if (n == 0) return 0;
    if (n != numel) return 2;
    memcpy(dst, src, (size_t)n * sizeof(float));
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* primitives                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* out[M,N] = x[M,K] . W[N,K]^T (+ b[N]); nn.Linear semantics; W row-major like the state_dict.
 * W is repacked once per call into 16-column panels [N/16][K][16]; blocks of 96 rows are distributed
 * over threads and each (row block, panel) pair runs a 6x16 register tile with k ascending -- one
 * fp32 accumulation chain per output element, 8-wide vector FMAs. */
typedef float v8f __attribute__((vector_size(32), aligned(4)));
#define MR 6
#define NR 16
static void linear(const float *x, const float *W, const float *b, int M, int K, int N, float *out) {
    int npan = (N + NR - 1) / NR;
    float *Wp = (float *)aligned_alloc(64, ((size_t)npan * K * NR * sizeof(float) + 63) / 64 * 64);
    if (!Wp) abort();
#pragma omp parallel for schedule(static)
    for (int pn = 0; pn < npan; ++pn) {
        float *dst = Wp + (size_t)pn * K * NR;
        for (int j = 0; j < NR; ++j) {
            int n = pn * NR + j;
            if (n < N) for (int k = 0; k < K; ++k) dst[(size_t)k * NR + j] = W[(size_t)n * K + k];
            else for (int k = 0; k < K; ++k) dst[(size_t)k * NR + j] = 0.f;
        }
    }
    int nthr = 1;
#ifdef _OPENMP
    nthr = omp_get_max_threads();
#endif
    int MB = (M / (2 * nthr)) / MR * MR;          /* >= 2 row blocks per thread when M allows */
    if (MB > 96) MB = 96;
    if (MB < MR) MB = MR;
    int mblocks = (M + MB - 1) / MB;
#pragma omp parallel for schedule(dynamic, 1)
    for (int mb = 0; mb < mblocks; ++mb) {
        int mlo = mb * MB, mhi = mlo + MB < M ? mlo + MB : M;
        for (int pn = 0; pn < npan; ++pn) {
            const float *wp = Wp + (size_t)pn * K * NR;
            int n0 = pn * NR;
            for (int m0 = mlo; m0 < mhi; m0 += MR) {
                const float *xr[MR];
                for (int i = 0; i < MR; ++i) {
                    int r = m0 + i < mhi ? m0 + i : mhi - 1;   /* clamp: duplicate rows are not stored */
                    xr[i] = x + (size_t)r * K;
                }
                v8f acc[MR][2];
                for (int i = 0; i < MR; ++i) { acc[i][0] = (v8f){0}; acc[i][1] = (v8f){0}; }
                for (int k = 0; k < K; ++k) {
                    v8f b0 = *(const v8f *)(wp + (size_t)k * NR);
                    v8f b1 = *(const v8f *)(wp + (size_t)k * NR + 8);
                    for (int i = 0; i < MR; ++i) {
                        float a = xr[i][k];
                        v8f av = {a, a, a, a, a, a, a, a};
                        acc[i][0] += av * b0;
                        acc[i][1] += av * b1;
                    }
                }
                for (int i = 0; i < MR && m0 + i < mhi; ++i) {
                    float tmp[NR];
                    *(v8f *)tmp = acc[i][0]; *(v8f *)(tmp + 8) = acc[i][1];
                    float *o = out + (size_t)(m0 + i) * N + n0;
                    int lim = N - n0 < NR ? N - n0 : NR;
                    for (int j = 0; j < lim; ++j) o[j] = tmp[j] + (b ? b[n0 + j] : 0.f);
                }
            }
        }
    }
    free(Wp);
}

/* nn.LayerNorm: biased variance, eps 1e-5, affine (Appendix A of SURVEY.md) */
static void layernorm(const float *x, const float *g, const float *b, long M, int D, float *out) {
#pragma omp parallel for schedule(static)
    for (long m = 0; m < M; ++m) {
        const float *r = x + (size_t)m * D;
        float *o = out + (size_t)m * D;
        double s = 0.0;
        for (int k = 0; k < D; ++k) s += r[k];
        float mean = (float)(s / D);
        double v = 0.0;
        for (int k = 0; k < D; ++k) { double t = (double)r[k] - mean; v += t * t; }
        float rstd = 1.0f / sqrtf((float)(v / D) + 1e-5f);
        for (int k = 0; k < D; ++k) o[k] = (r[k] - mean) * rstd * g[k] + b[k];
    }
}

static inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

/* SinusoidalEmbedding.forward: cat[sin(w*s), cos(w*s)]  (transformer_blocks.py:17-21) */
TLD_O_EXPORT void tld_o_sinusoid(const tld_o_model *m, const float *sigma, int B, float *out) {
    int h = m->noise_embed_dims / 2;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < h; ++k) {
            float ph = m->angular[k] * sigma[b];     /* float32 product, as torch does */
            out[(size_t)b * 2 * h + k] = sinf(ph);
            out[(size_t)b * 2 * h + h + k] = cosf(ph);
        }
}

/* conditioning tokens y[B,2,d] = LN(cat[fourier(sigma), label_proj(label)])  (denoiser.py:117-122) */
TLD_O_EXPORT void tld_o_cond(const tld_o_model *m, const float *sigma, const float *label, int B,
                             float *sin_emb_out /* [B,ne] or NULL */, float *y /* [B,2,d] */) {
    int d = m->embed_dim, ne = m->noise_embed_dims;
    float *se = falloc((size_t)B * ne), *h1 = falloc((size_t)B * d), *h2 = falloc((size_t)B * d),
          *lp = falloc((size_t)B * d), *cat = falloc((size_t)B * 2 * d);
    tld_o_sinusoid(m, sigma, B, se);
    if (sin_emb_out) memcpy(sin_emb_out, se, (size_t)B * ne * sizeof(float));
    linear(se, m->ff1_w, m->ff1_b, B, ne, d, h1);
    for (size_t i = 0; i < (size_t)B * d; ++i) h1[i] = gelu_erf(h1[i]);
    linear(h1, m->ff3_w, m->ff3_b, B, d, d, h2);
    linear(label, m->label_w, m->label_b, B, m->text_emb_size, d, lp);
    for (int b = 0; b < B; ++b) {                      /* token order [noise, label] */
        memcpy(cat + (size_t)b * 2 * d, h2 + (size_t)b * d, d * sizeof(float));
        memcpy(cat + (size_t)b * 2 * d + d, lp + (size_t)b * d, d * sizeof(float));
    }
    layernorm(cat, m->norm_w, m->norm_b, (long)B * 2, d, y);
    free(se); free(h1); free(h2); free(lp); free(cat);
}

/* patchify_and_embed + pos_embed -> tokens [B,N,d]  (denoiser.py:34-45, 75-77) */
TLD_O_EXPORT void tld_o_embed(const tld_o_model *m, const float *x, int B, float *tok) {
    int C = m->n_channels, S = m->image_size, p = m->patch_size, g = m->grid, pd = m->patch_dim,
        d = m->embed_dim, N = m->seq_len;
    float *pt = falloc((size_t)B * N * pd), *ptn = falloc((size_t)B * N * pd),
          *e = falloc((size_t)B * N * d);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < N; ++t) {
            int i = t / g, j = t % g;                   /* Rearrange "bs d h w -> bs (h w) d" */
            for (int o = 0; o < pd; ++o) {
                float s = m->pconv_b[o];
                for (int c = 0; c < C; ++c)
                    for (int u = 0; u < p; ++u)
                        for (int v = 0; v < p; ++v)
                            s += m->pconv_w[((o * C + c) * p + u) * p + v] *
                                 x[(((size_t)b * C + c) * S + (i * p + u)) * S + (j * p + v)];
                pt[((size_t)b * N + t) * pd + o] = s;
            }
        }
    layernorm(pt, m->pln1_w, m->pln1_b, (long)B * N, pd, ptn);
    linear(ptn, m->plin_w, m->plin_b, B * N, pd, d, e);
    layernorm(e, m->pln2_w, m->pln2_b, (long)B * N, d, tok);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)B * N; ++r) {
        int t = (int)(r % N);
        for (int k = 0; k < d; ++k) tok[(size_t)r * d + k] += m->pos[(size_t)t * d + k];
    }
    free(pt); free(ptn); free(e);
}

/* softmax(q k^T / sqrt(hd)) v, heads = contiguous hd-wide column groups
 * ("bs n (h d) -> bs h n d", transformer_blocks.py:35); q:[B,Nq,*] k,v:[B,Nk,*] with row strides. */
static void mha(const float *q, int ldq, const float *k, int ldk, const float *v, int ldv, int B,
                int Nq, int Nk, int H, int hd, float *out, int ldo) {
    float scale = 1.0f / sqrtf((float)hd);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h) {
            float *s = (float *)malloc(sizeof(float) * Nk);
            for (int i = 0; i < Nq; ++i) {
                const float *qi = q + ((size_t)b * Nq + i) * ldq + h * hd;
                float mx = -INFINITY;
                for (int j = 0; j < Nk; ++j) {
                    const float *kj = k + ((size_t)b * Nk + j) * ldk + h * hd;
                    float a = 0.f;
                    for (int e = 0; e < hd; ++e) a += qi[e] * kj[e];
                    a *= scale; s[j] = a; if (a > mx) mx = a;
                }
                float den = 0.f;
                for (int j = 0; j < Nk; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
                float inv = 1.0f / den;
                float *o = out + ((size_t)b * Nq + i) * ldo + h * hd;
                for (int e = 0; e < hd; ++e) o[e] = 0.f;
                for (int j = 0; j < Nk; ++j) {
                    const float *vj = v + ((size_t)b * Nk + j) * ldv + h * hd;
                    float pj = s[j] * inv;
                    for (int e = 0; e < hd; ++e) o[e] += pj * vj[e];
                }
            }
            free(s);
        }
}

/* MLPSepConv on token-major data: tokens viewed as [B,h,w,d] (channels-last view of
 * "bs (h w) d -> bs d h w", transformer_blocks.py:108-112). */
static void mlp_sepconv(const tld_o_model *m, const tld_o_layer *L, const float *xin, int B, float *out) {
    int d = m->embed_dim, hid = m->hidden, N = m->seq_len;
    int g = (int)sqrt((double)N);                       /* w = h = int(np.sqrt(x.size(1))) */
    float *u = falloc((size_t)B * N * hid), *c = falloc((size_t)B * N * hid);
    linear(xin, L->up_w, L->up_b, B * N, d, hid, u);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < N; ++t) {
            int i = t / g, j = t % g;
            float *o = c + ((size_t)b * N + t) * hid;
            for (int ch = 0; ch < hid; ++ch) o[ch] = L->dw_b[ch];
            for (int du = -1; du <= 1; ++du)
                for (int dv = -1; dv <= 1; ++dv) {
                    int ii = i + du, jj = j + dv;
                    if (ii < 0 || ii >= g || jj < 0 || jj >= g) continue;   /* zero padding */
                    const float *src = u + ((size_t)b * N + ii * g + jj) * hid;
                    int widx = (du + 1) * 3 + (dv + 1);                    /* cross-correlation */
                    for (int ch = 0; ch < hid; ++ch) o[ch] += L->dw_w[ch * 9 + widx] * src[ch];
                }
            for (int ch = 0; ch < hid; ++ch) o[ch] = gelu_erf(o[ch]);
        }
    linear(c, L->down_w, L->down_b, B * N, hid, d, out);
    free(u); free(c);
}

/* one DecoderBlock in place on x[B,N,d] with cond y[B,2,d]  (transformer_blocks.py:135-139).
 * after_sa / after_ca / after_mlp: optional snapshots of x after each residual add. */
static void decoder_block(const tld_o_model *m, const tld_o_layer *L, float *x, const float *y, int B,
                          float *after_sa, float *after_ca, float *after_mlp) {
    int d = m->embed_dim, N = m->seq_len, H = m->n_heads, hd = m->head_dim;
    size_t T = (size_t)B * N;
    float *xn = falloc(T * d), *qkv = falloc(T * 3 * d), *att = falloc(T * d), *qc = falloc(T * d),
          *kv = falloc((size_t)B * 2 * 2 * d);
    /* x = SA(LN1 x) + x */
    layernorm(x, L->n1_w, L->n1_b, (long)T, d, xn);
    linear(xn, L->qkv_w, NULL, (int)T, d, 3 * d, qkv);
    mha(qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, B, N, N, H, hd, att, d);
    for (size_t i = 0; i < T * d; ++i) x[i] += att[i];
    if (after_sa) memcpy(after_sa, x, T * d * sizeof(float));
    /* x = CA(LN2 x, y) + x */
    layernorm(x, L->n2_w, L->n2_b, (long)T, d, xn);
    linear(xn, L->q_w, NULL, (int)T, d, d, qc);
    linear(y, L->kv_w, NULL, B * 2, d, 2 * d, kv);
    mha(qc, d, kv, 2 * d, kv + d, 2 * d, B, N, 2, H, hd, att, d);
    for (size_t i = 0; i < T * d; ++i) x[i] += att[i];
    if (after_ca) memcpy(after_ca, x, T * d * sizeof(float));
    /* x = MLPSepConv(LN3 x) + x */
    layernorm(x, L->n3_w, L->n3_b, (long)T, d, xn);
    mlp_sepconv(m, L, xn, B, att);
    for (size_t i = 0; i < T * d; ++i) x[i] += att[i];
    if (after_mlp) memcpy(after_mlp, x, T * d * sizeof(float));
    free(xn); free(qkv); free(att); free(qc); free(kv);
}

/* out_proj + unpatchify: "b (h w) (c p1 p2) -> b c (h p1) (w p2)"  (denoiser.py:47-52,72,82) */
TLD_O_EXPORT void tld_o_unembed(const tld_o_model *m, const float *tok, int B, float *out) {
    int C = m->n_channels, S = m->image_size, p = m->patch_size, g = m->grid, pd = m->patch_dim,
        d = m->embed_dim, N = m->seq_len;
    float *pr = falloc((size_t)B * N * pd);
    linear(tok, m->out_w, m->out_b, B * N, d, pd, pr);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < N; ++t) {
            int i = t / g, j = t % g;
            for (int c = 0; c < C; ++c)
                for (int u = 0; u < p; ++u)
                    for (int v = 0; v < p; ++v)
                        out[(((size_t)b * C + c) * S + (i * p + u)) * S + (j * p + v)] =
                            pr[((size_t)b * N + t) * pd + (c * p + u) * p + v];
        }
    free(pr);
}

/* Denoiser.forward with optional stage dumps (any pointer may be NULL):
 *   sin_emb[B,ne] cond_y[B,2,d] tokens0[B,N,d] blk0_sa/ca/mlp[B,N,d] tokens_final[B,N,d] */
TLD_O_EXPORT void tld_o_forward_debug(const tld_o_model *m, const float *x, const float *sigma,
                                      const float *label, int B, float *out, float *sin_emb,
                                      float *cond_y, float *tokens0, float *blk0_sa, float *blk0_ca,
                                      float *blk0_mlp, float *tokens_final) {
    int d = m->embed_dim, N = m->seq_len;
    float *y = falloc((size_t)B * 2 * d), *tok = falloc((size_t)B * N * d);
    tld_o_cond(m, sigma, label, B, sin_emb, y);
    if (cond_y) memcpy(cond_y, y, (size_t)B * 2 * d * sizeof(float));
    tld_o_embed(m, x, B, tok);
    if (tokens0) memcpy(tokens0, tok, (size_t)B * N * d * sizeof(float));
    for (int l = 0; l < m->n_layers; ++l)
        decoder_block(m, &m->layers[l], tok, y, B, l == 0 ? blk0_sa : NULL, l == 0 ? blk0_ca : NULL,
                      l == 0 ? blk0_mlp : NULL);
    if (tokens_final) memcpy(tokens_final, tok, (size_t)B * N * d * sizeof(float));
    tld_o_unembed(m, tok, B, out);
    free(y); free(tok);
}

TLD_O_EXPORT void tld_o_forward(const tld_o_model *m, const float *x, const float *sigma,
                                const float *label, int B, float *out) {
    tld_o_forward_debug(m, x, sigma, label, B, out, 0, 0, 0, 0, 0, 0, 0);
}

/* ------------------------------------------------------------------------------------------ */
/* sampler: DiffusionGenerator.generate minus RNG and VAE  (diffusion.py:54-92)                */
/*   x_T[B,C,S,S], labels[B,text] (cond only; the zero uncond half is appended here, :61)      */
/*   noise_levels[n_levels] float64 -- already with noise_levels[0]=0.99 applied by the caller  */
/*   trace_x0 / trace_xt: optional [n_levels-1, B,C,S,S] per-loop-iteration dumps               */
/* ------------------------------------------------------------------------------------------ */
TLD_O_EXPORT void tld_o_sample(const tld_o_model *m, const float *x_T, const float *labels, int B,
                               const double *noise_levels, int n_levels, double class_guidance,
                               int use_ddpm_plus, double sharp_f, double bright_f, float *out_latent,
                               float *trace_x0, float *trace_xt) {
    int C = m->n_channels, S = m->image_size, te = m->text_emb_size;
    size_t img = (size_t)C * S * S, tot = (size_t)B * img;
    float *xt = falloc(tot), *x2 = falloc(2 * tot), *o2 = falloc(2 * tot), *x0 = falloc(tot),
          *x0_prev = falloc(tot), *lab2 = falloc((size_t)2 * B * te), *sig = falloc(2 * B);
    memcpy(xt, x_T, tot * sizeof(float));
    memcpy(lab2, labels, (size_t)B * te * sizeof(float));   /* second half stays zero */
    double *rs = NULL;
    if (use_ddpm_plus && n_levels > 2) {                     /* diffusion.py:54-57 */
        double *lam = (double *)malloc(sizeof(double) * n_levels);
        double *hs = (double *)malloc(sizeof(double) * (n_levels - 1));
        rs = (double *)malloc(sizeof(double) * (n_levels - 2));
        for (int i = 0; i < n_levels; ++i) lam[i] = log((1.0 - noise_levels[i]) / noise_levels[i]);
        for (int i = 1; i < n_levels; ++i) hs[i - 1] = lam[i] - lam[i - 1];
        for (int i = 1; i < n_levels - 1; ++i) rs[i - 1] = hs[i - 1] / hs[i];
        free(lam); free(hs);
    }
    int have_prev = 0;
    double next_noise = noise_levels[0];
    for (int step = 0; step <= n_levels - 1; ++step) {
        int final = (step == n_levels - 1);
        double curr = final ? next_noise : noise_levels[step];
        /* pred_image: model(cat[x_t,x_t], full(sigma), [labels;0]) then CFG (diffusion.py:94-103) */
        memcpy(x2, xt, tot * sizeof(float)); memcpy(x2 + tot, xt, tot * sizeof(float));
        for (int b = 0; b < 2 * B; ++b) sig[b] = (float)curr;
        tld_o_forward(m, x2, sig, lab2, 2 * B, o2);
        float g = (float)class_guidance, g1 = (float)(1.0 - class_guidance);
        for (size_t i = 0; i < tot; ++i) x0[i] = g * o2[i] + g1 * o2[tot + i];   /* :124-125 */
        if (final) break;
        next_noise = noise_levels[step + 1];
        if (trace_x0) memcpy(trace_x0 + (size_t)step * tot, x0, tot * sizeof(float));
        /* update, evaluated left-to-right in float32 tensors with python-float scalars
           (scalars are rounded to float32 when they meet a float32 tensor) */
        float a = (float)(curr - next_noise), bnx = (float)next_noise, cc = (float)curr;
        if (!have_prev) {                                                      /* :71-72 */
            for (size_t i = 0; i < tot; ++i) xt[i] = (a * x0[i] + bnx * xt[i]) / cc;
        } else {
            if (use_ddpm_plus) {                                               /* :74-76 */
                float c1 = (float)(1.0 + 1.0 / (2.0 * rs[step - 1]));
                float c2 = (float)(1.0 / (2.0 * rs[step - 1]));
                for (size_t i = 0; i < tot; ++i) {
                    float D = c1 * x0[i] - c2 * x0_prev[i];
                    xt[i] = (a * D + bnx * xt[i]) / cc;                        /* :81 */
                }
            } else {
                for (size_t i = 0; i < tot; ++i) xt[i] = (a * x0[i] + bnx * xt[i]) / cc;
            }
        }
        memcpy(x0_prev, x0, tot * sizeof(float)); have_prev = 1;               /* :83 */
        if (trace_xt) memcpy(trace_xt + (size_t)step * tot, xt, tot * sizeof(float));
    }
    /* latent shifts on channels 3 and 0 (diffusion.py:88-89) */
    for (int b = 0; b < B; ++b) {
        if (C > 3) for (int i = 0; i < S * S; ++i) x0[((size_t)b * C + 3) * S * S + i] += (float)sharp_f;
        for (int i = 0; i < S * S; ++i) x0[((size_t)b * C + 0) * S * S + i] += (float)bright_f;
    }
    memcpy(out_latent, x0, tot * sizeof(float));
    free(xt); free(x2); free(o2); free(x0); free(x0_prev); free(lab2); free(sig); free(rs);
}

/* bench.py's cpu_baseline leg only: cap the OpenMP team (all hardware threads is the default; SMT siblings hurt here) */
TLD_O_EXPORT void tld_o_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

TLD_O_EXPORT int tld_o_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
