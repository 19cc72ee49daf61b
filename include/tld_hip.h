/*
 * tld_hip.h -- C ABI of libtld_hip.so: the MI355X (gfx950) denoising engine.
 *
 * The reference has no FFI/plugin interface; its boundary is the Python nn.Module call contract
 * between the sampler and the model.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference checkout).  Signatures use plain
 * pointers and sizes only (no torch types): device buffers belong to the caller (PyTorch), packed
 * weights and workspace belong to the engine.  All kernels are enqueued on the caller's HIP stream
 * with no hidden synchronisation.  Every function returns 0 on success or a non-zero status;
 * tld_last_error() returns the thread-local message.  Nothing throws across this boundary.
 */
#ifndef TLD_HIP_H
#define TLD_HIP_H

#include <stdint.h>

#if defined(__GNUC__)
#define TLD_API __attribute__((visibility("default")))
#else
#define TLD_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tld_engine tld_engine;

/* Mirrors tld/configs.py:21-31 DenoiserConfig (dropout is identity at inference and not carried),
 * plus engine sizing.  Replaces the kwargs of Denoiser.__init__ (tld/denoiser.py:86-97). */
typedef struct tld_config {
    int32_t image_size;       /* image_size / patch_size (the token grid's side) must be a multiple of 4; the reference takes any square
                                 grid (tld/transformer_blocks.py:109) -- tld_engine_create says so when it refuses one */
    int32_t noise_embed_dims; /* even */
    int32_t patch_size;       /* n_channels * patch_size^2 (the patch vector) <= 64 */
    int32_t embed_dim;        /* any multiple of the head width 64 up to 1024: heads = embed_dim / 64 (tld/transformer_blocks.py:126-128);
                                 the training engine (tld_train_*) takes the same widths */
    int32_t n_layers;
    int32_t text_emb_size;
    int32_t n_channels;
    int32_t mlp_multiplier;
    int32_t max_batch;        /* largest model batch (CFG-doubled) a forward will see */
    int32_t device_id;        /* HIP device ordinal */
} tld_config;

enum { TLD_DTYPE_F32 = 0, TLD_DTYPE_BF16 = 1, TLD_DTYPE_F16 = 2 };

enum {
    TLD_OK = 0,
    TLD_ERR_INVALID = 1,      /* bad argument / unsupported configuration */
    TLD_ERR_KEY = 2,          /* unknown state_dict key */
    TLD_ERR_SHAPE = 3,        /* tensor shape does not match the configuration */
    TLD_ERR_STATE = 4,        /* call order (weights not finalized, missing tensors) */
    TLD_ERR_HIP = 5           /* a HIP runtime call failed */
};

/* Denoiser(**asdict(cfg)) -- tld/denoiser.py:85-114, tld/diffusion.py:145
 * (No entry point changes the calling thread's current HIP device: each one switches to cfg.device_id
 * for its own duration and restores the previous device before returning.) */
TLD_API int tld_engine_create(const tld_config* cfg, tld_engine** out);

/* Denoiser.load_state_dict, one entry at a time -- tld/diffusion.py:152-153.
 * key is the reference state_dict key; host_ptr is contiguous host memory of `dtype`
 * (TLD_DTYPE_F32; int64 buffers such as precomputed_pos_enc are passed with ndim/shape and ignored). */
TLD_API int tld_engine_load_tensor(tld_engine* e, const char* key, const void* host_ptr, const int64_t* shape,
                           int32_t ndim, int32_t dtype);

/* Packs weights into device layouts (bf16 GEMM operands, folded tables).  Must follow the loads;
 * fails with TLD_ERR_STATE and names the first missing key if the state_dict was incomplete. */
TLD_API int tld_engine_finalize_weights(tld_engine* e);

/* Operand type of the QKV / MLP GEMMs: 0 = bf16 (default), 1 = MX-fp8 (OCP e4m3 elements with one E8M0 scale per 32
 * K-elements, v_mfma_scale_f32_32x32x64_f8f6f4).  Call between tld_engine_create and tld_engine_finalize_weights.
 * Not in the reference (its model_dtype is fp32 / fp16 / bf16, tld/configs.py:33-37): BASELINE config C4. */
TLD_API int tld_engine_set_gemm_dtype(tld_engine* e, int32_t dtype);

/* Low-latency capacity classes (round 5 / 6; no counterpart in the reference, whose serving path -- tld/app.py:48-65 -- runs one prompt per call on whatever
 * kernels PyTorch picks): the MLP down projection of every block runs as K-splits + a finishing kernel.
 *   on = 1: four K-splits, engines of at most 4096 token rows (max_batch x tokens; e.g. 8 images = 16 CFG-doubled samples at 256 px)
 *   on = 2: eight K-splits, engines of at most 1024 token rows (one or two images per CFG call at 256 px -- the one-prompt-per-call pattern)
 *   on = 0: the default class
 * A one-image 35-step generate takes 37 ms in the default class, 31 ms in class 1, 30 ms in class 2; eight images 46 ms in either.  Results of a class differ from
 * the default class (and from the other class) in the fp32 summation order of that product (same tolerances against the reference); a class is chosen by the CALLER for the engine,
 * never by the batch of a call: inside a class results are bit-identical across batch sizes.
 * May be called any time after tld_engine_create; fails (TLD_ERR_INVALID) on larger engines, on widths other than 384 / 768, on hidden widths that do not split
 * into that many multiples of 64, and on engines in the MX-fp8 GEMM mode (bf16 operands only; tld_engine_set_gemm_dtype(fp8) likewise refuses an engine of these classes). */
TLD_API int tld_engine_set_low_latency(tld_engine* e, int32_t on);

/* Denoiser.forward(x, noise_level, label) -- tld/denoiser.py:116-126 (called at tld/diffusion.py:97-101).
 *   x      [batch, C, S, S]     device, io_dtype
 *   noise  [batch, 1]           device, io_dtype
 *   label  [batch, text_emb]    device, io_dtype
 *   out    [batch, C, S, S]     device, io_dtype (may not alias x)
 * Inputs are not modified. */
TLD_API int tld_denoiser_forward(tld_engine* e, const void* x, const void* noise, const void* label, void* out,
                         int32_t batch, int32_t io_dtype, void* hip_stream);

/* DiffusionGenerator.generate minus RNG and VAE decode -- tld/diffusion.py:54-92 with pred_image
 * (:94-103) and apply_classifier_free_guidance (:122-125) fused on device.
 *   x_T     [batch, C, S, S] fp32 device: initial noise (initialize_image, :105-120, done by caller)
 *   labels  [batch, text_emb] fp32 device: conditional embeddings only; the zero "uncond" half of
 *           :61 is implicit
 *   coeffs  [n_levels, 6] fp32 HOST: (sigma, a, b, c, c1, c2) per forward, see
 *           transformer_latent_diffusion_amd/schedule.py (host float64 algebra of :50-57,:72-81)
 *   out_latent [batch, C, S, S] fp32 device: x0_pred incl. sharp_f/bright_f shifts (:88-89)
 *   trace_x0 / trace_xt: optional device buffers [n_levels-1, batch, C, S, S] fp32 (NULL to skip)
 * batch*2 must be <= max_batch.  The call does NOT synchronise the stream: its host-built tables (sigma per level, token-row
 * indices) travel through a pinned staging buffer owned by the engine, and only a later tld_sample on the same engine waits (on
 * an event, long complete by then) before refilling it; all steps are enqueued asynchronously on hip_stream. */
TLD_API int tld_sample(tld_engine* e, const void* x_T, const void* labels, const float* coeffs, int32_t n_levels,
               float class_guidance, float sharp_f, float bright_f, void* out_latent, int32_t batch,
               void* trace_x0, void* trace_xt, void* hip_stream);

/* Test hook: copy a named internal stage of the LAST forward to host fp32 (synchronises).
 * names: "cond_y" [T,d] (T = batch noise rows then batch label rows), "tokens0", "blk0_sa",
 * "blk0_ca", "blk0_mlp", "tokens_final" (each [batch*N, d]).  Stage capture must have been
 * enabled with tld_engine_set_debug(e, 1) before the forward. */
TLD_API int tld_engine_set_debug(tld_engine* e, int32_t enable);
TLD_API int tld_engine_read_stage(tld_engine* e, const char* name, float* host_out, int64_t numel);

/* Test hook: C[M,N] = A[M,K] . W[N,K]^T with the engine's bf16 MFMA GEMM (fp32 accumulate), bf16
 * device inputs, fp32 device output.  K % 64 == 0. */
TLD_API int tld_debug_gemm_bf16(const void* a_bf16, const void* w_bf16, float* c_f32, int32_t M, int32_t N,
                        int32_t K, void* hip_stream);

/* Test hook: the same product as `ksplit` K-slices, c_slices[s] = A[:, s K/ksplit : (s+1) K/ksplit] . W[:, same]^T (fp32 [ksplit][M][N]).  K % (64 ksplit) == 0. */
TLD_API int tld_debug_gemm_splitk(const void* a_bf16, const void* w_bf16, float* c_slices_f32, int32_t M, int32_t N, int32_t K, int32_t ksplit,
                          void* hip_stream);

/* Test hooks of the MX-fp8 path.  quant_mx8: bf16 device matrix [M,K] -> e4m3 bytes [M,K] + E8M0 block scales laid out
 * [K/128][M][4] (device); quant_mx8_host: the weight-side quantiser (fp32 host matrix, host outputs, no GPU needed);
 * gemm_mx8: C[M,N] = dequant(A) . dequant(W)^T in fp32 from such operands (device).  K % 128 == 0, M % 4 == N % 4 == 0. */
TLD_API int tld_debug_quant_mx8(const void* in_bf16, void* out_e4m3, void* out_scale, int32_t M, int32_t K, void* hip_stream);
TLD_API int tld_debug_quant_mx8_host(const float* w, int32_t rows, int32_t K, void* out_e4m3, void* out_scale);
TLD_API int tld_debug_gemm_mx8(const void* a_e4m3, const void* a_scale, const void* w_e4m3, const void* w_scale, float* c_f32,
                               int32_t M, int32_t N, int32_t K, void* hip_stream);

/* Test/bench hook: time the engine's GEMM on self-allocated, pseudo-randomly filled device buffers.
 * epilogue: 0 fp32 out, 1 QKV (q|k row-major + V^T; N = 3*d, ntok tokens per sample), 2 bias+bf16,
 * 3 bias + fp32 residual add.  Returns the average kernel time over `iters` launches (HIP events). */
TLD_API int tld_debug_gemm_bench(int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t ntok, int32_t iters,
                                 double* avg_ms);

/* Live per-kernel-class timing with HIP events recorded on the launch stream around every launch
 * of the selected classes (bit k of class_mask).  Classes: 0 gemm_qkv, 1 gemm_up, 2 gemm_down,
 * 3 attention, 4 cross_row, 5 dwconv_gelu, 6 layernorm, 7 embed, 8 tail, 9 update, 10 conditioning.
 * set_profile forgets previously recorded timings; get_profile synchronises the device and returns
 * the summed elapsed time and the number of launches of one class since set_profile.
 * profile_reserve pre-creates `launches` event pairs for one class so that a timed region records into
 * existing events only (no hipEventCreate between the caller's fences). */
TLD_API int tld_engine_set_profile(tld_engine* e, uint32_t class_mask);
TLD_API int tld_engine_profile_reserve(tld_engine* e, int32_t kclass, int64_t launches);
TLD_API int tld_engine_get_profile(tld_engine* e, int32_t kclass, double* total_ms, int64_t* launches);

/* bytes of packed weights resident on the device */
TLD_API int64_t tld_engine_weight_bytes(const tld_engine* e);

TLD_API int tld_engine_destroy(tld_engine* e);

/* ---- VAE decode of the final latents --------------------------------------------------
 * Replaces `self.vae.decode(latents)[0]` at tld/diffusion.py:91, where `vae` is diffusers' AutoencoderKL
 * ("madebyollin/sdxl-vae-fp16-fix", tld/configs.py:39-43; a third-party dependency that is not part of the reference
 * checkout -- the algorithm restated here is AutoencoderKL.decode of diffusers 0.2x: post_quant_conv -> Decoder
 * (conv_in, UNetMidBlock2D with one single-head attention, UpDecoderBlock2D x n, GroupNorm + SiLU, conv_out)).
 * Activations are bf16 channels-last on the device, GroupNorm statistics / softmax / accumulation fp32. */
typedef struct tld_vae tld_vae;

typedef struct tld_vae_config {
    int32_t latent_channels;        /* 4 */
    int32_t out_channels;           /* 3 */
    int32_t n_blocks;               /* entries of block_out_channels in use (<= 4) */
    int32_t block_out_channels[4];  /* AutoencoderKL order, e.g. 128, 256, 512, 512; each in {64,128,256,512,1024} */
    int32_t layers_per_block;       /* 2: every decoder up block has layers_per_block + 1 resnets */
    int32_t norm_num_groups;        /* 32 */
    int32_t mid_block_attention;    /* 1 */
    int32_t use_post_quant_conv;    /* 1 */
    int32_t latent_size;            /* h = w of the latent image (32 for 256 px output) */
    int32_t max_batch;              /* largest batch one tld_vae_decode call will see */
    int32_t device_id;
} tld_vae_config;

TLD_API int tld_vae_create(const tld_vae_config* cfg, tld_vae** out);

/* AutoencoderKL.load_state_dict, one entry at a time (diffusers key names: "decoder.conv_in.weight",
 * "decoder.mid_block.attentions.0.to_q.weight" or its pre-0.19 spelling "...query.weight", "post_quant_conv.bias" ...).
 * "encoder.*" and "quant_conv.*" entries are accepted and ignored.  Host fp32 data. */
TLD_API int tld_vae_load_tensor(tld_vae* v, const char* key, const void* host_ptr, const int64_t* shape, int32_t ndim,
                                int32_t dtype);
TLD_API int tld_vae_finalize_weights(tld_vae* v);

/* AutoencoderKL.decode(z)[0] -- tld/diffusion.py:91.
 *   z    [batch, latent_channels, h, w]   device, io_dtype (already multiplied by the caller's scale factor)
 *   out  [batch, out_channels, 8h, 8w]    device, fp32 (2^(n_blocks-1) x upsampling) */
TLD_API int tld_vae_decode(tld_vae* v, const void* z, float* out, int32_t batch, int32_t io_dtype, void* hip_stream);

/* Test hook: with debug enabled, decode keeps a copy of the activation after every stage; read_stage converts one to
 * host fp32 [batch, C, H, W].  names: "conv_in", "mid.res0", "mid.attn", "mid.res1", "up<i>.res<j>", "up<i>.upsample",
 * "norm_out".  shape4 (optional) receives batch, C, H, W. */
TLD_API int tld_vae_set_debug(tld_vae* v, int32_t enable);
TLD_API int tld_vae_read_stage(tld_vae* v, const char* name, float* host_out, int64_t numel, int64_t* shape4);

/* Live timing of one kernel class of the decoder (HIP events around every launch, like tld_engine_set_profile).
 * classes: 0 conv3x3, 1 gemm (1x1 / attention), 2 groupnorm, 3 other */
TLD_API int tld_vae_set_profile(tld_vae* v, int32_t enable);
TLD_API int tld_vae_get_profile(tld_vae* v, int32_t kclass, double* total_ms, int64_t* launches);

/* Test hook: the implicit-GEMM 3x3 convolution alone (zero padding 1, stride 1; up = 1: nearest 2x upsampling folded in).
 *   in  bf16 channels-last [B, H >> up, W >> up, cin] (device);  w  bf16 [cout][3][3][cin] (device)
 *   out fp32 [B*H*W][cout] (device).  cin % 64 == 0.  Synchronises the stream. */
TLD_API int tld_debug_conv3x3(const void* in_bf16, const void* w_bf16, float* out_f32, int32_t B, int32_t H, int32_t W,
                              int32_t cin, int32_t cout, int32_t up, void* hip_stream);

TLD_API int64_t tld_vae_weight_bytes(const tld_vae* v);
TLD_API int tld_vae_destroy(tld_vae* v);

/* ---- CLIP text tower: the front edge ---------------------------------------------------
 * Replaces `model.encode_text(text_tokens)` in encode_text, tld/diffusion.py:136-140, where `model` is OpenAI CLIP
 * "ViT-L/14" from `clip.load` (tld/diffusion.py:160, tld/configs.py:46-48; third-party, not in the reference checkout).
 * Restated: CLIP.encode_text of openai/CLIP clip/model.py (token + positional embedding, pre-LN residual blocks with
 * causal nn.MultiheadAttention and a QuickGELU MLP, ln_final, the EOT token's row times text_projection).  Tokenisation
 * stays on the host (clip.tokenize); the engine takes token ids. */
typedef struct tld_clip tld_clip;

typedef struct tld_clip_config {
    int32_t vocab_size;         /* 49408 */
    int32_t context_length;     /* 77 (<= 128) */
    int32_t width;              /* 768: transformer width, multiple of 64 */
    int32_t heads;              /* width / 64 */
    int32_t layers;             /* 12 */
    int32_t embed_dim;          /* 768: columns of text_projection */
    int32_t max_batch;
    int32_t device_id;
} tld_clip_config;

TLD_API int tld_clip_create(const tld_clip_config* cfg, tld_clip** out);
/* CLIP.state_dict() entries, one at a time (host fp32): "token_embedding.weight", "positional_embedding", "text_projection",
 * "ln_final.*", "transformer.resblocks.<i>.{ln_1,ln_2}.*", ".attn.in_proj_{weight,bias}", ".attn.out_proj.*", ".mlp.c_fc.*",
 * ".mlp.c_proj.*".  "visual.*", "logit_scale" and the archive's metadata entries are accepted and ignored. */
TLD_API int tld_clip_load_tensor(tld_clip* c, const char* key, const void* host_ptr, const int64_t* shape, int32_t ndim, int32_t dtype);
TLD_API int tld_clip_finalize_weights(tld_clip* c);
/* CLIP.encode_text(text):  tokens [batch, context_length] int32 (device), eot_index [batch] int32 (device) = text.argmax(-1)
 * (the EOT token has the largest id), out [batch, embed_dim] fp32 (device). */
TLD_API int tld_clip_encode_text(tld_clip* c, const int32_t* tokens, const int32_t* eot_index, float* out, int32_t batch, void* hip_stream);
/* Test hook: copy a workspace buffer of the LAST encode_text to host fp32 (synchronises): "x", "tmp" ([T, width] fp32), "pooled"
 * ([batch, width]), "h", "att" ([T, width]), "qkv" ([T, 3 width]), "f" ([T, 4 width]) -- the state after the last block. */
TLD_API int tld_clip_read_buffer(tld_clip* c, const char* name, float* host_out, int64_t numel);
TLD_API int64_t tld_clip_weight_bytes(const tld_clip* c);
TLD_API int tld_clip_destroy(tld_clip* c);

TLD_API const char* tld_last_error(void);

/* ---- Training step: tld/train.py:162-173 for one batch on one device -------------------------------------------
 * The model's parameters, their gradients, Adam's two moment vectors and the EMA copy are FLAT fp32 device vectors owned by the
 * caller, in the order of Denoiser.named_parameters() (tld/denoiser.py:85-114); tld_train_param_layout enumerates (key, offset,
 * numel).  The gradient all-reduce of the reference's accelerate/DDP wrapper (tld/train.py:114,168) is the caller's
 * torch.distributed all_reduce on the flat gradient vector between tld_train_forward_backward and tld_train_adam_ema.
 * cfg.max_batch = largest batch (NOT doubled); 256-token latents (image_size / patch_size == 16) only. */
typedef struct tld_train tld_train;
TLD_API int tld_train_create(const tld_config* cfg, tld_train** out);
TLD_API int64_t tld_train_param_count(const tld_train* e);
TLD_API int32_t tld_train_tensor_count(const tld_train* e);
TLD_API int tld_train_param_layout(const tld_train* e, int32_t index, char* key_out, int32_t key_cap, int64_t* offset, int64_t* numel);
/* registered buffer "fourier_feats.0.angular_speeds" (tld/transformer_blocks.py:11-15): host fp32 [noise_embed_dims / 2] */
TLD_API int tld_train_set_angular_speeds(tld_train* e, const float* host, int32_t n);
/* params / grads: device fp32 [tld_train_param_count] */
TLD_API int tld_train_bind(tld_train* e, float* params, float* grads);
/* bf16 GEMM operands (and their transposes) of the current parameters; called automatically after an optimizer step */
TLD_API int tld_train_refresh_weights(tld_train* e, void* hip_stream);
/* model.train(); pred = model(x_noisy, noise_level.view(-1, 1), label); loss = nn.MSELoss()(pred, target); loss.backward()
 * -- tld/train.py:160,166-168.  x_noisy / target [batch, C, S, S], noise_level [batch], label [batch, text_emb]: device fp32.
 * Writes every gradient into the bound grads vector (overwritten, like zero_grad + backward), the scalar loss to loss_out (device
 * fp32[1]) and the prediction to pred_out (device fp32 [batch, C, S, S]). */
TLD_API int tld_train_forward_backward(tld_train* e, const float* x_noisy, const float* noise_level, const float* label, const float* target,
                                       int32_t batch, float* loss_out, float* pred_out, void* hip_stream);
/* The same step with a host callback for overlapping the data-parallel gradient reduction with the backward pass -- what the bucketed
 * all-reduce hooks of the reference's DDP wrapper do (accelerate.prepare -> torch DDP, tld/train.py:109,168).  grad_ready(user, offset,
 * numel) is invoked on the calling thread each time the kernels that FINISH a contiguous range [offset, offset + numel) of the flat
 * gradient vector have been enqueued on hip_stream: once per decoder block, last block first (a block's 15 tensors are contiguous in
 * named_parameters() order), then the ranges before and after the blocks (embedding / position table; out_proj, norm, label_proj --
 * final only at the end of the backward).  The ranges tile the vector exactly.  The callback typically records an event on hip_stream
 * and launches an asynchronous all-reduce of that slice on a second stream; it must not synchronise hip_stream.  NULL: no callbacks. */
typedef void (*tld_grad_ready_fn)(void* user, int64_t offset, int64_t numel);
TLD_API int tld_train_forward_backward_cb(tld_train* e, const float* x_noisy, const float* noise_level, const float* label, const float* target,
                                          int32_t batch, float* loss_out, float* pred_out, void* hip_stream, tld_grad_ready_fn grad_ready, void* user);
/* optimizer.step() of torch.optim.Adam(lr) + update_ema(ema_model, model, alpha) -- tld/train.py:87,169,55-58,172.  step counts from 1;
 * ema may be NULL; grad_scale multiplies the gradient first (1 / world_size after a SUM all-reduce). */
TLD_API int tld_train_adam_ema(tld_train* e, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema, int64_t numel,
                               float lr, float beta1, float beta2, float eps, int32_t step, float ema_alpha, float grad_scale, void* hip_stream);
/* Test hook: self-attention backward alone (head_dim 64; ntok = 64, 128 or a multiple of 256): qk [M, 2d] bf16 (q | k), vt [B, H, 64, ntok]
 * bf16, o [M, d] bf16 (forward output), g [M, d] fp32 (dL/dO) -> dqkv [M, 3d] bf16 (dq | dk | dv).  scratch: 2 * batch * heads * ntok floats
 * (row statistics between the two kernels of the ntok > 256 path; may be NULL otherwise).  Device pointers. */
TLD_API int tld_debug_attention_bwd(const void* qk, const void* vt, const void* o, const float* g, void* dqkv, float* scratch, int32_t batch,
                                    int32_t ntok, int32_t heads, void* hip_stream);
/* Test hook: a weight gradient of the training step, dW[n_out, k_in] = dY^T X (dY [rows, n_out], X [rows, k_in] bf16 row-major; fp32 out):
 * both operands are read as they are (the contraction index is the row), split-K partial sums go through `slices` (slice_floats fp32).
 * n_out, k_in multiples of 256, rows a multiple of 64.  Device pointers. */
TLD_API int tld_debug_wgrad(const void* dy, const void* x, float* dw, float* slices, int64_t slice_floats, int32_t rows, int32_t n_out, int32_t k_in,
                            void* hip_stream);
/* Test / measurement hook: self-attention forward alone, softmax(q k^T / 8) v per head (head_dim 64; MHAttention.forward,
 * tld/transformer_blocks.py:31-48).  qk [batch * ntok, 2 d] bf16 (q | k), vt [batch, heads * 64, ntok] bf16 (V transposed per head),
 * att [batch * ntok, d] bf16 out, d = 64 heads.  iters launches back to back; *ms_per_launch (host pointer, may be NULL) receives the
 * HIP-event average.  Device pointers. */
TLD_API int tld_debug_attention_fwd(const void* qk, const void* vt, void* att, int32_t batch, int32_t ntok, int32_t heads, int32_t iters,
                                    float* ms_per_launch, void* hip_stream);
/* Test hook: the MLP's depthwise 3x3 convolution + GELU alone (nn.Conv2d(hid, hid, 3, padding=1, groups=hid) then nn.GELU() on the
 * "b (h w) c -> b c h w" view, tld/transformer_blocks.py:95-103) on channels-last tokens: in / out [batch, grid * grid, channels] bf16
 * (device), weight [channels, 9] and bias [channels] fp32 (HOST, the reference's conv.weight / conv.bias).  Runs the kernel the engine
 * would pick for that grid (whole-image / tiled / row-streaming).  channels % 64 == 0; grid <= 16 or a multiple of 16. */
TLD_API int tld_debug_dwconv_gelu(const void* in_bf16, const float* weight_host, const float* bias_host, void* out_bf16, int32_t batch,
                                  int32_t grid, int32_t channels, void* hip_stream);
TLD_API int tld_train_destroy(tld_train* e);

#ifdef __cplusplus
}
#endif
#endif /* TLD_HIP_H */
