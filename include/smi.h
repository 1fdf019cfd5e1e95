/* smi.h -- C ABI of libsmi_hip.so: the MI355X-native (gfx950) hot path of the concept-slider LoRA trainer.
 *
 * The reference (ntc-ai/sliders-conceptmod) is pure Python and has no FFI of its own; the seam this library sits
 * under is the duck-typed model call the reference's step helpers make,
 *     unet(sample, timestep, encoder_hidden_states=..., added_cond_kwargs=...).sample
 * (conceptmod/textsliders/train_util.py:290-294 and :471-476), together with the autograd backward that
 * `loss.backward()` triggers through it (train_lora.py:298, train_lora_xl.py:348) and the elementwise step ops
 * around it.  Each entry point below names the reference code it replaces.  The Python binding a maintainer adds on
 * the reference side is shown in INTEGRATION.md (ctypes; the in-tree one is sliders_conceptmod_amd/_native.py).
 *
 * Conventions: plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host"; the caller
 * owns all memory (weights, LoRA parameters, gradients, latents, the workspace) and the library borrows it for the
 * duration of a call (the workspace: for the lifetime of the engine).  All work is enqueued on the `stream` given at
 * creation (a hipStream_t passed as void*, NULL = default stream); no call synchronises with the host.  Every
 * function returns 0 on success and a negative value on error; smi_last_error() then returns a message.
 * Nothing throws across the boundary.
 */
#ifndef SMI_H_
#define SMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMI_MAX_LEVELS 8
#define SMI_DTYPE_F16 0
#define SMI_DTYPE_BF16 1

/* Architecture of the diffusers UNet2DConditionModel being replaced (the values of its public config). */
typedef struct smi_unet_config {
  int dtype;                 /* storage / MFMA input type of weights and activations; accumulation is always fp32 */
  int in_channels;           /* 4 */
  int out_channels;          /* 4 */
  int n_levels;              /* len(block_out_channels) */
  int block_out_channels[SMI_MAX_LEVELS];
  int down_has_attn[SMI_MAX_LEVELS]; /* CrossAttnDownBlock2D (1) or DownBlock2D (0) */
  int up_has_attn[SMI_MAX_LEVELS];   /* CrossAttnUpBlock2D (1) or UpBlock2D (0), in up_blocks order */
  int layers_per_block;
  int transformer_layers[SMI_MAX_LEVELS]; /* per down level */
  int num_heads[SMI_MAX_LEVELS];          /* per down level (diffusers "attention_head_dim") */
  int mid_transformer_layers;
  int cross_attention_dim;
  int norm_num_groups;
  int use_linear_projection;
  int addition_embed;        /* 1 = SD-XL "text_time" conditioning */
  int addition_time_embed_dim;
  int projection_class_embeddings_input_dim;
} smi_unet_config;

/* One frozen weight tensor, named as in the diffusers state_dict ("down_blocks.0.resnets.0.conv1.weight"),
 * already in `dtype`, contiguous, torch layout. */
typedef struct smi_weight {
  const char* name;
  const void* data;
  int ndim;
  int64_t shape[4];
} smi_weight;

/* One LoRA-adapted Linear (reference: LoRAModule, conceptmod/textsliders/lora.py:76-138).
 * `target` is the dotted module path ("down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q").
 * lora_down.weight [rank, in] lives at down_flat + off_down, lora_up.weight [out, rank] at up_flat + off_up
 * (element offsets into the two flat fp32 buffers passed per call). scale = alpha / rank (lora.py:118-119). */
typedef struct smi_lora_site {
  const char* target;
  int64_t off_down;
  int64_t off_up;
  int rank;
  float scale;
  /* DoRA (conceptmod/textsliders/dora.py:53-162): element offset of this site's dora_scale [1, in] inside the UP flat
   * buffer (gradients accumulate at the same offset of the up-gradient buffer); < 0: a plain LoRA site.
   * dW = (W + up down) * (dora_scale / ||W + up down||_col) - W, y += scale * multiplier * x dW^T. */
  int64_t off_dora;
} smi_lora_site;

/* Architecture of the diffusers AutoencoderKL ENCODER half (image sliders encode their image pairs with it every step:
 * trainscripts/imagesliders/train_util.py:213-222). */
typedef struct smi_vae_config {
  int dtype;
  int in_channels;      /* 3 */
  int latent_channels;  /* 4: the encoder emits 2 x latent_channels moments (mean | logvar) */
  int n_levels;         /* len(block_out_channels) */
  int block_out_channels[SMI_MAX_LEVELS]; /* (128, 256, 512, 512) */
  int layers_per_block; /* 2 */
  int norm_num_groups;  /* 32 */
} smi_vae_config;

/* Architecture of a transformers CLIPTextModel / CLIPTextModelWithProjection (the prompt front end:
 * conceptmod/textsliders/train_util.py:108-155 `text_encode`, `text_encode_xl`). */
typedef struct smi_clip_config {
  int dtype;
  int vocab_size;        /* 49408 */
  int hidden_size;       /* 768 (CLIP ViT-L/14 text) / 1280 (OpenCLIP bigG) */
  int num_layers;        /* 12 / 32 */
  int num_heads;         /* 12 / 20 */
  int intermediate_size; /* 3072 / 5120 */
  int max_positions;     /* 77 */
  int hidden_act;        /* 0 quick_gelu, 1 gelu */
  int projection_dim;    /* 0: none (CLIPTextModel); > 0: text_projection (CLIPTextModelWithProjection) */
} smi_clip_config;

typedef struct smi_engine smi_engine;

const char* smi_last_error(void);

/* Bytes of device workspace an engine needs for UNet batch `batch` (= 2B of the reference's CFG-doubled batch,
 * train_util.py:285, or 4 x 2B when the four guidance passes run as one batched pass), of which at most
 * `batch_adapted` samples are LoRA-adapted and differentiated, latent h x w, context length ctx_len, and the given
 * set of adapted layers. */
int smi_workspace_bytes(const smi_unet_config* cfg, const smi_lora_site* sites, int n_sites, int batch,
                        int batch_adapted, int h, int w, int ctx_len, size_t* bytes);

/* Builds the engine: packs the frozen weights into MFMA-friendly layouts inside `workspace` (transposed copies for
 * the activation-gradient GEMMs, (ky,kx,ci)-ordered conv filters, fused q|k|v) and lays out the activation arenas.
 * Replaces: unet.to(device, dtype); unet.requires_grad_(False); unet.eval(); LoRANetwork(...).apply_to()
 * (train_lora.py:67-78). */
int smi_create(const smi_unet_config* cfg, const smi_weight* weights, int n_weights, const smi_lora_site* sites,
               int n_sites, int batch, int batch_adapted, int h, int w, int ctx_len, void* workspace,
               size_t workspace_bytes, void* stream, smi_engine** out);
void smi_destroy(smi_engine* e);

/* Engine lifetime across shapes.  The packed weights do not depend on the batch or the latent size; only the two
 * activation arenas do.  smi_workspace_bytes == smi_weights_bytes + smi_arena_bytes (+ alignment slack).
 * smi_replan switches a live engine to another (batch, batch_adapted, h, w, ctx_len) WITHOUT re-packing any weight:
 * `arena` is a caller-owned buffer of >= smi_arena_bytes(...) bytes that must outlive its use (NULL: reuse the arena
 * region of the creation workspace, if large enough).  It invalidates a saved forward (smi_unet_backward then fails
 * with -4).  This is what the reference's `dynamic_resolution` (a random bucket every step, train_util.py:1085-1097;
 * train_lora.py:172-178) needs: keep one arena per bucket and replan between them. */
int smi_weights_bytes(const smi_unet_config* cfg, const smi_lora_site* sites, int n_sites, size_t* bytes);
int smi_arena_bytes(const smi_unet_config* cfg, const smi_lora_site* sites, int n_sites, int batch, int batch_adapted,
                    int h, int w, int ctx_len, size_t* bytes);
int smi_replan(smi_engine* e, int batch, int batch_adapted, int h, int w, int ctx_len, void* arena,
               size_t arena_bytes);
/* out[0] = weight-packing kernel launches since creation, out[1] = replans, out[2] = generation number of the saved
 * forward whose tape is live (0: none; each save_for_backward forward gets a new number -- callers keep it next to the
 * output they will differentiate and compare before smi_unet_backward), out[3] = bytes of the packed-weight region. */
int smi_engine_stats(const smi_engine* e, int64_t out[4]);

/* ---- AutoencoderKL encoder (image sliders) ------------------------------------------------------------------------
 * moments = quant_conv(encoder(image)): replaces `vae.encode(image)` up to the posterior's parameters
 * (trainscripts/imagesliders/train_util.py:218: `vae.encode(image).latent_dist`); the caller draws the sample
 * (mean + exp(0.5 clamp(logvar, -30, 20)) eps) and multiplies by scaling_factor (:219-220).
 *   weights: diffusers AutoencoderKL state_dict entries `encoder.*` and `quant_conv.*`, dtype T
 *   image   f32 [n, 3, h, w], already preprocessed to [-1, 1] (VaeImageProcessor.preprocess)
 *   moments f32 [n, 2 * latent_channels, h / 8, w / 8]   (mean | logvar, unclamped)
 * h, w are the IMAGE size (multiples of 8 x 2^(n_levels-1)/... : each level but the last halves it).  Same ownership
 * rules as the UNet engine; smi_destroy frees it. */
int smi_vae_workspace_bytes(const smi_vae_config* cfg, int batch, int h, int w, size_t* bytes);
int smi_vae_create(const smi_vae_config* cfg, const smi_weight* weights, int n_weights, int batch, int h, int w,
                   void* workspace, size_t workspace_bytes, void* stream, smi_engine** out);
int smi_vae_encode(smi_engine* e, int n, const float* image, float* moments_out);

/* ---- CLIP text encoder (prompt front end) -----------------------------------------------------------------------------
 * Replaces `text_encoder(tokens)[0]` (train_util.py:119-120) and `text_encoder(tokens, output_hidden_states=True)`
 * -> `[0]`, `.hidden_states[-2]` (train_util.py:139-144).  Causal self-attention, pre-LayerNorm blocks.
 *   weights: the transformers state_dict (`text_model.embeddings.token_embedding.weight`, ...,
 *            `text_model.final_layer_norm.*`, `text_projection.weight` when projection_dim > 0), dtype T
 *   ids         int32 [n, max_positions]   token ids (host tokenizer output)
 *   eos_pos     int32 [n]                  position of the pooled token (transformers: the EOS token)
 *   last_hidden T [n, L, hidden]  final_layer_norm(output of the last layer)            (may be NULL)
 *   penultimate T [n, L, hidden]  output of layer num_layers-1, no final norm = hidden_states[-2]   (may be NULL)
 *   pooled      T [n, projection_dim or hidden]  last_hidden[eos_pos] (x text_projection)           (may be NULL) */
int smi_clip_workspace_bytes(const smi_clip_config* cfg, int batch, size_t* bytes);
int smi_clip_create(const smi_clip_config* cfg, const smi_weight* weights, int n_weights, int batch, void* workspace,
                    size_t workspace_bytes, void* stream, smi_engine** out);
int smi_clip_encode(smi_engine* e, int n, const int32_t* ids, const int32_t* eos_pos, void* last_hidden,
                    void* penultimate, void* pooled);

/* eps = unet(sample, t, ctx[, text_embeds, time_ids]).sample           (train_util.py:290-294, 471-476)
 *   sample      f32 [n, 4, h, w]  (NCHW, already scale_model_input-ed)
 *   ctx         T   [n, ctx_len, cross_attention_dim]
 *   text_embeds T   [n, P] or NULL ; time_ids f32 [n, 6] or NULL        (SD-XL added_cond_kwargs)
 *   lora_down_flat / lora_up_flat: flat fp32 LoRA parameters, NULL or multiplier == 0 -> adaptor off
 *                 (LoRANetwork.__exit__, lora.py:299-301); multiplier = 1.0 * lora_scale inside `with network`.
 *   save_for_backward != 0 keeps the activations the backward needs (the pass that runs with grad enabled).
 *   eps_out     f32 [n, 4, h, w]
 * n may be smaller than the creation batch. */
int smi_unet_forward(smi_engine* e, int n, const float* sample, float timestep, const void* ctx,
                     const void* text_embeds, const float* time_ids, const float* lora_down_flat,
                     const float* lora_up_flat, float multiplier, int save_for_backward, float* eps_out);

/* The four guidance passes of one slider step as ONE UNet pass: samples are independent through the network, so the
 * positive / neutral / negative batches (adaptor off) and the target batch (adaptor on) are stacked into a batch of
 * n = 4 x 2B with the `n_adapted` = 2B target samples LAST.  Only those samples receive the LoRA delta and only their
 * activations are differentiated by smi_unet_backward (whose d_eps then has n_adapted samples).  Same arithmetic per
 * sample as four smi_unet_forward calls; 4x larger GEMM M, ~60 % fewer launches.  n_adapted == n gives
 * smi_unet_forward. */
int smi_unet_forward_batched(smi_engine* e, int n, int n_adapted, const float* sample, float timestep, const void* ctx,
                             const void* text_embeds, const float* time_ids, const float* lora_down_flat,
                             const float* lora_up_flat, float multiplier, int save_for_backward, float* eps_out);

/* smi_unet_forward_batched with ONE ADAPTOR MULTIPLIER PER ADAPTED SAMPLE (`multipliers`: host array of n_adapted
 * floats): the image-slider step runs the adaptor at +s on the `high` image and at -s on the `low` one
 * (trainscripts/imagesliders/train_lora-scale-xl.py:317-381) -- with per-sample multipliers both sides share one UNet
 * pass and one backward.  Implemented for Linear LoRA sites (attention projections, time_emb_proj, conv_shortcut); with
 * conv (c3lier) or DoRA sites and unequal multipliers it returns an error (run the samples in separate passes).  Equal
 * multipliers are exactly smi_unet_forward_batched. */
int smi_unet_forward_multi(smi_engine* e, int n, int n_adapted, const float* sample, float timestep, const void* ctx,
                           const void* text_embeds, const float* time_ids, const float* lora_down_flat,
                           const float* lora_up_flat, const float* multipliers, int save_for_backward, float* eps_out);

/* Backward of the last save_for_backward forward: accumulates (+=) d(loss)/d(lora_down), d(loss)/d(lora_up) into
 * the flat fp32 gradient buffers (same offsets as the parameters).  Activation gradients are propagated only as far
 * as the first adapted layer; no frozen-weight gradients exist (unet.requires_grad_(False), train_lora.py:69).
 * Replaces loss.backward() through the UNet (train_lora.py:298). */
int smi_unet_backward(smi_engine* e, const float* d_eps, float* d_lora_down_flat, float* d_lora_up_flat);

/* Per-kernel-class timing, measured with HIP events recorded on the engine's stream around every launch
 * (measurement aid for bench.py's roofline; off by default, adds two event records per launch when on).
 * smi_profile_read synchronises with the host and returns, per class, the summed device time (ms), the algorithmic
 * FLOPs (2*M*N*K per GEMM/conv; 4*B*H*Nq*Nk*D per attention forward, 10*... per attention backward), the
 * algorithmic HBM bytes (operands read once + result written once) and the launch count since the last enable. */
#define SMI_PROF_GEMM 0  /* gemm_nt_kernel, dense operand (all Linear layers, 1x1 convs)  */
#define SMI_PROF_CONV 1  /* gemm_nt_kernel, implicit-GEMM 3x3 conv (+ the small-channel direct conv) */
#define SMI_PROF_ATTN 2  /* attention forward / backward kernels */
#define SMI_PROF_NORM 3  /* GroupNorm / LayerNorm */
#define SMI_PROF_ELEM 4  /* GEGLU, SiLU, add, concat/split copies, pooling */
#define SMI_PROF_LORA 5  /* LoRA down-projection and weight-gradient reductions */
#define SMI_PROF_NCAT 6
int smi_profile_enable(smi_engine* e, int enable);
int smi_profile_read(smi_engine* e, double* ms, double* flops, double* bytes, int64_t* launches);

/* out[i] = u[i] + g * (t[i] - u[i]) over the two halves of a CFG-doubled prediction   (train_util.py:297-300) */
int smi_cfg_combine(const float* eps_2n, float* out_n, int64_t n_half, float guidance_scale, void* stream);

/* PromptEmbedsPair.loss (prompt_util.py:134-174): loss = mean((target - (neutral + sign_eta*(positive-negative)))^2)
 * sign_eta = +eta for "enhance", -eta for "erase".  Writes the scalar to loss_out[0] and, if dtarget != NULL,
 * d(loss)/d(target).  scratch: >= 256 floats. */
int smi_slider_loss(const float* target, const float* positive, const float* neutral, const float* negative,
                    float sign_eta, int64_t n, float* loss_out, float* dtarget, float* scratch, void* stream);

/* torch.nn.utils.clip_grad_norm_(params, max_norm) (train_lora_xl.py:349; max_norm <= 0: no clipping) followed by
 * one torch.optim.AdamW step (train_lora_xl.py:104,350; train_util.py:1040) over flat fp32 buffers.
 * step counts from 1.  scratch: >= 1025 floats. */
int smi_clip_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float max_norm, float* scratch,
                   void* stream);

/* x = c_x * x + c_eps * eps + c_noise * noise   -- the update of scheduler.step(...).prev_sample
 * (train_util.py:324,705) for DDIM (eta = 0; noise = NULL) and Euler-ancestral; coefficients computed on the host. */
int smi_sched_step(float* x, const float* eps, const float* noise, float c_x, float c_eps, float c_noise, int64_t n,
                   void* stream);

/* ---- single-kernel entry points (used by the parity tests; same launchers the engine uses) ---------------- */
/* Lends `bytes` of device memory at `ws` to the GEMM / conv launchers of THIS host thread as split-K scratch (fp32 partial
 * slabs) until called again with NULL: with it, smi_op_gemm / smi_op_conv3x3 take the same split-K rule the engine's
 * launches take (the engine lends a region of its own workspace per call); without it they never split. */
int smi_op_gemm_scratch(void* ws, size_t bytes);
int smi_op_gemm(int dtype, const void* A, const void* W, void* C, int M, int N, int K, const void* bias,
                const void* res, const float* lora_xa, const float* lora_up, int lora_r, float lora_scale,
                int out_f32, void* stream);
/* The batched-pass form of the adapted Linear (T/lora.py:134-138 under T/train_util.py:449-489's batched UNet call):
 * rows m >= lora_row0 get + lora_scale * xa[m - lora_row0] up^T, rows below none (the frozen samples of the pass).
 * lora_seg > 0: fused projections (q|k|v) -- column block n / lora_seg has its own [lora_seg, r] up matrix (adjacent in
 * lora_up) and its own r columns of xa (row stride r * N / lora_seg). */
int smi_op_gemm_rows(int dtype, const void* A, const void* W, void* C, int M, int N, int K, const void* bias,
                     const void* res, const float* lora_xa, const float* lora_up, int lora_r, float lora_scale,
                     int lora_row0, int lora_seg, void* stream);
int smi_op_conv3x3(int dtype, const void* in, const void* w_packed, const void* bias, void* out, int nb, int hin,
                   int win, int cin, int cout, int stride, int upsample, int transposed, int hout, int wout,
                   void* stream);
int smi_op_attention_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int b, int h,
                         int nq, int nk, int d, float scale, void* stream);
int smi_op_attention_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const float* lse,
                         const void* d_o, void* dq, void* dk, void* dv, float* delta, int b, int h, int nq, int nk,
                         int d, float scale, void* stream);
int smi_op_groupnorm(int dtype, const void* x, const void* gamma, const void* beta, void* y, const void* dy, void* dx,
                     float* scratch, int nb, int hw, int c, int g, float eps, int silu, void* stream);
/* Workgroups of the one-launch (cooperative) GroupNorm that gave up waiting for their sample's other workgroups since the
 * library was loaded: 0 unless the device lost workgroups (the wait is bounded, csrc/norm.hip); -1 if it cannot be read.
 * A diagnostic for tests -- the reference has no counterpart (torch.nn.GroupNorm is one ATen call). */
int smi_gn_coop_timeouts(void);
int smi_op_layernorm(int dtype, const void* x, const void* gamma, const void* beta, void* y, const void* dy, void* dx,
                     float* mean_rstd, int m, int c, float eps, void* stream);
int smi_op_geglu(int dtype, const void* proj, void* out, const void* dout, void* dproj, int m, int c4, void* stream);
/* ff.net.0 with the GEGLU gate fused into the GEMM epilogue (the engine's forward path for
 * diffusers GEGLU: proj = x W^T + b [M, N]; out[M, N/2] = proj[:, :N/2] * gelu(proj[:, N/2:])).  Rows >= proj_row0 of
 * proj are also written (row m at proj + m*N); pass proj_row0 = M to keep none.  Fails when the layout is not
 * supported by the fused kernel (N % 256 != 0, misaligned operands). */
int smi_op_gemm_geglu(int dtype, const void* A, const void* W, const void* bias, void* out, void* proj, int M, int N,
                      int K, int proj_row0, void* stream);
int smi_op_lora_down(int dtype, const void* x, const float* a, float* xa, int m, int k, int r, void* stream);
/* out[m, r] (fp32) = x[m, k] * s[r, k]^T on 16-bit operands, r = 16 or 32, k % 128 == 0: the engine's kernel for
 * xa = x * lora_down^T and dxa = dy * lora_up (T/lora.py:134-138 and its backward) on the 16-bit shadow parameters */
int smi_op_lora_skinny(int dtype, const void* x, const void* s, float* out, int m, int r, int k, void* stream);
/* dw[r, k] += alpha * p[m, r]^T x[m, k]: the engine's grouped weight-gradient reduction (T/lora.py:134-138 backward)
 * run on a one-job table.  scratch: ceil(m / 64) * r * k + 256 floats (partials + the table). */
int smi_op_lora_wgrad(int dtype, const float* p, const void* x, float* dw, int m, int k, int r, float alpha,
                      float* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMI_H_ */
