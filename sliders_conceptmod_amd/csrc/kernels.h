// Internal launcher API of the HIP kernels (host side).  All launchers enqueue on `stream`, never synchronise,
// return 0 on success or a negative code after smi::set_error().  Activations are token-major ("NHWC"):
// an image tensor is [N, H*W, C] with C contiguous.
#pragma once
#include <vector>

#include "smi_common.h"

namespace smi {

// ---------------------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] * W[N,K]^T  (+ epilogue), W rows K-contiguous (torch nn.Linear layout).
// With conv != 0, A is gathered on the fly from an NHWC image (implicit GEMM, K = 9*Cin ordered (kh,kw,ci)).
// Epilogue (all fp32, one rounding at the store):  + bias[n] + rowvec[m / rows_per_vec][n]
//     + lora_scale * sum_q xa[m][q] * up[n][q] + res[m][n]
// ---------------------------------------------------------------------------------------------------------------
struct GemmParams {
  int dtype = DT_F16;
  const void* A = nullptr;
  int64_t lda = 0;
  const void* W = nullptr;
  void* C = nullptr;
  int64_t ldc = 0;
  int out_f32 = 0;
  int M = 0, N = 0, K = 0;
  const void* bias = nullptr;    // T [N]
  const void* res = nullptr;     // T [M, ldr]
  int64_t ldr = 0;
  const void* rowvec = nullptr;  // T [M / rows_per_vec, N]
  int rows_per_vec = 1;
  int64_t ld_rowvec = 0;         // row stride of rowvec in elements (0 = N): a column block of a wider matrix
  // LoRA rank-r delta: + lora_scale * sum_q xa[m*ld_xa + seg*r + q] * up[n*up_sn + q*up_sq], seg = n / lora_seg.
  // Forward: xa = x*A^T, up = lora_up.weight [N, r] (up_sn = r, up_sq = 1); with fused q|k|v projections the three
  // [C, r] up matrices are adjacent in the flat buffer and lora_seg = C selects each column block's own xa.
  // Backward (dX): xa = dY*up, "up" = lora_down.weight [r, K] read transposed (up_sn = 1, up_sq = K).
  const float* lora_xa = nullptr;
  int64_t ld_xa = 0;
  const float* lora_up = nullptr;
  int64_t up_sn = 0, up_sq = 1;
  int lora_r = 0;
  int lora_seg = 0;
  float lora_scale = 0.f;
  int lora_row0 = 0;  // rows m < lora_row0 get no delta (frozen samples of a batched pass); xa row = m - lora_row0
  // fused GEGLU (diffusers GEGLU: proj(x).chunk(2) -> hidden * gelu(gate)), dense v2 kernel only: N = 2 * N_half,
  // every 128-column tile holds 64 hidden + their 64 gate columns; geglu_out [M, N/2] receives hidden * gelu(gate)
  // computed from the 16-bit-rounded projection (bit-identical to projection + separate GEGLU kernel); the projection
  // itself is stored to C only for rows m >= geglu_row0 (the rows whose backward needs it).  bias only.
  void* geglu_out = nullptr;
  int geglu_row0 = 0;
  // implicit-GEMM 3x3 convolution (pad 1)
  int conv = 0;
  int Nb = 0, Hin = 0, Win = 0, Cin = 0, Hout = 0, Wout = 0;
  int stride = 1;      // forward: in = out*stride + k - pad
  int pad = 1;         // 1 everywhere in the UNet; 0 for the VAE encoder's (0,1,0,1)-padded stride-2 downsampler
  int upsample = 0;    // nearest-2x upsample of the input folded into the gather
  int transposed = 0;  // gradient of a strided conv: out = (in + 1 - k) / stride
  // split-K (internal to launch_gemm, gemm2 kernels only): the workgroups of slice s accumulate K-tiles
  // [s nk / ksplit, (s + 1) nk / ksplit) and store fp32 partials to C + s * M * ldc (out_f32 form, no epilogue)
  int ksplit = 0;
};
int launch_gemm(const GemmParams& p, hipStream_t stream);
// Scratch for split-K partial sums (fp32 slabs), set by whoever owns memory (the engine, per call; tests through
// smi_op_gemm_scratch) for the calling host thread; without it launch_gemm never splits.  Launches that use it are ordered
// on their stream, so one buffer serves every launch of a pass.
void set_gemm_scratch(void* ws, size_t bytes);
bool gemm_geglu_supported(const GemmParams& p);  // can launch_gemm take p.geglu_out?

// direct 3x3 conv for tiny channel counts (conv_in: Cin=4; conv_out: Cout=4 and their gradients)
// in [Nb,H,W,Cin] T ; w f32-free: T [Cout][9*Cin] ; out [Nb,H,W,Cout] (T or f32), stride 1 pad 1
int launch_conv3x3_small(int dtype, const void* in, const void* w, const void* bias, void* out, int out_f32, int Nb,
                         int H, int W, int Cin, int Cout, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// attention: Q [B, Nq, ldq] (head h at column h*D), K/V [B, Nk, ldk/ldv], O [B, Nq, ldo]; lse f32 [B, H, Nq]
// ---------------------------------------------------------------------------------------------------------------
struct AttnParams {
  int dtype = DT_F16;
  const void *Q = nullptr, *K = nullptr, *V = nullptr;
  void* O = nullptr;
  float* lse = nullptr;
  int64_t ldq = 0, ldk = 0, ldv = 0, ldo = 0;
  int B = 0, H = 0, Nq = 0, Nk = 0, D = 0;
  float scale = 1.f;
  int causal = 0;  // forward only: key j is visible to query i iff j <= i (CLIP text encoder)
  // backward
  const void* dO = nullptr;
  int64_t lddo = 0;
  float* delta = nullptr;  // f32 [B, H, Nq] workspace
  void *dQ = nullptr, *dK = nullptr, *dV = nullptr;
  int64_t lddq = 0, lddk = 0, lddv = 0;
};
int launch_attn_fwd(const AttnParams& p, hipStream_t stream);
int launch_attn_bwd(const AttnParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// normalisation
// ---------------------------------------------------------------------------------------------------------------
// GroupNorm over [Nb, HW, C]; writes per-(n,c) affine a,b (f32 [Nb,C] each: y = x*a + b) into `ab` ([2,Nb,C]) and
// y = (silu?)(x*a+b).  `partial` is scratch f32, gn_partial_floats(Nb, HW, G) long ([Nb, nchunk, G, 2], nchunk from gn_num_chunks(HW), + the fold).
int gn_num_chunks(int HW);
int gn_coop_timeouts();  // workgroups of the cooperative one-launch form that gave up waiting (0), norm.hip
size_t gn_partial_floats(int Nb, int HW, int G);  // floats of `partial` scratch (chunk partials + their fold for maps > 128 x 128)
// ab: [2][Nb][C] -- a = ab, b = ab + Nb*C.  The backward takes the two halves as separate pointers so that it can run
// on a sample sub-range of a larger forward batch.
int launch_groupnorm_fwd(int dtype, const void* x, const void* gamma, const void* beta, void* y, float* ab,
                         float* mean_rstd, float* partial, int Nb, int HW, int C, int G, float eps, int silu,
                         hipStream_t stream);
// dx for y = silu?(GN(x)); needs x, gamma, beta, ab and mean_rstd from the forward.
// `partial`: scratch f32, gn_partial_floats(Nb, HW, G) long
// `add` (optional, may alias dx): gradient already accumulated for x, added in the same pass
int launch_groupnorm_bwd(int dtype, const void* x, const void* dy, const void* gamma, const void* beta,
                         const float* a, const float* b, const float* mean_rstd, const void* add, void* dx,
                         float* partial, int Nb, int HW, int C, int G, int silu, hipStream_t stream);
int launch_layernorm_fwd(int dtype, const void* x, const void* gamma, const void* beta, void* y, float* mean_rstd,
                         int M, int C, float eps, hipStream_t stream);
int launch_layernorm_bwd(int dtype, const void* x, const void* dy, const void* gamma, const float* mean_rstd,
                         const void* add, void* dx, int M, int C, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// elementwise / data movement
// ---------------------------------------------------------------------------------------------------------------
int launch_geglu_fwd(int dtype, const void* proj, void* out, int M, int C4, hipStream_t stream);  // proj [M, 2*C4]
int launch_geglu_bwd(int dtype, const void* proj, const void* dout, void* dproj, int M, int C4, hipStream_t stream);
int launch_silu(int dtype, const void* x, void* y, int64_t n, hipStream_t stream);
int launch_add(int dtype, const void* a, const void* b, void* y, int64_t n, hipStream_t stream);  // y = a + b
// CLIP text encoder pieces: y = quick_gelu(x) (kind 0) / gelu(x) (kind 1); token + position embedding lookup; row gather
int launch_act(int dtype, const void* x, void* y, int64_t n, int kind, hipStream_t stream);
int launch_embed(int dtype, const int* ids, const void* tok, const void* pos, void* out, int64_t rows, int L, int d,
                 int vocab, hipStream_t stream);
int launch_gather_rows(int dtype, const void* src, const int* idx, void* out, int n, int L, int d, hipStream_t stream);
// copy [M, C] (src row stride lds) into dst columns [col0, col0+C) of a [M, ldd] tensor (concat / split)
int launch_copy_cols(int dtype, const void* src, int64_t lds, void* dst, int64_t ldd, int col0, int M, int C,
                     hipStream_t stream);
int launch_nchw_to_nhwc(int dtype, const void* src, int src_f32, void* dst, int Nb, int C, int HW, int Cpad,
                        float scale, hipStream_t stream);
// f32 NCHW -> T token-major, sample n multiplied by scale_dev[n] (backward entry: loss-scaled d_eps)
int launch_nchw_to_nhwc_scaled(int dtype, const float* src, void* dst, int Nb, int C, int HW, int Cpad,
                               const float* scale_dev, hipStream_t stream);
int launch_nhwc_to_nchw_f32(const float* src, float* dst, int Nb, int C, int HW, hipStream_t stream);
// sinusoidal embedding (cos | sin, flip_sin_to_cos, freq shift 0): vals f32 [n] -> out T [n, dim]
int launch_timestep_embed(int dtype, const float* vals, void* out, int n, int dim, hipStream_t stream);
// dx[n,h,w,c] = sum_{2x2} du[n,2h+a,2w+b,c]
int launch_pool2x2_sum(int dtype, const void* du, void* dx, int Nb, int H, int W, int C, hipStream_t stream);
// P[r][:] = softmax(scale * S[r][:]) for rows of fp32 scores (materialised attention of the VAE's single 512-wide head)
int launch_softmax_rows(int dtype, const float* S, void* P, int rows, int cols, float scale, hipStream_t stream);
// y[m][j] = b[j] + sum_i x[m][i] W[j][i] on fp32 rows of width C <= 16 (the VAE's 1x1 quant_conv), W / b of dtype T
int launch_chan_mix(int dtype, const float* x, const void* W, const void* b, float* y, int64_t M, int C,
                    hipStream_t stream);
// out[n][c] = mul * sum over the HW rows of sample n of x[n][hw][c]   (deterministic)
size_t colsum_scratch_floats(int Nb, int C);  // fp32 scratch launch_colsum needs
int launch_colsum(int dtype, const void* x, void* out, float* scratch, int Nb, int HW, int C, float mul,
                  hipStream_t stream);
// dst[M][cpad] (T) = src[M][0..cols) (fp32, row stride lds) * mul, zero padded
int launch_f32_to_padded(int dtype, const float* src, int lds, int cols, void* dst, int cpad, int64_t M, float mul,
                         hipStream_t stream);
// power-of-two loss scales chosen on device, ONE PER SAMPLE, from max|d_eps[sample]| (keeps 16-bit activation
// gradients in range and makes a sample's backward independent of its batch mates):
// scale_out[j] = scale, scale_out[inv_off + j] = 1 / scale
int launch_grad_scale(const float* d_eps, int n_samples, int64_t per_sample, float* scale_out, int inv_off,
                      hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// LoRA skinny kernels (rank r <= 32)
// ---------------------------------------------------------------------------------------------------------------
// xa[M, r] = X[M, K] (T, row stride ldx) * A[r, K]^T (f32)                       (lora_down / dXA = dY * up)
int launch_lora_down(int dtype, const void* X, int64_t ldx, const float* A, int64_t lda_r, int64_t lda_k, float* xa,
                     int64_t ld_xa, int M, int K, int r, hipStream_t stream);
// dW (+)= alpha * P[M, r]^T (f32) * X[M, K] (T) -- weight-gradient reductions over M.
// Grouped form (lora.hip): one table entry per rank-r reduction dW (+)= alpha * P^T X of a backward pass
struct WgradJob {
  const void* X;           // [M, K] 16-bit operand (dY or the layer input), row stride ldx
  const float* P;          // [M, >= nseg * r] fp32 (xa or dxa), row stride ldp
  float* dW;               // output base; element (q, k) at dW[q * so_r + k * so_k]
  float* partial;          // scratch [nsplit][r][K]
  const float* row_scale;  // per-sample factors applied to the rows of P (nullptr: none)
  int64_t ldx, ldp, so_r, so_k;
  int M, K, r, seg_cols, rows_per_sample;
  float alpha;
  // conv_tap >= 0: X is an image [n][Hin][Win][K] and row m = (n, oy, ox) of the OUTPUT grid reads the input pixel of
  // filter tap (ky, kx) = (conv_tap / 3, conv_tap % 3) of a 3x3 / pad-1 conv (zero outside): the k x k LoRA down filter
  int conv_tap, Hin, Win, Hout, Wout, conv_stride, conv_ups;
  int cw, ncolblk, rows_per_wg, nsplit;  // geometry (wgrad_job_plan)
  int wg0, fb0;                          // first workgroup in the partial / final grid (wgrad_grouped_finish)
};
void wgrad_job_plan(WgradJob& j);
size_t wgrad_job_scratch_floats(const WgradJob& j);
// 16-bit shadow copies of the LoRA matrices as GEMM operands (see lora.hip); sites_dev: device array of HostLoraPrepSite
struct HostLoraPrepSite {
  int64_t off_down, off_up, dst_down, dst_up;
  int r, nseg, K, cs, rows_pad;
  // conv site (c3lier): 0 = Linear; 1 / 2 = 3x3 conv down [r][K][3][3] (K = Cin, cs = Cout) whose gradient filter
  // [K][9*64] at dst_gw has flipped (1) or plain (2: stride-2 layer) taps
  int conv;
  int64_t dst_gw;
};
// out[M, R] (fp32, row stride ldo) = X[M, K] * S[R, K]^T for R = 16 / 32 (LoRA shadow products), K % 128 == 0
int launch_row_scale_f32(float* x, int ld, int M, int N, const float* row_mul, int rows_per_mul, hipStream_t stream);
bool lora_skinny_supported(const void* X, int64_t ldx, const void* S, const float* out, int ldo, int M, int R, int K);
int launch_lora_skinny(int dtype, const void* X, int64_t ldx, const void* S, float* out, int ldo, int M, int R, int K,
                       hipStream_t stream, const float* row_mul = nullptr, int rows_per_mul = 1);
// ---- DoRA (T/dora.py:124-162): dW = (W + up down) * (g / ||W + up down||_col) - W per adapted Linear (fused segments share W)
struct DoraSite {
  const void* W;                       // frozen [nseg * cs, K], 16-bit
  int64_t off_down, off_up, off_dora;  // of segment 0; segment s at + s*r*K, + s*cs*r, + s*K
  void* dW;                            // out: [nseg * cs, K] 16-bit, = lscale * dW
  void* dWt;                           // out (launch_dora_transpose): [K, nseg * cs], the dX GEMM's operand
  float* cnorm;                        // out: [nseg, K] column norms (detached in the backward)
  int r, nseg, K, cs;
  float scale;                         // alpha / rank
};
// per adapted forward: column norms, then the scaled delta weights (lscale = mult * site.scale folded in)
int launch_dora_prep(int dtype, const DoraSite* sites_dev, const DoraSite* sites_host, int n_sites, const float* down,
                     const float* up, float mult, hipStream_t stream);
// dWt = dW^T of every site in ONE launch (the backward of a saved forward needs them; 280 launches of 12 us before)
int launch_dora_transpose(int dtype, const DoraSite* sites_dev, const DoraSite* sites_host, int n_sites,
                          hipStream_t stream);
// dst[c][m] = src[m][c] * (f ? f[m / rows_per_sample] : 1) for m < M, 0 for M <= m < Mp      (dst [C, Mp], 16-bit)
int launch_transpose_scaled(int dtype, const void* src, int64_t lds, void* dst, int M, int C, int Mp, const float* f,
                            int rows_per_sample, hipStream_t stream);
// gradients of one DoRA Linear from G = dY^T X (fp32 [nseg*cs, K]): d(dora_scale), d(down) (column-wise), d(up) (row-wise);
// everything x alpha x *alpha_dev; accumulates (+=) into the flat gradient buffers
// `scratch`: dora_grad_scratch_floats(site) floats (row-slice partials of the column-wise sums)
size_t dora_grad_scratch_floats(const DoraSite& site);
int launch_dora_grads(int dtype, const DoraSite& site, const float* G, const float* down, const float* up,
                      float* d_down, float* d_up, float alpha, const float* alpha_dev, float* scratch,
                      hipStream_t stream);
// scale_buf[0..n) per-sample loss scales -> out[0] = min, out[1] = 1 / min, out[2 + j] = min / scale_j
int launch_scale_min(const float* scale_buf, int n, float* out, hipStream_t stream);
int wgrad_grouped_finish(std::vector<WgradJob>& jobs);
int launch_lora_wgrad_grouped(int dtype, const std::vector<WgradJob>& jobs, const WgradJob* jobs_dev,
                              hipStream_t stream);
int launch_lora_prep(int dtype, const void* sites_dev, int n_sites, const float* down, const float* up, void* shadow,
                     hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// slider-step elementwise ops (K9-K11)
// ---------------------------------------------------------------------------------------------------------------
int launch_cfg_combine(const float* eps2 /*[2B,...]*/, float* out /*[B,...]*/, int64_t n_half, float g,
                       hipStream_t stream);
// loss = mean((target - (neutral + sign*eta*(positive - negative)))^2) ; dtarget = 2*(target-goal)/n
int launch_slider_loss(const float* target, const float* positive, const float* neutral, const float* negative,
                       float sign_eta, int64_t n, float* loss_out, float* dtarget, float* scratch,
                       hipStream_t stream);
int launch_axpby(float* y, const float* x, float a, float b, int64_t n, hipStream_t stream);  // y = a*x + b*y
// global-norm clip (max_norm<=0: none) + AdamW on flat f32 buffers; state m,v; step>=1
int launch_clip_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, float max_norm, float* scratch /*>= 1024+2 floats*/,
                      hipStream_t stream);
// DDIM (eta=0): x = c0*x + c1*eps ; Euler-a: x = x + eps*dt + noise*sigma_up
int launch_sched_affine(float* x, const float* eps, const float* noise, float c_x, float c_eps, float c_noise,
                        int64_t n, hipStream_t stream);

}  // namespace smi
