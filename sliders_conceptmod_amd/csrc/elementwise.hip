// HBM-bound elementwise / data-movement kernels (16-byte lanes, grid-stride) and the slider-step ops
// (CFG combine, guidance loss + gradient, global-norm clip + AdamW, scheduler update).
#include "kernels.h"
#include <cstdlib>

namespace smi {
namespace {

inline int ew_grid(int64_t n_threads) {
  int64_t g = (n_threads + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
#define GSTRIDE(i, total)                                                         \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); \
       i += (int64_t)gridDim.x * blockDim.x)

// GEGLU: out[m, c] = proj[m, c] * gelu(proj[m, C4 + c])                 (diffusers GEGLU, exact erf gelu)
template <typename T>
__global__ void geglu_fwd_kernel(const T* __restrict__ proj, T* __restrict__ out, int64_t M, int C4) {
  const int c8 = C4 / 8;
  GSTRIDE(i, M * c8) {
    const int64_t m = i / c8;
    const int c = (int)(i - m * c8) * 8;
    Pack8<T> h, g, o;
    h.u = *reinterpret_cast<const u32x4*>(proj + m * 2 * C4 + c);
    g.u = *reinterpret_cast<const u32x4*>(proj + m * 2 * C4 + C4 + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = from_f<T>(to_f(h.e[e]) * gelu_f(to_f(g.e[e])));
    *reinterpret_cast<u32x4*>(out + m * C4 + c) = o.u;
  }
}
template <typename T>
__global__ void geglu_bwd_kernel(const T* __restrict__ proj, const T* __restrict__ dout, T* __restrict__ dproj,
                                 int64_t M, int C4) {
  const int c8 = C4 / 8;
  GSTRIDE(i, M * c8) {
    const int64_t m = i / c8;
    const int c = (int)(i - m * c8) * 8;
    Pack8<T> h, g, d, oh, og;
    h.u = *reinterpret_cast<const u32x4*>(proj + m * 2 * C4 + c);
    g.u = *reinterpret_cast<const u32x4*>(proj + m * 2 * C4 + C4 + c);
    d.u = *reinterpret_cast<const u32x4*>(dout + m * C4 + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gv = to_f(g.e[e]), dv = to_f(d.e[e]);
      oh.e[e] = from_f<T>(dv * gelu_f(gv));
      og.e[e] = from_f<T>(dv * to_f(h.e[e]) * dgelu_f(gv));
    }
    *reinterpret_cast<u32x4*>(dproj + m * 2 * C4 + c) = oh.u;
    *reinterpret_cast<u32x4*>(dproj + m * 2 * C4 + C4 + c) = og.u;
  }
}

template <typename T, int OP>  // 0: silu(a)   1: a + b   2: quick_gelu(a) = a * sigmoid(1.702 a)   3: gelu(a) (erf)
__global__ void ew_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n8) {
  GSTRIDE(i, n8) {
    Pack8<T> x, z, o;
    x.u = *reinterpret_cast<const u32x4*>(a + i * 8);
    if (OP == 1) z.u = *reinterpret_cast<const u32x4*>(b + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = to_f(x.e[e]);
      o.e[e] = from_f<T>(OP == 0 ? silu_f(v) : OP == 1 ? v + to_f(z.e[e]) : OP == 2 ? v / (1.f + __expf(-1.702f * v))
                                                                                      : gelu_f(v));
    }
    *reinterpret_cast<u32x4*>(y + i * 8) = o.u;
  }
}

template <typename T>
__global__ void copy_cols_kernel(const T* __restrict__ src, int64_t lds, T* __restrict__ dst, int64_t ldd, int col0,
                                 int64_t M, int C) {
  const int c8 = C / 8;
  GSTRIDE(i, M * c8) {
    const int64_t m = i / c8;
    const int c = (int)(i - m * c8) * 8;
    *reinterpret_cast<u32x4*>(dst + m * ldd + col0 + c) = *reinterpret_cast<const u32x4*>(src + m * lds + c);
  }
}

// NCHW (T or f32) -> token-major [Nb, HW, Cpad] T (channels >= C zero filled), value * scale (or * *scale_dev)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const void* __restrict__ src, int src_f32, T* __restrict__ dst, int Nb, int C,
                                    int HW, int Cpad, float scale, const float* __restrict__ scale_dev) {
  GSTRIDE(i, (int64_t)Nb * HW * Cpad) {
    const int c = (int)(i % Cpad);
    const int64_t pix = i / Cpad;
    const int64_t n = pix / HW, hw = pix - n * HW;
    const float sc = scale_dev ? scale * scale_dev[n] : scale;  // scale_dev: one factor per sample
    float v = 0.f;
    if (c < C) {
      const int64_t si = (n * C + c) * HW + hw;
      v = src_f32 ? reinterpret_cast<const float*>(src)[si] : to_f(reinterpret_cast<const T*>(src)[si]);
    }
    dst[i] = from_f<T>(v * sc);
  }
}
__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int Nb, int C,
                                        int HW) {
  GSTRIDE(i, (int64_t)Nb * C * HW) {
    const int64_t hw = i % HW;
    const int64_t nc = i / HW;
    const int64_t n = nc / C, c = nc - n * C;
    dst[i] = src[(n * HW + hw) * C + c];
  }
}

// diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos(t f_i) | sin(t f_i)]
template <typename T>
__global__ void timestep_embed_kernel(const float* __restrict__ vals, T* __restrict__ out, int n, int dim) {
  const int half = dim / 2;
  GSTRIDE(i, (int64_t)n * dim) {
    const int j = (int)(i % dim);
    const int64_t r = i / dim;
    const int k = j < half ? j : j - half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
    const float a = vals[r] * freq;
    out[i] = from_f<T>(j < half ? cosf(a) : sinf(a));
  }
}

template <typename T>
__global__ void pool2x2_sum_kernel(const T* __restrict__ du, T* __restrict__ dx, int Nb, int H, int W, int C) {
  const int c8 = C / 8;
  GSTRIDE(i, (int64_t)Nb * H * W * c8) {
    const int c = (int)(i % c8) * 8;
    const int64_t pix = i / c8;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        Pack8<T> v;
        v.u = *reinterpret_cast<const u32x4*>(du + ((n * 2 * H + 2 * y + a) * 2 * W + 2 * x + b) * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += to_f(v.e[e]);
      }
    Pack8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = from_f<T>(acc[e]);
    *reinterpret_cast<u32x4*>(dx + pix * C + c) = o.u;
  }
}

// out[n][c] = mul * sum_{hw} x[n][hw][c]  (gradient of a per-sample row vector added to every pixel: the time-embedding
// projection under a c3lier adaptor).  Two launches: partial sums over COLSUM_RS row slices per sample (a thread owns 8
// columns = one 16-byte load per row; 8 row lanes per workgroup meet in LDS in a fixed order), then a fixed-order sum of the
// slices -- deterministic.  (The one-launch form had C / 64 workgroups per sample walk all HW rows with 2-byte loads:
// 80 us per call at 4096 x 320, 1.75 ms per SD-1.4 c3lier step.)
constexpr int COLSUM_RS = 16;
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ part, int HW,
                                                             int C) {
  __shared__ float red[8][32][8];
  const int n = blockIdx.z, rs = blockIdx.y;
  const int cv = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cv) * 8;
  const int rows = (HW + COLSUM_RS - 1) / COLSUM_RS;
  const int r0 = rs * rows, r1 = min(HW, r0 + rows);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const T* p = x + (int64_t)n * HW * C + c;
    for (int r = r0 + rl; r < r1; r += 8) {
      Pack8<T> v;
      v.u = *reinterpret_cast<const u32x4*>(p + (int64_t)r * C);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += to_f(v.e[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cv][e] = acc[e];
  __syncthreads();
  if (rl == 0 && c < C) {
    float* o = part + ((int64_t)n * COLSUM_RS + rs) * C + c;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) t += red[j][cv][e];
      o[e] = t;
    }
  }
}
template <typename T>
__global__ void colsum_final_kernel(const float* __restrict__ part, T* __restrict__ out, int Nb, int C, float mul) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Nb * C) return;
  const int n = i / C, c = i - n * C;
  float t = 0.f;
  for (int rs = 0; rs < COLSUM_RS; ++rs) t += part[((int64_t)n * COLSUM_RS + rs) * C + c];
  out[i] = from_f<T>(t * mul);
}
// CLIP text embeddings: out[n*L + l][:] = tok[ids[n*L + l]][:] + pos[l][:]
template <typename T>
__global__ void embed_kernel(const int* __restrict__ ids, const T* __restrict__ tok, const T* __restrict__ pos,
                             T* __restrict__ out, int64_t rows, int L, int d8, int vocab) {
  GSTRIDE(i, rows * d8) {
    const int c = (int)(i % d8);
    const int64_t r = i / d8;
    int id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    Pack8<T> a, b, o;
    a.u = *reinterpret_cast<const u32x4*>(tok + ((int64_t)id * d8 + c) * 8);
    b.u = *reinterpret_cast<const u32x4*>(pos + ((int64_t)(r % L) * d8 + c) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = from_f<T>(to_f(a.e[e]) + to_f(b.e[e]));
    *reinterpret_cast<u32x4*>(out + i * 8) = o.u;
  }
}
// out[n][:] = src[n * L + idx[n]][:]
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, const int* __restrict__ idx, T* __restrict__ out, int n,
                                   int L, int d8) {
  GSTRIDE(i, (int64_t)n * d8) {
    const int c = (int)(i % d8);
    const int64_t r = i / d8;
    int j = idx[r];
    j = j < 0 ? 0 : (j >= L ? L - 1 : j);
    *reinterpret_cast<u32x4*>(out + i * 8) = *reinterpret_cast<const u32x4*>(src + ((r * L + j) * d8 + c) * 8);
  }
}
// one workgroup per row: max, sum of exp, normalised write (three passes over a row that sits in L2)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, T* __restrict__ P, int cols,
                                                           float scale) {
  __shared__ float sh[4];
  const float* s = S + (int64_t)blockIdx.x * cols;
  T* p = P + (int64_t)blockIdx.x * cols;
  float m = -3.0e38f;
  for (int i = threadIdx.x; i < cols; i += 256) m = fmaxf(m, s[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) * scale;
  __syncthreads();
  float sum = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) sum += __expf(s[i] * scale - m);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / ((sh[0] + sh[1]) + (sh[2] + sh[3]));
  for (int i = threadIdx.x; i < cols; i += 256) p[i] = from_f<T>(__expf(s[i] * scale - m) * inv);
}
template <typename T>
__global__ void chan_mix_kernel(const float* __restrict__ x, const T* __restrict__ W, const T* __restrict__ b,
                                float* __restrict__ y, int64_t M, int C) {
  GSTRIDE(i, M * C) {
    const int j = (int)(i % C);
    const int64_t m = i / C;
    float acc = to_f(b[j]);
    for (int k = 0; k < C; ++k) acc += x[m * C + k] * to_f(W[j * C + k]);
    y[i] = acc;
  }
}
// dst[m][0..cpad) = (T)(src[m][0..cols) * mul), zero beyond cols   (fp32 rank-r rows -> a 64-channel MFMA operand)
template <typename T>
__global__ void f32_to_padded_kernel(const float* __restrict__ src, int lds, int cols, T* __restrict__ dst, int cpad,
                                     int64_t M, float mul) {
  GSTRIDE(i, M * cpad) {
    const int c = (int)(i % cpad);
    const int64_t m = i / cpad;
    dst[i] = from_f<T>(c < cols ? src[m * lds + c] * mul : 0.f);
  }
}

// ---- reductions to a scalar: block partials into scratch, last-stage by one block (deterministic order)
template <int OP>  // 0: max|x|   1: sum x^2
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float* __restrict__ x, int64_t n,
                                                             float* __restrict__ partial) {
  __shared__ float sh[4];
  float acc = 0.f;
  GSTRIDE(i, n) {
    const float v = x[i];
    acc = OP == 0 ? fmaxf(acc, fabsf(v)) : acc + v * v;
  }
  acc = OP == 0 ? wave_max(acc) : wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    partial[blockIdx.x] = OP == 0 ? fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) : (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// one block per sample: scale[j] = 2^floor(log2(target / max|x_j|)) (power of two: exact), inv[j] = 1 / scale[j]
__global__ __launch_bounds__(256) void grad_scale_kernel(const float* __restrict__ x, int64_t per, float* __restrict__ out,
                                                         int inv_off, float target) {
  __shared__ float sh[4];
  const float* xs = x + (int64_t)blockIdx.x * per;
  float amax = 0.f;
  for (int64_t i = threadIdx.x; i < per; i += 256) amax = fmaxf(amax, fabsf(xs[i]));
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    amax = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    float s = 1.f;
    if (amax > 0.f && isfinite(amax)) {
      int e = (int)floorf(log2f(target / amax));
      e = e > 40 ? 40 : (e < -40 ? -40 : e);
      s = exp2f((float)e);
    }
    out[blockIdx.x] = s;
    out[inv_off + blockIdx.x] = 1.f / s;
  }
}

__global__ void cfg_combine_kernel(const float* __restrict__ e2, float* __restrict__ out, int64_t nh, float g) {
  GSTRIDE(i, nh) {
    const float u = e2[i], t = e2[nh + i];
    out[i] = u + g * (t - u);
  }
}

__global__ __launch_bounds__(256) void slider_loss_partial_kernel(const float* __restrict__ tg,
                                                                  const float* __restrict__ po,
                                                                  const float* __restrict__ ne,
                                                                  const float* __restrict__ ng, float sign_eta,
                                                                  int64_t n, float* __restrict__ dtarget,
                                                                  float* __restrict__ partial) {
  __shared__ float sh[4];
  float acc = 0.f;
  const float k = 2.f / (float)n;
  GSTRIDE(i, n) {
    const float goal = ne[i] + sign_eta * (po[i] - ng[i]);
    const float d = tg[i] - goal;
    acc += d * d;
    if (dtarget) dtarget[i] = k * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void sum_finalize_kernel(const float* __restrict__ partial, int np, float mul, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < np; ++i) s += partial[i];
    out[0] = s * mul;
  }
}

__global__ void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, float a, float b, int64_t n) {
  GSTRIDE(i, n) y[i] = a * x[i] + b * y[i];
}

// torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (decoupled weight decay), single fused pass.
// norm_sq_dev: device scalar holding sum g^2 (only read when max_norm > 0)
__global__ void clip_adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                  float bc1, float bc2_sqrt, float max_norm, const float* __restrict__ norm_sq_dev) {
  float coef = 1.f;
  if (max_norm > 0.f) {
    const float total = sqrtf(norm_sq_dev[0]);
    coef = fminf(max_norm / (total + 1e-6f), 1.f);
  }
  GSTRIDE(i, n) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}

__global__ void sched_affine_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                    const float* __restrict__ noise, float cx, float ce, float cn, int64_t n) {
  GSTRIDE(i, n) {
    float v = cx * x[i] + ce * eps[i];
    if (noise) v += cn * noise[i];
    x[i] = v;
  }
}

}  // namespace

int launch_geglu_fwd(int dtype, const void* proj, void* out, int M, int C4, hipStream_t stream) {
  SMI_CHECK(C4 % 8 == 0, "geglu: C4 %% 8");
  const int grid = ew_grid((int64_t)M * C4 / 8);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(geglu_fwd_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)proj, (f16*)out, (int64_t)M, C4);
  else
    hipLaunchKernelGGL(geglu_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)proj, (bf16*)out, (int64_t)M, C4);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_geglu_bwd(int dtype, const void* proj, const void* dout, void* dproj, int M, int C4, hipStream_t stream) {
  SMI_CHECK(C4 % 8 == 0, "geglu: C4 %% 8");
  const int grid = ew_grid((int64_t)M * C4 / 8);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(geglu_bwd_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)proj, (const f16*)dout, (f16*)dproj, (int64_t)M, C4);
  else
    hipLaunchKernelGGL(geglu_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)proj, (const bf16*)dout, (bf16*)dproj, (int64_t)M, C4);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_silu(int dtype, const void* x, void* y, int64_t n, hipStream_t stream) {
  SMI_CHECK(n % 8 == 0, "silu: n %% 8");
  const int grid = ew_grid(n / 8);
  if (dtype == DT_F16)
    hipLaunchKernelGGL((ew_kernel<f16, 0>), dim3(grid), dim3(256), 0, stream, (const f16*)x, nullptr, (f16*)y, n / 8);
  else
    hipLaunchKernelGGL((ew_kernel<bf16, 0>), dim3(grid), dim3(256), 0, stream, (const bf16*)x, nullptr, (bf16*)y, n / 8);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_act(int dtype, const void* x, void* y, int64_t n, int kind, hipStream_t stream) {
  SMI_CHECK(n % 8 == 0 && (kind == 0 || kind == 1), "act: n %% 8, kind 0 (quick_gelu) or 1 (gelu)");
  const int grid = ew_grid(n / 8);
#define L(TT_, OP_) hipLaunchKernelGGL((ew_kernel<TT_, OP_>), dim3(grid), dim3(256), 0, stream, (const TT_*)x, nullptr, (TT_*)y, n / 8)
  if (dtype == DT_F16) {
    if (kind == 0) L(f16, 2); else L(f16, 3);
  } else {
    if (kind == 0) L(bf16, 2); else L(bf16, 3);
  }
#undef L
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_embed(int dtype, const int* ids, const void* tok, const void* pos, void* out, int64_t rows, int L, int d,
                 int vocab, hipStream_t stream) {
  SMI_CHECK(d % 8 == 0, "embed: d %% 8");
  const int grid = ew_grid(rows * (d / 8));
  if (dtype == DT_F16)
    hipLaunchKernelGGL(embed_kernel<f16>, dim3(grid), dim3(256), 0, stream, ids, (const f16*)tok, (const f16*)pos, (f16*)out, rows, L, d / 8, vocab);
  else
    hipLaunchKernelGGL(embed_kernel<bf16>, dim3(grid), dim3(256), 0, stream, ids, (const bf16*)tok, (const bf16*)pos, (bf16*)out, rows, L, d / 8, vocab);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_gather_rows(int dtype, const void* src, const int* idx, void* out, int n, int L, int d, hipStream_t stream) {
  SMI_CHECK(d % 8 == 0, "gather_rows: d %% 8");
  const int grid = ew_grid((int64_t)n * (d / 8));
  if (dtype == DT_F16)
    hipLaunchKernelGGL(gather_rows_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)src, idx, (f16*)out, n, L, d / 8);
  else
    hipLaunchKernelGGL(gather_rows_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)src, idx, (bf16*)out, n, L, d / 8);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_add(int dtype, const void* a, const void* b, void* y, int64_t n, hipStream_t stream) {
  SMI_CHECK(n % 8 == 0, "add: n %% 8");
  const int grid = ew_grid(n / 8);
  if (dtype == DT_F16)
    hipLaunchKernelGGL((ew_kernel<f16, 1>), dim3(grid), dim3(256), 0, stream, (const f16*)a, (const f16*)b, (f16*)y, n / 8);
  else
    hipLaunchKernelGGL((ew_kernel<bf16, 1>), dim3(grid), dim3(256), 0, stream, (const bf16*)a, (const bf16*)b, (bf16*)y, n / 8);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_copy_cols(int dtype, const void* src, int64_t lds, void* dst, int64_t ldd, int col0, int M, int C,
                     hipStream_t stream) {
  (void)dtype;
  SMI_CHECK(C % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && col0 % 8 == 0, "copy_cols: 16-byte alignment");
  const int grid = ew_grid((int64_t)M * C / 8);
  hipLaunchKernelGGL(copy_cols_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)src, lds, (f16*)dst, ldd, col0, (int64_t)M, C);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_nchw_to_nhwc(int dtype, const void* src, int src_f32, void* dst, int Nb, int C, int HW, int Cpad,
                        float scale, hipStream_t stream) {
  const int grid = ew_grid((int64_t)Nb * HW * Cpad);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16>, dim3(grid), dim3(256), 0, stream, src, src_f32, (f16*)dst, Nb, C, HW, Cpad, scale, (const float*)nullptr);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(grid), dim3(256), 0, stream, src, src_f32, (bf16*)dst, Nb, C, HW, Cpad, scale, (const float*)nullptr);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_nchw_to_nhwc_scaled(int dtype, const float* src, void* dst, int Nb, int C, int HW, int Cpad,
                               const float* scale_dev, hipStream_t stream) {
  const int grid = ew_grid((int64_t)Nb * HW * Cpad);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const void*)src, 1, (f16*)dst, Nb, C, HW, Cpad, 1.f, scale_dev);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const void*)src, 1, (bf16*)dst, Nb, C, HW, Cpad, 1.f, scale_dev);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_nhwc_to_nchw_f32(const float* src, float* dst, int Nb, int C, int HW, hipStream_t stream) {
  hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(ew_grid((int64_t)Nb * C * HW)), dim3(256), 0, stream, src, dst, Nb, C, HW);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_timestep_embed(int dtype, const float* vals, void* out, int n, int dim, hipStream_t stream) {
  SMI_CHECK(dim % 2 == 0, "timestep_embed: odd dim");
  const int grid = ew_grid((int64_t)n * dim);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(timestep_embed_kernel<f16>, dim3(grid), dim3(256), 0, stream, vals, (f16*)out, n, dim);
  else
    hipLaunchKernelGGL(timestep_embed_kernel<bf16>, dim3(grid), dim3(256), 0, stream, vals, (bf16*)out, n, dim);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_pool2x2_sum(int dtype, const void* du, void* dx, int Nb, int H, int W, int C, hipStream_t stream) {
  SMI_CHECK(C % 8 == 0, "pool2x2: C %% 8");
  const int grid = ew_grid((int64_t)Nb * H * W * C / 8);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(pool2x2_sum_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)du, (f16*)dx, Nb, H, W, C);
  else
    hipLaunchKernelGGL(pool2x2_sum_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)du, (bf16*)dx, Nb, H, W, C);
  SMI_HIP(hipGetLastError());
  return 0;
}
// scale_out[j] = scale of sample j, scale_out[inv_off + j] = its inverse
int launch_grad_scale(const float* d_eps, int n_samples, int64_t per_sample, float* scale_out, int inv_off,
                      hipStream_t stream) {
  static const float target = [] {
    const char* v = getenv("SMI_GRAD_TARGET_LOG2");
    return exp2f(v ? (float)atoi(v) : 4.f);
  }();
  hipLaunchKernelGGL(grad_scale_kernel, dim3(n_samples), dim3(256), 0, stream, d_eps, per_sample, scale_out, inv_off,
                     target);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_cfg_combine(const float* eps2, float* out, int64_t n_half, float g, hipStream_t stream) {
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(ew_grid(n_half)), dim3(256), 0, stream, eps2, out, n_half, g);
  SMI_HIP(hipGetLastError());
  return 0;
}
// scratch: >= 256 floats
int launch_slider_loss(const float* target, const float* positive, const float* neutral, const float* negative,
                       float sign_eta, int64_t n, float* loss_out, float* dtarget, float* scratch,
                       hipStream_t stream) {
  const int np = ew_grid(n) > 256 ? 256 : ew_grid(n);
  hipLaunchKernelGGL(slider_loss_partial_kernel, dim3(np), dim3(256), 0, stream, target, positive, neutral, negative,
                     sign_eta, n, dtarget, scratch);
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(64), 0, stream, scratch, np, 1.f / (float)n, loss_out);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_axpby(float* y, const float* x, float a, float b, int64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, y, x, a, b, n);
  SMI_HIP(hipGetLastError());
  return 0;
}
// scratch: >= 1 + 1024 floats ([0] = sum g^2, [1..] block partials)
int launch_clip_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, float max_norm, float* scratch, hipStream_t stream) {
  SMI_CHECK(step >= 1, "adamw: step must start at 1");
  if (max_norm > 0.f) {
    const int np = ew_grid(n) > 1024 ? 1024 : ew_grid(n);
    hipLaunchKernelGGL(reduce_partial_kernel<1>, dim3(np), dim3(256), 0, stream, g, n, scratch + 1);
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(64), 0, stream, scratch + 1, np, 1.f, scratch);
  }
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(clip_adamw_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, sqrtf(bc2), max_norm, scratch);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_sched_affine(float* x, const float* eps, const float* noise, float c_x, float c_eps, float c_noise,
                        int64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(sched_affine_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, x, eps, noise, c_x, c_eps, c_noise, n);
  SMI_HIP(hipGetLastError());
  return 0;
}

int launch_softmax_rows(int dtype, const float* S, void* P, int rows, int cols, float scale, hipStream_t stream) {
  if (dtype == DT_F16)
    hipLaunchKernelGGL(softmax_rows_kernel<f16>, dim3(rows), dim3(256), 0, stream, S, (f16*)P, cols, scale);
  else
    hipLaunchKernelGGL(softmax_rows_kernel<bf16>, dim3(rows), dim3(256), 0, stream, S, (bf16*)P, cols, scale);
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_chan_mix(int dtype, const float* x, const void* W, const void* b, float* y, int64_t M, int C,
                    hipStream_t stream) {
  const int grid = ew_grid(M * C);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(chan_mix_kernel<f16>, dim3(grid), dim3(256), 0, stream, x, (const f16*)W, (const f16*)b, y, M, C);
  else
    hipLaunchKernelGGL(chan_mix_kernel<bf16>, dim3(grid), dim3(256), 0, stream, x, (const bf16*)W, (const bf16*)b, y, M, C);
  SMI_HIP(hipGetLastError());
  return 0;
}
size_t colsum_scratch_floats(int Nb, int C) { return (size_t)Nb * COLSUM_RS * C; }
int launch_colsum(int dtype, const void* x, void* out, float* scratch, int Nb, int HW, int C, float mul,
                  hipStream_t stream) {
  SMI_CHECK(C % 8 == 0 && scratch, "colsum: C %% 8 != 0 or no scratch");
  dim3 grid(cdiv(C, 256), COLSUM_RS, Nb);
  const int fb = cdiv((int64_t)Nb * C, 256);
  if (dtype == DT_F16) {
    hipLaunchKernelGGL(colsum_partial_kernel<f16>, grid, dim3(256), 0, stream, (const f16*)x, scratch, HW, C);
    hipLaunchKernelGGL(colsum_final_kernel<f16>, dim3(fb), dim3(256), 0, stream, scratch, (f16*)out, Nb, C, mul);
  } else {
    hipLaunchKernelGGL(colsum_partial_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)x, scratch, HW, C);
    hipLaunchKernelGGL(colsum_final_kernel<bf16>, dim3(fb), dim3(256), 0, stream, scratch, (bf16*)out, Nb, C, mul);
  }
  SMI_HIP(hipGetLastError());
  return 0;
}
int launch_f32_to_padded(int dtype, const float* src, int lds, int cols, void* dst, int cpad, int64_t M, float mul,
                         hipStream_t stream) {
  const int grid = ew_grid(M * cpad);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(f32_to_padded_kernel<f16>, dim3(grid), dim3(256), 0, stream, src, lds, cols, (f16*)dst, cpad, M, mul);
  else
    hipLaunchKernelGGL(f32_to_padded_kernel<bf16>, dim3(grid), dim3(256), 0, stream, src, lds, cols, (bf16*)dst, cpad, M, mul);
  SMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace smi
