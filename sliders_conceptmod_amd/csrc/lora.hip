// Rank-r (r <= 32) LoRA kernels: the two skinny products that cannot ride inside the big MFMA GEMM.
//
//  lora_down : xa[M, r] = X[M, K] * A[r, K]^T          one wave per row, K split over the lanes in 16-byte
//              vectors, r accumulators per lane, butterfly (wavefront shuffle) reduction.  Also used for
//              dXA = dY * up (A given with strides).  HBM-bound: reads X once.
//  lora_wgrad: dW[r, K] += alpha * P[M, r]^T * X[M, K]   weight-gradient reduction over the 4*N tokens:
//              each thread owns one 16-byte column vector of X and r x 8 f32 accumulators, workgroups split M;
//              per-workgroup partials are combined through LDS and a second fixed-order pass (deterministic).
// The up-projection delta itself (xa * up^T) is fused into the GEMM epilogue (gemm.hip).
#include "kernels.h"

namespace smi {
namespace {

template <typename T, int R>
__global__ __launch_bounds__(256) void lora_down_kernel(const T* __restrict__ X, int64_t ldx,
                                                        const float* __restrict__ A, int64_t sa_r, int64_t sa_k,
                                                        float* __restrict__ xa, int64_t ld_xa, int M, int K,
                                                        int r) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float acc[R];
#pragma unroll
  for (int q = 0; q < R; ++q) acc[q] = 0.f;
  const int nvec = K / 8;
  for (int v = lane; v < nvec; v += 64) {
    Pack8<T> xv;
    xv.u = *reinterpret_cast<const u32x4*>(X + (int64_t)row * ldx + v * 8);
    float xf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xf[e] = to_f(xv.e[e]);
#pragma unroll
    for (int q = 0; q < R; ++q) {
      if (q < r) {
        const float* a = A + q * sa_r + (int64_t)v * 8 * sa_k;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[q] += xf[e] * a[e * sa_k];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const float s = wave_sum(acc[q]);
    if (lane == 0 && q < r) xa[(int64_t)row * ld_xa + q] = s;
  }
}

// rows of M per workgroup in the weight-gradient reduction: small enough that even M = 4096 fills the chip
__host__ __device__ inline int wg_rows(int M) { return M >= 8192 ? 32 : 16; }

template <typename T, int R>
__global__ __launch_bounds__(256) void lora_wgrad_partial_kernel(const float* __restrict__ P, int64_t ldp,
                                                                 const T* __restrict__ X, int64_t ldx,
                                                                 float* __restrict__ partial, int M, int K, int r) {
  extern __shared__ float red[];  // [rpar * ncb][8]
  const int cols8 = K / 8;
  const int colblk = blockIdx.y;  // blocks of up to 256 column vectors
  const int ncb = min(256, cols8 - colblk * 256);
  const int rpar = 256 / ncb >= 1 ? 256 / ncb : 1;
  const int tid = threadIdx.x;
  const int rsub = tid / ncb;
  const int cl = tid - rsub * ncb;
  const bool active = rsub < rpar;
  const int col = colblk * 256 + cl;
  const int rows = wg_rows(M);
  const int row0 = blockIdx.x * rows;
  const int row1 = min(M, row0 + rows);
  float acc[R][8];
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
  if (active) {
    // 4 independent row loads in flight per thread (the loop is otherwise a chain of dependent HBM latencies)
    for (int m0 = row0 + rsub; m0 < row1; m0 += 4 * rpar) {
      Pack8<T> xv[4];
      float pv[4][R];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int m = m0 + u * rpar;
        const bool ok = m < row1;
        xv[u].u = ok ? *reinterpret_cast<const u32x4*>(X + (int64_t)m * ldx + col * 8) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < R; ++q) pv[u][q] = (ok && q < r) ? P[(int64_t)m * ldp + q] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[q][e] += pv[u][q] * to_f(xv[u].e[e]);
    }
  }
  // combine the rpar row-slices through LDS in a fixed order
#pragma unroll
  for (int q = 0; q < R; ++q) {
    if (q < r) {  // r is block-uniform: the barriers below are reached by every thread or by none
      __syncthreads();
      if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(rsub * ncb + cl) * 8 + e] = acc[q][e];
      }
      __syncthreads();
      if (tid < ncb) {
        float* out = partial + ((int64_t)blockIdx.x * r + q) * K + (colblk * 256 + tid) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float s = 0.f;
          for (int rs = 0; rs < rpar; ++rs) s += red[(rs * ncb + tid) * 8 + e];
          out[e] = s;
        }
      }
    }
  }
}

// dW[q*so_r + k*so_k] += alpha * alpha_dev * sum_s partial[s][q][k]   (fixed summation order: deterministic)
// block = 32 consecutive outputs x 8 split-groups; group g sums splits g, g+8, ...; LDS combines the 8 groups.
__global__ __launch_bounds__(256) void lora_wgrad_final_kernel(const float* __restrict__ partial, int nsplit,
                                                               float* __restrict__ dW, int64_t so_r, int64_t so_k,
                                                               int K, int r, float alpha,
                                                               const float* __restrict__ alpha_dev) {
  __shared__ float red[8][33];
  const float a = alpha * (alpha_dev ? alpha_dev[0] : 1.f);
  const int64_t total = (int64_t)r * K;
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + e;
  float s = 0.f;
  if (i < total) {
    for (int sp = g; sp < nsplit; sp += 8) s += partial[(int64_t)sp * total + i];
  }
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && i < total) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[j][e];
    const int q = (int)(i / K);
    const int k = (int)(i - (int64_t)q * K);
    dW[q * so_r + k * so_k] += a * t;
  }
}

// fp32 flat LoRA parameters -> 16-bit GEMM operands, once per forward, for every adapted (possibly fused) GEMM:
//   downT[rows_pad, K]        = lora_down rows of the site (zero rows up to rows_pad)
//   upT  [rows_pad, nseg*cs]  block-diagonal: upT[s*r + q][s*cs + n] = lora_up_s[n][q]
struct LoraPrepSite {
  int64_t off_down, off_up;  // into the flat fp32 buffers
  int64_t dst_down, dst_up;  // element offsets into the 16-bit shadow buffer
  int r, nseg, K, cs, rows_pad;
};
template <typename T>
__global__ void lora_prep_kernel(const LoraPrepSite* __restrict__ sites, const float* __restrict__ down,
                                 const float* __restrict__ up, T* __restrict__ shadow) {
  const LoraPrepSite st = sites[blockIdx.y];
  const int rtot = st.r * st.nseg;
  const int64_t n_down = (int64_t)st.rows_pad * st.K;
  const int64_t n_up = (int64_t)st.rows_pad * st.nseg * st.cs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_down + n_up;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n_down) {
      const int row = (int)(i / st.K);
      shadow[st.dst_down + i] = from_f<T>(row < rtot ? down[st.off_down + i] : 0.f);
    } else {
      const int64_t j = i - n_down;
      const int ncols = st.nseg * st.cs;
      const int row = (int)(j / ncols), col = (int)(j - (int64_t)row * ncols);
      float v = 0.f;
      if (row < rtot) {
        const int sgm = row / st.r, q = row - sgm * st.r;
        if (col / st.cs == sgm) v = up[st.off_up + ((int64_t)sgm * st.cs + (col - sgm * st.cs)) * st.r + q];
      }
      shadow[st.dst_up + j] = from_f<T>(v);
    }
  }
}

template <typename T>
int down_t(const void* X, int64_t ldx, const float* A, int64_t sr, int64_t sk, float* xa, int64_t ld_xa, int M, int K,
           int r, hipStream_t st) {
  const int grid = cdiv(M, 4);
#define L(RR) hipLaunchKernelGGL((lora_down_kernel<T, RR>), dim3(grid), dim3(256), 0, st, (const T*)X, ldx, A, sr, sk, xa, ld_xa, M, K, r)
  if (r <= 4) L(4);
  else if (r <= 8) L(8);
  else if (r <= 16) L(16);
  else L(32);
#undef L
  SMI_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int wgrad_t(const float* P, int64_t ldp, const void* X, int64_t ldx, float* dW, int64_t so_r, int64_t so_k, int M, int K, int r,
            float alpha, const float* alpha_dev, float* scratch, hipStream_t st) {
  const int nsplit = cdiv(M, wg_rows(M));
  const int cols8 = K / 8;
  dim3 grid(nsplit, cdiv(cols8, 256));
  const size_t sm = 256 * 8 * sizeof(float);
#define L(RR) hipLaunchKernelGGL((lora_wgrad_partial_kernel<T, RR>), grid, dim3(256), sm, st, P, ldp, (const T*)X, ldx, scratch, M, K, r)
  if (r <= 4) L(4);
  else if (r <= 8) L(8);
  else if (r <= 16) L(16);
  else L(32);
#undef L
  const int64_t total = (int64_t)r * K;
  hipLaunchKernelGGL(lora_wgrad_final_kernel, dim3((int)((total + 31) / 32)), dim3(256), 0, st, scratch, nsplit, dW,
                     so_r, so_k, K, r, alpha, alpha_dev);
  SMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_lora_down(int dtype, const void* X, int64_t ldx, const float* A, int64_t lda_r, int64_t lda_k, float* xa,
                     int64_t ld_xa, int M, int K, int r, hipStream_t stream) {
  SMI_CHECK(r >= 1 && r <= 32 && K % 8 == 0 && ldx % 8 == 0, "lora_down: r=%d K=%d ldx=%lld", r, K, (long long)ldx);
  return dtype == DT_F16 ? down_t<f16>(X, ldx, A, lda_r, lda_k, xa, ld_xa, M, K, r, stream)
                         : down_t<bf16>(X, ldx, A, lda_r, lda_k, xa, ld_xa, M, K, r, stream);
}

// scratch: lora_wgrad_scratch_floats(M, K, r) floats
size_t lora_wgrad_scratch_floats(int M, int K, int r) { return (size_t)cdiv(M, wg_rows(M)) * r * K; }

int launch_lora_wgrad(int dtype, const float* P, int64_t ldp, const void* X, int64_t ldx, float* dW, int64_t so_r,
                      int64_t so_k, int M, int K, int r, float alpha, const float* alpha_dev, float* scratch,
                      hipStream_t stream) {
  SMI_CHECK(r >= 1 && r <= 32 && K % 8 == 0 && ldx % 8 == 0, "lora_wgrad: r=%d K=%d", r, K);
  return dtype == DT_F16 ? wgrad_t<f16>(P, ldp, X, ldx, dW, so_r, so_k, M, K, r, alpha, alpha_dev, scratch, stream)
                         : wgrad_t<bf16>(P, ldp, X, ldx, dW, so_r, so_k, M, K, r, alpha, alpha_dev, scratch, stream);
}

int launch_lora_prep(int dtype, const void* sites_dev, int n_sites, const float* down, const float* up, void* shadow,
                     hipStream_t stream) {
  if (n_sites <= 0) return 0;
  dim3 grid(16, n_sites);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(lora_prep_kernel<f16>, grid, dim3(256), 0, stream, (const LoraPrepSite*)sites_dev, down, up, (f16*)shadow);
  else
    hipLaunchKernelGGL(lora_prep_kernel<bf16>, grid, dim3(256), 0, stream, (const LoraPrepSite*)sites_dev, down, up, (bf16*)shadow);
  SMI_HIP(hipGetLastError());
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// Skinny MFMA product for the LoRA shadow operands:  out[M, R] (fp32) = X[M, K] * S[R, K]^T,  R = 16 or 32.
//
// These sit on the critical path of every adapted Linear (xa = x * down^T before the main GEMM, dxa = dy * up before
// the dX GEMM) and are pure latency: 10-40 MB of X read once, a few MFLOP.  Through the tiled GEMM (32 workgroups of
// 128 rows, 20 serial K-steps) a launch took ~28 us.  Here a workgroup owns 16 rows and its 4 waves split K, every
// wave issues ALL of its fragment loads up front (no LDS staging, no K loop dependency), one MFMA chain, and the
// four partial tiles meet in LDS: one memory round trip per launch, M/16 workgroups.
// Operand maps as in the GEMMs: mfma16(S frag, X frag) -> lane (fr = l & 15, fq = l >> 4) holds out[m = fr][n = 4 fq + j].
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int NF, int STEPS>  // NF = R / 16 column fragments, STEPS = 32-deep K-steps per wave and chunk
__global__ __launch_bounds__(256) void lora_skinny_kernel(const T* __restrict__ X, int64_t ldx, const T* __restrict__ S,
                                                          float* __restrict__ out, int ldo, int M, int K) {
  __shared__ f32x4 red[3][NF][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int m = min(m0 + fr, M - 1);           // rows past M are clamped; their results are not stored
  const int kw = K >> 2;                       // K range of this wave: [wave * kw, (wave + 1) * kw)
  const T* xp = X + (int64_t)m * ldx + wave * kw + fq * 8;
  const T* sp = S + (int64_t)fr * K + wave * kw + fq * 8;
  f32x4 acc[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < kw; k0 += 32 * STEPS) {
    typename TT<T>::v8 xa[STEPS], sb[STEPS][NF];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
      const bool ok = k0 + 32 * i < kw;  // wave-uniform
      Pack8<T> t;
      t.u = ok ? *reinterpret_cast<const u32x4*>(xp + k0 + 32 * i) : u32x4{0u, 0u, 0u, 0u};
      xa[i] = t.v;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        Pack8<T> w;
        w.u = ok ? *reinterpret_cast<const u32x4*>(sp + (int64_t)f * 16 * K + k0 + 32 * i) : u32x4{0u, 0u, 0u, 0u};
        sb[i][f] = w.v;
      }
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i)
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[f] = TT<T>::mfma16(sb[i][f], xa[i], acc[f]);
  }
  if (wave > 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f) red[wave - 1][f][lane] = acc[f];
  }
  __syncthreads();
  if (wave == 0 && m0 + fr < M) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      f32x4 v = acc[f];
#pragma unroll
      for (int w = 0; w < 3; ++w) {  // fixed order: deterministic
        const f32x4 o = red[w][f][lane];
        v[0] += o[0];
        v[1] += o[1];
        v[2] += o[2];
        v[3] += o[3];
      }
      *reinterpret_cast<f32x4*>(out + (int64_t)(m0 + fr) * ldo + f * 16 + fq * 4) = v;
    }
  }
}

bool lora_skinny_supported(const void* X, int64_t ldx, const void* S, const float* out, int ldo, int M, int R, int K) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return (R == 16 || R == 32) && K % 128 == 0 && ldx % 8 == 0 && ldo % 4 == 0 && ldo >= R && M > 0 && al16(X) &&
         al16(S) && al16(out);
}

int launch_lora_skinny(int dtype, const void* X, int64_t ldx, const void* S, float* out, int ldo, int M, int R, int K,
                       hipStream_t stream) {
  SMI_CHECK(lora_skinny_supported(X, ldx, S, out, ldo, M, R, K), "lora_skinny: unsupported layout (R=%d K=%d)", R, K);
  const int grid = cdiv(M, 16);
#define GO(TT_, NF_)                                                                                          \
  hipLaunchKernelGGL((lora_skinny_kernel<TT_, NF_, 8>), dim3(grid), dim3(256), 0, stream, (const TT_*)X, ldx, \
                     (const TT_*)S, out, ldo, M, K)
  if (dtype == DT_F16) {
    if (R == 16) GO(f16, 1); else GO(f16, 2);
  } else {
    if (R == 16) GO(bf16, 1); else GO(bf16, 2);
  }
#undef GO
  SMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace smi
