// Rank-r (r <= 32) LoRA kernels: the two skinny products that cannot ride inside the big MFMA GEMM.
//
//  lora_down : xa[M, r] = X[M, K] * A[r, K]^T          one wave per row, K split over the lanes in 16-byte
//              vectors, r accumulators per lane, butterfly (wavefront shuffle) reduction.  Also used for
//              dXA = dY * up (A given with strides).  HBM-bound: reads X once.
//  lora_wgrad: dW[r, K] += alpha * P[M, r]^T * X[M, K]   weight-gradient reduction over the 4*N tokens, GROUPED: all
//              reductions of a backward pass in one launch from a job table (see "Grouped weight-gradient reduction").
// The up-projection delta itself (xa * up^T) is fused into the GEMM epilogue (gemm.hip).
#include <stdlib.h>

#include "kernels.h"

#include <algorithm>
#include <vector>

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

// fp32 flat LoRA parameters -> 16-bit GEMM operands, once per forward, for every adapted (possibly fused) GEMM:
//   downT[rows_pad, K]        = lora_down rows of the site (zero rows up to rows_pad)
//   upT  [rows_pad, nseg*cs]  block-diagonal: upT[s*r + q][s*cs + n] = lora_up_s[n][q]
struct LoraPrepSite {
  int64_t off_down, off_up;  // into the flat fp32 buffers
  int64_t dst_down, dst_up;  // element offsets into the 16-bit shadow buffer
  int r, nseg, K, cs, rows_pad;
  int conv;                  // see HostLoraPrepSite
  int64_t dst_gw;
};
template <typename T>
__global__ void lora_prep_kernel(const LoraPrepSite* __restrict__ sites, const float* __restrict__ down,
                                 const float* __restrict__ up, T* __restrict__ shadow) {
  const LoraPrepSite st = sites[blockIdx.y];
  const int rtot = st.r * st.nseg;
  const int64_t kd = st.conv ? 9 * (int64_t)st.K : st.K;  // row length of the down operand
  const int64_t n_down = (int64_t)st.rows_pad * kd;
  const int64_t n_up = (int64_t)st.rows_pad * st.nseg * st.cs;
  const int64_t n_gw = st.conv ? (int64_t)st.K * 9 * 64 : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_down + n_up + n_gw;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n_down) {
      const int row = (int)(i / kd);
      float v = 0.f;
      if (row < rtot) {
        if (st.conv) {  // torch [r][Cin][3][3] -> (ky,kx,ci)-ordered row
          const int rem = (int)(i - (int64_t)row * kd);
          const int t = rem / st.K, ci = rem - t * st.K;
          v = down[st.off_down + ((int64_t)row * st.K + ci) * 9 + t];
        } else {
          v = down[st.off_down + i];
        }
      }
      shadow[st.dst_down + i] = from_f<T>(v);
    } else if (i < n_down + n_up) {
      const int64_t j = i - n_down;
      const int ncols = st.nseg * st.cs;
      const int row = (int)(j / ncols), col = (int)(j - (int64_t)row * ncols);
      float v = 0.f;
      if (row < rtot) {
        const int sgm = row / st.r, q = row - sgm * st.r;
        if (col / st.cs == sgm) v = up[st.off_up + ((int64_t)sgm * st.cs + (col - sgm * st.cs)) * st.r + q];
      }
      shadow[st.dst_up + j] = from_f<T>(v);
    } else {  // gradient filter of the down conv: gw[ci][tt][q] = down[q][ci][t], tt = 8 - t (flipped) or t (stride 2)
      const int64_t j = i - n_down - n_up;
      const int ci = (int)(j / 576), rem = (int)(j - (int64_t)ci * 576);
      const int tt = rem >> 6, q = rem & 63;
      const int t = st.conv == 1 ? 8 - tt : tt;
      shadow[st.dst_gw + j] = from_f<T>(q < st.r ? down[st.off_down + ((int64_t)q * st.K + ci) * 9 + t] : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Grouped weight-gradient reduction: EVERY rank-r reduction of a backward pass in one launch (per accumulator class).
//
// The engine does not launch a reduction per site any more (that was ~700 latency-bound launches per step): each
// adapted Linear appends jobs to a table while the backward runs -- d(up) = alpha * dY^T xa and d(down) = alpha *
// dxa^T x, for all segments of a fused projection at once -- and, because the arenas never recycle memory inside a
// backward, all operands are still there when the dX chain has finished.  Then two launches do everything:
//   partial: workgroup = (job, M-split, column block); thread = one 16-byte column vector x one of `rpar` row slices,
//            R x 8 fp32 accumulators; the row slices meet in LDS in a fixed order; partial[split][q][k] is written.
//   final  : fixed-order sum over the splits, dW (+)= alpha * sum.            No atomics: bit-reproducible.
// P rows are multiplied by row_scale[m / rows_per_sample] on load (the per-sample loss scale divided out in fp32).
// Fused projections: column c belongs to segment c / seg_cols and pairs with P columns [seg * r, seg * r + r).
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int R>
__global__ __launch_bounds__(256) void lora_wgrad_grouped_partial_kernel(const WgradJob* __restrict__ jobs, int njobs) {
  __shared__ float red[256 * 8];
  // the workgroup's rows of P (x row scale), staged once: [row][all segments' r columns].  Sized by the rank class: the
  // rank <= 4 class (three fused segments = 12 columns at most) keeps 20 KB of LDS per workgroup instead of 40 KB, i.e.
  // 7 instead of 3 workgroups per CU in flight for a kernel that only waits on HBM
  constexpr int PSW = R <= 4 ? 12 : 32;
  __shared__ float ps[256 * PSW];
  __shared__ int jsel;
  if (threadIdx.x == 0) {  // last job whose first workgroup is <= blockIdx.x (jobs of one class are contiguous)
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].wg0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    jsel = lo;
  }
  __syncthreads();
  const WgradJob jb = jobs[jsel];
  const int local = blockIdx.x - jb.wg0;
  const int split = local / jb.ncolblk, colblk = local - split * jb.ncolblk;
  const int cw = jb.cw;                 // column vectors per workgroup (power of two <= 256)
  const int rpar = 256 / cw;            // row slices
  const int tid = threadIdx.x;
  const int rsub = tid / cw, cl = tid - rsub * cw;
  const int col8 = colblk * cw + cl;    // this thread's 8-column vector
  const bool active = col8 * 8 < jb.K;
  const int row0 = split * jb.rows_per_wg;
  const int row1 = min(jb.M, row0 + jb.rows_per_wg);
  const int r = jb.r;
  const T* X = reinterpret_cast<const T*>(jb.X);
  const int segoff = jb.seg_cols ? ((col8 * 8) / jb.seg_cols) * r : 0;
  const float* P = jb.P + segoff;
  // P is tiny (r floats per row) but was 17 load instructions per row and thread against ONE for X: stage the rows of
  // this workgroup in LDS (all segments' columns, row scale folded in) and read them back as LDS broadcasts
  const int pw = jb.seg_cols ? (jb.K / jb.seg_cols) * r : r;
  const bool staged = pw <= PSW && jb.rows_per_wg <= 256;
  if (staged) {
    const int nrow = row1 - row0;
    for (int i = tid; i < nrow * pw; i += 256) {
      const int rr = i / pw, q = i - rr * pw;
      const int m = row0 + rr;
      const float rs = jb.row_scale ? jb.row_scale[m / jb.rows_per_sample] : 1.f;
      ps[i] = jb.P[(int64_t)m * jb.ldp + q] * rs;
    }
    __syncthreads();
  }
  float acc[R][8];
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
  if (active) {
    for (int m0 = row0 + rsub; m0 < row1; m0 += 4 * rpar) {
      Pack8<T> xv[4];
      float pv[4][R];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int m = m0 + u * rpar;
        bool ok = m < row1;
        int64_t xrow = m;
        if (jb.conv_tap >= 0 && ok) {  // output pixel m -> input pixel under this filter tap (3x3, pad 1)
          const int hw = jb.Hout * jb.Wout;
          const int img = m / hw, rem = m - img * hw;
          const int oy = rem / jb.Wout, ox = rem - oy * jb.Wout;
          int iy = oy * jb.conv_stride + jb.conv_tap / 3 - 1, ix = ox * jb.conv_stride + jb.conv_tap % 3 - 1;
          const int hl = jb.conv_ups ? 2 * jb.Hin : jb.Hin, wl = jb.conv_ups ? 2 * jb.Win : jb.Win;
          const bool inside = (unsigned)iy < (unsigned)hl && (unsigned)ix < (unsigned)wl;
          if (jb.conv_ups) {
            iy >>= 1;
            ix >>= 1;
          }
          xrow = ((int64_t)img * jb.Hin + iy) * jb.Win + ix;
          if (!inside) xrow = -1;
        }
        const bool ld = ok && xrow >= 0;
        xv[u].u = ld ? *reinterpret_cast<const u32x4*>(X + xrow * jb.ldx + col8 * 8) : u32x4{0, 0, 0, 0};
        if (staged) {
          const float* pr = ps + (ok ? (m - row0) * pw + segoff : 0);
#pragma unroll
          for (int q = 0; q < R; ++q) pv[u][q] = (ok && q < r) ? pr[q] : 0.f;
        } else {
          const float rs = (ok && jb.row_scale) ? jb.row_scale[m / jb.rows_per_sample] : 1.f;
#pragma unroll
          for (int q = 0; q < R; ++q) pv[u][q] = (ok && q < r) ? P[(int64_t)m * jb.ldp + q] * rs : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[q][e] += pv[u][q] * to_f(xv[u].e[e]);
    }
  }
#pragma unroll
  for (int q = 0; q < R; ++q) {
    if (q < r) {  // r is block-uniform: the barriers below are reached by every thread or by none
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(rsub * cw + cl) * 8 + e] = acc[q][e];
      __syncthreads();
      if (tid < cw && (colblk * cw + tid) * 8 < jb.K) {
        float* out = jb.partial + ((int64_t)split * r + q) * jb.K + (colblk * cw + tid) * 8;
        float sum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] = 0.f;
        for (int rs = 0; rs < rpar; ++rs)
#pragma unroll
          for (int e = 0; e < 8; ++e) sum[e] += red[(rs * cw + tid) * 8 + e];
        *reinterpret_cast<f32x4*>(out) = f32x4{sum[0], sum[1], sum[2], sum[3]};
        *reinterpret_cast<f32x4*>(out + 4) = f32x4{sum[4], sum[5], sum[6], sum[7]};
      }
    }
  }
}

// block = 32 consecutive outputs x 8 split-groups (group g sums splits g, g+8, ... in order; LDS combines the groups
// in order): deterministic.  dW[q*so_r + k*so_k] += alpha * sum.
__global__ __launch_bounds__(256) void lora_wgrad_grouped_final_kernel(const WgradJob* __restrict__ jobs, int njobs) {
  __shared__ float red[8][33];
  __shared__ int jsel;
  if (threadIdx.x == 0) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].fb0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    jsel = lo;
  }
  __syncthreads();
  const WgradJob jb = jobs[jsel];
  const int64_t total = (int64_t)jb.r * jb.K;
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t i = (int64_t)(blockIdx.x - jb.fb0) * 32 + e;
  float s = 0.f;
  if (i < total) {
    for (int sp = g; sp < jb.nsplit; sp += 8) s += jb.partial[(int64_t)sp * total + i];
  }
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && i < total) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[j][e];
    const int q = (int)(i / jb.K);
    const int k = (int)(i - (int64_t)q * jb.K);
    jb.dW[q * jb.so_r + k * jb.so_k] += jb.alpha * t;
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

}  // namespace

int launch_lora_down(int dtype, const void* X, int64_t ldx, const float* A, int64_t lda_r, int64_t lda_k, float* xa,
                     int64_t ld_xa, int M, int K, int r, hipStream_t stream) {
  SMI_CHECK(r >= 1 && r <= 32 && K % 8 == 0 && ldx % 8 == 0, "lora_down: r=%d K=%d ldx=%lld", r, K, (long long)ldx);
  return dtype == DT_F16 ? down_t<f16>(X, ldx, A, lda_r, lda_k, xa, ld_xa, M, K, r, stream)
                         : down_t<bf16>(X, ldx, A, lda_r, lda_k, xa, ld_xa, M, K, r, stream);
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
                                                          float* __restrict__ out, int ldo, int M, int K,
                                                          const float* __restrict__ row_mul, int rows_per_mul) {
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
      if (row_mul) {  // per-sample adaptor multipliers (image sliders: +s / -s in one pass): out row *= m[sample] / m_ref
        const float rm = row_mul[(m0 + fr) / rows_per_mul];
        v[0] *= rm;
        v[1] *= rm;
        v[2] *= rm;
        v[3] *= rm;
      }
      *reinterpret_cast<f32x4*>(out + (int64_t)(m0 + fr) * ldo + f * 16 + fq * 4) = v;
    }
  }
}

// out[m, 0..N) *= row_mul[m / rows_per_mul]: the per-sample multipliers for shapes the skinny kernel does not take
__global__ void row_scale_f32_kernel(float* __restrict__ x, int ld, int M, int N, const float* __restrict__ row_mul,
                                     int rows_per_mul) {
  const int64_t total = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
    x[(int64_t)m * ld + n] *= row_mul[m / rows_per_mul];
  }
}
int launch_row_scale_f32(float* x, int ld, int M, int N, const float* row_mul, int rows_per_mul, hipStream_t stream) {
  SMI_CHECK(x && row_mul && rows_per_mul > 0 && M > 0 && N > 0, "row_scale: bad arguments");
  const int64_t total = (int64_t)M * N;
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(row_scale_f32_kernel, dim3(grid), dim3(256), 0, stream, x, ld, M, N, row_mul, rows_per_mul);
  SMI_HIP(hipGetLastError());
  return 0;
}

bool lora_skinny_supported(const void* X, int64_t ldx, const void* S, const float* out, int ldo, int M, int R, int K) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return (R == 16 || R == 32) && K % 128 == 0 && ldx % 8 == 0 && ldo % 4 == 0 && ldo >= R && M > 0 && al16(X) &&
         al16(S) && al16(out);
}

int launch_lora_skinny(int dtype, const void* X, int64_t ldx, const void* S, float* out, int ldo, int M, int R, int K,
                       hipStream_t stream, const float* row_mul, int rows_per_mul) {
  SMI_CHECK(!row_mul || rows_per_mul > 0, "lora_skinny: rows_per_mul must be positive");
  SMI_CHECK(lora_skinny_supported(X, ldx, S, out, ldo, M, R, K), "lora_skinny: unsupported layout (R=%d K=%d)", R, K);
  const int grid = cdiv(M, 16);
  // STEPS = 32-wide k-steps whose loads a wave issues before its first MFMA.  The grid is one 4-wave workgroup per 16 rows
  // (M = 4096: one per CU), so the launch lasts as many memory round trips as the k loop has iterations: with the UNet's
  // widths (K = 1280 / 640: 10 / 5 steps per wave) ONE iteration holds the whole row slice in flight -- 7.1 -> ~4 us per
  // launch, 280 launches per step.  Same MFMA chain per wave whatever the grouping: results do not change.
  const int ksteps = (K >> 2) / 32;
#define GO3(TT_, NF_, ST_)                                                                                       \
  hipLaunchKernelGGL((lora_skinny_kernel<TT_, NF_, ST_>), dim3(grid), dim3(256), 0, stream, (const TT_*)X, ldx, \
                     (const TT_*)S, out, ldo, M, K, row_mul, rows_per_mul)
#define GO(TT_, NF_)                         \
  do {                                       \
    if (ksteps == 10) GO3(TT_, NF_, 10);     \
    else if (ksteps == 5) GO3(TT_, NF_, 5);  \
    else GO3(TT_, NF_, 8);                   \
  } while (0)
  if (dtype == DT_F16) {
    if (R == 16) GO(f16, 1); else GO(f16, 2);
  } else {
    if (R == 16) GO(bf16, 1); else GO(bf16, 2);
  }
#undef GO
#undef GO3
  SMI_HIP(hipGetLastError());
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// DoRA kernels.  V = W + up down is never stored: r <= 32 multiply-adds per element rebuild it where it is needed.
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr int DORA_RMAX = 32;
constexpr int DORA_OS = 64;  // row slices of the column-gradient reduction

// grid (ceil(K / 256), nseg, n_sites); a lane owns FOUR adjacent columns (8-byte loads: a wave reads 512 contiguous bytes
// of a row instead of 128), the 4 waves split the rows of a segment; fixed-order LDS combine
template <typename T, int RC>  // RC: rank class (>= every site's rank): rank loops unroll, arrays stay in registers
__global__ __launch_bounds__(256) void dora_norm_kernel(const DoraSite* __restrict__ sites, const float* __restrict__ down,
                                                        const float* __restrict__ up) {
  __shared__ float red[4][4][64];
  const DoraSite st = sites[blockIdx.z];
  const int sgm = blockIdx.y;
  if (sgm >= st.nseg || (int)blockIdx.x * 256 >= st.K) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = blockIdx.x * 256 + lane * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (k < st.K) {  // K % 8 == 0: a lane's four columns exist together
    f32x4 a[RC];
#pragma unroll
    for (int q = 0; q < RC; ++q)
      a[q] = q < st.r ? *reinterpret_cast<const f32x4*>(down + st.off_down + ((int64_t)sgm * st.r + q) * st.K + k)
                      : f32x4{0.f, 0.f, 0.f, 0.f};
    const T* W = reinterpret_cast<const T*>(st.W) + (int64_t)sgm * st.cs * st.K + k;
    const float* B = up + st.off_up + (int64_t)sgm * st.cs * st.r;
    for (int o = w; o < st.cs; o += 4) {
      union { u32x2 u; T e[4]; } wv;
      wv.u = *reinterpret_cast<const u32x2*>(W + (int64_t)o * st.K);
      float v[4] = {to_f(wv.e[0]), to_f(wv.e[1]), to_f(wv.e[2]), to_f(wv.e[3])};
#pragma unroll
      for (int q = 0; q < RC; ++q) {
        if (q < st.r) {
          const float b = B[(int64_t)o * st.r + q];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b * a[q][e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += v[e] * v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[w][e][lane] = acc[e];
  __syncthreads();
  if (w == 0 && k < st.K) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      st.cnorm[(int64_t)sgm * st.K + k + e] =
          sqrtf((red[0][e][lane] + red[1][e][lane]) + (red[2][e][lane] + red[3][e][lane]));
  }
}

// every site's dWt[k][o] = dW[o][k] through a 64 x 64 LDS tile; grid (max tiles, n_sites)
template <typename T>
__global__ __launch_bounds__(256) void dora_transpose_kernel(const DoraSite* __restrict__ sites) {
  __shared__ T tile[64][66];
  const DoraSite st = sites[blockIdx.y];
  const int R = st.nseg * st.cs, K = st.K;
  const int tk = (K + 63) / 64, tr = (R + 63) / 64;
  if ((int)blockIdx.x >= tk * tr) return;
  const int r0 = ((int)blockIdx.x / tk) * 64, k0 = ((int)blockIdx.x % tk) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const T* src = reinterpret_cast<const T*>(st.dW);
  T* dst = reinterpret_cast<T*>(st.dWt);
  for (int i = ty; i < 64; i += 4)
    if (r0 + i < R && k0 + tx < K) tile[i][tx] = src[(int64_t)(r0 + i) * K + k0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4)
    if (k0 + i < K && r0 + tx < R) dst[(int64_t)(k0 + i) * R + r0 + tx] = tile[tx][i];
}

// grid (blocks, n_sites): dW[o][k] = lscale * ((W + up down)[o][k] * g[k] / n[k] - W[o][k]), 8 columns per thread
template <typename T>
__global__ __launch_bounds__(256) void dora_delta_kernel(const DoraSite* __restrict__ sites, const float* __restrict__ down,
                                                         const float* __restrict__ up, float mult) {
  const DoraSite st = sites[blockIdx.y];
  const int k8n = st.K / 8;
  const int64_t total = (int64_t)st.nseg * st.cs * k8n;
  const float lscale = mult * st.scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k0 = (int)(i % k8n) * 8;
    const int64_t ot = i / k8n;
    const int sgm = (int)(ot / st.cs);
    Pack8<T> wv, o;
    wv.u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(st.W) + ot * st.K + k0);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = to_f(wv.e[e]);
    const float* B = up + st.off_up + ot * st.r;
    for (int q = 0; q < st.r; ++q) {
      const float b = B[q];
      const float* a = down + st.off_down + ((int64_t)sgm * st.r + q) * st.K + k0;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += b * a[e];
    }
    const float* g = up + st.off_dora + (int64_t)sgm * st.K + k0;
    const float* n = st.cnorm + (int64_t)sgm * st.K + k0;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = from_f<T>(lscale * (v[e] * (g[e] / n[e]) - to_f(wv.e[e])));
    *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(st.dW) + ot * st.K + k0) = o.u;
  }
}

// dst[c][m] = src[m][c] * f[m / rps] through a 64 x 64 LDS tile (coalesced both ways); columns m >= M are zero
template <typename T>
__global__ __launch_bounds__(256) void transpose_scaled_kernel(const T* __restrict__ src, int64_t lds, T* __restrict__ dst,
                                                               int M, int C, int Mp, const float* __restrict__ f, int rps) {
  __shared__ float tile[64][65];
  const int tilesC = (C + 63) / 64;
  const int tm = (blockIdx.x / tilesC) * 64, tc = (blockIdx.x % tilesC) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int m = tm + i, c = tc + tx;
    float v = 0.f;
    if (m < M && c < C) v = to_f(src[(int64_t)m * lds + c]) * (f ? f[m / rps] : 1.f);
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = tc + i, m = tm + tx;
    if (c < C && m < Mp) dst[(int64_t)c * Mp + m] = from_f<T>(tile[tx][i]);
  }
}

// the same with 16-byte accesses on both sides (C % 8 == 0, lds % 8 == 0, 16-byte aligned src; Mp % 64 == 0 always)
template <typename T>
__global__ __launch_bounds__(256) void transpose_scaled_vec_kernel(const T* __restrict__ src, int64_t lds, T* __restrict__ dst,
                                                                   int M, int C, int Mp, const float* __restrict__ f, int rps) {
  __shared__ float tile[64][65];
  const int tilesC = (C + 63) / 64;
  const int tm = (blockIdx.x / tilesC) * 64, tc = (blockIdx.x % tilesC) * 64;
  for (int v = threadIdx.x; v < 512; v += 256) {
    const int i = v >> 3, c0 = (v & 7) * 8;
    const int m = tm + i, c = tc + c0;
    Pack8<T> t;
    t.u = u32x4{0u, 0u, 0u, 0u};
    float sc = 0.f;
    if (m < M && c < C) {
      t.u = *reinterpret_cast<const u32x4*>(src + (int64_t)m * lds + c);
      sc = f ? f[m / rps] : 1.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[i][c0 + e] = to_f(t.e[e]) * sc;
  }
  __syncthreads();
  for (int v = threadIdx.x; v < 512; v += 256) {
    const int ci = v >> 3, m0 = (v & 7) * 8;
    const int c = tc + ci, m = tm + m0;
    if (c < C && m < Mp) {
      Pack8<T> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.e[e] = from_f<T>(tile[m0 + e][ci]);
      *reinterpret_cast<u32x4*>(dst + (int64_t)c * Mp + m) = o.u;
    }
  }
}

// column-wise gradients: d(g)[k] = a * sum_o G V / n;  d(down)[q][k] = a * (g / n) * sum_o up[o][q] G[o][k].
// Two stages, fixed order (no atomics): grid (ceil(K / 256), nseg, DORA_OS) -- slice z sums the rows [z cs / OS, (z + 1) cs / OS)
// of its 256 columns into part[(sgm OS + z)(1 + r) + j][k]; the final kernel adds the slices in order and scales.  (One stage
// with grid (K / 64, nseg) was 20-60 workgroups walking a 6.5-20 MB matrix: 236 us per site, 33 ms per SD-XL step.)
template <typename T, int RC>
__global__ __launch_bounds__(256) void dora_col_grad_kernel(DoraSite st, const float* __restrict__ G,
                                                            const float* __restrict__ down, const float* __restrict__ up,
                                                            float* __restrict__ part) {
  // grid (ceil(K / 256), nseg, DORA_OS): a lane owns four adjacent columns (16-byte loads of G, 8-byte loads of W)
  __shared__ f32x4 red[4][64];
  const int sgm = blockIdx.y, os = blockIdx.z;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = blockIdx.x * 256 + lane * 4;
  const bool ok = k < st.K;  // K % 8 == 0: a lane's four columns exist together
  const int o0 = (int)((int64_t)os * st.cs / DORA_OS), o1 = (int)((int64_t)(os + 1) * st.cs / DORA_OS);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 dg = zero, da[RC], a[RC];
#pragma unroll
  for (int q = 0; q < RC; ++q) da[q] = zero, a[q] = zero;
  if (ok) {
#pragma unroll
    for (int q = 0; q < RC; ++q)
      if (q < st.r) a[q] = *reinterpret_cast<const f32x4*>(down + st.off_down + ((int64_t)sgm * st.r + q) * st.K + k);
    const T* W = reinterpret_cast<const T*>(st.W) + (int64_t)sgm * st.cs * st.K + k;
    const float* B = up + st.off_up + (int64_t)sgm * st.cs * st.r;
    const float* Gs = G + (int64_t)sgm * st.cs * st.K + k;
    for (int o = o0 + w; o < o1; o += 4) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(Gs + (int64_t)o * st.K);
      union { u32x2 u; T e[4]; } wv;
      wv.u = *reinterpret_cast<const u32x2*>(W + (int64_t)o * st.K);
      f32x4 v = {to_f(wv.e[0]), to_f(wv.e[1]), to_f(wv.e[2]), to_f(wv.e[3])};
#pragma unroll
      for (int q = 0; q < RC; ++q) {
        if (q < st.r) {
          const float b = B[(int64_t)o * st.r + q];
          v += b * a[q];
          da[q] += b * gv;
        }
      }
      dg += gv * v;
    }
  }
  // the four waves' sums, one quantity at a time (fixed order); wave jx & 3 writes quantity jx
  float* out = part + ((int64_t)sgm * DORA_OS + os) * (1 + st.r) * st.K + k;
#pragma unroll
  for (int jx = 0; jx <= RC; ++jx) {
    if (jx <= st.r) {  // block-uniform
      red[w][lane] = jx == 0 ? dg : da[jx > 0 ? jx - 1 : 0];
      __syncthreads();
      if (w == (jx & 3) && ok)
        *reinterpret_cast<f32x4*>(out + (int64_t)jx * st.K) = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
      __syncthreads();
    }
  }
}

// grid (ceil(K / 64), nseg), 256 threads: wave w takes the quantities j = w, w + 4, ... (j = 0: d(g); j = 1 + q: d(down)[q]);
// also leaves g / n per column behind the partials for the row-wise kernel (one division per column, not per element)
__global__ __launch_bounds__(256) void dora_col_final_kernel(DoraSite st, float* __restrict__ part,
                                                             const float* __restrict__ up, float* __restrict__ d_down,
                                                             float* __restrict__ d_up, float alpha,
                                                             const float* __restrict__ alpha_dev) {
  const int sgm = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  if (k >= st.K) return;
  const float al = alpha * (alpha_dev ? alpha_dev[0] : 1.f);
  const float n = st.cnorm[(int64_t)sgm * st.K + k];
  const float g = up[st.off_dora + (int64_t)sgm * st.K + k];
  if (w == 0) part[(int64_t)st.nseg * DORA_OS * (1 + st.r) * st.K + (int64_t)sgm * st.K + k] = g / n;
  for (int jx = w; jx <= st.r; jx += 4) {
    float s = 0.f;
    for (int os = 0; os < DORA_OS; ++os) s += part[(((int64_t)sgm * DORA_OS + os) * (1 + st.r) + jx) * st.K + k];
    if (jx == 0) d_up[st.off_dora + (int64_t)sgm * st.K + k] += al * s / n;
    else d_down[st.off_down + ((int64_t)sgm * st.r + (jx - 1)) * st.K + k] += al * (g / n) * s;
  }
}

// row-wise gradient: one wave per output row: d(up)[o][q] = a * sum_k G[o][k] (g[k] / n[k]) down[q][k]; gn = g / n per column.
// A lane takes four adjacent columns per step (16-byte loads; K % 8 == 0): 5 steps for K = 1280 instead of 20.
template <int RC>
__global__ __launch_bounds__(256) void dora_row_grad_kernel(DoraSite st, const float* __restrict__ G,
                                                            const float* __restrict__ down, const float* __restrict__ gn_all,
                                                            float* __restrict__ d_up, float alpha,
                                                            const float* __restrict__ alpha_dev) {
  const int lane = threadIdx.x & 63;
  const int64_t ot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ot >= (int64_t)st.nseg * st.cs) return;
  const int sgm = (int)(ot / st.cs);
  float acc[RC];
#pragma unroll
  for (int q = 0; q < RC; ++q) acc[q] = 0.f;
  const float* gn = gn_all + (int64_t)sgm * st.K;
  const float* Gr = G + ot * st.K;
  const float* dn = down + st.off_down + (int64_t)sgm * st.r * st.K;
  for (int k = lane * 4; k < st.K; k += 256) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(Gr + k);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(gn + k);
    const f32x4 t = gv * rv;
#pragma unroll
    for (int q = 0; q < RC; ++q) {
      if (q < st.r) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dn + (int64_t)q * st.K + k);
        acc[q] += (t[0] * d[0] + t[1] * d[1]) + (t[2] * d[2] + t[3] * d[3]);
      }
    }
  }
  const float al = alpha * (alpha_dev ? alpha_dev[0] : 1.f);
#pragma unroll
  for (int q = 0; q < RC; ++q) {
    if (q < st.r) {
      const float sum = wave_sum(acc[q]);
      if (lane == 0) d_up[st.off_up + ot * st.r + q] += al * sum;
    }
  }
}

__global__ void scale_min_kernel(const float* __restrict__ sc, int n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float m = sc[0];
    for (int j = 1; j < n; ++j) m = fminf(m, sc[j]);
    out[0] = m;
    out[1] = 1.f / m;
    for (int j = 0; j < n; ++j) out[2 + j] = m / sc[j];
  }
}
}  // namespace

int launch_dora_prep(int dtype, const DoraSite* sites_dev, const DoraSite* sites_host, int n_sites, const float* down,
                     const float* up, float mult, hipStream_t stream) {
  if (n_sites <= 0) return 0;
  int kmax = 0, segmax = 1;
  for (int i = 0; i < n_sites; ++i) {
    SMI_CHECK(sites_host[i].r >= 1 && sites_host[i].r <= DORA_RMAX && sites_host[i].K % 8 == 0, "dora: rank %d, K %d",
              sites_host[i].r, sites_host[i].K);
    kmax = std::max(kmax, sites_host[i].K);
    segmax = std::max(segmax, sites_host[i].nseg);
  }
  dim3 gn(cdiv(kmax, 256), segmax, n_sites), gd(128, n_sites);
  int rmax = 1;
  for (int i = 0; i < n_sites; ++i) rmax = std::max(rmax, sites_host[i].r);
#define NORM(TT_, RC_) hipLaunchKernelGGL((dora_norm_kernel<TT_, RC_>), gn, dim3(256), 0, stream, sites_dev, down, up)
#define NORMS(TT_)                                                                         \
  do {                                                                                     \
    if (rmax <= 4) NORM(TT_, 4); else if (rmax <= 8) NORM(TT_, 8);                         \
    else if (rmax <= 16) NORM(TT_, 16); else NORM(TT_, 32);                                \
  } while (0)
  if (dtype == DT_F16) {
    NORMS(f16);
    hipLaunchKernelGGL(dora_delta_kernel<f16>, gd, dim3(256), 0, stream, sites_dev, down, up, mult);
  } else {
    NORMS(bf16);
    hipLaunchKernelGGL(dora_delta_kernel<bf16>, gd, dim3(256), 0, stream, sites_dev, down, up, mult);
  }
#undef NORMS
#undef NORM
  SMI_HIP(hipGetLastError());
  return 0;
}

int launch_dora_transpose(int dtype, const DoraSite* sites_dev, const DoraSite* sites_host, int n_sites,
                          hipStream_t stream) {
  if (n_sites <= 0) return 0;
  int tmax = 1;
  for (int i = 0; i < n_sites; ++i)
    tmax = std::max(tmax, cdiv(sites_host[i].K, 64) * cdiv(sites_host[i].nseg * sites_host[i].cs, 64));
  if (dtype == DT_F16)
    hipLaunchKernelGGL(dora_transpose_kernel<f16>, dim3(tmax, n_sites), dim3(256), 0, stream, sites_dev);
  else
    hipLaunchKernelGGL(dora_transpose_kernel<bf16>, dim3(tmax, n_sites), dim3(256), 0, stream, sites_dev);
  SMI_HIP(hipGetLastError());
  return 0;
}

int launch_transpose_scaled(int dtype, const void* src, int64_t lds, void* dst, int M, int C, int Mp, const float* f,
                            int rows_per_sample, hipStream_t stream) {
  const int grid = cdiv(Mp, 64) * cdiv(C, 64);
  if (C % 8 == 0 && lds % 8 == 0 && Mp % 64 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    if (dtype == DT_F16)
      hipLaunchKernelGGL(transpose_scaled_vec_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)src, lds, (f16*)dst, M, C, Mp, f, rows_per_sample);
    else
      hipLaunchKernelGGL(transpose_scaled_vec_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)src, lds, (bf16*)dst, M, C, Mp, f, rows_per_sample);
    SMI_HIP(hipGetLastError());
    return 0;
  }
  if (dtype == DT_F16)
    hipLaunchKernelGGL(transpose_scaled_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)src, lds, (f16*)dst, M, C, Mp, f, rows_per_sample);
  else
    hipLaunchKernelGGL(transpose_scaled_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)src, lds, (bf16*)dst, M, C, Mp, f, rows_per_sample);
  SMI_HIP(hipGetLastError());
  return 0;
}

size_t dora_grad_scratch_floats(const DoraSite& site) {  // row-slice partials + g / n per column
  return (size_t)site.nseg * DORA_OS * (1 + site.r) * site.K + (size_t)site.nseg * site.K;
}

int launch_dora_grads(int dtype, const DoraSite& site, const float* G, const float* down, const float* up,
                      float* d_down, float* d_up, float alpha, const float* alpha_dev, float* scratch,
                      hipStream_t stream) {
  SMI_CHECK(site.r >= 1 && site.r <= DORA_RMAX, "dora: rank %d", site.r);
  dim3 gc(cdiv(site.K, 256), site.nseg, DORA_OS);
  const float* gn = scratch + (size_t)site.nseg * DORA_OS * (1 + site.r) * site.K;
  const dim3 gr(cdiv(site.nseg * site.cs, 4));
#define COL(TT_, RC_) hipLaunchKernelGGL((dora_col_grad_kernel<TT_, RC_>), gc, dim3(256), 0, stream, site, G, down, up, scratch)
#define ROW(RC_) hipLaunchKernelGGL((dora_row_grad_kernel<RC_>), gr, dim3(256), 0, stream, site, G, down, gn, d_up, alpha, alpha_dev)
#define BYRANK(X_)                                                                               \
  do {                                                                                           \
    if (site.r <= 4) { X_(4); } else if (site.r <= 8) { X_(8); } else if (site.r <= 16) { X_(16); } else { X_(32); } \
  } while (0)
#define COLF16(RC_) COL(f16, RC_)
#define COLBF16(RC_) COL(bf16, RC_)
  if (dtype == DT_F16) BYRANK(COLF16); else BYRANK(COLBF16);
  hipLaunchKernelGGL(dora_col_final_kernel, dim3(cdiv(site.K, 64), site.nseg), dim3(256), 0, stream, site, scratch, up,
                     d_down, d_up, alpha, alpha_dev);
  BYRANK(ROW);
#undef COLF16
#undef COLBF16
#undef BYRANK
#undef ROW
#undef COL
  SMI_HIP(hipGetLastError());
  return 0;
}

int launch_scale_min(const float* scale_buf, int n, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(scale_min_kernel, dim3(1), dim3(64), 0, stream, scale_buf, n, out);
  SMI_HIP(hipGetLastError());
  return 0;
}

// Fills the launch geometry of one job (host): column-block width, M-split, scratch need.
void wgrad_job_plan(WgradJob& j) {
  const int cols8 = j.K / 8;
  int cw = 256;
  while (cw > 16 && cols8 % cw != 0) cw >>= 1;  // largest power of two (16..256) dividing the vector count
  if (cols8 % cw != 0) {                        // none does: one power-of-two block >= the count (tail lanes idle)
    cw = 1;
    while (cw < cols8 && cw < 256) cw <<= 1;
  }
  j.cw = cw;
  j.ncolblk = cdiv(cols8, cw);
  // rows per workgroup: the per-workgroup fixed cost (P staging, 2 r barriers of the LDS combine, the partial store) is
  // paid once per rows_per_wg rows; SMI_WGRAD_ROWS overrides (experiments)
  static const int rows_env = []() { const char* e = getenv("SMI_WGRAD_ROWS"); return e ? atoi(e) : 0; }();
  // (measured at the headline shape, M = 4096: lora class 6.07 / 5.54 / 5.07 ms with 64 / 128 / 256 rows)
  j.rows_per_wg = rows_env > 0 ? rows_env : (j.M >= 2048 ? 256 : 64);
  j.nsplit = cdiv(j.M, j.rows_per_wg);
}
size_t wgrad_job_scratch_floats(const WgradJob& j) { return (size_t)j.nsplit * j.r * j.K; }

// jobs_host: the table as the device will see it (wg0 / fb0 filled here); jobs_dev: its device copy.  Jobs must be
// sorted by accumulator class (r <= 4, <= 8, <= 16, <= 32): one partial launch per class present, one final launch.
int wgrad_grouped_finish(std::vector<WgradJob>& jobs) {
  std::stable_sort(jobs.begin(), jobs.end(), [](const WgradJob& a, const WgradJob& b) {
    auto cls = [](int r) { return r <= 4 ? 0 : r <= 8 ? 1 : r <= 16 ? 2 : 3; };
    return cls(a.r) < cls(b.r);
  });
  int fb = 0;
  int cur_cls = -1, wg = 0;
  for (auto& j : jobs) {
    const int c = j.r <= 4 ? 0 : j.r <= 8 ? 1 : j.r <= 16 ? 2 : 3;
    if (c != cur_cls) {
      cur_cls = c;
      wg = 0;
    }
    j.wg0 = wg;
    wg += j.nsplit * j.ncolblk;
    j.fb0 = fb;
    fb += (int)(((int64_t)j.r * j.K + 31) / 32);
  }
  return 0;
}

int launch_lora_wgrad_grouped(int dtype, const std::vector<WgradJob>& jobs, const WgradJob* jobs_dev,
                              hipStream_t stream) {
  if (jobs.empty()) return 0;
  size_t i = 0;
  int final_blocks = 0;
  while (i < jobs.size()) {
    auto cls = [](int r) { return r <= 4 ? 0 : r <= 8 ? 1 : r <= 16 ? 2 : 3; };
    const int c = cls(jobs[i].r);
    size_t k = i;
    int wgs = 0;
    while (k < jobs.size() && cls(jobs[k].r) == c) {
      SMI_CHECK(jobs[k].r >= 1 && jobs[k].r <= 32 && jobs[k].K % 8 == 0 && jobs[k].ldx % 8 == 0,
                "lora_wgrad: r=%d K=%d", jobs[k].r, jobs[k].K);
      wgs += jobs[k].nsplit * jobs[k].ncolblk;
      final_blocks += (int)(((int64_t)jobs[k].r * jobs[k].K + 31) / 32);
      ++k;
    }
    const int n = (int)(k - i);
#define L(TT_, RR) hipLaunchKernelGGL((lora_wgrad_grouped_partial_kernel<TT_, RR>), dim3(wgs), dim3(256), 0, stream, jobs_dev + i, n)
    if (dtype == DT_F16) {
      if (c == 0) L(f16, 4); else if (c == 1) L(f16, 8); else if (c == 2) L(f16, 16); else L(f16, 32);
    } else {
      if (c == 0) L(bf16, 4); else if (c == 1) L(bf16, 8); else if (c == 2) L(bf16, 16); else L(bf16, 32);
    }
#undef L
    i = k;
  }
  hipLaunchKernelGGL(lora_wgrad_grouped_final_kernel, dim3(final_blocks), dim3(256), 0, stream, jobs_dev,
                     (int)jobs.size());
  SMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace smi
