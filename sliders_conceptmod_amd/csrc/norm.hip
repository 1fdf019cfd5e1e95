// GroupNorm (+SiLU) and LayerNorm, forward and activation-gradient, on token-major [N, HW, C] tensors.
// HBM-bound: every pass reads/writes each element once with 16-byte lanes; statistics are two-stage and
// deterministic (per-chunk partials reduced in a fixed order -- no float atomics).
//
// GroupNorm forward:  partial(sum, sumsq per (n, chunk, g)) -> apply: every workgroup first reduces the (<= 64) chunk
//                     partials of its sample to mean / rstd in LDS (a few KB from L2; a separate "finalize" launch was
//                     25 us of pure latency, 736 times per step), then y = silu?(x*a + b) for its rows; the workgroup
//                     of chunk 0 also stores mean / rstd and the per-(n,c) affine a, b for the backward
// GroupNorm backward: z = x*a+b, dz = dy*silu'(z);  S1 = sum dz*gamma, S2 = sum dz*(z-beta)  per (n,g)
//                     dx = a*dz + c2*x + c3   with  c2 = -r^2*S2/cnt,  c3 = -r*S1/cnt - c2*mu
#include "kernels.h"

namespace smi {
namespace {

// rows per chunk: at most 64 chunks per sample (the apply kernels re-reduce the chunk partials), at least 16 rows
// (the 32x32 maps then still give 64 workgroups per sample).  Depends on HW only, so a sample's arithmetic does not
// depend on the batch it travels in.
// Maps beyond 128 x 128 (the VAE encoder: 1024 x 1024 x 128 channels of ONE image) get 256-row chunks instead -- 64
// workgroups on 256 CUs ran those at 0.64 TB/s -- and their partials are folded down to 64 slots by gn_fold_kernel
// before the apply kernels re-reduce them.
constexpr int GN_MAX_SLOTS = 64;
__host__ __device__ inline int gn_rows_per_chunk(int HW) {
  if (HW > 16384) return 256;
  const int r = (HW + 63) / 64;
  return r < 16 ? 16 : (r + 15) / 16 * 16;
}

struct GnGeom {
  int cols8;  // C / 8
  int ncol;   // column vectors per thread (ceil(cols8 / 256))
  int rpar;   // rows processed in parallel by one block
};
__host__ __device__ inline GnGeom gn_geom(int C) {
  GnGeom g;
  g.cols8 = C / 8;
  g.ncol = (g.cols8 + 255) / 256;
  g.rpar = g.cols8 >= 256 ? 1 : 256 / g.cols8;
  return g;
}

// MODE 0: forward stats (sum x, sum x^2).  MODE 1: backward stats (S1, S2), SILU selects dz = dy*silu'(z)
template <typename T, int MODE, bool SILU>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const T* __restrict__ gamma, const T* __restrict__ beta,
                                                         const float* __restrict__ aa, const float* __restrict__ bb,
                                                         float* __restrict__ partial, int Nb, int HW, int C, int G,
                                                         int nchunk) {
  extern __shared__ float red[];  // [rpar][C][2]
  const GnGeom gg = gn_geom(C);
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int rsub = gg.cols8 >= 256 ? 0 : tid / gg.cols8;
  const int col_base = gg.cols8 >= 256 ? tid : tid - rsub * gg.cols8;
  const bool active = rsub < gg.rpar;
  const int row0 = chunk * gn_rows_per_chunk(HW);
  const int row1 = min(HW, row0 + gn_rows_per_chunk(HW));
  const float* an = aa + (int64_t)n * C;  // only dereferenced in MODE 1
  const float* bn = bb + (int64_t)n * C;

  for (int j = 0; j < gg.ncol; ++j) {
    const int col = col_base + 256 * j;
    const bool cok = active && col < gg.cols8;
    float s0[8], s1[8], av[8], bv[8], gv[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
    if (cok && MODE == 1) {
      Pack8<T> g8, b8;
      g8.u = *reinterpret_cast<const u32x4*>(gamma + col * 8);
      b8.u = *reinterpret_cast<const u32x4*>(beta + col * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        av[e] = an[col * 8 + e];
        bv[e] = bn[col * 8 + e];
        gv[e] = to_f(g8.e[e]);
        be[e] = to_f(b8.e[e]);
      }
    }
    if (cok) {
      // 4 independent row loads in flight per thread (otherwise a chain of dependent HBM latencies)
      for (int r0 = row0 + rsub; r0 < row1; r0 += 4 * gg.rpar) {
        Pack8<T> xv4[4], dv4[4];
        bool ok4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + u * gg.rpar;
          ok4[u] = r < row1;
          const int64_t off = ((int64_t)n * HW + (ok4[u] ? r : row0)) * C + col * 8;
          xv4[u].u = *reinterpret_cast<const u32x4*>(x + off);
          if (MODE == 1) dv4[u].u = *reinterpret_cast<const u32x4*>(dy + off);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok4[u]) continue;
          if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = to_f(xv4[u].e[e]);
              s0[e] += v;
              s1[e] += v * v;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float z = to_f(xv4[u].e[e]) * av[e] + bv[e];
              float dz = to_f(dv4[u].e[e]);
              if (SILU) dz *= dsilu_f(z);
              s0[e] += dz * gv[e];
              s1[e] += dz * (z - be[e]);
            }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[((int64_t)rsub * C + col * 8 + e) * 2 + 0] = s0[e];
        red[((int64_t)rsub * C + col * 8 + e) * 2 + 1] = s1[e];
      }
    }
  }
  __syncthreads();
  const int cpg = C / G;
  for (int g = tid; g < G; g += 256) {
    float a0 = 0.f, a1 = 0.f;
    for (int rs = 0; rs < gg.rpar; ++rs)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        a0 += red[((int64_t)rs * C + c) * 2 + 0];
        a1 += red[((int64_t)rs * C + c) * 2 + 1];
      }
    float* out = partial + (((int64_t)n * nchunk + chunk) * G + g) * 2;
    out[0] = a0;
    out[1] = a1;
  }
}

// partial [Nb][nchunk][G][2] -> folded [Nb][GN_MAX_SLOTS][G][2]: slot s = sum of chunks [s per, (s + 1) per) in order
__global__ __launch_bounds__(256) void gn_fold_kernel(const float* __restrict__ partial, float* __restrict__ folded,
                                                      int nchunk, int G2) {
  const int n = blockIdx.y, s = blockIdx.x;
  const int per = (nchunk + GN_MAX_SLOTS - 1) / GN_MAX_SLOTS;
  const int c0 = s * per, c1 = min(nchunk, c0 + per);
  for (int t = threadIdx.x; t < G2; t += 256) {
    float a = 0.f;
    for (int c = c0; c < c1; ++c) a += partial[((int64_t)n * nchunk + c) * G2 + t];
    folded[((int64_t)n * GN_MAX_SLOTS + s) * G2 + t] = a;
  }
}

// MODE 0: y = silu?(x*a+b).   MODE 1: dx = a*dz + c2*x + c3 (+ add).
// Grid (chunk, n) as the partial kernel.  Head: the chunk partials of sample n -> per-group statistics in LDS
// (8 lanes per group walk the partials, xor tree: fixed order, deterministic), then the rows of this chunk.
template <typename T, int MODE, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                       const T* __restrict__ gamma, const T* __restrict__ beta,
                                                       const float* __restrict__ partial, float* __restrict__ aa,
                                                       float* __restrict__ bb, float* __restrict__ mean_rstd,
                                                       const T* add, T* out, int Nb,
                                                       int HW, int C, int G, int nchunk, float eps) {
  // nchunk = slots of `partial` per sample (the grid's chunk count, or GN_MAX_SLOTS after a fold)
  extern __shared__ float st[];  // [G][2]: MODE 0 (mean, rstd); MODE 1 (c2, c3)
  const GnGeom gg = gn_geom(C);
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int cpg = C / G;
  const float cnt = (float)HW * (float)cpg;
  for (int g = tid >> 3; g < G; g += 32) {
    float s = 0.f, sq = 0.f;
    for (int ch = tid & 7; ch < nchunk; ch += 8) {
      const float* p = partial + (((int64_t)n * nchunk + ch) * G + g) * 2;
      s += p[0];
      sq += p[1];
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      sq += __shfl_xor(sq, o);
    }
    if (tid & 7) continue;
    if (MODE == 0) {
      const float mean = s / cnt;
      const float var = fmaxf(sq / cnt - mean * mean, 0.f);
      const float rstd = rsqrtf(var + eps);
      st[g * 2] = mean;
      st[g * 2 + 1] = rstd;
      if (chunk == 0) {
        mean_rstd[((int64_t)n * G + g) * 2] = mean;
        mean_rstd[((int64_t)n * G + g) * 2 + 1] = rstd;
      }
    } else {
      const float mu = mean_rstd[((int64_t)n * G + g) * 2];
      const float r = mean_rstd[((int64_t)n * G + g) * 2 + 1];
      const float c2 = -r * r * sq / cnt;  // (s, sq) = (S1, S2)
      st[g * 2] = c2;
      st[g * 2 + 1] = -r * s / cnt - c2 * mu;
    }
  }
  __syncthreads();
  const int rsub = gg.cols8 >= 256 ? 0 : tid / gg.cols8;
  const int col_base = gg.cols8 >= 256 ? tid : tid - rsub * gg.cols8;
  if (rsub >= gg.rpar) return;
  const int rpc = gn_rows_per_chunk(HW);
  const int row0 = chunk * rpc;
  const int row1 = min(HW, row0 + rpc);
  float* an = aa + (int64_t)n * C;
  float* bn = bb + (int64_t)n * C;
  for (int j = 0; j < gg.ncol; ++j) {
    const int col = col_base + 256 * j;
    if (col >= gg.cols8) break;
    float av[8], bv[8], c2[8], c3[8];
    if (MODE == 0) {
      Pack8<T> g8, b8;
      g8.u = *reinterpret_cast<const u32x4*>(gamma + col * 8);
      b8.u = *reinterpret_cast<const u32x4*>(beta + col * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (col * 8 + e) / cpg;
        av[e] = st[g * 2 + 1] * to_f(g8.e[e]);
        bv[e] = to_f(b8.e[e]) - st[g * 2] * av[e];
      }
      if (chunk == 0 && rsub == 0) {  // the backward reads the affine back
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          an[col * 8 + e] = av[e];
          bn[col * 8 + e] = bv[e];
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (col * 8 + e) / cpg;
        av[e] = an[col * 8 + e];
        bv[e] = bn[col * 8 + e];
        c2[e] = st[g * 2];
        c3[e] = st[g * 2 + 1];
      }
    }
    for (int r0 = row0 + rsub; r0 < row1; r0 += 4 * gg.rpar) {
      Pack8<T> xv4[4], dv4[4], ad4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * gg.rpar;
        const int64_t off = ((int64_t)n * HW + (r < row1 ? r : row0)) * C + col * 8;
        xv4[u].u = *reinterpret_cast<const u32x4*>(x + off);
        if (MODE == 1) {
          dv4[u].u = *reinterpret_cast<const u32x4*>(dy + off);
          if (add) ad4[u].u = *reinterpret_cast<const u32x4*>(add + off);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * gg.rpar;
        if (r >= row1) continue;
        Pack8<T> o;
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float z = to_f(xv4[u].e[e]) * av[e] + bv[e];
            if (SILU) z = silu_f(z);
            o.e[e] = from_f<T>(z);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xe = to_f(xv4[u].e[e]);
            float dz = to_f(dv4[u].e[e]);
            if (SILU) dz *= dsilu_f(xe * av[e] + bv[e]);
            o.e[e] = from_f<T>(av[e] * dz + c2[e] * xe + c3[e] + (add ? to_f(ad4[u].e[e]) : 0.f));
          }
        }
        *reinterpret_cast<u32x4*>(out + ((int64_t)n * HW + r) * C + col * 8) = o.u;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, C <= 8*64*LN_MAXV
// ---------------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 4;  // C <= 2048

template <typename T, int MODE>  // 0 fwd, 1 bwd
__global__ __launch_bounds__(256) void ln_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                 const T* __restrict__ gamma, const T* beta, T* out,
                                                 float* __restrict__ mean_rstd, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nvec = C / 8;
  float xv[LN_MAXV][8];
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int v = lane + 64 * j;
    if (v < nvec) {
      Pack8<T> t;
      t.u = *reinterpret_cast<const u32x4*>(x + (int64_t)row * C + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[j][e] = to_f(t.e[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[j][e] = 0.f;
    }
  }
  float mean, rstd;
  if (MODE == 0) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += xv[j][e];
    mean = wave_sum(s) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = xv[j][e] - mean;
          sq += d * d;
        }
      }
    }
    rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    if (lane == 0) {
      mean_rstd[(int64_t)row * 2] = mean;
      mean_rstd[(int64_t)row * 2 + 1] = rstd;
    }
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        Pack8<T> g8, b8, o;
        g8.u = *reinterpret_cast<const u32x4*>(gamma + v * 8);
        b8.u = *reinterpret_cast<const u32x4*>(beta + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = from_f<T>((xv[j][e] - mean) * rstd * to_f(g8.e[e]) + to_f(b8.e[e]));
        *reinterpret_cast<u32x4*>(out + (int64_t)row * C + v * 8) = o.u;
      }
    }
  } else {
    mean = mean_rstd[(int64_t)row * 2];
    rstd = mean_rstd[(int64_t)row * 2 + 1];
    float dg[LN_MAXV][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        Pack8<T> g8, d8;
        g8.u = *reinterpret_cast<const u32x4*>(gamma + v * 8);
        d8.u = *reinterpret_cast<const u32x4*>(dy + (int64_t)row * C + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xv[j][e] = (xv[j][e] - mean) * rstd;  // xhat
          dg[j][e] = to_f(d8.e[e]) * to_f(g8.e[e]);
          s1 += dg[j][e];
          s2 += dg[j][e] * xv[j][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) dg[j][e] = 0.f;
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        Pack8<T> o, ad;
        if (beta) ad.u = *reinterpret_cast<const u32x4*>(beta + (int64_t)row * C + v * 8);  // bwd: beta = `add`
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o.e[e] = from_f<T>(rstd * (dg[j][e] - s1 - xv[j][e] * s2) + (beta ? to_f(ad.e[e]) : 0.f));
        *reinterpret_cast<u32x4*>(out + (int64_t)row * C + v * 8) = o.u;
      }
    }
  }
}

inline int ew_grid(int64_t n_threads) {
  int64_t g = (n_threads + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// more than GN_MAX_SLOTS chunks per sample: fold them into the slots behind the raw partials; returns what to re-reduce
inline const float* gn_fold(float* partial, int Nb, int nchunk, int G, hipStream_t st) {
  if (nchunk <= GN_MAX_SLOTS) return partial;
  float* folded = partial + (size_t)Nb * nchunk * G * 2;
  hipLaunchKernelGGL(gn_fold_kernel, dim3(GN_MAX_SLOTS, Nb), dim3(256), 0, st, partial, folded, nchunk, G * 2);
  return folded;
}

template <typename T>
int gn_fwd_t(const void* x, const void* gamma, const void* beta, void* y, float* ab, float* mean_rstd, float* partial,
             int Nb, int HW, int C, int G, float eps, int silu, hipStream_t st) {
  const int nchunk = gn_num_chunks(HW);
  const GnGeom gg = gn_geom(C);
  const size_t sm = (size_t)gg.rpar * C * 2 * sizeof(float);
  hipLaunchKernelGGL((gn_partial_kernel<T, 0, false>), dim3(nchunk, Nb), dim3(256), sm, st, (const T*)x, nullptr,
                     nullptr, nullptr, ab, ab, partial, Nb, HW, C, G, nchunk);
  const size_t sa = (size_t)G * 2 * sizeof(float);
  const float* red = gn_fold(partial, Nb, nchunk, G, st);
  const int nslot = nchunk > GN_MAX_SLOTS ? GN_MAX_SLOTS : nchunk;
  if (silu)
    hipLaunchKernelGGL((gn_apply_kernel<T, 0, true>), dim3(nchunk, Nb), dim3(256), sa, st, (const T*)x, nullptr,
                       (const T*)gamma, (const T*)beta, red, ab, ab + (size_t)Nb * C, mean_rstd, nullptr, (T*)y, Nb, HW, C, G, nslot, eps);
  else
    hipLaunchKernelGGL((gn_apply_kernel<T, 0, false>), dim3(nchunk, Nb), dim3(256), sa, st, (const T*)x, nullptr,
                       (const T*)gamma, (const T*)beta, red, ab, ab + (size_t)Nb * C, mean_rstd, nullptr, (T*)y, Nb, HW, C, G, nslot, eps);
  SMI_HIP(hipGetLastError());
  return 0;
}

// aa / bb: the forward's per-(n,c) affine of the Nb samples being differentiated ([Nb][C] each)
template <typename T>
int gn_bwd_t(const void* x, const void* dy, const void* gamma, const void* beta, const float* aa, const float* bb,
             const float* mean_rstd, const void* add, void* dx, float* partial, int Nb, int HW, int C, int G, int silu,
             hipStream_t st) {
  const int nchunk = gn_num_chunks(HW);
  const GnGeom gg = gn_geom(C);
  const size_t sm = (size_t)gg.rpar * C * 2 * sizeof(float);
  if (silu)
    hipLaunchKernelGGL((gn_partial_kernel<T, 1, true>), dim3(nchunk, Nb), dim3(256), sm, st, (const T*)x,
                       (const T*)dy, (const T*)gamma, (const T*)beta, aa, bb, partial, Nb, HW, C, G, nchunk);
  else
    hipLaunchKernelGGL((gn_partial_kernel<T, 1, false>), dim3(nchunk, Nb), dim3(256), sm, st, (const T*)x,
                       (const T*)dy, (const T*)gamma, (const T*)beta, aa, bb, partial, Nb, HW, C, G, nchunk);
  const size_t sa = (size_t)G * 2 * sizeof(float);
  float* abp = const_cast<float*>(aa);
  float* bbp = const_cast<float*>(bb);
  float* mrp = const_cast<float*>(mean_rstd);
  const float* red = gn_fold(partial, Nb, nchunk, G, st);
  const int nslot = nchunk > GN_MAX_SLOTS ? GN_MAX_SLOTS : nchunk;
  if (silu)
    hipLaunchKernelGGL((gn_apply_kernel<T, 1, true>), dim3(nchunk, Nb), dim3(256), sa, st, (const T*)x, (const T*)dy,
                       (const T*)gamma, (const T*)beta, red, abp, bbp, mrp, (const T*)add, (T*)dx, Nb, HW, C, G, nslot, 0.f);
  else
    hipLaunchKernelGGL((gn_apply_kernel<T, 1, false>), dim3(nchunk, Nb), dim3(256), sa, st, (const T*)x, (const T*)dy,
                       (const T*)gamma, (const T*)beta, red, abp, bbp, mrp, (const T*)add, (T*)dx, Nb, HW, C, G, nslot, 0.f);
  SMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int gn_num_chunks(int HW) { return (HW + gn_rows_per_chunk(HW) - 1) / gn_rows_per_chunk(HW); }
size_t gn_partial_floats(int Nb, int HW, int G) {
  const int nchunk = gn_num_chunks(HW);
  return (size_t)Nb * (nchunk + (nchunk > GN_MAX_SLOTS ? GN_MAX_SLOTS : 0)) * G * 2;
}

int launch_groupnorm_fwd(int dtype, const void* x, const void* gamma, const void* beta, void* y, float* ab,
                         float* mean_rstd, float* partial, int Nb, int HW, int C, int G, float eps, int silu,
                         hipStream_t stream) {
  SMI_CHECK(C % 8 == 0 && C % G == 0, "groupnorm: C=%d must be a multiple of 8 and of G=%d", C, G);
  SMI_CHECK((size_t)gn_geom(C).rpar * C * 8 <= 65536, "groupnorm: C=%d too large for the LDS reduction", C);
  return dtype == DT_F16 ? gn_fwd_t<f16>(x, gamma, beta, y, ab, mean_rstd, partial, Nb, HW, C, G, eps, silu, stream)
                         : gn_fwd_t<bf16>(x, gamma, beta, y, ab, mean_rstd, partial, Nb, HW, C, G, eps, silu, stream);
}

// `partial` must hold Nb*nchunk*G*2 floats.  `a`, `b`: the forward's affine of these Nb samples ([Nb][C] each).
int launch_groupnorm_bwd(int dtype, const void* x, const void* dy, const void* gamma, const void* beta,
                         const float* a, const float* b, const float* mean_rstd, const void* add, void* dx,
                         float* partial, int Nb, int HW, int C, int G, int silu, hipStream_t stream) {
  SMI_CHECK(C % 8 == 0 && C % G == 0, "groupnorm bwd: C=%d G=%d", C, G);
  return dtype == DT_F16
             ? gn_bwd_t<f16>(x, dy, gamma, beta, a, b, mean_rstd, add, dx, partial, Nb, HW, C, G, silu, stream)
             : gn_bwd_t<bf16>(x, dy, gamma, beta, a, b, mean_rstd, add, dx, partial, Nb, HW, C, G, silu, stream);
}

int launch_layernorm_fwd(int dtype, const void* x, const void* gamma, const void* beta, void* y, float* mean_rstd,
                         int M, int C, float eps, hipStream_t stream) {
  SMI_CHECK(C % 8 == 0 && C <= 8 * 64 * LN_MAXV, "layernorm: C=%d unsupported", C);
  const int grid = cdiv(M, 4);
  if (dtype == DT_F16)
    hipLaunchKernelGGL((ln_kernel<f16, 0>), dim3(grid), dim3(256), 0, stream, (const f16*)x, nullptr,
                       (const f16*)gamma, (const f16*)beta, (f16*)y, mean_rstd, M, C, eps);
  else
    hipLaunchKernelGGL((ln_kernel<bf16, 0>), dim3(grid), dim3(256), 0, stream, (const bf16*)x, nullptr,
                       (const bf16*)gamma, (const bf16*)beta, (bf16*)y, mean_rstd, M, C, eps);
  SMI_HIP(hipGetLastError());
  return 0;
}

int launch_layernorm_bwd(int dtype, const void* x, const void* dy, const void* gamma, const float* mean_rstd,
                         const void* add, void* dx, int M, int C, hipStream_t stream) {
  SMI_CHECK(C % 8 == 0 && C <= 8 * 64 * LN_MAXV, "layernorm: C=%d unsupported", C);
  const int grid = cdiv(M, 4);
  if (dtype == DT_F16)
    hipLaunchKernelGGL((ln_kernel<f16, 1>), dim3(grid), dim3(256), 0, stream, (const f16*)x, (const f16*)dy,
                       (const f16*)gamma, (const f16*)add, (f16*)dx, const_cast<float*>(mean_rstd), M, C, 0.f);
  else
    hipLaunchKernelGGL((ln_kernel<bf16, 1>), dim3(grid), dim3(256), 0, stream, (const bf16*)x, (const bf16*)dy,
                       (const bf16*)gamma, (const bf16*)add, (bf16*)dx, const_cast<float*>(mean_rstd), M, C, 0.f);
  SMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace smi
