// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950.
//
//   C[M,N] = A[M,K] * W[N,K]^T + epilogue            (both operands K-contiguous: "NT")
//
// Tile: 128(M) x 128(N) x 64(K), 256 threads = 4 waves in a 2x2 grid, each wave 64x64 as 4x4 tiles of
// v_mfma_f32_16x16x32_{f16,bf16}.  Operands are staged global -> registers -> LDS (one 16-byte chunk per lane per
// load, XOR-swizzled 128-byte LDS rows, conflict-free ds_read_b128 fragment reads), with the next K-tile's global
// loads in flight under the current tile's MFMAs.  The MFMA is issued as D = W_frag x A_frag so that each lane
// ends up with 4 consecutive output columns (n) of one row (m): 8-byte (T) / 16-byte (f32) stores and a vector
// epilogue (bias, per-image row vector, LoRA rank-r delta, residual) in fp32 before the single rounding.
//
// Convolution: the A operand is gathered straight from the NHWC activation (no im2col buffer):
// row m = (image, oy, ox), K index = (ky, kx, ci).  Cin % 64 == 0 keeps every 64-wide K step inside one filter tap,
// so the tap geometry is wave-uniform scalar work; padding, tails and strided/transposed/up-sampled taps turn
// into out-of-range buffer offsets, which the hardware range check returns as zeros.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "kernels.h"
#include <type_traits>

#include <mutex>

namespace smi {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

struct ConvRow {
  int base;  // image index * Hin * Win  (pixels)
  int oy, ox;
};

template <typename T, bool CONV>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BM * BK * 2];
  unsigned char* As = smem;
  unsigned char* Bs = smem + BM * BK * 2;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap of the linear block id: blocks that share an XCD (id % 8) get a contiguous
  // range of tiles, so neighbouring tiles (same A rows / same W rows) hit the same L2.
  const int nbn = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int wg;
  {
    const int orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int bm0 = (wg / nbn) * BM;
  const int bn0 = (wg % nbn) * BN;

  const uint32_t esz = 2;
  __amdgpu_buffer_rsrc_t rA, rW;
  if (CONV) {
    rA = make_rsrc(p.A, (uint32_t)((int64_t)p.Nb * p.Hin * p.Win * p.Cin * esz));
  } else {
    rA = make_rsrc(p.A, (uint32_t)(((int64_t)(p.M - 1) * p.lda + p.K) * esz));
  }
  rW = make_rsrc(p.W, (uint32_t)((int64_t)p.N * p.K * esz));

  // ---- staging assignment: thread -> chunk (16 B) c of rows r0 + 32*i
  const int sc = tid & 7;
  const int sr = tid >> 3;
  ConvRow crow[4];
  uint32_t a_rowoff[4];
  uint32_t w_rowoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = bm0 + sr + 32 * i;
    if (CONV) {
      if (m < p.M) {
        const int hw = p.Hout * p.Wout;
        const int img = m / hw;
        const int rem = m - img * hw;
        crow[i].base = img * p.Hin * p.Win;
        crow[i].oy = rem / p.Wout;
        crow[i].ox = rem - crow[i].oy * p.Wout;
      } else {
        crow[i].base = 0;
        crow[i].oy = -(1 << 20);
        crow[i].ox = -(1 << 20);
      }
      a_rowoff[i] = 0;
    } else {
      a_rowoff[i] = (m < p.M) ? (uint32_t)((int64_t)m * p.lda * esz) : OOB;
    }
    const int n = bn0 + sr + 32 * i;
    w_rowoff[i] = (n < p.N) ? (uint32_t)((int64_t)n * p.K * esz) : OOB;
  }

  u32x4 ra[4], rb[4];
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    const int kc = k0 + sc * 8;
    const bool kok = kc < p.K;
    if (CONV) {
      const int tap = k0 / p.Cin;
      const int c0 = k0 - tap * p.Cin;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int iy, ix;
        bool ok;
        if (p.transposed) {
          const int ty = crow[i].oy + 1 - ky, tx = crow[i].ox + 1 - kx;
          iy = ty / p.stride;
          ix = tx / p.stride;
          ok = (ty >= 0) && (tx >= 0) && (iy * p.stride == ty) && (ix * p.stride == tx) && (iy < p.Hin) &&
               (ix < p.Win);
        } else {
          iy = crow[i].oy * p.stride + ky - p.pad;
          ix = crow[i].ox * p.stride + kx - p.pad;
          if (p.upsample) {
            ok = (iy >= 0) && (ix >= 0) && (iy < 2 * p.Hin) && (ix < 2 * p.Win);
            iy >>= 1;
            ix >>= 1;
          } else {
            ok = (iy >= 0) && (ix >= 0) && (iy < p.Hin) && (ix < p.Win);
          }
        }
        const uint32_t off = ok ? (uint32_t)(((crow[i].base + iy * p.Win + ix) * (int64_t)p.Cin + c0 + sc * 8) * esz)
                                : OOB;
        ra[i] = buf_load16(rA, off);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t off = (kok && a_rowoff[i] != OOB) ? a_rowoff[i] + (uint32_t)kc * esz : OOB;
        ra[i] = buf_load16(rA, off);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t off = (kok && w_rowoff[i] != OOB) ? w_rowoff[i] + (uint32_t)kc * esz : OOB;
      rb[i] = buf_load16(rW, off);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = sr + 32 * i;
      const int off = r * 128 + ((sc ^ (r & 7)) << 4);
      *reinterpret_cast<u32x4*>(As + off) = ra[i];
      *reinterpret_cast<u32x4*>(Bs + off) = rb[i];
    }
  };

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  const int fr = lane & 15;
  const int fq = lane >> 4;

  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      typename TT<T>::v8 xa[4], wb[4];
      const int ch = kk * 4 + fq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rm = wm * 64 + i * 16 + fr;
        Pack8<T> t;
        t.u = *reinterpret_cast<const u32x4*>(As + rm * 128 + ((ch ^ (rm & 7)) << 4));
        xa[i] = t.v;
        const int rn = wn * 64 + i * 16 + fr;
        Pack8<T> s;
        s.u = *reinterpret_cast<const u32x4*>(Bs + rn * 128 + ((ch ^ (rn & 7)) << 4));
        wb[i] = s.v;
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = TT<T>::mfma16(wb[ni], xa[mi], acc[ni][mi]);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds, per (ni, mi), rows n = nb + 4*fq + {0..3} (MFMA row) of column m = mb + fr
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = bm0 + wm * 64 + mi * 16 + fr;
    if (m >= p.M) continue;
    const bool lora_on = p.lora_r > 0 && m >= p.lora_row0;
    const float* xrow0 = p.lora_xa + (int64_t)(m - p.lora_row0) * p.ld_xa;
    const int64_t vrow = p.rowvec ? (int64_t)(m / p.rows_per_vec) * (p.ld_rowvec ? p.ld_rowvec : (int64_t)p.N) : 0;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = bn0 + wn * 64 + ni * 16 + fq * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
      if (p.bias) {
        Pack4<T> b;
        b.u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.bias) + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += to_f(b.e[j]);
      }
      if (p.rowvec) {
        Pack4<T> b;
        b.u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.rowvec) + vrow + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += to_f(b.e[j]);
      }
      if (lora_on) {
        const float* xrow = xrow0 + (p.lora_seg ? (n / p.lora_seg) * p.lora_r : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* up = p.lora_up + (int64_t)(n + j) * p.up_sn;
          float d = 0.f;
          for (int q = 0; q < p.lora_r; ++q) d = __builtin_fmaf(xrow[q], up[q * p.up_sq], d);
          v[j] = __builtin_fmaf(d, p.lora_scale, v[j]);
        }
      }
      if (p.res) {
        Pack4<T> b;
        b.u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.res) + (int64_t)m * p.ldr + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += to_f(b.e[j]);
      }
      if (p.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) =
            f32x4{v[0], v[1], v[2], v[3]};
      } else {
        Pack4<T> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.e[j] = from_f<T>(v[j]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n) = o.u;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// direct 3x3 conv, stride 1, pad 1, for very small Cin or Cout (conv_in 4->C, conv_out C->4 and gradients).
// One thread = one output pixel x 4 output channels; weights T [Cout][9*Cin].
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const T* __restrict__ in, const T* __restrict__ w,
                                                            const T* __restrict__ bias, void* out, int out_f32,
                                                            int Nb, int H, int W, int Cin, int Cout) {
  const int cg = (Cout + 3) / 4;
  const int64_t total = (int64_t)Nb * H * W * cg;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % cg);
    const int64_t pix = idx / cg;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int n = (int)(pix / ((int64_t)W * H));
    const int co0 = g * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ky - 1;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = x + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const T* ip = in + (((int64_t)n * H + iy) * W + ix) * Cin;
        const int kb = (ky * 3 + kx) * Cin;
        for (int ci = 0; ci < Cin; ++ci) {
          const float v = to_f(ip[ci]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co0 + j < Cout) acc[j] += v * to_f(w[(int64_t)(co0 + j) * 9 * Cin + kb + ci]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (co0 + j >= Cout) continue;
      float v = acc[j] + (bias ? to_f(bias[co0 + j]) : 0.f);
      if (out_f32)
        reinterpret_cast<float*>(out)[pix * Cout + co0 + j] = v;
      else
        reinterpret_cast<T*>(out)[pix * Cout + co0 + j] = from_f<T>(v);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// split-K finish: C[m, n] = sum_s slab[s][m][n] (slices in ascending order: deterministic) + the GEMM epilogue
// (bias, row vector, rank-r delta as the canonical fmaf chain of smi_common.h, residual), one rounding at the store.
// One thread = 4 consecutive columns of one row.
// ------------------------------------------------------------------------------------------------------------
// NS > 0: the slice count as a compile-time constant -- with the run-time count hipcc keeps the slab loop rolled, one
// dependent L2 round trip per slice (6.9 us per launch, 7236 launches in a 22-step SD-1.4 run); unrolled, all slices of an
// element are in flight at once.  The sum still runs over ascending slices: same bits.
template <typename T, int NS>
__global__ __launch_bounds__(256) void splitk_finish_kernel(GemmParams p, const float* __restrict__ slab, int nsplit_rt) {
  const int nsplit = NS > 0 ? NS : nsplit_rt;
  const int n4 = p.N >> 2;
  const int64_t total = (int64_t)p.M * n4;
  const int64_t plane = (int64_t)p.M * p.N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / n4);
    const int n = (int)(idx - (int64_t)m * n4) * 4;
    const float* sp = slab + (int64_t)m * p.N + n;
    f32x4 v = *reinterpret_cast<const f32x4*>(sp);
    if constexpr (NS > 0) {
      f32x4 w[NS > 1 ? NS - 1 : 1];
#pragma unroll
      for (int s = 1; s < NS; ++s) w[s - 1] = *reinterpret_cast<const f32x4*>(sp + s * plane);
#pragma unroll
      for (int s = 1; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += w[s - 1][j];
    } else {
      for (int s = 1; s < nsplit; ++s) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sp + s * plane);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += w[j];
      }
    }
    if (p.bias) {
      Pack4<T> b;
      b.u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.bias) + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += to_f(b.e[j]);
    }
    if (p.rowvec) {
      Pack4<T> b;
      b.u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.rowvec) + (int64_t)(m / p.rows_per_vec) * (p.ld_rowvec ? p.ld_rowvec : (int64_t)p.N) + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += to_f(b.e[j]);
    }
    if (p.lora_r > 0 && m >= p.lora_row0) {
      // (lora_seg is a multiple of 4: the thread's four columns share one segment, hence one xa row)
      const float* xr = p.lora_xa + (int64_t)(m - p.lora_row0) * p.ld_xa + (p.lora_seg ? (n / p.lora_seg) * p.lora_r : 0);
      const float* up0 = p.lora_up + (int64_t)n * p.up_sn;
      // ranks 4 / 8 unrolled: every operand of the delta is requested before the first fmaf (the rolled loop with its
      // run-time strides made 2 r dependent round trips per column: +18 us on a 2048 x 1280 launch).  Same canonical chain.
      auto delta = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        float x[R], u[4][R];
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = xr[r];
        const bool al = (reinterpret_cast<uintptr_t>(p.lora_up) & 15) == 0;
        if (al && p.up_sq == 1 && p.up_sn == R) {  // [N, r] rows: 16-byte pieces (a lane's dword loads at a 4 r-byte
          // stride touch 64 lines per wave-instruction: 16 of them per thread made the delta cost 17 of the finish's 24 us)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < R / 4; ++q) {
              const f32x4 t = *reinterpret_cast<const f32x4*>(up0 + j * R + q * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) u[j][q * 4 + e] = t[e];
            }
        } else if (al && p.up_sn == 1 && (p.up_sq & 3) == 0) {  // [r, K] read transposed: four columns of one rank row
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(up0 + (int64_t)r * p.up_sq);
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j][r] = t[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < R; ++r) u[j][r] = up0[(int64_t)j * p.up_sn + (int64_t)r * p.up_sq];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float d = 0.f;
#pragma unroll
          for (int r = 0; r < R; ++r) d = __builtin_fmaf(x[r], u[j][r], d);
          v[j] = __builtin_fmaf(d, p.lora_scale, v[j]);
        }
      };
      if (p.lora_r == 4) delta(std::integral_constant<int, 4>{});
      else if (p.lora_r == 8) delta(std::integral_constant<int, 8>{});
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* up = up0 + (int64_t)j * p.up_sn;
          float d = 0.f;
          for (int r = 0; r < p.lora_r; ++r) d = __builtin_fmaf(xr[r], up[r * p.up_sq], d);
          v[j] = __builtin_fmaf(d, p.lora_scale, v[j]);
        }
      }
    }
    if (p.res) {
      Pack4<T> b;
      b.u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.res) + (int64_t)m * p.ldr + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += to_f(b.e[j]);
    }
    if (p.out_f32) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = v;
    } else {
      Pack4<T> o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o.e[j] = from_f<T>(v[j]);
      *reinterpret_cast<u32x2*>(reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n) = o.u;
    }
  }
}

}  // namespace

bool gemm2_supported(const GemmParams& p);
bool gemm2_geglu_supported(const GemmParams& p);
int launch_gemm2(const GemmParams& p, int variant, hipStream_t stream);
bool gemm3_supported(const GemmParams& p);
int launch_gemm3(const GemmParams& p, hipStream_t stream);
bool gemm4_supported(const GemmParams& p);
int launch_gemm4(const GemmParams& p, hipStream_t stream);

// SMI_GEMM=v1 forces the register-staged kernel, SMI_GEMM=128 / 256 forces a v2 tile height (A/B experiments)
static int gemm_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("SMI_GEMM");
    mode = !e ? 0 : (!strcmp(e, "v1") ? 3 : (!strcmp(e, "128") ? 1 : (!strcmp(e, "256") ? 2 : (!strcmp(e, "convv1") ? 4 : (!strcmp(e, "64") ? 5 : (!strcmp(e, "8ph") ? 6 : (!strcmp(e, "no8ph") ? 7 : (!strcmp(e, "160") ? 8 : (!strcmp(e, "8w") ? 9 : (!strcmp(e, "no8w") ? 10 : (!strcmp(e, "64w") ? 11 : (!strcmp(e, "5ph") ? 12 : (!strcmp(e, "no5ph") ? 13 : (!strcmp(e, "160w") ? 14 : (!strcmp(e, "64x160") ? 15 : 0)))))))))))))));
  }
  return mode;
}

// shapes where the 256x256 tile is the better choice (measured, tools/bench_gemm.py)
// (one 256x256 workgroup per CU: it needs >= 2 full rounds of tiles to amortise its longer fill / epilogue; below that
// the 128-row v2 tiles with 2-3 resident workgroups per CU win)
// convs: only where Cout fills whole 256-column tiles (320 / 640 output channels lose 17-38 % of a tile row)
static bool gemm3_wanted(const GemmParams& p) {
  if (p.conv && p.N % 256 != 0) return false;
  return (int64_t)cdiv(p.M, 256) * cdiv(p.N, 256) >= 512;  // (N = 1280 convs at 16384 rows: 320 tiles -> gemm2 128x160)
}

bool gemm_geglu_supported(const GemmParams& p) { return gemm_mode() != 3 && gemm2_geglu_supported(p); }

// ---------------------------------------------------------------------------------------------------------------
// Tile autotuning.  All kernel generations / tile layouts give bit-identical results (same K order, same epilogue
// arithmetic: tests/test_fullsize_gpu.py), so which one runs is purely a speed question, and the heuristics above were
// fitted to a handful of SD-XL shapes.  The first time a (shape, epilogue) key is launched the candidates its layout
// allows are timed with HIP events on the launch stream (2 launches each after one warm-up; this synchronises with
// the host, once per key, during the first step) and the winner is cached for the life of the process.  A candidate
// must beat the heuristic choice by 3 % to replace it.  Tuning launches are harmless: every launch fully rewrites
// its outputs from read-only inputs, except in-place accumulation (res == C), which is timed into a scratch output.
// SMI_GEMM_TUNE=0 turns it off; any SMI_GEMM override does too.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct TuneKey {
  int v[18];
  bool operator==(const TuneKey& o) const { return memcmp(v, o.v, sizeof(v)) == 0; }
};
struct TuneHash {
  size_t operator()(const TuneKey& k) const {
    size_t h = 1469598103934665603ull;
    for (int x : k.v) h = (h ^ (size_t)(unsigned)x) * 1099511628211ull;
    return h;
  }
};
// SMI_TUNE_FILE=<path>: persist the tuner's decisions.  Lines of 18 key integers + the chosen candidate are loaded when
// the cache is first touched and appended whenever a key is tuned, so a second process (a rocprofv3 run of the same
// workload) launches no timing candidates at all: its kernel trace is the steady state from the first step.
const char* tune_file() {
  static const char* f = getenv("SMI_TUNE_FILE");
  return (f && f[0]) ? f : nullptr;
}
std::unordered_map<TuneKey, int, TuneHash>& tune_cache() {
  static std::unordered_map<TuneKey, int, TuneHash> c;
  static bool loaded = false;
  if (!loaded) {
    loaded = true;
    if (const char* path = tune_file()) {
      if (FILE* fp = fopen(path, "r")) {
        for (;;) {
          TuneKey k;
          int choice = 0, got = 0;
          for (int i = 0; i < 18; ++i) got += fscanf(fp, "%d", &k.v[i]) == 1;
          got += fscanf(fp, "%d", &choice) == 1;
          if (got != 19) break;
          c[k] = choice;
        }
        fclose(fp);
      }
    }
  }
  return c;
}
void tune_persist(const TuneKey& k, int choice) {
  const char* path = tune_file();
  if (!path) return;
  if (FILE* fp = fopen(path, "a")) {
    for (int i = 0; i < 18; ++i) fprintf(fp, "%d ", k.v[i]);
    fprintf(fp, "%d\n", choice);
    fclose(fp);
  }
}
bool tune_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("SMI_GEMM_TUNE");
    on = (e && !strcmp(e, "0")) ? 0 : 1;
  }
  return on == 1;
}
// candidate codes: 0 = heuristic, 1 = v2 128x128 (8 waves), 2 = v2 256x128, 4 = v2 128x160, 7 = v2 64x128, 10 = v2
// 128x160 with 8 waves, 12 = v2 64x160,
// 100 = v3 8-phase, 200 = v4 256x320 persistent
int launch_candidate(const GemmParams& p, int cand, hipStream_t stream) {
  if (cand == 200 && gemm4_supported(p)) return launch_gemm4(p, stream);
  if (cand == 100 && gemm3_supported(p)) return launch_gemm3(p, stream);  // (layout re-checked: the key is shape-only)
  if (cand == 0 || cand == 100 || cand == 200) {
    if (gemm3_wanted(p) && gemm3_supported(p)) return launch_gemm3(p, stream);
    return launch_gemm2(p, 0, stream);
  }
  return launch_gemm2(p, cand, stream);
}
int tuned_choice(const GemmParams& p, hipStream_t stream) {
  TuneKey key;
  const int kv[18] = {(int)p.dtype, (int)p.conv, p.M, p.N, p.K, (int)(p.conv ? p.Cin : p.lda),
                      (int)(p.stride * 4 + p.upsample * 2 + p.transposed), (int)(p.res != nullptr), p.lora_r,
                      (int)(p.geglu_out != nullptr), (int)p.out_f32, (int)(p.bias != nullptr),
                      (int)(p.rowvec != nullptr), (int)(p.conv ? p.Hin * 65536 + p.Win : p.ldc),
                      (int)(p.res ? p.ldr : 0), p.lora_seg, p.pad, (int)(p.res == p.C) + 2 * (int)(p.lora_r > 0 && p.up_sn == 1)};
  memcpy(key.v, kv, sizeof(kv));
  // one tuner at a time (several engines / host threads may share the process-wide cache); the timing launches run on
  // the caller's stream -- since the engine dropped its second stream nothing else executes beside them
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  auto& cache = tune_cache();
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int cands[14], nc = 0;
  cands[nc++] = 0;
  const bool plain_conv = p.conv && p.stride == 1 && !p.upsample && !p.transposed;
  if (!p.conv || plain_conv) {
    cands[nc++] = 1;
    if ((int64_t)cdiv(p.M, 256) * cdiv(p.N, 128) >= 128) cands[nc++] = 2;
    if (p.N % 160 == 0 && (!p.geglu_out || p.N % 320 == 0)) cands[nc++] = 4;
    if (gemm3_supported(p) && (int64_t)cdiv(p.M, 256) * cdiv(p.N, 256) >= 192) cands[nc++] = 100;
    if ((int64_t)cdiv(p.M, 128) * cdiv(p.N, 128) <= 512) cands[nc++] = 7;  // 64 x 128 tiles for small grids
    // 64 x 160 tiles (4 waves of 32 x 80) where 128 x 160 tiles leave CUs idle
    if (!p.geglu_out && p.N % 160 == 0 && (int64_t)cdiv(p.M, 128) * (p.N / 160) <= 256) cands[nc++] = 12;
    if (!p.geglu_out && p.N % 160 == 0 && p.K >= 1280 && (int64_t)cdiv(p.M, 128) * (p.N / 160) <= 128) cands[nc++] = 11;
    if (gemm4_supported(p) && (int64_t)(p.M / 256) * (p.N / 320) >= 128) cands[nc++] = 200;
    // about one 128 x 160 tile per CU: the eight-wave form of that tile
    if (!p.geglu_out && p.N % 160 == 0 && p.N % 8 == 0 && (int64_t)cdiv(p.M, 128) * (p.N / 160) <= 512) cands[nc++] = 10;
  } else if (p.conv && p.upsample && gemm4_supported(p) && (int64_t)(p.M / 256) * (p.N / 320) >= 128) {
    cands[nc++] = 200;  // up-sampler convs: gemm2's general gather or gemm4's row-aligned one
  }
  int best = 0;
  if (nc > 1) {
    GemmParams q = p;
    void* tmp = nullptr;
    if (q.res == q.C) {  // in-place accumulation is not repeatable: the tuning launches write to a scratch output
      const size_t bytes = ((size_t)(p.M - 1) * p.ldc + p.N) * (p.out_f32 ? 4 : 2);
      if (hipMalloc(&tmp, bytes) != hipSuccess) {
        (void)hipGetLastError();
        cache[key] = 0;
        return 0;
      }
      q.C = tmp;
    }
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      (void)hipGetLastError();
      if (tmp) (void)hipFree(tmp);
      cache[key] = 0;
      return 0;
    }
    float tbest = 0.f, t0 = 0.f;
    static const bool dump = getenv("SMI_TUNE_DUMP") != nullptr;  // candidate timings of every tuned key on stderr
    if (dump) fprintf(stderr, "[smi tune]");
    for (int i = 0; i < nc; ++i) {
      if (launch_candidate(q, cands[i], stream) != 0) continue;  // warm-up (also sets the LDS attribute once)
      (void)hipEventRecord(a, stream);
      (void)launch_candidate(q, cands[i], stream);
      (void)launch_candidate(q, cands[i], stream);
      (void)hipEventRecord(b, stream);
      if (hipEventSynchronize(b) != hipSuccess) break;
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, a, b);
      if (dump) fprintf(stderr, " %d:%.1fus", cands[i], ms * 500.f);
      if (i == 0) { t0 = tbest = ms; continue; }
      if (ms < 0.97f * t0 && ms < tbest) { tbest = ms; best = cands[i]; }
    }
    if (dump)
      fprintf(stderr, "  -> %d   [%s M=%d N=%d K=%d%s%s%s%s]\n", best, p.conv ? "conv" : "gemm", p.M, p.N, p.K,
              p.bias ? " bias" : "", p.res ? " res" : "", p.lora_r ? " lora" : "", p.geglu_out ? " geglu" : "");
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (tmp) {
      (void)hipStreamSynchronize(stream);
      (void)hipFree(tmp);
    }
    (void)hipGetLastError();
  }
  cache[key] = best;
  tune_persist(key, best);
  return best;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Split-K by RULE for grids that leave most CUs idle behind a long K loop: SD-1.x at batch 1-4 (conv 128 x 1280 x 11520:
// ten 128 x 128 tiles on 256 CUs, 180 K-tiles each), the deepest UNet level of the pre-roll, the time-embedding MLPs.
// The slice count is a function of the shape alone -- never of a timing -- so results stay reproducible run to run, and
// the rule sits in front of every generation override: each slice accumulates its K-tiles in ascending order on the
// gemm2 kernels (bit-identical among themselves) and one finish kernel adds the slices in order and applies the epilogue,
// so the SMI_GEMM overrides still agree bit for bit with the default selection.
//   tiles = 128 x 128 output tiles, nk = 64-deep K-tiles:  tiles <= 256, nk >= 16 (nk >= 32 from 96 tiles on: round 4, below)
//   S = largest power of two <= min(16, 512 / tiles, nk / 4), limited by the scratch the caller provides
//   (measured 160 / 384 against 256 / 512: SD-1.4 step 24.5 -> 24.2 ms, SD-1.5 B = 4 59.6 -> 59.1, headline 178.9 -> 178.2)
// SMI_GEMM_SPLITK=0 turns the rule off.
// NOTE (batch dependence): `tiles` counts the launch's rows, so the SAME sample's product is summed over K in S(batch)
// slices -- its fp32 summation order changes with the batch it travels in (never from run to run).  Everything that is
// compared bitwise runs both sides at one batch (batched pass vs separate passes at equal per-launch M of the split
// shapes, dedup vs full step, tile-selection digests); comparisons across batch sizes (VAE pair vs single, 2 ranks vs 1)
// carry a tolerance.
// ---------------------------------------------------------------------------------------------------------------
namespace {
thread_local void* t_scratch = nullptr;
thread_local size_t t_scratch_bytes = 0;
int splitk_slices(const GemmParams& p) {
  static const bool on = []() { const char* e = getenv("SMI_GEMM_SPLITK"); return !(e && !strcmp(e, "0")); }();
  if (!on || !t_scratch || p.geglu_out || p.ksplit || !gemm2_supported(p) || p.N % 4 != 0) return 1;
  const int64_t tiles = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128);
  const int nk = cdiv(p.K, 64);
  static const int max_tiles = []() { const char* e = getenv("SMI_SPLITK_MAXTILES"); return e ? atoi(e) : 256; }();
  if (tiles > max_tiles || nk < 16) return 1;
  // K = 1024-1984 (16-31 K-tiles) splits only while the un-split launch's 64-row tiles would leave the chip under-filled
  // (fewer than 96 of the 128 x 128 tiles, i.e. fewer than 192 workgroups).  Measured inside the pre-roll passes
  // (SMI_BENCH_PROFILE_PREROLL, HIP events): 2048 x 1280 x 1280 (160 tiles) 26.1 us un-split vs 29.5 split in two,
  // 512 x 5120 x 1280 (160) 24.5 vs 28.5, 8192 x 320 x 1280 (192) 28.4 vs 30.2, 512 x 3840 x 1280 (120) 26.3 vs 27.6 -- but
  // 512 x 1280 x 1280 (40) 23-25 vs 18-20, 128 x 3840 x 1280 (30) 25.9 vs 17.3.  (Back to back with hot operands the
  // un-split form wins at 40 tiles too, tools/bench_splitk.py: cold weights want the workgroups.)
  static const int few_tiles = []() { const char* e = getenv("SMI_SPLITK_FEWTILES"); return e ? atoi(e) : 95; }();
  static const int long_k = []() { const char* e = getenv("SMI_SPLITK_LONGK"); return e ? atoi(e) : 32; }();
  if (nk < long_k && tiles > few_tiles) return 1;
  int64_t lim = 16;
  const int64_t budget = 512;  // workgroup-tiles per launch after the split (two per CU)
  if (nk / 4 < lim) lim = nk / 4;
  const int64_t fit = (int64_t)(t_scratch_bytes / ((size_t)p.M * p.N * sizeof(float)));
  if (fit < lim) lim = fit;
  static const bool pow2 = []() { const char* e = getenv("SMI_SPLITK_POW2"); return e && e[0] == '1'; }();
  if (pow2) {  // rounds 3: the largest power of two that keeps the 128 x 128 tiles within the budget
    if (budget / tiles < lim) lim = budget / tiles;
    int s = 1;
    while (2 * s <= lim) s *= 2;
    return s;
  }
  // Any slice count, chosen to FILL the budget with the slice kernel launch_splitk will take: 64 x 128 tiles when two or
  // more slices of them fit, else 128 x 128.  2048 x 1280 x 5120 (160 tiles): two slices were 320 workgroups on 256 CUs
  // (50.7 us), three are 480 (43.7); 2048 x 640 x 2560: four slices of 128 x 128 (320 workgroups, 22.9 us) -> three of
  // 64 x 128 (480, 21.0); 1024 x 1280 x 5120: 31.2 -> 29.8 (tools/bench_splitk.py q).
  const int64_t tiles64 = (int64_t)cdiv(p.M, 64) * cdiv(p.N, 128);
  int64_t s = budget / tiles64 < lim ? budget / tiles64 : lim;
  if (s < 2) s = budget / tiles < lim ? budget / tiles : lim;
  return s < 1 ? 1 : (int)s;
}
int launch_splitk(const GemmParams& p, int S, hipStream_t stream, int variant = -1) {
  if (p.conv) {
    SMI_CHECK(p.K == 9 * p.Cin && p.M == p.Nb * p.Hout * p.Wout && (p.stride == 1 || p.stride == 2),
              "conv: inconsistent geometry");
  }
  GemmParams q = p;  // the slices: plain accumulation into fp32 slabs [S][M][N]
  q.C = t_scratch;
  q.ldc = p.N;
  q.out_f32 = 1;
  q.bias = q.res = q.rowvec = nullptr;
  q.lora_r = 0;
  q.lora_xa = q.lora_up = nullptr;
  q.ksplit = S;
  // 64 x 128 tiles (4 waves) while that still leaves CUs idle, else the 128 x 128 eight-wave tile.  (SMI_SPLITK_T160=1: the
  // 128 x 160 eight-wave tile where N allows -- back to back it wins, 2048 x 1280 x 5120 in two slices 46.5 vs 47.3 us, inside
  // a pass it loses: 61.4 vs 58.2 us there, SD-1.4 pre-roll 174.0 vs 172.3 ms.)
  const int64_t wg64 = (int64_t)cdiv(p.M, 64) * cdiv(p.N, 128) * S;
  static const bool t160 = []() { const char* e = getenv("SMI_SPLITK_T160"); return e && e[0] == '1'; }();
  const int v = wg64 <= 512 ? 7 : (t160 && p.N % 160 == 0 ? 10 : 5);
  if (launch_gemm2(q, variant >= 0 ? variant : v, stream) != 0) return -1;
  const int64_t total = (int64_t)p.M * (p.N / 4);
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
#define FIN(TT_, NS_) hipLaunchKernelGGL((splitk_finish_kernel<TT_, NS_>), dim3(grid), dim3(256), 0, stream, p, (const float*)t_scratch, S)
#define FIN_T(TT_)                                                                                              \
  do {                                                                                                          \
    switch (S) {                                                                                                \
      case 2: FIN(TT_, 2); break;   case 3: FIN(TT_, 3); break;   case 4: FIN(TT_, 4); break;                   \
      case 5: FIN(TT_, 5); break;   case 6: FIN(TT_, 6); break;   case 7: FIN(TT_, 7); break;                   \
      case 8: FIN(TT_, 8); break;   case 9: FIN(TT_, 9); break;   case 10: FIN(TT_, 10); break;                 \
      case 11: FIN(TT_, 11); break; case 12: FIN(TT_, 12); break; case 13: FIN(TT_, 13); break;                 \
      case 14: FIN(TT_, 14); break; case 15: FIN(TT_, 15); break; case 16: FIN(TT_, 16); break;                 \
      default: FIN(TT_, 0);                                                                                     \
    }                                                                                                           \
  } while (0)
  if (p.dtype == DT_F16) FIN_T(f16); else FIN_T(bf16);
#undef FIN_T
#undef FIN
  SMI_HIP(hipGetLastError());
  return 0;
}
}  // namespace

void set_gemm_scratch(void* ws, size_t bytes) {
  t_scratch = ws;
  t_scratch_bytes = ws ? bytes : 0;
}

int launch_gemm(const GemmParams& p, hipStream_t stream) {
  SMI_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "gemm: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
  SMI_CHECK(!p.geglu_out || gemm_geglu_supported(p), "gemm: fused GEGLU not available for this shape/layout");
  // SMI_SPLITK_DEBUG (tools/bench_splitk.py): slice count and slice-kernel variant of THIS call from SMI_SPLITK_S / _V
  static const bool splitk_debug = getenv("SMI_SPLITK_DEBUG") != nullptr;
  if (splitk_debug && gemm2_supported(p) && !p.geglu_out) {
    const char* es = getenv("SMI_SPLITK_S");
    const char* ev = getenv("SMI_SPLITK_V");
    const int S = es ? atoi(es) : 0, V = ev ? atoi(ev) : 0;
    if (S > 1 && t_scratch && (size_t)S * p.M * p.N * sizeof(float) <= t_scratch_bytes) return launch_splitk(p, S, stream, V);
    if (S == 1) return launch_gemm2(p, V, stream);
  }
  if (gemm_mode() != 3) {
    const int S = splitk_slices(p);
    if (S > 1) return launch_splitk(p, S, stream);
  }
  // One 128 x 160 tile per CU with a long K (the 4096-row backward GEMMs and convs): the eight-wave tile with the
  // four-stage deep-prefetch loop, by RULE and not by tuning -- the tuner times candidates back to back with their
  // operands hot in L2 / the Infinity Cache, where prefetch depth buys nothing (121 vs 120 us at 4096 x 1280 x 10240);
  // inside the backward pass the operands come from HBM and the deep form wins (153 -> 129 us, conv K = 11520:
  // 193 -> 160 us; measured in the step, A/B on one device).  Bit-identical to every other form.  SMI_GEMM_DEEP=0 turns
  // the rule off, SMI_GEMM_DEEP_K moves the K threshold.
  if (gemm_mode() == 0) {
    // grids of at most `small_grid` 128 x 160 tiles (half the CUs idle: SD-1.x at batch 1-2, the deepest UNet level) are
    // left to the tuner, which can give them 64-row tiles (twice the workgroups) -- or this same deep form (candidate 11)
    static int deep = -1, deep_k = 1280, small_grid = 128;
    if (deep < 0) {
      const char* e = getenv("SMI_GEMM_DEEP");
      deep = (e && !strcmp(e, "0")) ? 0 : 1;
      if (const char* k = getenv("SMI_GEMM_DEEP_K")) deep_k = atoi(k);
      if (const char* k = getenv("SMI_GEMM_SMALL_GRID")) small_grid = atoi(k);
    }
    if (deep && gemm2_supported(p) && !p.geglu_out && p.N % 160 == 0 && p.K >= deep_k &&
        (int64_t)cdiv(p.M, 128) * (p.N / 160) <= 320 && (int64_t)cdiv(p.M, 128) * (p.N / 160) > small_grid &&
        (!p.conv || (p.stride == 1 && !p.upsample && !p.transposed))) {
      if (p.conv) {
        SMI_CHECK(p.K == 9 * p.Cin && p.M == p.Nb * p.Hout * p.Wout, "conv: inconsistent geometry");
      }
      static const int deep_variant = []() { const char* e = getenv("SMI_GEMM_DEEP_VARIANT"); return e ? atoi(e) : 11; }();
      return launch_gemm2(p, deep_variant, stream);  // (11 = eight-wave 128 x 160 deep loop; others for A/B experiments)
    }
  }
  if (gemm_mode() == 0 && tune_enabled() && gemm2_supported(p) && (int64_t)p.M * p.N >= (1 << 17)) {
    if (p.conv) {
      SMI_CHECK(p.K == 9 * p.Cin && p.M == p.Nb * p.Hout * p.Wout && (p.stride == 1 || p.stride == 2),
                "conv: inconsistent geometry");
    }
    return launch_candidate(p, tuned_choice(p, stream), stream);
  }
  // v2 (LDS-DMA staging, full-row epilogue) serves dense GEMMs and convs; SMI_GEMM=convv1 keeps convs on v1, =v1 all
  // v3 (256x256 tile, 8-phase schedule): SMI_GEMM=8ph forces it wherever its layout rules hold, =no8ph disables it
  // v4 (256x320 tile, persistent): SMI_GEMM=5ph forces it wherever its layout rules hold
  if (gemm_mode() == 12 && gemm4_supported(p)) return launch_gemm4(p, stream);
  if (gemm_mode() == 14 && gemm2_supported(p) && !p.geglu_out && p.N % 160 == 0) return launch_gemm2(p, 10, stream);
  if (gemm_mode() == 15 && gemm2_supported(p) && !p.geglu_out && p.N % 160 == 0) return launch_gemm2(p, 12, stream);
  if ((gemm_mode() == 6 || ((gemm_mode() == 0 || gemm_mode() == 10) && gemm3_wanted(p))) && gemm3_supported(p))
    return launch_gemm3(p, stream);
  if (gemm_mode() != 3 && gemm2_supported(p) && (!p.conv || gemm_mode() != 4)) {
    if (p.conv) {
      SMI_CHECK(p.K == 9 * p.Cin && p.M == p.Nb * p.Hout * p.Wout && (p.stride == 1 || p.stride == 2),
                "conv: inconsistent geometry");
    }
    return launch_gemm2(p, gemm_mode() == 11 ? 7 : gemm_mode() == 10 ? 6 : gemm_mode() == 9 ? 5 : gemm_mode() == 8 ? 4 : ((gemm_mode() == 4 || gemm_mode() >= 6) ? 0 : (gemm_mode() == 5 ? 3 : gemm_mode())), stream);
  }
  SMI_CHECK(p.K % 8 == 0 && p.N % 4 == 0, "gemm: K %% 8 and N %% 4 must be 0 (K=%d N=%d)", p.K, p.N);
  SMI_CHECK(p.ldc % 4 == 0, "gemm: ldc %% 4 != 0");
  SMI_CHECK(p.lora_seg % 4 == 0, "gemm: lora_seg %% 4 != 0");
  SMI_CHECK(p.res == nullptr || p.ldr % 4 == 0, "gemm: ldr %% 4 != 0");
  if (p.conv) {
    SMI_CHECK(p.Cin % 64 == 0 && p.K == 9 * p.Cin, "conv: Cin %% 64 != 0 or K != 9*Cin (Cin=%d K=%d)", p.Cin, p.K);
    SMI_CHECK(p.M == p.Nb * p.Hout * p.Wout, "conv: M != Nb*Hout*Wout");
    SMI_CHECK((int64_t)p.Nb * p.Hin * p.Win * p.Cin * 2 < 0xFFFFFFF0ll, "conv: input larger than 4 GiB");
    SMI_CHECK(p.stride == 1 || p.stride == 2, "conv: stride");
  } else {
    SMI_CHECK(p.lda % 8 == 0, "gemm: lda %% 8 != 0");
    SMI_CHECK(((int64_t)(p.M - 1) * p.lda + p.K) * 2 < 0xFFFFFFF0ll, "gemm: A larger than 4 GiB");
  }
  SMI_CHECK((int64_t)p.N * p.K * 2 < 0xFFFFFFF0ll, "gemm: W larger than 4 GiB");
  const int grid = cdiv(p.M, BM) * cdiv(p.N, BN);
  if (p.dtype == DT_F16) {
    if (p.conv)
      hipLaunchKernelGGL((gemm_nt_kernel<f16, true>), dim3(grid), dim3(256), 0, stream, p);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<f16, false>), dim3(grid), dim3(256), 0, stream, p);
  } else {
    if (p.conv)
      hipLaunchKernelGGL((gemm_nt_kernel<bf16, true>), dim3(grid), dim3(256), 0, stream, p);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<bf16, false>), dim3(grid), dim3(256), 0, stream, p);
  }
  SMI_HIP(hipGetLastError());
  return 0;
}

int launch_conv3x3_small(int dtype, const void* in, const void* w, const void* bias, void* out, int out_f32, int Nb,
                         int H, int W, int Cin, int Cout, hipStream_t stream) {
  const int64_t total = (int64_t)Nb * H * W * ((Cout + 3) / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == DT_F16)
    hipLaunchKernelGGL(conv3x3_small_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)in, (const f16*)w,
                       (const f16*)bias, out, out_f32, Nb, H, W, Cin, Cout);
  else
    hipLaunchKernelGGL(conv3x3_small_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)in,
                       (const bf16*)w, (const bf16*)bias, out, out_f32, Nb, H, W, Cin, Cout);
  SMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace smi
