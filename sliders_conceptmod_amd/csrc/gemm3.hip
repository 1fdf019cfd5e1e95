// MFMA GEMM, third generation: 256x256 tile, 8 waves, four phases per 64-deep K-tile, LDS-DMA kept in flight across
// raw barriers (counted vmcnt) -- the deep-pipelined structure of cdna_hip_programming.md section 5 ("256^2 8-phase
// template"), written for this engine's operands and epilogues.  gemm2.hip's loop (one barrier per K-step, drain
// vmcnt(0) each step) tops out at ~850-950 TF/s; its waves spend ~43 % of their cycles parked at that drain.
//
// Geometry.  Block tile 256 x 256 x 64, 8 waves as 2 (M) x 4 (N), per-wave output 128 x 64 = acc[4 n-frags][8 m-frags]
// (128 accumulator registers).  Per K-tile a wave runs 4 phases of 16 MFMAs (one 64 x 32 quadrant x K = 64):
//     ph1  read A-sub0 (8 ds_read_b128) + B-sub0 (4)   mma (A0, B0)
//     ph2  read B-sub1 (4)                              mma (A0, B1)
//     ph3  read A-sub1 (8)                              mma (A1, B1)
//     ph4  --                                           mma (A1, B0)
// and every phase also issues the LDS-DMA of ONE 16-KiB half-tile (2 x global_load_lds_dwordx4 per lane) of a later
// K-tile.  LDS holds 2 K-tiles x 4 half-tiles; half-tile A-h{s} collects sub-tile s of BOTH wave rows (B-h{s}: of all
// four wave columns), so each half-tile is read in exactly one phase and can be restaged soon after:
//     ph1(t): stage A-h1 of tile t+1      (last read ph3(t-1))
//     ph2(t): stage B-h0 of tile t+2      (read ph1(t); those 4 reads are retired by the lgkmcnt(8) before ph1's barrier)
//     ph3(t): stage A-h0 of tile t+2      (read ph1(t))
//     ph4(t): stage B-h1 of tile t+2      (read ph2(t));  then s_waitcnt vmcnt(6): everything but these last three
//             half-tiles has landed, i.e. tile t+1 is complete; it is first read one phase (two barriers) later.
// The two wave rows run staggered by one barrier (wave row 1 executes one extra s_barrier up front, wave row 0 one at
// the end): between two barriers one group issues its loads while the other group -- its SIMD partner -- runs MFMAs.
//
// Rows beyond M / N are clamped to the last valid row (their results are never stored), so the source of every DMA is
// a per-lane pointer that just advances 128 bytes per K-tile.  Requires K % 64 == 0 (all UNet layers); others use v2.
// CONV = implicit-GEMM 3x3 / stride 1 / pad 1: K runs over (tap, channel); an A row is a pixel, its source for a K-tile
// is the centre-tap pointer + a wave-uniform tap offset, or a zero page when the tap falls outside the image.
#include "kernels.h"

#include <type_traits>

namespace smi {
namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 128 * BK * 2;  // one half-tile: 128 rows x 128 B
constexpr int BUF = 4 * HALF;       // A-h0 | A-h1 | B-h0 | B-h1
constexpr int OLD = BN + 8;         // staged output row length (elements)
constexpr int SMEM3 = (BM * OLD * 2 > 2 * BUF) ? BM * OLD * 2 : 2 * BUF;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_dst, 16, 0, 0);
}

#define SMI_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __attribute__((aligned(256))) unsigned char g_zero_page3[256];  // zero-initialised (conv padding source)

template <typename T, bool CONV>
__global__ __launch_bounds__(512) void gemm_8ph_kernel(GemmParams p) {
  typedef typename TT<T>::v8 v8;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int nbn = (p.N + BN - 1) / BN;
  const int nbm = (p.M + BM - 1) / BM;
  int wg;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  // grouped rasterisation (see gemm2.hip): the 32 tiles co-resident on an XCD form a ~(32/GW) x GW patch
  const int ngrp = (nbn + 7) / 8;
  const int GW = (nbn + ngrp - 1) / ngrp;
  const int grp = wg / (GW * nbm);
  const int gw = min(nbn - grp * GW, GW);
  const int lw = wg - grp * GW * nbm;
  const int bm0 = (lw / gw) * BM;
  const int bn0 = (grp * GW + lw % gw) * BN;

  const bool geglu = p.geglu_out != nullptr;
  const int nhalf = p.N >> 1;
  auto gcol = [&](int nl) { return geglu ? (nl < 128 ? (bn0 >> 1) + nl : nhalf + (bn0 >> 1) + nl - 128) : bn0 + nl; };

  // ---- LDS-DMA sources.  Half-tile instruction j of wave w covers LDS rows r = 16 w + 8 j + (lane >> 3); the lane's
  //      16-byte slot (lane & 7) holds source chunk slot ^ (r & 7)  (the read side applies the same XOR).
  const int lrow = lane >> 3, lslot = lane & 7;
  const T* aS[2][2];   // dense: running source pointer; conv: centre-tap pointer of the row's pixel
  int amask[2][2];     // conv: bit (3 ky + kx) set when that tap of the row's pixel lies inside the image
  const T* wS[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wave * 16 + j * 8 + lrow;
      const int chunk = lslot ^ lrow;
      // A-h{h}: LDS row r = 64 wr' + local  <-  tile row 128 wr' + 64 h + local
      const int mrow = bm0 + (r >> 6) * 128 + h * 64 + (r & 63);
      if (CONV) {  // 3x3, stride 1, pad 1 (the other gather modes stay on gemm2.hip): Hout == Hin, Wout == Win
        const int m = min(mrow, p.M - 1);
        const int hw = p.Hin * p.Win;
        const int img = m / hw;
        const int rem = m - img * hw;
        const int oy = rem / p.Win, ox = rem - oy * p.Win;
        int mk = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const bool ok = (unsigned)(oy + ky - 1) < (unsigned)p.Hin && (unsigned)(ox + kx - 1) < (unsigned)p.Win;
            mk |= ok ? (1 << (3 * ky + kx)) : 0;
          }
        amask[h][j] = mrow < p.M ? mk : 0;
        aS[h][j] = reinterpret_cast<const T*>(p.A) + (int64_t)m * p.Cin + chunk * 8;
      } else {
        const int m = min(mrow, p.M - 1);
        amask[h][j] = 0;
        aS[h][j] = reinterpret_cast<const T*>(p.A) + (int64_t)m * p.lda + chunk * 8;
      }
      // B-h{h}: LDS row r = 32 wc' + 16 nip + fr  <-  W row (output column) 64 wc' + 32 h + 8 (fr >> 2) + 4 nip + (fr & 3):
      // after the two MFMAs of a pair a lane owns 8 consecutive output columns (16-byte epilogue accesses)
      const int fr_ = r & 15, nip = (r >> 4) & 1;
      const int n = min(gcol((r >> 5) * 64 + 32 * h + 8 * (fr_ >> 2) + 4 * nip + (fr_ & 3)), p.N - 1);
      wS[h][j] = reinterpret_cast<const T*>(p.W) + (int64_t)n * p.K + chunk * 8;
    }
  // conv: K position of the next A half-tile to stage, per half (wave-uniform): channel offset and tap
  int cA[2] = {0, 0}, kyA[2] = {0, 0}, kxA[2] = {0, 0};
  // dst: half-tile base + (2 wave + j) KiB
  auto stage_a = [&](int h, unsigned char* dst) {
    if (CONV) {
      const int toff = ((kyA[h] - 1) * p.Win + (kxA[h] - 1)) * p.Cin + cA[h];
      const int tap = kyA[h] * 3 + kxA[h];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const void* src = ((amask[h][j] >> tap) & 1) ? (const void*)(aS[h][j] + toff) : (const void*)g_zero_page3;
        glds16(src, dst + (wave * 2 + j) * 1024);
      }
      cA[h] += BK;
      if (cA[h] == p.Cin) {
        cA[h] = 0;
        if (++kxA[h] == 3) {
          kxA[h] = 0;
          ++kyA[h];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        glds16(aS[h][j], dst + (wave * 2 + j) * 1024);
        aS[h][j] += BK;
      }
    }
  };
  auto stage_w = [&](int h, unsigned char* dst) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      glds16(wS[h][j], dst + (wave * 2 + j) * 1024);
      wS[h][j] += BK;
    }
  };

  f32x4 acc[4][8];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  // fragment (i, kk) of a half-tile: row = base + 16 i + fr, 16-byte chunk (4 kk + fq) ^ (fr & 7)
  const int swz0 = (fq ^ (fr & 7)) << 4;
  const int aoff0 = (wr * 64 + fr) * 128 + swz0, aoff1 = aoff0 ^ 64;
  const int boff0 = (wc * 32 + fr) * 128 + swz0, boff1 = boff0 ^ 64;

  v8 xa0[2][4], xa1[2][4], wb0[2][2], wb1[2][2];
  auto rdA = [&](const unsigned char* half, v8 (&x)[2][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Pack8<T> t0, t1;
      t0.u = *reinterpret_cast<const u32x4*>(half + aoff0 + i * 2048);
      t1.u = *reinterpret_cast<const u32x4*>(half + aoff1 + i * 2048);
      x[0][i] = t0.v;
      x[1][i] = t1.v;
    }
  };
  auto rdB = [&](const unsigned char* half, v8 (&w)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      Pack8<T> t0, t1;
      t0.u = *reinterpret_cast<const u32x4*>(half + boff0 + i * 2048);
      t1.u = *reinterpret_cast<const u32x4*>(half + boff1 + i * 2048);
      w[0][i] = t0.v;
      w[1][i] = t1.v;
    }
  };
#define SMI_MMA(XA, WB, MH, NH)                                                                        \
  do {                                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                     \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)  \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) acc[2 * (NH) + ni][4 * (MH) + mi] =           \
            TT<T>::mfma16(WB[kk][ni], XA[kk][mi], acc[2 * (NH) + ni][4 * (MH) + mi]);                  \
    __builtin_amdgcn_s_setprio(0);                                                                     \
  } while (0)

  const int nk = p.K / BK;

  // ---- prologue: tile 0 complete, then B-h0 / A-h0 / B-h1 of tile 1 left in flight
  stage_a(0, smem + 0 * HALF);
  stage_a(1, smem + 1 * HALF);
  stage_w(0, smem + 2 * HALF);
  stage_w(1, smem + 3 * HALF);
  if (nk > 1) {
    stage_w(0, smem + BUF + 2 * HALF);
    stage_a(0, smem + BUF + 0 * HALF);
    stage_w(1, smem + BUF + 3 * HALF);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: wave row 1 runs one barrier behind wave row 0

  // one K-tile = 4 phases.  s1: tile t+1 exists (its A-h1 is still to be staged); s2: tile t+2 exists.
  auto tile = [&](auto Bc, const bool s1, const bool s2) {
    constexpr int b = decltype(Bc)::value;
    unsigned char* cur = smem + b * BUF;
    unsigned char* oth = smem + (b ^ 1) * BUF;
    // -------- phase 1
    SMI_FENCE();
    rdB(cur + 2 * HALF, wb0);
    SMI_FENCE();
    rdA(cur + 0 * HALF, xa0);
    SMI_FENCE();
    if (s1) stage_a(1, oth + 1 * HALF);
    SMI_FENCE();
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the 4 B-h0 reads (issued first) have returned
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SMI_FENCE();
    SMI_MMA(xa0, wb0, 0, 0);
    SMI_FENCE();
    __builtin_amdgcn_s_barrier();
    // -------- phase 2
    SMI_FENCE();
    rdB(cur + 3 * HALF, wb1);
    SMI_FENCE();
    if (s2) stage_w(0, cur + 2 * HALF);
    SMI_FENCE();
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SMI_FENCE();
    SMI_MMA(xa0, wb1, 0, 1);
    SMI_FENCE();
    __builtin_amdgcn_s_barrier();
    // -------- phase 3
    SMI_FENCE();
    rdA(cur + 1 * HALF, xa1);
    SMI_FENCE();
    if (s2) stage_a(0, cur + 0 * HALF);
    SMI_FENCE();
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SMI_FENCE();
    SMI_MMA(xa1, wb1, 1, 1);
    SMI_FENCE();
    __builtin_amdgcn_s_barrier();
    // -------- phase 4
    SMI_FENCE();
    if (s2) {
      stage_w(1, cur + 3 * HALF);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // all but the last three half-tiles: tile t+1 has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SMI_FENCE();
    __builtin_amdgcn_s_barrier();
    SMI_FENCE();
    SMI_MMA(xa1, wb0, 1, 0);
    SMI_FENCE();
    __builtin_amdgcn_s_barrier();
  };
  for (int t = 0; t < nk; t += 2) {
    tile(std::integral_constant<int, 0>{}, t + 1 < nk, t + 2 < nk);
    if (t + 1 < nk) tile(std::integral_constant<int, 1>{}, t + 2 < nk, t + 3 < nk);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // re-join the two wave rows
  SMI_FENCE();

  // ---- epilogue (as gemm2.hip): lane holds 8 consecutive columns per (mi, pair q); 16-bit results staged through
  //      LDS and written as whole 512-byte rows
  const bool stage_out = !p.out_f32 && (p.N % 8 == 0);
  T* otile = reinterpret_cast<T*>(smem);
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int ml = wr * 128 + mi * 16 + fr;
    const int m = bm0 + ml;
    if (m >= p.M) continue;
    const bool lora_on = p.lora_r > 0 && m >= p.lora_row0;
    const float* xrow0 = p.lora_xa + (int64_t)(m - p.lora_row0) * p.ld_xa;
    const int64_t vrow = p.rowvec ? (int64_t)(m / p.rows_per_vec) * (p.ld_rowvec ? p.ld_rowvec : (int64_t)p.N) : 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int nl = wc * 64 + q * 32 + fq * 8;
      const int n = gcol(nl);
      if (n >= p.N) continue;
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[2 * q][mi][j];
        v[4 + j] = acc[2 * q + 1][mi][j];
      }
      if (p.bias) {
        Pack8<T> b;
        b.u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bias) + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += to_f(b.e[j]);
      }
      if (p.rowvec) {
        Pack8<T> b;
        b.u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.rowvec) + vrow + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += to_f(b.e[j]);
      }
      if (lora_on) {
        const float* xrow = xrow0 + (p.lora_seg ? (n / p.lora_seg) * p.lora_r : 0);
        if (p.up_sq == 1 && p.up_sn == p.lora_r && (p.lora_r & 3) == 0) {
          float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          for (int r0 = 0; r0 < p.lora_r; r0 += 4) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + r0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const f32x4 uv = *reinterpret_cast<const f32x4*>(p.lora_up + (int64_t)(n + j) * p.lora_r + r0);
              d[j] = lora_fma4(d[j], xv, uv);
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(d[j], p.lora_scale, v[j]);
        } else if (p.up_sn == 1 && (p.up_sq & 3) == 0) {
          float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          for (int r = 0; r < p.lora_r; ++r) {
            const float xq = xrow[r];
            const float* ar = p.lora_up + (int64_t)r * p.up_sq + n;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ar), a1 = *reinterpret_cast<const f32x4*>(ar + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              d[j] = __builtin_fmaf(xq, a0[j], d[j]);
              d[4 + j] = __builtin_fmaf(xq, a1[j], d[4 + j]);
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(d[j], p.lora_scale, v[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float* xr = xrow0 + (p.lora_seg ? ((n + j) / p.lora_seg) * p.lora_r : 0);
            const float* up = p.lora_up + (int64_t)(n + j) * p.up_sn;
            float d = 0.f;
            for (int r = 0; r < p.lora_r; ++r) d = __builtin_fmaf(xr[r], up[r * p.up_sq], d);
            v[j] = __builtin_fmaf(d, p.lora_scale, v[j]);
          }
        }
      }
      if (p.res) {
        Pack8<T> b;
        b.u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.res) + (int64_t)m * p.ldr + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += to_f(b.e[j]);
      }
      if (p.out_f32) {
        float* op = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
        *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        Pack8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = from_f<T>(v[j]);
        *reinterpret_cast<u32x4*>(otile + ml * OLD + nl) = o.u;
      }
    }
  }
  if (stage_out && geglu) {
    __syncthreads();
    {  // hidden * gelu(gate): 16 chunks of 8 output columns per row; a wave writes 4 rows x 256 contiguous bytes
      const int c = tid & 15;
      T* gout = reinterpret_cast<T*>(p.geglu_out);
#pragma unroll
      for (int i = 0; i < BM / 32; ++i) {
        const int r = (tid >> 4) + i * 32;
        const int m = bm0 + r;
        if (m < p.M) {
          Pack8<T> h, g, o;
          h.u = *reinterpret_cast<const u32x4*>(otile + r * OLD + c * 8);
          g.u = *reinterpret_cast<const u32x4*>(otile + r * OLD + 128 + c * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) o.e[e] = from_f<T>(to_f(h.e[e]) * gelu_f(to_f(g.e[e])));
          *reinterpret_cast<u32x4*>(gout + (int64_t)m * nhalf + (bn0 >> 1) + c * 8) = o.u;
        }
      }
    }
    if (bm0 + BM > p.geglu_row0) {  // projection kept only for the rows that will be differentiated
      const int c = tid & 31;
      const int n = gcol(c * 8);
#pragma unroll
      for (int i = 0; i < BM / 16; ++i) {
        const int r = (tid >> 5) + i * 16;
        const int m = bm0 + r;
        if (m < p.M && m >= p.geglu_row0)
          *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n) =
              *reinterpret_cast<const u32x4*>(otile + r * OLD + c * 8);
      }
    }
  } else if (stage_out) {
    __syncthreads();
    const int c = tid & 31;  // 16-byte chunk within the 256-column tile row
    const int n = bn0 + c * 8;
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
      const int r = (tid >> 5) + i * 16;
      const int m = bm0 + r;
      if (m < p.M && n < p.N)
        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n) =
            *reinterpret_cast<const u32x4*>(otile + r * OLD + c * 8);
    }
  }
}

template <typename T, bool CONV>
int launch_t(const GemmParams& p, hipStream_t stream) {
  static DynLdsOnce once;  // (per instantiation, per device)
  if (int rc = once.set((const void*)gemm_8ph_kernel<T, CONV>, SMEM3)) return rc;
  const int grid = cdiv(p.M, BM) * cdiv(p.N, BN);
  hipLaunchKernelGGL((gemm_8ph_kernel<T, CONV>), dim3(grid), dim3(512), SMEM3, stream, p);
  SMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace

bool gemm2_supported(const GemmParams& p);

// dense GEMMs whose layout the 8-phase kernel takes: everything gemm2 takes, plus K % 64 == 0 and N % 8 == 0
bool gemm3_supported(const GemmParams& p) {
  if (!gemm2_supported(p)) return false;
  if (p.conv && (p.stride != 1 || p.upsample || p.transposed || p.Cin % BK != 0 || p.Hout != p.Hin || p.Wout != p.Win ||
                 p.K != 9 * p.Cin || p.M != p.Nb * p.Hout * p.Wout))
    return false;
  if (p.K % BK != 0 || p.N % 8 != 0 || p.lora_seg % 8 != 0) return false;
  if (p.out_f32 && p.ldc % 4 != 0) return false;
  if (p.geglu_out && p.N % 512 != 0) return false;
  return true;
}

int launch_gemm3(const GemmParams& p, hipStream_t stream) {
  if (p.dtype == DT_F16) return p.conv ? launch_t<f16, true>(p, stream) : launch_t<f16, false>(p, stream);
  return p.conv ? launch_t<bf16, true>(p, stream) : launch_t<bf16, false>(p, stream);
}

}  // namespace smi
