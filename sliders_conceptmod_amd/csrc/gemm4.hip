// MFMA GEMM, fourth generation: 256 x 320 tile, persistent, five phases per 64-deep K-tile, ONE LDS-DMA pipeline that
// runs on across output tiles.
//
// Why another tile.  Every channel count of the SD / SD-XL UNets is a multiple of 320 (320 / 640 / 1280 and the fused
// 1920 / 2560 / 3840 / 5120 / 10240 widths) and none of the narrow ones is a multiple of 256: gemm3.hip's 256 x 256 tile
// turns the N = 1280 layers into 64 x 5 = 320 tiles = 1.25 rounds of the 256 CUs (they ran on gemm2.hip's 128 x 160 tile
// at 600-765 TF/s instead), N = 640 into 2.5 column tiles, and q|k|v (N = 3840) into 3.75 rounds.  With BN = 320 the
// level-2 GEMMs of the batched step are exactly 1 (N = 1280), 3 (q|k|v) and 4 (ff.net.2: K = 5120, still 1 round)
// rounds.  What the persistent form adds: a workgroup keeps its DMA pipeline full across the tile boundary (the next
// tile's first two K-tiles are already staged when the epilogue starts), and the epilogue writes straight from the
// accumulator registers (a lane owns 8 consecutive output columns), so it needs neither LDS nor a barrier.
//
// Geometry.  8 waves as 4 (M) x 2 (N); per-wave output 64 x 160 = acc[10 n-frags][4 m-frags] (160 accumulator
// registers).  LDS: two K-tile buffers of A (256 rows x 128 B = 4 units of 8 KiB) + B (320 rows = 5 units); a unit is
// what ONE global_load_lds_dwordx4 of the whole workgroup fills (64 rows).  B unit u holds the u-th pair of 16-column
// fragments of BOTH wave columns, so phase u + 1 reads exactly one B unit:
//     ph1  read B1 (4 ds_read_b128), A (8)    stage B4, A-u3 of tile t+1       16 MFMAs (4 m-frags x pair 0 x K = 64)
//     ph2  read B2                             stage B5 of tile t+1              16 MFMAs
//     ph3  read B3                             stage B1, A-u0 of tile t+2        16 MFMAs
//     ph4  read B4                             stage B2, A-u1 of tile t+2        16 MFMAs
//     ph5  read B5                             stage B3, A-u2 of tile t+2        16 MFMAs;  s_waitcnt vmcnt(6) first
// Every unit is restaged two phases after its last read (cdna guide: "restage a buffer >= 2 phases after its last
// ds_read"); the counted wait in ph5 leaves the three youngest stage groups (6 DMAs) in flight, i.e. K-tile t+1 has
// landed, and t+1 is first read one phase (two barriers) later.  The two wave groups (waves 0-3 / 4-7: SIMD partners)
// run staggered by one barrier as in gemm3.hip; they re-join for the epilogue and re-stagger after it.
// "Tile t+1 / t+2" run on into the next output tile of this workgroup: the source of a stage is
// (current tile, k) or (next tile, k - K).
//
// Requires M % 256 == 0, N % 320 == 0, K % 64 == 0, K >= 128 (everything else stays on gemm2 / gemm3).  Results are
// bit-identical to the other generations: same K order per output element, same epilogue arithmetic.
#include <mutex>

#include <stdlib.h>

#include "kernels.h"

#include <type_traits>

namespace smi {
namespace {

constexpr int BM = 256, BN = 320, BK = 64;
constexpr int UNIT = 64 * BK * 2;        // 8 KiB
constexpr int ABYTES = 4 * UNIT;         // 32 KiB
constexpr int KBUF = ABYTES + 5 * UNIT;  // 72 KiB per K-tile
constexpr int SMEM4 = 2 * KBUF;          // 144 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_dst, 16, 0, 0);
}

#define SMI_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __attribute__((aligned(256))) unsigned char g_zero_page4[256];  // zero-initialised (conv padding source)

// CONV = implicit-GEMM 3x3 / stride 1 / pad 1 (as gemm3.hip): K runs over (tap, channel); an A row is an output pixel,
// its source for a K-tile is the pixel's own address + a wave-uniform tap offset, or the zero page when the tap falls
// outside the image (one 9-bit mask per lane and A unit, for the current and for the next output tile).
// MODE: 0 dense, 1 conv, 2 dense with the fused GEGLU epilogue, 3 conv with the 2x up-sampling folded into the gather
// (separate instantiations: each keeps only its own epilogue,
// which is what keeps the 160 accumulators + epilogue temporaries inside 256 registers)
template <typename T, int MODE>
__global__ __launch_bounds__(512) void gemm_5ph_kernel(GemmParams p) {
  constexpr bool CONV = MODE == 1;
  constexpr bool GEGLU = MODE == 2;
  constexpr bool UPS = MODE == 3;  // 3x3 conv on the nearest-2x up-sampled input (the UNet's up-samplers), Wout % 64 == 0
  typedef typename TT<T>::v8 v8;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1, grp = wave >> 2;

  const int nbn = p.N / BN, nbm = p.M / BM;
  const int ntiles = nbn * nbm;
  const int nk = p.K / BK;
  const int G = gridDim.x;

  // virtual tile id -> tile origin: XCD-contiguous chunks (ids congruent mod 8 share an XCD: G % 8 == 0 or one round),
  // then gemm2's grouped rasterisation inside the chunk
  const int ngrp = (nbn + 7) / 8;
  const int GW = (nbn + ngrp - 1) / ngrp;
  auto tile_origin = [&](int v, int& bm0, int& bn0) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    const int g = wg / (GW * nbm);
    const int gw = min(nbn - g * GW, GW);
    const int lw = wg - g * GW * nbm;
    bm0 = __builtin_amdgcn_readfirstlane((lw / gw) * BM);
    bn0 = __builtin_amdgcn_readfirstlane((g * GW + lw % gw) * BN);
  };

  // ---- LDS-DMA lane constants.  One instruction of the workgroup fills a unit: wave w covers its rows 8 w + (lane >> 3),
  //      the lane's 16-byte slot (lane & 7) holds source chunk slot ^ (row & 7) (the read side applies the same XOR).
  const int lrow = lane >> 3, lslot = lane & 7;
  const int r64 = wave * 8 + lrow;
  const int chunk = lslot ^ lrow;
  // "mix": a batched forward whose last quarter of the rows carries the LoRA delta (3 frozen passes + 1 adapted, the
  // slider step) takes every tile as 192 frozen + 64 adapted rows -- A unit u (= wave row u) is 48 frozen rows + 16
  // adapted ones, so each wave's fourth m-fragment is the adapted one and the delta's epilogue cost is spread evenly
  // over all tiles and waves instead of quadrupling the epilogue of a quarter of the tiles (the launch waits for those).
  const bool mix = !CONV && !UPS && p.lora_r > 0 && p.lora_row0 > 0 && 4 * (int64_t)p.lora_row0 == 3 * (int64_t)p.M;
  const int lda = (CONV || UPS) ? p.Cin : (int)p.lda;  // elements between consecutive A rows (conv: NHWC pixels are contiguous)
  const bool arow = mix && wave >= 6;  // this wave stages the adapted rows of a unit (wave-uniform)
  const uint32_t voffA = (uint32_t)(arow ? r64 - 48 : r64) * (uint32_t)(lda * 2) + chunk * 16;
  // B unit u, LDS row r = 32 wc' + 16 nip + fr  <-  W row (output column) 160 wc' + 32 u + 8 (fr >> 2) + 4 nip + (fr & 3):
  // after the two MFMAs of a pair a lane owns 8 consecutive output columns
  // fused GEGLU (N = 2 x N_half, tile = 160 hidden columns + their 160 gate columns): the pair's first MFMA takes 16
  // hidden columns, the second the SAME 16 gate columns, so a lane ends up with hidden and gate of 4 consecutive
  // columns of one row and forms hidden * gelu(gate) in registers:  row r  <-  W row (nip ? N_half : 0) + 80 wc' + 16 u + fr
  constexpr bool geglu = GEGLU;
  const int nhalf = p.N >> 1;
  const uint32_t voffB =
      (geglu ? (uint32_t)(((r64 >> 4) & 1) * nhalf + (r64 >> 5) * 80 + (r64 & 15))
             : (uint32_t)((r64 >> 5) * 160 + 8 * ((r64 & 15) >> 2) + 4 * ((r64 >> 4) & 1) + (r64 & 3))) *
          (uint32_t)(p.K * 2) +
      chunk * 16;
  const int64_t a_unit = (int64_t)(mix ? (arow ? 16 : 48) : 64) * lda * 2;  // bytes between this wave's rows of A units
  const int64_t w_unit = (int64_t)(geglu ? 16 : 32) * p.K * 2;  // bytes between B units (32 output columns)
  const int ldsw = wave * 1024;                    // this wave's slice of a unit

  // output tile state (wave-uniform)
  int bm0, bn0, bm1 = 0, bn1 = 0;
  int v = blockIdx.x;
  tile_origin(v, bm0, bn0);
  // first row this wave stages for tile origin bm: plain bm; mix: frozen block 3/4 bm or adapted block row0 + bm/4
  auto a_row0 = [&](int bm) { return mix ? (arow ? p.lora_row0 + (bm >> 2) : 3 * (bm >> 2)) : bm; };
  const char* a_cur = reinterpret_cast<const char*>(p.A) + (int64_t)a_row0(bm0) * lda * 2;
  const char* w_cur = reinterpret_cast<const char*>(p.W) + (int64_t)(geglu ? bn0 >> 1 : bn0) * p.K * 2;
  const char* a_nxt = a_cur;
  const char* w_nxt = w_cur;
  bool has_next = false;

  // conv: bit (3 ky + kx) of am[u] is set when that tap of the lane's pixel of unit u lies inside the image
  int am_cur[4] = {0, 0, 0, 0}, am_nxt[4] = {0, 0, 0, 0};
  auto conv_masks = [&](int bm, int (&am)[4]) {
    const int hw = p.Hin * p.Win;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = bm + u * 64 + r64;
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
      am[u] = mk;
    }
  };
  if (CONV) conv_masks(bm0, am_cur);
  const int cpt = (CONV || UPS) ? p.Cin / BK : 1;  // K-tiles per filter tap
  const int cpt_magic = 65536 / cpt + 1;         // tap = (kt * magic) >> 16, exact for kt < 9 * cpt <= 65536 / 9

  // UPS: a unit is 64 consecutive output pixels of ONE output row (Wout % 64 == 0), so the row, the image and the first
  // column are wave-uniform per unit and only the column is per lane: input column = (ox0 + r + kx - 1) >> 1, i.e. the
  // lane offsets depend on kx alone (three constants) and the horizontal border test is two VALU operations.
  int ub_cur[4] = {0, 0, 0, 0}, uy_cur[4] = {0, 0, 0, 0}, ux_cur[4] = {0, 0, 0, 0};
  int ub_nxt[4] = {0, 0, 0, 0}, uy_nxt[4] = {0, 0, 0, 0}, ux_nxt[4] = {0, 0, 0, 0};
  auto ups_units = [&](int bm, int (&ub)[4], int (&uy)[4], int (&ux)[4]) {
    const int hw = p.Hout * p.Wout;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m0 = bm + u * 64;
      const int img = m0 / hw;
      const int rem = m0 - img * hw;
      const int oy = rem / p.Wout;
      ub[u] = __builtin_amdgcn_readfirstlane(img * p.Hin * p.Win);
      uy[u] = __builtin_amdgcn_readfirstlane(oy);
      ux[u] = __builtin_amdgcn_readfirstlane(rem - oy * p.Wout);
    }
  };
  int voffU[3] = {0, 0, 0};
  if (UPS) {
    ups_units(bm0, ub_cur, uy_cur, ux_cur);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) voffU[kx] = ((r64 + kx - 1) >> 1) * (p.Cin * 2) + chunk * 16;
  }

  // stage unit u of K-tile tp (counted from the current output tile's first; tp >= nk runs into the next tile)
  auto stage_a = [&](int u, int tp, unsigned char* buf) {
    const bool nx = tp >= nk;
    const int tt = nx ? tp - nk : tp;
    if (UPS) {
      const int tap = (tt * cpt_magic) >> 16;
      const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
      const int oy = nx ? uy_nxt[u] : uy_cur[u], ox0 = nx ? ux_nxt[u] : ux_cur[u], ub = nx ? ub_nxt[u] : ub_cur[u];
      const int iy2 = oy + ky - 1;
      const bool vok = (unsigned)iy2 < (unsigned)(2 * p.Hin);
      const char* base = reinterpret_cast<const char*>(p.A) +
                         ((int64_t)(ub + (iy2 >> 1) * p.Win + (ox0 >> 1)) * p.Cin + (tt - tap * cpt) * BK) * 2;
      const int vo = kx == 0 ? voffU[0] : (kx == 1 ? voffU[1] : voffU[2]);
      const bool ok = vok && (unsigned)(ox0 + r64 + kx - 1) < (unsigned)(2 * p.Win);
      glds16(ok ? (const void*)(base + vo) : (const void*)g_zero_page4, buf + u * UNIT + ldsw);
    } else if (CONV) {
      const int tap = (tt * cpt_magic) >> 16;
      const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
      const int64_t toff = ((int64_t)((ky - 1) * p.Win + (kx - 1)) * p.Cin + (tt - tap * cpt) * BK) * 2;
      const int mk = nx ? am_nxt[u] : am_cur[u];
      const char* src = (nx ? a_nxt : a_cur) + u * a_unit + toff + voffA;
      glds16(((mk >> tap) & 1) ? (const void*)src : (const void*)g_zero_page4, buf + u * UNIT + ldsw);
    } else {
      const char* base = (nx ? a_nxt : a_cur) + (int64_t)tt * (BK * 2);
      glds16(base + u * a_unit + voffA, buf + u * UNIT + ldsw);
    }
  };
  auto stage_b = [&](int u, int tp, unsigned char* buf) {
    const char* base = tp < nk ? w_cur + (int64_t)tp * (BK * 2) : w_nxt + (int64_t)(tp - nk) * (BK * 2);
    glds16(base + u * w_unit + voffB, buf + ABYTES + u * UNIT + ldsw);
  };

  f32x4 acc[10][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 10; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  // fragment (i, kk) of a buffer: row = base + 16 i + fr, 16-byte chunk (4 kk + fq) ^ (fr & 7)
  const int swz0 = (fq ^ (fr & 7)) << 4;
  const int aoff0 = (wr * 64 + fr) * 128 + swz0, aoff1 = aoff0 ^ 64;
  const int boff0 = ABYTES + (wc * 32 + fr) * 128 + swz0, boff1 = boff0 ^ 64;

  v8 xa[2][4], wb[2][2];
  auto rdA = [&](const unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Pack8<T> t0, t1;
      t0.u = *reinterpret_cast<const u32x4*>(buf + aoff0 + i * 2048);
      t1.u = *reinterpret_cast<const u32x4*>(buf + aoff1 + i * 2048);
      xa[0][i] = t0.v;
      xa[1][i] = t1.v;
    }
  };
  auto rdB = [&](const unsigned char* buf, int u) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      Pack8<T> t0, t1;
      t0.u = *reinterpret_cast<const u32x4*>(buf + boff0 + u * UNIT + i * 2048);
      t1.u = *reinterpret_cast<const u32x4*>(buf + boff1 + u * UNIT + i * 2048);
      wb[0][i] = t0.v;
      wb[1][i] = t1.v;
    }
  };
#define SMI_MMA5(U)                                                                                   \
  do {                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) acc[2 * (U) + ni][mi] =                      \
            TT<T>::mfma16(wb[kk][ni], xa[kk][mi], acc[2 * (U) + ni][mi]);                             \
    __builtin_amdgcn_s_setprio(0);                                                                    \
  } while (0)

  // ---- prologue: K-tile 0 complete; B1 / A-u0, B2 / A-u1, B3 / A-u2 of K-tile 1 left in flight
  {
    unsigned char* b0 = smem;
    unsigned char* b1 = smem + KBUF;
#pragma unroll
    for (int u = 0; u < 4; ++u) stage_a(u, 0, b0);
#pragma unroll
    for (int u = 0; u < 5; ++u) stage_b(u, 0, b0);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      stage_b(u, 1, b1);
      stage_a(u, 1, b1);
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger: waves 4-7 run one barrier behind waves 0-3
  }

  unsigned gk = 0;  // K-tiles done by this workgroup (buffer parity)
  while (true) {
    {
      const int vn = v + G;
      has_next = vn < ntiles;
      if (has_next) {
        tile_origin(vn, bm1, bn1);
        a_nxt = reinterpret_cast<const char*>(p.A) + (int64_t)a_row0(bm1) * lda * 2;
        if (CONV) conv_masks(bm1, am_nxt);
        if (UPS) ups_units(bm1, ub_nxt, uy_nxt, ux_nxt);
        w_nxt = reinterpret_cast<const char*>(p.W) + (int64_t)(geglu ? bn1 >> 1 : bn1) * p.K * 2;
      }
    }
    for (int t = 0; t < nk; ++t, ++gk) {
      unsigned char* cur = smem + (gk & 1) * KBUF;
      unsigned char* oth = smem + ((gk & 1) ^ 1) * KBUF;
      const bool s1 = t + 1 < nk || has_next;  // K-tile t+1 exists (its B4 / A-u3 / B5 are still to be staged)
      const bool s2 = t + 2 < nk || has_next;  // K-tile t+2 exists
      // -------- phase 1
      SMI_FENCE();
      rdB(cur, 0);
      SMI_FENCE();
      rdA(cur);
      SMI_FENCE();
      if (s1) {
        stage_b(3, t + 1, oth);
        stage_a(3, t + 1, oth);
      }
      SMI_FENCE();
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SMI_FENCE();
      SMI_MMA5(0);
      SMI_FENCE();
      __builtin_amdgcn_s_barrier();
      // -------- phase 2
      SMI_FENCE();
      rdB(cur, 1);
      SMI_FENCE();
      if (s1) stage_b(4, t + 1, oth);
      SMI_FENCE();
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SMI_FENCE();
      SMI_MMA5(1);
      SMI_FENCE();
      __builtin_amdgcn_s_barrier();
      // -------- phase 3
      SMI_FENCE();
      rdB(cur, 2);
      SMI_FENCE();
      if (s2) {
        stage_b(0, t + 2, cur);
        stage_a(0, t + 2, cur);
      }
      SMI_FENCE();
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SMI_FENCE();
      SMI_MMA5(2);
      SMI_FENCE();
      __builtin_amdgcn_s_barrier();
      // -------- phase 4
      SMI_FENCE();
      rdB(cur, 3);
      SMI_FENCE();
      if (s2) {
        stage_b(1, t + 2, cur);
        stage_a(1, t + 2, cur);
      }
      SMI_FENCE();
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SMI_FENCE();
      SMI_MMA5(3);
      SMI_FENCE();
      __builtin_amdgcn_s_barrier();
      // -------- phase 5
      SMI_FENCE();
      rdB(cur, 4);
      SMI_FENCE();
      if (s2) {
        stage_b(2, t + 2, cur);
        stage_a(2, t + 2, cur);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // all but the last three stage groups: K-tile t+1 has landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      SMI_FENCE();
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SMI_FENCE();
      SMI_MMA5(4);
      SMI_FENCE();
      __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // re-join: both wave groups run the epilogue together
    SMI_FENCE();

    // ---- epilogue, straight from the accumulators: per (mi, pair u) a lane holds 8 consecutive columns of one row
    // tile row of fragment mi: plain 64 wr + 16 mi + fr; mix: fragments 0-2 frozen rows, fragment 3 adapted rows
    const int rowf = mix ? 3 * (bm0 >> 2) + wr * 48 + fr : bm0 + wr * 64 + fr;
    const int rowa = mix ? p.lora_row0 + (bm0 >> 2) + wr * 16 + fr : bm0 + wr * 64 + 48 + fr;
    auto row_of = [&](int mi) { return mi < 3 ? rowf + mi * 16 : rowa; };
    const bool lora_tile = p.lora_r > 0 && (mix || bm0 + BM > p.lora_row0);
    // forward-form delta (up [N, r] row-major, r % 4 == 0, r <= 16, a tile inside one fused segment) on the fp32 MFMA:
    // D[16 cols][16 rows] = up-fragment [16 x 4] * xa-fragment [4 x 16] + D per rank block -- the canonical fmaf chain
    // (smi_common.h lora_fma4) in the accumulators' own register layout, one dword load per operand and lane
    const bool lora_mfma = p.up_sq == 1 && p.up_sn == p.lora_r && (p.lora_r & 3) == 0 && p.lora_r <= 16 &&
                           (p.lora_seg == 0 || p.lora_seg % BN == 0);
    const int nblk = p.lora_r >> 2;
    const int xoff_t = p.lora_seg ? (bn0 / p.lora_seg) * p.lora_r : 0;  // (lora_mfma: the whole tile is one segment)
    auto epilogue_pair = [&](auto Uc) {
      constexpr int u = decltype(Uc)::value;
      const int n = bn0 + wc * 160 + u * 32 + fq * 8;
      float bv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = 0.f;
      if (p.bias) {
        Pack8<T> b;
        b.u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bias) + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = to_f(b.e[j]);
      }
      // MFMA A operand: lane (i = lane & 15, k = lane >> 4) holds up[column of fragment row i][4 blk + k]; fragment row i
      // of the pair's MFMA nip is output column 8 (i >> 2) + 4 nip + (i & 3) of the pair's 32
      float au[2][4];
#pragma unroll
      for (int nip = 0; nip < 2; ++nip)
#pragma unroll
        for (int b = 0; b < 4; ++b) au[nip][b] = 0.f;
      if (lora_tile && lora_mfma) {
        const int ncol = bn0 + wc * 160 + u * 32 + 8 * (fr >> 2) + (fr & 3);
#pragma unroll
        for (int nip = 0; nip < 2; ++nip)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (b < nblk) au[nip][b] = p.lora_up[(int64_t)(ncol + 4 * nip) * p.lora_r + 4 * b + fq];
      }
      // the residual rows of all four m-fragments of the pair are requested up front: written inside the mi loop each load
      // sat in its own basic block (the epilogue terms are run-time switches) with its wait right behind it -- 20 dependent
      // round trips per lane and tile, the whole chip in them at once in the one-round N = 1280 launches
      // (not in the plain conv mode: its tap masks leave no registers for the four rows -- 5 spills -- and its few
      // launches with a residual are multi-round, where the next tile's prefetch already runs under the epilogue)
      constexpr bool RES_AHEAD = !CONV;
      Pack8<T> rbuf[4];
      if (RES_AHEAD && p.res) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          rbuf[mi].u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.res) + (int64_t)row_of(mi) * p.ldr + n);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = row_of(mi);
        float vv[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vv[j] = acc[2 * u][mi][j];
          vv[4 + j] = acc[2 * u + 1][mi][j];
          acc[2 * u][mi][j] = 0.f;
          acc[2 * u + 1][mi][j] = 0.f;
        }
        if (p.bias) {
#pragma unroll
          for (int j = 0; j < 8; ++j) vv[j] += bv[j];
        }
        if (p.rowvec) {
          Pack8<T> b;
          b.u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.rowvec) +
                                                (int64_t)(m / p.rows_per_vec) * (p.ld_rowvec ? p.ld_rowvec : (int64_t)p.N) + n);
#pragma unroll
          for (int j = 0; j < 8; ++j) vv[j] += to_f(b.e[j]);
        }
        if (lora_tile && (!mix || mi == 3)) {  // (wave-uniform)
          const bool on = m >= p.lora_row0;
          if (lora_mfma) {
            f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = d0;
            const float* xr = p.lora_xa + (int64_t)(on ? m - p.lora_row0 : 0) * p.ld_xa + xoff_t + fq;
#pragma unroll
            for (int b = 0; b < 4; ++b)
              if (b < nblk) {
                const float bx = on ? xr[4 * b] : 0.f;  // B operand: lane (k = lane >> 4, j = lane & 15) holds xa[row j][4 blk + k]
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(au[0][b], bx, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(au[1][b], bx, d1, 0, 0, 0);
              }
            if (on) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                vv[j] = __builtin_fmaf(d0[j], p.lora_scale, vv[j]);
                vv[4 + j] = __builtin_fmaf(d1[j], p.lora_scale, vv[4 + j]);
              }
            }
          } else if (on) {
            const float* xrow0 = p.lora_xa + (int64_t)(m - p.lora_row0) * p.ld_xa;
            if (p.up_sn == 1 && (p.up_sq & 3) == 0) {  // dX form: "up" is lora_down [r, K] read along K
              const float* xrow = xrow0 + (p.lora_seg ? (n / p.lora_seg) * p.lora_r : 0);
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
              for (int j = 0; j < 8; ++j) vv[j] = __builtin_fmaf(d[j], p.lora_scale, vv[j]);
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float* xr = xrow0 + (p.lora_seg ? ((n + j) / p.lora_seg) * p.lora_r : 0);
                const float* up = p.lora_up + (int64_t)(n + j) * p.up_sn;
                float d = 0.f;
                for (int r = 0; r < p.lora_r; ++r) d = __builtin_fmaf(xr[r], up[r * p.up_sq], d);
                vv[j] = __builtin_fmaf(d, p.lora_scale, vv[j]);
              }
            }
          }
        }
        if (p.res) {
          if constexpr (!RES_AHEAD)
            rbuf[mi].u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.res) + (int64_t)m * p.ldr + n);
#pragma unroll
          for (int j = 0; j < 8; ++j) vv[j] += to_f(rbuf[mi].e[j]);
        }
        if (p.out_f32) {
          float* op = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
          *reinterpret_cast<f32x4*>(op) = f32x4{vv[0], vv[1], vv[2], vv[3]};
          *reinterpret_cast<f32x4*>(op + 4) = f32x4{vv[4], vv[5], vv[6], vv[7]};
        } else {
          Pack8<T> o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o.e[j] = from_f<T>(vv[j]);
          *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n) = o.u;
        }
      }
    };
    // fused GEGLU (bias only; same arithmetic as gemm2 / gemm3: the projection is rounded to 16 bits first, the gate is
    // applied to the rounded values): 8-byte stores of 4 consecutive columns
    auto epilogue_geglu = [&](auto Uc) {
      constexpr int u = decltype(Uc)::value;
      const int hc = (bn0 >> 1) + wc * 80 + u * 16 + fq * 4;  // hidden column; its gate is column nhalf + hc
      float bh[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
        Pack4<T> b0, b1;
        b0.u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.bias) + hc);
        b1.u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.bias) + nhalf + hc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bh[j] = to_f(b0.e[j]);
          bg[j] = to_f(b1.e[j]);
        }
      }
      T* gout = reinterpret_cast<T*>(p.geglu_out);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = bm0 + wr * 64 + mi * 16 + fr;
        Pack4<T> h, g, o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float vh = acc[2 * u][mi][j], vg = acc[2 * u + 1][mi][j];
          acc[2 * u][mi][j] = 0.f;
          acc[2 * u + 1][mi][j] = 0.f;
          if (p.bias) {
            vh += bh[j];
            vg += bg[j];
          }
          h.e[j] = from_f<T>(vh);
          g.e[j] = from_f<T>(vg);
          o.e[j] = from_f<T>(to_f(h.e[j]) * gelu_f(to_f(g.e[j])));
        }
        *reinterpret_cast<u32x2*>(gout + (int64_t)m * nhalf + hc) = o.u;
        if (m >= p.geglu_row0) {  // projection kept only for the rows that will be differentiated
          T* cp = reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc;
          *reinterpret_cast<u32x2*>(cp + hc) = h.u;
          *reinterpret_cast<u32x2*>(cp + nhalf + hc) = g.u;
        }
      }
    };
    if constexpr (GEGLU) {
      epilogue_geglu(std::integral_constant<int, 0>{});
      epilogue_geglu(std::integral_constant<int, 1>{});
      epilogue_geglu(std::integral_constant<int, 2>{});
      epilogue_geglu(std::integral_constant<int, 3>{});
      epilogue_geglu(std::integral_constant<int, 4>{});
    } else {
      epilogue_pair(std::integral_constant<int, 0>{});
      epilogue_pair(std::integral_constant<int, 1>{});
      epilogue_pair(std::integral_constant<int, 2>{});
      epilogue_pair(std::integral_constant<int, 3>{});
      epilogue_pair(std::integral_constant<int, 4>{});
    }
    SMI_FENCE();
    if (!has_next) break;
    if (grp == 1) __builtin_amdgcn_s_barrier();  // re-stagger
    v += G;
    bm0 = bm1;
    bn0 = bn1;
    a_cur = a_nxt;
    w_cur = w_nxt;
    if (CONV) {
#pragma unroll
      for (int u = 0; u < 4; ++u) am_cur[u] = am_nxt[u];
    }
    if (UPS) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ub_cur[u] = ub_nxt[u];
        uy_cur[u] = uy_nxt[u];
        ux_cur[u] = ux_nxt[u];
      }
    }
  }
}

template <typename T, int MODE>
int launch_t(const GemmParams& p, hipStream_t stream) {
  // per device (one bit per ordinal, set once under a mutex): the attribute belongs to the function ON the current
  // device, and engines on two devices / two host threads may share this process; the persistent grid is the device's
  // CU count, read from the device instead of assumed
  static std::mutex mu;
  static uint64_t attr_done = 0;
  static int cus[64] = {0};
  int dev = 0;
  SMI_HIP(hipGetDevice(&dev));
  int ncu = 256;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64 || !((attr_done >> dev) & 1)) {
      SMI_HIP(hipFuncSetAttribute((const void*)gemm_5ph_kernel<T, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SMEM4));
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      if (dev >= 0 && dev < 64) {
        cus[dev] = n;
        attr_done |= 1ull << dev;
      }
      ncu = n;
    } else {
      ncu = cus[dev];
    }
  }
  const int ntiles = (p.M / BM) * (p.N / BN);
  const int grid = ntiles < ncu ? ntiles : ncu;
  hipLaunchKernelGGL((gemm_5ph_kernel<T, MODE>), dim3(grid), dim3(512), SMEM4, stream, p);
  SMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace

bool gemm2_supported(const GemmParams& p);

// dense GEMMs and plain 3x3 convs on whole 256 x 320 tiles
bool gemm4_supported(const GemmParams& p) {
  if (!gemm2_supported(p)) return false;
  {  // SMI_G4_DENY (debugging): bit mask of launch classes kept off this kernel
    static const int deny = []() { const char* e = getenv("SMI_G4_DENY"); return e ? atoi(e) : 0; }();
    if ((deny & 1) && p.conv && !p.upsample && p.lora_r > 0) return false;
    if ((deny & 2) && p.conv && p.Cin == 64) return false;
    if ((deny & 4) && p.conv && p.upsample && p.lora_r > 0) return false;
    if ((deny & 8) && !p.conv && p.lora_r > 0 && p.lora_row0 > 0) return false;
    if ((deny & 16) && p.conv && p.res == p.C && p.res) return false;
    if ((deny & 32) && p.conv && p.upsample) return false;
    if ((deny & 64) && p.conv) return false;
    if ((deny & 128) && !p.conv && p.lora_r > 0) return false;
  }
  if (p.geglu_out && (p.conv || p.out_f32 || p.res || p.rowvec || p.lora_r > 0 ||
                      (reinterpret_cast<uintptr_t>(p.geglu_out) & 15) != 0))
    return false;
  if (p.conv && (p.stride != 1 || p.pad != 1 || p.transposed || p.Cin % BK != 0 || p.K != 9 * p.Cin ||
                 p.M != p.Nb * p.Hout * p.Wout))
    return false;
  if (p.conv && !p.upsample && (p.Hout != p.Hin || p.Wout != p.Win)) return false;
  if (p.conv && p.upsample && (p.Hout != 2 * p.Hin || p.Wout != 2 * p.Win || p.Wout % 64 != 0)) return false;
  if (p.M % BM != 0 || p.N % BN != 0 || p.K % BK != 0 || p.K < 2 * BK) return false;
  if (p.lora_seg % 8 != 0) return false;
  if (p.out_f32 && p.ldc % 4 != 0) return false;
  const int64_t lda = p.conv ? p.Cin : p.lda;
  if ((int64_t)BM * lda * 2 >= (1ll << 31) || (int64_t)BN * p.K * 2 >= (1ll << 31)) return false;  // 32-bit lane offsets
  return true;
}

int launch_gemm4(const GemmParams& p, hipStream_t stream) {
  const int mode = p.conv ? (p.upsample ? 3 : 1) : (p.geglu_out ? 2 : 0);
  if (p.dtype == DT_F16)
    return mode == 1   ? launch_t<f16, 1>(p, stream)
           : mode == 2 ? launch_t<f16, 2>(p, stream)
           : mode == 3 ? launch_t<f16, 3>(p, stream)
                       : launch_t<f16, 0>(p, stream);
  return mode == 1   ? launch_t<bf16, 1>(p, stream)
         : mode == 2 ? launch_t<bf16, 2>(p, stream)
         : mode == 3 ? launch_t<bf16, 3>(p, stream)
                     : launch_t<bf16, 0>(p, stream);
}

}  // namespace smi
