// MFMA GEMM / implicit-GEMM conv, second generation: direct-to-LDS staging.
//
// What changed against gemm.hip (kept as the fallback for unaligned operands) and why -- measured on MI355X with
// rocprofv3 counters on the v1 kernel: MFMA pipe 37 % busy, 22 % of wave-cycles stalled on LDS issue, 0 bank
// conflicts.  The VGPR->LDS store path (ds_write_b128: 13 cycles per wave-instruction) cost more LDS-pipe time per
// K-step than the MFMAs take on a 128x128 tile.  Here:
//   * operands go HBM/L2 -> LDS by LDS-DMA (`global_load_lds_dwordx4`): no staging registers, no ds_write;
//     the LDS image is lane-linear per wave-instruction (1 KiB = 8 rows x 128 B), so the XOR swizzle that keeps
//     ds_read_b128 conflict-free is applied to the per-lane SOURCE address (cdna guide rule 21);
//   * two LDS stages, ONE barrier per K-step: the DMA of tile t+1 is in flight under the MFMAs of tile t;
//   * rows that do not exist (M/N tails, conv padding, strided / transposed / up-sampled taps) read a 256-byte
//     zero page instead of relying on the buffer range check (LDS-DMA has a per-lane source address but no
//     per-lane predicate);
//   * optional 256-row tile (8 waves) for the large-M layers: fewer LDS bytes staged per FLOP;
//   * epilogue lane remap: the W-fragment rows of each MFMA pair are permuted so that a lane owns 8 CONSECUTIVE
//     output columns -> 16-byte stores / residual / bias accesses instead of 8-byte ones.
#include "kernels.h"

namespace smi {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];  // zero-initialised

namespace {

constexpr int BK = 64;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_dst, 16, 0, 0);
}

// Block tile = (64 WM) x BN with 2 WM waves.  NL = 0: BN = 128, waves as WM (M) x 2 (N), 64 x 64 per wave.
// NL = 1: BN = 160, waves as 2 WM (M) x 1 (N), 32 x 160 per wave -- every layer width of the UNets (320 / 640 / 1280 and
// their multiples) is a multiple of 160 but not of 128 (320 output channels = 2.5 tiles of 128), and N = 1280 splits
// into 8 column tiles so that e.g. 4096 x 1280 is exactly 256 workgroups, one per CU.
// NL = 2: BN = 128 with TWICE the waves (4 WM, as 2 WM (M) x 2 (N), 32 x 64 per wave): for grids of about one
// workgroup per CU, where a lone wave per SIMD serialises its DMA issue, fragment reads and MFMAs -- two waves per
// SIMD from the same workgroup overlap them.
// NL = 3: BN = 160 with twice the waves, as 2 WM (M) x 2 (N), 32 x 80 per wave = FIVE column fragments: two pairs (8
// consecutive columns per lane, as everywhere) and one single fragment (4 columns per lane; its epilogue is the 4-column
// path that exists for conv_out).  For N = 1280 at 4096 rows this is 32 x 8 = 256 tiles -- one round of the CUs with
// two waves per SIMD -- where the 128 x 128 eight-wave tile needs 320 tiles = 1.25 rounds.
// DEEP: four LDS stages, three K-tiles of DMA in flight across raw barriers with a counted vmcnt (bit-identical: same K
// order).  In isolation (operands hot in L2 / the Infinity Cache, which is also what the tile tuner sees) it buys
// nothing; the question is the cold operands of the real backward pass.
template <typename T, bool CONV, int WM, int NL, bool DEEP = false>
__global__ __launch_bounds__((NL >= 2 ? 4 : 2) * WM * 64) void gemm_glds_kernel(GemmParams p) {
  constexpr int BN = (NL == 1 || NL == 3) ? 160 : 128;
  constexpr int BM = 64 * WM;
  constexpr int NW = (NL >= 2 ? 4 : 2) * WM;
  constexpr int MI = NL ? 2 : 4;          // 16-row fragments per wave
  constexpr int NI = NL == 1 ? 10 : (NL == 3 ? 5 : 4);  // 16-column fragments per wave
  static_assert(NL == 3 || ((BN / 8) % NW) == 0, "W tile rows must split evenly over the waves");
  constexpr int STAGE = (BM + BN) * BK * 2;  // bytes
  constexpr int A_INSTR = (BM / 8) / NW;     // 1-KiB wave-instructions per wave for the A tile (= 4)
  // (NL = 3: 20 B pieces over 8 waves -- piece j * NW + wave, waves 0-3 issue three, waves 4-7 two)
  constexpr int B_INSTR = (BN / 8 + NW - 1) / NW;  // (= 4, 2 or 3)
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = NL == 1 ? wave : wave >> 1, wn = NL == 1 ? 0 : wave & 1;
  auto b_piece = [&](int j) { return NL == 3 ? j * NW + wave : wave * B_INSTR + j; };
  auto b_valid = [&](int j) { return NL != 3 || j * NW + wave < BN / 8; };

  constexpr int WROWS = 16 * MI, WCOLS = 16 * NI;  // wave tile

  const int nbn = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int wg;
  {
    const int orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  // split-K: the grid holds ksplit workgroups per output tile (adjacent after the remap); slice s owns a contiguous range
  // of K-tiles and writes fp32 partials to its own slab
  const int nsplit = p.ksplit > 1 ? p.ksplit : 1;
  int kslice = 0;
  if (nsplit > 1) {
    kslice = wg % nsplit;
    wg /= nsplit;
  }
  // grouped rasterisation: consecutive workgroups (= the ones co-resident on one XCD after the remap above) sweep a
  // band of GW column tiles before moving down a row tile, so the 32-64 tiles sharing an L2 form a ~8x8 patch
  // (8 + 8 operand panels per K-step instead of 1 + 64 for a wide-N GEMM walked row-major)
  constexpr int GW = 8;
  const int nbm = (p.M + BM - 1) / BM;
  const int grp = wg / (GW * nbm);
  const int gw = min(nbn - grp * GW, GW);
  const int lw = wg - grp * GW * nbm;
  const int bm0 = (lw / gw) * BM;
  const int bn0 = (grp * GW + lw % gw) * BN;
  const bool geglu = p.geglu_out != nullptr;
  const int nhalf = p.N >> 1;
  // tile-local column nl (0..127) -> global output column
  auto gcol = [&](int nl) {
    return geglu ? (nl < BN / 2 ? (bn0 >> 1) + nl : nhalf + (bn0 >> 1) + nl - BN / 2) : bn0 + nl;
  };

  const int nk_all = (p.K + BK - 1) / BK;
  const int kbase = (int)((int64_t)kslice * nk_all / nsplit);                    // first K-tile of this slice
  const int nk = (int)((int64_t)(kslice + 1) * nk_all / nsplit) - kbase;         // its K-tile count (all of K unsplit)
  const unsigned char* zero = g_zero_page;
  const T* Ap = reinterpret_cast<const T*>(p.A);
  const T* Wp = reinterpret_cast<const T*>(p.W);

  // ---- per-lane source rows.  Wave w issues A instructions j = 0..A_INSTR-1 covering tile rows
  //      (w*A_INSTR + j)*8 + (lane>>3); lane slot (lane&7) holds source chunk slot ^ (row & 7).
  const int lrow = lane >> 3, lslot = lane & 7;
  const T* a_ptr[A_INSTR];   // dense: row base pointer (or null -> zero page); conv fast path: centre-tap pointer
  int c_base[A_INSTR], c_oy[A_INSTR], c_ox[A_INSTR];
  // plain 3x3 / stride 1 convs (all but 5 launches per pass): the tap only adds a wave-uniform element offset
  const bool conv_fast = CONV && p.stride == 1 && !p.upsample && !p.transposed;
  int a_chunk[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (wave * A_INSTR + j) * 8 + lrow;
    const int m = bm0 + row;
    a_chunk[j] = lslot ^ (row & 7);
    if (CONV) {
      if (m < p.M) {
        const int hw = p.Hout * p.Wout;
        const int img = m / hw;
        const int rem = m - img * hw;
        c_base[j] = img * p.Hin * p.Win;
        c_oy[j] = rem / p.Wout;
        c_ox[j] = rem - c_oy[j] * p.Wout;
      } else {
        c_base[j] = 0;
        c_oy[j] = -(1 << 20);
        c_ox[j] = -(1 << 20);
      }
      a_ptr[j] = Ap + ((int64_t)(c_base[j] + c_oy[j] * p.Win + c_ox[j]) * p.Cin + a_chunk[j] * 8);
    } else {
      a_ptr[j] = (m < p.M) ? Ap + (int64_t)m * p.lda : nullptr;
    }
  }
  const T* w_ptr[B_INSTR];
  int w_chunk[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int row = b_piece(j) * 8 + lrow;
    // epilogue lane remap, applied at staging time so fragment reads stay on consecutive (conflict-free) LDS rows:
    // LDS row 16*ni + fr (ni counted over the whole tile) holds W row 32*(ni>>1) + 8*(fr>>2) + 4*(ni&1) + (fr&3);
    // after the MFMAs of the pair (2q, 2q+1) a lane then owns the 8 consecutive columns 32q + 8*fq + {0..7}.
    // NL = 3 pairs inside each wave column's five fragments; the fifth keeps its columns in order (4 per lane).
    const int ni_ = row >> 4, fr_ = row & 15;
    int col;
    if (NL == 3) {
      const int wn_ = row / 80, l = row - wn_ * 80, nl_ = l >> 4;
      col = wn_ * 80 + (nl_ < 4 ? (nl_ >> 1) * 32 + 8 * (fr_ >> 2) + 4 * (nl_ & 1) + (fr_ & 3) : 64 + (l & 15));
    } else {
      col = (ni_ >> 1) * 32 + 8 * (fr_ >> 2) + 4 * (ni_ & 1) + (fr_ & 3);
    }
    const int n = gcol(col);
    w_chunk[j] = lslot ^ (row & 7);
    w_ptr[j] = (b_valid(j) && n < p.N) ? Wp + (int64_t)n * p.K : nullptr;
  }

  // dense operands with K % 64 == 0 (every layer of the UNets): the source of each LDS-DMA is a per-lane pointer that
  // simply advances by 64 elements per K-step (rows that do not exist keep pointing at the zero page with step 0):
  // 2 VALU per load instead of ~10 -- the K loop was issuing 3.9 VALU per MFMA, more than the MFMA leaves issue
  // slots for
  const bool inc_path = !CONV && (p.K % BK) == 0;
  const T* a_src[A_INSTR];
  const T* w_src[B_INSTR];
  int a_step[A_INSTR], w_step[B_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const bool ok = !CONV && a_ptr[j] != nullptr;
    a_src[j] = ok ? a_ptr[j] + a_chunk[j] * 8 + (int64_t)kbase * BK : reinterpret_cast<const T*>(zero);
    a_step[j] = ok ? BK : 0;
  }
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const bool ok = w_ptr[j] != nullptr;
    w_src[j] = ok ? w_ptr[j] + w_chunk[j] * 8 + (int64_t)kbase * BK : reinterpret_cast<const T*>(zero);
    w_step[j] = ok ? BK : 0;
  }
  auto stage_inc = [&](int buf) {
    unsigned char* As = smem + buf * STAGE;
    unsigned char* Bs = As + BM * BK * 2;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      glds16(a_src[j], As + (wave * A_INSTR + j) * 1024);
      a_src[j] += a_step[j];
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      if (b_valid(j)) glds16(w_src[j], Bs + b_piece(j) * 1024);
      w_src[j] += w_step[j];
    }
  };
  auto stage = [&](int kt, int buf) {
    if (inc_path) {
      stage_inc(buf);
      return;
    }
    unsigned char* As = smem + buf * STAGE;
    unsigned char* Bs = As + BM * BK * 2;
    const int k0 = (kbase + kt) * BK;
    if (CONV) {
      const int tap = k0 / p.Cin;
      const int c0 = k0 - tap * p.Cin;
      const int ky = tap / 3, kx = tap - ky * 3;
      if (conv_fast) {
        const int toff = ((ky - 1) * p.Win + (kx - 1)) * p.Cin + c0;  // wave-uniform
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
          const int iy = c_oy[j] + ky - 1, ix = c_ox[j] + kx - 1;
          const bool ok = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
          const void* src = ok ? (const void*)(a_ptr[j] + toff) : (const void*)zero;
          glds16(src, As + (wave * A_INSTR + j) * 1024);
        }
      } else
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        int iy, ix;
        bool ok;
        if (p.transposed) {
          const int ty = c_oy[j] + 1 - ky, tx = c_ox[j] + 1 - kx;
          iy = ty / p.stride;
          ix = tx / p.stride;
          ok = (ty >= 0) && (tx >= 0) && (iy * p.stride == ty) && (ix * p.stride == tx) && (iy < p.Hin) &&
               (ix < p.Win);
        } else {
          iy = c_oy[j] * p.stride + ky - p.pad;
          ix = c_ox[j] * p.stride + kx - p.pad;
          if (p.upsample) {
            ok = (iy >= 0) && (ix >= 0) && (iy < 2 * p.Hin) && (ix < 2 * p.Win);
            iy >>= 1;
            ix >>= 1;
          } else {
            ok = (iy >= 0) && (ix >= 0) && (iy < p.Hin) && (ix < p.Win);
          }
        }
        const void* src = ok ? (const void*)(Ap + ((int64_t)(c_base[j] + iy * p.Win + ix) * p.Cin + c0 + a_chunk[j] * 8))
                             : (const void*)zero;
        glds16(src, As + (wave * A_INSTR + j) * 1024);
      }
    } else {
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        const int kc = k0 + a_chunk[j] * 8;
        const void* src = (a_ptr[j] && kc < p.K) ? (const void*)(a_ptr[j] + kc) : (const void*)zero;
        glds16(src, As + (wave * A_INSTR + j) * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      const int kc = k0 + w_chunk[j] * 8;
      const void* src = (w_ptr[j] && kc < p.K) ? (const void*)(w_ptr[j] + kc) : (const void*)zero;
      if (b_valid(j)) glds16(src, Bs + b_piece(j) * 1024);
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int fq = lane >> 4;

  // per-lane LDS byte offsets of the 2 x (4 + 4) fragments of a stage, computed once
  int offA[2][MI], offB[2][NI];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ch = kk * 4 + fq;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int rm = wm * WROWS + i * 16 + fr;
      offA[kk][i] = rm * 128 + ((ch ^ (rm & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rn = wn * WCOLS + i * 16 + fr;
      offB[kk][i] = BM * BK * 2 + rn * 128 + ((ch ^ (rn & 7)) << 4);
    }
  }

  // (A 4-stage variant -- three tiles in flight across raw barriers, counted vmcnt, one workgroup per CU -- was built
  // and measured for the small grids: 4096 x 1280 x 1280 24-36 us against 22.8 us here.  Not kept.)
  // DMA instructions this wave issues per K-tile (NL = 3: waves 4-7 one B piece fewer)
  const int per = A_INSTR + (NL == 3 ? (wave < (BN / 8) % NW ? B_INSTR : B_INSTR - 1) : B_INSTR);
  // wait until at most n of this wave's staged K-tiles are still in flight (DEEP only; n = 0, 1, 2)
  auto wait_tiles = [&](int n) {
    const int c = n * per;  // wave-uniform; s_waitcnt takes an immediate
    if (c >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (c == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (c == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (c >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (c == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if (DEEP) {
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 2) stage(2, 2);
    wait_tiles(min(nk, 3) - 1);
    __builtin_amdgcn_s_barrier();
  } else {
    stage(0, 0);
    __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) in front of the barrier
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = DEEP ? (kt & 3) : (kt & 1);
    if (DEEP) {
      // buffer (kt + 3) & 3 held K-tile kt - 1: every wave finished reading it before the barrier that ended iteration kt - 1
      if (kt + 3 < nk) stage(kt + 3, (kt + 3) & 3);
    } else {
      if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
    }
    const unsigned char* As = smem + buf * STAGE;
    // fragments of both 32-deep halves are fetched up front (two register sets) so the second half's LDS latency
    // hides under the first half's MFMAs
    typename TT<T>::v8 xa[2][MI], wb[2][NI];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        Pack8<T> t;
        t.u = *reinterpret_cast<const u32x4*>(As + offA[kk][i]);
        xa[kk][i] = t.v;
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        Pack8<T> s;
        s.u = *reinterpret_cast<const u32x4*>(As + offB[kk][i]);
        wb[kk][i] = s.v;
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = TT<T>::mfma16(wb[kk][ni], xa[kk][mi], acc[ni][mi]);
    if (DEEP) {
      // K-tile kt + 1 must have landed (this wave's pieces; the barrier covers the other waves'): leave kt + 2, kt + 3
      wait_tiles(min(nk - 1, kt + 3) - (kt + 1));
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
  }
  if (DEEP) __syncthreads();  // (nothing is in flight any more; the epilogue reuses the staging memory)

  // ---- epilogue: per (mi, pair q) the lane holds 8 consecutive columns n = nb + 32q + 8fq + {0..7} of row m.
  // 16-bit results are staged through LDS (free after the last barrier) and written out as whole rows: one wave
  // store instruction then covers 4 rows x 256 contiguous bytes instead of 16 rows x 64 bytes -- half-line writes are
  // cheap while the output fits the Infinity Cache and cost ~2x once it streams to HBM (measured: ff1 at M = 65536).
  constexpr int OLD = BN + 8;  // LDS row length (elements) of the staged output tile: +16 B pad
  const bool stage_out = !p.out_f32 && (p.N % 8 == 0);
  T* otile = reinterpret_cast<T*>(smem);

  // NL = 3 (the tile of the 4096-row backward GEMMs, whose dX carries the rank-r delta on every row): the delta of all
  // ten (fragment, row-fragment) pairs on the fp32 MFMA, in uniform control flow ahead of the per-lane epilogue --
  // D[16 cols x 16 rows] = "up"-fragment [16 x 4] * xa-fragment [4 x 16] + D per rank block, the canonical fmaf chain of
  // smi_common.h in the accumulators' register layout (as in gemm4.hip), for both operand forms (forward: up [N, r];
  // dX: lora_down [r, K] read along K).  One dword load per operand and lane instead of 3 r 16-byte loads per 8 columns.
  f32x4 dlt[NL == 3 ? NI : 1][NL == 3 ? MI : 1];
  bool delta_mfma = false;
  if constexpr (NL == 3) {
    const bool fwd_form = p.up_sq == 1 && p.up_sn == p.lora_r;
    const bool dx_form = p.up_sn == 1 && !fwd_form;
    delta_mfma = p.lora_r > 0 && (p.lora_r & 3) == 0 && p.lora_r <= 16 && (fwd_form || dx_form) &&
                 (p.lora_seg == 0 || p.lora_seg % BN == 0) && bm0 + BM > p.lora_row0 && p.N % BN == 0;
    if (delta_mfma) {
      const int nblk = p.lora_r >> 2;
      const int xoff = p.lora_seg ? (bn0 / p.lora_seg) * p.lora_r : 0;
      float bx[MI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = bm0 + wm * WROWS + mi * 16 + fr;
        const bool on = m < p.M && m >= p.lora_row0;
        const float* xr = p.lora_xa + (int64_t)(on ? m - p.lora_row0 : 0) * p.ld_xa + xoff + fq;
#pragma unroll
        for (int b = 0; b < 4; ++b) bx[mi][b] = (on && b < nblk) ? xr[4 * b] : 0.f;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = bn0 + wn * WCOLS +
                        (ni < NI - 1 ? (ni >> 1) * 32 + 8 * (fr >> 2) + 4 * (ni & 1) + (fr & 3) : (NI - 1) * 16 + fr);
        float au[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
          au[b] = b < nblk ? (fwd_form ? p.lora_up[(int64_t)col * p.lora_r + 4 * b + fq]
                                       : p.lora_up[(int64_t)(4 * b + fq) * p.up_sq + col])
                           : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (b < nblk) d = __builtin_amdgcn_mfma_f32_16x16x4f32(au[b], bx[mi][b], d, 0, 0, 0);
          dlt[ni][mi] = d;
        }
      }
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int ml = wm * WROWS + mi * 16 + fr;
    const int m = bm0 + ml;
    if (m >= p.M) continue;
    const bool lora_on = p.lora_r > 0 && m >= p.lora_row0;
    const float* xrow0 = p.lora_xa + (int64_t)(m - p.lora_row0) * p.ld_xa;
    const int64_t vrow = p.rowvec ? (int64_t)(m / p.rows_per_vec) * (p.ld_rowvec ? p.ld_rowvec : (int64_t)p.N) : 0;
#pragma unroll
    for (int q = 0; q < (NI + 1) / 2; ++q) {
      constexpr bool ODD = (NI & 1) != 0;
      const bool single = ODD && q == NI / 2;  // NL = 3: the wave's fifth fragment, 4 columns per lane
      const int nl = single ? wn * WCOLS + (NI - 1) * 16 + fq * 4 : wn * WCOLS + q * 32 + fq * 8;
      const int n = gcol(nl);
      if (n >= p.N) continue;
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[2 * q][mi][j];
        v[4 + j] = acc[(2 * q + 1 < NI) ? 2 * q + 1 : 0][mi][j];  // (unused for the single fragment)
      }
      const bool full = !single && n + 8 <= p.N;  // N % 8 may be 4 (conv_out): second half masked
      if (p.bias) {
        const T* bp = reinterpret_cast<const T*>(p.bias) + n;
        if (full) {
          Pack8<T> b;
          b.u = *reinterpret_cast<const u32x4*>(bp);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += to_f(b.e[j]);
        } else {
          Pack4<T> b;
          b.u = *reinterpret_cast<const u32x2*>(bp);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += to_f(b.e[j]);
        }
      }
      if (p.rowvec) {
        const T* bp = reinterpret_cast<const T*>(p.rowvec) + vrow + n;
        if (full) {
          Pack8<T> b;
          b.u = *reinterpret_cast<const u32x4*>(bp);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += to_f(b.e[j]);
        } else {
          Pack4<T> b;
          b.u = *reinterpret_cast<const u32x2*>(bp);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += to_f(b.e[j]);
        }
      }
      if (NL == 3 && delta_mfma) {
        if (lora_on) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = __builtin_fmaf(dlt[NL == 3 ? 2 * q : 0][NL == 3 ? mi : 0][j], p.lora_scale, v[j]);
            if (!single)
              v[4 + j] = __builtin_fmaf(dlt[(NL == 3 && 2 * q + 1 < NI) ? 2 * q + 1 : 0][NL == 3 ? mi : 0][j],
                                        p.lora_scale, v[4 + j]);
          }
        }
      } else if (lora_on) {
        const float* xrow = xrow0 + (p.lora_seg ? (n / p.lora_seg) * p.lora_r : 0);
        if ((full || single) && p.up_sq == 1 && p.up_sn == p.lora_r && (p.lora_r & 3) == 0) {
          // forward, rank 4 / 8 / ...: 16-byte loads of xa and of each output column's row of lora_up [N, r]
          // (the single fragment of the 32 x 80 wave tile: its 4 columns only)
          float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          for (int r0 = 0; r0 < p.lora_r; r0 += 4) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + r0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (j < 4 || full) {
                const f32x4 uv = *reinterpret_cast<const f32x4*>(p.lora_up + (int64_t)(n + j) * p.lora_r + r0);
                d[j] = lora_fma4(d[j], xv, uv);
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(d[j], p.lora_scale, v[j]);
        } else if ((full || single) && p.up_sn == 1 && (p.up_sq & 3) == 0) {
          // backward (dX): "up" is lora_down [r_tot, K] read along K: 8 consecutive columns = two 16-byte loads per q
          float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          for (int r = 0; r < p.lora_r; ++r) {
            const float xq = xrow[r];
            const float* ar = p.lora_up + (int64_t)r * p.up_sq + n;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ar);
            const f32x4 a1 = full ? *reinterpret_cast<const f32x4*>(ar + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
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
            if ((full || j < 4) && n + j < p.N) {
              const float* xr = xrow0 + (p.lora_seg ? ((n + j) / p.lora_seg) * p.lora_r : 0);
              const float* up = p.lora_up + (int64_t)(n + j) * p.up_sn;
              float d = 0.f;
              for (int r = 0; r < p.lora_r; ++r) d = __builtin_fmaf(xr[r], up[r * p.up_sq], d);
              v[j] = __builtin_fmaf(d, p.lora_scale, v[j]);
            }
          }
        }
      }
      if (p.res) {
        const T* rp = reinterpret_cast<const T*>(p.res) + (int64_t)m * p.ldr + n;
        if (full) {
          Pack8<T> b;
          b.u = *reinterpret_cast<const u32x4*>(rp);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += to_f(b.e[j]);
        } else {
          Pack4<T> b;
          b.u = *reinterpret_cast<const u32x2*>(rp);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += to_f(b.e[j]);
        }
      }
      if (p.out_f32) {
        float* op = reinterpret_cast<float*>(p.C) + ((int64_t)kslice * p.M + m) * p.ldc + n;  // (kslice = 0 unsplit)
        *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
        if (full) *reinterpret_cast<f32x4*>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else if (stage_out && full) {
        Pack8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = from_f<T>(v[j]);
        *reinterpret_cast<u32x4*>(otile + ml * OLD + nl) = o.u;
      } else if (stage_out) {
        Pack4<T> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.e[j] = from_f<T>(v[j]);
        *reinterpret_cast<u32x2*>(otile + ml * OLD + nl) = o.u;
      } else {
        T* op = reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n;
        Pack4<T> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.e[j] = from_f<T>(v[j]);
        *reinterpret_cast<u32x2*>(op) = o.u;
      }
    }
  }
  if (stage_out && geglu) {
    __syncthreads();
    constexpr int NT = NW * 64;
    {  // hidden * gelu(gate): BN/16 chunks of 8 output columns per row
      constexpr int CH = BN / 16;
      T* gout = reinterpret_cast<T*>(p.geglu_out);
#pragma unroll
      for (int i = 0; i < (BM * CH) / NT; ++i) {
        const int idx = tid + i * NT;
        const int r = idx / CH, c = idx - r * CH;
        const int m = bm0 + r;
        if (m < p.M) {
          Pack8<T> h, g, o;
          h.u = *reinterpret_cast<const u32x4*>(otile + r * OLD + c * 8);
          g.u = *reinterpret_cast<const u32x4*>(otile + r * OLD + BN / 2 + c * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) o.e[e] = from_f<T>(to_f(h.e[e]) * gelu_f(to_f(g.e[e])));
          *reinterpret_cast<u32x4*>(gout + (int64_t)m * nhalf + (bn0 >> 1) + c * 8) = o.u;
        }
      }
    }
    if (bm0 + BM > p.geglu_row0) {  // projection kept only for the rows that will be differentiated
      constexpr int CH = BN / 8;
#pragma unroll
      for (int i = 0; i < (BM * CH) / NT; ++i) {
        const int idx = tid + i * NT;
        const int r = idx / CH, c = idx - r * CH;
        const int n = gcol(c * 8);
        const int m = bm0 + r;
        if (m < p.M && m >= p.geglu_row0)
          *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n) =
              *reinterpret_cast<const u32x4*>(otile + r * OLD + c * 8);
      }
    }
  } else if (stage_out) {
    __syncthreads();
    constexpr int NT = NW * 64;
    constexpr int CH = BN / 8;  // 16-byte chunks per tile row
#pragma unroll
    for (int i = 0; i < (BM * CH) / NT; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / CH, c = idx - r * CH;
      const int n = bn0 + c * 8;
      const int m = bm0 + r;
      if (m < p.M && n < p.N)
        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n) =
            *reinterpret_cast<const u32x4*>(otile + r * OLD + c * 8);
    }
  }
}

template <typename T, bool CONV, int WM, int NL, bool DEEP = false>
int launch_t(const GemmParams& p, hipStream_t stream) {
  constexpr int BN = (NL == 1 || NL == 3) ? 160 : 128;
  constexpr int BM = 64 * WM;
  constexpr int NW = (NL >= 2 ? 4 : 2) * WM;
  constexpr int SMEM = (DEEP ? 4 : 2) * (BM + BN) * BK * 2;
  if (SMEM > 65536) {
    static DynLdsOnce once;  // (per instantiation, per device)
    if (int rc = once.set((const void*)gemm_glds_kernel<T, CONV, WM, NL, DEEP>, SMEM)) return rc;
  }
  const int grid = cdiv(p.M, BM) * cdiv(p.N, BN) * (p.ksplit > 1 ? p.ksplit : 1);
  hipLaunchKernelGGL((gemm_glds_kernel<T, CONV, WM, NL, DEEP>), dim3(grid), dim3(NW * 64), SMEM, stream, p);
  SMI_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// true when every operand meets the 16-byte alignment / multiple-of-8 layout the LDS-DMA path and the 16-byte
// epilogue need; anything else goes to the v1 kernel (gemm.hip)
bool gemm2_supported(const GemmParams& p) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (p.K % 8 != 0 || !al16(p.A) || !al16(p.W) || !al16(p.C)) return false;
  if (p.conv ? (p.Cin % 64 != 0) : (p.lda % 8 != 0)) return false;
  if (p.N % 8 == 0) {
    if (p.ldc % 8 != 0) return false;
    if (p.res && (p.ldr % 8 != 0 || !al16(p.res))) return false;
    if (p.bias && !al16(p.bias)) return false;
    if (p.rowvec && !al16(p.rowvec)) return false;
    return true;
  }
  return p.N == 4 && p.ldc % 4 == 0 && !p.res && !p.rowvec;  // conv_out: one half-width column group
}

// fused GEGLU: a tile must hold matching hidden / gate column groups, i.e. N is a multiple of 2 x the tile width of
// whichever variant launch_gemm2 will pick (N % 256 for the 128-column tiles, N % 320 for the 160-column ones)
bool gemm2_geglu_supported(const GemmParams& p) {
  return gemm2_supported(p) && !p.conv && !p.out_f32 && !p.res && !p.rowvec && p.lora_r == 0 && p.N % 256 == 0 &&
         (reinterpret_cast<uintptr_t>(p.geglu_out) & 15) == 0;
}

int launch_gemm2(const GemmParams& p, int variant, hipStream_t stream) {
  // variant: 0 = auto, 1 = 128-row tile, 2 = 256-row tile, 3 = 64-row tile, 4 = 128 x 160 tile,
  //          5 = 128 x 128 tile with 8 waves
  int wm = 2, nl = 0;
  const bool ok160 = p.N % 8 == 0 && (!p.geglu_out || p.N % 320 == 0);
  if (variant == 2) wm = 4;
  else if (variant == 0 || variant == 6) {
    // measured (tools/bench_gemm.py): the 256-row tile only pays for very wide outputs (N >= 4096: fewer LDS bytes
    // staged per FLOP); at small N or short K the 128-row tile's extra resident workgroup per CU wins
    const int64_t tiles256 = (int64_t)cdiv(p.M, 256) * cdiv(p.N, 128);
    if (tiles256 >= 512 && p.N >= 4096 && !p.conv) wm = 4;
    if (p.conv && tiles256 >= 512) wm = 4;
    // 128 x 160 tiles for the big grids (>= 4 tiles per CU; below that a launch is latency-bound and the smaller
    // tile wins): cost = tiles per CU x tile area / relative tile efficiency -- reproduces the measured ratios
    // (conv 262144 x 320: 858 vs 790 TF/s; 65536 x 640: 932 vs 1028; 16384 x 1280: 955 vs 867)
    if (ok160 && p.N % 160 == 0 && (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128) >= 1024) {
      auto cost = [&](int bm, int bn, double eff) {
        const int64_t tiles = (int64_t)cdiv(p.M, bm) * cdiv(p.N, bn);
        return (double)((tiles + 255) / 256) * bm * bn / eff;
      };
      if (cost(128, 160, 1.04) < cost(64 * wm, 128, wm == 4 ? 1.10 : 1.0)) { wm = 2; nl = 1; }
    }
  }
  // the 128 x 128 tile runs with 8 waves (measured better than 4 on every shape of tools/bench_gemm.py, most with
  // an epilogue: 4096 x 1280 x 1280 + bias + residual 24.7 vs 27.8 us); variant 6 = auto with the 4-wave form
  if ((variant == 0 || variant == 1) && wm == 2 && nl == 0) nl = 2;
  if (variant == 3) wm = 1;
  if (variant == 4 && ok160) { wm = 2; nl = 1; }
  if (variant == 5) { wm = 2; nl = 2; }
  if (variant == 7) { wm = 1; nl = 2; }  // 64 x 128 tile, 4 waves of 32 x 64: twice the tiles for small grids
  if (variant == 12 && ok160 && !p.geglu_out) {  // 64 x 160 tile, 4 waves of 32 x 80: small grids with N % 160 == 0
    if (p.dtype == DT_F16) return p.conv ? launch_t<f16, true, 1, 3>(p, stream) : launch_t<f16, false, 1, 3>(p, stream);
    return p.conv ? launch_t<bf16, true, 1, 3>(p, stream) : launch_t<bf16, false, 1, 3>(p, stream);
  }
  if ((variant == 10 || variant == 11) && ok160 && !p.geglu_out) {  // 128 x 160 tile with eight waves of 32 x 80 (NL = 3)
    if (variant == 11) {  // ... with the deep-prefetch loop
      if (p.dtype == DT_F16)
        return p.conv ? launch_t<f16, true, 2, 3, true>(p, stream) : launch_t<f16, false, 2, 3, true>(p, stream);
      return p.conv ? launch_t<bf16, true, 2, 3, true>(p, stream) : launch_t<bf16, false, 2, 3, true>(p, stream);
    }
    if (p.dtype == DT_F16) return p.conv ? launch_t<f16, true, 2, 3>(p, stream) : launch_t<f16, false, 2, 3>(p, stream);
    return p.conv ? launch_t<bf16, true, 2, 3>(p, stream) : launch_t<bf16, false, 2, 3>(p, stream);
  }
#define GO(TT_, CV, W_, NL_) return launch_t<TT_, CV, W_, NL_>(p, stream)
#define PICK(TT_, CV)                            \
  do {                                           \
    if (nl == 2 && wm == 1) GO(TT_, CV, 1, 2);   \
    if (nl == 2) GO(TT_, CV, 2, 2);              \
    if (nl) GO(TT_, CV, 2, 1);                   \
    if (wm == 4) GO(TT_, CV, 4, 0);              \
    if (wm == 1 && !CV) GO(TT_, false, 1, 0);    \
    GO(TT_, CV, 2, 0);                           \
  } while (0)
  if (p.dtype == DT_F16) {
    if (p.conv) PICK(f16, true); else PICK(f16, false);
  } else {
    if (p.conv) PICK(bf16, true); else PICK(bf16, false);
  }
#undef PICK
#undef GO
  return -1;
}

}  // namespace smi
