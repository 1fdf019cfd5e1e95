// UNet2DCondition engine: executes the SD-1.x / SD-XL denoising forward and the activation/LoRA-gradient backward
// as a stream of hand-written gfx950 kernels (gemm.hip, attention.hip, norm.hip, elementwise.hip, lora.hip).
//
// Structure
//   * Activations are token-major [n*H*W, C] in the model dtype (fp16/bf16); accumulation is fp32 everywhere.
//   * Frozen weights are borrowed from the caller in torch layout ([out, in] is already the K-contiguous "NT" operand)
//     and complemented at creation by packed copies inside the workspace: transposed matrices for the dX GEMMs,
//     (ky,kx,ci)-ordered conv filters (+ flipped/transposed gradient filters), fused q|k|v and k|v projections.
//   * Memory: two bump arenas sized by a dry run of the same code (no reuse inside a pass -- 288 GB of HBM makes
//     that the simple and fast choice): arena 0 serves no-grad passes, arena 1 the pass that is differentiated and
//     its backward, so a frozen pass can never clobber saved activations.
//   * Autograd: every op pushes a closure on a tape when its output depends on an adapted layer; the backward pops
//     them in reverse.  Gradients exist only downstream of the first adapted layer; frozen weights get none.
//     Residual / skip fan-in is accumulated inside the producing kernel (GEMM `res` epilogue, norm `add` operand).
//   * 16-bit gradient range: d_eps is multiplied by a power-of-two loss scale chosen on the device from max|d_eps|;
//     the LoRA weight-gradient kernels divide it out (they accumulate in fp32).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>
#include <deque>

#include "../../include/smi.h"
#include "kernels.h"

namespace smi {

static thread_local char g_err[2048] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int launch_nchw_to_nhwc_scaled(int dtype, const float* src, void* dst, int Nb, int C, int HW, int Cpad,
                               const float* scale_dev, hipStream_t stream);

namespace {

// ------------------------------------------------------------------------------------------------------------
// weight packing kernels (run once at creation)
// ------------------------------------------------------------------------------------------------------------
// dst[c][col0 + r] = src[r][c] through a 64x64 LDS tile: coalesced reads and writes
template <typename T>
__global__ __launch_bounds__(256) void pack_transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int R,
                                                             int C, int ldd, int col0) {
  __shared__ T tile[64][66];
  const int tilesC = (C + 63) / 64;
  const int tr = (blockIdx.x / tilesC) * 64, tc = (blockIdx.x % tilesC) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = tr + i, c = tc + tx;
    if (r < R && c < C) tile[i][tx] = src[(int64_t)r * C + c];
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = tc + i, r = tr + tx;
    if (r < R && c < C) dst[(int64_t)c * ldd + col0 + r] = tile[tx][i];
  }
}
// conv weight [Cout, Cin, 3, 3] -> forward pack  dst[co][(ky*3+kx)*Cin + ci]
//                               -> gradient pack dst[ci][(ty*3+tx)*Cout + co] = w[co][ci][ky][kx],
//                                  (ty,tx) = flip ? (2-ky, 2-kx) : (ky, kx)
template <typename T>
__global__ void pack_conv_kernel(const T* __restrict__ src, T* __restrict__ dst, int Cout, int Cin, int mode,
                                 int pad) {  // pad: padded inner channel count (Cin for mode 0, Cout otherwise)
  const int64_t total = (int64_t)Cout * Cin * 9;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % 9);
    const int ci = (int)((i / 9) % Cin);
    const int co = (int)(i / (9 * (int64_t)Cin));
    const int ky = t / 3, kx = t % 3;
    if (mode == 0) {
      dst[((int64_t)co * 9 + t) * pad + ci] = src[i];
    } else {
      const int tt = mode == 2 ? (2 - ky) * 3 + (2 - kx) : t;
      dst[((int64_t)ci * 9 + tt) * pad + co] = src[i];
    }
  }
}
// up to 64 floats handed over BY VALUE in the kernel arguments (copied into the kernarg segment at launch), written to one
// or two device tables: host values reach the device without a host buffer that must outlive the call
struct F64Args {
  float v[64];
};
__global__ void set_floats_kernel(float* dst0, float* dst1, F64Args a, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    dst0[i] = a.v[i];
    if (dst1) dst1[i] = a.v[i];
  }
}
__global__ void fill_kernel(float* p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool overflow = false;
  void reset() { off = 0; }
  void* alloc(size_t bytes) {
    off = align_up(off, 256);
    void* p = base + off;
    off += bytes;
    if (off > peak) peak = off;
    if (off > cap) overflow = true;
    return p;
  }
};

struct Ten {
  void* p = nullptr;
  void* g = nullptr;
  int64_t rows = 0;
  int cols = 0;
  int n = 0, H = 0, W = 0;  // rows = n*H*W for image tensors
  // first row of the ADAPTED samples: a batched pass carries frozen samples first and the adapted (LoRA on,
  // differentiated) samples last; gradients `g` cover rows [arow0, rows) only
  int64_t arow0 = 0;
  bool ng = false;
  float gmul = 1.f;  // the stored gradient g is gmul x the (loss-scaled) gradient; a power of two, 1 but for row vectors
};

struct Lin {
  std::string name;
  const void* W = nullptr;   // [out, in]
  const void* Wt = nullptr;  // [in, out]
  const void* b = nullptr;
  int in = 0, out = 0;
  // LoRA (fused projections carry nseg adjacent sites)
  int nseg = 1;
  int nsite = 0;
  int rank = 0;
  float scale = 0.f;
  int64_t off_down = 0, off_up = 0;
  // 16-bit GEMM-operand copies of the LoRA matrices, refreshed once per adapted forward (lora.hip lora_prep_kernel)
  int rows_pad = 0;
  const void* sh_down = nullptr;  // [rows_pad, in]
  const void* sh_up = nullptr;    // [rows_pad, out] block-diagonal over the fused segments
  // DoRA (T/dora.py): the adapted rows get a second GEMM against dW = lscale * ((W + up down) g / ||.||_col - W)
  bool dora = false;
  int dora_idx = -1;  // into smi_engine::dora_sites
};
struct Conv {
  const void* Wp = nullptr;   // forward pack [Cout][9*Cin]
  const void* Wg = nullptr;   // gradient pack [Cin][9*Cout]
  const void* b = nullptr;
  int Cin = 0, Cout = 0;
  int mode = 0;  // 0 stride 1, 1 stride-2 downsample, 2 nearest-2x upsample + conv
  int pad = 1;   // 0: the VAE encoder's downsampler (input padded (0,1,0,1), stride 2)
  // c3lier LoRA (T/lora.py:100-114): 3x3 down conv [r][Cin][3][3] with the layer's stride / padding, then 1x1 up [Cout][r]
  std::string name;
  int nsite = 0, rank = 0, rows_pad = 0;
  float scale = 0.f;
  int64_t off_down = 0, off_up = 0;
  const void* sh_down = nullptr;  // 16-bit [rows_pad][9*Cin], (ky,kx,ci)-ordered: filter of the xa conv
  const void* sh_up = nullptr;    // 16-bit [rows_pad][Cout] = up^T (dxa = dy * up)
  const void* sh_gw = nullptr;    // 16-bit [Cin][9*64]: gradient filter of the down conv, rank zero-padded to 64
};
struct Norm {
  const void* gamma = nullptr;
  const void* beta = nullptr;
  int C = 0;
  float eps = 1e-5f;
};
struct Resnet {
  Norm n1, n2;
  Conv c1, c2;
  Lin temb, sc;
  bool has_sc = false;
  bool has_temb = true;  // false in the VAE encoder
  int64_t temb_col = 0;  // column offset of this resnet's time_emb_proj inside the grouped projection
};
struct TBlock {
  Norm n1, n2, n3;
  Lin qkv1, out1, q2, kv2, out2, ff1, ff2;
  int64_t kv_col = 0;  // column offset of this block's k|v inside the grouped projection
};
struct Transformer {
  Norm norm;
  Lin proj_in, proj_out;
  std::vector<TBlock> blocks;
  int heads = 0, C = 0;
};
struct Level {
  std::vector<Resnet> res;
  std::vector<Transformer> att;
  bool has_att = false;
  bool has_samp = false;
  Conv samp;
};

}  // namespace
}  // namespace smi

using namespace smi;

struct smi_engine {
  smi_unet_config cfg{};
  int dtype = 0;
  hipStream_t stream = nullptr;
  // LoRA weight gradients are DEFERRED: every adapted Linear appends its rank-r reductions to `wjobs` while the
  // backward runs and one grouped launch at the end does them all (lora.hip).  The arenas never recycle memory inside a
  // backward, so x, xa and dxa are still there; the one operand at risk is dY, which can live on as another tensor's
  // gradient (accumulate() aliases it) and would then be added to IN PLACE: such buffers are `pinned`, and a write
  // that would land on a pinned buffer goes to a fresh one instead (grad_slot / accumulate).
  std::vector<WgradJob> wjobs;
  std::vector<WgradJob> wjobs_uploaded;  // what the device table holds (same addresses every step of a plan: no re-upload)
  WgradJob* wjobs_dev = nullptr;
  static constexpr size_t SPLITK_WS_BYTES = (size_t)512 * 128 * 128 * sizeof(float);  // 33.5 MB
  void* splitk_ws = nullptr;
  // per-sample adaptor multipliers (smi_unet_forward_multi): sigma_i = m_i / m_ref per ADAPTED sample, applied to the rows
  // of xa = x down^T (forward) and dxa = dy up (backward) -- every other use of the multiplier stays the scalar m_ref.
  // Slot 0 serves the pass in flight, slot 1 keeps the saved pass's values for its backward.
  float* samp_mult_dev = nullptr;  // [2][MAXS]
  bool samp_on = false, bw_samp_on = false;
  int n_conv_sites = 0;
  const float* samp_ptr(bool bwd) const {
    return (bwd ? bw_samp_on : samp_on) ? samp_mult_dev + (bwd ? MAXS : 0) : nullptr;
  }
  size_t wjobs_cap = 0;
  std::vector<const void*> pinned;
  WgradJob& push_wjob(const void* X, int64_t ldx, const float* P, int64_t ldp, float* dW, int64_t so_r, int64_t so_k,
                      int M, int K, int rr, int seg_cols, int rows_per_sample, float alpha) {
    WgradJob j;
    memset(&j, 0, sizeof(j));  // padding too: the table is compared bytewise with the uploaded copy
    j.X = X;
    j.ldx = ldx;
    j.P = P;
    j.ldp = ldp;
    j.dW = dW;
    j.so_r = so_r;
    j.so_k = so_k;
    j.M = M;
    j.K = K;
    j.r = rr;
    j.seg_cols = seg_cols;
    j.rows_per_sample = rows_per_sample;
    j.row_scale = gscale + MAXS;
    j.alpha = alpha;
    j.conv_tap = -1;
    wgrad_job_plan(j);
    j.partial = alloc_f32(wgrad_job_scratch_floats(j));
    wjobs.push_back(j);
    return wjobs.back();
  }
  bool is_pinned(const void* g) const {
    for (const void* q : pinned)
      if (q == g) return true;
    return false;
  }
  bool dry = false;
  bool err = false;
  int max_n = 0, lat_h = 0, lat_w = 0, ctx_len = 0;

  // workspace layout
  char* ws = nullptr;
  size_t ws_bytes = 0;
  char* arena_home = nullptr;  // the arena region inside the creation workspace (smi_replan with arena == NULL)
  size_t arena_home_bytes = 0;
  Arena wpack;     // packed weights + persistent small buffers
  Arena arena[2];  // 0: no-grad passes, 1: differentiated pass + backward
  Arena* cur = nullptr;
  // per-sample power-of-two loss scales: [0, MAXS) scale of adapted sample j, [MAXS, 2 MAXS) its inverse, then scratch
  static constexpr int MAXS = 1024;
  float* gscale = nullptr;

  std::unordered_map<std::string, const smi_weight*> wmap;
  std::unordered_map<std::string, const smi_lora_site*> smap;
  std::vector<smi_lora_site> sites;
  std::vector<std::string> site_names;
  std::vector<char> site_used;

  // DoRA sites: delta weights [out, in] (+ their transposes for dX) and column norms, rebuilt per adapted forward
  std::vector<DoraSite> dora_sites;
  std::vector<void*> dora_dWt;  // [in, out] per site
  DoraSite* dora_sites_dev = nullptr;
  float bw_mult = 0.f;
  // what the delta-weight buffers currently hold: set by dora_prepare, so that the backward of a saved forward does not
  // rebuild what that forward built (nothing ran in between); `with_t`: the transposed copies for the dX GEMMs exist too
  const float* dora_ready_down = nullptr;
  const float* dora_ready_up = nullptr;
  float dora_ready_mult = 0.f;
  bool dora_ready_t = false;
  void dora_prepare(const float* down, const float* up, float m, bool with_t, bool reuse) {
    if (dora_sites.empty() || dry || err) return;
    const bool same = reuse && dora_ready_down == down && dora_ready_up == up && dora_ready_mult == m;
    if (!same) {
      if (launch_dora_prep(dtype, dora_sites_dev, dora_sites.data(), (int)dora_sites.size(), down, up, m, stream) != 0) {
        err = true;
        return;
      }
      dora_ready_down = down;
      dora_ready_up = up;
      dora_ready_mult = m;
      dora_ready_t = false;
    }
    if (with_t && !dora_ready_t) {  // a no-grad pass (pre-roll) never reads the transposed copies
      if (launch_dora_transpose(dtype, dora_sites_dev, dora_sites.data(), (int)dora_sites.size(), stream) != 0) {
        err = true;
        return;
      }
      dora_ready_t = true;
    }
  }
  // LoRA shadow operands
  std::vector<HostLoraPrepSite> prep_sites;
  void* prep_sites_dev = nullptr;
  char* lora_shadow = nullptr;
  size_t lora_shadow_elems = 0;
  // grouped cross-attention K|V projection: every block's to_k|to_v stacked into one [sum N, D_ctx] matrix
  Lin kv_all;
  bool kv_grouped = false;
  Ten* kv_all_out = nullptr;
  Lin temb_all;  // every resnet's time_emb_proj stacked (they all read silu(emb)): one GEMM per pass
  bool temb_grouped = false;
  Ten* temb_all_out = nullptr;

  // model
  Conv conv_in, conv_out;
  Norm norm_out;
  Lin time1, time2, add1, add2;
  std::vector<Level> down, up;
  Level mid;

  // per-call state
  std::deque<Ten> tens_[2];  // per arena: a no-grad pass must not invalidate the tensors the tape refers to
  std::deque<Ten>* tens = &tens_[0];
  std::vector<std::function<void()>> tape;
  bool saving = false;
  bool tape_valid = false;
  // every saved forward gets a generation number; the caller keeps it with the output it will differentiate and
  // smi_unet_backward_checked refuses a stale one (a later saved forward, a replan or a backward dropped the tape)
  uint64_t tape_gen = 0;
  int64_t pack_launches = 0;  // weight-packing kernels launched so far (creation only: a replan must not add any)
  int64_t replans = 0;
  const float* lora_down = nullptr;  // parameters of the forward in flight
  const float* lora_up = nullptr;
  const float* bw_down = nullptr;    // parameters of the saved (differentiated) forward, used by its backward
  const float* bw_up = nullptr;
  float mult = 0.f;
  float* d_down = nullptr;
  float* d_up = nullptr;
  Ten* out_ten = nullptr;  // conv_out result (f32 [rows, 4])
  int n_ad = 0;     // adapted samples of the pass in flight (the last n_ad of n)
  int bw_n_ad = 0;  // ... of the saved pass
  int max_n_ad = 0;

  size_t esz() const { return 2; }

  // ---------------------------------------------------------------------------------------------------------
  void fail(const char* what) {
    if (!err) set_error("%s: %s", what, g_err);
    err = true;
  }
#define RUN(call)                          \
  do {                                     \
    if (!dry && !err) {                    \
      if ((call) != 0) {                   \
        err = true;                        \
      }                                    \
      bound_queue_depth();                 \
    }                                      \
  } while (0)
  // same, attributing the launch (and its algorithmic FLOPs / HBM bytes) to a kernel class for smi_profile_*
#define RUNP(cat, flops, bytes, call)      \
  do {                                     \
    if (!dry && !err) {                    \
      prof_begin(cat, flops, bytes);       \
      if (trace_launches) trace_before(#call); \
      if ((call) != 0) {                   \
        err = true;                        \
      }                                    \
      if (trace_launches) trace_after();   \
      prof_end();                          \
      bound_queue_depth();                 \
    }                                      \
  } while (0)
  // SMI_TRACE_LAUNCH=1 (debugging a hang): names every launch on stderr and waits for it, so the last line printed is the
  // launch that never came back
  bool trace_launches = getenv("SMI_TRACE_LAUNCH") != nullptr;
  // SMI_SYNC_EVERY=N: wait for the stream after every N launches, i.e. never more than N (x 3 with the profiling events)
  // AQL packets of ours outstanding.  For `rocprofv3 --pmc` runs: counter collection serialises dispatches behind its own
  // start / stop / read packets, the HIP runtime's queue backs up at our launch rate, and with some hundred packets
  // outstanding the profiler's intercept queue aborted once on a packet it had not finished rewriting
  int sync_every = getenv("SMI_SYNC_EVERY") ? atoi(getenv("SMI_SYNC_EVERY")) : 0;
  int launches_since_sync = 0;
  void bound_queue_depth() {
    if (sync_every > 0 && ++launches_since_sync >= sync_every) {
      launches_since_sync = 0;
      (void)hipStreamSynchronize(stream);
    }
  }
  void trace_before(const char* what) {
    fprintf(stderr, "[smi launch] %.110s ...", what);
    fflush(stderr);
  }
  void trace_after() {
    const hipError_t e = hipStreamSynchronize(stream);
    fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
    fflush(stderr);
  }

  // ---- per-kernel-class timing with HIP events on the engine's stream (profiling mode only)
  bool prof_on = false;
  struct ProfEv {
    int cat;
    hipEvent_t a, b;
    std::string tag;
    double flops;
  };
  // SMI_PROF_DUMP=1: per-shape table of the GEMM / conv launches of the profiled step on stderr (smi_profile_read)
  struct TagAcc {
    double ms = 0, flops = 0;
    int n = 0;
  };
  std::unordered_map<std::string, TagAcc> prof_tags;
  bool prof_dump = getenv("SMI_PROF_DUMP") != nullptr;
  std::string gemm_tag(const GemmParams& p) const {
    char b[160];
    snprintf(b, sizeof(b), "%s M=%d N=%d K=%d%s%s%s%s%s%s", p.conv ? "conv" : "gemm", p.M, p.N, p.K, p.bias ? " bias" : "",
             p.res ? " res" : "", p.lora_r ? " lora" : "", p.geglu_out ? " geglu" : "", p.rowvec ? " rowvec" : "",
             p.out_f32 ? " f32" : "");
    return b;
  }
  std::string next_tag;
  std::vector<ProfEv> prof_ev;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[SMI_PROF_NCAT] = {0};
  double prof_flops[SMI_PROF_NCAT] = {0};
  double prof_bytes[SMI_PROF_NCAT] = {0};
  int64_t prof_launches[SMI_PROF_NCAT] = {0};
  hipEvent_t prof_event() {
    if (!prof_pool.empty()) {
      hipEvent_t e = prof_pool.back();
      prof_pool.pop_back();
      return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
  void prof_begin(int cat, double flops, double bytes) {
    if (!prof_on) return;
    ProfEv pe;
    pe.cat = cat;
    pe.a = prof_event();
    pe.b = prof_event();
    pe.flops = flops;
    (void)hipEventRecord(pe.a, stream);
    prof_ev.push_back(pe);
    prof_flops[cat] += flops;
    prof_bytes[cat] += bytes;
    prof_launches[cat] += 1;
  }
  void prof_end() {
    if (!prof_on) return;
    prof_ev.back().tag.swap(next_tag);  // set while the call's arguments were evaluated
    next_tag.clear();
    (void)hipEventRecord(prof_ev.back().b, stream);
  }
  void prof_collect() {  // host-synchronising
    for (auto& pe : prof_ev) {
      (void)hipEventSynchronize(pe.b);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, pe.a, pe.b);
      prof_ms[pe.cat] += ms;
      if (prof_dump && !pe.tag.empty()) {
        TagAcc& t = prof_tags[pe.tag];
        t.ms += ms;
        t.flops += pe.flops;
        t.n += 1;
      }
      prof_pool.push_back(pe.a);
      prof_pool.push_back(pe.b);
    }
    prof_ev.clear();
  }

  // ---- liveness in NO-GRAD passes (arena 0: pre-roll forwards, frozen-only passes).  Nothing is saved there, so a tensor
  // is dead once the block after its producer has run.  Allocation inside a block (resnet [+ transformer], sampler conv,
  // the final norm + conv_out) comes from one of two scratch regions that alternate per block -- block k writes region
  // k & 1 while it reads block k - 1's output in the other one, and re-starts the region block k - 2 used -- except the
  // block's OUTPUT when the up path will read it again as a skip connection: that one tensor, and everything allocated
  // outside blocks (embeddings, grouped projections, conv_in), goes to the persistent bump region.  Arena 0 is then
  // persistent + 2 x the largest block instead of the sum of all tensors of a pass (SD-XL 1024^2, 16 samples: 80 -> ~10 GB);
  // the saved pass (arena 1) keeps every tensor: its backward reads them.
  Arena scr[2];
  int scr_i = 0;
  bool in_block = false, persist_next = false;
  void begin_block() {
    if (saving || is_vae || is_clip) return;
    scr_i ^= 1;
    scr[scr_i].reset();
    in_block = true;
  }
  void end_block() { in_block = false; persist_next = false; }
  void set_arena0_regions(size_t persistent, size_t scratch) {  // sizes from the dry run (plan)
    arena[0].cap = persistent;
    for (int k = 0; k < 2; ++k) {
      scr[k] = Arena();
      scr[k].base = arena[0].base + persistent + k * scratch;
      scr[k].cap = scratch;
    }
  }
  // the next tensor created is the current block's output and must outlive the scratch regions (a skip connection)
  void keep_next_output() { persist_next = in_block; }

  Ten* new_ten(int64_t rows, int cols, int n = 0, int H = 0, int W = 0, size_t elt = 0) {
    tens->emplace_back();
    Ten* t = &tens->back();
    t->rows = rows;
    t->cols = cols;
    t->n = n;
    t->H = H;
    t->W = W;
    t->arow0 = n > 0 ? (int64_t)(n - n_ad) * (rows / n) : 0;
    const bool keep = persist_next;
    persist_next = false;
    t->p = arena_alloc((size_t)rows * cols * (elt ? elt : esz()), keep);
    return t;
  }
  int64_t MA(const Ten* t) const { return t->rows - t->arow0; }  // adapted rows
  char* PA(const Ten* t) const { return (char*)t->p + (size_t)t->arow0 * t->cols * esz(); }
  void* arena_alloc(size_t bytes, bool persistent = false) {
    Arena* a = (in_block && !persistent) ? &scr[scr_i] : cur;
    void* p = a->alloc(bytes);
    if (!dry && a->overflow && !err) {  // never launch a kernel on memory we do not own
      set_error("workspace arena overflow (%zu > %zu bytes)", a->off, a->cap);
      err = true;
    }
    return p;
  }
  float* alloc_f32(size_t count) { return (float*)arena_alloc(count * sizeof(float)); }
  void* alloc_t(int64_t rows, int cols) { return arena_alloc((size_t)rows * cols * esz()); }

  // gradient slot of t: `out` is where the producer writes, `add` (may be null, may equal out) what is already
  // accumulated there and has to be added in the same pass.  Out of place when the old buffer is pinned.
  struct Slot {
    void* add;
    void* out;
  };
  Slot grad_slot(Ten* t) {
    Slot s;
    if (!t->g) {
      s.add = nullptr;
      s.out = t->g = alloc_t(MA(t), t->cols);
    } else if (is_pinned(t->g)) {
      s.add = t->g;
      s.out = t->g = alloc_t(MA(t), t->cols);
    } else {
      s.add = s.out = t->g;
    }
    return s;
  }
  void accumulate(Ten* t, void* g) {
    if (!t->g) {
      t->g = g;  // alias: g is dead after its producer's closure (unless pinned: then nobody writes to it)
    } else {
      Slot s = grad_slot(t);
      RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_add(dtype, s.add, g, s.out, MA(t) * t->cols, stream));
    }
  }

  // ---------------------------------------------------------------------------------------------------------
  // model construction
  // ---------------------------------------------------------------------------------------------------------
  const smi_weight* W(const std::string& name, bool required = true) {
    auto it = wmap.find(name);
    if (it == wmap.end()) {
      if (required && !dry) {
        set_error("missing weight '%s'", name.c_str());
        err = true;
      }
      return nullptr;
    }
    return it->second;
  }
  const void* Wd(const std::string& name, bool required = true) {
    const smi_weight* w = W(name, required);
    return w ? w->data : nullptr;
  }
  void check_shape(const std::string& name, std::initializer_list<int64_t> shp) {
    if (dry) return;
    const smi_weight* w = W(name);
    if (!w) return;
    int64_t want = 1, have = 1;
    for (auto v : shp) want *= v;
    for (int i = 0; i < w->ndim; ++i) have *= w->shape[i];
    if (want != have) {
      set_error("weight '%s': expected %lld elements, got %lld", name.c_str(), (long long)want, (long long)have);
      err = true;
    }
  }
  void* pack_alloc(size_t bytes) { return wpack.alloc(bytes); }

  void transpose_into(const void* src, void* dst, int R, int C, int ldd, int col0, bool packing = true) {
    if (dry || err || !src) return;
    const int grid = ((R + 63) / 64) * ((C + 63) / 64);
    if (packing) ++pack_launches;
    if (dtype == DT_F16)
      hipLaunchKernelGGL(pack_transpose_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)src, (f16*)dst, R, C, ldd, col0);
    else
      hipLaunchKernelGGL(pack_transpose_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)src, (bf16*)dst, R, C, ldd, col0);
  }
  void pack_conv_into(const void* src, void* dst, int Cout, int Cin, int mode, int pad = 0) {
    if (dry || err || !src) return;
    if (pad == 0) pad = mode == 0 ? Cin : Cout;
    else (void)hipMemsetAsync(dst, 0, (size_t)(mode == 0 ? Cout : Cin) * 9 * pad * esz(), stream);
    const int64_t total = (int64_t)Cout * Cin * 9;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 8192);
    ++pack_launches;
    if (dtype == DT_F16)
      hipLaunchKernelGGL(pack_conv_kernel<f16>, dim3(grid), dim3(256), 0, stream, (const f16*)src, (f16*)dst, Cout, Cin, mode, pad);
    else
      hipLaunchKernelGGL(pack_conv_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)src, (bf16*)dst, Cout, Cin, mode, pad);
  }

  void attach_lora(Lin& L, const std::vector<std::string>& targets) {
    // targets: the module paths fused into this GEMM (1, 2 (k|v) or 3 (q|k|v)); all adapted or none
    int found = 0;
    const smi_lora_site* first = nullptr;
    for (size_t i = 0; i < targets.size(); ++i) {
      auto it = smap.find(targets[i]);
      if (it == smap.end()) continue;
      ++found;
      const smi_lora_site* s = it->second;
      site_used[s - sites.data()] = 1;
      if (!first) first = s;
      if (i > 0 && found == (int)i + 1) {
        const int cs = L.out / (int)targets.size();
        if (s->rank != first->rank || s->scale != first->scale ||
            s->off_down != first->off_down + (int64_t)i * first->rank * L.in ||
            s->off_up != first->off_up + (int64_t)i * cs * first->rank) {
          set_error("LoRA sites fused into '%s' must be adjacent in the flat buffers with equal rank/scale",
                    L.name.c_str());
          err = true;
        }
      }
    }
    if (found == 0) return;
    if (found != (int)targets.size()) {
      set_error("LoRA must adapt all or none of the projections fused into '%s'", L.name.c_str());
      err = true;
      return;
    }
    L.nsite = found;
    L.rank = first->rank;
    L.scale = first->scale;
    L.off_down = first->off_down;
    L.off_up = first->off_up;
    if (first->off_dora >= 0) {  // DoRA site(s): all fused segments must be DoRA with adjacent scale vectors
      for (size_t i = 0; i < targets.size(); ++i) {
        const smi_lora_site* si = smap.find(targets[i])->second;
        if (si->off_dora != first->off_dora + (int64_t)i * L.in) {
          set_error("DoRA sites fused into '%s' must have adjacent dora_scale vectors", L.name.c_str());
          err = true;
          return;
        }
      }
      if (L.rank > 32) {
        set_error("DoRA rank %d > 32 on '%s'", L.rank, L.name.c_str());
        err = true;
        return;
      }
      L.dora = true;
      L.dora_idx = (int)dora_sites.size();
      DoraSite d{};
      d.W = L.W;
      d.off_down = L.off_down;
      d.off_up = L.off_up;
      d.off_dora = first->off_dora;
      d.r = L.rank;
      d.nseg = L.nseg;
      d.K = L.in;
      d.cs = L.out / L.nseg;
      d.scale = L.scale;
      d.dW = pack_alloc((size_t)L.out * L.in * esz());
      d.cnorm = (float*)pack_alloc((size_t)L.nseg * L.in * sizeof(float));
      d.dWt = pack_alloc((size_t)L.out * L.in * esz());
      dora_sites.push_back(d);
      dora_dWt.push_back(d.dWt);
      return;  // no 16-bit shadow operands: the delta is a dense second GEMM
    }
    const int rtot = L.rank * L.nseg;
    L.rows_pad = (rtot + 15) / 16 * 16;
    HostLoraPrepSite ps{};
    ps.off_down = L.off_down;
    ps.off_up = L.off_up;
    ps.dst_down = (int64_t)lora_shadow_elems;
    lora_shadow_elems += (size_t)L.rows_pad * L.in;
    ps.dst_up = (int64_t)lora_shadow_elems;
    lora_shadow_elems += (size_t)L.rows_pad * L.out;
    ps.r = L.rank;
    ps.nseg = L.nseg;
    ps.K = L.in;
    ps.cs = L.out / L.nseg;
    ps.rows_pad = L.rows_pad;
    prep_sites.push_back(ps);
    L.sh_down = reinterpret_cast<const void*>((uintptr_t)ps.dst_down);  // offsets for now; rebased in finish_lora()
    L.sh_up = reinterpret_cast<const void*>((uintptr_t)ps.dst_up);
  }
  void attach_lora_conv(Conv& c) {
    auto it = smap.find(c.name);
    if (it == smap.end()) return;
    const smi_lora_site* s = it->second;
    site_used[s - sites.data()] = 1;
    ++n_conv_sites;
    c.nsite = 1;
    c.rank = s->rank;
    c.scale = s->scale;
    c.off_down = s->off_down;
    c.off_up = s->off_up;
    c.rows_pad = (c.rank + 15) / 16 * 16;
    if (c.rank > 64) {
      set_error("conv LoRA rank %d > 64 on '%s'", c.rank, c.name.c_str());
      err = true;
      return;
    }
    HostLoraPrepSite ps{};
    ps.off_down = c.off_down;
    ps.off_up = c.off_up;
    ps.dst_down = (int64_t)lora_shadow_elems;
    lora_shadow_elems += (size_t)c.rows_pad * 9 * c.Cin;
    ps.dst_up = (int64_t)lora_shadow_elems;
    lora_shadow_elems += (size_t)c.rows_pad * c.Cout;
    ps.dst_gw = (int64_t)lora_shadow_elems;
    lora_shadow_elems += (size_t)c.Cin * 9 * 64;
    ps.r = c.rank;
    ps.nseg = 1;
    ps.K = c.Cin;
    ps.cs = c.Cout;
    ps.rows_pad = c.rows_pad;
    ps.conv = c.mode == 1 ? 2 : 1;  // 1: flipped taps in the gradient filter, 2: stride-2 (unflipped, transposed gather)
    prep_sites.push_back(ps);
    c.sh_down = reinterpret_cast<const void*>((uintptr_t)ps.dst_down);
    c.sh_up = reinterpret_cast<const void*>((uintptr_t)ps.dst_up);
    c.sh_gw = reinterpret_cast<const void*>((uintptr_t)ps.dst_gw);
  }
  void finish_lora() {  // after build(): all Lin objects are at their final addresses
    dora_sites_dev = (DoraSite*)pack_alloc(std::max<size_t>(dora_sites.size(), 1) * sizeof(DoraSite));
    if (!dry && !err && !dora_sites.empty())
      (void)hipMemcpyAsync(dora_sites_dev, dora_sites.data(), dora_sites.size() * sizeof(DoraSite), hipMemcpyHostToDevice,
                           stream);
    lora_shadow = (char*)pack_alloc(std::max<size_t>(lora_shadow_elems, 8) * esz());
    prep_sites_dev = pack_alloc(std::max<size_t>(prep_sites.size(), 1) * sizeof(HostLoraPrepSite));
    if (!dry && !err && !prep_sites.empty())
      (void)hipMemcpyAsync(prep_sites_dev, prep_sites.data(), prep_sites.size() * sizeof(HostLoraPrepSite),
                           hipMemcpyHostToDevice, stream);
  }
  const void* shadow_ptr(const void* off) const { return lora_shadow + (uintptr_t)off * 2; }

  // need_t: the layer lies on the gradient path (gets a transposed copy) and may carry a LoRA adaptor
  // attach_only: the layer may carry a LoRA adaptor although no gradient flows to its input (time_emb_proj)
  Lin make_lin(const std::string& name, int in, int out, bool bias, bool need_t = true, bool attach_only = false) {
    Lin L;
    L.name = name;
    L.in = in;
    L.out = out;
    L.W = Wd(name + ".weight");
    check_shape(name + ".weight", {out, in});
    L.b = bias ? Wd(name + ".bias") : nullptr;
    if (need_t) {
      void* t = pack_alloc((size_t)in * out * esz());
      transpose_into(L.W, t, out, in, out, 0);
      L.Wt = t;
      attach_lora(L, {name});
    } else if (attach_only) {
      attach_lora(L, {name});
    }
    return L;
  }
  // fused projection: rows of the parts concatenated ([sum out, in]); Wt [in, sum out]
  Lin make_fused(const std::string& base, const std::vector<std::string>& parts, int in, int out_each, bool need_t) {
    Lin L;
    L.name = base + ".{" + parts[0] + "..}";
    L.in = in;
    L.out = out_each * (int)parts.size();
    L.nseg = (int)parts.size();
    char* w = (char*)pack_alloc((size_t)L.out * in * esz());
    char* t = need_t ? (char*)pack_alloc((size_t)L.out * in * esz()) : nullptr;
    std::vector<std::string> targets;
    for (size_t i = 0; i < parts.size(); ++i) {
      const std::string nm = base + "." + parts[i];
      const void* src = Wd(nm + ".weight");
      check_shape(nm + ".weight", {out_each, in});
      if (!dry && !err && src)
        (void)hipMemcpyAsync(w + i * (size_t)out_each * in * esz(), src, (size_t)out_each * in * esz(),
                             hipMemcpyDeviceToDevice, stream);
      if (t) transpose_into(src, t, out_each, in, L.out, (int)i * out_each);
      targets.push_back(nm);
    }
    L.W = w;
    L.Wt = t;
    attach_lora(L, targets);
    return L;
  }
  Conv make_conv(const std::string& name, int Cin, int Cout, int mode, bool grad_pack) {
    Conv c;
    c.Cin = Cin;
    c.Cout = Cout;
    c.mode = mode;
    c.b = Wd(name + ".bias");
    c.name = name;
    check_shape(name + ".weight", {Cout, Cin, 3, 3});
    attach_lora_conv(c);
    void* wp = pack_alloc((size_t)Cout * Cin * 9 * esz());
    pack_conv_into(Wd(name + ".weight"), wp, Cout, Cin, 0);
    c.Wp = wp;
    if (grad_pack) {
      void* wg = pack_alloc((size_t)Cout * Cin * 9 * esz());
      pack_conv_into(Wd(name + ".weight"), wg, Cout, Cin, mode == 1 ? 1 : 2);
      c.Wg = wg;
    }
    return c;
  }
  Norm make_norm(const std::string& name, int C, float eps) {
    Norm n;
    n.C = C;
    n.eps = eps;
    n.gamma = Wd(name + ".weight");
    n.beta = Wd(name + ".bias");
    check_shape(name + ".weight", {C});
    return n;
  }
  // grad: the block lies on a gradient path (gradient filter packs, transposed shortcut); eps 1e-5 UNet / 1e-6 VAE
  Resnet make_resnet(const std::string& name, int Cin, int Cout, bool temb = true, float eps = 1e-5f, bool grad = true) {
    Resnet r;
    const int ted = cfg.block_out_channels[0] * 4;
    r.n1 = make_norm(name + ".norm1", Cin, eps);
    r.c1 = make_conv(name + ".conv1", Cin, Cout, 0, grad);
    r.has_temb = temb;
    if (temb) r.temb = make_lin(name + ".time_emb_proj", ted, Cout, true, false, true);
    r.n2 = make_norm(name + ".norm2", Cout, eps);
    r.c2 = make_conv(name + ".conv2", Cout, Cout, 0, grad);
    r.has_sc = Cin != Cout;
    if (r.has_sc) r.sc = make_lin(name + ".conv_shortcut", Cin, Cout, true, grad);
    return r;
  }
  Transformer make_transformer(const std::string& name, int C, int heads, int layers) {
    Transformer t;
    t.C = C;
    t.heads = heads;
    t.norm = make_norm(name + ".norm", C, 1e-6f);
    t.proj_in = make_lin(name + ".proj_in", C, C, true);
    for (int k = 0; k < layers; ++k) {
      const std::string b = name + ".transformer_blocks." + std::to_string(k);
      TBlock tb;
      tb.n1 = make_norm(b + ".norm1", C, 1e-5f);
      tb.qkv1 = make_fused(b + ".attn1", {"to_q", "to_k", "to_v"}, C, C, true);
      tb.out1 = make_lin(b + ".attn1.to_out.0", C, C, true);
      tb.n2 = make_norm(b + ".norm2", C, 1e-5f);
      tb.q2 = make_lin(b + ".attn2.to_q", C, C, false);
      tb.kv2 = make_fused(b + ".attn2", {"to_k", "to_v"}, cfg.cross_attention_dim, C, false);
      tb.out2 = make_lin(b + ".attn2.to_out.0", C, C, true);
      tb.n3 = make_norm(b + ".norm3", C, 1e-5f);
      tb.ff1 = make_lin(b + ".ff.net.0.proj", C, 8 * C, true);
      tb.ff2 = make_lin(b + ".ff.net.2", 4 * C, C, true);
      t.blocks.push_back(tb);
    }
    t.proj_out = make_lin(name + ".proj_out", C, C, true);
    return t;
  }

  void build() {
    const int L = cfg.n_levels;
    const int* boc = cfg.block_out_channels;
    const int ted = boc[0] * 4;
    // conv_in: direct small-Cin kernel, weights (ky,kx,ci)-ordered
    {  // conv_in on the MFMA conv path: input channels zero-padded to 64 (one K step per filter tap)
      conv_in.Cin = 64;
      conv_in.Cout = boc[0];
      conv_in.b = Wd("conv_in.bias");
      check_shape("conv_in.weight", {boc[0], cfg.in_channels, 3, 3});
      void* wp = pack_alloc((size_t)boc[0] * 9 * 64 * esz());
      pack_conv_into(Wd("conv_in.weight"), wp, boc[0], cfg.in_channels, 0, 64);
      conv_in.Wp = wp;
    }
    time1 = make_lin("time_embedding.linear_1", boc[0], ted, true, false);
    time2 = make_lin("time_embedding.linear_2", ted, ted, true, false);
    if (cfg.addition_embed) {
      add1 = make_lin("add_embedding.linear_1", cfg.projection_class_embeddings_input_dim, ted, true, false);
      add2 = make_lin("add_embedding.linear_2", ted, ted, true, false);
    }
    int out_ch = boc[0];
    down.resize(L);
    for (int i = 0; i < L; ++i) {
      const int in_ch = out_ch;
      out_ch = boc[i];
      Level& lv = down[i];
      lv.has_att = cfg.down_has_attn[i] != 0;
      const std::string b = "down_blocks." + std::to_string(i);
      for (int j = 0; j < cfg.layers_per_block; ++j) {
        lv.res.push_back(make_resnet(b + ".resnets." + std::to_string(j), j == 0 ? in_ch : out_ch, out_ch));
        if (lv.has_att)
          lv.att.push_back(make_transformer(b + ".attentions." + std::to_string(j), out_ch, cfg.num_heads[i],
                                            cfg.transformer_layers[i]));
      }
      lv.has_samp = i != L - 1;
      if (lv.has_samp) lv.samp = make_conv(b + ".downsamplers.0.conv", out_ch, out_ch, 1, true);
    }
    {
      const int ch = boc[L - 1];
      mid.res.push_back(make_resnet("mid_block.resnets.0", ch, ch));
      mid.att.push_back(make_transformer("mid_block.attentions.0", ch, cfg.num_heads[L - 1],
                                         cfg.mid_transformer_layers));
      mid.res.push_back(make_resnet("mid_block.resnets.1", ch, ch));
    }
    up.resize(L);
    out_ch = boc[L - 1];
    for (int i = 0; i < L; ++i) {
      const int prev = out_ch;
      out_ch = boc[L - 1 - i];
      const int in_ch = boc[L - 1 - std::min(i + 1, L - 1)];
      Level& lv = up[i];
      lv.has_att = cfg.up_has_attn[i] != 0;
      const std::string b = "up_blocks." + std::to_string(i);
      const int nl = cfg.layers_per_block + 1;
      for (int j = 0; j < nl; ++j) {
        const int skip = j == nl - 1 ? in_ch : out_ch;
        const int rin = j == 0 ? prev : out_ch;
        lv.res.push_back(make_resnet(b + ".resnets." + std::to_string(j), rin + skip, out_ch));
        if (lv.has_att)
          lv.att.push_back(make_transformer(b + ".attentions." + std::to_string(j), out_ch,
                                            cfg.num_heads[L - 1 - i], cfg.transformer_layers[L - 1 - i]));
      }
      lv.has_samp = i != L - 1;
      if (lv.has_samp) lv.samp = make_conv(b + ".upsamplers.0.conv", out_ch, out_ch, 2, true);
    }
    norm_out = make_norm("conv_norm_out", boc[0], 1e-5f);
    conv_out = make_conv("conv_out", boc[0], cfg.out_channels, 0, false);
    {  // conv_out gradient filter [C0][9*64]: flipped taps, d_eps channels zero-padded to 64 (MFMA conv path)
      void* wg = pack_alloc((size_t)boc[0] * 9 * 64 * esz());
      pack_conv_into(Wd("conv_out.weight"), wg, cfg.out_channels, boc[0], 2, 64);
      conv_out.Wg = wg;
    }
    build_kv_group();
    build_temb_group();
    finish_lora();
    gscale = (float*)pack_alloc((3 * MAXS + 258) * sizeof(float));  // + [2 MAXS..): min, 1/min, min / scale_j (DoRA)
    samp_mult_dev = (float*)pack_alloc(2 * MAXS * sizeof(float));
    splitk_ws = pack_alloc(SPLITK_WS_BYTES);  // fp32 slabs of the split-K launches (gemm.hip: S x tiles <= 384 tiles)
    wjobs_cap = 11 * (sites.size() + 8);  // a conv site pushes 10 jobs (d_up + one d_down job per filter tap)
    wjobs_dev = (WgradJob*)pack_alloc(wjobs_cap * sizeof(WgradJob));
    for (size_t i = 0; i < sites.size(); ++i)
      if (!site_used[i] && !err) {
        set_error("LoRA target '%s' is not a layer this engine adapts (attention projections; with c3lier: resnet "
                  "conv1 / conv2 / time_emb_proj / conv_shortcut and the up / down sampler convs)",
                  site_names[i].c_str());
        err = true;
      }
  }

  // every cross-attention k|v projection reads the same ctx: stack them into ONE GEMM per pass (70 launches of a
  // 3x10-tile grid -> one launch of a 3x1300-tile grid on SD-XL).  Only when no LoRA adapts to_k / to_v.
  template <typename F>
  void for_each_tblock(F f) {
    for (auto& lv : down) for (auto& t : lv.att) for (auto& tb : t.blocks) f(tb);
    for (auto& t : mid.att) for (auto& tb : t.blocks) f(tb);
    for (auto& lv : up) for (auto& t : lv.att) for (auto& tb : t.blocks) f(tb);
  }
  void build_kv_group() {
    int64_t total = 0;
    bool any_lora = false;
    for_each_tblock([&](TBlock& tb) {
      tb.kv_col = total;
      total += tb.kv2.out;
      any_lora |= tb.kv2.nsite > 0;
    });
    kv_grouped = !any_lora && total > 0 && total * (int64_t)cfg.cross_attention_dim * 2 < 0xFFFFFFF0ll;
    if (!kv_grouped) return;
    const int D = cfg.cross_attention_dim;
    char* w = (char*)pack_alloc((size_t)total * D * esz());
    for_each_tblock([&](TBlock& tb) {
      if (!dry && !err && tb.kv2.W)
        (void)hipMemcpyAsync(w + (size_t)tb.kv_col * D * esz(), tb.kv2.W, (size_t)tb.kv2.out * D * esz(),
                             hipMemcpyDeviceToDevice, stream);
    });
    kv_all.name = "grouped attn2.to_k|to_v";
    kv_all.W = w;
    kv_all.in = D;
    kv_all.out = (int)total;
  }

  // every resnet's time_emb_proj reads the same silu(emb) [n, 1280]: stack them into ONE GEMM per pass (SD-1.x: 22
  // launches of n <= 32 rows, each a split-K pair, -> one pair); the conv1 epilogues read their column block as a row
  // vector with a leading dimension.  Only when no adaptor sits on a time_emb_proj (c3lier has one on each).
  template <typename F>
  void for_each_resnet(F f) {
    for (auto& lv : down) for (auto& r : lv.res) f(r);
    for (auto& r : mid.res) f(r);
    for (auto& lv : up) for (auto& r : lv.res) f(r);
  }
  void build_temb_group() {
    int64_t total = 0;
    bool any_lora = false, same_in = true;
    int in = 0;
    for_each_resnet([&](Resnet& r) {
      if (!r.has_temb) return;
      r.temb_col = total;
      total += r.temb.out;
      any_lora |= r.temb.nsite > 0;
      if (in == 0) in = r.temb.in;
      same_in &= r.temb.in == in;
    });
    static const bool off = getenv("SMI_TEMB_GROUP") && atoi(getenv("SMI_TEMB_GROUP")) == 0;
    temb_grouped = !off && !any_lora && same_in && total > 0 && total % 8 == 0;
    if (!temb_grouped) return;
    char* w = (char*)pack_alloc((size_t)total * in * esz());
    char* b = (char*)pack_alloc((size_t)total * esz());
    if (!dry && !err) (void)hipMemsetAsync(b, 0, (size_t)total * esz(), stream);  // (a projection without a bias adds zero)
    for_each_resnet([&](Resnet& r) {
      if (!r.has_temb || dry || err || !r.temb.W) return;
      (void)hipMemcpyAsync(w + (size_t)r.temb_col * in * esz(), r.temb.W, (size_t)r.temb.out * in * esz(),
                           hipMemcpyDeviceToDevice, stream);
      if (r.temb.b)
        (void)hipMemcpyAsync(b + (size_t)r.temb_col * esz(), r.temb.b, (size_t)r.temb.out * esz(),
                             hipMemcpyDeviceToDevice, stream);
    });
    temb_all.name = "grouped time_emb_proj";
    temb_all.W = w;
    temb_all.b = b;
    temb_all.in = in;
    temb_all.out = (int)total;
  }

  // ---------------------------------------------------------------------------------------------------------
  // ops
  // ---------------------------------------------------------------------------------------------------------
  bool lora_active(const Lin& L) const { return L.nsite > 0 && (dry || (lora_down && lora_up && mult != 0.f)); }

  Ten* linear(Ten* x, const Lin& L, Ten* res = nullptr) {
    Ten* y = new_ten(x->rows, L.out, x->n, x->H, x->W);
    const bool lon = lora_active(L);
    const int rtot = L.rank * L.nseg;
    float* xa = nullptr;
    const float lscale = mult * L.scale;
    if (lon && !L.dora) {  // xa[M, rows_pad] = x * down^T as one MFMA GEMM on the 16-bit shadow copy (fp32 result)
      xa = alloc_f32((size_t)MA(x) * L.rows_pad);
      GemmParams g;
      g.dtype = dtype;
      g.A = PA(x);
      g.lda = x->cols;
      g.W = shadow_ptr(L.sh_down);
      g.C = xa;
      g.ldc = L.rows_pad;
      g.out_f32 = 1;
      g.M = (int)MA(x);
      g.N = L.rows_pad;
      g.K = L.in;
      if (dry || lora_skinny_supported(g.A, g.lda, g.W, xa, g.ldc, g.M, g.N, g.K))
        RUNP(SMI_PROF_LORA, 2.0 * g.M * rtot * g.K, 0.0,
             launch_lora_skinny(dtype, g.A, g.lda, g.W, xa, g.ldc, g.M, g.N, g.K, stream, samp_ptr(false),
                                std::max(1, g.M / std::max(n_ad, 1))));
      else {
        RUNP(SMI_PROF_LORA, 2.0 * g.M * rtot * g.K, 0.0, (prof_on && prof_dump ? (next_tag = gemm_tag(g), 0) : 0, launch_gemm(g, stream)));
        if (samp_on)
          RUNP(SMI_PROF_LORA, 0.0, 0.0, launch_row_scale_f32(xa, g.ldc, g.M, g.N, samp_ptr(false),
                                                             std::max(1, g.M / std::max(n_ad, 1)), stream));
      }
    }
    GemmParams p;
    p.dtype = dtype;
    p.A = x->p;
    p.lda = x->cols;
    p.W = L.W;
    p.C = y->p;
    p.ldc = L.out;
    p.M = (int)x->rows;
    p.N = L.out;
    p.K = L.in;
    p.bias = L.b;
    if (res) {
      p.res = res->p;
      p.ldr = res->cols;
    }
    if (lon && !L.dora) {
      p.lora_xa = xa;
      p.ld_xa = L.rows_pad;
      p.lora_up = lora_up + L.off_up;
      p.up_sn = L.rank;
      p.up_sq = 1;
      p.lora_r = L.rank;
      p.lora_seg = L.nseg > 1 ? L.out / L.nseg : 0;
      p.lora_scale = lscale;
      p.lora_row0 = (int)x->arow0;
    }
    RUNP(SMI_PROF_GEMM, 2.0 * p.M * p.N * p.K, 2.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N),
         (prof_on && prof_dump ? (next_tag = gemm_tag(p), 0) : 0, launch_gemm(p, stream)));
    y->arow0 = x->arow0;
    if (lon && L.dora && MA(x) > 0) {  // adapted rows: y += x dW^T (dW holds lscale), accumulated onto the main result
      GemmParams d;
      d.dtype = dtype;
      d.A = PA(x);
      d.lda = x->cols;
      d.W = dora_sites[L.dora_idx].dW;
      d.C = PA(y);
      d.ldc = L.out;
      d.res = PA(y);
      d.ldr = L.out;
      d.M = (int)MA(x);
      d.N = L.out;
      d.K = L.in;
      RUNP(SMI_PROF_LORA, 2.0 * d.M * d.N * d.K, 0.0, launch_gemm(d, stream));
    }
    y->ng = lon || x->ng || (res && res->ng);
    if (saving && y->ng) {
      const Lin* Lp = &L;
      tape.push_back([=]() { linear_bwd(x, Lp, res, y, xa, lon, lscale); });
    }
    return y;
  }
  void linear_bwd(Ten* x, const Lin* L, Ten* res, Ten* y, float* xa, bool lon, float lscale) {
    void* dy = y->g;
    if (!dy) return;
    if (res && res->ng) accumulate(res, dy);
    const int M = (int)MA(x);
    const int rtot = L->rank * L->nseg;
    float* dxa = nullptr;
    if (lon && L->dora) {
      // G = dY^T X (dense fp32 [out, in]) from transposed 16-bit copies; rows of different samples carry different
      // power-of-two loss scales, so dY is brought to the smallest one while it is transposed (min / scale_j <= 1)
      const DoraSite& ds = dora_sites[L->dora_idx];
      const int Mp = (M + 63) / 64 * 64;
      const int rps = (int)(M / std::max(n_ad, 1));
      void* dyT = alloc_t(L->out, Mp);
      void* xT = alloc_t(L->in, Mp);
      float* G = alloc_f32((size_t)L->out * L->in);
      RUNP(SMI_PROF_LORA, 0.0, 0.0, launch_transpose_scaled(dtype, dy, L->out, dyT, M, L->out, Mp, gscale + 2 * MAXS + 2, rps, stream));
      RUNP(SMI_PROF_LORA, 0.0, 0.0, launch_transpose_scaled(dtype, PA(x), x->cols, xT, M, L->in, Mp, nullptr, rps, stream));
      GemmParams g;
      g.dtype = dtype;
      g.A = dyT;
      g.lda = Mp;
      g.W = xT;
      g.C = G;
      g.ldc = L->in;
      g.out_f32 = 1;
      g.M = L->out;
      g.N = L->in;
      g.K = Mp;
      RUNP(SMI_PROF_LORA, 2.0 * g.M * g.N * M, 0.0, launch_gemm(g, stream));
      // dW was built with lscale folded in: d(dW_unscaled) = lscale * G; the common loss scale (the minimum) divides out
      float* dscr = alloc_f32(dora_grad_scratch_floats(ds));
      RUNP(SMI_PROF_LORA, 0.0, 0.0, launch_dora_grads(dtype, ds, G, bw_down, bw_up, d_down, d_up, lscale / y->gmul, gscale + 2 * MAXS + 1, dscr, stream));
    }
    if (lon && !L->dora) {
      const int cs = L->out / L->nseg;
      const int r = L->rank;
      const int rp = L->rows_pad;
      dxa = alloc_f32((size_t)M * rp);
      {  // dxa[M, rows_pad] = dy * upT^T : one MFMA GEMM against the block-diagonal 16-bit shadow of lora_up
        GemmParams g;
        g.dtype = dtype;
        g.A = dy;
        g.lda = L->out;
        g.W = shadow_ptr(L->sh_up);
        g.C = dxa;
        g.ldc = rp;
        g.out_f32 = 1;
        g.M = M;
        g.N = rp;
        g.K = L->out;
        if (dry || lora_skinny_supported(g.A, g.lda, g.W, dxa, g.ldc, g.M, g.N, g.K))
          RUNP(SMI_PROF_LORA, 2.0 * M * rtot * cs, 0.0,
               launch_lora_skinny(dtype, g.A, g.lda, g.W, dxa, g.ldc, g.M, g.N, g.K, stream, samp_ptr(true),
                                  std::max(1, M / std::max(n_ad, 1))));
        else {
          RUNP(SMI_PROF_LORA, 2.0 * M * rtot * cs, 0.0, (prof_on && prof_dump ? (next_tag = gemm_tag(g), 0) : 0, launch_gemm(g, stream)));
          if (bw_samp_on)
            RUNP(SMI_PROF_LORA, 0.0, 0.0, launch_row_scale_f32(dxa, g.ldc, g.M, g.N, samp_ptr(true),
                                                               std::max(1, M / std::max(n_ad, 1)), stream));
        }
      }
      // deferred (see `wjobs`): d(up)[n][q] += lscale/S * sum_m dy[m][n] * xa[m][seg(n)*r + q]  for all segments,
      //                         d(down)[q'][k] += lscale/S * sum_m dxa[m][q'] * x[m][k]       (q' over the r_tot rows)
      const int rps = (int)(M / std::max(n_ad, 1));
      const float alpha = lscale / y->gmul;  // y->gmul: power-of-two factor the stored gradient carries (1 but for row vectors)
      auto push = [&](const void* X, int64_t ldx, const float* P, float* dW, int64_t so_r, int64_t so_k, int K, int rr,
                      int seg_cols) {
        WgradJob& j = push_wjob(X, ldx, P, rp, dW, so_r, so_k, M, K, rr, seg_cols, rps, alpha);
        (void)j;
      };
      push(dy, L->out, xa, d_up ? d_up + L->off_up : nullptr, 1, r, L->out, r, L->nseg > 1 ? cs : 0);
      // all segments of a fused projection in one job only while their rank rows fit the 8-accumulator class; beyond that
      // one job per segment: x is read once per segment, but the reduction kernel keeps R x 8 fp32 accumulators per thread
      // and its speed falls with R -- rank 8 on q|k|v (24 rows -> the 32-row class, 256 accumulators) took 7.5 ms per step
      // (lora class 12.7 -> 5.9 ms split), rank 4 (12 rows -> the 16-row class) 5.1 -> 4.7 ms.  SMI_WGRAD_SEG_MAX overrides.
      static const int seg_max = []() { const char* e = getenv("SMI_WGRAD_SEG_MAX"); return e ? atoi(e) : 8; }();
      if (rtot <= seg_max) {
        push(PA(x), x->cols, dxa, d_down ? d_down + L->off_down : nullptr, L->in, 1, L->in, rtot, 0);
      } else {
        for (int sg = 0; sg < L->nseg; ++sg) {
          push(PA(x), x->cols, dxa + sg * r, d_down ? d_down + L->off_down + (int64_t)sg * r * L->in : nullptr, L->in, 1,
               L->in, r, 0);
        }
      }
      pinned.push_back(dy);
    }
    if (x->ng) {
      if (!L->Wt && !dry) {
        set_error("internal: no transposed weight for '%s'", L->name.c_str());
        err = true;
        return;
      }
      const Slot gs = grad_slot(x);
      GemmParams p;
      p.dtype = dtype;
      p.A = dy;
      p.lda = L->out;
      p.W = L->Wt;
      p.C = gs.out;
      p.ldc = L->in;
      p.M = M;
      p.N = L->in;
      p.K = L->out;
      if (gs.add) {
        p.res = gs.add;
        p.ldr = L->in;
      }
      if (lon && !L->dora) {
        p.lora_xa = dxa;
        p.ld_xa = L->rows_pad;
        p.lora_up = bw_down + L->off_down;  // A_cat [rtot, in] read as [in][rtot]
        p.up_sn = 1;
        p.up_sq = L->in;
        p.lora_r = rtot;
        p.lora_seg = 0;
        p.lora_scale = lscale;
      }
      RUNP(SMI_PROF_GEMM, 2.0 * p.M * p.N * p.K, 2.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N),
           (prof_on && prof_dump ? (next_tag = gemm_tag(p), 0) : 0, launch_gemm(p, stream)));
      if (lon && L->dora) {  // dX += dY dW (dW^T [in, out] is the K-contiguous operand), accumulated in place
        GemmParams d;
        d.dtype = dtype;
        d.A = dy;
        d.lda = L->out;
        d.W = dora_dWt[L->dora_idx];
        d.C = gs.out;
        d.ldc = L->in;
        d.res = gs.out;
        d.ldr = L->in;
        d.M = M;
        d.N = L->in;
        d.K = L->out;
        RUNP(SMI_PROF_LORA, 2.0 * d.M * d.N * d.K, 0.0, launch_gemm(d, stream));
      }
    }
  }

  Ten* layernorm(Ten* x, const Norm& nm) {
    Ten* y = new_ten(x->rows, x->cols, x->n, x->H, x->W);
    float* st = alloc_f32((size_t)x->rows * 2);
    RUNP(SMI_PROF_NORM, 0.0, 4.0 * x->rows * x->cols, launch_layernorm_fwd(dtype, x->p, nm.gamma, nm.beta, y->p, st, (int)x->rows, x->cols, nm.eps, stream));
    y->ng = x->ng;
    if (saving && y->ng) {
      const Norm* np = &nm;
      tape.push_back([=]() {
        if (!y->g) return;
        const Slot gs = grad_slot(x);
        RUNP(SMI_PROF_NORM, 0.0, 6.0 * MA(x) * x->cols, launch_layernorm_bwd(dtype, PA(x), y->g, np->gamma, st + x->arow0 * 2, gs.add, gs.out, (int)MA(x), x->cols,
                                 stream));
      });
    }
    return y;
  }

  Ten* groupnorm(Ten* x, const Norm& nm, bool silu) {
    const int HW = x->H * x->W, G = cfg.norm_num_groups, C = x->cols;
    Ten* y = new_ten(x->rows, C, x->n, x->H, x->W);
    float* ab = alloc_f32((size_t)2 * x->n * C);
    float* mr = alloc_f32((size_t)x->n * G * 2);
    const size_t npart = gn_partial_floats(x->n, HW, G);
    float* part = alloc_f32(npart);
    RUNP(SMI_PROF_NORM, 0.0, 4.0 * x->rows * x->cols, launch_groupnorm_fwd(dtype, x->p, nm.gamma, nm.beta, y->p, ab, mr, part, x->n, HW, C, G, nm.eps, silu ? 1 : 0,
                             stream));
    y->ng = x->ng;
    if (saving && y->ng) {
      const Norm* np = &nm;
      tape.push_back([=]() {
        if (!y->g) return;
        const Slot gs = grad_slot(x);
        float* scr = alloc_f32(npart);
        const int n0 = (int)(x->arow0 / HW);  // first adapted sample
        RUNP(SMI_PROF_NORM, 0.0, 6.0 * MA(x) * x->cols, launch_groupnorm_bwd(dtype, PA(x), y->g, np->gamma, np->beta, ab + (size_t)n0 * C, ab + (size_t)(x->n + n0) * C, mr + (size_t)n0 * G * 2, gs.add, gs.out, scr, x->n - n0, HW,
                                 C, G, silu ? 1 : 0, stream));
      });
    }
    return y;
  }

  // ff.net.0 (GEGLU): proj = x W^T + b ; out = proj[:, :4C] * gelu(proj[:, 4C:]) as ONE GEMM whose epilogue applies the
  // gate; the projection is written only for the adapted rows (the backward's geglu_bwd needs it).
  Ten* linear_geglu(Ten* x, const Lin& L) {
    GemmParams p;
    p.dtype = dtype;
    p.A = x->p;
    p.lda = x->cols;
    p.W = L.W;
    p.ldc = L.out;
    p.M = (int)x->rows;
    p.N = L.out;
    p.K = L.in;
    p.bias = L.b;
    const bool need_proj = saving && x->ng;
    p.geglu_out = reinterpret_cast<void*>(16);  // placeholder for the capability query
    p.C = reinterpret_cast<void*>(16);
    if (lora_active(L) || !(dry || gemm_geglu_supported(p))) return geglu(linear(x, L));  // generic path
    Ten* out = new_ten(x->rows, L.out / 2, x->n, x->H, x->W);
    p.geglu_out = out->p;
    // the projection buffer holds the adapted rows only; virtual base so that row m lands at (m - arow0)
    tens->emplace_back();
    Ten* pj = &tens->back();
    *pj = *x;
    pj->cols = L.out;
    pj->g = nullptr;
    char* pbuf = (char*)arena_alloc((size_t)(need_proj ? MA(x) : 1) * L.out * esz());
    pj->p = pbuf - (size_t)x->arow0 * L.out * esz();
    p.C = pj->p;
    p.geglu_row0 = need_proj ? (int)x->arow0 : p.M;
    RUNP(SMI_PROF_GEMM, 2.0 * p.M * p.N * p.K, 2.0 * ((double)p.M * p.K + (double)p.N * p.K + 0.5 * (double)p.M * p.N),
         (prof_on && prof_dump ? (next_tag = gemm_tag(p), 0) : 0, launch_gemm(p, stream)));
    out->ng = x->ng;
    pj->ng = x->ng;
    if (saving && out->ng) {
      const Lin* Lp = &L;
      const int C4 = L.out / 2;
      tape.push_back([=]() {  // linear backward (runs after the GEGLU backward below has produced pj->g)
        linear_bwd(x, Lp, nullptr, pj, nullptr, false, 0.f);
      });
      tape.push_back([=]() {
        if (!out->g) return;
        void* dp = grad_slot(pj).out;  // the projection has a single consumer: nothing accumulated yet
        RUNP(SMI_PROF_ELEM, 0.0, 10.0 * MA(pj) * C4, launch_geglu_bwd(dtype, PA(pj), out->g, dp, (int)MA(pj), C4, stream));
      });
    }
    return out;
  }

  Ten* geglu(Ten* pj) {
    const int C4 = pj->cols / 2;
    Ten* y = new_ten(pj->rows, C4, pj->n, pj->H, pj->W);
    RUNP(SMI_PROF_ELEM, 0.0, 6.0 * pj->rows * C4, launch_geglu_fwd(dtype, pj->p, y->p, (int)pj->rows, C4, stream));
    y->ng = pj->ng;
    if (saving && y->ng) {
      tape.push_back([=]() {
        if (!y->g) return;
        const Slot gs = grad_slot(pj);
        if (gs.add) {  // never happens in this graph (proj has a single consumer); kept for safety
          void* tmp = alloc_t(MA(pj), pj->cols);
          RUNP(SMI_PROF_ELEM, 0.0, 10.0 * MA(pj) * C4, launch_geglu_bwd(dtype, PA(pj), y->g, tmp, (int)MA(pj), C4, stream));
          RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_add(dtype, gs.add, tmp, gs.out, MA(pj) * pj->cols, stream));
        } else {
          RUNP(SMI_PROF_ELEM, 0.0, 10.0 * MA(pj) * C4, launch_geglu_bwd(dtype, PA(pj), y->g, gs.out, (int)MA(pj), C4, stream));
        }
      });
    }
    return y;
  }

  // self-attention on a fused [M, 3C] q|k|v tensor, or cross-attention on q [M, C] and kv [n*L, 2C]
  Ten* attention(Ten* qkv, Ten* q, Ten* kv, int heads, int C, int nbatch, int Nq, int Nk,
                 const void* kv_ext = nullptr, int64_t ld_ext = 0) {
    Ten* src_q = qkv ? qkv : q;
    Ten* o = new_ten(src_q->rows, C, src_q->n, src_q->H, src_q->W);
    float* lse = alloc_f32((size_t)nbatch * heads * Nq);
    AttnParams p;
    p.dtype = dtype;
    p.B = nbatch;
    p.H = heads;
    p.Nq = Nq;
    p.Nk = Nk;
    p.D = C / heads;
    p.scale = 1.f / sqrtf((float)p.D);
    p.causal = attn_causal ? 1 : 0;
    if (qkv) {
      p.Q = qkv->p;
      p.K = (char*)qkv->p + (size_t)C * esz();
      p.V = (char*)qkv->p + (size_t)2 * C * esz();
      p.ldq = p.ldk = p.ldv = 3 * C;
    } else {
      p.Q = q->p;
      p.ldq = C;
      p.K = kv_ext ? kv_ext : kv->p;
      p.V = (const char*)p.K + (size_t)C * esz();
      p.ldk = p.ldv = kv_ext ? ld_ext : 2 * C;
    }
    p.O = o->p;
    p.ldo = C;
    p.lse = lse;
    RUNP(SMI_PROF_ATTN, 4.0 * nbatch * heads * (double)Nq * Nk * p.D, 0.0, launch_attn_fwd(p, stream));
    o->ng = qkv ? qkv->ng : (q->ng || (kv && kv->ng));
    if (saving && o->ng) {
      tape.push_back([=]() {
        if (!o->g) return;
        AttnParams b = p;
        const int b0 = (int)(o->arow0 / Nq);  // first adapted sample
        b.B = nbatch - b0;
        b.Q = (const char*)p.Q + (size_t)b0 * Nq * p.ldq * esz();
        b.K = (const char*)p.K + (size_t)b0 * Nk * p.ldk * esz();
        b.V = (const char*)p.V + (size_t)b0 * Nk * p.ldv * esz();
        b.O = PA(o);
        b.lse = p.lse + (size_t)b0 * heads * Nq;
        b.dO = o->g;
        b.lddo = C;
        b.delta = alloc_f32((size_t)b.B * heads * Nq);
        // q|k|v, q and kv tensors have a single consumer (this attention): their slots are fresh
        if (qkv) {
          char* g = (char*)grad_slot(qkv).out;
          b.dQ = g;
          b.dK = g + (size_t)C * esz();
          b.dV = g + (size_t)2 * C * esz();
          b.lddq = b.lddk = b.lddv = 3 * C;
        } else {
          if (q->ng) {
            b.dQ = grad_slot(q).out;
            b.lddq = C;
          }
          if (kv && kv->ng) {
            char* g = (char*)grad_slot(kv).out;
            b.dK = g;
            b.dV = g + (size_t)C * esz();
            b.lddk = b.lddv = 2 * C;
          }
        }
        RUNP(SMI_PROF_ATTN, 10.0 * b.B * heads * (double)Nq * Nk * b.D, 0.0, launch_attn_bwd(b, stream));
      });
    }
    return o;
  }

  // 3x3 conv (pad 1): mode 0 stride 1, 1 stride 2, 2 nearest-2x upsample then stride 1
  Ten* conv3x3(Ten* x, const Conv& c, Ten* rowvec, Ten* res, int64_t ld_rowvec = 0) {
    const int Hin = x->H, Win = x->W;
    const int Hout = c.mode == 1 ? (Hin + 1) / 2 : (c.mode == 2 ? Hin * 2 : Hin);
    const int Wout = c.mode == 1 ? (Win + 1) / 2 : (c.mode == 2 ? Win * 2 : Win);
    Ten* y = new_ten((int64_t)x->n * Hout * Wout, c.Cout, x->n, Hout, Wout);
    GemmParams p;
    p.dtype = dtype;
    p.conv = 1;
    p.A = x->p;
    p.W = c.Wp;
    p.C = y->p;
    p.ldc = c.Cout;
    p.M = (int)y->rows;
    p.N = c.Cout;
    p.K = 9 * c.Cin;
    p.bias = c.b;
    p.Nb = x->n;
    p.Hin = Hin;
    p.Win = Win;
    p.Cin = c.Cin;
    p.Hout = Hout;
    p.Wout = Wout;
    p.stride = c.mode == 1 ? 2 : 1;
    p.pad = c.pad;
    p.upsample = c.mode == 2 ? 1 : 0;
    if (rowvec) {
      p.rowvec = rowvec->p;
      p.rows_per_vec = Hout * Wout;
      p.ld_rowvec = ld_rowvec;
    }
    if (res) {
      p.res = res->p;
      p.ldr = res->cols;
    }
    // c3lier adaptor: y += lscale * up(down_conv(x)) on the adapted samples.  xa = down_conv(x) is one more implicit-GEMM
    // conv with the (16-row padded) shadow filter and fp32 output; the 1x1 up-projection rides in the main epilogue.
    const bool lon = c.nsite > 0 && (dry || (lora_down && lora_up && mult != 0.f));
    const float lscale = mult * c.scale;
    float* xa = nullptr;
    if (lon) {
      xa = alloc_f32((size_t)MA(y) * c.rows_pad);
      GemmParams g = p;
      g.A = PA(x);
      g.W = shadow_ptr(c.sh_down);
      g.C = xa;
      g.ldc = c.rows_pad;
      g.out_f32 = 1;
      g.M = (int)MA(y);
      g.N = c.rows_pad;
      g.bias = nullptr;
      g.rowvec = nullptr;
      g.res = nullptr;
      g.Nb = x->n - (int)(x->arow0 / (Hin * Win));
      RUNP(SMI_PROF_LORA, 2.0 * g.M * c.rank * g.K, 0.0, launch_gemm(g, stream));
      p.lora_xa = xa;
      p.ld_xa = c.rows_pad;
      p.lora_up = lora_up + c.off_up;
      p.up_sn = c.rank;
      p.up_sq = 1;
      p.lora_r = c.rank;
      p.lora_seg = 0;
      p.lora_scale = lscale;
      p.lora_row0 = (int)y->arow0;
    }
    RUNP(SMI_PROF_CONV, 2.0 * p.M * p.N * p.K, 2.0 * ((double)x->rows * c.Cin + (double)p.N * p.K + (double)p.M * p.N),
         (prof_on && prof_dump ? (next_tag = gemm_tag(p), 0) : 0, launch_gemm(p, stream)));
    y->ng = lon || x->ng || (res && res->ng) || (rowvec && rowvec->ng);
    if (saving && y->ng) {
      const Conv* cp = &c;
      tape.push_back([=]() {
        void* dy = y->g;
        if (!dy) return;
        if (res && res->ng) accumulate(res, dy);
        const int nb_ad = x->n - (int)(x->arow0 / (Hin * Win));  // adapted samples
        if (rowvec && rowvec->ng) {
          // d(rowvec)[n][c] = sum over the pixels of sample n of dy; stored x 2^-k (k ~ log2(HW) / 2) so that a coherent sum
          // cannot leave the fp16 range -- linear_bwd divides it out again (Ten::gmul)
          int k = 0;
          while ((1 << (2 * (k + 1))) <= Hout * Wout) ++k;
          rowvec->gmul = exp2f((float)-k);
          rowvec->g = alloc_t(MA(rowvec), rowvec->cols);
          float* cs_scratch = alloc_f32(colsum_scratch_floats(nb_ad, cp->Cout));
          RUNP(SMI_PROF_ELEM, 0.0, 0.0,
               launch_colsum(dtype, dy, rowvec->g, cs_scratch, nb_ad, Hout * Wout, cp->Cout, rowvec->gmul, stream));
        }
        float* dxa = nullptr;
        if (lon) {
          const int M = (int)MA(y), rp = cp->rows_pad, r = cp->rank;
          dxa = alloc_f32((size_t)M * rp);
          {  // dxa[M, rows_pad] = dy * up
            GemmParams g;
            g.dtype = dtype;
            g.A = dy;
            g.lda = cp->Cout;
            g.W = shadow_ptr(cp->sh_up);
            g.C = dxa;
            g.ldc = rp;
            g.out_f32 = 1;
            g.M = M;
            g.N = rp;
            g.K = cp->Cout;
            if (dry || lora_skinny_supported(g.A, g.lda, g.W, dxa, g.ldc, g.M, g.N, g.K))
              RUNP(SMI_PROF_LORA, 2.0 * M * r * cp->Cout, 0.0,
                   launch_lora_skinny(dtype, g.A, g.lda, g.W, dxa, g.ldc, g.M, g.N, g.K, stream));
            else
              RUNP(SMI_PROF_LORA, 2.0 * M * r * cp->Cout, 0.0, launch_gemm(g, stream));
          }
          const int rps = Hout * Wout;
          // d(up)[n][q] += lscale/S * sum_m dy[m][n] xa[m][q];  d(down)[q][ci][tap] += lscale/S * sum_m dxa[m][q] x[pixel(m, tap)][ci]
          push_wjob(dy, cp->Cout, xa, rp, d_up ? d_up + cp->off_up : nullptr, 1, r, M, cp->Cout, r, 0, rps, lscale);
          for (int t = 0; t < 9; ++t) {
            WgradJob& j = push_wjob(PA(x), cp->Cin, dxa, rp, d_down ? d_down + cp->off_down + t : nullptr,
                                    (int64_t)9 * cp->Cin, 9, M, cp->Cin, r, 0, rps, lscale);
            j.conv_tap = t;
            j.Hin = Hin;
            j.Win = Win;
            j.Hout = Hout;
            j.Wout = Wout;
            j.conv_stride = cp->mode == 1 ? 2 : 1;
            j.conv_ups = cp->mode == 2 ? 1 : 0;
          }
          pinned.push_back(dy);
        }
        if (!x->ng) return;
        const Slot gs = grad_slot(x);
        // the adaptor's share of dX: the same gradient conv once more on dxa (fp32 -> 64-channel 16-bit image, x lscale)
        // with the down filter's gradient pack, accumulated onto the main result
        void* dxa16 = nullptr;
        if (lon) {
          dxa16 = alloc_t(MA(y), 64);
          RUNP(SMI_PROF_LORA, 0.0, 0.0, launch_f32_to_padded(dtype, dxa, cp->rows_pad, cp->rank, dxa16, 64, MA(y), lscale, stream));
        }
        auto lora_dx = [&](GemmParams b, void* acc) {  // b: the main gradient conv's parameters
          if (!lon) return;
          b.A = dxa16;
          b.W = shadow_ptr(cp->sh_gw);
          b.K = 9 * 64;
          b.Cin = 64;
          b.C = acc;
          b.res = acc;
          b.ldr = cp->Cin;
          RUNP(SMI_PROF_LORA, 2.0 * b.M * b.N * 9 * cp->rank, 0.0, launch_gemm(b, stream));
        };
        GemmParams b;
        b.dtype = dtype;
        b.conv = 1;
        b.A = dy;
        b.W = cp->Wg;
        b.N = cp->Cin;
        b.K = 9 * cp->Cout;
        b.Cin = cp->Cout;
        b.Nb = x->n - (int)(x->arow0 / (Hin * Win));
        b.ldc = cp->Cin;
        if (cp->mode == 0) {
          b.Hin = b.Hout = Hin;
          b.Win = b.Wout = Win;
          b.M = (int)MA(x);
          b.C = gs.out;
          if (gs.add) {
            b.res = gs.add;
            b.ldr = cp->Cin;
          }
          RUNP(SMI_PROF_CONV, 2.0 * b.M * b.N * b.K, 0.0, (prof_on && prof_dump ? (next_tag = gemm_tag(b), 0) : 0, launch_gemm(b, stream)));
          lora_dx(b, gs.out);
        } else if (cp->mode == 1) {  // gradient of the stride-2 conv: gather dY at (i + 1 - k) / 2
          b.Hin = Hout;
          b.Win = Wout;
          b.Hout = Hin;
          b.Wout = Win;
          b.stride = 2;
          b.transposed = 1;
          b.M = (int)MA(x);
          b.C = gs.out;
          if (gs.add) {
            b.res = gs.add;
            b.ldr = cp->Cin;
          }
          RUNP(SMI_PROF_CONV, 2.0 * b.M * b.N * b.K, 0.0, (prof_on && prof_dump ? (next_tag = gemm_tag(b), 0) : 0, launch_gemm(b, stream)));
          lora_dx(b, gs.out);
        } else {  // upsample + conv: gradient on the 2x grid, then 2x2 sum-pool
          void* du = alloc_t(MA(y), cp->Cin);
          b.Hin = b.Hout = Hout;
          b.Win = b.Wout = Wout;
          b.M = (int)MA(y);
          b.C = du;
          RUNP(SMI_PROF_CONV, 2.0 * b.M * b.N * b.K, 0.0, (prof_on && prof_dump ? (next_tag = gemm_tag(b), 0) : 0, launch_gemm(b, stream)));
          lora_dx(b, du);
          if (gs.add) {
            void* tmp = alloc_t(MA(x), cp->Cin);
            RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_pool2x2_sum(dtype, du, tmp, b.Nb, Hin, Win, cp->Cin, stream));
            RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_add(dtype, gs.add, tmp, gs.out, MA(x) * x->cols, stream));
          } else {
            RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_pool2x2_sum(dtype, du, gs.out, b.Nb, Hin, Win, cp->Cin, stream));
          }
        }
      });
    }
    return y;
  }

  Ten* concat(Ten* a, Ten* b) {
    Ten* y = new_ten(a->rows, a->cols + b->cols, a->n, a->H, a->W);
    RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_copy_cols(dtype, a->p, a->cols, y->p, y->cols, 0, (int)a->rows, a->cols, stream));
    RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_copy_cols(dtype, b->p, b->cols, y->p, y->cols, a->cols, (int)a->rows, b->cols, stream));
    y->ng = a->ng || b->ng;
    if (saving && y->ng) {
      tape.push_back([=]() {
        if (!y->g) return;
        Ten* parts[2] = {a, b};
        int col0 = 0;
        for (int i = 0; i < 2; ++i) {
          Ten* t = parts[i];
          if (t->ng) {
            const Slot gs = grad_slot(t);
            if (gs.add) {
              void* tmp = alloc_t(MA(t), t->cols);
              RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_copy_cols(dtype, (char*)y->g + (size_t)col0 * esz(), y->cols, tmp, t->cols, 0, (int)MA(t),
                                   t->cols, stream));
              RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_add(dtype, gs.add, tmp, gs.out, MA(t) * t->cols, stream));
            } else {
              RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_copy_cols(dtype, (char*)y->g + (size_t)col0 * esz(), y->cols, gs.out, t->cols, 0, (int)MA(t),
                                   t->cols, stream));
            }
          }
          col0 += t->cols;
        }
      });
    }
    return y;
  }

  Ten* silu(Ten* x) {  // only used on the (gradient-free) time embedding path
    Ten* y = new_ten(x->rows, x->cols, x->n, x->H, x->W);
    RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_silu(dtype, x->p, y->p, x->rows * x->cols, stream));
    return y;
  }

  Ten* resnet(Ten* x, const Resnet& r, Ten* temb_act, bool keep_out = false) {
    Ten* h = groupnorm(x, r.n1, true);
    Ten* t = nullptr;
    int64_t ldt = 0;
    if (r.has_temb && temb_all_out) {  // a column block of the grouped projection (no gradient flows into it)
      tens->emplace_back();
      t = &tens->back();
      t->p = (char*)temb_all_out->p + (size_t)r.temb_col * esz();
      t->rows = temb_all_out->rows;
      t->cols = r.temb.out;
      t->n = temb_all_out->n;
      ldt = temb_all.out;
    } else if (r.has_temb) {
      t = linear(temb_act, r.temb);
    }
    h = conv3x3(h, r.c1, t, nullptr, ldt);
    h = groupnorm(h, r.n2, true);
    Ten* sc = r.has_sc ? linear(x, r.sc) : x;
    if (keep_out) keep_next_output();  // (conv3x3 creates its output tensor first)
    return conv3x3(h, r.c2, nullptr, sc);
  }

  Ten* transformer(Ten* x, const Transformer& t, Ten* ctx, bool keep_out = false) {
    const int C = t.C, nb = x->n, Nq = x->H * x->W;
    Ten* h = groupnorm(x, t.norm, false);
    h = linear(h, t.proj_in);
    for (const TBlock& tb : t.blocks) {
      Ten* nrm = layernorm(h, tb.n1);
      Ten* qkv = linear(nrm, tb.qkv1);
      Ten* o = attention(qkv, nullptr, nullptr, t.heads, C, nb, Nq, Nq);
      h = linear(o, tb.out1, h);
      nrm = layernorm(h, tb.n2);
      Ten* q = linear(nrm, tb.q2);
      if (kv_grouped) {
        o = attention(nullptr, q, nullptr, t.heads, C, nb, Nq, ctx_len,
                      (const char*)kv_all_out->p + (size_t)tb.kv_col * esz(), kv_all.out);
      } else {
        Ten* kv = linear(ctx, tb.kv2);
        o = attention(nullptr, q, kv, t.heads, C, nb, Nq, ctx_len);
      }
      h = linear(o, tb.out2, h);
      nrm = layernorm(h, tb.n3);
      Ten* gg = linear_geglu(nrm, tb.ff1);
      h = linear(gg, tb.ff2, h);
    }
    if (keep_out) keep_next_output();  // (linear creates its output tensor first)
    return linear(h, t.proj_out, x);
  }

  // ---------------------------------------------------------------------------------------------------------
  // AutoencoderKL encoder (image sliders: I/train_util.py:213-222 `vae.encode(image).latent_dist`), forward only.
  // Same kernels as the UNet: GroupNorm(+SiLU), implicit-GEMM 3x3 convs (the downsampler with its (0,1,0,1) padding as
  // `pad = 0`), 1x1 shortcuts as GEMMs.  The mid block's attention is ONE head of width 512 over all pixels: run as
  // materialised GEMMs (scores in fp32, row softmax, P V), a few GFLOP per image.
  // ---------------------------------------------------------------------------------------------------------
  bool is_vae = false;
  smi_vae_config vcfg{};
  struct VLevel {
    std::vector<Resnet> res;
    bool has_samp = false;
    Conv samp;
  };
  std::vector<VLevel> v_down;
  Resnet v_mid0, v_mid1;
  Norm v_attn_norm, v_norm_out;
  Lin v_qkv, v_o;
  Conv v_conv_in, v_conv_out;
  const void* v_quant_w = nullptr;
  const void* v_quant_b = nullptr;

  void build_vae() {
    const int L = vcfg.n_levels;
    const int* boc = vcfg.block_out_channels;
    {
      v_conv_in.Cin = 64;
      v_conv_in.Cout = boc[0];
      v_conv_in.b = Wd("encoder.conv_in.bias");
      check_shape("encoder.conv_in.weight", {boc[0], vcfg.in_channels, 3, 3});
      void* wp = pack_alloc((size_t)boc[0] * 9 * 64 * esz());
      pack_conv_into(Wd("encoder.conv_in.weight"), wp, boc[0], vcfg.in_channels, 0, 64);
      v_conv_in.Wp = wp;
    }
    v_down.resize(L);
    int ch = boc[0];
    for (int i = 0; i < L; ++i) {
      const std::string b = "encoder.down_blocks." + std::to_string(i);
      for (int j = 0; j < vcfg.layers_per_block; ++j)
        v_down[i].res.push_back(make_resnet(b + ".resnets." + std::to_string(j), j == 0 ? ch : boc[i], boc[i], false, 1e-6f, false));
      ch = boc[i];
      v_down[i].has_samp = i != L - 1;
      if (v_down[i].has_samp) {
        v_down[i].samp = make_conv(b + ".downsamplers.0.conv", ch, ch, 1, false);
        v_down[i].samp.pad = 0;
      }
    }
    v_mid0 = make_resnet("encoder.mid_block.resnets.0", ch, ch, false, 1e-6f, false);
    v_attn_norm = make_norm("encoder.mid_block.attentions.0.group_norm", ch, 1e-6f);
    v_qkv = make_fused("encoder.mid_block.attentions.0", {"to_q", "to_k", "to_v"}, ch, ch, false);
    {  // fused q|k|v bias
      char* bq = (char*)pack_alloc((size_t)3 * ch * esz());
      const char* parts[3] = {"to_q", "to_k", "to_v"};
      for (int i = 0; i < 3; ++i) {
        const void* src = Wd(std::string("encoder.mid_block.attentions.0.") + parts[i] + ".bias");
        if (!dry && !err && src)
          (void)hipMemcpyAsync(bq + (size_t)i * ch * esz(), src, (size_t)ch * esz(), hipMemcpyDeviceToDevice, stream);
      }
      v_qkv.b = bq;
    }
    v_o = make_lin("encoder.mid_block.attentions.0.to_out.0", ch, ch, true, false);
    v_mid1 = make_resnet("encoder.mid_block.resnets.1", ch, ch, false, 1e-6f, false);
    v_norm_out = make_norm("encoder.conv_norm_out", ch, 1e-6f);
    v_conv_out = make_conv("encoder.conv_out", ch, 2 * vcfg.latent_channels, 0, false);
    v_quant_w = Wd("quant_conv.weight");
    v_quant_b = Wd("quant_conv.bias");
    check_shape("quant_conv.weight", {2 * vcfg.latent_channels, 2 * vcfg.latent_channels, 1, 1});
    gscale = (float*)pack_alloc(256 * sizeof(float));
    finish_lora();
  }

  // image f32 [n, 3, h, w] in [-1, 1]  ->  moments f32 [n, 2 * latent, h/8, w/8] (mean | logvar, before the clamp)
  int forward_vae(int n, const float* image, float* moments_out) {
    n_ad = 0;
    cur = &arena[0];
    cur->reset();
    tens = &tens_[0];
    tens->clear();
    saving = false;
    lora_down = lora_up = nullptr;
    mult = 0.f;
    const int H = lat_h, Wd_ = lat_w;  // image size for this engine kind
    Ten* x0 = new_ten((int64_t)n * H * Wd_, 64, n, H, Wd_);
    RUN(launch_nchw_to_nhwc(dtype, image, 1, x0->p, n, vcfg.in_channels, H * Wd_, 64, 1.f, stream));
    Ten* h = conv3x3(x0, v_conv_in, nullptr, nullptr);
    for (auto& lv : v_down) {
      for (auto& r : lv.res) h = resnet(h, r, nullptr);
      if (lv.has_samp) h = conv3x3(h, lv.samp, nullptr, nullptr);
    }
    h = resnet(h, v_mid0, nullptr);
    {  // single-head attention over the pixels, with its own GroupNorm and residual
      const int C = h->cols, N = h->H * h->W;
      Ten* nrm = groupnorm(h, v_attn_norm, false);
      Ten* qkv = linear(nrm, v_qkv);
      Ten* o = new_ten(h->rows, C, h->n, h->H, h->W);
      const float sc = 1.f / sqrtf((float)C);
      for (int i = 0; i < n; ++i) {
        const char* base = (const char*)qkv->p + (size_t)i * N * 3 * C * esz();
        float* S = alloc_f32((size_t)N * N);
        void* P = alloc_t(N, N);
        void* Vt = alloc_t(C, N);
        GemmParams g;
        g.dtype = dtype;
        g.A = base;
        g.lda = 3 * C;
        g.W = base + (size_t)C * esz();  // K rows, row stride 3C: not a dense [N, C] operand -> copy below
        // the GEMM's W operand is dense [N_out, K]: K (and V) are column blocks of the fused tensor, so stage them
        void* Kd = alloc_t(N, C);
        RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_copy_cols(dtype, base + (size_t)C * esz(), 3 * C, Kd, C, 0, N, C, stream));
        g.W = Kd;
        g.C = S;
        g.ldc = N;
        g.out_f32 = 1;
        g.M = N;
        g.N = N;
        g.K = C;
        RUNP(SMI_PROF_ATTN, 2.0 * N * N * C, 0.0, launch_gemm(g, stream));
        RUNP(SMI_PROF_ATTN, 0.0, 0.0, launch_softmax_rows(dtype, S, P, N, N, sc, stream));
        void* Vd = alloc_t(N, C);
        RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_copy_cols(dtype, base + (size_t)2 * C * esz(), 3 * C, Vd, C, 0, N, C, stream));
        transpose_into(Vd, Vt, N, C, N, 0, false);
        GemmParams pv;
        pv.dtype = dtype;
        pv.A = P;
        pv.lda = N;
        pv.W = Vt;
        pv.C = (char*)o->p + (size_t)i * N * C * esz();
        pv.ldc = C;
        pv.M = N;
        pv.N = C;
        pv.K = N;
        RUNP(SMI_PROF_ATTN, 2.0 * N * N * C, 0.0, launch_gemm(pv, stream));
      }
      h = linear(o, v_o, h);
    }
    h = resnet(h, v_mid1, nullptr);
    Ten* hn = groupnorm(h, v_norm_out, true);
    const int C2 = 2 * vcfg.latent_channels;
    const int Ho = hn->H, Wo = hn->W;
    Ten* y = new_ten((int64_t)n * Ho * Wo, C2, n, Ho, Wo, sizeof(float));
    {
      GemmParams p;
      p.dtype = dtype;
      p.conv = 1;
      p.A = hn->p;
      p.W = v_conv_out.Wp;
      p.C = y->p;
      p.ldc = C2;
      p.out_f32 = 1;
      p.M = (int)y->rows;
      p.N = C2;
      p.K = 9 * v_conv_out.Cin;
      p.bias = v_conv_out.b;
      p.Nb = n;
      p.Hin = p.Hout = Ho;
      p.Win = p.Wout = Wo;
      p.Cin = v_conv_out.Cin;
      RUNP(SMI_PROF_CONV, 2.0 * p.M * p.N * p.K, 0.0, launch_gemm(p, stream));
    }
    float* q = alloc_f32((size_t)y->rows * C2);
    RUN(launch_chan_mix(dtype, (const float*)y->p, v_quant_w, v_quant_b, q, y->rows, C2, stream));
    RUN(launch_nhwc_to_nchw_f32(q, moments_out, n, C2, Ho * Wo, stream));
    if (cur->overflow && !dry) {
      set_error("workspace too small for this call (needs %zu bytes, has %zu)", cur->peak, cur->cap);
      return -3;
    }
    return err ? -1 : 0;
  }

  // ---------------------------------------------------------------------------------------------------------
  // CLIP text encoder (transformers CLIPTextModel[WithProjection]; T/train_util.py:108-155), forward only: token +
  // position embedding, pre-LayerNorm blocks with CAUSAL self-attention (the fused attention kernel with its causal
  // mask), quick_gelu / gelu MLP, final LayerNorm, pooled EOS row (x text_projection).
  // ---------------------------------------------------------------------------------------------------------
  bool attn_causal = false;
  bool is_clip = false;
  smi_clip_config ccfg{};
  struct CLayer {
    Norm n1, n2;
    Lin qkv, out, fc1, fc2;
  };
  std::vector<CLayer> c_layers;
  Norm c_final;
  Lin c_proj;
  const void* c_tok = nullptr;
  const void* c_pos = nullptr;

  void build_clip() {
    const int d = ccfg.hidden_size;
    c_tok = Wd("text_model.embeddings.token_embedding.weight");
    c_pos = Wd("text_model.embeddings.position_embedding.weight");
    check_shape("text_model.embeddings.token_embedding.weight", {ccfg.vocab_size, d});
    check_shape("text_model.embeddings.position_embedding.weight", {ccfg.max_positions, d});
    c_layers.resize(ccfg.num_layers);
    for (int i = 0; i < ccfg.num_layers; ++i) {
      const std::string b = "text_model.encoder.layers." + std::to_string(i);
      CLayer& L = c_layers[i];
      L.n1 = make_norm(b + ".layer_norm1", d, 1e-5f);
      L.qkv = make_fused(b + ".self_attn", {"q_proj", "k_proj", "v_proj"}, d, d, false);
      char* bq = (char*)pack_alloc((size_t)3 * d * esz());
      const char* parts[3] = {"q_proj", "k_proj", "v_proj"};
      for (int j = 0; j < 3; ++j) {
        const void* src = Wd(b + ".self_attn." + parts[j] + ".bias");
        if (!dry && !err && src)
          (void)hipMemcpyAsync(bq + (size_t)j * d * esz(), src, (size_t)d * esz(), hipMemcpyDeviceToDevice, stream);
      }
      L.qkv.b = bq;
      L.out = make_lin(b + ".self_attn.out_proj", d, d, true, false);
      L.n2 = make_norm(b + ".layer_norm2", d, 1e-5f);
      L.fc1 = make_lin(b + ".mlp.fc1", d, ccfg.intermediate_size, true, false);
      L.fc2 = make_lin(b + ".mlp.fc2", ccfg.intermediate_size, d, true, false);
    }
    c_final = make_norm("text_model.final_layer_norm", d, 1e-5f);
    if (ccfg.projection_dim > 0) c_proj = make_lin("text_projection", d, ccfg.projection_dim, false, false);
    gscale = (float*)pack_alloc(256 * sizeof(float));
    finish_lora();
  }

  int forward_clip(int n, const int* ids, const int* eos_pos, void* last_hidden, void* penultimate, void* pooled) {
    n_ad = 0;
    cur = &arena[0];
    cur->reset();
    tens = &tens_[0];
    tens->clear();
    saving = false;
    lora_down = lora_up = nullptr;
    mult = 0.f;
    attn_causal = true;
    const int L = ccfg.max_positions, d = ccfg.hidden_size;
    Ten* h = new_ten((int64_t)n * L, d, n, L, 1);
    RUN(launch_embed(dtype, ids, c_tok, c_pos, h->p, (int64_t)n * L, L, d, ccfg.vocab_size, stream));
    for (int i = 0; i < ccfg.num_layers; ++i) {
      const CLayer& ly = c_layers[i];
      if (i == ccfg.num_layers - 1 && penultimate)  // hidden_states[-2]: what enters the last layer
        RUN(launch_copy_cols(dtype, h->p, d, penultimate, d, 0, (int)h->rows, d, stream));
      Ten* qkv = linear(layernorm(h, ly.n1), ly.qkv);
      Ten* o = attention(qkv, nullptr, nullptr, ccfg.num_heads, d, n, L, L);
      h = linear(o, ly.out, h);
      Ten* f = linear(layernorm(h, ly.n2), ly.fc1);
      Ten* a = new_ten(f->rows, f->cols, n, L, 1);
      RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_act(dtype, f->p, a->p, f->rows * f->cols, ccfg.hidden_act, stream));
      h = linear(a, ly.fc2, h);
    }
    attn_causal = false;
    Ten* fin = layernorm(h, c_final);
    if (last_hidden) RUN(launch_copy_cols(dtype, fin->p, d, last_hidden, d, 0, (int)fin->rows, d, stream));
    if (pooled) {
      Ten* pl = new_ten(n, d, n, 1, 1);
      RUN(launch_gather_rows(dtype, fin->p, eos_pos, pl->p, n, L, d, stream));
      if (ccfg.projection_dim > 0) pl = linear(pl, c_proj);
      RUN(launch_copy_cols(dtype, pl->p, pl->cols, pooled, pl->cols, 0, n, pl->cols, stream));
    }
    if (cur->overflow && !dry) {
      set_error("workspace too small for this call (needs %zu bytes, has %zu)", cur->peak, cur->cap);
      return -3;
    }
    return err ? -1 : 0;
  }

  // ---------------------------------------------------------------------------------------------------------
  // whole passes
  // ---------------------------------------------------------------------------------------------------------
  int forward(int n, int n_adapted, const float* sample, float timestep, const void* ctxp, const void* text_embeds,
              const float* time_ids, bool save, float* eps_out) {
    n_ad = n_adapted;
    cur = &arena[save ? 1 : 0];
    cur->reset();
    in_block = persist_next = false;
    if (!save) {
      scr[0].reset();
      scr[1].reset();
      if (dry) scr[0].cap = scr[1].cap = (size_t)-1;
    }
    if (save) {
      tape.clear();
      tape_valid = false;
      ++tape_gen;
    }
    tens = &tens_[save ? 1 : 0];
    tens->clear();
    saving = save;
    const int H = lat_h, Wd_ = lat_w, HW = H * Wd_;
    const int C0 = cfg.block_out_channels[0];
    const int ted = C0 * 4;

    // ---- time / added-condition embedding (no gradient flows here: not adapted under lierla)
    float* tvals = alloc_f32(n);
    if (!dry && !err) hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, stream, tvals, timestep, n);
    Ten* te = new_ten(n, C0, n, 1, 1);  // [n, C] tensors carry n, H = W = 1: the adapted samples are the last rows
    RUN(launch_timestep_embed(dtype, tvals, te->p, n, C0, stream));
    Ten* emb = linear(silu(linear(te, time1)), time2);
    if (cfg.addition_embed) {
      const int P = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim;
      Ten* tid = new_ten(n, 6 * cfg.addition_time_embed_dim, n, 1, 1);
      RUN(launch_timestep_embed(dtype, time_ids, tid->p, n * 6, cfg.addition_time_embed_dim, stream));
      Ten* cat = new_ten(n, cfg.projection_class_embeddings_input_dim, n, 1, 1);
      Ten pooled;
      pooled.p = const_cast<void*>(text_embeds);
      RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_copy_cols(dtype, pooled.p, P, cat->p, cat->cols, 0, n, P, stream));
      RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_copy_cols(dtype, tid->p, tid->cols, cat->p, cat->cols, P, n, tid->cols, stream));
      Ten* aug = linear(silu(linear(cat, add1)), add2);
      Ten* sum = new_ten(n, ted, n, 1, 1);
      RUNP(SMI_PROF_ELEM, 0.0, 0.0, launch_add(dtype, emb->p, aug->p, sum->p, (int64_t)n * ted, stream));
      emb = sum;
    }
    Ten* temb_act = silu(emb);
    if (!prep_sites.empty() && (dry || (lora_down && lora_up && mult != 0.f)))
      RUNP(SMI_PROF_LORA, 0.0, 0.0, launch_lora_prep(dtype, prep_sites_dev, (int)prep_sites.size(), lora_down, lora_up, lora_shadow, stream));
    // (a forward always rebuilds: the parameters behind the same pointers change with every optimiser step)
    if (lora_down && lora_up && mult != 0.f) dora_prepare(lora_down, lora_up, mult, saving, false);

    // ---- context as a [n*L, D] tensor (borrowed)
    tens->emplace_back();
    Ten* ctx = &tens->back();
    ctx->p = const_cast<void*>(ctxp);
    ctx->rows = (int64_t)n * ctx_len;
    ctx->cols = cfg.cross_attention_dim;
    ctx->n = n;
    ctx->arow0 = (int64_t)(n - n_ad) * ctx_len;

    if (kv_grouped) kv_all_out = linear(ctx, kv_all);
    // (only while nothing upstream of emb is trained: the column-block views carry no gradient)
    temb_all_out = (temb_grouped && !temb_act->ng) ? linear(temb_act, temb_all) : nullptr;

    // ---- conv_in (MFMA implicit GEMM on the 64-channel zero-padded latent)
    Ten* x0 = new_ten((int64_t)n * HW, 64, n, H, Wd_);
    RUN(launch_nchw_to_nhwc(dtype, sample, 1, x0->p, n, cfg.in_channels, HW, 64, 1.f, stream));
    Ten* h = new_ten((int64_t)n * HW, C0, n, H, Wd_);
    {
      GemmParams p;
      p.dtype = dtype;
      p.conv = 1;
      p.A = x0->p;
      p.W = conv_in.Wp;
      p.C = h->p;
      p.ldc = C0;
      p.M = (int)h->rows;
      p.N = C0;
      p.K = 9 * 64;
      p.bias = conv_in.b;
      p.Nb = n;
      p.Hin = p.Hout = H;
      p.Win = p.Wout = Wd_;
      p.Cin = 64;
      RUNP(SMI_PROF_CONV, 2.0 * p.M * p.N * 9 * cfg.in_channels, 0.0, (prof_on && prof_dump ? (next_tag = gemm_tag(p), 0) : 0, launch_gemm(p, stream)));
    }

    // (begin_block / end_block: scratch-region liveness of the no-grad passes, see new_ten; no-ops in a saved pass)
    std::vector<Ten*> skips;
    skips.push_back(h);
    for (size_t i = 0; i < down.size(); ++i) {
      Level& lv = down[i];
      for (size_t j = 0; j < lv.res.size(); ++j) {
        begin_block();  // every down-path block output is a skip connection: kept
        h = resnet(h, lv.res[j], temb_act, !lv.has_att);
        if (lv.has_att) h = transformer(h, lv.att[j], ctx, true);
        end_block();
        skips.push_back(h);
      }
      if (lv.has_samp) {
        begin_block();
        keep_next_output();
        h = conv3x3(h, lv.samp, nullptr, nullptr);
        end_block();
        skips.push_back(h);
      }
    }
    begin_block();
    h = resnet(h, mid.res[0], temb_act);
    end_block();
    begin_block();
    h = transformer(h, mid.att[0], ctx);
    end_block();
    begin_block();
    h = resnet(h, mid.res[1], temb_act);
    end_block();
    for (size_t i = 0; i < up.size(); ++i) {
      Level& lv = up[i];
      for (size_t j = 0; j < lv.res.size(); ++j) {
        Ten* sk = skips.back();
        skips.pop_back();
        begin_block();
        h = concat(h, sk);
        h = resnet(h, lv.res[j], temb_act);
        if (lv.has_att) h = transformer(h, lv.att[j], ctx);
        end_block();
      }
      if (lv.has_samp) {
        begin_block();
        h = conv3x3(h, lv.samp, nullptr, nullptr);
        end_block();
      }
    }
    begin_block();
    Ten* hn = groupnorm(h, norm_out, true);

    // ---- conv_out (MFMA path, fp32 result) + NHWC->NCHW
    Ten* y = new_ten((int64_t)n * HW, cfg.out_channels, n, H, Wd_, sizeof(float));
    {
      GemmParams p;
      p.dtype = dtype;
      p.conv = 1;
      p.A = hn->p;
      p.W = conv_out.Wp;
      p.C = y->p;
      p.ldc = cfg.out_channels;
      p.out_f32 = 1;
      p.M = (int)y->rows;
      p.N = cfg.out_channels;
      p.K = 9 * C0;
      p.bias = conv_out.b;
      p.Nb = n;
      p.Hin = p.Hout = H;
      p.Win = p.Wout = Wd_;
      p.Cin = C0;
      RUNP(SMI_PROF_CONV, 2.0 * p.M * p.N * p.K, 0.0, (prof_on && prof_dump ? (next_tag = gemm_tag(p), 0) : 0, launch_gemm(p, stream)));
    }
    RUN(launch_nhwc_to_nchw_f32((const float*)y->p, eps_out, n, cfg.out_channels, HW, stream));
    end_block();
    y->ng = hn->ng;
    if (saving && y->ng) {
      tape.push_back([=]() {
        if (!y->g) return;
        void* dx = grad_slot(hn).out;
        GemmParams b;
        b.dtype = dtype;
        b.conv = 1;
        b.A = y->g;  // [adapted rows, 64]: d_eps channels zero-padded
        b.W = conv_out.Wg;
        b.C = dx;
        b.ldc = C0;
        b.M = (int)MA(hn);
        b.N = C0;
        b.K = 9 * 64;
        b.Nb = n_adapted;
        b.Hin = b.Hout = H;
        b.Win = b.Wout = Wd_;
        b.Cin = 64;
        RUNP(SMI_PROF_CONV, 2.0 * b.M * b.N * 9 * cfg.out_channels, 0.0, (prof_on && prof_dump ? (next_tag = gemm_tag(b), 0) : 0, launch_gemm(b, stream)));
      });
    }
    if (save) {
      out_ten = y;
      bw_n_ad = n_adapted;
      bw_down = lora_down;
      bw_up = lora_up;
      bw_mult = mult;
      bw_samp_on = samp_on;
      tape_valid = true;
    }
    saving = false;
    if (cur->overflow && !dry) {
      set_error("workspace too small for this call (arena %d needs %zu bytes, has %zu)", save ? 1 : 0, cur->peak,
                cur->cap);
      return -3;
    }
    return err ? -1 : 0;
  }

  int backward(const float* d_eps, float* dd, float* du) {
    if (!tape_valid && !dry) {
      set_error("smi_unet_backward: no saved forward pass (call smi_unet_forward with save_for_backward=1 first)");
      return -4;
    }
    cur = &arena[1];
    tens = &tens_[1];
    d_down = dd;
    d_up = du;
    const int n = bw_n_ad, HW = lat_h * lat_w;  // d_eps covers the adapted samples only
    n_ad = bw_n_ad;
    Ten* y = out_ten;
    if (!y->ng) {  // adaptor off: nothing depends on the LoRA parameters
      tape.clear();
      tape_valid = false;
      return 0;
    }
    if (n > MAXS && !dry) {
      set_error("smi_unet_backward: %d adapted samples exceed the %d per-sample loss scales", n, MAXS);
      return -1;
    }
    wjobs.clear();
    pinned.clear();
    // one power-of-two loss scale PER SAMPLE (from max|d_eps[sample]|): a sample's backward arithmetic then does not
    // depend on which other samples share the batch -- W ranks on shards == one rank on the global batch
    RUN(launch_grad_scale(d_eps, n, (int64_t)cfg.out_channels * HW, gscale, MAXS, stream));
    // the 16-bit LoRA operands / DoRA delta weights belong to the SAVED forward: an adapted no-grad pass in between
    // (pre-roll) may have rebuilt them with other parameters or another multiplier -> rebuild from the saved ones
    if (!prep_sites.empty() && bw_down && bw_up && bw_mult != 0.f)
      RUNP(SMI_PROF_LORA, 0.0, 0.0, launch_lora_prep(dtype, prep_sites_dev, (int)prep_sites.size(), bw_down, bw_up, lora_shadow, stream));
    if (!dora_sites.empty() && bw_down && bw_up && bw_mult != 0.f) {
      dora_prepare(bw_down, bw_up, bw_mult, true, true);
      RUN(launch_scale_min(gscale, n, gscale + 2 * MAXS, stream));
    }
    mult = bw_mult;
    y->g = alloc_t(MA(y), 64);
    RUN(launch_nchw_to_nhwc_scaled(dtype, d_eps, y->g, n, cfg.out_channels, HW, 64, gscale, stream));
    for (auto it = tape.rbegin(); it != tape.rend(); ++it) (*it)();
    if (!wjobs.empty() && !err) {  // every LoRA weight gradient of this pass: one grouped launch per accumulator class
      wgrad_grouped_finish(wjobs);
      if (!dry) {
        if (wjobs.size() > wjobs_cap) {
          set_error("internal: %zu weight-gradient jobs exceed the table (%zu)", wjobs.size(), wjobs_cap);
          return -1;
        }
        const bool same = wjobs_uploaded.size() == wjobs.size() &&
                          memcmp(wjobs_uploaded.data(), wjobs.data(), wjobs.size() * sizeof(WgradJob)) == 0;
        if (!same) {  // bump allocation gives the same addresses every step of a plan: uploaded once, then reused
          wjobs_uploaded = wjobs;
          if (hipMemcpyAsync(wjobs_dev, wjobs_uploaded.data(), wjobs.size() * sizeof(WgradJob), hipMemcpyHostToDevice,
                             stream) != hipSuccess) {
            set_error("hipMemcpyAsync (weight-gradient job table) failed");
            return -1;
          }
        }
        double bytes = 0.0, flops = 0.0;
        for (const auto& j : wjobs) {
          bytes += (double)j.M * j.K * 2.0;
          flops += 2.0 * j.M * j.K * j.r;
        }
        RUNP(SMI_PROF_LORA, flops, bytes, launch_lora_wgrad_grouped(dtype, wjobs, wjobs_dev, stream));
      }
    }
    pinned.clear();
    tape.clear();
    tape_valid = false;
    if (cur->overflow && !dry) {
      set_error("workspace too small for the backward (needs %zu bytes, has %zu)", cur->peak, cur->cap);
      return -3;
    }
    return err ? -1 : 0;
  }
};

// ==============================================================================================================
// C ABI
// ==============================================================================================================
namespace {

int check_cfg(const smi_unet_config* c) {
  SMI_CHECK(c != nullptr, "config is NULL");
  SMI_CHECK(c->dtype == SMI_DTYPE_F16 || c->dtype == SMI_DTYPE_BF16, "dtype must be f16 (0) or bf16 (1)");
  SMI_CHECK(c->n_levels >= 1 && c->n_levels <= SMI_MAX_LEVELS, "n_levels out of range");
  SMI_CHECK(c->in_channels >= 1 && c->in_channels <= 16 && c->out_channels == 4, "in/out channels unsupported");
  for (int i = 0; i < c->n_levels; ++i) {
    SMI_CHECK(c->block_out_channels[i] % 64 == 0, "block_out_channels must be multiples of 64 (MFMA K tile)");
    SMI_CHECK(c->block_out_channels[i] % c->norm_num_groups == 0, "channels must divide by norm_num_groups");
    SMI_CHECK(c->num_heads[i] > 0 && c->block_out_channels[i] % c->num_heads[i] == 0, "heads must divide channels");
    SMI_CHECK((c->block_out_channels[i] / c->num_heads[i]) % 8 == 0, "head_dim must be a multiple of 8");
  }
  SMI_CHECK(c->cross_attention_dim % 8 == 0, "cross_attention_dim %% 8");
  return 0;
}

int setup(smi_engine* e, const smi_unet_config* cfg, const smi_weight* weights, int n_weights,
          const smi_lora_site* sites, int n_sites, int batch, int batch_adapted, int h, int w, int ctx_len) {
  e->cfg = *cfg;
  e->dtype = cfg->dtype;
  e->max_n = batch;
  e->max_n_ad = batch_adapted;
  e->lat_h = h;
  e->lat_w = w;
  e->ctx_len = ctx_len;
  for (int i = 0; i < n_weights; ++i) e->wmap[weights[i].name] = &weights[i];
  e->sites.assign(sites, sites + n_sites);
  e->site_names.resize(n_sites);
  e->site_used.assign(n_sites, 0);
  for (int i = 0; i < n_sites; ++i) {
    e->site_names[i] = sites[i].target;
    e->sites[i].target = e->site_names[i].c_str();
  }
  for (int i = 0; i < n_sites; ++i) e->smap[e->site_names[i]] = &e->sites[i];
  return 0;
}

// dry run: sizes of the three regions
int plan(const smi_unet_config* cfg, const smi_lora_site* sites, int n_sites, int batch, int batch_adapted, int h, int w,
         int ctx_len, size_t out[5]) {
  smi_engine e;
  e.dry = true;
  setup(&e, cfg, nullptr, 0, sites, n_sites, batch, batch_adapted, h, w, ctx_len);
  e.build();
  if (e.err) return -1;
  out[0] = align_up(e.wpack.peak, 4096);
  e.forward(batch, batch_adapted, nullptr, 0.f, nullptr, nullptr, nullptr, false, nullptr);
  // arena 0 = [persistent | scratch 0 | scratch 1] (no-grad liveness, smi_engine::begin_block)
  out[3] = align_up(e.arena[0].peak, 4096);
  out[4] = align_up(std::max(e.scr[0].peak, e.scr[1].peak), 4096);
  out[1] = out[3] + 2 * out[4];
  e.forward(batch, batch_adapted, nullptr, 0.f, nullptr, nullptr, nullptr, true, nullptr);
  e.backward(nullptr, nullptr, nullptr);
  out[2] = align_up(e.arena[1].peak, 4096);
  return e.err ? -1 : 0;
}

}  // namespace

extern "C" {

const char* smi_last_error(void) { return g_err; }

int smi_workspace_bytes(const smi_unet_config* cfg, const smi_lora_site* sites, int n_sites, int batch,
                        int batch_adapted, int h, int w, int ctx_len, size_t* bytes) {
  if (check_cfg(cfg)) return -1;
  SMI_CHECK(bytes && batch > 0 && batch_adapted >= 0 && batch_adapted <= batch && h > 0 && w > 0 && ctx_len > 0,
            "bad arguments");
  size_t r[5];
  if (plan(cfg, sites, n_sites, batch, batch_adapted, h, w, ctx_len, r)) return -1;
  *bytes = r[0] + r[1] + r[2] + 3 * 4096;
  return 0;
}

int smi_create(const smi_unet_config* cfg, const smi_weight* weights, int n_weights, const smi_lora_site* sites,
               int n_sites, int batch, int batch_adapted, int h, int w, int ctx_len, void* workspace,
               size_t workspace_bytes, void* stream, smi_engine** out) {
  if (check_cfg(cfg)) return -1;
  SMI_CHECK(out && workspace && weights && n_weights > 0 && batch_adapted >= 0 && batch_adapted <= batch,
            "bad arguments");
  size_t r[5];
  if (plan(cfg, sites, n_sites, batch, batch_adapted, h, w, ctx_len, r)) return -1;
  SMI_CHECK(r[0] + r[1] + r[2] + 3 * 4096 <= workspace_bytes, "workspace too small: need %zu bytes, got %zu",
            r[0] + r[1] + r[2] + 3 * 4096, workspace_bytes);
  smi_engine* e = new smi_engine();
  e->stream = (hipStream_t)stream;
  setup(e, cfg, weights, n_weights, sites, n_sites, batch, batch_adapted, h, w, ctx_len);
  char* base = (char*)align_up((size_t)workspace, 4096);
  e->ws = (char*)workspace;
  e->ws_bytes = workspace_bytes;
  e->wpack.base = base;
  e->wpack.cap = r[0];
  e->arena[0].base = base + r[0];
  e->arena[1].base = base + r[0] + r[1];
  e->arena[1].cap = r[2];
  e->set_arena0_regions(r[3], r[4]);
  e->arena_home = base + r[0];
  e->arena_home_bytes = (size_t)((char*)workspace + workspace_bytes - e->arena_home);
  e->build();
  if (!e->err) (void)hipStreamSynchronize(e->stream);
  if (e->err || hipGetLastError() != hipSuccess) {
    if (!e->err) set_error("HIP error while packing weights");
    delete e;
    return -1;
  }
  // the weight table is only borrowed during creation
  e->wmap.clear();
  *out = e;
  return 0;
}

void smi_destroy(smi_engine* e) {
  if (!e) return;
  delete e;
}

// ---- AutoencoderKL encoder engine ------------------------------------------------------------------------------------
static int check_vae_cfg(const smi_vae_config* c, int batch, int h, int w) {
  SMI_CHECK(c != nullptr, "config is NULL");
  SMI_CHECK(c->dtype == SMI_DTYPE_F16 || c->dtype == SMI_DTYPE_BF16, "dtype must be f16 (0) or bf16 (1)");
  SMI_CHECK(c->n_levels >= 1 && c->n_levels <= SMI_MAX_LEVELS && c->in_channels >= 1 && c->in_channels <= 16 &&
                c->latent_channels >= 1 && c->latent_channels <= 8 && c->layers_per_block >= 1,
            "VAE config out of range");
  for (int i = 0; i < c->n_levels; ++i)
    SMI_CHECK(c->block_out_channels[i] % 64 == 0 && c->block_out_channels[i] % c->norm_num_groups == 0,
              "VAE block_out_channels must be multiples of 64 and of norm_num_groups");
  const int f = 1 << (c->n_levels - 1);
  SMI_CHECK(batch > 0 && h > 0 && w > 0 && h % f == 0 && w % f == 0, "image size must be a multiple of %d", f);
  return 0;
}
static void vae_setup(smi_engine* e, const smi_vae_config* cfg, const smi_weight* weights, int n_weights, int batch, int h,
                      int w) {
  e->is_vae = true;
  e->vcfg = *cfg;
  e->dtype = cfg->dtype;
  e->cfg.norm_num_groups = cfg->norm_num_groups;
  e->cfg.block_out_channels[0] = cfg->block_out_channels[0];
  e->max_n = batch;
  e->lat_h = h;
  e->lat_w = w;
  for (int i = 0; i < n_weights; ++i) e->wmap[weights[i].name] = &weights[i];
}
static int vae_plan(const smi_vae_config* cfg, int batch, int h, int w, size_t out[2]) {
  smi_engine e;
  e.dry = true;
  vae_setup(&e, cfg, nullptr, 0, batch, h, w);
  e.build_vae();
  if (e.err) return -1;
  out[0] = align_up(e.wpack.peak, 4096);
  e.forward_vae(batch, nullptr, nullptr);
  out[1] = align_up(e.arena[0].peak, 4096);
  return e.err ? -1 : 0;
}

int smi_vae_workspace_bytes(const smi_vae_config* cfg, int batch, int h, int w, size_t* bytes) {
  if (check_vae_cfg(cfg, batch, h, w)) return -1;
  SMI_CHECK(bytes != nullptr, "bad arguments");
  size_t r[2];
  if (vae_plan(cfg, batch, h, w, r)) return -1;
  *bytes = r[0] + r[1] + 2 * 4096;
  return 0;
}

int smi_vae_create(const smi_vae_config* cfg, const smi_weight* weights, int n_weights, int batch, int h, int w,
                   void* workspace, size_t workspace_bytes, void* stream, smi_engine** out) {
  if (check_vae_cfg(cfg, batch, h, w)) return -1;
  SMI_CHECK(out && workspace && weights && n_weights > 0, "bad arguments");
  size_t r[2];
  if (vae_plan(cfg, batch, h, w, r)) return -1;
  SMI_CHECK(r[0] + r[1] + 2 * 4096 <= workspace_bytes, "workspace too small: need %zu bytes, got %zu",
            r[0] + r[1] + 2 * 4096, workspace_bytes);
  smi_engine* e = new smi_engine();
  e->stream = (hipStream_t)stream;
  vae_setup(e, cfg, weights, n_weights, batch, h, w);
  char* base = (char*)align_up((size_t)workspace, 4096);
  e->ws = (char*)workspace;
  e->ws_bytes = workspace_bytes;
  e->wpack.base = base;
  e->wpack.cap = r[0];
  e->arena[0].base = base + r[0];
  e->arena[0].cap = r[1];
  e->build_vae();
  if (!e->err) (void)hipStreamSynchronize(e->stream);
  if (e->err || hipGetLastError() != hipSuccess) {
    if (!e->err) set_error("HIP error while packing the VAE weights");
    delete e;
    return -1;
  }
  e->wmap.clear();
  *out = e;
  return 0;
}

int smi_vae_encode(smi_engine* e, int n, const float* image, float* moments_out) {
  SMI_CHECK(e && e->is_vae && image && moments_out, "smi_vae_encode: NULL argument or not a VAE engine");
  SMI_CHECK(n >= 1 && n <= e->max_n, "batch %d outside [1, %d] the engine was created for", n, e->max_n);
  e->err = false;
  return e->forward_vae(n, image, moments_out);
}

// ---- CLIP text encoder engine ------------------------------------------------------------------------------------------
static int check_clip_cfg(const smi_clip_config* c, int batch) {
  SMI_CHECK(c != nullptr, "config is NULL");
  SMI_CHECK(c->dtype == SMI_DTYPE_F16 || c->dtype == SMI_DTYPE_BF16, "dtype must be f16 (0) or bf16 (1)");
  SMI_CHECK(c->hidden_size % 64 == 0 && c->hidden_size <= 2048 && c->num_heads > 0 && c->hidden_size % c->num_heads == 0 &&
                (c->hidden_size / c->num_heads) % 8 == 0 && c->intermediate_size % 64 == 0 && c->num_layers >= 1 &&
                c->vocab_size > 0 && c->max_positions > 0 && (c->hidden_act == 0 || c->hidden_act == 1) &&
                c->projection_dim >= 0 && c->projection_dim % 8 == 0 && batch > 0,
            "CLIP config out of range (hidden %% 64, hidden <= 2048, head_dim %% 8, intermediate %% 64)");
  return 0;
}
static void clip_setup(smi_engine* e, const smi_clip_config* cfg, const smi_weight* weights, int n_weights, int batch) {
  e->is_clip = true;
  e->ccfg = *cfg;
  e->dtype = cfg->dtype;
  e->max_n = batch;
  for (int i = 0; i < n_weights; ++i) e->wmap[weights[i].name] = &weights[i];
}
static int clip_plan(const smi_clip_config* cfg, int batch, size_t out[2]) {
  smi_engine e;
  e.dry = true;
  clip_setup(&e, cfg, nullptr, 0, batch);
  e.build_clip();
  if (e.err) return -1;
  out[0] = align_up(e.wpack.peak, 4096);
  e.forward_clip(batch, nullptr, nullptr, (void*)16, (void*)16, (void*)16);
  out[1] = align_up(e.arena[0].peak, 4096);
  return e.err ? -1 : 0;
}

int smi_clip_workspace_bytes(const smi_clip_config* cfg, int batch, size_t* bytes) {
  if (check_clip_cfg(cfg, batch)) return -1;
  SMI_CHECK(bytes != nullptr, "bad arguments");
  size_t r[2];
  if (clip_plan(cfg, batch, r)) return -1;
  *bytes = r[0] + r[1] + 2 * 4096;
  return 0;
}

int smi_clip_create(const smi_clip_config* cfg, const smi_weight* weights, int n_weights, int batch, void* workspace,
                    size_t workspace_bytes, void* stream, smi_engine** out) {
  if (check_clip_cfg(cfg, batch)) return -1;
  SMI_CHECK(out && workspace && weights && n_weights > 0, "bad arguments");
  size_t r[2];
  if (clip_plan(cfg, batch, r)) return -1;
  SMI_CHECK(r[0] + r[1] + 2 * 4096 <= workspace_bytes, "workspace too small: need %zu bytes, got %zu",
            r[0] + r[1] + 2 * 4096, workspace_bytes);
  smi_engine* e = new smi_engine();
  e->stream = (hipStream_t)stream;
  clip_setup(e, cfg, weights, n_weights, batch);
  char* base = (char*)align_up((size_t)workspace, 4096);
  e->ws = (char*)workspace;
  e->ws_bytes = workspace_bytes;
  e->wpack.base = base;
  e->wpack.cap = r[0];
  e->arena[0].base = base + r[0];
  e->arena[0].cap = r[1];
  e->build_clip();
  if (!e->err) (void)hipStreamSynchronize(e->stream);
  if (e->err || hipGetLastError() != hipSuccess) {
    if (!e->err) set_error("HIP error while packing the CLIP weights");
    delete e;
    return -1;
  }
  e->wmap.clear();
  *out = e;
  return 0;
}

int smi_clip_encode(smi_engine* e, int n, const int32_t* ids, const int32_t* eos_pos, void* last_hidden,
                    void* penultimate, void* pooled) {
  SMI_CHECK(e && e->is_clip && ids, "smi_clip_encode: NULL argument or not a CLIP engine");
  SMI_CHECK(n >= 1 && n <= e->max_n, "batch %d outside [1, %d] the engine was created for", n, e->max_n);
  SMI_CHECK(!pooled || eos_pos, "pooled output needs eos_pos");
  e->err = false;
  return e->forward_clip(n, ids, eos_pos, last_hidden, penultimate, pooled);
}

int smi_weights_bytes(const smi_unet_config* cfg, const smi_lora_site* sites, int n_sites, size_t* bytes) {
  if (check_cfg(cfg)) return -1;
  SMI_CHECK(bytes != nullptr, "bad arguments");
  size_t r[5];
  if (plan(cfg, sites, n_sites, 1, 1, 8, 8, 8, r)) return -1;  // the packed region does not depend on the shape
  *bytes = r[0] + 4096;
  return 0;
}

int smi_arena_bytes(const smi_unet_config* cfg, const smi_lora_site* sites, int n_sites, int batch, int batch_adapted,
                    int h, int w, int ctx_len, size_t* bytes) {
  if (check_cfg(cfg)) return -1;
  SMI_CHECK(bytes && batch > 0 && batch_adapted >= 0 && batch_adapted <= batch && h > 0 && w > 0 && ctx_len > 0,
            "bad arguments");
  size_t r[5];
  if (plan(cfg, sites, n_sites, batch, batch_adapted, h, w, ctx_len, r)) return -1;
  *bytes = r[1] + r[2] + 2 * 4096;
  return 0;
}

int smi_replan(smi_engine* e, int batch, int batch_adapted, int h, int w, int ctx_len, void* arena,
               size_t arena_bytes) {
  SMI_CHECK(e && batch > 0 && batch_adapted >= 0 && batch_adapted <= batch && h > 0 && w > 0 && ctx_len > 0,
            "bad arguments");
  SMI_CHECK(!e->is_vae && !e->is_clip, "smi_replan: only UNet engines re-plan");
  size_t r[5];
  if (plan(&e->cfg, e->sites.data(), (int)e->sites.size(), batch, batch_adapted, h, w, ctx_len, r)) return -1;
  char* base = arena ? (char*)align_up((size_t)arena, 4096) : e->arena_home;
  const size_t have = arena ? (size_t)((char*)arena + arena_bytes - base) : e->arena_home_bytes;
  SMI_CHECK(r[1] + r[2] <= have, "arena too small for batch %d (%d adapted), %dx%d latents: need %zu bytes, got %zu", batch,
            batch_adapted, h, w, r[1] + r[2] + 2 * 4096, arena ? arena_bytes : e->arena_home_bytes);
  e->wjobs_uploaded.clear();
  e->max_n = batch;
  e->max_n_ad = batch_adapted;
  e->lat_h = h;
  e->lat_w = w;
  e->ctx_len = ctx_len;
  e->arena[0] = Arena();
  e->arena[1] = Arena();
  e->arena[0].base = base;
  e->arena[1].base = base + r[1];
  e->arena[1].cap = r[2];
  e->set_arena0_regions(r[3], r[4]);
  e->tape.clear();  // the saved activations lived in the old arena
  e->tape_valid = false;
  e->tens_[0].clear();
  e->tens_[1].clear();
  e->out_ten = nullptr;
  ++e->replans;
  return 0;
}

int smi_engine_stats(const smi_engine* e, int64_t out[4]) {
  SMI_CHECK(e && out, "NULL argument");
  out[0] = e->pack_launches;
  out[1] = e->replans;
  out[2] = e->tape_valid ? (int64_t)e->tape_gen : 0;
  out[3] = (int64_t)e->wpack.peak;
  return 0;
}

namespace {
// the engine lends its split-K scratch to launch_gemm for the duration of one call on this host thread
struct GemmScratchScope {
  GemmScratchScope(void* ws, size_t bytes) { smi::set_gemm_scratch(ws, bytes); }
  ~GemmScratchScope() { smi::set_gemm_scratch(nullptr, 0); }
};
}  // namespace

int smi_unet_forward_batched(smi_engine* e, int n, int n_adapted, const float* sample, float timestep, const void* ctx,
                             const void* text_embeds, const float* time_ids, const float* lora_down_flat,
                             const float* lora_up_flat, float multiplier, int save_for_backward, float* eps_out) {
  SMI_CHECK(e && sample && ctx && eps_out, "NULL argument");
  SMI_CHECK(!e->is_vae && !e->is_clip, "this engine is a VAE / CLIP encoder (use smi_vae_encode / smi_clip_encode)");
  SMI_CHECK(n >= 1 && n <= e->max_n, "batch %d outside [1, %d] the engine was created for", n, e->max_n);
  SMI_CHECK(n_adapted >= 0 && n_adapted <= n && n_adapted <= e->max_n_ad,
            "adapted batch %d outside [0, min(%d, %d)]", n_adapted, n, e->max_n_ad);
  SMI_CHECK(!e->cfg.addition_embed || (text_embeds && time_ids), "SD-XL engine needs text_embeds and time_ids");
  e->err = false;
  e->lora_down = lora_down_flat;
  e->lora_up = lora_up_flat;
  e->mult = (lora_down_flat && lora_up_flat && n_adapted > 0) ? multiplier : 0.f;
  GemmScratchScope scratch(e->splitk_ws, smi_engine::SPLITK_WS_BYTES);
  return e->forward(n, n_adapted, sample, timestep, ctx, text_embeds, time_ids, save_for_backward != 0, eps_out);
}

int smi_unet_forward_multi(smi_engine* e, int n, int n_adapted, const float* sample, float timestep, const void* ctx,
                           const void* text_embeds, const float* time_ids, const float* lora_down_flat,
                           const float* lora_up_flat, const float* multipliers, int save_for_backward, float* eps_out) {
  SMI_CHECK(e && multipliers, "NULL argument");
  SMI_CHECK(n_adapted >= 1 && n_adapted <= smi_engine::MAXS, "per-sample multipliers: 1..%d adapted samples", smi_engine::MAXS);
  float mref = 0.f;
  bool same = true;
  for (int i = 0; i < n_adapted; ++i) {
    if (fabsf(multipliers[i]) > mref) mref = fabsf(multipliers[i]);
    same = same && multipliers[i] == multipliers[0];
  }
  if (same || mref == 0.f)
    return smi_unet_forward_batched(e, n, n_adapted, sample, timestep, ctx, text_embeds, time_ids, lora_down_flat,
                                    lora_up_flat, multipliers[0], save_for_backward, eps_out);
  SMI_CHECK(e->n_conv_sites == 0 && e->dora_sites.empty(),
            "per-sample multipliers are implemented for Linear LoRA sites (attention projections, time_emb_proj, "
            "conv_shortcut); this network has conv or DoRA sites -- run the samples in separate passes");
  // the per-sample ratios travel in kernel arguments, 64 per launch, ordered on the engine's stream (an async
  // copy from a stack array is only correct while the runtime stages pageable sources before returning)
  for (int i0 = 0; i0 < n_adapted; i0 += 64) {
    F64Args a;
    const int cnt = n_adapted - i0 < 64 ? n_adapted - i0 : 64;
    for (int i = 0; i < 64; ++i) a.v[i] = i < cnt ? multipliers[i0 + i] / mref : 0.f;
    hipLaunchKernelGGL(set_floats_kernel, dim3(1), dim3(64), 0, e->stream, e->samp_mult_dev + i0,
                       save_for_backward ? e->samp_mult_dev + smi_engine::MAXS + i0 : (float*)nullptr, a, cnt);
  }
  SMI_HIP(hipGetLastError());
  e->samp_on = true;
  const int rc = smi_unet_forward_batched(e, n, n_adapted, sample, timestep, ctx, text_embeds, time_ids, lora_down_flat,
                                          lora_up_flat, mref, save_for_backward, eps_out);
  e->samp_on = false;
  return rc;
}

int smi_unet_forward(smi_engine* e, int n, const float* sample, float timestep, const void* ctx,
                     const void* text_embeds, const float* time_ids, const float* lora_down_flat,
                     const float* lora_up_flat, float multiplier, int save_for_backward, float* eps_out) {
  SMI_CHECK(e != nullptr, "NULL argument");
  return smi_unet_forward_batched(e, n, n < e->max_n_ad ? n : e->max_n_ad, sample, timestep, ctx, text_embeds, time_ids,
                                  lora_down_flat, lora_up_flat, multiplier, save_for_backward, eps_out);
}

int smi_unet_backward(smi_engine* e, const float* d_eps, float* d_lora_down_flat, float* d_lora_up_flat) {
  SMI_CHECK(e && d_eps && d_lora_down_flat && d_lora_up_flat, "NULL argument");
  SMI_CHECK(!e->is_vae && !e->is_clip, "this engine is a VAE / CLIP encoder: it has no backward");
  e->err = false;
  GemmScratchScope scratch(e->splitk_ws, smi_engine::SPLITK_WS_BYTES);
  return e->backward(d_eps, d_lora_down_flat, d_lora_up_flat);
}

int smi_profile_enable(smi_engine* e, int enable) {
  SMI_CHECK(e != nullptr, "NULL engine");
  e->prof_collect();
  e->prof_on = enable != 0;
  for (int i = 0; i < SMI_PROF_NCAT; ++i) {
    e->prof_ms[i] = 0;
    e->prof_flops[i] = 0;
    e->prof_bytes[i] = 0;
    e->prof_launches[i] = 0;
  }
  return 0;
}
int smi_profile_read(smi_engine* e, double* ms, double* flops, double* bytes, int64_t* launches) {
  SMI_CHECK(e && ms && flops && bytes && launches, "NULL argument");
  e->prof_collect();
  if (e->prof_dump && !e->prof_tags.empty()) {
    std::vector<std::pair<std::string, smi_engine::TagAcc>> v(e->prof_tags.begin(), e->prof_tags.end());
    std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.second.ms > b.second.ms; });
    double tot = 0;
    for (auto& kv : v) tot += kv.second.ms;
    fprintf(stderr, "[smi profile] GEMM / conv launches by shape (%.2f ms total)\n", tot);
    for (auto& kv : v)
      fprintf(stderr, "  %8.3f ms %5d x %8.1f us %7.0f TF/s  %s\n", kv.second.ms, kv.second.n,
              kv.second.ms * 1e3 / kv.second.n, kv.second.flops / (kv.second.ms * 1e-3) / 1e12, kv.first.c_str());
    e->prof_tags.clear();
  }
  for (int i = 0; i < SMI_PROF_NCAT; ++i) {
    ms[i] = e->prof_ms[i];
    flops[i] = e->prof_flops[i];
    bytes[i] = e->prof_bytes[i];
    launches[i] = e->prof_launches[i];
  }
  return 0;
}

int smi_cfg_combine(const float* eps_2n, float* out_n, int64_t n_half, float g, void* stream) {
  return launch_cfg_combine(eps_2n, out_n, n_half, g, (hipStream_t)stream);
}
int smi_slider_loss(const float* target, const float* positive, const float* neutral, const float* negative,
                    float sign_eta, int64_t n, float* loss_out, float* dtarget, float* scratch, void* stream) {
  return launch_slider_loss(target, positive, neutral, negative, sign_eta, n, loss_out, dtarget, scratch,
                            (hipStream_t)stream);
}
int smi_clip_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float max_norm, float* scratch,
                   void* stream) {
  return launch_clip_adamw(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, max_norm,
                           scratch, (hipStream_t)stream);
}
int smi_sched_step(float* x, const float* eps, const float* noise, float c_x, float c_eps, float c_noise, int64_t n,
                   void* stream) {
  return launch_sched_affine(x, eps, noise, c_x, c_eps, c_noise, n, (hipStream_t)stream);
}

// ---- single-kernel entry points for the parity tests -----------------------------------------------------------
int smi_op_gemm_scratch(void* ws, size_t bytes) {
  smi::set_gemm_scratch(ws, bytes);
  return 0;
}
int smi_op_gemm(int dtype, const void* A, const void* W, void* C, int M, int N, int K, const void* bias,
                const void* res, const float* lora_xa, const float* lora_up, int lora_r, float lora_scale,
                int out_f32, void* stream) {
  GemmParams p;
  p.dtype = dtype;
  p.A = A;
  p.lda = K;
  p.W = W;
  p.C = C;
  p.ldc = N;
  p.out_f32 = out_f32;
  p.M = M;
  p.N = N;
  p.K = K;
  p.bias = bias;
  p.res = res;
  p.ldr = N;
  p.lora_xa = lora_xa;
  p.ld_xa = lora_r;
  p.lora_up = lora_up;
  p.up_sn = lora_r;
  p.up_sq = 1;
  p.lora_r = lora_xa ? lora_r : 0;
  p.lora_scale = lora_scale;
  return launch_gemm(p, (hipStream_t)stream);
}
int smi_op_gemm_rows(int dtype, const void* A, const void* W, void* C, int M, int N, int K, const void* bias,
                     const void* res, const float* lora_xa, const float* lora_up, int lora_r, float lora_scale,
                     int lora_row0, int lora_seg, void* stream) {
  GemmParams p;
  p.dtype = dtype;
  p.A = A;
  p.lda = K;
  p.W = W;
  p.C = C;
  p.ldc = N;
  p.M = M;
  p.N = N;
  p.K = K;
  p.bias = bias;
  p.res = res;
  p.ldr = N;
  p.lora_xa = lora_xa;
  p.ld_xa = (int64_t)lora_r * (lora_seg > 0 ? N / lora_seg : 1);
  p.lora_up = lora_up;
  p.up_sn = lora_r;
  p.up_sq = 1;
  p.lora_r = lora_xa ? lora_r : 0;
  p.lora_seg = lora_seg;
  p.lora_scale = lora_scale;
  p.lora_row0 = lora_row0;
  return launch_gemm(p, (hipStream_t)stream);
}
int smi_op_gemm_geglu(int dtype, const void* A, const void* W, const void* bias, void* out, void* proj, int M, int N,
                      int K, int proj_row0, void* stream) {
  GemmParams p;
  p.dtype = dtype;
  p.A = A;
  p.lda = K;
  p.W = W;
  p.C = proj;
  p.ldc = N;
  p.M = M;
  p.N = N;
  p.K = K;
  p.bias = bias;
  p.geglu_out = out;
  p.geglu_row0 = proj_row0;
  return launch_gemm(p, (hipStream_t)stream);
}
int smi_op_conv3x3(int dtype, const void* in, const void* w_packed, const void* bias, void* out, int nb, int hin,
                   int win, int cin, int cout, int stride, int upsample, int transposed, int hout, int wout,
                   void* stream) {
  GemmParams p;
  p.dtype = dtype;
  p.conv = 1;
  p.A = in;
  p.W = w_packed;
  p.C = out;
  p.ldc = cout;
  p.M = nb * hout * wout;
  p.N = cout;
  p.K = 9 * cin;
  p.bias = bias;
  p.Nb = nb;
  p.Hin = hin;
  p.Win = win;
  p.Cin = cin;
  p.Hout = hout;
  p.Wout = wout;
  p.stride = stride;
  p.upsample = upsample;
  p.transposed = transposed;
  return launch_gemm(p, (hipStream_t)stream);
}
int smi_op_attention_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int b, int h,
                         int nq, int nk, int d, float scale, void* stream) {
  AttnParams p;
  p.dtype = dtype;
  p.Q = q;
  p.K = k;
  p.V = v;
  p.O = o;
  p.lse = lse;
  p.ldq = p.ldk = p.ldv = p.ldo = (int64_t)h * d;
  p.B = b;
  p.H = h;
  p.Nq = nq;
  p.Nk = nk;
  p.D = d;
  p.scale = scale;
  return launch_attn_fwd(p, (hipStream_t)stream);
}
int smi_op_attention_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const float* lse,
                         const void* d_o, void* dq, void* dk, void* dv, float* delta, int b, int h, int nq, int nk,
                         int d, float scale, void* stream) {
  AttnParams p;
  p.dtype = dtype;
  p.Q = q;
  p.K = k;
  p.V = v;
  p.O = const_cast<void*>(o);
  p.lse = const_cast<float*>(lse);
  p.ldq = p.ldk = p.ldv = p.ldo = p.lddo = p.lddq = p.lddk = p.lddv = (int64_t)h * d;
  p.dO = d_o;
  p.dQ = dq;
  p.dK = dk;
  p.dV = dv;
  p.delta = delta;
  p.B = b;
  p.H = h;
  p.Nq = nq;
  p.Nk = nk;
  p.D = d;
  p.scale = scale;
  return launch_attn_bwd(p, (hipStream_t)stream);
}
// scratch layout (floats): ab[2*nb*c] | mean_rstd[nb*g*2] | partial[...]
int smi_gn_coop_timeouts(void) { return smi::gn_coop_timeouts(); }
int smi_op_groupnorm(int dtype, const void* x, const void* gamma, const void* beta, void* y, const void* dy, void* dx,
                     float* scratch, int nb, int hw, int c, int g, float eps, int silu, void* stream) {
  float* ab = scratch;
  float* mr = ab + (size_t)2 * nb * c;
  float* part = mr + (size_t)nb * g * 2;
  int rc = launch_groupnorm_fwd(dtype, x, gamma, beta, y, ab, mr, part, nb, hw, c, g, eps, silu, (hipStream_t)stream);
  if (rc || !dy) return rc;
  return launch_groupnorm_bwd(dtype, x, dy, gamma, beta, ab, ab + (size_t)nb * c, mr, nullptr, dx, part, nb, hw, c, g, silu,
                              (hipStream_t)stream);
}
int smi_op_layernorm(int dtype, const void* x, const void* gamma, const void* beta, void* y, const void* dy, void* dx,
                     float* mean_rstd, int m, int c, float eps, void* stream) {
  int rc = launch_layernorm_fwd(dtype, x, gamma, beta, y, mean_rstd, m, c, eps, (hipStream_t)stream);
  if (rc || !dy) return rc;
  return launch_layernorm_bwd(dtype, x, dy, gamma, mean_rstd, nullptr, dx, m, c, (hipStream_t)stream);
}
int smi_op_geglu(int dtype, const void* proj, void* out, const void* dout, void* dproj, int m, int c4, void* stream) {
  int rc = launch_geglu_fwd(dtype, proj, out, m, c4, (hipStream_t)stream);
  if (rc || !dout) return rc;
  return launch_geglu_bwd(dtype, proj, dout, dproj, m, c4, (hipStream_t)stream);
}
int smi_op_lora_down(int dtype, const void* x, const float* a, float* xa, int m, int k, int r, void* stream) {
  return launch_lora_down(dtype, x, k, a, k, 1, xa, r, m, k, r, (hipStream_t)stream);
}
int smi_op_lora_skinny(int dtype, const void* x, const void* s, float* out, int m, int r, int k, void* stream) {
  return launch_lora_skinny(dtype, x, k, s, out, r, m, r, k, (hipStream_t)stream);
}
int smi_op_lora_wgrad(int dtype, const float* p, const void* x, float* dw, int m, int k, int r, float alpha,
                      float* scratch, void* stream) {
  // the engine's grouped reduction with a one-job table (scratch: the job's partials followed by the table itself)
  std::vector<WgradJob> jobs(1);
  WgradJob& j = jobs[0];
  memset(&j, 0, sizeof(j));
  j.conv_tap = -1;
  j.X = x;
  j.ldx = k;
  j.P = p;
  j.ldp = r;
  j.dW = dw;
  j.so_r = k;
  j.so_k = 1;
  j.M = m;
  j.K = k;
  j.r = r;
  j.rows_per_sample = m;
  j.alpha = alpha;
  wgrad_job_plan(j);
  j.partial = scratch;
  wgrad_grouped_finish(jobs);
  WgradJob* dev = reinterpret_cast<WgradJob*>(scratch + ((wgrad_job_scratch_floats(j) + 63) / 64) * 64);
  SMI_HIP(hipMemcpyAsync(dev, jobs.data(), sizeof(WgradJob), hipMemcpyHostToDevice, (hipStream_t)stream));
  return launch_lora_wgrad_grouped(dtype, jobs, dev, (hipStream_t)stream);
}

}  // extern "C"
