// Common device/host helpers for the gfx950 (MI355X, CDNA4) slider-trainer kernels.
// Everything here is written for wave64 / MFMA / buffer addressing on gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>

namespace smi {

typedef _Float16 f16;
typedef __bf16 bf16;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

enum DType : int { DT_F16 = 0, DT_BF16 = 1 };

// ---------------------------------------------------------------------------------------------------
// per-dtype traits: 8-wide fragment type and the two MFMA shapes used (16x16x32 for GEMM/conv,
// 32x32x16 for attention).  Operand maps (cdna_hip_programming.md section 3):
//   16x16x32: lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], j=0..7
//             D: col = l&15, row = 4(l>>4)+reg
//   32x32x16: lane l holds A[row l&31][k = 8(l>>5)+j], B[k = 8(l>>5)+j][col l&31]
//             D: col = l&31, row = (reg&3) + 8(reg>>2) + 4(l>>5)
// ---------------------------------------------------------------------------------------------------
template <typename T> struct TT;
template <> struct TT<f16> {
  typedef f16x8 v8;
  typedef f16x4 v4;
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct TT<bf16> {
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <typename T> __device__ __forceinline__ float to_f(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x) { return (T)x; }

// 16-byte / 8-byte reinterpretation helpers
template <typename T> union Pack8 {
  u32x4 u;
  typename TT<T>::v8 v;
  T e[8];
};
template <typename T> union Pack4 {
  u32x2 u;
  typename TT<T>::v4 v;
  T e[4];
};

// ---------------------------------------------------------------------------------------------------
// buffer addressing: hardware range check -> out-of-range loads return 0, stores are dropped.
// Used for conv zero padding, M/N tails, and as a guard against wild accesses.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ u32x2 buf_load8(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store8(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)byte_off, 0, 0);
}
static constexpr uint32_t OOB = 0xFFFFFFF0u;  // voffset that is out of range for every descriptor we build (< 4 GiB - 16)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
  float s = 1.f / (1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the 16-bit output rounding): one v_rcp, one v_exp
// and five FMAs instead of libm erff's ~40 instructions -- the exact-erf GELU sits in a GEMM epilogue's critical path
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
// gelu(x) = x Phi(x) with Phi from the same A&S 7.1.26 erf, folded so that the sign needs no copysign / select:
//   x >= 0: x (1 - q),  x < 0: x q,  q = 0.5 poly(t) exp(-x^2 / 2)   ==   max(x, 0) - |x| q
// (14 VALU operations, v_rcp + v_exp among them; the bare v_exp_f32 is enough -- where its result would be subnormal q
// is far below the rounding of x).  This sits in the epilogue of the largest GEMM of the step (ff.net.0: 80 gates per lane
// and 256 x 320 tile), which is pure VALU time the MFMAs wait for.
__device__ __forceinline__ float gelu_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.3275911f * 0.70710678118654752f, 1.f));
  const float poly =
      t * (0.5f * 0.254829592f +
           t * (0.5f * -0.284496736f + t * (0.5f * 1.421413741f + t * (0.5f * -1.453152027f + t * (0.5f * 1.061405429f)))));
  const float q = poly * __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));
  return __builtin_fmaf(-ax, q, fmaxf(x, 0.f));
}
__device__ __forceinline__ float dgelu_f(float x) {
  return 0.5f * (1.f + erf_as(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

// Canonical arithmetic of the LoRA delta in every GEMM epilogue: ONE sequential fp32 FMA chain over the rank index
// (ascending, from zero), then one FMA with the scale onto the running value.  Written with explicit fmaf so that the
// result does not depend on how hipcc contracts / vectorises a sum of products in a given kernel, and because this is
// exactly what v_mfma_f32_16x16x4_f32 computes (a k-ordered fmaf chain, cdna guide section 3) -- a VALU epilogue and an
// MFMA epilogue give the same bits.
__device__ __forceinline__ float lora_fma4(float d, f32x4 x, f32x4 u) {
  d = __builtin_fmaf(x[0], u[0], d);
  d = __builtin_fmaf(x[1], u[1], d);
  d = __builtin_fmaf(x[2], u[2], d);
  return __builtin_fmaf(x[3], u[3], d);
}

// host side ------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define SMI_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      ::smi::set_error(__VA_ARGS__);    \
      return -1;                        \
    }                                   \
  } while (0)
#define SMI_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) {                                                           \
      ::smi::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return -2;                                                                       \
    }                                                                                  \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device): the attribute belongs to the
// function ON the current device, and engines on several devices / host threads may share one process.
// Use: `static DynLdsOnce once; if (int rc = once.set((const void*)kernel, bytes)) return rc;`
struct DynLdsOnce {
  std::mutex mu;
  uint64_t done = 0;  // one bit per device ordinal
  int set(const void* fn, int bytes) {
    int dev = 0;
    SMI_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && ((done >> dev) & 1)) return 0;
    SMI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev >= 0 && dev < 64) done |= 1ull << dev;
    return 0;
  }
};

}  // namespace smi
