// Shared device/host helpers for libdsrg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include "../../include/dsrg_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits

using f32x4 = __attribute__((ext_vector_type(4))) float;
using s16x8 = __attribute__((ext_vector_type(8))) short;

// f32 -> bf16, round-to-nearest-even, NaN stays NaN (compiler emits v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int PER16 = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int PER16 = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// load / store 4 consecutive elements as float4 (pointer 8B-aligned for bf16, 16B for f32)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                     __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  uint2 u;
  u.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
  u.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}

// raw (unconverted) 4-element loads: keeping the conversion away from the load lets a batch of loads go out together
__device__ __forceinline__ float4 raw_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 raw_ld4(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
template <typename T> struct Raw4 { typedef float4 type; };
template <> struct Raw4<bf16_t> { typedef uint2 type; };
__device__ __forceinline__ float4 raw4(float4 v) { return v; }
__device__ __forceinline__ float4 raw4(uint2 u) {     // 4 packed bf16
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

// non-temporal forms (the GEMM's C tiles are read next by another kernel on other XCDs: keep them out of the way of A/W in L2)
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(p));
}
__device__ __forceinline__ void st4_nt(bf16_t* p, float4 v) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u2 u;
  u.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
  u.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
  __builtin_nontemporal_store(u, reinterpret_cast<u2*>(p));
}

// Wave-wide (64-lane) all-reduce.  Four DPP steps fold each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror: ~8 cycles each, no LDS), then the four row results are read as scalars and combined in the butterfly's own
// association ((r0 + r1) + (r2 + r3)), so the value is bit-identical to the __shfl_xor butterfly this replaces - whose six
// ds_bpermute round trips (~60-100 cycles each, dependent) were the critical path of every row kernel: LayerNorm backward
// spends four reductions per row.  Every lane of the wave must be active (as with the butterfly).
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_f32(float x, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov_f32<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_mov_f32<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_mov_f32<0x141>(v);      // row_half_mirror
  v += dpp_mov_f32<0x140>(v);      // row_mirror
  return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov_f32<0xB1>(v));
  v = fmaxf(v, dpp_mov_f32<0x4E>(v));
  v = fmaxf(v, dpp_mov_f32<0x141>(v));
  v = fmaxf(v, dpp_mov_f32<0x140>(v));
  return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}

__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
// The same erf-form GELU through erfc(y) = t*P4(t)*exp(-y^2), t = 1/(1 + 0.3275911 y) (Abramowitz & Stegun 7.1.26):
// |error| <= 4.3e-7 absolute over the whole line (checked in float32 against the double erf), i.e. far below one bf16
// ulp of the result, for ~14 VALU operations instead of libm erff's ~45.  Used where the result is stored as bf16:
// with K = 768 the FFN-in epilogue is otherwise VALU-bound (7-21 M erf per launch).
__device__ __forceinline__ float gelu_erf_short(float x) {
  const float y = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, y, 1.0f));   // v_rcp_f32 (1 ulp), not the 11-instruction IEEE division
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float q = p * t * __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // erfc(y)
  const float phi = x >= 0.f ? fmaf(-0.5f, q, 1.0f) : 0.5f * q;
  return x * phi;
}

// Two at a time on packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32: 13 packed + 8 scalar instructions per PAIR instead of 2 x 16);
// the same operations in the same order as gelu_erf_short, so the results are bit-identical to it.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf_short2(f32x2_t x) {
  const f32x2_t ax = {fabsf(x.x), fabsf(x.y)};
  const f32x2_t y = ax * 0.70710678118654752440f;
  const f32x2_t d = y * 0.3275911f + 1.0f;
  const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2_t p = t * 1.061405429f + (-1.453152027f);
  p = t * p + 1.421413741f;
  p = t * p + (-0.284496736f);
  p = t * p + 0.254829592f;
  const f32x2_t ea = x * x * -0.72134752044448170368f;
  const f32x2_t e = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
  const f32x2_t q = p * t * e;
  const f32x2_t pos = q * -0.5f + 1.0f, neg = q * 0.5f;
  const f32x2_t phi = {x.x >= 0.f ? pos.x : neg.x, x.y >= 0.f ? pos.y : neg.y};
  return x * phi;
}

// host side
void dsrg_set_error(const char* fmt, ...);
// device pointer registered with dsrg_set_dropout_offset() for the current device (NULL = none): a uint64 that every
// dropout kernel adds to its seed, so that a captured training step draws new masks on every replay
const unsigned long long* dsrg_dropout_offset_ptr();
// hipFuncAttributeMaxDynamicSharedMemorySize, set once per (call site, device) under std::call_once: the entry points are
// re-entrant (include/dsrg_hip.h), and the attribute belongs to each device's copy of the code object
#define DSRG_MAX_DEVICES 16
#define DSRG_SET_MAX_LDS_ONCE(fn_ptr, bytes)                                                                         \
  do {                                                                                                               \
    static std::once_flag once_[DSRG_MAX_DEVICES];                                                                   \
    int dev_ = 0;                                                                                                    \
    (void)hipGetDevice(&dev_);                                                                                       \
    auto set_ = [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn_ptr), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); }; \
    if (dev_ >= 0 && dev_ < DSRG_MAX_DEVICES) std::call_once(once_[dev_], set_);                                     \
    else set_();                                                                                                     \
  } while (0)
#define DSRG_CHECK(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      dsrg_set_error(__VA_ARGS__);            \
      return DSRG_EINVAL;                     \
    }                                         \
  } while (0)
#define DSRG_LAUNCH_CHECK(name)                                              \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      dsrg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return DSRG_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

// Stage the feature projection Wf [H,F] transposed into LDS (WT[k*HS + h]) with NT threads.  Eight loads are issued
// before the first LDS store: a plain `WT[..] = Wf[j]` loop with a run-time trip count compiles to load / wait / store
// per iteration, i.e. one memory latency per 4 bytes per thread (21 in a row for H*F = 5376 and 256 threads).
// Staged as 16-byte loads of the flat [H*F] array (H % 4 == 0, so H*F is a multiple of 4 and every piece is whole), all of a
// thread's pieces in flight at once; j / F is a multiply-high by ceil(2^32 / F) (exact for j < 2^28), not the ~35-instruction
// run-time division.  (Round 3 stamps: the staging's own cost is small next to the memory latency in front of it; the kernel
// that used it for every launch was LDS-read-bound in its projection loop - see embed_ln_regw_kernel in rowops.hip.)
template <int NT>
__device__ __forceinline__ void stage_wt(float* WT, const float* __restrict__ Wf, int H, int F, int HS, int tid) {
  const int n4 = (H * F) >> 2;
  const unsigned magic = F > 1 ? 0xFFFFFFFFu / (unsigned)F + 1u : 0u;
  const float4* __restrict__ W4 = reinterpret_cast<const float4*>(Wf);
  constexpr int UB = 12;       // pieces per thread and round: one round (one memory latency) for H*F <= 48 NT floats
  for (int j0 = tid; j0 < n4; j0 += NT * UB) {
    float4 v[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) v[u] = W4[min(j0 + u * NT, n4 - 1)];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int j4 = j0 + u * NT;
      if (j4 < n4) {
        const int j = j4 << 2;
        int h = F > 1 ? (int)__umulhi((unsigned)j, magic) : j, k = j - h * F;
        const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          WT[k * HS + h] = e[t];
          if (++k == F) { k = 0; ++h; }
        }
      }
    }
  }
}

// Counter-based dropout: the keep decision of element `idx` of a tensor is a hash (splitmix64 finaliser) of (seed, idx), so a
// backward kernel recomputes the mask and no mask tensor exists.  Returns inv_keep = 1/(1-p) for a kept element, 0 for a
// dropped one.  Used on the attention probabilities (attention.hip) and on the hidden states in front of a residual
// LayerNorm (rowops.hip / train.hip).
__device__ __forceinline__ float drop_keep(unsigned long long seed, size_t idx, float p, float inv_keep) {
  unsigned long long z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f) >= p ? inv_keep : 0.f;
}

// d act(z) / dz for DSRG_ACT_* (erf-GELU of FT/vilmodel.py:30-31, ReLU, tanh)
__device__ __forceinline__ float act_grad(float z, int act) {
  switch (act) {
    case DSRG_ACT_GELU: return 0.5f * (1.0f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * __expf(-0.5f * z * z);
    case DSRG_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case DSRG_ACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
    default: return 1.f;
  }
}
