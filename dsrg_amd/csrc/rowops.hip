// Row-wise kernels: LayerNorm, token-embedding sums, GAA pooling, head tails, logit fusion, IGL gate.
// One 64-lane wave owns one row; statistics by wave reduction in fp32.  HBM-bound (rows are read once,
// 8/16 bytes per lane), so the only rules that matter are coalescing and enough waves in flight.
#include <stdarg.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"

static thread_local char g_err[512] = "";
void dsrg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dsrg_last_error(void) { return g_err; }
extern "C" int dsrg_abi_version(void) { return DSRG_ABI_VERSION; }

// ---- dropout seed offset (include/dsrg_hip.h: dsrg_set_dropout_offset), one slot per device
#include <atomic>
static std::atomic<const unsigned long long*> g_drop_off[DSRG_MAX_DEVICES];
const unsigned long long* dsrg_dropout_offset_ptr() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DSRG_MAX_DEVICES) return nullptr;
  return g_drop_off[dev].load(std::memory_order_acquire);
}
extern "C" int dsrg_set_dropout_offset(const uint64_t* dev_ptr) {
  int dev = 0;
  DSRG_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < DSRG_MAX_DEVICES, "dsrg_set_dropout_offset: no current device");
  DSRG_CHECK(((uintptr_t)dev_ptr & 7) == 0, "dsrg_set_dropout_offset: pointer must be 8-byte aligned");
  g_drop_off[dev].store(reinterpret_cast<const unsigned long long*>(dev_ptr), std::memory_order_release);
  return DSRG_OK;
}

namespace {

constexpr int WPB = 4;        // waves (rows) per 256-thread block
constexpr int MAXV = 4;       // float4 per lane: H <= 64*4*4 = 1024

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// LayerNorm of a row held as NV float4 per lane (element index = (i*64+lane)*4 .. +3)
template <int NV>
__device__ __forceinline__ void ln_row(float4 (&v)[NV], int nv, int lane, int H, const float* gamma, const float* beta, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv && (i * 64 + lane) * 4 < H) s += v[i].x + v[i].y + v[i].z + v[i].w;
  const float mean = wave_sum(s) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv && (i * 64 + lane) * 4 < H) {
      float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss += a * a + b * b + c * c + d * d;
    }
  const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv && (i * 64 + lane) * 4 < H) {
      const int c = (i * 64 + lane) * 4;
      float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
      v[i] = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                         (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
    }
}

struct LnArgs {
  dsrg_ln_problem_t p[DSRG_MAX_GROUPS];
  int blk_start[DSRG_MAX_GROUPS + 1];     // first workgroup of each problem (WPB rows per workgroup)
  int nprob, H;
  float eps;
  // hidden-state dropout in front of the residual add (training): x is scaled by drop_keep(seed, (tag << 40) + row * H + col)
  float p_drop, inv_keep;
  unsigned long long seed;
  int tag[DSRG_MAX_GROUPS];
  const unsigned long long* seed_off;     // optional device word added to the seed (dsrg_set_dropout_offset)
};

// NV = float4 chunks per lane (ceil(H / 256)), a template parameter so that every load of the row - x, the residual,
// gamma, beta - is issued up front without a per-lane condition: behind `if (c < H)` the compiler waited for each load
// before issuing the next (six exposed latencies for a 768-wide row with residual).  Lanes past H load a clamped column.
template <typename TI, typename TO, int NV, bool DROP = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnArgs a) {
  const int lane = threadIdx.x & 63;
  // problem record in one batch of scalar loads at entry (problem 0 at fixed offsets, pinned into SGPRs; blk_start[] is
  // INT_MAX past the last problem) - see gemm.hip: the compiler otherwise chains three dependent load -> wait round trips
  dsrg_ln_problem_t P = a.p[0];
  asm volatile("" ::"s"(P.x), "s"(P.residual), "s"(P.gamma), "s"(P.beta), "s"(P.out), "s"(P.M), "s"(P.ldx), "s"(P.ldr), "s"(P.ldo), "s"(a.H), "s"(a.eps));
  asm volatile("" ::"s"(P.o_seg), "s"(P.o_seg_ld), "s"(P.aux0), "s"(P.aux1), "s"(P.aux_seg), "s"(P.aux_ld0), "s"(P.aux_ld1));
  int g = 0, bstart = 0;
#pragma unroll
  for (int i = 1; i < DSRG_MAX_GROUPS; ++i)
    if ((int)blockIdx.x >= a.blk_start[i]) { g = i; bstart = a.blk_start[i]; }
  if (g != 0) {
    P = a.p[g];
    asm volatile("" ::"s"(P.x), "s"(P.residual), "s"(P.gamma), "s"(P.beta), "s"(P.out), "s"(P.M), "s"(P.ldx), "s"(P.ldr), "s"(P.ldo));
    asm volatile("" ::"s"(P.o_seg), "s"(P.o_seg_ld), "s"(P.aux0), "s"(P.aux1), "s"(P.aux_seg), "s"(P.aux_ld0), "s"(P.aux_ld1));
  }
  const int row = ((int)blockIdx.x - bstart) * WPB + (threadIdx.x >> 6);
  if (row >= P.M) return;
  // output placement (dsrg_ln_problem_t): segmented rows, and a second / third copy of every aux_seg-th row
  size_t ooff = (size_t)row * P.ldo;
  if (P.o_seg > 0) { const int sg = row / P.o_seg; ooff = (size_t)sg * P.o_seg_ld + (size_t)(row - sg * P.o_seg) * P.ldo; }
  TO* __restrict__ ax0 = nullptr;
  TO* __restrict__ ax1 = nullptr;
  if (P.aux_seg > 0) {
    const int ai = row / P.aux_seg;
    if (row - ai * P.aux_seg == 0) {
      if (P.aux0) ax0 = reinterpret_cast<TO*>(P.aux0) + (size_t)ai * P.aux_ld0;
      if (P.aux1) ax1 = reinterpret_cast<TO*>(P.aux1) + (size_t)ai * P.aux_ld1;
    }
  }
  const TI* __restrict__ x = reinterpret_cast<const TI*>(P.x) + (size_t)row * P.ldx;
  const TO* __restrict__ res = reinterpret_cast<const TO*>(P.residual);
  TO* __restrict__ out = reinterpret_cast<TO*>(P.out);
  const int H = a.H;
  float4 v[NV], r[NV], gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = ld4(x + min((i * 64 + lane) * 4, H - 4));
  if (res) {
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = ld4(res + (size_t)row * P.ldr + min((i * 64 + lane) * 4, H - 4));
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = min((i * 64 + lane) * 4, H - 4);
    gm[i] = *reinterpret_cast<const float4*>(P.gamma + c);
    bt[i] = *reinterpret_cast<const float4*>(P.beta + c);
  }
  if constexpr (DROP) {       // dropout(x) before the residual add (BertSelfOutput / BertOutput, vilmodel.py:173-177, 213-217)
    const unsigned long long seed = a.seed + (a.seed_off ? *a.seed_off : 0ull);
    const size_t base = ((size_t)a.tag[g] << 40) + (size_t)row * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const size_t c = base + min((i * 64 + lane) * 4, H - 4);
      v[i].x *= drop_keep(seed, c, a.p_drop, a.inv_keep); v[i].y *= drop_keep(seed, c + 1, a.p_drop, a.inv_keep);
      v[i].z *= drop_keep(seed, c + 2, a.p_drop, a.inv_keep); v[i].w *= drop_keep(seed, c + 3, a.p_drop, a.inv_keep);
    }
  }
  if (res) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = f4add(v[i], r[i]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if ((i * 64 + lane) * 4 < H) s += v[i].x + v[i].y + v[i].z + v[i].w;
  const float mean = wave_sum(s) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if ((i * 64 + lane) * 4 < H) {
      const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
      ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    }
  const float rstd = rsqrtf(wave_sum(ss) / (float)H + a.eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const float4 o4 = make_float4((v[i].x - mean) * rstd * gm[i].x + bt[i].x, (v[i].y - mean) * rstd * gm[i].y + bt[i].y,
                                    (v[i].z - mean) * rstd * gm[i].z + bt[i].z, (v[i].w - mean) * rstd * gm[i].w + bt[i].w);
      st4(out + ooff + c, o4);
      if (ax0) st4(ax0 + c, o4);
      if (ax1) st4(ax1 + c, o4);
    }
  }
}

#ifdef DSRG_EMBED_STAMPS   // diagnostic build only (tools/micro/embed_time.py --stamps): per-workgroup s_memtime stamps
__device__ unsigned long long g_embed_stamps[4096 * 8];
#define EMBED_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_embed_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define EMBED_STAMP(i) do {} while (0)
#endif
struct EmbedIdx { const void* p; int kind, stride, div; };
struct EmbedArgs {
  const void* base; int ldb;
  const float* feat; int ldf; int F;
  const float* Wf; const float* bf; const float* lnf_g; const float* lnf_b;
  const float* E1; EmbedIdx i1; const float* E2; EmbedIdx i2;
  const float* vec; const float* lno_g; const float* lno_b;
  float eps; void* out; int ldo; int M; int H;
  // base row gathered from two tables by per-episode lengths (dsrg_embed_ln_fwd_seg); seg_T = 0: base row = row
  const void* base2; int ldb2; int seg_T, seg_V, seg_No; const int64_t* lens0; const int64_t* lens1;
  // optional LayerNorm on the base row before it is added (dsrg_embed_ln_fwd_lnb; rows without a base stay zero)
  const float* lnb_g; const float* lnb_b;
};

__device__ __forceinline__ int embed_index(const EmbedIdx& ix, int row) {
  const size_t o = (size_t)(ix.div == 1 ? row : row / ix.div) * ix.stride;      // (uniform branch: the division is ~35 instructions)
  if (ix.kind == DSRG_IDX_I32) return reinterpret_cast<const int32_t*>(ix.p)[o];
  if (ix.kind == DSRG_IDX_I64) return (int)reinterpret_cast<const int64_t*>(ix.p)[o];
  return (int)reinterpret_cast<const float*>(ix.p)[o];      // class id carried as a float feature column
}

// 8 waves per workgroup, one row per wave (16 left 104 of the 256 CUs idle at the nav-step's 1280-2432 rows, and the kernel is
// VALU-bound: four waves per SIMD on 152 CUs took 14.7 us for the 2432 viewpoint tokens).  The feature projection Wf [H,F] (F <= 16) is staged once per workgroup,
// transposed, in LDS (WT[k][h], rows padded by 4 floats) so that a lane's 4 output columns are one ds_read_b128 per k.
// The step is latency-bound (a few KB per row), so every independent global load of the row - features, indices,
// base row, both table rows - is issued before the staging barrier and the row's arithmetic starts with all of
// them in flight.
constexpr int EWAVES = 4;

// LayerNorm of a row held as NV float4 per lane with gamma / beta already in registers (same arithmetic, in the same order, as
// ln_row); ok[i]: this lane's chunk i lies inside the row.
template <int NV>
__device__ __forceinline__ void ln_row_pre(float4 (&v)[NV], const bool (&ok)[NV], int H, const float4 (&g)[NV], const float4 (&b)[NV], float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (ok[i]) s += v[i].x + v[i].y + v[i].z + v[i].w;
  const float mean = wave_sum(s) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (ok[i]) {
      float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss += a * a + bb * bb + c * c + d * d;
    }
  const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    v[i] = make_float4((v[i].x - mean) * rstd * g[i].x + b[i].x, (v[i].y - mean) * rstd * g[i].y + b[i].y,
                       (v[i].z - mean) * rstd * g[i].z + b[i].z, (v[i].w - mean) * rstd * g[i].w + b[i].w);
}

// NV = float4 chunks per lane (ceil(H / 256)), a template parameter (as in layernorm_kernel): EVERY global load of the row -
// features, indices, base row, table rows, and the per-column vectors (projection bias, both LayerNorms' gamma / beta) - is
// issued before the staging barrier, unconditionally per lane (columns clamped).  In-kernel stamps (round 3,
// profiles/r03s_embed_stamps.txt): with the bias load inside each chunk's `if (c < H)` and gamma / beta loaded inside the
// LayerNorms, a workgroup spent 8.3 k cycles in the projection and 1.9 k in the first LayerNorm waiting for five exposed
// memory latencies; the kernel took 13-15 us for 1280-2432 rows against 4 us for a plain LayerNorm of the same rows.
template <typename T, int NV>
__global__ __launch_bounds__(64 * EWAVES) void embed_ln_kernel(const EmbedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float WT[];
  EMBED_STAMP(0);
  // the whole launch record in one batch of scalar loads (the compiler otherwise sinks them into four dependent rounds)
  asm volatile("" ::"s"(a.base), "s"(a.ldb), "s"(a.feat), "s"(a.ldf), "s"(a.F), "s"(a.Wf), "s"(a.bf), "s"(a.lnf_g), "s"(a.lnf_b), "s"(a.E1),
               "s"(a.i1.p), "s"(a.i1.kind), "s"(a.i1.stride), "s"(a.i1.div), "s"(a.E2), "s"(a.i2.p), "s"(a.i2.kind), "s"(a.i2.stride),
               "s"(a.i2.div), "s"(a.vec), "s"(a.lno_g), "s"(a.lno_b), "s"(a.eps), "s"(a.out), "s"(a.ldo), "s"(a.M), "s"(a.H));
  asm volatile("" ::"s"(a.base2), "s"(a.ldb2), "s"(a.seg_T), "s"(a.seg_V), "s"(a.seg_No), "s"(a.lens0), "s"(a.lens1), "s"(a.lnb_g), "s"(a.lnb_b));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.H, HS = H + 4;
  const int row = blockIdx.x * EWAVES + wave;
  const bool live = row < a.M;
  const T* base = reinterpret_cast<const T*>(a.base);
  const int rowc = min(row, a.M - 1);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const T* brow = base ? base + (size_t)rowc * a.ldb : nullptr;
  if (a.seg_T > 0) {         // (uniform per wave) token t of episode b: a view row, an object row, or zero padding
    const int eb = rowc / a.seg_T, t = rowc - eb * a.seg_T;
    const int l0 = (int)a.lens0[eb], l1 = (int)a.lens1[eb];
    if (t < l0) brow = base + ((size_t)eb * a.seg_V + t) * a.ldb;
    else if (t < l0 + l1) brow = reinterpret_cast<const T*>(a.base2) + ((size_t)eb * a.seg_No + (t - l0)) * a.ldb2;
    else brow = nullptr;
  }
  int cc[NV];
  bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { cc[i] = min((i * 64 + lane) * 4, H - 4); ok[i] = (i * 64 + lane) * 4 < H; }
  // the row's features: ONE load (lane k holds feature k), broadcast below - F conditional loads were F latencies in a row
  const float fl = a.feat ? a.feat[(size_t)rowc * a.ldf + min(lane, a.F - 1)] : 0.f;
  const int i1 = a.E1 ? embed_index(a.i1, rowc) : 0, i2 = a.E2 ? embed_index(a.i2, rowc) : 0;
  typename Raw4<T>::type rb[NV];
  float4 l1[NV], l2[NV], lv[NV], bfv[NV], g1[NV], b1[NV], g2[NV], b2[NV], g0[NV], b0[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) l1[i] = l2[i] = lv[i] = bfv[i] = g1[i] = b1[i] = g2[i] = b2[i] = g0[i] = b0[i] = z4;
  // (every condition below is uniform: a scalar branch around a batch of loads, nothing waits inside it)
  if (brow) {
#pragma unroll
    for (int i = 0; i < NV; ++i) rb[i] = raw_ld4(brow + cc[i]);
  }
  if (a.lnb_g) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { g0[i] = *reinterpret_cast<const float4*>(a.lnb_g + cc[i]); b0[i] = *reinterpret_cast<const float4*>(a.lnb_b + cc[i]); }
  }
  if (a.E1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) l1[i] = *reinterpret_cast<const float4*>(a.E1 + (size_t)i1 * H + cc[i]);
  }
  if (a.E2) {
#pragma unroll
    for (int i = 0; i < NV; ++i) l2[i] = *reinterpret_cast<const float4*>(a.E2 + (size_t)i2 * H + cc[i]);
  }
  if (a.vec) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lv[i] = *reinterpret_cast<const float4*>(a.vec + cc[i]);
  }
  if (a.feat && a.bf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) bfv[i] = *reinterpret_cast<const float4*>(a.bf + cc[i]);
  }
  if (a.feat && a.lnf_g) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { g1[i] = *reinterpret_cast<const float4*>(a.lnf_g + cc[i]); b1[i] = *reinterpret_cast<const float4*>(a.lnf_b + cc[i]); }
  }
  if (a.lno_g) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { g2[i] = *reinterpret_cast<const float4*>(a.lno_g + cc[i]); b2[i] = *reinterpret_cast<const float4*>(a.lno_b + cc[i]); }
  }
  EMBED_STAMP(1);
  if (a.feat) {
    stage_wt<64 * EWAVES>(WT, a.Wf, H, a.F, HS, threadIdx.x);
    __syncthreads();
  }
  EMBED_STAMP(2);
  if (!live) return;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = z4;
  if (a.feat) {
    float f[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) f[k] = k < a.F ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fl), k)) : 0.f;
    // Branch-free over 8 or 16 features (rows of WT past F are read clamped and meet f[k] = 0): the LDS reads of a chunk go
    // out together instead of one round trip per k
    auto project = [&](auto kn_tag) {
      constexpr int KN = decltype(kn_tag)::value;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float4 o = bfv[i];
        float4 w[KN];
#pragma unroll
        for (int k = 0; k < KN; ++k) w[k] = *reinterpret_cast<const float4*>(&WT[min(k, a.F - 1) * HS + cc[i]]);
#pragma unroll
        for (int k = 0; k < KN; ++k) {
          o.x = fmaf(w[k].x, f[k], o.x); o.y = fmaf(w[k].y, f[k], o.y); o.z = fmaf(w[k].z, f[k], o.z); o.w = fmaf(w[k].w, f[k], o.w);
        }
        v[i] = o;
      }
    };
    if (a.F == 7) project(std::integral_constant<int, 7>{});            // (the reference's position features: 7 and 14 wide)
    else if (a.F <= 8) project(std::integral_constant<int, 8>{});
    else if (a.F == 14) project(std::integral_constant<int, 14>{});
    else project(std::integral_constant<int, 16>{});
    EMBED_STAMP(3);
    if (a.lnf_g) ln_row_pre<NV>(v, ok, H, g1, b1, a.eps);
  }
  EMBED_STAMP(4);
  float4 lb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) lb[i] = brow ? raw4(rb[i]) : z4;
  if (a.lnb_g && brow) ln_row_pre<NV>(lb, ok, H, g0, b0, a.eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = f4add(f4add(f4add(lb[i], l1[i]), f4add(l2[i], lv[i])), v[i]);
  if (a.lno_g) ln_row_pre<NV>(v, ok, H, g2, b2, a.eps);
  EMBED_STAMP(5);
  T* out = reinterpret_cast<T*>(a.out);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (ok[i]) st4(out + (size_t)row * a.ldo + cc[i], v[i]);
  EMBED_STAMP(6);
}

// Register-resident projection (the reference's position features: F = 7 or 14): the kernel above re-reads the whole staged
// W^T from LDS for every row - 43 KB per viewpoint token, 9.5 rows per CU - and is LDS-bandwidth-bound there (stamps: 8.4 k of a
// workgroup's 17 k cycles; the rest is one memory latency for the row and one for the staging).  Here a lane keeps the weights
// of ITS columns in registers: columns c..c+3 of chunk i are rows c..c+3 of Wf [H,F], i.e. 4F contiguous floats = F float4 as
// they lie (no transposition, no LDS, no barrier), read once per wave and used for ER rows.  Every global load of the wave -
// weights, per-column vectors, the ER rows' base / feature / table rows - is issued up front.
// Element (hh, k) of a lane's chunk is float hh*F + k of its F float4.
constexpr int ER = 3;          // rows per wave: 2432 viewpoint tokens = 203 workgroups of 4 waves (one round on 256 CUs)
template <typename T, int NV, int F>
__global__ __launch_bounds__(256) void embed_ln_regw_kernel(const EmbedArgs a) {
  asm volatile("" ::"s"(a.base), "s"(a.ldb), "s"(a.feat), "s"(a.ldf), "s"(a.F), "s"(a.Wf), "s"(a.bf), "s"(a.lnf_g), "s"(a.lnf_b), "s"(a.E1),
               "s"(a.i1.p), "s"(a.i1.kind), "s"(a.i1.stride), "s"(a.i1.div), "s"(a.E2), "s"(a.i2.p), "s"(a.i2.kind), "s"(a.i2.stride),
               "s"(a.i2.div), "s"(a.vec), "s"(a.lno_g), "s"(a.lno_b), "s"(a.eps), "s"(a.out), "s"(a.ldo), "s"(a.M), "s"(a.H));
  asm volatile("" ::"s"(a.base2), "s"(a.ldb2), "s"(a.seg_T), "s"(a.seg_V), "s"(a.seg_No), "s"(a.lens0), "s"(a.lens1), "s"(a.lnb_g), "s"(a.lnb_b));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.H;
  const int row0 = (blockIdx.x * 4 + wave) * ER;
  if (row0 >= a.M) return;                                   // wave-uniform; no barrier in this kernel
  const T* base = reinterpret_cast<const T*>(a.base);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int cc[NV];
  bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { cc[i] = min((i * 64 + lane) * 4, H - 4); ok[i] = (i * 64 + lane) * 4 < H; }
  // ---- per row: indices and features first (the table rows depend on the indices)
  int rowc[ER], i1[ER], i2[ER];
  float fl[ER];
  const T* brow[ER];
#pragma unroll
  for (int r = 0; r < ER; ++r) {
    rowc[r] = min(row0 + r, a.M - 1);
    fl[r] = a.feat[(size_t)rowc[r] * a.ldf + min(lane, F - 1)];
    i1[r] = a.E1 ? embed_index(a.i1, rowc[r]) : 0;
    i2[r] = a.E2 ? embed_index(a.i2, rowc[r]) : 0;
    brow[r] = base ? base + (size_t)rowc[r] * a.ldb : nullptr;
    if (a.seg_T > 0) {
      const int eb = rowc[r] / a.seg_T, t = rowc[r] - eb * a.seg_T;
      const int l0 = (int)a.lens0[eb], l1 = (int)a.lens1[eb];
      if (t < l0) brow[r] = base + ((size_t)eb * a.seg_V + t) * a.ldb;
      else if (t < l0 + l1) brow[r] = reinterpret_cast<const T*>(a.base2) + ((size_t)eb * a.seg_No + (t - l0)) * a.ldb2;
      else brow[r] = nullptr;
    }
  }
  // ---- weights of this lane's columns, per-column vectors
  float4 w[NV][F];
  const float4* __restrict__ W4 = reinterpret_cast<const float4*>(a.Wf);
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int k = 0; k < F; ++k) w[i][k] = W4[(size_t)(cc[i] >> 2) * F + k];
  float4 lv[NV], bfv[NV], g1[NV], b1[NV], g2[NV], b2[NV], g0[NV], b0[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) lv[i] = bfv[i] = g1[i] = b1[i] = g2[i] = b2[i] = g0[i] = b0[i] = z4;
  if (a.lnb_g) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { g0[i] = *reinterpret_cast<const float4*>(a.lnb_g + cc[i]); b0[i] = *reinterpret_cast<const float4*>(a.lnb_b + cc[i]); }
  }
  if (a.vec) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lv[i] = *reinterpret_cast<const float4*>(a.vec + cc[i]);
  }
  if (a.bf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) bfv[i] = *reinterpret_cast<const float4*>(a.bf + cc[i]);
  }
  if (a.lnf_g) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { g1[i] = *reinterpret_cast<const float4*>(a.lnf_g + cc[i]); b1[i] = *reinterpret_cast<const float4*>(a.lnf_b + cc[i]); }
  }
  if (a.lno_g) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { g2[i] = *reinterpret_cast<const float4*>(a.lno_g + cc[i]); b2[i] = *reinterpret_cast<const float4*>(a.lno_b + cc[i]); }
  }
  // ---- the rows' base and table rows
  typename Raw4<T>::type rb[ER][NV];
  float4 l1[ER][NV], l2[ER][NV];
#pragma unroll
  for (int r = 0; r < ER; ++r) {
#pragma unroll
    for (int i = 0; i < NV; ++i) l1[r][i] = l2[r][i] = z4;
    if (brow[r]) {
#pragma unroll
      for (int i = 0; i < NV; ++i) rb[r][i] = raw_ld4(brow[r] + cc[i]);
    }
    if (a.E1) {
#pragma unroll
      for (int i = 0; i < NV; ++i) l1[r][i] = *reinterpret_cast<const float4*>(a.E1 + (size_t)i1[r] * H + cc[i]);
    }
    if (a.E2) {
#pragma unroll
      for (int i = 0; i < NV; ++i) l2[r][i] = *reinterpret_cast<const float4*>(a.E2 + (size_t)i2[r] * H + cc[i]);
    }
  }
  T* out = reinterpret_cast<T*>(a.out);
#pragma unroll
  for (int r = 0; r < ER; ++r) {
    if (row0 + r >= a.M) break;                              // wave-uniform
    float f[F];
#pragma unroll
    for (int k = 0; k < F; ++k) f[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fl[r]), k));
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float o[4] = {bfv[i].x, bfv[i].y, bfv[i].z, bfv[i].w};
      const float* wf = reinterpret_cast<const float*>(&w[i][0]);      // (compile-time indices: stays in registers)
#pragma unroll
      for (int k = 0; k < F; ++k)                            // k outermost: the same fma chain per column as the staged kernel
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) o[hh] = fmaf(wf[hh * F + k], f[k], o[hh]);
      v[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (a.lnf_g) ln_row_pre<NV>(v, ok, H, g1, b1, a.eps);
    float4 lb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) lb[i] = brow[r] ? raw4(rb[r][i]) : z4;
    if (a.lnb_g && brow[r]) ln_row_pre<NV>(lb, ok, H, g0, b0, a.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = f4add(f4add(f4add(lb[i], l1[r][i]), f4add(l2[r][i], lv[i])), v[i]);
    if (a.lno_g) ln_row_pre<NV>(v, ok, H, g2, b2, a.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ok[i]) st4(out + (size_t)(row0 + r) * a.ldo + cc[i], v[i]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void group_mean_kernel(const T* __restrict__ x, int ldx, T* __restrict__ out, int ldo, int M, int G, int H) {
  const int idx = blockIdx.x * 256 + threadIdx.x;    // one thread per (row, 4 columns)
  const int hv = H / 4;
  if (idx >= M * hv) return;
  const int m = idx / hv, c = (idx % hv) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = 0; g < G; ++g) s = f4add(s, ld4(x + (size_t)(m * G + g) * ldx + c));
  const float inv = 1.0f / (float)G;
  st4(out + (size_t)m * ldo + c, make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv));
}

// One 1024-thread workgroup per episode, every view row read from HBM exactly once: wave w loads views w, w+16, ..
// (all loads issued before anything depends on them), scores them, and after the softmax over the LDS score vector
// forms the weighted sum of ITS OWN rows from registers; the 16 partial sums are combined through LDS.
constexpr int GWAVES = 16, GVPW = 4;      // V <= 64
template <typename T>
__global__ __launch_bounds__(1024) void gaa_pool_kernel(const T* __restrict__ x, long x_bs, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ out, T* __restrict__ out2, long out2_bs, const int64_t* __restrict__ idx2,
                                                        int idx2_max, int V, int H, const float* __restrict__ ln_g = nullptr,
                                                        const float* __restrict__ ln_b = nullptr, float ln_eps = 0.f,
                                                        T* __restrict__ rows_out = nullptr, long rows_bs = 0) {
  __shared__ float sc[64];
  extern __shared__ __attribute__((aligned(16))) float part[];     // [GWAVES][H]
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = (H + 255) / 256;
  const T* xb = x + (size_t)b * x_bs;
  // every load of the wave goes out first, unconditionally (view / column clamped), and is masked afterwards: a
  // `cond ? load : 0` consumes each load at once and the compiler then waits for it before issuing the next
  // (16 memory latencies in a row in the ISA of the first version)
  float4 row[GVPW][MAXV], ww[MAXV];
  typename Raw4<T>::type raw[GVPW][MAXV];          // unconverted: the bf16 -> f32 shifts would consume each load at once
#pragma unroll
  for (int r = 0; r < GVPW; ++r)
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      raw[r][i] = raw_ld4(xb + (size_t)min(wave + GWAVES * r, V - 1) * H + min((i * 64 + lane) * 4, H - 4));
#pragma unroll
  for (int i = 0; i < MAXV; ++i) ww[i] = *reinterpret_cast<const float4*>(w + min((i * 64 + lane) * 4, H - 4));
  const float b0 = bias[0];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const bool okc = i < nv && (i * 64 + lane) * 4 < H;
    if (!okc) ww[i] = z4;
#pragma unroll
    for (int r = 0; r < GVPW; ++r) row[r][i] = (okc && wave + GWAVES * r < V) ? raw4(raw[r][i]) : z4;
  }
  // dsrg_gaa_pool_fwd_ln: the rows come un-normalised (the vis_obj_linear GEMM's output as stored); every wave normalises ITS
  // rows (the whole row is in the wave), stores them to rows_out in the storage dtype and pools the stored values - what a
  // LayerNorm launch in front of this kernel would have produced, without the launch
  if (ln_g) {
    float4 lg[MAXV], lb[MAXV];
    bool ok[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int cc = min((i * 64 + lane) * 4, H - 4);
      ok[i] = i < nv && (i * 64 + lane) * 4 < H;
      lg[i] = *reinterpret_cast<const float4*>(ln_g + cc);
      lb[i] = *reinterpret_cast<const float4*>(ln_b + cc);
    }
#pragma unroll
    for (int r = 0; r < GVPW; ++r) {
      const int v = wave + GWAVES * r;
      if (v < V) {                                             // wave-uniform
        ln_row_pre<MAXV>(row[r], ok, H, lg, lb, ln_eps);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
          if (ok[i]) {
            T* dst = rows_out + (size_t)b * rows_bs + (size_t)v * H + (i * 64 + lane) * 4;
            st4(dst, row[r][i]);
            if constexpr (sizeof(T) == 2)                           // pool what was stored (rounded to bf16), as the two-launch path does
              row[r][i] = make_float4(bf2f(f2bf(row[r][i].x)), bf2f(f2bf(row[r][i].y)), bf2f(f2bf(row[r][i].z)), bf2f(f2bf(row[r][i].w)));
          } else row[r][i] = z4;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < GVPW; ++r) {
    const int v = wave + GWAVES * r;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += row[r][i].x * ww[i].x + row[r][i].y * ww[i].y + row[r][i].z * ww[i].z + row[r][i].w * ww[i].w;
    s = wave_sum(s);
    if (lane == 0 && v < V) sc[v] = tanhf(s + b0);
  }
  __syncthreads();
  // softmax statistics over the V <= 64 scores: lane <-> view, one expf per lane and two wave reductions (every thread
  // looping over all V scores cost 2 V dependent LDS-read + expf steps, 40 % of the kernel)
  const float mine = lane < V ? sc[lane] : -INFINITY;
  float mx = mine;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  const float den = wave_sum(lane < V ? expf(mine - mx) : 0.f);
  const float inv = 1.0f / den;
  float4 acc[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int r = 0; r < GVPW; ++r) {
    const int v = wave + GWAVES * r;
    const float p = v < V ? expf(sc[v] - mx) * inv : 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      acc[i].x += row[r][i].x * p; acc[i].y += row[r][i].y * p; acc[i].z += row[r][i].z * p; acc[i].w += row[r][i].w * p;
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < H) *reinterpret_cast<float4*>(&part[wave * H + c]) = acc[i];
  }
  __syncthreads();
  const int c = threadIdx.x * 4;
  if (c < H) {
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < GWAVES; ++g) sum = f4add(sum, *reinterpret_cast<const float4*>(&part[g * H + c]));
    st4(out + (size_t)b * H + c, sum);
    if (out2) st4(out2 + (size_t)b * out2_bs + (size_t)min(max((int)idx2[b], 0), idx2_max - 1) * H + c, sum);   // (index clamped: never outside the batch's rows)     // the current node's slot of the graph-map token buffer
  }
}

__global__ __launch_bounds__(256) void head_tail_kernel(const float* __restrict__ h, int ldh, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, float eps, float* __restrict__ score, int M, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = (H + 255) / 256;
  float4 v[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < H) v[i] = ld4(h + (size_t)row * ldh + c);
  }
  ln_row<MAXV>(v, nv, lane, H, gamma, beta, eps);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < H) {
      float4 ww = *reinterpret_cast<const float4*>(w2 + c);
      s += v[i].x * ww.x + v[i].y * ww.y + v[i].z * ww.z + v[i].w * ww.w;
    }
  }
  s = wave_sum(s);
  if (lane == 0) score[row] = s + b2[0];
}

struct HeadTailGroup {
  dsrg_head_tail_problem_t p[DSRG_MAX_GROUPS];
  int blk_start[DSRG_MAX_GROUPS + 1];
  int n, H;
  float eps;
};
__global__ __launch_bounds__(256) void head_tail_grouped_kernel(const HeadTailGroup a) {
  const int lane = threadIdx.x & 63;
  int g = 0;
#pragma unroll
  for (int i = 1; i < DSRG_MAX_GROUPS; ++i)
    if (i < a.n && (int)blockIdx.x >= a.blk_start[i]) g = i;
  const dsrg_head_tail_problem_t& P = a.p[g];
  const int row = ((int)blockIdx.x - a.blk_start[g]) * WPB + (threadIdx.x >> 6);
  if (row >= P.M) return;
  const int H = a.H, nv = (H + 255) / 256;
  float4 v[MAXV], ww[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = min((i * 64 + lane) * 4, H - 4);
    v[i] = *reinterpret_cast<const float4*>(P.h + (size_t)row * P.ldh + c);
    ww[i] = *reinterpret_cast<const float4*>(P.w2 + c);
  }
  ln_row<MAXV>(v, nv, lane, H, P.gamma, P.beta, a.eps);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (i < nv && (i * 64 + lane) * 4 < H) s += v[i].x * ww[i].x + v[i].y * ww[i].y + v[i].z * ww[i].z + v[i].w * ww[i].w;
  s = wave_sum(s);
  if (lane == 0) P.score[row] = s + P.b2[0];
}

// one wave per episode; local logits are staged in LDS for the gather by graph node
constexpr int NAV_MAX_VP = 1024;
__global__ __launch_bounds__(64) void nav_logits_kernel(const float* __restrict__ g_raw, const float* __restrict__ l_raw,
                                                        const float* __restrict__ fuse_raw, const uint8_t* __restrict__ gmask,
                                                        const uint8_t* __restrict__ gvis, const uint8_t* __restrict__ vnav,
                                                        const int32_t* __restrict__ gsrc, const uint8_t* __restrict__ bwm,
                                                        float* __restrict__ gl, float* __restrict__ ll, float* __restrict__ fl,
                                                        const float* __restrict__ o_raw, const uint8_t* __restrict__ vobj,
                                                        float* __restrict__ ol, int G, int Vp) {
  __shared__ float sl[NAV_MAX_VP];
  __shared__ uint8_t sb[NAV_MAX_VP];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (ol) {          // REVERIE object grounding: masked_fill(~vp_obj_masks, -inf)
    for (int k = lane; k < Vp; k += 64) ol[(size_t)b * Vp + k] = vobj[(size_t)b * Vp + k] ? o_raw[(size_t)b * Vp + k] : -INFINITY;
  }
  const float fw = fuse_raw ? 1.0f / (1.0f + expf(-fuse_raw[b])) : 0.5f;
  for (int k = lane; k < Vp; k += 64) {
    float v = l_raw[(size_t)b * Vp + k] * (1.0f - fw);
    if (!vnav[(size_t)b * Vp + k]) v = -INFINITY;
    ll[(size_t)b * Vp + k] = v;
    sl[k] = v;
    sb[k] = bwm[(size_t)b * Vp + k];
  }
  __syncthreads();
  float bw = 0.f;    // sequential, in candidate order, like the reference's running sum
  for (int k = 0; k < Vp; ++k)
    if (sb[k]) bw += sl[k];
  const float l0 = sl[0];
  for (int j = lane; j < G; j += 64) {
    float v = g_raw[(size_t)b * G + j] * fw;
    if (gvis[(size_t)b * G + j] || !gmask[(size_t)b * G + j]) v = -INFINITY;
    gl[(size_t)b * G + j] = v;
    const int s = gsrc[(size_t)b * G + j];
    float f = v;
    if (j == 0) f += l0;
    else if (s >= 0 && s < Vp) f += sl[s];
    else if (s == -2) f += bw;
    fl[(size_t)b * G + j] = f;
  }
}

// ---- action selection + loss tail (agent.py:484-516): one wave per episode row of nav logits ([B,N], -inf = masked).
//   probs = softmax(row); stop_prob = probs[0]; argmax (first maximum, like torch.max); entropy = -sum p log p;
//   sample = inverse-CDF draw with the caller's uniform u[b]; CE row loss = lse - logit[target] (0 when target == ignore);
//   dlogits = probs - onehot(target) (0 row when ignored): the gradient of the summed cross-entropy.
__global__ __launch_bounds__(64) void action_tail_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ targets,
                                                         int64_t ignore, const float* __restrict__ u, float* __restrict__ stop_prob,
                                                         int32_t* __restrict__ amax, int32_t* __restrict__ sample,
                                                         float* __restrict__ entropy, float* __restrict__ loss_rows,
                                                         float* __restrict__ dlogits, int N) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* row = logits + (size_t)b * ld;
  float mx = -INFINITY;
  int ai = 0x7fffffff;
  for (int j = lane; j < N; j += 64) {
    const float v = row[j];
    if (v > mx) { mx = v; ai = j; }              // strided scan keeps the first index of this lane's maximum
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o);
    const int oi = __shfl_xor(ai, o);
    if (om > mx || (om == mx && oi < ai)) { mx = om; ai = oi; }
  }
  float se = 0.f;
  for (int j = lane; j < N; j += 64) se += __expf(row[j] - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se), inv = 1.0f / se;
  float ent = 0.f;
  for (int j = lane; j < N; j += 64) {
    const float v = row[j];
    if (v > -INFINITY) { const float p = __expf(v - mx) * inv; ent -= p * (v - lse); }
  }
  ent = wave_sum(ent);
  const int64_t t = targets ? targets[b] : ignore;
  const bool live = targets && t != ignore && t >= 0 && t < N;
  if (dlogits)
    for (int j = lane; j < N; j += 64)
      dlogits[(size_t)b * N + j] = live ? __expf(row[j] - mx) * inv - (j == (int)t ? 1.f : 0.f) : 0.f;
  if (sample) {
    // sequential inverse CDF over 64-wide chunks: pick the first j with cdf(j) > u (the last unmasked one if rounding
    // leaves cdf(N-1) <= u)
    const float uu = u[b];
    float base = 0.f;
    int pick = -1, lastpos = -1;
    for (int j0 = 0; j0 < N && pick < 0; j0 += 64) {
      const int j = j0 + lane;
      const float p = j < N ? __expf(row[j] - mx) * inv : 0.f;
      float c = p;                                  // inclusive scan across the wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const float y = __shfl_up(c, o); if (lane >= o) c += y; }
      c += base;
      const unsigned long long hit = __ballot(p > 0.f && c > uu), pos = __ballot(p > 0.f);
      if (pos) lastpos = j0 + 63 - __builtin_clzll(pos);
      if (hit) pick = j0 + __builtin_ctzll(hit);
      base = __shfl(c, 63);
    }
    if (lane == 0) sample[b] = pick >= 0 ? pick : lastpos;
  }
  if (lane == 0) {
    stop_prob[b] = __expf(row[0] - mx) * inv;
    amax[b] = ai;
    if (entropy) entropy[b] = ent;
    if (loss_rows) loss_rows[b] = live ? lse - row[t] : 0.f;
  }
}

// ---- cross-entropy rows (the pre-training loss tails: F.cross_entropy(.., reduction='none') over the 30,522-word MLM
// scores and the SAP logits, pretrain_cmt.py:160-167, 286-294).  One 256-thread workgroup per row.
//   forward : ONE pass over the row with an online (max, sum exp) per thread, combined through LDS:
//             lse_b, loss_b = lse_b - logit[b, target_b] (0 when the target is ignore_index / out of range);
//   backward: dlogits[b, j] = g_b * (exp(logit[b, j] - lse_b) - [j == target_b])  (0 row when ignored), recomputed from the
//             logits and lse: no log-softmax matrix is ever stored.
__device__ __forceinline__ void ms_combine(float& m, float& s, float om, float os) {
  const float nm = fmaxf(m, om);
  if (nm == -INFINITY) { s = 0.f; return; }
  s = s * __expf(m - nm) + os * __expf(om - nm);
  m = nm;
}
__global__ __launch_bounds__(256) void ce_rows_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ targets,
                                                          int64_t ignore, float* __restrict__ lse_rows, float* __restrict__ loss_rows, int N) {
  __shared__ float sm[4], ss[4];
  __shared__ int snan[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (size_t)b * ld;
  float m = -INFINITY, s = 0.f;
  bool bad = false;           // a NaN logit fails both comparisons below: it is carried separately and poisons the row's lse,
  for (int j = tid; j < N; j += 256) {      // as F.cross_entropy's log_softmax does (a diverged step then shows in the logged loss)
    const float v = row[j];
    bad = bad || (v != v);
    if (v > m) { s = s * __expf(m - v) + 1.f; m = v; }      // exp(-inf - v) = 0 on the first finite value
    else if (v > -INFINITY) s += __expf(v - m);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ms_combine(m, s, __shfl_xor(m, o), __shfl_xor(s, o));
  const bool wbad = __any((int)bad) != 0;
  if (lane == 0) { sm[wave] = m; ss[wave] = s; snan[wave] = wbad ? 1 : 0; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) ms_combine(m, s, sm[w], ss[w]);     // fixed order
    const bool anybad = (snan[0] | snan[1] | snan[2] | snan[3]) != 0;
    const float lse = anybad ? __builtin_nanf("") : m + logf(s);
    const int64_t t = targets[b];
    const bool live = t != ignore && t >= 0 && t < N;
    lse_rows[b] = lse;
    loss_rows[b] = live ? lse - row[t] : 0.f;
  }
}
template <typename OutT>
__global__ __launch_bounds__(256) void ce_rows_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ targets,
                                                          int64_t ignore, const float* __restrict__ lse_rows, const float* __restrict__ g,
                                                          OutT* __restrict__ dlogits, int ldd, int N) {
  const int b = blockIdx.x;
  const float* row = logits + (size_t)b * ld;
  OutT* out = dlogits + (size_t)b * ldd;
  const int64_t t = targets[b];
  const bool live = t != ignore && t >= 0 && t < N;
  const float lse = lse_rows[b], gb = live ? g[b] : 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    const float p = live ? __expf(row[j] - lse) - (j == (int)t ? 1.f : 0.f) : 0.f;
    Elem<OutT>::st(out + j, gb * p);
  }
  for (int j = N + threadIdx.x; j < ldd; j += 256) Elem<OutT>::st(out + j, 0.f);     // padding columns of a wider buffer
}

// ---- KL rows (the MRC loss tail, pretrain_cmt.py:215-226: F.kl_div(F.log_softmax(logits, -1), target, reduction='none').sum(1)).
// One wave per row (4 rows per workgroup); N = 1000 / 1601 classes.
//   forward : lse_r; loss_r = sum_c [t > 0] t * (log t - (x - lse_r))             (xlogy(t, t) - t * log_softmax(x))
//   backward: dx[r, c] = g_r * (exp(x - lse_r) * sum_c t - t)
__global__ __launch_bounds__(256) void kl_rows_fwd_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ target, int ldt,
                                                          float* __restrict__ lse_rows, float* __restrict__ loss_rows, int R, int N) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* row = logits + (size_t)r * ld;
  const float* tr = target + (size_t)r * ldt;
  float m = -INFINITY;
  for (int j = lane; j < N; j += 64) m = fmaxf(m, row[j]);
  m = wave_max(m);
  float s = 0.f, a = 0.f, st = 0.f;            // sum exp, sum t * (log t - x), sum t
  for (int j = lane; j < N; j += 64) {
    const float x = row[j], t = tr[j];
    s += __expf(x - m);
    if (t > 0.f) a += t * (logf(t) - x);
    st += t;
  }
  s = wave_sum(s); a = wave_sum(a); st = wave_sum(st);
  const float lse = m + logf(s);
  if (lane == 0) { lse_rows[r] = lse; loss_rows[r] = a + st * lse; }
}
__global__ __launch_bounds__(256) void kl_rows_bwd_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ target, int ldt,
                                                          const float* __restrict__ lse_rows, const float* __restrict__ g, float* __restrict__ dlogits,
                                                          int ldd, int R, int N) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* row = logits + (size_t)r * ld;
  const float* tr = target + (size_t)r * ldt;
  float st = 0.f;
  for (int j = lane; j < N; j += 64) st += tr[j];
  st = wave_sum(st);
  const float lse = lse_rows[r], gr = g[r];
  for (int j = lane; j < N; j += 64) dlogits[(size_t)r * ldd + j] = gr * (__expf(row[j] - lse) * st - tr[j]);
  for (int j = N + lane; j < ldd; j += 64) dlogits[(size_t)r * ldd + j] = 0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void igl_blend_kernel(const T* __restrict__ aug, const T* __restrict__ x, const float* __restrict__ wa,
                                                        const float* __restrict__ ba, const float* __restrict__ wo, const float* __restrict__ bo,
                                                        T* __restrict__ out, int M, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = (H + 255) / 256;
  float4 va[MAXV], vx[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < H) {
      va[i] = ld4(aug + (size_t)row * H + c);
      vx[i] = ld4(x + (size_t)row * H + c);
      float4 a = *reinterpret_cast<const float4*>(wa + c), o = *reinterpret_cast<const float4*>(wo + c);
      s += va[i].x * a.x + va[i].y * a.y + va[i].z * a.z + va[i].w * a.w;
      s += vx[i].x * o.x + vx[i].y * o.y + vx[i].z * o.z + vx[i].w * o.w;
    }
  }
  s = wave_sum(s) + ba[0] + bo[0];
  const float w = 1.0f / (1.0f + expf(-s));
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < H)
      st4(out + (size_t)row * H + c, make_float4(w * va[i].x + (1.f - w) * vx[i].x, w * va[i].y + (1.f - w) * vx[i].y,
                                                   w * va[i].z + (1.f - w) * vx[i].z, w * va[i].w + (1.f - w) * vx[i].w));
  }
}

inline bool ok_dtype(int d) { return d == DSRG_F32 || d == DSRG_BF16; }
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

static int layernorm_fwd_impl(const dsrg_ln_problem_t* problems, int nproblems, int H, float eps, int in_dtype,
                              int out_dtype, float p_drop, unsigned long long seed, void* stream) {
  DSRG_CHECK(p_drop >= 0.f && p_drop < 1.f, "dsrg_layernorm_fwd: p_drop=%g out of [0,1)", (double)p_drop);
  DSRG_CHECK(problems && nproblems >= 1 && nproblems <= DSRG_MAX_GROUPS, "dsrg_layernorm_fwd: nproblems=%d out of [1,%d]", nproblems,
             DSRG_MAX_GROUPS);
  DSRG_CHECK(ok_dtype(in_dtype) && ok_dtype(out_dtype), "dsrg_layernorm_fwd: bad dtype");
  DSRG_CHECK(in_dtype == DSRG_F32 || in_dtype == out_dtype, "dsrg_layernorm_fwd: input must be f32 or the output dtype");
  DSRG_CHECK(H > 0 && H <= 1024 && H % 64 == 0, "dsrg_layernorm_fwd: H=%d must be a multiple of 64 and <= 1024", H);
  LnArgs a;
  a.nprob = 0; a.H = H; a.eps = eps;
  a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
  a.seed_off = p_drop > 0.f ? dsrg_dropout_offset_ptr() : nullptr;
  int blocks = 0;
  for (int i = 0; i < nproblems; ++i) {
    const dsrg_ln_problem_t& p = problems[i];
    DSRG_CHECK(p.M >= 0, "dsrg_layernorm_fwd: problem %d has M < 0", i);
    if (p.M == 0) continue;
    DSRG_CHECK(p.x && p.gamma && p.beta && p.out, "dsrg_layernorm_fwd: problem %d has a null pointer", i);
    DSRG_CHECK(p.ldx % 4 == 0 && p.ldo % 4 == 0 && (!p.residual || p.ldr % 4 == 0) && p.ldx >= H && p.ldo >= H && (!p.residual || p.ldr >= H),
               "dsrg_layernorm_fwd: problem %d leading dims must be >= H and multiples of 4", i);
    DSRG_CHECK(al16(p.gamma) && al16(p.beta) && ((uintptr_t)p.x & 7) == 0 && ((uintptr_t)p.out & 7) == 0 && ((uintptr_t)p.residual & 7) == 0 &&
                   (in_dtype == DSRG_BF16 || al16(p.x)) && (out_dtype == DSRG_BF16 || (al16(p.out) && al16(p.residual))),
               "dsrg_layernorm_fwd: problem %d has a misaligned pointer", i);
    DSRG_CHECK(p.o_seg >= 0 && (p.o_seg == 0 || (p.o_seg_ld % 4 == 0 && (size_t)p.o_seg_ld >= (size_t)(p.o_seg - 1) * p.ldo + H)),
               "dsrg_layernorm_fwd: problem %d: o_seg_ld must be a multiple of 4 and cover a segment of o_seg rows", i);
    DSRG_CHECK(p.aux_seg >= 0 && (p.aux_seg == 0 || ((!p.aux0 || (p.aux_ld0 % 4 == 0 && p.aux_ld0 >= H && ((uintptr_t)p.aux0 & (out_dtype == DSRG_BF16 ? 7 : 15)) == 0)) &&
                                                      (!p.aux1 || (p.aux_ld1 % 4 == 0 && p.aux_ld1 >= H && ((uintptr_t)p.aux1 & (out_dtype == DSRG_BF16 ? 7 : 15)) == 0)))),
               "dsrg_layernorm_fwd: problem %d: aux destinations need leading dims >= H, multiples of 4, and aligned pointers", i);
    a.blk_start[a.nprob] = blocks;
    a.tag[a.nprob] = i;
    a.p[a.nprob++] = p;
    blocks += (p.M + WPB - 1) / WPB;
  }
  if (a.nprob == 0) return DSRG_OK;
  a.blk_start[a.nprob] = blocks;
  for (int i = a.nprob + 1; i <= DSRG_MAX_GROUPS; ++i) a.blk_start[i] = 0x7fffffff;     // (the kernel's problem search)
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(blocks), block(256);
#define DSRG_LN_LAUNCH(NV_)                                                                                              \
  do {                                                                                                                  \
    if (p_drop > 0.f) {                                                                                                 \
      if (out_dtype == DSRG_F32) hipLaunchKernelGGL((layernorm_kernel<float, float, NV_, true>), grid, block, 0, s, a);      \
      else if (in_dtype == DSRG_F32) hipLaunchKernelGGL((layernorm_kernel<float, bf16_t, NV_, true>), grid, block, 0, s, a); \
      else hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t, NV_, true>), grid, block, 0, s, a);                          \
    } else if (out_dtype == DSRG_F32) hipLaunchKernelGGL((layernorm_kernel<float, float, NV_>), grid, block, 0, s, a);  \
    else if (in_dtype == DSRG_F32) hipLaunchKernelGGL((layernorm_kernel<float, bf16_t, NV_>), grid, block, 0, s, a);    \
    else hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t, NV_>), grid, block, 0, s, a);                             \
  } while (0)
  switch ((a.H + 255) / 256) {
    case 1: DSRG_LN_LAUNCH(1); break;
    case 2: DSRG_LN_LAUNCH(2); break;
    case 3: DSRG_LN_LAUNCH(3); break;
    default: DSRG_LN_LAUNCH(4); break;
  }
#undef DSRG_LN_LAUNCH
  DSRG_LAUNCH_CHECK("dsrg_layernorm_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_layernorm_fwd_grouped(const dsrg_ln_problem_t* problems, int nproblems, int H, float eps, int in_dtype,
                                          int out_dtype, void* stream) {
  return layernorm_fwd_impl(problems, nproblems, H, eps, in_dtype, out_dtype, 0.f, 0, stream);
}

extern "C" int dsrg_layernorm_fwd_grouped_drop(const dsrg_ln_problem_t* problems, int nproblems, int H, float eps, int in_dtype,
                                               int out_dtype, float p_drop, uint64_t seed, void* stream) {
  return layernorm_fwd_impl(problems, nproblems, H, eps, in_dtype, out_dtype, p_drop, seed, stream);
}

extern "C" int dsrg_layernorm_fwd(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const float* beta,
                                  void* out, int ldo, int M, int H, float eps, int in_dtype, int out_dtype, void* stream) {
  dsrg_ln_problem_t p{};
  p.x = x; p.residual = residual; p.gamma = gamma; p.beta = beta; p.out = out;
  p.M = M; p.ldx = ldx; p.ldr = ldr; p.ldo = ldo;
  return dsrg_layernorm_fwd_grouped(&p, 1, H, eps, in_dtype, out_dtype, stream);
}

#ifdef DSRG_EMBED_STAMPS
extern "C" int dsrg_debug_embed_stamps(void* host_dst, int nbytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_embed_stamps), nbytes, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int dsrg_embed_ln_fwd(const void* base, int ldb, const float* feat, int ldf, int F, const float* Wf, const float* bf,
                                 const float* lnf_gamma, const float* lnf_beta,
                                 const float* E1, const void* idx1, int idx1_kind, int idx1_stride, int idx1_div,
                                 const float* E2, const void* idx2, int idx2_kind, int idx2_stride, int idx2_div,
                                 const float* vec, const float* lno_gamma, const float* lno_beta, float eps,
                                 void* out, int ldo, int M, int H, int dtype, void* stream) {
  return dsrg_embed_ln_fwd_seg(base, ldb, nullptr, 0, 0, 0, 0, nullptr, nullptr, feat, ldf, F, Wf, bf, lnf_gamma, lnf_beta, E1, idx1, idx1_kind,
                               idx1_stride, idx1_div, E2, idx2, idx2_kind, idx2_stride, idx2_div, vec, lno_gamma, lno_beta, eps, out, ldo, M, H,
                               dtype, stream);
}

static int embed_ln_impl(const void* base, int ldb, const void* base2, int ldb2, int seg_T, int seg_V, int seg_No,
                         const int64_t* lens0, const int64_t* lens1,
                         const float* feat, int ldf, int F, const float* Wf, const float* bf,
                         const float* lnf_gamma, const float* lnf_beta,
                         const float* E1, const void* idx1, int idx1_kind, int idx1_stride, int idx1_div,
                         const float* E2, const void* idx2, int idx2_kind, int idx2_stride, int idx2_div,
                         const float* vec, const float* lno_gamma, const float* lno_beta, float eps,
                         void* out, int ldo, int M, int H, int dtype, void* stream, const float* lnb_gamma, const float* lnb_beta) {
  DSRG_CHECK((!lnb_gamma && !lnb_beta) || (lnb_gamma && lnb_beta && base && al16(lnb_gamma) && al16(lnb_beta)),
             "dsrg_embed_ln_fwd_lnb: the base LayerNorm needs a base, gamma and beta (16-byte aligned)");
  DSRG_CHECK(seg_T >= 0 && (seg_T == 0 || (base && base2 && lens0 && lens1 && seg_V >= 1 && seg_No >= 1 && M % seg_T == 0 && ldb2 % 4 == 0 &&
                                           ldb2 >= H && ((uintptr_t)base2 & 7) == 0)),
             "dsrg_embed_ln_fwd_seg: the gathered base needs both tables, both length vectors, M %% seg_T == 0 and an aligned base2");
  DSRG_CHECK(ok_dtype(dtype), "dsrg_embed_ln_fwd: bad dtype %d", dtype);
  DSRG_CHECK(out && M >= 0, "dsrg_embed_ln_fwd: null out");
  DSRG_CHECK(H > 0 && H <= 1024 && H % 64 == 0, "dsrg_embed_ln_fwd: H=%d must be a multiple of 64 and <= 1024", H);
  DSRG_CHECK(!feat || (Wf && F >= 1 && F <= 16 && ldf >= F && ((uintptr_t)Wf & 15) == 0), "dsrg_embed_ln_fwd: feature term needs a 16-byte aligned Wf and 1 <= F <= 16");
  DSRG_CHECK((!E1 || idx1) && (!E2 || idx2), "dsrg_embed_ln_fwd: embedding table without indices");
  DSRG_CHECK((!E1 || (idx1_kind >= DSRG_IDX_I32 && idx1_kind <= DSRG_IDX_F32 && idx1_stride >= 1 && idx1_div >= 1)) &&
                 (!E2 || (idx2_kind >= DSRG_IDX_I32 && idx2_kind <= DSRG_IDX_F32 && idx2_stride >= 1 && idx2_div >= 1)),
             "dsrg_embed_ln_fwd: bad index kind / stride / div");
  DSRG_CHECK((!lnf_gamma || lnf_beta) && (!lno_gamma || lno_beta), "dsrg_embed_ln_fwd: LayerNorm gamma without beta");
  DSRG_CHECK(ldo % 4 == 0 && ldo >= H && (!base || (ldb % 4 == 0 && ldb >= H)), "dsrg_embed_ln_fwd: bad leading dims");
  DSRG_CHECK(al16(out) && ((uintptr_t)base & 7) == 0 && al16(E1) && al16(E2) && al16(vec) && al16(lno_gamma) && al16(lno_beta) &&
                 al16(lnf_gamma) && al16(lnf_beta) && al16(bf), "dsrg_embed_ln_fwd: misaligned pointer");
  if (M == 0) return DSRG_OK;
  EmbedArgs a{base, ldb, feat, ldf, F, Wf, bf, lnf_gamma, lnf_beta, E1, {idx1, idx1_kind, idx1_stride, idx1_div},
              E2, {idx2, idx2_kind, idx2_stride, idx2_div}, vec, lno_gamma, lno_beta, eps, out, ldo, M, H,
              base2, ldb2, seg_T, seg_V, seg_No, lens0, lens1, lnb_gamma, lnb_beta};
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((M + EWAVES - 1) / EWAVES), block(64 * EWAVES);
  const size_t lds = feat ? (size_t)F * (H + 4) * sizeof(float) : 0;     // <= 16 * 1028 * 4 = 65,792 B
  // position features of the reference's widths on 768- / 128-wide rows: weights in registers, no LDS (embed_ln_regw_kernel)
  static const bool regw_on = []() { const char* e = getenv("DSRG_EMBED_REGW"); return !(e && e[0] == '0'); }();
  if (regw_on && feat && (F == 7 || F == 14) && (H == 768 || H == 128)) {
    dim3 g2((M + 4 * ER - 1) / (4 * ER)), b2(256);
#define DSRG_REGW_LAUNCH(T_)                                                                               \
    do {                                                                                                   \
      if (H == 768 && F == 14) hipLaunchKernelGGL((embed_ln_regw_kernel<T_, 3, 14>), g2, b2, 0, s, a);      \
      else if (H == 768) hipLaunchKernelGGL((embed_ln_regw_kernel<T_, 3, 7>), g2, b2, 0, s, a);             \
      else if (F == 14) hipLaunchKernelGGL((embed_ln_regw_kernel<T_, 1, 14>), g2, b2, 0, s, a);             \
      else hipLaunchKernelGGL((embed_ln_regw_kernel<T_, 1, 7>), g2, b2, 0, s, a);                           \
    } while (0)
    if (dtype == DSRG_F32) DSRG_REGW_LAUNCH(float);
    else DSRG_REGW_LAUNCH(bf16_t);
#undef DSRG_REGW_LAUNCH
    DSRG_LAUNCH_CHECK("dsrg_embed_ln_fwd");
    return DSRG_OK;
  }
#define DSRG_EMBED_LAUNCH(T_, NV_)                                                      \
  do {                                                                                  \
    DSRG_SET_MAX_LDS_ONCE((&embed_ln_kernel<T_, NV_>), 66 * 1024);                      \
    hipLaunchKernelGGL((embed_ln_kernel<T_, NV_>), grid, block, lds, s, a);             \
  } while (0)
  const int nvc = (H + 255) / 256;
  if (dtype == DSRG_F32) {
    if (nvc == 1) DSRG_EMBED_LAUNCH(float, 1); else if (nvc == 2) DSRG_EMBED_LAUNCH(float, 2);
    else if (nvc == 3) DSRG_EMBED_LAUNCH(float, 3); else DSRG_EMBED_LAUNCH(float, 4);
  } else {
    if (nvc == 1) DSRG_EMBED_LAUNCH(bf16_t, 1); else if (nvc == 2) DSRG_EMBED_LAUNCH(bf16_t, 2);
    else if (nvc == 3) DSRG_EMBED_LAUNCH(bf16_t, 3); else DSRG_EMBED_LAUNCH(bf16_t, 4);
  }
#undef DSRG_EMBED_LAUNCH
  DSRG_LAUNCH_CHECK("dsrg_embed_ln_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_embed_ln_fwd_seg(const void* base, int ldb, const void* base2, int ldb2, int seg_T, int seg_V, int seg_No,
                                     const int64_t* lens0, const int64_t* lens1,
                                     const float* feat, int ldf, int F, const float* Wf, const float* bf,
                                     const float* lnf_gamma, const float* lnf_beta,
                                     const float* E1, const void* idx1, int idx1_kind, int idx1_stride, int idx1_div,
                                     const float* E2, const void* idx2, int idx2_kind, int idx2_stride, int idx2_div,
                                     const float* vec, const float* lno_gamma, const float* lno_beta, float eps,
                                     void* out, int ldo, int M, int H, int dtype, void* stream) {
  return embed_ln_impl(base, ldb, base2, ldb2, seg_T, seg_V, seg_No, lens0, lens1, feat, ldf, F, Wf, bf, lnf_gamma, lnf_beta, E1, idx1, idx1_kind,
                       idx1_stride, idx1_div, E2, idx2, idx2_kind, idx2_stride, idx2_div, vec, lno_gamma, lno_beta, eps, out, ldo, M, H, dtype,
                       stream, nullptr, nullptr);
}

extern "C" int dsrg_embed_ln_fwd_lnb(const void* base, int ldb, const float* lnb_gamma, const float* lnb_beta,
                                     const float* feat, int ldf, int F, const float* Wf, const float* bf,
                                     const float* lnf_gamma, const float* lnf_beta,
                                     const float* E1, const void* idx1, int idx1_kind, int idx1_stride, int idx1_div,
                                     const float* E2, const void* idx2, int idx2_kind, int idx2_stride, int idx2_div,
                                     const float* vec, const float* lno_gamma, const float* lno_beta, float eps,
                                     void* out, int ldo, int M, int H, int dtype, void* stream) {
  return embed_ln_impl(base, ldb, nullptr, 0, 0, 0, 0, nullptr, nullptr, feat, ldf, F, Wf, bf, lnf_gamma, lnf_beta, E1, idx1, idx1_kind,
                       idx1_stride, idx1_div, E2, idx2, idx2_kind, idx2_stride, idx2_div, vec, lno_gamma, lno_beta, eps, out, ldo, M, H, dtype,
                       stream, lnb_gamma, lnb_beta);
}

extern "C" int dsrg_group_mean_fwd(const void* x, int ldx, void* out, int ldo, int M, int G, int H, int dtype, void* stream) {
  DSRG_CHECK(ok_dtype(dtype) && x && out, "dsrg_group_mean_fwd: bad dtype or null pointer");
  DSRG_CHECK(G >= 1 && H > 0 && H % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldx >= H && ldo >= H, "dsrg_group_mean_fwd: bad dims");
  DSRG_CHECK(((uintptr_t)x & 7) == 0 && ((uintptr_t)out & 7) == 0 && (dtype == DSRG_BF16 || (al16(x) && al16(out))), "dsrg_group_mean_fwd: misaligned pointer");
  if (M <= 0) return DSRG_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((M * (H / 4) + 255) / 256), block(256);
  if (dtype == DSRG_F32) hipLaunchKernelGGL((group_mean_kernel<float>), grid, block, 0, s, (const float*)x, ldx, (float*)out, ldo, M, G, H);
  else hipLaunchKernelGGL((group_mean_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, M, G, H);
  DSRG_LAUNCH_CHECK("dsrg_group_mean_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_gaa_pool_fwd(const void* x, const float* w, const float* bias, void* out, int B, int V, int H, int dtype, void* stream) {
  return dsrg_gaa_pool_fwd_ex(x, (int64_t)V * H, w, bias, out, nullptr, 0, nullptr, 0, B, V, H, dtype, stream);
}

extern "C" int dsrg_gaa_pool_fwd_ex(const void* x, int64_t x_bs, const float* w, const float* bias, void* out, void* out2, int64_t out2_bs,
                                    const int64_t* idx2, int idx2_max, int B, int V, int H, int dtype, void* stream) {
  DSRG_CHECK(ok_dtype(dtype) && x && w && bias && out, "dsrg_gaa_pool_fwd: bad dtype or null pointer");
  DSRG_CHECK(V >= 1 && V <= 64 && H > 0 && H % 4 == 0 && H <= 1024, "dsrg_gaa_pool_fwd: need 1 <= V <= 64 (got %d), H %% 4 == 0 and H <= 1024", V);
  DSRG_CHECK(al16(x) && al16(out) && al16(w), "dsrg_gaa_pool_fwd: misaligned pointer");
  DSRG_CHECK(x_bs >= (int64_t)V * H && x_bs % 8 == 0, "dsrg_gaa_pool_fwd_ex: x_bs must cover V rows and keep 16-byte alignment");
  DSRG_CHECK(!out2 || (idx2 && idx2_max >= 1 && out2_bs >= (int64_t)idx2_max * H && out2_bs % 8 == 0 && al16(out2)),
             "dsrg_gaa_pool_fwd_ex: out2 needs idx2, idx2_max >= 1 and a batch stride that covers idx2_max rows");
  if (B <= 0) return DSRG_OK;
  hipStream_t s = (hipStream_t)stream;
  DSRG_SET_MAX_LDS_ONCE(&gaa_pool_kernel<float>, 65536);     // 16 x 1024 floats + the score vector is just over the 64 KiB default
  DSRG_SET_MAX_LDS_ONCE(&gaa_pool_kernel<bf16_t>, 65536);
  if (dtype == DSRG_F32) hipLaunchKernelGGL((gaa_pool_kernel<float>), dim3(B), dim3(1024), (size_t)GWAVES * H * 4, s, (const float*)x, (long)x_bs, w, bias, (float*)out, (float*)out2, (long)out2_bs, idx2, idx2_max, V, H, (const float*)nullptr, (const float*)nullptr, 0.f, (float*)nullptr, 0L);
  else hipLaunchKernelGGL((gaa_pool_kernel<bf16_t>), dim3(B), dim3(1024), (size_t)GWAVES * H * 4, s, (const bf16_t*)x, (long)x_bs, w, bias, (bf16_t*)out, (bf16_t*)out2, (long)out2_bs, idx2, idx2_max, V, H, (const float*)nullptr, (const float*)nullptr, 0.f, (bf16_t*)nullptr, 0L);
  DSRG_LAUNCH_CHECK("dsrg_gaa_pool_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_gaa_pool_fwd_ln(const void* x, int64_t x_bs, const float* ln_gamma, const float* ln_beta, float ln_eps, void* rows_out,
                                    int64_t rows_bs, const float* w, const float* bias, void* out, void* out2, int64_t out2_bs,
                                    const int64_t* idx2, int idx2_max, int B, int V, int H, int dtype, void* stream) {
  DSRG_CHECK(ok_dtype(dtype) && x && w && bias && out && ln_gamma && ln_beta && rows_out, "dsrg_gaa_pool_fwd_ln: bad dtype or null pointer");
  DSRG_CHECK(V >= 1 && V <= 64 && H > 0 && H % 64 == 0 && H <= 1024, "dsrg_gaa_pool_fwd_ln: need 1 <= V <= 64 (got %d), H %% 64 == 0 and H <= 1024", V);
  DSRG_CHECK(al16(x) && al16(out) && al16(w) && al16(ln_gamma) && al16(ln_beta) && al16(rows_out), "dsrg_gaa_pool_fwd_ln: misaligned pointer");
  DSRG_CHECK(x_bs >= (int64_t)V * H && x_bs % 8 == 0 && rows_bs >= (int64_t)V * H && rows_bs % 8 == 0,
             "dsrg_gaa_pool_fwd_ln: x_bs / rows_bs must cover V rows and keep 16-byte alignment");
  DSRG_CHECK(!out2 || (idx2 && idx2_max >= 1 && out2_bs >= (int64_t)idx2_max * H && out2_bs % 8 == 0 && al16(out2)),
             "dsrg_gaa_pool_fwd_ln: out2 needs idx2, idx2_max >= 1 and a batch stride that covers idx2_max rows");
  if (B <= 0) return DSRG_OK;
  hipStream_t s = (hipStream_t)stream;
  DSRG_SET_MAX_LDS_ONCE(&gaa_pool_kernel<float>, 65536);
  DSRG_SET_MAX_LDS_ONCE(&gaa_pool_kernel<bf16_t>, 65536);
  if (dtype == DSRG_F32) hipLaunchKernelGGL((gaa_pool_kernel<float>), dim3(B), dim3(1024), (size_t)GWAVES * H * 4, s, (const float*)x, (long)x_bs, w, bias, (float*)out, (float*)out2, (long)out2_bs, idx2, idx2_max, V, H, ln_gamma, ln_beta, ln_eps, (float*)rows_out, (long)rows_bs);
  else hipLaunchKernelGGL((gaa_pool_kernel<bf16_t>), dim3(B), dim3(1024), (size_t)GWAVES * H * 4, s, (const bf16_t*)x, (long)x_bs, w, bias, (bf16_t*)out, (bf16_t*)out2, (long)out2_bs, idx2, idx2_max, V, H, ln_gamma, ln_beta, ln_eps, (bf16_t*)rows_out, (long)rows_bs);
  DSRG_LAUNCH_CHECK("dsrg_gaa_pool_fwd_ln");
  return DSRG_OK;
}

extern "C" int dsrg_head_tail_fwd(const float* h, int ldh, const float* gamma, const float* beta, const float* w2, const float* b2,
                                  float eps, float* score, int M, int H, void* stream) {
  DSRG_CHECK(h && gamma && beta && w2 && b2 && score, "dsrg_head_tail_fwd: null pointer");
  DSRG_CHECK(H > 0 && H <= 1024 && H % 64 == 0 && ldh % 4 == 0 && ldh >= H, "dsrg_head_tail_fwd: bad H/ldh");
  DSRG_CHECK(al16(h) && al16(gamma) && al16(beta) && al16(w2), "dsrg_head_tail_fwd: misaligned pointer");
  if (M <= 0) return DSRG_OK;
  hipLaunchKernelGGL(head_tail_kernel, dim3((M + WPB - 1) / WPB), dim3(256), 0, (hipStream_t)stream, h, ldh, gamma, beta, w2, b2, eps, score, M, H);
  DSRG_LAUNCH_CHECK("dsrg_head_tail_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_head_tail_fwd_grouped(const dsrg_head_tail_problem_t* problems, int nproblems, int H, float eps, void* stream) {
  DSRG_CHECK(problems && nproblems >= 1 && nproblems <= DSRG_MAX_GROUPS, "dsrg_head_tail_fwd_grouped: nproblems=%d out of [1,%d]", nproblems, DSRG_MAX_GROUPS);
  DSRG_CHECK(H > 0 && H <= 1024 && H % 64 == 0, "dsrg_head_tail_fwd_grouped: bad H=%d", H);
  HeadTailGroup g;
  g.n = 0; g.H = H; g.eps = eps;
  int blocks = 0;
  for (int i = 0; i < nproblems; ++i) {
    const dsrg_head_tail_problem_t& p = problems[i];
    DSRG_CHECK(p.M >= 0, "dsrg_head_tail_fwd_grouped: problem %d has M < 0", i);
    if (p.M == 0) continue;
    DSRG_CHECK(p.h && p.gamma && p.beta && p.w2 && p.b2 && p.score, "dsrg_head_tail_fwd_grouped: problem %d has a null pointer", i);
    DSRG_CHECK(p.ldh % 4 == 0 && p.ldh >= H && al16(p.h) && al16(p.gamma) && al16(p.beta) && al16(p.w2),
               "dsrg_head_tail_fwd_grouped: problem %d: bad ldh or misaligned pointer", i);
    g.p[g.n] = p;
    g.blk_start[g.n] = blocks;
    blocks += (p.M + WPB - 1) / WPB;
    ++g.n;
  }
  if (g.n == 0) return DSRG_OK;
  g.blk_start[g.n] = blocks;
  hipLaunchKernelGGL(head_tail_grouped_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
  DSRG_LAUNCH_CHECK("dsrg_head_tail_fwd_grouped");
  return DSRG_OK;
}

extern "C" int dsrg_nav_logits_fwd(const float* g_raw, const float* l_raw, const float* fuse_raw, const uint8_t* gmap_masks,
                                   const uint8_t* gmap_visited, const uint8_t* vp_nav_masks, const int32_t* gmap_src,
                                   const uint8_t* bw_mask, float* global_logits, float* local_logits, float* fused_logits,
                                   int B, int G, int Vp, void* stream) {
  return dsrg_nav_logits_fwd_og(g_raw, l_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask, global_logits, local_logits,
                                fused_logits, nullptr, nullptr, nullptr, B, G, Vp, stream);
}

extern "C" int dsrg_nav_logits_fwd_og(const float* g_raw, const float* l_raw, const float* fuse_raw, const uint8_t* gmap_masks,
                                      const uint8_t* gmap_visited, const uint8_t* vp_nav_masks, const int32_t* gmap_src,
                                      const uint8_t* bw_mask, float* global_logits, float* local_logits, float* fused_logits,
                                      const float* o_raw, const uint8_t* vp_obj_masks, float* obj_logits, int B, int G, int Vp, void* stream) {
  DSRG_CHECK(g_raw && l_raw && gmap_masks && gmap_visited && vp_nav_masks && gmap_src && bw_mask && global_logits && local_logits && fused_logits,
             "dsrg_nav_logits_fwd: null pointer");
  DSRG_CHECK((o_raw && vp_obj_masks && obj_logits) || (!o_raw && !vp_obj_masks && !obj_logits), "dsrg_nav_logits_fwd_og: o_raw, vp_obj_masks and obj_logits come together");
  DSRG_CHECK(G >= 1 && Vp >= 1 && Vp <= NAV_MAX_VP, "dsrg_nav_logits_fwd: need G >= 1 and 1 <= Vp <= %d", NAV_MAX_VP);
  if (B <= 0) return DSRG_OK;
  hipLaunchKernelGGL(nav_logits_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, g_raw, l_raw, fuse_raw, gmap_masks, gmap_visited,
                     vp_nav_masks, gmap_src, bw_mask, global_logits, local_logits, fused_logits, o_raw, vp_obj_masks, obj_logits, G, Vp);
  DSRG_LAUNCH_CHECK("dsrg_nav_logits_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_igl_blend_fwd(const void* aug, const void* x, const float* wa, const float* ba, const float* wo, const float* bo,
                                  void* out, int M, int H, int dtype, void* stream) {
  DSRG_CHECK(ok_dtype(dtype) && aug && x && wa && ba && wo && bo && out, "dsrg_igl_blend_fwd: bad dtype or null pointer");
  DSRG_CHECK(H > 0 && H <= 1024 && H % 64 == 0, "dsrg_igl_blend_fwd: bad H");
  DSRG_CHECK(al16(aug) && al16(x) && al16(out) && al16(wa) && al16(wo), "dsrg_igl_blend_fwd: misaligned pointer");
  if (M <= 0) return DSRG_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((M + WPB - 1) / WPB), block(256);
  if (dtype == DSRG_F32) hipLaunchKernelGGL((igl_blend_kernel<float>), grid, block, 0, s, (const float*)aug, (const float*)x, wa, ba, wo, bo, (float*)out, M, H);
  else hipLaunchKernelGGL((igl_blend_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)aug, (const bf16_t*)x, wa, ba, wo, bo, (bf16_t*)out, M, H);
  DSRG_LAUNCH_CHECK("dsrg_igl_blend_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_ce_rows_fwd(const float* logits, int ld, const int64_t* targets, int64_t ignore_index, float* lse_rows,
                                float* loss_rows, int B, int N, void* stream) {
  DSRG_CHECK(logits && targets && lse_rows && loss_rows && B >= 0 && N >= 1 && ld >= N, "dsrg_ce_rows_fwd: bad arguments");
  if (B == 0) return DSRG_OK;
  hipLaunchKernelGGL(ce_rows_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ld, targets, ignore_index, lse_rows, loss_rows, N);
  DSRG_LAUNCH_CHECK("dsrg_ce_rows_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_kl_rows_fwd(const float* logits, int ld, const float* target, int ldt, float* lse_rows, float* loss_rows, int R, int N,
                               void* stream) {
  DSRG_CHECK(logits && target && lse_rows && loss_rows && R >= 0 && N >= 1 && ld >= N && ldt >= N, "dsrg_kl_rows_fwd: bad arguments");
  if (R == 0) return DSRG_OK;
  hipLaunchKernelGGL(kl_rows_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, target, ldt, lse_rows, loss_rows, R, N);
  DSRG_LAUNCH_CHECK("dsrg_kl_rows_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_kl_rows_bwd(const float* logits, int ld, const float* target, int ldt, const float* lse_rows, const float* grad_rows,
                               float* dlogits, int ldd, int R, int N, void* stream) {
  DSRG_CHECK(logits && target && lse_rows && grad_rows && dlogits && R >= 0 && N >= 1 && ld >= N && ldt >= N && ldd >= N, "dsrg_kl_rows_bwd: bad arguments");
  if (R == 0) return DSRG_OK;
  hipLaunchKernelGGL(kl_rows_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, target, ldt, lse_rows, grad_rows, dlogits, ldd, R, N);
  DSRG_LAUNCH_CHECK("dsrg_kl_rows_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_ce_rows_bwd(const float* logits, int ld, const int64_t* targets, int64_t ignore_index, const float* lse_rows,
                                const float* grad_rows, void* dlogits, int ldd, int out_dtype, int B, int N, void* stream) {
  DSRG_CHECK(logits && targets && lse_rows && grad_rows && dlogits && B >= 0 && N >= 1 && ld >= N && ldd >= N, "dsrg_ce_rows_bwd: bad arguments");
  DSRG_CHECK(ok_dtype(out_dtype), "dsrg_ce_rows_bwd: bad out_dtype %d", out_dtype);
  if (B == 0) return DSRG_OK;
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == DSRG_F32) hipLaunchKernelGGL((ce_rows_bwd_kernel<float>), dim3(B), dim3(256), 0, s, logits, ld, targets, ignore_index, lse_rows, grad_rows, (float*)dlogits, ldd, N);
  else hipLaunchKernelGGL((ce_rows_bwd_kernel<bf16_t>), dim3(B), dim3(256), 0, s, logits, ld, targets, ignore_index, lse_rows, grad_rows, (bf16_t*)dlogits, ldd, N);
  DSRG_LAUNCH_CHECK("dsrg_ce_rows_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_action_tail_fwd(const float* logits, int ld, const int64_t* targets, int64_t ignore_index, const float* u,
                                    float* stop_prob, int32_t* argmax, int32_t* sample, float* entropy, float* loss_rows,
                                    float* dlogits, int B, int N, void* stream) {
  DSRG_CHECK(logits && stop_prob && argmax && B >= 0 && N >= 1 && ld >= N, "dsrg_action_tail_fwd: bad arguments");
  DSRG_CHECK(!sample || u, "dsrg_action_tail_fwd: sampling needs the uniform draws u[B]");
  DSRG_CHECK(!(loss_rows || dlogits) || targets, "dsrg_action_tail_fwd: loss / gradient output needs targets");
  if (B == 0) return DSRG_OK;
  hipLaunchKernelGGL(action_tail_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, ld, targets, ignore_index, u, stop_prob, argmax,
                     sample, entropy, loss_rows, dlogits, N);
  DSRG_LAUNCH_CHECK("dsrg_action_tail_fwd");
  return DSRG_OK;
}
