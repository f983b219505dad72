// Row-wise kernels: LayerNorm, token-embedding sums, GAA pooling, head tails, logit fusion, IGL gate.
// One 64-lane wave owns one row; statistics by wave reduction in fp32.  HBM-bound (rows are read once,
// 8/16 bytes per lane), so the only rules that matter are coalescing and enough waves in flight.
#include <stdarg.h>
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

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, int ldx, const TO* __restrict__ res, int ldr,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        TO* __restrict__ out, int ldo, int M, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = (H + 255) / 256;
  float4 v[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < H) {
      v[i] = ld4(x + (size_t)row * ldx + c);
      if (res) v[i] = f4add(v[i], ld4(res + (size_t)row * ldr + c));
    }
  }
  ln_row<MAXV>(v, nv, lane, H, gamma, beta, eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < H) st4(out + (size_t)row * ldo + c, v[i]);
  }
}

struct EmbedArgs {
  const void* base; int ldb;
  const float* feat; int ldf; int F;
  const float* Wf; const float* bf; const float* lnf_g; const float* lnf_b;
  const float* E1; const int32_t* idx1; const float* E2; const int32_t* idx2;
  const float* vec; const float* lno_g; const float* lno_b;
  float eps; void* out; int ldo; int M; int H;
};

template <typename T>
__global__ __launch_bounds__(256) void embed_ln_kernel(const EmbedArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const int H = a.H, nv = (H + 255) / 256;
  float4 v[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.feat) {
    float f[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) f[k] = (k < a.F) ? a.feat[(size_t)row * a.ldf + k] : 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < H) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* w = a.Wf + (size_t)(c + e) * a.F;
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (k < a.F) s = fmaf(w[k], f[k], s);
          o[e] = s + (a.bf ? a.bf[c + e] : 0.f);
        }
        v[i] = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    if (a.lnf_g) ln_row<MAXV>(v, nv, lane, H, a.lnf_g, a.lnf_b, a.eps);
  }
  const T* base = reinterpret_cast<const T*>(a.base);
  const int i1 = a.E1 ? a.idx1[row] : 0, i2 = a.E2 ? a.idx2[row] : 0;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < H) {
      if (base) v[i] = f4add(ld4(base + (size_t)row * a.ldb + c), v[i]);
      if (a.E1) v[i] = f4add(v[i], *reinterpret_cast<const float4*>(a.E1 + (size_t)i1 * H + c));
      if (a.E2) v[i] = f4add(v[i], *reinterpret_cast<const float4*>(a.E2 + (size_t)i2 * H + c));
      if (a.vec) v[i] = f4add(v[i], *reinterpret_cast<const float4*>(a.vec + c));
    }
  }
  if (a.lno_g) ln_row<MAXV>(v, nv, lane, H, a.lno_g, a.lno_b, a.eps);
  T* out = reinterpret_cast<T*>(a.out);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < H) st4(out + (size_t)row * a.ldo + c, v[i]);
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

// one block per episode; wave w scores views w, w+4, ...; then all threads form the weighted sum
template <typename T>
__global__ __launch_bounds__(256) void gaa_pool_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       T* __restrict__ out, int V, int H) {
  __shared__ float sc[64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* xb = x + (size_t)b * V * H;
  for (int v = wave; v < V; v += 4) {
    float s = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
      float4 a = ld4(xb + (size_t)v * H + c), ww = *reinterpret_cast<const float4*>(w + c);
      s += a.x * ww.x + a.y * ww.y + a.z * ww.z + a.w * ww.w;
    }
    s = wave_sum(s);
    if (lane == 0) sc[v] = tanhf(s + bias[0]);
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int v = 0; v < V; ++v) mx = fmaxf(mx, sc[v]);
  float den = 0.f;
  for (int v = 0; v < V; ++v) den += expf(sc[v] - mx);
  const float inv = 1.0f / den;
  for (int c = threadIdx.x * 4; c < H; c += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int v = 0; v < V; ++v) {
      const float p = expf(sc[v] - mx) * inv;
      float4 a = ld4(xb + (size_t)v * H + c);
      acc.x += a.x * p; acc.y += a.y * p; acc.z += a.z * p; acc.w += a.w * p;
    }
    st4(out + (size_t)b * H + c, acc);
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

// one wave per episode; local logits are staged in LDS for the gather by graph node
constexpr int NAV_MAX_VP = 1024;
__global__ __launch_bounds__(64) void nav_logits_kernel(const float* __restrict__ g_raw, const float* __restrict__ l_raw,
                                                        const float* __restrict__ fuse_raw, const uint8_t* __restrict__ gmask,
                                                        const uint8_t* __restrict__ gvis, const uint8_t* __restrict__ vnav,
                                                        const int32_t* __restrict__ gsrc, const uint8_t* __restrict__ bwm,
                                                        float* __restrict__ gl, float* __restrict__ ll, float* __restrict__ fl,
                                                        int G, int Vp) {
  __shared__ float sl[NAV_MAX_VP];
  __shared__ uint8_t sb[NAV_MAX_VP];
  const int b = blockIdx.x, lane = threadIdx.x;
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

extern "C" int dsrg_layernorm_fwd(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const float* beta,
                                  void* out, int ldo, int M, int H, float eps, int in_dtype, int out_dtype, void* stream) {
  DSRG_CHECK(ok_dtype(in_dtype) && ok_dtype(out_dtype), "dsrg_layernorm_fwd: bad dtype");
  DSRG_CHECK(in_dtype == DSRG_F32 || in_dtype == out_dtype, "dsrg_layernorm_fwd: in_dtype must be f32 or equal out_dtype");
  DSRG_CHECK(x && gamma && beta && out && M >= 0, "dsrg_layernorm_fwd: null argument");
  DSRG_CHECK(H > 0 && H <= 1024 && H % 64 == 0, "dsrg_layernorm_fwd: H=%d must be a multiple of 64 and <= 1024", H);
  DSRG_CHECK(ldx % 4 == 0 && ldo % 4 == 0 && (!residual || ldr % 4 == 0) && ldx >= H && ldo >= H, "dsrg_layernorm_fwd: bad leading dims");
  DSRG_CHECK(al16(x) && al16(out) && al16(gamma) && al16(beta) && ((uintptr_t)residual & 7) == 0, "dsrg_layernorm_fwd: misaligned pointer");
  if (M == 0) return DSRG_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((M + WPB - 1) / WPB), block(256);
  if (in_dtype == DSRG_F32 && out_dtype == DSRG_F32)
    hipLaunchKernelGGL((layernorm_kernel<float, float>), grid, block, 0, s, (const float*)x, ldx, (const float*)residual, ldr, gamma, beta, (float*)out, ldo, M, H, eps);
  else if (in_dtype == DSRG_F32)
    hipLaunchKernelGGL((layernorm_kernel<float, bf16_t>), grid, block, 0, s, (const float*)x, ldx, (const bf16_t*)residual, ldr, gamma, beta, (bf16_t*)out, ldo, M, H, eps);
  else
    hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)x, ldx, (const bf16_t*)residual, ldr, gamma, beta, (bf16_t*)out, ldo, M, H, eps);
  DSRG_LAUNCH_CHECK("dsrg_layernorm_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_embed_ln_fwd(const void* base, int ldb, const float* feat, int ldf, int F, const float* Wf, const float* bf,
                                 const float* lnf_gamma, const float* lnf_beta, const float* E1, const int32_t* idx1,
                                 const float* E2, const int32_t* idx2, const float* vec, const float* lno_gamma,
                                 const float* lno_beta, float eps, void* out, int ldo, int M, int H, int dtype, void* stream) {
  DSRG_CHECK(ok_dtype(dtype), "dsrg_embed_ln_fwd: bad dtype %d", dtype);
  DSRG_CHECK(out && M >= 0, "dsrg_embed_ln_fwd: null out");
  DSRG_CHECK(H > 0 && H <= 1024 && H % 64 == 0, "dsrg_embed_ln_fwd: H=%d must be a multiple of 64 and <= 1024", H);
  DSRG_CHECK(!feat || (Wf && F >= 1 && F <= 16 && ldf >= F), "dsrg_embed_ln_fwd: feature term needs Wf and 1 <= F <= 16");
  DSRG_CHECK((!E1 || idx1) && (!E2 || idx2), "dsrg_embed_ln_fwd: embedding table without indices");
  DSRG_CHECK((!lnf_gamma || lnf_beta) && (!lno_gamma || lno_beta), "dsrg_embed_ln_fwd: LayerNorm gamma without beta");
  DSRG_CHECK(ldo % 4 == 0 && ldo >= H && (!base || (ldb % 4 == 0 && ldb >= H)), "dsrg_embed_ln_fwd: bad leading dims");
  DSRG_CHECK(al16(out) && ((uintptr_t)base & 7) == 0 && al16(E1) && al16(E2) && al16(vec) && al16(lno_gamma) && al16(lno_beta) &&
                 al16(lnf_gamma) && al16(lnf_beta), "dsrg_embed_ln_fwd: misaligned pointer");
  if (M == 0) return DSRG_OK;
  EmbedArgs a{base, ldb, feat, ldf, F, Wf, bf, lnf_gamma, lnf_beta, E1, idx1, E2, idx2, vec, lno_gamma, lno_beta, eps, out, ldo, M, H};
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((M + WPB - 1) / WPB), block(256);
  if (dtype == DSRG_F32) hipLaunchKernelGGL((embed_ln_kernel<float>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((embed_ln_kernel<bf16_t>), grid, block, 0, s, a);
  DSRG_LAUNCH_CHECK("dsrg_embed_ln_fwd");
  return DSRG_OK;
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
  DSRG_CHECK(ok_dtype(dtype) && x && w && bias && out, "dsrg_gaa_pool_fwd: bad dtype or null pointer");
  DSRG_CHECK(V >= 1 && V <= 64 && H > 0 && H % 4 == 0, "dsrg_gaa_pool_fwd: need 1 <= V <= 64 (got %d) and H %% 4 == 0", V);
  DSRG_CHECK(al16(x) && al16(out) && al16(w), "dsrg_gaa_pool_fwd: misaligned pointer");
  if (B <= 0) return DSRG_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DSRG_F32) hipLaunchKernelGGL((gaa_pool_kernel<float>), dim3(B), dim3(256), 0, s, (const float*)x, w, bias, (float*)out, V, H);
  else hipLaunchKernelGGL((gaa_pool_kernel<bf16_t>), dim3(B), dim3(256), 0, s, (const bf16_t*)x, w, bias, (bf16_t*)out, V, H);
  DSRG_LAUNCH_CHECK("dsrg_gaa_pool_fwd");
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

extern "C" int dsrg_nav_logits_fwd(const float* g_raw, const float* l_raw, const float* fuse_raw, const uint8_t* gmap_masks,
                                   const uint8_t* gmap_visited, const uint8_t* vp_nav_masks, const int32_t* gmap_src,
                                   const uint8_t* bw_mask, float* global_logits, float* local_logits, float* fused_logits,
                                   int B, int G, int Vp, void* stream) {
  DSRG_CHECK(g_raw && l_raw && gmap_masks && gmap_visited && vp_nav_masks && gmap_src && bw_mask && global_logits && local_logits && fused_logits,
             "dsrg_nav_logits_fwd: null pointer");
  DSRG_CHECK(G >= 1 && Vp >= 1 && Vp <= NAV_MAX_VP, "dsrg_nav_logits_fwd: need G >= 1 and 1 <= Vp <= %d", NAV_MAX_VP);
  if (B <= 0) return DSRG_OK;
  hipLaunchKernelGGL(nav_logits_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, g_raw, l_raw, fuse_raw, gmap_masks, gmap_visited,
                     vp_nav_masks, gmap_src, bw_mask, global_logits, local_logits, fused_logits, G, Vp);
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
