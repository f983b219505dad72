// Backward of the row kernels of rowops.hip (training path): ASV group mean, GAA pooling, ClsPrediction tail,
// dynamic logit fusion, IGL gate and the token-embedding sums.  Same ownership as the forward (one wave per row /
// one workgroup per episode); parameter gradients are summed per workgroup (registers -> LDS) and leave as ONE float
// atomic per element per workgroup into zeroed fp32 buffers.  All arithmetic fp32, forward values are recomputed.
#include "common.h"

namespace {

constexpr int MAXV = 4;       // float4 per lane: H <= 1024
constexpr int WPB = 4;

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator*(float a, float4 b) { return make_float4(a * b.x, a * b.y, a * b.z, a * b.w); }
__device__ __forceinline__ float hsum(float4 a) { return a.x + a.y + a.z + a.w; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void lds_add4(float* p, float4 v) { atomicAdd(p, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w); }

#define ROWV(i, c) for (int i = 0; i < MAXV; ++i) { const int c = (i * 64 + lane) * 4; if (i < nv && c < H)
#define ENDV }

// normalise a row in place: v <- (v - mean) * rstd, returns rstd
__device__ __forceinline__ float ln_hat(float4 (&v)[MAXV], int nv, int lane, int H, float eps) {
  float s = 0.f;
#pragma unroll
  ROWV(i, c) s += hsum(v[i]); ENDV
  const float mean = wave_sum(s) / (float)H;
  float ss = 0.f;
#pragma unroll
  ROWV(i, c) { v[i] = v[i] - f4(mean); ss += dot4(v[i], v[i]); } ENDV
  const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
#pragma unroll
  ROWV(i, c) v[i] = rstd * v[i]; ENDV
  return rstd;
}

// dz of LayerNorm given g = dy*gamma and the normalised row zh:  rstd * (g - mean(g) - zh * mean(g*zh))
__device__ __forceinline__ void ln_dz(float4 (&g)[MAXV], const float4 (&zh)[MAXV], float rstd, int nv, int lane, int H) {
  float a = 0.f, b = 0.f;
#pragma unroll
  ROWV(i, c) { a += hsum(g[i]); b += dot4(g[i], zh[i]); } ENDV
  a = wave_sum(a) / (float)H;
  b = wave_sum(b) / (float)H;
#pragma unroll
  ROWV(i, c) g[i] = rstd * (g[i] - f4(a) - b * zh[i]); ENDV
}

// ---- ASV group mean: dx[m*G+g, :] = dy[m, :] / G
template <typename T>
__global__ __launch_bounds__(256) void group_mean_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int M, int G, int H) {
  const int idx = blockIdx.x * 256 + threadIdx.x, hv = H / 4;
  if (idx >= M * G * hv) return;
  const int r = idx / hv, c = (idx % hv) * 4;
  const float4 g = ld4(dy + (size_t)(r / G) * lddy + c);
  st4(dx + (size_t)r * lddx + c, (1.0f / (float)G) * g);
}

// ---- GAA pooling backward: a = softmax_v(tanh(x_v.w + b)), out = sum_v a_v x_v
//   da_v = dout.x_v;  dt_v = a_v (da_v - sum_u a_u da_u);  ds_v = dt_v (1 - t_v^2);  dx_v = a_v dout + ds_v w
//   dw = sum_{b,v} ds_v x_v;  db = sum ds_v.   One 16-wave workgroup per episode, rows held in registers (V <= 64).
constexpr int GWAVES = 16, GVPW = 4;
template <typename T>
__global__ __launch_bounds__(1024) void gaa_pool_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            const T* __restrict__ dout, T* __restrict__ dx, float* __restrict__ dw,
                                                            float* __restrict__ db, int V, int H) {
  __shared__ float tv[64], av[64], da[64];
  extern __shared__ __attribute__((aligned(16))) float part[];      // [H] workgroup sum of ds_v x_v
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = (H + 255) / 256;
  const T* xb = x + (size_t)b * V * H;
  float4 row[GVPW][MAXV], ww[MAXV], go[MAXV];
#pragma unroll
  for (int r = 0; r < GVPW; ++r)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = wave + GWAVES * r, c = (i * 64 + lane) * 4;
      row[r][i] = (v < V && i < nv && c < H) ? ld4(xb + (size_t)v * H + c) : f4(0.f);
    }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const bool ok = i < nv && c < H;
    ww[i] = ok ? *reinterpret_cast<const float4*>(w + c) : f4(0.f);
    go[i] = ok ? ld4(dout + (size_t)b * H + c) : f4(0.f);
  }
  for (int c = threadIdx.x; c < H; c += 1024) part[c] = 0.f;
  const float b0 = bias[0];
#pragma unroll
  for (int r = 0; r < GVPW; ++r) {
    const int v = wave + GWAVES * r;
    float s = 0.f, d = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { s += dot4(row[r][i], ww[i]); d += dot4(row[r][i], go[i]); }
    s = wave_sum(s); d = wave_sum(d);
    if (lane == 0 && v < V) { tv[v] = tanhf(s + b0); da[v] = d; }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int v = 0; v < V; ++v) mx = fmaxf(mx, tv[v]);
  float den = 0.f;
  for (int v = 0; v < V; ++v) den += expf(tv[v] - mx);
  const float inv = 1.0f / den;
  float mean_da = 0.f;
  for (int v = 0; v < V; ++v) mean_da += expf(tv[v] - mx) * inv * da[v];
  float4 accw[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) accw[i] = f4(0.f);
  float dbs = 0.f;
#pragma unroll
  for (int r = 0; r < GVPW; ++r) {
    const int v = wave + GWAVES * r;
    if (v >= V) continue;
    const float a = expf(tv[v] - mx) * inv, t = tv[v];
    const float ds = a * (da[v] - mean_da) * (1.f - t * t);
    dbs += ds;
#pragma unroll
    ROWV(i, c) {
      st4(dx + ((size_t)b * V + v) * H + c, a * go[i] + ds * ww[i]);
      accw[i] = accw[i] + ds * row[r][i];
    } ENDV
  }
#pragma unroll
  ROWV(i, c) lds_add4(part + c, accw[i]); ENDV
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 1024) atomicAdd(dw + c, part[c]);
  if (lane == 0 && dbs != 0.f) atomicAdd(db, dbs);
}

// ---- ClsPrediction tail backward: score = (LN(h) * gamma + beta) . w2 + b2
template <bool UNUSED = false>
__global__ __launch_bounds__(256) void head_tail_bwd_kernel(const float* __restrict__ h, int ldh, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ w2,
                                                            const float* __restrict__ dscore, float eps, float* __restrict__ dh, int lddh,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dw2,
                                                            float* __restrict__ db2, int M, int H, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float acc[];       // [3][H]: dgamma, dbeta, dw2
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nv = (H + 255) / 256;
  for (int c = threadIdx.x; c < 3 * H; c += 256) acc[c] = 0.f;
  __syncthreads();
  float4 gm[MAXV], bt[MAXV], wv[MAXV], ag[MAXV], ab[MAXV], aw[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const bool ok = i < nv && c < H;
    gm[i] = ok ? *reinterpret_cast<const float4*>(gamma + c) : f4(0.f);
    bt[i] = ok ? *reinterpret_cast<const float4*>(beta + c) : f4(0.f);
    wv[i] = ok ? *reinterpret_cast<const float4*>(w2 + c) : f4(0.f);
    ag[i] = ab[i] = aw[i] = f4(0.f);
  }
  float sds = 0.f;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  for (int row = m0 + wave; row < m1; row += WPB) {
    float4 zh[MAXV], g[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      zh[i] = (i < nv && c < H) ? ld4(h + (size_t)row * ldh + c) : f4(0.f);
    }
    const float rstd = ln_hat(zh, nv, lane, H, eps);
    const float ds = dscore[row];
    sds += ds;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const float4 dy = ds * wv[i];
      g[i] = dy * gm[i];
      ag[i] = ag[i] + dy * zh[i];
      ab[i] = ab[i] + dy;
      aw[i] = aw[i] + ds * (zh[i] * gm[i] + bt[i]);
    }
    ln_dz(g, zh, rstd, nv, lane, H);
#pragma unroll
    ROWV(i, c) st4(dh + (size_t)row * lddh + c, g[i]); ENDV
  }
#pragma unroll
  ROWV(i, c) { lds_add4(acc + c, ag[i]); lds_add4(acc + H + c, ab[i]); lds_add4(acc + 2 * H + c, aw[i]); } ENDV
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    atomicAdd(dgamma + c, acc[c]); atomicAdd(dbeta + c, acc[H + c]); atomicAdd(dw2 + c, acc[2 * H + c]);
  }
  if (lane == 0 && sds != 0.f) atomicAdd(db2, sds);
}

// ---- dynamic fusion + logit fusion backward (forward: nav_logits_kernel, rowops.hip)
//   gl_j = g_j fw (masked -> -inf),  ll_k = l_k (1 - fw) (masked -> -inf),  fl_j = gl_j + [j = 0] ll_0 + [src_j >= 0] ll_src_j
//   + [src_j = -2] sum_{k: bw_k} ll_k  (j >= 1 for the two src terms).  One wave per episode.
constexpr int NAV_MAX = 1024;
__global__ __launch_bounds__(64) void nav_logits_bwd_kernel(const float* __restrict__ g_raw, const float* __restrict__ l_raw,
                                                            const float* __restrict__ fuse_raw, const uint8_t* __restrict__ gmask,
                                                            const uint8_t* __restrict__ gvis, const uint8_t* __restrict__ vnav,
                                                            const int32_t* __restrict__ gsrc, const uint8_t* __restrict__ bwm,
                                                            const float* __restrict__ dgl, const float* __restrict__ dll,
                                                            const float* __restrict__ dfl, float* __restrict__ dg_raw,
                                                            float* __restrict__ dl_raw, float* __restrict__ dfuse_raw, int G, int Vp) {
  __shared__ float dl_acc[NAV_MAX];
  const int b = blockIdx.x, lane = threadIdx.x;
  const float fw = fuse_raw ? 1.0f / (1.0f + expf(-fuse_raw[b])) : 0.5f;
  for (int k = lane; k < Vp; k += 64) dl_acc[k] = 0.f;
  __syncthreads();
  float bws = 0.f, dfw = 0.f;
  for (int j = lane; j < G; j += 64) {
    const size_t o = (size_t)b * G + j;
    const bool dead = gvis[o] || !gmask[o];
    const float df = dfl ? dfl[o] : 0.f;
    const float dg = dead ? 0.f : (dgl ? dgl[o] : 0.f) + df;      // masked_fill: no gradient through a masked gl
    dg_raw[o] = dg * fw;
    dfw += dg * g_raw[o];
    if (j == 0) atomicAdd(&dl_acc[0], df);
    else {
      const int s = gsrc[o];
      if (s >= 0 && s < Vp) atomicAdd(&dl_acc[s], df);
      else if (s == -2) bws += df;
    }
  }
  bws = wave_sum(bws);
  __syncthreads();
  for (int k = lane; k < Vp; k += 64) {
    const size_t o = (size_t)b * Vp + k;
    const float d = vnav[o] ? (dll ? dll[o] : 0.f) + dl_acc[k] + (bwm[o] ? bws : 0.f) : 0.f;
    dl_raw[o] = d * (1.0f - fw);
    dfw -= d * l_raw[o];
  }
  dfw = wave_sum(dfw);
  if (lane == 0 && dfuse_raw) dfuse_raw[b] = dfw * fw * (1.0f - fw);
}

// ---- IGL gate backward: w = sigmoid(aug.wa + ba + x.wo + bo), out = w aug + (1 - w) x
template <typename T>
__global__ __launch_bounds__(256) void igl_blend_bwd_kernel(const T* __restrict__ aug, const T* __restrict__ x, const float* __restrict__ wa,
                                                            const float* __restrict__ ba, const float* __restrict__ wo,
                                                            const float* __restrict__ bo, const T* __restrict__ dout, T* __restrict__ daug,
                                                            T* __restrict__ dx, float* __restrict__ dwa, float* __restrict__ dwo,
                                                            float* __restrict__ dbias /*[2]: dba, dbo*/, int M, int H, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float acc[];       // [2][H]: dwa, dwo
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nv = (H + 255) / 256;
  for (int c = threadIdx.x; c < 2 * H; c += 256) acc[c] = 0.f;
  __syncthreads();
  float4 a4[MAXV], o4[MAXV], ga[MAXV], go[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const bool ok = i < nv && c < H;
    a4[i] = ok ? *reinterpret_cast<const float4*>(wa + c) : f4(0.f);
    o4[i] = ok ? *reinterpret_cast<const float4*>(wo + c) : f4(0.f);
    ga[i] = go[i] = f4(0.f);
  }
  const float bsum = ba[0] + bo[0];
  float sds = 0.f;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  for (int row = m0 + wave; row < m1; row += WPB) {
    float4 va[MAXV], vx[MAXV], gd[MAXV];
    float s = 0.f, dwg = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      const bool ok = i < nv && c < H;
      va[i] = ok ? ld4(aug + (size_t)row * H + c) : f4(0.f);
      vx[i] = ok ? ld4(x + (size_t)row * H + c) : f4(0.f);
      gd[i] = ok ? ld4(dout + (size_t)row * H + c) : f4(0.f);
      s += dot4(va[i], a4[i]) + dot4(vx[i], o4[i]);
      dwg += dot4(gd[i], va[i] - vx[i]);
    }
    s = wave_sum(s) + bsum;
    const float w = 1.0f / (1.0f + expf(-s));
    const float ds = wave_sum(dwg) * w * (1.f - w);
    sds += ds;
#pragma unroll
    ROWV(i, c) {
      st4(daug + (size_t)row * H + c, w * gd[i] + ds * a4[i]);
      st4(dx + (size_t)row * H + c, (1.f - w) * gd[i] + ds * o4[i]);
      ga[i] = ga[i] + ds * va[i];
      go[i] = go[i] + ds * vx[i];
    } ENDV
  }
#pragma unroll
  ROWV(i, c) { lds_add4(acc + c, ga[i]); lds_add4(acc + H + c, go[i]); } ENDV
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) { atomicAdd(dwa + c, acc[c]); atomicAdd(dwo + c, acc[H + c]); }
  if (lane == 0 && sds != 0.f) { atomicAdd(dbias, sds); atomicAdd(dbias + 1, sds); }
}

// ---- token-embedding sums backward (forward: embed_ln_kernel, rowops.hip)
//   out = LNo( base + LNf(Wf.feat + bf) + E1[i1] + E2[i2] + vec )     (every term optional)
// One wave per row, a strip of rows per workgroup.  A lane owns the same 4*NV columns in every row, so every column-vector
// gradient, dWf (FR = F rounded up to 8 / 16 columns) and the gradient of a table with a handful of rows (nav / token
// types: TS of them) are summed over the wave's rows in REGISTERS.  At the end the four waves' sums meet in a [4][H] LDS
// slab, one vector at a time, with plain stores (no LDS atomics: a wave-wide ds_add_f32 costs ~1000 cycles here - the
// first version spent ~15 of them per element per ROW and took 110-750 us per launch), and leave with one global float
// atomic per element per workgroup.  Rows of large tables get their dz by global atomics (dense fp32 tables).
constexpr int EMB_SMALL = 4;
#define ROWN(i, c) for (int i = 0; i < NV; ++i) { const int c = (i * 64 + lane) * 4; if (c < H)
__device__ __forceinline__ int embed_index_b(const void* p, int kind, int stride, int div, int row) {
  const size_t o = (size_t)(row / div) * stride;
  if (kind == DSRG_IDX_I32) return reinterpret_cast<const int32_t*>(p)[o];
  if (kind == DSRG_IDX_I64) return (int)reinterpret_cast<const int64_t*>(p)[o];
  return (int)reinterpret_cast<const float*>(p)[o];
}

// The column-vector gradients a launch produces, in the order the kernel flushes them: where each one goes (dst + c * stride).
constexpr int EMB_MAX_SLOTS = 32;
struct EmbSlots { float* dst[EMB_MAX_SLOTS]; int stride[EMB_MAX_SLOTS]; int n; };

// dst[c * stride] += sum_b partials[b][slot][c]: 32 columns x 32 row groups per workgroup, LDS tree, fixed order
__global__ __launch_bounds__(1024) void embed_bwd_reduce_kernel(const float* __restrict__ partials, int nblk, int H, const EmbSlots sl) {
  __shared__ float red[32][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int per = H / 32;                           // column groups per slot (H % 64 == 0)
  const int slot = blockIdx.x / per, c = (blockIdx.x - slot * per) * 32 + cl;
  const size_t rs = (size_t)sl.n * H;
  const float* src = partials + (size_t)slot * H + c;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int b = rg;
  for (; b + 96 < nblk; b += 128) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] += src[(size_t)(b + 32 * u) * rs];
  }
  for (; b < nblk; b += 32) a[0] += src[(size_t)b * rs];
  red[rg][cl] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    if (rg < o) red[rg][cl] += red[rg + o][cl];
    __syncthreads();
  }
  if (rg == 0) sl.dst[slot][(size_t)c * sl.stride[slot]] += red[0][cl];
}

template <typename T, int NV, int FR, int TS>
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(const dsrg_embed_bwd_args_t a, int rows_per_block, int nslots) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.H, F = a.F, nv = NV, HS = H + 4;
  float* slab = lds;                                 // [WPB][H] cross-wave reduction slab
  float* WT = lds + WPB * H;                         // [F][HS] forward weight, transposed
  const bool small1 = a.dE1 && a.E1_rows <= EMB_SMALL, small2 = a.dE2 && a.E2_rows <= EMB_SMALL;
  constexpr int FA = FR ? FR : 1, TA = TS ? EMB_SMALL : 1, TB = TS > 1 ? EMB_SMALL : 1;
  float4 a_bf[NV], a_lfg[NV], a_lfb[NV], a_vec[NV], a_log[NV], a_lob[NV], a_w[FA][NV], a_t1[TA][NV], a_t2[TB][NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    a_bf[i] = a_lfg[i] = a_lfb[i] = a_vec[i] = a_log[i] = a_lob[i] = f4(0.f);
#pragma unroll
    for (int k = 0; k < FA; ++k) a_w[k][i] = f4(0.f);
#pragma unroll
    for (int r = 0; r < TA; ++r) a_t1[r][i] = f4(0.f);
#pragma unroll
    for (int r = 0; r < TB; ++r) a_t2[r][i] = f4(0.f);
  }
  if (FR && a.feat) {
    stage_wt<256>(WT, a.Wf, H, F, HS, threadIdx.x);
    __syncthreads();
  }
  const T* base = reinterpret_cast<const T*>(a.base);
  const T* dout = reinterpret_cast<const T*>(a.dout);
  T* dbase = reinterpret_cast<T*>(a.dbase);
  const int m0 = blockIdx.x * rows_per_block, m1 = min(a.M, m0 + rows_per_block);
  float* dE1w = a.dE1 + (a.E1_copies > 1 ? (size_t)(blockIdx.x % a.E1_copies) * a.E1_rows * H : 0);   // this workgroup's copy
  for (int row = m0 + wave; row < m1; row += WPB) {
    // the row's features: ONE load (lane k holds feature k), then broadcasts - F conditional loads in a row were F memory
    // latencies in a row
    const float fl = (FR && a.feat) ? a.feat[(size_t)row * a.ldf + min(lane, F - 1)] : 0.f;
    const int i1 = a.E1 ? embed_index_b(a.idx1, a.idx1_kind, a.idx1_stride, a.idx1_div, row) : 0;
    const int i2 = a.E2 ? embed_index_b(a.idx2, a.idx2_kind, a.idx2_stride, a.idx2_div, row) : 0;
    float4 zf[MAXV], z[MAXV], g[MAXV];
    float rstd_f = 1.f, rstd_o = 1.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) zf[i] = z[i] = g[i] = f4(0.f);
    // every global load of the row first
#pragma unroll
    ROWN(i, c) {
      g[i] = ld4(dout + (size_t)row * a.lddo + c);
      if (base) z[i] = ld4(base + (size_t)row * a.ldb + c);
    } ENDV
    if (a.lno_gamma) {                                   // the forward sum is only needed to redo the output LayerNorm
      if (a.E1) {
#pragma unroll
        ROWN(i, c) z[i] = z[i] + *reinterpret_cast<const float4*>(a.E1 + (size_t)i1 * H + c); ENDV
      }
      if (a.E2) {
#pragma unroll
        ROWN(i, c) z[i] = z[i] + *reinterpret_cast<const float4*>(a.E2 + (size_t)i2 * H + c); ENDV
      }
      if (a.vec) {
#pragma unroll
        ROWN(i, c) z[i] = z[i] + *reinterpret_cast<const float4*>(a.vec + c); ENDV
      }
    }
    float f[FA];
#pragma unroll
    for (int k = 0; k < FA; ++k) f[k] = k < F ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fl), k)) : 0.f;
    if (FR && a.feat) {
#pragma unroll
      ROWN(i, c) {
        float4 o = a.bf ? *reinterpret_cast<const float4*>(a.bf + c) : f4(0.f);
#pragma unroll
        for (int k = 0; k < FA; ++k)
          if (k < F) o = o + f[k] * *reinterpret_cast<const float4*>(&WT[k * HS + c]);
        zf[i] = o;
      } ENDV
      if (a.lnf_gamma) {
        rstd_f = ln_hat(zf, nv, lane, H, a.eps);
#pragma unroll
        ROWN(i, c) z[i] = z[i] + zf[i] * *reinterpret_cast<const float4*>(a.lnf_gamma + c) + *reinterpret_cast<const float4*>(a.lnf_beta + c); ENDV
      } else {
#pragma unroll
        ROWN(i, c) z[i] = z[i] + zf[i]; ENDV
      }
    }
    if (a.lno_gamma) {
      rstd_o = ln_hat(z, nv, lane, H, a.eps);            // z <- normalised row
#pragma unroll
      ROWN(i, c) {
        a_log[i] = a_log[i] + g[i] * z[i];
        a_lob[i] = a_lob[i] + g[i];
        g[i] = g[i] * *reinterpret_cast<const float4*>(a.lno_gamma + c);
      } ENDV
      ln_dz(g, z, rstd_o, nv, lane, H);
    }
    // g = dz: gradient of the pre-LayerNorm sum
#pragma unroll
    ROWN(i, c) {
      if (dbase) st4(dbase + (size_t)row * a.lddb + c, g[i]);
      if (a.dE1 && !small1 && i1 != a.E1_padding_idx) lds_add4(dE1w + (size_t)i1 * H + c, g[i]);     // global float atomics
      if (a.dE2 && !small2) lds_add4(a.dE2 + (size_t)i2 * H + c, g[i]);
    } ENDV
    if constexpr (TS > 0) {                              // the row's table index is wave-uniform: a scalar branch per slot
      if (small1) {
#pragma unroll
        for (int r = 0; r < TA; ++r)
          if (i1 == r) {
#pragma unroll
            ROWN(i, c) a_t1[r][i] = a_t1[r][i] + g[i]; ENDV
          }
      }
    }
    if constexpr (TS > 1) {
      if (small2) {
#pragma unroll
        for (int r = 0; r < TB; ++r)
          if (i2 == r) {
#pragma unroll
            ROWN(i, c) a_t2[r][i] = a_t2[r][i] + g[i]; ENDV
          }
      }
    }
    if (a.vec) {
#pragma unroll
      ROWN(i, c) a_vec[i] = a_vec[i] + g[i]; ENDV
    }
    if (FR && a.feat) {
      if (a.lnf_gamma) {
#pragma unroll
        ROWN(i, c) {
          a_lfg[i] = a_lfg[i] + g[i] * zf[i];
          a_lfb[i] = a_lfb[i] + g[i];
          g[i] = g[i] * *reinterpret_cast<const float4*>(a.lnf_gamma + c);
        } ENDV
        ln_dz(g, zf, rstd_f, nv, lane, H);
      }
#pragma unroll
      ROWN(i, c) {
        a_bf[i] = a_bf[i] + g[i];
#pragma unroll
        for (int k = 0; k < FA; ++k)
          if (k < F) a_w[k][i] = a_w[k][i] + f[k] * g[i];
      } ENDV
    }
  }
  // the four waves' register sums -> one global atomic per element per workgroup.  `on` and the loop bounds are
  // workgroup-uniform, so every thread reaches the barriers.
  int slot = 0;
  auto flush = [&](const float4 (&acc)[NV], float* dst, int dst_stride, bool on) {
    if (!on) return;
    // LDS-only barriers: __syncthreads() also waits for the previous vector's global atomics (a ~2.5 us round trip per
    // flush, 10-20 flushes per workgroup)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    ROWN(i, c) *reinterpret_cast<float4*>(slab + wave * H + c) = acc[i]; ENDV
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (a.partials) {        // this workgroup's row of the partial table; embed_bwd_reduce_kernel adds the table onto dst
      float* prow = a.partials + ((size_t)blockIdx.x * nslots + slot) * H;
      for (int c = threadIdx.x; c < H; c += 256) prow[c] = (slab[c] + slab[H + c]) + (slab[2 * H + c] + slab[3 * H + c]);
      ++slot;
      return;
    }
    for (int c = threadIdx.x; c < H; c += 256)
      atomicAdd(dst + (size_t)c * dst_stride, (slab[c] + slab[H + c]) + (slab[2 * H + c] + slab[3 * H + c]));
  };
  if (FR && a.feat) {
#pragma unroll
    for (int k = 0; k < FA; ++k) flush(a_w[k], a.dWf + k, F, k < F);
    flush(a_bf, a.dbf, 1, a.dbf != nullptr);
    flush(a_lfg, a.dlnf_gamma, 1, a.lnf_gamma != nullptr);
    flush(a_lfb, a.dlnf_beta, 1, a.lnf_gamma != nullptr);
  }
  if constexpr (TS > 0) {
#pragma unroll
    for (int r = 0; r < TA; ++r) flush(a_t1[r], a.dE1 + (size_t)r * H, 1, small1 && r < a.E1_rows);
  }
  if constexpr (TS > 1) {
#pragma unroll
    for (int r = 0; r < TB; ++r) flush(a_t2[r], a.dE2 + (size_t)r * H, 1, small2 && r < a.E2_rows);
  }
  flush(a_vec, a.dvec, 1, a.vec != nullptr);
  flush(a_log, a.dlno_gamma, 1, a.lno_gamma != nullptr);
  flush(a_lob, a.dlno_beta, 1, a.lno_gamma != nullptr);
}
#undef ROWN

// ---- segment-weighted row gather (pre-training trajectory aggregation, PT/model/vilmodel.py:557-571, 597-635):
//   out[r, :] = sum_{e in [seg_ptr[r], seg_ptr[r+1])} w[e] * in[idx[e], :]
// The inputs are two row tables addressed as one (idx < n0 -> in0, else in1[idx - n0]) and so are the outputs
// (r < m0 -> out0, else out1[r - m0]): the forward gathers panorama-view rows and GAA-pooled rows into graph-map slots,
// its backward is the same kernel on the transposed incidence (built on the host), so no atomics anywhere.
template <typename T>
__global__ __launch_bounds__(256) void segment_wsum_kernel(const T* __restrict__ in0, int ld0, int n0, const T* __restrict__ in1, int ld1,
                                                           T* __restrict__ out0, int ldo0, int m0, T* __restrict__ out1, int ldo1, int m1,
                                                           const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ idx,
                                                           const float* __restrict__ w, int H) {
  const int lane = threadIdx.x & 63, nv = (H + 255) / 256;
  const int r = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (r >= m0 + m1) return;
  float4 acc[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) acc[i] = f4(0.f);
  for (int e = seg_ptr[r]; e < seg_ptr[r + 1]; ++e) {
    const int j = idx[e];
    const float we = w[e];
    const T* row = j < n0 ? in0 + (size_t)j * ld0 : in1 + (size_t)(j - n0) * ld1;
#pragma unroll
    ROWV(i, c) acc[i] = acc[i] + we * ld4(row + c); ENDV
  }
  T* o = r < m0 ? out0 + (size_t)r * ldo0 : out1 + (size_t)(r - m0) * ldo1;
#pragma unroll
  ROWV(i, c) st4(o + c, acc[i]); ENDV
}

inline bool okdt(int d) { return d == DSRG_F32 || d == DSRG_BF16; }
inline bool al(const void* p, uintptr_t m) { return ((uintptr_t)p & m) == 0; }
inline int strip(int M) { int r = (M + 511) / 512; return r < WPB ? WPB : (r + WPB - 1) / WPB * WPB; }   // <= ~512 workgroups

}  // namespace

extern "C" int dsrg_group_mean_bwd(const void* dy, int lddy, void* dx, int lddx, int M, int G, int H, int dtype, void* stream) {
  DSRG_CHECK(dy && dx && okdt(dtype) && M >= 0 && G >= 1 && H > 0 && H % 4 == 0 && lddy >= H && lddx >= H && lddy % 4 == 0 && lddx % 4 == 0,
             "dsrg_group_mean_bwd: bad arguments");
  DSRG_CHECK(al(dy, dtype == DSRG_BF16 ? 7 : 15) && al(dx, dtype == DSRG_BF16 ? 7 : 15), "dsrg_group_mean_bwd: misaligned pointer");
  if (M == 0) return DSRG_OK;
  const long n = (long)M * G * (H / 4);
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (dtype == DSRG_F32) hipLaunchKernelGGL((group_mean_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)dy, lddy, (float*)dx, lddx, M, G, H);
  else hipLaunchKernelGGL((group_mean_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, M, G, H);
  DSRG_LAUNCH_CHECK("dsrg_group_mean_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_gaa_pool_bwd(const void* x, const float* w, const float* bias, const void* dout, void* dx, float* dw, float* db,
                                 int B, int V, int H, int dtype, void* stream) {
  DSRG_CHECK(okdt(dtype) && x && w && bias && dout && dx && dw && db, "dsrg_gaa_pool_bwd: bad dtype or null pointer");
  DSRG_CHECK(V >= 1 && V <= 64 && H > 0 && H % 4 == 0 && H <= 1024, "dsrg_gaa_pool_bwd: need 1 <= V <= 64 (got %d), H %% 4 == 0 and H <= 1024", V);
  DSRG_CHECK(al(x, 15) && al(dout, 15) && al(dx, 15) && al(w, 15), "dsrg_gaa_pool_bwd: misaligned pointer");
  if (B <= 0) return DSRG_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DSRG_F32)
    hipLaunchKernelGGL((gaa_pool_bwd_kernel<float>), dim3(B), dim3(1024), (size_t)H * 4, s, (const float*)x, w, bias, (const float*)dout, (float*)dx, dw, db, V, H);
  else
    hipLaunchKernelGGL((gaa_pool_bwd_kernel<bf16_t>), dim3(B), dim3(1024), (size_t)H * 4, s, (const bf16_t*)x, w, bias, (const bf16_t*)dout, (bf16_t*)dx, dw, db, V, H);
  DSRG_LAUNCH_CHECK("dsrg_gaa_pool_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_head_tail_bwd(const float* h, int ldh, const float* gamma, const float* beta, const float* w2, const float* dscore,
                                  float eps, float* dh, int lddh, float* dgamma, float* dbeta, float* dw2, float* db2, int M, int H,
                                  void* stream) {
  DSRG_CHECK(h && gamma && beta && w2 && dscore && dh && dgamma && dbeta && dw2 && db2, "dsrg_head_tail_bwd: null pointer");
  DSRG_CHECK(M >= 0 && H > 0 && H <= 1024 && H % 64 == 0 && ldh >= H && lddh >= H && ldh % 4 == 0 && lddh % 4 == 0, "dsrg_head_tail_bwd: bad dims");
  DSRG_CHECK(al(h, 15) && al(dh, 15) && al(gamma, 15) && al(beta, 15) && al(w2, 15), "dsrg_head_tail_bwd: misaligned pointer");
  if (M == 0) return DSRG_OK;
  const int rpb = strip(M);
  hipLaunchKernelGGL((head_tail_bwd_kernel<false>), dim3((M + rpb - 1) / rpb), dim3(256), (size_t)3 * H * 4, (hipStream_t)stream, h, ldh, gamma, beta,
                     w2, dscore, eps, dh, lddh, dgamma, dbeta, dw2, db2, M, H, rpb);
  DSRG_LAUNCH_CHECK("dsrg_head_tail_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_nav_logits_bwd(const float* g_raw, const float* l_raw, const float* fuse_raw, const uint8_t* gmap_masks,
                                   const uint8_t* gmap_visited, const uint8_t* vp_nav_masks, const int32_t* gmap_src,
                                   const uint8_t* bw_mask, const float* dglobal, const float* dlocal, const float* dfused,
                                   float* dg_raw, float* dl_raw, float* dfuse_raw, int B, int G, int Vp, void* stream) {
  DSRG_CHECK(g_raw && l_raw && gmap_masks && gmap_visited && vp_nav_masks && gmap_src && bw_mask && dg_raw && dl_raw,
             "dsrg_nav_logits_bwd: null pointer");
  DSRG_CHECK(B >= 0 && G >= 1 && Vp >= 1 && Vp <= NAV_MAX, "dsrg_nav_logits_bwd: bad dims (Vp <= %d)", NAV_MAX);
  DSRG_CHECK(!fuse_raw == !dfuse_raw, "dsrg_nav_logits_bwd: fuse_raw and dfuse_raw go together");
  if (B == 0) return DSRG_OK;
  hipLaunchKernelGGL(nav_logits_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, g_raw, l_raw, fuse_raw, gmap_masks, gmap_visited,
                     vp_nav_masks, gmap_src, bw_mask, dglobal, dlocal, dfused, dg_raw, dl_raw, dfuse_raw, G, Vp);
  DSRG_LAUNCH_CHECK("dsrg_nav_logits_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_igl_blend_bwd(const void* aug, const void* x, const float* wa, const float* ba, const float* wo, const float* bo,
                                  const void* dout, void* daug, void* dx, float* dwa, float* dwo, float* dbias, int M, int H, int dtype,
                                  void* stream) {
  DSRG_CHECK(okdt(dtype) && aug && x && wa && ba && wo && bo && dout && daug && dx && dwa && dwo && dbias, "dsrg_igl_blend_bwd: bad dtype or null pointer");
  DSRG_CHECK(M >= 0 && H > 0 && H <= 1024 && H % 64 == 0, "dsrg_igl_blend_bwd: bad dims");
  const uintptr_t m = dtype == DSRG_BF16 ? 7 : 15;
  DSRG_CHECK(al(aug, m) && al(x, m) && al(dout, m) && al(daug, m) && al(dx, m) && al(wa, 15) && al(wo, 15), "dsrg_igl_blend_bwd: misaligned pointer");
  if (M == 0) return DSRG_OK;
  const int rpb = strip(M);
  dim3 grid((M + rpb - 1) / rpb), block(256);
  if (dtype == DSRG_F32)
    hipLaunchKernelGGL((igl_blend_bwd_kernel<float>), grid, block, (size_t)2 * H * 4, (hipStream_t)stream, (const float*)aug, (const float*)x, wa, ba, wo,
                       bo, (const float*)dout, (float*)daug, (float*)dx, dwa, dwo, dbias, M, H, rpb);
  else
    hipLaunchKernelGGL((igl_blend_bwd_kernel<bf16_t>), grid, block, (size_t)2 * H * 4, (hipStream_t)stream, (const bf16_t*)aug, (const bf16_t*)x, wa, ba,
                       wo, bo, (const bf16_t*)dout, (bf16_t*)daug, (bf16_t*)dx, dwa, dwo, dbias, M, H, rpb);
  DSRG_LAUNCH_CHECK("dsrg_igl_blend_bwd");
  return DSRG_OK;
}

// ~256 workgroups: with the column sums in registers a longer strip per wave means fewer flushes at the end, a shorter one
// more rows in flight (measured best for 4932 rows with the atomic flush: 89 us, against 111 us with ~512 and 98 us with ~128)
static int embed_bwd_rows_per_block(const dsrg_embed_bwd_args_t& a) {
  // (with the partial table instead of atomics, twice the workgroups: no difference, 10.42 vs 10.38 ms per pre-training step)
  const int wgs = a.M >= 4096 ? 256 : a.M >= 1024 ? 128 : 64;
  const int rpb0 = ((a.M + wgs - 1) / wgs + WPB - 1) / WPB * WPB;
  return rpb0 < WPB ? WPB : rpb0;
}
// the kernel's flush order, mirrored: dWf columns, dbf, feature-LayerNorm gain / offset, rows of the small tables, the
// constant vector, output-LayerNorm gain / offset
static EmbSlots embed_bwd_slots(const dsrg_embed_bwd_args_t& a) {
  EmbSlots s;
  s.n = 0;
  auto add = [&](float* dst, int stride) { if (s.n < EMB_MAX_SLOTS) { s.dst[s.n] = dst; s.stride[s.n] = stride; } ++s.n; };
  const bool small1 = a.dE1 && a.E1_rows <= EMB_SMALL, small2 = a.dE2 && a.E2_rows <= EMB_SMALL;
  if (a.feat) {
    for (int k = 0; k < a.F; ++k) add(a.dWf + k, a.F);
    if (a.dbf) add(a.dbf, 1);
    if (a.lnf_gamma) { add(a.dlnf_gamma, 1); add(a.dlnf_beta, 1); }
  }
  if (small1) for (int r = 0; r < a.E1_rows; ++r) add(a.dE1 + (size_t)r * a.H, 1);
  if (small2) for (int r = 0; r < a.E2_rows; ++r) add(a.dE2 + (size_t)r * a.H, 1);
  if (a.vec) add(a.dvec, 1);
  if (a.lno_gamma) { add(a.dlno_gamma, 1); add(a.dlno_beta, 1); }
  for (int i = s.n; i < EMB_MAX_SLOTS; ++i) { s.dst[i] = nullptr; s.stride[i] = 0; }
  return s;
}

extern "C" int64_t dsrg_embed_ln_bwd_workspace(const dsrg_embed_bwd_args_t* args) {
  if (!args || args->M <= 0 || args->H <= 0) return 0;
  dsrg_embed_bwd_args_t b = *args;
  b.partials = reinterpret_cast<float*>(sizeof(float));      // the strip length depends on whether a workspace is given: size it for "given"
  const int rpb = embed_bwd_rows_per_block(b);
  const EmbSlots s = embed_bwd_slots(*args);
  if (s.n > EMB_MAX_SLOTS) return 0;              // (cannot happen with F <= 16 and tables <= 4 rows: 16 + 3 + 4 + 4 + 3 = 30)
  return (int64_t)((args->M + rpb - 1) / rpb) * s.n * args->H;
}

extern "C" int dsrg_embed_ln_bwd(const dsrg_embed_bwd_args_t* args, void* stream) {
  DSRG_CHECK(args, "dsrg_embed_ln_bwd: null args");
  const dsrg_embed_bwd_args_t& a = *args;
  DSRG_CHECK(okdt(a.dtype) && a.dout && a.M >= 0, "dsrg_embed_ln_bwd: bad dtype or null dout");
  DSRG_CHECK(a.H > 0 && a.H <= 1024 && a.H % 64 == 0, "dsrg_embed_ln_bwd: H=%d must be a multiple of 64 and <= 1024", a.H);
  DSRG_CHECK(!a.feat || (a.Wf && a.dWf && a.F >= 1 && a.F <= 16 && a.ldf >= a.F && (!a.bf == !a.dbf)), "dsrg_embed_ln_bwd: feature term needs Wf, dWf, 1 <= F <= 16");
  DSRG_CHECK(!a.feat || ((uintptr_t)a.Wf & 15) == 0, "dsrg_embed_ln_bwd: Wf must be 16-byte aligned");
  DSRG_CHECK((!a.lnf_gamma || (a.feat && a.lnf_beta && a.dlnf_gamma && a.dlnf_beta)) && (!a.lno_gamma || (a.lno_beta && a.dlno_gamma && a.dlno_beta)),
             "dsrg_embed_ln_bwd: LayerNorm operands / gradients incomplete");
  DSRG_CHECK((!a.E1 || a.idx1) && (!a.E2 || a.idx2) && (!a.dE1 || a.E1) && (!a.dE2 || a.E2) && (!a.vec == !a.dvec), "dsrg_embed_ln_bwd: table / index / gradient mismatch");
  DSRG_CHECK((!a.E1 || (a.idx1_kind >= DSRG_IDX_I32 && a.idx1_kind <= DSRG_IDX_F32 && a.idx1_stride >= 1 && a.idx1_div >= 1)) &&
                 (!a.E2 || (a.idx2_kind >= DSRG_IDX_I32 && a.idx2_kind <= DSRG_IDX_F32 && a.idx2_stride >= 1 && a.idx2_div >= 1)),
             "dsrg_embed_ln_bwd: bad index kind / stride / div");
  DSRG_CHECK(a.lddo >= a.H && a.lddo % 4 == 0 && (!a.base || (a.ldb >= a.H && a.ldb % 4 == 0)) && (!a.dbase || (a.lddb >= a.H && a.lddb % 4 == 0)),
             "dsrg_embed_ln_bwd: bad leading dims");
  const uintptr_t m = a.dtype == DSRG_BF16 ? 7 : 15;
  DSRG_CHECK(al(a.dout, m) && al(a.base, m) && al(a.dbase, m) && al(a.E1, 15) && al(a.E2, 15) && al(a.vec, 15) && al(a.bf, 15) && al(a.lnf_gamma, 15) &&
                 al(a.lnf_beta, 15) && al(a.lno_gamma, 15) && al(a.lno_beta, 15), "dsrg_embed_ln_bwd: misaligned pointer");
  if (a.M == 0) return DSRG_OK;
  // ~256 workgroups: with the column sums in registers a longer strip per wave means fewer LDS / global atomics
  // ~256 workgroups (measured best for 4932 rows: 89 us, against 111 us with ~512 and 98 us with ~128): a longer strip per
  // wave means fewer same-address global atomics at the end, a shorter one more rows in flight
  // (inside a pre-training step the 888-row viewpoint launch took 180 us with 222 workgroups of one row per wave: ~23
  // flush vectors x 222 same-address atomics each; hence fewer, longer strips for the small launches)
  const int rpb = embed_bwd_rows_per_block(a);
  DSRG_CHECK((!a.dE1 || a.E1_rows >= 1) && (!a.dE2 || a.E2_rows >= 1), "dsrg_embed_ln_bwd: table row counts missing");
  DSRG_CHECK(a.E1_copies >= 0 && a.E1_copies <= 1024 && (a.E1_copies <= 1 || (a.dE1 && a.E1_rows > EMB_SMALL)),
             "dsrg_embed_ln_bwd: E1_copies=%d needs dE1 and a table of more than %d rows", a.E1_copies, EMB_SMALL);
  const bool small1 = a.dE1 && a.E1_rows <= EMB_SMALL, small2 = a.dE2 && a.E2_rows <= EMB_SMALL;
  const int ts = small2 ? 2 : small1 ? 1 : 0, fr = !a.feat ? 0 : a.F <= 8 ? 8 : 16;
  const size_t lds = ((size_t)WPB * a.H + (a.feat ? (size_t)a.F * (a.H + 4) : 0)) * sizeof(float);
  dim3 grid((a.M + rpb - 1) / rpb), block(256);
  hipStream_t st = (hipStream_t)stream;
  const EmbSlots sl = embed_bwd_slots(a);
  DSRG_CHECK(!a.partials || (al(a.partials, 15) && a.H % 64 == 0), "dsrg_embed_ln_bwd: partials needs 16-byte alignment and H %% 64 == 0");
#define EMB_BWD(T, NV, FR, TS)                                                                                             \
  do {                                                                                                                     \
    DSRG_SET_MAX_LDS_ONCE((&embed_ln_bwd_kernel<T, NV, FR, TS>), 96 * 1024);                                               \
    hipLaunchKernelGGL((embed_ln_bwd_kernel<T, NV, FR, TS>), grid, block, lds, st, a, rpb, sl.n);                          \
  } while (0)
#define EMB_TS(T, NV, FR) do { if (ts == 0) EMB_BWD(T, NV, FR, 0); else if (ts == 1) EMB_BWD(T, NV, FR, 1); else EMB_BWD(T, NV, FR, 2); } while (0)
#define EMB_FR(T, NV) do { if (fr == 0) EMB_TS(T, NV, 0); else if (fr == 8) EMB_TS(T, NV, 8); else EMB_TS(T, NV, 16); } while (0)
  if (a.H <= 768) { if (a.dtype == DSRG_F32) EMB_FR(float, 3); else EMB_FR(bf16_t, 3); }
  else { if (a.dtype == DSRG_F32) EMB_FR(float, 4); else EMB_FR(bf16_t, 4); }
#undef EMB_FR
#undef EMB_TS
#undef EMB_BWD
  if (a.partials && sl.n > 0)
    hipLaunchKernelGGL(embed_bwd_reduce_kernel, dim3(sl.n * (a.H / 32)), dim3(1024), 0, st, a.partials, (int)grid.x, a.H, sl);
  DSRG_LAUNCH_CHECK("dsrg_embed_ln_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_segment_wsum(const void* in0, int ld0, int n0, const void* in1, int ld1, int n1, void* out0, int ldo0, int m0,
                                 void* out1, int ldo1, int m1, const int32_t* seg_ptr, const int32_t* idx, const float* w, int H,
                                 int dtype, void* stream) {
  DSRG_CHECK(okdt(dtype) && seg_ptr && n0 >= 0 && n1 >= 0 && m0 >= 0 && m1 >= 0, "dsrg_segment_wsum: bad dtype / null seg_ptr / negative sizes");
  DSRG_CHECK((n0 == 0 || in0) && (n1 == 0 || in1) && (m0 == 0 || out0) && (m1 == 0 || out1), "dsrg_segment_wsum: null table");
  DSRG_CHECK(H > 0 && H <= 1024 && H % 4 == 0, "dsrg_segment_wsum: H=%d must be a multiple of 4 and <= 1024", H);
  DSRG_CHECK((n0 == 0 || (ld0 >= H && ld0 % 4 == 0)) && (n1 == 0 || (ld1 >= H && ld1 % 4 == 0)) && (m0 == 0 || (ldo0 >= H && ldo0 % 4 == 0)) &&
                 (m1 == 0 || (ldo1 >= H && ldo1 % 4 == 0)), "dsrg_segment_wsum: bad leading dims");
  const uintptr_t m = dtype == DSRG_BF16 ? 7 : 15;
  DSRG_CHECK(al(in0, m) && al(in1, m) && al(out0, m) && al(out1, m), "dsrg_segment_wsum: misaligned pointer");
  if (m0 + m1 == 0) return DSRG_OK;
  DSRG_CHECK(idx && w, "dsrg_segment_wsum: null idx / w");
  dim3 grid((m0 + m1 + WPB - 1) / WPB), block(256);
  if (dtype == DSRG_F32)
    hipLaunchKernelGGL((segment_wsum_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)in0, ld0, n0, (const float*)in1, ld1, (float*)out0,
                       ldo0, m0, (float*)out1, ldo1, m1, seg_ptr, idx, w, H);
  else
    hipLaunchKernelGGL((segment_wsum_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)in0, ld0, n0, (const bf16_t*)in1, ld1,
                       (bf16_t*)out0, ldo0, m0, (bf16_t*)out1, ldo1, m1, seg_ptr, idx, w, H);
  DSRG_LAUNCH_CHECK("dsrg_segment_wsum");
  return DSRG_OK;
}
