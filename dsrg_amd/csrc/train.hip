// Backward-pass kernels.  The GEMM gradients reuse the forward NT kernel (gemm.hip):
//   dX = dY · W          = gemm_nt(dY [M,N], Wt [K,N])          Wt  = transpose(W)
//   dW = dYᵀ · X         = gemm_nt(dYt [N,Mp], Xt [K,Mp])        dYt = transpose(dY), Xt = transpose(X), M zero-padded to 64
// so the only extra data movement is the (HBM-bound) transpose below; db is a column sum.
// LayerNorm / activation backward are row / element kernels with fp32 statistics.
#include "common.h"

namespace {

// out[c][r] = in[r][c]; out rows are zero-filled up to Rp >= R.  64x64 tiles through LDS (+1 pad: conflict-free).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const TI* __restrict__ in, int ld_in, TO* __restrict__ out, int ld_out,
                                                            int R, int C, int Rp, float* __restrict__ colsum) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (r < R && c < C) ? Elem<TI>::ld(in + (size_t)r * ld_in + c) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < C && r < Rp) Elem<TO>::st(out + (size_t)c * ld_out + r, tile[tx][ty + 4 * i]);
  }
  if (colsum) {        // the bias gradient rides along: column sums of this 64-row strip, one atomic per column
    __shared__ float part[4][64];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += tile[ty + 4 * i][tx];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c0 + tx < C) atomicAdd(colsum + c0 + tx, part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]);
  }
}

// ---- grouped form: every operand transpose of one backward call in a single launch (1-D grid of 64x64 tiles)
struct TransposeGroup {
  dsrg_transpose_problem_t p[DSRG_MAX_TRANSPOSES];
  int tile_start[DSRG_MAX_TRANSPOSES + 1];
  int n;
};
__device__ __forceinline__ float ld_any(const void* p, size_t i, int dt) {
  return dt == DSRG_F32 ? reinterpret_cast<const float*>(p)[i] : Elem<bf16_t>::ld(reinterpret_cast<const bf16_t*>(p) + i);
}
__device__ __forceinline__ void st_any(void* p, size_t i, int dt, float v) {
  if (dt == DSRG_F32) reinterpret_cast<float*>(p)[i] = v;
  else Elem<bf16_t>::st(reinterpret_cast<bf16_t*>(p) + i, v);
}
__global__ __launch_bounds__(256) void transpose_pad_grouped_kernel(const TransposeGroup grp) {
  __shared__ float tile[64][65];
  __shared__ float part[4][64];
  int g = 0;
  for (int i = 1; i < grp.n; ++i)
    if ((int)blockIdx.x >= grp.tile_start[i]) g = i;
  const dsrg_transpose_problem_t& P = grp.p[g];
  const int t = blockIdx.x - grp.tile_start[g];
  const int tc = (P.C + 63) / 64;
  const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int R = P.R, C = P.C, Rp = P.Rp;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (r < R && c < C) ? ld_any(P.in, (size_t)r * P.ld_in + c, P.in_dtype) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < C && r < Rp) st_any(P.out, (size_t)c * P.ld_out + r, P.out_dtype, tile[tx][ty + 4 * i]);
  }
  if (P.colsum) {                                  // uniform per workgroup
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += tile[ty + 4 * i][tx];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c0 + tx < C) atomicAdd(P.colsum + c0 + tx, part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]);
  }
}

// out[n] += sum over a strip of rows of in[m][n]   (out pre-zeroed; one atomic per column per strip)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, int ld, float* __restrict__ out, int M, int N, int rows_per_block) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float s = 0.f;
  if (c < N)
    for (int m = m0 + w; m < m1; m += 4) s += Elem<T>::ld(in + (size_t)m * ld + c);
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < N) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

__device__ __forceinline__ float act_fwd(float z, int act) {
  switch (act) {
    case DSRG_ACT_GELU: return gelu_erf(z);
    case DSRG_ACT_RELU: return fmaxf(z, 0.f);
    case DSRG_ACT_TANH: return tanhf(z);
    default: return z;
  }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void act_kernel(const T* __restrict__ z, const T* __restrict__ dy, T* __restrict__ out, size_t n4, int act) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 a = ld4(z + i * 4);
    float4 r;
    if (BWD) {
      const float4 g = ld4(dy + i * 4);
      r = make_float4(g.x * act_grad(a.x, act), g.y * act_grad(a.y, act), g.z * act_grad(a.z, act), g.w * act_grad(a.w, act));
    } else {
      r = make_float4(act_fwd(a.x, act), act_fwd(a.y, act), act_fwd(a.z, act), act_fwd(a.w, act));
    }
    st4(out + i * 4, r);
  }
}

// out[i] = sum_s in[s * stride + i] (fixed order): the fix-up of a K-split wgrad whose S problems left fp32 partial tiles
template <typename OutT>
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ in, int S, size_t stride, OutT* __restrict__ out, size_t n) {
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 a = *reinterpret_cast<const float4*>(in + i * 4);
    for (int s = 1; s < S; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(in + (size_t)s * stride + i * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    st4(out + i * 4, a);
  }
  if (blockIdx.x == 0) {
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      float a = in[i];
      for (int s = 1; s < S; ++s) a += in[(size_t)s * stride + i];
      Elem<OutT>::st(out + i, a);
    }
  }
}

// out = (res ? res : 0) + x * keep / (1 - p): nn.Dropout on a hidden state (+ the residual add behind it), the keep decision
// from the counter hash of (seed, element index) like every other dropout of the library; the backward is the same kernel
// on dy (res = NULL).
template <typename T>
__global__ __launch_bounds__(256) void dropout_add_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ out, size_t n4,
                                                           float p, float inv_keep, unsigned long long seed, const unsigned long long* __restrict__ seed_off) {
  if (seed_off) seed += *seed_off;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 a = ld4(x + i * 4);
    const size_t e = i * 4;
    a.x *= drop_keep(seed, e, p, inv_keep); a.y *= drop_keep(seed, e + 1, p, inv_keep);
    a.z *= drop_keep(seed, e + 2, p, inv_keep); a.w *= drop_keep(seed, e + 3, p, inv_keep);
    if (res) { const float4 r = ld4(res + i * 4); a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
    st4(out + i * 4, a);
  }
}

// LayerNorm backward.  z = x (+ res); zh = (z - mean) * rstd; y = zh*gamma + beta.
//   dz = rstd * (g - mean(g) - zh * mean(g*zh)),  g = dy*gamma;   dgamma += sum_rows dy*zh;  dbeta += sum_rows dy
// One wave per row; a block keeps its dgamma/dbeta partials in registers, then one atomic per column per block.
constexpr int LNB_MAXV = 4;
// DROP: the forward ran  z = dropout(x) + res  (dsrg_layernorm_fwd_grouped_drop): the mask is recomputed from (seed, tag, row,
// col); dz is the gradient of the residual, dzd = dz * keep / (1 - p) the gradient of x (the dense output in front of it).
struct LnDrop { float p, inv_keep; unsigned long long seed; int tag; const unsigned long long* seed_off; };
// v2 (round 3): 8 waves per block, R rows per wave held in registers at once and processed in ONE pass - all of a wave's loads
// are in flight together, then the four row reductions, then the stores - instead of 4 waves walking 4 rows each one after the
// other (one memory round trip per row: 12-17 us per launch for 2-18 MB of traffic).  R = 1 for short inputs (more blocks).
constexpr int LNB_WAVES = 8;
template <typename TX, typename T, bool DROP, int R>
__global__ __launch_bounds__(LNB_WAVES * 64) void layernorm_bwd_kernel(const TX* __restrict__ x, int ldx, const T* __restrict__ res, int ldr,
                                                            const float* __restrict__ gamma, const T* __restrict__ dy, int lddy,
                                                            T* __restrict__ dz, int lddz, float* __restrict__ partials,
                                                            int M, int H, float eps, LnDrop dr, T* __restrict__ dzd, int lddzd) {
  __shared__ float red[LNB_WAVES][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = (H + 255) / 256;
  const int row0 = (blockIdx.x * LNB_WAVES + wave) * R;
  float4 ag[LNB_MAXV], ab[LNB_MAXV], gm[LNB_MAXV];
  if constexpr (DROP) {
    if (dr.seed_off) dr.seed += *dr.seed_off;          // (dsrg_set_dropout_offset: the forward added the same word)
  }
#pragma unroll
  for (int i = 0; i < LNB_MAXV; ++i) {
    ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = (i * 64 + lane) * 4;
    gm[i] = (i < nv && c < H) ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 z[R][LNB_MAXV], g[R][LNB_MAXV], rs[R][LNB_MAXV];
  // every load of the wave's rows first
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int row = row0 + j;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      z[j][i] = g[j][i] = rs[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < nv && c < H && row < M) {
        z[j][i] = ld4(x + (size_t)row * ldx + c);
        if (res) rs[j][i] = ld4(res + (size_t)row * ldr + c);
        g[j][i] = ld4(dy + (size_t)row * lddy + c);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int row = row0 + j;
    if (row >= M) continue;                     // wave-uniform
    float4 kp[DROP ? LNB_MAXV : 1];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < H) {
        float4& zz = z[j][i];
        if constexpr (DROP) {
          const size_t e = ((size_t)dr.tag << 40) + (size_t)row * H + c;
          kp[i] = make_float4(drop_keep(dr.seed, e, dr.p, dr.inv_keep), drop_keep(dr.seed, e + 1, dr.p, dr.inv_keep),
                              drop_keep(dr.seed, e + 2, dr.p, dr.inv_keep), drop_keep(dr.seed, e + 3, dr.p, dr.inv_keep));
          zz.x *= kp[i].x; zz.y *= kp[i].y; zz.z *= kp[i].z; zz.w *= kp[i].w;
        }
        zz.x += rs[j][i].x; zz.y += rs[j][i].y; zz.z += rs[j][i].z; zz.w += rs[j][i].w;
        s += zz.x + zz.y + zz.z + zz.w;
      }
    }
    const float mean = wave_sum(s) / (float)H;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < H) {
        float4& zz = z[j][i];
        zz.x -= mean; zz.y -= mean; zz.z -= mean; zz.w -= mean;
        ss += zz.x * zz.x + zz.y * zz.y + zz.z * zz.z + zz.w * zz.w;
      }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
    float sg = 0.f, sgz = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < H) {
        float4& zz = z[j][i];
        float4& gg = g[j][i];
        zz.x *= rstd; zz.y *= rstd; zz.z *= rstd; zz.w *= rstd;          // zh
        ab[i].x += gg.x; ab[i].y += gg.y; ab[i].z += gg.z; ab[i].w += gg.w;
        ag[i].x += gg.x * zz.x; ag[i].y += gg.y * zz.y; ag[i].z += gg.z * zz.z; ag[i].w += gg.w * zz.w;
        gg.x *= gm[i].x; gg.y *= gm[i].y; gg.z *= gm[i].z; gg.w *= gm[i].w;
        sg += gg.x + gg.y + gg.z + gg.w;
        sgz += gg.x * zz.x + gg.y * zz.y + gg.z * zz.z + gg.w * zz.w;
      }
    }
    const float mg = wave_sum(sg) / (float)H, mgz = wave_sum(sgz) / (float)H;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < H) {
        const float4 zz = z[j][i], gg = g[j][i];
        const float4 d = make_float4(rstd * (gg.x - mg - zz.x * mgz), rstd * (gg.y - mg - zz.y * mgz),
                                     rstd * (gg.z - mg - zz.z * mgz), rstd * (gg.w - mg - zz.w * mgz));
        if (dz) st4(dz + (size_t)row * lddz + c, d);
        if constexpr (DROP) st4(dzd + (size_t)row * lddzd + c, make_float4(d.x * kp[i].x, d.y * kp[i].y, d.z * kp[i].z, d.w * kp[i].w));
      }
    }
  }
  // combine the waves' partials (dgamma, then dbeta through the same LDS panel) into this block's row of the partial table
  // [blocks][2][H]; layernorm_bwd_reduce_kernel sums the table.  (Float atomics on dgamma / dbeta - 2H addresses, 48 cache
  // lines - were the whole cost of this kernel: 15-34 us per launch, proportional to the number of blocks.)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < H) *reinterpret_cast<float4*>(&red[wave][c]) = pass ? ab[i] : ag[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += LNB_WAVES * 64) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < LNB_WAVES; ++w) a += red[w][c];
      partials[((size_t)blockIdx.x * 2 + pass) * H + c] = a;
    }
  }
}

// out[c] = sum_b partials[b][c] for the 2H columns (dgamma | dbeta): 32 columns x 32 row groups per block (a thread reads
// nblk / 32 values, all loads independent), LDS tree over the groups; fixed order
constexpr int LNR_COLS = 32, LNR_GROUPS = 32;
__global__ __launch_bounds__(LNR_COLS * LNR_GROUPS) void layernorm_bwd_reduce_kernel(const float* __restrict__ partials, int nblk, int H,
                                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[LNR_GROUPS][LNR_COLS + 1];
  const int cl = threadIdx.x % LNR_COLS, rg = threadIdx.x / LNR_COLS;
  const int c = blockIdx.x * LNR_COLS + cl;            // 2H is a multiple of 64
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int b = rg;
  for (; b + 3 * LNR_GROUPS < nblk; b += 4 * LNR_GROUPS) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] += partials[(size_t)(b + LNR_GROUPS * u) * 2 * H + c];
  }
  for (; b < nblk; b += LNR_GROUPS) a[0] += partials[(size_t)b * 2 * H + c];
  red[rg][cl] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
#pragma unroll
  for (int o = LNR_GROUPS / 2; o > 0; o >>= 1) {
    if (rg < o) red[rg][cl] += red[rg + o][cl];
    __syncthreads();
  }
  if (rg == 0) {
    if (c < H) dgamma[c] = red[0][cl]; else dbeta[c - H] = red[0][cl];
  }
}

inline int lnb_rows_per_wave(int M) { return M <= 8192 ? 1 : 2; }        // one row per wave until the partial table outgrows the rows it saves
inline int lnb_blocks(int M) { const int r = LNB_WAVES * lnb_rows_per_wave(M); return (M + r - 1) / r; }
inline bool okdt(int d) { return d == DSRG_F32 || d == DSRG_BF16; }

}  // namespace

extern "C" int dsrg_transpose_pad(const void* in, int ld_in, void* out, int ld_out, int R, int C, int Rp, int in_dtype, int out_dtype,
                                  float* colsum, void* stream) {
  DSRG_CHECK(in && out && okdt(in_dtype) && okdt(out_dtype), "dsrg_transpose_pad: bad pointer or dtype");
  DSRG_CHECK(R >= 0 && C >= 0 && Rp >= R && ld_in >= C && ld_out >= Rp, "dsrg_transpose_pad: bad dims R=%d C=%d Rp=%d ld_in=%d ld_out=%d", R, C, Rp, ld_in, ld_out);
  if (C == 0 || Rp == 0) return DSRG_OK;
  dim3 grid((C + 63) / 64, (Rp + 63) / 64), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == DSRG_F32 && out_dtype == DSRG_F32) hipLaunchKernelGGL((transpose_pad_kernel<float, float>), grid, block, 0, s, (const float*)in, ld_in, (float*)out, ld_out, R, C, Rp, colsum);
  else if (in_dtype == DSRG_F32) hipLaunchKernelGGL((transpose_pad_kernel<float, bf16_t>), grid, block, 0, s, (const float*)in, ld_in, (bf16_t*)out, ld_out, R, C, Rp, colsum);
  else if (out_dtype == DSRG_F32) hipLaunchKernelGGL((transpose_pad_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)in, ld_in, (float*)out, ld_out, R, C, Rp, colsum);
  else hipLaunchKernelGGL((transpose_pad_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, R, C, Rp, colsum);
  DSRG_LAUNCH_CHECK("dsrg_transpose_pad");
  return DSRG_OK;
}

extern "C" int dsrg_transpose_pad_grouped(const dsrg_transpose_problem_t* problems, int nproblems, void* stream) {
  DSRG_CHECK(problems && nproblems >= 1 && nproblems <= DSRG_MAX_TRANSPOSES, "dsrg_transpose_pad_grouped: nproblems=%d out of [1,%d]", nproblems, DSRG_MAX_TRANSPOSES);
  TransposeGroup g;
  g.n = 0;
  int tiles = 0;
  for (int i = 0; i < nproblems; ++i) {
    const dsrg_transpose_problem_t& p = problems[i];
    DSRG_CHECK(p.in && p.out && okdt(p.in_dtype) && okdt(p.out_dtype), "dsrg_transpose_pad_grouped: problem %d: bad pointer or dtype", i);
    DSRG_CHECK(p.R >= 0 && p.C >= 0 && p.Rp >= p.R && p.ld_in >= p.C && p.ld_out >= p.Rp,
               "dsrg_transpose_pad_grouped: problem %d: bad dims R=%d C=%d Rp=%d ld_in=%d ld_out=%d", i, p.R, p.C, p.Rp, p.ld_in, p.ld_out);
    if (p.C == 0 || p.Rp == 0) continue;
    g.p[g.n] = p;
    g.tile_start[g.n] = tiles;
    tiles += ((p.C + 63) / 64) * ((p.Rp + 63) / 64);
    ++g.n;
  }
  if (g.n == 0) return DSRG_OK;
  g.tile_start[g.n] = tiles;
  hipLaunchKernelGGL(transpose_pad_grouped_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, g);
  DSRG_LAUNCH_CHECK("dsrg_transpose_pad_grouped");
  return DSRG_OK;
}

extern "C" int dsrg_colsum_acc(const void* in, int ld, float* out, int M, int N, int dtype, void* stream) {
  DSRG_CHECK(in && out && okdt(dtype) && M >= 0 && N >= 0 && ld >= N, "dsrg_colsum_acc: bad arguments");
  if (M == 0 || N == 0) return DSRG_OK;
  const int rpb = 128;
  dim3 grid((N + 63) / 64, (M + rpb - 1) / rpb), block(256);
  if (dtype == DSRG_F32) hipLaunchKernelGGL((colsum_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)in, ld, out, M, N, rpb);
  else hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)in, ld, out, M, N, rpb);
  DSRG_LAUNCH_CHECK("dsrg_colsum_acc");
  return DSRG_OK;
}

extern "C" int dsrg_act_fwd(const void* z, void* y, int64_t n, int act, int dtype, void* stream) {
  DSRG_CHECK(z && y && okdt(dtype) && n >= 0 && n % 4 == 0 && act >= DSRG_ACT_NONE && act <= DSRG_ACT_TANH, "dsrg_act_fwd: bad arguments (n must be a multiple of 4)");
  DSRG_CHECK((((uintptr_t)z | (uintptr_t)y) & (dtype == DSRG_F32 ? 15 : 7)) == 0, "dsrg_act_fwd: misaligned pointer");
  if (n == 0) return DSRG_OK;
  const size_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
  if (dtype == DSRG_F32) hipLaunchKernelGGL((act_kernel<float, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)z, (const float*)nullptr, (float*)y, n4, act);
  else hipLaunchKernelGGL((act_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, (const bf16_t*)nullptr, (bf16_t*)y, n4, act);
  DSRG_LAUNCH_CHECK("dsrg_act_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_act_bwd(const void* z, const void* dy, void* dz, int64_t n, int act, int dtype, void* stream) {
  DSRG_CHECK(z && dy && dz && okdt(dtype) && n >= 0 && n % 4 == 0 && act >= DSRG_ACT_NONE && act <= DSRG_ACT_TANH, "dsrg_act_bwd: bad arguments (n must be a multiple of 4)");
  DSRG_CHECK((((uintptr_t)z | (uintptr_t)dy | (uintptr_t)dz) & (dtype == DSRG_F32 ? 15 : 7)) == 0, "dsrg_act_bwd: misaligned pointer");
  if (n == 0) return DSRG_OK;
  const size_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
  if (dtype == DSRG_F32) hipLaunchKernelGGL((act_kernel<float, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)z, (const float*)dy, (float*)dz, n4, act);
  else hipLaunchKernelGGL((act_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, (const bf16_t*)dy, (bf16_t*)dz, n4, act);
  DSRG_LAUNCH_CHECK("dsrg_act_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_layernorm_bwd_blocks(int M) { return M > 0 ? lnb_blocks(M) : 0; }

extern "C" int dsrg_sum_slabs(const float* in, int S, int64_t slab_stride, void* out, int64_t n, int out_dtype, void* stream) {
  DSRG_CHECK(in && out && okdt(out_dtype) && S >= 1 && n >= 0 && slab_stride >= n && slab_stride % 4 == 0, "dsrg_sum_slabs: bad arguments (slab_stride %% 4 == 0, >= n)");
  DSRG_CHECK((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & (out_dtype == DSRG_F32 ? 15 : 7)) == 0, "dsrg_sum_slabs: misaligned pointer");
  if (n == 0) return DSRG_OK;
  const size_t n4 = (size_t)n / 4;
  const int grid = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256 < 1 ? 1 : (n4 + 255) / 256);
  if (out_dtype == DSRG_F32) hipLaunchKernelGGL((sum_slabs_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, in, S, (size_t)slab_stride, (float*)out, (size_t)n);
  else hipLaunchKernelGGL((sum_slabs_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, in, S, (size_t)slab_stride, (bf16_t*)out, (size_t)n);
  DSRG_LAUNCH_CHECK("dsrg_sum_slabs");
  return DSRG_OK;
}

extern "C" int dsrg_dropout_add(const void* x, const void* residual, void* out, int64_t n, int dtype, float p_drop, uint64_t seed, void* stream) {
  DSRG_CHECK(x && out && okdt(dtype) && n >= 0 && n % 4 == 0, "dsrg_dropout_add: bad arguments (n must be a multiple of 4)");
  DSRG_CHECK((((uintptr_t)x | (uintptr_t)out | (uintptr_t)residual) & (dtype == DSRG_F32 ? 15 : 7)) == 0, "dsrg_dropout_add: misaligned pointer");
  DSRG_CHECK(p_drop >= 0.f && p_drop < 1.f, "dsrg_dropout_add: p_drop=%g out of [0,1)", (double)p_drop);
  if (n == 0) return DSRG_OK;
  const size_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
  const float inv_keep = 1.0f / (1.0f - p_drop);
  const unsigned long long* off = p_drop > 0.f ? dsrg_dropout_offset_ptr() : nullptr;
  if (dtype == DSRG_F32) hipLaunchKernelGGL((dropout_add_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)residual, (float*)out, n4, p_drop, inv_keep, (unsigned long long)seed, off);
  else hipLaunchKernelGGL((dropout_add_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)out, n4, p_drop, inv_keep, (unsigned long long)seed, off);
  DSRG_LAUNCH_CHECK("dsrg_dropout_add");
  return DSRG_OK;
}

static int layernorm_bwd_impl(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const void* dy, int lddy,
                              void* dz, int lddz, void* dzd, int lddzd, float* dgamma, float* dbeta, float* partials, int M, int H, float eps,
                              int in_dtype, int dtype, float p_drop, unsigned long long seed, int tag, void* stream) {
  DSRG_CHECK(x && gamma && dy && dgamma && dbeta && partials && okdt(in_dtype) && okdt(dtype), "dsrg_layernorm_bwd: bad pointer or dtype");
  DSRG_CHECK(in_dtype == DSRG_F32 || in_dtype == dtype, "dsrg_layernorm_bwd: in_dtype must be f32 or equal dtype");
  DSRG_CHECK(H > 0 && H <= 1024 && H % 64 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddz % 4 == 0 && (!residual || ldr % 4 == 0), "dsrg_layernorm_bwd: bad H / leading dims");
  DSRG_CHECK(p_drop >= 0.f && p_drop < 1.f && tag >= 0 && tag < DSRG_MAX_GROUPS, "dsrg_layernorm_bwd: p_drop=%g out of [0,1) or bad tag %d", (double)p_drop, tag);
  const bool drop = p_drop > 0.f;
  DSRG_CHECK(drop ? (dzd != nullptr && lddzd >= H && lddzd % 4 == 0) : dz != nullptr, "dsrg_layernorm_bwd: missing gradient output");
  hipStream_t s = (hipStream_t)stream;
  if (M <= 0) {
    (void)hipMemsetAsync(dgamma, 0, (size_t)H * 4, s);
    (void)hipMemsetAsync(dbeta, 0, (size_t)H * 4, s);
    return DSRG_OK;
  }
  const LnDrop dr{p_drop, 1.0f / (1.0f - p_drop), seed, tag, p_drop > 0.f ? dsrg_dropout_offset_ptr() : nullptr};
  const int R = lnb_rows_per_wave(M);
  dim3 grid(lnb_blocks(M)), block(LNB_WAVES * 64);
#define DSRG_LNB2(TX_, T_, DROP_, R_)                                                                                                   \
  hipLaunchKernelGGL((layernorm_bwd_kernel<TX_, T_, DROP_, R_>), grid, block, 0, s, (const TX_*)x, ldx, (const T_*)residual, ldr, gamma, \
                     (const T_*)dy, lddy, (T_*)dz, lddz, partials, M, H, eps, dr, (T_*)dzd, lddzd)
#define DSRG_LNB(TX_, T_, DROP_) do { if (R == 1) DSRG_LNB2(TX_, T_, DROP_, 1); else DSRG_LNB2(TX_, T_, DROP_, 2); } while (0)
  if (in_dtype == DSRG_F32 && dtype == DSRG_F32) { if (drop) DSRG_LNB(float, float, true); else DSRG_LNB(float, float, false); }
  else if (in_dtype == DSRG_F32) { if (drop) DSRG_LNB(float, bf16_t, true); else DSRG_LNB(float, bf16_t, false); }
  else { if (drop) DSRG_LNB(bf16_t, bf16_t, true); else DSRG_LNB(bf16_t, bf16_t, false); }
#undef DSRG_LNB2
  hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3(2 * H / LNR_COLS), dim3(LNR_COLS * LNR_GROUPS), 0, s, partials, (int)grid.x, H, dgamma, dbeta);
#undef DSRG_LNB
  DSRG_LAUNCH_CHECK("dsrg_layernorm_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_layernorm_bwd(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const void* dy, int lddy,
                                  void* dz, int lddz, float* dgamma, float* dbeta, float* partials, int M, int H, float eps, int in_dtype,
                                  int dtype, void* stream) {
  return layernorm_bwd_impl(x, ldx, residual, ldr, gamma, dy, lddy, dz, lddz, nullptr, 0, dgamma, dbeta, partials, M, H, eps, in_dtype, dtype, 0.f, 0, 0, stream);
}

extern "C" int dsrg_layernorm_bwd_drop(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const void* dy, int lddy,
                                       void* dz, int lddz, void* dzd, int lddzd, float* dgamma, float* dbeta, float* partials, int M, int H,
                                       float eps, int in_dtype, int dtype, float p_drop, uint64_t seed, int tag, void* stream) {
  return layernorm_bwd_impl(x, ldx, residual, ldr, gamma, dy, lddy, dz, lddz, dzd, lddzd, dgamma, dbeta, partials, M, H, eps, in_dtype, dtype, p_drop, seed, tag, stream);
}
