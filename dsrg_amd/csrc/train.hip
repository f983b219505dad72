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

// LayerNorm backward.  z = x (+ res); zh = (z - mean) * rstd; y = zh*gamma + beta.
//   dz = rstd * (g - mean(g) - zh * mean(g*zh)),  g = dy*gamma;   dgamma += sum_rows dy*zh;  dbeta += sum_rows dy
// One wave per row, each block walks a strip of rows and keeps its dgamma/dbeta partials in registers, then one
// atomic per column per block.
constexpr int LNB_MAXV = 4;
// DROP: the forward ran  z = dropout(x) + res  (dsrg_layernorm_fwd_grouped_drop): the mask is recomputed from (seed, tag, row,
// col); dz is the gradient of the residual, dzd = dz * keep / (1 - p) the gradient of x (the dense output in front of it).
struct LnDrop { float p, inv_keep; unsigned long long seed; int tag; const unsigned long long* seed_off; };
template <typename TX, typename T, bool DROP = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TX* __restrict__ x, int ldx, const T* __restrict__ res, int ldr,
                                                            const float* __restrict__ gamma, const T* __restrict__ dy, int lddy,
                                                            T* __restrict__ dz, int lddz, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int M, int H, float eps, int rows_per_block, LnDrop dr = LnDrop{},
                                                            T* __restrict__ dzd = nullptr, int lddzd = 0) {
  __shared__ float red[2][4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = (H + 255) / 256;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
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
  for (int row = m0 + wave; row < m1; row += 4) {
    float4 z[LNB_MAXV], g[LNB_MAXV], kp[DROP ? LNB_MAXV : 1];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < H) {
        z[i] = ld4(x + (size_t)row * ldx + c);
        if constexpr (DROP) {
          const size_t e = ((size_t)dr.tag << 40) + (size_t)row * H + c;
          kp[i] = make_float4(drop_keep(dr.seed, e, dr.p, dr.inv_keep), drop_keep(dr.seed, e + 1, dr.p, dr.inv_keep),
                              drop_keep(dr.seed, e + 2, dr.p, dr.inv_keep), drop_keep(dr.seed, e + 3, dr.p, dr.inv_keep));
          z[i].x *= kp[i].x; z[i].y *= kp[i].y; z[i].z *= kp[i].z; z[i].w *= kp[i].w;
        }
        if (res) { const float4 r = ld4(res + (size_t)row * ldr + c); z[i].x += r.x; z[i].y += r.y; z[i].z += r.z; z[i].w += r.w; }
        g[i] = ld4(dy + (size_t)row * lddy + c);
        s += z[i].x + z[i].y + z[i].z + z[i].w;
      }
    }
    const float mean = wave_sum(s) / (float)H;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < H) {
        z[i].x -= mean; z[i].y -= mean; z[i].z -= mean; z[i].w -= mean;
        ss += z[i].x * z[i].x + z[i].y * z[i].y + z[i].z * z[i].z + z[i].w * z[i].w;
      }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
    float sg = 0.f, sgz = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < H) {
        z[i].x *= rstd; z[i].y *= rstd; z[i].z *= rstd; z[i].w *= rstd;          // zh
        ab[i].x += g[i].x; ab[i].y += g[i].y; ab[i].z += g[i].z; ab[i].w += g[i].w;
        ag[i].x += g[i].x * z[i].x; ag[i].y += g[i].y * z[i].y; ag[i].z += g[i].z * z[i].z; ag[i].w += g[i].w * z[i].w;
        g[i].x *= gm[i].x; g[i].y *= gm[i].y; g[i].z *= gm[i].z; g[i].w *= gm[i].w;
        sg += g[i].x + g[i].y + g[i].z + g[i].w;
        sgz += g[i].x * z[i].x + g[i].y * z[i].y + g[i].z * z[i].z + g[i].w * z[i].w;
      }
    }
    const float mg = wave_sum(sg) / (float)H, mgz = wave_sum(sgz) / (float)H;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (i < nv && c < H) {
        const float4 d = make_float4(rstd * (g[i].x - mg - z[i].x * mgz), rstd * (g[i].y - mg - z[i].y * mgz),
                                     rstd * (g[i].z - mg - z[i].z * mgz), rstd * (g[i].w - mg - z[i].w * mgz));
        if (dz) st4(dz + (size_t)row * lddz + c, d);
        if constexpr (DROP) st4(dzd + (size_t)row * lddzd + c, make_float4(d.x * kp[i].x, d.y * kp[i].y, d.z * kp[i].z, d.w * kp[i].w));
      }
    }
  }
  // combine the 4 waves' partials, one atomic per column
#pragma unroll
  for (int i = 0; i < LNB_MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (i < nv && c < H) {
      *reinterpret_cast<float4*>(&red[0][wave][c]) = ag[i];
      *reinterpret_cast<float4*>(&red[1][wave][c]) = ab[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

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

static int layernorm_bwd_impl(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const void* dy, int lddy,
                              void* dz, int lddz, void* dzd, int lddzd, float* dgamma, float* dbeta, int M, int H, float eps, int in_dtype,
                              int dtype, float p_drop, unsigned long long seed, int tag, void* stream) {
  DSRG_CHECK(x && gamma && dy && dgamma && dbeta && okdt(in_dtype) && okdt(dtype), "dsrg_layernorm_bwd: bad pointer or dtype");
  DSRG_CHECK(in_dtype == DSRG_F32 || in_dtype == dtype, "dsrg_layernorm_bwd: in_dtype must be f32 or equal dtype");
  DSRG_CHECK(H > 0 && H <= 1024 && H % 64 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddz % 4 == 0 && (!residual || ldr % 4 == 0), "dsrg_layernorm_bwd: bad H / leading dims");
  DSRG_CHECK(p_drop >= 0.f && p_drop < 1.f && tag >= 0 && tag < DSRG_MAX_GROUPS, "dsrg_layernorm_bwd: p_drop=%g out of [0,1) or bad tag %d", (double)p_drop, tag);
  const bool drop = p_drop > 0.f;
  DSRG_CHECK(drop ? (dzd != nullptr && lddzd >= H && lddzd % 4 == 0) : dz != nullptr, "dsrg_layernorm_bwd: missing gradient output");
  if (M <= 0) return DSRG_OK;
  const int rpb = 16;
  dim3 grid((M + rpb - 1) / rpb), block(256);
  hipStream_t s = (hipStream_t)stream;
  const LnDrop dr{p_drop, 1.0f / (1.0f - p_drop), seed, tag, p_drop > 0.f ? dsrg_dropout_offset_ptr() : nullptr};
#define DSRG_LNB(TX_, T_, DROP_)                                                                                                   \
  hipLaunchKernelGGL((layernorm_bwd_kernel<TX_, T_, DROP_>), grid, block, 0, s, (const TX_*)x, ldx, (const T_*)residual, ldr, gamma, \
                     (const T_*)dy, lddy, (T_*)dz, lddz, dgamma, dbeta, M, H, eps, rpb, dr, (T_*)dzd, lddzd)
  if (in_dtype == DSRG_F32 && dtype == DSRG_F32) { if (drop) DSRG_LNB(float, float, true); else DSRG_LNB(float, float, false); }
  else if (in_dtype == DSRG_F32) { if (drop) DSRG_LNB(float, bf16_t, true); else DSRG_LNB(float, bf16_t, false); }
  else { if (drop) DSRG_LNB(bf16_t, bf16_t, true); else DSRG_LNB(bf16_t, bf16_t, false); }
#undef DSRG_LNB
  DSRG_LAUNCH_CHECK("dsrg_layernorm_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_layernorm_bwd(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const void* dy, int lddy,
                                  void* dz, int lddz, float* dgamma, float* dbeta, int M, int H, float eps, int in_dtype, int dtype,
                                  void* stream) {
  return layernorm_bwd_impl(x, ldx, residual, ldr, gamma, dy, lddy, dz, lddz, nullptr, 0, dgamma, dbeta, M, H, eps, in_dtype, dtype, 0.f, 0, 0, stream);
}

extern "C" int dsrg_layernorm_bwd_drop(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const void* dy, int lddy,
                                       void* dz, int lddz, void* dzd, int lddzd, float* dgamma, float* dbeta, int M, int H, float eps,
                                       int in_dtype, int dtype, float p_drop, uint64_t seed, int tag, void* stream) {
  return layernorm_bwd_impl(x, ldx, residual, ldr, gamma, dy, lddy, dz, lddz, dzd, lddzd, dgamma, dbeta, M, H, eps, in_dtype, dtype, p_drop, seed, tag, stream);
}
