// Optimizer side of the pre-training step (include/dsrg_hip.h: dsrg_grad_sumsq, dsrg_adamw_prepare, dsrg_adamw_step):
// gradient-norm clipping (torch.nn.utils.clip_grad_norm_, pretrain_src/train_r2r.py:283-289) and the reference's own AdamW
// (pretrain_src/optim/adamw.py:30-92) over ALL parameters of a step in three launches, driven by device-resident tables.
//
// HBM-bound streaming work: per parameter element the update reads p, g, m, v and writes p, m, v (28 B), the norm reads g
// (4 B).  One workgroup owns one 4096-element chunk of one tensor (block table: tensor index, chunk index), 256 lanes x
// 4 float4 each, fully coalesced 16-byte accesses; tensors whose pointers are not 16-byte aligned take a scalar path.
// Nothing here allocates; the tables, partial sums, step counters and hyper-parameter words are the caller's buffers.
#include "common.h"

namespace {

constexpr int OPT_THREADS = 256;
constexpr int OPT_VEC = 4;                                   // float4 per lane
constexpr int OPT_CHUNK = OPT_THREADS * OPT_VEC * 4;         // 4096 elements per workgroup

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < OPT_THREADS / 64; ++i) s += red[i];
  return s;
}

__global__ __launch_bounds__(OPT_THREADS) void grad_sumsq_kernel(const dsrg_opt_tensor_t* __restrict__ tab, const int2* __restrict__ blk,
                                                                  float* __restrict__ partials) {
  __shared__ float red[OPT_THREADS / 64];
  const int2 b = blk[blockIdx.x];
  const dsrg_opt_tensor_t t = tab[b.x];
  const long long base = (long long)b.y * OPT_CHUNK;
  const long long n = t.numel - base < OPT_CHUNK ? t.numel - base : OPT_CHUNK;
  const float* g = t.g ? reinterpret_cast<const float*>(t.g) + base : nullptr;
  const bf16_t* g2 = t.g2 ? reinterpret_cast<const bf16_t*>(t.g2) + base : nullptr;     // the gradient (or a second share of it) in bf16
  float acc = 0.f;
  if (t.vec_ok && n == OPT_CHUNK) {
    float4 v[OPT_VEC];
#pragma unroll
    for (int i = 0; i < OPT_VEC; ++i) {
      const int j = threadIdx.x + i * OPT_THREADS;
      v[i] = g ? reinterpret_cast<const float4*>(g)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (g2) { const float4 h = ld4(g2 + 4 * j); v[i].x += h.x; v[i].y += h.y; v[i].z += h.z; v[i].w += h.w; }
    }
#pragma unroll
    for (int i = 0; i < OPT_VEC; ++i) acc += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
  } else {
    for (long long i = threadIdx.x; i < n; i += OPT_THREADS) {
      const float x = (g ? g[i] : 0.f) + (g2 ? bf2f(g2[i]) : 0.f);
      acc += x * x;
    }
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// one workgroup: total norm from the per-chunk partial sums (fixed order, double), the clip coefficient, and the step counters
// of the tensors this step updates
__global__ __launch_bounds__(1024) void adamw_prepare_kernel(const float* __restrict__ partials, int n_partials, float max_norm, float gscale,
                                                             float* __restrict__ norm_out, const dsrg_opt_tensor_t* __restrict__ tab, int n_tensors,
                                                             int* __restrict__ steps) {
  __shared__ double red[1024];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += 1024) acc += (double)partials[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]) * gscale;                   // norm of the gradients as the update will see them
    float coef = gscale;
    if (n_partials > 0 && max_norm > 0.f) coef *= fminf(max_norm / (norm + 1e-6f), 1.0f);      // clip_grad_norm_: clamp(max_norm / (total + 1e-6), max = 1)
    norm_out[0] = norm;
    norm_out[1] = coef;
  }
  for (int i = threadIdx.x; i < n_tensors; i += 1024) steps[tab[i].step_idx] += 1;
}

struct AdamConst { double b1d, b2d; float b1, b2, omb1, omb2, eps; int correct_bias; };     // omb = 1 - beta, rounded from double like the reference's Python scalars

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float coef, const AdamConst& c, float step_size, float decay) {
  const float b1 = c.b1, b2 = c.b2, eps = c.eps;
  g *= coef;
  m = m * b1 + c.omb1 * g;                 // exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)            adamw.py:66
  v = v * b2 + c.omb2 * g * g;                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)  :67
  const float denom = sqrtf(v) + eps;           // :68
  p = p - step_size * (m / denom);              // p.addcdiv_(exp_avg, denom, value=-step_size)                  :76
  p = p - decay * p;                            // p.add_(p, alpha=-lr * weight_decay), after the update         :87  (decay = 0: p - 0 = p)
}

__global__ __launch_bounds__(OPT_THREADS) void adamw_step_kernel(const dsrg_opt_tensor_t* __restrict__ tab, const int2* __restrict__ blk,
                                                                  const float* __restrict__ hp, const float* __restrict__ norm, const int* __restrict__ steps,
                                                                  const AdamConst c) {
  const int2 b = blk[blockIdx.x];
  const dsrg_opt_tensor_t t = tab[b.x];
  const long long base = (long long)b.y * OPT_CHUNK;
  const long long n = t.numel - base < OPT_CHUNK ? t.numel - base : OPT_CHUNK;
  const float lr = hp[2 * t.group], wd = hp[2 * t.group + 1];
  const float coef = norm ? norm[1] : 1.0f;
  float step_size = lr;
  if (c.correct_bias) {                         // adamw.py:71-74; the counter was advanced by adamw_prepare_kernel
    const double s = (double)steps[t.step_idx];
    step_size = (float)((double)lr * sqrt(1.0 - pow(c.b2d, s)) / (1.0 - pow(c.b1d, s)));
  }
  const float decay = wd > 0.f ? lr * wd : 0.f;
  float* p = reinterpret_cast<float*>(t.p) + base;
  const float* g = t.g ? reinterpret_cast<const float*>(t.g) + base : nullptr;
  const bf16_t* g2 = t.g2 ? reinterpret_cast<const bf16_t*>(t.g2) + base : nullptr;
  float* m = t.m + base;
  float* v = t.v + base;
  bf16_t* pk = t.pack ? reinterpret_cast<bf16_t*>(t.pack) + base : nullptr;      // compute-dtype copy of the parameter (the next forward's operand)
  if (t.vec_ok && n == OPT_CHUNK) {
    float4 pv[OPT_VEC], gv[OPT_VEC], mv[OPT_VEC], vv[OPT_VEC];
#pragma unroll
    for (int i = 0; i < OPT_VEC; ++i) {
      const int j = threadIdx.x + i * OPT_THREADS;
      pv[i] = reinterpret_cast<const float4*>(p)[j];
      gv[i] = g ? reinterpret_cast<const float4*>(g)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      mv[i] = reinterpret_cast<const float4*>(m)[j]; vv[i] = reinterpret_cast<const float4*>(v)[j];
      if (g2) { const float4 h = ld4(g2 + 4 * j); gv[i].x += h.x; gv[i].y += h.y; gv[i].z += h.z; gv[i].w += h.w; }
    }
#pragma unroll
    for (int i = 0; i < OPT_VEC; ++i) {
      const int j = threadIdx.x + i * OPT_THREADS;
      adamw_one(pv[i].x, gv[i].x, mv[i].x, vv[i].x, coef, c, step_size, decay);
      adamw_one(pv[i].y, gv[i].y, mv[i].y, vv[i].y, coef, c, step_size, decay);
      adamw_one(pv[i].z, gv[i].z, mv[i].z, vv[i].z, coef, c, step_size, decay);
      adamw_one(pv[i].w, gv[i].w, mv[i].w, vv[i].w, coef, c, step_size, decay);
      reinterpret_cast<float4*>(p)[j] = pv[i]; reinterpret_cast<float4*>(m)[j] = mv[i]; reinterpret_cast<float4*>(v)[j] = vv[i];
      if (pk) st4(pk + 4 * j, pv[i]);
    }
  } else {
    for (long long i = threadIdx.x; i < n; i += OPT_THREADS) {
      float pp = p[i], mm = m[i], vv = v[i];
      adamw_one(pp, (g ? g[i] : 0.f) + (g2 ? bf2f(g2[i]) : 0.f), mm, vv, coef, c, step_size, decay);
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (pk) pk[i] = f2bf(pp);
    }
  }
}

}  // namespace

extern "C" int dsrg_opt_chunk(void) { return OPT_CHUNK; }

extern "C" int dsrg_grad_sumsq(const dsrg_opt_tensor_t* table, const int32_t* blocks, int n_blocks, float* partials, hipStream_t stream) {
  DSRG_CHECK(n_blocks >= 0, "dsrg_grad_sumsq: n_blocks < 0");
  if (n_blocks == 0) return DSRG_OK;
  DSRG_CHECK(table && blocks && partials, "dsrg_grad_sumsq: null table / blocks / partials");
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(n_blocks), dim3(OPT_THREADS), 0, stream, table, reinterpret_cast<const int2*>(blocks), partials);
  DSRG_LAUNCH_CHECK("dsrg_grad_sumsq");
  return DSRG_OK;
}

extern "C" int dsrg_adamw_prepare(const float* partials, int n_partials, float max_norm, float grad_scale, float* norm_out,
                                  const dsrg_opt_tensor_t* table, int n_tensors, int32_t* steps, hipStream_t stream) {
  DSRG_CHECK(norm_out && steps && (table || n_tensors == 0), "dsrg_adamw_prepare: null norm_out / steps / table");
  DSRG_CHECK(n_partials >= 0 && n_tensors >= 0 && (partials || n_partials == 0), "dsrg_adamw_prepare: bad partials");
  DSRG_CHECK(grad_scale > 0.f, "dsrg_adamw_prepare: grad_scale must be positive");
  hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(1024), 0, stream, partials, n_partials, max_norm, grad_scale, norm_out, table, n_tensors, steps);
  DSRG_LAUNCH_CHECK("dsrg_adamw_prepare");
  return DSRG_OK;
}

extern "C" int dsrg_adamw_step(const dsrg_opt_tensor_t* table, const int32_t* blocks, int n_blocks, const float* hyper, const float* norm,
                               const int32_t* steps, double beta1, double beta2, double eps, int correct_bias, hipStream_t stream) {
  DSRG_CHECK(n_blocks >= 0, "dsrg_adamw_step: n_blocks < 0");
  if (n_blocks == 0) return DSRG_OK;
  DSRG_CHECK(table && blocks && hyper && steps, "dsrg_adamw_step: null table / blocks / hyper / steps");
  DSRG_CHECK(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "dsrg_adamw_step: betas in [0, 1), eps >= 0");
  const AdamConst c{beta1, beta2, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, correct_bias};
  hipLaunchKernelGGL(adamw_step_kernel, dim3(n_blocks), dim3(OPT_THREADS), 0, stream, table, reinterpret_cast<const int2*>(blocks), hyper, norm, steps, c);
  DSRG_LAUNCH_CHECK("dsrg_adamw_step");
  return DSRG_OK;
}
