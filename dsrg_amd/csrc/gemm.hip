// Grouped NT GEMM with fused bias / activation / residual epilogue for gfx950.
//   C[m,n] = act( sum_k A[m,k] * W[n,k] + bias[n] ) + residual[m,n]
// bf16 storage -> v_mfma_f32_16x16x32_bf16; f32 storage -> v_mfma_f32_16x16x4_f32 (exact fp32
// fma chain: the fp32 parity mode).  Both dtypes share one byte-level data path: an LDS tile row
// is 128 bytes (64 bf16 or 32 f32 of K), moved in 16-byte chunks, XOR-swizzled so that the
// ds_read_b128 fragment reads of a 16-row x 4-chunk wave access are bank-conflict free
// (chunk' = chunk ^ ((row >> 1) & 7); derivation in DESIGN.md).
// Operands are passed to the MFMA swapped (W as the A operand, the activations as B) so that each
// lane's 4 accumulator registers are 4 CONSECUTIVE output columns n of one row m: the epilogue then
// reads bias/residual and writes C with 8/16-byte vector accesses.
#include "common.h"

namespace {

constexpr int ROWB = 128;  // bytes of K per LDS tile row

struct GemmArgs {
  dsrg_gemm_problem_t p[DSRG_MAX_GROUPS];
  int tile_start[DSRG_MAX_GROUPS + 1];
  int nprob;
  int act;
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case DSRG_ACT_GELU: return gelu_erf(v);
    case DSRG_ACT_RELU: return fmaxf(v, 0.0f);
    case DSRG_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

template <typename T, typename OutT, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs args) {
  constexpr int PER16 = Elem<T>::PER16;
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int WM = BM / 2, WN = BN / 2;     // wave tile (2x2 waves)
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int TILE_BYTES = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- which problem / tile
  int bid = blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < DSRG_MAX_GROUPS; ++i)
    if (i < args.nprob && bid >= args.tile_start[i]) g = i;
  const dsrg_gemm_problem_t& P = args.p[g];
  const int M = P.M, N = P.N, K = P.K;
  const int t = bid - args.tile_start[g];
  const int tiles_m = (M + BM - 1) / BM;
  const int tm = t % tiles_m, tn = t / tiles_m;   // consecutive blocks share the W panel
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, q = lane >> 4;

  const T* __restrict__ A = reinterpret_cast<const T*>(P.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(P.W);

  // ---- staging map: thread -> (row lr + 32*pass, 16B chunk lc)
  const int lr = tid >> 3, lc = tid & 7;
  constexpr int PA = BM / 32, PW = BN / 32;
  uint4 ra[PA], rw[PW];

  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      int m = m0 + lr + 32 * p;
      ra[p] = (m < M) ? *reinterpret_cast<const uint4*>(A + (size_t)m * P.lda + k0 + lc * PER16)
                      : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < PW; ++p) {
      int n = n0 + lr + 32 * p;
      rw[p] = (n < N) ? *reinterpret_cast<const uint4*>(W + (size_t)n * P.ldw + k0 + lc * PER16)
                      : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
    char* sa = smem + buf * TILE_BYTES;
    char* sw = sa + BM * ROWB;
#pragma unroll
    for (int p = 0; p < PA; ++p) *reinterpret_cast<uint4*>(sa + swz(lr + 32 * p, lc)) = ra[p];
#pragma unroll
    for (int p = 0; p < PW; ++p) *reinterpret_cast<uint4*>(sw + swz(lr + 32 * p, lc)) = rw[p];
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const char* sa = smem + cur * TILE_BYTES;
    const char* sw = sa + BM * ROWB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int chunk = q + 4 * kk;
      uint4 fa[MT], fw[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const uint4*>(sa + swz(wm * WM + i * 16 + r16, chunk));
#pragma unroll
      for (int j = 0; j < NT; ++j) fw[j] = *reinterpret_cast<const uint4*>(sw + swz(wn * WN + j * 16 + r16, chunk));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fw[j]),
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[i]), acc[i][j], 0, 0, 0);
          } else {
            const float* pw = reinterpret_cast<const float*>(&fw[j]);
            const float* pa = reinterpret_cast<const float*>(&fa[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[e], pa[e], acc[i][j], 0, 0, 0);
          }
        }
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n..n+3]
  const float* __restrict__ bias = P.bias;
  const T* __restrict__ R = reinterpret_cast<const T*>(P.residual);
  OutT* __restrict__ C = reinterpret_cast<OutT*>(P.C);
  const int act = args.act;
  const bool vec_ok = ((P.ldc & 3) == 0) && (R == nullptr || (P.ldr & 3) == 0);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * WM + i * 16 + r16;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * WN + j * 16 + q * 4;
      if (n >= N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (n + 3 < N && vec_ok) {
        if (bias) {
          float4 b = *reinterpret_cast<const float4*>(bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
        if (R) {
          float4 r = ld4(R + (size_t)m * P.ldr + n);
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        st4(C + (size_t)m * P.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
      } else {
        for (int e = 0; e < 4 && n + e < N; ++e) {
          float x = v[e] + (bias ? bias[n + e] : 0.f);
          x = apply_act(x, act);
          if (R) x += Elem<T>::ld(R + (size_t)m * P.ldr + n + e);
          Elem<OutT>::st(C + (size_t)m * P.ldc + n + e, x);
        }
      }
    }
  }
}

template <typename T, typename OutT, int BM, int BN>
int launch(const GemmArgs& a, int total_tiles, hipStream_t s) {
  constexpr int smem = 2 * (BM + BN) * ROWB;
  hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, BM, BN>), dim3(total_tiles), dim3(256), smem, s, a);
  DSRG_LAUNCH_CHECK("dsrg_gemm_nt");
  return DSRG_OK;
}

// tile choice: fill >= ~1 wave of the 256 CUs when the problem allows it
int plan_tiles(GemmArgs& a, int& bm, int& bn) {
  auto count = [&](int m, int n) {
    int tot = 0;
    for (int i = 0; i < a.nprob; ++i) {
      a.tile_start[i] = tot;
      tot += ((a.p[i].M + m - 1) / m) * ((a.p[i].N + n - 1) / n);
    }
    a.tile_start[a.nprob] = tot;
    return tot;
  };
  int maxM = 0;
  for (int i = 0; i < a.nprob; ++i) maxM = a.p[i].M > maxM ? a.p[i].M : maxM;
  if (maxM > 64 && count(128, 128) >= 400) { bm = 128; bn = 128; }
  else if (maxM > 64 && count(64, 128) >= 128) { bm = 64; bn = 128; }
  else { bm = 64; bn = 64; }
  return count(bm, bn);
}

template <typename T, typename OutT>
int dispatch_tile(GemmArgs& a, hipStream_t s) {
  int bm, bn;
  const int tiles = plan_tiles(a, bm, bn);
  if (bm == 128) return launch<T, OutT, 128, 128>(a, tiles, s);
  if (bn == 128) return launch<T, OutT, 64, 128>(a, tiles, s);
  return launch<T, OutT, 64, 64>(a, tiles, s);
}

}  // namespace

extern "C" int dsrg_gemm_nt(const dsrg_gemm_problem_t* problems, int nproblems, int dtype, int out_dtype,
                            int act, void* stream) {
  DSRG_CHECK(problems && nproblems >= 1 && nproblems <= DSRG_MAX_GROUPS, "dsrg_gemm_nt: nproblems=%d out of [1,%d]",
             nproblems, DSRG_MAX_GROUPS);
  DSRG_CHECK(dtype == DSRG_F32 || dtype == DSRG_BF16, "dsrg_gemm_nt: bad dtype %d", dtype);
  DSRG_CHECK(out_dtype == DSRG_F32 || out_dtype == dtype, "dsrg_gemm_nt: out_dtype must be f32 or the input dtype");
  DSRG_CHECK(act >= DSRG_ACT_NONE && act <= DSRG_ACT_TANH, "dsrg_gemm_nt: bad act %d", act);
  const int bk = dtype == DSRG_BF16 ? 64 : 32, al = dtype == DSRG_BF16 ? 8 : 4;
  GemmArgs a;
  a.nprob = nproblems;
  a.act = act;
  int live = 0;
  for (int i = 0; i < nproblems; ++i) {
    const dsrg_gemm_problem_t& p = problems[i];
    DSRG_CHECK(p.M >= 0 && p.N >= 0 && p.K > 0, "dsrg_gemm_nt: problem %d has bad dims %dx%dx%d", i, p.M, p.N, p.K);
    if (p.M == 0 || p.N == 0) continue;
    DSRG_CHECK(p.A && p.W && p.C, "dsrg_gemm_nt: problem %d has a null A/W/C", i);
    DSRG_CHECK(p.K % bk == 0, "dsrg_gemm_nt: problem %d K=%d must be a multiple of %d", i, p.K, bk);
    DSRG_CHECK(p.lda % al == 0 && p.ldw % al == 0 && p.lda >= p.K && p.ldw >= p.K,
               "dsrg_gemm_nt: problem %d lda=%d/ldw=%d must be >= K and multiples of %d", i, p.lda, p.ldw, al);
    DSRG_CHECK(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0, "dsrg_gemm_nt: problem %d A/W not 16-byte aligned", i);
    DSRG_CHECK(p.ldc >= p.N && (!p.residual || p.ldr >= p.N), "dsrg_gemm_nt: problem %d ldc/ldr < N", i);
    DSRG_CHECK(((uintptr_t)p.C & 15) == 0 && ((uintptr_t)p.residual & 7) == 0 && ((uintptr_t)p.bias & 15) == 0,
               "dsrg_gemm_nt: problem %d C/residual/bias misaligned", i);
    a.p[live++] = p;
  }
  if (live == 0) return DSRG_OK;
  a.nprob = live;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DSRG_BF16) {
    if (out_dtype == DSRG_F32) return dispatch_tile<bf16_t, float>(a, s);
    return dispatch_tile<bf16_t, bf16_t>(a, s);
  }
  return dispatch_tile<float, float>(a, s);
}

extern "C" int dsrg_gemm_nt_plan(const dsrg_gemm_problem_t* problems, int nproblems, int dtype, int* bm, int* bn, int* ntiles) {
  DSRG_CHECK(problems && nproblems >= 1 && nproblems <= DSRG_MAX_GROUPS && bm && bn && ntiles, "dsrg_gemm_nt_plan: bad arguments");
  (void)dtype;
  GemmArgs a;
  a.nprob = 0;
  for (int i = 0; i < nproblems; ++i)
    if (problems[i].M > 0 && problems[i].N > 0) a.p[a.nprob++] = problems[i];
  *bm = *bn = *ntiles = 0;
  if (a.nprob) *ntiles = plan_tiles(a, *bm, *bn);
  return DSRG_OK;
}
