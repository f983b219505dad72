// Micro-benchmark (round 2): the operand traffic of one NT GEMM launch WITHOUT the arithmetic - every workgroup pulls
// the (BM + BN) x K operand panel of its output tile from L2 / Infinity Cache, K-step by K-step (128 B of K per row),
//   mode 0: LDS-DMA (global_load_lds_dwordx4) into an NST-deep LDS ring, counted vmcnt + barrier per K-step
//   mode 1: global_load_dwordx4 into registers only (two register sets in flight), nothing written to LDS
//   mode 2: global_load_dwordx4 into registers, then ds_write_b128 into a 2-deep LDS ring + barrier per K-step
// for the tile shapes a nav-step GEMM could use.  Prints us per launch and B/clk/CU; answers "what bounds the K loop".
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/ingest2.hip -o gpurun_out/ingest2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ROWB = 128;

template <int BM, int BN, int MODE, int NST>
__global__ __launch_bounds__(256) void ingest(const char* __restrict__ A, const char* __restrict__ W, int M, int N, int ldb, int nk,
                                              int tiles_m, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TR = BM + BN, LPS = TR / 32, LPA = BM / 32, STAGE = TR * ROWB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const char* p[LPS];
#pragma unroll
  for (int i = 0; i < LPS; ++i) {
    const int row = (i * 4 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (i < LPA) p[i] = A + (size_t)min(tm * BM + row, M - 1) * ldb + chunk * 16;
    else p[i] = W + (size_t)min(tn * BN + row - BM, N - 1) * ldb + chunk * 16;
  }
  unsigned acc = 0;
  if (MODE == 0) {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int stage, int k) {
#pragma unroll
      for (int i = 0; i < LPS; ++i) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE + (i * 4 + wave) * 1024);
        const char* g = p[i] + (size_t)k * ROWB;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
      }
    };
    for (int d = 0; d < NST - 1; ++d) if (d < nk) issue(d, d);
    int cur = 0;
    for (int k = 0; k < nk; ++k) {
      const int younger = min(NST - 2, nk - 1 - k);
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(2 * LPS) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(LPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (k + NST - 1 < nk) issue(cur == 0 ? NST - 1 : cur - 1, k + NST - 1);
      acc += *reinterpret_cast<const unsigned*>(smem + cur * STAGE + threadIdx.x * 4);
      cur = cur == NST - 1 ? 0 : cur + 1;
    }
  } else {
    uint4 r0[LPS], r1[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) r0[i] = *reinterpret_cast<const uint4*>(p[i]);
#pragma unroll
    for (int i = 0; i < LPS; ++i) r1[i] = *reinterpret_cast<const uint4*>(p[i] + (size_t)(nk > 1 ? 1 : 0) * ROWB);
    for (int k = 0; k < nk; k += 2) {
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) *reinterpret_cast<uint4*>(smem + (i * 4 + wave) * 1024 + lane * 16) = r0[i];
      } else {
#pragma unroll
        for (int i = 0; i < LPS; ++i) acc ^= r0[i].x ^ r0[i].y ^ r0[i].z ^ r0[i].w;
      }
      if (k + 2 < nk) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) r0[i] = *reinterpret_cast<const uint4*>(p[i] + (size_t)(k + 2) * ROWB);
      }
      if (MODE == 2) { __syncthreads(); acc += *reinterpret_cast<const unsigned*>(smem + threadIdx.x * 4); }
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) *reinterpret_cast<uint4*>(smem + STAGE + (i * 4 + wave) * 1024 + lane * 16) = r1[i];
      } else {
#pragma unroll
        for (int i = 0; i < LPS; ++i) acc ^= r1[i].x ^ r1[i].y ^ r1[i].z ^ r1[i].w;
      }
      if (k + 3 < nk) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) r1[i] = *reinterpret_cast<const uint4*>(p[i] + (size_t)(k + 3) * ROWB);
      }
      if (MODE == 2) { __syncthreads(); acc += *reinterpret_cast<const unsigned*>(smem + STAGE + threadIdx.x * 4); }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void empty_kernel(unsigned* sink) { if (threadIdx.x == 999) sink[0] = 1; }

static char *gA, *gW, *gFlush, *gA2, *gW2;
static hipStream_t s1, s2;
static unsigned* gsink;

template <int BM, int BN, int MODE, int NST>
int run(int M, int N, int K) {
  const int ldb = K * 2, nk = ldb / ROWB;
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN, blocks = tm * tn;
  const int smem = MODE == 1 ? 1024 : (MODE == 2 ? 2 : NST) * (BM + BN) * ROWB;
  CK(hipFuncSetAttribute((const void*)ingest<BM, BN, MODE, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float tot = 0;
  const int rep = 10;
  for (int w = 0; w < rep + 2; ++w) {
    CK(hipMemsetAsync(gFlush, w, 64 << 20));      // another kernel's traffic between launches: operands come from beyond the L2s
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((ingest<BM, BN, MODE, NST>), dim3(blocks), dim3(256), smem, 0, gA, gW, M, N, ldb, nk, tm, gsink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (w >= 2) tot += ms;
  }
  const double us = tot * 1e3 / rep, bytes = (double)blocks * nk * (BM + BN) * ROWB;
  const double percu = (double)((blocks + 255) / 256) * nk * (BM + BN) * ROWB;
  printf("M %4d N %4d K %4d tile %3dx%3d mode %d nst %d blocks %4d: %6.1f us  %6.2f TB/s  max-per-CU %4.0f KB -> %5.1f B/clk/CU(2.1GHz)\n", M, N, K, BM, BN,
         MODE, NST, blocks, us, bytes / us / 1e6, percu / 1024, percu / (us * 1e-6 * 2.1e9));
  return 0;
}

// the same launch on ONE stream twice (different operands) vs on TWO streams at once: do two launches overlap?
template <int BM, int BN, int NST>
int run_pair(int M, int N, int K) {
  const int ldb = K * 2, nk = ldb / ROWB;
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN, blocks = tm * tn;
  const int smem = NST * (BM + BN) * ROWB;
  CK(hipFuncSetAttribute((const void*)ingest<BM, BN, 0, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1, ef; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ef));
  const int rep = 10;
  for (int mode = 0; mode < 3; ++mode) {
    float tot = 0;
    for (int w = 0; w < rep + 2; ++w) {
      CK(hipMemsetAsync(gFlush, w, 64 << 20, s1));
      CK(hipEventRecord(e0, s1));
      if (mode == 2) { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); }
      hipLaunchKernelGGL((ingest<BM, BN, 0, NST>), dim3(blocks), dim3(256), smem, s1, gA, gW, M, N, ldb, nk, tm, gsink);
      if (mode >= 1) hipLaunchKernelGGL((ingest<BM, BN, 0, NST>), dim3(blocks), dim3(256), smem, mode == 2 ? s2 : s1, gA2, gW2, M, N, ldb, nk, tm, gsink);
      if (mode == 2) { CK(hipEventRecord(ef, s2)); CK(hipStreamWaitEvent(s1, ef, 0)); }
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (w >= 2) tot += ms;
    }
    printf("M %4d N %4d K %4d tile %3dx%3d nst %d blocks %4d  %s: %6.1f us\n", M, N, K, BM, BN, NST, blocks,
           mode == 0 ? "one launch          " : mode == 1 ? "two, same stream    " : "two, on two streams ", tot * 1e3 / rep);
  }
  return 0;
}

int main() {
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  CK(hipMalloc(&gA2, (size_t)4096 * 3072 * 2)); CK(hipMemset(gA2, 1, (size_t)4096 * 3072 * 2));
  CK(hipMalloc(&gW2, (size_t)4096 * 3072 * 2)); CK(hipMemset(gW2, 2, (size_t)4096 * 3072 * 2));

  CK(hipMalloc(&gA, (size_t)4096 * 3072 * 2)); CK(hipMemset(gA, 1, (size_t)4096 * 3072 * 2));
  CK(hipMalloc(&gW, (size_t)4096 * 3072 * 2)); CK(hipMemset(gW, 2, (size_t)4096 * 3072 * 2));
  CK(hipMalloc(&gFlush, 64 << 20)); CK(hipMalloc(&gsink, 4));
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tot = 0;
    for (int w = 0; w < 12; ++w) {
      CK(hipMemsetAsync(gFlush, w, 64 << 20));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(empty_kernel, dim3(456), dim3(256), 0, 0, gsink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (w >= 2) tot += ms;
    }
    printf("empty kernel, 456 blocks, same event bracket: %.1f us\n", tot * 1e3 / 10);
  }
  printf("--- do two launches overlap?\n");
  run_pair<64, 64, 2>(2432, 768, 768); run_pair<64, 64, 2>(1216, 768, 768); run_pair<128, 64, 4>(2432, 768, 768);
  run_pair<128, 128, 2>(2432, 3072, 768); run_pair<64, 64, 3>(2432, 768, 3072);
  printf("--- local-branch projection 2432 x 768 x 768\n");
  run<64, 64, 0, 2>(2432, 768, 768);  run<64, 64, 0, 4>(2432, 768, 768);  run<64, 64, 1, 2>(2432, 768, 768);  run<64, 64, 2, 2>(2432, 768, 768);
  run<128, 64, 0, 2>(2432, 768, 768); run<128, 64, 0, 3>(2432, 768, 768); run<128, 64, 0, 4>(2432, 768, 768); run<128, 64, 1, 2>(2432, 768, 768); run<128, 64, 2, 2>(2432, 768, 768);
  run<128, 128, 0, 2>(2432, 768, 768); run<128, 128, 0, 4>(2432, 768, 768); run<128, 128, 1, 2>(2432, 768, 768);
  printf("--- local-branch QKV 2432 x 2304 x 768\n");
  run<64, 128, 0, 2>(2432, 2304, 768); run<128, 128, 0, 2>(2432, 2304, 768); run<128, 128, 0, 3>(2432, 2304, 768);
  run<128, 192, 0, 2>(2432, 2304, 768); run<128, 192, 0, 3>(2432, 2304, 768); run<128, 192, 0, 4>(2432, 2304, 768); run<128, 192, 1, 2>(2432, 2304, 768);
  printf("--- local-branch FFN-in 2432 x 3072 x 768\n");
  run<128, 128, 0, 2>(2432, 3072, 768); run<128, 256, 0, 2>(2432, 3072, 768); run<128, 256, 0, 3>(2432, 3072, 768); run<128, 256, 1, 2>(2432, 3072, 768);
  printf("--- local-branch FFN-out 2432 x 768 x 3072\n");
  run<64, 64, 0, 3>(2432, 768, 3072); run<128, 64, 0, 3>(2432, 768, 3072); run<128, 64, 0, 4>(2432, 768, 3072); run<128, 64, 1, 2>(2432, 768, 3072);
  printf("--- global-branch 1280 rows\n");
  run<64, 64, 0, 2>(1280, 768, 768); run<64, 64, 0, 4>(1280, 768, 768); run<64, 64, 1, 2>(1280, 768, 768);
  run<64, 64, 0, 2>(1280, 2304, 768); run<128, 96, 0, 3>(1280, 2304, 768); run<64, 192, 0, 3>(1280, 2304, 768);
  run<64, 64, 0, 2>(1280, 3072, 768); run<128, 128, 0, 3>(1280, 3072, 768); run<64, 256, 0, 3>(1280, 3072, 768);
  return 0;
}
