// Micro-benchmark: how fast can a CU pull GEMM-shaped tile rows (8 rows x 128 B per wave-instruction) from L2/MALL
// into LDS, (a) with LDS-DMA (global_load_lds_dwordx4), (b) with global_load_dwordx4 -> ds_write_b128.
// usage: ingest_bench   (prints GB/s and B/clk/CU for a few occupancies)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ROWB = 128;
// each block: `rows` tile rows per stage (rows/32 wave-instructions per wave, 4 waves), K sweep of nk stages
template <int ROWS, int MODE, int NST, int ROT, int NW = 4>
__global__ __launch_bounds__(NW * 64) void ingest(const char* __restrict__ src, int ld_bytes, int nk, int npanel, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LPS = ROWS / (8 * NW), STAGE = ROWS * ROWB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int panel = blockIdx.x % npanel;
  const char* p[LPS];
#pragma unroll
  for (int i = 0; i < LPS; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    p[i] = src + (size_t)(panel * ROWS + row) * ld_bytes + (lane & 7) * 16;
  }
  const int kofs = ROT ? (int)((blockIdx.x * 7u) % (unsigned)nk) : 0;
  auto kk = [&](int k) { int x = k + kofs; return x >= nk ? x - nk : x; };
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned acc = 0;
  if (MODE == 0) {
    auto issue = [&](int stage, int k) {
#pragma unroll
      for (int i = 0; i < LPS; ++i) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE + (i * NW + wave) * 1024);
        const char* g = p[i] + (size_t)kk(k) * ROWB;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
      }
    };
    for (int d = 0; d < NST - 1; ++d) issue(d, d);
    int cur = 0;
    for (int k = 0; k < nk; ++k) {
      if (k + NST - 1 < nk) {
        if (NST == 2) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((NST - 2) * LPS) : "memory");
        issue(cur == 0 ? NST - 1 : cur - 1, k + NST - 1);
      } else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      acc += *reinterpret_cast<const unsigned*>(smem + cur * STAGE + threadIdx.x * 4);   // touch the stage
      cur = cur == NST - 1 ? 0 : cur + 1;
    }
  } else if (MODE == 2) {
    constexpr int PFD = 3;
    auto issue = [&](int stage, int k) {
#pragma unroll
      for (int i = 0; i < LPS; ++i) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE + (i * NW + wave) * 1024);
        const char* g = p[i] + (size_t)kk(k) * ROWB;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
      }
    };
    // prefetcher lanes: waves with wave*64 < ROWS, lane l -> row wave*64+l, one dword of the line of K-step k
    const bool pfw = wave * 64 < ROWS;
    const char* prow = src + (size_t)(panel * ROWS + (wave * 64 + lane) % ROWS) * ld_bytes;
    // The prefetch load lands asynchronously: its destination register must stay reserved until the final
    // s_waitcnt, so every asm below takes it as a read-write operand (one live range for the whole loop).
    unsigned dummy = 0;
    issue(0, 0);
    if (pfw) asm volatile("global_load_dword %0, %1, off" : "+v"(dummy) : "v"(prow) : "memory");
    int cur = 0;
    for (int k = 0; k < nk; ++k) {
      if (pfw) asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (k + 1 < nk) issue(cur ^ 1, k + 1);
      if (pfw) {
        const char* g = prow + (size_t)kk(k + PFD < nk ? k + PFD : nk - 1) * ROWB;
        asm volatile("global_load_dword %0, %1, off" : "+v"(dummy) : "v"(g) : "memory");
      }
      acc += *reinterpret_cast<const unsigned*>(smem + cur * STAGE + threadIdx.x * 4);
      cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(dummy) :: "memory");
    acc += dummy & 1;
  } else {
    uint4 r0[LPS], r1[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) r0[i] = *reinterpret_cast<const uint4*>(p[i] + (size_t)kk(0) * ROWB);
#pragma unroll
    for (int i = 0; i < LPS; ++i) r1[i] = *reinterpret_cast<const uint4*>(p[i] + (size_t)kk(1) * ROWB);
    for (int k = 0; k < nk; k += 2) {
#pragma unroll
      for (int i = 0; i < LPS; ++i) *reinterpret_cast<uint4*>(smem + (i * NW + wave) * 1024 + lane * 16) = r0[i];
      if (k + 2 < nk) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) r0[i] = *reinterpret_cast<const uint4*>(p[i] + (size_t)kk(k + 2) * ROWB);
      }
      __syncthreads();
      acc += *reinterpret_cast<const unsigned*>(smem + threadIdx.x * 4);
#pragma unroll
      for (int i = 0; i < LPS; ++i) *reinterpret_cast<uint4*>(smem + STAGE + (i * NW + wave) * 1024 + lane * 16) = r1[i];
      if (k + 3 < nk) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) r1[i] = *reinterpret_cast<const uint4*>(p[i] + (size_t)kk(k + 3) * ROWB);
      }
      __syncthreads();
      acc += *reinterpret_cast<const unsigned*>(smem + STAGE + threadIdx.x * 4);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int ROWS, int MODE, int NST, int ROT = 0, int NW = 4>
int run(const char* name, const char* src, int ld, int nk, int npanel, int blocks, int lds_pad, unsigned* sink) {
  const int smem = NST * ROWS * ROWB + lds_pad;
  CK(hipFuncSetAttribute((const void*)ingest<ROWS, MODE, NST, ROT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((ingest<ROWS, MODE, NST, ROT, NW>), dim3(blocks), dim3(NW * 64), smem, 0, src, ld, nk, npanel, sink);
  CK(hipEventRecord(e0));
  const int rep = 10;
  for (int w = 0; w < rep; ++w) hipLaunchKernelGGL((ingest<ROWS, MODE, NST, ROT, NW>), dim3(blocks), dim3(NW * 64), smem, 0, src, ld, nk, npanel, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / rep, bytes = (double)blocks * nk * ROWS * ROWB;
  printf("%-30s nw %d rot %d rows/stage %3d nst %d blocks %5d (lds %6d): %7.1f us  %6.2f TB/s  %5.1f B/clk/CU\n", name, NW, ROT, ROWS, NST, blocks, smem, us,
         bytes / us / 1e6, bytes / (us * 1e-6) / 2.4e9 / 256);
  return 0;
}

int main() {
  const int K = 3072, ld = K * 2, nk = ld / ROWB, rows_total = 4480;   // x.ff2-like: 3712 + 768 rows of 6 KiB
  char* src; unsigned* sink;
  CK(hipMalloc(&src, (size_t)rows_total * ld)); CK(hipMemset(src, 1, (size_t)rows_total * ld)); CK(hipMalloc(&sink, 4));
  const int np1 = rows_total / 128;
  run<128, 0, 2, 0, 4>("dma", src, ld, nk, np1, 696, 0, sink);
  run<128, 2, 2, 0, 4>("dma + L2 prefetch", src, ld, nk, np1, 696, 0, sink);
  run<128, 0, 2, 0, 4>("dma L2-hot", src, ld, nk, 2, 696, 0, sink);
  run<128, 2, 2, 0, 4>("dma + L2 prefetch, L2-hot", src, ld, nk, 2, 696, 0, sink);
  run<256, 0, 2, 0, 4>("dma 128x128-like 348", src, ld, nk, rows_total / 256, 348, 0, sink);
  run<256, 2, 2, 0, 4>("dma + L2 prefetch 128x128-like 348", src, ld, nk, rows_total / 256, 348, 0, sink);
  return 0;
}
