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
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#ifndef DSRG_ABLATE
#define DSRG_ABLATE 0   // diagnostic builds: 1 = no LDS-DMA in the K loop, 2 = no fragment reads / MFMA
#endif
#include "common.h"

#ifdef DSRG_GEMM_STAMPS   // diagnostic build only (tools/gemm_stamps.py): per-workgroup s_memtime stamps of the tile's phases
__device__ unsigned long long g_gemm_stamps[8192 * 8];
#define GEMM_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int dsrg_debug_gemm_stamps(void* host_dst, int nbytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_gemm_stamps), nbytes, 0, hipMemcpyDeviceToHost);
}
extern "C" int dsrg_debug_gemm_stamps_clear(void) {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_gemm_stamps)) != hipSuccess) return -1;
  return (int)hipMemset(p, 0, sizeof(g_gemm_stamps));
}
#else
#define GEMM_STAMP(i) do {} while (0)
#endif

namespace {

constexpr int ROWB = 128;  // bytes of K per LDS tile row

struct GemmArgs {
  dsrg_gemm_problem_t p[DSRG_MAX_GROUPS];
  int tile_start[DSRG_MAX_GROUPS + 1];
  int nprob;
  int act;
  int gm;          // M-tiles per super-row of the tile walk
  int pw[DSRG_MAX_GROUPS];   // N-tiles per panel of the tile walk (per problem)
  int nt_store;    // 1: C leaves with non-temporal stores
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }
// Swizzle of a TRANSPOSED operand's stage (rows = contraction index): a 32-lane half of ds_read_b64_tr_b16 takes the same
// 32-byte column pair of rows r..r+3 and r+8..r+11 (r % 4 == 0); even / odd rows already sit in different halves of the 64
// banks, so the pair index is XORed with (row bit 1, row bit 3): the 8 rows then cover 8 different 32-byte bank groups.
__device__ __forceinline__ int swt(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }

// Sum over groups of L (8 or 16) consecutive lanes, every lane of a group receiving the total: DPP moves inside a row of 16
// lanes (xor 1, xor 2 by quad_perm; the two quads of an 8-lane half by row_half_mirror; the halves by row_mirror) instead of
// ds_bpermute round trips.  Lanes that are switched off contribute 0 (bound_ctrl).
template <int L>
__device__ __forceinline__ float dpp_group_sum(float v) {
  static_assert(L == 8 || L == 16, "8 or 16 lanes");
  auto mv = [](float x, auto ctrl) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true)); };
  v += mv(v, std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
  v += mv(v, std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
  v += mv(v, std::integral_constant<int, 0x141>{});      // row_half_mirror
  if constexpr (L == 16) v += mv(v, std::integral_constant<int, 0x140>{});     // row_mirror
  return v;
}

template <bool SHORT_GELU>
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case DSRG_ACT_GELU: return SHORT_GELU ? gelu_erf_short(v) : gelu_erf(v);
    case DSRG_ACT_RELU: return fmaxf(v, 0.0f);
    case DSRG_ACT_TANH: return tanhf(v);
    default: return v;
  }
}
// Epilogue specialisations (in-kernel stamps, tools/gemm_stamps.py): with the activation a run-time switch per ELEMENT the
// row pass of the epilogue was ~35 scalar branches per 8 columns, 1100-1300 cycles per pass even with no activation at all
// (a 128x128 tile: 10 k cycles of "rows out" against an 11 k-cycle K loop).  The common cases are therefore compiled as
// straight-line code and chosen by ONE uniform branch per tile: ACT / RACT >= 0 are compile-time activations, -1 keeps the
// run-time switch (ReLU / tanh heads and poolers, rare combinations).
template <int ACT, bool SHORT_GELU>
__device__ __forceinline__ float apply_act_c(float v, int act) {
  if constexpr (ACT == DSRG_ACT_NONE) return v;
  else if constexpr (ACT == DSRG_ACT_GELU) return SHORT_GELU ? gelu_erf_short(v) : gelu_erf(v);
  else return apply_act<SHORT_GELU>(v, act);
}
template <int RACT>
__device__ __forceinline__ float act_grad_c(float z, int res_act) {
  if constexpr (RACT == DSRG_ACT_GELU) return act_grad(z, DSRG_ACT_GELU);
  else return act_grad(z, res_act);
}

// FUSE: the LayerNorm-fusion epilogue (include/dsrg_hip.h, dsrg_gemm_problem_t), launched only when a problem of the group
// asks for it, so the plain kernels carry none of its registers.  Two kinds, so that each carries only its own operands:
//   FUSE_A  (1): the A operand is an un-normalised row y with partial statistics (a_stats): the projections that read LN(y);
//   FUSE_RS (2): the residual is un-normalised (r_stats) and / or the launch writes the partial statistics of what it stores
//                (stats_out): the dense + residual launches that produce the next y.
// Partial statistics are float2 (sum, sum of squares) per row and 64-column block, stored BLOCK-major ([N/64][M]) so that a
// wave's loads and stores of one block are contiguous (round 3 stamps, profiles/r03a_gemm_stamps_fuse.txt: with the row-major
// layout every partial load touched 64 lines, 1100-1700 cycles of prologue in front of the first stage).
constexpr int FUSE_NONE = 0, FUSE_A = 1, FUSE_RS = 2;
// WAVES_M x WAVES_N waves (2x2 = 256 threads for every tile but the 256x128 one, which runs 4x2 = 512 threads: two waves
// per SIMD of 64x64 each, so that a workgroup that owns its CU still has a second wave per SIMD to issue LDS reads and
// DMAs while the first one's MFMAs run).
// TRANS: operand storage of the backward GEMMs (dsrg_gemm_problem_t::trans; bf16, 64x64 tile only).  Bit 0: A is stored
// [K, M] (contraction-major rows), bit 1: W is stored [K, N].  A transposed operand's stage is 64 contraction rows x 64
// output columns (again 128-byte rows, the same DMA pieces); its MFMA fragments come out of LDS through the gfx950
// transposing read ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16-lane group, two reads = the 8 k values a lane holds),
// so dX = dY.W (TRANS 2) and dW = dY^T.X (TRANS 3) need no transposed copy of W, dY or X in HBM.
template <typename T, typename OutT, int BM, int BN, int NST, int FUSE = 0, int WAVES_M = 2, int WAVES_N = 2, int TRANS = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_nt_kernel(const GemmArgs args) {
  GEMM_STAMP(0);
  constexpr bool TA = (TRANS & 1) != 0, TW = (TRANS & 2) != 0;
  // A transposed operand of a tile wider than 64 is staged as BM/64 (BN/64) blocks of 64 contraction rows x 64 columns, each
  // with the 64x64 tile's own layout, one after the other (a "row" of the stage below is block * 64 + contraction row).
  static_assert(TRANS == 0 || (BM % 64 == 0 && BN % 64 == 0 && sizeof(T) == 2 && FUSE == 0), "transposed operands: bf16, tiles of 64-column blocks");
  constexpr int PER16 = Elem<T>::PER16;
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int NWAVES = WAVES_M * WAVES_N, THREADS = 64 * NWAVES;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;     // wave tile
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int TILE_BYTES = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- which problem / tile.  Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2), so
  // the hardware block id is first remapped to give every XCD one CONTIGUOUS run of logical tile ids (bijective
  // for any grid size), and logical ids walk the tile grid in super-rows of GM M-tiles with N fastest-but-one:
  // the ~32 tiles an XCD runs at a time then form a compact GM x (32/GM) patch that shares A and W panels in L2.
  int bid;
#define DSRG_REMAP_BID()                                                                                   \
  do {                                                                                                     \
    const int nwg = gridDim.x, orig = blockIdx.x;                                                          \
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;                                                 \
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);                     \
  } while (0)
  // The launch record comes in through scalar loads from the kernarg segment, 300-500+ cycles each when they miss.  Left to
  // the compiler they are sunk to their first uses: 7 dependent load -> wait round trips before the first DMA could be
  // issued (in-kernel stamps: 2500-5300 cycles from kernel entry to the first stage, a third of a 64x64 tile's lifetime).
  // So: everything at a fixed offset (tile starts, panel widths, problem 0) is read in ONE batch at entry - the asm
  // statements below only pin the values into SGPRs at that point - and only the workgroups of problems 1.. pay a second
  // batch.  tile_start[] past the last problem is INT_MAX (host), so the search needs no problem count.
  dsrg_gemm_problem_t P = args.p[0];
#define DSRG_PIN_PROBLEM(P_)                                                                                                        \
  do {                                                                                                                              \
    asm volatile("" ::"s"(P_.A), "s"(P_.W), "s"(P_.bias), "s"(P_.residual), "s"(P_.C), "s"(P_.M), "s"(P_.N), "s"(P_.K), "s"(P_.lda), \
                 "s"(P_.ldw), "s"(P_.ldr), "s"(P_.ldc), "s"(P_.c_pre), "s"(P_.ldcp), "s"(P_.res_act), "s"(P_.a_colsum), "s"(P_.reserved0_)); \
    /* (ln_eps is NOT pinned: the allocator gave it a register of the wide load in flight above it, which split the entry   \
       batch into two dependent round trips, ~1000 cycles of prologue in the round-3 stamps; it is read in the epilogue) */    \
    if constexpr (FUSE == FUSE_A) asm volatile("" ::"s"(P_.a_stats), "s"(P_.a_c1), "s"(P_.a_np));                                   \
    if constexpr (FUSE == FUSE_RS)                                                                                                  \
      asm volatile("" ::"s"(P_.r_stats), "s"(P_.r_gamma), "s"(P_.r_beta), "s"(P_.stats_out), "s"(P_.r_np));                          \
  } while (0)
  // ... together with the grid size (gridDim.x is a scalar load from the kernarg segment too: with the block-id remap first,
  // every launch paid that round trip alone before the batch was issued) and the tile starts of the problem search
  {
    const int nwg_ = gridDim.x;
    asm volatile("" ::"s"(nwg_), "s"(args.tile_start[1]), "s"(args.tile_start[2]), "s"(args.tile_start[3]), "s"(args.tile_start[4]),
                 "s"(args.tile_start[5]), "s"(args.tile_start[6]), "s"(args.tile_start[7]), "s"(args.tile_start[8]), "s"(args.pw[0]));
    // (every dword of the wide scalar loads is named here or in DSRG_PIN_PROBLEM - reserved0_ and tile_start[8] included:
    // the allocator otherwise hands a dead dword's register to a later load, which then has to wait for the wide one and
    // splits the batch into two dependent round trips)
  }
  DSRG_PIN_PROBLEM(P);                      // (in front of the search: its loads go out with the tile starts')
  DSRG_REMAP_BID();
#undef DSRG_REMAP_BID
  int g = 0, tstart = 0, PW = args.pw[0];
#pragma unroll
  for (int i = 1; i < DSRG_MAX_GROUPS; ++i)
    if (bid >= args.tile_start[i]) { g = i; tstart = args.tile_start[i]; PW = args.pw[i]; }
  if (g != 0) {
    P = args.p[g];
    DSRG_PIN_PROBLEM(P);
  }
#undef DSRG_PIN_PROBLEM
  const int M = P.M, N = P.N, K = P.K;
  const int t = bid - tstart;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  // Walk: N-panels of PW tiles (outermost), inside a panel super-rows of GM M-tiles, inside a super-row N, then M
  // fastest.  The ~T/8 consecutive tiles one XCD runs then cover a compact (T/8/PW) x PW patch of the tile grid whose
  // A rows + W rows fit that XCD's 4 MiB L2 (PW is chosen on the host for that), instead of a stripe that needs all of W.
  const int GM = args.gm;
  const int full = tiles_m * PW;                        // tiles in a full panel
  const int pnl = t / full, in_pnl = t - pnl * full;
  const int wn_p = min(PW, tiles_n - pnl * PW);         // width of this (possibly narrower, last) panel
  const int per_group = GM * wn_p;
  const int grp = in_pnl / per_group, in_grp = in_pnl - grp * per_group;
  const int gm = min(GM, tiles_m - grp * GM);          // rows in this (possibly short) super-row
  const int tm = grp * GM + in_grp % gm, tn = pnl * PW + in_grp / gm;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int r16 = lane & 15, q = lane >> 4;

  const T* __restrict__ A = reinterpret_cast<const T*>(P.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(P.W);

  // ---- this thread's place in the epilogue (fixed 16-B column chunk, rows orow0 + k*RPP) and its bias values, loaded
  // here so that their latency hides behind the first stage's
  const float* __restrict__ bias = P.bias;
  const T* __restrict__ Rp = reinterpret_cast<const T*>(P.residual);
  constexpr int CW = sizeof(OutT) == 2 ? 8 : 4;     // output columns per thread = one 16-byte store (stamps: with 8-byte
                                                    // bf16 stores the epilogue was store-issue-bound, ~7 B/clk per CU)
  constexpr int CPR = BN / CW;                      // threads per image row
  constexpr int RPP = THREADS / CPR;                // rows per pass (threads past RPP*CPR idle when THREADS % CPR != 0)
  const int oc = tid % CPR, orow0 = tid / CPR;
  const int n = n0 + oc * CW;
  const bool vec = (n + CW - 1 < N) && (P.ldc % CW == 0) && (Rp == nullptr || (P.ldr % CW == 0 && ((uintptr_t)Rp & 15) == 0));
  // training forward: the pre-activation (accumulator + bias) also leaves, in the input dtype, for the backward (c_pre);
  // training backward: res_act > 0 multiplies by act'(residual) instead of adding the residual (dX = (dY.W) * act'(z))
  T* __restrict__ Cpre = reinterpret_cast<T*>(P.c_pre);
  const int res_act = P.res_act;
  float bv[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) bv[e] = 0.f;
  if (bias && n < N) {
    if (n + CW - 1 < N) {
#pragma unroll
      for (int h = 0; h < CW / 4; ++h) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + n + 4 * h);
        bv[4 * h] = b4.x; bv[4 * h + 1] = b4.y; bv[4 * h + 2] = b4.z; bv[4 * h + 3] = b4.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < CW; ++e)
        if (n + e < N) bv[e] = bias[n + e];
    }
  }

  // ---- staging: LDS-DMA (global_load_lds_dwordx4), NST-deep ring, counted vmcnt.
  // One wave-instruction moves 8 tile rows (64 lanes x 16 B = 1 KiB, lane-linear in LDS); the XOR swizzle is
  // applied to the per-lane SOURCE chunk, the LDS image stays linear (row*128 + pos*16), and the fragment reads
  // use the same involution.  Rows past M / N are clamped to the last valid row (their results are never stored).
  // Addresses are scalar base (A or W, advanced along K in SGPRs) + a per-lane 32-bit byte offset that never changes,
  // so one DMA costs three scalar instructions and no vector ALU work.
  constexpr int TR = BM + BN;           // tile rows per stage (A rows then W rows)
  constexpr int LPS = TR / (8 * NWAVES);   // LDS-DMA instructions per wave per stage
  constexpr int LPA = BM / (8 * NWAVES);   // ... of which the first LPA fetch A rows
  static_assert(TR % (8 * NWAVES) == 0 && BM % (8 * NWAVES) == 0, "a DMA piece is 8 rows per wave");
  unsigned voff[LPS];
#pragma unroll
  for (int i = 0; i < LPS; ++i) {
    const int row = (i * NWAVES + wave) * 8 + (lane >> 3);  // row of the stage
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (i < LPA) {
      if constexpr (TA) {     // row = (block, contraction row) of the stage; the 16-byte chunk holds 8 output rows (clamped whole: M % 8 == 0)
        const int kr = row & 63, blk = row >> 6;
        const int col = min(m0 + blk * 64 + (((lane & 7) ^ swt(kr)) << 3), M - 8);
        voff[i] = (unsigned)kr * (unsigned)(P.lda * (int)sizeof(T)) + (unsigned)col * (unsigned)sizeof(T);
      } else voff[i] = (unsigned)min(m0 + row, M - 1) * (unsigned)(P.lda * (int)sizeof(T)) + chunk * 16;
    } else {
      if constexpr (TW) {
        const int rw = row - BM, kr = rw & 63, blk = rw >> 6;
        const int col = min(n0 + blk * 64 + (((lane & 7) ^ swt(kr)) << 3), N - 8);
        voff[i] = (unsigned)kr * (unsigned)(P.ldw * (int)sizeof(T)) + (unsigned)col * (unsigned)sizeof(T);
      } else voff[i] = (unsigned)min(n0 + row - BM, N - 1) * (unsigned)(P.ldw * (int)sizeof(T)) + chunk * 16;
    }
  }
  // TRANS 3 only: K (the rows of both operands) need not be a multiple of 64.  The last, partial K-step fetches rows
  // >= krem from row 0 of the step instead (valid memory, finite unless the operand itself is not) and the A rows among
  // them are zeroed in LDS once they have landed (land()): 0 x finite adds nothing.
  const int krem = TRANS == 3 ? (K & (BK - 1)) : 0;
  unsigned voff_last[TRANS == 3 ? LPS : 1];
  if constexpr (TRANS == 3) {
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
      const int row = ((i * NWAVES + wave) * 8 + (lane >> 3)) & (BK - 1);
      voff_last[i] = (krem && row >= krem) ? voff[i] - (unsigned)row * (unsigned)((i < LPA ? P.lda : P.ldw) * (int)sizeof(T)) : voff[i];
    }
  }
  // The DMA is issued from inline asm on purpose: hipcc does not count asm memory operations, so it cannot
  // drain the ring with a vmcnt(0) in front of the fragment reads (it does for the builtin form); completion is
  // tracked by the counted s_waitcnt below.  M0 (LDS destination base) is written in the statement that uses it.
  const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024);
  auto issue1 = [&](int i, int stage, int k0) {
    const unsigned dst = ldsw + stage * TILE_BYTES + i * (NWAVES * 1024);
    const T* base = i < LPA ? (TA ? A + (size_t)k0 * P.lda : A + k0) : (TW ? W + (size_t)k0 * P.ldw : W + k0);
    unsigned keep, vo = voff[i];
    if constexpr (TRANS == 3) vo = (k0 + BK > K) ? voff_last[i] : vo;       // uniform condition: the partial last K-step
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(base), "s"(dst) : "memory");
  };
  auto issue = [&](int stage, int k0) {
#pragma unroll
    for (int i = 0; i < LPS; ++i) issue1(i, stage, k0);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;          // K % BK != 0 only with TRANS 3 (host-checked)
  // TRANS bit 0: column sums of the A operand (the bias gradient that goes with dW = dY^T.X), by the waves that own the
  // tile's rows in the first column of tiles: one more MFMA per A fragment against a fragment of ones.
  f32x4 accdb[TA ? MT : 1];
  bool dbw = false;
  if constexpr (TA) {
#pragma unroll
    for (int i = 0; i < MT; ++i) accdb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    dbw = __builtin_amdgcn_readfirstlane((int)(P.a_colsum != nullptr && tn == 0 && wn == 0)) != 0;
  }
  const uint4 ones4 = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
  constexpr int D = NST - 1;                 // prefetch distance in K-steps
  // One fragment read (index r < MT+NT: A-fragment r, or W-fragment r-MT) of half K-step `kk` of a stage.
  auto rd1 = [&](uint4 (&fa)[MT], uint4 (&fw)[NT], int r, int stage, int kk) {
    const char* sa = smem + stage * TILE_BYTES;
    const int chunk = q + 4 * kk;
    // transposed stage: this lane's group (q) takes k rows 32kk+8q .. +7 as two 4-row blocks of the fragment's 16 columns;
    // the lane supplies the address of block row (r16 >> 2), columns 4 (r16 & 3) .. +3, and receives column r16
    auto trd = [&](const char* tile, int col16) -> uint4 {
      typedef short v4s __attribute__((ext_vector_type(4)));
      const int krow = 32 * kk + 8 * q + (r16 >> 2);
      const int c = (col16 >> 3) + ((r16 & 3) >> 1);
      const char* p = tile + (col16 >> 6) * (64 * ROWB) + krow * ROWB + (((c & 7) ^ swt(krow)) << 4) + ((r16 & 1) << 3);     // rows +4: same swizzle, +512 bytes
      const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
      const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 4 * ROWB));
      const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };
    if (r < MT) {
      if constexpr (TA) fa[r] = trd(sa, wm * WM + r * 16);
      else fa[r] = *reinterpret_cast<const uint4*>(sa + swz(wm * WM + r * 16 + r16, chunk));
    } else {
      if constexpr (TW) fw[r - MT] = trd(sa + BM * ROWB, wn * WN + (r - MT) * 16);
      else fw[r - MT] = *reinterpret_cast<const uint4*>(sa + BM * ROWB + swz(wn * WN + (r - MT) * 16 + r16, chunk));
    }
  };
  // The bf16 MFMA is an asm statement with the accumulator tied to an AGPR: (1) the register allocator otherwise
  // shuttles the 128-192 accumulator registers of the 256-row tiles between the two register files once the
  // fragments are double-buffered; (2) volatile statements keep the order written below, which IS the schedule.
  auto mfma1 = [&](const uint4& fa, const uint4& fw, f32x4& c) {
    if constexpr (sizeof(T) == 2) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                   : "+a"(c) : "v"(__builtin_bit_cast(u32x4, fw)), "v"(__builtin_bit_cast(u32x4, fa)));
    } else {
      const float* pw = reinterpret_cast<const float*>(&fw);
      const float* pa = reinterpret_cast<const float*>(&fa);
#pragma unroll
      for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[e], pa[e], c, 0, 0, 0);
    }
  };
  // Column-sum MFMA (ones x A fragment).  The compiler materialises the constant fragment with VALU moves right in front of
  // the statement and cannot see that an MFMA reads it: the s_nop covers the VALU-write -> MFMA-read distance (measured:
  // without it the first of two such MFMAs read stale registers).
  auto mfma_db = [&](const uint4& fa, f32x4& c) {
    if constexpr (sizeof(T) == 2) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                   : "+a"(c) : "v"(__builtin_bit_cast(u32x4, ones4)), "v"(__builtin_bit_cast(u32x4, fa)));
    }
  };
  // Half a K-step: the MT*NT MFMAs on (fa, fw), with the MT+NT fragment reads of the NEXT half step (into na, nw) and
  // the refill DMAs [P0, P1) of a stage spread evenly between them - a workgroup that is alone on its CU (the 256-row
  // tiles) has no other wave to hide LDS latency or DMA issue behind, so the overlap has to be inside the wave.  A DMA
  // piece holds the issuing wave for 60-180 cycles while the CU's one address path (64 B/clk: 16 clk per 1 KiB piece,
  // shared by the four waves) works through the queue, hence at most one piece per ~4 MFMAs.
  auto half = [&](const uint4 (&fa)[MT], const uint4 (&fw)[NT], uint4 (&na)[MT], uint4 (&nw)[NT], int rstage, int rkk,
                  auto p0_tag, auto p1_tag, int dstage, int dk0) {
    constexpr int P0 = decltype(p0_tag)::value, P1 = decltype(p1_tag)::value, NP = P1 - P0;
    constexpr int NM = MT * NT, R = MT + NT;
    int r = 0, d = 0;
    // The column-sum MFMAs go FIRST: the compiler may move accdb between registers at a phase boundary, and it does not
    // know an asm MFMA is still writing it - the NM MFMAs below are the cover for that.
    if constexpr (TA) {
      if (dbw) {
#pragma unroll
        for (int i = 0; i < MT; ++i) mfma_db(fa[i], accdb[i]);
      }
    }
#pragma unroll
    for (int idx = 0; idx < NM; ++idx) {
#if DSRG_ABLATE != 2
      mfma1(fa[idx / NT], fw[idx % NT], acc[idx / NT][idx % NT]);
#endif
#if DSRG_ABLATE != 1
      if (d < NP && d * NM <= idx * NP) { issue1(P0 + d, dstage, dk0); ++d; }
#endif
      if (r < R && r * NM <= idx * R) { rd1(na, nw, r, rstage, rkk); ++r; }
    }
#pragma unroll
    for (; r < R; ++r) rd1(na, nw, r, rstage, rkk);
#if DSRG_ABLATE != 1
#pragma unroll
    for (; d < NP; ++d) issue1(P0 + d, dstage, dk0);
#endif
  };
  // stage `st` has landed once all but the DMAs of the (up to D-1) younger stages are done; wait + barrier are one
  // statement with a memory clobber (no LDS access may move across it).  After the barrier every wave has also
  // finished READING stage st-1 (its fragment loads are waited for first), so that buffer is refilled right away.
  auto land = [&](int st) {
    const int younger = min(D - 1, nk - 1 - st);
    if constexpr (TRANS == 3) {
      if (krem && st == nk - 1) {              // this wave's own DMAs are done: zero the A rows past K that it fetched
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        char* sa = smem + (st % NST) * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < LPA; ++i) {
          const int row = (i * NWAVES + wave) * 8 + (lane >> 3);
          if ((row & (BK - 1)) >= krem) *reinterpret_cast<uint4*>(sa + row * ROWB + ((lane & 7) << 4)) = make_uint4(0u, 0u, 0u, 0u);
        }
      }
    }
    if (D >= 3 && younger == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * LPS) : "memory");
    else if (D >= 2 && younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  // The refill of stage s goes into the buffer of stage s-1-D, free from the barrier in the middle of K-step s-1-D on.
  // With D >= 2 its pieces are split over the second half of that K-step ([0,H)) and the first half of the next one
  // ([H,LPS)); a 2-deep ring needs the stage one half step later, so all of them go into the second half.
  constexpr int H = D >= 2 ? LPS / 2 : LPS;
  using Z = std::integral_constant<int, 0>;
  using PH = std::integral_constant<int, H>;
  using PL = std::integral_constant<int, LPS>;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nk) issue(d, d * BK);
  uint4 fa0[MT], fw0[NT], fa1[MT], fw1[NT];
  land(0);
  GEMM_STAMP(1);
  // ---- FUSE: this thread's share of the LayerNorm partials of its tile's rows (row = tid % BM, every TG-th partial from
  // tid / BM on), issued once the first stage has LANDED (block-major table: a wave's 64 rows of one block are 512 contiguous bytes) and
  // summed after the K loop.  Completion is counted in issue order: in front of the ring's first DMAs these loads held the
  // first stage back by their own misses, right behind them they made land(0) wait for the second stage as well; here they
  // are older than every refill, so the counted waits of the K loop cover them (stricter by a few pieces, never wrong).
  constexpr int TG = THREADS / BM > 0 ? THREADS / BM : 1;     // threads per tile row (4 at BM = 64, 2 at BM = 128)
  constexpr int SP = 16 / TG;                                 // partial loads per thread: a_np, r_np <= 16
  float2 pl[FUSE ? SP : 1];
  const float* __restrict__ stp = nullptr;
  int snp = 1;
  if constexpr (FUSE == FUSE_A) { stp = P.a_stats; snp = P.a_np; }
  if constexpr (FUSE == FUSE_RS) { stp = P.r_stats; snp = P.r_np; }
  if constexpr (FUSE != 0) {
    static_assert(THREADS % BM == 0 && TG * SP == 16, "the fused epilogue sums 16 / TG partials per thread");
    const int srow = min(m0 + (tid % BM), M - 1), sg = tid / BM;
#pragma unroll
    for (int j = 0; j < SP; ++j) pl[j] = make_float2(0.f, 0.f);
    if (stp) {
      const float2* ps_ = reinterpret_cast<const float2*>(stp) + srow;
#pragma unroll
      for (int j = 0; j < SP; ++j) pl[j] = ps_[(size_t)min(sg + TG * j, snp - 1) * M];
    }
  }
  // ... and the per-column vectors of this thread's CW output columns (row sums of the folded weight; residual gamma / beta)
  float c1v[FUSE == FUSE_A ? CW : 1], rgv[FUSE == FUSE_RS ? CW : 1], rbv[FUSE == FUSE_RS ? CW : 1];
  if constexpr (FUSE == FUSE_A) {
#pragma unroll
    for (int e = 0; e < CW; ++e) c1v[e] = 0.f;
    if (stp && n + CW - 1 < N) {
#pragma unroll
      for (int h = 0; h < CW / 4; ++h) {
        const float4 t4 = *reinterpret_cast<const float4*>(P.a_c1 + n + 4 * h);
        c1v[4 * h] = t4.x; c1v[4 * h + 1] = t4.y; c1v[4 * h + 2] = t4.z; c1v[4 * h + 3] = t4.w;
      }
    }
  }
  if constexpr (FUSE == FUSE_RS) {
#pragma unroll
    for (int e = 0; e < CW; ++e) { rgv[e] = 1.f; rbv[e] = 0.f; }
    if (stp && n + CW - 1 < N) {
#pragma unroll
      for (int h = 0; h < CW / 4; ++h) {
        const float4 g4 = *reinterpret_cast<const float4*>(P.r_gamma + n + 4 * h), b4 = *reinterpret_cast<const float4*>(P.r_beta + n + 4 * h);
        rgv[4 * h] = g4.x; rgv[4 * h + 1] = g4.y; rgv[4 * h + 2] = g4.z; rgv[4 * h + 3] = g4.w;
        rbv[4 * h] = b4.x; rbv[4 * h + 1] = b4.y; rbv[4 * h + 2] = b4.z; rbv[4 * h + 3] = b4.w;
      }
    }
  }

#if DSRG_ABLATE != 1
  if (D < nk) {
#pragma unroll
    for (int i = 0; i < H; ++i) issue1(i, D, D * BK);
  }
#endif
#pragma unroll
  for (int r = 0; r < MT + NT; ++r) rd1(fa0, fw0, r, 0, 0);
  // Straight-line phases (steady state, last refill, drain, last step) rather than one loop with branches: every
  // control-flow merge inside the loop makes the register allocator copy the whole accumulator.
  int cur = 0, prv = NST - 1, kt = 0;
  for (; kt + 1 + D < nk; ++kt) {
    const int nxt = cur == NST - 1 ? 0 : cur + 1;
    half(fa0, fw0, fa1, fw1, cur, 1, PH{}, PL{}, prv, (kt + D) * BK);          // rest of stage kt+D
    if (D >= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * LPS) : "memory");
    else if (D == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    half(fa1, fw1, fa0, fw0, nxt, 0, Z{}, PH{}, cur, (kt + 1 + D) * BK);        // start of stage kt+1+D
    prv = cur; cur = nxt;
  }
  if (kt + D < nk && kt + 1 < nk) {                                              // K-step nk-1-D: the last refill ends here
    const int nxt = cur == NST - 1 ? 0 : cur + 1;
    half(fa0, fw0, fa1, fw1, cur, 1, PH{}, PL{}, prv, (kt + D) * BK);
    land(kt + 1);
    half(fa1, fw1, fa0, fw0, nxt, 0, Z{}, Z{}, 0, 0);
    prv = cur; cur = nxt; ++kt;
  }
  for (; kt + 1 < nk; ++kt) {
    const int nxt = cur == NST - 1 ? 0 : cur + 1;
    half(fa0, fw0, fa1, fw1, cur, 1, Z{}, Z{}, 0, 0);
    land(kt + 1);
    half(fa1, fw1, fa0, fw0, nxt, 0, Z{}, Z{}, 0, 0);
    cur = nxt;
  }
  half(fa0, fw0, fa1, fw1, cur, 1, Z{}, Z{}, 0, 0);
  if constexpr (TA) {
    if (dbw) {
#pragma unroll
      for (int i = 0; i < MT; ++i) mfma_db(fa1[i], accdb[i]);
    }
  }
#if DSRG_ABLATE != 2
#pragma unroll
  for (int idx = 0; idx < MT * NT; ++idx) mfma1(fa1[idx / NT], fw1[idx % NT], acc[idx / NT][idx % NT]);
#endif
  GEMM_STAMP(2);
  // the asm MFMAs are invisible to the compiler's hazard recogniser: cover the XDL-write -> VALU/LDS-read distance
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  if constexpr (TA) {        // D[i][j] = sum_k ones[i][k] * A[k][j]: every register of a lane is the column sum of its A column r16
    if (dbw && q == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * WM + i * 16 + r16;
        if (m < M) P.a_colsum[m] = accdb[i][0];
      }
    }
  }
  // ---- epilogue.  A lane holds C[m][n..n+3] for 16 different rows per store, i.e. 32/64-byte row pieces: written
  // straight to HBM that pattern, not the K loop, bounds these mid-size GEMMs.  The raw accumulators therefore go
  // through LDS (fp32 [BM][BN] image over the drained ring, 16-B chunk index XOR (row & 7): conflict-free both ways)
  // and leave as whole rows: 32 consecutive lanes cover BN consecutive columns (256 B bf16 / 512 B f32 per row),
  // with bias, activation and the residual (read with the same coalescing) applied on the way out.
  OutT* __restrict__ C = reinterpret_cast<OutT*>(P.C);
  const int act = args.act;
  // Tiles whose fp32 image does not fit the ring (256x192: 192 KiB) leave in two passes, one per wave row.
  constexpr int EPASS = (BM * BN * 4 > NST * TILE_BYTES) ? 2 : 1;
  constexpr int EROWS = BM / EPASS;
  static_assert(EPASS == 1 || EROWS % WM == 0, "two-pass epilogue splits the tile between wave rows");
  static_assert(EPASS == 1 || EROWS * BN * 4 <= NST * TILE_BYTES, "a half image must fit the ring");
  float* cimg = reinterpret_cast<float*>(smem);
  // The residual pieces of this thread's rows are fetched NOW, ahead of the two barriers and the image round trip
  // (in-kernel stamps: the epilogue was 30-45 % of a tile's lifetime, a third of it exposed global-load latency).
  constexpr int NPASS = (EROWS + RPP - 1) / RPP;
  constexpr bool PRE_RES = EPASS == 1 && NPASS <= 16;
  typedef typename std::conditional<CW == 8, uint4, typename std::conditional<sizeof(T) == 2, uint2, float4>::type>::type RawRes;
  RawRes rres[PRE_RES ? NPASS : 1];
  if (PRE_RES && Rp && vec && n < N && orow0 < RPP) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int r = orow0 + ps * RPP, m = m0 + r;
      if (r < EROWS && m < M) rres[ps] = *reinterpret_cast<const RawRes*>(Rp + (size_t)m * P.ldr + n);
    }
  }
  // FUSE: the TG threads of a tile row exchange their partial sums through LDS.  The tables live BEHIND ring and image
  // (launch() adds their bytes), so they are written here, before the barrier that frees the ring: [4][BM] partial sums,
  // then [BM] (mean, rstd), finished by the first BM threads between the epilogue's two barriers.
  constexpr int LNOFF = (NST * TILE_BYTES > BM * BN * 4) ? NST * TILE_BYTES : BM * BN * 4;
  // The summation order is the same for every tile shape (results do not depend on the plan): partial p belongs to group
  // p % 4, a group is summed in increasing p, the four group sums are added in order 0..3.
  constexpr int NG = 4;
  float2* lnpart = reinterpret_cast<float2*>(smem + LNOFF);              // [NG][BM]
  float2* lnfin = lnpart + NG * BM;                                      // [BM]
  if constexpr (FUSE != 0) {
    static_assert(NG % TG == 0, "a thread owns NG / TG groups");
    constexpr int GPT = NG / TG;                   // groups per thread (1 at BM = 64, 2 at BM = 128)
    const int sg = tid / BM;
    // The partials loaded in the prologue become visible to the optimiser only HERE: left alone it hoisted the first
    // addition up to the loads and waited for them (s_waitcnt vmcnt(3) in front of the first DMA: a whole cold-miss latency,
    // 1200-2000 cycles of prologue in the round-3 stamps).
#pragma unroll
    for (int j = 0; j < SP; ++j) asm volatile("" : "+v"(pl[j].x), "+v"(pl[j].y));
    float2 sa[GPT];
#pragma unroll
    for (int gi = 0; gi < GPT; ++gi) sa[gi] = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < SP; ++j) {                 // partial sg + TG*j: group (sg + TG*j) % NG = sg + TG*(j % GPT)
      if (sg + TG * j < snp) { sa[j % GPT].x += pl[j].x; sa[j % GPT].y += pl[j].y; }
    }
#pragma unroll
    for (int gi = 0; gi < GPT; ++gi) lnpart[(sg + TG * gi) * BM + (tid % BM)] = sa[gi];
  }
  float2* __restrict__ sto = nullptr;
  if constexpr (FUSE == FUSE_RS) sto = reinterpret_cast<float2*>(P.stats_out);
  auto rows_out = [&](auto act_tag, auto ract_tag) {
  constexpr int ACT = decltype(act_tag)::value, RACT = decltype(ract_tag)::value;   // -1: run-time (see apply_act_c)
#pragma unroll
  for (int ep = 0; ep < EPASS; ++ep) {
    __syncthreads();                                // every wave is done reading the last stage / the previous image
    GEMM_STAMP(5);
    if constexpr (FUSE != 0) {
      static_assert(FUSE == 0 || EPASS == 1, "fused epilogue: one pass");
      if (tid < BM) {                              // (mean, rstd) of tile row `tid`: over K for the A operand, over N for the residual
        float2 t0 = make_float2(0.f, 0.f);
#pragma unroll
        for (int g4 = 0; g4 < NG; ++g4) { const float2 u0 = lnpart[g4 * BM + tid]; t0.x += u0.x; t0.y += u0.y; }
        const float inv = 1.0f / (float)(FUSE == FUSE_A ? K : N);
        const float mu = t0.x * inv;
        lnfin[tid] = stp ? make_float2(mu, rsqrtf(fmaxf(t0.y * inv - mu * mu, 0.f) + P.ln_eps)) : make_float2(0.f, 1.f);
      }
    }
    if (EPASS == 1 || (wm * WM) / EROWS == ep) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * WM - (EPASS == 1 ? 0 : ep * EROWS) + i * 16 + r16;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int chunk = (wn * WN + j * 16) / 4 + q;
          *reinterpret_cast<f32x4*>(cimg + row * BN + ((chunk ^ (row & 7)) << 2)) = acc[i][j];
        }
      }
    }
    __syncthreads();
    GEMM_STAMP(6);
    if (n < N && orow0 < RPP) {
      // the image chunks of all passes are read up front (unconditionally: rows clamped into the image), so that the passes'
      // LDS latencies overlap - the early exit below otherwise keeps every pass's read behind the previous pass's branch
      constexpr int NCH = CW / 4;
      constexpr bool PRE_IMG = NPASS * NCH <= 16 && NWAVES <= 4;
      f32x4 img[PRE_IMG ? NPASS : 1][NCH];
      float2 msv[(PRE_IMG && FUSE != 0) ? NPASS : 1];
      if constexpr (PRE_IMG) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int rr = min(orow0 + ps * RPP, EROWS - 1);
#pragma unroll
          for (int h = 0; h < NCH; ++h) img[ps][h] = *reinterpret_cast<const f32x4*>(cimg + rr * BN + (((oc * NCH + h) ^ (rr & 7)) << 2));
          if constexpr (FUSE != 0) msv[ps] = lnfin[rr];
        }
      }
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int r = orow0 + ps * RPP;
        const int m = m0 + ep * EROWS + r;
        if (r >= EROWS || m >= M) break;
        float2 ms = make_float2(0.f, 1.f);           // (mean, rstd) of this row: A operand over K (FUSE_A), residual over N (FUSE_RS)
        if constexpr (FUSE != 0) {
          if constexpr (PRE_IMG) ms = msv[ps];
          else ms = lnfin[r];
        }
        float v[CW];
#pragma unroll
        for (int h = 0; h < CW / 4; ++h) {
          const int ch = oc * (CW / 4) + h;           // 4-float chunk of the image row
          f32x4 a4;
          if constexpr (PRE_IMG) a4 = img[ps][h];
          else a4 = *reinterpret_cast<const f32x4*>(cimg + r * BN + ((ch ^ (r & 7)) << 2));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = a4[e];
            if constexpr (FUSE == FUSE_A) x = ms.y * (x - ms.x * c1v[4 * h + e]);     // no a_stats: ms = (0, 1), c1v = 0 -> x, exactly
            v[4 * h + e] = x + bv[4 * h + e];
          }
        }
        if (Cpre) {                          // host-checked: whole vectors only (N, ldcp multiples of 8, 16-byte aligned)
          if (n + CW - 1 < N) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
              for (int h = 0; h < CW / 4; ++h) {
                uint2 u;
                u.x = (uint32_t)f2bf(v[4 * h]) | ((uint32_t)f2bf(v[4 * h + 1]) << 16);
                u.y = (uint32_t)f2bf(v[4 * h + 2]) | ((uint32_t)f2bf(v[4 * h + 3]) << 16);
                *reinterpret_cast<uint2*>(Cpre + (size_t)m * P.ldcp + n + 4 * h) = u;
              }
            } else {
#pragma unroll
              for (int h = 0; h < CW / 4; ++h) st4(Cpre + (size_t)m * P.ldcp + n + 4 * h, make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]));
            }
          }
        }
        if constexpr (ACT == DSRG_ACT_GELU && sizeof(OutT) == 2) {
#pragma unroll
          for (int e = 0; e < CW; e += 2) {
            const f32x2_t g2 = gelu_erf_short2(f32x2_t{v[e], v[e + 1]});
            v[e] = g2.x; v[e + 1] = g2.y;
          }
        } else {
#pragma unroll
          for (int e = 0; e < CW; ++e) v[e] = apply_act_c<ACT, sizeof(OutT) == 2>(v[e], act);
        }
        if (vec) {
          if (Rp) {
            RawRes raw;
            if constexpr (PRE_RES) raw = rres[ps];
            else raw = *reinterpret_cast<const RawRes*>(Rp + (size_t)m * P.ldr + n);
            float rv[CW];
            if constexpr (CW == 8) {
              const float4 r0 = raw4(make_uint2(raw.x, raw.y)), r1 = raw4(make_uint2(raw.z, raw.w));
              rv[0] = r0.x; rv[1] = r0.y; rv[2] = r0.z; rv[3] = r0.w; rv[4] = r1.x; rv[5] = r1.y; rv[6] = r1.z; rv[7] = r1.w;
            } else {
              const float4 rr = raw4(raw);
              rv[0] = rr.x; rv[1] = rr.y; rv[2] = rr.z; rv[3] = rr.w;
            }
            if constexpr (FUSE == FUSE_RS) {       // no r_stats: ms = (0, 1), gamma = 1, beta = 0 -> the residual itself, exactly (no branch)
#pragma unroll
              for (int e = 0; e < CW; ++e) rv[e] = (rv[e] - ms.x) * ms.y * rgv[e] + rbv[e];
            }
            if (RACT > 0 || (RACT < 0 && res_act > 0)) {
#pragma unroll
              for (int e = 0; e < CW; ++e) v[e] *= act_grad_c<RACT>(rv[e], res_act);
            } else {
#pragma unroll
              for (int e = 0; e < CW; ++e) v[e] += rv[e];
            }
          }
          if constexpr (FUSE == FUSE_RS) {
            // partial (sum, sum of squares) of this row over each 64-column block, from the fp32 values; always computed
            // (straight-line code), stored by the block's first lane when the problem asks for it
            static_assert(FUSE != FUSE_RS || BN % 64 == 0, "statistics are kept per 64-column block");
            constexpr int L64 = 64 / CW;          // consecutive lanes that hold one row's 64 columns (8 or 16): DPP, not LDS shuffles
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < CW; ++e) { s1 += v[e]; s2 += v[e] * v[e]; }
            s1 = dpp_group_sum<L64>(s1); s2 = dpp_group_sum<L64>(s2);
            if (sto && (oc & (L64 - 1)) == 0) sto[(size_t)(n >> 6) * M + m] = make_float2(s1, s2);
          }
          if constexpr (CW == 8) {
            uint4 u;
            u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
            u.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
            u.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
            *reinterpret_cast<uint4*>(C + (size_t)m * P.ldc + n) = u;
          } else {
            if (args.nt_store) st4_nt(C + (size_t)m * P.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
            else st4(C + (size_t)m * P.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
          }
        } else {
          for (int e = 0; e < CW && n + e < N; ++e) {
            float x = v[e];
            if (Rp) {
              const float rr = Elem<T>::ld(Rp + (size_t)m * P.ldr + n + e);
              x = res_act > 0 ? x * act_grad(rr, res_act) : x + rr;
            }
            Elem<OutT>::st(C + (size_t)m * P.ldc + n + e, x);
          }
        }
      }
    }
  }
  };
  using RT = std::integral_constant<int, -1>;
  using A0 = std::integral_constant<int, DSRG_ACT_NONE>;
  using AG = std::integral_constant<int, DSRG_ACT_GELU>;
  // (the 8-wave tile with fp32 output keeps the run-time form only: with the copies its 128-register budget makes hipcc
  // hand the K loop's scalar DMA bases to vector registers, which the asm statements cannot take)
  constexpr bool SPEC = !(NWAVES == 8 && sizeof(OutT) == 4);
  if (SPEC && res_act <= 0 && act == DSRG_ACT_NONE) rows_out(A0{}, A0{});
  else if (SPEC && res_act <= 0 && act == DSRG_ACT_GELU) rows_out(AG{}, A0{});
  else {
    bool done = false;
    if constexpr (TRANS != 0) {               // the dgrad of an FFN's second layer: (dY.W2) * gelu'(z)
      if (res_act == DSRG_ACT_GELU && act == DSRG_ACT_NONE) { rows_out(A0{}, AG{}); done = true; }
    }
    if (!done) rows_out(RT{}, RT{});
  }
  GEMM_STAMP(3);
#ifdef DSRG_GEMM_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_stamps[blockIdx.x * 8 + 4] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
#endif
}

template <typename T, typename OutT, int BM, int BN, int NST, int FUSE = 0, int WAVES_M = 2, int WAVES_N = 2, int TRANS = 0>
int launch(const GemmArgs& a, int total_tiles, hipStream_t s) {
  constexpr int ring = NST * (BM + BN) * ROWB, cimg = BM * BN * 4;
  constexpr int smem_main = ring > cimg ? ring : (ring > cimg / 2 ? ring : cimg / 2);   // an image larger than the ring leaves in two passes
  // fused epilogue: the partial-sum and (mean, rstd) tables sit behind ring and (whole) image
  constexpr int smem = FUSE ? (ring > cimg ? ring : cimg) + (4 + 1) * BM * 8 : smem_main;
  DSRG_SET_MAX_LDS_ONCE((gemm_nt_kernel<T, OutT, BM, BN, NST, FUSE, WAVES_M, WAVES_N, TRANS>), smem);
  hipLaunchKernelGGL((gemm_nt_kernel<T, OutT, BM, BN, NST, FUSE, WAVES_M, WAVES_N, TRANS>), dim3(total_tiles), dim3(64 * WAVES_M * WAVES_N), smem, s, a);
  DSRG_LAUNCH_CHECK("dsrg_gemm_nt");
  return DSRG_OK;
}

// DSRG_GEMM_TILE="BMxBN[xNST]" (tuning aid), parsed once (thread-safe function-local static), read-only afterwards
struct TileOverride { int bm, bn, nst; bool set; };
const TileOverride& tile_override() {
  static const TileOverride o = [] {
    TileOverride t{0, 0, 0, false};
    const char* force = getenv("DSRG_GEMM_TILE");     // "128x128" | "64x128" | "64x64" ... [x ring depth]
    if (force && sscanf(force, "%dx%dx%d", &t.bm, &t.bn, &t.nst) >= 2) t.set = true;
    return t;
  }();
  return o;
}

// The tile menu.  Every (bm, bn) has 2x2 waves of (bm/2) x (bn/2); `nst` lists the ring depths built for it (0-terminated,
// first = default).  The wide tiles exist for bf16 only (fp32 parity mode keeps the three square-ish ones).
struct TileShape { int bm, bn, nst[4]; bool f32_ok; };
constexpr TileShape kTiles[] = {
  {64, 64, {2, 3, 4, 0}, true},   {64, 128, {2, 3, 0, 0}, true},  {64, 192, {3, 0, 0, 0}, false}, {64, 256, {3, 0, 0, 0}, false},
  {128, 64, {3, 2, 4, 0}, false}, {128, 96, {3, 0, 0, 0}, false}, {128, 128, {2, 3, 0, 0}, true}, {128, 192, {3, 2, 0, 0}, false},
  {128, 256, {3, 0, 0, 0}, false}, {256, 128, {3, 2, 0, 0}, false},      // 256x128: 8 waves (4x2)
};
const TileShape* find_tile(int bm, int bn, bool bf16) {
  for (const TileShape& t : kTiles)
    if (t.bm == bm && t.bn == bn && (bf16 || t.f32_ok)) return &t;
  return nullptr;
}
bool tile_has_nst(const TileShape& t, int nst) {
  for (int i = 0; i < 4 && t.nst[i]; ++i)
    if (t.nst[i] == nst) return true;
  return false;
}

int count_tiles(GemmArgs& a, int m, int n) {
  int tot = 0;
  for (int i = 0; i < a.nprob; ++i) {
    a.tile_start[i] = tot;
    tot += ((a.p[i].M + m - 1) / m) * ((a.p[i].N + n - 1) / n);
  }
  for (int i = a.nprob; i <= DSRG_MAX_GROUPS; ++i) a.tile_start[i] = i == a.nprob ? tot : 0x7fffffff;   // (the kernel's problem search)
  return tot;
}

// Tile choice, from measurements on MI355X (profiles/r02b_gemm_sweep_cold.txt, r02d_plan_ab.txt).
//  * One launch at a time on operands beyond the XCD L2s (tools/bench_gemm.py --cold --sweep): a nav-step GEMM (174..1400
//    tiles of 64x64) is bound by per-workgroup latency and by the operand bytes a CU pulls from L2 into LDS (~36 B/clk per
//    CU), not by the MFMA rate.  One round of 200-256 large tiles (128x192, 128x256: one workgroup per CU, one wave per
//    SIMD, nothing to hide prologue / epilogue / GELU behind) is 1.5-2.5x SLOWER than 2-4 co-resident workgroups of small
//    tiles on every x-layer shape; per launch the best tile is 64x64 for N = 768, 64x128 / 128x64 for N = 2304 and 128x128
//    for the local FFN-in, each with the ring depth that the tile count leaves LDS for.
//  * Inside the step the ranking changes: the captured graph runs two chains of launches side by side (global / local
//    branch, object / view encoder), and what counts is how many workgroups of BOTH kernels a CU holds.  A/B of the whole
//    step on one box (bench.py, ms per step): round 1's plan 1.233 | per-launch-best tiles and depths 1.255 | 64x64 with a
//    3-deep ring everywhere 1.201 | 64x128x2 everywhere 1.356 | 128x64x3 1.253 | 128x128x2 1.535.  So: 64x64, 3-deep ring
//    (48 KiB: three workgroups per CU), 2-deep from 1024 tiles on (4+ rounds: occupancy hides the latency), and the
//    8-wave 256x128 tile (one workgroup per CU, two 64x64 waves per SIMD) only for the big once-per-episode GEMMs (text
//    K/V panel, >= 1000 tiles of 128x128: 158 -> 124 us; 4096^3: 190 -> 111 us = 1.24 PFLOP/s).  The same tile on the
//    panorama FFN-in (2 launches per step, 24.6 -> 22.6 us each on their own) makes the STEP 20 us slower: it takes the
//    whole CU's LDS while the object-token encoder's launches on the other stream wait.
int plan_tiles(GemmArgs& a, int& bm, int& bn, int& nst, bool bf16, int fbm, int fbn, int fnst) {
  nst = 0;
  if (fbm <= 0) {
    const TileOverride& ov = tile_override();
    if (ov.set) { fbm = ov.bm; fbn = ov.bn; fnst = ov.nst; }
  }
  if (fbm > 0) {
    const TileShape* t = find_tile(fbm, fbn, bf16);
    if (t) {
      bm = fbm; bn = fbn;
      nst = tile_has_nst(*t, fnst) ? fnst : t->nst[0];
      return count_tiles(a, bm, bn);
    }
  }
  int maxM = 0, minK = 1 << 30;
  for (int i = 0; i < a.nprob; ++i) {
    maxM = a.p[i].M > maxM ? a.p[i].M : maxM;
    minK = a.p[i].K < minK ? a.p[i].K : minK;
  }
  // wgrad launches (every problem dW = dY^T.X on operands as they lie, trans = 3; bf16) can run 128-row tiles (a transposed
  // operand staged as 64-column blocks): DSRG_GEMM_WGRAD = "BMxBNxNST:mintiles" (128x128x2 | 128x64x3).  OFF by default -
  // measured on the pre-training step (one box, ms per step): 64x64 11.71 | 128x64x3 from 150 tiles 11.67 | 128x128x2 from 100
  // tiles 12.00 | from 30 tiles 12.37: dW of the packed QKV / FFN weights is only 108 / 144 tiles of 128x128, and as with the
  // forward launches a half-empty chip of large tiles loses to 4 rounds of small ones.
  {
    bool all_t3 = a.nprob > 0 && bf16;
    for (int i = 0; i < a.nprob; ++i) all_t3 = all_t3 && a.p[i].trans == 3;
    if (all_t3) {
      static const char* wg_env = getenv("DSRG_GEMM_WGRAD");
      const char* e = wg_env ? wg_env : "0";
      int tbm = 0, tbn = 0, tnst = 0, tmin = 0;
      if (sscanf(e, "%dx%dx%d:%d", &tbm, &tbn, &tnst, &tmin) == 4 &&
          ((tbm == 128 && tbn == 128 && tnst == 2) || (tbm == 128 && tbn == 64 && tnst == 3)) && count_tiles(a, tbm, tbn) >= tmin) {
        bm = tbm; bn = tbn; nst = tnst;
        return count_tiles(a, bm, bn);
      }
    }
  }
  // LayerNorm-fusion kinds (FUSE_A: 64x64 and 128x128x2; FUSE_RS: 64x64 and 128x64x4) and transposed operands (64x64, wgrad above)
  bool fuse_a = false, fuse_rs = false, only64 = false, dgrad = false;
  // DSRG_GEMM_DGRAD=1: dgrad launches (trans = 2) take the forward rules' wide tiles.  OFF by default - measured on the
  // pre-training step (one stream, one box): 64x64 10.49 ms | wide 10.58 (bit-identical results either way).
  static const char* dg_env = getenv("DSRG_GEMM_DGRAD");
  for (int i = 0; i < a.nprob; ++i) {
    fuse_a = fuse_a || a.p[i].a_stats;
    fuse_rs = fuse_rs || a.p[i].r_stats || a.p[i].stats_out;
    // dgrad (dX = dY.W, W as it lies: trans = 2) takes the forward launches' wide / long-K rules below (128x128x2, 128x64x4)
    if (a.p[i].trans == 2 && bf16 && dg_env && dg_env[0] == '1') dgrad = true;
    else only64 = only64 || a.p[i].trans;
  }
  only64 = only64 || ((fuse_a || fuse_rs) && !bf16);        // the wide fused tiles are built for bf16
  const bool fuse = only64 || fuse_a || fuse_rs || dgrad;    // no 8-wave tile, no tuning overrides
  // Wide launches (every problem N >= 2304 and K <= 1024: packed QKV, FFN-in; 20 per nav-step): 128x128 with a 2-deep ring
  // (64 KiB: two workgroups per CU) from 200 tiles on - r02t A/B in DESIGN.md section 4.0.  DSRG_GEMM_WIDE /
  // DSRG_GEMM_LONGK (K >= 2048) = "BMxBNxNST:mintiles[,...]" override the candidates (tuning aid; first that reaches its
  // minimum tile count wins, "0" = none).
  {
    int minN = 1 << 30, maxK = 0;
    for (int i = 0; i < a.nprob; ++i) { minN = a.p[i].N < minN ? a.p[i].N : minN; maxK = a.p[i].K > maxK ? a.p[i].K : maxK; }
    static const char* wide_env = getenv("DSRG_GEMM_WIDE");
    static const char* longk_env = getenv("DSRG_GEMM_LONGK");
    const char* e = nullptr;
    const bool big = !fuse && maxM > 64 && minK >= 512 && count_tiles(a, 128, 128) >= 1000;      // the 8-wave 256x128 tile below
    if (big) e = nullptr;
    else if (!only64 && !fuse_rs && maxM > 64 && minN >= 2304 && maxK <= 1024) {
      // (DSRG_GEMM_WIDE_MIN: the tile count from which the default wide tile applies, LayerNorm-fused launches included)
      static const char* wmin_env = getenv("DSRG_GEMM_WIDE_MIN");
      static char wide_def[32];
      static const bool wide_def_init = (snprintf(wide_def, sizeof(wide_def), "128x128x2:%d", wmin_env ? atoi(wmin_env) : 200), true);
      (void)wide_def_init;
      e = (wide_env && !fuse) ? wide_env : (bf16 ? wide_def : nullptr);      // fp32 parity mode: 64x64x3 (3.95 vs 4.28 ms per step)
    }
    else if (!only64 && !fuse_a && maxM > 64 && minK >= 2048) e = (longk_env && !fuse) ? longk_env : (bf16 || !fuse ? "128x64x4:1" : nullptr);    // FFN-out: 1.136 -> 1.128 ms per c2 step
    // Narrow-N launches without a LayerNorm epilogue (the pre-training step's plain projections over thousands of token rows):
    // 128x128x2 once that still gives 300 tiles - c4 10.10 -> 10.00 ms same box, nav-steps unchanged (their plain launches are
    // smaller); DSRG_GEMM_NARROW overrides ("0" = 64x64 only).
    else if (!fuse && maxM > 64 && maxK <= 1024) { static const char* narrow_env = getenv("DSRG_GEMM_NARROW"); e = narrow_env ? narrow_env : (bf16 ? "128x128x2:300" : nullptr); }
    while (e && *e) {
      int tbm = 0, tbn = 0, tnst = 0, tmin = 0;
      if (sscanf(e, "%dx%dx%d:%d", &tbm, &tbn, &tnst, &tmin) == 4) {
        const TileShape* t = find_tile(tbm, tbn, bf16);
        if (t && tile_has_nst(*t, tnst) && count_tiles(a, tbm, tbn) >= tmin) { bm = tbm; bn = tbn; nst = tnst; return count_tiles(a, bm, bn); }
      }
      const char* c = strchr(e, ',');
      e = c ? c + 1 : nullptr;
    }
  }
  if (!fuse && maxM > 64 && minK >= 512 && count_tiles(a, 128, 128) >= 1000) {
    if (bf16) { bm = 256; bn = 128; nst = 3; } else { bm = 128; bn = 128; nst = 2; }
  } else {
    bm = 64; bn = 64;
    nst = count_tiles(a, 64, 64) > 1024 ? 2 : 3;
  }
  return count_tiles(a, bm, bn);
}

template <typename T, typename OutT>
int dispatch_tile(GemmArgs& a, hipStream_t s, int fbm, int fbn, int fnst) {
  int bm, bn, nst;
  int tiles = plan_tiles(a, bm, bn, nst, sizeof(T) == 2, fbm, fbn, fnst);
  {   // the fused epilogue's wide tiles exist for bf16 storage (128x128x2: bf16 output too): anything else runs on 64x64
    bool fa = false, frs = false;
    for (int i = 0; i < a.nprob; ++i) { fa = fa || a.p[i].a_stats; frs = frs || a.p[i].r_stats || a.p[i].stats_out; }
    const bool wide_ok = sizeof(T) == 2 && ((fa && sizeof(OutT) == 2 && bm == 128 && bn == 128 && nst == 2) || (frs && bm == 128 && bn == 64 && nst == 4));
    if ((fa || frs) && fbm <= 0 && !(bm == 64 && bn == 64 && (nst == 2 || nst == 3)) && !wide_ok) {
      bm = 64; bn = 64;
      tiles = count_tiles(a, 64, 64);
      nst = tiles > 1024 ? 2 : 3;
    }
  }
  static const char* gm_env = getenv("DSRG_GEMM_GM");      // tuning aid
  a.gm = gm_env ? atoi(gm_env) : (bm >= 128 ? 1 : 2);      // r02w A/B, c2 step: 4 everywhere 1.0847 ms | 2 everywhere 1.0816 | 1 everywhere 1.0803 | 8: 1.099
  if (a.gm < 1) a.gm = 1;
  static const char* np_env = getenv("DSRG_GEMM_NP");      // tuning aid: force the number of N-panels (1 = stripes)
  static const char* nt_env = getenv("DSRG_GEMM_NT");      // tuning aid: 0/1 non-temporal C stores
  a.nt_store = nt_env ? atoi(nt_env) : 0;   // 3-9 % faster launch by launch (tools/bench_gemm.py), 0.2 % slower inside the step: the consumer finds C in L2 less often
  const int per_xcd = (tiles + 7) / 8;
  for (int i = 0; i < a.nprob; ++i) {
    const int tm_ = (a.p[i].M + bm - 1) / bm, tn_ = (a.p[i].N + bn - 1) / bn;
    int best_pw = tn_;
    long best = -1;
    for (int np = 1; np <= tn_ && np <= 8; ++np) {
      if (np_env && np != atoi(np_env) && atoi(np_env) >= 1 && atoi(np_env) <= tn_) continue;
      const int pw = (tn_ + np - 1) / np;
      int mt = (per_xcd + pw - 1) / pw;
      if (mt > tm_) mt = tm_;
      const long rows = (long)mt * bm + (long)pw * bn;   // A rows + W rows an XCD's patch touches
      if (best < 0 || rows < best) { best = rows; best_pw = pw; }
    }
    a.pw[i] = best_pw;
  }
  bool fuse_a = false, fuse_rs = false;
  for (int i = 0; i < a.nprob; ++i) { fuse_a = fuse_a || a.p[i].a_stats; fuse_rs = fuse_rs || a.p[i].r_stats || a.p[i].stats_out; }
  if (a.p[0].trans) {      // backward GEMMs on operands as they lie (no transposed copies); bf16 checked by the entry
    const bool wg_tile = a.p[0].trans == 3 && ((bm == 128 && bn == 128 && nst == 2) || (bm == 128 && bn == 64 && nst == 3));
    const bool dg_tile = a.p[0].trans == 2 && ((bm == 128 && bn == 128 && nst == 2) || (bm == 128 && bn == 64 && nst == 4));
    DSRG_CHECK((bm == 64 && bn == 64 && (nst == 2 || nst == 3)) || wg_tile || dg_tile,
               "dsrg_gemm_nt: transposed operands exist for the 64x64 tile (ring 2 or 3), trans = 3 also 128x128x2 / 128x64x3, trans = 2 also 128x128x2 / 128x64x4, not %dx%dx%d", bm, bn, nst);
    if constexpr (sizeof(T) == 2) {
      const int tr = a.p[0].trans;
      if (tr == 2 && bm == 128 && bn == 128) return launch<T, OutT, 128, 128, 2, false, 2, 2, 2>(a, tiles, s);
      if (tr == 2 && bm == 128 && bn == 64) return launch<T, OutT, 128, 64, 4, false, 2, 2, 2>(a, tiles, s);
      if (tr == 3 && bm == 128 && bn == 128) return launch<T, OutT, 128, 128, 2, false, 2, 2, 3>(a, tiles, s);
      if (tr == 3 && bm == 128 && bn == 64) return launch<T, OutT, 128, 64, 3, false, 2, 2, 3>(a, tiles, s);
#define DSRG_TRANS_CASE(TR_, NST_) if (tr == TR_ && nst == NST_) return launch<T, OutT, 64, 64, NST_, false, 2, 2, TR_>(a, tiles, s)
      DSRG_TRANS_CASE(1, 2); DSRG_TRANS_CASE(1, 3); DSRG_TRANS_CASE(2, 2); DSRG_TRANS_CASE(2, 3); DSRG_TRANS_CASE(3, 2); DSRG_TRANS_CASE(3, 3);
#undef DSRG_TRANS_CASE
    }
    DSRG_CHECK(false, "dsrg_gemm_nt: no kernel for trans=%d", a.p[0].trans);
  }
  if (fuse_a || fuse_rs) {
    DSRG_CHECK(!(fuse_a && fuse_rs), "dsrg_gemm_nt: a_stats and r_stats / stats_out problems cannot share a launch");
    if (fuse_a) {
      if (bm == 64 && bn == 64 && nst == 2) return launch<T, OutT, 64, 64, 2, FUSE_A>(a, tiles, s);
      if (bm == 64 && bn == 64 && nst == 3) return launch<T, OutT, 64, 64, 3, FUSE_A>(a, tiles, s);
      if constexpr (sizeof(T) == 2 && sizeof(OutT) == 2) {
        if (bm == 128 && bn == 128 && nst == 2) return launch<T, OutT, 128, 128, 2, FUSE_A>(a, tiles, s);
      }
    } else {
      if (bm == 64 && bn == 64 && nst == 2) return launch<T, OutT, 64, 64, 2, FUSE_RS>(a, tiles, s);
      if (bm == 64 && bn == 64 && nst == 3) return launch<T, OutT, 64, 64, 3, FUSE_RS>(a, tiles, s);
      if constexpr (sizeof(T) == 2) {
        if (bm == 128 && bn == 64 && nst == 4) return launch<T, OutT, 128, 64, 4, FUSE_RS>(a, tiles, s);
      }
    }
    DSRG_CHECK(false, "dsrg_gemm_nt: the LayerNorm-fusion epilogue exists for 64x64 (ring 2 or 3), 128x128x2 (a_stats, bf16 in and out) and "
               "128x64x4 (r_stats / stats_out, bf16) only, not %dx%dx%d", bm, bn, nst);
  }
#define DSRG_TILE_CASE(BM_, BN_, NST_) if (bm == BM_ && bn == BN_ && nst == NST_) return launch<T, OutT, BM_, BN_, NST_>(a, tiles, s)
  DSRG_TILE_CASE(64, 64, 2); DSRG_TILE_CASE(64, 64, 3); DSRG_TILE_CASE(64, 64, 4);
  DSRG_TILE_CASE(64, 128, 2); DSRG_TILE_CASE(64, 128, 3);
  DSRG_TILE_CASE(128, 128, 2); DSRG_TILE_CASE(128, 128, 3);
  if constexpr (sizeof(T) == 2) {
    DSRG_TILE_CASE(64, 192, 3); DSRG_TILE_CASE(64, 256, 3);
    DSRG_TILE_CASE(128, 64, 2); DSRG_TILE_CASE(128, 64, 3); DSRG_TILE_CASE(128, 64, 4);
    DSRG_TILE_CASE(128, 96, 3);
    DSRG_TILE_CASE(128, 192, 2); DSRG_TILE_CASE(128, 192, 3);
    DSRG_TILE_CASE(128, 256, 3);
    if (bm == 256 && bn == 128 && nst == 3) return launch<T, OutT, 256, 128, 3, false, 4, 2>(a, tiles, s);
    if (bm == 256 && bn == 128 && nst == 2) return launch<T, OutT, 256, 128, 2, false, 4, 2>(a, tiles, s);
  }
#undef DSRG_TILE_CASE
  DSRG_CHECK(false, "dsrg_gemm_nt: no kernel for tile %dx%d ring %d", bm, bn, nst);
  return DSRG_EINVAL;
}

int gemm_entry(const dsrg_gemm_problem_t* problems, int nproblems, int dtype, int out_dtype, int act, int fbm, int fbn, int fnst, void* stream);

}  // namespace

namespace {
int gemm_entry(const dsrg_gemm_problem_t* problems, int nproblems, int dtype, int out_dtype, int act, int fbm, int fbn, int fnst, void* stream) {
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
    DSRG_CHECK(p.K % bk == 0 || p.trans == 3, "dsrg_gemm_nt: problem %d K=%d must be a multiple of %d (any K only with trans = 3)", i, p.K, bk);
    DSRG_CHECK(!p.a_colsum || (p.trans & 1), "dsrg_gemm_nt: problem %d: a_colsum needs a transposed A (trans bit 0)", i);
    DSRG_CHECK(!p.c_pre || (p.N % 8 == 0 && p.ldcp % 8 == 0 && p.ldcp >= p.N && ((uintptr_t)p.c_pre & 15) == 0 && !p.a_stats && !p.r_stats && !p.stats_out),
               "dsrg_gemm_nt: problem %d: c_pre needs N and ldcp multiples of 8, a 16-byte aligned pointer and no LayerNorm fusion", i);
    DSRG_CHECK(p.res_act >= DSRG_ACT_NONE && p.res_act <= DSRG_ACT_TANH && (p.res_act == DSRG_ACT_NONE || (p.residual && !p.r_stats)),
               "dsrg_gemm_nt: problem %d: res_act=%d needs a residual (the pre-activation) and no r_stats", i, p.res_act);
    DSRG_CHECK(p.trans >= 0 && p.trans <= 3 && p.trans == problems[0].trans, "dsrg_gemm_nt: problem %d: trans=%d must be 0..3 and the same for the whole group", i, p.trans);
    DSRG_CHECK(p.trans == 0 || (dtype == DSRG_BF16 && !p.a_stats && !p.r_stats && !p.stats_out &&
                                (fbm <= 0 || (p.trans >= 2 && fbm == 128 && (fbn == 128 || fbn == 64)))),
               "dsrg_gemm_nt: problem %d: transposed operands are bf16 only, without LayerNorm fusion; a pinned tile only with trans = 2 / 3 (128x128x2, 128x64x3|4)", i);
    const bool ta = (p.trans & 1) != 0, tw = (p.trans & 2) != 0;
    // a transposed operand is [K, M] / [K, N]: whole 16-byte chunks of 8 output rows / columns
    DSRG_CHECK(p.lda % al == 0 && p.lda >= (ta ? p.M : p.K) && (!ta || (p.M % 8 == 0 && p.M >= 8)),
               "dsrg_gemm_nt: problem %d lda=%d must be a multiple of %d and >= %s (and M %% 8 == 0 when A is stored transposed)", i, p.lda, al, ta ? "M" : "K");
    DSRG_CHECK(p.ldw % al == 0 && p.ldw >= (tw ? p.N : p.K) && (!tw || (p.N % 8 == 0 && p.N >= 8)),
               "dsrg_gemm_nt: problem %d ldw=%d must be a multiple of %d and >= %s (and N %% 8 == 0 when W is stored transposed)", i, p.ldw, al, tw ? "N" : "K");
    DSRG_CHECK(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0, "dsrg_gemm_nt: problem %d A/W not 16-byte aligned", i);
    {   // the kernel addresses A and W rows with 32-bit byte offsets from the operand base (a transposed operand: from the K-step's base)
      const size_t esz = dtype == DSRG_BF16 ? 2 : 4;
      DSRG_CHECK((size_t)(ta ? 64 : p.M) * p.lda * esz < ((size_t)1 << 32) && (size_t)(tw ? 64 : p.N) * p.ldw * esz < ((size_t)1 << 32),
                 "dsrg_gemm_nt: problem %d: an operand spans 4 GiB or more (M*lda or N*ldw too large)", i);
    }
    DSRG_CHECK(p.ldc >= p.N && (!p.residual || p.ldr >= p.N), "dsrg_gemm_nt: problem %d ldc/ldr < N", i);
    DSRG_CHECK(((uintptr_t)p.C & 15) == 0 && ((uintptr_t)p.residual & 7) == 0 && ((uintptr_t)p.bias & 15) == 0,
               "dsrg_gemm_nt: problem %d C/residual/bias misaligned", i);
    if (p.a_stats || p.r_stats || p.stats_out) {   // LayerNorm fusion: whole 16-byte column groups only, partials 8-byte aligned
      DSRG_CHECK(p.N % 8 == 0 && p.ldc % 8 == 0 && (!p.residual || (p.ldr % 8 == 0 && ((uintptr_t)p.residual & 15) == 0)),
                 "dsrg_gemm_nt: problem %d: LayerNorm fusion needs N, ldc, ldr multiples of 8 and a 16-byte aligned residual", i);
      DSRG_CHECK(!p.a_stats || (p.a_c1 && p.a_np >= 1 && p.a_np <= 16 && ((uintptr_t)p.a_stats & 7) == 0 && ((uintptr_t)p.a_c1 & 15) == 0),
                 "dsrg_gemm_nt: problem %d: a_stats needs a_c1 and 1 <= a_np <= 16", i);
      DSRG_CHECK(!p.r_stats || (p.residual && p.r_gamma && p.r_beta && p.r_np >= 1 && p.r_np <= 16 && ((uintptr_t)p.r_stats & 7) == 0 &&
                                ((uintptr_t)p.r_gamma & 15) == 0 && ((uintptr_t)p.r_beta & 15) == 0),
                 "dsrg_gemm_nt: problem %d: r_stats needs a residual, r_gamma, r_beta and 1 <= r_np <= 16", i);
      DSRG_CHECK(!p.stats_out || (p.N % 64 == 0 && ((uintptr_t)p.stats_out & 7) == 0), "dsrg_gemm_nt: problem %d: stats_out needs N %% 64 == 0", i);
      DSRG_CHECK(p.ln_eps > 0.f || (!p.a_stats && !p.r_stats), "dsrg_gemm_nt: problem %d: ln_eps must be positive", i);
      DSRG_CHECK(!(p.a_stats && (p.r_stats || p.stats_out)), "dsrg_gemm_nt: problem %d: a_stats cannot be combined with r_stats / stats_out", i);
      DSRG_CHECK(p.M > 0 && (size_t)p.M * 16 * 8 < ((size_t)1 << 31), "dsrg_gemm_nt: problem %d: too many rows for the statistics tables", i);
    }
    a.p[live++] = p;
  }
  if (live == 0) return DSRG_OK;
  a.nprob = live;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DSRG_BF16) {
    if (out_dtype == DSRG_F32) return dispatch_tile<bf16_t, float>(a, s, fbm, fbn, fnst);
    return dispatch_tile<bf16_t, bf16_t>(a, s, fbm, fbn, fnst);
  }
  return dispatch_tile<float, float>(a, s, fbm, fbn, fnst);
}
}  // namespace

extern "C" int dsrg_gemm_nt(const dsrg_gemm_problem_t* problems, int nproblems, int dtype, int out_dtype,
                            int act, void* stream) {
  return gemm_entry(problems, nproblems, dtype, out_dtype, act, 0, 0, 0, stream);
}

extern "C" int dsrg_gemm_nt_tiled(const dsrg_gemm_problem_t* problems, int nproblems, int dtype, int out_dtype,
                                  int act, int bm, int bn, int nst, void* stream) {
  if (bm > 0) {
    DSRG_CHECK(find_tile(bm, bn, dtype == DSRG_BF16) != nullptr, "dsrg_gemm_nt_tiled: no %dx%d tile for dtype %d", bm, bn, dtype);
    DSRG_CHECK(nst == 0 || tile_has_nst(*find_tile(bm, bn, dtype == DSRG_BF16), nst), "dsrg_gemm_nt_tiled: tile %dx%d has no ring depth %d", bm, bn, nst);
  }
  return gemm_entry(problems, nproblems, dtype, out_dtype, act, bm, bn, nst, stream);
}

extern "C" int dsrg_gemm_nt_plan(const dsrg_gemm_problem_t* problems, int nproblems, int dtype, int* bm, int* bn, int* ntiles, int* nst_out) {
  DSRG_CHECK(problems && nproblems >= 1 && nproblems <= DSRG_MAX_GROUPS && bm && bn && ntiles, "dsrg_gemm_nt_plan: bad arguments");
  GemmArgs a;
  a.nprob = 0;
  for (int i = 0; i < nproblems; ++i)
    if (problems[i].M > 0 && problems[i].N > 0) a.p[a.nprob++] = problems[i];
  *bm = *bn = *ntiles = 0;
  int nst = 0;
  if (a.nprob) *ntiles = plan_tiles(a, *bm, *bn, nst, dtype == DSRG_BF16, 0, 0, 0);
  if (nst_out) *nst_out = nst;
  return DSRG_OK;
}
