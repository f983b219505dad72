// Fused short-sequence attention (head dim 64), fp32 arithmetic on the vector ALU with K/V tiles
// staged once per workgroup in LDS and an online softmax over 64-key tiles, so any Nk works and no
// score matrix ever reaches HBM.  This is the exact-fp32 path (parity mode) and the generic fallback;
// the mask term, the sprel bias w*d+b and the "+1 bool mask" quirk are formed in-kernel.
//   grid = (ceil(Nq/16), NH, B), 256 threads = 4 waves, each wave owns 4 query rows.
//   scores : lane <-> key   (K tile rows padded to 68 floats: conflict-free ds_read_b128)
//   P·V    : lane <-> d     (V tile row-major: conflict-free ds_read_b32, P broadcast from LDS)
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int DH = 64, KT = 64, QW = 4, QPB = 16, KPAD = 68;

struct AttnArgs {
  const void *Q, *K, *V; void* O;
  int64_t q_bs, k_bs, v_bs, o_bs;
  int q_rs, k_rs, v_rs, o_rs;
  const uint8_t* kmask; int mask_mode;
  const float* dists; const float* sprel_w; const float* sprel_b;
  int Nq, Nk; float scale;
  float* lse;   // [B,NH,Nq] log-sum-exp of the scores (NULL unless a backward pass will follow)
  int NH;
  float p_drop;               // attention-probability dropout (training); 0 = off
  unsigned long long seed;
  // key validity from lengths instead of a bool mask (forward kernels): key k is valid iff k % kseg < klens[b, k / kseg]
  // (dsrg_attn_problem_t::klens: no mask-building kernel in front of the attention launch)
  const int64_t* klens; int kseg;
  const int64_t* klens2;      // optional second lengths tensor, added to the first (REVERIE: view_lens + obj_lens)
  const unsigned long long* seed_off;     // optional device word added to the seed (dsrg_set_dropout_offset)
};
// valid[b, key] from the bool mask or from the lengths
__device__ __forceinline__ bool key_valid(const AttnArgs& a, int b, int key) {
  if (a.klens) {
    const int sg = key / a.kseg, nseg = (a.Nk + a.kseg - 1) / a.kseg;
    return key - sg * a.kseg < (int)a.klens[(size_t)b * nseg + sg] + (a.klens2 ? (int)a.klens2[(size_t)b * nseg + sg] : 0);
  }
  return a.kmask[(size_t)b * a.Nk + key] != 0;
}

// Dropout on the attention probabilities (vilmodel.py:115,150; nn.MultiheadAttention(dropout=)): drop_keep() of common.h on
// the flat index of element (b, h, q, k); the backward recomputes the same mask and no mask tensor exists.

// up to DSRG_ATTN_MAX_GROUPS independent problems (same B, NH, mask mode, dtype) in one launch: blockIdx.x walks the
// query blocks of problem 0, then problem 1, ...
struct AttnGroup {
  AttnArgs a[DSRG_ATTN_MAX_GROUPS];
  int start[DSRG_ATTN_MAX_GROUPS + 1];
  int n;
};
// The problem record is read from the kernarg segment in ONE batch of scalar loads at kernel entry (problem 0 sits at fixed
// offsets; the empty asm statement pins the values into SGPRs there) instead of a chain of dependent load -> wait round
// trips sunk to the first uses (9 in the forward kernel's ISA); only workgroups of problems 1.. pay a second batch.
// start[] past the last problem is INT_MAX (host), so the search needs no problem count.  Same scheme as gemm.hip.
#define ATTN_PIN_PROBLEM(a_)                                                                                                   \
  asm volatile("" ::"s"(a_.Q), "s"(a_.K), "s"(a_.V), "s"(a_.O), "s"(a_.q_bs), "s"(a_.k_bs), "s"(a_.v_bs), "s"(a_.o_bs),         \
               "s"(a_.q_rs), "s"(a_.k_rs), "s"(a_.v_rs), "s"(a_.o_rs), "s"(a_.kmask), "s"(a_.mask_mode), "s"(a_.dists),          \
               "s"(a_.sprel_w), "s"(a_.sprel_b), "s"(a_.Nq), "s"(a_.Nk), "s"(a_.scale), "s"(a_.lse), "s"(a_.NH), "s"(a_.p_drop), \
               "s"(a_.seed), "s"(a_.klens), "s"(a_.kseg), "s"(a_.klens2))
#define ATTN_PICK_PROBLEM(grp)                                   \
  AttnArgs a = grp.a[0];                                         \
  ATTN_PIN_PROBLEM(a);                                           \
  int gsel = 0, gstart = 0;                                      \
  _Pragma("unroll") for (int i_ = 1; i_ < DSRG_ATTN_MAX_GROUPS; ++i_) \
    if ((int)blockIdx.x >= grp.start[i_]) { gsel = i_; gstart = grp.start[i_]; } \
  if (gsel != 0) {                                               \
    a = grp.a[gsel];                                             \
    ATTN_PIN_PROBLEM(a);                                         \
  }                                                              \
  const int bx = (int)blockIdx.x - gstart;

template <typename T, bool DROP>
__global__ __launch_bounds__(256) void attention_kernel(const AttnGroup grp) {
  ATTN_PICK_PROBLEM(grp)
  if constexpr (DROP) { if (a.seed_off) a.seed += *a.seed_off; }
  __shared__ __attribute__((aligned(16))) float Ks[KT * KPAD];
  __shared__ __attribute__((aligned(16))) float Vs[KT * DH];
  __shared__ __attribute__((aligned(16))) float Qs[QPB * DH];
  __shared__ __attribute__((aligned(16))) float Ps[4 * KT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, q0 = bx * QPB;
  const T* Qp = reinterpret_cast<const T*>(a.Q) + b * a.q_bs + h * DH;
  const T* Kp = reinterpret_cast<const T*>(a.K) + b * a.k_bs + h * DH;
  const T* Vp = reinterpret_cast<const T*>(a.V) + b * a.v_bs + h * DH;
  T* Op = reinterpret_cast<T*>(a.O) + b * a.o_bs + h * DH;

  // stage the block's 16 query rows (zero beyond Nq)
  {
    const int r = tid >> 4, c = (tid & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + r < a.Nq) v = ld4(Qp + (size_t)(q0 + r) * a.q_rs + c);
    *reinterpret_cast<float4*>(&Qs[r * DH + c]) = v;
  }
  const float sw = a.dists ? a.sprel_w[0] : 0.f, sb = a.dists ? a.sprel_b[0] : 0.f;

  float m[QW], l[QW], o[QW];
#pragma unroll
  for (int i = 0; i < QW; ++i) { m[i] = -INFINITY; l[i] = 0.f; o[i] = 0.f; }

  for (int k0 = 0; k0 < a.Nk; k0 += KT) {
    __syncthreads();   // previous tile fully consumed (also orders the Qs writes on the first pass)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = (tid >> 4) + 16 * p, c = (tid & 15) * 4;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (k0 + r < a.Nk) {
        kv = ld4(Kp + (size_t)(k0 + r) * a.k_rs + c);
        vv = ld4(Vp + (size_t)(k0 + r) * a.v_rs + c);
      }
      *reinterpret_cast<float4*>(&Ks[r * KPAD + c]) = kv;
      *reinterpret_cast<float4*>(&Vs[r * DH + c]) = vv;
    }
    __syncthreads();

    const int key = k0 + lane;
    float mterm = 0.f;
    bool dead = key >= a.Nk;
    if (!dead && (a.kmask || a.klens)) {
      const bool mk = key_valid(a, b, key);
      if (a.mask_mode == DSRG_MASK_NEG10000) mterm = mk ? 0.f : -10000.0f;
      else if (a.mask_mode == DSRG_MASK_NEGINF) dead = !mk;
      else if (a.mask_mode == DSRG_MASK_PLUS1) mterm = mk ? 1.0f : 0.f;
    }
#pragma unroll
    for (int i = 0; i < QW; ++i) {
      const int ql = wave * QW + i, qg = q0 + ql;
      if (qg >= a.Nq) continue;                     // wave-uniform
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 kk = *reinterpret_cast<const float4*>(&Ks[lane * KPAD + c]);
        const float4 qq = *reinterpret_cast<const float4*>(&Qs[ql * DH + c]);
        s = fmaf(qq.x, kk.x, s); s = fmaf(qq.y, kk.y, s); s = fmaf(qq.z, kk.z, s); s = fmaf(qq.w, kk.w, s);
      }
      s = s * a.scale + mterm;
      if (a.dists && !dead) s += a.dists[((size_t)b * a.Nq + qg) * a.Nk + key] * sw + sb;
      if (dead) s = -INFINITY;
      const float mn = fmaxf(m[i], wave_max(s));
      const float alpha = (m[i] == -INFINITY) ? 0.f : __expf(m[i] - mn);
      const float p = (s == -INFINITY) ? 0.f : __expf(s - mn);
      l[i] = l[i] * alpha + wave_sum(p);
      m[i] = mn;
      const float pk = (DROP && p != 0.f)
                           ? p * drop_keep(a.seed, (((size_t)b * a.NH + h) * a.Nq + qg) * a.Nk + key, a.p_drop, 1.0f / (1.0f - a.p_drop)) : p;
      Ps[wave * KT + lane] = pk;                    // wave-private row: no barrier needed, wave is in lockstep
      __builtin_amdgcn_wave_barrier();
      float acc = o[i] * alpha;
#pragma unroll 8
      for (int j = 0; j < KT; j += 4) {
        const float4 pp = *reinterpret_cast<const float4*>(&Ps[wave * KT + j]);
        acc = fmaf(pp.x, Vs[(j + 0) * DH + lane], acc);
        acc = fmaf(pp.y, Vs[(j + 1) * DH + lane], acc);
        acc = fmaf(pp.z, Vs[(j + 2) * DH + lane], acc);
        acc = fmaf(pp.w, Vs[(j + 3) * DH + lane], acc);
      }
      o[i] = acc;
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int i = 0; i < QW; ++i) {
    const int qg = q0 + wave * QW + i;
    if (qg >= a.Nq) continue;
    Elem<T>::st(Op + (size_t)qg * a.o_rs + lane, o[i] / l[i]);
    if (a.lse && lane == 0) a.lse[((size_t)b * a.NH + h) * a.Nq + qg] = m[i] + logf(l[i]);
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA attention.  One workgroup per (episode, head, 64-query group); wave w owns the 16-query
// tile w.  Keys are processed in chunks of 128 with an online softmax, K staged row-major (128-B rows,
// same XOR swizzle as the GEMM), V staged the same way and read back TRANSPOSED by ds_read_b64_tr_b16 (tfrag_rm).
//   S^T tile  = K · Q^T   (v_mfma_f32_16x16x32_bf16, A = K rows, B = Q rows): lane (c = lane&15, g = lane>>4)
//               holds S[query c][key 16t + 4g + r], r = 0..3  -> the softmax max/sum over keys is in-lane
//               plus two xor-shuffles (16, 32).
//   O^T      += V^T · P^T with P taken STRAIGHT from those accumulators as the B operand: k-slot 8g+e of the
//               MFMA is key 16*t0+4g+e (e<4) / 16*t1+4g+e-4 (e>=4) of a tile pair (t0,t1); the V^T fragment
//               is read with the same key permutation (two transposing reads 16 keys apart).  No P round trip.
//   lane ends with O[query c][d = 16*dt + 4g + 0..3]  -> 8-byte stores.
constexpr int KC = 128, NKT = KC / 16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ int kswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Fragment of the TRANSPOSE of a row-major image (st_rows: 128-byte rows, kswz swizzle) for the MFMA whose B operand comes from a
// pair (2j, 2j+1) of score tiles, through the gfx950 transposing read ds_read_b64_tr_b16: the 16 lanes of group g supply rows 32j + 4g .. +3 (lane: row c16 >> 2, columns 4 (c16 & 3) .. +3 of the
// d-tile) and each receives ITS column d = 16 dt + c16 of those four rows; the second read is the same rows 16 further on
// ((row + 16) has the same swizzle).  Round 3: replaces the transposed images (32 two-byte LDS writes per thread and image,
// and in the backward kernels a second set of global loads) in the tiled forward and both backward kernels.
__device__ __forceinline__ bf16x8 tfrag_rm(const char* S, int dt, int c16, int j, int g) {
  typedef short v4s __attribute__((ext_vector_type(4)));
  const int row = 32 * j + 4 * g + (c16 >> 2), col = dt * 16 + 4 * (c16 & 3);
  const char* p = S + kswz(row, col >> 3) + ((col & 7) << 1);
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 16 * 128));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
}

#ifdef DSRG_ATTN_STAMPS   // diagnostic build only (tools/attn_stamps.py)
__device__ unsigned long long g_attn_stamps[16384 * 4];
#define ATTN_STAMP(i) do { if (threadIdx.x == 0) { const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (wg < 16384) g_attn_stamps[wg * 4 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define ATTN_STAMP(i) do {} while (0)
#endif
#ifndef ATTN_MFMA_MIN_WAVES
#define ATTN_MFMA_MIN_WAVES 3
#endif
// SPREL: the spatial bias w * d + b of the workgroup's [64 queries x 128 keys] block is staged in LDS with the K / V tiles
// (coalesced 256-byte row pieces, one affine per element) and read back as one ds_read_b128 per score tile, instead of four
// scattered global loads per tile and lane (graph_sprels, vilmodel.py:757-759; launches without pair distances use the
// instantiation without the 33 KiB table).
constexpr int SPS = 132;      // floats per row of the bias table: rows 4 banks apart, conflict-free b128 reads
template <bool DROP, bool SPREL>
__global__ __launch_bounds__(256, (DROP || SPREL) ? 2 : ATTN_MFMA_MIN_WAVES) void attention_mfma_kernel(const AttnGroup grp) {
  ATTN_STAMP(0);
  ATTN_PICK_PROBLEM(grp)
  if constexpr (DROP) { if (a.seed_off) a.seed += *a.seed_off; }
  __shared__ __attribute__((aligned(16))) char Ks[KC * 128];
  __shared__ __attribute__((aligned(16))) char Vs[KC * 128];
  __shared__ float mt[KC];
  __shared__ __attribute__((aligned(16))) float sp[SPREL ? 64 * SPS : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = (bx * 4 + wave) * 16;
  const bool active = q0 < a.Nq;                               // wave-uniform
  const bf16_t* Qp = reinterpret_cast<const bf16_t*>(a.Q) + b * a.q_bs + h * DH;
  const bf16_t* Kp = reinterpret_cast<const bf16_t*>(a.K) + b * a.k_bs + h * DH;
  const bf16_t* Vp = reinterpret_cast<const bf16_t*>(a.V) + b * a.v_bs + h * DH;
  bf16_t* Op = reinterpret_cast<bf16_t*>(a.O) + b * a.o_bs + h * DH;

  const int qrow = q0 + c16;
  uint4 qf[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  if (active && qrow < a.Nq) {
    qf[0] = *reinterpret_cast<const uint4*>(Qp + (size_t)qrow * a.q_rs + g * 8);
    qf[1] = *reinterpret_cast<const uint4*>(Qp + (size_t)qrow * a.q_rs + 32 + g * 8);
  }
  const float sw = a.dists ? a.sprel_w[0] : 0.f, sb = a.dists ? a.sprel_b[0] : 0.f;
  const float* drow = (!SPREL && a.dists) ? a.dists + ((size_t)b * a.Nq + (qrow < a.Nq ? qrow : 0)) * a.Nk : nullptr;
  const bool use_sp = SPREL && a.dists != nullptr;              // workgroup-uniform (a grouped launch may mix problems)

  f32x4 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  // One chunk of <= 128 keys.  The first chunk is processed outside the loop: inside it the compiler waits for the Q
  // loads at the loop header, i.e. before the K/V loads are issued (one more exposed memory latency per workgroup).
  auto chunk = [&](const int k0) {
    const int nk = min(KC, a.Nk - k0);
    const int ntile = ((nk + 31) / 32) * 2;                    // even number of 16-key tiles
    // ---- stage K and V (both row-major, swizzled; V is read back transposed by tfrag_rm); rows >= nk are zero.  All nine global loads of a thread
    // are issued UNCONDITIONALLY (row index clamped) before the first LDS write: behind per-row conditions the compiler
    // waited for each load before issuing the next - nine memory latencies in a row, 9 k of a workgroup's 17 k cycles
    // (in-kernel stamps, tools/attn_stamps.py).
    {
      const int r = tid >> 3, c = tid & 7;
      uint4 kreg[4], vreg[4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
        kreg[p] = *reinterpret_cast<const uint4*>(Kp + (size_t)(k0 + min(r + 32 * p, nk - 1)) * a.k_rs + c * 8);
#pragma unroll
      for (int p = 0; p < 4; ++p)
        vreg[p] = *reinterpret_cast<const uint4*>(Vp + (size_t)(k0 + min(r + 32 * p, nk - 1)) * a.v_rs + c * 8);
      uint8_t mkb = 1;
      if (a.kmask || a.klens) mkb = key_valid(a, b, k0 + min(tid & (KC - 1), nk - 1)) ? 1 : 0;
      float dreg[SPREL ? 32 : 1];
      // rows of this wave that exist (a 20-node graph map: 16 for wave 0, 4 for wave 1, none for waves 2 and 3) and whether
      // the chunk has a second 64-key half at all: both wave-uniform, so the table costs 20 row loads instead of 128
      const int sp_rows = SPREL ? max(0, min(16, a.Nq - (bx * 64 + wave * 16))) : 0;
      const bool sp_hi = nk > 64;
      if constexpr (SPREL) {
        if (use_sp) {          // wave w stages rows 16w .. 16w+15 of the block: two 256-byte pieces per row
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            dreg[2 * i] = dreg[2 * i + 1] = 0.f;
            if (i < sp_rows) {
              const float* dr = a.dists + ((size_t)b * a.Nq + bx * 64 + wave * 16 + i) * a.Nk + k0;
              dreg[2 * i] = dr[min(lane, nk - 1)];
              if (sp_hi) dreg[2 * i + 1] = dr[min(lane + 64, nk - 1)];
            }
          }
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = r + 32 * p;
        if (row < ntile * 16) {
          *reinterpret_cast<uint4*>(Ks + kswz(row, c)) = row < nk ? kreg[p] : make_uint4(0, 0, 0, 0);
          *reinterpret_cast<uint4*>(Vs + kswz(row, c)) = row < nk ? vreg[p] : make_uint4(0, 0, 0, 0);
        }
      }
      if constexpr (SPREL) {
        if (use_sp) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (i < sp_rows) {
              sp[(wave * 16 + i) * SPS + lane] = dreg[2 * i] * sw + sb;
              if (sp_hi) sp[(wave * 16 + i) * SPS + 64 + lane] = dreg[2 * i + 1] * sw + sb;
            }
          }
        }
      }
      if (tid < KC) {
        float t = -INFINITY;
        if (tid < nk) {
          t = 0.f;
          if (a.kmask || a.klens) {
            const bool mk = mkb != 0;
            if (a.mask_mode == DSRG_MASK_NEG10000) t = mk ? 0.f : -10000.0f;
            else if (a.mask_mode == DSRG_MASK_NEGINF) t = mk ? 0.f : -INFINITY;
            else if (a.mask_mode == DSRG_MASK_PLUS1) t = mk ? 1.0f : 0.f;
          }
        }
        mt[tid] = t;
      }
    }
    __syncthreads();
    ATTN_STAMP(1);
    if (!active) return;

    // The score / softmax / P.V part is specialised on the number of 16-key tiles (2, 4, 6, 8): with a run-time tile
    // count every tile sat in its own basic block and its LDS reads could not be scheduled ahead of the previous tile's MFMAs.
    auto compute = [&](auto nt_tag) {
    constexpr int NTL = decltype(nt_tag)::value;
    // ---- S^T tiles
    f32x4 s[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const uint4 kf = *reinterpret_cast<const uint4*>(Ks + kswz(t * 16 + c16, g + 4 * kk));
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[kk]), s[t], 0, 0, 0);
        }
      }
    }
    // ---- scale + mask + sprel, running max
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      {
        const float4 mk = *reinterpret_cast<const float4*>(&mt[t * 16 + 4 * g]);
        float mm[4] = {mk.x, mk.y, mk.z, mk.w};
        if constexpr (SPREL) {
          if (use_sp) {        // padding keys carry -inf in mt: the (finite, clamped) table entry does not matter there
            const float4 d4 = *reinterpret_cast<const float4*>(&sp[(wave * 16 + c16) * SPS + t * 16 + 4 * g]);
            mm[0] += d4.x; mm[1] += d4.y; mm[2] += d4.z; mm[3] += d4.w;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = s[t][r] * a.scale + mm[r];
          if (drow) {
            const int key = k0 + t * 16 + 4 * g + r;
            if (key < a.Nk) v += drow[key] * sw + sb;
          }
          s[t][r] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_safe);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < NTL; ++t)
      {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(s[t][r] - m_safe);
          s[t][r] = p;
          psum += p;
        }
      }
    l_run = l_run * alpha + psum;
    if (DROP && qrow < a.Nq) {                        // the normaliser keeps the undropped sum; only P.V sees the mask
      const float ik = 1.0f / (1.0f - a.p_drop);
      const size_t rbase = (((size_t)b * a.NH + h) * a.Nq + qrow) * a.Nk + k0;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
        {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kl = t * 16 + 4 * g + r;
            if (kl < nk) s[t][r] *= drop_keep(a.seed, rbase + kl, a.p_drop, ik);
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] *= alpha;
    // ---- O^T += V^T · P^T
#pragma unroll
    for (int j = 0; j < NTL / 2; ++j) {
      {
        bf16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[r] = (__bf16)s[2 * j][r]; pf[4 + r] = (__bf16)s[2 * j + 1][r]; }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag_rm(Vs, dt, c16, j, g), pf, o[dt], 0, 0, 0);
      }
    }
    };
    switch (ntile) {
      case 2: compute(std::integral_constant<int, 2>{}); break;
      case 4: compute(std::integral_constant<int, 4>{}); break;
      case 6: compute(std::integral_constant<int, 6>{}); break;
      default: compute(std::integral_constant<int, 8>{}); break;
    }
  };
  chunk(0);
  for (int k0 = KC; k0 < a.Nk; k0 += KC) {
    __syncthreads();                                           // every wave is done with the previous chunk's K / V
    chunk(k0);
  }
  ATTN_STAMP(2);
  if (!active) return;
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  if (qrow < a.Nq) {
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      st4(Op + (size_t)qrow * a.o_rs + dt * 16 + 4 * g, make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv));
    if (a.lse && g == 0) a.lse[((size_t)b * a.NH + h) * a.Nq + qrow] = m_run + logf(l_run);
  }
  ATTN_STAMP(3);
}

// ------------------------------------------------------------------------------------------------
// Short-key variant of the bf16 MFMA attention (every problem of the launch has Nk <= 128): the nav-step's
// launches are 768 workgroups that each live for one dependent chain  global load -> LDS write -> barrier -> LDS read ->
// S -> softmax -> P.V -> store  (in-kernel stamps, profiles/r03s_attn_stamps_before.txt: 5.8 k cycles of staging + 3.9 k of
// compute per workgroup for ~30 MFMAs).  Here the chain is shorter:
//   * the K fragments of S^T = K.Q^T are the MFMA's A operand exactly as K lies in memory (lane (c, g) holds 8 consecutive
//     head-dim values of key row c), so they are loaded from global memory straight into registers - no LDS round trip and no
//     barrier in front of the scores;
//   * V is staged ROW-major with plain 16-byte LDS writes (160-byte rows: the 8 rows a 32-lane half of a transposing read
//     touches fall into 8 different 32-byte bank groups) and its transposed fragments V^T[d][8 keys] come out through the
//     gfx950 transposing read ds_read_b64_tr_b16 - instead of 32 two-byte LDS writes per thread for the transposed image;
//   * key validity is one ballot per 64 keys (lane <-> key: a mask byte or a length compare), kept in scalar registers: no
//     mask-term table in LDS, so the only barrier sits between the softmax and P.V, where V's loads have long landed;
//   * the V loads are issued first, then Q / K / pair distances: the LDS writes wait for the oldest loads only.
// NTL = number of 16-key tiles (2, 4, 6, 8) of the LONGEST problem of the launch, a template parameter so that a launch
// of 20-key problems carries 16 K-fragment registers, not 64.  Same arithmetic, in the same order, as attention_mfma_kernel.
constexpr int VDS = 160;
template <int NTL, bool SPREL, bool DROP = false>
__global__ __launch_bounds__(256, NTL <= 6 ? 3 : 2) void attention_mfma_direct_kernel(const AttnGroup grp) {
  ATTN_STAMP(0);
  ATTN_PICK_PROBLEM(grp)
  if constexpr (DROP) { if (a.seed_off) a.seed += *a.seed_off; }
  constexpr int NP = NTL / 2;                                   // 32-row passes of the V stage
  __shared__ __attribute__((aligned(16))) char Vs[NTL * 16 * VDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = (bx * 4 + wave) * 16;
  const bool active = q0 < a.Nq;                               // wave-uniform
  const int nk = a.Nk;
  const bf16_t* Qp = reinterpret_cast<const bf16_t*>(a.Q) + b * a.q_bs + h * DH;
  const bf16_t* Kp = reinterpret_cast<const bf16_t*>(a.K) + b * a.k_bs + h * DH;
  const bf16_t* Vp = reinterpret_cast<const bf16_t*>(a.V) + b * a.v_bs + h * DH;
  bf16_t* Op = reinterpret_cast<bf16_t*>(a.O) + b * a.o_bs + h * DH;
  const int qrow = q0 + c16;
  const int qr = min(qrow, a.Nq - 1);                           // rows past Nq repeat the last one (never stored)

  // ---- all global loads in the order they are consumed (completion is counted in issue order): the validity bytes /
  // lengths, V rows (whole workgroup), then this wave's Q / K fragments and pair distances
  bool kv0 = true, kv1 = true;
  if (a.kmask || a.klens) {
    kv0 = key_valid(a, b, min(lane, nk - 1));
    if (NTL > 4) kv1 = key_valid(a, b, min(lane + 64, nk - 1));
  }
  uint4 vreg[NP];
  {
    const int r = tid >> 3, c = tid & 7;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      vreg[p] = *reinterpret_cast<const uint4*>(Vp + (size_t)min(r + 32 * p, nk - 1) * a.v_rs + c * 8);
  }
  uint4 qf[2], kf[NTL][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) qf[kk] = *reinterpret_cast<const uint4*>(Qp + (size_t)qr * a.q_rs + kk * 32 + g * 8);
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      kf[t][kk] = *reinterpret_cast<const uint4*>(Kp + (size_t)min(t * 16 + c16, nk - 1) * a.k_rs + kk * 32 + g * 8);
  const bool use_sp = SPREL && a.dists != nullptr;              // workgroup-uniform (a grouped launch may mix problems)
  float dreg[SPREL ? NTL : 1][4];
  float sw = 0.f, sb = 0.f;
  if constexpr (SPREL) {
#pragma unroll
    for (int t = 0; t < NTL; ++t) dreg[t][0] = dreg[t][1] = dreg[t][2] = dreg[t][3] = 0.f;
    if (use_sp) {
      sw = a.sprel_w[0]; sb = a.sprel_b[0];
      const float* dr = a.dists + ((size_t)b * a.Nq + qr) * nk;
      const bool al = (nk & 3) == 0 && ((uintptr_t)a.dists & 15) == 0;      // rows of pair distances are 16-byte aligned
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const int k4 = t * 16 + 4 * g;
        if (al && k4 + 3 < nk) {
          const float4 d4 = *reinterpret_cast<const float4*>(dr + k4);
          dreg[t][0] = d4.x; dreg[t][1] = d4.y; dreg[t][2] = d4.z; dreg[t][3] = d4.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k4 + r < nk) dreg[t][r] = dr[k4 + r];
        }
      }
    }
  }
  // ---- key validity: lane <-> key (and key + 64), one ballot each
  unsigned long long vb[2] = {~0ull, ~0ull};
  if (a.kmask || a.klens) {
    vb[0] = __ballot(kv0);
    if (NTL > 4) vb[1] = __ballot(kv1);
  }
  const int mode = a.mask_mode;
  const float m_bad = mode == DSRG_MASK_NEG10000 ? -10000.0f : (mode == DSRG_MASK_NEGINF ? -INFINITY : 0.f);
  const float m_ok = mode == DSRG_MASK_PLUS1 ? 1.0f : 0.f;

  // ---- V rows into LDS (rows >= nk are zero: their probabilities are, too, and 0 x anything finite adds nothing)
  {
    const int r = tid >> 3, c = tid & 7;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int row = r + 32 * p;
      *reinterpret_cast<uint4*>(Vs + row * VDS + c * 16) = row < nk ? vreg[p] : make_uint4(0, 0, 0, 0);
    }
  }
  ATTN_STAMP(1);
  f32x4 s[NTL];
  float m_run = -INFINITY, l_run = 0.f;
  if (active) {
    // ---- S^T tiles
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[t][kk]), __builtin_bit_cast(bf16x8, qf[kk]), s[t], 0, 0, 0);
    }
    // ---- scale + mask + sprel, max
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const unsigned bits = (unsigned)(vb[t >> 2] >> ((t & 3) * 16 + 4 * g));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + 4 * g + r;
        float mm = ((bits >> r) & 1u) ? m_ok : m_bad;
        if (key >= nk) mm = -INFINITY;
        if constexpr (SPREL) {
          if (use_sp) mm += dreg[t][r] * sw + sb;
        }
        const float v = s[t][r] * a.scale + mm;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    m_run = mx;
    const float m_safe = (mx == -INFINITY) ? 0.f : mx;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[t][r] - m_safe);
        s[t][r] = p;
        psum += p;
      }
    l_run = psum;
    if (DROP && qrow < a.Nq) {                                  // the normaliser keeps the undropped sum; only P.V sees the mask
      const float ik = 1.0f / (1.0f - a.p_drop);
      const size_t rbase = (((size_t)b * a.NH + h) * a.Nq + qrow) * nk;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kl = t * 16 + 4 * g + r;
          if (kl < nk) s[t][r] *= drop_keep(a.seed, rbase + kl, a.p_drop, ik);
        }
    }
  }
  __syncthreads();                                              // V staged (every wave, active or not, wrote its rows)
  if (!active) return;
  // ---- O^T = V^T . P^T, P straight from the score accumulators (k-slot 8g+e <-> key 32j + 4g + e, e < 4; 32j + 16 + 4g + e - 4)
  f32x4 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef short v4s __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int j = 0; j < NTL / 2; ++j) {
    bf16x8 pf;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pf[r] = (__bf16)s[2 * j][r]; pf[4 + r] = (__bf16)s[2 * j + 1][r]; }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      // the 16 lanes of group g supply rows 32j + 4g .. +3 (lane: row c16 >> 2, columns 4 (c16 & 3) .. +3 of the d-tile) and
      // each receives ITS column d = 16 dt + c16 of those four rows; the second read is the same 16 keys further on
      const char* vp = Vs + (32 * j + 4 * g + (c16 >> 2)) * VDS + (dt * 16 + 4 * (c16 & 3)) * 2;
      const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(vp));
      const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(vp + 16 * VDS));
      const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      const uint4 vf = make_uint4(l2.x, l2.y, h2.x, h2.y);
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), pf, o[dt], 0, 0, 0);
    }
  }
  ATTN_STAMP(2);
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  if (qrow < a.Nq) {
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      st4(Op + (size_t)qrow * a.o_rs + dt * 16 + 4 * g, make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv));
    if (a.lse && g == 0) a.lse[((size_t)b * a.NH + h) * a.Nq + qrow] = m_run + logf(l_run);
  }
  ATTN_STAMP(3);
}

// ------------------------------------------------------------------------------------------------
// Attention backward (fp32 arithmetic, recompute from Q, K, V and the forward's log-sum-exp; no score matrix in HBM).
//   p_ij = exp(s_ij - lse_i),  dp_ij = dO_i . v_j,  delta_i = dO_i . O_i,  ds_ij = p_ij (dp_ij - delta_i)
//   dQ_i = scale * sum_j ds_ij k_j      dK_j = scale * sum_i ds_ij q_i      dV_j = sum_i p_ij dO_i
//   d(sprel_w) = sum ds_ij * dist_ij    d(sprel_b) = sum ds_ij
// Two kernels so that no gradient needs a cross-workgroup sum except the two sprel scalars (float atomics):
//   attention_bwd_dq : grid (ceil(Nq/16), NH, B), wave per query row, lane <-> key, 64-key tiles (also writes delta);
//   attention_bwd_dkv: grid (ceil(Nk/64), NH, B), lane <-> key of the tile, the 4 waves split the queries and each
//                      lane keeps dK_j, dV_j (2 x 64 floats) in registers; waves are combined through LDS.
struct AttnBwdArgs {
  AttnArgs f;                 // forward operands (O = forward output)
  const void* dO; int64_t do_bs; int do_rs;
  void *dQ, *dK, *dV;         // [B,N,NH*64] with the batch / row strides below (elements)
  int64_t dq_bs, dk_bs, dv_bs;
  int dq_rs, dk_rs, dv_rs;
  float* delta;               // [B,NH,Nq] workspace
  float* dsprel;              // [2] (w, b) accumulated with atomics, or NULL
};

__device__ __forceinline__ float mask_term(const AttnArgs& a, int b, int key, bool& dead) {
  float t = 0.f;
  dead = key >= a.Nk;
  if (!dead && a.kmask) {
    const bool mk = a.kmask[(size_t)b * a.Nk + key] != 0;
    if (a.mask_mode == DSRG_MASK_NEG10000) t = mk ? 0.f : -10000.0f;
    else if (a.mask_mode == DSRG_MASK_NEGINF) dead = !mk;
    else if (a.mask_mode == DSRG_MASK_PLUS1) t = mk ? 1.0f : 0.f;
  }
  return t;
}

template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(const AttnBwdArgs ba) {
  AttnArgs a = ba.f;
  if (a.p_drop > 0.f && a.seed_off) a.seed += *a.seed_off;
  __shared__ __attribute__((aligned(16))) float Ks[KT * KPAD];
  __shared__ __attribute__((aligned(16))) float Vs[KT * KPAD];
  __shared__ __attribute__((aligned(16))) float Qs[QPB * DH];
  __shared__ __attribute__((aligned(16))) float Gs[QPB * DH];   // dO rows
  __shared__ __attribute__((aligned(16))) float Ds[4 * KT];     // ds of the wave's current row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QPB;
  const T* Qp = reinterpret_cast<const T*>(a.Q) + b * a.q_bs + h * DH;
  const T* Kp = reinterpret_cast<const T*>(a.K) + b * a.k_bs + h * DH;
  const T* Vp = reinterpret_cast<const T*>(a.V) + b * a.v_bs + h * DH;
  const T* Op = reinterpret_cast<const T*>(a.O) + b * a.o_bs + h * DH;
  const T* Gp = reinterpret_cast<const T*>(ba.dO) + b * ba.do_bs + h * DH;
  T* dQp = reinterpret_cast<T*>(ba.dQ) + (size_t)b * ba.dq_bs + h * DH;
  // Global loads are issued unconditionally with clamped row indices and masked afterwards: behind `if (row < N)` the
  // compiler waits for each load before issuing the next, one exposed memory latency per row group.
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float sw = a.dists ? a.sprel_w[0] : 0.f, sb = a.dists ? a.sprel_b[0] : 0.f;
  float dq[QW], lse[QW], delta[QW], gsw = 0.f, gsb = 0.f;
  {
    const int r = tid >> 4, c = (tid & 15) * 4, rq = min(q0 + r, a.Nq - 1);
    const float4 qv = ld4(Qp + (size_t)rq * a.q_rs + c), gv = ld4(Gp + (size_t)rq * ba.do_rs + c);
    float gl[QW], ol[QW];
#pragma unroll
    for (int i = 0; i < QW; ++i) {
      const int qc = min(q0 + wave * QW + i, a.Nq - 1);
      gl[i] = Elem<T>::ld(Gp + (size_t)qc * ba.do_rs + lane);
      ol[i] = Elem<T>::ld(Op + (size_t)qc * a.o_rs + lane);
      lse[i] = a.lse[((size_t)b * a.NH + h) * a.Nq + qc];
    }
    *reinterpret_cast<float4*>(&Qs[r * DH + c]) = q0 + r < a.Nq ? qv : zero4;
    *reinterpret_cast<float4*>(&Gs[r * DH + c]) = q0 + r < a.Nq ? gv : zero4;
#pragma unroll
    for (int i = 0; i < QW; ++i) {
      const int qg = q0 + wave * QW + i;
      dq[i] = 0.f;
      delta[i] = wave_sum(gl[i] * ol[i]);
      if (qg >= a.Nq) { lse[i] = 0.f; delta[i] = 0.f; }
      else if (lane == 0) ba.delta[((size_t)b * a.NH + h) * a.Nq + qg] = delta[i];
    }
  }
  for (int k0 = 0; k0 < a.Nk; k0 += KT) {
    float4 kst[4], vst[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int rk = min(k0 + (tid >> 4) + 16 * p, a.Nk - 1), c = (tid & 15) * 4;
      kst[p] = ld4(Kp + (size_t)rk * a.k_rs + c);
      vst[p] = ld4(Vp + (size_t)rk * a.v_rs + c);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = (tid >> 4) + 16 * p, c = (tid & 15) * 4;
      *reinterpret_cast<float4*>(&Ks[r * KPAD + c]) = k0 + r < a.Nk ? kst[p] : zero4;
      *reinterpret_cast<float4*>(&Vs[r * KPAD + c]) = k0 + r < a.Nk ? vst[p] : zero4;
    }
    __syncthreads();
    const int key = k0 + lane;
    bool dead;
    const float mterm = mask_term(a, b, key, dead);
#pragma unroll
    for (int i = 0; i < QW; ++i) {
      const int ql = wave * QW + i, qg = q0 + ql;
      if (qg >= a.Nq) continue;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 kk = *reinterpret_cast<const float4*>(&Ks[lane * KPAD + c]);
        const float4 vv = *reinterpret_cast<const float4*>(&Vs[lane * KPAD + c]);
        const float4 qq = *reinterpret_cast<const float4*>(&Qs[ql * DH + c]);
        const float4 gg = *reinterpret_cast<const float4*>(&Gs[ql * DH + c]);
        s = fmaf(qq.x, kk.x, s); s = fmaf(qq.y, kk.y, s); s = fmaf(qq.z, kk.z, s); s = fmaf(qq.w, kk.w, s);
        dp = fmaf(gg.x, vv.x, dp); dp = fmaf(gg.y, vv.y, dp); dp = fmaf(gg.z, vv.z, dp); dp = fmaf(gg.w, vv.w, dp);
      }
      s = s * a.scale + mterm;
      float dist = 0.f;
      if (a.dists && !dead) { dist = a.dists[((size_t)b * a.Nq + qg) * a.Nk + key]; s += dist * sw + sb; }
      const float p = dead ? 0.f : __expf(s - lse[i]);
      if (a.p_drop > 0.f && !dead)
        dp *= drop_keep(a.seed, (((size_t)b * a.NH + h) * a.Nq + qg) * a.Nk + key, a.p_drop, 1.0f / (1.0f - a.p_drop));
      const float ds = p * (dp - delta[i]);
      gsw += ds * dist; gsb += ds;
      Ds[wave * KT + lane] = ds;
      __builtin_amdgcn_wave_barrier();
      float acc = dq[i];
#pragma unroll 8
      for (int j = 0; j < KT; j += 4) {
        const float4 dd = *reinterpret_cast<const float4*>(&Ds[wave * KT + j]);
        acc = fmaf(dd.x, Ks[(j + 0) * KPAD + lane], acc);
        acc = fmaf(dd.y, Ks[(j + 1) * KPAD + lane], acc);
        acc = fmaf(dd.z, Ks[(j + 2) * KPAD + lane], acc);
        acc = fmaf(dd.w, Ks[(j + 3) * KPAD + lane], acc);
      }
      dq[i] = acc;
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int i = 0; i < QW; ++i) {
    const int qg = q0 + wave * QW + i;
    if (qg < a.Nq) Elem<T>::st(dQp + (size_t)qg * ba.dq_rs + lane, dq[i] * a.scale);
  }
  if (ba.dsprel && a.dists) {
    gsw = wave_sum(gsw); gsb = wave_sum(gsb);
    if (lane == 0) { atomicAdd(ba.dsprel, gsw); atomicAdd(ba.dsprel + 1, gsb); }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(const AttnBwdArgs ba) {
  AttnArgs a = ba.f;
  if (a.p_drop > 0.f && a.seed_off) a.seed += *a.seed_off;
  __shared__ __attribute__((aligned(16))) float Ks[KT * KPAD];
  __shared__ __attribute__((aligned(16))) float Vs[KT * KPAD];
  __shared__ __attribute__((aligned(16))) float Qs[QPB * DH];
  __shared__ __attribute__((aligned(16))) float Gs[QPB * DH];
  __shared__ float Ls[QPB], Dl[QPB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * KT;
  const T* Qp = reinterpret_cast<const T*>(a.Q) + b * a.q_bs + h * DH;
  const T* Kp = reinterpret_cast<const T*>(a.K) + b * a.k_bs + h * DH;
  const T* Vp = reinterpret_cast<const T*>(a.V) + b * a.v_bs + h * DH;
  const T* Gp = reinterpret_cast<const T*>(ba.dO) + b * ba.do_bs + h * DH;
  // (loads unconditional with clamped rows, masked afterwards: see attention_bwd_dq_kernel)
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    float4 kst[4], vst[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int rk = min(k0 + (tid >> 4) + 16 * p, a.Nk - 1), c = (tid & 15) * 4;
      kst[p] = ld4(Kp + (size_t)rk * a.k_rs + c);
      vst[p] = ld4(Vp + (size_t)rk * a.v_rs + c);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = (tid >> 4) + 16 * p, c = (tid & 15) * 4;
      *reinterpret_cast<float4*>(&Ks[r * KPAD + c]) = k0 + r < a.Nk ? kst[p] : zero4;
      *reinterpret_cast<float4*>(&Vs[r * KPAD + c]) = k0 + r < a.Nk ? vst[p] : zero4;
    }
  }
  const int key = k0 + lane;
  bool dead;
  const float mterm = mask_term(a, b, key, dead);
  const float sw = a.dists ? a.sprel_w[0] : 0.f, sb = a.dists ? a.sprel_b[0] : 0.f;
  float dk[DH], dv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  for (int q0 = 0; q0 < a.Nq; q0 += QPB) {
    {
      const int r = tid >> 4, c = (tid & 15) * 4, rq = min(q0 + r, a.Nq - 1);
      const float4 qv = ld4(Qp + (size_t)rq * a.q_rs + c), gv = ld4(Gp + (size_t)rq * ba.do_rs + c);
      const size_t li = ((size_t)b * a.NH + h) * a.Nq + min(q0 + (tid & (QPB - 1)), a.Nq - 1);
      const float lv = a.lse[li], dl = ba.delta[li];
      __syncthreads();                               // the previous query block has been consumed
      *reinterpret_cast<float4*>(&Qs[r * DH + c]) = q0 + r < a.Nq ? qv : zero4;
      *reinterpret_cast<float4*>(&Gs[r * DH + c]) = q0 + r < a.Nq ? gv : zero4;
      if (tid < QPB) {
        const bool ok = q0 + tid < a.Nq;
        Ls[tid] = ok ? lv : 0.f;
        Dl[tid] = ok ? dl : 0.f;
      }
    }
    __syncthreads();
    for (int i = wave; i < QPB; i += 4) {          // the 4 waves split the query rows
      const int qg = q0 + i;
      if (qg >= a.Nq) break;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 kk = *reinterpret_cast<const float4*>(&Ks[lane * KPAD + c]);
        const float4 vv = *reinterpret_cast<const float4*>(&Vs[lane * KPAD + c]);
        const float4 qq = *reinterpret_cast<const float4*>(&Qs[i * DH + c]);
        const float4 gg = *reinterpret_cast<const float4*>(&Gs[i * DH + c]);
        s = fmaf(qq.x, kk.x, s); s = fmaf(qq.y, kk.y, s); s = fmaf(qq.z, kk.z, s); s = fmaf(qq.w, kk.w, s);
        dp = fmaf(gg.x, vv.x, dp); dp = fmaf(gg.y, vv.y, dp); dp = fmaf(gg.z, vv.z, dp); dp = fmaf(gg.w, vv.w, dp);
      }
      s = s * a.scale + mterm;
      if (a.dists && !dead) s += a.dists[((size_t)b * a.Nq + qg) * a.Nk + key] * sw + sb;
      const float p = dead ? 0.f : __expf(s - Ls[i]);
      float keep = 1.f;
      if (a.p_drop > 0.f && !dead)
        keep = drop_keep(a.seed, (((size_t)b * a.NH + h) * a.Nq + qg) * a.Nk + key, a.p_drop, 1.0f / (1.0f - a.p_drop));
      dp *= keep;
      const float pv = p * keep;                    // what multiplied V in the forward
      const float ds = p * (dp - Dl[i]) * a.scale;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 qq = *reinterpret_cast<const float4*>(&Qs[i * DH + c]);
        const float4 gg = *reinterpret_cast<const float4*>(&Gs[i * DH + c]);
        dk[c] = fmaf(ds, qq.x, dk[c]); dk[c + 1] = fmaf(ds, qq.y, dk[c + 1]); dk[c + 2] = fmaf(ds, qq.z, dk[c + 2]); dk[c + 3] = fmaf(ds, qq.w, dk[c + 3]);
        dv[c] = fmaf(pv, gg.x, dv[c]); dv[c + 1] = fmaf(pv, gg.y, dv[c + 1]); dv[c + 2] = fmaf(pv, gg.z, dv[c + 2]); dv[c + 3] = fmaf(pv, gg.w, dv[c + 3]);
      }
    }
  }
  // combine the 4 waves: reuse Ks / Vs as [64][68] accumulators
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        float4 ak = make_float4(dk[c], dk[c + 1], dk[c + 2], dk[c + 3]), av = make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
        if (w > 0) {
          const float4 pk = *reinterpret_cast<const float4*>(&Ks[lane * KPAD + c]), pv = *reinterpret_cast<const float4*>(&Vs[lane * KPAD + c]);
          ak.x += pk.x; ak.y += pk.y; ak.z += pk.z; ak.w += pk.w;
          av.x += pv.x; av.y += pv.y; av.z += pv.z; av.w += pv.w;
        }
        *reinterpret_cast<float4*>(&Ks[lane * KPAD + c]) = ak;
        *reinterpret_cast<float4*>(&Vs[lane * KPAD + c]) = av;
      }
    }
    __syncthreads();
  }
  T* dKp = reinterpret_cast<T*>(ba.dK) + (size_t)b * ba.dk_bs + h * DH;
  T* dVp = reinterpret_cast<T*>(ba.dV) + (size_t)b * ba.dv_bs + h * DH;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = (tid >> 4) + 16 * p, c = (tid & 15) * 4;
    if (k0 + r < a.Nk) {
      st4(dKp + (size_t)(k0 + r) * ba.dk_rs + c, *reinterpret_cast<const float4*>(&Ks[r * KPAD + c]));
      st4(dVp + (size_t)(k0 + r) * ba.dv_rs + c, *reinterpret_cast<const float4*>(&Vs[r * KPAD + c]));
    }
  }
}


// ------------------------------------------------------------------------------------------------
// bf16 attention backward on the matrix cores (the fp32 kernels above stay the exact-fp32 parity path and the fallback
// for operands that are not 16-byte aligned).  The same split into a dQ part and a dK / dV part and the same recompute from the
// forward's log-sum-exp, but as two ROLES of one launch (attention_bwd_mfma_kernel below); both reuse the forward kernel's
// operand tricks:
//   * a score tile leaves the MFMA with one operand's row index in the lane (c = lane & 15) and the other's in the
//     register index (4g + r), so per-lane scalars (lse / delta of the lane's query, or the mask term of the lane's key)
//     apply to all of a lane's accumulators;
//   * the element-wise result (dS, or P) goes STRAIGHT from those accumulators into the B operand of the next MFMA whose
//     contraction runs over the register index; the other operand is the transpose of a row-major LDS image, read with the
//     matching k-slot permutation by two ds_read_b64_tr_b16 sixteen rows apart (tfrag_rm), as the direct forward does for P.V.
// attention_bwd_mfma_dq_body : workgroups [0, ceil(Nq/64)) x NH x B, wave <-> 16 queries, keys in chunks of 128:
//     S^T = K Q^T, dP^T = V dO^T (accumulators: rows = keys, lane = query), dS^T -> B operand of dQ^T += K^T dS^T.
// attention_bwd_mfma_dkv_body: the next ceil(Nk/64) workgroups x NH x B, wave <-> 16 keys, queries in chunks of 128:
//     S = Q K^T, dP = dO V^T (rows = queries, lane = key), P and dS -> B operands of dV^T += dO^T P, dK^T += Q^T dS.
struct RowRegs { uint4 r[4]; };

// 128 rows x 64 columns of one head, row-major: thread (r = tid >> 3, c = tid & 7) holds rows r + 32p, 16-byte chunk c
__device__ __forceinline__ void ld_rows(RowRegs& L, const bf16_t* P, int rs, int r0, int nr, int tid) {
  const int r = tid >> 3, c = tid & 7;
#pragma unroll
  for (int p = 0; p < 4; ++p) L.r[p] = *reinterpret_cast<const uint4*>(P + (size_t)(r0 + min(r + 32 * p, nr - 1)) * rs + c * 8);
}
__device__ __forceinline__ void st_rows(char* S, const RowRegs& L, int nr, int ntile, int tid) {
  const int r = tid >> 3, c = tid & 7;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = r + 32 * p;
    if (row < ntile * 16) *reinterpret_cast<uint4*>(S + kswz(row, c)) = row < nr ? L.r[p] : make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ float dot8(const uint4& x, const uint4& y) {
  const float4 a0 = raw4(make_uint2(x.x, x.y)), a1 = raw4(make_uint2(x.z, x.w)), b0 = raw4(make_uint2(y.x, y.y)), b1 = raw4(make_uint2(y.z, y.w));
  return a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w + a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
}
// (device functions: both run as roles of ONE launch, attention_bwd_mfma_kernel below; `smem` = 2 row images + 2 x KC floats)
template <bool DROP>
__device__ __forceinline__ void attention_bwd_mfma_dq_body(const AttnBwdArgs& ba, const int bx, char* smem) {
  AttnArgs a = ba.f;
  if (a.p_drop > 0.f && a.seed_off) a.seed += *a.seed_off;
  char* Ks = smem;
  char* Vs = smem + KC * 128;
  float* mt = reinterpret_cast<float*>(smem + 2 * KC * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = (bx * 4 + wave) * 16;
  const bool active = q0 < a.Nq;                               // wave-uniform
  const bf16_t* Qp = reinterpret_cast<const bf16_t*>(a.Q) + b * a.q_bs + h * DH;
  const bf16_t* Kp = reinterpret_cast<const bf16_t*>(a.K) + b * a.k_bs + h * DH;
  const bf16_t* Vp = reinterpret_cast<const bf16_t*>(a.V) + b * a.v_bs + h * DH;
  const bf16_t* Op = reinterpret_cast<const bf16_t*>(a.O) + b * a.o_bs + h * DH;
  const bf16_t* Gp = reinterpret_cast<const bf16_t*>(ba.dO) + b * ba.do_bs + h * DH;
  const int qrow = q0 + c16, qc = min(qrow, a.Nq - 1);
  const bool qok = active && qrow < a.Nq;
  // this lane's query row: Q, dO, O (16 of its 64 columns each), log-sum-exp; delta = dO . O over the 4 lanes of the row
  uint4 qf[2], gf[2], of[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    qf[kk] = *reinterpret_cast<const uint4*>(Qp + (size_t)qc * a.q_rs + 32 * kk + g * 8);
    gf[kk] = *reinterpret_cast<const uint4*>(Gp + (size_t)qc * ba.do_rs + 32 * kk + g * 8);
    of[kk] = *reinterpret_cast<const uint4*>(Op + (size_t)qc * a.o_rs + 32 * kk + g * 8);
  }
  const size_t srow = ((size_t)b * a.NH + h) * a.Nq + qc;
  const float lse_q = a.lse[srow];
  float delta = dot8(gf[0], of[0]) + dot8(gf[1], of[1]);
  delta += __shfl_xor(delta, 16);
  delta += __shfl_xor(delta, 32);
  if (qok && g == 0) ba.delta[srow] = delta;
  if (!qok) { qf[0] = qf[1] = gf[0] = gf[1] = make_uint4(0, 0, 0, 0); }
  const float sw = a.dists ? a.sprel_w[0] : 0.f, sb = a.dists ? a.sprel_b[0] : 0.f;
  const float* drow = a.dists ? a.dists + ((size_t)b * a.Nq + qc) * a.Nk : nullptr;
  f32x4 dq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float gsw = 0.f, gsb = 0.f;

  for (int k0 = 0; k0 < a.Nk; k0 += KC) {
    const int nk = min(KC, a.Nk - k0);
    const int ntile = ((nk + 31) / 32) * 2;
    {
      RowRegs kr, vr;
      ld_rows(kr, Kp, a.k_rs, k0, nk, tid);
      ld_rows(vr, Vp, a.v_rs, k0, nk, tid);
      uint8_t mkb = 1;
      if (a.kmask) mkb = a.kmask[(size_t)b * a.Nk + k0 + min(tid & (KC - 1), nk - 1)];
      if (k0) __syncthreads();                                 // every wave is done with the previous chunk
      st_rows(Ks, kr, nk, ntile, tid);
      st_rows(Vs, vr, nk, ntile, tid);
      if (tid < KC) {
        float t = -INFINITY;
        if (tid < nk) {
          t = 0.f;
          if (a.kmask) {
            const bool mk = mkb != 0;
            if (a.mask_mode == DSRG_MASK_NEG10000) t = mk ? 0.f : -10000.0f;
            else if (a.mask_mode == DSRG_MASK_NEGINF) t = mk ? 0.f : -INFINITY;
            else if (a.mask_mode == DSRG_MASK_PLUS1) t = mk ? 1.0f : 0.f;
          }
        }
        mt[tid] = t;
      }
    }
    __syncthreads();
    if (!active) continue;
    auto compute = [&](auto nt_tag) {
      constexpr int NTL = decltype(nt_tag)::value;
      f32x4 s[NTL], dp[NTL];
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const uint4 kf = *reinterpret_cast<const uint4*>(Ks + kswz(t * 16 + c16, g + 4 * kk));
          const uint4 vf = *reinterpret_cast<const uint4*>(Vs + kswz(t * 16 + c16, g + 4 * kk));
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[kk]), s[t], 0, 0, 0);
          dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, gf[kk]), dp[t], 0, 0, 0);
        }
      }
      const float ik = DROP ? 1.0f / (1.0f - a.p_drop) : 1.0f;
      const size_t rbase = (((size_t)b * a.NH + h) * a.Nq + qc) * a.Nk + k0;
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const float4 mk = *reinterpret_cast<const float4*>(&mt[t * 16 + 4 * g]);
        const float mm[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kl = t * 16 + 4 * g + r;
          float v = s[t][r] * a.scale + mm[r], dist = 0.f;
          if (drow && kl < nk) { dist = drow[k0 + kl]; v += dist * sw + sb; }
          const float pr = __expf(v - lse_q);                  // 0 for padded / -inf-masked keys
          float d = dp[t][r];
          if (DROP && kl < nk) d *= drop_keep(a.seed, rbase + kl, a.p_drop, ik);
          const float ds = qok ? pr * (d - delta) : 0.f;
          gsw += ds * dist; gsb += ds;
          s[t][r] = ds;
        }
      }
#pragma unroll
      for (int j = 0; j < NTL / 2; ++j) {
        bf16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[r] = (__bf16)s[2 * j][r]; pf[4 + r] = (__bf16)s[2 * j + 1][r]; }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag_rm(Ks, dt, c16, j, g), pf, dq[dt], 0, 0, 0);
      }
    };
    switch (ntile) {
      case 2: compute(std::integral_constant<int, 2>{}); break;
      case 4: compute(std::integral_constant<int, 4>{}); break;
      case 6: compute(std::integral_constant<int, 6>{}); break;
      default: compute(std::integral_constant<int, 8>{}); break;
    }
  }
  if (qok) {
    bf16_t* dQp = reinterpret_cast<bf16_t*>(ba.dQ) + (size_t)b * ba.dq_bs + (size_t)qrow * ba.dq_rs + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      st4(dQp + dt * 16 + 4 * g, make_float4(dq[dt][0] * a.scale, dq[dt][1] * a.scale, dq[dt][2] * a.scale, dq[dt][3] * a.scale));
  }
  if (ba.dsprel && a.dists) {
    gsw = wave_sum(gsw); gsb = wave_sum(gsb);
    if (lane == 0 && active) { atomicAdd(ba.dsprel, gsw); atomicAdd(ba.dsprel + 1, gsb); }
  }
}

template <bool DROP>
__device__ __forceinline__ void attention_bwd_mfma_dkv_body(const AttnBwdArgs& ba, const int bx, char* smem) {
  AttnArgs a = ba.f;
  if (a.p_drop > 0.f && a.seed_off) a.seed += *a.seed_off;
  char* Qs = smem;
  char* Gs = smem + KC * 128;
  float* Ls = reinterpret_cast<float*>(smem + 2 * KC * 128);
  float* Dl = Ls + KC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kb = (bx * 4 + wave) * 16;
  const bool active = kb < a.Nk;                               // wave-uniform
  const bf16_t* Qp = reinterpret_cast<const bf16_t*>(a.Q) + b * a.q_bs + h * DH;
  const bf16_t* Kp = reinterpret_cast<const bf16_t*>(a.K) + b * a.k_bs + h * DH;
  const bf16_t* Vp = reinterpret_cast<const bf16_t*>(a.V) + b * a.v_bs + h * DH;
  const bf16_t* Gp = reinterpret_cast<const bf16_t*>(ba.dO) + b * ba.do_bs + h * DH;
  const bf16_t* Op = reinterpret_cast<const bf16_t*>(a.O) + b * a.o_bs + h * DH;
  const int key = kb + c16, kc_ = min(key, a.Nk - 1);
  // this lane's key row of K and V; its mask term (-inf: key beyond Nk or masked with -inf -> P = 0)
  uint4 kf[2], vf[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    kf[kk] = *reinterpret_cast<const uint4*>(Kp + (size_t)kc_ * a.k_rs + 32 * kk + g * 8);
    vf[kk] = *reinterpret_cast<const uint4*>(Vp + (size_t)kc_ * a.v_rs + 32 * kk + g * 8);
  }
  float mterm = 0.f;
  {
    uint8_t mkb = 1;
    if (a.kmask) mkb = a.kmask[(size_t)b * a.Nk + kc_];
    if (a.kmask) {
      const bool mk = mkb != 0;
      if (a.mask_mode == DSRG_MASK_NEG10000) mterm = mk ? 0.f : -10000.0f;
      else if (a.mask_mode == DSRG_MASK_NEGINF) mterm = mk ? 0.f : -INFINITY;
      else if (a.mask_mode == DSRG_MASK_PLUS1) mterm = mk ? 1.0f : 0.f;
    }
    if (!active || key >= a.Nk) mterm = -INFINITY;
  }
  const float sw = a.dists ? a.sprel_w[0] : 0.f, sb = a.dists ? a.sprel_b[0] : 0.f;
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { dk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int q0 = 0; q0 < a.Nq; q0 += KC) {
    const int nq = min(KC, a.Nq - q0);
    const int ntile = ((nq + 31) / 32) * 2;
    {
      // delta = dO . O per query row is formed HERE, from the dO rows that are staged anyway and the matching O rows (a thread
      // holds one 8-column chunk of rows r, r+32, r+64, r+96; the 8 lanes of a row add their pieces in the dQ role's order:
      // chunks c and c+4 first, then pairs), so this role does not wait for the dQ role's delta and both run in one launch
      RowRegs qr, gr, orr;
      ld_rows(qr, Qp, a.q_rs, q0, nq, tid);
      ld_rows(gr, Gp, ba.do_rs, q0, nq, tid);
      ld_rows(orr, Op, a.o_rs, q0, nq, tid);
      const size_t li = ((size_t)b * a.NH + h) * a.Nq + q0 + min(tid & (KC - 1), nq - 1);
      const float lv = a.lse[li];
      float dl[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float d = dot8(gr.r[p], orr.r[p]);
        d += __shfl_xor(d, 4);
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        dl[p] = d;
      }
      if (q0) __syncthreads();                                 // every wave is done with the previous chunk
      st_rows(Qs, qr, nq, ntile, tid);
      st_rows(Gs, gr, nq, ntile, tid);
      if (tid < KC) Ls[tid] = tid < nq ? lv : INFINITY;        // exp(s - inf) = 0 for the padding rows
      if ((tid & 7) == 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) Dl[(tid >> 3) + 32 * p] = (tid >> 3) + 32 * p < nq ? dl[p] : 0.f;
      }
    }
    __syncthreads();
    if (!active) continue;
    auto compute = [&](auto nt_tag) {
      constexpr int NTL = decltype(nt_tag)::value;
      f32x4 s[NTL], dp[NTL];
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const uint4 qv = *reinterpret_cast<const uint4*>(Qs + kswz(t * 16 + c16, g + 4 * kk));
          const uint4 gv = *reinterpret_cast<const uint4*>(Gs + kswz(t * 16 + c16, g + 4 * kk));
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qv), __builtin_bit_cast(bf16x8, kf[kk]), s[t], 0, 0, 0);
          dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gv), __builtin_bit_cast(bf16x8, vf[kk]), dp[t], 0, 0, 0);
        }
      }
      const float ik = DROP ? 1.0f / (1.0f - a.p_drop) : 1.0f;
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const float4 l4 = *reinterpret_cast<const float4*>(&Ls[t * 16 + 4 * g]), d4 = *reinterpret_cast<const float4*>(&Dl[t * 16 + 4 * g]);
        const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = t * 16 + 4 * g + r, qg = q0 + ql;
          float v = s[t][r] * a.scale + mterm;
          if (a.dists && ql < nq && key < a.Nk) v += a.dists[((size_t)b * a.Nq + qg) * a.Nk + key] * sw + sb;
          const float pr = __expf(v - ll[r]);
          float keep = 1.f;
          if (DROP && ql < nq && key < a.Nk) keep = drop_keep(a.seed, (((size_t)b * a.NH + h) * a.Nq + qg) * a.Nk + key, a.p_drop, ik);
          const float d = dp[t][r] * keep;
          s[t][r] = pr * keep;                                 // what multiplied V in the forward
          dp[t][r] = pr * (d - dd[r]) * a.scale;
        }
      }
#pragma unroll
      for (int j = 0; j < NTL / 2; ++j) {
        bf16x8 pf, df;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pf[r] = (__bf16)s[2 * j][r]; pf[4 + r] = (__bf16)s[2 * j + 1][r];
          df[r] = (__bf16)dp[2 * j][r]; df[4 + r] = (__bf16)dp[2 * j + 1][r];
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag_rm(Gs, dt, c16, j, g), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tfrag_rm(Qs, dt, c16, j, g), df, dk[dt], 0, 0, 0);
        }
      }
    };
    switch (ntile) {
      case 2: compute(std::integral_constant<int, 2>{}); break;
      case 4: compute(std::integral_constant<int, 4>{}); break;
      case 6: compute(std::integral_constant<int, 6>{}); break;
      default: compute(std::integral_constant<int, 8>{}); break;
    }
  }
  if (active && key < a.Nk) {
    bf16_t* dKp = reinterpret_cast<bf16_t*>(ba.dK) + (size_t)b * ba.dk_bs + (size_t)key * ba.dk_rs + h * DH;
    bf16_t* dVp = reinterpret_cast<bf16_t*>(ba.dV) + (size_t)b * ba.dv_bs + (size_t)key * ba.dv_rs + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      st4(dKp + dt * 16 + 4 * g, make_float4(dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]));
      st4(dVp + dt * 16 + 4 * g, make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]));
    }
  }
}

// One launch for both roles (late round 3): workgroups [0, nqb) compute dQ (and store delta, an output of the entry points),
// the others dK / dV.  As two launches the second waited for the first (it read the first one's delta) and each ran on
// 2-3 waves per SIMD: 26 launch pairs of ~20 us each per pre-training step, latency- not throughput-bound.
template <bool DROP>
__global__ __launch_bounds__(256) void attention_bwd_mfma_kernel(const AttnBwdArgs ba, const int nqb) {
  __shared__ __attribute__((aligned(16))) char smem[2 * KC * 128 + 2 * KC * 4];
  if ((int)blockIdx.x < nqb) attention_bwd_mfma_dq_body<DROP>(ba, blockIdx.x, smem);
  else attention_bwd_mfma_dkv_body<DROP>(ba, blockIdx.x - nqb, smem);
}

}  // namespace

#ifdef DSRG_ATTN_STAMPS
extern "C" int dsrg_debug_attn_stamps(void* host_dst, int nbytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_attn_stamps), nbytes, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int dsrg_attention_fwd_grouped(const dsrg_attn_problem_t* problems, int nproblems, int mask_mode,
                                          const float* sprel_w, const float* sprel_b, int B, int NH, float scale, float p_drop,
                                          uint64_t seed, int dtype, void* stream) {
  DSRG_CHECK(p_drop >= 0.f && p_drop < 1.f, "dsrg_attention_fwd: p_drop=%g out of [0,1)", (double)p_drop);
  DSRG_CHECK(problems && nproblems >= 1 && nproblems <= DSRG_ATTN_MAX_GROUPS, "dsrg_attention_fwd: nproblems=%d out of [1,%d]", nproblems,
             DSRG_ATTN_MAX_GROUPS);
  DSRG_CHECK(dtype == DSRG_F32 || dtype == DSRG_BF16, "dsrg_attention_fwd: bad dtype %d", dtype);
  DSRG_CHECK(B >= 0 && NH >= 1 && B <= 65535 && NH <= 65535, "dsrg_attention_fwd: bad dims B=%d NH=%d", B, NH);
  DSRG_CHECK(mask_mode >= DSRG_MASK_NONE && mask_mode <= DSRG_MASK_PLUS1, "dsrg_attention_fwd: bad mask_mode %d", mask_mode);
  const uintptr_t am = dtype == DSRG_BF16 ? 7 : 15;
  AttnGroup grp;
  grp.n = 0;
  bool mfma_ok = dtype == DSRG_BF16;
  for (int i = 0; i < nproblems; ++i) {
    const dsrg_attn_problem_t& p = problems[i];
    DSRG_CHECK(p.Nq >= 0 && p.Nk >= 1, "dsrg_attention_fwd: problem %d has bad dims Nq=%d Nk=%d", i, p.Nq, p.Nk);
    if (p.Nq == 0) continue;
    DSRG_CHECK(p.Q && p.K && p.V && p.O, "dsrg_attention_fwd: problem %d has a null Q/K/V/O", i);
    DSRG_CHECK(mask_mode == DSRG_MASK_NONE || p.kmask || p.klens, "dsrg_attention_fwd: mask_mode %d needs kmask or klens (problem %d)", mask_mode, i);
    DSRG_CHECK(!p.klens || (p.kseg >= 1 && !p.kmask), "dsrg_attention_fwd: problem %d: klens needs kseg >= 1 and no kmask", i);
    DSRG_CHECK(!p.klens2 || p.klens, "dsrg_attention_fwd: problem %d: klens2 without klens", i);
    DSRG_CHECK(!p.pair_dists || (sprel_w && sprel_b), "dsrg_attention_fwd: pair_dists needs sprel_w and sprel_b");
    DSRG_CHECK(p.q_rs % 4 == 0 && p.k_rs % 4 == 0 && p.v_rs % 4 == 0 && p.q_bs % 4 == 0 && p.k_bs % 4 == 0 && p.v_bs % 4 == 0,
               "dsrg_attention_fwd: problem %d strides must be multiples of 4 elements", i);
    DSRG_CHECK(p.q_rs >= NH * 64 && p.k_rs >= NH * 64 && p.v_rs >= NH * 64 && p.o_rs >= NH * 64,
               "dsrg_attention_fwd: problem %d row strides smaller than NH*64", i);
    DSRG_CHECK(((uintptr_t)p.Q & am) == 0 && ((uintptr_t)p.K & am) == 0 && ((uintptr_t)p.V & am) == 0, "dsrg_attention_fwd: problem %d misaligned Q/K/V", i);
    mfma_ok = mfma_ok && p.q_rs % 8 == 0 && p.k_rs % 8 == 0 && p.v_rs % 8 == 0 && p.q_bs % 8 == 0 && p.k_bs % 8 == 0 && p.v_bs % 8 == 0 &&
              p.o_rs % 4 == 0 && p.o_bs % 4 == 0 && (((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V) & 15) == 0 && ((uintptr_t)p.O & 7) == 0;
    grp.a[grp.n++] = AttnArgs{p.Q, p.K, p.V, p.O, p.q_bs, p.k_bs, p.v_bs, p.o_bs, p.q_rs, p.k_rs, p.v_rs, p.o_rs,
                              mask_mode == DSRG_MASK_NONE ? nullptr : p.kmask, mask_mode, p.pair_dists, sprel_w, sprel_b, p.Nq, p.Nk, scale,
                              p.lse, NH, p_drop, seed + 0x632BE59BD9B4E019ull * (unsigned long long)i,
                              mask_mode == DSRG_MASK_NONE ? nullptr : p.klens, p.kseg, mask_mode == DSRG_MASK_NONE ? nullptr : p.klens2,
                              p_drop > 0.f ? dsrg_dropout_offset_ptr() : nullptr};
  }
  if (B == 0 || grp.n == 0) return DSRG_OK;
  const int qpb = mfma_ok ? 64 : QPB;           // query rows per workgroup
  int blocks = 0;
  for (int i = 0; i < grp.n; ++i) { grp.start[i] = blocks; blocks += (grp.a[i].Nq + qpb - 1) / qpb; }
  grp.start[grp.n] = blocks;
  for (int i = grp.n + 1; i <= DSRG_ATTN_MAX_GROUPS; ++i) grp.start[i] = 0x7fffffff;      // (the kernels' problem search)
  dim3 grid(blocks, NH, B), block(256);
  const bool drop = p_drop > 0.f;      // the dropout variants are separate instantiations: the inference kernels carry none of it
  hipStream_t st = (hipStream_t)stream;
  bool any_dists = false;
  for (int i = 0; i < grp.n; ++i) any_dists = any_dists || grp.a[i].dists != nullptr;
  int max_nk = 0;
  for (int i = 0; i < grp.n; ++i) max_nk = grp.a[i].Nk > max_nk ? grp.a[i].Nk : max_nk;
  // DSRG_ATTN_DIRECT (A/B aid): 0 = never, 2 = whenever it applies, default = only when the K rows lie in a packed q|k|v or
  // k|v buffer: three waves of a workgroup each fetch every K row, which is cheap out of L2 (the projection just written)
  // and a loss for the text K/V panel, whose 15.7 MB per launch come cold from HBM (r03s stamps: cross-attention workgroups
  // 11.3-11.9 k cycles direct vs 10.5-10.8 k staged; self-attention 5.5-7.5 k vs 7.3-8.8 k)
  static const int direct_mode = []() { const char* e = getenv("DSRG_ATTN_DIRECT"); return e ? atoi(e) : 1; }();
  bool k_near = true;
  for (int i = 0; i < grp.n; ++i) k_near = k_near && grp.a[i].k_rs <= 4 * NH * 64;
  if (mfma_ok && max_nk <= KC && (direct_mode == 2 || (direct_mode == 1 && k_near))) {
    const int ntl = ((max_nk + 31) / 32) * 2;
#define DSRG_ATTN_DIRECT_LAUNCH(NTL_)                                                                                       \
    do {                                                                                                                    \
      if (any_dists && drop) hipLaunchKernelGGL((attention_mfma_direct_kernel<NTL_, true, true>), grid, block, 0, st, grp);   \
      else if (any_dists) hipLaunchKernelGGL((attention_mfma_direct_kernel<NTL_, true, false>), grid, block, 0, st, grp);     \
      else if (drop) hipLaunchKernelGGL((attention_mfma_direct_kernel<NTL_, false, true>), grid, block, 0, st, grp);          \
      else hipLaunchKernelGGL((attention_mfma_direct_kernel<NTL_, false, false>), grid, block, 0, st, grp);                   \
    } while (0)
    if (ntl == 2) DSRG_ATTN_DIRECT_LAUNCH(2);
    else if (ntl == 4) DSRG_ATTN_DIRECT_LAUNCH(4);
    else if (ntl == 6) DSRG_ATTN_DIRECT_LAUNCH(6);
    else DSRG_ATTN_DIRECT_LAUNCH(8);
#undef DSRG_ATTN_DIRECT_LAUNCH
  } else if (mfma_ok) {
    if (any_dists) {
      if (drop) hipLaunchKernelGGL((attention_mfma_kernel<true, true>), grid, block, 0, st, grp);
      else hipLaunchKernelGGL((attention_mfma_kernel<false, true>), grid, block, 0, st, grp);
    } else {
      if (drop) hipLaunchKernelGGL((attention_mfma_kernel<true, false>), grid, block, 0, st, grp);
      else hipLaunchKernelGGL((attention_mfma_kernel<false, false>), grid, block, 0, st, grp);
    }
  } else if (dtype == DSRG_F32) {
    if (drop) hipLaunchKernelGGL((attention_kernel<float, true>), grid, block, 0, st, grp);
    else hipLaunchKernelGGL((attention_kernel<float, false>), grid, block, 0, st, grp);
  } else {
    if (drop) hipLaunchKernelGGL((attention_kernel<bf16_t, true>), grid, block, 0, st, grp);
    else hipLaunchKernelGGL((attention_kernel<bf16_t, false>), grid, block, 0, st, grp);
  }
  DSRG_LAUNCH_CHECK("dsrg_attention_fwd");
  return DSRG_OK;
}

extern "C" int dsrg_attention_fwd(const void* Q, int64_t q_bs, int q_rs, const void* K, int64_t k_bs, int k_rs,
                                  const void* V, int64_t v_bs, int v_rs, void* O, int64_t o_bs, int o_rs,
                                  const uint8_t* kmask, int mask_mode, const float* pair_dists,
                                  const float* sprel_w, const float* sprel_b,
                                  int B, int NH, int Nq, int Nk, float scale, float* lse, float p_drop, uint64_t seed, int dtype,
                                  void* stream) {
  const dsrg_attn_problem_t p{Q, K, V, O, q_bs, k_bs, v_bs, o_bs, q_rs, k_rs, v_rs, o_rs, kmask, pair_dists, lse, Nq, Nk, nullptr, 0, nullptr};
  return dsrg_attention_fwd_grouped(&p, 1, mask_mode, sprel_w, sprel_b, B, NH, scale, p_drop, seed, dtype, stream);
}

static int attention_bwd_impl(const void* Q, int64_t q_bs, int q_rs, const void* K, int64_t k_bs, int k_rs,
                              const void* V, int64_t v_bs, int v_rs, const void* O, int64_t o_bs, int o_rs,
                              const void* dO, int64_t do_bs, int do_rs, const float* lse,
                              const uint8_t* kmask, int mask_mode, const float* pair_dists,
                              const float* sprel_w, const float* sprel_b,
                              void* dQ, void* dK, void* dV, float* delta_ws, float* dsprel,
                              int B, int NH, int Nq, int Nk, float scale, float p_drop, uint64_t seed, int dtype, void* stream,
                              const int64_t* gst = nullptr /* dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs or contiguous */) {
  const int64_t hh = (int64_t)NH * DH;
  const int64_t dq_bs = gst ? gst[0] : Nq * hh, dq_rs = gst ? gst[1] : hh, dk_bs = gst ? gst[2] : Nk * hh, dk_rs = gst ? gst[3] : hh,
                dv_bs = gst ? gst[4] : Nk * hh, dv_rs = gst ? gst[5] : hh;
  DSRG_CHECK(dq_rs >= hh && dk_rs >= hh && dv_rs >= hh && dq_rs % 4 == 0 && dk_rs % 4 == 0 && dv_rs % 4 == 0 && dq_bs % 4 == 0 && dk_bs % 4 == 0 &&
                 dv_bs % 4 == 0 && dq_rs < (1 << 30) && dk_rs < (1 << 30) && dv_rs < (1 << 30) &&
                 (B <= 1 || (dq_bs >= (int64_t)Nq * hh && dk_bs >= (int64_t)Nk * hh && dv_bs >= (int64_t)Nk * hh)),
             "dsrg_attention_bwd: gradient strides must be multiples of 4 elements, rows >= NH*64 and batches >= N*NH*64");
  DSRG_CHECK(p_drop >= 0.f && p_drop < 1.f, "dsrg_attention_bwd: p_drop=%g out of [0,1)", (double)p_drop);
  DSRG_CHECK(dtype == DSRG_F32 || dtype == DSRG_BF16, "dsrg_attention_bwd: bad dtype %d", dtype);
  DSRG_CHECK(Q && K && V && O && dO && lse && dQ && dK && dV && delta_ws, "dsrg_attention_bwd: null pointer");
  DSRG_CHECK(B >= 0 && NH >= 1 && Nq >= 0 && Nk >= 1 && B <= 65535 && NH <= 65535, "dsrg_attention_bwd: bad dims");
  DSRG_CHECK(mask_mode >= DSRG_MASK_NONE && mask_mode <= DSRG_MASK_PLUS1 && (mask_mode == DSRG_MASK_NONE || kmask), "dsrg_attention_bwd: bad mask");
  DSRG_CHECK(!pair_dists || (sprel_w && sprel_b), "dsrg_attention_bwd: pair_dists needs sprel_w and sprel_b");
  DSRG_CHECK(q_rs % 4 == 0 && k_rs % 4 == 0 && v_rs % 4 == 0 && o_rs % 4 == 0 && do_rs % 4 == 0 && q_bs % 4 == 0 && k_bs % 4 == 0 &&
                 v_bs % 4 == 0 && o_bs % 4 == 0 && do_bs % 4 == 0, "dsrg_attention_bwd: strides must be multiples of 4 elements");
  const uintptr_t am = dtype == DSRG_BF16 ? 7 : 15;
  DSRG_CHECK((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO | (uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & am) == 0,
             "dsrg_attention_bwd: misaligned pointer");
  if (B == 0 || Nq == 0) return DSRG_OK;
  AttnBwdArgs ba{{Q, K, V, const_cast<void*>(O), q_bs, k_bs, v_bs, o_bs, q_rs, k_rs, v_rs, o_rs, mask_mode == DSRG_MASK_NONE ? nullptr : kmask,
                  mask_mode, pair_dists, sprel_w, sprel_b, Nq, Nk, scale, const_cast<float*>(lse), NH, p_drop, seed, nullptr, 0, nullptr,
                  p_drop > 0.f ? dsrg_dropout_offset_ptr() : nullptr},
                 dO, do_bs, do_rs, dQ, dK, dV, dq_bs, dk_bs, dv_bs, (int)dq_rs, (int)dk_rs, (int)dv_rs, delta_ws, dsprel};
  hipStream_t s = (hipStream_t)stream;
  dim3 gq((Nq + QPB - 1) / QPB, NH, B), gk((Nk + KT - 1) / KT, NH, B), block(256);
  // bf16 with 16-byte addressable rows: the matrix-core kernels (64 queries / 64 keys per workgroup)
  const bool mfma_ok = dtype == DSRG_BF16 && q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 8 == 0 && do_rs % 8 == 0 && q_bs % 8 == 0 &&
                       k_bs % 8 == 0 && v_bs % 8 == 0 && o_bs % 8 == 0 && do_bs % 8 == 0 &&
                       (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO) & 15) == 0;
  if (mfma_ok) {
    const int nqb = (Nq + 63) / 64, nkb = (Nk + 63) / 64;
    static const char* merged_env = getenv("DSRG_ATTN_BWD_MERGED");      // 0: the two roles as two launches (A/B)
    if (!merged_env || merged_env[0] != '0') {
      dim3 mg(nqb + nkb, NH, B);
      if (p_drop > 0.f) hipLaunchKernelGGL((attention_bwd_mfma_kernel<true>), mg, block, 0, s, ba, nqb);
      else hipLaunchKernelGGL((attention_bwd_mfma_kernel<false>), mg, block, 0, s, ba, nqb);
    } else {
      dim3 mq(nqb, NH, B), mk(nkb, NH, B);
      if (p_drop > 0.f) {
        hipLaunchKernelGGL((attention_bwd_mfma_kernel<true>), mq, block, 0, s, ba, nqb);
        hipLaunchKernelGGL((attention_bwd_mfma_kernel<true>), mk, block, 0, s, ba, 0);
      } else {
        hipLaunchKernelGGL((attention_bwd_mfma_kernel<false>), mq, block, 0, s, ba, nqb);
        hipLaunchKernelGGL((attention_bwd_mfma_kernel<false>), mk, block, 0, s, ba, 0);
      }
    }
    DSRG_LAUNCH_CHECK("dsrg_attention_bwd");
    return DSRG_OK;
  }
  if (dtype == DSRG_F32) {
    hipLaunchKernelGGL((attention_bwd_dq_kernel<float>), gq, block, 0, s, ba);
    hipLaunchKernelGGL((attention_bwd_dkv_kernel<float>), gk, block, 0, s, ba);
  } else {
    hipLaunchKernelGGL((attention_bwd_dq_kernel<bf16_t>), gq, block, 0, s, ba);
    hipLaunchKernelGGL((attention_bwd_dkv_kernel<bf16_t>), gk, block, 0, s, ba);
  }
  DSRG_LAUNCH_CHECK("dsrg_attention_bwd");
  return DSRG_OK;
}

extern "C" int dsrg_attention_bwd(const void* Q, int64_t q_bs, int q_rs, const void* K, int64_t k_bs, int k_rs,
                                  const void* V, int64_t v_bs, int v_rs, const void* O, int64_t o_bs, int o_rs,
                                  const void* dO, int64_t do_bs, int do_rs, const float* lse,
                                  const uint8_t* kmask, int mask_mode, const float* pair_dists,
                                  const float* sprel_w, const float* sprel_b,
                                  void* dQ, void* dK, void* dV, float* delta_ws, float* dsprel,
                                  int B, int NH, int Nq, int Nk, float scale, float p_drop, uint64_t seed, int dtype, void* stream) {
  return attention_bwd_impl(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, O, o_bs, o_rs, dO, do_bs, do_rs, lse, kmask, mask_mode, pair_dists,
                            sprel_w, sprel_b, dQ, dK, dV, delta_ws, dsprel, B, NH, Nq, Nk, scale, p_drop, seed, dtype, stream);
}

extern "C" int dsrg_attention_bwd_strided(const void* Q, int64_t q_bs, int q_rs, const void* K, int64_t k_bs, int k_rs,
                                          const void* V, int64_t v_bs, int v_rs, const void* O, int64_t o_bs, int o_rs,
                                          const void* dO, int64_t do_bs, int do_rs, const float* lse,
                                          const uint8_t* kmask, int mask_mode, const float* pair_dists,
                                          const float* sprel_w, const float* sprel_b,
                                          void* dQ, int64_t dq_bs, int dq_rs, void* dK, int64_t dk_bs, int dk_rs, void* dV, int64_t dv_bs, int dv_rs,
                                          float* delta_ws, float* dsprel,
                                          int B, int NH, int Nq, int Nk, float scale, float p_drop, uint64_t seed, int dtype, void* stream) {
  const int64_t gst[6] = {dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs};
  return attention_bwd_impl(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, O, o_bs, o_rs, dO, do_bs, do_rs, lse, kmask, mask_mode, pair_dists,
                            sprel_w, sprel_b, dQ, dK, dV, delta_ws, dsprel, B, NH, Nq, Nk, scale, p_drop, seed, dtype, stream, gst);
}

// ---- the allocation-free variant (include/dsrg_hip.h: dsrg_attention_bwd_noalloc).  The delta workspace
// (delta_i = dO_i . O_i per (episode, head, query row), written by the dQ kernel, read by the dK/dV kernel) lives in a
// file-scope device array instead of caller memory.  The array's name is a HOST-side handle: the device address is
// taken once per device with hipGetSymbolAddress and only that pointer is ever handed to a kernel.
static __device__ __align__(256) float scratch[DSRG_ATTN_BWD_NOALLOC_MAX_ROWS];     // 4 MiB on every GPU the library runs on

static int scratch_ptr(float** out, hipStream_t capturing_on = nullptr, bool check_capture = false) {
  static std::mutex mu;
  static float* ptr[DSRG_MAX_DEVICES] = {};
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DSRG_MAX_DEVICES) {
    dsrg_set_error("dsrg_attention_bwd_noalloc: no current device, or device index %d >= %d", dev, DSRG_MAX_DEVICES);
    return DSRG_EINVAL;
  }
  std::lock_guard<std::mutex> g(mu);
  if (!ptr[dev]) {
    if (check_capture) {     // the symbol lookup below may load the code object: never inside a stream capture
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(capturing_on, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        dsrg_set_error("dsrg_attention_bwd_noalloc: first use on device %d inside a stream capture; call dsrg_init() on this device before capturing", dev);
        return DSRG_EINVAL;
      }
    }
    float* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));     // this device's copy
    if (e != hipSuccess || !p) {
      dsrg_set_error("dsrg_attention_bwd_noalloc: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
      return DSRG_ELAUNCH;
    }
    ptr[dev] = p;
  }
  *out = ptr[dev];
  return DSRG_OK;
}

extern "C" int dsrg_attention_bwd_noalloc(const void* Q, int64_t q_bs, int q_rs, const void* K, int64_t k_bs, int k_rs,
                                          const void* V, int64_t v_bs, int v_rs, const void* O, int64_t o_bs, int o_rs,
                                          const void* dO, int64_t do_bs, int do_rs, const float* lse,
                                          const uint8_t* kmask, int mask_mode, const float* pair_dists,
                                          const float* sprel_w, const float* sprel_b,
                                          void* dQ, void* dK, void* dV, float* dsprel,
                                          int B, int NH, int Nq, int Nk, float scale, float p_drop, uint64_t seed, int dtype, void* stream) {
  DSRG_CHECK(B >= 0 && NH >= 1 && Nq >= 0, "dsrg_attention_bwd_noalloc: bad dims");
  DSRG_CHECK((int64_t)B * NH * Nq <= (int64_t)DSRG_ATTN_BWD_NOALLOC_MAX_ROWS,
             "dsrg_attention_bwd_noalloc: B*NH*Nq = %lld exceeds the static workspace of %d rows; use dsrg_attention_bwd with a caller workspace",
             (long long)B * NH * Nq, DSRG_ATTN_BWD_NOALLOC_MAX_ROWS);
  float* ws = nullptr;
  const int rc = scratch_ptr(&ws, (hipStream_t)stream, true);
  if (rc != DSRG_OK) return rc;
  // nothing to clear: rows [0, B*NH*Nq) are all written by the dQ kernel before the dK/dV kernel (same stream) reads them
  return attention_bwd_impl(Q, q_bs, q_rs, K, k_bs, k_rs, V, v_bs, v_rs, O, o_bs, o_rs, dO, do_bs, do_rs, lse, kmask, mask_mode, pair_dists,
                            sprel_w, sprel_b, dQ, dK, dV, ws, dsprel, B, NH, Nq, Nk, scale, p_drop, seed, dtype, stream);
}

// Per-device set-up that must not happen inside a stream capture (include/dsrg_hip.h): resolves the static workspace's
// device address for the CURRENT device.  Idempotent, thread-safe.
extern "C" int dsrg_init(void) {
  float* ws = nullptr;
  return scratch_ptr(&ws);
}
