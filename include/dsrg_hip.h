/* dsrg_hip.h — C-ABI of libdsrg_hip.so, the MI355X (gfx950) kernel library under the DSRG
 * navigation-policy network.
 *
 * The reference (CrystalSixone/DSRG) has no native boundary on this path: every op below is a
 * torch.nn call inside finetune_src/models/vilmodel.py (FT/) or pretrain_src/model/ (PT/).  Each entry
 * point cites the reference lines whose arithmetic it replaces.  Conventions:
 *   - every function returns 0 on success, a negative DSRG_E* code on failure; dsrg_last_error()
 *     returns a thread-local message for the last failure on the calling thread;
 *   - all pointers are DEVICE pointers unless the parameter comment says "host";
 *   - `stream` is a hipStream_t passed as void*; the call only enqueues work on it;
 *   - no entry point allocates, frees or synchronises; they are safe under hipGraph capture;
 *   - `dtype` is DSRG_F32 or DSRG_BF16 and names the storage type of activations and weights;
 *     biases, LayerNorm affine parameters, masks-derived terms and all statistics are fp32;
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef DSRG_HIP_H
#define DSRG_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DSRG_ABI_VERSION 14

enum { DSRG_F32 = 0, DSRG_BF16 = 1 };
enum { DSRG_OK = 0, DSRG_EINVAL = -1, DSRG_ELAUNCH = -2, DSRG_EUNSUPPORTED = -3 };

int dsrg_abi_version(void);
const char* dsrg_last_error(void);
/* Per-device set-up, for the CURRENT device (hipSetDevice first): resolves the device address of the library's static
 * workspace (dsrg_attention_bwd_noalloc).  Optional for eager callers - the first dsrg_attention_bwd_noalloc call on a
 * device does it - but REQUIRED before that entry point is used under hipGraph / stream capture on the device: the lookup
 * may load the code object and must not run inside a capture (the entry point then fails with DSRG_EINVAL and says so).
 * Idempotent and thread-safe.  No counterpart in the reference. */
int dsrg_init(void);
/* Dropout under hipGraph replay (training; no counterpart in the reference, whose eager torch.nn.Dropout draws from the
 * device generator): every dropout kernel of the library - attention probabilities, hidden states in front of a residual
 * LayerNorm, forward and backward - takes its seed by value, so a captured training step would replay ONE mask.  With a
 * device pointer registered here (current device; NULL = off) they add *dev_ptr to the seed at run time: the captured step
 * advances that word once per replay and draws a new mask each time, forward and backward still agreeing. */
int dsrg_set_dropout_offset(const uint64_t* dev_ptr);

/* ---- optimizer step of the pre-training path: gradient-norm clipping + AdamW over every parameter in three launches.
 * Replaces torch.nn.utils.clip_grad_norm_ (pretrain_src/train_r2r.py:283-289) and the reference's own AdamW.step()
 * (pretrain_src/optim/adamw.py:30-92: m, v moving averages, denom = sqrt(v) + eps, step = lr * sqrt(1 - b2^t) / (1 - b1^t)
 * when correct_bias, decoupled weight decay p -= lr * wd * p applied after the update).
 * All tables live in device memory (the caller builds them once per set of tensors):
 *   table   one entry per parameter tensor that has a gradient this step (fp32, contiguous);
 *   blocks  int32 pairs (tensor index, chunk index), one per workgroup; a chunk is dsrg_opt_chunk() elements;
 *   hyper   float pairs (lr, weight_decay) per parameter group - device words, so a schedule can move lr without recapture;
 *   steps   int32 update counters, one per parameter (entry.step_idx), advanced by dsrg_adamw_prepare;
 *   norm    float[2] written by dsrg_adamw_prepare: total gradient norm (after grad_scale), coefficient applied to g.
 * Order per step: dsrg_grad_sumsq (optional: only when clipping) -> dsrg_adamw_prepare -> dsrg_adamw_step.  The gradients
 * themselves are left unscaled (the coefficient is applied where the update reads them). */
typedef struct {
  void* p;              /* parameter, fp32 [numel] */
  const void* g;        /* gradient, fp32 [numel] (may be NULL when g2 is given) */
  float* m;             /* exp_avg */
  float* v;             /* exp_avg_sq */
  int64_t numel;
  int32_t group;        /* index into hyper */
  int32_t step_idx;     /* index into steps */
  int32_t vec_ok;       /* p, g, m, v all 16-byte aligned (and g2, pack 8-byte aligned) */
  int32_t reserved_;
  const void* g2;       /* optional bf16 [numel]: a (further) share of the gradient, added to g where it is read - the slice of a
                           packed weight's bf16 gradient, consumed as it lies instead of being cast into an fp32 .grad first.
                           At least one of g, g2 is non-NULL */
  void* pack;           /* optional bf16 [numel]: receives the updated parameter rounded to bf16 - the packed / cast copy the next
                           forward reads, so that no separate cast pass over the parameters is needed; NULL = none */
} dsrg_opt_tensor_t;
int dsrg_opt_chunk(void);
/* partials[b] = sum of g^2 over block b's chunk */
int dsrg_grad_sumsq(const dsrg_opt_tensor_t* table, const int32_t* blocks, int n_blocks, float* partials, hipStream_t stream);
/* norm_out[0] = grad_scale * sqrt(sum partials); norm_out[1] = grad_scale * min(1, max_norm / (norm + 1e-6)) (n_partials = 0
 * or max_norm <= 0: no clipping, coefficient = grad_scale); steps[table[i].step_idx] += 1 for i < n_tensors.
 * grad_scale folds a pending division of the gradients (1 / world size after a summing all-reduce) into the same read. */
int dsrg_adamw_prepare(const float* partials, int n_partials, float max_norm, float grad_scale, float* norm_out,
                       const dsrg_opt_tensor_t* table, int n_tensors, int32_t* steps, hipStream_t stream);
/* norm may be NULL (coefficient 1).  betas / eps are doubles: the reference rounds 1 - beta from double (Python scalars) */
int dsrg_adamw_step(const dsrg_opt_tensor_t* table, const int32_t* blocks, int n_blocks, const float* hyper, const float* norm,
                    const int32_t* steps, double beta1, double beta2, double eps, int correct_bias, hipStream_t stream);

/* ---- GEMM: C = act(A · Wᵀ + bias) [+ residual] ------------------------------------------------
 * Replaces every nn.Linear on the path (FT/vilmodel.py:111-113,169,195,209,338-340,555,577,622;
 * FT/transformer.py:138-142 in_proj/out_proj/linear1/linear2; PT/pretrain_cmt.py:16-19,30-33;
 * PT/vilmodel.py BertLMPredictionHead decoder).  A is [M,K] (lda), W is [N,K] (ldw, nn.Linear layout),
 * bias is fp32 [N] or NULL, residual is [M,N] (ldr, storage dtype `dtype`) or NULL and is added
 * AFTER the activation, C is [M,N] (ldc) stored as `out_dtype`.
 * Up to DSRG_MAX_GROUPS independent problems run in one launch (grouped GEMM: the global- and
 * local-branch x-layers, the three RMF poolers, the SAP heads).
 * Requirements: K % 64 == 0 (bf16) / K % 32 == 0 (f32); A, W 16-byte aligned, lda/ldw multiples of
 * 8 (bf16) / 4 (f32); bias 16-byte aligned; M*lda and N*ldw below 4 GiB in bytes (32-bit row offsets in the kernel).
 * M, N arbitrary.  With act = GELU and a bf16 output the erf-form GELU is evaluated through a 5-term erfc expansion
 * (|error| <= 4.3e-7, far below one bf16 ulp); fp32 outputs use libm erff. */
enum { DSRG_ACT_NONE = 0, DSRG_ACT_GELU = 1, DSRG_ACT_RELU = 2, DSRG_ACT_TANH = 3 };
#define DSRG_MAX_GROUPS 8
typedef struct {
  const void* A; const void* W; const float* bias; const void* residual; void* C;
  int32_t M, N, K;
  int32_t lda, ldw, ldr, ldc;
  int32_t reserved0_;     /* (the alignment hole in front of the next pointer, named: set to 0) */
  /* Optional LayerNorm fusion (all NULL / 0 = plain GEMM).  A post-LN block  x1 = LN(y),  y = dense(..) + residual
   * (BertSelfOutput / BertOutput, FT/vilmodel.py:173-177, 213-217) followed by projections of x1 needs no standalone
   * LayerNorm launch and no fp32 copy of y:
   *  Partial statistics are float2 (sum, sum of squares) per row and per 64-column block, stored BLOCK-major: table[p * M + m]
   *  for block p of row m (ABI v12; v11 was row-major): a wave's 64 rows of one block are contiguous.
   *  - stats_out: float2 [N/64][M]: the launch that produces y also writes the partials of the fp32 values it stores
   *    (N % 64 == 0);
   *  - a_stats / a_np / a_c1: A holds the UN-normalised y [M,K] of a previous launch and a_stats its partials
   *    (a_np per row, a_np = K/64 <= 16).  With W' = W * diag(gamma) as the weight, a_c1[n] = sum_k W'[n,k] and
   *    bias[n] = b[n] + sum_k beta[k] W[n,k] the epilogue computes  rstd_m * (acc - mu_m * a_c1[n]) + bias[n]
   *    = LN(y) W^T + b  (exact in real arithmetic; mu, rstd over the K columns of A with ln_eps);
   *  - r_stats / r_np / r_gamma / r_beta: the residual holds an un-normalised y' [M,N] and r_stats its partials; the
   *    residual added is LN(y')[m,n] = (y' - mu'_m) * rstd'_m * r_gamma[n] + r_beta[n].
   *  A problem uses either a_stats or r_stats / stats_out, and all fused problems of one launch are of one kind.  Tiles built
   *  for the fused epilogue: 64x64 (both dtypes), 128x128 for a_stats (bf16 in and out) and 128x64 for r_stats / stats_out
   *  (bf16); the plan picks among them. */
  const float* a_stats; const float* a_c1;
  const float* r_stats; const float* r_gamma; const float* r_beta;
  float* stats_out;
  int32_t a_np, r_np;
  float ln_eps;
  /* Operand storage for the backward GEMMs of a linear layer (replaces torch.autograd's  dX = dY @ W,  dW = dY.T @ X  for
   * the nn.Linear calls of FT/vilmodel.py:111-113,195,209; PT/vilmodel.py twins) without transposed copies in HBM.
   * 0: A [M,K], W [N,K] as above.  Bit 0: A is stored [K, M] (row stride lda >= M); bit 1: W is stored [K, N] (ldw >= N).
   *   dX[M,K'] = dY[M,N'] . W[N',K']   -> A = dY, W = the weight as it lies ([N',K'] = [K, N] of this GEMM), trans = 2
   *   dW[N',K'] = dY^T . X             -> A = dY ([M',N'] = [K, M]), W = X ([M',K'] = [K, N]), trans = 3
   * bf16 only, M (N) % 8 == 0 for a transposed A (W), no LayerNorm fusion; one value per grouped launch.  K % 64 == 0
   * except with trans = 3 (any K >= 1: the rows past K of the last step are zero-filled on chip).
   * a_colsum (trans bit 0 only, optional): float [M], receives sum_k A[k, m] - the bias gradient db = column sums of dY
   * that belongs to dW = dY^T.X - written by plain stores (each m by exactly one workgroup: deterministic). */
  int32_t trans;
  float* a_colsum;
  /* Activation in a training step, without activation kernels (BertIntermediate, FT/vilmodel.py:190-199 and its autograd):
   *  - c_pre / ldcp (forward, with act != NONE): the pre-activation z = A W^T + bias is ALSO stored, in the input dtype, next
   *    to C = act(z) (N, ldcp multiples of 8; no LayerNorm fusion);
   *  - res_act (backward): DSRG_ACT_* > 0 makes the epilogue multiply by act'(residual[m,n]) instead of adding the residual:
   *    with residual = that stored z, the dgrad GEMM of the NEXT layer leaves as dZ = (dY W2) * act'(z). */
  void* c_pre;
  int32_t ldcp;
  int32_t res_act;
} dsrg_gemm_problem_t;
int dsrg_gemm_nt(const dsrg_gemm_problem_t* problems /*host*/, int nproblems, int dtype, int out_dtype,
                 int act, void* stream);
/* Which block tile (bm x bn), how many workgroups and which LDS ring depth (nst, may be NULL) dsrg_gemm_nt would use
 * for these problems (host-only query, no launch; used by bench.py to attribute measured durations to a kernel variant). */
int dsrg_gemm_nt_plan(const dsrg_gemm_problem_t* problems /*host*/, int nproblems, int dtype, int* bm, int* bn,
                      int* ntiles, int* nst);
/* dsrg_gemm_nt with the block tile (bm x bn) and LDS ring depth `nst` chosen by the caller instead of the library's
 * plan (bm = 0: the plan; nst = 0: the tile's default depth).  Same reference sites as dsrg_gemm_nt; exists so that a
 * host can tune / pin the tile per call without process-wide state (tools/bench_gemm.py --sweep).  Tiles built:
 * 64x64, 64x128, 128x128 (both dtypes); bf16 only: 64x192, 64x256, 128x64, 128x96, 128x192, 128x256.  DSRG_EINVAL for
 * a tile / depth that is not built. */
int dsrg_gemm_nt_tiled(const dsrg_gemm_problem_t* problems /*host*/, int nproblems, int dtype, int out_dtype,
                       int act, int bm, int bn, int nst, void* stream);

/* ---- fused short-sequence attention ------------------------------------------------------------
 * O[b,q,h,:] = softmax_k( scale·Q[b,q,h,:]·K[b,k,h,:] + mask_term(b,k) + sprel(b,q,k) ) · V[b,k,h,:]
 * Replaces BertSelfAttention / BertOutAttention (FT/vilmodel.py:125-164, 349-376) and the
 * nn.MultiheadAttention core of the pano encoder (FT/transformer.py:176-177).  Head dim is 64.
 * Q/K/V/O are addressed as base + b*?_bs + row*?_rs + h*64 (+d), strides in elements, so packed
 * QKV projections and hoisted text K/V panels are read in place.
 * kmask: uint8 [B,Nk] (torch.bool) or NULL.  mask_mode:
 *   DSRG_MASK_NONE; DSRG_MASK_NEG10000: + (1-m)·(-10000)  (FT/ops.py:24-33);
 *   DSRG_MASK_NEGINF: -inf where !m (key_padding_mask, FT/vilmodel.py:727);
 *   DSRG_MASK_PLUS1: + m·1.0 — the object self-encoder quirk, a bool mask ADDED to the scores
 *   (FT/vilmodel.py:711-712 -> :143).
 * pair_dists: fp32 [B,Nq,Nk] or NULL; sprel_w/sprel_b: device pointers to the two scalars of
 * sprel_linear; the bias w·d+b (FT/vilmodel.py:757-759, 462-463) is formed in the kernel.
 * lse: fp32 [B,NH,Nq] or NULL; when given, the log-sum-exp of each score row is stored for dsrg_attention_bwd.
 * p_drop / seed: dropout on the attention probabilities (training; FT/vilmodel.py:115,150,342, transformer.py:138): the
 * keep decision of element (b,h,q,k) is a counter-based hash of (seed, index), recomputed by dsrg_attention_bwd when
 * it is given the same p_drop and seed; 0 = off. */
enum { DSRG_MASK_NONE = 0, DSRG_MASK_NEG10000 = 1, DSRG_MASK_NEGINF = 2, DSRG_MASK_PLUS1 = 3 };
int dsrg_attention_fwd(const void* Q, int64_t q_bs, int q_rs, const void* K, int64_t k_bs, int k_rs,
                       const void* V, int64_t v_bs, int v_rs, void* O, int64_t o_bs, int o_rs,
                       const uint8_t* kmask, int mask_mode, const float* pair_dists,
                       const float* sprel_w, const float* sprel_b,
                       int B, int NH, int Nq, int Nk, float scale, float* lse, float p_drop, uint64_t seed, int dtype,
                       void* stream);

/* The same for up to DSRG_ATTN_MAX_GROUPS independent problems that share B, NH, mask mode, scale and dtype, in ONE
 * launch: the global and the local branch of an x-layer attend at the same point (FT/vilmodel.py:540-549). */
#define DSRG_ATTN_MAX_GROUPS 4
typedef struct {
  const void *Q, *K, *V;
  void* O;
  int64_t q_bs, k_bs, v_bs, o_bs;   /* batch strides (elements) */
  int32_t q_rs, k_rs, v_rs, o_rs;   /* row strides (elements) */
  const uint8_t* kmask;             /* [B,Nk] or NULL (mask_mode NONE) */
  const float* pair_dists;          /* [B,Nq,Nk] or NULL */
  float* lse;                       /* [B,NH,Nq] or NULL */
  int32_t Nq, Nk;
  /* Key validity from lengths instead of kmask (ABI v12; kmask must then be NULL): key k of episode b is valid iff
   * k % kseg < klens[b, k / kseg], klens int64 [B, ceil(Nk / kseg)].  kseg = Nk: one length per episode (gen_seq_masks,
   * FT/ops.py:35-43: the panorama's view_lens); kseg = O: one length per view for the V*O object tokens (obj_lens,
   * FT/vilmodel.py:711).  The reference builds these masks with an arange < len comparison per call. */
  const int64_t* klens;
  int32_t kseg;
  const int64_t* klens2;            /* optional, same shape: the limit is klens + klens2 (REVERIE: views + objects per episode) */
} dsrg_attn_problem_t;
int dsrg_attention_fwd_grouped(const dsrg_attn_problem_t* problems /*host*/, int nproblems, int mask_mode,
                               const float* sprel_w, const float* sprel_b, int B, int NH, float scale, float p_drop,
                               uint64_t seed /* problem i uses seed + i * 0x632BE59BD9B4E019 */, int dtype, void* stream);

/* ---- LayerNorm over the last dim (wave reduction, fp32 statistics) -------------------------------
 * out = LN(x [+ residual]) * gamma + beta.  Replaces BertLayerNorm / nn.LayerNorm
 * (FT/vilmodel.py:73,170,176,210,216; FT/transformer.py:144-145,174,179; FT/ops.py:19).
 * x is `in_dtype` (fp32 pre-LN sums written by dsrg_gemm_nt, or storage dtype), residual is
 * `out_dtype` or NULL, H <= 1024 and H % 64 == 0. */
int dsrg_layernorm_fwd(const void* x, int ldx, const void* residual, int ldr, const float* gamma,
                       const float* beta, void* out, int ldo, int M, int H, float eps, int in_dtype,
                       int out_dtype, void* stream);

/* The same over up to DSRG_MAX_GROUPS independent row sets in ONE launch (the global and local branch of a
 * CrossmodalEncoder layer normalise at the same point, FT/vilmodel.py:540-549; a launch costs more than the rows). */
typedef struct {
  const void* x;        /* [M,H] rows, leading dim ldx */
  const void* residual; /* NULL or [M,H], leading dim ldr */
  const float* gamma;
  const float* beta;
  void* out;            /* [M,H], leading dim ldo */
  int32_t M, ldx, ldr, ldo;
  /* Optional output placement (ABI v12; all 0 / NULL = plain), so that a serving loop needs no copy kernels between calls
   * (the reference's agent builds the next call's token buffers with torch.cat / slot assignments, FT/r2r/agent.py:231-237,
   * 185-188, 472; FT/vilmodel.py:773 concatenates the two [CLS] rows for sap_fuse_linear):
   *  - o_seg > 0: rows come in segments of o_seg rows; row r is stored at out + (r / o_seg) * o_seg_ld + (r % o_seg) * ldo
   *    (e.g. the 36 panorama tokens of episode b into slots 2.. of a [B, 38, H] viewpoint-token buffer);
   *  - aux_seg > 0: every aux_seg-th row (r % aux_seg == 0) is ALSO stored at aux0 + (r / aux_seg) * aux_ld0 and / or
   *    aux1 + (r / aux_seg) * aux_ld1 (NULL = skip): row 0 of every sequence into a [B, 2H] concat buffer, the RMF memory
   *    into the [MEM] slots of both token buffers. */
  int32_t o_seg, o_seg_ld;
  void* aux0; void* aux1;
  int32_t aux_seg, aux_ld0, aux_ld1, reserved_;
} dsrg_ln_problem_t;
int dsrg_layernorm_fwd_grouped(const dsrg_ln_problem_t* problems /*host*/, int nproblems, int H, float eps,
                               int in_dtype, int out_dtype, void* stream);

/* ---- token embedding: sum of parts + LayerNorm ---------------------------------------------------
 * out[r,:] = LNo( base[r,:] + LNf( Wf·feat[r,:] + bf ) + E1[idx1[r],:] + E2[idx2[r],:] + vec[:] )
 * Every term is optional (NULL).  One kernel covers:
 *   text embeddings (word + position + token-type -> LN), FT/vilmodel.py:76-91;
 *   ASV object tokens (bbox linear + name embedding + nav-type embedding -> LN(128)), :698-708;
 *   panorama tokens (img + LN(loc_linear) + nav_type + token_type[1] -> LN), :719-723;
 *   graph-map tokens (img + step embedding + LN(Linear(7))), :753-755;
 *   viewpoint tokens (img + LN(Linear(14))), :769.
 * feat is fp32 [M,F] (ldf), F <= 16; Wf is fp32 [H,F] (the nn.Linear layout, 16-byte aligned); E1/E2/vec/gammas/betas fp32;
 * base/out `dtype`.  lnf_* NULL => no LayerNorm on the feature term; lno_* NULL => no output LayerNorm.
 * Row r takes table index idx[(r / div) * stride], read as int32, int64 or as a float that carries a class id
 * (the reference keeps the object class in the last feature column, :701, and repeats the per-view nav type over
 * the view's objects, :703-705) - so no index tensor has to be converted or expanded before the call. */
enum { DSRG_IDX_I32 = 0, DSRG_IDX_I64 = 1, DSRG_IDX_F32 = 2 };
int dsrg_embed_ln_fwd(const void* base, int ldb, const float* feat, int ldf, int F, const float* Wf,
                      const float* bf, const float* lnf_gamma, const float* lnf_beta,
                      const float* E1, const void* idx1, int idx1_kind, int idx1_stride, int idx1_div,
                      const float* E2, const void* idx2, int idx2_kind, int idx2_stride, int idx2_div,
                      const float* vec, const float* lno_gamma, const float* lno_beta, float eps,
                      void* out, int ldo, int M, int H, int dtype, void* stream);

/* ---- mean over groups of G consecutive rows: out[m,:] = mean_g x[m*G+g,:] -------------------------
 * The per-view mean over ASV object tokens, padded slots included (FT/vilmodel.py:714). */
int dsrg_group_mean_fwd(const void* x, int ldx, void* out, int ldo, int M, int G, int H, int dtype,
                        void* stream);

/* dsrg_embed_ln_fwd with the base row GATHERED from two tables by per-episode lengths (REVERIE's per-episode token
 * assembly, FT/vilmodel_reverie.py:727-735: a Python loop of torch.cat / zero padding): rows come in episodes of seg_T
 * tokens; token t of episode b takes its base row from  base[b*seg_V + t]  if t < lens0[b],  from
 * base2[b*seg_No + (t - lens0[b])]  if t < lens0[b] + lens1[b],  and is zero otherwise.  lens0 / lens1 int64 [B];
 * base2 has leading dim ldb2.  seg_T = 0: plain dsrg_embed_ln_fwd. */
int dsrg_embed_ln_fwd_seg(const void* base, int ldb, const void* base2, int ldb2, int seg_T, int seg_V, int seg_No,
                          const int64_t* lens0, const int64_t* lens1,
                          const float* feat, int ldf, int F, const float* Wf, const float* bf,
                          const float* lnf_gamma, const float* lnf_beta,
                          const float* E1, const void* idx1, int idx1_kind, int idx1_stride, int idx1_div,
                          const float* E2, const void* idx2, int idx2_kind, int idx2_stride, int idx2_div,
                          const float* vec, const float* lno_gamma, const float* lno_beta, float eps,
                          void* out, int ldo, int M, int H, int dtype, void* stream);

/* dsrg_embed_ln_fwd with a LayerNorm on the BASE row before it is added (same eps):
 *   out[r,:] = LNo( LNb(base[r,:]) + LNf( Wf.feat[r,:] + bf ) + E1[..] + E2[..] + vec )
 * The panorama tokens of FT/vilmodel.py:693-695, 719-724 - img_layer_norm(img_linear(view features)) is the base of the token
 * sum - without a LayerNorm launch between the projection and the token kernel: base is the projection's output as the
 * GEMM stored it (storage dtype).  lnb_gamma / lnb_beta fp32 [H], 16-byte aligned, both or neither. */
int dsrg_embed_ln_fwd_lnb(const void* base, int ldb, const float* lnb_gamma, const float* lnb_beta,
                          const float* feat, int ldf, int F, const float* Wf, const float* bf,
                          const float* lnf_gamma, const float* lnf_beta,
                          const float* E1, const void* idx1, int idx1_kind, int idx1_stride, int idx1_div,
                          const float* E2, const void* idx2, int idx2_kind, int idx2_stride, int idx2_div,
                          const float* vec, const float* lno_gamma, const float* lno_beta, float eps,
                          void* out, int ldo, int M, int H, int dtype, void* stream);

/* ---- GAA: global adaptive aggregation of the panorama ---------------------------------------------
 * a[b,v] = softmax_v( tanh( x[b,v,:]·w + bias ) ) over ALL V slots (unmasked);
 * out[b,:] = sum_v a[b,v]·x[b,v,:].  FT/vilmodel.py:735-740 (REVERIE vilmodel_reverie.py:753-758).
 * x is [B,V,H] contiguous in `dtype`, w fp32 [H], bias fp32 [1], out [B,H] `dtype`; V <= 64. */
int dsrg_gaa_pool_fwd(const void* x, const float* w, const float* bias, void* out, int B, int V,
                      int H, int dtype, void* stream);
/* The same with placement for a serving loop (no copy kernels between the panorama and the navigation call): x is read with
 * a batch stride x_bs (elements; the panorama tokens as they lie in slots 2.. of the viewpoint-token buffer), and the pooled
 * row is ALSO stored at out2 + b*out2_bs + idx2[b]*H (NULL = skip): the current node's slot of the graph-map token buffer
 * (FT/r2r/agent.py:443-447 writes the GAA-pooled panorama into the graph map).  idx2 int64 [B] with 0 <= idx2[b] < idx2_max
 * (the caller's guarantee; out2_bs >= idx2_max*H is checked). */
int dsrg_gaa_pool_fwd_ex(const void* x, int64_t x_bs, const float* w, const float* bias, void* out, void* out2,
                         int64_t out2_bs, const int64_t* idx2, int idx2_max, int B, int V, int H, int dtype,
                         void* stream);
/* dsrg_gaa_pool_fwd_ex on rows that are still UN-normalised: x [B,V,H] (batch stride x_bs) is the output of vis_obj_linear's
 * Linear as the GEMM stored it; the kernel applies that Sequential's LayerNorm (FT/vilmodel.py:731-732; gamma / beta fp32 [H])
 * to every row, stores the normalised rows - the call's pano_embeds - to rows_out (batch stride rows_bs, storage dtype) and
 * pools the stored values: the LayerNorm launch between the projection and the pooling is gone.  H % 64 == 0. */
int dsrg_gaa_pool_fwd_ln(const void* x, int64_t x_bs, const float* ln_gamma, const float* ln_beta, float ln_eps, void* rows_out,
                         int64_t rows_bs, const float* w, const float* bias, void* out, void* out2, int64_t out2_bs,
                         const int64_t* idx2, int idx2_max, int B, int V, int H, int dtype, void* stream);

/* ---- ClsPrediction tail: score[r] = LN(h[r,:])·w2 + b2 ---------------------------------------------
 * h = relu(Linear(x)) comes from dsrg_gemm_nt (fp32, DSRG_ACT_RELU).  FT/vilmodel.py:616-628
 * (net.2 LayerNorm eps 1e-12, net.3 Linear(H,1)).  h fp32 [M,H] (ldh); score fp32 [M]. */
int dsrg_head_tail_fwd(const float* h, int ldh, const float* gamma, const float* beta, const float* w2,
                       const float* b2, float eps, float* score, int M, int H, void* stream);

/* The same for up to DSRG_MAX_GROUPS heads in ONE launch (global, local, fuse and object heads of a navigation step,
 * vilmodel.py:773-785): per problem its own rows, LayerNorm affine, projection and output. */
typedef struct {
  const float* h; const float* gamma; const float* beta; const float* w2; const float* b2; float* score;
  int32_t M, ldh;
} dsrg_head_tail_problem_t;
int dsrg_head_tail_fwd_grouped(const dsrg_head_tail_problem_t* problems, int nproblems, int H, float eps, void* stream);

/* ---- navigation logits: dynamic fusion + masking + global/local logit fusion ------------------------
 * fw = sigmoid(fuse_raw[b]);  gl = g_raw*fw, ll = l_raw*(1-fw);  -inf masking
 * (FT/vilmodel.py:773-785), then fused = gl; fused[:,0] += ll[:,0]; for each real, unvisited graph
 * node j: += ll[cand slot of j] if j is a current candidate else += sum of ll over candidates that are
 * visited nodes (FT/vilmodel.py:788-805; pre-training variant PT/pretrain_cmt.py:269-285).
 * The viewpoint-id string matching of the reference is done once on the host and arrives as
 *   gmap_src int32 [B,G]: k >= 0 -> add ll[b,k];  -2 -> add the back-track sum;  -1 -> add nothing;
 *   bw_mask uint8 [B,Vp]: 1 where local token k is a candidate that is an already-visited node.
 * fuse_raw may be NULL (fw = 0.5).  Masks are torch.bool (uint8).  All logits fp32. */
int dsrg_nav_logits_fwd(const float* g_raw, const float* l_raw, const float* fuse_raw,
                        const uint8_t* gmap_masks, const uint8_t* gmap_visited, const uint8_t* vp_nav_masks,
                        const int32_t* gmap_src, const uint8_t* bw_mask, float* global_logits,
                        float* local_logits, float* fused_logits, int B, int G, int Vp, void* stream);
/* The same plus the object-grounding logits of REVERIE in the same launch (FT/vilmodel_reverie.py:823-825:
 * obj_logits = og_head(vp_embeds).masked_fill(~vp_obj_masks, -inf)): o_raw fp32 [B,Vp], vp_obj_masks bool [B,Vp],
 * obj_logits fp32 [B,Vp]; all three NULL = dsrg_nav_logits_fwd. */
int dsrg_nav_logits_fwd_og(const float* g_raw, const float* l_raw, const float* fuse_raw,
                           const uint8_t* gmap_masks, const uint8_t* gmap_visited, const uint8_t* vp_nav_masks,
                           const int32_t* gmap_src, const uint8_t* bw_mask, float* global_logits,
                           float* local_logits, float* fused_logits, const float* o_raw, const uint8_t* vp_obj_masks,
                           float* obj_logits, int B, int G, int Vp, void* stream);

/* ---- action selection + loss tail of a rollout step -------------------------------
 * One pass over the nav logits [B,N] (fp32, -inf = masked) replaces FT/r2r/agent.py:484 (softmax), :491 (stop
 * probability = probs[:,0]), :502 (CrossEntropyLoss(ignore_index, reduction='sum'), agent_base.py:150), :508 (argmax),
 * :511-514 (Categorical entropy / sample).  sample[b] = first j with cdf_b(j) > u[b] (the caller draws u ~ U[0,1), so
 * the result is a deterministic function of u); loss_rows[b] = lse_b - logits[b,target_b] (0 if ignored), their sum is
 * the reference's loss; dlogits = softmax - onehot(target) is its gradient.  Optional outputs may be NULL. */
int dsrg_action_tail_fwd(const float* logits, int ld, const int64_t* targets, int64_t ignore_index, const float* u,
                         float* stop_prob, int32_t* argmax, int32_t* sample, float* entropy, float* loss_rows,
                         float* dlogits, int B, int N, void* stream);

/* Cross-entropy rows = F.cross_entropy(logits, targets, ignore_index, reduction='none') of the pre-training losses
 * (PT/model/pretrain_cmt.py:160-167 MLM over the vocabulary, :286-294 the three SAP terms, :325 OG).  logits fp32 [B,N]
 * (row stride ld, -inf = masked); fwd: lse_rows[b] = logsumexp, loss_rows[b] = lse - logits[b,target] (0 when the target is
 * ignore_index or out of range), one pass over the row; bwd: dlogits[b,j] = grad_rows[b] * (softmax - onehot) recomputed
 * from logits and lse_rows (0 row when ignored), written in out_dtype with row stride ldd (columns N..ldd-1 zeroed). */
int dsrg_ce_rows_fwd(const float* logits, int ld, const int64_t* targets, int64_t ignore_index, float* lse_rows,
                     float* loss_rows, int B, int N, void* stream);
int dsrg_ce_rows_bwd(const float* logits, int ld, const int64_t* targets, int64_t ignore_index, const float* lse_rows,
                     const float* grad_rows, void* dlogits, int ldd, int out_dtype, int B, int N, void* stream);

/* ---- device-resident graph map + navigation-input assembly ------------------------
 * Replaces GraphMap / FloydGraph (FT/models/graph_utils.py:42-148), the per-step graph / node-embedding updates of the
 * rollout (FT/r2r/agent.py:406-408, 434-451, 565-569) and the builders _nav_gmap_variable / _nav_vp_variable (:157-278).
 * The host keeps only the viewpoint-id -> slot dictionaries (slot = insertion order into node_positions; slot 0 is the
 * start viewpoint) and hands over, per step, slots + positions of the current node and its candidates.  Distances are
 * double, compared exactly as the reference's Python floats.  N <= 128 nodes per episode. */
typedef struct {
  double* pos;        /* [B,N,3] */
  double* dist;       /* [B,N,N] 95959595 = unknown */
  int32_t* point;     /* [B,N,N] via-node, -1 = direct / unknown */
  uint8_t* visited;   /* [B,N] */
  int32_t* step_id;   /* [B,N] */
  float* emb_sum;     /* [B,N,H] */
  int32_t* emb_cnt;   /* [B,N] */
  int32_t* n_nodes;   /* [B] */
  int32_t N, H;
} dsrg_graph_state_t;
typedef struct {
  void* gmap_img;          /* [B,G,H] storage dtype: zeros, [MEM], mean node embeddings (visited, then unvisited), zero pad */
  int64_t* gmap_step_ids;  /* [B,G] */
  float* gmap_pos_fts;     /* [B,G,7] */
  float* gmap_pair_dists;  /* [B,G,G] */
  uint8_t* gmap_masks;     /* [B,G] column 1 ([MEM]) false, FT/r2r/agent.py:211 */
  uint8_t* gmap_visited;   /* [B,G] */
  int32_t* gmap_slots;     /* [B,G] node slot of each token, -1 for [stop] / [MEM] / padding */
  int32_t* gmap_src;       /* [B,G] fusion indices for dsrg_nav_logits_fwd */
  uint8_t* bw_mask;        /* [B,Vp] */
  float* vp_pos_fts;       /* [B,Vp,14] */
  uint8_t* no_vp_left;     /* [B] */
  int32_t* gmap_len;       /* [B] */
  int32_t G, Vp;           /* padded widths chosen by the caller (G >= max gmap_len) */
} dsrg_graph_nav_out_t;
int dsrg_graph_reset(double* pos, double* dist, int32_t* point, uint8_t* visited, int32_t* step_id, float* emb_sum,
                     int32_t* emb_cnt, int32_t* n_nodes, int B, int N, int H, void* stream);
/* update_graph(ob) for episodes with graph_active, then node_step_ids[cur] = step_value for those with step_active.
 * cand_slot [B,Cmax], cur_pos [B,3], cand_pos [B,Cmax,3] (double); n_nodes_new = node count after this observation. */
int dsrg_graph_observe(const dsrg_graph_state_t* st, const uint8_t* graph_active, const uint8_t* step_active,
                       const int32_t* cur_slot, const int32_t* n_nodes_new, const int32_t* ncand, const int32_t* cand_slot,
                       const double* cur_pos, const double* cand_pos, int Cmax, int step_value, int B, void* stream);
/* node embeddings: current node <- fused (rewrite); every unvisited candidate j accumulates pano[b, j, :]. */
int dsrg_graph_update_embeds(const dsrg_graph_state_t* st, const uint8_t* active, const int32_t* cur_slot, const int32_t* ncand,
                             const int32_t* cand_slot, int Cmax, const void* pano, int64_t pano_bs, int pano_rs,
                             const void* fused, int fused_rs, int B, int dtype, void* stream);
/* all graph-side inputs of forward(mode='navigation') in one launch; mem = last cls_embeds [B,H] or NULL (step 0). */
int dsrg_graph_nav_inputs(const dsrg_graph_state_t* st, const dsrg_graph_nav_out_t* out, const int32_t* cur_slot,
                          const double* heading, const double* elevation, const int32_t* ncand, const int32_t* cand_slot,
                          int Cmax, const void* mem, int mem_rs, int B, int dtype, void* stream);

/* ---- IGL gate: w = sigmoid(aug·wa + ba + x·wo + bo); out = w·aug + (1-w)·x --------------------------
 * FT/vilmodel.py:525-528.  aug/x/out [M,H] `dtype`; wa/wo fp32 [H]; ba/bo fp32 [1]. */
int dsrg_igl_blend_fwd(const void* aug, const void* x, const float* wa, const float* ba, const float* wo,
                       const float* bo, void* out, int M, int H, int dtype, void* stream);

/* ======================= backward pass =======================================================================
 * The reference has no explicit backward code: these replace what torch.autograd derives for the ops above
 * (loss.backward() at finetune_src/r2r/agent_base.py:205, pretrain_src/train_r2r.py:251-261).
 * GEMM gradients reuse dsrg_gemm_nt:  dX = gemm_nt(dY, Wt),  dW = gemm_nt(dYt, Xt)  with the transposes below. */

/* out[c, r] = in[r, c] for r < R (zero for R <= r < Rp), c < C.  in is [R,C] (ld_in) of in_dtype, out is [C,Rp] (ld_out)
 * of out_dtype.  Rp is the row count padded to the GEMM's K granularity (64).
 * colsum: NULL, or fp32 [C] accumulated with float atomics (zeroed by the caller): colsum[c] += sum_r in[r, c] - the
 * bias gradient, taken while dY is being transposed for the weight gradient. */
int dsrg_transpose_pad(const void* in, int ld_in, void* out, int ld_out, int R, int C, int Rp, int in_dtype,
                       int out_dtype, float* colsum, void* stream);

/* The same for up to DSRG_MAX_TRANSPOSES matrices in ONE launch: the backward of one (grouped) linear layer needs W^T,
 * dY^T and X^T per problem - autograd of the nn.Linear sites listed at dsrg_gemm_nt (loss.backward(),
 * PT/train_r2r.py:253, FT/r2r/agent.py:552). */
#define DSRG_MAX_TRANSPOSES 24
typedef struct {
  const void* in; void* out; float* colsum;   /* colsum may be NULL */
  int32_t ld_in, ld_out, R, C, Rp, in_dtype, out_dtype, pad_;
} dsrg_transpose_problem_t;
int dsrg_transpose_pad_grouped(const dsrg_transpose_problem_t* problems, int nproblems, void* stream);

/* out[n] += sum_m in[m, n]  (bias gradient; out is fp32 and must be zeroed by the caller; float atomics). */
int dsrg_colsum_acc(const void* in, int ld, float* out, int M, int N, int dtype, void* stream);

/* y = act(z) and dz = dy * act'(z) element-wise (act = DSRG_ACT_*; erf-GELU of FT/vilmodel.py:30-31, ReLU of
 * ClsPrediction :623, tanh of BertPooler :279).  n elements, n % 4 == 0. */
int dsrg_act_fwd(const void* z, void* y, int64_t n, int act, int dtype, void* stream);
int dsrg_act_bwd(const void* z, const void* dy, void* dz, int64_t n, int act, int dtype, void* stream);

/* Attention backward: dQ, dK, dV (contiguous [B,N,NH*64], storage dtype) and, when pair_dists is given, the two
 * gradients of sprel_linear accumulated with float atomics into dsprel[2] (zeroed by the caller).  Scores are
 * recomputed from Q, K, V and the forward's lse; delta_ws is an fp32 [B,NH,Nq] workspace that receives delta = dO . O per
 * query row (the fp32 path's dK / dV launch reads it back; the bf16 matrix-core path is ONE launch whose dK / dV workgroups
 * form their own delta from the dO and O rows they stage, and only stores it).  O / dO are the forward output and its
 * gradient (strides like the forward). */
int dsrg_attention_bwd(const void* Q, int64_t q_bs, int q_rs, const void* K, int64_t k_bs, int k_rs,
                       const void* V, int64_t v_bs, int v_rs, const void* O, int64_t o_bs, int o_rs,
                       const void* dO, int64_t do_bs, int do_rs, const float* lse,
                       const uint8_t* kmask, int mask_mode, const float* pair_dists,
                       const float* sprel_w, const float* sprel_b,
                       void* dQ, void* dK, void* dV, float* delta_ws, float* dsprel,
                       int B, int NH, int Nq, int Nk, float scale, float p_drop, uint64_t seed, int dtype, void* stream);
/* The same with batch / row strides (elements) on the three gradients, so that they can be column blocks of ONE packed
 * buffer - d(qkv) [B,N,3*NH*64] of a BertSelfAttention whose query / key / value projections are one GEMM
 * (FT/vilmodel.py:111-113), d(kv) [B,Nk,2*NH*64] of a cross attention - and torch.autograd needs no zero-fill, copy and
 * add per slice to put them together.  Rows >= NH*64 elements, strides multiples of 4 elements. */
int dsrg_attention_bwd_strided(const void* Q, int64_t q_bs, int q_rs, const void* K, int64_t k_bs, int k_rs,
                               const void* V, int64_t v_bs, int v_rs, const void* O, int64_t o_bs, int o_rs,
                               const void* dO, int64_t do_bs, int do_rs, const float* lse,
                               const uint8_t* kmask, int mask_mode, const float* pair_dists,
                               const float* sprel_w, const float* sprel_b,
                               void* dQ, int64_t dq_bs, int dq_rs, void* dK, int64_t dk_bs, int dk_rs, void* dV, int64_t dv_bs, int dv_rs,
                               float* delta_ws, float* dsprel,
                               int B, int NH, int Nq, int Nk, float scale, float p_drop, uint64_t seed, int dtype, void* stream);

/* dsrg_attention_bwd without the workspace argument, for callers that cannot allocate: code under HIP-graph capture,
 * tight serving / fine-tuning loops, harnesses that forbid hipMalloc.  NO counterpart in the reference (its autograd
 * allocates freely); everything else is dsrg_attention_bwd (same kernels, same results bit for bit).  The delta
 * workspace is a static device array of DSRG_ATTN_BWD_NOALLOC_MAX_ROWS floats (4 MiB, held on every GPU the library
 * runs on until the process ends): B*NH*Nq above that is rejected with DSRG_EINVAL.  The entry point calls no
 * hipMalloc / hipFree / allocating routine.  Every call on a GPU shares that GPU's array: two calls that overlap on one
 * GPU (different streams, or different threads) RACE - serialise them on one stream.  Nothing needs clearing between
 * calls (each call writes every row it later reads). */
#define DSRG_ATTN_BWD_NOALLOC_MAX_ROWS (1 << 20)
int dsrg_attention_bwd_noalloc(const void* Q, int64_t q_bs, int q_rs, const void* K, int64_t k_bs, int k_rs,
                               const void* V, int64_t v_bs, int v_rs, const void* O, int64_t o_bs, int o_rs,
                               const void* dO, int64_t do_bs, int do_rs, const float* lse,
                               const uint8_t* kmask, int mask_mode, const float* pair_dists,
                               const float* sprel_w, const float* sprel_b,
                               void* dQ, void* dK, void* dV, float* dsprel,
                               int B, int NH, int Nq, int Nk, float scale, float p_drop, uint64_t seed, int dtype, void* stream);

/* Backward of the row kernels (each recomputes its forward values; parameter gradients are accumulated with float
 * atomics into fp32 buffers the caller has zeroed; activation gradients are written in the storage dtype). */
int dsrg_group_mean_bwd(const void* dy, int lddy, void* dx, int lddx, int M, int G, int H, int dtype, void* stream);
/* GAA pooling: dx [B,V,H], dw [H], db [1] from dout [B,H] (all of x / dout / dx contiguous). */
int dsrg_gaa_pool_bwd(const void* x, const float* w, const float* bias, const void* dout, void* dx, float* dw, float* db,
                      int B, int V, int H, int dtype, void* stream);
/* ClsPrediction tail: dh [M,H] fp32, dgamma / dbeta / dw2 [H], db2 [1] from dscore [M]. */
int dsrg_head_tail_bwd(const float* h, int ldh, const float* gamma, const float* beta, const float* w2, const float* dscore,
                       float eps, float* dh, int lddh, float* dgamma, float* dbeta, float* dw2, float* db2, int M, int H,
                       void* stream);
/* Dynamic fusion + logit fusion: gradients of the three raw score inputs from the gradients of the three logit outputs
 * (any of dglobal / dlocal / dfused may be NULL = zero); masked (-inf) positions pass no gradient, like masked_fill_. */
int dsrg_nav_logits_bwd(const float* g_raw, const float* l_raw, const float* fuse_raw, const uint8_t* gmap_masks,
                        const uint8_t* gmap_visited, const uint8_t* vp_nav_masks, const int32_t* gmap_src,
                        const uint8_t* bw_mask, const float* dglobal, const float* dlocal, const float* dfused,
                        float* dg_raw, float* dl_raw, float* dfuse_raw, int B, int G, int Vp, void* stream);
/* IGL gate: daug / dx [M,H], dwa / dwo [H], dbias[2] = (dba, dbo). */
int dsrg_igl_blend_bwd(const void* aug, const void* x, const float* wa, const float* ba, const float* wo, const float* bo,
                       const void* dout, void* daug, void* dx, float* dwa, float* dwo, float* dbias, int M, int H, int dtype,
                       void* stream);

/* Token-embedding sums backward (forward operands as dsrg_embed_ln_fwd).  Gradient pointers may be NULL where the
 * forward term is absent; all of them are fp32 buffers the caller has zeroed (accumulated with float atomics), except
 * dbase ([M,H], storage dtype, written).  dE1 / dE2 are dense tables like the forward ones. */
typedef struct {
  const void* base; const float* feat; const float* Wf; const float* bf; const float* lnf_gamma; const float* lnf_beta;
  const float* E1; const void* idx1; const float* E2; const void* idx2; const float* vec;
  const float* lno_gamma; const float* lno_beta; const void* dout;
  void* dbase; float* dWf; float* dbf; float* dlnf_gamma; float* dlnf_beta; float* dE1; float* dE2; float* dvec;
  float* dlno_gamma; float* dlno_beta;
  int32_t ldb, ldf, F, idx1_kind, idx1_stride, idx1_div, idx2_kind, idx2_stride, idx2_div, lddo, lddb, M, H, dtype;
  float eps;
  int32_t E1_rows, E2_rows;   /* rows of the tables (tables of <= 4 rows are summed in registers first) */
  int32_t E1_padding_idx;     /* row of E1 that receives no gradient, as nn.Embedding(padding_idx=0) of the reference's
                               * word embeddings (finetune_src/models/vilmodel.py:65); -1 = none */
  int32_t E1_copies;          /* > 1: dE1 is [E1_copies][E1_rows][H] and workgroup w adds into copy w % E1_copies; the
                               * caller sums the copies.  Spreads the same-address atomics of a table with a few dozen
                               * rows (the 45 object classes take 14796 rows of a pre-training batch).  0 / 1 = one copy */
  int32_t reserved_;
  float* partials;            /* optional fp32 workspace of dsrg_embed_ln_bwd_workspace(args) elements (uninitialised): every
                               * workgroup stores its column-vector sums there with plain stores and a second launch adds the
                               * table, in a fixed order, onto the gradient buffers - instead of one float atomic per element
                               * per workgroup onto the same few cache lines (0.4 ms of a pre-training step).  NULL = atomics */
} dsrg_embed_bwd_args_t;
int64_t dsrg_embed_ln_bwd_workspace(const dsrg_embed_bwd_args_t* args /*host*/);
int dsrg_embed_ln_bwd(const dsrg_embed_bwd_args_t* args /*host*/, void* stream);

/* Segment-weighted row gather: out[r,:] = sum_{e in [seg_ptr[r], seg_ptr[r+1])} w[e] * in[idx[e],:].  Replaces the
 * pre-training graph-map aggregation (dict-of-tensors pooling with Python loops, PT/model/vilmodel.py:597-635: visited
 * node = the GAA-pooled panorama of its step, unvisited node = mean of the candidate views that saw it) and the
 * viewpoint-token assembly (:557-571).  Inputs are two row tables addressed as one (idx < n0 -> in0, else in1), outputs
 * likewise (r < m0 -> out0, else out1); rows of an empty segment are zero.  The backward is the same call on the
 * transposed incidence (host-built), so no atomics.  seg_ptr has m0 + m1 + 1 entries. */
int dsrg_segment_wsum(const void* in0, int ld0, int n0, const void* in1, int ld1, int n1, void* out0, int ldo0, int m0,
                      void* out1, int ldo1, int m1, const int32_t* seg_ptr, const int32_t* idx, const float* w, int H,
                      int dtype, void* stream);

/* LayerNorm backward for out = LN(x [+ residual]) * gamma + beta:
 *   dz (= dx = dresidual, stored as `dtype`); dgamma / dbeta fp32 [H] are WRITTEN (ABI 13; they were accumulated with float
 *   atomics before, and those atomics - 2H addresses shared by every workgroup - were the kernel's whole cost): each workgroup
 *   stores its column sums into `partials` (uninitialised fp32 workspace of dsrg_layernorm_bwd_blocks(M) * 2 * H elements) and
 *   a second launch sums the table in a fixed order, so the result is also run-to-run deterministic.
 * x is `in_dtype` (fp32 pre-LN sums or `dtype`), residual / dy / dz are `dtype`. */
int dsrg_layernorm_bwd_blocks(int M);
int dsrg_layernorm_bwd(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const void* dy,
                       int lddy, void* dz, int lddz, float* dgamma, float* dbeta, float* partials, int M, int H, float eps,
                       int in_dtype, int dtype, void* stream);
/* Hidden-state dropout fused into the residual LayerNorm (training; BertSelfOutput / BertOutput, FT/vilmodel.py:173-177,
 * 213-217: LayerNorm(dropout(dense(h)) + input)).  The keep decision of element (problem tag, row, col) is a counter-based
 * hash of (seed, (tag << 40) + row * H + col) - tag = the problem's index in the grouped forward call - so no mask tensor is
 * stored and the backward recomputes it from the SAME x (the un-dropped dense output):
 *   fwd: out = LN(x * keep / (1 - p) + residual);   bwd: dz = dLN (the residual's gradient, may be NULL), dzd = dz * keep / (1 - p)
 *   (the dense output's gradient).  p_drop = 0 is dsrg_layernorm_fwd_grouped / dsrg_layernorm_bwd. */
int dsrg_layernorm_fwd_grouped_drop(const dsrg_ln_problem_t* problems, int nproblems, int H, float eps, int in_dtype,
                                    int out_dtype, float p_drop, uint64_t seed, void* stream);
int dsrg_layernorm_bwd_drop(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const void* dy, int lddy,
                            void* dz, int lddz, void* dzd, int lddzd, float* dgamma, float* dbeta, float* partials, int M, int H,
                            float eps, int in_dtype, int dtype, float p_drop, uint64_t seed, int tag, void* stream);

/* The MRC loss tail, PT/model/pretrain_cmt.py:215-226: F.kl_div(F.log_softmax(logits, -1), target, reduction='none').sum(1)
 * over fp32 [R, N] rows (N = 1000 view / 1601 object classes): loss_r = sum_c [t > 0] t (log t - x) + lse_r sum_c t; backward
 * dlogits[r, c] = grad_r (exp(x - lse_r) sum_c t - t), recomputed from the logits and lse_rows (no log-softmax matrix). */
int dsrg_kl_rows_fwd(const float* logits, int ld, const float* target, int ldt, float* lse_rows, float* loss_rows, int R, int N,
                     void* stream);
int dsrg_kl_rows_bwd(const float* logits, int ld, const float* target, int ldt, const float* lse_rows, const float* grad_rows,
                     float* dlogits, int ldd, int R, int N, void* stream);

/* out[i] = sum_{s < S} in[s * slab_stride + i], i < n, summed in slab order; out is fp32 or bf16.  The fix-up of a K-split
 * weight gradient: dW = dY^T.X of a small weight over many token rows (the object-token encoder: 128 x 128 ... 3072 x 128
 * outputs over 14,796 rows) runs as S problems of ONE grouped dsrg_gemm_nt launch (trans = 3, fp32 outputs), each over a
 * slice of the rows; no counterpart in the reference (autograd's `grad_output.t().mm(input)`). */
int dsrg_sum_slabs(const float* in, int S, int64_t slab_stride, void* out, int64_t n, int out_dtype, void* stream);

/* out = (residual ? residual : 0) + x * keep / (1 - p_drop) over n contiguous elements (n % 4 == 0): nn.Dropout on a hidden
 * state with the residual add behind it - the embedding dropout (FT/vilmodel.py:89-90) and the pre-norm panorama encoder's
 * `src + dropout(...)` (models/transformer.py:153-160).  Keep decision of element i = the library's counter hash of (seed, i)
 * (+ the dropout offset word), so the backward is the same call on dy with residual = NULL and no mask tensor exists. */
int dsrg_dropout_add(const void* x, const void* residual, void* out, int64_t n, int dtype, float p_drop, uint64_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSRG_HIP_H */
