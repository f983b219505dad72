"""Raw launches of the libdsrg_hip.so kernels on torch tensors (no autograd; see functional.py).

Tensors must live on a ROCm device; the work is enqueued on torch's current stream.  Outputs are
allocated with torch.empty (caching allocator: capturable in a HIP graph)."""
import ctypes as C
import torch

from . import _lib
from ._lib import F32, BF16

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _dt(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise _lib.DsrgHipError('unsupported dtype %s (float32 / bfloat16 only)' % (t,))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """the current HIP stream of the current device as a void* (torch.cuda.current_stream() costs ~8 us of Python per
    call; the raw getter is one C call)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else t.data_ptr()


_ZERO_POOL = {}
_ZERO_POOL_ON = __import__('os').environ.get('DSRG_ZERO_POOL', '1') == '1'
_ZERO_CHUNK = 1 << 20          # fp32 elements per chunk (4 MiB)


def zeros_f32(shape, device):
    """Small zero-initialised fp32 accumulator (column sums, LayerNorm / bias gradients, sprel gradients ...).  A training
    step needs ~300 of them; each torch.zeros is a fill launch plus ~8 us of host time, so they are carved out of 4 MiB
    chunks that ONE fill zeroes (a chunk is handed out once and never reused: slices may live on as .grad).  For large
    requests this is torch.zeros; under HIP-graph capture too, unless the capture runs inside zero_pool_capture_scope()."""
    n = 1
    for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
        n *= int(d)
    capturing = torch.cuda.is_current_stream_capturing()
    if n > 65536 or not _ZERO_POOL_ON or (capturing and _CAPTURE_SCOPE[0] is None):
        return torch.zeros(shape, device=device, dtype=torch.float32)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), _raw_stream(torch.cuda.current_device()) if _raw_stream else 0)
    if capturing:
        key = key + (_CAPTURE_SCOPE[0],)
    st = _ZERO_POOL.get(key)
    na = (n + 63) // 64 * 64
    if st is None or st[1] + na > _ZERO_CHUNK:
        st = _ZERO_POOL[key] = [torch.zeros(_ZERO_CHUNK, device=device, dtype=torch.float32), 0]
    out = st[0][st[1]:st[1] + n].view(shape)
    st[1] += na
    return out


_CAPTURE_SCOPE = [None]


class zero_pool_capture_scope:
    """`with zero_pool_capture_scope():` around ONE stream capture: zeros_f32 then carves its accumulators out of chunks that
    are allocated and filled INSIDE the capture (graph-pool memory, one fill node per chunk, replayed before any of the
    chunk's users), and that no other capture and no eager code will ever be handed."""
    _n = 0

    def __enter__(self):
        zero_pool_capture_scope._n += 1
        self.prev, _CAPTURE_SCOPE[0] = _CAPTURE_SCOPE[0], ('capture', zero_pool_capture_scope._n)
        return self

    def __exit__(self, *exc):
        for k in [k for k in _ZERO_POOL if k[-1] == _CAPTURE_SCOPE[0]]:
            del _ZERO_POOL[k]
        _CAPTURE_SCOPE[0] = self.prev
        return False


def rows_ok(t):
    """t as-is if its leading dims collapse onto one row stride (what every row kernel needs), else a contiguous copy:
    incoming gradients can be arbitrary views (e.g. the slice of a longer token sequence)."""
    if t.stride(-1) != 1:
        return t.contiguous()
    try:
        _rows(t)
        return t
    except _lib.DsrgHipError:
        return t.contiguous()


def _dev(t):
    if not t.is_cuda:
        raise _lib.DsrgHipError('dsrg_amd kernels need device tensors (got %s); there is no CPU path' % t.device)
    return t


def _rows(t):
    """[..., H] tensor -> (M, ld) treating it as rows; requires unit stride on the last dim and
    uniform row stride."""
    nd = t.dim()
    if nd == 2 and t.stride(1) == 1:             # the common case on the training path (one call per operand per launch)
        return t.shape[0], t.stride(0)
    if nd == 1:
        return 1, t.shape[0]
    if t.is_contiguous():
        H = t.shape[-1]
        return (t.numel() // H if H else 0), H
    if t.stride(-1) != 1:
        raise _lib.DsrgHipError('last dim must be contiguous')
    if t.dim() == 2:
        return t.shape[0], t.stride(0)
    ld = t.stride(-2)
    M = 1
    for d in range(t.dim() - 1):
        M *= t.shape[d]
    exp = ld
    for d in range(t.dim() - 2, -1, -1):       # check the leading dims collapse onto one stride
        if t.shape[d] != 1 and t.stride(d) != exp:
            raise _lib.DsrgHipError('tensor rows are not uniformly strided: %s %s' % (tuple(t.shape), t.stride()))
        exp *= t.shape[d]
    return M, ld


def gemm(problems, act=_lib.ACT_NONE, out_f32=False, outs=None, tile=None, fuse=None, trans=0, colsums=None, want_pre=False, res_acts=None):
    """problems: list of (x [..,K], w [N,K], bias fp32 [N] | None, residual [..,N] | None).
    Returns list of outputs [.., N] (dtype = x dtype, or fp32 if out_f32).  tile=(bm, bn, nst) pins the block tile
    (dsrg_gemm_nt_tiled; tuning only).
    fuse: optional list (one entry per problem, None = plain) of dicts for the LayerNorm-fusion epilogue of
    dsrg_gemm_problem_t: 'a_stats': (partials [K/64, M, 2] fp32, c1 [N] fp32) - x holds an un-normalised row and w / bias
    are the gamma-folded weight and beta-folded bias; 'r_stats': (partials [N/64, M, 2], gamma [N], beta [N]) - the residual
    is un-normalised; 'stats_out': fp32 [N/64, M, 2] to receive this launch's partials; 'eps'.  (Block-major tables: one
    64-column block of all rows is contiguous.)
    trans (dsrg_gemm_problem_t::trans, bf16 only): bit 0 - x is stored [K, M]; bit 1 - w is stored [K, N] (2-D tensors, row
    stride free): the backward GEMMs of a linear layer on dY, W and X as they lie.  colsums (trans bit 0): optional list of
    fp32 [M] tensors (or None) that receive the column sums of the stored x (= the bias gradient with dW = dY^T.X).
    want_pre: also return the pre-activations z = x w^T + b (input dtype) -> (outputs, pres) (dsrg_gemm_problem_t::c_pre).
    res_acts: per problem, an activation id > 0 makes the epilogue MULTIPLY by act'(residual) instead of adding the residual."""
    lib = _lib.load()
    n = len(problems)
    arr = (_lib.GemmProblem * n)()
    res, pres = [], []
    dt = _dt(problems[0][0])
    for i, (x, w, b, r) in enumerate(problems):
        _dev(x)
        if _dt(x) != dt or w.dtype != x.dtype:
            raise _lib.DsrgHipError('gemm: mixed dtypes in one grouped launch')
        if trans:
            if x.dim() != 2 or w.dim() != 2 or x.stride(1) != 1:
                raise _lib.DsrgHipError('gemm: transposed operands must be 2-D with a contiguous last dim')
            (K, M), lda = (x.shape, x.stride(0)) if trans & 1 else ((x.shape[1], x.shape[0]), x.stride(0))
            N, Kw = (w.shape[1], w.shape[0]) if trans & 2 else w.shape
            oshape = (M, N)
        else:
            M, lda = _rows(x)
            N, Kw = w.shape
            K = x.shape[-1]
            oshape = x.shape[:-1] + (N,)
        if Kw != K:
            raise _lib.DsrgHipError('gemm: x contracts over %d, w over %d (trans=%d)' % (K, Kw, trans))
        if outs is not None and outs[i] is not None:
            out = outs[i]
        else:
            out = torch.empty(oshape, device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
        Mo, ldc = _rows(out)
        ldr = 0
        if r is not None:
            Mr, ldr = _rows(r)
            if r.dtype != x.dtype or Mr != M:
                raise _lib.DsrgHipError('gemm: residual must match x rows/dtype')
        if b is not None and (b.dtype != torch.float32 or not b.is_contiguous()):
            raise _lib.DsrgHipError('gemm: bias must be contiguous fp32')
        if Mo != M or w.stride(1) != 1:
            raise _lib.DsrgHipError('gemm: bad out rows / weight layout')
        arr[i] = _lib.GemmProblem(_p(x), _p(w), _p(b), _p(r), _p(out), M, N, K, lda, w.stride(0), ldr, ldc)
        arr[i].trans = trans
        if want_pre:
            pre = torch.empty(oshape, device=x.device, dtype=x.dtype)
            arr[i].c_pre, arr[i].ldcp = _p(pre), _rows(pre)[1]
            pres.append(pre)
        if res_acts is not None and res_acts[i]:
            arr[i].res_act = int(res_acts[i])
        if colsums is not None and colsums[i] is not None:
            cs = colsums[i]
            if not (trans & 1) or cs.dtype != torch.float32 or cs.numel() != M or not cs.is_contiguous():
                raise _lib.DsrgHipError('gemm: colsums[i] must be contiguous fp32 [M] and needs trans bit 0')
            arr[i].a_colsum = _p(cs)
        f = fuse[i] if fuse is not None else None
        if f:
            g = arr[i]
            g.ln_eps = float(f.get('eps', 1e-12))
            if f.get('a_stats') is not None:
                st, c1 = f['a_stats']
                if st.dtype != torch.float32 or not st.is_contiguous() or st.shape != (K // 64, M, 2) or K % 64 or c1.shape != (N,) \
                        or c1.dtype != torch.float32 or not c1.is_contiguous():
                    raise _lib.DsrgHipError('gemm: a_stats must be contiguous fp32 [K/64, M, 2] with c1 fp32 [N]')
                g.a_stats, g.a_c1, g.a_np = _p(st), _p(c1), K // 64
            if f.get('r_stats') is not None:
                st, ga, be = f['r_stats']
                if r is None or st.dtype != torch.float32 or not st.is_contiguous() or st.shape != (N // 64, M, 2) or N % 64 \
                        or any(t.shape != (N,) or t.dtype != torch.float32 or not t.is_contiguous() for t in (ga, be)):
                    raise _lib.DsrgHipError('gemm: r_stats must be contiguous fp32 [N/64, M, 2] with gamma, beta fp32 [N] and a residual')
                g.r_stats, g.r_gamma, g.r_beta, g.r_np = _p(st), _p(ga), _p(be), N // 64
            if f.get('stats_out') is not None:
                st = f['stats_out']
                if st.dtype != torch.float32 or not st.is_contiguous() or st.shape != (N // 64, M, 2) or N % 64:
                    raise _lib.DsrgHipError('gemm: stats_out must be contiguous fp32 [N/64, M, 2]')
                g.stats_out = _p(st)
        res.append(out)
    odt = F32 if out_f32 else dt
    if tile is not None:
        _lib.check(lib.dsrg_gemm_nt_tiled(arr, n, dt, odt, act, tile[0], tile[1], tile[2] if len(tile) > 2 else 0, _stream()), 'dsrg_gemm_nt_tiled')
        return res
    _lib.check(lib.dsrg_gemm_nt(arr, n, dt, odt, act, _stream()), 'dsrg_gemm_nt')
    if GEMM_TRACE is not None:      # (before the want_pre return: the training-forward FFN-in launches are recorded too)
        bm, bn, nt, nst = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib.dsrg_gemm_nt_plan(arr, n, dt, C.byref(bm), C.byref(bn), C.byref(nt), C.byref(nst))
        # everything a replay of this record reads or writes stays alive with it: operands, outputs, pre-activations,
        # LayerNorm-fusion operands and the column-sum targets (slices of pooled zero chunks)
        keep = [t for p in problems for t in p if t is not None] + res + pres + [fuse] + [c for c in (colsums or []) if c is not None]
        GEMM_TRACE.append({'arr': arr, 'n': n, 'dt': dt, 'odt': odt, 'act': act, 'tile': (bm.value, bn.value, nst.value), 'fuse': 0 if not (fuse and any(f for f in fuse)) else (1 if any(f and f.get('a_stats') is not None for f in fuse) else 2), 'trans': int(arr[0].trans),
                           'wgs': nt.value, 'flops': sum(2.0 * a.M * a.N * a.K for a in arr), 'keep': keep})
    if want_pre:
        return res, pres
    return res


GEMM_TRACE = None   # bench.py sets this to a list to record the GEMM launches of one step


def gemm_replay(rec):
    """Re-issue a recorded dsrg_gemm_nt launch (same pointers; outputs are overwritten with the same values)."""
    _lib.check(_lib.load().dsrg_gemm_nt(rec['arr'], rec['n'], rec['dt'], rec['odt'], rec['act'], _stream()), 'dsrg_gemm_nt')


def _check_heads(nh, *ts):
    """the attention kernels are built for head dim 64: every operand must be [.., .., nh*64]"""
    for t in ts:
        if t.dim() != 3 or t.shape[2] != nh * 64:
            raise _lib.DsrgHipError('attention: operand %s is not [B, N, nh*64] with nh=%d (head dim must be 64)' % (tuple(t.shape), nh))


def attention(q, k, v, nh, kmask=None, mask_mode=_lib.MASK_NONE, dists=None, sprel_w=None, sprel_b=None, scale=0.125,
              want_lse=False, p_drop=0.0, seed=0):
    """q [B,Nq,nh*64], k/v [B,Nk,nh*64] views (row/batch strides free, last dim contiguous)."""
    lib = _lib.load()
    _dev(q)
    B, Nq, _ = q.shape
    Nk = k.shape[1]
    if isinstance(kmask, KeyLens):            # lengths instead of a mask tensor: the grouped entry carries them
        if want_lse:
            kmask = kmask.mask(Nk).contiguous()
        else:
            return attention_grouped([(q, k, v, kmask, dists)], nh, mask_mode, sprel_w, sprel_b, scale, p_drop, seed)[0]
    for t in (q, k, v):
        if t.stride(2) != 1 or t.dtype != q.dtype:
            raise _lib.DsrgHipError('attention: last dim must be contiguous, dtypes equal')
    _check_heads(nh, q, k, v)
    out = torch.empty(B, Nq, nh * 64, device=q.device, dtype=q.dtype)
    if kmask is not None:
        if kmask.dtype != torch.bool or kmask.shape != (B, Nk) or not kmask.is_contiguous():
            raise _lib.DsrgHipError('attention: kmask must be contiguous bool [B,Nk]')
    if dists is not None:
        if dists.dtype != torch.float32 or dists.shape != (B, Nq, Nk) or not dists.is_contiguous():
            raise _lib.DsrgHipError('attention: pair_dists must be contiguous fp32 [B,Nq,Nk]')
    lse = torch.empty(B, nh, Nq, device=q.device, dtype=torch.float32) if want_lse else None
    bs = lambda t: t.stride(0) if B > 1 else t.shape[1] * t.stride(1)
    _lib.check(lib.dsrg_attention_fwd(
        _p(q), bs(q), q.stride(1), _p(k), bs(k), k.stride(1), _p(v), bs(v), v.stride(1),
        _p(out), out.stride(0), out.stride(1), _p(kmask), mask_mode, _p(dists), _p(sprel_w), _p(sprel_b),
        B, nh, Nq, Nk, float(scale), _p(lse), float(p_drop), int(seed), _dt(q), _stream()), 'dsrg_attention_fwd')
    return (out, lse) if want_lse else out


class KeyLens:
    """Key validity given by lengths instead of a bool mask: key k of episode b is valid iff k % seg < lens[b, k // seg]
    (lens int64 [B, ceil(Nk / seg)] or [B] with seg = Nk).  Accepted by attention_grouped in place of kmask (inference)."""
    __slots__ = ('lens', 'seg', 'lens2')

    def __init__(self, lens, seg, lens2=None):
        self.lens, self.seg, self.lens2 = lens, int(seg), lens2      # lens2: optional second lengths tensor, added to the first

    def mask(self, Nk):
        """the bool [B, Nk] mask these lengths stand for (for the paths that need a tensor)"""
        B = self.lens.shape[0]
        nseg = (Nk + self.seg - 1) // self.seg
        k = torch.arange(Nk, device=self.lens.device)
        tot = self.lens if self.lens2 is None else self.lens + self.lens2
        return (k % self.seg)[None] < tot.reshape(B, nseg)[:, k // self.seg]


def attention_grouped(problems, nh, mask_mode=_lib.MASK_NONE, sprel_w=None, sprel_b=None, scale=0.125, p_drop=0.0, seed=0):
    """problems: list (<= 4) of (q, k, v, kmask|None, dists|None) sharing B, nh, mask mode and dtype -> outputs, ONE launch."""
    lib = _lib.load()
    n = len(problems)
    if not 1 <= n <= _lib.ATTN_MAX_GROUPS:
        raise _lib.DsrgHipError('attention_grouped: %d problems (max %d)' % (n, _lib.ATTN_MAX_GROUPS))
    q0 = problems[0][0]
    _dev(q0)
    B = q0.shape[0]
    arr = (_lib.AttnProblem * n)()
    outs = []
    for i, (q, k, v, kmask, dists) in enumerate(problems):
        Nq, Nk = q.shape[1], k.shape[1]
        for t in (q, k, v):
            if t.stride(2) != 1 or t.dtype != q0.dtype or t.shape[0] != B:
                raise _lib.DsrgHipError('attention: last dim must be contiguous, dtypes and batch equal')
        _check_heads(nh, q, k, v)
        klens, kseg, klens2 = None, 0, None
        if isinstance(kmask, KeyLens):       # validity from lengths (dsrg_attn_problem_t::klens): no mask tensor at all
            klens, kseg, klens2, kmask = kmask.lens, kmask.seg, kmask.lens2, None
            if klens2 is not None and (klens2.dtype != torch.int64 or not klens2.is_contiguous() or klens2.numel() != klens.numel()):
                raise _lib.DsrgHipError('attention: KeyLens.lens2 must match lens')
            if klens.dtype != torch.int64 or not klens.is_contiguous() or klens.numel() != B * ((Nk + kseg - 1) // kseg) or kseg < 1:
                raise _lib.DsrgHipError('attention: KeyLens needs contiguous int64 lengths [B, ceil(Nk / seg)]')
        if kmask is not None and (kmask.dtype != torch.bool or kmask.shape != (B, Nk) or not kmask.is_contiguous()):
            raise _lib.DsrgHipError('attention: kmask must be contiguous bool [B,Nk]')
        if (kmask is None and klens is None) != (mask_mode == _lib.MASK_NONE):
            raise _lib.DsrgHipError('attention_grouped: kmask and mask_mode disagree')
        if dists is not None and (dists.dtype != torch.float32 or dists.shape != (B, Nq, Nk) or not dists.is_contiguous()):
            raise _lib.DsrgHipError('attention: pair_dists must be contiguous fp32 [B,Nq,Nk]')
        out = torch.empty(B, Nq, nh * 64, device=q.device, dtype=q.dtype)
        outs.append(out)
        bs = lambda t: t.stride(0) if B > 1 else t.shape[1] * t.stride(1)
        a = arr[i]
        a.Q, a.K, a.V, a.O = _p(q), _p(k), _p(v), _p(out)
        a.q_bs, a.k_bs, a.v_bs, a.o_bs = bs(q), bs(k), bs(v), out.stride(0)
        a.q_rs, a.k_rs, a.v_rs, a.o_rs = q.stride(1), k.stride(1), v.stride(1), out.stride(1)
        a.kmask, a.pair_dists, a.lse, a.Nq, a.Nk = _p(kmask), _p(dists), None, Nq, Nk
        a.klens, a.kseg, a.klens2 = _p(klens), kseg, _p(klens2)
    _lib.check(lib.dsrg_attention_fwd_grouped(arr, n, mask_mode, _p(sprel_w), _p(sprel_b), B, nh, float(scale), float(p_drop), int(seed), _dt(q0),
                                              _stream()), 'dsrg_attention_fwd_grouped')
    return outs


def attention_bwd(q, k, v, o, do, lse, nh, kmask, mask_mode, dists, sprel_w, sprel_b, scale=0.125, p_drop=0.0, seed=0, grads=None):
    """-> dq, dk, dv, dsprel fp32 [2] (or None).  grads: optional (dq, dk, dv) to write into - [B,N,nh*64] views with unit
    column stride, e.g. the column blocks of one packed d(qkv) buffer (dsrg_attention_bwd_strided); else fresh contiguous."""
    lib = _lib.load()
    B, Nq, _ = q.shape
    Nk = k.shape[1]
    _check_heads(nh, q, k, v, o, do)
    do = do if do.stride(2) == 1 else do.contiguous()
    if grads is None:
        dq = torch.empty(B, Nq, nh * 64, device=q.device, dtype=q.dtype)
        dk = torch.empty(B, Nk, nh * 64, device=q.device, dtype=q.dtype)
        dv = torch.empty_like(dk)
    else:
        dq, dk, dv = grads
        _check_heads(nh, dq, dk, dv)
        if dq.shape[:2] != (B, Nq) or dk.shape[:2] != (B, Nk) or dv.shape[:2] != (B, Nk) or any(t.stride(2) != 1 or t.dtype != q.dtype for t in grads):
            raise _lib.DsrgHipError('attention_bwd: grads must be [B,Nq|Nk,nh*64] views of the input dtype with unit column stride')
    delta = torch.empty(B, nh, Nq, device=q.device, dtype=torch.float32)
    dsp = zeros_f32((2,), q.device) if dists is not None else None
    bs = lambda t: t.stride(0) if B > 1 else t.shape[1] * t.stride(1)
    if grads is None:
        _lib.check(lib.dsrg_attention_bwd(
            _p(q), bs(q), q.stride(1), _p(k), bs(k), k.stride(1), _p(v), bs(v), v.stride(1), _p(o), bs(o), o.stride(1),
            _p(do), bs(do), do.stride(1), _p(lse), _p(kmask), mask_mode if kmask is not None else _lib.MASK_NONE, _p(dists),
            _p(sprel_w), _p(sprel_b), _p(dq), _p(dk), _p(dv), _p(delta), _p(dsp), B, nh, Nq, Nk, float(scale), float(p_drop), int(seed),
            _dt(q), _stream()), 'dsrg_attention_bwd')
        return dq, dk, dv, dsp
    _lib.check(lib.dsrg_attention_bwd_strided(
        _p(q), bs(q), q.stride(1), _p(k), bs(k), k.stride(1), _p(v), bs(v), v.stride(1), _p(o), bs(o), o.stride(1),
        _p(do), bs(do), do.stride(1), _p(lse), _p(kmask), mask_mode if kmask is not None else _lib.MASK_NONE, _p(dists),
        _p(sprel_w), _p(sprel_b), _p(dq), bs(dq), dq.stride(1), _p(dk), bs(dk), dk.stride(1), _p(dv), bs(dv), dv.stride(1),
        _p(delta), _p(dsp), B, nh, Nq, Nk, float(scale), float(p_drop), int(seed),
        _dt(q), _stream()), 'dsrg_attention_bwd_strided')
    return dq, dk, dv, dsp


def sum_slabs(part, out=None, out_dtype=torch.float32):
    """part fp32 [S, ...] contiguous -> sum over dim 0 in slab order (dsrg_sum_slabs); out: optional contiguous target"""
    lib = _lib.load()
    _dev(part)
    if part.dtype != torch.float32 or not part.is_contiguous():
        raise _lib.DsrgHipError('sum_slabs: fp32 contiguous input')
    S, n = part.shape[0], part[0].numel()
    if out is None:
        out = torch.empty(part.shape[1:], device=part.device, dtype=out_dtype)
    if out.numel() != n or not out.is_contiguous() or n % 4:
        raise _lib.DsrgHipError('sum_slabs: out must be contiguous with the slab size (a multiple of 4 elements)')
    _lib.check(lib.dsrg_sum_slabs(_p(part), S, n, _p(out), n, _dt(out), _stream()), 'dsrg_sum_slabs')
    return out


def dropout_add(x, residual, p_drop, seed):
    """residual + dropout(x) (residual may be None) with the counter-hash mask of (seed, element index); dsrg_dropout_add"""
    lib = _lib.load()
    _dev(x)
    x = x if x.is_contiguous() else x.contiguous()
    if residual is not None:
        residual = residual if residual.is_contiguous() else residual.contiguous()
        if residual.shape != x.shape or residual.dtype != x.dtype:
            raise _lib.DsrgHipError('dropout_add: residual must have the shape and dtype of x')
    if x.numel() % 4:
        raise _lib.DsrgHipError('dropout_add: element count must be a multiple of 4')
    out = torch.empty_like(x)
    _lib.check(lib.dsrg_dropout_add(_p(x), _p(residual), _p(out), x.numel(), _dt(x), float(p_drop), int(seed), _stream()), 'dsrg_dropout_add')
    return out


_DROP_OFF = {}


def set_dropout_offset(t):
    """Register (or, with None, clear) the int64 device word that every dropout kernel adds to its seed on `t`'s device
    (dsrg_set_dropout_offset): a captured training step advances it once per replay.  The tensor is kept alive here."""
    lib = _lib.load()
    dev = torch.cuda.current_device() if t is None else _dev(t).device.index
    if t is not None and (t.dtype != torch.int64 or t.numel() != 1):
        raise _lib.DsrgHipError('set_dropout_offset: one int64 element')
    with torch.cuda.device(dev):
        _lib.check(lib.dsrg_set_dropout_offset(_p(t)), 'dsrg_set_dropout_offset')
    _DROP_OFF[dev] = t


def init():
    """dsrg_init() for the current device: resolve the library's static workspace outside any stream capture (required
    before attention_bwd_noalloc is captured into a HIP graph on this device)."""
    _lib.check(_lib.load().dsrg_init(), 'dsrg_init')


def attention_bwd_noalloc(q, k, v, o, do, lse, nh, kmask, mask_mode, dists, sprel_w, sprel_b, dq, dk, dv, dsp=None, scale=0.125,
                          p_drop=0.0, seed=0):
    """dsrg_attention_bwd_noalloc: the caller passes every output (dq, dk, dv contiguous, dsp zeroed fp32 [2] or None); the
    delta workspace is the library's static device array.  Nothing is allocated here or in the library."""
    lib = _lib.load()
    B, Nq, _ = q.shape
    Nk = k.shape[1]
    _check_heads(nh, q, k, v, o, do, dq, dk, dv)
    for t in (dq, dk, dv):
        if not t.is_contiguous() or t.dtype != q.dtype:
            raise _lib.DsrgHipError('attention_bwd_noalloc: dq / dk / dv must be contiguous and of the input dtype')
    if do.stride(2) != 1:
        raise _lib.DsrgHipError('attention_bwd_noalloc: dO rows must be contiguous (no copy is made here)')
    bs = lambda t: t.stride(0) if B > 1 else t.shape[1] * t.stride(1)
    _lib.check(lib.dsrg_attention_bwd_noalloc(
        _p(q), bs(q), q.stride(1), _p(k), bs(k), k.stride(1), _p(v), bs(v), v.stride(1), _p(o), bs(o), o.stride(1),
        _p(do), bs(do), do.stride(1), _p(lse), _p(kmask), mask_mode if kmask is not None else _lib.MASK_NONE, _p(dists),
        _p(sprel_w), _p(sprel_b), _p(dq), _p(dk), _p(dv), _p(dsp), B, nh, Nq, Nk, float(scale), float(p_drop), int(seed),
        _dt(q), _stream()), 'dsrg_attention_bwd_noalloc')
    return dq, dk, dv, dsp


def layernorm(x, gamma, beta, eps, residual=None, out_dtype=None, out=None):
    lib = _lib.load()
    _dev(x)
    M, ldx = _rows(x)
    H = x.shape[-1]
    out_dtype = out_dtype or (residual.dtype if residual is not None else x.dtype)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    Mo, ldo = _rows(out)
    ldr = _rows(residual)[1] if residual is not None else 0
    _lib.check(lib.dsrg_layernorm_fwd(_p(x), ldx, _p(residual), ldr, _p(gamma), _p(beta), _p(out), ldo, M, H,
                                      float(eps), _dt(x), _dt(out), _stream()), 'dsrg_layernorm_fwd')
    return out


def layernorm_grouped(problems, eps, out_dtype=None, p_drop=0.0, seed=0, outs=None, aux=None):
    """problems: list of (x, gamma, beta, residual|None) with one H and one input dtype -> list of outputs, ONE launch.
    p_drop > 0: out = LN(dropout(x) + residual) with the counter-based mask of (seed, problem index, row, col)
    (dsrg_layernorm_fwd_grouped_drop; layernorm_bwd recomputes it).
    outs: optional list (None = fresh) of destinations: a tensor with uniformly strided rows, or a 3-D [S, L, H] view whose
    S segments of L rows are strided apart (dsrg_ln_problem_t::o_seg: e.g. slots 2.. of a [B, 38, H] token buffer).
    aux: optional list (None = none) of (seg, t0, t1 | None): every seg-th row is also stored into row r // seg of the 2-D
    views t0 / t1 (row 0 of every sequence into a concat buffer; the RMF memory into both [MEM] slots)."""
    lib = _lib.load()
    n = len(problems)
    if not 1 <= n <= _lib.MAX_GROUPS:
        raise _lib.DsrgHipError('layernorm_grouped: %d problems (max %d)' % (n, _lib.MAX_GROUPS))
    x0 = problems[0][0]
    _dev(x0)
    H = x0.shape[-1]
    out_dtype = out_dtype or (problems[0][3].dtype if problems[0][3] is not None else x0.dtype)
    arr = (_lib.LnProblem * n)()
    outs_in, outs = outs, []
    for i, (x, g, b, r) in enumerate(problems):
        if x.shape[-1] != H or x.dtype != x0.dtype:
            raise _lib.DsrgHipError('layernorm_grouped: problems must share H and dtype')
        M, ldx = _rows(x)
        out = outs_in[i] if (outs_in is not None and outs_in[i] is not None) else torch.empty(x.shape, device=x.device, dtype=out_dtype)
        if out.dtype != out_dtype or out.shape[-1] != H or out.numel() != M * H or out.stride(-1) != 1:
            raise _lib.DsrgHipError('layernorm_grouped: destination must hold the %d x %d rows in the output dtype' % (M, H))
        outs.append(out)
        arr[i].x, arr[i].residual, arr[i].gamma, arr[i].beta, arr[i].out = _p(x), _p(r), _p(g), _p(b), _p(out)
        try:
            ldo = _rows(out)[1]
        except _lib.DsrgHipError:            # segments of rows strided apart
            if out.dim() != 3:
                raise
            arr[i].o_seg, arr[i].o_seg_ld, ldo = out.shape[1], out.stride(0), out.stride(1)
        arr[i].M, arr[i].ldx, arr[i].ldr, arr[i].ldo = M, ldx, (_rows(r)[1] if r is not None else 0), ldo
        if aux is not None and aux[i] is not None:
            seg, t0, t1 = aux[i]
            for t in (t0, t1):
                if t is not None and (t.dim() != 2 or t.shape != ((M + seg - 1) // seg, H) or t.stride(1) != 1 or t.dtype != out_dtype):
                    raise _lib.DsrgHipError('layernorm_grouped: aux destinations must be 2-D [ceil(M / seg), H] views of the output dtype')
            arr[i].aux_seg, arr[i].aux0, arr[i].aux1 = int(seg), _p(t0), _p(t1)
            arr[i].aux_ld0, arr[i].aux_ld1 = (t0.stride(0) if t0 is not None else 0), (t1.stride(0) if t1 is not None else 0)
    if p_drop > 0.0:
        _lib.check(lib.dsrg_layernorm_fwd_grouped_drop(arr, n, H, float(eps), _dt(x0), _dt(outs[0]), float(p_drop), int(seed), _stream()),
                   'dsrg_layernorm_fwd_grouped_drop')
        return outs
    _lib.check(lib.dsrg_layernorm_fwd_grouped(arr, n, H, float(eps), _dt(x0), _dt(outs[0]), _stream()), 'dsrg_layernorm_fwd_grouped')
    return outs


def _idx(t, M, what):
    """index operand of embed_ln -> (ptr, kind, stride, div).  `t` is a tensor (int32 / int64 / float32, 1-D, any
    stride) with M entries, or a tuple (tensor, div) with ceil(M / div) entries: row r uses entry r // div."""
    if t is None:
        return None, 0, 1, 1
    t, div = t if isinstance(t, tuple) else (t, 1)
    kind = {torch.int32: _lib.IDX_I32, torch.int64: _lib.IDX_I64, torch.float32: _lib.IDX_F32}.get(t.dtype)
    if kind is None or t.dim() != 1 or t.numel() * div < M or div < 1 or (t.numel() > 1 and t.stride(0) < 1):
        raise _lib.DsrgHipError('embed_ln: %s must be a 1-D int32/int64/float32 tensor covering %d rows' % (what, M))
    _dev(t)
    return t, kind, (t.stride(0) if t.numel() > 1 else 1), div


def embed_ln(M, H, dtype, device, base=None, feat=None, Wf=None, bf=None, lnf=None, E1=None, idx1=None,
             E2=None, idx2=None, vec=None, lno=None, eps=1e-12, out=None, seg=None, lnb=None):
    """seg = (base2 [B*No, H], T, V, No, lens0 int64 [B], lens1 int64 [B]): the base row of token t of episode b is gathered
    from base (view rows, V per episode) / base2 (object rows, No per episode) / zero by the two lengths
    (dsrg_embed_ln_fwd_seg).  lnb = (gamma, beta): LayerNorm on the base row before it is added (dsrg_embed_ln_fwd_lnb)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(M, H, device=device, dtype=dtype)
    _dev(out)
    ldo = _rows(out)[1]
    ldb = _rows(base)[1] if base is not None else 0
    F_, ldf = (feat.shape[-1], _rows(feat)[1]) if feat is not None else (0, 0)
    if feat is not None and feat.dtype != torch.float32:
        raise _lib.DsrgHipError('embed_ln: feat must be fp32')
    i1, k1, s1, d1 = _idx(idx1 if E1 is not None else None, M, 'idx1')
    i2, k2, s2, d2 = _idx(idx2 if E2 is not None else None, M, 'idx2')
    if base is not None and base.dtype != out.dtype:
        raise _lib.DsrgHipError('embed_ln: base dtype must match out dtype')
    lnf_g, lnf_b = lnf if lnf is not None else (None, None)
    lno_g, lno_b = lno if lno is not None else (None, None)
    if seg is not None:
        if lnb is not None:
            raise _lib.DsrgHipError('embed_ln: lnb and seg are exclusive')
        base2, T_, V_, No_, l0, l1 = seg
        B_ = M // T_
        if base is None or base2.dtype != out.dtype or _rows(base)[0] != B_ * V_ or _rows(base2)[0] != B_ * No_ or \
                any(t.dtype != torch.int64 or t.shape != (B_,) or not t.is_contiguous() for t in (l0, l1)):
            raise _lib.DsrgHipError('embed_ln: seg needs base [B*V,H], base2 [B*No,H] of the output dtype and int64 lengths [B]')
        _lib.check(lib.dsrg_embed_ln_fwd_seg(_p(base), ldb, _p(base2), _rows(base2)[1], T_, V_, No_, _p(l0), _p(l1), _p(feat), ldf, F_, _p(Wf), _p(bf),
                                             _p(lnf_g), _p(lnf_b), _p(E1), _p(i1), k1, s1, d1, _p(E2), _p(i2), k2, s2, d2, _p(vec), _p(lno_g), _p(lno_b),
                                             float(eps), _p(out), ldo, M, H, _dt(out), _stream()), 'dsrg_embed_ln_fwd_seg')
        return out
    if lnb is not None:
        if base is None or any(t.dtype != torch.float32 or t.shape != (H,) or not t.is_contiguous() for t in lnb):
            raise _lib.DsrgHipError('embed_ln: lnb needs a base and fp32 [H] gamma / beta')
        _lib.check(lib.dsrg_embed_ln_fwd_lnb(_p(base), ldb, _p(lnb[0]), _p(lnb[1]), _p(feat), ldf, F_, _p(Wf), _p(bf), _p(lnf_g), _p(lnf_b),
                                             _p(E1), _p(i1), k1, s1, d1, _p(E2), _p(i2), k2, s2, d2, _p(vec), _p(lno_g), _p(lno_b),
                                             float(eps), _p(out), ldo, M, H, _dt(out), _stream()), 'dsrg_embed_ln_fwd_lnb')
        return out
    _lib.check(lib.dsrg_embed_ln_fwd(_p(base), ldb, _p(feat), ldf, F_, _p(Wf), _p(bf), _p(lnf_g), _p(lnf_b),
                                     _p(E1), _p(i1), k1, s1, d1, _p(E2), _p(i2), k2, s2, d2, _p(vec), _p(lno_g), _p(lno_b),
                                     float(eps), _p(out), ldo, M, H, _dt(out), _stream()), 'dsrg_embed_ln_fwd')
    return out


def group_mean(x, G, out=None):
    """x [M*G, H] -> [M, H] mean over consecutive groups of G rows."""
    lib = _lib.load()
    _dev(x)
    MG, ldx = _rows(x)
    H = x.shape[-1]
    M = MG // G
    if out is None:
        out = torch.empty(M, H, device=x.device, dtype=x.dtype)
    _lib.check(lib.dsrg_group_mean_fwd(_p(x), ldx, _p(out), _rows(out)[1], M, G, H, _dt(x), _stream()),
               'dsrg_group_mean_fwd')
    return out


def gaa_pool(x, w, b, scatter=None, ln=None, rows_out=None):
    """x [B,V,H] (rows contiguous, batch stride free) -> pooled [B,H].  scatter = (buf [B,G,H], idx int64 [B]): the pooled row
    of episode b is also stored into buf[b, idx[b]] (dsrg_gaa_pool_fwd_ex).
    ln = (gamma, beta, eps): x is un-normalised; the kernel normalises every row, stores the rows to rows_out ([B,V,H], rows
    contiguous, batch stride free; fresh if None) and pools them -> (rows_out, pooled) (dsrg_gaa_pool_fwd_ln)."""
    lib = _lib.load()
    _dev(x)
    B, V, H = x.shape
    if x.stride(2) != 1 or x.stride(1) != H:
        raise _lib.DsrgHipError('gaa_pool: x must be [B,V,H] with contiguous rows')
    out = torch.empty(B, H, device=x.device, dtype=x.dtype)
    if ln is not None:
        g_, b_, eps_ = ln
        if rows_out is None:
            rows_out = torch.empty(B, V, H, device=x.device, dtype=x.dtype)
        if rows_out.shape != (B, V, H) or rows_out.dtype != x.dtype or rows_out.stride(2) != 1 or rows_out.stride(1) != H or \
                any(t.dtype != torch.float32 or t.shape != (H,) or not t.is_contiguous() for t in (g_, b_)):
            raise _lib.DsrgHipError('gaa_pool: ln needs fp32 [H] gamma / beta and a [B,V,H] rows_out of the input dtype with contiguous rows')
        buf = idx = None
        if scatter is not None:
            buf, idx = scatter
            if buf.dim() != 3 or buf.shape[0] != B or buf.shape[2] != H or buf.stride(2) != 1 or buf.stride(1) != H or buf.dtype != x.dtype \
                    or idx.dtype != torch.int64 or idx.shape != (B,) or not idx.is_contiguous():
                raise _lib.DsrgHipError('gaa_pool: scatter needs a [B,G,H] buffer of the input dtype with contiguous rows and int64 indices [B]')
        bs = lambda t: t.stride(0) if B > 1 else t.shape[1] * H
        _lib.check(lib.dsrg_gaa_pool_fwd_ln(_p(x), bs(x), _p(g_), _p(b_), float(eps_), _p(rows_out), bs(rows_out), _p(w), _p(b), _p(out), _p(buf),
                                            (bs(buf) if buf is not None else 0), _p(idx), (buf.shape[1] if buf is not None else 0), B, V, H,
                                            _dt(x), _stream()), 'dsrg_gaa_pool_fwd_ln')
        return rows_out, out
    buf = idx = None
    if scatter is not None:
        buf, idx = scatter
        if buf.dim() != 3 or buf.shape[0] != B or buf.shape[2] != H or buf.stride(2) != 1 or buf.stride(1) != H or buf.dtype != x.dtype \
                or idx.dtype != torch.int64 or idx.shape != (B,) or not idx.is_contiguous():
            raise _lib.DsrgHipError('gaa_pool: scatter needs a [B,G,H] buffer of the input dtype with contiguous rows and int64 indices [B]')
    xbs = x.stride(0) if B > 1 else V * H
    _lib.check(lib.dsrg_gaa_pool_fwd_ex(_p(x), xbs, _p(w), _p(b), _p(out), _p(buf), (buf.stride(0) if buf is not None and B > 1 else (buf.shape[1] * H if buf is not None else 0)),
                                        _p(idx), (buf.shape[1] if buf is not None else 0), B, V, H, _dt(x), _stream()), 'dsrg_gaa_pool_fwd_ex')
    return out


def head_tail(h, gamma, beta, w2, b2, eps=1e-12):
    lib = _lib.load()
    _dev(h)
    M, ldh = _rows(h)
    H = h.shape[-1]
    if h.dtype != torch.float32:
        raise _lib.DsrgHipError('head_tail: h must be fp32')
    score = torch.empty(h.shape[:-1], device=h.device, dtype=torch.float32)
    _lib.check(lib.dsrg_head_tail_fwd(_p(h), ldh, _p(gamma), _p(beta), _p(w2), _p(b2), float(eps), _p(score), M, H,
                                      _stream()), 'dsrg_head_tail_fwd')
    return score


def head_tail_grouped(items, eps=1e-12):
    """items: list (<= MAX_GROUPS) of (h fp32 [..,H], gamma, beta, w2, b2) sharing H -> list of scores [..], ONE launch"""
    lib = _lib.load()
    n = len(items)
    arr = (_lib.HeadTailProblem * n)()
    outs = []
    H = items[0][0].shape[-1]
    for i, (h, gamma, beta, w2, b2) in enumerate(items):
        _dev(h)
        if h.dtype != torch.float32 or h.shape[-1] != H:
            raise _lib.DsrgHipError('head_tail_grouped: h must be fp32 with a common width')
        M, ldh = _rows(h)
        score = torch.empty(h.shape[:-1], device=h.device, dtype=torch.float32)
        arr[i] = _lib.HeadTailProblem(_p(h), _p(gamma), _p(beta), _p(w2), _p(b2), _p(score), M, ldh)
        outs.append(score)
    _lib.check(lib.dsrg_head_tail_fwd_grouped(arr, n, H, float(eps), _stream()), 'dsrg_head_tail_fwd_grouped')
    return outs


def nav_logits(g_raw, l_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask, o_raw=None, vp_obj_masks=None):
    """-> (global, local, fused logits[, obj_logits when o_raw / vp_obj_masks are given: REVERIE's masked object-grounding scores])"""
    lib = _lib.load()
    _dev(g_raw)
    B, G = g_raw.shape
    Vp = l_raw.shape[1]
    for t in (g_raw, l_raw, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask):
        if not t.is_contiguous():
            raise _lib.DsrgHipError('nav_logits: inputs must be contiguous')
    gl, ll, fl = torch.empty_like(g_raw), torch.empty_like(l_raw), torch.empty_like(g_raw)
    if o_raw is not None:
        if o_raw.shape != l_raw.shape or o_raw.dtype != torch.float32 or not o_raw.is_contiguous() or vp_obj_masks.dtype != torch.bool \
                or vp_obj_masks.shape != l_raw.shape or not vp_obj_masks.is_contiguous():
            raise _lib.DsrgHipError('nav_logits: o_raw fp32 [B,Vp] and vp_obj_masks bool [B,Vp], contiguous')
        ol = torch.empty_like(l_raw)
        _lib.check(lib.dsrg_nav_logits_fwd_og(_p(g_raw), _p(l_raw), _p(fuse_raw), _p(gmap_masks), _p(gmap_visited), _p(vp_nav_masks), _p(gmap_src),
                                              _p(bw_mask), _p(gl), _p(ll), _p(fl), _p(o_raw), _p(vp_obj_masks), _p(ol), B, G, Vp, _stream()),
                   'dsrg_nav_logits_fwd_og')
        return gl, ll, fl, ol
    _lib.check(lib.dsrg_nav_logits_fwd(_p(g_raw), _p(l_raw), _p(fuse_raw), _p(gmap_masks), _p(gmap_visited),
                                       _p(vp_nav_masks), _p(gmap_src), _p(bw_mask), _p(gl), _p(ll), _p(fl),
                                       B, G, Vp, _stream()), 'dsrg_nav_logits_fwd')
    return gl, ll, fl


def igl_blend(aug, x, wa, ba, wo, bo):
    lib = _lib.load()
    _dev(x)
    if not (aug.is_contiguous() and x.is_contiguous()):
        raise _lib.DsrgHipError('igl_blend: inputs must be contiguous')
    H = x.shape[-1]
    M = x.numel() // H
    out = torch.empty_like(x)
    _lib.check(lib.dsrg_igl_blend_fwd(_p(aug), _p(x), _p(wa), _p(ba), _p(wo), _p(bo), _p(out), M, H, _dt(x),
                                      _stream()), 'dsrg_igl_blend_fwd')
    return out


# ----------------------------------------------------------------------------- backward-pass kernels
def transpose_pad(x, out_dtype=None, pad_to=64, colsum=None):
    """x [R,C] (row stride free) -> [C, Rp] with Rp = R rounded up to pad_to, zero filled.
    colsum: optional zeroed fp32 [C]; receives the column sums of x (the bias gradient) in the same pass."""
    lib = _lib.load()
    _dev(x)
    R, C_ = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    out = torch.empty(C_, Rp, device=x.device, dtype=out_dtype or x.dtype)
    if colsum is not None and (colsum.dtype != torch.float32 or colsum.numel() != C_ or not colsum.is_contiguous()):
        raise _lib.DsrgHipError('transpose_pad: colsum must be contiguous fp32 [C]')
    _lib.check(lib.dsrg_transpose_pad(_p(x), x.stride(0), _p(out), Rp, R, C_, Rp, _dt(x), _dt(out), _p(colsum), _stream()), 'dsrg_transpose_pad')
    return out


def transpose_pad_grouped(items, pad_to=64):
    """items: list of (x [R,C], colsum | None) -> list of [C, Rp] transposes, ONE launch per MAX_TRANSPOSES items."""
    lib = _lib.load()
    outs = []
    for s0 in range(0, len(items), _lib.MAX_TRANSPOSES):
        chunk = items[s0:s0 + _lib.MAX_TRANSPOSES]
        arr = (_lib.TransposeProblem * len(chunk))()
        for i, (x, cs) in enumerate(chunk):
            _dev(x)
            R, C_ = x.shape
            if x.stride(1) != 1:
                raise _lib.DsrgHipError('transpose_pad_grouped: last dim must be contiguous')
            Rp = (R + pad_to - 1) // pad_to * pad_to
            out = torch.empty(C_, Rp, device=x.device, dtype=x.dtype)
            if cs is not None and (cs.dtype != torch.float32 or cs.numel() != C_ or not cs.is_contiguous()):
                raise _lib.DsrgHipError('transpose_pad_grouped: colsum must be contiguous fp32 [C]')
            d = _dt(x)
            arr[i] = _lib.TransposeProblem(_p(x), _p(out), _p(cs), x.stride(0), Rp, R, C_, Rp, d, d, 0)
            outs.append(out)
        _lib.check(lib.dsrg_transpose_pad_grouped(arr, len(chunk), _stream()), 'dsrg_transpose_pad_grouped')
    return outs


def colsum(x, out=None):
    """x [M,N] -> fp32 [N] (added into `out` when given: a zeroed slice of a gradient buffer)"""
    lib = _lib.load()
    _dev(x)
    M, N = x.shape
    if out is None:
        out = zeros_f32((N,), x.device)
    elif out.dtype != torch.float32 or out.numel() != N or not out.is_contiguous():
        raise _lib.DsrgHipError('colsum: out must be contiguous fp32 [N]')
    _lib.check(lib.dsrg_colsum_acc(_p(x), x.stride(0), _p(out), M, N, _dt(x), _stream()), 'dsrg_colsum_acc')
    return out


def act_fwd(z, act):
    lib = _lib.load()
    _dev(z)
    z = z.contiguous()
    y = torch.empty_like(z)
    _lib.check(lib.dsrg_act_fwd(_p(z), _p(y), z.numel(), act, _dt(z), _stream()), 'dsrg_act_fwd')
    return y


def act_bwd(z, dy, act):
    lib = _lib.load()
    z, dy = z.contiguous(), dy.contiguous()
    dz = torch.empty_like(z)
    _lib.check(lib.dsrg_act_bwd(_p(z), _p(dy), _p(dz), z.numel(), act, _dt(z), _stream()), 'dsrg_act_bwd')
    return dz


def layernorm_bwd(x, residual, gamma, dy, eps, p_drop=0.0, seed=0, tag=0, want_dz=True):
    """returns dz (dtype of dy), dgamma, dbeta (fp32).  p_drop > 0 (the forward was layernorm_grouped(.., p_drop, seed) and x
    was problem `tag` of it): returns (dz | None, dzd, dgamma, dbeta) - dz the residual's gradient (skipped when not want_dz),
    dzd = dz * keep / (1 - p) the gradient of x."""
    lib = _lib.load()
    _dev(x)
    M, ldx = _rows(x)
    H = x.shape[-1]
    dy = rows_ok(dy)
    dgb = torch.empty((2, H), device=x.device, dtype=torch.float32)          # written, not accumulated (ABI 13)
    dg, db = dgb[0], dgb[1]
    part = torch.empty(max(1, lib.dsrg_layernorm_bwd_blocks(M)) * 2 * H, device=x.device, dtype=torch.float32)
    ldr = _rows(residual)[1] if residual is not None else 0
    if p_drop > 0.0:
        dz = torch.empty(x.shape, device=x.device, dtype=dy.dtype) if want_dz else None
        dzd = torch.empty(x.shape, device=x.device, dtype=dy.dtype)
        _lib.check(lib.dsrg_layernorm_bwd_drop(_p(x), ldx, _p(residual), ldr, _p(gamma), _p(dy), _rows(dy)[1], _p(dz), _rows(dzd)[1], _p(dzd),
                                               _rows(dzd)[1], _p(dg), _p(db), _p(part), M, H, float(eps), _dt(x), _dt(dy), float(p_drop), int(seed), int(tag),
                                               _stream()), 'dsrg_layernorm_bwd_drop')
        return dz, dzd, dg, db
    dz = torch.empty(x.shape, device=x.device, dtype=dy.dtype)
    _lib.check(lib.dsrg_layernorm_bwd(_p(x), ldx, _p(residual), ldr, _p(gamma), _p(dy), _rows(dy)[1], _p(dz), _rows(dz)[1],
                                      _p(dg), _p(db), _p(part), M, H, float(eps), _dt(x), _dt(dy), _stream()), 'dsrg_layernorm_bwd')
    return dz, dg, db


def action_tail(logits, targets=None, u=None, ignore_index=-100, want_grad=False):
    """nav logits [B,N] fp32 (-inf masked) -> dict(stop_prob, argmax, entropy[, sample][, loss_rows][, dlogits])."""
    lib = _lib.load()
    _dev(logits)
    if logits.dtype != torch.float32 or logits.dim() != 2 or logits.stride(1) != 1:
        raise _lib.DsrgHipError('action_tail: logits must be fp32 [B,N] with contiguous rows')
    B, N = logits.shape
    dev = logits.device
    if targets is not None and (targets.dtype != torch.int64 or targets.shape != (B,) or not targets.is_contiguous()):
        raise _lib.DsrgHipError('action_tail: targets must be contiguous int64 [B]')
    if u is not None and (u.dtype != torch.float32 or u.shape != (B,) or not u.is_contiguous()):
        raise _lib.DsrgHipError('action_tail: u must be contiguous fp32 [B]')
    out = {'stop_prob': torch.empty(B, device=dev), 'argmax': torch.empty(B, device=dev, dtype=torch.int32),
           'entropy': torch.empty(B, device=dev)}
    if u is not None:
        out['sample'] = torch.empty(B, device=dev, dtype=torch.int32)
    if targets is not None:
        out['loss_rows'] = torch.empty(B, device=dev)
        if want_grad:
            out['dlogits'] = torch.empty(B, N, device=dev)
    _lib.check(lib.dsrg_action_tail_fwd(_p(logits), logits.stride(0), _p(targets), int(ignore_index), _p(u), _p(out['stop_prob']),
                                        _p(out['argmax']), _p(out.get('sample')), _p(out['entropy']), _p(out.get('loss_rows')),
                                        _p(out.get('dlogits')), B, N, _stream()), 'dsrg_action_tail_fwd')
    return out


def ce_rows(logits, targets, ignore_index=-100):
    """fp32 logits [B,N] -> (loss_rows [B], lse_rows [B]) = F.cross_entropy(reduction='none') and the row log-sum-exp"""
    lib = _lib.load()
    _dev(logits)
    if logits.dtype != torch.float32 or logits.dim() != 2 or logits.stride(1) != 1:
        raise _lib.DsrgHipError('ce_rows: logits must be fp32 [B,N] with contiguous rows')
    B, N = logits.shape
    if targets.dtype != torch.int64 or targets.shape != (B,) or not targets.is_contiguous():
        raise _lib.DsrgHipError('ce_rows: targets must be contiguous int64 [B]')
    loss, lse = torch.empty(B, device=logits.device), torch.empty(B, device=logits.device)
    _lib.check(lib.dsrg_ce_rows_fwd(_p(logits), logits.stride(0), _p(targets), int(ignore_index), _p(lse), _p(loss), B, N, _stream()), 'dsrg_ce_rows_fwd')
    return loss, lse


def kl_rows(logits, target):
    """fp32 logits, target [R,N] -> (loss_rows [R], lse_rows [R]): F.kl_div(log_softmax(logits), target, 'none').sum(1)"""
    lib = _lib.load()
    _dev(logits)
    for t in (logits, target):
        if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise _lib.DsrgHipError('kl_rows: logits and target must be fp32 [R,N] with contiguous rows')
    if logits.shape != target.shape:
        raise _lib.DsrgHipError('kl_rows: logits and target must have one shape')
    R, N = logits.shape
    loss, lse = torch.empty(R, device=logits.device), torch.empty(R, device=logits.device)
    _lib.check(lib.dsrg_kl_rows_fwd(_p(logits), logits.stride(0), _p(target), target.stride(0), _p(lse), _p(loss), R, N, _stream()), 'dsrg_kl_rows_fwd')
    return loss, lse


def kl_rows_bwd(logits, target, lse, grad_rows):
    lib = _lib.load()
    R, N = logits.shape
    g = grad_rows if (grad_rows.dtype == torch.float32 and grad_rows.is_contiguous()) else grad_rows.float().contiguous()
    d = torch.empty(R, N, device=logits.device, dtype=torch.float32)
    _lib.check(lib.dsrg_kl_rows_bwd(_p(logits), logits.stride(0), _p(target), target.stride(0), _p(lse), _p(g), _p(d), N, R, N, _stream()), 'dsrg_kl_rows_bwd')
    return d


def ce_rows_bwd(logits, targets, lse, grad_rows, ignore_index=-100, out_dtype=torch.float32):
    """-> dlogits [B,N] (out_dtype) = grad_rows[:,None] * (softmax(logits) - onehot(targets))"""
    lib = _lib.load()
    B, N = logits.shape
    g = grad_rows if (grad_rows.dtype == torch.float32 and grad_rows.is_contiguous()) else grad_rows.float().contiguous()
    d = torch.empty(B, N, device=logits.device, dtype=out_dtype)
    _lib.check(lib.dsrg_ce_rows_bwd(_p(logits), logits.stride(0), _p(targets), int(ignore_index), _p(lse), _p(g), _p(d), N, _dt(d), B, N, _stream()),
               'dsrg_ce_rows_bwd')
    return d


# ----------------------------------------------------------------------------- backward of the row kernels
def group_mean_bwd(dy, G):
    lib = _lib.load()
    dy = rows_ok(dy)
    M, lddy = _rows(dy)
    H = dy.shape[-1]
    dx = torch.empty(M * G, H, device=dy.device, dtype=dy.dtype)
    _lib.check(lib.dsrg_group_mean_bwd(_p(dy), lddy, _p(dx), H, M, G, H, _dt(dy), _stream()), 'dsrg_group_mean_bwd')
    return dx


def gaa_pool_bwd(x, w, b, dout):
    lib = _lib.load()
    B, V, H = x.shape
    dout = dout.to(x.dtype).contiguous()
    dx = torch.empty_like(x)
    dwb = zeros_f32((H + 1,), x.device)
    _lib.check(lib.dsrg_gaa_pool_bwd(_p(x), _p(w), _p(b), _p(dout), _p(dx), _p(dwb), dwb.data_ptr() + 4 * H, B, V, H, _dt(x), _stream()),
               'dsrg_gaa_pool_bwd')
    return dx, dwb[:H], dwb[H:]


def head_tail_bwd(h, gamma, beta, w2, dscore, eps=1e-12):
    lib = _lib.load()
    M, ldh = _rows(h)
    H = h.shape[-1]
    dscore = dscore.float().contiguous()
    dh = torch.empty(h.shape, device=h.device, dtype=torch.float32)
    acc = zeros_f32((3 * H + 1,), h.device)
    base = acc.data_ptr()
    _lib.check(lib.dsrg_head_tail_bwd(_p(h), ldh, _p(gamma), _p(beta), _p(w2), _p(dscore), float(eps), _p(dh), H, base, base + 4 * H,
                                      base + 8 * H, base + 12 * H, M, H, _stream()), 'dsrg_head_tail_bwd')
    return dh, acc[:H], acc[H:2 * H], acc[2 * H:3 * H], acc[3 * H:]


def nav_logits_bwd(g_raw, l_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask, dgl, dll, dfl):
    lib = _lib.load()
    B, G = g_raw.shape
    Vp = l_raw.shape[1]
    c = lambda t: None if t is None else t.float().contiguous()
    dgl, dll, dfl = c(dgl), c(dll), c(dfl)
    dg, dl = torch.empty_like(g_raw), torch.empty_like(l_raw)
    df = torch.empty(B, device=g_raw.device, dtype=torch.float32) if fuse_raw is not None else None
    _lib.check(lib.dsrg_nav_logits_bwd(_p(g_raw), _p(l_raw), _p(fuse_raw), _p(gmap_masks), _p(gmap_visited), _p(vp_nav_masks), _p(gmap_src),
                                       _p(bw_mask), _p(dgl), _p(dll), _p(dfl), _p(dg), _p(dl), _p(df), B, G, Vp, _stream()),
               'dsrg_nav_logits_bwd')
    return dg, dl, df


def igl_blend_bwd(aug, x, wa, ba, wo, bo, dout):
    lib = _lib.load()
    H = x.shape[-1]
    M = x.numel() // H
    dout = dout.to(x.dtype).contiguous()
    daug, dx = torch.empty_like(aug), torch.empty_like(x)
    acc = zeros_f32((2 * H + 2,), x.device)
    base = acc.data_ptr()
    _lib.check(lib.dsrg_igl_blend_bwd(_p(aug), _p(x), _p(wa), _p(ba), _p(wo), _p(bo), _p(dout), _p(daug), _p(dx), base, base + 4 * H,
                                      base + 8 * H, M, H, _dt(x), _stream()), 'dsrg_igl_blend_bwd')
    return daug, dx, acc[:H], acc[2 * H:2 * H + 1], acc[H:2 * H], acc[2 * H + 1:]


def embed_ln_bwd(M, H, dout, base=None, feat=None, Wf=None, bf=None, lnf=None, E1=None, idx1=None, E2=None, idx2=None,
                 vec=None, lno=None, eps=1e-12, need_dbase=True, E1_padding_idx=None):
    """Backward of embed_ln -> dict with the gradients of the terms that are present (fp32, table gradients dense).
    E1_padding_idx: row of E1 that gets no gradient (nn.Embedding's padding_idx)."""
    lib = _lib.load()
    dout = rows_ok(dout)
    _dev(dout)
    dev = dout.device
    a = _lib.EmbedBwdArgs()
    lnf_g, lnf_b = lnf if lnf is not None else (None, None)
    lno_g, lno_b = lno if lno is not None else (None, None)
    i1, k1, s1, d1 = _idx(idx1 if E1 is not None else None, M, 'idx1')
    i2, k2, s2, d2 = _idx(idx2 if E2 is not None else None, M, 'idx2')
    z = lambda t: None if t is None else zeros_f32(tuple(t.shape), dev)
    g = {'Wf': z(Wf) if feat is not None else None, 'bf': z(bf) if feat is not None else None,
         'lnf_g': z(lnf_g), 'lnf_b': z(lnf_b), 'E1': z(E1), 'E2': z(E2), 'vec': z(vec), 'lno_g': z(lno_g), 'lno_b': z(lno_b),
         'base': torch.empty(M, H, device=dev, dtype=dout.dtype) if (base is not None and need_dbase) else None}
    a.base, a.feat, a.Wf, a.bf, a.lnf_gamma, a.lnf_beta = _p(base), _p(feat), _p(Wf), _p(bf), _p(lnf_g), _p(lnf_b)
    a.E1, a.idx1, a.E2, a.idx2, a.vec, a.lno_gamma, a.lno_beta, a.dout = _p(E1), _p(i1), _p(E2), _p(i2), _p(vec), _p(lno_g), _p(lno_b), _p(dout)
    a.dbase, a.dWf, a.dbf, a.dlnf_gamma, a.dlnf_beta = _p(g['base']), _p(g['Wf']), _p(g['bf']), _p(g['lnf_g']), _p(g['lnf_b'])
    a.dE1, a.dE2, a.dvec, a.dlno_gamma, a.dlno_beta = _p(g['E1']), _p(g['E2']), _p(g['vec']), _p(g['lno_g']), _p(g['lno_b'])
    a.ldb = _rows(base)[1] if base is not None else 0
    a.ldf, a.F = (_rows(feat)[1], feat.shape[-1]) if feat is not None else (0, 0)
    a.idx1_kind, a.idx1_stride, a.idx1_div, a.idx2_kind, a.idx2_stride, a.idx2_div = k1, s1, d1, k2, s2, d2
    a.lddo, a.lddb, a.M, a.H, a.dtype, a.eps = _rows(dout)[1], H, M, H, _dt(dout), float(eps)
    a.E1_rows, a.E2_rows = (E1.shape[0] if E1 is not None else 0), (E2.shape[0] if E2 is not None else 0)
    a.E1_padding_idx = -1 if E1_padding_idx is None else int(E1_padding_idx)
    # a table of a few dozen rows hit by thousands of token rows (the 45 object classes): spread the same-address float
    # atomics over 32 copies of the gradient table and sum them afterwards
    copies = 32 if (E1 is not None and 4 < E1.shape[0] <= 256 and M >= 2048) else 1
    if copies > 1:
        g['E1'] = torch.zeros(copies, E1.shape[0] * H, device=dev, dtype=torch.float32)
        a.dE1 = _p(g['E1'])
    a.E1_copies = copies
    nws = int(lib.dsrg_embed_ln_bwd_workspace(C.byref(a)))
    part = torch.empty(nws, device=dev, dtype=torch.float32) if nws > 0 else None      # per-workgroup column sums (no atomics)
    a.partials = _p(part)
    _lib.check(lib.dsrg_embed_ln_bwd(C.byref(a), _stream()), 'dsrg_embed_ln_bwd')
    if copies > 1:
        g['E1'] = colsum(g['E1']).view(E1.shape)
    return g


def segment_wsum(in0, in1, seg_ptr, idx, w, m0, m1=0):
    """out[r] = sum_e w[e] * cat(in0, in1)[idx[e]] over the entries of segment r -> (out0 [m0,H], out1 [m1,H] or None)."""
    lib = _lib.load()
    ref = in0 if in0 is not None else in1
    _dev(ref)
    H = ref.shape[-1]
    n0, ld0 = _rows(in0) if in0 is not None else (0, 0)
    n1, ld1 = _rows(in1) if in1 is not None else (0, 0)
    if in0 is not None and in1 is not None and in0.dtype != in1.dtype:
        raise _lib.DsrgHipError('segment_wsum: both tables must have one dtype')
    for t, dt in ((seg_ptr, torch.int32), (idx, torch.int32), (w, torch.float32)):
        if t.dtype != dt or not t.is_contiguous():
            raise _lib.DsrgHipError('segment_wsum: seg_ptr / idx must be contiguous int32, w contiguous fp32')
    if seg_ptr.numel() != m0 + m1 + 1:
        raise _lib.DsrgHipError('segment_wsum: seg_ptr must have m0 + m1 + 1 entries')
    out0 = torch.empty(m0, H, device=ref.device, dtype=ref.dtype) if m0 else None
    out1 = torch.empty(m1, H, device=ref.device, dtype=ref.dtype) if m1 else None
    _lib.check(lib.dsrg_segment_wsum(_p(in0), ld0, n0, _p(in1), ld1, n1, _p(out0), H, m0, _p(out1), H, m1, _p(seg_ptr), _p(idx), _p(w),
                                     H, _dt(ref), _stream()), 'dsrg_segment_wsum')
    return out0, out1
