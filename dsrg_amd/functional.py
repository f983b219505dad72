"""Differentiable ops of the DSRG hot path.

Forward ALWAYS runs the hand-written HIP kernel (kernels.py -> libdsrg_hip.so); there is no other forward, and every
backward is a hand-written HIP kernel as well:
  * linear / grouped linear, dense(+dropout)+residual+LayerNorm, LayerNorm, activations: dgrad / wgrad on the MFMA GEMM
    with transposed operands, LayerNorm-backward and activation-backward kernels, bias column sums fused into the dY
    transpose -- >= 97 % of the backward FLOPs;
  * attention: two recompute kernels (dQ; dK/dV) from the forward's log-sum-exp, fp32 arithmetic;
  * token-embedding sums, GAA pooling, ASV mean, head tails, logit fusion, IGL gate: train_rows.hip.
ATen is used for bookkeeping only (dropout masks, views, zero-filled gradient buffers)."""
import math
import torch
import torch.nn.functional as F

from . import _lib, kernels as K
from ._lib import ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH, MASK_NONE, MASK_NEG10000, MASK_NEGINF, MASK_PLUS1


# ----------------------------------------------------------------------------- weight packing
def _foreach_copy_grouped(dsts, srcs):
    """torch._foreach_copy_ per (destination dtype, source dtype) group: its multi-tensor fast path needs one dtype
    per list."""
    groups = {}
    for d, s_ in zip(dsts, srcs):
        g = groups.setdefault((d.dtype, s_.dtype), ([], []))
        g[0].append(d); g[1].append(s_)
    for d, s_ in groups.values():
        torch._foreach_copy_(d, s_)


# DistributedDataParallel (find_unused_parameters=True) expects a gradient for every parameter that is REACHABLE in the
# autograd graph; all parameters reach the single _PackFn node, so under DDP the ones a task did not touch must get an
# explicit zero gradient (views of one zero buffer: one fill kernel) or the reducer never finishes its buckets.
# dsrg_amd.distributed.wrap_model switches this on.
MATERIALIZE_UNUSED_GRADS = False

# Packed / cast copies of the parameters are cached per parameter version.  Tensor._version does NOT move under fused
# optimizers (torch.optim.AdamW(fused=True)) nor under optimizers that write through `.data` (the reference's own AdamW,
# pretrain_src/optim/adamw.py), so every cache key also carries this counter, which every backward pass through a pack
# advances: an optimizer step is always preceded by one.
PACK_EPOCH = [0]


def pack_version(srcs):
    """cache key of a pack: the epoch above, every source's version counter, and every source's storage address (a
    parameter re-seated with `p.data = ...` / load_state_dict(assign=True) moves neither of the first two)"""
    return (PACK_EPOCH[0],) + tuple(p._version for p in srcs) + tuple(p.data_ptr() for p in srcs)


class _PackFn(torch.autograd.Function):
    """All compute-dtype / packed copies of the parameters in one autograd node: forward = one multi-tensor copy
    (torch._foreach_copy_, casts included) of every source into its row range of its packed buffer; backward = one
    multi-tensor copy of the packed gradients' row ranges back into fresh parameter-dtype gradients.  Replaces one
    cast (+ one cat) kernel per weight and their autograd mirrors (~1000 tiny launches per training step)."""

    @staticmethod
    def forward(ctx, spec, *srcs):
        plan = getattr(spec, 'plan', None)
        if plan is not None or (isinstance(spec, _Spec) and all(s.is_contiguous() for s in srcs)):
            # Everything that does not change from step to step (detached flat aliases of the parameters, the layout of the
            # packed buffers, the grouping by dtype pair) is computed once per packer: a step then costs one allocation and one
            # split per output dtype, the multi-tensor copies, and one view per output - not ~4 tensor ops per parameter.
            # the plan aliases the parameters' storage: a parameter whose storage was swapped outside Module._apply
            # (`p.data = ...`, vector_to_parameters, an EMA swap, load_state_dict(assign=True)) invalidates it
            if plan is None or plan['ptrs'] != [t.data_ptr() for t in srcs]:
                managed = spec.managed
                plan = spec.plan = _pack_plan(spec, srcs)
                if managed is not None:            # the parameters moved: the optimizer's destinations moved with the buffers
                    managed.attach_packs(managed._pack_specs)
            if spec.managed is None:
                flats = {d: torch.empty(n, dtype=d, device=plan['dev']) for d, n in plan['total'].items()}
                parts = {d: flats[d].split(plan['sizes'][d]) for d in flats}
                for (dd, sd), (pidx, dets, _) in plan['copies'].items():
                    pp = parts[dd]
                    torch._foreach_copy_([pp[j] for j in pidx], dets)
            else:
                # optimizer-managed pack (manage_pack): persistent buffers; the optimizer's update kernel writes every changed
                # parameter's bf16 copy, so the cast pass runs only when something else touched the parameters (version
                # counters) and for the sources the optimizer does not cover
                flats = plan['pflats']
                parts = {d: flats[d].split(plan['sizes'][d]) for d in flats}
                vers = tuple(t._version for t in srcs)
                full = plan.get('synced') != vers
                for (dd, sd), (pidx, dets, _) in plan['copies'].items():
                    keep = range(len(pidx)) if full else plan['unmanaged'].get((dd, sd), ())
                    if len(keep):
                        torch._foreach_copy_([parts[dd][pidx[i]] for i in keep], [dets[i] for i in keep])
                plan['synced'] = vers
            outs = [torch.as_strided(flats[d], shape, strides, off) for d, shape, strides, off in plan['outs']]
            if isinstance(ctx, torch.autograd.function.FunctionCtx):
                ctx.set_materialize_grads(False)
            ctx.spec = spec
            ctx.meta = plan['meta']
            return tuple(outs)
        outs, dst = [], []
        for dtype, idxs in spec:
            rows = sum(srcs[i].shape[0] for i in idxs)
            o = torch.empty((rows,) + tuple(srcs[idxs[0]].shape[1:]), device=srcs[idxs[0]].device, dtype=dtype)
            r = 0
            for i in idxs:
                dst.append((i, o[r:r + srcs[i].shape[0]]))
                r += srcs[i].shape[0]
            outs.append(o)
        _foreach_copy_grouped([d for _, d in dst], [srcs[i].detach() for i, _ in dst])
        if isinstance(ctx, torch.autograd.function.FunctionCtx):
            ctx.set_materialize_grads(False)       # weights a task does not touch get no gradient, not a zero tensor
        ctx.spec = spec
        ctx.meta = [(t.shape, t.dtype, t.device) for t in srcs]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        PACK_EPOCH[0] += 1           # the parameters are about to change: no cached pack survives this backward
        grads = [None] * len(ctx.meta)
        dst, src = [], []
        # optimizer-managed pack with packed gradients on (optim.AdamW.packed_grads): the bf16 gradient of a packed weight is
        # handed to the optimizer AS IT LIES (address of the source's row range) instead of being cast into an fp32 .grad
        opt = ctx.spec.managed
        stash = opt._gstash if (opt is not None and opt.packed_grads and ctx.spec.plan is not None) else None
        msrc = ctx.spec.plan.get('managed_src', {}) if stash is not None else {}
        for (dtype, idxs), g in zip(ctx.spec, gouts):
            if g is None:
                continue
            r = 0
            lies = stash is not None and g.dtype == torch.bfloat16 and g.is_contiguous()
            for i in idxs:
                shape, dt, dev = ctx.meta[i]
                if lies and i in msrc and ctx.needs_input_grad[1 + i]:
                    row = 1
                    for d_ in shape[1:]:
                        row *= int(d_)
                    stash[msrc[i]] = (g.data_ptr() + 2 * r * row, g)
                elif ctx.needs_input_grad[1 + i]:
                    grads[i] = torch.empty(shape, dtype=dt, device=dev)
                    dst.append(grads[i]); src.append(g[r:r + shape[0]])
                r += shape[0]
        if dst:
            _foreach_copy_grouped(dst, src)
        if MATERIALIZE_UNUSED_GRADS:
            miss = [i for i in range(len(grads)) if grads[i] is None and ctx.needs_input_grad[1 + i]]
            if miss:
                by = {}
                for i in miss:
                    by.setdefault((ctx.meta[i][1], ctx.meta[i][2]), []).append(i)
                for (dt, dev), idxs in by.items():
                    sizes = [int(torch.Size(ctx.meta[i][0]).numel()) for i in idxs]
                    zero = torch.zeros(sum(sizes), dtype=dt, device=dev)
                    o = 0
                    for i, n in zip(idxs, sizes):
                        grads[i] = zero[o:o + n].view(ctx.meta[i][0]); o += n
        return (None,) + tuple(grads)


class _Spec(tuple):
    """the packer's (dtype, source indices) groups; a tuple subclass so that the static part of the pack (`plan`) can ride on it"""
    plan = None
    managed = None          # the optimizer that keeps this pack's buffers current (manage_pack)


def manage_pack(spec, optimizer):
    """Hand a pack's bf16 copies over to `optimizer` (dsrg_amd.optim.AdamW.attach_packs) - or, with None, take them back.
    Gives the pack persistent buffers and returns [(parameter address, address of its bf16 copy)] for every fp32 source that
    is a plain cast into a bf16 buffer; the optimizer's update kernel then writes those copies and _PackFn.forward skips them.
    Needs a plan, i.e. one forward in grad mode must have run."""
    plan = spec.plan
    if plan is None:
        raise RuntimeError('manage_pack: this pack has not run yet (one forward first)')
    if optimizer is None:
        spec.managed = None
        plan.pop('pflats', None); plan.pop('synced', None); plan.pop('unmanaged', None); plan.pop('managed_src', None)
        return []
    if 'pflats' not in plan:
        plan['pflats'] = {d: torch.empty(n, dtype=d, device=plan['dev']) for d, n in plan['total'].items()}
    plan['synced'] = None
    pairs, unmanaged, seen = [], {}, set()
    owned = {p.data_ptr() for p in optimizer._index}
    managed_src = {}
    for (dd, sd), (pidx, dets, sidx) in plan['copies'].items():
        offs, o = [], 0
        for n in plan['sizes'][dd]:
            offs.append(o); o += n
        base = plan['pflats'][dd].data_ptr()
        es = plan['pflats'][dd].element_size()
        left = []
        for i, (j, det) in enumerate(zip(pidx, dets)):
            sp = det.data_ptr()
            if dd == torch.bfloat16 and sd == torch.float32 and sp in owned and sp not in optimizer._pack_dst and sp not in seen:
                pairs.append((sp, base + offs[j] * es)); seen.add(sp); managed_src[sidx[i]] = sp
            else:
                left.append(i)
        unmanaged[(dd, sd)] = left
    plan['unmanaged'], plan['managed_src'] = unmanaged, managed_src
    spec.managed = optimizer
    return pairs


def _pack_plan(spec, srcs):
    dev = srcs[0].device
    total, sizes, copies, outs = {}, {}, {}, []
    for dtype, idxs in spec:
        off = total.get(dtype, 0)
        sz = sizes.setdefault(dtype, [])
        pad = (-off) % 64                      # every packed tensor starts on a 64-element boundary (16-byte aligned rows for the GEMM)
        if pad:
            sz.append(pad); off += pad
        rows = sum(srcs[i].shape[0] for i in idxs)
        shape = (rows,) + tuple(srcs[idxs[0]].shape[1:])
        st, acc = [], 1
        for d_ in reversed(shape):
            st.append(acc); acc *= d_
        outs.append((dtype, shape, tuple(reversed(st)), off))
        for i in idxs:
            n = srcs[i].numel()
            g = copies.setdefault((dtype, srcs[i].dtype), ([], [], []))
            g[0].append(len(sz)); g[1].append(srcs[i].detach().view(-1)); g[2].append(i)
            sz.append(n); off += n
        total[dtype] = off
    return {'dev': dev, 'total': total, 'sizes': sizes, 'copies': copies, 'outs': outs, 'meta': [(t.shape, t.dtype, t.device) for t in srcs],
            'ptrs': [t.data_ptr() for t in srcs]}


class WeightPacker:
    """Collects cast / concat requests while a weight dict is being built and serves them with one _PackFn call.
    `cast(t)` / `cat(ts)` return placeholders; `resolve(W)` swaps them for the packed tensors."""

    class _Lazy:
        __slots__ = ('k',)

        def __init__(self, k):
            self.k = k

    class _View:
        __slots__ = ('src', 'shape')

        def __init__(self, src, shape):
            self.src, self.shape = src, shape

    def __init__(self):
        self.srcs, self.spec = [], []

    @staticmethod
    def view(t, shape):
        """placeholder for t.reshape(shape), t a parameter or a cast()/cat() placeholder.  The view is taken at resolve()
        time, in the caller's grad mode: a view of a parameter that was taken while the weight dict was being built would
        carry that moment's grad mode (none: the weight silently stops training) and, with autograd on, would keep the
        parameter's AccumulateGrad node - pinned to the stream of that first forward - alive for as long as the plan lives
        (which breaks a later HIP-graph capture of the step on another stream)."""
        return WeightPacker._View(t, tuple(shape))

    def cat(self, ts, dtype):
        idxs = []
        for t in ts:
            idxs.append(len(self.srcs)); self.srcs.append(t)
        self.spec.append((dtype, tuple(idxs)))
        return WeightPacker._Lazy(len(self.spec) - 1)

    def cast(self, t, dtype):
        if t.dtype == dtype and t.is_contiguous():
            return t
        return self.cat([t], dtype)

    def resolve(self, W):
        outs = ()
        if self.spec:
            spec = getattr(self, '_spec_t', None)
            if spec is None or len(spec) != len(self.spec):
                spec = self._spec_t = _Spec(self.spec)
            if _needs_grad(self.srcs):
                outs = _PackFn.apply(spec, *self.srcs)
            else:
                with torch.no_grad():
                    outs = _PackFn.forward(_NoCtx(), spec, *self.srcs)
        for k, v in W.items():
            if isinstance(v, WeightPacker._Lazy):
                W[k] = outs[v.k]
            elif isinstance(v, WeightPacker._View):
                W[k] = (outs[v.src.k] if isinstance(v.src, WeightPacker._Lazy) else v.src).reshape(v.shape)
        return W


class _NoCtx:
    pass


# ----------------------------------------------------------------------------- linear (grouped), hand-written backward
def _as2d(t):
    """[..., H] view with uniformly strided rows -> 2-D [M, H] view (no copy)."""
    if t.dim() == 2:
        return t
    M, ld = K._rows(t)
    return t.as_strided((M, t.shape[-1]), (ld, 1), t.storage_offset())


BWD_AS_LIES = __import__('os').environ.get('DSRG_BWD_AS_LIES', '1') == '1'    # A/B knob: 0 = always transposed copies


def _lies_ok(t):
    """a 2-D bf16 operand the transposed-operand GEMM can take as it lies: unit column stride, 16-byte rows and base"""
    return t.dtype == torch.bfloat16 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0


def _gemm_grads(items):
    """items: list of (x2d, w, dy2d (compute dtype), need_dx, need_dw, need_db[, z2d, act]) -> (dx list, dw list, db list).
    With (z2d, act) the dgrad leaves as dX * act'(z) (the GEMM epilogue's res_act: the activation backward of the layer
    below costs no kernel).
    bf16 with friendly shapes (the transformer layers): dX = dY.W and dW = dY^T.X run on dY, W and X AS THEY LIE
    (dsrg_gemm_problem_t::trans = 2 / 3, ds_read_b64_tr_b16 fragments): one grouped launch for the dgrads, one for the wgrads
    (any row count: the partial last K-step is zero-filled on chip), no transposed copies; db = column sums of dY out of the
    wgrad launch (a_colsum).  Anything else (fp32 parity mode, ragged rows such as the masked-token
    heads) goes through the NT kernel on transposed, zero-padded copies (kernels.transpose_pad_grouped, which also takes the
    bias gradients)."""
    nb = sum(it[1].shape[0] for it in items if it[5])
    dball = K.zeros_f32((nb,), items[0][2].device) if nb else None
    dxp, dwp, dbs, o = [], [], [], 0
    fdx, fdw, fdb = [], [], []                 # problems on operands as they lie (fdb: where a wgrad's column sums of dY go)
    sdw = []                                   # wgrads that run K-split (few output tiles, thousands of rows)
    fra, tra = [], []                          # res_act of the dgrad problems (fast / transposed-copy path)
    tr = []                                    # every transpose of this call: one grouped launch
    slots = []                                 # per item: where its dx / dw come from
    for it in items:
        x, w, dy, ndx, ndw, ndb = it[:6]
        zr, ract = (it[6], it[7]) if len(it) > 6 else (None, 0)
        N, Kd = w.shape
        M = dy.shape[0]
        db = None
        if ndb:
            db = dball[o:o + N]; o += N
        lies = BWD_AS_LIES and _lies_ok(dy) and _lies_ok(w) and N % 8 == 0 and Kd % 8 == 0 and M > 0
        sx = sw = None
        if ndx:
            if lies and N % 64 == 0:
                sx = ('f', len(fdx)); fdx.append((dy, w, None, zr)); fra.append(ract)    # [M,N] x ([N,K] stored [K',N'])
            else:
                Np = (N + 63) // 64 * 64
                dyp = dy if Np == N else torch.nn.functional.pad(dy, (0, Np - N))
                sx = ('t', len(dxp)); dxp.append([dyp, len(tr), None, zr]); tra.append(ract); tr.append((w, None))   # [M,Np] x [K,Np]^T
        if ndw:
            if lies and _lies_ok(x) and _wgrad_splits(M, N, Kd) > 1:
                sw = ('s', len(sdw)); sdw.append((dy, x, db))     # a small dW over many rows: K-split, below
            elif lies and _lies_ok(x):
                sw = ('f', len(fdw)); fdw.append((dy, x, None, None)); fdb.append(db)    # dY [M,N] = [K',M'], X [M,K] = [K',N']
            else:
                sw = ('t', len(dwp)); dwp.append([len(tr), len(tr) + 1, None, None]); tr += [(dy, db), (x, None)]   # [N,Mp] x [K,Mp]^T
        elif ndb:
            K.colsum(dy, out=db)
        dbs.append(db)
        slots.append((sx, sw))
    res = {'f': [None, None], 't': [None, None], 's': [None, [_wgrad_split(dy, x, db) for dy, x, db in sdw]]}
    # dX feeds the next backward node, dW has no reader until the optimizer: with both present the wgrad launches go to a side
    # stream (a parallel branch of a captured step) beside the dgrad launches and are joined before this call returns (10.40 ->
    # 10.27 ms per pre-training step; joining later would race with autograd's accumulation of a weight used twice).  Their
    # outputs are allocated here, on the calling stream, so the caching allocator never hands them out on the side stream.
    side = cur = None
    if fdx and fdw and WGRAD_STREAM and torch.cuda.is_current_stream_capturing():      # (launched eagerly the step is host-bound: a fork / join per call only costs)
        cur = torch.cuda.current_stream()
        side = _wgrad_side_stream(cur)
        wouts = [torch.empty(dy.shape[1], x.shape[1], device=dy.device, dtype=dy.dtype) for dy, x, _, _ in fdw]
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for i in range(0, len(fdw), _lib.MAX_GROUPS):
                K.gemm(fdw[i:i + _lib.MAX_GROUPS], trans=3, colsums=fdb[i:i + _lib.MAX_GROUPS], outs=wouts[i:i + _lib.MAX_GROUPS])
        res['f'][1] = wouts
    if fdx:
        res['f'][0] = [K.gemm(fdx[i:i + _lib.MAX_GROUPS], trans=2, res_acts=fra[i:i + _lib.MAX_GROUPS]) for i in range(0, len(fdx), _lib.MAX_GROUPS)]
        res['f'][0] = [t for g in res['f'][0] for t in g]
    if side is not None:
        cur.wait_stream(side)
    elif fdw:
        res['f'][1] = [K.gemm(fdw[i:i + _lib.MAX_GROUPS], trans=3, colsums=fdb[i:i + _lib.MAX_GROUPS]) for i in range(0, len(fdw), _lib.MAX_GROUPS)]
        res['f'][1] = [t for g in res['f'][1] for t in g]
    if tr:
        trs = K.transpose_pad_grouped(tr)
        dxp = [(p[0], trs[p[1]], None, p[3]) for p in dxp]
        dwp = [(trs[p[0]], trs[p[1]], None, None) for p in dwp]
        if dxp and dwp and len(dxp) + len(dwp) <= _lib.MAX_GROUPS:
            both = K.gemm(dxp + dwp, res_acts=tra + [0] * len(dwp))            # dgrad and wgrad problems of this backward in ONE grouped launch
            res['t'][0], res['t'][1] = both[:len(dxp)], both[len(dxp):]
        else:
            res['t'][0] = K.gemm(dxp, res_acts=tra) if dxp else []
            res['t'][1] = K.gemm(dwp) if dwp else []
    outx = [None if sx is None else res[sx[0]][0][sx[1]] for sx, _ in slots]
    outw = [None if sw is None else res[sw[0]][1][sw[1]] for _, sw in slots]
    return outx, outw, dbs


WGRAD_SPLIT = __import__('os').environ.get('DSRG_WGRAD_SPLIT', '1') == '1'
WGRAD_STREAM = __import__('os').environ.get('DSRG_WGRAD_STREAM', '1') == '1'
_WGRAD_SIDE = {}


def _wgrad_side_stream(cur):
    """one side stream per calling stream (a captured step and eager code must not share one)"""
    s = _WGRAD_SIDE.get(cur.cuda_stream)
    if s is None:
        s = _WGRAD_SIDE[cur.cuda_stream] = torch.cuda.Stream(device=cur.device)
    return s


def _wgrad_splits(M, N, Kd):
    """K-split factor of dW [N, Kd] = dY^T.X over M token rows: 1 unless the launch would leave most of the chip idle for a
    long time - fewer than 128 tiles of 64x64 and at least 4096 rows (the object-token encoder's weights: 4-96 tiles, 14,796
    rows, 80-96 us per launch on 4-96 workgroups, profiles/r03n_c4_pmc_fetch.csv).  Each slice gets >= 1024 rows."""
    if not WGRAD_SPLIT or M < 4096 or N % 64 or Kd % 4:
        return 1
    tiles = ((N + 63) // 64) * ((Kd + 63) // 64)
    if tiles >= 128:
        return 1
    return max(1, min(_lib.MAX_GROUPS, M // 1024))


def _wgrad_split(dy, x, db):
    """dW = dY^T.X as S problems of one grouped launch over row slices (fp32 partial tiles) + one dsrg_sum_slabs; the bias
    gradient (column sums of dY) the same way."""
    M, N = dy.shape
    Kd = x.shape[1]
    S = _wgrad_splits(M, N, Kd)
    step = ((M + S - 1) // S + 63) // 64 * 64
    cuts = [(r, min(M, r + step)) for r in range(0, M, step)]
    part = torch.empty(len(cuts), N, Kd, device=dy.device, dtype=torch.float32)
    dbp = torch.empty(len(cuts), N, device=dy.device, dtype=torch.float32) if db is not None else None
    K.gemm([(dy[a:b], x[a:b], None, None) for a, b in cuts], out_f32=True, trans=3, outs=[part[i] for i in range(len(cuts))],
           colsums=None if db is None else [dbp[i] for i in range(len(cuts))])
    if db is not None:
        K.sum_slabs(dbp, out=db)
    return K.sum_slabs(part, out_dtype=dy.dtype)


class _LinearFn(torch.autograd.Function):
    """y_i = x_i W_i^T + b_i (+ r_i), grouped.  Backward: dX = dY.W and dW = dY^T.X on the MFMA GEMM (transposed
    operands, kernels.transpose_pad), db = column sums, dr = dY."""

    @staticmethod
    def forward(ctx, n, out_f32, *flat):
        probs = [tuple(flat[4 * i:4 * i + 4]) for i in range(n)]
        ctx.n = n
        ctx.shapes = [p[0].shape for p in probs]
        ctx.rdt = [None if p[3] is None else p[3].dtype for p in probs]
        ctx.hasb = [p[2] is not None for p in probs]
        ctx.save_for_backward(*[t for p in probs for t in (p[0], p[1])])
        with torch.no_grad():
            return tuple(K.gemm(probs, ACT_NONE, out_f32))

    @staticmethod
    def backward(ctx, *dys):
        sv = ctx.saved_tensors
        items, res = [], [None, None]
        for i in range(ctx.n):
            x, w = sv[2 * i], sv[2 * i + 1]
            dy = dys[i]
            if dy is None:
                items.append(None)
                continue
            dy2 = _as2d(K.rows_ok(dy))
            if dy2.dtype != w.dtype:
                dy2 = dy2.to(w.dtype)
            items.append((_as2d(x), w, dy2, ctx.needs_input_grad[2 + 4 * i], ctx.needs_input_grad[3 + 4 * i],
                          ctx.hasb[i] and ctx.needs_input_grad[4 + 4 * i]))
        live = [it for it in items if it is not None]
        dxs, dws, dbs = _gemm_grads(live) if live else ([], [], [])
        for i in range(ctx.n):
            if items[i] is None:
                res += [None, None, None, None]
                continue
            dx, dw, db = dxs.pop(0), dws.pop(0), dbs.pop(0)
            dr = None
            if ctx.rdt[i] is not None and ctx.needs_input_grad[5 + 4 * i]:
                dr = dys[i].to(ctx.rdt[i])
            res += [dx.view(ctx.shapes[i]) if dx is not None else None, dw, db, dr]
        return tuple(res)


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, act):
        ctx.act = act
        ctx.save_for_backward(z)
        return K.act_fwd(z, act)

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        return K.act_bwd(z, dy.to(z.dtype), ctx.act), None


def activation(z, act):
    if act == ACT_NONE:
        return z
    if torch.is_grad_enabled() and z.requires_grad:
        return _ActFn.apply(z, act)
    return K.act_fwd(z, act)


def _needs_grad(ts):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


def linear_grouped(problems, act=ACT_NONE, out_f32=False):
    """problems: list of (x, w, bias|None, residual|None) -> list of outputs; one kernel launch.
    Inference: the activation is fused in the GEMM epilogue.  Under autograd the pre-activation is kept (the GEMM
    runs with no activation and a separate element-wise kernel applies it) so the backward needs no recompute."""
    flat = [t for p in problems for t in p]
    if not _needs_grad(flat):
        return K.gemm(problems, act, out_f32)
    if act != ACT_NONE and any(p[3] is not None for p in problems):
        raise _lib.DsrgHipError('linear_grouped: activation together with a residual is not used on this path')
    # out_f32: the pre-activation stays fp32 and the activation runs on it, as the fused epilogue of the no-grad path
    # does on the fp32 accumulator (train-mode and eval-mode head scores then agree)
    zs = _LinearFn.apply(len(problems), out_f32, *flat)
    if act == ACT_NONE:
        return list(zs)
    return [activation(z, act) for z in zs]


def linear(x, w, b=None, act=ACT_NONE, residual=None, out_f32=False):
    return linear_grouped([(x, w, b, residual)], act, out_f32)[0]


# ----------------------------------------------------------------------------- dense (+dropout) + residual + LayerNorm
FUSED_HIDDEN_DROPOUT = __import__('os').environ.get('DSRG_FUSED_HIDDEN_DROPOUT', '1') == '1'    # 0: torch.native_dropout + mask tensors (A/B)


class _LinearLNFn(torch.autograd.Function):
    """y_i = LN(drop(x_i W_i^T + b_i) + r_i) * gamma_i + beta_i, grouped (BertSelfOutput / BertOutput and every
    Linear->LayerNorm pair).  The fp32 pre-LN sum is kept for the backward; backward = LN-backward kernel, then the
    dgrad / wgrad GEMMs and the bias column sums."""

    @staticmethod
    def forward(ctx, n, eps, p_drop, *flat):
        probs = [tuple(flat[6 * i:6 * i + 6]) for i in range(n)]      # x, w, b, r, gamma, beta
        ctx.n, ctx.eps = n, eps
        ctx.shapes = [p[0].shape for p in probs]
        with torch.no_grad():
            masks = [None] * n
            ctx.seed = 0
            if p_drop > 0 and FUSED_HIDDEN_DROPOUT:
                # dropout inside the LayerNorm kernel (counter-based mask, recomputed in backward): z stays the un-dropped dense output
                ctx.seed = dropout_seed()
                zs = K.gemm([(p[0], p[1], p[2], None) for p in probs], ACT_NONE, True)
                ys = K.layernorm_grouped([(z, p[4], p[5], p[3]) for z, p in zip(zs, probs)], eps, probs[0][1].dtype, p_drop, ctx.seed)
            elif p_drop > 0:
                zs = K.gemm([(p[0], p[1], p[2], None) for p in probs], ACT_NONE, True)
                masks = []
                for i, z in enumerate(zs):
                    zs[i], m = torch.native_dropout(z, p_drop, True)      # one kernel: dropped+scaled z and the bool mask
                    masks.append(m)
                ys = K.layernorm_grouped([(z, p[4], p[5], p[3]) for z, p in zip(zs, probs)], eps, probs[0][1].dtype)
            else:
                zs = K.gemm([(p[0], p[1], p[2], p[3]) for p in probs], ACT_NONE, True)
                ys = K.layernorm_grouped([(z, p[4], p[5], None) for z, p in zip(zs, probs)], eps, probs[0][1].dtype)
        ctx.drop = p_drop > 0
        ctx.p_drop = p_drop
        ctx.fused_drop = p_drop > 0 and FUSED_HIDDEN_DROPOUT
        ctx.scale = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
        ctx.hasr = [p[3] is not None for p in probs]
        ctx.hasb = [p[2] is not None for p in probs]
        saved = []
        for p, z, m in zip(probs, zs, masks):
            saved += [p[0], p[1], p[4], z] + ([p[3]] if (ctx.drop and p[3] is not None) else []) + ([m] if m is not None else [])
        ctx.save_for_backward(*saved)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        sv = list(ctx.saved_tensors)
        items, keep = [], []
        for i in range(ctx.n):
            x, w, gamma, z = sv.pop(0), sv.pop(0), sv.pop(0), sv.pop(0)
            r = sv.pop(0) if (ctx.drop and ctx.hasr[i]) else None
            m = sv.pop(0) if (ctx.drop and not ctx.fused_drop) else None
            dy = dys[i]
            if dy is None:
                items.append(None); keep.append(None)
                continue
            dy = dy if dy.dtype == w.dtype else dy.to(w.dtype)
            if ctx.fused_drop:
                dz, dlin, dg, dbt = K.layernorm_bwd(z, r, gamma, dy, ctx.eps, ctx.p_drop, ctx.seed, i,
                                                    want_dz=ctx.hasr[i] and ctx.needs_input_grad[6 + 6 * i])
            else:
                dz, dg, dbt = K.layernorm_bwd(z, r, gamma, dy, ctx.eps)
                dlin = dz if m is None else torch.ops.aten.native_dropout_backward(dz, m, ctx.scale)
            keep.append((dz, dg, dbt, dlin))
            items.append((_as2d(x), w, _as2d(dlin), ctx.needs_input_grad[3 + 6 * i], ctx.needs_input_grad[4 + 6 * i],
                          ctx.hasb[i] and ctx.needs_input_grad[5 + 6 * i]))
        live = [it for it in items if it is not None]
        dxs, dws, dbs = _gemm_grads(live) if live else ([], [], [])
        res = [None, None, None]
        for i in range(ctx.n):
            if items[i] is None:
                res += [None] * 6
                continue
            dz, dg, dbt, dlin = keep[i]
            dx, dw, db = dxs.pop(0), dws.pop(0), dbs.pop(0)
            dr = dz if (ctx.hasr[i] and ctx.needs_input_grad[6 + 6 * i]) else None
            res += [dx.view(ctx.shapes[i]) if dx is not None else None, dw, db, dr, dg, dbt]
        return tuple(res)


class _FFNFn(torch.autograd.Function):
    """y_i = LN(drop(act(x_i W1_i^T + b1_i) W2_i^T + b2_i) + r_i) * gamma_i + beta_i, grouped: BertIntermediate + BertOutput
    (vilmodel.py:190-217) as ONE autograd node.  Forward: the first GEMM's epilogue writes act(z) and z (c_pre); backward: the
    second layer's dgrad leaves as (dY W2) * act'(z) (res_act): no activation kernel in either direction, and one node
    instead of three."""

    @staticmethod
    def forward(ctx, n, eps, p_drop, act, *flat):
        probs = [tuple(flat[8 * i:8 * i + 8]) for i in range(n)]      # x, w1, b1, w2, b2, r, gamma, beta
        ctx.n, ctx.eps, ctx.act, ctx.p_drop = n, eps, act, p_drop
        ctx.shapes = [p[0].shape for p in probs]
        ctx.seed = dropout_seed() if p_drop > 0 else 0
        with torch.no_grad():
            hs, zs = K.gemm([(p[0], p[1], p[2], None) for p in probs], act, False, want_pre=True)
            if p_drop > 0:
                z2 = K.gemm([(h, p[3], p[4], None) for h, p in zip(hs, probs)], ACT_NONE, True)
                ys = K.layernorm_grouped([(z, p[6], p[7], p[5]) for z, p in zip(z2, probs)], eps, probs[0][1].dtype, p_drop, ctx.seed)
            else:
                z2 = K.gemm([(h, p[3], p[4], p[5]) for h, p in zip(hs, probs)], ACT_NONE, True)
                ys = K.layernorm_grouped([(z, p[6], p[7], None) for z, p in zip(z2, probs)], eps, probs[0][1].dtype)
        ctx.hasr = [p[5] is not None for p in probs]
        ctx.same = [p[5] is p[0] for p in probs]       # the residual IS the block input (BertOutput): one gradient, summed in the dgrad epilogue
        ctx.hasb = [(p[2] is not None, p[4] is not None) for p in probs]
        saved = []
        for p, z, h, zz in zip(probs, zs, hs, z2):
            saved += [p[0], p[1], p[3], p[6], z, h, zz] + ([p[5]] if (p_drop > 0 and p[5] is not None) else [])
        ctx.save_for_backward(*saved)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        sv = list(ctx.saved_tensors)
        it2, it1, keep = [], [], []
        for i in range(ctx.n):
            x, w1, w2, gamma, z, h, z2 = (sv.pop(0) for _ in range(7))
            r = sv.pop(0) if (ctx.p_drop > 0 and ctx.hasr[i]) else None
            dy = dys[i]
            if dy is None:
                keep.append(None)
                continue
            dy = dy if dy.dtype == w1.dtype else dy.to(w1.dtype)
            ni = ctx.needs_input_grad
            want_dr = ctx.hasr[i] and ni[9 + 8 * i]
            if ctx.p_drop > 0:
                dz, dlin, dg, dbt = K.layernorm_bwd(z2, r, gamma, dy, ctx.eps, ctx.p_drop, ctx.seed, i, want_dz=want_dr)
            else:
                dz, dg, dbt = K.layernorm_bwd(z2, None, gamma, dy, ctx.eps)
                dlin = dz
            keep.append((x, w1, z, dz if want_dr else None, dg, dbt))
            # second layer: dW2 = dlin^T h, db2, and dZ = (dlin W2) * act'(z) straight out of the dgrad epilogue
            it2.append((_as2d(h), w2, _as2d(dlin), True, ni[7 + 8 * i], ctx.hasb[i][1] and ni[8 + 8 * i], _as2d(z), ctx.act))
        res = [None, None, None, None]
        if not it2:
            return tuple(res + [None] * (8 * ctx.n))
        dzs, dw2s, db2s = _gemm_grads(it2)
        merged = set()
        live = [k for k in keep if k is not None]
        for (x, w1, z, dr, dg, dbt), dz1, i in zip(live, dzs, [i for i, k in enumerate(keep) if k is not None]):
            ni = ctx.needs_input_grad
            if ctx.same[i] and dr is not None and ni[4 + 8 * i] and dr.dtype == dz1.dtype:
                # dX = dZ1 W1 + dR out of one epilogue (torch.autograd would add the two with a kernel of its own)
                it1.append((_as2d(x), w1, dz1, True, ni[5 + 8 * i], ctx.hasb[i][0] and ni[6 + 8 * i], _as2d(dr), 0))
                merged.add(i)
            else:
                it1.append((_as2d(x), w1, dz1, ni[4 + 8 * i], ni[5 + 8 * i], ctx.hasb[i][0] and ni[6 + 8 * i]))
        dxs, dw1s, db1s = _gemm_grads(it1)
        j = 0
        for i in range(ctx.n):
            if keep[i] is None:
                res += [None] * 8
                continue
            x, w1, z, dr, dg, dbt = keep[i]
            dx = dxs[j]
            res += [dx.view(ctx.shapes[i]) if dx is not None else None, dw1s[j], db1s[j], dw2s[j], db2s[j], None if i in merged else dr, dg, dbt]
            j += 1
        return tuple(res)


class _AttnBlockFn(torch.autograd.Function):
    """y_i = LN(drop(attention(..) Wo_i^T + bo_i) + x_i), grouped: BertAttention / BertXAttention + BertSelfOutput
    (vilmodel.py:111-177, 349-392) as ONE autograd node instead of three (projection, attention, dense+LN).
      self  (kv_i is None): q | k | v = x_i Wq_i^T + bq_i with Wq_i the packed [3H, H] projection;
      cross (kv_i given)  : q = x_i Wq_i^T + bq_i, k | v = kv_i (a packed [B, Nk, 2H] projection of the other modality).
    Backward: LayerNorm backward (hidden dropout recomputed), the output projection's dgrad / wgrad, the attention backward
    writing into packed gradients, and the input projection's dgrad with the residual's gradient added in its epilogue."""

    @staticmethod
    def forward(ctx, n, cfg, *flat):
        nh, mask_mode, scale, p_attn, p_hid, eps = cfg
        H = nh * 64
        probs = [tuple(flat[10 * i:10 * i + 10]) for i in range(n)]    # x, wq, bq, wo, bo, gamma, beta, kv, kmask, dists
        sw, sb = flat[10 * n], flat[10 * n + 1]
        ctx.n, ctx.cfg = n, cfg
        ctx.hid_seed = dropout_seed() if p_hid > 0 else 0
        ctx.att_seeds = [dropout_seed() if p_attn > 0 else 0 for _ in range(n)]
        with torch.no_grad():
            qs = K.gemm([(p[0], p[1], p[2], None) for p in probs])
            outs, lses = [], []
            for q, p, sd in zip(qs, probs, ctx.att_seeds):
                kv, km, d = p[7], p[8], p[9]
                qq, k, v = (q[..., :H], q[..., H:2 * H], q[..., 2 * H:]) if kv is None else (q, kv[..., :H], kv[..., H:])
                o, l = K.attention(qq, k, v, nh, km, mask_mode if km is not None else MASK_NONE, d, sw if d is not None else None,
                                   sb if d is not None else None, scale, want_lse=True, p_drop=p_attn, seed=sd)
                outs.append(o); lses.append(l)
            if p_hid > 0:
                z2 = K.gemm([(o, p[3], p[4], None) for o, p in zip(outs, probs)], ACT_NONE, True)
                ys = K.layernorm_grouped([(z, p[5], p[6], p[0]) for z, p in zip(z2, probs)], eps, probs[0][1].dtype, p_hid, ctx.hid_seed)
            else:
                z2 = K.gemm([(o, p[3], p[4], p[0]) for o, p in zip(outs, probs)], ACT_NONE, True)
                ys = K.layernorm_grouped([(z, p[5], p[6], None) for z, p in zip(z2, probs)], eps, probs[0][1].dtype)
        ctx.hasb = [(p[2] is not None, p[4] is not None) for p in probs]
        ctx.cross = [p[7] is not None for p in probs]
        ctx.hasd = [p[9] is not None for p in probs]
        saved = [sw, sb]
        for p, q, o, l, z in zip(probs, qs, outs, lses, z2):
            saved += [p[0], p[1], p[3], p[5], p[7], p[8], p[9], q, o, l, z]
        ctx.save_for_backward(*saved)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        nh, mask_mode, scale, p_attn, p_hid, eps = ctx.cfg
        H = nh * 64
        sv = list(ctx.saved_tensors)
        sw, sb = sv.pop(0), sv.pop(0)
        ni = ctx.needs_input_grad
        st, it_o = [], []
        for i in range(ctx.n):
            x, wq, wo, gamma, kv, km, d, q, o, l, z2 = (sv.pop(0) for _ in range(11))
            dy = dys[i]
            if dy is None:
                st.append(None)
                continue
            dy = dy if dy.dtype == wq.dtype else dy.to(wq.dtype)
            if p_hid > 0:
                dz, dlin, dg, dbt = K.layernorm_bwd(z2, x, gamma, dy, eps, p_hid, ctx.hid_seed, i, want_dz=ni[2 + 10 * i])
            else:
                dz, dg, dbt = K.layernorm_bwd(z2, None, gamma, dy, eps)
                dlin = dz
            st.append([x, wq, kv, km, d, q, o, l, dz if ni[2 + 10 * i] else None, dg, dbt])
            it_o.append((_as2d(o), wo, _as2d(dlin), True, ni[5 + 10 * i], ctx.hasb[i][1] and ni[6 + 10 * i]))
        res = [None, None]
        if not it_o:
            return tuple(res + [None] * (10 * ctx.n + 2))
        douts, dwos, dbos = _gemm_grads(it_o)
        it_q, dkvs, dsp_sum = [], [], None
        j = 0
        for i in range(ctx.n):
            if st[i] is None:
                dkvs.append(None)
                continue
            x, wq, kv, km, d, q, o, l, dz, dg, dbt = st[i]
            do = douts[j].view(o.shape)
            dqp = torch.empty(q.shape, device=q.device, dtype=q.dtype)
            if kv is None:
                qq, k, v = q[..., :H], q[..., H:2 * H], q[..., 2 * H:]
                grads, dkv = (dqp[..., :H], dqp[..., H:2 * H], dqp[..., 2 * H:]), None
            else:
                qq, k, v = q, kv[..., :H], kv[..., H:]
                dkv = torch.empty(kv.shape, device=kv.device, dtype=kv.dtype)
                grads = (dqp, dkv[..., :H], dkv[..., H:])
            _, _, _, dsp = K.attention_bwd(qq, k, v, o, do, l, nh, km, mask_mode if km is not None else MASK_NONE, d, sw if d is not None else None,
                                           sb if d is not None else None, scale, p_attn, ctx.att_seeds[i], grads=grads)
            if dsp is not None:
                dsp_sum = dsp if dsp_sum is None else dsp_sum + dsp
            dkvs.append(dkv)
            if dz is not None and ni[2 + 10 * i] and dz.dtype == dqp.dtype:
                it_q.append((_as2d(x), wq, _as2d(dqp), True, ni[3 + 10 * i], ctx.hasb[i][0] and ni[4 + 10 * i], _as2d(dz), 0))    # dX = dQ Wq + dR
            else:
                it_q.append((_as2d(x), wq, _as2d(dqp), ni[2 + 10 * i], ni[3 + 10 * i], ctx.hasb[i][0] and ni[4 + 10 * i]))
            j += 1
        dxs, dwqs, dbqs = _gemm_grads(it_q)
        j = 0
        for i in range(ctx.n):
            if st[i] is None:
                res += [None] * 10
                continue
            x, dg, dbt = st[i][0], st[i][9], st[i][10]
            dx = dxs[j]
            res += [dx.view(x.shape) if dx is not None else None, dwqs[j], dbqs[j], dwos[j], dbos[j], dg, dbt, dkvs[i], None, None]
            j += 1
        dsw = dsb = None
        if dsp_sum is not None and sw is not None:
            dsw, dsb = dsp_sum[0:1].reshape(sw.shape).to(sw.dtype), dsp_sum[1:2].reshape(sb.shape).to(sb.dtype)
        return tuple(res + [dsw, dsb])


FUSED_ATTN_BLOCK = __import__('os').environ.get('DSRG_FUSED_ATTN_BLOCK', '1') == '1'     # 0: projection / attention / dense+LN nodes (A/B)


def attn_block_grouped(problems, nh, mask_mode, eps, p_attn=0.0, p_hid=0.0, sprel_w=None, sprel_b=None, scale=0.125):
    """problems: list of (x, wq, bq, wo, bo, gamma, beta, kv | None, kmask | None, dists | None) ->
    list of LN(drop(attention Wo^T + bo) + x); see _AttnBlockFn.  Without autograd: the grouped launches of the parts."""
    flat = [t for p in problems for t in p] + [sprel_w, sprel_b]
    if FUSED_ATTN_BLOCK and _needs_grad(flat) and all(p[0].dim() == 3 and (p[7] is None or p[7].stride(-1) == 1) for p in problems):
        return list(_AttnBlockFn.apply(len(problems), (nh, mask_mode, scale, float(p_attn), float(p_hid), eps), *flat))
    q = linear_grouped([(p[0], p[1], p[2], None) for p in problems])
    a = attention_grouped_packed([(None if p[7] is None else q[j], q[j] if p[7] is None else p[7], p[8], p[9]) for j, p in enumerate(problems)],
                                 nh, mask_mode, sprel_w, sprel_b, scale, p_attn)
    return linear_ln_grouped([(a[j], p[3], p[4], p[0], p[5], p[6]) for j, p in enumerate(problems)], eps, p_hid)


FUSED_FFN = __import__('os').environ.get('DSRG_FUSED_FFN', '1') == '1'       # 0: linear + activation + linear_ln nodes (A/B)


def ffn_grouped(problems, eps, p_drop=0.0, act=ACT_GELU, outs=None, aux=None):
    """problems: list of (x, w1, b1, w2, b2, residual|None, gamma, beta) -> list of LN(drop(act(x W1^T + b1) W2^T + b2) + residual).
    outs / aux: output placement of the final LayerNorm (kernels.layernorm_grouped; inference only)."""
    flat = [t for p in problems for t in p]
    if FUSED_FFN and _needs_grad(flat) and all(p[1].shape[0] % 8 == 0 for p in problems):
        _no_placement(outs, aux)
        return list(_FFNFn.apply(len(problems), eps, float(p_drop), act, *flat))
    inter = linear_grouped([(p[0], p[1], p[2], None) for p in problems], act=act)
    return linear_ln_grouped([(h, p[3], p[4], p[5], p[6], p[7]) for h, p in zip(inter, problems)], eps, p_drop, outs, aux)


def _no_placement(outs, aux):
    if outs is not None or aux is not None:
        raise _lib.DsrgHipError('output placement (outs / aux) is an inference-path feature: no gradient flows through it')


def linear_ln_grouped(problems, eps, p_drop=0.0, outs=None, aux=None):
    """problems: list of (x, w, b, residual|None, gamma, beta) -> list of LN(drop(xW^T+b) + residual).
    outs / aux: output placement of the LayerNorm (kernels.layernorm_grouped; inference only)."""
    flat = [t for p in problems for t in p]
    if not _needs_grad(flat) and p_drop == 0.0:
        zs = K.gemm([(p[0], p[1], p[2], p[3]) for p in problems], ACT_NONE, True)
        return K.layernorm_grouped([(z, p[4], p[5], None) for z, p in zip(zs, problems)], eps, problems[0][1].dtype, outs=outs, aux=aux)
    _no_placement(outs, aux)
    return list(_LinearLNFn.apply(len(problems), eps, float(p_drop), *flat))


def linear_ln(x, w, b, gamma, beta, eps, residual=None, p_drop=0.0, out=None, aux=None):
    return linear_ln_grouped([(x, w, b, residual, gamma, beta)], eps, p_drop, None if out is None else [out], None if aux is None else [aux])[0]


# ----------------------------------------------------------------------------- attention
class _DropoutAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, p, seed):
        ctx.cfg = (p, seed, residual is not None)
        return K.dropout_add(x, residual, p, seed)

    @staticmethod
    def backward(ctx, dy):
        p, seed, has_res = ctx.cfg
        dy = dy if dy.is_contiguous() else dy.contiguous()
        return K.dropout_add(dy, None, p, seed) if ctx.needs_input_grad[0] else None, dy if (has_res and ctx.needs_input_grad[1]) else None, None, None


def dropout_add(x, p_drop, residual=None, training=True):
    """residual + nn.Dropout(p)(x): one HIP kernel forward and one backward, no mask tensor (counter-hash mask, as the
    attention and LayerNorm dropouts); identity / plain add when not training or p = 0."""
    if not training or p_drop <= 0.0:
        return x if residual is None else x + residual
    seed = dropout_seed()
    if _needs_grad((x, residual)):
        return _DropoutAddFn.apply(x, residual, float(p_drop), seed)
    return K.dropout_add(x, residual, float(p_drop), seed)


def dropout_seed():
    """63-bit seed for the in-kernel attention dropout, drawn from torch's CPU generator (follows torch.manual_seed)."""
    return int(torch.empty((), dtype=torch.int64).random_().item())


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, kmask, dists, sw, sb, nh, mask_mode, scale, p_drop, seed):
        out, lse = K.attention(q, k, v, nh, kmask, mask_mode if kmask is not None else MASK_NONE, dists, sw, sb, scale, want_lse=True,
                               p_drop=p_drop, seed=seed)
        ctx.cfg = (nh, mask_mode, scale, p_drop, seed)
        ctx.save_for_backward(q, k, v, out, lse, kmask, dists, sw, sb)
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, out, lse, kmask, dists, sw, sb = ctx.saved_tensors
        nh, mask_mode, scale, p_drop, seed = ctx.cfg
        dq, dk, dv, dsp = K.attention_bwd(q, k, v, out, do.to(q.dtype), lse, nh, kmask, mask_mode, dists, sw, sb, scale, p_drop, seed)
        dsw = dsb = None
        if dsp is not None:
            dsw, dsb = dsp[0:1].reshape(sw.shape).to(sw.dtype), dsp[1:2].reshape(sb.shape).to(sb.dtype)
        return dq, dk, dv, None, None, dsw, dsb, None, None, None, None, None


def attention(q, k, v, nh, kmask=None, mask_mode=MASK_NONE, dists=None, sprel_w=None, sprel_b=None, scale=0.125, p_drop=0.0, seed=None):
    """Fused attention.  p_drop > 0: dropout on the attention probabilities inside the kernel (training)."""
    if p_drop > 0.0 and seed is None:
        seed = dropout_seed()
    if _needs_grad((q, k, v, sprel_w, sprel_b)):
        return _AttentionFn.apply(q, k, v, kmask, dists, sprel_w, sprel_b, nh, mask_mode, scale, float(p_drop), seed or 0)
    return K.attention(q, k, v, nh, kmask, mask_mode if kmask is not None else MASK_NONE, dists, sprel_w, sprel_b, scale,
                       p_drop=p_drop, seed=seed or 0)


class _AttentionPackedFn(torch.autograd.Function):
    """Attention on a packed projection: `packed` is [B,N,3H] = q|k|v (self attention, q is None) or [B,Nk,2H] = k|v (cross
    attention, separate q).  The backward kernels write dq / dk / dv straight into the column blocks of ONE packed gradient
    (dsrg_attention_bwd_strided): with plain slices torch.autograd builds that gradient from a zero-fill, a copy and an add
    per slice - ~250 of the ~1130 launches of a pre-training step."""

    @staticmethod
    def forward(ctx, packed, q, kmask, dists, sw, sb, nh, mask_mode, scale, p_drop, seed):
        H = nh * 64
        if q is None:
            qq, k, v = packed[..., :H], packed[..., H:2 * H], packed[..., 2 * H:]
        else:
            qq, k, v = q, packed[..., :H], packed[..., H:]
        out, lse = K.attention(qq, k, v, nh, kmask, mask_mode if kmask is not None else MASK_NONE, dists, sw, sb, scale, want_lse=True,
                               p_drop=p_drop, seed=seed)
        ctx.cfg = (nh, mask_mode, scale, p_drop, seed)
        ctx.save_for_backward(packed, q, out, lse, kmask, dists, sw, sb)
        return out

    @staticmethod
    def backward(ctx, do):
        packed, q, out, lse, kmask, dists, sw, sb = ctx.saved_tensors
        nh, mask_mode, scale, p_drop, seed = ctx.cfg
        H = nh * 64
        dp = torch.empty(packed.shape, device=packed.device, dtype=packed.dtype)
        if q is None:
            qq, k, v = packed[..., :H], packed[..., H:2 * H], packed[..., 2 * H:]
            grads = (dp[..., :H], dp[..., H:2 * H], dp[..., 2 * H:])
        else:
            qq, k, v = q, packed[..., :H], packed[..., H:]
            grads = (torch.empty(q.shape, device=q.device, dtype=q.dtype), dp[..., :H], dp[..., H:])
        _, _, _, dsp = K.attention_bwd(qq, k, v, out, do.to(qq.dtype), lse, nh, kmask, mask_mode, dists, sw, sb, scale, p_drop, seed, grads=grads)
        dsw = dsb = None
        if dsp is not None:
            dsw, dsb = dsp[0:1].reshape(sw.shape).to(sw.dtype), dsp[1:2].reshape(sb.shape).to(sb.dtype)
        return dp, (grads[0] if q is not None else None), None, None, dsw, dsb, None, None, None, None, None


def attention_packed(packed, nh, q=None, kmask=None, mask_mode=MASK_NONE, dists=None, sprel_w=None, sprel_b=None, scale=0.125, p_drop=0.0,
                     seed=None):
    """attention(q, k, v) with q|k|v (q is None) or k|v the column blocks of `packed` (see _AttentionPackedFn)."""
    H = nh * 64
    if packed.shape[-1] != (3 if q is None else 2) * H:
        raise _lib.DsrgHipError('attention_packed: packed last dim %d is not %d x nh*64' % (packed.shape[-1], 3 if q is None else 2))
    if p_drop > 0.0 and seed is None:
        seed = dropout_seed()
    if _needs_grad((packed, q, sprel_w, sprel_b)) and packed.stride(-1) == 1:
        return _AttentionPackedFn.apply(packed, q, kmask, dists, sprel_w, sprel_b, nh, mask_mode, scale, float(p_drop), seed or 0)
    if q is None:
        return attention(packed[..., :H], packed[..., H:2 * H], packed[..., 2 * H:], nh, kmask, mask_mode, dists, sprel_w, sprel_b, scale, p_drop, seed)
    return attention(q, packed[..., :H], packed[..., H:], nh, kmask, mask_mode, dists, sprel_w, sprel_b, scale, p_drop, seed)


def attention_grouped_packed(problems, nh, mask_mode=MASK_NONE, sprel_w=None, sprel_b=None, scale=0.125, p_drop=0.0):
    """problems: list of (q | None, packed, kmask, dists | None) - attention_grouped for packed projections: one launch when no
    gradient is needed, one attention_packed call per problem otherwise."""
    H = nh * 64
    flat = [t for p in problems for t in p[:2]] + [sprel_w, sprel_b]
    if _needs_grad(flat) or len(problems) > _lib.ATTN_MAX_GROUPS:
        return [attention_packed(pk, nh, q, km, mask_mode, d, sprel_w if d is not None else None, sprel_b if d is not None else None, scale,
                                 p_drop) for q, pk, km, d in problems]
    sl = [((pk[..., :H], pk[..., H:2 * H], pk[..., 2 * H:]) if q is None else (q, pk[..., :H], pk[..., H:])) + (km, d) for q, pk, km, d in problems]
    return K.attention_grouped(sl, nh, mask_mode, sprel_w, sprel_b, scale, p_drop, dropout_seed() if p_drop > 0.0 else 0)


class _SplitColsFn(torch.autograd.Function):
    """x [.., sum(widths)] -> column blocks (views); backward = ONE concatenation of the blocks' gradients (plain slices
    cost a zero-fill + copy per slice and an add per pair in backward)."""

    @staticmethod
    def forward(ctx, x, widths):
        ctx.widths, ctx.meta = widths, (x.shape, x.dtype, x.device)
        ctx.set_materialize_grads(False)
        outs, o = [], 0
        for w in widths:
            outs.append(x[..., o:o + w]); o += w
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        shape, dtype, device = ctx.meta
        if all(g is None for g in gs):
            return None, None
        gs = [g.to(dtype) if g is not None else torch.zeros(shape[:-1] + (w,), dtype=dtype, device=device) for g, w in zip(gs, ctx.widths)]
        return torch.cat(gs, -1), None


def split_cols(x, widths):
    """column blocks of x (views), with a one-kernel backward under autograd"""
    widths = tuple(int(w) for w in widths)
    if sum(widths) != x.shape[-1]:
        raise _lib.DsrgHipError('split_cols: widths %s do not add up to %d' % (widths, x.shape[-1]))
    if _needs_grad((x,)):
        return list(_SplitColsFn.apply(x, widths))
    outs, o = [], 0
    for w in widths:
        outs.append(x[..., o:o + w]); o += w
    return outs


def attention_grouped(problems, nh, mask_mode=MASK_NONE, sprel_w=None, sprel_b=None, scale=0.125, p_drop=0.0):
    """problems: list of (q, k, v, kmask, dists|None): one launch when no gradient is needed (the two branches of an
    x-layer), one differentiable call per problem otherwise."""
    flat = [t for p in problems for t in p[:3]] + [sprel_w, sprel_b]
    if _needs_grad(flat) or len(problems) > _lib.ATTN_MAX_GROUPS:
        return [attention(q, k, v, nh, km, mask_mode, d, sprel_w if d is not None else None, sprel_b if d is not None else None, scale,
                          p_drop) for q, k, v, km, d in problems]
    return K.attention_grouped(problems, nh, mask_mode, sprel_w, sprel_b, scale, p_drop, dropout_seed() if p_drop > 0.0 else 0)


# ----------------------------------------------------------------------------- layernorm
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, eps, odt):
        ctx.eps = eps
        ctx.save_for_backward(x, gamma, residual)
        return K.layernorm(x, gamma, beta, eps, residual, odt)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, residual = ctx.saved_tensors
        dz, dg, db = K.layernorm_bwd(x, residual, gamma, dy, ctx.eps)
        dx = dz if dz.dtype == x.dtype else dz.to(x.dtype)
        dr = None if residual is None else (dz if dz.dtype == residual.dtype else dz.to(residual.dtype))
        return dx, dg, db, dr, None, None


def layernorm(x, gamma, beta, eps, residual=None, out_dtype=None, out=None):
    odt = out_dtype or (residual.dtype if residual is not None else x.dtype)
    if out is None and _needs_grad((x, gamma, beta, residual)):
        return _LayerNormFn.apply(x, gamma, beta, residual, eps, odt)
    return K.layernorm(x, gamma, beta, eps, residual, odt, out)


# ----------------------------------------------------------------------------- token embedding
class _EmbedLnFn(torch.autograd.Function):
    """out = LNo(base + LNf(Wf.feat + bf) + E1[i1] + E2[i2] + vec); backward = dsrg_embed_ln_bwd (one kernel)."""

    @staticmethod
    def forward(ctx, cfg, base, feat, Wf, bf, lnf_g, lnf_b, E1, E2, vec, lno_g, lno_b):
        M, H, dtype, device, idx1, idx2, eps, _ = cfg
        ctx.cfg = cfg
        ctx.save_for_backward(base, feat, Wf, bf, lnf_g, lnf_b, E1, E2, vec, lno_g, lno_b)
        return K.embed_ln(M, H, dtype, device, base, feat, Wf, bf, (lnf_g, lnf_b) if lnf_g is not None else None,
                          E1, idx1, E2, idx2, vec, (lno_g, lno_b) if lno_g is not None else None, eps)

    @staticmethod
    def backward(ctx, dout):
        M, H, dtype, device, idx1, idx2, eps, pad1 = ctx.cfg
        base, feat, Wf, bf, lnf_g, lnf_b, E1, E2, vec, lno_g, lno_b = ctx.saved_tensors
        if feat is not None and ctx.needs_input_grad[2]:
            raise _lib.DsrgHipError('embed_ln: no gradient w.r.t. the raw feature input')
        g = K.embed_ln_bwd(M, H, dout.to(dtype), base, feat, Wf, bf, (lnf_g, lnf_b) if lnf_g is not None else None, E1, idx1, E2, idx2,
                           vec, (lno_g, lno_b) if lno_g is not None else None, eps, need_dbase=ctx.needs_input_grad[1],
                           E1_padding_idx=pad1)
        db = g['base']
        if db is not None and base is not None:
            db = db.view(base.shape)
        return (None, db, None, g['Wf'], g['bf'], g['lnf_g'], g['lnf_b'], g['E1'], g['E2'], g['vec'], g['lno_g'], g['lno_b'])


def embed_ln(M, H, dtype, device, base=None, feat=None, Wf=None, bf=None, lnf=None, E1=None, idx1=None,
             E2=None, idx2=None, vec=None, lno=None, eps=1e-12, out=None, E1_padding_idx=None, seg=None, lnb=None):
    """Token-embedding sum + LayerNorm(s) (BertEmbeddings and the four visual token sites, see include/dsrg_hip.h).
    E1_padding_idx: row of E1 that receives no gradient (the word table is nn.Embedding(padding_idx=0) in the reference).
    lnb = (gamma, beta): LayerNorm on the base row before it is added (inference path, as seg)."""
    lnf_g, lnf_b = lnf if lnf is not None else (None, None)
    lno_g, lno_b = lno if lno is not None else (None, None)
    ts = (base, Wf, bf, lnf_g, lnf_b, E1, E2, vec, lno_g, lno_b)
    if _needs_grad(ts):
        if seg is not None or lnb is not None:
            raise _lib.DsrgHipError('embed_ln: the gathered base (seg) and the base LayerNorm (lnb) are inference-path features')
        if base is not None and base.dim() != 2:
            base = base.reshape(M, H)
        return _EmbedLnFn.apply((M, H, dtype, device, idx1, idx2, eps, E1_padding_idx), base, feat, Wf, bf, lnf_g, lnf_b, E1, E2, vec, lno_g, lno_b)
    return K.embed_ln(M, H, dtype, device, base, feat, Wf, bf, lnf, E1, idx1, E2, idx2, vec, lno, eps, out, seg, lnb)


class _GroupMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, G):
        ctx.G, ctx.shape = G, x.shape
        return K.group_mean(x, G)

    @staticmethod
    def backward(ctx, dy):
        return K.group_mean_bwd(dy, ctx.G).view(ctx.shape), None


def group_mean(x, G, out=None):
    if _needs_grad((x,)):
        return _GroupMeanFn.apply(x, G)
    return K.group_mean(x, G, out)


class _GaaPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w, b)
        return K.gaa_pool(x, w, b)

    @staticmethod
    def backward(ctx, dout):
        x, w, b = ctx.saved_tensors
        dx, dw, db = K.gaa_pool_bwd(x, w, b, dout)
        return dx, dw.view(w.shape), db.view(b.shape)


def gaa_pool(x, w, b, scatter=None):
    """GAA (vilmodel.py:735-740): softmax_v(tanh(x_v.w + b))-weighted sum over the views; hand-written backward.
    scatter = (buf [B,G,H], idx int64 [B]) (no-grad only): the pooled row also lands in buf[b, idx[b]]."""
    if _needs_grad((x, w, b)):
        if scatter is not None:
            raise _lib.DsrgHipError('gaa_pool: scatter is an inference-path placement (no gradient flows through it)')
        return _GaaPoolFn.apply(x if x.is_contiguous() else x.contiguous(), w, b)
    return K.gaa_pool(x, w, b, scatter)


class _HeadTailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, gamma, beta, w2, b2, eps):
        ctx.eps = eps
        ctx.save_for_backward(h, gamma, beta, w2)
        ctx.b2_shape = b2.shape
        return K.head_tail(h, gamma, beta, w2, b2, eps)

    @staticmethod
    def backward(ctx, dscore):
        h, gamma, beta, w2 = ctx.saved_tensors
        dh, dg, dbt, dw2, db2 = K.head_tail_bwd(h, gamma, beta, w2, dscore, ctx.eps)
        return dh, dg, dbt, dw2.view(w2.shape), db2.view(ctx.b2_shape), None


def head_tail(h, gamma, beta, w2, b2, eps=1e-12):
    """ClsPrediction tail (vilmodel.py:616-628): LayerNorm + Linear(H, 1) on the fp32 hidden rows."""
    if _needs_grad((h, gamma, beta, w2, b2)):
        return _HeadTailFn.apply(h, gamma, beta, w2, b2, eps)
    return K.head_tail(h, gamma, beta, w2, b2, eps)


def head_tail_grouped(items, eps=1e-12):
    """items: list of (h, gamma, beta, w2, b2): all ClsPrediction tails of a step in ONE launch (no-grad); with autograd on
    each head keeps its own node."""
    if any(_needs_grad(it) for it in items):
        return [head_tail(*it, eps) for it in items]
    return K.head_tail_grouped(items, eps)


class _NavLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, l, f, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask):
        ctx.save_for_backward(g, l, f, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask)
        ctx.set_materialize_grads(False)
        return K.nav_logits(g, l, f, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask)

    @staticmethod
    def backward(ctx, dgl, dll, dfl):
        g, l, f, gm, gv, vn, src, bw = ctx.saved_tensors
        dg, dl, df = K.nav_logits_bwd(g, l, f, gm, gv, vn, src, bw, dgl, dll, dfl)
        return dg, dl, df, None, None, None, None, None


def nav_logits(g_raw, l_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask, o_raw=None, vp_obj_masks=None):
    """dynamic fusion weight, the three masked_fill_ and the logit-fusion loop (vilmodel.py:773-805) -> global, local, fused
    [, obj_logits = o_raw.masked_fill(~vp_obj_masks, -inf) (vilmodel_reverie.py:823-825): in the same launch without autograd]."""
    if _needs_grad((g_raw, l_raw, fuse_raw, o_raw)):
        out = _NavLogitsFn.apply(g_raw, l_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask)
        return out if o_raw is None else tuple(out) + (o_raw.masked_fill(vp_obj_masks.logical_not(), -float('inf')),)
    return K.nav_logits(g_raw, l_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask, o_raw, vp_obj_masks)


class _IglBlendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aug, x, wa, ba, wo, bo):
        ctx.save_for_backward(aug, x, wa, ba, wo, bo)
        return K.igl_blend(aug, x, wa, ba, wo, bo)

    @staticmethod
    def backward(ctx, dout):
        aug, x, wa, ba, wo, bo = ctx.saved_tensors
        daug, dx, dwa, dba, dwo, dbo = K.igl_blend_bwd(aug, x, wa, ba, wo, bo, dout)
        return daug, dx, dwa.view(wa.shape), dba.view(ba.shape), dwo.view(wo.shape), dbo.view(bo.shape)


def igl_blend(aug, x, wa, ba, wo, bo):
    """IGL gate (vilmodel.py:525-528): w = sigmoid(aug.wa + ba + x.wo + bo); out = w aug + (1 - w) x."""
    if _needs_grad((aug, x, wa, ba, wo, bo)):
        c = lambda t: t if t.is_contiguous() else t.contiguous()
        return _IglBlendFn.apply(c(aug), c(x), wa, ba, wo, bo)
    return K.igl_blend(aug, x, wa, ba, wo, bo)


# ----------------------------------------------------------------------------- action selection + loss tail (agent.py:484-516)
class _ActionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        o = K.action_tail(logits, targets, None, ignore_index, want_grad=True)
        ctx.save_for_backward(o['dlogits'])
        ctx.mark_non_differentiable(o['stop_prob'], o['argmax'], o['entropy'])
        return o['loss_rows'].sum(), o['stop_prob'], o['argmax'], o['entropy']

    @staticmethod
    def backward(ctx, gl, *_):
        (d,) = ctx.saved_tensors
        return d * gl, None, None


class _CERowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        loss, lse = K.ce_rows(logits, targets, ignore_index)
        ctx.ignore = ignore_index
        ctx.save_for_backward(logits, targets, lse)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, targets, lse = ctx.saved_tensors
        return K.ce_rows_bwd(logits, targets, lse, g, ctx.ignore), None, None


def cross_entropy_rows(logits, targets, ignore_index=-100):
    """F.cross_entropy(logits, targets, ignore_index=ignore_index, reduction='none') for fp32 [B,N] logits (-inf = masked):
    one pass over each row in forward (dsrg_ce_rows_fwd), the gradient recomputed from the logits and the row log-sum-exp in
    backward (dsrg_ce_rows_bwd); no log-softmax matrix (173 x 30,522 for the MLM head) is stored."""
    lg = logits if logits.dtype == torch.float32 else logits.float()
    lg = lg if lg.stride(-1) == 1 else lg.contiguous()
    if torch.is_grad_enabled() and lg.requires_grad:
        return _CERowsFn.apply(lg, targets.contiguous(), ignore_index)
    return K.ce_rows(lg.detach(), targets.contiguous(), ignore_index)[0]


class _KLRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        loss, lse = K.kl_rows(logits, target)
        ctx.save_for_backward(logits, target, lse)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target, lse = ctx.saved_tensors
        return K.kl_rows_bwd(logits, target, lse, g), None


def kl_div_rows(logits, target):
    """F.kl_div(F.log_softmax(logits, -1), target, reduction='none').sum(1) for fp32 [R,N] (the MRC loss, pretrain_cmt.py:215-226):
    one kernel forward (dsrg_kl_rows_fwd), the gradient recomputed from the logits and the row log-sum-exp in backward."""
    lg = logits if logits.dtype == torch.float32 else logits.float()
    lg = lg if lg.stride(-1) == 1 else lg.contiguous()
    tg = target if (target.dtype == torch.float32 and target.stride(-1) == 1) else target.float().contiguous()
    if lg.shape[0] == 0:
        return lg.new_zeros((0,))
    if torch.is_grad_enabled() and lg.requires_grad:
        return _KLRowsFn.apply(lg, tg)
    return K.kl_rows(lg.detach(), tg)[0]


def action_tail(nav_logits, targets=None, u=None, ignore_index=-100):
    """One kernel for what the rollout does with a step's logits: stop probability (softmax[:,0]), greedy action, entropy,
    optionally an inverse-CDF sample for the caller's uniforms `u`, and with `targets` the summed cross-entropy
    (CrossEntropyLoss(ignore_index, reduction='sum'), differentiable w.r.t. the logits).  Returns a dict."""
    lg = nav_logits if nav_logits.dtype == torch.float32 else nav_logits.float()
    if targets is not None and torch.is_grad_enabled() and lg.requires_grad:
        loss, sp, am, ent = _ActionLossFn.apply(lg, targets, ignore_index)
        out = {'loss': loss, 'stop_prob': sp, 'argmax': am, 'entropy': ent}
        if u is not None:
            out['sample'] = K.action_tail(lg.detach(), None, u)['sample']
        return out
    o = K.action_tail(lg.detach(), targets, u, ignore_index)
    if targets is not None:
        o['loss'] = o.pop('loss_rows').sum()
    return o


# ----------------------------------------------------------------------------- segment-weighted row gather (PT aggregation)
def csr_by(keys, n, other, w, device):
    """host: entries (key_e, other_e, w_e) -> CSR over key: (seg_ptr int32 [n+1], idx int32, w fp32) on `device`."""
    import numpy as np
    keys, other, w = np.asarray(keys, np.int64), np.asarray(other, np.int64), np.asarray(w, np.float32)
    order = np.argsort(keys, kind='stable')
    ptr = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(keys, minlength=n), out=ptr[1:])
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(device)
    return t(ptr, np.int32), t(other[order], np.int32), t(w[order], np.float32)


class _SegmentWsumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, in0, in1, fwd_csr, bwd_csr, m):
        ctx.bwd_csr = bwd_csr
        ctx.n = (in0.shape[0], in1.shape[0] if in1 is not None else 0)
        ctx.shapes = (in0.shape, None if in1 is None else in1.shape)
        return K.segment_wsum(in0, in1, *fwd_csr, m)[0]

    @staticmethod
    def backward(ctx, dout):
        d0, d1 = K.segment_wsum(K.rows_ok(dout), None, *ctx.bwd_csr, ctx.n[0], ctx.n[1])
        return d0, d1, None, None, None


def segment_plan(dst, src, w, m, n, device):
    """Device-resident incidence of a segment_wsum, built once on the host: the CSR over the m destinations (forward) and
    the transposed one over the n source rows (backward).  A plan depends only on the index arrays, so a caller that
    replays the same batch (a captured training step) builds it outside the replayed region and passes it in."""
    return csr_by(dst, m, src, w, device), csr_by(src, n, dst, w, device)


def segment_wsum(in0, in1, dst, src, w, m, plan=None):
    """out[d] = sum_{e: dst_e = d} w_e * cat(in0, in1)[src_e]  (dst, src, w: host index arrays, or a `plan` from segment_plan;
    m output rows).  One kernel forward, the same kernel on the transposed incidence backward."""
    dev = in0.device
    in0 = in0.reshape(-1, in0.shape[-1])
    if in1 is not None:
        in1 = in1.reshape(-1, in1.shape[-1])
    n = in0.shape[0] + (in1.shape[0] if in1 is not None else 0)
    if plan is None:
        plan = (csr_by(dst, m, src, w, dev), csr_by(src, n, dst, w, dev) if _needs_grad((in0, in1)) else None)
    if _needs_grad((in0, in1)):
        return _SegmentWsumFn.apply(in0, in1, plan[0], plan[1], m)
    return K.segment_wsum(in0, in1, *plan[0], m)[0]
