"""Differentiable ops of the DSRG hot path.

Forward ALWAYS runs the hand-written HIP kernel (kernels.py -> libdsrg_hip.so); there is no other
forward.  When autograd needs a gradient, the op is wrapped in `_Fused`: its backward re-expresses the
op with ATen ops on the saved inputs and differentiates that (rocBLAS GEMMs + ATen elementwise).
Round-1 status: backward is correct but not yet hand-written; see DESIGN.md "Backward".
"""
import math
import torch
import torch.nn.functional as F

from . import _lib, kernels as K
from ._lib import ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH, MASK_NONE, MASK_NEG10000, MASK_NEGINF, MASK_PLUS1


class _Fused(torch.autograd.Function):
    """forward = hip_fn(*args); backward = autograd through aten_fn(*args) recomputed on saved inputs."""

    @staticmethod
    def forward(ctx, hip_fn, aten_fn, *args):
        ctx.aten_fn = aten_fn
        ctx.is_t = [isinstance(a, torch.Tensor) for a in args]
        ctx.consts = [None if t else a for a, t in zip(args, ctx.is_t)]
        ctx.save_for_backward(*[a for a in args if isinstance(a, torch.Tensor)])
        ctx.req = [isinstance(a, torch.Tensor) and a.requires_grad for a in args]
        with torch.no_grad():
            out = hip_fn(*args)
        ctx.multi = isinstance(out, (tuple, list))
        return tuple(out) if ctx.multi else out

    @staticmethod
    def backward(ctx, *grads):
        saved = list(ctx.saved_tensors)
        args, leaves = [], []
        for is_t, c, r in zip(ctx.is_t, ctx.consts, ctx.req):
            if is_t:
                t = saved.pop(0).detach()
                if r:
                    t.requires_grad_(True)
                    leaves.append(t)
                args.append(t)
            else:
                args.append(c)
        with torch.enable_grad():
            outs = ctx.aten_fn(*args)
            outs = list(outs) if ctx.multi else [outs]
            pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o.requires_grad]
            gin = torch.autograd.grad([o for o, _ in pairs], leaves, [g.to(o.dtype) for o, g in pairs],
                                      allow_unused=True) if pairs and leaves else [None] * len(leaves)
        gin = list(gin)
        res = [None, None]
        for r in ctx.req:
            res.append(gin.pop(0) if r else None)
        return tuple(res)


def _call(hip_fn, aten_fn, *args):
    if torch.is_grad_enabled() and any(isinstance(a, torch.Tensor) and a.requires_grad for a in args):
        return _Fused.apply(hip_fn, aten_fn, *args)
    return hip_fn(*args)


# ----------------------------------------------------------------------------- linear (grouped)
def _act(z, act):
    if act == ACT_GELU:
        return F.gelu(z)
    if act == ACT_RELU:
        return torch.relu(z)
    if act == ACT_TANH:
        return torch.tanh(z)
    return z


def linear_grouped(problems, act=ACT_NONE, out_f32=False):
    """problems: list of (x, w, bias|None, residual|None) -> list of outputs; one kernel launch."""
    n = len(problems)
    flat = [t for p in problems for t in p]

    def hip(*a):
        return K.gemm([tuple(a[4 * i:4 * i + 4]) for i in range(n)], act, out_f32)

    def aten(*a):
        outs = []
        for i in range(n):
            x, w, b, r = a[4 * i:4 * i + 4]
            z = F.linear(x, w).float()
            if b is not None:
                z = z + b
            z = _act(z, act)
            if r is not None:
                z = z + r.float()
            outs.append(z if out_f32 else z.to(x.dtype))
        return outs

    return list(_call(hip, aten, *flat))


def linear(x, w, b=None, act=ACT_NONE, residual=None, out_f32=False):
    return linear_grouped([(x, w, b, residual)], act, out_f32)[0]


# ----------------------------------------------------------------------------- attention
def attention(q, k, v, nh, kmask=None, mask_mode=MASK_NONE, dists=None, sprel_w=None, sprel_b=None, scale=0.125):
    def hip(q, k, v, kmask, dists, sw, sb):
        return K.attention(q, k, v, nh, kmask, mask_mode if kmask is not None else MASK_NONE, dists, sw, sb, scale)

    def aten(q, k, v, kmask, dists, sw, sb):
        B, Nq, _ = q.shape
        sp = lambda t: t.reshape(B, t.shape[1], nh, 64).permute(0, 2, 1, 3).float()
        s = sp(q) @ sp(k).transpose(-1, -2) * scale
        if kmask is not None:
            m = kmask[:, None, None, :]
            if mask_mode == MASK_NEG10000:
                s = s + (1.0 - m.float()) * -10000.0
            elif mask_mode == MASK_NEGINF:
                s = s.masked_fill(~m, float('-inf'))
            elif mask_mode == MASK_PLUS1:
                s = s + m.float()
        if dists is not None:
            s = s + (dists * sw.reshape(()) + sb.reshape(()))[:, None]
        o = torch.softmax(s, -1) @ sp(v)
        return o.permute(0, 2, 1, 3).reshape(B, Nq, nh * 64).to(q.dtype)

    return _call(hip, aten, q, k, v, kmask, dists, sprel_w, sprel_b)


# ----------------------------------------------------------------------------- layernorm
def layernorm(x, gamma, beta, eps, residual=None, out_dtype=None, out=None):
    odt = out_dtype or (residual.dtype if residual is not None else x.dtype)

    def hip(x, g, b, r):
        return K.layernorm(x, g, b, eps, r, odt, out)

    def aten(x, g, b, r):
        z = x.float() if r is None else x.float() + r.float()
        return F.layer_norm(z, (z.shape[-1],), g, b, eps).to(odt)

    return _call(hip, aten, x, gamma, beta, residual)


# ----------------------------------------------------------------------------- token embedding
def embed_ln(M, H, dtype, device, base=None, feat=None, Wf=None, bf=None, lnf=None, E1=None, idx1=None,
             E2=None, idx2=None, vec=None, lno=None, eps=1e-12, out=None):
    lnf_g, lnf_b = lnf if lnf is not None else (None, None)
    lno_g, lno_b = lno if lno is not None else (None, None)

    def hip(base, feat, Wf, bf, lnf_g, lnf_b, E1, E2, vec, lno_g, lno_b):
        return K.embed_ln(M, H, dtype, device, base, feat, Wf, bf, (lnf_g, lnf_b) if lnf_g is not None else None,
                          E1, idx1, E2, idx2, vec, (lno_g, lno_b) if lno_g is not None else None, eps, out)

    def aten(base, feat, Wf, bf, lnf_g, lnf_b, E1, E2, vec, lno_g, lno_b):
        z = torch.zeros(M, H, device=device)
        if feat is not None:
            t = F.linear(feat.reshape(M, -1), Wf, bf)
            z = F.layer_norm(t, (H,), lnf_g, lnf_b, eps) if lnf_g is not None else t
        if base is not None:
            z = z + base.reshape(M, H).float()
        if E1 is not None:
            z = z + E1[idx1.long()]
        if E2 is not None:
            z = z + E2[idx2.long()]
        if vec is not None:
            z = z + vec
        if lno_g is not None:
            z = F.layer_norm(z, (H,), lno_g, lno_b, eps)
        return z.to(dtype)

    return _call(hip, aten, base, feat, Wf, bf, lnf_g, lnf_b, E1, E2, vec, lno_g, lno_b)


def group_mean(x, G, out=None):
    return _call(lambda x: K.group_mean(x, G, out),
                 lambda x: x.float().reshape(-1, G, x.shape[-1]).mean(1).to(x.dtype), x)


def gaa_pool(x, w, b):
    def aten(x, w, b):
        xf = x.float()
        a = torch.softmax(torch.tanh(xf @ w + b), 1)
        return (xf * a[..., None]).sum(1).to(x.dtype)
    return _call(K.gaa_pool, aten, x, w, b)


def head_tail(h, gamma, beta, w2, b2, eps=1e-12):
    def aten(h, g, be, w2, b2):
        return F.layer_norm(h, (h.shape[-1],), g, be, eps) @ w2 + b2
    return _call(lambda h, g, be, w2, b2: K.head_tail(h, g, be, w2, b2, eps), aten, h, gamma, beta, w2, b2)


def nav_logits(g_raw, l_raw, fuse_raw, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask):
    def hip(g, l, f):
        return K.nav_logits(g, l, f, gmap_masks, gmap_visited, vp_nav_masks, gmap_src, bw_mask)

    def aten(g, l, f):
        fw = torch.sigmoid(f)[:, None] if f is not None else 0.5
        ninf = float('-inf')
        gl = (g * fw).masked_fill(gmap_visited | ~gmap_masks, ninf)
        ll = (l * (1 - fw)).masked_fill(~vp_nav_masks, ninf)
        bw = torch.where(bw_mask.bool(), ll, torch.zeros_like(ll)).sum(1, keepdim=True)
        src = gmap_src.long()
        picked = torch.gather(ll, 1, src.clamp(min=0))
        add = torch.where(src >= 0, picked, torch.where(src == -2, bw.expand_as(picked), torch.zeros_like(picked)))
        add = torch.cat([ll[:, :1], add[:, 1:]], 1)
        return gl, ll, gl + add

    return _call(hip, aten, g_raw, l_raw, fuse_raw)


def igl_blend(aug, x, wa, ba, wo, bo):
    def aten(aug, x, wa, ba, wo, bo):
        a, xf = aug.float(), x.float()
        w = torch.sigmoid(a @ wa + ba + xf @ wo + bo)[..., None]
        return (w * a + (1 - w) * xf).to(x.dtype)
    return _call(K.igl_blend, aten, aug, x, wa, ba, wo, bo)
