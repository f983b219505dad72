"""REVERIE variant of the navigation-policy network (reference: finetune_src/models/vilmodel_reverie.py).

Differences from R2R that this file carries (all in the panorama mode; language IGL uses object words only
and is handled by the REVERIE flag in the base class):
  * ASV object embedding is fused into each view BEFORE the pano encoder (vilmodel_reverie.py:719-720);
  * 768-d "objv1" object tokens go through obj_reverie_linear + LN and are appended to each episode's
    view sequence (:722-736), so the pano encoder runs over 36 + No tokens;
  * GAA pools the pano-encoder output over all padded slots (:753-758);
  * the object-grounding head is live in navigation mode (base class, vp_obj_masks).
"""
import torch

from . import functional as Fn, kernels as K
from .vilmodel import GlocalTextPathNavCMT as _Base


class GlocalTextPathNavCMT(_Base):
    REVERIE = True

    def forward_panorama_per_step(self, view_img_fts, objv2_img_fts, loc_fts, nav_types, view_lens, objv2_lens,
                                  obj_img_fts=None, obj_lens=None, objv2_nav_types=None, pano_out=None, gmap_scatter=None,
                                  want_masks=True):
        """pano_out / gmap_scatter / want_masks: the serving-loop extensions of the base class's panorama call"""
        W, cfg, cd = self._weights(), self.config, self.compute_dtype
        B, V, _ = view_img_fts.shape
        H, d = cfg.hidden_size, cfg.obj_v2_size
        dev = view_img_fts.device
        objs = self._fork(lambda: self._objv2_tokens(W, objv2_img_fts, objv2_lens, objv2_nav_types))
        z = Fn.linear(view_img_fts.to(cd), W['img.lin.w'], W['img.lin.b'], out_f32=True)
        o1 = None
        if obj_img_fts is not None:
            o1 = Fn.linear_ln(obj_img_fts.to(cd), W['objv1.lin.w'], W['objv1.lin.b'], W['objv1.ln.g'], W['objv1.ln.b'], 1e-12)
        obj_rows, O = objs()
        g, b = W['img.ln.g'], W['img.ln.b']
        cat = self._concat2(lambda o: Fn.layernorm(z, g, b, 1e-12, out_dtype=cd, out=o) if o is not None
                            else Fn.layernorm(z, g, b, 1e-12, out_dtype=cd),
                            lambda o: K.group_mean(obj_rows, O, out=o) if o is not None else Fn.group_mean(obj_rows, O),
                            B, V, H, d, dev)
        view = Fn.linear_ln(cat, W['visobj.lin.w'], W['visobj.lin.b'], W['visobj.ln.g'], W['visobj.ln.b'], 1e-12)
        lens2 = None
        if obj_img_fts is not None:
            No = obj_img_fts.shape[1]
            T = nav_types.shape[1]
            lens2 = obj_lens
            if not torch.is_grad_enabled() and view_lens.dtype == torch.int64 and obj_lens.dtype == torch.int64:
                # per-episode [views[:vl] ; objects[:ol] ; zero pad] (reference: Python loop :727-735) inside the token-embedding
                # kernel: its base row is gathered from the view rows / the object rows / zero by the two lengths
                x = self._pano_tokens(W, view, loc_fts, nav_types, seg=(o1.reshape(B * No, H), No, view_lens.contiguous(), obj_lens.contiguous()))
            else:       # (autograd path: one gather)
                lens = view_lens + obj_lens
                pos = torch.arange(T, device=dev)[None]
                vl, tl = view_lens[:, None], lens[:, None]
                idx = torch.where(pos < vl, pos, torch.where(pos < tl, V + pos - vl, torch.full_like(pos, V + No)))
                src = torch.cat([view, o1, torch.zeros(B, 1, H, device=dev, dtype=cd)], 1)
                x = self._pano_tokens(W, torch.gather(src, 1, idx[..., None].expand(B, T, H)), loc_fts, nav_types)
        else:
            T = V
            x = self._pano_tokens(W, view, loc_fts, nav_types)
        km, pano_masks = self._len_masks(view_lens, T, want_masks, lens2)
        x = self._pano_encoder(W, x, km)
        if pano_out is not None and not torch.is_grad_enabled():
            pano = K.layernorm_grouped([(x, W['pano.norm.g'], W['pano.norm.b'], None)], 1e-12, cd, outs=[pano_out])[0]
        else:
            pano = Fn.layernorm(x, W['pano.norm.g'], W['pano.norm.b'], 1e-12)
        fused = Fn.gaa_pool(pano, W['gaa.w'], W['gaa.b'], scatter=gmap_scatter)
        return pano, pano_masks, fused

    def forward(self, mode, batch, **kwargs):
        if mode == 'panorama':
            get = batch.get if hasattr(batch, 'get') else (lambda k, d=None: batch[k])
            return self.forward_panorama_per_step(
                batch['view_img_fts'], batch['objv2_img_fts'], batch['loc_fts'], batch['nav_types'], batch['view_lens'],
                batch['objv2_lens'], batch['obj_img_fts'], batch['obj_lens'], batch['objv2_nav_types'],
                pano_out=get('pano_out'), gmap_scatter=get('gmap_scatter'), want_masks=get('want_pano_masks', True) is not False)
        return super().forward(mode, batch, **kwargs)
