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
                                  obj_img_fts=None, obj_lens=None, objv2_nav_types=None):
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
        if obj_img_fts is not None:
            No = obj_img_fts.shape[1]
            # per-episode [views[:vl] ; objects[:ol] ; zero pad] as one gather (reference: Python loop :727-735)
            T = nav_types.shape[1]
            lens = view_lens + obj_lens
            pos = torch.arange(T, device=dev)[None]
            vl, tl = view_lens[:, None], lens[:, None]
            idx = torch.where(pos < vl, pos, torch.where(pos < tl, V + pos - vl, torch.full_like(pos, V + No)))
            src = torch.cat([view, o1, torch.zeros(B, 1, H, device=dev, dtype=cd)], 1)
            x = torch.gather(src, 1, idx[..., None].expand(B, T, H))
        else:
            x, lens, T = view, view_lens, V
        x = self._pano_tokens(W, x, loc_fts, nav_types)
        pano_masks = torch.arange(T, device=dev)[None] < lens[:, None]
        x = self._pano_encoder(W, x, pano_masks.contiguous())
        pano = Fn.layernorm(x, W['pano.norm.g'], W['pano.norm.b'], 1e-12)
        fused = Fn.gaa_pool(pano, W['gaa.w'], W['gaa.b'])
        return pano, pano_masks, fused

    def forward(self, mode, batch, **kwargs):
        if mode == 'panorama':
            return self.forward_panorama_per_step(
                batch['view_img_fts'], batch['objv2_img_fts'], batch['loc_fts'], batch['nav_types'], batch['view_lens'],
                batch['objv2_lens'], batch['obj_img_fts'], batch['obj_lens'], batch['objv2_nav_types'])
        return super().forward(mode, batch, **kwargs)
