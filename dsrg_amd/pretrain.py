"""MI355X-native pre-training model: drop-in for the reference's `GlocalTextPathCMTPreTraining`
(pretrain_src/model/pretrain_cmt.py:38-392) over its backbone `GlocalTextPathCMT`
(pretrain_src/model/vilmodel.py:716-840).

Same constructor (a config with `pretrain_tasks`), same `forward(batch, task, compute_loss)` with task prefixes
mlm / mrc / sap (+ `ValueError('invalid task')`), same per-sample un-reduced loss vectors, same `state_dict()` keys
(624 tensors: `bert.*` backbone, `mlm_head.*` with the decoder tied to the word embeddings, `image_classifier.*`,
`global_sap_head.*`, `local_sap_head.*`, `sap_fuse_linear.*`).  All arithmetic of the backbone and heads runs in the
libdsrg_hip.so kernels through dsrg_amd.functional; only the final scalar losses (log-softmax / CE / KL on a few
fp32 logits) are ATen.

Two host-side loops of the reference become index tensors consumed on the device:
  * `_aggregate_gmap_features` (vilmodel.py:597-635, dict-of-tensors pooling over the trajectory) -> one
    gather + index_add with per-contribution weights (`gmap_aggregation_indices`);
  * the SAP logit fusion (pretrain_cmt.py:269-285) -> `fusion_indices(skip=1)` + the nav_logits kernel.
"""
import numpy as np
import torch
import torch.nn as nn

from . import functional as Fn
from ._lib import ACT_GELU, ACT_RELU, MASK_NEG10000
from .vilmodel import GlocalTextPathNavCMT as _Backbone, _cls_head, _dense_ln, _mod


def _memo(b, key, build):
    """Host-built index tensors of a batch (aggregation incidences, selections of masked tokens, the last step of every
    trajectory, logit-fusion sources) depend only on the batch's own index data.  A batch that carries a dict under
    '_prep' keeps them there: the first forward over it fills the dict, later ones - the captured training step of
    dsrg_amd.train_graph replays the batch - find them and issue no host work, no host-to-device copy and no
    data-dependent shape query.  Without '_prep' everything is built per call, as before."""
    p = b.get('_prep')
    if p is None:
        return build()
    if key not in p:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("batch['_prep'] lacks %r: run one eager forward over this batch before capturing it" % key)
        p[key] = build()
    return p[key]


class GlocalTextPathCMT(_Backbone):
    PRETRAIN = True

    # ------------------------------------------------------------------ pieces
    def _text(self, txt_ids, txt_lens):
        L = txt_ids.shape[1]
        txt_masks = torch.arange(L, device=txt_ids.device)[None] < txt_lens[:, None]
        return self.forward_text(txt_ids, txt_masks), txt_masks

    def _trajectory(self, b):
        """ImageEmbeddings.forward over all steps of all trajectories at once (vilmodel.py:497-546): identical
        arithmetic to the fine-tune panorama mode, so it IS that code path.  Also returns the per-step token counts
        (views, or views + objv1 objects for REVERIE) that the viewpoint encoder masks with."""
        pano, masks, fused = self.forward_panorama_per_step(
            b['traj_view_img_fts'], b['traj_objv2_fts'], b['traj_loc_fts'], b['traj_nav_types'],
            b['traj_vp_view_lens'], b['traj_objv2_lens'])
        return pano, fused, b['traj_vp_view_lens']

    @staticmethod
    def gmap_aggregation_indices(traj_step_lens, traj_vpids, traj_cand_vpids, gmap_vpids, V, G):
        """(dst, src, w): graph-map slot b*G+j  +=  w * row src of cat([pano.view(S*V,H), fused]).
        Visited node = GAA-pooled panorama of (the last) step that stood on it; unvisited node = mean of every
        candidate-view embedding that saw it before it was visited (vilmodel.py:605-627)."""
        dst, src, w = [], [], []
        S = int(sum(traj_step_lens))
        s0 = 0
        for i, T in enumerate(traj_step_lens):
            vis, unv = {}, {}
            for t in range(T):
                vis[traj_vpids[i][t]] = S * V + s0 + t
                for j, vp in enumerate(traj_cand_vpids[i][t]):
                    if vp not in vis:
                        unv.setdefault(vp, []).append((s0 + t) * V + j)
            for j, vp in enumerate(gmap_vpids[i][1:], start=1):
                if vp in vis:
                    dst.append(i * G + j); src.append(vis[vp]); w.append(1.0)
                else:
                    for r in unv[vp]:
                        dst.append(i * G + j); src.append(r); w.append(1.0 / len(unv[vp]))
            s0 += T
        return np.asarray(dst, np.int64), np.asarray(src, np.int64), np.asarray(w, np.float32)

    def _build_gmap_plan(self, b, S, V, B, G, dev):
        return Fn.segment_plan(*self.gmap_aggregation_indices(b['traj_step_lens'], b['traj_vpids'], b['traj_cand_vpids'], b['gmap_vpids'], V, G),
                               B * G, S * V + S, dev)

    @staticmethod
    def _build_vp_plan(b, S, V, T, dev):
        last_np = np.cumsum(np.asarray(b['traj_step_lens'], np.int64)) - 1
        # token (i, 0) = zeros ([stop]), token (i, t >= 1) = view t-1 of trajectory i's last step: the same gather kernel
        tt = np.arange(1, T)
        dst = (np.arange(last_np.shape[0])[:, None] * T + tt[None]).reshape(-1)
        src = (last_np[:, None] * V + (tt - 1)[None]).reshape(-1)
        return (torch.from_numpy(last_np).to(dev),
                Fn.segment_plan(dst, src, np.ones(dst.shape[0], np.float32), last_np.shape[0] * T, S * V, dev))

    def _gmap_input(self, W, pano, fused, b):
        cfg, cd = self.config, self.compute_dtype
        B, G = b['gmap_step_ids'].shape
        S, V, H = pano.shape
        dev = pano.device
        plan = _memo(b, 'gmap_plan', lambda: self._build_gmap_plan(b, S, V, B, G, dev))
        img = Fn.segment_wsum(pano.reshape(S * V, H), fused, None, None, None, B * G, plan=plan)      # one gather kernel, no atomics
        g = Fn.embed_ln(B * G, H, cd, dev, base=img, feat=b['gmap_pos_fts'].float().reshape(B * G, -1).contiguous(),
                        Wf=W['gmap.pos.w'], bf=W['gmap.pos.b'], lnf=(W['gmap.pos.ln.g'], W['gmap.pos.ln.b']),
                        E1=W['gmap.step'], idx1=b['gmap_step_ids'].reshape(-1)).view(B, G, H)
        gmask = (torch.arange(G, device=dev)[None] < b['gmap_lens'][:, None]).contiguous()
        return g, gmask

    def _vp_input(self, W, pano, b, tok_lens):
        cfg, cd = self.config, self.compute_dtype
        S, V, H = pano.shape
        dev = pano.device
        T = b['vp_pos_fts'].shape[1]                 # = max(last view_len) + 1  ([stop] first)
        last, plan = _memo(b, 'vp_plan', lambda: self._build_vp_plan(b, S, V, T, dev))
        B = last.shape[0]
        vp_lens = tok_lens.index_select(0, last) + 1
        img = Fn.segment_wsum(pano.reshape(S * V, H), None, None, None, None, B * T, plan=plan)
        v = Fn.embed_ln(B * T, H, cd, dev, base=img.reshape(B * T, H), feat=b['vp_pos_fts'].float().reshape(B * T, -1).contiguous(),
                        Wf=W['vp.pos.w'], bf=W['vp.pos.b'], lnf=(W['vp.pos.ln.g'], W['vp.pos.ln.b'])).view(B, T, H)
        vmask = (torch.arange(T, device=dev)[None] < vp_lens[:, None]).contiguous()
        return v, vmask, last

    # ------------------------------------------------------------------ vilmodel.py:730-779
    def forward_backbone(self, b, return_gmap_embeds=True):
        W = self._weights()
        txt, tm = self._text(b['txt_ids'], b['txt_lens'])
        pano, fused, tok_lens = self._trajectory(b)
        kv = self.text_kv(txt) if not torch.is_grad_enabled() else Fn.linear(txt, W['x.txtkv.w'], W['x.txtkv.b'])
        v, vmask, last = self._vp_input(W, pano, b, tok_lens)
        tm = tm.contiguous()
        if return_gmap_embeds:
            g, gmask = self._gmap_input(W, pano, fused, b)
            g, v = self._x_layers(W, kv, tm, [g, v], [gmask, vmask], b['gmap_pair_dists'].float().contiguous())
            return g, v, last
        (v,) = self._x_layers(W, kv, tm, [v], [vmask], None, brs=('l',))
        return None, v, last

    # ------------------------------------------------------------------ vilmodel.py:782-840
    def forward_mlm(self, b):
        """Text updated by the visual context: per x-layer, text queries attend over the (un-updated) graph-map /
        viewpoint input tokens, then the text-side self-attention + FFN sub-layers; the two branches are summed."""
        W, cfg = self._weights(), self.config
        H, nh, eps, nl = cfg.hidden_size, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.num_x_layers
        txt, tm = self._text(b['txt_ids'], b['txt_lens'])
        tm = tm.contiguous()
        pano, fused, tok_lens = self._trajectory(b)
        g, gmask = self._gmap_input(W, pano, fused, b)
        v, vmask, _ = self._vp_input(W, pano, b, tok_lens)
        ctx, cmask, brs = [g, v], [gmask, vmask], ('g', 'l')
        t = [txt, txt]
        for i in range(nl):
            ks = ['x.%s.%d' % (br, i) for br in brs]
            ph = cfg.hidden_dropout_prob if self.training else 0.0
            kv = Fn.linear_grouped([(ctx[j], W[ks[j] + '.ckv.w'], W[ks[j] + '.ckv.b'], None) for j in range(2)])
            t = Fn.attn_block_grouped([(t[j], W[ks[j] + '.cq.w'], W[ks[j] + '.cq.b'], W[ks[j] + '.co.w'], W[ks[j] + '.co.b'], W[ks[j] + '.cln.g'],
                                        W[ks[j] + '.cln.b'], kv[j], cmask[j], None) for j in range(2)], nh, MASK_NEG10000, eps, self._attn_p(), ph)
            t = Fn.attn_block_grouped([(t[j], W[ks[j] + '.lqkv.w'], W[ks[j] + '.lqkv.b'], W[ks[j] + '.lso.w'], W[ks[j] + '.lso.b'], W[ks[j] + '.lsln.g'],
                                        W[ks[j] + '.lsln.b'], None, tm, None) for j in range(2)], nh, MASK_NEG10000, eps, self._attn_p(), ph)
            t = Fn.ffn_grouped([(t[j], W[ks[j] + '.lff1.w'], W[ks[j] + '.lff1.b'], W[ks[j] + '.lff2.w'], W[ks[j] + '.lff2.b'], t[j],
                                 W[ks[j] + '.lfln.g'], W[ks[j] + '.lfln.b']) for j in range(2)], eps, cfg.hidden_dropout_prob if self.training else 0.0)
        return t[0].float() + t[1].float()

    def forward(self, *a, **k):
        raise NotImplementedError('use GlocalTextPathCMTPreTraining.forward(batch, task) or forward_backbone / forward_mlm')

    def _head_weights(self, W, lin, ln, f):      # heads live on the wrapper (pretrain_cmt.py:45-61)
        pass


def _region_head(h, n_out):   # RegionClassification, pretrain_cmt.py:12-23
    return _cls_head(h, out=n_out)


class GlocalTextPathCMTPreTraining(nn.Module):
    BACKBONE = GlocalTextPathCMT

    def __init__(self, config, compute_dtype=torch.bfloat16):
        super().__init__()
        self.config = cfg = config
        h = cfg.hidden_size
        self.bert = self.BACKBONE(cfg, compute_dtype=compute_dtype)
        tasks = list(getattr(cfg, 'pretrain_tasks', ()))
        if 'mlm' in tasks:
            pred = _mod(transform=_dense_ln(h, h, cfg.layer_norm_eps), decoder=nn.Linear(h, cfg.vocab_size, bias=False))
            pred.bias = nn.Parameter(torch.zeros(cfg.vocab_size))
            self.mlm_head = _mod(predictions=pred)
        if 'mrc' in tasks:
            self.image_classifier = _region_head(h, cfg.image_prob_size)
            op = getattr(cfg, 'obj_prob_size', 0)
            self.obj_classifier = _region_head(h, op) if (op > 0 and op != cfg.image_prob_size) else None
        if 'sap' in tasks:
            self.global_sap_head, self.local_sap_head = _cls_head(h), _cls_head(h)
            self.sap_fuse_linear = _cls_head(h, inp=2 * h) if cfg.glocal_fuse else None
        if 'og' in tasks:
            self.og_head = _cls_head(h)
        for m in self.modules():
            if m is not self.bert and not any(m is x for x in self.bert.modules()):
                if isinstance(m, nn.Linear):
                    nn.init.normal_(m.weight, 0.0, getattr(cfg, 'initializer_range', 0.02))
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)
        self.tie_weights()

    def tie_weights(self):   # pretrain_cmt.py:66-69
        if hasattr(self, 'mlm_head'):
            self.mlm_head.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, config=None, state_dict=None, **kw):
        model = cls(config, **kw)
        if state_dict:
            own = model.state_dict()
            model.load_state_dict({k: v for k, v in state_dict.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}, strict=False)
        return model

    # ------------------------------------------------------------------ heads
    def _hw(self):
        """compute-dtype views of the head weights (differentiable; small)."""
        plan = getattr(self, '_hwplan', None)
        if plan is not None and plan[2] is self.bert._wplan_token():
            return self._hw_resolve(plan)
        cd = self.bert.compute_dtype
        W = {}
        pk = Fn.WeightPacker()
        c = lambda t: pk.cast(t, cd)
        f = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else pk.cast(t, torch.float32)
        lin = lambda k, m: W.update({k + '.w': c(m.weight), k + '.b': f(m.bias) if m.bias is not None else None})
        ln = lambda k, m: W.update({k + '.g': f(m.weight), k + '.b': f(m.bias)})
        for k in ('global_sap_head', 'local_sap_head', 'sap_fuse_linear', 'og_head', 'image_classifier', 'obj_classifier'):
            hd = getattr(self, k, None)
            if hd is not None:
                self.bert._cls_head_weights(W, k, hd, lin, ln, f)
        if hasattr(self, 'mlm_head'):
            p = self.mlm_head.predictions
            lin('mlm.t', p.transform.dense); ln('mlm.ln', p.transform.LayerNorm)
            W['mlm.dec.w'], W['mlm.dec.b'] = c(p.decoder.weight), f(p.bias)
        self._hwplan = plan = (W, pk, self.bert._wplan_token())
        self._hwres = None
        return self._hw_resolve(plan)

    def weight_packs(self):
        """the model's weight packs (functional._Spec, each with its plan) once a grad-mode forward has run: what
        dsrg_amd.optim.AdamW.attach_packs takes"""
        out = []
        for plan in (self.bert._wplan, getattr(self, '_hwplan', None)):
            spec = getattr(plan[1], '_spec_t', None) if plan is not None else None
            if spec is not None and spec.plan is not None:
                out.append(spec)
        return out

    def sync_packs(self):
        """run the packs now (grad mode), so that optimizer-managed buffers are current before a capture"""
        self.drop_grad_packs()
        with torch.enable_grad():
            self.bert._weights()
            self._hw()

    def drop_grad_packs(self):
        """forget the differentiable weight packs (and with them the autograd nodes of the step that made them)"""
        self.bert._wgrad = None
        self._hwres = None

    def _hw_resolve(self, plan):
        """one pack per parameter version and grad mode (see GlocalTextPathNavCMT._resolve_weights)"""
        W0, pk, _ = plan
        key = (torch.is_grad_enabled(), Fn.pack_version(pk.srcs))
        if self._hwres is not None and self._hwres[0] == key:
            return self._hwres[1]
        W = pk.resolve(dict(W0))
        self._hwres = (key, W)
        return W

    def _scores(self, W, key, x):
        """ClsPrediction -> [..] raw scores (Linear+ReLU in the GEMM epilogue, LN+Linear(H,1) tail kernel)."""
        h = Fn.linear(x, W[key + '.l1.w'], W[key + '.l1.b'], act=ACT_RELU, out_f32=True)
        return Fn.head_tail(h, W[key + '.ln.g'], W[key + '.ln.b'], W[key + '.w2'], W[key + '.b2'])

    # ------------------------------------------------------------------ tasks
    def forward(self, batch, task, compute_loss=True):
        b = batch if hasattr(batch, 'get') else dict(batch)
        if task.startswith('mlm'):
            return self.forward_mlm(b, compute_loss)
        elif task.startswith('mrc'):
            return self.forward_mrc(b, compute_loss)
        elif task.startswith('sap'):
            return self.forward_sap(b, compute_loss)
        elif task.startswith('og'):
            return self.forward_og(b, compute_loss)
        elif task.startswith('valid_sap_og'):
            return self.forward_sap_og(b)
        raise ValueError('invalid task')

    def forward_mlm(self, b, compute_loss=True):   # pretrain_cmt.py:142-167
        W, cd = self._hw(), self.bert.compute_dtype
        txt = self.bert.forward_mlm(b)
        idx, labels = _memo(b, 'mlm_sel', lambda: self._build_mlm_sel(b))
        h = txt.reshape(-1, txt.shape[-1]).index_select(0, idx).to(cd)
        h = Fn.linear(h, W['mlm.t.w'], W['mlm.t.b'], act=ACT_GELU, out_f32=True)
        h = Fn.layernorm(h, W['mlm.ln.g'], W['mlm.ln.b'], self.config.layer_norm_eps, out_dtype=cd)
        scores = Fn.linear(h, W['mlm.dec.w'], W['mlm.dec.b'], out_f32=True)
        if compute_loss:
            return Fn.cross_entropy_rows(scores, labels)
        return scores

    @staticmethod
    def _build_mlm_sel(b):                            # the masked positions, row-major as txt[labels != -1] lists them
        idx = (b['txt_labels'].reshape(-1) != -1).nonzero().reshape(-1)
        return idx, b['txt_labels'].reshape(-1).index_select(0, idx)

    def _region_logits(self, W, key, x):
        """RegionClassification (pretrain_cmt.py:12-23): Linear+ReLU in the GEMM epilogue -> LayerNorm -> Linear(H, n_classes)"""
        cd = self.bert.compute_dtype
        h = Fn.linear(x, W[key + '.l1.w'], W[key + '.l1.b'], act=ACT_RELU, out_f32=True)
        h = Fn.layernorm(h, W[key + '.ln.g'], W[key + '.ln.b'], 1e-12, out_dtype=cd)
        return Fn.linear(h, W[key + '.l2.w'], W[key + '.l2.b'], out_f32=True)

    @staticmethod
    def _token_slices(b, key, v, start, lens, T, mask=None):
        """[B, T, H]: row i = v[i, start_i : start_i + lens_i] zero-padded (the reference's pad_tensors_wgrad over per-sample
        slices, pretrain_cmt.py:191-194, 203-206, 313-317; slices are clipped at the end of v like Python slicing).  With a
        [B, T] bool `mask` only the masked rows, in row-major order: [n_masked, H] = slices[mask].  One row-gather kernel
        (dsrg_segment_wsum) forward and one backward over an incidence built on the host from the lengths."""
        B, N, H = v.shape
        plan, m = _memo(b, key, lambda: GlocalTextPathCMTPreTraining._build_slices(start, lens, T, B, N, mask, v.device))
        if m == 0:
            return v.new_zeros((0, H))
        rows = Fn.segment_wsum(v.reshape(B * N, H), None, None, None, None, m, plan=plan)
        return rows if mask is not None else rows.view(B, T, H)

    @staticmethod
    def _build_slices(start, lens, T, B, N, mask, dev):
        st, ln = start.cpu().numpy().astype(np.int64), lens.cpu().numpy().astype(np.int64)
        t = np.arange(T)[None]
        ok = (t < ln[:, None]) & (st[:, None] + t < N)
        src = (np.arange(B)[:, None] * N + st[:, None] + t)
        if mask is None:
            dst, m = (np.arange(B)[:, None] * T + t), B * T
        else:
            mk = mask.cpu().numpy().astype(bool)
            dst, m = np.cumsum(mk.reshape(-1)).reshape(B, T) - 1, int(mk.sum())
            ok &= mk
        return Fn.segment_plan(dst[ok], src[ok], np.ones(int(ok.sum()), np.float32), m, B * N, dev), m

    def _last_lens(self, b, last):
        vl = b['traj_vp_view_lens'].index_select(0, last)
        ol = b['traj_vp_obj_lens'].index_select(0, last) if b.get('traj_vp_obj_lens') is not None else torch.zeros_like(vl)
        return vl, ol

    def forward_mrc(self, b, compute_loss=True):   # pretrain_cmt.py:175-228
        W = self._hw()
        _, v, last = self.bert.forward_backbone(b, return_gmap_embeds=False)
        vl, ol = self._last_lens(b, last)
        m = b['vp_view_mrc_masks']
        ve = self._token_slices(b, 'mrc_view_rows', v, torch.ones_like(vl), vl, m.shape[1], mask=m)     # [stop] at 0; = slices[m]
        logits = self._region_logits(W, 'image_classifier', ve)
        tgt = _memo(b, 'mrc_view_tgt', lambda: b['vp_view_probs'][m])
        ologits = otgt = None
        if b.get('traj_obj_img_fts') is not None:                                    # object regions (REVERIE), :202-213
            om = b['vp_obj_mrc_masks']
            oe = self._token_slices(b, 'mrc_obj_rows', v, vl + 1, ol, om.shape[1], mask=om)
            key = 'obj_classifier' if getattr(self, 'obj_classifier', None) is not None else 'image_classifier'
            ologits = self._region_logits(W, key, oe)
            otgt = _memo(b, 'mrc_obj_tgt', lambda: b['vp_obj_probs'][om])
        if not compute_loss:
            return logits, tgt, ologits, otgt
        loss = Fn.kl_div_rows(logits, tgt)
        if ologits is not None:
            loss = torch.cat([loss, Fn.kl_div_rows(ologits, otgt)], 0)
        return loss

    def _og_logits(self, W, b, v, last):   # pretrain_cmt.py:313-321
        vl, ol = self._last_lens(b, last)
        T = _memo(b, 'og_T', lambda: int(ol.max()))
        oe = self._token_slices(b, 'og_rows', v, vl + 1, ol, T)
        sc = self._scores(W, 'og_head', oe)
        return sc.masked_fill(torch.arange(T, device=v.device)[None] >= ol[:, None], -float('inf'))

    def forward_og(self, b, compute_loss=True):   # pretrain_cmt.py:296-327
        W = self._hw()
        _, v, last = self.bert.forward_backbone(b, return_gmap_embeds=False)
        obj_logits = self._og_logits(W, b, v, last)
        if compute_loss:
            return Fn.cross_entropy_rows(obj_logits, b['obj_labels'])
        return obj_logits

    def forward_sap_og(self, b):   # pretrain_cmt.py:329-392
        """NB the reference's own call here drops the objv2 features (`self.bert(...)` without traj_objv2_fts, :337-341), which
        its objv2 backbones cannot run without (AttributeError at vilmodel.py:502 / vilmodel_reverie.py:526), and no driver
        ever uses the task; this is the evident intent: SAP logits and OG logits from ONE backbone pass."""
        W = self._hw()
        g, v, last = self.bert.forward_backbone(b)
        gl, ll, fl = self._sap_logits(W, b, g, v, last)
        return gl, ll, fl, self._og_logits(W, b, v, last)

    def forward_sap(self, b, compute_loss=True):   # pretrain_cmt.py:230-294
        W = self._hw()
        g, v, last = self.bert.forward_backbone(b)
        gl, ll, fl = self._sap_logits(W, b, g, v, last)
        if not compute_loss:
            return gl, ll, fl, b['global_act_labels'], b['local_act_labels']
        return Fn.cross_entropy_rows(gl, b['global_act_labels']) + Fn.cross_entropy_rows(ll, b['local_act_labels']) \
            + Fn.cross_entropy_rows(fl, b['global_act_labels'])

    def _build_sap_fusion(self, b, G, T, dev):
        cands = [[None] + list(c[-1]) for c in b['traj_cand_vpids']]
        src, bw = self.bert.fusion_indices(b['gmap_vpids'], b['gmap_visited_masks'], cands, G, T, skip=1)
        return torch.from_numpy(src).to(dev), torch.from_numpy(bw).to(dev)

    def prepare(self, batch, task):
        """Fill batch['_prep'] with every host-built index tensor `forward(batch, task)` consumes (see _memo), from the batch
        alone: what a data-loader worker runs ahead of the step so that the step itself issues kernels only
        (dsrg_amd.train_graph replays it as a HIP graph).  Idempotent; returns the batch."""
        b = batch
        p = b.setdefault('_prep', {})
        dev = b['traj_loc_fts'].device
        S, V = b['traj_loc_fts'].shape[:2]
        T = b['vp_pos_fts'].shape[1]
        B, G = b['gmap_step_ids'].shape
        memo = lambda key, build: _memo(b, key, build)
        last, _ = memo('vp_plan', lambda: self.bert._build_vp_plan(b, S, V, T, dev))
        if task.startswith('mlm') or task.startswith('sap') or task.startswith('valid_sap_og'):
            memo('gmap_plan', lambda: self.bert._build_gmap_plan(b, S, V, B, G, dev))
        if task.startswith('mlm'):
            memo('mlm_sel', lambda: self._build_mlm_sel(b))
        if task.startswith('sap') or task.startswith('valid_sap_og'):
            memo('sap_fusion', lambda: self._build_sap_fusion(b, G, T, dev))
        if task.startswith('mrc') or task.startswith('og') or task.startswith('valid_sap_og'):
            vl, ol = self._last_lens(b, last)
            if task.startswith('mrc'):
                m = b['vp_view_mrc_masks']
                memo('mrc_view_rows', lambda: self._build_slices(torch.ones_like(vl), vl, m.shape[1], B, T, m, dev))
                memo('mrc_view_tgt', lambda: b['vp_view_probs'][m])
                if b.get('traj_obj_img_fts') is not None:
                    om = b['vp_obj_mrc_masks']
                    memo('mrc_obj_rows', lambda: self._build_slices(vl + 1, ol, om.shape[1], B, T, om, dev))
                    memo('mrc_obj_tgt', lambda: b['vp_obj_probs'][om])
            else:
                To = memo('og_T', lambda: int(ol.max()))
                memo('og_rows', lambda: self._build_slices(vl + 1, ol, To, B, T, None, dev))
        return b

    def _sap_logits(self, W, b, g, v, last):   # pretrain_cmt.py:248-285
        B, G, _ = g.shape
        T = v.shape[1]
        dev = g.device
        sg, sl = self._scores(W, 'global_sap_head', g), self._scores(W, 'local_sap_head', v)
        fr = self._scores(W, 'sap_fuse_linear', torch.cat([g[:, 0], v[:, 0]], 1)) if self.sap_fuse_linear is not None else None
        gmask = (torch.arange(G, device=dev)[None] < b['gmap_lens'][:, None]).contiguous()
        nav_last = b['traj_nav_types'].index_select(0, last)                      # [B,V]
        nav_ok = torch.cat([torch.ones(B, 1, dtype=torch.bool, device=dev), nav_last[:, :T - 1] == 1], 1).contiguous()
        src, bw = _memo(b, 'sap_fusion', lambda: self._build_sap_fusion(b, G, T, dev))
        return Fn.nav_logits(sg, sl, fr, gmask, b['gmap_visited_masks'].contiguous(), nav_ok, src, bw)
