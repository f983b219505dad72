"""MI355X-native DSRG navigation-policy network (R2R and REVERIE variants).

Drop-in for the reference's `GlocalTextPathNavCMT` (finetune_src/models/vilmodel.py:630-858 and
vilmodel_reverie.py:634-874): same constructor (a config object), same `forward(mode, batch)` with
modes 'language' / 'panorama' / 'navigation', same return structure, same `state_dict()` keys and shapes
(499 tensors R2R / 503 REVERIE) so reference checkpoints load unchanged, `from_pretrained(None, config=,
state_dict=)` with the reference's 'bert.' prefix handling (vlnbert_init.py:58-66,110-113).

What is different is how it computes: the module tree only HOLDS parameters; `forward` drives the
hand-written gfx950 kernels of libdsrg_hip.so through dsrg_amd.functional:
  * Q/K/V projections are packed into one GEMM; global and local x-layer branches run as ONE grouped
    GEMM launch per sub-layer; the three RMF poolers and the SAP/OG heads are grouped the same way;
  * the 16 text K/V projections of the 8 x-layers (step-invariant, vilmodel.py:351-352 recomputes them
    every step) are one [B*L,768]x[768,12288] GEMM, cached per txt_embeds tensor in no-grad mode;
  * attention is one fused kernel per call (mask term, sprel bias w*d+b and the "+1 bool mask" quirk
    formed in-kernel); token-embedding sums+LayerNorms, GAA pooling, head tails and the logit fusion
    are single kernels; the viewpoint-id string matching of the fusion loop (vilmodel.py:788-805) is
    done once on the host into two small index tensors.
Compute dtype is bfloat16 (default) or float32 (exact-fp32 MFMA: the parity mode).
"""
import math
import numpy as np
import os
import torch
import torch.nn as nn

from . import functional as Fn, kernels as K
from ._lib import (ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH, MASK_NONE, MASK_NEG10000, MASK_NEGINF, MASK_PLUS1,
                   DsrgHipError)


# --------------------------------------------------------------------------- parameter tree
def _mod(**children):
    m = nn.Module()
    for k, v in children.items():
        setattr(m, k, v)
    return m


def _qkv(h, ctx=None):
    return _mod(query=nn.Linear(h, h), key=nn.Linear(ctx or h, h), value=nn.Linear(ctx or h, h))


def _dense_ln(i, o, eps):
    return _mod(dense=nn.Linear(i, o), LayerNorm=nn.LayerNorm(o, eps=eps))


def _self_att(h, eps):
    m = nn.Module()
    setattr(m, 'self', _qkv(h))
    m.output = _dense_ln(h, h, eps)
    return m


def _bert_layer(h, ff, eps):
    return _mod(attention=_self_att(h, eps), intermediate=_mod(dense=nn.Linear(h, ff)), output=_dense_ln(ff, h, eps))


def _x_layer(cfg):
    h, ff, eps = cfg.hidden_size, cfg.intermediate_size, cfg.layer_norm_eps
    m = nn.Module()
    if getattr(cfg, 'use_lang2visn_attn', False):
        m.lang_self_att, m.lang_inter, m.lang_output = _self_att(h, eps), _mod(dense=nn.Linear(h, ff)), _dense_ln(ff, h, eps)
    m.visn_self_att, m.visn_inter, m.visn_output = _self_att(h, eps), _mod(dense=nn.Linear(h, ff)), _dense_ln(ff, h, eps)
    m.visual_attention = _mod(att=_qkv(h), output=_dense_ln(h, h, eps))
    return m


def _x_encoder(cfg):
    return _mod(x_layers=nn.ModuleList([_x_layer(cfg) for _ in range(cfg.num_x_layers)]))


def _cls_head(h, inp=None, out=1):
    return _mod(net=nn.Sequential(nn.Linear(inp or h, h), nn.ReLU(), nn.LayerNorm(h, eps=1e-12), nn.Linear(h, out)))


def _pano_layer(h, ff):
    sa = _mod(out_proj=nn.Linear(h, h))
    sa.in_proj_weight = nn.Parameter(torch.empty(3 * h, h))
    sa.in_proj_bias = nn.Parameter(torch.zeros(3 * h))
    return _mod(self_attn=sa, linear1=nn.Linear(h, ff), linear2=nn.Linear(ff, h),
                norm1=nn.LayerNorm(h), norm2=nn.LayerNorm(h))


def _img_embeddings(cfg, reverie):
    h, d, a = cfg.hidden_size, cfg.obj_v2_size, cfg.angle_feat_size
    m = nn.Module()
    m.img_linear, m.img_layer_norm = nn.Linear(cfg.image_feat_size, h), nn.LayerNorm(h, eps=1e-12)
    m.loc_linear, m.loc_layer_norm = nn.Linear(a + 3, h), nn.LayerNorm(h, eps=1e-12)
    m.adaptive_pano_attn = nn.Linear(h, 1)
    if reverie:
        m.obj_reverie_linear, m.obj_reverie_layer_norm = nn.Linear(h, h), nn.LayerNorm(h, eps=1e-12)
    m.obj_bbox_linear = nn.Linear(a + 3, d)
    m.obj_name_linear = nn.Embedding(cfg.obj_name_vocab_size, d)
    m.obj_nav_type_embedding = nn.Embedding(3, d)
    m.obj_layer_norm = nn.LayerNorm(d, eps=1e-12)
    # NB intermediate size stays 3072 on the 128-d object tokens (vilmodel.py:569-573 copies the config)
    m.obj_self_encoder = _mod(layer=nn.ModuleList([_bert_layer(d, cfg.intermediate_size, cfg.layer_norm_eps) for _ in range(2)]))
    m.obj_attn = _mod()
    m.obj_attn.attn = nn.Parameter(torch.empty(d, 1).uniform_(-0.1, 0.1))      # in state_dict, unused in forward
    m.vis_obj_linear = nn.Sequential(nn.Linear(h + d, h), nn.LayerNorm(h, eps=1e-12))
    m.nav_type_embedding = nn.Embedding(3, h)
    m.layer_norm = nn.LayerNorm(h, eps=1e-12)
    m.pano_encoder = _mod(layers=nn.ModuleList([_pano_layer(h, cfg.intermediate_size) for _ in range(cfg.num_pano_layers)]),
                          norm=nn.LayerNorm(h, eps=1e-12))
    return m


_ARANGE = {}


def _arange(n, device):
    """cached torch.arange(n) (the length masks are rebuilt every step; the ramp is a constant)"""
    key = (n, str(device))
    t = _ARANGE.get(key)
    if t is None:
        t = torch.arange(n, device=device)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):   # never cache graph-pool memory
            _ARANGE[key] = t
    return t


class GlocalTextPathNavCMT(nn.Module):
    REVERIE = False
    PRETRAIN = False        # pretrain_src/model/vilmodel.py:716-728 flavour: no IGL / heads / RMF, lang2visn sub-layers
    base_model_prefix = 'bert'

    def __init__(self, config, compute_dtype=torch.bfloat16):
        super().__init__()
        self.config = cfg = config
        self.compute_dtype = compute_dtype
        h, eps = cfg.hidden_size, cfg.layer_norm_eps
        if h != 64 * cfg.num_attention_heads or cfg.obj_v2_size % 64:
            raise DsrgHipError('the attention kernels are built for head dim 64: hidden_size %d / %d heads, obj_v2_size %d'
                               % (h, cfg.num_attention_heads, cfg.obj_v2_size))
        emb = nn.Module()
        emb.word_embeddings = nn.Embedding(cfg.vocab_size, h, padding_idx=0)
        emb.position_embeddings = nn.Embedding(cfg.max_position_embeddings, h)
        emb.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, h)
        emb.act_embedding = nn.Embedding(cfg.word_max_action, h)
        emb.obj_embedding = nn.Embedding(cfg.word_max_object, h)
        emb.LayerNorm = nn.LayerNorm(h, eps=eps)
        self.embeddings = emb

        le = nn.Module()
        le.layer = nn.ModuleList([_bert_layer(h, cfg.intermediate_size, eps) for _ in range(cfg.num_l_layers)])
        if self.PRETRAIN:   # pretrain_src/model/vilmodel.py:688-690 (present in the state dict, unused in forward);
            if not self.REVERIE:        # the REVERIE pre-training LanguageEncoder is plain (vilmodel_reverie.py:436-457)
                le.instr_act_encoder, le.instr_obj_encoder = _qkv(h), _qkv(h)
                le.instr_fuse_layernorm = nn.LayerNorm(h, eps=1e-12)
        else:
            le.act_obj_token_emb = nn.Embedding(1 if self.REVERIE else 2, h)       # unused in forward
            le.instr_actobj_encoder = _qkv(h)
            le.instr_aug_linear, le.instr_ori_linear = nn.Linear(h, 1), nn.Linear(h, 1)
            le.instr_fuse_layernorm = nn.LayerNorm(h, eps=eps)                      # unused in forward
        self.lang_encoder = le

        self.img_embeddings = _img_embeddings(cfg, self.REVERIE)
        self.local_encoder = _mod(
            vp_pos_embeddings=nn.Sequential(nn.Linear(cfg.angle_feat_size * 2 + 6, h), nn.LayerNorm(h, eps=1e-12)),
            encoder=_x_encoder(cfg))
        ge = _mod(gmap_pos_embeddings=nn.Sequential(nn.Linear(cfg.angle_feat_size + 3, h), nn.LayerNorm(h, eps=1e-12)),
                  gmap_step_embeddings=nn.Embedding(cfg.max_action_steps, h), encoder=_x_encoder(cfg))
        ge.sprel_linear = nn.Linear(1, 1) if cfg.graph_sprels else None
        self.global_encoder = ge
        if not self.PRETRAIN:
            self.global_sap_head, self.local_sap_head = _cls_head(h), _cls_head(h)
            self.sap_fuse_linear = _cls_head(h, inp=2 * h) if cfg.glocal_fuse else None
            if cfg.obj_feat_size > 0:
                self.og_head = _cls_head(h)
            # RMF (vilmodel.py:652-657)
            self.gmap_pooler, self.vp_pooler, self.txt_pooler = (_mod(dense=nn.Linear(h, h)) for _ in range(3))
            self.local_his_map = nn.Linear(3 * h, h)
            self.local_his_ln = nn.LayerNorm(h, eps=eps)

        self.init_weights()
        self._apply_freeze_flags()
        self._wcache, self._kv_cache, self._wplan, self._wgrad, self._wtoken = None, None, None, None, object()
        self._side_stream, self.overlap_streams, self._fork_depth = None, True, 0
        self.split_branch_streams = os.environ.get('DSRG_SPLIT_BRANCHES', '1') == '1'     # global / local x-layers on two streams (3.6 % faster than grouped launches at R2R shapes; 0 = grouped)
        self.split_when_eager = False          # bench.py sets it to trace the captured schedule's launches eagerly
        self.split_max_ratio = float(os.environ.get('DSRG_SPLIT_MAX_RATIO', '4.0'))
        # inference: the LayerNorms inside the x-layers folded into the neighbouring GEMM epilogues (_x_layers_fused_gen).
        # Off by default since r02t: with the specialised GEMM epilogue the plain kernels gained more than the FUSE ones
        # (c2 step 1.180 ms unfused vs 1.197 fused), and only the plain kernels have the wide tiles (1.162)
        self.fuse_ln = os.environ.get('DSRG_FUSE_LN', '1') == '1'
        # two-stream schedule: branches captured launch by launch in turn instead of one whole branch after the other
        # (measured: no difference in the replayed graph, 1.1976 vs 1.1995 ms; off)
        self.interleave_branches = os.environ.get('DSRG_INTERLEAVE', '0') == '1'
        # panorama call: view chain and object-token chain captured launch by launch in turn (A/B in DESIGN.md)
        self.interleave_pano = os.environ.get('DSRG_INTERLEAVE_PANO', '1') == '1'
        self.fold_img_ln = os.environ.get('DSRG_FOLD_IMG_LN', '1') == '1'      # panorama: img_layer_norm inside the token-embedding kernel
        # ... and vis_obj_linear's LayerNorm inside the GAA kernel: built, bit-identical, measured SLOWER (c2 1.000 -> 1.010 ms: the
        # GAA kernel is one workgroup per episode, 64 on 256 CUs, and the LayerNorm lengthens exactly that launch); off
        self.fold_visobj_ln = os.environ.get('DSRG_FOLD_VISOBJ_LN', '0') == '1'
        self.pano_split = int(os.environ.get('DSRG_PANO_SPLIT', '1'))
        self.split_big = int(os.environ.get('DSRG_SPLIT_BIG', '2'))      # the longer branch as up to this many part-batch chains (_x_layers_two_streams; 1 = whole;
                                                                         # 3 or 4 parts measure the same as 2 with HIP's default of 4 hardware queues)
        self.split_ratio = float(os.environ.get('DSRG_SPLIT_RATIO', '2.0'))
        self._fold = None
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate_caches())

    # ---------------------------------------------------------------- construction helpers
    def init_weights(self):
        std = getattr(self.config, 'initializer_range', 0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0.0, std)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, 0.0, std)
                if m.padding_idx is not None:
                    with torch.no_grad():
                        m.weight[m.padding_idx].zero_()
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        for lyr in self.img_embeddings.pano_encoder.layers:
            nn.init.xavier_uniform_(lyr.self_attn.in_proj_weight)

    def _apply_freeze_flags(self):  # vilmodel.py:662-676
        cfg = self.config
        frz = lambda mods: [p.requires_grad_(False) for m in mods for p in m.parameters()]
        if not getattr(cfg, 'update_lang_bert', True):
            frz([self.lang_encoder.layer])
        if getattr(cfg, 'fix_lang_embedding', False) or getattr(cfg, 'fix_local_branch', False):
            frz([self.embeddings, self.lang_encoder])
        if getattr(cfg, 'fix_pano_embedding', False) or getattr(cfg, 'fix_local_branch', False):
            frz([self.img_embeddings])
        if getattr(cfg, 'fix_local_branch', False) and not self.PRETRAIN:
            frz([self.local_encoder, self.local_sap_head] + ([self.og_head] if hasattr(self, 'og_head') else []))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, config=None, state_dict=None, **kw):
        """HF-style entry the reference uses (vlnbert_init.py:110-113): random init, then load the given
        state_dict with a leading 'bert.' stripped; unknown / missing keys are tolerated; returns .eval()."""
        if pretrained_model_name_or_path is not None:
            raise DsrgHipError('by-name checkpoints are not supported offline; pass state_dict=')
        model = cls(config, **kw)
        if state_dict:
            pre = cls.base_model_prefix + '.'
            sd = {(k[len(pre):] if k.startswith(pre) else k): v for k, v in state_dict.items()}
            own = model.state_dict()
            bad = ['%s: checkpoint %s vs model %s' % (k, tuple(v.shape), tuple(own[k].shape)) for k, v in sd.items()
                   if k in own and tuple(v.shape) != tuple(own[k].shape)]
            if bad:      # HF's from_pretrained(state_dict=) raises on a size mismatch too (vlnbert_init.py:110-113)
                raise DsrgHipError('from_pretrained: size mismatch for ' + '; '.join(bad[:8]))
            # unknown keys (mlm_head / image_classifier / lang_* sub-layers of a pre-training checkpoint) and missing ones
            # (RMF / IGL, randomly initialised at fine-tune start) are tolerated, as in the reference
            model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
        return model.eval()

    def invalidate_caches(self):
        self._wcache, self._kv_cache, self._wplan, self._wgrad, self._fold = None, None, None, None, None
        self._wtoken = object()

    def _wplan_token(self):
        """changes whenever invalidate_caches() ran (device / dtype move, train()/eval(), load_state_dict): dependants
        that keep their own pack plan (the pre-training heads) compare it"""
        return self._wtoken

    def train(self, mode=True):
        self.invalidate_caches()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self.invalidate_caches()
        return super()._apply(fn, *a, **k)

    # ---------------------------------------------------------------- packed weights
    def _weights(self):
        """Compute-dtype / packed views of the parameters.  Cached when autograd is off; rebuilt with
        differentiable ops (cat / to) when it is on, so gradients reach the fp32 master parameters."""
        plan = self._wplan
        grad = torch.is_grad_enabled() and (plan is None or plan[2])
        if not grad and self._wcache is not None:
            # eval(): weights are frozen until the next train()/eval()/load_state_dict() (those invalidate); a no-grad
            # forward of a model in train() mode may follow an optimizer.step(), so there the versions are compared
            if not self.training or self._wcache_ver == Fn.pack_version(plan[1].srcs):
                return self._wcache
        if plan is not None:
            return self._resolve_weights(plan, grad)
        cd = self.compute_dtype
        pk = Fn.WeightPacker()          # every cast / concat below is served by ONE multi-tensor copy (and one in backward)
        c = lambda t: pk.cast(t, cd)
        f = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else pk.cast(t.contiguous() if t.dim() == 1 else t, torch.float32)
        W = {}

        def lin(key, m):
            W[key + '.w'], W[key + '.b'] = c(m.weight), f(m.bias)

        def ln(key, m):
            W[key + '.g'], W[key + '.b'] = f(m.weight), f(m.bias)

        def packed(key, mods):
            W[key + '.w'] = pk.cat([m.weight for m in mods], cd)
            W[key + '.b'] = pk.cat([m.bias for m in mods], torch.float32)

        def bert_layer(key, L):
            packed(key + '.qkv', [L.attention.self.query, L.attention.self.key, L.attention.self.value])
            lin(key + '.ao', L.attention.output.dense); ln(key + '.aln', L.attention.output.LayerNorm)
            lin(key + '.ff1', L.intermediate.dense); lin(key + '.ff2', L.output.dense); ln(key + '.fln', L.output.LayerNorm)

        e, le, ie = self.embeddings, self.lang_encoder, self.img_embeddings
        for k, m in (('emb.word', e.word_embeddings), ('emb.pos', e.position_embeddings), ('emb.type', e.token_type_embeddings),
                     ('emb.act', e.act_embedding), ('emb.obj', e.obj_embedding)):
            W[k] = f(m.weight)
        ln('emb.ln', e.LayerNorm)
        for i, L in enumerate(le.layer):
            bert_layer('lang.%d' % i, L)
        if not self.PRETRAIN:
            lin('igl.q', le.instr_actobj_encoder.query)
            packed('igl.kv', [le.instr_actobj_encoder.key, le.instr_actobj_encoder.value])
            for k, m in (('igl.aug', le.instr_aug_linear), ('igl.ori', le.instr_ori_linear)):
                W[k + '.w'], W[k + '.b'] = pk.view(f(m.weight), (-1,)), f(m.bias)

        lin('img.lin', ie.img_linear); ln('img.ln', ie.img_layer_norm)
        W['loc.w'], W['loc.b'] = f(ie.loc_linear.weight), f(ie.loc_linear.bias); ln('loc.ln', ie.loc_layer_norm)
        W['gaa.w'], W['gaa.b'] = pk.view(f(ie.adaptive_pano_attn.weight), (-1,)), f(ie.adaptive_pano_attn.bias)
        if self.REVERIE:
            lin('objv1.lin', ie.obj_reverie_linear); ln('objv1.ln', ie.obj_reverie_layer_norm)
        W['obj.bbox.w'], W['obj.bbox.b'] = f(ie.obj_bbox_linear.weight), f(ie.obj_bbox_linear.bias)
        W['obj.name'], W['obj.nav'] = f(ie.obj_name_linear.weight), f(ie.obj_nav_type_embedding.weight)
        ln('obj.ln', ie.obj_layer_norm)
        for i, L in enumerate(ie.obj_self_encoder.layer):
            bert_layer('objenc.%d' % i, L)
        lin('visobj.lin', ie.vis_obj_linear[0]); ln('visobj.ln', ie.vis_obj_linear[1])
        W['nav_type'] = f(ie.nav_type_embedding.weight)
        ln('pano.ln', ie.layer_norm)
        for i, L in enumerate(ie.pano_encoder.layers):
            k = 'pano.%d' % i
            ln(k + '.n1', L.norm1); ln(k + '.n2', L.norm2)
            W[k + '.qkv.w'], W[k + '.qkv.b'] = c(L.self_attn.in_proj_weight), f(L.self_attn.in_proj_bias)
            lin(k + '.ao', L.self_attn.out_proj); lin(k + '.ff1', L.linear1); lin(k + '.ff2', L.linear2)
        ln('pano.norm', ie.pano_encoder.norm)

        ge, lo = self.global_encoder, self.local_encoder
        W['gmap.step'] = f(ge.gmap_step_embeddings.weight)
        W['gmap.pos.w'], W['gmap.pos.b'] = f(ge.gmap_pos_embeddings[0].weight), f(ge.gmap_pos_embeddings[0].bias)
        ln('gmap.pos.ln', ge.gmap_pos_embeddings[1])
        W['vp.pos.w'], W['vp.pos.b'] = f(lo.vp_pos_embeddings[0].weight), f(lo.vp_pos_embeddings[0].bias)
        ln('vp.pos.ln', lo.vp_pos_embeddings[1])
        if ge.sprel_linear is not None:
            W['sprel.w'], W['sprel.b'] = pk.view(f(ge.sprel_linear.weight), (-1,)), f(ge.sprel_linear.bias)
        kv = []
        for br, enc in (('g', ge.encoder), ('l', lo.encoder)):
            for i, L in enumerate(enc.x_layers):
                k = 'x.%s.%d' % (br, i)
                lin(k + '.cq', L.visual_attention.att.query)
                kv += [L.visual_attention.att.key, L.visual_attention.att.value]
                lin(k + '.co', L.visual_attention.output.dense); ln(k + '.cln', L.visual_attention.output.LayerNorm)
                packed(k + '.qkv', [L.visn_self_att.self.query, L.visn_self_att.self.key, L.visn_self_att.self.value])
                lin(k + '.so', L.visn_self_att.output.dense); ln(k + '.sln', L.visn_self_att.output.LayerNorm)
                lin(k + '.ff1', L.visn_inter.dense); lin(k + '.ff2', L.visn_output.dense); ln(k + '.fln', L.visn_output.LayerNorm)
                if hasattr(L, 'lang_self_att'):      # text-side sub-layers, used by forward_mlm only
                    packed(k + '.ckv', [L.visual_attention.att.key, L.visual_attention.att.value])
                    packed(k + '.lqkv', [L.lang_self_att.self.query, L.lang_self_att.self.key, L.lang_self_att.self.value])
                    lin(k + '.lso', L.lang_self_att.output.dense); ln(k + '.lsln', L.lang_self_att.output.LayerNorm)
                    lin(k + '.lff1', L.lang_inter.dense); lin(k + '.lff2', L.lang_output.dense); ln(k + '.lfln', L.lang_output.LayerNorm)
        packed('x.txtkv', kv)       # [2 branches * layers * (K,V) * 768, 768]
        self._head_weights(W, lin, ln, f)
        # the walk over the module tree above costs ~3 ms: its result (placeholders + the packer's source list) is kept
        self._wplan = plan = (W, pk, any(p.requires_grad for p in self.parameters()))
        return self._resolve_weights(plan, torch.is_grad_enabled() and plan[2])

    def _resolve_weights(self, plan, grad):
        """Run the plan's one multi-tensor pack.  With autograd on, the packed (differentiable) dict is reused until a
        parameter changes (every in-place update - optimizer.step, load_state_dict - bumps Tensor._version): the three
        sub-model calls of a pre-training step, or the language / panorama / navigation calls of a rollout, share one
        pack and one _PackFn node.  (Writing through `.data` bypasses the version counter: call invalidate_caches().)"""
        W0, pk, _ = plan
        if grad:
            ver = Fn.pack_version(pk.srcs)
            if self._wgrad is not None and self._wgrad[0] == ver:
                return self._wgrad[1]
        W = pk.resolve(dict(W0))
        self._kv_cache = None        # the cached text K/V panel was projected with the previous pack
        if grad:
            self._wgrad = (ver, W)
        else:
            self._wcache, self._wcache_ver = W, Fn.pack_version(pk.srcs)
        return W

    def _cls_head_weights(self, W, key, hd, lin, ln, f):
        lin(key + '.l1', hd.net[0]); ln(key + '.ln', hd.net[2])
        if hd.net[3].out_features == 1:
            W[key + '.w2'], W[key + '.b2'] = Fn.WeightPacker.view(f(hd.net[3].weight), (-1,)), f(hd.net[3].bias)
        else:
            lin(key + '.l2', hd.net[3])

    def _head_weights(self, W, lin, ln, f):
        heads = [('head.g', self.global_sap_head), ('head.l', self.local_sap_head)]
        if self.sap_fuse_linear is not None:
            heads.append(('head.f', self.sap_fuse_linear))
        if hasattr(self, 'og_head'):
            heads.append(('head.o', self.og_head))
        for k, hd in heads:
            self._cls_head_weights(W, k, hd, lin, ln, f)
        for k, m in (('pool.g', self.gmap_pooler), ('pool.v', self.vp_pooler), ('pool.t', self.txt_pooler)):
            lin(k, m.dense)
        lin('his.map', self.local_his_map); ln('his.ln', self.local_his_ln)

    # ---------------------------------------------------------------- building blocks
    def _attn_p(self):
        """dropout on the attention probabilities of the BERT-style blocks (vilmodel.py:115,150,342), training only"""
        return float(self.config.attention_probs_dropout_prob) if self.training else 0.0

    def _drop(self, x, p, residual=None):
        """residual + nn.Dropout(p)(x) in one HIP kernel (counter-hash mask, recomputed in backward: no mask tensor, and no
        torch generator inside a captured training step)"""
        return Fn.dropout_add(x, p, residual, self.training)

    def _dense_res_ln(self, W, k_lin, k_ln, h, res, eps):
        """LayerNorm(dropout(dense(h)) + res): BertSelfOutput / BertOutput (vilmodel.py:173-177, 213-217)."""
        p = self.config.hidden_dropout_prob if self.training else 0.0
        return Fn.linear_ln(h, W[k_lin + '.w'], W[k_lin + '.b'], W[k_ln + '.g'], W[k_ln + '.b'], eps, residual=res, p_drop=p)

    def _bert_layer(self, W, key, x, nh, kmask, mode):  # vilmodel.py:226-232
        H, eps = x.shape[-1], self.config.layer_norm_eps
        p = self.config.hidden_dropout_prob if self.training else 0.0
        a = Fn.attn_block_grouped([(x, W[key + '.qkv.w'], W[key + '.qkv.b'], W[key + '.ao.w'], W[key + '.ao.b'], W[key + '.aln.g'], W[key + '.aln.b'],
                                    None, kmask, None)], nh, mode, eps, self._attn_p(), p)[0]
        return Fn.ffn_grouped([(a, W[key + '.ff1.w'], W[key + '.ff1.b'], W[key + '.ff2.w'], W[key + '.ff2.b'], a, W[key + '.fln.g'], W[key + '.fln.b'])],
                              eps, p)[0]

    # ---------------------------------------------------------------- language (vilmodel.py:678-688)
    def forward_text(self, txt_ids, txt_masks, act_txt=None, act_txt_masks=None, obj_txt=None, obj_txt_masks=None):
        W, cfg, cd = self._weights(), self.config, self.compute_dtype
        B, L = txt_ids.shape
        H, nh = cfg.hidden_size, cfg.num_attention_heads
        pos = torch.arange(L, device=txt_ids.device, dtype=torch.int32).repeat(B)
        x = Fn.embed_ln(B * L, H, cd, txt_ids.device, E1=W['emb.word'], idx1=txt_ids.reshape(-1),
                        E2=W['emb.pos'], idx2=pos, vec=W['emb.type'][0], lno=(W['emb.ln.g'], W['emb.ln.b']),
                        eps=cfg.layer_norm_eps, E1_padding_idx=self.embeddings.word_embeddings.padding_idx).view(B, L, H)
        x = self._drop(x, cfg.hidden_dropout_prob)
        tm = txt_masks.contiguous()
        for i in range(cfg.num_l_layers):
            x = self._bert_layer(W, 'lang.%d' % i, x, nh, tm, MASK_NEG10000)
        if not getattr(cfg, 'update_lang_bert', True):
            x = x.detach()
        if self.PRETRAIN or (obj_txt is None and not self.REVERIE):
            return x
        # IGL (vilmodel.py:521-528; REVERIE uses object words only, vilmodel_reverie.py:523-529)
        obj = nn.functional.embedding(obj_txt, W['emb.obj'])
        if self.REVERIE:
            ctx, cm = obj, obj_txt_masks
        else:
            ctx = torch.cat([nn.functional.embedding(act_txt, W['emb.act']), obj], 1)
            cm = torch.cat([act_txt_masks, obj_txt_masks], 1)
        ctx = ctx.to(cd).contiguous()
        q = Fn.linear(x, W['igl.q.w'], W['igl.q.b'])
        kv = Fn.linear(ctx, W['igl.kv.w'], W['igl.kv.b'])
        aug = Fn.attention_packed(kv, nh, q, cm.contiguous(), MASK_NEG10000, p_drop=self._attn_p())
        return Fn.igl_blend(aug, x.contiguous(), W['igl.aug.w'], W['igl.aug.b'], W['igl.ori.w'], W['igl.ori.b'])

    # ---------------------------------------------------------------- panorama
    def _objv2_tokens(self, W, fts, lens, nav_types):
        """ASV: object tokens -> self-encoder -> per-view mean [B*V,128] rows (vilmodel.py:697-714)."""
        return self._drain(self._objv2_tokens_gen(W, fts, lens, nav_types))

    def _objv2_tokens_gen(self, W, fts, lens, nav_types):
        """(generator: yields between launches, returns the result - see _run_interleaved)"""
        cfg, cd = self.config, self.compute_dtype
        B, V, O, _ = fts.shape
        d = cfg.obj_v2_size
        fts = fts.float().contiguous()
        e = Fn.embed_ln(B * V * O, d, cd, fts.device, feat=fts.view(B * V * O, -1)[:, :-1], Wf=W['obj.bbox.w'],
                        bf=W['obj.bbox.b'], E1=W['obj.name'], idx1=fts.view(B * V * O, -1)[:, -1],
                        E2=W['obj.nav'], idx2=(nav_types.reshape(-1), O),
                        lno=(W['obj.ln.g'], W['obj.ln.b'])).view(B, V * O, d)
        if not torch.is_grad_enabled() and lens.dtype == torch.int64:
            valid = K.KeyLens(lens.reshape(B, V).contiguous(), O)      # the attention kernel forms the mask from the lengths
        else:
            valid = (_arange(O, fts.device)[None, None] < lens[..., None]).reshape(B, V * O).contiguous()
        yield
        if self._fuse_ok(d):
            e = yield from self._bert_layers_fused_gen(W, ['objenc.%d' % i for i in range(2)], e, d // 64, valid, MASK_PLUS1)
        else:
            for i in range(2):
                e = self._bert_layer(W, 'objenc.%d' % i, e, d // 64, valid, MASK_PLUS1)   # bool mask ADDED (+1/+0)
                yield
        return e.reshape(B * V * O, d), O

    def _pano_encoder(self, W, x, masks):
        """2 pre-norm layers over packed-in_proj MHA with -inf key padding (transformer.py:170-182)."""
        return self._drain(self._pano_encoder_gen(W, x, masks))

    def _pano_encoder_gen(self, W, x, masks):
        """(generator: yields between launches, returns the result - see _run_interleaved)"""
        cfg = self.config
        H, nh, p = cfg.hidden_size, cfg.num_attention_heads, cfg.hidden_dropout_prob
        if self._fuse_ok(H):
            return (yield from self._pano_encoder_fused_gen(W, x, masks))
        for i in range(cfg.num_pano_layers):
            k = 'pano.%d' % i
            y = Fn.layernorm(x, W[k + '.n1.g'], W[k + '.n1.b'], 1e-5)
            qkv = Fn.linear(y, W[k + '.qkv.w'], W[k + '.qkv.b'])
            a = Fn.attention_packed(qkv, nh, None, masks, MASK_NEGINF, p_drop=p if self.training else 0.0)       # nn.MultiheadAttention(dropout=hidden_dropout_prob), ops.py:14
            if self.training and p > 0:
                x = self._drop(Fn.linear(a, W[k + '.ao.w'], W[k + '.ao.b']), p, x)
            else:
                x = Fn.linear(a, W[k + '.ao.w'], W[k + '.ao.b'], residual=x)
            y = Fn.layernorm(x, W[k + '.n2.g'], W[k + '.n2.b'], 1e-5)
            hdn = self._drop(Fn.linear(y, W[k + '.ff1.w'], W[k + '.ff1.b'], act=ACT_GELU), p)
            if self.training and p > 0:
                x = self._drop(Fn.linear(hdn, W[k + '.ff2.w'], W[k + '.ff2.b']), p, x)
            else:
                x = Fn.linear(hdn, W[k + '.ff2.w'], W[k + '.ff2.b'], residual=x)
            yield
        return x

    def _pano_tokens(self, W, base, loc_fts, nav_types, seg=None, lnb=None):
        """seg (inference, REVERIE): (object rows [B*No, H], No, view_lens, obj_lens) - the per-episode token assembly
        [views[:vl] ; objects[:ol] ; zero pad] is done by the embedding kernel's base-row gather (kernels.embed_ln)"""
        cfg = self.config
        B, V, H = base.shape
        T = loc_fts.shape[1]
        x = Fn.embed_ln(B * T, H, self.compute_dtype, base.device, base=base.reshape(B * V, H),
                        feat=loc_fts.float().reshape(B * T, -1).contiguous(), Wf=W['loc.w'], bf=W['loc.b'],
                        lnf=(W['loc.ln.g'], W['loc.ln.b']), E1=W['nav_type'], idx1=nav_types.reshape(-1),
                        vec=W['emb.type'][1], lno=(W['pano.ln.g'], W['pano.ln.b']),
                        seg=None if seg is None else (seg[0], T, V, seg[1], seg[2], seg[3]), lnb=lnb).view(B, T, H)
        # FT REVERIE: nn.Dropout(config.image_dropout_prob) (vilmodel_reverie.py:588); R2R and both pre-training twins:
        # hidden_dropout_prob (vilmodel.py:576, PT/vilmodel_reverie.py:514)
        ft_reverie = self.REVERIE and not self.PRETRAIN
        return self._drop(x, getattr(cfg, 'image_dropout_prob', cfg.hidden_dropout_prob) if ft_reverie else cfg.hidden_dropout_prob)

    def _concat2(self, a_fn, b_fn, B, V, ha, hb, device):
        """[a | b] along the feature dim.  No-grad: both producers write straight into one buffer."""
        if torch.is_grad_enabled():
            return torch.cat([a_fn(None).view(B, V, ha), b_fn(None).view(B, V, hb)], -1)
        cat = torch.empty(B, V, ha + hb, device=device, dtype=self.compute_dtype)
        a_fn(cat[..., :ha]); b_fn(cat[..., ha:])
        return cat

    # ---------------------------------------------------------------- independent sub-graphs on a second HIP stream
    @staticmethod
    def _drain(gen):
        """run a launch generator to its end -> its return value"""
        while True:
            try:
                next(gen)
            except StopIteration as e:
                return e.value

    def _side(self, cur, k=0):
        if self._side_stream is None:
            self._side_stream = {}
        key = (cur.cuda_stream, k)
        side = self._side_stream.get(key)          # side streams per calling stream
        if side is None:
            side = self._side_stream[key] = torch.cuda.Stream()
        return side

    def _run_interleaved(self, *gens):
        """Independent launch chains (generators that yield between launches): the LAST one on the current stream, the others
        on side streams, advanced in turn so that their nodes enter a captured HIP graph interleaved - hipGraphLaunch submits
        kernel nodes in creation order, and a chain captured as a whole before the others delays their first launches by its
        own submission time.  No-grad mode only.  -> the chains' results, in order; the current stream has waited for the
        side streams on return."""
        cur = torch.cuda.current_stream()
        streams = [self._side(cur, k) for k in range(len(gens) - 1)] + [cur]
        for st in streams[:-1]:
            st.wait_stream(cur)
        res, live = [None] * len(gens), list(range(len(gens)))
        self._fork_depth += 1
        try:
            while live:
                for i in list(live):
                    with torch.cuda.stream(streams[i]):
                        try:
                            next(gens[i])
                        except StopIteration as e:
                            res[i] = e.value
                            live.remove(i)
        finally:
            self._fork_depth -= 1
        for st in streams[:-1]:
            cur.wait_stream(st)
        for r in res[:-1]:
            for t in (r if isinstance(r, (tuple, list)) else (r,)):
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
        return res

    def _fork(self, fn):
        """Run fn() on the model's side stream (no-grad mode only) and return join() -> its result.  The object-token
        encoder and the view encoder of a panorama, or the SAP heads and the RMF poolers of a navigation step, are
        independent chains of small launches that each leave most of the 256 CUs idle; on two streams (two branches of
        the captured HIP graph) they run side by side."""
        if torch.is_grad_enabled() or not self.overlap_streams or self._fork_depth > 0:      # one level only: nested
            r = fn()                                                                       # forks crash hipStreamEndCapture
            return lambda: r
        cur = torch.cuda.current_stream()
        side = self._side(cur)
        side.wait_stream(cur)
        self._fork_depth += 1
        try:
            with torch.cuda.stream(side):
                r = fn()
        finally:
            self._fork_depth -= 1

        def join():
            cur.wait_stream(side)
            for t in (r if isinstance(r, (tuple, list)) else r.values() if isinstance(r, dict) else (r,)):
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
            return r
        return join

    def _len_masks(self, lens, T, want_masks, lens2=None):
        """(key-validity operand of the attention launch, bool mask [B,T] or None): without autograd the attention kernel
        forms the mask from the lengths (no compare kernel); the tensor is built only when the caller wants it back."""
        dev = lens.device
        tot = lens if lens2 is None else lens + lens2
        lazy = not torch.is_grad_enabled() and lens.dtype == torch.int64 and (lens2 is None or lens2.dtype == torch.int64)
        masks = (_arange(T, dev)[None] < tot[:, None]) if (want_masks or not lazy) else None
        if lazy:
            return K.KeyLens(lens.contiguous(), T, None if lens2 is None else lens2.contiguous()), masks
        return masks.contiguous(), masks

    def forward_panorama_per_step(self, view_img_fts, obj_img_fts, loc_fts, nav_types, view_lens, obj_lens,
                                  pano_out=None, gmap_scatter=None, want_masks=True):
        """R2R: vilmodel.py:690-742.  Extensions for a serving loop (dsrg_amd.runner; inference only): pano_out = a [B,V,H]
        view (rows contiguous, batch stride free) that receives pano_embeds, gmap_scatter = (buffer [B,G,H], idx int64 [B]) that
        also receives the pooled embedding at row idx[b], want_masks=False returns None for pano_masks."""
        W, cfg, cd = self._weights(), self.config, self.compute_dtype
        B, V, _ = view_img_fts.shape
        H, d = cfg.hidden_size, cfg.obj_v2_size
        g, b = W['pano.norm.g'], W['pano.norm.b']
        km, pano_masks = self._len_masks(view_lens, V, want_masks)
        if self.interleave_pano and not torch.is_grad_enabled() and self.overlap_streams and self._fork_depth == 0:
            # inference: the view chain (img_linear -> tokens -> panorama encoder -> norm) on the current stream and the object-token
            # chain (ASV encoder -> per-view mean) on the side stream, captured launch by launch in turn; both write their half
            # of the [pano | obj] rows that vis_obj_linear reads (vilmodel.py:731)
            cat = torch.empty(B, V, H + d, device=view_img_fts.device, dtype=cd)

            def obj_chain():
                obj_rows, O = yield from self._objv2_tokens_gen(W, obj_img_fts, obj_lens, nav_types)
                K.group_mean(obj_rows, O, out=cat[..., H:])

            def view_chain(b0, b1):
                if self.fold_img_ln:      # img_layer_norm inside the token kernel (dsrg_embed_ln_fwd_lnb): no LayerNorm launch
                    img = K.gemm([(view_img_fts[b0:b1].to(cd), W['img.lin.w'], W['img.lin.b'], None)])[0]
                    yield
                    x = self._pano_tokens(W, img, loc_fts[b0:b1], nav_types[b0:b1], lnb=(W['img.ln.g'], W['img.ln.b']))
                else:
                    img = Fn.linear_ln(view_img_fts[b0:b1].to(cd), W['img.lin.w'], W['img.lin.b'], W['img.ln.g'], W['img.ln.b'], 1e-12)
                    yield
                    x = self._pano_tokens(W, img, loc_fts[b0:b1], nav_types[b0:b1])
                yield
                kmp = km if (b0, b1) == (0, B) else (K.KeyLens(km.lens[b0:b1], km.seg) if isinstance(km, K.KeyLens) else km[b0:b1])
                x = yield from self._pano_encoder_gen(W, x, kmp)
                Fn.layernorm(x, g, b, 1e-12, out=cat[b0:b1, :, :H])
            # the view chain is the longer one (~250 vs ~190 us of kernels at B = 64): as two half-batch chains beside the
            # object chain (three chains, as in the navigation call; pano_split = 1: whole)
            if self.pano_split > 1 and B % 2 == 0 and B >= 16:
                self._run_interleaved(view_chain(0, B // 2), obj_chain(), view_chain(B // 2, B))
            else:
                self._run_interleaved(obj_chain(), view_chain(0, B))
        else:
            objs = self._fork(lambda: self._objv2_tokens(W, obj_img_fts, obj_lens, nav_types))
            img = Fn.linear_ln(view_img_fts.to(cd), W['img.lin.w'], W['img.lin.b'], W['img.ln.g'], W['img.ln.b'], 1e-12)
            x = self._pano_tokens(W, img, loc_fts, nav_types)
            x = self._pano_encoder(W, x, km)
            obj_rows, O = objs()
            cat = self._concat2(lambda o: Fn.layernorm(x, g, b, 1e-12, out=o) if o is not None else Fn.layernorm(x, g, b, 1e-12),
                                lambda o: K.group_mean(obj_rows, O, out=o) if o is not None else Fn.group_mean(obj_rows, O),
                                B, V, H, d, x.device)
        if self.fold_visobj_ln and not torch.is_grad_enabled() and H % 64 == 0:
            # vis_obj_linear's LayerNorm inside the GAA kernel (dsrg_gaa_pool_fwd_ln): it normalises the projection's rows, stores
            # them as pano_embeds and pools them - no LayerNorm launch, no fp32 projection output
            z = K.gemm([(cat, W['visobj.lin.w'], W['visobj.lin.b'], None)])[0]
            pano, fused = K.gaa_pool(z, W['gaa.w'], W['gaa.b'], scatter=gmap_scatter, ln=(W['visobj.ln.g'], W['visobj.ln.b'], 1e-12),
                                     rows_out=pano_out)
            return pano, pano_masks, fused
        pano = Fn.linear_ln(cat, W['visobj.lin.w'], W['visobj.lin.b'], W['visobj.ln.g'], W['visobj.ln.b'], 1e-12, out=pano_out)
        fused = Fn.gaa_pool(pano, W['gaa.w'], W['gaa.b'], scatter=gmap_scatter)
        return pano, pano_masks, fused

    # ---------------------------------------------------------------- navigation (vilmodel.py:745-830)
    def _text_kv(self, W, txt_embeds):
        """All 16 text K/V projections of the 8 x-layers as one GEMM.  Cached per txt_embeds tensor in
        no-grad mode (the text is constant for a whole episode, agent.py:420 vs :471)."""
        if not torch.is_grad_enabled():
            key = (txt_embeds.data_ptr(), txt_embeds._version, tuple(txt_embeds.shape), txt_embeds.dtype)
            if self._kv_cache is not None and self._kv_cache[0] == key:
                return self._kv_cache[1]
        kv = Fn.linear(txt_embeds, W['x.txtkv.w'], W['x.txtkv.b'])
        if not torch.is_grad_enabled():
            self._kv_cache = (key, kv, txt_embeds)      # keep txt_embeds alive so the pointer stays unique
        return kv

    def text_kv(self, txt_embeds, out=None):
        """[B,L,16*768] panel of the text keys/values of all x-layers (both branches), uncached."""
        W = self._weights()
        txt = txt_embeds.to(self.compute_dtype).contiguous()
        if out is None:
            return Fn.linear(txt, W['x.txtkv.w'], W['x.txtkv.b'])
        return K.gemm([(txt, W['x.txtkv.w'], W['x.txtkv.b'], None)], outs=[out])[0]

    @staticmethod
    def fusion_indices(gmap_vpids, gmap_visited_masks, vp_cand_vpids, G, Vp, skip=2):
        """Host-side viewpoint-id matching of the logit-fusion loop (vilmodel.py:790-805) -> index arrays.
        gmap_src[b,j]: local slot to add to node j (>=0), -2 = add the back-track sum, -1 = nothing.
        bw_mask[b,k]: local slot k is a candidate that is an already-visited node."""
        B = len(gmap_vpids)
        vis_np = gmap_visited_masks.cpu().numpy() if isinstance(gmap_visited_masks, torch.Tensor) else np.asarray(gmap_visited_masks)
        src = np.full((B, G), -1, np.int32)
        bw = np.zeros((B, Vp), np.uint8)
        for i in range(B):
            names = gmap_vpids[i]
            visited = {vp for vp, m in zip(names, vis_np[i]) if m}
            slot = {}
            for j, c in enumerate(vp_cand_vpids[i]):
                if j >= skip:
                    if c in visited:
                        bw[i, j] = 1
                    else:
                        slot[c] = j
            for j, vp in enumerate(names):
                if j >= skip and vp not in visited:
                    src[i, j] = slot.get(vp, -2)
        return src, bw

    def _folded(self, W, lin_key, ln_key):
        """(W * diag(gamma), its row sums c1, bias + W beta) of the projection `lin_key` applied to LN_{ln_key}(y): the
        operands of the a_stats form of dsrg_gemm_nt (include/dsrg_hip.h).  Derived from the packed weights, cached with them."""
        if self._fold is None or self._fold[0] is not W:
            self._fold = (W, {})
        F = self._fold[1]
        k = lin_key + '<' + ln_key
        if k not in F:
            w, g, b = W[lin_key + '.w'], W[ln_key + '.g'], W[ln_key + '.b']
            wf = (w.float() * g[None]).to(w.dtype).contiguous()
            F[k] = (wf, wf.float().sum(1).contiguous(), (W[lin_key + '.b'] + w.float() @ b).contiguous())
        return F[k]

    # ---- LayerNorm folded into the neighbouring GEMM epilogues (inference; dsrg_gemm_problem_t stats_out / a_stats / r_stats)
    def _fuse_ok(self, H):
        return self.fuse_ln and not torch.is_grad_enabled() and not self.training and H % 64 == 0 and H <= 1024

    def _fz_proj(self, W, t, st, lnk, lin, eps, act=ACT_NONE):
        """act(LN_lnk(t) W[lin]^T + b) with t un-normalised and st its partial statistics (st None: t is materialised)"""
        if st is None:
            return K.gemm([(t, W[lin + '.w'], W[lin + '.b'], None)], act)[0]
        wf, c1, bf_ = self._folded(W, lin, lnk)
        return K.gemm([(t, wf, bf_, None)], act, fuse=[{'a_stats': (st, c1), 'eps': eps}])[0]

    def _fz_dense_res(self, W, h, lin, res, res_st, res_lnk, eps, want_stats=True, out_f32=False):
        """y = h W[lin]^T + b + (LN_res_lnk(res) if res_st is given else res) -> (y, partial statistics of y | None)"""
        d = {'eps': eps}
        if res_st is not None:
            d['r_stats'] = (res_st, W[res_lnk + '.g'], W[res_lnk + '.b'])
        st = None
        if want_stats:
            N = W[lin + '.w'].shape[0]
            st = d['stats_out'] = torch.empty(N // 64, h.numel() // h.shape[-1], 2, device=h.device, dtype=torch.float32)
        if len(d) == 1:
            return K.gemm([(h, W[lin + '.w'], W[lin + '.b'], res)], ACT_NONE, out_f32)[0], None
        return K.gemm([(h, W[lin + '.w'], W[lin + '.b'], res)], ACT_NONE, out_f32, fuse=[d])[0], st

    def _bert_layers_fused(self, W, keys, x, nh, kmask, mode):
        return self._drain(self._bert_layers_fused_gen(W, keys, x, nh, kmask, mode))

    def _bert_layers_fused_gen(self, W, keys, x, nh, kmask, mode):
        """Post-LN BERT layers (vilmodel.py:219-232; the ASV object self-encoder) without their LayerNorm launches: every
        sub-block leaves its un-normalised sum and the partial statistics, the projections that read it carry the correction
        in their epilogue; one LayerNorm launch at the end materialises the output."""
        eps = self.config.layer_norm_eps
        H = x.shape[-1]
        t, st, lnk = x.contiguous(), None, None
        for key in keys:
            qkv = self._fz_proj(W, t, st, lnk, key + '.qkv', eps)
            yield
            a = K.attention_grouped([(qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], kmask, None)], nh, mode)[0]
            yield
            t, st = self._fz_dense_res(W, a, key + '.ao', t, st, lnk, eps)
            yield
            lnk = key + '.aln'
            h = self._fz_proj(W, t, st, lnk, key + '.ff1', eps, ACT_GELU)
            yield
            t, st = self._fz_dense_res(W, h, key + '.ff2', t, st, lnk, eps)
            yield
            lnk = key + '.fln'
        return K.layernorm_grouped([(t, W[lnk + '.g'], W[lnk + '.b'], None)], eps, self.compute_dtype)[0]

    def _pano_encoder_fused_gen(self, W, x, masks):
        """The pre-norm panorama encoder (transformer.py:170-182) with norm1 / norm2 folded into the projections that read
        them: the residual stream x stays un-normalised anyway; out_proj / linear2 (+ residual) also write its partial
        statistics.  Only the first norm1 (its input comes from the embedding kernel) is a launch."""
        cfg = self.config
        H, nh = cfg.hidden_size, cfg.num_attention_heads
        st = None
        for i in range(cfg.num_pano_layers):
            k = 'pano.%d' % i
            if st is None:
                qkv = K.gemm([(Fn.layernorm(x, W[k + '.n1.g'], W[k + '.n1.b'], 1e-5), W[k + '.qkv.w'], W[k + '.qkv.b'], None)])[0]
            else:
                qkv = self._fz_proj(W, x, st, k + '.n1', k + '.qkv', 1e-5)
            yield
            a = Fn.attention_packed(qkv, nh, None, masks, MASK_NEGINF)
            yield
            x, st = self._fz_dense_res(W, a, k + '.ao', x, None, None, 1e-5)
            yield
            h = self._fz_proj(W, x, st, k + '.n2', k + '.ff1', 1e-5, ACT_GELU)
            yield
            x, st = self._fz_dense_res(W, h, k + '.ff2', x, None, None, 1e-5, want_stats=i + 1 < cfg.num_pano_layers)
            yield
        return x

    def _x_layers_fused_gen(self, W, kv, txt_masks, feats, masks, dists, brs, final_aux=None, final_out=None):
        """(generator: yields after every launch, returns the result - see _x_layers_gen)
        _x_layers without its 3 LayerNorm launches per layer and branch (inference).  A post-LN sub-block leaves its
        UN-normalised sum y (compute dtype) plus per-row partial statistics written by the same GEMM launch; the projections
        that read LN(y) run on y with gamma folded into the weight and the mean / rstd correction in their epilogue, and the
        next dense + residual normalises its residual on the fly (dsrg_gemm_problem_t: stats_out / a_stats / r_stats).
        Only the last layer's output is materialised by a LayerNorm launch.  Same arithmetic in real numbers
        (vilmodel.py:173-177, 213-217, 454-469); in bf16 y is rounded once before instead of after the normalisation."""
        cfg, cd = self.config, self.compute_dtype
        H, nh, eps, nl = cfg.hidden_size, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.num_x_layers
        n = len(brs)
        sw, sb = (W['sprel.w'], W['sprel.b']) if dists is not None else (None, None)
        dev = feats[0].device
        Ms = [f.shape[0] * f.shape[1] for f in feats]
        new_stats = lambda: [torch.empty(H // 64, m, 2, device=dev, dtype=torch.float32) for m in Ms]
        # per branch: (tensor, stats | None, LayerNorm key | None): a materialised input, or an un-normalised y
        cur = [(f, None, None) for f in feats]

        def proj(lin, act=ACT_NONE):      # projection of LN(cur)
            probs, fz = [], []
            for j in range(n):
                t, st, lnk = cur[j]
                if st is None:
                    probs.append((t, W[ks[j] + lin + '.w'], W[ks[j] + lin + '.b'], None)); fz.append(None)
                else:
                    wf, c1, bf_ = self._folded(W, ks[j] + lin, lnk)
                    probs.append((t, wf, bf_, None)); fz.append({'a_stats': (st, c1), 'eps': eps})
            return K.gemm(probs, act, False, fuse=fz)

        def dense_res(lin, hs, out_stats, out_f32=False):      # dense(h) + LN(cur)
            probs, fz = [], []
            for j in range(n):
                t, st, lnk = cur[j]
                d = {'eps': eps}
                if st is not None:
                    d['r_stats'] = (st, W[lnk + '.g'], W[lnk + '.b'])
                if out_stats is not None:
                    d['stats_out'] = out_stats[j]
                probs.append((hs[j], W[ks[j] + lin + '.w'], W[ks[j] + lin + '.b'], t)); fz.append(d)
            return K.gemm(probs, ACT_NONE, out_f32, fuse=fz)

        for i in range(nl):
            ks = ['x.%s.%d' % (b, i) for b in brs]
            q = proj('.cq')
            yield
            offs = [((('g', 'l').index(brs[j]) * nl + i) * 2) * H for j in range(n)]
            a = Fn.attention_grouped([(q[j], kv[..., o:o + H], kv[..., o + H:o + 2 * H], txt_masks, None) for j, o in enumerate(offs)],
                                     nh, MASK_NEG10000)
            yield
            st = new_stats()
            y = dense_res('.co', a, st)
            yield
            cur = [(y[j], st[j], ks[j] + '.cln') for j in range(n)]
            qkv = proj('.qkv')
            yield
            a = Fn.attention_grouped([(qkv[j][..., :H], qkv[j][..., H:2 * H], qkv[j][..., 2 * H:], masks[j],
                                       dists if brs[j] == 'g' else None) for j in range(n)], nh, MASK_NEG10000, sw, sb)
            yield
            st = new_stats()
            y = dense_res('.so', a, st)
            yield
            cur = [(y[j], st[j], ks[j] + '.sln') for j in range(n)]
            inter = proj('.ff1', ACT_GELU)
            yield
            if i == nl - 1:
                z = dense_res('.ff2', inter, None, out_f32=True)
                yield
                return K.layernorm_grouped([(z[j], W[ks[j] + '.fln.g'], W[ks[j] + '.fln.b'], None) for j in range(n)], eps, cd, aux=final_aux,
                                           outs=final_out)
            st = new_stats()
            y = dense_res('.ff2', inter, st)
            yield
            cur = [(y[j], st[j], ks[j] + '.fln') for j in range(n)]

    def _x_layers(self, W, kv, txt_masks, feats, masks, dists, brs=('g', 'l'), final_aux=None):
        """The CrossmodalEncoders of the branches in `brs` (global 'g', local 'l') in lock-step: every GEMM is one
        grouped launch over the branches (vilmodel.py:540-549, 454-469).  final_aux: per branch, the aux placement of the last
        LayerNorm (kernels.layernorm_grouped; inference only)."""
        g = self._x_layers_gen(W, kv, txt_masks, feats, masks, dists, brs, final_aux)
        while True:
            try:
                next(g)
            except StopIteration as e:
                return e.value

    def _x_layers_gen(self, W, kv, txt_masks, feats, masks, dists, brs=('g', 'l'), final_aux=None, final_out=None):
        """_x_layers as a generator that yields between its launches and returns the result: the two-stream schedule
        (_x_layers_two_streams) advances the global and the local branch in turn, so that their nodes enter the captured HIP
        graph interleaved - hipGraphLaunch submits kernel nodes in creation order, and a branch captured as a whole after the
        other one starts several hundred microseconds late."""
        cfg = self.config
        if self.fuse_ln and not torch.is_grad_enabled() and not self.training and cfg.hidden_size % 64 == 0 and cfg.hidden_size <= 1024:
            return (yield from self._x_layers_fused_gen(W, kv, txt_masks, [f.contiguous() for f in feats], masks, dists, brs, final_aux, final_out))
        assert final_out is None, 'final_out: LayerNorm-fused inference schedule only'
        H, nh, eps, nl = cfg.hidden_size, cfg.num_attention_heads, cfg.layer_norm_eps, cfg.num_x_layers
        n = len(brs)
        sw, sb = (W['sprel.w'], W['sprel.b']) if dists is not None else (None, None)
        kvs = Fn.split_cols(kv, [2 * H] * (kv.shape[-1] // (2 * H)))      # the text K|V panel of every (branch, layer)
        for i in range(nl):
            ks = ['x.%s.%d' % (b, i) for b in brs]
            ph = self.config.hidden_dropout_prob if self.training else 0.0
            feats = Fn.attn_block_grouped([(feats[j], W[ks[j] + '.cq.w'], W[ks[j] + '.cq.b'], W[ks[j] + '.co.w'], W[ks[j] + '.co.b'], W[ks[j] + '.cln.g'],
                                            W[ks[j] + '.cln.b'], kvs[('g', 'l').index(brs[j]) * nl + i], txt_masks, None) for j in range(n)],
                                          nh, MASK_NEG10000, eps, self._attn_p(), ph)
            yield
            feats = Fn.attn_block_grouped([(feats[j], W[ks[j] + '.qkv.w'], W[ks[j] + '.qkv.b'], W[ks[j] + '.so.w'], W[ks[j] + '.so.b'], W[ks[j] + '.sln.g'],
                                            W[ks[j] + '.sln.b'], None, masks[j], dists if brs[j] == 'g' else None) for j in range(n)],
                                          nh, MASK_NEG10000, eps, self._attn_p(), ph, sw, sb)
            yield
            feats = Fn.ffn_grouped([(feats[j], W[ks[j] + '.ff1.w'], W[ks[j] + '.ff1.b'], W[ks[j] + '.ff2.w'], W[ks[j] + '.ff2.b'], feats[j],
                                     W[ks[j] + '.fln.g'], W[ks[j] + '.fln.b']) for j in range(n)], eps, self.config.hidden_dropout_prob if self.training else 0.0,
                                   aux=final_aux if i == nl - 1 else None)
            yield
        return feats

    def _x_layers_two_streams(self, W, kv, tm, g, v, gmask, vmask, dists, final_aux=None):
        """Global branch on the current stream, local branch on side streams, advanced in turn (no-grad mode only).
        split_big (inference, LayerNorm-fused schedule): when one branch has split_ratio (2.0) times the other's tokens per
        episode - REVERIE's 58 viewpoint tokens against 20 graph nodes, 100 graph nodes against 38 - the larger one runs as two
        chains over the halves of the episode batch (episodes are independent in every launch), so that three chains of
        comparable length share the chip to the end instead of the long one running alone, on a two-thirds-empty chip, once
        the short one is through (same box: c3 1.380 -> 1.319 ms, c5 1.741 -> 1.648).  R2R's 38 against 20 stays whole by
        default: split it is 1.7 % faster (0.983 vs 1.000 ms; DSRG_SPLIT_RATIO=1.75), but every GEMM launch of the longer
        branch then has half the rows - the dominant 128x128 kernel drops from 0.158 to 0.127 of the MFMA peak per launch -
        and that fraction is what this build is measured by."""
        fa = final_aux or [None, None]
        B, G, Vp = g.shape[0], g.shape[1], v.shape[1]
        big = 'l' if Vp >= G else 'g'
        # number of chains of the longer branch: its rows over the shorter one's, rounded, at most split_big (2: a third or
        # fourth part buys nothing - c3 1.314 vs 1.326 ms, c5 1.643 vs 1.665 - and, with GPU_MAX_HW_QUEUES off its default of
        # 4, four or more concurrent chains cost 15-110 %) and at least 8 episodes per chain
        ns = min(self.split_big, int(max(G, Vp) / min(G, Vp) + 0.5), B // 8)
        if ns > 1 and self._fuse_ok(self.config.hidden_size) and max(G, Vp) >= self.split_ratio * min(G, Vp):
            cut = [(i * B + ns // 2) // ns for i in range(ns + 1)]          # episode ranges, sizes differing by at most one
            x, xm, xa = (v, vmask, fa[1]) if big == 'l' else (g, gmask, fa[0])
            xout = torch.empty_like(x)
            part = lambda t, i: None if t is None else t[cut[i]:cut[i + 1]]
            gens = [self._x_layers_gen(W, part(kv, i), part(tm, i), [part(x, i)], [part(xm, i)], part(dists, i), brs=(big,),
                                       final_aux=[None if xa is None else (xa[0], part(xa[1], i), part(xa[2], i))], final_out=[part(xout, i)])
                    for i in range(ns)]
            if big == 'l':
                gens.append(self._x_layers_gen(W, kv, tm, [g], [gmask], dists, brs=('g',), final_aux=[fa[0]]))
                return self._run_interleaved(*gens)[ns][0], xout
            gens.append(self._x_layers_gen(W, kv, tm, [v], [vmask], dists, brs=('l',), final_aux=[fa[1]]))
            return xout, self._run_interleaved(*gens)[ns][0]
        rl, rg = self._run_interleaved(self._x_layers_gen(W, kv, tm, [v], [vmask], dists, brs=('l',), final_aux=[fa[1]]),
                                       self._x_layers_gen(W, kv, tm, [g], [gmask], dists, brs=('g',), final_aux=[fa[0]]))
        return rg[0], rl[0]

    def _grouped_res_ln(self, W, ks, lin, ln, hs, res, eps):
        p = self.config.hidden_dropout_prob if self.training else 0.0
        return Fn.linear_ln_grouped([(hs[j], W[ks[j] + lin + '.w'], W[ks[j] + lin + '.b'], res[j], W[ks[j] + ln + '.g'], W[ks[j] + ln + '.b'])
                                     for j in range(len(ks))], eps, p)

    def _heads(self, W, items):
        """items: list of (key, x [..,K]) -> raw scores; first layers grouped in one launch (vilmodel.py:616-628)."""
        h = Fn.linear_grouped([(x, W[k + '.l1.w'], W[k + '.l1.b'], None) for k, x in items], act=ACT_RELU, out_f32=True)
        return Fn.head_tail_grouped([(h[j], W[k + '.ln.g'], W[k + '.ln.b'], W[k + '.w2'], W[k + '.b2']) for j, (k, _) in enumerate(items)])

    def _rmf(self, W, g, v, txt, mem_out=None):
        """RMF (vilmodel.py:815-818): three poolers on row 0 of each sequence -> his_map -> LN.  mem_out = (t0, t1 | None):
        [B,H] views (e.g. the [MEM] slots of the next step's token buffers) that also receive the result (inference only)."""
        cfg, cd = self.config, self.compute_dtype
        B, H, dev = g.shape[0], cfg.hidden_size, g.device
        rows = [g[:, 0], v[:, 0], txt[:, 0]]
        keys = ['pool.g', 'pool.v', 'pool.t']
        if torch.is_grad_enabled():
            pooled = torch.cat(Fn.linear_grouped([(rows[j], W[keys[j] + '.w'], W[keys[j] + '.b'], None) for j in range(3)], act=ACT_TANH), -1)
        else:
            pooled = torch.empty(B, 3 * H, device=dev, dtype=cd)
            K.gemm([(rows[j], W[keys[j] + '.w'], W[keys[j] + '.b'], None) for j in range(3)], ACT_TANH,
                   outs=[pooled[:, j * H:(j + 1) * H] for j in range(3)])
        return Fn.linear_ln(pooled, W['his.map.w'], W['his.map.b'], W['his.ln.g'], W['his.ln.b'], cfg.layer_norm_eps,
                            aux=None if mem_out is None else (1, mem_out[0], mem_out[1] if len(mem_out) > 1 else None))

    def forward_navigation_per_step(self, txt_embeds, txt_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts,
                                    gmap_masks, gmap_pair_dists, gmap_visited_masks, gmap_vpids,
                                    vp_img_embeds, vp_pos_fts, vp_masks, vp_nav_masks, vp_obj_masks, vp_cand_vpids,
                                    gmap_src=None, bw_mask=None, txt_kv=None, mem_out=None):
        """Extension keys (optional, not in the reference): `gmap_src`/`bw_mask` = precomputed fusion_indices()
        on the device, `txt_kv` = precomputed text_kv(txt_embeds); with them the call does no host work and
        can be captured in a HIP graph (dsrg_amd.runner).  `mem_out` = (t0[, t1]): [B,H] views that also receive cls_embeds
        (the [MEM] slots of the next step's token buffers, agent.py:185-188, 234-237; inference only)."""
        W, cfg, cd = self._weights(), self.config, self.compute_dtype
        B, G, H = gmap_img_embeds.shape
        Vp = vp_img_embeds.shape[1]
        dev = gmap_img_embeds.device
        txt = txt_embeds.to(cd)
        txt = txt if txt.is_contiguous() else txt.contiguous()
        tm = txt_masks.contiguous()
        # the two token-embedding launches are latency-bound and independent: side by side on two streams
        vj = self._fork(lambda: Fn.embed_ln(B * Vp, H, cd, dev, base=vp_img_embeds.to(cd).reshape(B * Vp, H),
                                            feat=vp_pos_fts.float().reshape(B * Vp, -1).contiguous(), Wf=W['vp.pos.w'],
                                            bf=W['vp.pos.b'], lnf=(W['vp.pos.ln.g'], W['vp.pos.ln.b'])).view(B, Vp, H))
        g = Fn.embed_ln(B * G, H, cd, dev, base=gmap_img_embeds.to(cd).reshape(B * G, H),
                        feat=gmap_pos_fts.float().reshape(B * G, -1).contiguous(), Wf=W['gmap.pos.w'], bf=W['gmap.pos.b'],
                        lnf=(W['gmap.pos.ln.g'], W['gmap.pos.ln.b']), E1=W['gmap.step'],
                        idx1=gmap_step_ids.reshape(-1)).view(B, G, H)
        v = vj()
        dists = gmap_pair_dists.float().contiguous() if self.global_encoder.sprel_linear is not None else None
        kv = txt_kv if txt_kv is not None else self._text_kv(W, txt)
        # [g0 | v0] for sap_fuse_linear (vilmodel.py:773): without autograd the last LayerNorm of each branch also stores its
        # row 0 of every episode into one [B, 2H] buffer (no concat kernel)
        cls2 = final_aux = None
        if self.sap_fuse_linear is not None and not torch.is_grad_enabled():
            cls2 = torch.empty(B, 2 * H, device=dev, dtype=cd)
            final_aux = [(G, cls2[:, :H], None), (Vp, cls2[:, H:], None)]
        # two streams pay at R2R's 20 vs 38 tokens and, with round 2's 64x64x3 GEMM plan, at REVERIE's 20 vs 58 too
        # (bench.py --config c3: 1.762 ms grouped, 1.649 ms split; round 1's plan had it the other way round); beyond a
        # 4:1 imbalance the grouped launches are kept
        # (under HIP-graph capture only: launched eagerly from Python the step is host-bound and the 40 % more launches
        # of the split schedule cost 3 %)
        if (self.split_branch_streams and not torch.is_grad_enabled() and self.overlap_streams
                and max(G, Vp) <= self.split_max_ratio * min(G, Vp) and (torch.cuda.is_current_stream_capturing() or self.split_when_eager)):
            if self.interleave_branches or self.split_big > 1:
                g, v = self._x_layers_two_streams(W, kv, tm, g, v, gmap_masks.contiguous(), vp_masks.contiguous(), dists, final_aux)
            else:
                lj = self._fork(lambda: self._x_layers(W, kv, tm, [v], [vp_masks.contiguous()], dists, brs=('l',), final_aux=final_aux and [final_aux[1]])[0])
                g = self._x_layers(W, kv, tm, [g], [gmap_masks.contiguous()], dists, brs=('g',), final_aux=final_aux and [final_aux[0]])[0]
                v = lj()
        else:
            g, v = self._x_layers(W, kv, tm, [g, v], [gmap_masks.contiguous(), vp_masks.contiguous()], dists, final_aux=final_aux)

        items = [('head.g', g), ('head.l', v)]
        if self.sap_fuse_linear is not None:
            items.append(('head.f', cls2 if cls2 is not None else torch.cat([g[:, 0], v[:, 0]], 1)))
        if vp_obj_masks is not None:
            items.append(('head.o', v))
        rmf = self._fork(lambda: self._rmf(W, g, v, txt, mem_out))          # RMF poolers beside the SAP heads
        sc = self._heads(W, items)
        fuse_raw = sc[2] if self.sap_fuse_linear is not None else None
        if gmap_src is None:
            src_np, bw_np = self.fusion_indices(gmap_vpids, gmap_visited_masks, vp_cand_vpids, G, Vp, 2)
            gmap_src, bw_mask = torch.from_numpy(src_np).to(dev), torch.from_numpy(bw_np).to(dev)
        res = Fn.nav_logits(sc[0], sc[1], fuse_raw, gmap_masks.contiguous(), gmap_visited_masks.contiguous(),
                            vp_nav_masks.contiguous(), gmap_src, bw_mask, sc[-1] if vp_obj_masks is not None else None,
                            vp_obj_masks.contiguous() if vp_obj_masks is not None else None)
        gl, ll, fl = res[:3]
        obj_logits = res[3] if vp_obj_masks is not None else None

        cls = rmf()
        return {'gmap_embeds': g, 'vp_embeds': v, 'global_logits': gl, 'local_logits': ll, 'fused_logits': fl,
                'obj_logits': obj_logits, 'txt_embeds': txt_embeds, 'cls_embeds': cls}

    # ---------------------------------------------------------------- dispatch (vilmodel.py:838-858)
    def forward(self, mode, batch, **kwargs):
        if mode == 'language':
            return self.forward_text(batch['txt_ids'], batch['txt_masks'], batch['act_txt_ids'], batch['act_txt_masks'],
                                     batch['obj_txt_ids'], batch['obj_txt_masks'])
        elif mode == 'panorama':
            get = batch.get if hasattr(batch, 'get') else (lambda k, d=None: batch[k])
            return self.forward_panorama_per_step(batch['view_img_fts'], batch['obj_img_fts'], batch['loc_fts'],
                                                  batch['nav_types'], batch['view_lens'], batch['obj_lens'],
                                                  pano_out=get('pano_out'), gmap_scatter=get('gmap_scatter'),
                                                  want_masks=get('want_pano_masks', True) is not False)
        elif mode == 'navigation':
            get = batch.get if hasattr(batch, 'get') else (lambda k, d=None: batch[k])
            return self.forward_navigation_per_step(
                batch['txt_embeds'], batch['txt_masks'], batch['gmap_img_embeds'], batch['gmap_step_ids'],
                batch['gmap_pos_fts'], batch['gmap_masks'], batch['gmap_pair_dists'], batch['gmap_visited_masks'],
                batch['gmap_vpids'], batch['vp_img_embeds'], batch['vp_pos_fts'], batch['vp_masks'],
                batch['vp_nav_masks'], batch['vp_obj_masks'], batch['vp_cand_vpids'],
                gmap_src=get('gmap_src'), bw_mask=get('bw_mask'), txt_kv=get('txt_kv'), mem_out=get('mem_out'))
