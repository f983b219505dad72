"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not product code.

A CPU (torch fp32, functional, weight-dict driven) restatement of the reference's
navigation-policy forward pass, written from the reference's source as a numerical
specification.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; nothing under dsrg_amd/ does.  It shares no code with the product path.

Pinning: checked against golden vectors produced by running the reference's own model
files in the build container (tools/make_golden.py -> tests/golden/*.npz; the reference
ships no tests or fixtures of its own).

Reference citations are `path:line` under the reference root, FT = finetune_src/models.
"""
import math
import torch
import torch.nn.functional as F

NEG = -10000.0


# ---------------------------------------------------------------- primitives
def lin(sd, p, x):
    return F.linear(x, sd[p + '.weight'], sd.get(p + '.bias'))


def ln(sd, p, x, eps=1e-12):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], eps)


def gelu(x):  # FT/vilmodel.py:30-31 (erf form)
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def ext_neg(mask):  # FT/ops.py:24-33
    return (1.0 - mask[:, None, None, :].float()) * NEG


def seq_masks(lens, max_len=None):  # FT/ops.py:35-43
    max_len = int(lens.max()) if max_len is None else max_len
    return torch.arange(max_len)[None] < lens[:, None]


def mha(sd, pq, pk, pv, x, ctx, add_mask, nh):
    """BertSelfAttention / BertOutAttention core, FT/vilmodel.py:125-164, 349-376."""
    B, Nq, _ = x.shape
    q, k, v = lin(sd, pq, x), lin(sd, pk, ctx), lin(sd, pv, ctx)
    dh = q.shape[-1] // nh
    sp = lambda t: t.view(B, -1, nh, dh).permute(0, 2, 1, 3)
    s = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dh)
    if add_mask is not None:
        s = s + add_mask
    p = torch.softmax(s, -1)
    return (p @ sp(v)).permute(0, 2, 1, 3).reshape(B, Nq, nh * dh)


def self_output(sd, p, h, inp, eps=1e-12):  # BertSelfOutput / BertOutput, FT/vilmodel.py:173-177, 213-217
    return ln(sd, p + '.LayerNorm', lin(sd, p + '.dense', h) + inp, eps)


def bert_layer(sd, p, x, add_mask, nh):  # FT/vilmodel.py:226-232
    a = mha(sd, p + '.attention.self.query', p + '.attention.self.key', p + '.attention.self.value', x, x, add_mask, nh)
    a = self_output(sd, p + '.attention.output', a, x)
    i = gelu(lin(sd, p + '.intermediate.dense', a))
    return self_output(sd, p + '.output', i, a)


def x_layer(sd, p, txt, txt_add, vis, vis_add, sprels, nh):  # GraphLXRTXLayer.forward, FT/vilmodel.py:454-469
    a = mha(sd, p + '.visual_attention.att.query', p + '.visual_attention.att.key',
            p + '.visual_attention.att.value', vis, txt, txt_add, nh)
    a = self_output(sd, p + '.visual_attention.output', a, vis)
    m = vis_add if sprels is None else vis_add + sprels
    s = mha(sd, p + '.visn_self_att.self.query', p + '.visn_self_att.self.key',
            p + '.visn_self_att.self.value', a, a, m, nh)
    s = self_output(sd, p + '.visn_self_att.output', s, a)
    i = gelu(lin(sd, p + '.visn_inter.dense', s))
    return self_output(sd, p + '.visn_output', i, s)


def x_layer_lang2visn(sd, p, txt, txt_add, vis, vis_add, nh):  # PT/vilmodel.py:423-434
    a = mha(sd, p + '.visual_attention.att.query', p + '.visual_attention.att.key',
            p + '.visual_attention.att.value', txt, vis, vis_add, nh)
    a = self_output(sd, p + '.visual_attention.output', a, txt)
    s = mha(sd, p + '.lang_self_att.self.query', p + '.lang_self_att.self.key',
            p + '.lang_self_att.self.value', a, a, txt_add, nh)
    s = self_output(sd, p + '.lang_self_att.output', s, a)
    i = gelu(lin(sd, p + '.lang_inter.dense', s))
    return self_output(sd, p + '.lang_output', i, s)


def crossmodal(sd, p, txt, txt_masks, vis, vis_masks, sprels, cfg):  # FT/vilmodel.py:540-549
    ta, va = ext_neg(txt_masks), ext_neg(vis_masks)
    for i in range(cfg.num_x_layers):
        vis = x_layer(sd, '%s.x_layers.%d' % (p, i), txt, ta, vis, va, sprels, cfg.num_attention_heads)
    return vis


def cls_head(sd, p, x):  # ClsPrediction, FT/vilmodel.py:616-628
    h = torch.relu(lin(sd, p + '.net.0', x))
    return lin(sd, p + '.net.3', ln(sd, p + '.net.2', h))


def pano_layer(sd, p, x, key_pad, nh):
    """DETR-style pre-norm layer over nn.MultiheadAttention, FT/transformer.py:170-182."""
    B, N, H = x.shape
    y = F.layer_norm(x, (H,), sd[p + '.norm1.weight'], sd[p + '.norm1.bias'], 1e-5)
    qkv = F.linear(y, sd[p + '.self_attn.in_proj_weight'], sd[p + '.self_attn.in_proj_bias'])
    q, k, v = qkv.split(H, -1)
    dh = H // nh
    sp = lambda t: t.view(B, N, nh, dh).permute(0, 2, 1, 3)
    s = (sp(q) * (dh ** -0.5)) @ sp(k).transpose(-1, -2)
    s = s.masked_fill(key_pad[:, None, None, :], float('-inf'))
    a = (torch.softmax(s, -1) @ sp(v)).permute(0, 2, 1, 3).reshape(B, N, H)
    x = x + lin(sd, p + '.self_attn.out_proj', a)
    y = F.layer_norm(x, (H,), sd[p + '.norm2.weight'], sd[p + '.norm2.bias'], 1e-5)
    return x + lin(sd, p + '.linear2', F.gelu(lin(sd, p + '.linear1', y)))


def pano_encoder(sd, p, x, masks, cfg):  # FT/transformer.py:71-89, FT/ops.py:10-22
    for i in range(cfg.num_pano_layers):
        x = pano_layer(sd, '%s.layers.%d' % (p, i), x, masks.logical_not(), cfg.num_attention_heads)
    return ln(sd, p + '.norm', x)


def gaa(sd, x):  # FT/vilmodel.py:735-740 (softmax over ALL view slots, unmasked)
    w = torch.softmax(torch.tanh(lin(sd, 'img_embeddings.adaptive_pano_attn', x)), 1)
    return (x * w).sum(1)


def objv2_embed(sd, fts, lens, nav_types, cfg):
    """ASV object tokens -> per-view 128-d embedding, FT/vilmodel.py:697-714.
    NB the 'mask' is a bool added to the scores (+1 valid / +0 padded), line 711-712 -> 143."""
    B, V, O, _ = fts.shape
    e = lin(sd, 'img_embeddings.obj_bbox_linear', fts[..., :-1]) \
        + F.embedding(fts[..., -1].int().long(), sd['img_embeddings.obj_name_linear.weight']) \
        + F.embedding(nav_types, sd['img_embeddings.obj_nav_type_embedding.weight']).unsqueeze(-2)
    e = ln(sd, 'img_embeddings.obj_layer_norm', e).reshape(B, V * O, -1)
    m = seq_masks(lens.reshape(-1), O).reshape(B, 1, 1, V * O).float()
    for i in range(2):
        e = bert_layer(sd, 'img_embeddings.obj_self_encoder.layer.%d' % i, e, m, cfg.obj_v2_size // 64)
    return e.reshape(B, V, O, -1).mean(2)


# ---------------------------------------------------------------- the three modes
def forward_text(sd, cfg, b, reverie=False):
    """FT/vilmodel.py:678-688, 76-96, 509-530 (REVERIE IGL: vilmodel_reverie.py:523-529)."""
    ids = b['txt_ids']
    L = ids.shape[1]
    x = F.embedding(ids, sd['embeddings.word_embeddings.weight'], padding_idx=0) \
        + sd['embeddings.position_embeddings.weight'][:L][None] \
        + sd['embeddings.token_type_embeddings.weight'][0][None, None]
    x = ln(sd, 'embeddings.LayerNorm', x, cfg.layer_norm_eps)
    ta = ext_neg(b['txt_masks'])
    for i in range(cfg.num_l_layers):
        x = bert_layer(sd, 'lang_encoder.layer.%d' % i, x, ta, cfg.num_attention_heads)
    if b.get('obj_txt_ids') is None:
        return x
    act = F.embedding(b['act_txt_ids'], sd['embeddings.act_embedding.weight'])
    obj = F.embedding(b['obj_txt_ids'], sd['embeddings.obj_embedding.weight'])
    if reverie:
        ctx, cm = obj, b['obj_txt_masks']
    else:
        ctx, cm = torch.cat([act, obj], 1), torch.cat([b['act_txt_masks'], b['obj_txt_masks']], 1)
    p = 'lang_encoder.instr_actobj_encoder'
    aug = mha(sd, p + '.query', p + '.key', p + '.value', x, ctx, ext_neg(cm), cfg.num_attention_heads)
    w = torch.sigmoid(lin(sd, 'lang_encoder.instr_aug_linear', aug) + lin(sd, 'lang_encoder.instr_ori_linear', x))
    return w * aug + (1 - w) * x


def forward_panorama(sd, cfg, b, reverie=False):
    """R2R FT/vilmodel.py:690-742; REVERIE vilmodel_reverie.py:689-759."""
    img = ln(sd, 'img_embeddings.img_layer_norm', lin(sd, 'img_embeddings.img_linear', b['view_img_fts']))
    type1 = sd['embeddings.token_type_embeddings.weight'][1][None, None]
    if not reverie:
        obj = objv2_embed(sd, b['obj_img_fts'], b['obj_lens'], b['nav_types'], cfg)
        x = img + ln(sd, 'img_embeddings.loc_layer_norm', lin(sd, 'img_embeddings.loc_linear', b['loc_fts'])) \
            + F.embedding(b['nav_types'], sd['img_embeddings.nav_type_embedding.weight']) + type1
        x = ln(sd, 'img_embeddings.layer_norm', x)
        masks = seq_masks(b['view_lens'])
        x = pano_encoder(sd, 'img_embeddings.pano_encoder', x, masks, cfg)
        x = ln(sd, 'img_embeddings.vis_obj_linear.1', lin(sd, 'img_embeddings.vis_obj_linear.0', torch.cat([x, obj], -1)))
        return x, masks, gaa(sd, x)
    obj = objv2_embed(sd, b['objv2_img_fts'], b['objv2_lens'], b['objv2_nav_types'], cfg)
    view = ln(sd, 'img_embeddings.vis_obj_linear.1', lin(sd, 'img_embeddings.vis_obj_linear.0', torch.cat([img, obj], -1)))
    if b.get('obj_img_fts') is not None:
        o1 = ln(sd, 'img_embeddings.obj_reverie_layer_norm', lin(sd, 'img_embeddings.obj_reverie_linear', b['obj_img_fts']))
        lens = b['view_lens'] + b['obj_lens']
        T = int(lens.max())
        x = torch.zeros(view.shape[0], T, view.shape[2])
        for i in range(view.shape[0]):  # vilmodel_reverie.py:727-735
            vl, ol = int(b['view_lens'][i]), int(b['obj_lens'][i])
            x[i, :vl] = view[i, :vl]
            x[i, vl:vl + ol] = o1[i, :ol]
    else:
        x, lens = view, b['view_lens']
    x = x + ln(sd, 'img_embeddings.loc_layer_norm', lin(sd, 'img_embeddings.loc_linear', b['loc_fts'])) \
        + F.embedding(b['nav_types'], sd['img_embeddings.nav_type_embedding.weight']) + type1
    x = ln(sd, 'img_embeddings.layer_norm', x)
    masks = seq_masks(lens)
    x = pano_encoder(sd, 'img_embeddings.pano_encoder', x, masks, cfg)
    return x, masks, gaa(sd, x)


def fuse_logits(global_logits, local_logits, gmap_vpids, visited_masks, cand_vpids, skip):
    """FT/vilmodel.py:788-805 (skip=2: [stop],[MEM]); PT/pretrain_cmt.py:269-285 (skip=1)."""
    fused = global_logits.clone()
    fused[:, 0] += local_logits[:, 0]
    for i in range(fused.shape[0]):
        vis = set(vp for vp, m in zip(gmap_vpids[i], visited_masks[i]) if m)
        tmp, bw = {}, 0
        for j, c in enumerate(cand_vpids[i]):
            if j >= skip:
                if c in vis:
                    bw = bw + local_logits[i, j]
                else:
                    tmp[c] = local_logits[i, j]
        for j, vp in enumerate(gmap_vpids[i]):
            if j >= skip and vp not in vis:
                fused[i, j] += tmp[vp] if vp in tmp else bw
    return fused


def forward_navigation(sd, cfg, b):
    """FT/vilmodel.py:745-830."""
    txt, tm = b['txt_embeds'], b['txt_masks']
    g = b['gmap_img_embeds'] + F.embedding(b['gmap_step_ids'], sd['global_encoder.gmap_step_embeddings.weight']) \
        + ln(sd, 'global_encoder.gmap_pos_embeddings.1', lin(sd, 'global_encoder.gmap_pos_embeddings.0', b['gmap_pos_fts']))
    sprels = None
    if cfg.graph_sprels:
        sprels = (b['gmap_pair_dists'] * sd['global_encoder.sprel_linear.weight'].reshape(())
                  + sd['global_encoder.sprel_linear.bias'].reshape(()))[:, None]
    g = crossmodal(sd, 'global_encoder.encoder', txt, tm, g, b['gmap_masks'], sprels, cfg)
    v = b['vp_img_embeds'] + ln(sd, 'local_encoder.vp_pos_embeddings.1',
                                lin(sd, 'local_encoder.vp_pos_embeddings.0', b['vp_pos_fts']))
    v = crossmodal(sd, 'local_encoder.encoder', txt, tm, v, b['vp_masks'], None, cfg)
    fw = torch.sigmoid(cls_head(sd, 'sap_fuse_linear', torch.cat([g[:, 0], v[:, 0]], 1))) if cfg.glocal_fuse else 0.5
    gl = cls_head(sd, 'global_sap_head', g).squeeze(2) * fw
    ll = cls_head(sd, 'local_sap_head', v).squeeze(2) * (1 - fw)
    gl = gl.masked_fill(b['gmap_visited_masks'], float('-inf')).masked_fill(~b['gmap_masks'], float('-inf'))
    ll = ll.masked_fill(~b['vp_nav_masks'], float('-inf'))
    fused = fuse_logits(gl, ll, b['gmap_vpids'], b['gmap_visited_masks'], b['vp_cand_vpids'], 2)
    ol = None
    if b.get('vp_obj_masks') is not None:
        ol = cls_head(sd, 'og_head', v).squeeze(2).masked_fill(~b['vp_obj_masks'], float('-inf'))
    pool = lambda p, x: torch.tanh(lin(sd, p + '.dense', x[:, 0]))  # BertPooler FT/vilmodel.py:274-280
    cls = ln(sd, 'local_his_ln', lin(sd, 'local_his_map', torch.cat(
        [pool('gmap_pooler', g), pool('vp_pooler', v), pool('txt_pooler', txt)], -1)), cfg.layer_norm_eps)
    return {'gmap_embeds': g, 'vp_embeds': v, 'global_logits': gl, 'local_logits': ll,
            'fused_logits': fused, 'obj_logits': ol, 'txt_embeds': txt, 'cls_embeds': cls}


def forward(sd, cfg, mode, batch, reverie=False, grad=False):
    """VLNBert.forward / GlocalTextPathNavCMT.forward dispatch, FT/model.py:21-41, FT/vilmodel.py:838-858.
    grad=True keeps the autograd graph (gradient parity tests differentiate the oracle)."""
    with torch.set_grad_enabled(bool(grad)):
        if mode == 'language':
            return forward_text(sd, cfg, batch, reverie)
        if mode == 'panorama':
            return forward_panorama(sd, cfg, batch, reverie)
        if mode == 'navigation':
            return forward_navigation(sd, cfg, batch)
    raise NotImplementedError('wrong mode: %s' % mode)
