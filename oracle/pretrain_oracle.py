"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/dsrg_oracle.py header).

CPU torch-fp32 restatement of the reference's pre-training model: GlocalTextPathCMT
(pretrain_src/model/vilmodel.py:716-840) + GlocalTextPathCMTPreTraining heads and losses
(pretrain_src/model/pretrain_cmt.py:38-392), and of the REVERIE twin (vilmodel_reverie.py / pretrain_cmt_reverie.py: objv1
object tokens, MRC object branch, OG).  Pinned by tests/golden/r2r_pt.npz and reverie_pt.npz, produced by the
reference's own files (tools/make_golden_pt.py).  PT = pretrain_src/model.  `valid_sap_og` has no golden: the
reference's own call omits the objv2 features and raises (pretrain_cmt.py:337-341); here it is SAP + OG logits of one
backbone pass, each of which is pinned by its own task's golden."""
import torch
import torch.nn.functional as F

from .dsrg_oracle import (lin, ln, gelu, ext_neg, seq_masks, bert_layer, x_layer, x_layer_lang2visn, cls_head,
                          pano_encoder, objv2_embed, fuse_logits)


class _Sub(dict):
    """view of a state dict under a prefix"""
    def __init__(self, sd, prefix):
        super().__init__({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})


def text(sd, cfg, b):  # PT/vilmodel.py:738-748, 79-100 (IGL branch unreachable: collators pass act_txt_ids=None)
    ids = b['txt_ids']
    L = ids.shape[1]
    x = F.embedding(ids, sd['embeddings.word_embeddings.weight'], padding_idx=0) + sd['embeddings.position_embeddings.weight'][:L][None] \
        + sd['embeddings.token_type_embeddings.weight'][0][None, None]
    x = ln(sd, 'embeddings.LayerNorm', x, cfg.layer_norm_eps)
    tm = seq_masks(b['txt_lens'], L)
    for i in range(cfg.num_l_layers):
        x = bert_layer(sd, 'lang_encoder.layer.%d' % i, x, ext_neg(tm), cfg.num_attention_heads)
    return x, tm


def img_embeddings(sd, cfg, b):  # PT/vilmodel.py:497-546
    fts, lens = b['traj_objv2_fts'], b['traj_objv2_lens']
    img = ln(sd, 'img_embeddings.img_layer_norm', lin(sd, 'img_embeddings.img_linear', b['traj_view_img_fts']))
    obj = objv2_embed(sd, fts, lens, b['traj_nav_types'], cfg)
    x = img + ln(sd, 'img_embeddings.loc_layer_norm', lin(sd, 'img_embeddings.loc_linear', b['traj_loc_fts'])) \
        + F.embedding(b['traj_nav_types'], sd['img_embeddings.nav_type_embedding.weight']) \
        + sd['embeddings.token_type_embeddings.weight'][1][None, None]
    x = ln(sd, 'img_embeddings.layer_norm', x)
    masks = seq_masks(b['traj_vp_view_lens'], x.shape[1])
    x = pano_encoder(sd, 'img_embeddings.pano_encoder', x, masks, cfg)
    x = ln(sd, 'img_embeddings.vis_obj_linear.1', lin(sd, 'img_embeddings.vis_obj_linear.0', torch.cat([x, obj], -1)))
    w = torch.softmax(torch.tanh(lin(sd, 'img_embeddings.adaptive_pano_attn', x)), 1)
    fused = (x * w).sum(1)
    sl = b['traj_step_lens']
    return torch.split(x, sl, 0), torch.split(b['traj_vp_view_lens'], sl, 0), torch.split(fused, sl, 0)


def img_embeddings_reverie(sd, cfg, b):  # PT/vilmodel_reverie.py:520-593
    V = b['traj_view_img_fts'].shape[1]
    img = ln(sd, 'img_embeddings.img_layer_norm', lin(sd, 'img_embeddings.img_linear', b['traj_view_img_fts']))
    obj = objv2_embed(sd, b['traj_objv2_fts'], b['traj_objv2_lens'], b['traj_nav_types'][:, :V], cfg)
    view = ln(sd, 'img_embeddings.vis_obj_linear.1', lin(sd, 'img_embeddings.vis_obj_linear.0', torch.cat([img, obj], -1)))
    lens = b['traj_vp_view_lens']
    if b.get('traj_obj_img_fts') is not None:  # :551-564
        o1 = ln(sd, 'img_embeddings.obj_reverie_layer_norm', lin(sd, 'img_embeddings.obj_reverie_linear', b['traj_obj_img_fts']))
        lens = b['traj_vp_view_lens'] + b['traj_vp_obj_lens']
        x = torch.zeros(view.shape[0], int(lens.max()), view.shape[2])
        for i in range(view.shape[0]):
            vl, ol = int(b['traj_vp_view_lens'][i]), int(b['traj_vp_obj_lens'][i])
            x[i, :vl] = view[i, :vl]
            x[i, vl:vl + ol] = o1[i, :ol]
    else:
        x = view
    x = x + ln(sd, 'img_embeddings.loc_layer_norm', lin(sd, 'img_embeddings.loc_linear', b['traj_loc_fts'])) \
        + F.embedding(b['traj_nav_types'], sd['img_embeddings.nav_type_embedding.weight']) \
        + sd['embeddings.token_type_embeddings.weight'][1][None, None]
    x = ln(sd, 'img_embeddings.layer_norm', x)
    x = pano_encoder(sd, 'img_embeddings.pano_encoder', x, seq_masks(lens, x.shape[1]), cfg)
    w = torch.softmax(torch.tanh(lin(sd, 'img_embeddings.adaptive_pano_attn', x)), 1)
    fused = (x * w).sum(1)
    sl = b['traj_step_lens']
    return torch.split(x, sl, 0), torch.split(lens, sl, 0), torch.split(fused, sl, 0)


def _img(sd, cfg, b):
    return img_embeddings_reverie(sd, cfg, b) if 'img_embeddings.obj_reverie_linear.weight' in sd else img_embeddings(sd, cfg, b)


def aggregate_gmap(split_embeds, split_lens, split_fused, b):  # PT/vilmodel.py:597-635
    B = len(split_embeds)
    rows = []
    for i in range(B):
        vis, unv = {}, {}
        m = seq_masks(split_lens[i])
        e = split_embeds[i][:, :int(split_lens[i].max())] * m.unsqueeze(2)
        for t in range(len(split_embeds[i])):
            vis[b['traj_vpids'][i][t]] = split_fused[i][t]
            for j, vp in enumerate(b['traj_cand_vpids'][i][t]):
                if vp not in vis:
                    unv.setdefault(vp, []).append(e[t][j])
        rows.append(torch.stack([vis[vp] if vp in vis else torch.stack(unv[vp], 0).mean(0) for vp in b['gmap_vpids'][i][1:]], 0))
    G = max(r.shape[0] for r in rows) + 1
    out = torch.zeros(B, G, rows[0].shape[1])
    for i, r in enumerate(rows):
        out[i, 1:1 + r.shape[0]] = r
    return out


def gmap_input(sd, cfg, se, sl, sf, b):  # PT/vilmodel.py:637-650
    g = aggregate_gmap(se, sl, sf, b) + F.embedding(b['gmap_step_ids'], sd['global_encoder.gmap_step_embeddings.weight']) \
        + ln(sd, 'global_encoder.gmap_pos_embeddings.1', lin(sd, 'global_encoder.gmap_pos_embeddings.0', b['gmap_pos_fts']))
    return g, seq_masks(b['gmap_lens'], g.shape[1])


def vp_input(sd, cfg, se, sl, b):  # PT/vilmodel.py:557-571
    B = len(se)
    lens = torch.stack([x[-1] + 1 for x in sl], 0)
    T = int(lens.max())
    v = torch.zeros(B, T, se[0].shape[-1])
    for i in range(B):
        n = min(T - 1, se[i].shape[1])
        v[i, 1:1 + n] = se[i][-1][:n]
    v = v + ln(sd, 'local_encoder.vp_pos_embeddings.1', lin(sd, 'local_encoder.vp_pos_embeddings.0', b['vp_pos_fts']))
    return v, seq_masks(lens, T)


def bert_forward(sd, cfg, b, return_gmap=True):  # PT/vilmodel.py:730-779
    txt, tm = text(sd, cfg, b)
    se, sl, sf = _img(sd, cfg, b)
    ta = ext_neg(tm)
    g = None
    if return_gmap:
        g, gm = gmap_input(sd, cfg, se, sl, sf, b)
        sprels = (b['gmap_pair_dists'] * sd['global_encoder.sprel_linear.weight'].reshape(())
                  + sd['global_encoder.sprel_linear.bias'].reshape(()))[:, None]
        for i in range(cfg.num_x_layers):
            g = x_layer(sd, 'global_encoder.encoder.x_layers.%d' % i, txt, ta, g, ext_neg(gm), sprels, cfg.num_attention_heads)
    v, vm = vp_input(sd, cfg, se, sl, b)
    for i in range(cfg.num_x_layers):
        v = x_layer(sd, 'local_encoder.encoder.x_layers.%d' % i, txt, ta, v, ext_neg(vm), None, cfg.num_attention_heads)
    return g, v


def bert_forward_mlm(sd, cfg, b):  # PT/vilmodel.py:782-840
    txt, tm = text(sd, cfg, b)
    se, sl, sf = _img(sd, cfg, b)
    ta = ext_neg(tm)
    g, gm = gmap_input(sd, cfg, se, sl, sf, b)
    v, vm = vp_input(sd, cfg, se, sl, b)
    tg, tv = txt, txt
    for i in range(cfg.num_x_layers):
        tg = x_layer_lang2visn(sd, 'global_encoder.encoder.x_layers.%d' % i, tg, ta, g, ext_neg(gm), cfg.num_attention_heads)
        tv = x_layer_lang2visn(sd, 'local_encoder.encoder.x_layers.%d' % i, tv, ta, v, ext_neg(vm), cfg.num_attention_heads)
    return tg + tv


def region_head(sd, p, x):  # RegionClassification, PT/pretrain_cmt.py:12-23
    return lin(sd, p + '.net.3', ln(sd, p + '.net.2', torch.relu(lin(sd, p + '.net.0', x))))


def forward(sd_full, cfg, batch, task, compute_loss=True):
    """GlocalTextPathCMTPreTraining.forward, PT/pretrain_cmt.py:71-140."""
    sd = _Sub(sd_full, 'bert.')
    b = batch
    if task.startswith('mlm'):  # :142-167
        t = bert_forward_mlm(sd, cfg, b)
        h = t[b['txt_labels'] != -1]
        p = 'mlm_head.predictions'
        h = ln(sd_full, p + '.transform.LayerNorm', gelu(lin(sd_full, p + '.transform.dense', h)), cfg.layer_norm_eps)
        scores = F.linear(h, sd_full[p + '.decoder.weight']) + sd_full[p + '.bias']
        return F.cross_entropy(scores, b['txt_labels'][b['txt_labels'] != -1], reduction='none') if compute_loss else scores
    if task.startswith('mrc'):  # :175-228 (R2R: traj_obj_img_fts is None -> views only)
        _, v = bert_forward(sd, cfg, b, return_gmap=False)
        last = [x[-1] for x in torch.split(b['traj_vp_view_lens'], b['traj_step_lens'])]
        ve = _pad_slices([v[i, 1:int(n) + 1] for i, n in enumerate(last)])
        m = b['vp_view_mrc_masks']
        logits = region_head(sd_full, 'image_classifier', ve[m])
        tgt = b['vp_view_probs'][m]
        ologits = otgt = None
        if b.get('traj_obj_img_fts') is not None:  # :202-213
            lo = [x[-1] for x in torch.split(b['traj_vp_obj_lens'], b['traj_step_lens'])]
            oe = _pad_slices([v[i, int(n) + 1:int(n) + int(k) + 1] for i, (n, k) in enumerate(zip(last, lo))])
            om = b['vp_obj_mrc_masks']
            head = 'obj_classifier' if 'obj_classifier.net.0.weight' in sd_full else 'image_classifier'
            ologits = region_head(sd_full, head, oe[om])
            otgt = b['vp_obj_probs'][om]
        if not compute_loss:
            return logits, tgt, ologits, otgt
        loss = F.kl_div(F.log_softmax(logits, -1), tgt, reduction='none').sum(1)
        if ologits is not None:
            loss = torch.cat([loss, F.kl_div(F.log_softmax(ologits, -1), otgt, reduction='none').sum(1)], 0)
        return loss
    if task.startswith('og'):  # :296-327
        _, v = bert_forward(sd, cfg, b, return_gmap=False)
        ol = og_logits(sd_full, b, v)
        return F.cross_entropy(ol, b['obj_labels'], reduction='none') if compute_loss else ol
    if task.startswith('valid_sap_og'):  # :329-392 (see the header)
        g, v = bert_forward(sd, cfg, b)
        return sap_logits(sd_full, b, g, v) + (og_logits(sd_full, b, v),)
    if task.startswith('sap'):  # :230-294
        g, v = bert_forward(sd, cfg, b)
        gl, ll, fused = sap_logits(sd_full, b, g, v)
        if not compute_loss:
            return gl, ll, fused, b['global_act_labels'], b['local_act_labels']
        return F.cross_entropy(gl, b['global_act_labels'], reduction='none') \
            + F.cross_entropy(ll, b['local_act_labels'], reduction='none') \
            + F.cross_entropy(fused, b['global_act_labels'], reduction='none')
    raise ValueError('invalid task')


def _pad_slices(rows):  # PT/ops.py:45-67 pad_tensors_wgrad
    T = max(r.shape[0] for r in rows)
    out = torch.zeros(len(rows), T, rows[0].shape[-1])
    for i, r in enumerate(rows):
        out[i, :r.shape[0]] = r
    return out


def og_logits(sd_full, b, v):  # PT/pretrain_cmt.py:313-321
    lv = [x[-1] for x in torch.split(b['traj_vp_view_lens'], b['traj_step_lens'], 0)]
    lo = [x[-1] for x in torch.split(b['traj_vp_obj_lens'], b['traj_step_lens'], 0)]
    oe = _pad_slices([v[i, 1 + int(n):1 + int(n) + int(k)] for i, (n, k) in enumerate(zip(lv, lo))])
    om = seq_masks(torch.stack(lo, 0))
    return cls_head(sd_full, 'og_head', oe).squeeze(2).masked_fill(~om, float('-inf'))


def sap_logits(sd_full, b, g, v):  # PT/pretrain_cmt.py:248-285
    if True:
        fw = torch.sigmoid(cls_head(sd_full, 'sap_fuse_linear', torch.cat([g[:, 0], v[:, 0]], 1)))
        gl = cls_head(sd_full, 'global_sap_head', g).squeeze(2) * fw
        gl = gl.masked_fill(b['gmap_visited_masks'], float('-inf')).masked_fill(~seq_masks(b['gmap_lens'], g.shape[1]), float('-inf'))
        ll = cls_head(sd_full, 'local_sap_head', v).squeeze(2) * (1 - fw)
        nav = [x[-1] != 1 for x in torch.split(b['traj_nav_types'], b['traj_step_lens'])]
        nm = torch.zeros(ll.shape, dtype=torch.bool)
        for i, x in enumerate(nav):
            n = min(ll.shape[1] - 1, x.shape[0])
            nm[i, 1:1 + n] = x[:n]
        ll = ll.masked_fill(nm, float('-inf'))
        fused = fuse_logits(gl, ll, b['gmap_vpids'], b['gmap_visited_masks'],
                            [[None] + list(c[-1]) for c in b['traj_cand_vpids']], 1)
        return gl, ll, fused
