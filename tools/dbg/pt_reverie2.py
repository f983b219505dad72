import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
from dsrg_amd import synth, functional as Fn, kernels as K
from dsrg_amd.config import DSRGConfig
from dsrg_amd.pretrain_reverie import GlocalTextPathCMTPreTraining
from tests import common
from tests.test_oracle_pt_golden import PT_CFG_REVERIE
from oracle import pretrain_oracle as PO
from oracle.dsrg_oracle import lin, ln, objv2_embed, seq_masks, pano_encoder

gold = common.load_gold('reverie_pt')
B, L, O, seed = [int(x) for x in gold['meta']]
cfg = DSRGConfig(**PT_CFG_REVERIE)
m = GlocalTextPathCMTPreTraining(cfg, compute_dtype=torch.float32)
m.load_state_dict(common.synth_sd('reverie_pt'))
m = m.cuda().train()
sdf = common.synth_sd('reverie_pt')
sd = PO._Sub(sdf, 'bert.')
d = lambda a, b: float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())
task = 'mlm'
b = common.to_t(synth.pretrain_batch(B, L, O, seed, task, reverie=True))
bg = common.to_t(synth.pretrain_batch(B, L, O, seed, task, reverie=True), 'cuda')
for k, v in b.items():
    if hasattr(v, 'shape'): print(k, tuple(v.shape), v.dtype)
print('view lens', b['traj_vp_view_lens'].tolist(), 'obj lens', b['traj_vp_obj_lens'].tolist(), 'steps', b['traj_step_lens'])
with torch.no_grad():
    V = b['traj_view_img_fts'].shape[1]
    img = ln(sd, 'img_embeddings.img_layer_norm', lin(sd, 'img_embeddings.img_linear', b['traj_view_img_fts']))
    obj = objv2_embed(sd, b['traj_objv2_fts'], b['traj_objv2_lens'], b['traj_nav_types'][:, :V], cfg)
    view = ln(sd, 'img_embeddings.vis_obj_linear.1', lin(sd, 'img_embeddings.vis_obj_linear.0', torch.cat([img, obj], -1)))
    o1 = ln(sd, 'img_embeddings.obj_reverie_layer_norm', lin(sd, 'img_embeddings.obj_reverie_linear', b['traj_obj_img_fts']))
    lens = b['traj_vp_view_lens'] + b['traj_vp_obj_lens']
    x = torch.zeros(view.shape[0], int(lens.max()), view.shape[2])
    for i in range(view.shape[0]):
        vl, ol = int(b['traj_vp_view_lens'][i]), int(b['traj_vp_obj_lens'][i])
        x[i, :vl] = view[i, :vl]
        x[i, vl:vl + ol] = o1[i, :ol]
    x0 = x
    x = x + ln(sd, 'img_embeddings.loc_layer_norm', lin(sd, 'img_embeddings.loc_linear', b['traj_loc_fts'])) \
        + F.embedding(b['traj_nav_types'], sd['img_embeddings.nav_type_embedding.weight']) \
        + sd['embeddings.token_type_embeddings.weight'][1][None, None]
    x1 = ln(sd, 'img_embeddings.layer_norm', x)
    x2 = pano_encoder(sd, 'img_embeddings.pano_encoder', x1, seq_masks(lens, x.shape[1]), cfg)

    # product pieces
    self = m.bert
    W, cd = self._weights(), self.compute_dtype
    view_img_fts, objv2_img_fts, loc_fts, nav_types, view_lens, objv2_lens = bg['traj_view_img_fts'], bg['traj_objv2_fts'], bg['traj_loc_fts'], bg['traj_nav_types'], bg['traj_vp_view_lens'], bg['traj_objv2_lens']
    obj_img_fts, obj_lens, objv2_nav_types = bg['traj_obj_img_fts'], bg['traj_vp_obj_lens'], bg['traj_nav_types'][:, :V]
    Bs = view_img_fts.shape[0]
    H, dd = cfg.hidden_size, cfg.obj_v2_size
    dev = view_img_fts.device
    obj_rows, OO = self._objv2_tokens(W, objv2_img_fts, objv2_lens, objv2_nav_types)
    print('objv2 rows', d(Fn.group_mean(obj_rows, OO).view(Bs, V, dd), obj))
    z = Fn.linear(view_img_fts.to(cd), W['img.lin.w'], W['img.lin.b'], out_f32=True)
    o1g = Fn.linear_ln(obj_img_fts.to(cd), W['objv1.lin.w'], W['objv1.lin.b'], W['objv1.ln.g'], W['objv1.ln.b'], 1e-12)
    print('o1', d(o1g, o1))
    imgg = Fn.layernorm(z, W['img.ln.g'], W['img.ln.b'], 1e-12, out_dtype=cd)
    print('img', d(imgg.view(Bs, V, H), img))
    cat = torch.cat([imgg.view(Bs, V, H), Fn.group_mean(obj_rows, OO).view(Bs, V, dd)], -1)
    viewg = Fn.linear_ln(cat, W['visobj.lin.w'], W['visobj.lin.b'], W['visobj.ln.g'], W['visobj.ln.b'], 1e-12)
    print('view', d(viewg, view))
    No = obj_img_fts.shape[1]
    T = nav_types.shape[1]
    lensg = view_lens + obj_lens
    pos = torch.arange(T, device=dev)[None]
    vl, tl = view_lens[:, None], lensg[:, None]
    idx = torch.where(pos < vl, pos, torch.where(pos < tl, V + pos - vl, torch.full_like(pos, V + No)))
    src = torch.cat([viewg, o1g, torch.zeros(Bs, 1, H, device=dev, dtype=cd)], 1)
    xg = torch.gather(src, 1, idx[..., None].expand(Bs, T, H))
    print('x0', d(xg, x0), xg.shape, x0.shape)
    x1g = self._pano_tokens(W, xg, loc_fts, nav_types)
    print('x1', d(x1g, x1), 'per pos', (x1g.cpu() - x1).abs().amax(-1).amax(0).tolist()[:8])
    pm = (torch.arange(T, device=dev)[None] < lensg[:, None]).contiguous()
    x2g = self._pano_encoder(W, x1g, pm)
    x2g = Fn.layernorm(x2g, W['pano.norm.g'], W['pano.norm.b'], 1e-12)
    e = (x2g.cpu() - x2).abs().amax(-1)
    print('x2', float(e.max()), 'per step', e.amax(1).tolist())
    print('x2 per pos step0', [round(v, 3) for v in e[0].tolist()])
