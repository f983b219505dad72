"""debug aid: REVERIE pre-training twin on the GPU vs the CPU oracle, stage by stage"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dsrg_amd import synth
from dsrg_amd.config import DSRGConfig
from dsrg_amd.pretrain_reverie import GlocalTextPathCMTPreTraining
from tests import common
from tests.test_oracle_pt_golden import PT_CFG_REVERIE
from oracle import pretrain_oracle as PO

gold = common.load_gold('reverie_pt')
B, L, O, seed = [int(x) for x in gold['meta']]
cfg = DSRGConfig(**PT_CFG_REVERIE)
m = GlocalTextPathCMTPreTraining(cfg, compute_dtype=torch.float32)
m.load_state_dict(common.synth_sd('reverie_pt'))
m = m.cuda().train()
sd = common.synth_sd('reverie_pt')
sd['bert.embeddings.word_embeddings.weight'] = sd['mlm_head.predictions.decoder.weight']
sub = PO._Sub(sd, 'bert.')
d = lambda a, b: float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())
for task in ('mlm', 'mrc', 'sap', 'og'):
    bc = common.to_t(synth.pretrain_batch(B, L, O, seed, task, reverie=True))
    bg = common.to_t(synth.pretrain_batch(B, L, O, seed, task, reverie=True), 'cuda')
    with torch.no_grad():
        lo = PO.forward(sd, cfg, bc, task, True)
        lg = m(bg, task, compute_loss=True)
    print(task, 'loss err vs oracle', d(lg, lo), 'oracle vs gold', float(np.abs(lo.numpy() - gold[task + '_loss']).max()), flush=True)
    if task == 'mlm':
        with torch.no_grad():
            bert = m.bert
            W = bert._weights()
            txt_o, tm_o = PO.text(sub, cfg, bc)
            txt_g, tm_g = bert._text(bg['txt_ids'], bg['txt_lens'])
            print(' text', d(txt_g, txt_o))
            se, sl, sf = PO._img(sub, cfg, bc)
            pano, fused, tok = bert._trajectory(bg)
            po = torch.cat(se, 0)
            print(' pano', d(pano, po), 'fused', d(fused, torch.cat(sf, 0)), 'lens', (tok.cpu() == torch.cat(sl, 0)).all().item())
            g_o, gm_o = PO.gmap_input(sub, cfg, se, sl, sf, bc)
            g_g, gm_g = bert._gmap_input(W, pano, fused, bg)
            print(' gmap in', d(g_g, g_o), (gm_g.cpu() == gm_o).all().item(), g_g.shape, g_o.shape)
            v_o, vm_o = PO.vp_input(sub, cfg, se, sl, bc)
            v_g, vm_g, last = bert._vp_input(W, pano, bg, tok)
            print(' vp in', d(v_g, v_o), (vm_g.cpu() == vm_o).all().item(), v_g.shape, v_o.shape)
            t_o = PO.bert_forward_mlm(sub, cfg, bc)
            t_g = bert.forward_mlm(bg)
            print(' mlm txt', d(t_g, t_o))
            sel = bg['txt_labels'] != -1
            print(' n masked', int(sel.sum()))
