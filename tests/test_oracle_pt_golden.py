"""not-gpu: the pre-training CPU oracle reproduces the reference's own loss vectors / logits
(tests/golden/r2r_pt.npz and reverie_pt.npz from tools/make_golden_pt.py) and, through autograd, its gradient slices."""
import json
import numpy as np
import torch

from dsrg_amd import synth
from dsrg_amd.config import DSRGConfig
from oracle import pretrain_oracle
from tests import common

PT_CFG = dict(use_lang2visn_attn=True, obj_feat_size=0, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
              pretrain_tasks=('mlm', 'mrc', 'sap'))


def test_pt_oracle_matches_reference_golden():
    torch.set_num_threads(8)
    gold = common.load_gold('r2r_pt')
    B, L, O, seed = [int(x) for x in gold['meta']]
    cfg = DSRGConfig(**PT_CFG)
    sd = common.synth_sd('r2r_pt')
    # decoder and word embeddings are ONE tensor (pretrain_cmt.py:66-69); when the golden run loaded the synthetic
    # state dict into the tied parameter the later key (the decoder's) won, so that is the shared value
    sd['bert.embeddings.word_embeddings.weight'] = sd['mlm_head.predictions.decoder.weight']
    for k in sd:
        sd[k].requires_grad_(True)
    for task in ('mlm', 'mrc', 'sap'):
        b = common.to_t(synth.pretrain_batch(B, L, O, seed, task))
        loss = pretrain_oracle.forward(sd, cfg, b, task, True)
        assert np.abs(loss.detach().numpy() - gold[task + '_loss']).max() <= 2e-5, task
        for t in sd.values():
            t.grad = None
        loss.mean().backward()
        for k, g in gold.items():
            if k.startswith(task + '_grad:'):
                mine = sd[k.split(':', 1)[1]].grad.reshape(-1)[:64].numpy()
                assert np.abs(mine - g).max() <= 1e-5 + 1e-4 * np.abs(g).max(), k
        with torch.no_grad():
            o = pretrain_oracle.forward(sd, cfg, b, task, False)
        if task == 'mlm':
            assert np.abs(o[:, ::97].numpy() - gold['mlm_scores_s']).max() <= 1e-4
        elif task == 'mrc':
            assert np.abs(o[0][:, ::13].numpy() - gold['mrc_logits_s']).max() <= 1e-4
        else:
            for n, t in zip(('global_logits', 'local_logits', 'fused_logits'), o[:3]):
                g = gold['sap_' + n]
                fin = np.isfinite(g)
                assert (np.isfinite(t.numpy()) == fin).all()
                assert np.abs(t.numpy()[fin] - g[fin]).max() <= 1e-4


PT_CFG_REVERIE = dict(use_lang2visn_attn=True, obj_feat_size=768, obj_prob_size=1000, hidden_dropout_prob=0.0,
                      attention_probs_dropout_prob=0.0, pretrain_tasks=('mlm', 'mrc', 'sap', 'og'))


def check_reverie_outputs(task, o, gold, tol):
    """compute_loss=False outputs of one task against the REVERIE golden (shared with the gpu test)."""
    f = lambda t: t.detach().float().cpu().numpy()
    if task == 'mlm':
        assert np.abs(f(o)[:, ::97] - gold['mlm_scores_s']).max() <= tol
    elif task == 'mrc':
        assert np.abs(f(o[0])[:, ::13] - gold['mrc_logits_s']).max() <= tol
        assert np.abs(f(o[2])[:, ::13] - gold['mrc_obj_logits_s']).max() <= tol
    else:
        pairs = zip(('sap_global_logits', 'sap_local_logits', 'sap_fused_logits'), o[:3]) if task == 'sap' else [('og_logits', o)]
        for n, t in pairs:
            g, t = gold[n], f(t)
            fin = np.isfinite(g)
            assert t.shape == g.shape and (np.isfinite(t) == fin).all(), n
            assert np.abs(t[fin] - g[fin]).max() <= tol, n


def test_pt_reverie_oracle_matches_reference_golden():
    """REVERIE twin (pretrain_cmt_reverie.py / vilmodel_reverie.py): objv1 tokens in the trajectory encoder, MRC with the
    object-region branch, SAP, OG."""
    torch.set_num_threads(8)
    gold = common.load_gold('reverie_pt')
    B, L, O, seed = [int(x) for x in gold['meta']]
    cfg = DSRGConfig(**PT_CFG_REVERIE)
    sd = common.synth_sd('reverie_pt')
    sd['bert.embeddings.word_embeddings.weight'] = sd['mlm_head.predictions.decoder.weight']
    for k in sd:
        sd[k].requires_grad_(True)
    for task in ('mlm', 'mrc', 'sap', 'og'):
        b = common.to_t(synth.pretrain_batch(B, L, O, seed, task, reverie=True))
        loss = pretrain_oracle.forward(sd, cfg, b, task, True)
        assert loss.shape == gold[task + '_loss'].shape and np.abs(loss.detach().numpy() - gold[task + '_loss']).max() <= 2e-5, task
        for t in sd.values():
            t.grad = None
        loss.mean().backward()
        n = 0
        for k, g in gold.items():
            if k.startswith(task + '_grad:'):
                mine = sd[k.split(':', 1)[1]].grad.reshape(-1)[:64].numpy()
                assert np.abs(mine - g).max() <= 1e-5 + 1e-4 * np.abs(g).max(), k
                n += 1
        assert n >= 4
        with torch.no_grad():
            check_reverie_outputs(task, pretrain_oracle.forward(sd, cfg, b, task, False), gold, 1e-4)
    # valid_sap_og = the SAP logits and the OG logits of one backbone pass (no golden: see the oracle's header)
    b = common.to_t(synth.pretrain_batch(B, L, O, seed, 'valid_sap_og', reverie=True))
    with torch.no_grad():
        gl, ll, fl, ol = pretrain_oracle.forward(sd, cfg, b, 'valid_sap_og', False)
        sap = pretrain_oracle.forward(sd, cfg, b, 'sap', False)
        og = pretrain_oracle.forward(sd, cfg, b, 'og', False)
    assert torch.equal(torch.nan_to_num(fl, neginf=-9e9), torch.nan_to_num(sap[2], neginf=-9e9))
    assert torch.equal(torch.nan_to_num(ol, neginf=-9e9), torch.nan_to_num(og, neginf=-9e9))
