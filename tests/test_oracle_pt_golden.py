"""not-gpu: the pre-training CPU oracle reproduces the reference's own loss vectors / logits
(tests/golden/r2r_pt.npz from tools/make_golden_pt.py) and, through autograd, its gradient slices."""
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
