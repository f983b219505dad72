"""Golden vectors for the pre-training model (GlocalTextPathCMTPreTraining), produced by the reference's own
pretrain_src/model files.  Called from tools/make_golden.py (--only pt)."""
import json
import os
import numpy as np
import torch

from dsrg_amd import synth

PT_CFG = '/root/reference/pretrain_src/config/r2r_model_config_objv2.json'
GRAD_KEYS = ['bert.global_encoder.sprel_linear.weight', 'bert.img_embeddings.img_linear.bias',
             'bert.local_encoder.encoder.x_layers.3.visn_output.LayerNorm.weight',
             'bert.global_encoder.encoder.x_layers.0.visual_attention.att.query.bias',
             'bert.embeddings.LayerNorm.weight', 'bert.img_embeddings.adaptive_pano_attn.weight']


def T(b):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()}


def run_pt(import_pt, ref_config, load_synth, out):
    mod = import_pt()
    cfgd = json.load(open(PT_CFG))
    cfg = ref_config(**cfgd, pretrain_tasks=['mlm', 'mrc', 'sap'])
    cfg.hidden_dropout_prob = 0.0
    cfg.attention_probs_dropout_prob = 0.0
    model = mod.GlocalTextPathCMTPreTraining(cfg)
    shapes = load_synth(model)
    model.train()          # dropout p = 0 -> deterministic; grads flow
    gold = {}
    B, L, seed = 3, 40, 77
    for task in ('mlm', 'mrc', 'sap'):
        b = T(synth.pretrain_batch(B, L, 3, seed, task))
        model.zero_grad()
        loss = model(b, task, compute_loss=True)
        loss.mean().backward()
        gold[task + '_loss'] = loss.detach().numpy()
        named = dict(model.named_parameters())
        for k in GRAD_KEYS:
            g = named[k].grad
            if g is not None:
                gold[task + '_grad:' + k] = g.detach().reshape(-1)[:64].numpy().copy()
        gold[task + '_gradnorm'] = np.array([float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))])
        with torch.no_grad():
            o = model(b, task, compute_loss=False)
        if task == 'mlm':
            gold['mlm_scores_s'] = o[:, ::97].numpy()
        elif task == 'mrc':
            gold['mrc_logits_s'] = o[0][:, ::13].numpy()
        else:
            for n, t in zip(('global_logits', 'local_logits', 'fused_logits'), o[:3]):
                gold['sap_' + n] = t.numpy()
    gold['meta'] = np.array([B, L, 3, seed])
    np.savez_compressed(os.path.join(out, 'r2r_pt.npz'), **gold)
    with open(os.path.join(out, 'r2r_pt_state_dict_keys.txt'), 'w') as f:
        for k in sorted(shapes):
            f.write('%s %s\n' % (k, ','.join(map(str, shapes[k]))))
    print('r2r_pt ok: %d tensors, %.3f M params' % (len(shapes), sum(int(np.prod(v)) for v in shapes.values()) / 1e6))
    print({k: v for k, v in gold.items() if k.endswith('_loss') or k.endswith('gradnorm')})


PT_CFG_REVERIE = '/root/reference/pretrain_src/config/reverie_model_config.json'
GRAD_KEYS_REVERIE = GRAD_KEYS + ['bert.img_embeddings.obj_reverie_linear.weight', 'bert.img_embeddings.vis_obj_linear.0.bias',
                                 'og_head.net.0.bias', 'image_classifier.net.3.bias']


def run_pt_reverie(import_pt, ref_config, load_synth, out):
    """REVERIE pre-training twin (pretrain_cmt_reverie.py over vilmodel_reverie.py): tasks mlm / mrc (view + object regions) /
    sap / og on batches with 0..20 objv1 tokens per step.  `valid_sap_og` is not captured: the reference's own call omits
    the objv2 features its backbone needs (pretrain_cmt.py:337-341) and raises."""
    mod = import_pt(reverie=True)
    cfgd = json.load(open(PT_CFG_REVERIE))
    cfg = ref_config(**cfgd, pretrain_tasks=['mlm', 'mrc', 'sap', 'og'])
    cfg.hidden_dropout_prob = 0.0
    cfg.attention_probs_dropout_prob = 0.0
    model = mod.GlocalTextPathCMTPreTraining(cfg)
    shapes = load_synth(model)
    model.train()
    gold = {}
    B, L, seed = 3, 40, 78
    for task in ('mlm', 'mrc', 'sap', 'og'):
        b = T(synth.pretrain_batch(B, L, 3, seed, task, reverie=True))
        model.zero_grad()
        loss = model(b, task, compute_loss=True)
        loss.mean().backward()
        gold[task + '_loss'] = loss.detach().numpy()
        named = dict(model.named_parameters())
        for k in GRAD_KEYS_REVERIE:
            g = named[k].grad if k in named else None
            if g is not None:
                gold[task + '_grad:' + k] = g.detach().reshape(-1)[:64].numpy().copy()
        gold[task + '_gradnorm'] = np.array([float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))])
        with torch.no_grad():
            o = model(b, task, compute_loss=False)
        if task == 'mlm':
            gold['mlm_scores_s'] = o[:, ::97].numpy()
        elif task == 'mrc':
            gold['mrc_logits_s'] = o[0][:, ::13].numpy()
            gold['mrc_obj_logits_s'] = o[2][:, ::13].numpy()
        elif task == 'sap':
            for n, t in zip(('global_logits', 'local_logits', 'fused_logits'), o[:3]):
                gold['sap_' + n] = t.numpy()
        else:
            gold['og_logits'] = o.numpy()
    gold['meta'] = np.array([B, L, 3, seed])
    np.savez_compressed(os.path.join(out, 'reverie_pt.npz'), **gold)
    with open(os.path.join(out, 'reverie_pt_state_dict_keys.txt'), 'w') as f:
        for k in sorted(shapes):
            f.write('%s %s\n' % (k, ','.join(map(str, shapes[k]))))
    print('reverie_pt ok: %d tensors, %.3f M params' % (len(shapes), sum(int(np.prod(v)) for v in shapes.values()) / 1e6))
    print({k: v for k, v in gold.items() if k.endswith('_loss') or k.endswith('gradnorm')})
