import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
import common
from dsrg_amd import synth, functional as Fn
from test_gpu_pretrain import _model_reverie
gold = common.load_gold('reverie_pt')
B, L, O, seed = [int(x) for x in gold['meta']]
grads = {}
for dtype, fused in ((torch.float32, True), (torch.bfloat16, False), (torch.bfloat16, True)):
    Fn.FUSED_FFN = fused
    m = _model_reverie(dtype)
    b = common.to_t(synth.pretrain_batch(B, L, O, seed, 'sap', reverie=True), 'cuda')
    m.zero_grad()
    m(b, 'sap', compute_loss=True).mean().backward()
    grads[(dtype, fused)] = {k: p.grad.double().clone() for k, p in m.named_parameters() if p.grad is not None}
ref = grads[(torch.float32, True)]
for key in ('bert.img_embeddings.adaptive_pano_attn.weight', 'bert.img_embeddings.adaptive_pano_attn.bias', 'bert.embeddings.LayerNorm.weight',
            'bert.global_encoder.sprel_linear.weight', 'bert.local_encoder.encoder.x_layers.3.visn_output.LayerNorm.weight'):
    for fused in (False, True):
        g = grads[(torch.bfloat16, fused)].get(key)
        if g is None or key not in ref: continue
        r = ref[key]
        print('%-75s %-8s relL2 %.4f  max|d|/max|g| %.4f (first 64: %.4f)' % (key[-75:], 'fused' if fused else 'separate', float((g - r).norm() / r.norm()),
              float((g - r).abs().max() / r.abs().max()), float((g - r).reshape(-1)[:64].abs().max() / r.reshape(-1)[:64].abs().max())))
tot = {f: 0.0 for f in (False, True)}; nrm = 0.0
for k, r in ref.items():
    nrm += float(r.norm() ** 2)
    for f in (False, True):
        tot[f] += float((grads[(torch.bfloat16, f)][k] - r).norm() ** 2)
print('all parameters: relL2 separate %.4f fused %.4f' % ((tot[False] / nrm) ** 0.5, (tot[True] / nrm) ** 0.5))
