#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE model files (read-only, /root/reference)
on deterministic synthetic weights/inputs (dsrg_amd.synth).  Runs only in the build
container; the reference never travels.  Output: tests/golden/*.npz (data only).

Usage: python tools/make_golden.py [--out tests/golden]
Recipe: SURVEY.md section 8c (two ordinary-Python-error shims for transformers 5.x).
"""
import argparse
import importlib
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsrg_amd import synth  # noqa: E402
from dsrg_amd.config import DSRGConfig  # noqa: E402

REF = '/root/reference'


def ref_config(**kw):
    from transformers import BertConfig
    cfg = BertConfig()
    for k, v in DSRGConfig(**kw).to_dict().items():
        setattr(cfg, k, v)
    return cfg


def import_ft(reverie):
    for m in [k for k in sys.modules if k == 'models' or k.startswith('models.')]:
        del sys.modules[m]
    sys.path.insert(0, os.path.join(REF, 'finetune_src'))
    try:
        mod = importlib.import_module('models.vilmodel_reverie' if reverie else 'models.vilmodel')
    finally:
        sys.path.pop(0)
    mod.GlocalTextPathNavCMT.init_weights = lambda self: None   # shim 1 (weights are overwritten)
    return mod


def import_pt(reverie=False):
    for m in [k for k in sys.modules if k == 'model' or k.startswith('model.')]:
        del sys.modules[m]
    sys.path.insert(0, os.path.join(REF, 'pretrain_src'))
    try:
        mod = importlib.import_module('model.pretrain_cmt_reverie' if reverie else 'model.pretrain_cmt')
        vil = importlib.import_module('model.vilmodel_reverie' if reverie else 'model.vilmodel')
    finally:
        sys.path.pop(0)
    vil.GlocalTextPathCMT.init_weights = lambda self: None
    mod.GlocalTextPathCMTPreTraining.init_weights = lambda self: None
    mod.GlocalTextPathCMTPreTraining._tie_or_clone_weights = \
        lambda self, a, b: setattr(a, 'weight', b.weight)           # shim 2
    return mod


def load_synth(model, seed=0):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    return shapes


def T(b):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()}


def sample(x, step):
    """row sample + checksums of a [B,N,H] tensor."""
    x = x.detach().numpy()
    return x[:, ::step].copy(), np.array([x.astype(np.float64).sum(), np.abs(x).astype(np.float64).sum()])


def run_ft(reverie, B, L, G, O, seed, out, tag, No=20):
    mod = import_ft(reverie)
    cfg = ref_config()
    model = mod.GlocalTextPathNavCMT(cfg)
    shapes = load_synth(model)
    gold = {}
    with torch.no_grad():
        lb = synth.language_batch(B, L, seed)
        txt = model('language', T(lb))
        gold['txt_embeds_s'], gold['txt_embeds_ck'] = sample(txt, 9)
        gold['txt_cls'] = txt[:, 0].numpy()
        cls = None
        for t in range(3):                      # recurrent chain, agent.py:434-472
            pb = synth.panorama_batch(B, 36, O, seed + 10 * t, reverie=reverie, No=No)
            pe, pm, pf = model('panorama', T(pb))
            Vp = pe.shape[1] + 2
            vlens = pm.sum(1).numpy() + 2
            nb = synth.navigation_batch(B, G, Vp, seed + 10 * t, ncand=pb['ncand'], nav_types=pb['nav_types'],
                                        vp_lens=vlens, reverie=reverie)
            nb = T(nb)
            mem = torch.zeros(B, 768) if cls is None else cls
            nb['vp_img_embeds'] = torch.cat([torch.zeros(B, 1, 768), mem[:, None], pe], 1)
            nb['gmap_img_embeds'][:, 1] = mem
            nb['gmap_img_embeds'][:, 2] = pf
            nb['txt_embeds'], nb['txt_masks'] = txt, torch.from_numpy(lb['txt_masks'])
            o = model('navigation', nb)
            cls = o['cls_embeds']
            p = 's%d_' % t
            gold[p + 'pano_embeds_s'], gold[p + 'pano_embeds_ck'] = sample(pe, 5)
            gold[p + 'pano_masks'] = pm.numpy()
            gold[p + 'pano_fused'] = pf.numpy()
            for k in ('global_logits', 'local_logits', 'fused_logits', 'cls_embeds'):
                gold[p + k] = o[k].numpy()
            if o['obj_logits'] is not None:
                gold[p + 'obj_logits'] = o['obj_logits'].numpy()
            gold[p + 'gmap_embeds_s'], gold[p + 'gmap_embeds_ck'] = sample(o['gmap_embeds'], 3)
            gold[p + 'vp_embeds_s'], gold[p + 'vp_embeds_ck'] = sample(o['vp_embeds'], 5)
            gold[p + 'argmax'] = o['fused_logits'].argmax(1).numpy()
    gold['meta'] = np.array([B, L, G, O, seed, int(reverie), No])
    np.savez_compressed(os.path.join(out, tag + '.npz'), **gold)
    names = sorted(shapes)
    with open(os.path.join(out, tag + '_state_dict_keys.txt'), 'w') as f:
        for k in names:
            f.write('%s %s\n' % (k, ','.join(map(str, shapes[k]))))
    print(tag, 'ok: %d tensors, %.3f M params' % (len(names), sum(int(np.prod(shapes[k])) for k in names) / 1e6))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if a.only in ('', 'r2r'):
        run_ft(False, 8, 80, 20, 3, 1234, a.out, 'r2r_c1')
    if a.only in ('', 'reverie'):
        run_ft(True, 8, 80, 20, 3, 1234, a.out, 'reverie_c3')
    if a.only in ('', 'pt'):
        try:
            from make_golden_pt import run_pt
        except ImportError:
            run_pt = None
        if run_pt is not None:
            run_pt(import_pt, ref_config, load_synth, a.out)
    if a.only in ('', 'pt_reverie'):
        from make_golden_pt import run_pt_reverie
        run_pt_reverie(import_pt, ref_config, load_synth, a.out)
