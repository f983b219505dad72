"""Shared parity harness: replays the golden recurrent chain (tools/make_golden.py: run_ft)
through any implementation exposing fwd(mode, batch) and returns the same keys as the fixture."""
import os
import numpy as np
import torch

from dsrg_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_shapes(tag):
    shapes = {}
    with open(os.path.join(GOLDEN, tag + '_state_dict_keys.txt')) as f:
        for line in f:
            k, s = line.split()
            shapes[k] = tuple(int(x) for x in s.split(','))
    return shapes


def load_gold(tag):
    return dict(np.load(os.path.join(GOLDEN, tag + '.npz')))


def synth_sd(tag, device='cpu', seed=0):
    sd = synth.synth_state_dict(load_shapes(tag), seed)
    return {k: torch.from_numpy(v).to(device) for k, v in sd.items()}


def to_t(b, device='cpu'):
    return {k: (torch.from_numpy(v).to(device) if isinstance(v, np.ndarray) else v) for k, v in b.items()}


def _sample(x, step):
    x = x.detach().float().cpu().numpy()
    return x[:, ::step].copy(), np.array([x.astype(np.float64).sum(), np.abs(x).astype(np.float64).sum()])


def run_chain(fwd, meta, device='cpu', steps=3, dtype=torch.float32):
    B, L, G, O, seed, reverie, No = [int(x) for x in meta]
    reverie = bool(reverie)
    out = {}
    lb = synth.language_batch(B, L, seed)
    txt = fwd('language', to_t(lb, device))
    out['txt_embeds_s'], out['txt_embeds_ck'] = _sample(txt, 9)
    out['txt_cls'] = txt[:, 0].float().cpu().numpy()
    cls = None
    for t in range(steps):
        pb = synth.panorama_batch(B, 36, O, seed + 10 * t, reverie=reverie, No=No)
        pe, pm, pf = fwd('panorama', to_t(pb, device))
        Vp = pe.shape[1] + 2
        vlens = pm.sum(1).cpu().numpy() + 2
        nb = to_t(synth.navigation_batch(B, G, Vp, seed + 10 * t, ncand=pb['ncand'], nav_types=pb['nav_types'],
                                         vp_lens=vlens, reverie=reverie), device)
        mem = torch.zeros(B, 768, device=device, dtype=pe.dtype) if cls is None else cls
        nb['vp_img_embeds'] = torch.cat([torch.zeros(B, 1, 768, device=device, dtype=pe.dtype), mem[:, None], pe], 1)
        nb['gmap_img_embeds'] = nb['gmap_img_embeds'].to(pe.dtype)
        nb['gmap_img_embeds'][:, 1] = mem
        nb['gmap_img_embeds'][:, 2] = pf
        nb['txt_embeds'], nb['txt_masks'] = txt, torch.from_numpy(lb['txt_masks']).to(device)
        o = fwd('navigation', nb)
        cls = o['cls_embeds']
        p = 's%d_' % t
        out[p + 'pano_embeds_s'], out[p + 'pano_embeds_ck'] = _sample(pe, 5)
        out[p + 'pano_masks'] = pm.cpu().numpy()
        out[p + 'pano_fused'] = pf.float().cpu().numpy()
        for k in ('global_logits', 'local_logits', 'fused_logits', 'cls_embeds'):
            out[p + k] = o[k].float().cpu().numpy()
        if o['obj_logits'] is not None:
            out[p + 'obj_logits'] = o['obj_logits'].float().cpu().numpy()
        out[p + 'gmap_embeds_s'], out[p + 'gmap_embeds_ck'] = _sample(o['gmap_embeds'], 3)
        out[p + 'vp_embeds_s'], out[p + 'vp_embeds_ck'] = _sample(o['vp_embeds'], 5)
        out[p + 'argmax'] = o['fused_logits'].float().argmax(1).cpu().numpy()
    return out


def compare(out, gold, tol, logit_tol=None, check_argmax=True, steps=3):
    """max-abs on finite entries, -inf positions must coincide. Returns dict key -> max err."""
    errs = {}
    logit_tol = tol if logit_tol is None else logit_tol
    for k, g in gold.items():
        if k == 'meta' or k.endswith('_ck') or k not in out:
            continue
        o = out[k]
        if k.endswith('argmax'):
            continue
        if k.endswith('pano_masks'):
            assert (o == g).all(), k
            continue
        assert o.shape == g.shape, (k, o.shape, g.shape)
        fin = np.isfinite(g)
        assert (np.isfinite(o) == fin).all(), '%s: -inf positions differ' % k
        e = float(np.abs(o[fin] - g[fin]).max()) if fin.any() else 0.0
        errs[k] = e
        lim = logit_tol if 'logits' in k else tol
        assert e <= lim, '%s: max-abs %.3e > %.1e' % (k, e, lim)
    if check_argmax:
        for t in range(steps):
            assert (out['s%d_argmax' % t] == gold['s%d_argmax' % t]).all(), 'argmax step %d' % t
    return errs


def chain_loss(fwd, meta, device='cpu', steps=2, B=4):
    """Fine-tuning loss of a short rollout as the agent accumulates it (agent.py:502,578): sum over steps of
    CrossEntropyLoss(ignore_index=-100, reduction='sum') on the fused logits, with the RMF memory and the panorama
    embeddings carried from step to step INSIDE the autograd graph.  Targets are a fixed function of the -inf pattern."""
    _, L, G, O, seed, reverie, No = [int(x) for x in meta]
    reverie = bool(reverie)
    lb = synth.language_batch(B, L, seed)
    txt = fwd('language', to_t(lb, device))
    cls, loss = None, 0.0
    for t in range(steps):
        pb = synth.panorama_batch(B, 36, O, seed + 10 * t, reverie=reverie, No=No)
        pe, pm, pf = fwd('panorama', to_t(pb, device))
        Vp = pe.shape[1] + 2
        vlens = pm.sum(1).cpu().numpy() + 2
        nb = to_t(synth.navigation_batch(B, G, Vp, seed + 10 * t, ncand=pb['ncand'], nav_types=pb['nav_types'],
                                         vp_lens=vlens, reverie=reverie), device)
        mem = torch.zeros(B, 768, device=device, dtype=pe.dtype) if cls is None else cls
        nb['vp_img_embeds'] = torch.cat([torch.zeros(B, 1, 768, device=device, dtype=pe.dtype), mem[:, None], pe], 1)
        g = nb['gmap_img_embeds'].to(pe.dtype)
        nb['gmap_img_embeds'] = torch.cat([g[:, :1], mem[:, None], pf[:, None], g[:, 3:]], 1)
        nb['txt_embeds'], nb['txt_masks'] = txt, torch.from_numpy(lb['txt_masks']).to(device)
        o = fwd('navigation', nb)
        cls = o['cls_embeds']
        fl = o['fused_logits'].float()
        fin = torch.isfinite(fl.detach())
        order = torch.argsort((~fin).int(), dim=1, stable=True)            # finite positions first, in index order
        pick = (torch.arange(B, device=fl.device) + t) % fin.sum(1)
        tgt = order.gather(1, pick[:, None])[:, 0]
        if B > 2:
            tgt[B - 1] = -100
        loss = loss + torch.nn.functional.cross_entropy(fl, tgt, ignore_index=-100, reduction='sum')
    return loss
