"""Golden vectors for the optimizer step of the pre-training path: the reference's OWN AdamW (pretrain_src/optim/adamw.py,
loaded from /root/reference by file path) driven the way pretrain_src/train_r2r.py:270-295 drives it - lr set per step,
torch.nn.utils.clip_grad_norm_, optimizer.step() - on seeded fp32 tensors.  Writes tests/golden/adamw.npz (data only).

    python tools/make_golden_adamw.py
"""
import importlib.util
import os
import numpy as np
import torch

REF = '/root/reference/pretrain_src/optim/adamw.py'
HERE = os.path.dirname(os.path.abspath(__file__))

SHAPES = [(5000,), (37, 19), (4096,), (3,), (8193,), (64, 64)]           # odd sizes, exact chunk, one-over-a-chunk, tiny
DECAY = [True, False, True, False, True, True]                            # two parameter groups, as optim/misc.py:9-19 builds them
STEPS = 4
SKIP = {2: [1, 4]}                                                         # step 2: tensors 1 and 4 have no gradient (a task that leaves heads untouched)


def tensors(seed):
    g = np.random.default_rng(seed)
    return [g.standard_normal(s).astype(np.float32) for s in SHAPES]


def main():
    spec = importlib.util.spec_from_file_location('ref_adamw', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for name, kw, max_norm in (('clip', dict(betas=(0.9, 0.98), weight_decay=0.01), 5.0),
                               ('noclip_nobias', dict(betas=(0.9, 0.999), weight_decay=0.05, correct_bias=False), -1.0)):
        ps = [torch.nn.Parameter(torch.from_numpy(a.copy())) for a in tensors(1)]
        wd = kw.pop('weight_decay')
        opt = mod.AdamW([{'params': [p for p, d in zip(ps, DECAY) if d], 'weight_decay': wd},
                         {'params': [p for p, d in zip(ps, DECAY) if not d], 'weight_decay': 0.0}], lr=1e-3, **kw)
        for step in range(STEPS):
            lr = 1e-3 * (step + 1) / STEPS                 # a moving schedule, train_r2r.py:270-272
            for grp in opt.param_groups:
                grp['lr'] = lr
            gs = tensors(100 + step)
            for i, (p, g) in enumerate(zip(ps, gs)):
                p.grad = None if i in SKIP.get(step, []) else torch.from_numpy(g * (3.0 if step == 1 else 0.02))     # step 1 clips, the others do not
            if max_norm > 0:
                norm = torch.nn.utils.clip_grad_norm_(ps, max_norm)
                out['%s_norm_%d' % (name, step)] = np.array([float(norm)], np.float32)
            opt.step()
            if step in (1, STEPS - 1):                     # after the clipped step, and at the end
                for i, p in enumerate(ps):
                    out['%s_p_%d_%d' % (name, step, i)] = p.detach().numpy().copy()
        for i, p in enumerate(ps):
            out['%s_m_%d' % (name, i)] = opt.state[p]['exp_avg'].numpy().copy()
            out['%s_v_%d' % (name, i)] = opt.state[p]['exp_avg_sq'].numpy().copy()
    np.savez_compressed(os.path.join(HERE, '..', 'tests', 'golden', 'adamw.npz'), **out)
    print('adamw golden: %d arrays' % len(out))


if __name__ == '__main__':
    main()
