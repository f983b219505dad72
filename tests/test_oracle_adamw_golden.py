"""cpu: oracle/adamw_oracle.py (numpy restatement of clip_grad_norm_ + the reference's AdamW.step) against
tests/golden/adamw.npz, which tools/make_golden_adamw.py produced by running the reference's own optimizer class."""
import os
import numpy as np

from oracle import adamw_oracle
from tools.make_golden_adamw import SHAPES, DECAY, STEPS, SKIP, tensors

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'adamw.npz')
CASES = (('clip', dict(betas=(0.9, 0.98)), 0.01, 5.0), ('noclip_nobias', dict(betas=(0.9, 0.999), correct_bias=False), 0.05, -1.0))


def schedule(step):
    lr = 1e-3 * (step + 1) / STEPS
    gs = [None if i in SKIP.get(step, []) else (g * np.float32(3.0 if step == 1 else 0.02)).astype(np.float32)
          for i, g in enumerate(tensors(100 + step))]
    return lr, gs


def test_oracle_matches_reference_adamw():
    gold = np.load(GOLD)
    for name, kw, wd, max_norm in CASES:
        ps = tensors(1)
        opt = adamw_oracle.AdamW(ps, [0 if d else 1 for d in DECAY], [1e-3, 1e-3], [wd, 0.0], **kw)
        for step in range(STEPS):
            lr, gs = schedule(step)
            opt.lr = [lr, lr]
            if max_norm > 0:
                norm, gs = adamw_oracle.clip_grad_norm(gs, max_norm)
                assert abs(norm - gold['%s_norm_%d' % (name, step)][0]) <= 1e-5 * norm
            opt.step(gs)
            if step in (1, STEPS - 1):
                for i, p in enumerate(ps):
                    want = gold['%s_p_%d_%d' % (name, step, i)]
                    assert np.abs(p - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), (name, step, i)
        for i in range(len(ps)):
            for a, k in ((opt.m[i], 'm'), (opt.v[i], 'v')):
                want = gold['%s_%s_%d' % (name, k, i)]
                assert np.abs(a - want).max() <= 2e-6 * max(1e-3, np.abs(want).max()), (name, k, i)
