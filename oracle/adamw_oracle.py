"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of the optimizer step of the pre-training path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path (dsrg_amd.optim over
libdsrg_hip.so) never does.

  * clip_grad_norm: torch.nn.utils.clip_grad_norm_ as pretrain_src/train_r2r.py:283-289 calls it (2-norm over all
    gradients, coefficient clamp(max_norm / (total + 1e-6), max=1), gradients scaled in place);
  * AdamW.step: pretrain_src/optim/adamw.py:30-92 (moving averages :66-67, denom = sqrt(v) + eps :68, bias-corrected step
    size :70-74, update :76, decoupled weight decay applied AFTER the update :87-88; per-parameter step counters :62, a
    parameter without a gradient is skipped :44-45).

Pinned by tests/golden/adamw.npz, which tools/make_golden_adamw.py produced by running the reference's own AdamW class.
All arithmetic in float32, like the reference's tensors.
"""
import math
import numpy as np

F = np.float32


def clip_grad_norm(grads, max_norm):
    """grads: list of float32 arrays or None.  Returns (total_norm, scaled grads)."""
    live = [g for g in grads if g is not None]
    total = F(math.sqrt(sum(float(np.sum(g.astype(np.float64) ** 2)) for g in live)))
    coef = min(F(max_norm) / (total + F(1e-6)), F(1.0))
    return total, [None if g is None else (g * F(coef)).astype(F) for g in grads]


class AdamW:
    def __init__(self, params, group_of, lrs, weight_decays, betas=(0.9, 0.999), eps=1e-6, correct_bias=True):
        """params: list of float32 arrays (updated in place); group_of[i]: parameter i's group; lrs / weight_decays per group"""
        self.p, self.group_of = params, group_of
        self.lr, self.wd = list(lrs), list(weight_decays)
        self.b1, self.b2, self.eps, self.correct_bias = betas[0], betas[1], eps, correct_bias
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]
        self.step_count = [0] * len(params)

    def step(self, grads):
        for i, (p, g) in enumerate(zip(self.p, grads)):
            if g is None:
                continue
            lr, wd = self.lr[self.group_of[i]], self.wd[self.group_of[i]]
            self.step_count[i] += 1
            self.m[i] = (self.m[i] * F(self.b1) + g * F(1.0 - self.b1)).astype(F)
            self.v[i] = (self.v[i] * F(self.b2) + g * g * F(1.0 - self.b2)).astype(F)
            denom = np.sqrt(self.v[i]) + F(self.eps)
            step_size = lr
            if self.correct_bias:
                t = self.step_count[i]
                step_size = step_size * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
            p -= (F(step_size) * (self.m[i] / denom)).astype(F)
            if wd > 0.0:
                p += (p * F(-lr * wd)).astype(F)
