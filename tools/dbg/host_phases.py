"""GPU box only: host time per phase of the pre-training step (no syncs inside the step)."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from dsrg_amd import synth, functional as Fn
from dsrg_amd.config import DSRGConfig
from dsrg_amd.pretrain import GlocalTextPathCMTPreTraining
dev = torch.device('cuda', 0)
cfg = DSRGConfig(use_lang2visn_attn=True, obj_feat_size=0, pretrain_tasks=bench.PT_TASKS)
model = GlocalTextPathCMTPreTraining(cfg, compute_dtype=torch.bfloat16)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes).items()})
model = model.to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=5e-5, betas=(0.9, 0.98), fused=True)
batches = {t: bench.T(synth.pretrain_batch(bench.PT_B, bench.PT_L, 3, 200, t, min_steps=4, max_steps=7), dev) for t in bench.PT_TASKS}
acc = collections.defaultdict(float)
pc = time.perf_counter
orig_resolve = Fn.WeightPacker.resolve
def timed_resolve(self, W):
    t = pc(); r = orig_resolve(self, W); acc['  pack resolve (inside forward)'] += pc() - t; return r
Fn.WeightPacker.resolve = timed_resolve
orig_apply = Fn._PackFn.apply
def timed_pack(*a):
    t = pc(); r = orig_apply(*a); acc['  pack fwd (inside forward)'] += pc() - t; return r
Fn._PackFn.apply = timed_pack
orig_bwd = Fn._PackFn.backward
def timed_bwd(ctx, *g):
    t = pc(); r = orig_bwd(ctx, *g); acc['  pack bwd (inside backward)'] += pc() - t; return r
Fn._PackFn.backward = staticmethod(timed_bwd)
def step(i, rec):
    task = bench.PT_TASKS[i % len(bench.PT_TASKS)]
    t0 = pc(); opt.zero_grad(set_to_none=True)
    t1 = pc(); loss = model(batches[task], task, compute_loss=True).mean()
    t2 = pc(); loss.backward()
    t3 = pc(); torch.nn.utils.clip_grad_norm_(params, 5.0)
    t4 = pc(); opt.step()
    t5 = pc()
    if rec:
        for k, v in (('zero_grad', t1 - t0), ('forward', t2 - t1), ('backward', t3 - t2), ('clip', t4 - t3), ('opt.step', t5 - t4)):
            acc[k] += v
for i in range(6): step(i, False)
acc.clear()
torch.cuda.synchronize(); t = pc(); c0 = time.process_time()
N = 30
for i in range(N): step(i, True)
th = pc() - t; cpu = time.process_time() - c0
torch.cuda.synchronize()
print('host %.2f ms/step, wall %.2f ms/step, process CPU time %.2f ms/step' % (th / N * 1e3, (pc() - t) / N * 1e3, cpu / N * 1e3))
for k, v in acc.items():
    print('%-32s %.2f ms/step' % (k, v / N * 1e3))
