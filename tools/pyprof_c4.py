"""Host-side profile (cProfile) of the pre-training step of bench.py --config c4 (GPU box only)."""
import cProfile, pstats, sys, io, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dsrg_amd import synth
from dsrg_amd.config import DSRGConfig
from dsrg_amd.pretrain import GlocalTextPathCMTPreTraining
dev = torch.device('cuda', 0)
cfg = DSRGConfig(use_lang2visn_attn=True, obj_feat_size=0, pretrain_tasks=bench.PT_TASKS)
model = GlocalTextPathCMTPreTraining(cfg, compute_dtype=torch.bfloat16)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes).items()})
model = model.to(dev).train()
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=5e-5, betas=(0.9, 0.98), fused=True)
batches = {t: bench.T(synth.pretrain_batch(bench.PT_B, bench.PT_L, 3, 200, t, min_steps=4, max_steps=7), dev) for t in bench.PT_TASKS}
def step(i):
    task = bench.PT_TASKS[i % len(bench.PT_TASKS)]
    opt.zero_grad(set_to_none=True)
    loss = model(batches[task], task, compute_loss=True).mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    opt.step()
for i in range(6): step(i)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(12): step(i)
torch.cuda.synchronize()
print('ms/step', (time.perf_counter() - t) / 12 * 1e3)
# forward only / backward only host time
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(6):
    loss = model(batches[bench.PT_TASKS[i % 3]], bench.PT_TASKS[i % 3], compute_loss=True).mean()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('forward: host %.1f ms/step, +%.1f ms to drain' % ((t1 - t) / 6 * 1e3, (t2 - t1) / 6 * 1e3))
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile(); pr.enable()
for i in range(12): step(i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40); print(s.getvalue()[:9000])
