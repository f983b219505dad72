"""GPU box only: cProfile of the pre-training forward (host side), cumulative times of our own functions."""
import cProfile, pstats, sys, io, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
batches = {t: bench.T(synth.pretrain_batch(bench.PT_B, bench.PT_L, 3, 200, t, min_steps=4, max_steps=7), dev) for t in bench.PT_TASKS}
def fwd(i):
    task = bench.PT_TASKS[i % 3]
    return model(batches[task], task, compute_loss=True).mean()
for i in range(6): fwd(i)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(12): fwd(i)
print('forward host ms', (time.perf_counter() - t) / 12 * 1e3)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(12): fwd(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats('dsrg_amd|bench', 45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25); print(s.getvalue()[:5000])
