"""GPU box only: which Python lines issue the ATen glue kernels (fill / copy / add / mul ...) of one pre-training step."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from dsrg_amd import synth
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
def step(i):
    task = bench.PT_TASKS[i % len(bench.PT_TASKS)]
    opt.zero_grad(set_to_none=True)
    loss = model(batches[task], task, compute_loss=True).mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt.step()
for i in range(6): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(3): step(i)
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    for k in (ev.kernels or []):
        if 'FillFunctor<float' in k.name or 'copyBuffer' in k.name or 'direct_copy' in k.name:
            chain, e = [], ev
            while e is not None and len(chain) < 4:
                chain.append(e.name); e = e.cpu_parent
            cnt[(k.name[:60], ' < '.join(chain), str(ev.input_shapes)[:50])] += 1
for (kn, ch, shp), c in cnt.most_common(40):
    print('%6.1f/step  %-60s %s %s' % (c / 3, kn, ch, shp))
