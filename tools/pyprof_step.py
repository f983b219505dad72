import cProfile, pstats, sys, io, os
sys.path.insert(0, '.')
import torch, bench
from dsrg_amd.runner import GraphedNavStep
dev = torch.device('cuda', 0)
model, cfg, sd, lbs, pb, nb, dims = bench.build('c2', torch.bfloat16, dev)
with torch.no_grad():
    l = bench.T(lbs[0], dev)
    txt = model('language', l)
runner = GraphedNavStep(model, txt, l['txt_masks'], bench.T(pb, dev), bench.T(nb, dev), use_graph=False)
for _ in range(5): runner.step()
torch.cuda.synchronize()
import time
t=time.perf_counter()
for _ in range(20): runner.step()
torch.cuda.synchronize()
print('eager ms/step', (time.perf_counter()-t)/20*1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): runner.step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30); print(s.getvalue()[:7000])
