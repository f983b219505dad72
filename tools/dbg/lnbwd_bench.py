"""GPU box: time dsrg_layernorm_bwd(_drop) per launch for the row counts of the pre-training step (HIP events around 50 launches)."""
import sys, torch
sys.path.insert(0, '.')
from dsrg_amd import kernels as K

H = 768
for M in (648, 888, 1920, 4932):
    for drop in (0.0, 0.1):
        x = torch.randn(M, H, device='cuda')
        r = torch.randn(M, H, device='cuda').bfloat16()
        dy = torch.randn(M, H, device='cuda').bfloat16()
        g = torch.ones(H, device='cuda')
        f = lambda: K.layernorm_bwd(x, r, g, dy, 1e-12, drop, 1234, 0)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(gr, stream=s):
                for _ in range(50):
                    f()
            gr.replay(); torch.cuda.synchronize()
            e0.record(s); gr.replay(); e1.record(s)
        torch.cuda.synchronize()
        print('M=%5d drop=%.1f  %.1f us/launch' % (M, drop, e0.elapsed_time(e1) * 1e3 / 50), flush=True)
