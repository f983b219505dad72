"""GPU box: time dsrg_embed_ln_bwd alone on the four shapes of a pre-training step (text / panorama / graph-map / viewpoint
tokens) and on variants that isolate one cost (duplicate table rows, feature width)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from dsrg_amd import kernels as K  # noqa: E402

H = 768
dev = 'cuda'
g = torch.Generator(device='cpu').manual_seed(1)


def r(*s, dt=torch.float32):
    return torch.randn(*s, generator=g).to(dev).to(dt)


def timeit(name, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print('%-46s median %8.1f us  min %8.1f us (includes the zero-fills of the gradient buffers)' % (name, ts[len(ts) // 2], ts[0]), flush=True)


def main():
    bf = torch.bfloat16
    lno = (r(H) + 1, r(H))
    lnf = (r(H) + 1, r(H))
    vec = r(H)
    # text
    M = 24 * 80
    Ew, Ep = r(30522, H), r(512, H)
    ids = torch.randint(1000, 30000, (M,), generator=g)
    pad = (torch.arange(M) % 80) >= 45
    ids_pad = torch.where(pad, torch.zeros_like(ids), ids).to(dev)
    pos = (torch.arange(M) % 80).int().to(dev)
    dout = r(M, H, dt=bf)
    timeit('text M=1920, 44% padding ids', lambda: K.embed_ln_bwd(M, H, dout, E1=Ew, idx1=ids_pad, E2=Ep, idx2=pos, vec=vec, lno=lno))
    timeit('text M=1920, distinct ids', lambda: K.embed_ln_bwd(M, H, dout, E1=Ew, idx1=ids.to(dev), E2=Ep, idx2=pos, vec=vec, lno=lno))
    timeit('text M=1920, small word table (no 94 MB fill)', lambda: K.embed_ln_bwd(M, H, dout, E1=Ew[:1920], idx1=torch.arange(M, device=dev), E2=Ep, idx2=pos, vec=vec, lno=lno))
    timeit('text M=1920, no tables', lambda: K.embed_ln_bwd(M, H, dout, vec=vec, lno=lno))
    # panorama / graph-map / viewpoint
    for name, M, F in (('pano', 137 * 36, 7), ('gmap', 288, 7), ('vp', 216, 14)):
        base, feat, Wf, b = r(M, H, dt=bf), r(M, F), r(H, F), r(H)
        E1 = r(3, H)
        i1 = torch.randint(0, 3, (M,), generator=g).to(dev)
        dout = r(M, H, dt=bf)
        timeit('%s M=%d F=%d full' % (name, M, F), lambda: K.embed_ln_bwd(M, H, dout, base=base, feat=feat, Wf=Wf, bf=b, lnf=lnf, E1=E1, idx1=i1, lno=lno))
        timeit('%s M=%d F=%d no table' % (name, M, F), lambda: K.embed_ln_bwd(M, H, dout, base=base, feat=feat, Wf=Wf, bf=b, lnf=lnf, lno=lno))
        timeit('%s M=%d no feat' % (name, M), lambda: K.embed_ln_bwd(M, H, dout, base=base, E1=E1, idx1=i1, lno=lno))


if __name__ == '__main__':
    main()
