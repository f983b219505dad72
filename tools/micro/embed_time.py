"""GPU box only: isolated launch time of the nav-step's token-embedding launches (HIP events around a graph of 20 launches)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from dsrg_amd import functional as Fn


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev, cd, H = 'cuda', torch.bfloat16, 768
    r = lambda *s: torch.randn(*s, device=dev)
    with torch.no_grad():
        for name, M, F in (('vp tokens (2432 rows, F=14)', 2432, 14), ('gmap tokens (1280 rows, F=7 + step table)', 1280, 7),
                           ('pano tokens (2304 rows, F=7 + nav type + vec)', 2304, 7)):
            base, feat = r(M, H).to(cd), r(M, F)
            Wf, bf, g, b = r(H, F), r(H), r(H), r(H)
            E1, idx = r(100, H), torch.randint(0, 3, (M,), device=dev)
            kw = dict(base=base, feat=feat, Wf=Wf, bf=bf, lnf=(g, b))
            if 'gmap' in name or 'pano' in name:
                kw.update(E1=E1, idx1=idx)
            if 'pano' in name:
                kw.update(vec=r(H), lno=(g, b))
            print('%-50s %.2f us' % (name, timed(lambda: Fn.embed_ln(M, H, cd, dev, **kw))))
            if '--stamps' in sys.argv:      # DSRG_HIP_LIB = a -DDSRG_EMBED_STAMPS build
                import ctypes as C, numpy as np
                from dsrg_amd import _lib
                fn = _lib.load().dsrg_debug_embed_stamps
                fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
                torch.cuda.synchronize(); Fn.embed_ln(M, H, cd, dev, **kw); torch.cuda.synchronize()
                buf = np.zeros(4096 * 8, np.uint64)
                assert fn(buf.ctypes.data, buf.nbytes) == 0
                nwg = (M + 3) // 4
                st = buf.reshape(4096, 8)[:nwg, :7].astype(np.int64)
                names = ['loads issued', 'W staged', 'projection', 'LN feat', 'LN out', 'stores']
                print('    wgs %d  span %d clk | ' % (nwg, st[:, 6].max() - st[:, 0].min()) + ' | '.join('%s %d' % (n, np.median(st[:, i + 1] - st[:, i])) for i, n in enumerate(names))
                      + ' | total %d' % np.median(st[:, 6] - st[:, 0]))
        x = r(2432, H)
        print('%-50s %.2f us' % ('layernorm fp32 -> bf16, 2432 rows', timed(lambda: Fn.layernorm(x, g, b, 1e-12, out_dtype=cd))))


if __name__ == '__main__':
    main()
