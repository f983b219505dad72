#!/usr/bin/env python3
"""Micro-benchmark of dsrg_gemm_nt on the nav-step's GEMM shapes (GPU box only).
usage: python tools/bench_gemm.py [--blas] [--cold] [--sweep] [--only SUBSTR]
  --cold   rotate over enough operand sets (>= 96 MB) that a launch finds A / W / residual beyond the XCD L2s, as it does
           inside a nav-step (back-to-back replays of ONE launch on hot operands read 30-50 % low)
  --sweep  time every tile of the menu (dsrg_gemm_nt_tiled) next to the library's own plan"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsrg_amd import kernels as K

SHAPES = [  # name, [(M,N,K)...] grouped, act, out_f32, residual
    ('x.cq/co/so  g+l N=768 K=768 f32out+res', [(1280, 768, 768), (2432, 768, 768)], 0, True, True),
    ('x.qkv       g+l N=2304 K=768', [(1280, 2304, 768), (2432, 2304, 768)], 0, False, False),
    ('x.ff1 gelu  g+l N=3072 K=768', [(1280, 3072, 768), (2432, 3072, 768)], 1, False, False),
    ('x.ff2       g+l N=768 K=3072 f32out+res', [(1280, 768, 3072), (2432, 768, 3072)], 0, True, True),
    ('g.cq  M=1280 N=768 K=768 f32out+res', [(1280, 768, 768)], 0, True, True),
    ('g.qkv M=1280 N=2304 K=768', [(1280, 2304, 768)], 0, False, False),
    ('g.ff1 M=1280 N=3072 K=768 gelu', [(1280, 3072, 768)], 1, False, False),
    ('g.ff2 M=1280 N=768 K=3072 f32out+res', [(1280, 768, 3072)], 0, True, True),
    ('l.cq  M=2432 N=768 K=768 f32out+res', [(2432, 768, 768)], 0, True, True),
    ('l.qkv M=2432 N=2304 K=768', [(2432, 2304, 768)], 0, False, False),
    ('l.ff1 M=2432 N=3072 K=768 gelu', [(2432, 3072, 768)], 1, False, False),
    ('l.ff2 M=2432 N=768 K=3072 f32out+res', [(2432, 768, 3072)], 0, True, True),
    ('g.ff2 split-K 2', [(1280, 768, 1536)] * 2, 0, True, True),
    ('g.ff2 split-K 3', [(1280, 768, 1024)] * 3, 0, True, True),
    ('l.ff2 split-K 2', [(2432, 768, 1536)] * 2, 0, True, True),
    ('x.ff2 split-K 2 (4 problems K=1536)', [(1280, 768, 1536), (2432, 768, 1536)] * 2, 0, True, True),
    ('x.ff2 split-K 3 (6 problems K=1024)', [(1280, 768, 1024), (2432, 768, 1024)] * 3, 0, True, True),
    ('x.cq split-K 2 (4 problems K=384)', [(1280, 768, 384), (2432, 768, 384)] * 2, 0, True, True),
    ('pano.ff2 split-K 2', [(2304, 768, 1536)] * 2, 0, False, True),
    ('pano.qkv    M=2304 N=2304 K=768', [(2304, 2304, 768)], 0, False, False),
    ('pano.ff1    M=2304 N=3072 K=768', [(2304, 3072, 768)], 1, False, False),
    ('pano.ff2    M=2304 N=768 K=3072 +res', [(2304, 768, 3072)], 0, False, True),
    ('obj.ff1     M=6912 N=3072 K=128', [(6912, 3072, 128)], 1, False, False),
    ('obj.ff2     M=6912 N=128 K=3072 f32out+res', [(6912, 128, 3072)], 0, True, True),
    ('his_map     M=64 N=768 K=2304 f32out', [(64, 768, 2304)], 0, True, False),
    ('poolers     3x M=64 N=768 K=768 tanh', [(64, 768, 768)] * 3, 3, False, False),
    ('heads       g,l,fuse l1 relu f32out', [(1280, 768, 768), (2432, 768, 768), (64, 768, 1536)], 2, True, False),
    ('img_linear  M=2304 N=768 K=768 f32out', [(2304, 768, 768)], 0, True, False),
    ('txtkv       M=5120 N=12288 K=768', [(5120, 12288, 768)], 0, False, False),
    ('square 4096', [(4096, 4096, 4096)], 0, False, False),
]

TILES = [(64, 64, 2), (64, 64, 3), (64, 64, 4), (64, 128, 2), (64, 128, 3), (64, 192, 3), (64, 256, 3), (128, 64, 2), (128, 64, 3), (128, 64, 4),
         (128, 96, 3), (128, 128, 2), (128, 128, 3), (128, 192, 2), (128, 192, 3), (128, 256, 3), (256, 128, 2), (256, 128, 3)]


def time_graph(fn, inner=20, rep=5):
    """HIP-graph replay of `inner` launches: no Python/ctypes time in the measurement"""
    fn(0)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn(0)
    torch.cuda.current_stream().wait_stream(st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(inner):
            fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (rep * inner)


def main():
    dt = torch.bfloat16
    cold, sweep = '--cold' in sys.argv, '--sweep' in sys.argv
    only = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else None
    for name, probs, act, of32, res in SHAPES:
        if only and only not in name:
            continue
        per_set = sum(M * Kd * 2 + N * Kd * 2 + M * N * (4 if of32 else 2) + (M * N * 2 if res else 0) for (M, N, Kd) in probs)
        nsets = max(2, -(-96 * 2 ** 20 // per_set)) if cold else 1
        sets = []
        for _ in range(nsets):
            ps = []
            for (M, N, Kd) in probs:
                x = (torch.rand(M, Kd, device='cuda') * 2 - 1).to(dt)
                w = ((torch.rand(N, Kd, device='cuda') * 2 - 1) * 0.05).to(dt)
                b = torch.rand(N, device='cuda')
                r = (torch.rand(M, N, device='cuda') * 2 - 1).to(dt) if res else None
                ps.append((x, w, b, r))
            sets.append((ps, K.gemm(ps, act, of32)))
        fl = sum(2.0 * M * N * Kd for (M, N, Kd) in probs)
        us = time_graph(lambda i: K.gemm(sets[i % nsets][0], act, of32, outs=sets[i % nsets][1]))
        line = '%-44s %8.1f us  %7.1f TFLOP/s' % (name, us, fl / us / 1e6)
        if sweep:
            ref = [o.float() for o in sets[0][1]]
            best = None
            cells = []
            for t in TILES:
                try:
                    ut = time_graph(lambda i: K.gemm(sets[i % nsets][0], act, of32, outs=sets[i % nsets][1], tile=t))
                except Exception as e:      # tile not built for this dtype
                    continue
                err = max(float((o.float() - r_).abs().max()) for o, r_ in zip(sets[0][1], ref))
                cells.append('%dx%dx%d %.1f%s' % (t + (ut, '' if err < 2e-2 else ' ERR %.3g' % err)))
                if best is None or ut < best[0]:
                    best = (ut, t)
            line += '   | best %dx%dx%d %.1f us | ' % (best[1] + (best[0],)) + '  '.join(cells)
        if '--blas' in sys.argv:     # yardstick only: the vendor library's plain GEMM (no epilogue) on the same operands
            ys = [[torch.empty(x.shape[0], w.shape[0], device='cuda', dtype=dt) for (x, w, _, _) in ps] for ps, _ in sets]

            def blas(i):
                for (x, w, _, _), y in zip(sets[i % nsets][0], ys[i % nsets]):
                    torch.mm(x, w.t(), out=y)
            us2 = time_graph(blas)
            line += '   | vendor GEMM (no epilogue) %8.1f us %7.1f TFLOP/s' % (us2, fl / us2 / 1e6)
        print(line, flush=True)


if __name__ == '__main__':
    main()
