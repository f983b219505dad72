#!/usr/bin/env python3
"""Micro-benchmark of dsrg_gemm_nt on the nav-step's GEMM shapes (GPU box only).
usage: python tools/bench_gemm.py            (env DSRG_GEMM_TILE=128x128|64x128|64x64 forces a tile)"""
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

def main():
    dt = torch.bfloat16
    for name, probs, act, of32, res in SHAPES:
        ps = []
        for (M, N, Kd) in probs:
            x = (torch.rand(M, Kd, device='cuda') * 2 - 1).to(dt)
            w = ((torch.rand(N, Kd, device='cuda') * 2 - 1) * 0.05).to(dt)
            b = torch.rand(N, device='cuda')
            r = (torch.rand(M, N, device='cuda') * 2 - 1).to(dt) if res else None
            ps.append((x, w, b, r))
        outs = K.gemm(ps, act, of32)
        for _ in range(3):
            K.gemm(ps, act, of32, outs=outs)
        torch.cuda.synchronize()
        # HIP-graph replay of `inner` back-to-back launches: no Python/ctypes time in the measurement
        inner, rep = 20, 5
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            K.gemm(ps, act, of32, outs=outs)
        torch.cuda.current_stream().wait_stream(st)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(inner):
                K.gemm(ps, act, of32, outs=outs)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rep):
            g.replay()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (rep * inner)
        fl = sum(2.0 * M * N * Kd for (M, N, Kd) in probs)
        line = '%-44s %8.1f us  %7.1f TFLOP/s' % (name, us, fl / us / 1e6)
        if '--blas' in sys.argv:     # yardstick only: the vendor library's plain GEMM (no epilogue) on the same operands
            ys = [torch.empty(x.shape[0], w.shape[0], device='cuda', dtype=dt) for (x, w, _, _) in ps]
            def blas():
                for (x, w, _, _), y in zip(ps, ys):
                    torch.mm(x, w.t(), out=y)
            for _ in range(3):
                blas()
            torch.cuda.synchronize()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                for _ in range(inner):
                    blas()
            g2.replay(); torch.cuda.synchronize()
            e0.record()
            for _ in range(rep):
                g2.replay()
            e1.record(); e1.synchronize()
            us2 = e0.elapsed_time(e1) * 1e3 / (rep * inner)
            line += '   | vendor GEMM (no epilogue) %8.1f us %7.1f TFLOP/s' % (us2, fl / us2 / 1e6)
        print(line, flush=True)

if __name__ == '__main__':
    main()
