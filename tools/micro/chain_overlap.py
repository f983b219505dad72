#!/usr/bin/env python3
"""Do independent chains of x-layer-shaped GEMMs overlap when they are parallel branches of one HIP graph?
Each chain = 4 layers x (cq, co, qkv, so, ff1, ff2) for M rows; configs run the same total rows as 1, 2, 3 or 4 chains."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dsrg_amd import kernels as K

LAYER = [(768, 768, 0, False, False), (768, 768, 0, True, True), (2304, 768, 0, False, False), (768, 768, 0, True, True),
         (3072, 768, 1, False, False), (768, 3072, 0, True, True)]
dt = torch.bfloat16


def make_chain(M, nl=4):
    ops = []
    for _ in range(nl):
        for (N, Kd, act, f32, res) in LAYER:
            x = (torch.rand(M, Kd, device='cuda') * 2 - 1).to(dt)
            w = ((torch.rand(N, Kd, device='cuda') * 2 - 1) * 0.05).to(dt)
            b = torch.rand(N, device='cuda')
            r = (torch.rand(M, N, device='cuda') * 2 - 1).to(dt) if res else None
            ps = [(x, w, b, r)]
            ops.append((ps, act, f32, K.gemm(ps, act, f32)))
    return ops


def run_chain(ops):
    for ps, act, f32, outs in ops:
        K.gemm(ps, act, f32, outs=outs)


def time_cfg(name, Ms, parallel):
    chains = [make_chain(M) for M in Ms]
    streams = [torch.cuda.Stream() for _ in chains]
    main = torch.cuda.Stream()

    def body():
        cur = torch.cuda.current_stream()
        if not parallel:
            for c in chains:
                run_chain(c)
            return
        for s, c in zip(streams, chains):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                run_chain(c)
        for s in streams:
            cur.wait_stream(s)
    with torch.cuda.stream(main):
        body(); main.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            body()
        g.replay(); main.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(20):
            g.replay()
        e1.record(main); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('%-52s %8.1f us' % (name, us), flush=True)


time_cfg('global 1280 then local 2432, one stream', [1280, 2432], False)
time_cfg('global 1280 | local 2432, two branches', [1280, 2432], True)
time_cfg('1280 | 1216 | 1216, three branches', [1280, 1216, 1216], True)
time_cfg('640 | 640 | 1216 | 1216, four branches', [640, 640, 1216, 1216], True)
time_cfg('928 x 4, four branches', [928] * 4, True)
time_cfg('464 x 8, eight branches', [464] * 8, True)
time_cfg('3712 one chain', [3712], False)
time_cfg('local 2432 alone', [2432], False)
time_cfg('global 1280 alone', [1280], False)
time_cfg('1216 alone', [1216], False)
