#!/usr/bin/env python3
"""Per-workgroup phase timeline of dsrg_gemm_nt from in-kernel s_memtime stamps (diagnostic build only):
   tools/build_variant.sh stamps -DDSRG_GEMM_STAMPS
   DSRG_HIP_LIB=dsrg_amd/csrc/variants/libdsrg_hip_stamps.so python3 tools/gemm_stamps.py
Stamps: 0 kernel entry, 1 first stage landed, 2 K loop done, 3 epilogue done (shader clock, 100 MHz s_memtime ticks
are NOT used: s_memtime counts shader cycles)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsrg_amd import kernels as K, _lib

SHAPES = [
    ('x.cq g+l N=768 K=768 f32+res', [(1280, 768, 768), (2432, 768, 768)], 0, True, True),
    ('x.qkv g+l N=2304 K=768', [(1280, 2304, 768), (2432, 2304, 768)], 0, False, False),
    ('x.ff1 g+l N=3072 K=768 gelu', [(1280, 3072, 768), (2432, 3072, 768)], 1, False, False),
    ('x.ff2 g+l N=768 K=3072 f32+res', [(1280, 768, 3072), (2432, 768, 3072)], 0, True, True),
    ('pano.ff1 M=2304 N=3072 K=768 gelu', [(2304, 3072, 768)], 1, False, False),
    ('g.ff1 M=1280 N=3072 K=768 gelu', [(1280, 3072, 768)], 1, False, False),
    ('g.ff1 M=1280 N=3072 K=768 no act', [(1280, 3072, 768)], 0, False, False),
    ('l.ff1 M=2432 N=3072 K=768 gelu', [(2432, 3072, 768)], 1, False, False),
    ('l.qkv M=2432 N=2304 K=768', [(2432, 2304, 768)], 0, False, False),
]
# --tiles: every shape is also run with these pinned tiles (dsrg_gemm_nt_tiled)
TILES = [None, (64, 64, 3), (128, 128, 2), (128, 128, 3), (256, 128, 2), (256, 128, 3)] if '--tiles' in sys.argv else [None]


def fuse_main(lib, fn):
    """--fuse: the LayerNorm-fusion (FUSE) instantiations next to the plain kernels they replace, local-branch shapes
    (2432 rows): producer = dense + residual writing bf16 y + partial statistics (r_stats + stats_out) vs the plain fp32-out
    launch; consumer = projection of LN(y) (a_stats) on the 64x64 FUSE tile vs the plain launch on its planned tile."""
    dt = torch.bfloat16
    M, H = 2432, 768
    dev = 'cuda'
    rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1)
    cases = []
    for name, N, Kd, act in (('co/so N=768 K=768', 768, 768, 0), ('ff2 N=768 K=3072', 768, 3072, 0)):
        x, w, b, r = rnd(M, Kd).to(dt), (rnd(N, Kd) * 0.05).to(dt), rnd(N), rnd(M, N).to(dt)
        st_in, st_out = torch.rand(N // 64, M, 2, device=dev), torch.empty(N // 64, M, 2, device=dev)
        g, be = torch.rand(N, device=dev) + 0.5, rnd(N)
        cases.append((name + ' plain f32+res', [(x, w, b, r)], act, True, None, None))
        cases.append((name + ' plain bf16+res', [(x, w, b, r)], act, False, None, None))
        cases.append((name + ' FUSE stats_out', [(x, w, b, r)], act, False, [{'stats_out': st_out, 'eps': 1e-12}], None))
        cases.append((name + ' FUSE r_stats+stats_out', [(x, w, b, r)], act, False, [{'r_stats': (st_in, g, be), 'stats_out': st_out, 'eps': 1e-12}], None))
    for name, N, Kd, act in (('cq N=768 K=768', 768, 768, 0), ('qkv N=2304 K=768', 2304, 768, 0), ('ff1 N=3072 K=768 gelu', 3072, 768, 1)):
        x, w, b = rnd(M, Kd).to(dt), (rnd(N, Kd) * 0.05).to(dt), rnd(N)
        st_in, c1 = torch.rand(Kd // 64, M, 2, device=dev), rnd(N)
        cases.append((name + ' plain (planned tile)', [(x, w, b, None)], act, False, None, None))
        cases.append((name + ' plain 64x64x3', [(x, w, b, None)], act, False, None, (64, 64, 3)))
        cases.append((name + ' FUSE a_stats', [(x, w, b, None)], act, False, [{'a_stats': (st_in, c1), 'eps': 1e-12}], None))
    for name, ps, act, of32, fz, tile in cases:
        outs = K.gemm(ps, act, of32, fuse=fz, tile=tile)
        for _ in range(5):
            K.gemm(ps, act, of32, outs=outs, fuse=fz, tile=tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        assert lib.dsrg_debug_gemm_stamps_clear() == 0
        torch.cuda.synchronize()
        e0.record(); K.gemm(ps, act, of32, outs=outs, fuse=fz, tile=tile); e1.record(); e1.synchronize()
        report(name, fn, e0.elapsed_time(e1))


def report(name, fn, ms):
    buf = np.zeros(8192 * 8, np.uint64)
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(8192, 8)
    n = int((st[:, 3] > 0).sum())
    st = st[:n].astype(np.int64)
    t00 = st[:, 0].min()
    span = st[:, 3].max() - t00
    ph = [st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]]
    q = lambda a: '%6d %6d %6d' % tuple(np.percentile(a, [10, 50, 90]))
    print('%-44s wgs %5d  event %.1f us  span %7d clk' % (name, n, ms * 1e3, span))
    print('    prologue %s | K loop %s | epilogue %s | total %s  (clk, 10/50/90%%)' % (q(ph[0]), q(ph[1]), q(ph[2]), q(st[:, 3] - st[:, 0])))
    print('    epilogue split: wait-for-waves %s | image write + barrier %s | rows out %s' % (q(st[:, 5] - st[:, 2]), q(st[:, 6] - st[:, 5]), q(st[:, 3] - st[:, 6])))


def main():
    lib = _lib.load()
    fn = lib.dsrg_debug_gemm_stamps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
    if '--fuse' in sys.argv:
        return fuse_main(lib, fn)
    dt = torch.bfloat16
    for name in [(n_ + (' tile %dx%dx%d' % t_ if t_ else ''), p_, a_, o_, r_, t_) for (n_, p_, a_, o_, r_) in SHAPES for t_ in TILES]:
        name, probs, act, of32, res, tile = name
        ps = []
        for (M, N, Kd) in probs:
            x = (torch.rand(M, Kd, device='cuda') * 2 - 1).to(dt)
            w = ((torch.rand(N, Kd, device='cuda') * 2 - 1) * 0.05).to(dt)
            b = torch.rand(N, device='cuda')
            r = (torch.rand(M, N, device='cuda') * 2 - 1).to(dt) if res else None
            ps.append((x, w, b, r))
        outs = K.gemm(ps, act, of32)
        for _ in range(5):
            K.gemm(ps, act, of32, outs=outs, tile=tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        assert lib.dsrg_debug_gemm_stamps_clear() == 0
        torch.cuda.synchronize()
        e0.record(); K.gemm(ps, act, of32, outs=outs, tile=tile); e1.record(); e1.synchronize()
        plan = K.gemm_plan(ps) if hasattr(K, 'gemm_plan') else None
        buf = np.zeros(8192 * 8, np.uint64)
        assert fn(buf.ctypes.data, buf.nbytes) == 0
        st = buf.reshape(8192, 8)
        n = int((st[:, 3] > 0).sum())
        st = st[:n].astype(np.int64)
        t00 = st[:, 0].min()
        span = st[:, 3].max() - t00
        ph = [st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]]
        q = lambda a: '%6d %6d %6d' % tuple(np.percentile(a, [10, 50, 90]))
        print('%-36s wgs %5d  event %.1f us  span %7d clk' % (name, n, e0.elapsed_time(e1) * 1e3, span))
        print('    start-after-first [10/50/90%%]: %s   end-before-last: %s' % (q(st[:, 0] - t00), q(st[:, 3].max() - st[:, 3])))
        print('    prologue %s | K loop %s | epilogue %s | total %s  (clk, 10/50/90%%)' % (q(ph[0]), q(ph[1]), q(ph[2]), q(st[:, 3] - st[:, 0])))
        print('    epilogue split: wait-for-waves %s | image write + barrier %s | rows out %s' % (q(st[:, 5] - st[:, 2]), q(st[:, 6] - st[:, 5]), q(st[:, 3] - st[:, 6])))
        late = st[:, 0] - t00 > 0.3 * span
        print('    workgroups starting after 30%% of the span: %d (%.0f%%)' % (late.sum(), 100.0 * late.mean()))


if __name__ == '__main__':
    main()
