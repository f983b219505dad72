"""GPU box only: where the c2 step's wall time goes - the panorama graph and the navigation graph replayed alone
(GraphedNavStep(split=True)), next to the whole step.   usage: python tools/dbg/split_times.py [config]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch

import bench
from dsrg_amd.runner import GraphedNavStep


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    dev = torch.device('cuda', 0)
    model, cfg, sd, lbs, pb, nb, dims = bench.build(cfgname, torch.bfloat16, dev)
    with torch.no_grad():
        l = bench.T(lbs[0], dev)
        txt, tm = model('language', l), l['txt_masks']
    whole = GraphedNavStep(model, txt, tm, bench.T(pb, dev), bench.T(nb, dev), reverie=dims[0])
    split = GraphedNavStep(model, txt, tm, bench.T(pb, dev), bench.T(nb, dev), reverie=dims[0], split=True)
    print('%s whole step   %.4f ms' % (cfgname, timeit(whole.step)))
    print('%s panorama     %.4f ms' % (cfgname, timeit(split.pano_step)))
    print('%s navigation   %.4f ms' % (cfgname, timeit(split.nav_step)))
    model.overlap_streams = False
    one = GraphedNavStep(model, txt, tm, bench.T(pb, dev), bench.T(nb, dev), reverie=dims[0], split=True)
    print('%s panorama, one stream    %.4f ms' % (cfgname, timeit(one.pano_step)))
    print('%s navigation, one stream (grouped launches)  %.4f ms' % (cfgname, timeit(one.nav_step)))


if __name__ == '__main__':
    main()
