#!/usr/bin/env python3
"""Per-workgroup phase timeline of attention_mfma_kernel from in-kernel stamps (diagnostic build only):
   tools/build_variant.sh astamps -DDSRG_ATTN_STAMPS
   DSRG_HIP_LIB=dsrg_amd/csrc/variants/libdsrg_hip_astamps.so python3 tools/attn_stamps.py
Stamps: 0 entry, 1 K/V staged (after the barrier), 2 compute done, 3 stores issued (shader cycles)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsrg_amd import kernels as K, _lib
from dsrg_amd._lib import MASK_NEG10000

def main():
    lib = _lib.load()
    fn = lib.dsrg_debug_attn_stamps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
    B, nh, H = 64, 12, 768
    dt = torch.bfloat16
    for name, Nq, Nk, stride in (('cross g (Nq=20, Nk=80, K/V in the 12288-wide panel)', 20, 80, 12288),
                                 ('cross l (Nq=38, Nk=80)', 38, 80, 12288),
                                 ('self  l (Nq=Nk=38, packed QKV)', 38, 38, 2304),
                                 ('self  g (Nq=Nk=20, packed QKV)', 20, 20, 2304)):
        q = torch.randn(B, Nq, H, device='cuda').to(dt)
        kv = torch.randn(B, Nk, stride, device='cuda').to(dt)
        k, v = kv[..., :H], kv[..., H:2 * H]
        mask = torch.ones(B, Nk, device='cuda', dtype=torch.bool)
        for _ in range(3):
            o = K.attention(q, k, v, nh, mask, MASK_NEG10000)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); K.attention(q, k, v, nh, mask, MASK_NEG10000); e1.record(); e1.synchronize()
        nwg = ((Nq + 63) // 64) * nh * B
        buf = np.zeros(16384 * 4, np.uint64)
        assert fn(buf.ctypes.data, buf.nbytes) == 0
        st = buf.reshape(16384, 4)[:nwg].astype(np.int64)
        ph = [st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 3] - st[:, 0]]
        q3 = lambda a: '%6d %6d %6d' % tuple(np.percentile(a, [10, 50, 90]))
        print('%-52s wgs %4d event %.1f us | stage %s | compute %s | store %s | total %s' % (name, nwg, e0.elapsed_time(e1) * 1e3, q3(ph[0]), q3(ph[1]), q3(ph[2]), q3(ph[3])))

if __name__ == '__main__':
    main()
