"""not-gpu: the C-ABI library builds/loads on a CPU-only host and exports every symbol that
include/dsrg_hip.h declares; the ctypes binding covers exactly that set; calls without a GPU fail loudly."""
import os
import re
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'dsrg_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dsrg_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported_and_bound():
    from dsrg_amd import _lib
    names = _declared()
    assert len(names) >= 11
    lib = _lib.load()
    assert sorted(_lib.SIGNATURES) == names
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.dsrg_abi_version() == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    """host-side checks run before any launch, so they can be exercised on a CPU-only box."""
    import ctypes as C
    from dsrg_amd import _lib
    lib = _lib.load()
    arr = (_lib.GemmProblem * 1)()
    arr[0] = _lib.GemmProblem(16, 16, None, None, 16, 4, 8, 100, 100, 100, 0, 8)   # K=100 not a multiple of 32
    assert lib.dsrg_gemm_nt(arr, 1, _lib.F32, _lib.F32, 0, None) == -1
    assert b'multiple' in lib.dsrg_last_error()
    assert lib.dsrg_gemm_nt(arr, 99, _lib.F32, _lib.F32, 0, None) == -1
    assert lib.dsrg_layernorm_fwd(None, 0, None, 0, None, None, None, 0, 1, 768, 1e-12, 0, 0, None) == -1


def test_gemm_plan_and_tile_validation_without_gpu():
    """dsrg_gemm_nt_plan is a host-only query and dsrg_gemm_nt_tiled validates its tile before any launch: the nav-step
    N = 768 shapes get the 64x64 tile with a 3-deep ring, the wide ones (packed QKV, FFN-in) 128x128 with a 2-deep ring (bf16;
    fp32: 64x64, 2-deep above 1024 tiles), FFN-out (K >= 2048) 128x64 with a 4-deep ring, the once-per-episode text K/V panel
    the 8-wave 256x128 tile (bf16 only), a LayerNorm-fusion problem the 64x64 tile or its kind's wide tile (a_stats: 128x128x2, r_stats / stats_out: 128x64x4; bf16), and a tile that is not built is
    EINVAL."""
    import ctypes as C
    from dsrg_amd import _lib
    lib = _lib.load()

    def plan(M, N, K, dt=_lib.BF16, **kw):
        arr = (_lib.GemmProblem * 1)()
        arr[0] = _lib.GemmProblem(16, 16, None, None, 16, M, N, K, K, K, 0, N)
        for k, v in kw.items():
            setattr(arr[0], k, v)
        bm, bn, nt, nst = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert lib.dsrg_gemm_nt_plan(arr, 1, dt, C.byref(bm), C.byref(bn), C.byref(nt), C.byref(nst)) == 0
        return bm.value, bn.value, nst.value, nt.value
    assert plan(2432, 768, 768) == (64, 64, 3, 38 * 12)
    assert plan(2432, 2304, 768) == (128, 128, 2, 19 * 18)
    assert plan(2432, 2304, 768, _lib.F32) == (64, 64, 2, 38 * 36)
    assert plan(2432, 768, 3072) == (128, 64, 4, 19 * 12)
    assert plan(2432, 2304, 768, a_stats=16)[:3] == (128, 128, 2)       # LayerNorm-fusion consumers keep the wide tile (bf16)
    assert plan(2432, 768, 768, a_stats=16)[:3] == (64, 64, 3)
    assert plan(2432, 2304, 768, _lib.F32, a_stats=16)[:2] == (64, 64)
    assert plan(2432, 768, 3072, stats_out=16, r_stats=16)[:3] == (128, 64, 4)     # ... and the producers theirs
    assert plan(2432, 768, 768, stats_out=16)[:3] == (64, 64, 3)
    assert plan(5120, 12288, 768) == (256, 128, 3, 20 * 96)
    assert plan(5120, 12288, 768, _lib.F32)[:3] == (128, 128, 2)
    assert plan(5120, 12288, 768, stats_out=16)[:2] == (64, 64)
    arr = (_lib.GemmProblem * 1)()
    arr[0] = _lib.GemmProblem(16, 16, None, None, 16, 64, 64, 64, 64, 64, 0, 64)
    assert lib.dsrg_gemm_nt_tiled(arr, 1, _lib.BF16, _lib.BF16, 0, 96, 96, 0, None) == -1
    assert b'no 96x96 tile' in lib.dsrg_last_error()
    assert lib.dsrg_gemm_nt_tiled(arr, 1, _lib.F32, _lib.F32, 0, 256, 128, 0, None) == -1      # wide tiles are bf16 only
    assert lib.dsrg_gemm_nt_tiled(arr, 1, _lib.BF16, _lib.BF16, 0, 64, 64, 7, None) == -1
    assert b'ring depth' in lib.dsrg_last_error()
    # LayerNorm-fusion arguments are checked on the host too
    arr[0].a_stats = 16
    assert lib.dsrg_gemm_nt(arr, 1, _lib.BF16, _lib.BF16, 0, None) == -1 and b'a_stats' in lib.dsrg_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only check')
def test_no_cpu_fallback():
    from dsrg_amd import kernels
    from dsrg_amd._lib import DsrgHipError
    x, w = torch.zeros(4, 64), torch.zeros(8, 64)
    with pytest.raises(DsrgHipError):
        kernels.gemm([(x, w, None, None)])
