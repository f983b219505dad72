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


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only check')
def test_no_cpu_fallback():
    from dsrg_amd import kernels
    from dsrg_amd._lib import DsrgHipError
    x, w = torch.zeros(4, 64), torch.zeros(8, 64)
    with pytest.raises(DsrgHipError):
        kernels.gemm([(x, w, None, None)])
