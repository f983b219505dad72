"""gpu: dsrg_attention_bwd_noalloc (the one entry point that allocates nothing: its delta workspace is a static device
array inside the library) against dsrg_attention_bwd on the same operands, through the C-ABI.  Same kernels -> the bar is
bit-exact.  Two calls in a row, the second with a smaller input (the shared array keeps the first call's rows), all mask
modes incl. the sprel bias; sizes beyond the static maximum are rejected with a clear error."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, nh, Nq, Nk, dtype, mode, sprel, seed):
    from dsrg_amd import kernels as K
    g = torch.Generator(device='cpu').manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).cuda().to(dtype)
    q, k, v, do = r(B, Nq, nh * 64), r(B, Nk, nh * 64), r(B, Nk, nh * 64), r(B, Nq, nh * 64)
    kmask = None
    if mode != 0:
        kmask = (torch.rand(B, Nk, generator=g) > 0.3).cuda()
        kmask[:, 0] = True
    dists = sw = sb = None
    if sprel:
        dists = (torch.rand(B, Nq, Nk, generator=g) * 10).cuda()
        sw, sb = torch.tensor([0.3], device='cuda'), torch.tensor([-0.1], device='cuda')
    o, lse = K.attention(q, k, v, nh, kmask, mode, dists, sw, sb, want_lse=True)
    want = K.attention_bwd(q, k, v, o, do, lse, nh, kmask, mode, dists, sw, sb)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    dsp = torch.zeros(2, device='cuda') if sprel else None
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    K.attention_bwd_noalloc(q, k, v, o, do, lse, nh, kmask, mode, dists, sw, sb, dq, dk, dv, dsp)
    assert torch.cuda.memory_allocated() == before            # nothing allocated on torch's side either
    torch.cuda.synchronize()
    for a, b, n in zip((dq, dk, dv), want[:3], 'qkv'):
        assert torch.equal(a, b), 'd%s differs (B=%d Nq=%d Nk=%d mode=%d)' % (n, B, Nq, Nk, mode)
    if sprel:      # two float atomics per workgroup: order-dependent in the last bits
        assert torch.allclose(dsp, want[3], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_noalloc_matches_attention_bwd_two_calls_second_smaller(dtype):
    _case(24, 12, 38, 38, dtype, 1, False, 1)      # local-branch self-attention of a pre-training step
    _case(5, 12, 20, 20, dtype, 1, True, 2)        # smaller: the array still holds rows of the first call beyond these
    _case(3, 2, 108, 108, dtype, 3, False, 3)      # object encoder ("+1" mask), 2 heads
    _case(2, 12, 7, 80, dtype, 2, False, 4)        # cross-shaped, -inf padding


def test_noalloc_rejects_sizes_beyond_the_static_workspace():
    from dsrg_amd import kernels as K, _lib
    B, nh, Nq = 1, 12, (1 << 20) // 12 + 1
    q = torch.zeros(B, Nq, nh * 64, device='cuda', dtype=torch.bfloat16)
    k = torch.zeros(B, 8, nh * 64, device='cuda', dtype=torch.bfloat16)
    lse = torch.zeros(B, nh, Nq, device='cuda')
    with pytest.raises(_lib.DsrgHipError, match='exceeds the static workspace'):
        K.attention_bwd_noalloc(q, k, k, q, q, lse, nh, None, 0, None, None, None, torch.empty_like(q), torch.empty_like(k), torch.empty_like(k))


def test_noalloc_under_graph_capture_after_init():
    """The entry point is for callers that cannot allocate - HIP-graph capture among them: after dsrg_init() on the device
    (the symbol lookup must not run inside a capture) a captured call replays to the bits of dsrg_attention_bwd, also on new
    operand VALUES in the same buffers."""
    from dsrg_amd import kernels as K
    K.init()
    B, nh, Nq, Nk, dtype = 6, 12, 38, 80, torch.bfloat16
    g = torch.Generator(device='cpu').manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g).cuda().to(dtype)
    q, k, v, do = r(B, Nq, nh * 64), r(B, Nk, nh * 64), r(B, Nk, nh * 64), r(B, Nq, nh * 64)
    kmask = (torch.rand(B, Nk, generator=g) > 0.3).cuda()
    kmask[:, 0] = True
    o, lse = K.attention(q, k, v, nh, kmask, 1, want_lse=True)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        K.attention_bwd_noalloc(q, k, v, o, do, lse, nh, kmask, 1, None, None, None, dq, dk, dv)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        K.attention_bwd_noalloc(q, k, v, o, do, lse, nh, kmask, 1, None, None, None, dq, dk, dv)
    for rep in range(2):
        if rep == 1:                                   # new values in the captured buffers
            do.copy_(r(B, Nq, nh * 64))
        for t in (dq, dk, dv):
            t.zero_()
        gr.replay()
        torch.cuda.synchronize()
        want = K.attention_bwd(q, k, v, o, do, lse, nh, kmask, 1, None, None, None)
        for a, b in zip((dq, dk, dv), want[:3]):
            assert torch.equal(a, b)
