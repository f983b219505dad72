"""gpu: each libdsrg_hip.so entry point against a plain torch fp32 restatement of the same op,
called through the C-ABI (dsrg_amd.kernels -> ctypes).  fp32 storage must agree to ~1e-5 relative
(exact-fp32 MFMA, different summation order); bf16 storage to bf16 rounding of the result."""
import math
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from dsrg_amd import kernels
    return kernels


def _r(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device='cpu').manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


ACTS = {0: lambda z: z, 1: torch.nn.functional.gelu, 2: torch.relu, 3: torch.tanh}


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K_,act,res,of32', [
    (2432, 768, 768, 0, True, True), (1280, 2304, 768, 0, False, False), (2304, 3072, 768, 1, False, False),
    (2304, 768, 3072, 0, True, False), (64, 768, 2304, 0, False, True), (6912, 384, 128, 0, False, False),
    (100, 1000, 768, 0, False, True), (37, 30522, 768, 0, False, True), (1, 768, 1536, 2, False, True),
    (513, 130, 896, 3, False, False),
])
def test_gemm(K, dtype, M, N, K_, act, res, of32):
    x, w, b = _r(M, K_, dtype=dtype), _r(N, K_, seed=1, scale=0.05, dtype=dtype), _r(N, seed=2)
    r = _r(M, N, seed=3, dtype=dtype) if res else None
    out = K.gemm([(x, w, b, r)], act, of32)[0]
    ref = ACTS[act](x.float() @ w.float().T + b)
    if res:
        ref = ref + r.float()
    assert out.dtype == (torch.float32 if of32 else dtype)
    tol = 2e-5 * math.sqrt(K_) if dtype == torch.float32 else (1e-3 * math.sqrt(K_) / 16 if of32 else 3e-2)
    err = (out.float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * max(1.0, scale), (err, tol, scale)


TILE_MENU = [(64, 64, 2), (64, 64, 3), (64, 64, 4), (64, 128, 2), (64, 128, 3), (64, 192, 3), (64, 256, 3), (128, 64, 2), (128, 64, 3),
             (128, 64, 4), (128, 96, 3), (128, 128, 2), (128, 128, 3), (128, 192, 2), (128, 192, 3), (128, 256, 3), (256, 128, 2), (256, 128, 3)]


@pytest.mark.parametrize('M,N,K_,act,res,of32', [(513, 330, 896, 1, False, False), (2432, 768, 768, 0, True, True), (300, 1000, 128, 2, True, False),
                                                  (1, 64, 3072, 0, False, False)])
def test_gemm_tile_menu(K, M, N, K_, act, res, of32):
    """every tile / ring depth of the bf16 menu through dsrg_gemm_nt_tiled (the 8-wave 256x128 tile, the two-pass epilogue
    of its 2-deep form, the odd 96- and 192-column tiles): same result as the plan's tile, which test_gemm checks against
    torch; partial tiles on every edge."""
    dtype = torch.bfloat16
    x, w, b = _r(M, K_, dtype=dtype), _r(N, K_, seed=1, scale=0.05, dtype=dtype), _r(N, seed=2)
    r = _r(M, N, seed=3, dtype=dtype) if res else None
    ref = ACTS[act](x.float() @ w.float().T + b) + (r.float() if res else 0)
    tol = (1e-3 * math.sqrt(K_) / 16 if of32 else 3e-2) * max(1.0, ref.abs().max().item())
    base = K.gemm([(x, w, b, r)], act, of32)[0]
    assert (base.float() - ref).abs().max().item() <= tol
    for t in TILE_MENU:
        out = K.gemm([(x, w, b, r)], act, of32, tile=t)[0]
        assert out.dtype == base.dtype
        # same K order and fp32 accumulation in every tile: equal up to the GELU / rounding of identical sums
        assert (out.float() - base.float()).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item()) + (0 if of32 else 1e-2 * tol), t
    # grouped: two problems of different shape in one launch of the 8-wave tile
    x2, w2 = _r(77, K_, seed=5, dtype=dtype), _r(200, K_, seed=6, scale=0.05, dtype=dtype)
    o = K.gemm([(x, w, b, r), (x2, w2, None, None)], act, of32, tile=(256, 128, 3))
    assert (o[0].float() - ref).abs().max().item() <= tol
    assert (o[1].float() - ACTS[act](x2.float() @ w2.float().T)).abs().max().item() <= tol


@pytest.mark.parametrize('M,N,K_', [(2432, 768, 3072), (64, 64, 64), (1288, 3072, 768), (8, 24, 128), (776, 30528, 768), (200, 136, 4096)])
@pytest.mark.parametrize('of32', [False, True])
def test_gemm_transposed_operands(K, M, N, K_, of32):
    """dsrg_gemm_problem_t::trans - the backward GEMMs of a linear layer on operands as they lie: A stored [K, M] and / or W
    stored [K, N] (strided views included).  Bit-exact against the NT kernel on explicitly transposed copies (same K order,
    same MFMA), and within bf16 GEMM tolerance of torch."""
    dtype = torch.bfloat16
    a = _r(M, K_, dtype=dtype)
    w = _r(N, K_, seed=1, scale=0.05, dtype=dtype)
    b = _r(N, seed=2)
    ref = a.float() @ w.float().T + b
    base = K.gemm([(a, w, b, None)], 0, of32)[0]
    tol = (1e-3 * math.sqrt(K_) / 16 if of32 else 3e-2) * max(1.0, ref.abs().max().item())
    assert (base.float() - ref).abs().max().item() <= tol
    at_buf = torch.zeros(K_, M + 16, device=a.device, dtype=dtype)       # [K, M] inside a wider buffer: lda > M
    at_buf[:, :M] = a.T
    at, wt = at_buf[:, :M], w.T.contiguous()
    for trans, xa, xw in ((1, at, w), (2, a, wt), (3, at, wt)):
        out = K.gemm([(xa, xw, b, None)], 0, of32, trans=trans)[0]
        assert out.shape == base.shape and out.dtype == base.dtype
        assert torch.equal(out, base), (trans, (out.float() - base.float()).abs().max().item())
    # round 3: the wide tiles of the dgrad (trans = 2) and wgrad (trans = 3) forms, W / A staged as 64-column blocks
    for trans, xa, xw, tile in ((2, a, wt, (128, 128, 2)), (2, a, wt, (128, 64, 4)), (3, at, wt, (128, 128, 2)), (3, at, wt, (128, 64, 3))):
        out = K.gemm([(xa, xw, b, None)], 0, of32, trans=trans, tile=tile)[0]
        assert torch.equal(out, base), (trans, tile, (out.float() - base.float()).abs().max().item())
    # grouped: two problems of different shape in one launch
    a2, w2 = _r(72, K_, seed=5, dtype=dtype), _r(200, K_, seed=6, scale=0.05, dtype=dtype)
    o = K.gemm([(at, wt, b, None), (a2.T.contiguous(), w2.T.contiguous(), None, None)], 0, of32, trans=3)
    assert torch.equal(o[0], base)
    assert torch.equal(o[1], K.gemm([(a2, w2, None, None)], 0, of32)[0])


@pytest.mark.parametrize('M,N,K_', [(768, 768, 888), (3072, 768, 4932), (128, 384, 14796), (1000, 768, 1), (768, 1536, 24), (8, 8, 63), (200, 72, 130)])
def test_gemm_wgrad_any_rows_and_column_sums(K, M, N, K_):
    """trans = 3 (dW = dY^T.X) with a row count that is no multiple of 64 (the partial last K-step is zero-filled on chip)
    and the column sums of dY (the bias gradient) from the same launch; bit-exact against the NT kernel on zero-padded
    transposed copies, column sums against torch in fp32."""
    dtype = torch.bfloat16
    dy, x = _r(K_, M, dtype=dtype), _r(K_, N, seed=1, dtype=dtype)
    Kp = (K_ + 63) // 64 * 64
    dyt = torch.zeros(M, Kp, device=dy.device, dtype=dtype); dyt[:, :K_] = dy.T
    xt = torch.zeros(N, Kp, device=dy.device, dtype=dtype); xt[:, :K_] = x.T
    base = K.gemm([(dyt, xt, None, None)])[0]
    db = torch.full((M,), 7.0, device=dy.device)          # plain stores: whatever was there is overwritten
    out = K.gemm([(dy, x, None, None)], trans=3, colsums=[db])[0]
    assert torch.equal(out, base), (out.float() - base.float()).abs().max().item()
    ref = dy.float().sum(0)
    assert (db - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()) * math.sqrt(K_)
    # poisoned memory behind the operands must not leak in: the rows past K_ are never used
    big = torch.full((K_ + 64, M), float('nan'), device=dy.device, dtype=dtype); big[:K_] = dy
    bigx = torch.full((K_ + 64, N), float('nan'), device=dy.device, dtype=dtype); bigx[:K_] = x
    out2 = K.gemm([(big[:K_], bigx[:K_], None, None)], trans=3)[0]
    assert torch.equal(out2, base)
    # round 3: the wide wgrad tiles (a transposed operand staged as 64-column blocks) - same K order, same MFMA: bit-exact too
    for tile in ((128, 128, 2), (128, 64, 3)):
        db2 = torch.full((M,), 7.0, device=dy.device)
        o = K.gemm([(big[:K_], bigx[:K_], None, None)], trans=3, colsums=[db2], tile=tile)[0]
        assert torch.equal(o, base), (tile, (o.float() - base.float()).abs().max().item())
        assert torch.equal(db2, db), tile


def test_gemm_transposed_operands_rejects_bad_shapes(K):
    from dsrg_amd import _lib
    a, w = _r(64, 20, dtype=torch.bfloat16), _r(64, 64, seed=1, dtype=torch.bfloat16)       # M = 20 is not a multiple of 8
    with pytest.raises(_lib.DsrgHipError):
        K.gemm([(a, w, None, None)], trans=1)
    with pytest.raises(_lib.DsrgHipError):
        K.gemm([(_r(64, 64), _r(64, 64, seed=1), None, None)], trans=3)                       # fp32: no transposed kernels


def _ln_fold(w, b, gamma, beta, dtype):
    """(W * gamma in the storage dtype, c1 = row sums of that, bias + W beta): dsrg_gemm_problem_t, a_stats"""
    wf = (w.float() * gamma[None]).to(dtype)
    return wf, wf.float().sum(1).contiguous(), (b + w.float() @ beta).contiguous()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,H,N2,act', [(2432, 768, 2304, 0), (1280, 768, 3072, 1), (130, 128, 384, 0), (65, 1024, 64, 0)])
def test_gemm_layernorm_fusion(K, dtype, M, H, N2, act):
    """The LayerNorm-fusion epilogue (dsrg_gemm_problem_t a_stats / r_stats / stats_out) against torch: a post-LN block
    y = x W0^T + b0 + r, x1 = LN(y) followed by (a) a projection of x1 and (b) a dense + residual x1, with NO standalone
    LayerNorm: launch 1 stores y and its per-tile (sum, sum of squares), launches 2 and 3 normalise on the fly.  Random
    gamma / beta / bias; a row offset of 3 sigma so that mean != 0 is exercised."""
    eps = 1e-12
    x, w0, b0 = _r(M, H, dtype=dtype), _r(H, H, seed=1, scale=0.05, dtype=dtype), _r(H, seed=2)
    r = (_r(M, H, seed=3).float() + 3.0).to(dtype)
    gamma, beta = (1.0 + 0.3 * _r(H, seed=4)).contiguous(), 0.2 * _r(H, seed=5)
    w1, b1 = _r(N2, H, seed=6, scale=0.05, dtype=dtype), _r(N2, seed=7)
    w2, b2 = _r(H, N2, seed=8, scale=0.05, dtype=dtype), _r(H, seed=9)
    np_ = H // 64
    st = torch.zeros(np_, M, 2, device='cuda')          # block-major partial tables (ABI v12)
    # launch 1: y (storage dtype) + partials
    y = K.gemm([(x, w0, b0, r)], 0, False, fuse=[{'stats_out': st}])[0]
    y_ref = x.float() @ w0.float().T + b0 + r.float()
    tol = 2e-5 * math.sqrt(H) if dtype == torch.float32 else 3e-2
    assert (y.float() - y_ref).abs().max().item() <= tol * max(1.0, y_ref.abs().max().item())
    s_ref = torch.stack([y_ref.view(M, np_, 64).sum(-1), (y_ref.view(M, np_, 64) ** 2).sum(-1)], -1).transpose(0, 1)
    assert (st - s_ref).abs().max().item() <= (1e-3 if dtype == torch.float32 else 0.3) * max(1.0, s_ref.abs().max().item()) * 1e-2
    # the reference of what follows normalises the y that was STORED (bf16: its rounding is part of the design), with the
    # statistics of the fp32 values
    mu, var = y_ref.mean(-1, keepdim=True), y_ref.var(-1, unbiased=False, keepdim=True)
    x1 = (y.float() - mu) * torch.rsqrt(var + eps) * gamma + beta
    # launch 2: act(LN(y) W1^T + b1) from the un-normalised y
    w1f, c1, b1f = _ln_fold(w1, b1, gamma, beta, dtype)
    h = K.gemm([(y, w1f, b1f, None)], act, False, fuse=[{'a_stats': (st, c1), 'eps': eps}])[0]
    h_ref = ACTS[act](x1 @ w1.float().T + b1)
    tol2 = 3e-4 if dtype == torch.float32 else 6e-2
    assert (h.float() - h_ref).abs().max().item() <= tol2 * max(1.0, h_ref.abs().max().item()), (h.float() - h_ref).abs().max().item()
    # launch 3: h W2^T + b2 + LN(y) as the residual, fp32 out, and partials of ITS output
    st3 = torch.zeros(np_, M, 2, device='cuda')
    z = K.gemm([(h, w2, b2, y)], 0, True, fuse=[{'r_stats': (st, gamma, beta), 'stats_out': st3, 'eps': eps}])[0]
    z_ref = h.float() @ w2.float().T + b2 + x1
    tol3 = 2e-5 * math.sqrt(N2) if dtype == torch.float32 else 1e-3 * math.sqrt(N2) / 16
    assert (z - z_ref).abs().max().item() <= tol3 * max(1.0, z_ref.abs().max().item()), (z - z_ref).abs().max().item()
    s3 = torch.stack([z.view(M, np_, 64).sum(-1), (z.view(M, np_, 64) ** 2).sum(-1)], -1).transpose(0, 1)
    assert (st3 - s3).abs().max().item() <= 1e-4 * max(1.0, s3.abs().max().item())
    # and both fused features in one grouped launch next to a plain problem
    o = K.gemm([(y, w1f, b1f, None), (x, w0, b0, None)], act, False, fuse=[{'a_stats': (st, c1), 'eps': eps}, None])
    assert torch.equal(o[0], h) and (o[1].float() - ACTS[act](x.float() @ w0.float().T + b0)).abs().max().item() <= max(tol, 3e-2) * 10


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gemm_grouped_strided(K, dtype):
    """two branches in one launch; strided A (row 0 of each sequence) and strided C (concat buffer)."""
    B, G, Vp, H = 16, 20, 38, 768
    g, v = _r(B, G, H, dtype=dtype), _r(B, Vp, H, seed=1, dtype=dtype)
    w1, w2 = _r(H, H, seed=2, scale=0.05, dtype=dtype), _r(H, H, seed=3, scale=0.05, dtype=dtype)
    b1, b2 = _r(H, seed=4), _r(H, seed=5)
    o = K.gemm([(g, w1, b1, None), (v, w2, b2, None)])
    for out, x, w, b in ((o[0], g, w1, b1), (o[1], v, w2, b2)):
        ref = x.float() @ w.float().T + b
        assert (out.float() - ref).abs().max().item() <= (1e-3 if dtype == torch.float32 else 8e-2)
    cat = torch.zeros(B, 2 * H, device='cuda', dtype=dtype)
    K.gemm([(g[:, 0], w1, b1, None), (v[:, 0], w2, b2, None)], 3, outs=[cat[:, :H], cat[:, H:]])
    ref = torch.cat([torch.tanh(g[:, 0].float() @ w1.float().T + b1), torch.tanh(v[:, 0].float() @ w2.float().T + b2)], 1)
    assert (cat.float() - ref).abs().max().item() <= (1e-4 if dtype == torch.float32 else 2e-2)


def _attn_ref(q, k, v, nh, kmask, mode, dists, sw, sb, scale=0.125):
    B, Nq, _ = q.shape
    sp = lambda t: t.float().reshape(B, t.shape[1], nh, 64).permute(0, 2, 1, 3)
    s = sp(q) @ sp(k).transpose(-1, -2) * scale
    if mode == 1:
        s = s + (1.0 - kmask[:, None, None].float()) * -10000.0
    elif mode == 2:
        s = s.masked_fill(~kmask[:, None, None], float('-inf'))
    elif mode == 3:
        s = s + kmask[:, None, None].float()
    if dists is not None:
        s = s + (dists * sw + sb)[:, None]
    return (torch.softmax(s, -1) @ sp(v)).permute(0, 2, 1, 3).reshape(B, Nq, nh * 64)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,nh,Nq,Nk,mode,sprel', [
    (8, 12, 20, 80, 1, False), (8, 12, 38, 38, 1, False), (8, 12, 20, 20, 1, True), (4, 12, 36, 36, 2, False),
    (4, 2, 108, 108, 3, False), (2, 2, 720, 720, 3, False), (2, 12, 100, 250, 1, False), (3, 12, 1, 5, 0, False),
    # the short-key kernel (Nk <= 128: K fragments straight from global memory, transposing V reads): every tile count,
    # partial last tiles, two query blocks, pair distances with rows that are / are not 16-byte aligned
    (2, 12, 70, 127, 1, True), (2, 12, 5, 33, 2, False), (2, 12, 64, 128, 3, False), (2, 12, 17, 22, 1, True),
    (2, 12, 100, 100, 1, True), (3, 2, 36, 56, 1, False), (2, 12, 130, 96, 0, False),
])
def test_attention(K, dtype, B, nh, Nq, Nk, mode, sprel):
    H = nh * 64
    self_att = Nq == Nk
    if self_att:      # packed QKV, read in place through strides
        qkv = _r(B, Nq, 3 * H, dtype=dtype)
        q, k, v = qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:]
    else:
        q, kv = _r(B, Nq, H, dtype=dtype), _r(B, Nk, 2 * H, seed=1, dtype=dtype)
        k, v = kv[..., :H], kv[..., H:]
    lens = torch.randint(1, Nk + 1, (B,), generator=torch.Generator().manual_seed(5)); lens[0] = Nk
    kmask = (torch.arange(Nk)[None] < lens[:, None]).cuda() if mode else None
    dists = (_r(B, Nq, Nk, seed=7).abs() * 5) if sprel else None
    sw, sb = (torch.tensor([-0.21]).cuda(), torch.tensor([0.07]).cuda()) if sprel else (None, None)
    out = K.attention(q, k, v, nh, kmask, mode, dists, sw, sb)
    ref = _attn_ref(q, k, v, nh, kmask, mode, dists, sw, sb)
    err = (out.float() - ref).abs().max().item()
    assert err <= (2e-5 if dtype == torch.float32 else 2e-2), err


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_attention_grouped(K, dtype):
    """two branches of an x-layer in one launch == two launches (cross-attention on shared text K/V; self-attention
    with the sprel bias on the first branch only)."""
    B, nh, L = 5, 12, 80
    H = nh * 64
    kv = _r(B, L, 4 * H, seed=1, dtype=dtype)
    tmask = (torch.arange(L)[None] < torch.tensor([80, 33, 1, 64, 17])[:, None]).cuda()
    qs = [_r(B, 20, H, seed=2, dtype=dtype), _r(B, 70, H, seed=3, dtype=dtype)]
    probs = [(qs[0], kv[..., :H], kv[..., H:2 * H], tmask, None), (qs[1], kv[..., 2 * H:3 * H], kv[..., 3 * H:], tmask, None)]
    outs = K.attention_grouped(probs, nh, 1)
    for (q, k, v, m, _), o in zip(probs, outs):
        assert torch.equal(o, K.attention(q, k, v, nh, m, 1))
    sw, sb = torch.tensor([-0.21]).cuda(), torch.tensor([0.07]).cuda()
    probs = []
    for j, N in enumerate((20, 38)):
        qkv = _r(B, N, 3 * H, seed=4 + j, dtype=dtype)
        m = (torch.arange(N)[None] < torch.tensor([N, 3, 1, N - 1, 9])[:, None]).cuda()
        probs.append((qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], m, (_r(B, N, N, seed=7).abs() * 5) if j == 0 else None))
    outs = K.attention_grouped(probs, nh, 1, sw, sb)
    for (q, k, v, m, d), o in zip(probs, outs):
        ref = _attn_ref(q, k, v, nh, m, 1, d, sw, sb)
        assert torch.equal(o, K.attention(q, k, v, nh, m, 1, d, sw if d is not None else None, sb if d is not None else None))
        assert (o.float() - ref).abs().max().item() <= (2e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,nh,Nq,Nk,mode,sprel', [
    (4, 12, 20, 80, 1, False), (4, 12, 38, 38, 1, False), (4, 12, 20, 20, 1, True), (3, 12, 36, 36, 2, False),
    (3, 2, 108, 108, 3, False), (2, 12, 100, 250, 1, False), (3, 12, 1, 5, 0, False), (2, 2, 70, 130, 1, True),
    (2, 2, 200, 150, 1, False), (1, 3, 257, 257, 1, True),      # more than one 128-query chunk in the dK / dV kernel
])
def test_attention_bwd(K, dtype, B, nh, Nq, Nk, mode, sprel):
    """dsrg_attention_bwd (recompute from the forward's log-sum-exp) against torch autograd of the fp32 reference."""
    H = nh * 64
    if Nq == Nk:
        qkv = _r(B, Nq, 3 * H, dtype=dtype)
        q, k, v = qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:]
    else:
        q, kv = _r(B, Nq, H, dtype=dtype), _r(B, Nk, 2 * H, seed=1, dtype=dtype)
        k, v = kv[..., :H], kv[..., H:]
    lens = torch.randint(1, Nk + 1, (B,), generator=torch.Generator().manual_seed(5)); lens[0] = Nk
    kmask = (torch.arange(Nk)[None] < lens[:, None]).cuda() if mode else None
    dists = (_r(B, Nq, Nk, seed=7).abs() * 5) if sprel else None
    sw, sb = (torch.tensor([-0.21]).cuda(), torch.tensor([0.07]).cuda()) if sprel else (None, None)
    do = _r(B, Nq, H, seed=11, dtype=dtype)
    out, lse = K.attention(q, k, v, nh, kmask, mode, dists, sw, sb, want_lse=True)
    dq, dk, dv, dsp = K.attention_bwd(q, k, v, out, do, lse, nh, kmask, mode, dists, sw, sb)
    leaves = [t.float().clone().requires_grad_(True) for t in (q, k, v)]
    sp = [t.clone().requires_grad_(True) for t in (sw, sb)] if sprel else [None, None]
    ref = _attn_ref(*leaves, nh, kmask, mode, dists, *sp)
    ref.backward(do.float())
    tol = 2e-4 if dtype == torch.float32 else 4e-2
    for name, got, want in (('dq', dq, leaves[0].grad), ('dk', dk, leaves[1].grad), ('dv', dv, leaves[2].grad)):
        err = (got.float() - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        assert err <= tol, (name, err)
    if sprel:
        want = torch.cat([sp[0].grad, sp[1].grad])
        # d(sprel_b) is analytically 0 (a constant shift of a softmax row); judge both against the size of d(sprel_w)
        assert ((dsp - want).abs().max() / want.abs().max().clamp_min(1.0)).item() <= (1e-3 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('H', [768, 128])
def test_layernorm_and_embed(K, dtype, H):
    M = 777
    x, res = _r(M, H), _r(M, H, seed=1, dtype=dtype)
    g, b = 1 + 0.1 * _r(H, seed=2), 0.1 * _r(H, seed=3)
    out = K.layernorm(x, g, b, 1e-12, res, dtype)
    ref = torch.nn.functional.layer_norm(x + res.float(), (H,), g, b, 1e-12)
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    assert (out.float() - ref).abs().max().item() <= tol
    # token embedding: base + LN(feat linear) + two gathers + vec -> LN
    F_ = 7
    feat8 = _r(M, F_ + 1, seed=4)
    Wf, bf = _r(H, F_, seed=5, scale=0.3), _r(H, seed=6, scale=0.02)
    E1, E2, vec = _r(45, H, seed=7), _r(3, H, seed=8), _r(H, seed=9)
    i1 = torch.randint(0, 45, (M,), dtype=torch.int32).cuda(); i2 = torch.randint(0, 3, (M,), dtype=torch.int32).cuda()
    base = _r(M, H, seed=10, dtype=dtype)
    out = K.embed_ln(M, H, dtype, 'cuda', base=base, feat=feat8[:, :F_], Wf=Wf, bf=bf, lnf=(g, b), E1=E1, idx1=i1, E2=E2,
                     idx2=i2, vec=vec, lno=(b + 1, g - 1), eps=1e-12)
    t = torch.nn.functional.layer_norm(feat8[:, :F_] @ Wf.T + bf, (H,), g, b, 1e-12)
    ref = torch.nn.functional.layer_norm(base.float() + t + E1[i1.long()] + E2[i2.long()] + vec, (H,), b + 1, g - 1, 1e-12)
    assert (out.float() - ref).abs().max().item() <= (2e-5 if dtype == torch.float32 else 3e-2)
    out = K.embed_ln(M, H, dtype, 'cuda', feat=feat8[:, :F_], Wf=Wf, bf=bf, E1=E1, idx1=i1)
    ref = feat8[:, :F_] @ Wf.T + bf + E1[i1.long()]
    assert (out.float() - ref).abs().max().item() <= (2e-5 if dtype == torch.float32 else 3e-2)
    # index operands as the reference holds them: class id in the last (float) feature column, int64 ids, and one
    # id per group of 3 rows (the view's nav type repeated over its objects); 14 features (viewpoint position)
    feat8[:, -1] = i1.float()
    i2g = torch.randint(0, 3, ((M + 2) // 3,)).cuda()
    out = K.embed_ln(M, H, dtype, 'cuda', feat=feat8[:, :F_], Wf=Wf, bf=bf, E1=E1, idx1=feat8[:, -1], E2=E2, idx2=(i2g, 3),
                     lno=(g, b))
    ref = torch.nn.functional.layer_norm(feat8[:, :F_] @ Wf.T + bf + E1[i1.long()] + E2[i2g.repeat_interleave(3)[:M]], (H,), g, b, 1e-12)
    assert (out.float() - ref).abs().max().item() <= (2e-5 if dtype == torch.float32 else 3e-2)
    f14, W14 = _r(M, 14, seed=12), _r(H, 14, seed=13, scale=0.3)
    out = K.embed_ln(M, H, dtype, 'cuda', base=base, feat=f14, Wf=W14, bf=bf, lnf=(g, b), E1=E1, idx1=i1.long())
    ref = base.float() + torch.nn.functional.layer_norm(f14 @ W14.T + bf, (H,), g, b, 1e-12) + E1[i1.long()]
    # no output LayerNorm here: values reach ~8, so the bf16 bound is half an ulp of the largest value
    assert (out.float() - ref).abs().max().item() <= (2e-5 if dtype == torch.float32 else 2.0 ** -8 * ref.abs().max().item())
    # LayerNorm on the base row inside the kernel (dsrg_embed_ln_fwd_lnb): the register-weight kernel (F = 7, 14) and the
    # LDS-staged one (F = 5); equal to a LayerNorm launch in front of the plain call (bit for bit in fp32)
    gb, bb = 1 + 0.1 * _r(H, seed=20), 0.1 * _r(H, seed=21)
    for Fx, fx, Wx in ((7, feat8[:, :7], Wf), (14, f14, W14), (5, feat8[:, :5].contiguous(), _r(H, 5, seed=22, scale=0.3))):
        kw = dict(feat=fx, Wf=Wx, bf=bf, lnf=(g, b), E1=E1, idx1=i1, vec=vec, lno=(b + 1, g - 1))
        got = K.embed_ln(M, H, dtype, 'cuda', base=base, lnb=(gb, bb), **kw)
        two = K.embed_ln(M, H, dtype, 'cuda', base=K.layernorm(base, gb, bb, 1e-12, None, dtype), **kw)
        t = torch.nn.functional.layer_norm(fx @ Wx.T + bf, (H,), g, b, 1e-12)
        ref = torch.nn.functional.layer_norm(torch.nn.functional.layer_norm(base.float(), (H,), gb, bb, 1e-12) + t + E1[i1.long()] + vec,
                                             (H,), b + 1, g - 1, 1e-12)
        assert (got.float() - ref).abs().max().item() <= (2e-5 if dtype == torch.float32 else 3e-2), Fx
        if dtype == torch.float32:
            assert torch.equal(got, two), Fx


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_pool_mean_head_blend(K, dtype):
    B, V, H = 9, 36, 768
    x = _r(B, V, H, dtype=dtype)
    w, b = _r(H, scale=0.05, seed=1), _r(1, seed=2)
    out = K.gaa_pool(x, w, b)
    a = torch.softmax(torch.tanh(x.float() @ w + b), 1)
    ref = (x.float() * a[..., None]).sum(1)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert (out.float() - ref).abs().max().item() <= tol
    # the same with the LayerNorm in front of it inside the kernel (dsrg_gaa_pool_fwd_ln): rows and pooled vector equal to a
    # LayerNorm launch followed by the plain call, bit for bit; rows placed with a batch stride, pooled row scattered
    lg, lb = 1 + 0.1 * _r(H, seed=30), 0.1 * _r(H, seed=31)
    buf = torch.zeros(B, V + 2, H, device='cuda', dtype=dtype)
    gm = torch.zeros(B, 7, H, device='cuda', dtype=dtype)
    idx = torch.arange(B, device='cuda') % 7
    rows2, pooled2 = K.gaa_pool(x, w, b, scatter=(gm, idx), ln=(lg, lb, 1e-12), rows_out=buf[:, 2:])
    xn = K.layernorm(x, lg, lb, 1e-12, None, dtype)
    assert torch.equal(rows2, xn) and torch.equal(buf[:, 2:], xn) and not buf[:, :2].any()
    assert torch.equal(pooled2, K.gaa_pool(xn, w, b)) and torch.equal(gm[torch.arange(B), idx], pooled2)
    rows = _r(B * V * 3, 128, dtype=dtype)
    out = K.group_mean(rows, 3)
    assert (out.float() - rows.float().reshape(-1, 3, 128).mean(1)).abs().max().item() <= tol
    h = _r(B * V, H).relu()
    g, be, w2, b2 = 1 + 0.1 * _r(H, seed=3), 0.1 * _r(H, seed=4), _r(H, seed=5, scale=0.05), _r(1, seed=6)
    sc = K.head_tail(h, g, be, w2, b2)
    ref = torch.nn.functional.layer_norm(h, (H,), g, be, 1e-12) @ w2 + b2
    assert (sc - ref).abs().max().item() <= 2e-5
    # all heads of a step in one launch: different row counts, a 3-D input, per-head parameters
    h2, h3 = _r(7, 5, H, seed=11).relu(), _r(3, H, seed=12).relu()
    g2, be2, w22, b22 = 1 + 0.1 * _r(H, seed=13), 0.1 * _r(H, seed=14), _r(H, seed=15, scale=0.05), _r(1, seed=16)
    s1, s2, s3 = K.head_tail_grouped([(h, g, be, w2, b2), (h2, g2, be2, w22, b22), (h3, g, be, w22, b2)])
    assert s1.shape == sc.shape and s2.shape == (7, 5) and s3.shape == (3,)
    assert (s1 - sc).abs().max().item() <= 2e-6
    assert (s2 - K.head_tail(h2, g2, be2, w22, b22)).abs().max().item() <= 2e-6 and (s3 - K.head_tail(h3, g, be, w22, b2)).abs().max().item() <= 2e-6
    aug, xx = _r(50, H, dtype=dtype), _r(50, H, seed=1, dtype=dtype)
    wa, ba, wo, bo = _r(H, scale=0.05, seed=7), _r(1, seed=8), _r(H, scale=0.05, seed=9), _r(1, seed=10)
    out = K.igl_blend(aug, xx, wa, ba, wo, bo)
    ww = torch.sigmoid(aug.float() @ wa + ba + xx.float() @ wo + bo)[:, None]
    assert (out.float() - (ww * aug.float() + (1 - ww) * xx.float())).abs().max().item() <= tol * 4


def test_nav_logits_vs_reference_loop(K):
    """fusion kernel + host index builder against the reference's Python double loop semantics
    (restated in oracle/dsrg_oracle.py: fuse_logits)."""
    import numpy as np
    from dsrg_amd import synth
    from dsrg_amd.vilmodel import GlocalTextPathNavCMT as M
    from oracle.dsrg_oracle import fuse_logits
    B, G, Vp = 32, 20, 38
    nb = synth.navigation_batch(B, G, Vp, seed=3)
    g_raw, l_raw, f_raw = _r(B, G), _r(B, Vp, seed=1), _r(B, seed=2)
    gm, gv, vn = (torch.from_numpy(nb[k]).cuda() for k in ('gmap_masks', 'gmap_visited_masks', 'vp_nav_masks'))
    src, bw = M.fusion_indices(nb['gmap_vpids'], nb['gmap_visited_masks'], nb['vp_cand_vpids'], G, Vp)
    gl, ll, fl = K.nav_logits(g_raw, l_raw, f_raw, gm, gv, vn, torch.from_numpy(src).cuda(), torch.from_numpy(bw).cuda())
    fw = torch.sigmoid(f_raw)[:, None]
    rgl = (g_raw * fw).masked_fill(gv | ~gm, float('-inf'))
    rll = (l_raw * (1 - fw)).masked_fill(~vn, float('-inf'))
    rfl = fuse_logits(rgl.cpu(), rll.cpu(), nb['gmap_vpids'], nb['gmap_visited_masks'], nb['vp_cand_vpids'], 2)
    for a, r in ((gl, rgl), (ll, rll), (fl, rfl.cuda())):
        fin = torch.isfinite(r)
        assert (torch.isfinite(a) == fin).all()
        assert (a[fin] - r[fin]).abs().max().item() <= 1e-6
    assert (src == -2).any() and (src >= 0).any() and bw.any()


def test_errors_are_loud(K):
    from dsrg_amd._lib import DsrgHipError
    x, w = _r(4, 100), _r(8, 100)
    with pytest.raises(DsrgHipError):
        K.gemm([(x, w, None, None)])            # K not a multiple of 32
    with pytest.raises(DsrgHipError):
        K.gemm([(x.cpu(), w.cpu(), None, None)])  # no CPU path


# ----------------------------------------------------------------------------- backward kernels
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_transpose_colsum_act(K, dtype):
    x = _r(1000, 130, dtype=dtype)
    t = K.transpose_pad(x)
    assert t.shape == (130, 1024)
    assert torch.equal(t[:, :1000], x.t()) and (t[:, 1000:] == 0).all()
    t2 = K.transpose_pad(_r(77, 768), out_dtype=dtype)
    assert t2.dtype == dtype and t2.shape == (768, 128)
    s = K.colsum(x)
    assert (s - x.float().sum(0)).abs().max().item() <= (1e-3 if dtype == torch.float32 else 2e-1)
    cs = torch.zeros(130, device='cuda')           # column sums taken by the transposing kernel itself
    t3 = K.transpose_pad(x, colsum=cs)
    assert torch.equal(t3, t) and (cs - x.float().sum(0)).abs().max().item() <= 1e-3
    # grouped form: ragged shapes, a strided view, a column-sum rider, one launch
    a, b, c = _r(1000, 130, dtype=dtype), _r(77, 768, seed=2, dtype=dtype), _r(300, 200, seed=3, dtype=dtype)[:, :70]
    cs2 = torch.zeros(768, device='cuda')
    ta, tb, tc = K.transpose_pad_grouped([(a, None), (b, cs2), (c, None)])
    for src, dst in ((a, ta), (b, tb), (c, tc)):
        R = src.shape[0]
        assert dst.shape == (src.shape[1], (R + 63) // 64 * 64) and dst.dtype == dtype
        assert torch.equal(dst[:, :R], src.t()) and (dst[:, R:] == 0).all()
    assert (cs2 - b.float().sum(0)).abs().max().item() <= 1e-3
    many = [(_r(65 + i, 64 + i, seed=10 + i, dtype=dtype), None) for i in range(30)]       # > MAX_TRANSPOSES: two launches
    for (src, _), dst in zip(many, K.transpose_pad_grouped(many)):
        assert torch.equal(dst[:, :src.shape[0]], src.t())
    z = _r(64, 3072, dtype=dtype)
    dy = _r(64, 3072, seed=1, dtype=dtype)
    for act in (1, 2, 3):
        zz = z.float().clone().requires_grad_(True)
        yy = ACTS[act](zz)
        yy.backward(dy.float())
        tol = 1e-5 if dtype == torch.float32 else 3e-2
        assert (K.act_fwd(z, act).float() - yy).abs().max().item() <= tol
        assert (K.act_bwd(z, dy, act).float() - zz.grad).abs().max().item() <= tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('H,xf32,res', [(768, True, True), (768, False, False), (128, True, True)])
def test_layernorm_bwd(K, dtype, H, xf32, res):
    M = 333
    x = _r(M, H, dtype=torch.float32 if xf32 else dtype)
    r = _r(M, H, seed=1, dtype=dtype) if res else None
    g, b = 1 + 0.1 * _r(H, seed=2), 0.1 * _r(H, seed=3)
    dy = _r(M, H, seed=4, dtype=dtype)
    dz, dg, db = K.layernorm_bwd(x, r, g, dy, 1e-12)
    zz = (x.float() + (r.float() if res else 0)).requires_grad_(True)
    gg = g.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(zz, (H,), gg, bb, 1e-12).backward(dy.float())
    tol = 2e-5 if dtype == torch.float32 else 4e-2
    assert (dz.float() - zz.grad).abs().max().item() <= tol
    assert (dg - gg.grad).abs().max().item() <= tol * 30 and (db - bb.grad).abs().max().item() <= tol * 30


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_functional_grads_match_autograd(dtype):
    """grouped linear (+GELU), dense+residual+LN composite and LayerNorm: HIP backward vs torch autograd."""
    from dsrg_amd import functional as Fn
    tol = 2e-4 if dtype == torch.float32 else 6e-2

    def leaf(*shape, seed=0, scale=1.0, dt=dtype):
        return _r(*shape, seed=seed, scale=scale, dtype=dt).requires_grad_(True)
    xs = [leaf(4, 20, 768), leaf(4, 38, 768, seed=1)]
    ws = [leaf(3072, 768, seed=2, scale=0.03), leaf(3072, 768, seed=3, scale=0.03)]
    bs = [leaf(3072, seed=4, dt=torch.float32), leaf(3072, seed=5, dt=torch.float32)]
    w2 = [leaf(768, 3072, seed=6, scale=0.02), leaf(768, 3072, seed=7, scale=0.02)]
    b2 = [leaf(768, seed=8, dt=torch.float32), leaf(768, seed=9, dt=torch.float32)]
    gm = [leaf(768, seed=10, scale=0.1, dt=torch.float32), leaf(768, seed=11, scale=0.1, dt=torch.float32)]
    bt = [leaf(768, seed=12, scale=0.1, dt=torch.float32), leaf(768, seed=13, scale=0.1, dt=torch.float32)]
    leaves = xs + ws + bs + w2 + b2 + gm + bt
    hs = Fn.linear_grouped([(xs[j], ws[j], bs[j], None) for j in range(2)], act=1)
    ys = Fn.linear_ln_grouped([(hs[j], w2[j], b2[j], xs[j], gm[j] + 1, bt[j]) for j in range(2)], 1e-12)
    ys = [Fn.layernorm(y, gm[j] + 1, bt[j], 1e-5) for j, y in enumerate(ys)]
    up = [_r(*y.shape, seed=20 + j) for j, y in enumerate(ys)]
    sum((y.float() * u).sum() for y, u in zip(ys, up)).backward()
    mine = [t.grad.float().clone() for t in leaves]
    for t in leaves:
        t.grad = None
    F = torch.nn.functional
    ref_y = []
    for j in range(2):
        h = F.gelu(F.linear(xs[j].float(), ws[j].float(), bs[j]))
        y = F.layer_norm(F.linear(h, w2[j].float(), b2[j]) + xs[j].float(), (768,), gm[j] + 1, bt[j], 1e-12)
        ref_y.append(F.layer_norm(y, (768,), gm[j] + 1, bt[j], 1e-5))
    sum((y * u).sum() for y, u in zip(ref_y, up)).backward()
    for t, m in zip(leaves, mine):
        ref = t.grad.float()
        err = (m - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (tuple(t.shape), err, ref.abs().max().item())


def test_dense_dropout_residual_ln_train_path():
    """the p_drop > 0 branch of the dense+dropout+residual+LayerNorm composite (training only): with the weight set to
    zero the dropped term vanishes, so output and input gradients must equal the plain LayerNorm(residual + bias) ones."""
    from dsrg_amd import functional as Fn
    torch.manual_seed(0)
    M, H = 96, 768
    x = _r(M, H, dtype=torch.bfloat16).requires_grad_(True)
    r = _r(M, H, seed=1, dtype=torch.bfloat16).requires_grad_(True)
    w = torch.zeros(H, H, device='cuda', dtype=torch.bfloat16, requires_grad=True)
    b = torch.zeros(H, device='cuda', requires_grad=True)
    g, be = (1 + 0.1 * _r(H, seed=2)).requires_grad_(True), (0.1 * _r(H, seed=3)).requires_grad_(True)
    y = Fn.linear_ln(x, w, b, g, be, 1e-12, residual=r, p_drop=0.3)
    dy = _r(M, H, seed=4, dtype=torch.bfloat16)
    y.backward(dy)
    rr = r.detach().float().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(rr, (H,), g.detach(), be.detach(), 1e-12)
    ref.backward(dy.float())
    assert (y.float() - ref).abs().max().item() <= 3e-2
    assert (r.grad.float() - rr.grad).abs().max().item() <= 3e-2 * rr.grad.abs().max().item() + 1e-3
    assert torch.isfinite(w.grad).all() and torch.isfinite(b.grad).all() and x.grad.abs().max().item() == 0.0


@pytest.mark.parametrize('B,N', [(64, 20), (7, 100), (3, 1), (5, 250)])
def test_action_tail(K, B, N):
    """softmax / stop probability / argmax / entropy / summed cross-entropy and its gradient / inverse-CDF sample of the
    rollout's action tail (agent.py:484-516) against the torch ops the reference calls."""
    from dsrg_amd import functional as Fn
    g = torch.Generator().manual_seed(B * 1000 + N)
    logits = (torch.randn(B, N, generator=g) * 3).cuda()
    masked = (torch.rand(B, N, generator=g) < 0.3).cuda()
    masked[:, 0] = False
    logits = logits.masked_fill(masked, float('-inf'))
    targets = torch.randint(0, N, (B,), generator=g).cuda()
    targets = torch.where(masked.gather(1, targets[:, None])[:, 0], torch.zeros_like(targets), targets)
    if B > 2:
        targets[1] = -100
    u = torch.rand(B, generator=g).cuda()
    lg = logits.clone().requires_grad_(True)
    o = Fn.action_tail(lg, targets, u)
    ref_l = logits.clone().requires_grad_(True)
    probs = torch.softmax(ref_l, 1)
    loss = torch.nn.functional.cross_entropy(ref_l, targets, ignore_index=-100, reduction='sum')
    assert abs(o['loss'].item() - loss.item()) <= 1e-4 * max(1.0, abs(loss.item()))
    assert (o['stop_prob'] - probs[:, 0]).abs().max().item() <= 1e-6
    assert torch.equal(o['argmax'].long(), ref_l.max(1)[1])
    ent = torch.distributions.Categorical(probs).entropy()
    assert (o['entropy'] - ent).abs().max().item() <= 1e-5
    (o['loss'] * 0.5).backward()
    (loss * 0.5).backward()
    assert (lg.grad - ref_l.grad).abs().max().item() <= 1e-6
    cdf = probs.detach().double().cumsum(1)
    smp = o['sample'].long()
    assert (probs.detach().gather(1, smp[:, None])[:, 0] > 0).all()
    hi = cdf.gather(1, smp[:, None])[:, 0]
    lo = hi - probs.detach().double().gather(1, smp[:, None])[:, 0]
    assert ((u.double() < hi + 1e-6) & (u.double() >= lo - 1e-6)).all()
    # inference form: no targets, no gradient
    o2 = Fn.action_tail(logits)
    assert torch.equal(o2['argmax'], o['argmax']) and 'loss' not in o2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_row_op_backward_kernels(dtype):
    """hand-written backward of the row kernels (ASV mean, GAA pooling, ClsPrediction tail, logit fusion, IGL gate)
    vs torch autograd of a plain fp32 restatement of each op."""
    from dsrg_amd import functional as Fn
    F = torch.nn.functional
    fp32 = dtype == torch.float32
    tol = 2e-4 if fp32 else 5e-2

    def leaf(*shape, seed=0, scale=1.0, dt=dtype):
        return _r(*shape, seed=seed, scale=scale, dtype=dt).requires_grad_(True)

    def check(leaves, mine_fn, ref_fn, name):
        outs = mine_fn()
        outs = outs if isinstance(outs, (tuple, list)) else [outs]
        ups = [_r(*o.shape, seed=50 + i) for i, o in enumerate(outs)]
        sum((o.float() * u).sum() for o, u in zip(outs, ups)).backward()
        mine = [None if t.grad is None else t.grad.float().clone() for t in leaves]
        for t in leaves:
            t.grad = None
        refs = ref_fn()
        refs = refs if isinstance(refs, (tuple, list)) else [refs]
        for o, r in zip(outs, refs):
            fin = torch.isfinite(r)
            assert (torch.isfinite(o.float()) == fin).all(), name
            assert (o.float()[fin] - r[fin]).abs().max().item() <= (1e-4 if fp32 else 5e-2), name
        sum((torch.where(torch.isfinite(r), r, torch.zeros_like(r)) * u).sum() for r, u in zip(refs, ups)).backward()
        for i, (t, m) in enumerate(zip(leaves, mine)):
            ref = t.grad.float()
            assert m is not None, (name, i)
            err = (m - ref).abs().max().item()
            assert err <= tol * max(1.0, ref.abs().max().item()), (name, i, tuple(t.shape), err, ref.abs().max().item())
            t.grad = None

    # ASV mean over groups of 3 rows
    x = leaf(36 * 3 * 5, 128)
    check([x], lambda: Fn.group_mean(x, 3), lambda: x.float().reshape(-1, 3, 128).mean(1), 'group_mean')
    # GAA pooling
    x, w, b = leaf(9, 36, 768), leaf(768, seed=1, scale=0.05, dt=torch.float32), leaf(1, seed=2, dt=torch.float32)

    def gaa_ref():
        xf = x.float()
        a = torch.softmax(torch.tanh(xf @ w + b), 1)
        return (xf * a[..., None]).sum(1)
    check([x, w, b], lambda: Fn.gaa_pool(x, w, b), gaa_ref, 'gaa_pool')
    # ClsPrediction tail (fp32 rows)
    h = leaf(7, 38, 768, seed=3, dt=torch.float32)
    g, be = leaf(768, seed=4, scale=0.1, dt=torch.float32), leaf(768, seed=5, scale=0.1, dt=torch.float32)
    w2, b2 = leaf(768, seed=6, scale=0.05, dt=torch.float32), leaf(1, seed=7, dt=torch.float32)
    if fp32:
        check([h, g, be, w2, b2], lambda: Fn.head_tail(h, g + 1, be, w2, b2), lambda: F.layer_norm(h, (768,), g + 1, be, 1e-12) @ w2 + b2, 'head_tail')
    # IGL gate
    aug, xx = leaf(3, 80, 768, seed=8), leaf(3, 80, 768, seed=9)
    wa, wo = leaf(768, seed=10, scale=0.05, dt=torch.float32), leaf(768, seed=11, scale=0.05, dt=torch.float32)
    ba, bo = leaf(1, seed=12, dt=torch.float32), leaf(1, seed=13, dt=torch.float32)

    def igl_ref():
        a, xf = aug.float(), xx.float()
        gate = torch.sigmoid(a @ wa + ba + xf @ wo + bo)[..., None]
        return gate * a + (1 - gate) * xf
    check([aug, xx, wa, ba, wo, bo], lambda: Fn.igl_blend(aug, xx, wa, ba, wo, bo), igl_ref, 'igl_blend')
    # token-embedding sums: the panorama-token form (every term present) and the text form (two tables + vector)
    for M in (333, 2500):      # one row per wave, and strips of 3 rows per wave (sums carried in registers)
        H, Fd = 768, 7
        base = leaf(M, H, seed=20)
        feat = _r(M, Fd + 1, seed=21)
        Wf, bf = leaf(H, Fd, seed=22, scale=0.3, dt=torch.float32), leaf(H, seed=23, scale=0.02, dt=torch.float32)
        E1, E2, vec = leaf(45, H, seed=24, dt=torch.float32), leaf(3, H, seed=25, dt=torch.float32), leaf(H, seed=26, dt=torch.float32)
        gf, bfb = leaf(H, seed=27, scale=0.1, dt=torch.float32), leaf(H, seed=28, scale=0.1, dt=torch.float32)
        go_, bo_ = leaf(H, seed=29, scale=0.1, dt=torch.float32), leaf(H, seed=30, scale=0.1, dt=torch.float32)
        i1 = torch.randint(0, 45, (M,)).cuda()
        i2g = torch.randint(0, 3, ((M + 2) // 3,)).cuda()

        def emb_ref():
            t = F.layer_norm(feat[:, :Fd] @ Wf.T + bf, (H,), gf + 1, bfb, 1e-12)
            return F.layer_norm(base.float() + t + E1[i1] + E2[i2g.repeat_interleave(3)[:M]] + vec, (H,), go_ + 1, bo_, 1e-12)
        check([base, Wf, bf, E1, E2, vec, gf, bfb, go_, bo_],
              lambda: Fn.embed_ln(M, H, dtype, 'cuda', base=base, feat=feat[:, :Fd], Wf=Wf, bf=bf, lnf=(gf + 1, bfb), E1=E1, idx1=i1,
                                  E2=E2, idx2=(i2g, 3), vec=vec, lno=(go_ + 1, bo_)), emb_ref, 'embed_ln/pano')
        Ew = leaf(500, H, seed=31, dt=torch.float32)
        ids = torch.randint(0, 500, (M,)).cuda()
        pos = (torch.arange(M) % 80).int().cuda()
        Ep = leaf(80, H, seed=32, dt=torch.float32)
        check([Ew, Ep, vec, go_, bo_],
              lambda: Fn.embed_ln(M, H, dtype, 'cuda', E1=Ew, idx1=ids, E2=Ep, idx2=pos, vec=vec, lno=(go_ + 1, bo_)),
              lambda: F.layer_norm(Ew[ids] + Ep[pos.long()] + vec, (H,), go_ + 1, bo_, 1e-12), 'embed_ln/text')
        ids0 = torch.where(pos >= 50, torch.zeros_like(ids), ids)      # padded tail: row 0 gets no gradient (padding_idx)
        check([Ew, Ep, vec, go_, bo_],
              lambda: Fn.embed_ln(M, H, dtype, 'cuda', E1=Ew, idx1=ids0, E2=Ep, idx2=pos, vec=vec, lno=(go_ + 1, bo_), E1_padding_idx=0),
              lambda: F.layer_norm(F.embedding(ids0, Ew, padding_idx=0) + Ep[pos.long()] + vec, (H,), go_ + 1, bo_, 1e-12), 'embed_ln/text-pad')
        f14, W14 = _r(M, 14, seed=33), leaf(H, 14, seed=34, scale=0.3, dt=torch.float32)
        check([base, W14, bf, gf, bfb],
              lambda: Fn.embed_ln(M, H, dtype, 'cuda', base=base, feat=f14, Wf=W14, bf=bf, lnf=(gf + 1, bfb)),
              lambda: base.float() + F.layer_norm(f14 @ W14.T + bf, (H,), gf + 1, bfb, 1e-12), 'embed_ln/vp')
    # dynamic fusion + logit fusion (fp32 scores), with a fuse score and without
    if fp32:
        from dsrg_amd import synth
        import numpy as np
        B, G, Vp = 6, 20, 38
        nb = synth.navigation_batch(B, G, Vp, 7)
        from dsrg_amd.vilmodel import GlocalTextPathNavCMT
        src, bw = GlocalTextPathNavCMT.fusion_indices(nb['gmap_vpids'], nb['gmap_visited_masks'], nb['vp_cand_vpids'], G, Vp, 2)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        gm, gv, vn = dev(nb['gmap_masks']), dev(nb['gmap_visited_masks']), dev(nb['vp_nav_masks'])
        src_t, bw_t = dev(src), dev(bw)
        for use_f in (True, False):
            gr, lr = leaf(B, G, seed=14, dt=torch.float32), leaf(B, Vp, seed=15, dt=torch.float32)
            fr = leaf(B, seed=16, dt=torch.float32) if use_f else None

            def nav_ref():
                fw = torch.sigmoid(fr)[:, None] if fr is not None else 0.5
                ninf = float('-inf')
                gl = (gr * fw).masked_fill(gv | ~gm, ninf)
                ll = (lr * (1 - fw)).masked_fill(~vn, ninf)
                llz = torch.where(vn, ll, torch.zeros_like(ll))
                bws = torch.where(bw_t.bool(), ll, torch.zeros_like(ll)).sum(1, keepdim=True)
                s = src_t.long()
                picked = torch.gather(ll, 1, s.clamp(min=0))
                add = torch.where(s >= 0, picked, torch.where(s == -2, bws.expand_as(picked), torch.zeros_like(picked)))
                add = torch.cat([ll[:, :1], add[:, 1:]], 1)
                return gl, ll, gl + add
            check([gr, lr] + ([fr] if use_f else []), lambda: Fn.nav_logits(gr, lr, fr, gm, gv, vn, src_t, bw_t), nav_ref, 'nav_logits')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_segment_wsum(dtype):
    """segment-weighted row gather (pre-training graph-map aggregation) forward and backward vs index_select/index_add_."""
    import numpy as np
    from dsrg_amd import functional as Fn
    rng = np.random.RandomState(3)
    n0, n1, m, H, E = 180, 5, 40, 768, 300
    a = _r(n0, H, dtype=dtype).requires_grad_(True)
    b = _r(n1, H, seed=1, dtype=dtype).requires_grad_(True)
    dst = rng.randint(0, m - 3, E)            # the last rows stay empty -> zeros
    src = rng.randint(0, n0 + n1, E)
    w = rng.rand(E).astype(np.float32)
    out = Fn.segment_wsum(a, b, dst, src, w, m)
    up = _r(m, H, seed=2)
    (out.float() * up).sum().backward()
    ga, gb = a.grad.float().clone(), b.grad.float().clone()
    a.grad = b.grad = None
    flat = torch.cat([a.float(), b.float()], 0)
    rows = flat.index_select(0, torch.from_numpy(src).cuda()) * torch.from_numpy(w).cuda()[:, None]
    ref = torch.zeros(m, H, device='cuda').index_add_(0, torch.from_numpy(dst).cuda(), rows)
    (ref * up).sum().backward()
    tol = 1e-4 if dtype == torch.float32 else 5e-2
    assert (out.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert (out[m - 3:] == 0).all()
    for mine, t in ((ga, a), (gb, b)):
        assert (mine - t.grad.float()).abs().max().item() <= tol * max(1.0, t.grad.abs().max().item())


def _keep_mask(seed, B, nh, Nq, Nk, p, offset=0):
    """numpy restatement of drop_keep() (common.h): splitmix64 finaliser of seed + (idx+1)*golden, top 24 bits -> u"""
    import numpy as np
    idx = np.arange(B * nh * Nq * Nk, dtype=np.uint64) + np.uint64(offset)
    with np.errstate(over='ignore'):
        z = np.uint64(seed) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy((u >= np.float32(p)).reshape(B, nh, Nq, Nk)).cuda()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,nh,Nq,Nk,mode,sprel', [(3, 12, 20, 80, 1, False), (2, 12, 38, 38, 1, True), (2, 2, 108, 108, 3, False),
                                                    (2, 12, 36, 36, 2, False), (2, 12, 70, 250, 1, False)])
def test_attention_dropout(K, dtype, B, nh, Nq, Nk, mode, sprel):
    """in-kernel dropout on the attention probabilities: forward and backward against torch with the SAME keep mask (the
    mask is a counter-based hash, restated in numpy above); the normaliser uses the undropped probabilities, like
    nn.Dropout applied after the softmax."""
    H, p, seed = nh * 64, 0.1, 0x1234_5678_9ABC
    if Nq == Nk:
        qkv = _r(B, Nq, 3 * H, dtype=dtype)
        q, k, v = qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:]
    else:
        q, kv = _r(B, Nq, H, dtype=dtype), _r(B, Nk, 2 * H, seed=1, dtype=dtype)
        k, v = kv[..., :H], kv[..., H:]
    lens = torch.randint(1, Nk + 1, (B,), generator=torch.Generator().manual_seed(5)); lens[0] = Nk
    kmask = (torch.arange(Nk)[None] < lens[:, None]).cuda()
    dists = (_r(B, Nq, Nk, seed=7).abs() * 5) if sprel else None
    sw, sb = (torch.tensor([-0.21]).cuda(), torch.tensor([0.07]).cuda()) if sprel else (None, None)
    do = _r(B, Nq, H, seed=11, dtype=dtype)
    out, lse = K.attention(q, k, v, nh, kmask, mode, dists, sw, sb, want_lse=True, p_drop=p, seed=seed)
    dq, dk, dv, _ = K.attention_bwd(q, k, v, out, do, lse, nh, kmask, mode, dists, sw, sb, p_drop=p, seed=seed)
    keep = _keep_mask(seed, B, nh, Nq, Nk, p)
    assert 0.85 < keep.float().mean().item() < 0.95
    leaves = [t.float().clone().requires_grad_(True) for t in (q, k, v)]
    sp = lambda t: t.reshape(B, t.shape[1], nh, 64).permute(0, 2, 1, 3)
    s = sp(leaves[0]) @ sp(leaves[1]).transpose(-1, -2) * 0.125
    if mode == 1:
        s = s + (1.0 - kmask[:, None, None].float()) * -10000.0
    elif mode == 2:
        s = s.masked_fill(~kmask[:, None, None], float('-inf'))
    elif mode == 3:
        s = s + kmask[:, None, None].float()
    if dists is not None:
        s = s + (dists * sw + sb)[:, None]
    pr = torch.softmax(s, -1) * keep.float() / (1.0 - p)
    ref = (pr @ sp(leaves[2])).permute(0, 2, 1, 3).reshape(B, Nq, H)
    ref.backward(do.float())
    fp32 = dtype == torch.float32
    assert (out.float() - ref).abs().max().item() <= (3e-5 if fp32 else 3e-2)
    for name, got, want in (('dq', dq, leaves[0].grad), ('dk', dk, leaves[1].grad), ('dv', dv, leaves[2].grad)):
        err = (got.float() - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        assert err <= (2e-4 if fp32 else 4e-2), (name, err)
    # a different seed gives a different mask; p = 0 is exactly the plain kernel
    assert not torch.equal(out, K.attention(q, k, v, nh, kmask, mode, dists, sw, sb, p_drop=p, seed=seed + 1))
    assert torch.equal(K.attention(q, k, v, nh, kmask, mode, dists, sw, sb), K.attention(q, k, v, nh, kmask, mode, dists, sw, sb, p_drop=0.0, seed=99))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cross', [False, True])
def test_attention_packed_backward_matches_sliced(K, dtype, cross):
    """functional.attention_packed (gradients written straight into the packed d(qkv) / d(kv): dsrg_attention_bwd_strided)
    against functional.attention on slices of the same packed tensor (torch.autograd assembling the gradient), and
    functional.split_cols against plain slicing."""
    from dsrg_amd import functional as Fn
    B, Nq, Nk, nh = 5, 38, 80 if cross else 38, 12
    H = nh * 64
    torch.manual_seed(3)
    pk0 = (torch.randn(B, Nk, (2 if cross else 3) * H, device='cuda') * 0.5).to(dtype)
    q0 = (torch.randn(B, Nq, H, device='cuda') * 0.5).to(dtype) if cross else None
    kmask = torch.arange(Nk, device='cuda')[None] < torch.tensor([Nk, 7, 20, Nk - 1, 33], device='cuda')[:, None]
    dists = torch.rand(B, Nq, Nk, device='cuda') if not cross else None
    sw = torch.tensor([[0.3]], device='cuda', requires_grad=True) if not cross else None
    sb = torch.tensor([0.1], device='cuda', requires_grad=True) if not cross else None
    go = torch.randn(B, Nq, H, device='cuda').to(dtype)
    res = []
    for packed_api in (False, True):
        pk = pk0.clone().requires_grad_(True)
        q = q0.clone().requires_grad_(True) if cross else None
        if sw is not None:
            sw.grad = sb.grad = None
        if packed_api:
            o = Fn.attention_packed(pk, nh, q, kmask, 1, dists, sw, sb)
        elif cross:
            o = Fn.attention(q, pk[..., :H], pk[..., H:], nh, kmask, 1, dists, sw, sb)
        else:
            o = Fn.attention(pk[..., :H], pk[..., H:2 * H], pk[..., 2 * H:], nh, kmask, 1, dists, sw, sb)
        o.backward(go)
        res.append((o.detach(), pk.grad, q.grad if cross else None, sw.grad.clone() if sw is not None else None))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])           # same kernels, same values: only where they are written differs
    if cross:
        assert torch.equal(res[0][2], res[1][2])
    else:
        assert torch.allclose(res[0][3], res[1][3], rtol=1e-3, atol=1e-4)      # float atomics: order may differ
    # split_cols: gradient of the blocks = concatenation, missing blocks = zeros
    x = pk0.clone().requires_grad_(True)
    widths = [H, pk0.shape[-1] - H]
    a, b = Fn.split_cols(x, widths)
    (a.float() * 2).sum().backward()
    assert torch.equal(x.grad[..., :H], torch.full_like(x.grad[..., :H], 2.0)) and x.grad[..., H:].abs().max().item() == 0.0


@pytest.mark.parametrize('B,N', [(173, 30522), (24, 37), (1, 1), (64, 257)])
def test_cross_entropy_rows(K, B, N):
    """functional.cross_entropy_rows (dsrg_ce_rows_fwd / _bwd) against F.cross_entropy(reduction='none') and its autograd:
    -inf masked logits, ignored targets, per-row upstream gradients."""
    from dsrg_amd import functional as Fn
    torch.manual_seed(B + N)
    lg0 = torch.randn(B, N, device='cuda') * 3
    tg = torch.randint(0, N, (B,), device='cuda')
    if N > 4:
        lg0[:, 1::3] = -float('inf')
        lg0[torch.arange(B, device='cuda'), tg] = torch.randn(B, device='cuda')       # the target itself is never masked
        tg[::5] = -100
    gup = torch.rand(B, device='cuda') + 0.1
    a = lg0.clone().requires_grad_(True); b = lg0.clone().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(a, tg, reduction='none')
    got = Fn.cross_entropy_rows(b, tg)
    assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    (want * gup).sum().backward(); (got * gup).sum().backward()
    assert (a.grad - b.grad).abs().max().item() <= 2e-6
    with torch.no_grad():
        assert torch.equal(Fn.cross_entropy_rows(lg0, tg), got.detach())
    d16 = K.ce_rows_bwd(lg0, tg, K.ce_rows(lg0, tg)[1], gup, out_dtype=torch.bfloat16)
    assert (d16.float() - b.grad).abs().max().item() <= 8e-3
    # a NaN logit away from the target poisons its row's loss, as in F.cross_entropy (the other rows stay as they were)
    if N > 4:
        lgn = lg0.clone()
        row = 1 if B > 1 else 0
        col = (int(tg[row].item()) + 2) % N if int(tg[row].item()) >= 0 else 0
        lgn[row, col] = float('nan')
        wn = torch.nn.functional.cross_entropy(lgn, tg, reduction='none')
        gn = Fn.cross_entropy_rows(lgn, tg)
        assert torch.equal(torch.isnan(gn), torch.isnan(wn))
        ok = ~torch.isnan(wn)
        assert (gn[ok] - wn[ok]).abs().max().item() <= 1e-5 * max(1.0, wn[ok].abs().max().item())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,H', [(1920, 768), (37, 128), (300, 1024)])
def test_linear_ln_fused_hidden_dropout(K, dtype, M, H):
    """LayerNorm(dropout(dense(h)) + input) with the dropout inside the LayerNorm kernels (dsrg_layernorm_fwd_grouped_drop /
    dsrg_layernorm_bwd_drop): forward and every gradient against torch with the SAME keep mask (numpy restatement of the
    counter hash; problem 1 of a grouped call uses index offset 1 << 40)."""
    from dsrg_amd import functional as Fn
    p, eps, Kin = 0.1, 1e-12, 256
    torch.manual_seed(7)
    mk = lambda *s, sc=1.0: (torch.randn(*s, device='cuda') * sc)
    xs = [mk(M, Kin).to(dtype), mk(M // 2 + 1, Kin).to(dtype)]
    ws = [mk(H, Kin, sc=0.05).to(dtype) for _ in range(2)]
    bs = [mk(H, sc=0.1) for _ in range(2)]
    rs = [mk(x.shape[0], H).to(dtype) for x in xs]
    gs = [1 + mk(H, sc=0.1) for _ in range(2)]
    bt = [mk(H, sc=0.1) for _ in range(2)]
    leaves = [t.clone().requires_grad_(True) for t in xs + ws + bs + rs + gs + bt]
    X, W, Bv, R, G, Bt = leaves[0:2], leaves[2:4], leaves[4:6], leaves[6:8], leaves[8:10], leaves[10:12]
    seed = 123456789
    orig = Fn.dropout_seed
    Fn.dropout_seed = lambda: seed
    try:
        ys = Fn.linear_ln_grouped([(X[j], W[j], Bv[j], R[j], G[j], Bt[j]) for j in range(2)], eps, p)
    finally:
        Fn.dropout_seed = orig
    gos = [mk(*y.shape).to(dtype) for y in ys]
    torch.autograd.backward(ys, gos)
    got = [t.grad.clone() for t in leaves]
    # reference: same arithmetic in fp32 torch with the same mask (the dense output is fp32 in the kernel path too)
    ref_leaves = [t.detach().float().clone().requires_grad_(True) for t in leaves]
    X2, W2, B2, R2, G2, Bt2 = ref_leaves[0:2], ref_leaves[2:4], ref_leaves[4:6], ref_leaves[6:8], ref_leaves[8:10], ref_leaves[10:12]
    outs = []
    for j in range(2):
        Mj = X2[j].shape[0]
        keep = _keep_mask(seed, 1, 1, Mj, H, p, offset=j << 40)[0, 0].float() / (1 - p)
        z = X2[j] @ W2[j].T + B2[j]
        outs.append(torch.nn.functional.layer_norm(z * keep + R2[j], (H,), G2[j], Bt2[j], eps))
    torch.autograd.backward(outs, [g.float() for g in gos])
    tol = 2e-4 if dtype == torch.float32 else 4e-2
    for j in range(2):
        assert (ys[j].float() - outs[j]).abs().max().item() <= tol * max(1.0, outs[j].abs().max().item())
    for a, b, name in zip(got, [t.grad for t in ref_leaves], ['x0', 'x1', 'w0', 'w1', 'b0', 'b1', 'r0', 'r1', 'g0', 'g1', 'bt0', 'bt1']):
        sc = max(1.0, b.abs().max().item())
        assert (a.float() - b).abs().max().item() <= (5e-4 if dtype == torch.float32 else 6e-2) * sc, name


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K_,act', [(1920, 3072, 768, 1), (37, 128, 64, 2), (300, 776, 128, 3)])
def test_gemm_pre_activation_output_and_act_grad_epilogue(K, dtype, M, N, K_, act):
    """dsrg_gemm_problem_t::c_pre (forward: act(z) and z from one launch) and ::res_act (backward: (dY W) * act'(z) out of
    the dgrad epilogue) against the separate activation kernels."""
    x, w, b = _r(M, K_, dtype=dtype), _r(N, K_, seed=1, scale=0.05, dtype=dtype), _r(N, seed=2)
    (y,), (z,) = K.gemm([(x, w, b, None)], act, False, want_pre=True)
    z0 = K.gemm([(x, w, b, None)], 0, False)[0]
    y0 = K.gemm([(x, w, b, None)], act, False)[0]
    assert torch.equal(z, z0) and torch.equal(y, y0)
    # backward: dz = (dy2 @ w2) * act'(z)
    N2 = 256
    dy2, w2 = _r(M, N2, seed=3, dtype=dtype), _r(N2, N, seed=4, scale=0.05, dtype=dtype)
    ref = (dy2.float() @ w2.float()) * {1: lambda t: 0.5 * (1 + torch.erf(t / math.sqrt(2))) + t * torch.exp(-0.5 * t * t) / math.sqrt(2 * math.pi),
                                         2: lambda t: (t > 0).float(), 3: lambda t: 1 - torch.tanh(t) ** 2}[act](z.float())
    tol = (2e-4 if dtype == torch.float32 else 3e-2) * max(1.0, ref.abs().max().item())
    if dtype == torch.bfloat16:
        got = K.gemm([(dy2, w2, None, z)], trans=2, res_acts=[act])[0]
        assert (got.float() - ref).abs().max().item() <= tol
    got = K.gemm([(dy2, w2.T.contiguous(), None, z)], res_acts=[act])[0]
    assert (got.float() - ref).abs().max().item() <= tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('p_drop', [0.0, 0.1])
def test_ffn_fused_node_matches_separate_nodes(K, dtype, p_drop):
    """functional.ffn_grouped as one autograd node (pre-activation from the GEMM epilogue, activation backward inside the
    dgrad epilogue) against linear + activation + linear_ln nodes: outputs and every gradient, same dropout seed."""
    from dsrg_amd import functional as Fn
    M, H, F_ = 650, 128, 512
    torch.manual_seed(5)
    mk = lambda *s, sc=1.0: torch.randn(*s, device='cuda') * sc
    base = [mk(M, H).to(dtype), mk(F_, H, sc=0.05).to(dtype), mk(F_, sc=0.1), mk(H, F_, sc=0.05).to(dtype), mk(H, sc=0.1), 1 + mk(H, sc=0.1), mk(H, sc=0.1)]
    go = mk(M, H).to(dtype)
    outs = []
    for fused in (True, False):
        lv = [t.clone().requires_grad_(True) for t in base]
        x, w1, b1, w2, b2, g, bt = lv
        Fn.FUSED_FFN = fused
        orig = Fn.dropout_seed
        Fn.dropout_seed = lambda: 4242
        try:
            y = Fn.ffn_grouped([(x, w1, b1, w2, b2, x, g, bt)], 1e-12, p_drop)[0]
        finally:
            Fn.dropout_seed = orig
            Fn.FUSED_FFN = True
        y.backward(go)
        outs.append([y.detach()] + [t.grad for t in lv])
    tol = 1e-4 if dtype == torch.float32 else 5e-2      # bf16: the separate path applies GELU to the bf16-rounded pre-activation
    for a, b, name in zip(outs[0], outs[1], ['y', 'x', 'w1', 'b1', 'w2', 'b2', 'gamma', 'beta']):
        assert (a.float() - b.float()).abs().max().item() <= tol * max(1.0, b.float().abs().max().item()), name


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cross,p_hid', [(False, 0.0), (False, 0.1), (True, 0.0), (True, 0.1)])
def test_attention_block_fused_node_matches_separate_nodes(K, dtype, cross, p_hid):
    """functional.attn_block_grouped as one autograd node against projection + attention + dense/LayerNorm nodes: two
    grouped problems (one with the sprel bias), outputs and every gradient incl. d(kv) and d(sprel)."""
    from dsrg_amd import functional as Fn
    B, nh = 3, 2
    H = nh * 64
    Nq, Nk = (20, 38), (33, 38)
    torch.manual_seed(11)
    mk = lambda *s, sc=1.0: torch.randn(*s, device='cuda') * sc
    base, masks, dists = [], [], []
    for j in range(2):
        nk = Nk[j] if cross else Nq[j]
        base.append([mk(B, Nq[j], H).to(dtype), mk(H if cross else 3 * H, H, sc=0.08).to(dtype), mk(H if cross else 3 * H, sc=0.1), mk(H, H, sc=0.08).to(dtype),
                     mk(H, sc=0.1), 1 + mk(H, sc=0.1), mk(H, sc=0.1), mk(B, nk, 2 * H).to(dtype) if cross else None])
        masks.append(torch.arange(nk, device='cuda')[None] < torch.tensor([nk, 5, nk - 2], device='cuda')[:, None])
        dists.append(torch.rand(B, Nq[j], nk, device='cuda') if (j == 0 and not cross) else None)
    sw0, sb0 = torch.tensor([[0.3]], device='cuda'), torch.tensor([0.1], device='cuda')
    gos = [mk(B, Nq[j], H).to(dtype) for j in range(2)]
    outs = []
    for fused in (True, False):
        lv = [[None if t is None else t.clone().requires_grad_(True) for t in b] for b in base]
        sw, sb = sw0.clone().requires_grad_(True), sb0.clone().requires_grad_(True)
        Fn.FUSED_ATTN_BLOCK = fused
        orig = Fn.dropout_seed
        Fn.dropout_seed = lambda: 777
        try:
            ys = Fn.attn_block_grouped([tuple(lv[j][:7]) + (lv[j][7], masks[j], dists[j]) for j in range(2)], nh, 1, 1e-12, 0.0, p_hid,
                                       None if cross else sw, None if cross else sb)
        finally:
            Fn.dropout_seed = orig
            Fn.FUSED_ATTN_BLOCK = True
        torch.autograd.backward(ys, gos)
        outs.append([y.detach() for y in ys] + [t.grad for b in lv for t in b if t is not None] + ([] if cross else [sw.grad, sb.grad]))
    tol = 1e-4 if dtype == torch.float32 else 5e-2
    assert len(outs[0]) == len(outs[1])
    for idx, (a, b) in enumerate(zip(outs[0], outs[1])):
        assert a is not None and b is not None, idx
        assert (a.float() - b.float()).abs().max().item() <= tol * max(1.0, b.float().abs().max().item()), idx


# ----------------------------------------------------------------------------- round 3: output placement / length masks (no glue kernels)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_layernorm_output_placement(K, dtype):
    """dsrg_ln_problem_t o_seg / aux: segments of rows stored into slots 2.. of a larger token buffer, every seg-th row also
    into two more destinations, next to a plain problem in the same launch - all equal to the plain launch's rows."""
    B, V, Vp, H = 5, 36, 38, 768
    x, x2 = _r(B * V, H), _r(7, H, seed=1)
    g, b = 1 + 0.2 * _r(H, seed=2), 0.1 * _r(H, seed=3)
    ref, ref2 = K.layernorm_grouped([(x, g, b, None), (x2, g, b, None)], 1e-12, dtype)
    buf = torch.full((B, Vp, H), 7.0, device='cuda', dtype=dtype)
    cat = torch.zeros(B, 2 * H, device='cuda', dtype=dtype)
    mem = torch.zeros(B, 20, H, device='cuda', dtype=dtype)
    out, out2 = K.layernorm_grouped([(x, g, b, None), (x2, g, b, None)], 1e-12, dtype, outs=[buf[:, 2:], None],
                                    aux=[(V, cat[:, H:], mem[:, 1]), None])
    assert out.data_ptr() == buf[:, 2:].data_ptr()
    assert torch.equal(buf[:, 2:].reshape(B * V, H), ref) and torch.equal(out2, ref2)
    assert (buf[:, :2] == 7.0).all()                                        # the other slots are untouched
    assert torch.equal(cat[:, H:], ref.view(B, V, H)[:, 0]) and (cat[:, :H] == 0).all()
    assert torch.equal(mem[:, 1], ref.view(B, V, H)[:, 0]) and (mem[:, 0] == 0).all() and (mem[:, 2:] == 0).all()
    # aux with seg = 1 (every row): the RMF memory into both [MEM] slots
    y = _r(B, H, seed=4)
    r1 = K.layernorm_grouped([(y, g, b, None)], 1e-12, dtype)[0]
    K.layernorm_grouped([(y, g, b, None)], 1e-12, dtype, aux=[(1, buf[:, 1], mem[:, 3])])
    assert torch.equal(buf[:, 1], r1) and torch.equal(mem[:, 3], r1) and torch.equal(buf[:, 2:].reshape(B * V, H), ref)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gaa_pool_strided_input_and_scatter(K, dtype):
    B, V, G, H = 6, 36, 9, 768
    buf = _r(B, V + 2, H, dtype=dtype)
    w, b = _r(H, seed=1, scale=0.05), _r(1, seed=2)
    ref = K.gaa_pool(buf[:, 2:].contiguous(), w, b)
    gm = torch.zeros(B, G, H, device='cuda', dtype=dtype)
    idx = torch.tensor([2, 8, 3, 2, 5, 7], device='cuda')
    got = K.gaa_pool(buf[:, 2:], w, b, scatter=(gm, idx))
    assert torch.equal(got, ref)
    want = torch.zeros_like(gm)
    want[torch.arange(B, device='cuda'), idx] = ref
    assert torch.equal(gm, want)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('mode', [1, 2, 3])
def test_attention_key_lengths_equal_masks(K, dtype, mode):
    """dsrg_attn_problem_t klens / kseg / klens2: validity from lengths is bit-identical to the bool mask they stand for
    (one length per episode; one per view for the V*O object tokens; the sum of two lengths)."""
    B, nh, V, O = 5, 2, 36, 3
    q, k, v = (_r(B, V * O, nh * 64, seed=s, dtype=dtype) for s in range(3))
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(0, O + 1, (B, V), generator=g).cuda()
    lens[:, 0] = O                                               # (-inf mode: never a fully masked row)
    kl = K.KeyLens(lens.contiguous(), O)
    mask = (torch.arange(O, device='cuda')[None, None] < lens[..., None]).reshape(B, V * O).contiguous()
    assert torch.equal(kl.mask(V * O), mask)
    ref = K.attention(q, k, v, nh, mask, mode)
    assert torch.equal(K.attention(q, k, v, nh, kl, mode), ref)
    # one length per episode, two problems in one launch, the second with the sum of two lengths
    Nk = 56
    k2, v2 = _r(B, Nk, nh * 64, seed=7, dtype=dtype), _r(B, Nk, nh * 64, seed=8, dtype=dtype)
    l0 = torch.tensor([36, 30, 36, 1, 20], device='cuda')
    l1 = torch.tensor([20, 0, 3, 0, 11], device='cuda')
    m1 = (torch.arange(V * O, device='cuda')[None] < (l0 * 2)[:, None]).contiguous()
    m2 = (torch.arange(Nk, device='cuda')[None] < (l0 + l1)[:, None]).contiguous()
    r = K.attention_grouped([(q, k, v, m1, None), (q[:, :40], k2, v2, m2, None)], nh, mode)
    o = K.attention_grouped([(q, k, v, K.KeyLens((l0 * 2).contiguous(), V * O), None), (q[:, :40], k2, v2, K.KeyLens(l0, Nk, l1), None)], nh, mode)
    assert torch.equal(o[0], r[0]) and torch.equal(o[1], r[1])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_embed_ln_gathered_base(K, dtype):
    """dsrg_embed_ln_fwd_seg: REVERIE's per-episode [views[:vl] ; objects[:ol] ; zero pad] inside the token-embedding kernel
    against the same launch on a torch.gather-ed base."""
    B, V, No, H, F = 7, 36, 20, 768, 7
    T = V + No
    view, o1 = _r(B * V, H, dtype=dtype), _r(B * No, H, seed=1, dtype=dtype)
    vl = torch.tensor([36, 36, 30, 36, 1, 36, 20], device='cuda')
    ol = torch.tensor([20, 0, 5, 13, 0, 1, 20], device='cuda')
    feat, Wf, bf = _r(B * T, F, seed=2), _r(H, F, seed=3, scale=0.3), _r(H, seed=4, scale=0.1)
    lg, lb, og, ob = 1 + 0.1 * _r(H, seed=5), 0.1 * _r(H, seed=6), 1 + 0.1 * _r(H, seed=7), 0.1 * _r(H, seed=8)
    E1 = _r(3, H, seed=9)
    idx = torch.randint(0, 3, (B * T,), generator=torch.Generator().manual_seed(1)).cuda()
    vec = _r(H, seed=10)
    pos = torch.arange(T, device='cuda')[None]
    sel = torch.where(pos < vl[:, None], pos, torch.where(pos < (vl + ol)[:, None], V + pos - vl[:, None], torch.full_like(pos, V + No)))
    src = torch.cat([view.view(B, V, H), o1.view(B, No, H), torch.zeros(B, 1, H, device='cuda', dtype=dtype)], 1)
    base = torch.gather(src, 1, sel[..., None].expand(B, T, H)).reshape(B * T, H).contiguous()
    kw = dict(feat=feat, Wf=Wf, bf=bf, lnf=(lg, lb), E1=E1, idx1=idx, vec=vec, lno=(og, ob))
    ref = K.embed_ln(B * T, H, dtype, 'cuda', base=base, **kw)
    got = K.embed_ln(B * T, H, dtype, 'cuda', base=view, seg=(o1, T, V, No, vl, ol), **kw)
    assert torch.equal(got, ref)


def test_nav_logits_with_object_grounding(K):
    B, G, Vp = 6, 11, 40
    g, l, f, o = _r(B, G), _r(B, Vp, seed=1), _r(B, seed=2), _r(B, Vp, seed=3)
    gen = torch.Generator().manual_seed(3)
    rb = lambda *s: (torch.rand(*s, generator=gen) > 0.4).cuda()
    gm, gv, vn, vo = rb(B, G), rb(B, G), rb(B, Vp), rb(B, Vp)
    src = torch.randint(-2, Vp, (B, G), generator=gen).int().cuda()
    bw = (torch.rand(B, Vp, generator=gen) > 0.7).to(torch.uint8).cuda()
    a = K.nav_logits(g, l, f, gm, gv, vn, src, bw)
    b4 = K.nav_logits(g, l, f, gm, gv, vn, src, bw, o, vo)
    assert all(torch.equal(x, y) for x, y in zip(a, b4[:3]))
    assert torch.equal(b4[3], o.masked_fill(~vo, -float('inf')))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dropout_seed_offset_word(K, dtype):
    """dsrg_set_dropout_offset: every dropout kernel (attention forward / backward, hidden-state dropout in front of the
    residual LayerNorm forward / backward) run with seed s and the device word holding k must equal the same kernel run
    with seed s + k and no word: that is what lets a captured training step draw new masks on every replay."""
    from dsrg_amd import functional as Fn
    B, nh, Nq, Nk, p, seed, k_off = 2, 12, 38, 80, 0.1, 0x2345_6789, 5
    H = nh * 64
    q, kv = _r(B, Nq, H, dtype=dtype), _r(B, Nk, 2 * H, seed=1, dtype=dtype)
    k, v = kv[..., :H], kv[..., H:]
    kmask = (torch.arange(Nk)[None] < torch.tensor([Nk, 17])[:, None]).cuda()
    do = _r(B, Nq, H, seed=11, dtype=dtype)
    x, w, b = _r(300, 256, seed=3, dtype=dtype), _r(H, 256, seed=4, scale=0.05, dtype=dtype), _r(H, seed=5)
    res, g, bt = _r(300, H, seed=6, dtype=dtype), 1 + _r(H, seed=7, scale=0.1), _r(H, seed=8, scale=0.1)
    go = _r(300, H, seed=9, dtype=dtype)

    def run(s):
        out, lse = K.attention(q, k, v, nh, kmask, 1, None, None, None, want_lse=True, p_drop=p, seed=s)
        dq, dk, dv, _ = K.attention_bwd(q, k, v, out, do, lse, nh, kmask, 1, None, None, None, p_drop=p, seed=s)
        leaves = [t.clone().requires_grad_(True) for t in (x, w, res)]
        orig = Fn.dropout_seed
        Fn.dropout_seed = lambda: s
        try:
            y = Fn.linear_ln(leaves[0], leaves[1], b, g, bt, 1e-12, residual=leaves[2], p_drop=p)
        finally:
            Fn.dropout_seed = orig
        y.backward(go)
        return [out, dq, dk, dv, y.detach()] + [t.grad for t in leaves]

    want = run(seed + k_off)
    word = torch.full((1,), k_off, dtype=torch.int64, device='cuda')
    K.set_dropout_offset(word)
    try:
        got = run(seed)
        word.fill_(0)
        plain = run(seed)
    finally:
        K.set_dropout_offset(None)
    for a, c in zip(got, want):
        assert torch.equal(a, c)
    assert not torch.equal(plain[0], want[0]) and not torch.equal(plain[4], want[4])
    for a, c in zip(run(seed), plain):         # cleared: the seed alone again
        assert torch.equal(a, c)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dropout_add(K, dtype):
    """functional.dropout_add (dsrg_dropout_add): residual + dropout(x) against torch with the SAME keep mask (numpy
    restatement of the counter hash), forward and both gradients; identity when not training."""
    from dsrg_amd import functional as Fn
    B, N, H, p, seed = 3, 37, 768, 0.1, 987654321
    x0, r0 = _r(B, N, H, dtype=dtype), _r(B, N, H, seed=1, dtype=dtype)
    go = _r(B, N, H, seed=2, dtype=dtype)
    keep = _keep_mask(seed, 1, 1, B * N, H, p)[0, 0].reshape(B, N, H).float() / (1 - p)
    assert 0.85 < (keep > 0).float().mean().item() < 0.95
    orig = Fn.dropout_seed
    Fn.dropout_seed = lambda: seed
    try:
        for with_res in (True, False):
            x, r = x0.clone().requires_grad_(True), (r0.clone().requires_grad_(True) if with_res else None)
            y = Fn.dropout_add(x, p, r, True)
            y.backward(go)
            want = x0.float() * keep + (r0.float() if with_res else 0.0)
            tol = 1e-6 if dtype == torch.float32 else 2e-2
            assert (y.float() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
            assert (x.grad.float() - go.float() * keep).abs().max().item() <= tol * max(1.0, go.float().abs().max().item())
            if with_res:
                assert torch.equal(r.grad, go)
        with torch.no_grad():
            assert torch.equal(Fn.dropout_add(x0, p, None, False), x0)
            assert torch.equal(Fn.dropout_add(x0, p, r0, False), x0 + r0)
    finally:
        Fn.dropout_seed = orig


@pytest.mark.parametrize('R,N', [(131, 1000), (5, 1601), (1, 7)])
def test_kl_div_rows(K, R, N):
    """functional.kl_div_rows (dsrg_kl_rows_fwd / _bwd) against F.kl_div(F.log_softmax(x), t, 'none').sum(1) and its autograd;
    targets with exact zeros (xlogy(0, 0) = 0) and rows that do not sum to one."""
    from dsrg_amd import functional as Fn
    torch.manual_seed(R + N)
    x0 = torch.randn(R, N, device='cuda') * 3
    t = torch.softmax(torch.randn(R, N, device='cuda') * 2, -1)
    t[:, ::3] = 0.0
    t[0] *= 0.5
    g = torch.rand(R, device='cuda') + 0.1
    a, b = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    want = torch.nn.functional.kl_div(torch.log_softmax(a, -1), t, reduction='none').sum(1)
    got = Fn.kl_div_rows(b, t)
    assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    (want * g).sum().backward(); (got * g).sum().backward()
    assert (a.grad - b.grad).abs().max().item() <= 2e-6
    with torch.no_grad():
        assert torch.equal(Fn.kl_div_rows(x0, t), got.detach())


def test_wgrad_k_split_matches_single_launch(K):
    """functional._gemm_grads: dW of a small weight over thousands of rows runs K-split (S row slices as problems of one
    grouped trans = 3 launch with fp32 partial tiles + dsrg_sum_slabs); against the unsplit launch (fp32 accumulation order
    differs, so a bf16 ulp at most) and torch in fp32; the bias gradient too."""
    from dsrg_amd import functional as Fn
    for M, N, Kd in ((14796, 128, 128), (5000, 384, 128), (4100, 3072, 128 // 2 * 2)):
        dy, x = _r(M, N, dtype=torch.bfloat16), _r(M, Kd, seed=1, dtype=torch.bfloat16)
        w = _r(N, Kd, seed=2, dtype=torch.bfloat16)
        assert Fn._wgrad_splits(M, N, Kd) > 1
        _, dws, dbs = Fn._gemm_grads([(x, w, dy, False, True, True)])
        Fn.WGRAD_SPLIT = False
        try:
            _, dw1, db1 = Fn._gemm_grads([(x, w, dy, False, True, True)])
        finally:
            Fn.WGRAD_SPLIT = True
        ref = dy.float().T @ x.float()
        tol = 1e-2 * ref.abs().max().item()
        assert (dws[0].float() - ref).abs().max().item() <= tol
        assert (dws[0].float() - dw1[0].float()).abs().max().item() <= 2 * tol / 1e-2 * 2 ** -8
        refb = dy.float().sum(0)
        assert (dbs[0] - refb).abs().max().item() <= 1e-4 * max(1.0, refb.abs().max().item()) * math.sqrt(M)
        assert (dbs[0] - db1[0]).abs().max().item() <= 1e-3 * max(1.0, refb.abs().max().item())
    assert Fn._wgrad_splits(1920, 768, 768) == 1 and Fn._wgrad_splits(14796, 3072, 768) == 1
    # sum_slabs on its own, odd tail
    p = torch.randn(5, 1000, device='cuda')
    assert torch.allclose(K.sum_slabs(p), p.sum(0), atol=1e-5)
