"""gpu: dsrg_amd.optim.AdamW (dsrg_grad_sumsq / dsrg_adamw_prepare / dsrg_adamw_step through the C-ABI) against
tests/golden/adamw.npz - the reference's own AdamW class + torch's clip_grad_norm_, run by tools/make_golden_adamw.py -
and against the numpy oracle at a size beyond the fixture.  fp32 arithmetic; tolerance 2e-6 relative (fused multiply-add
contraction and the order of the norm's partial sums are the only differences)."""
import numpy as np
import pytest
import torch

from oracle import adamw_oracle
from tests.test_oracle_adamw_golden import GOLD, CASES, schedule
from tools.make_golden_adamw import DECAY, STEPS, tensors

pytestmark = pytest.mark.gpu


def _close(a, want, floor=1.0):
    return np.abs(a - want).max() <= 2e-6 * max(floor, np.abs(want).max())


@pytest.mark.parametrize('misalign', [False, True])
def test_adamw_matches_reference_golden(misalign):
    from dsrg_amd.optim import AdamW
    gold = np.load(GOLD)
    for name, kw, wd, max_norm in CASES:
        hold = []

        def dev(a):                       # misalign: 4-byte aligned storage only -> the kernels' scalar path
            if not misalign:
                return torch.from_numpy(a.copy()).cuda()
            buf = torch.empty(a.size + 1, dtype=torch.float32, device='cuda'); hold.append(buf)
            v = buf[1:].view(a.shape); v.copy_(torch.from_numpy(a)); return v
        ps = [torch.nn.Parameter(dev(a)) for a in tensors(1)]
        opt = AdamW([{'params': [p for p, d in zip(ps, DECAY) if d], 'weight_decay': wd},
                     {'params': [p for p, d in zip(ps, DECAY) if not d], 'weight_decay': 0.0}], lr=1e-3, **kw)
        for step in range(STEPS):
            lr, gs = schedule(step)
            for grp in opt.param_groups:
                grp['lr'] = lr
            for p, g in zip(ps, gs):
                p.grad = None if g is None else dev(g)
            if max_norm > 0:
                norm = float(opt.clip_grad_norm(max_norm))
                want = float(gold['%s_norm_%d' % (name, step)][0])
                assert abs(norm - want) <= 2e-6 * want
            opt.step()
            if step in (1, STEPS - 1):
                for i, p in enumerate(ps):
                    assert _close(p.detach().cpu().numpy(), gold['%s_p_%d_%d' % (name, step, i)]), (name, step, i)
        for i, p in enumerate(ps):
            assert _close(opt.state[p]['exp_avg'].cpu().numpy(), gold['%s_m_%d' % (name, i)], 1e-3)
            assert _close(opt.state[p]['exp_avg_sq'].cpu().numpy(), gold['%s_v_%d' % (name, i)], 1e-3)
        steps = [int(opt.state[p]['step']) for p in ps]
        assert steps == [STEPS, STEPS - 1, STEPS, STEPS, STEPS - 1, STEPS]          # tensors 1 and 4 sat out one step


def test_adamw_large_against_oracle_and_grad_scale():
    """3.2 M elements over ~800 chunks (the block table at work), grad_scale (the 1 / world size of a summed all-reduce)"""
    from dsrg_amd.optim import AdamW
    rng = np.random.default_rng(3)
    shapes = [(1000, 768), (3072, 768), (768,), (30522,), (17, 1)]
    p0 = [rng.standard_normal(s).astype(np.float32) * 0.05 for s in shapes]
    grp = [0, 0, 1, 1, 0]
    ps = [torch.nn.Parameter(torch.from_numpy(a.copy()).cuda()) for a in p0]
    opt = AdamW([{'params': [p for p, g in zip(ps, grp) if g == 0], 'weight_decay': 0.01},
                 {'params': [p for p, g in zip(ps, grp) if g == 1], 'weight_decay': 0.0}], lr=5e-5, betas=(0.9, 0.98))
    ora = adamw_oracle.AdamW([a.copy() for a in p0], grp, [5e-5, 5e-5], [0.01, 0.0], betas=(0.9, 0.98))
    opt.grad_scale = 0.5
    for step in range(3):
        gs = [rng.standard_normal(s).astype(np.float32) * (0.2 if step == 1 else 0.001) for s in shapes]
        for p, g in zip(ps, gs):
            p.grad = torch.from_numpy(g).cuda()
        norm = float(opt.clip_grad_norm(5.0))
        opt.step()
        want_norm, clipped = adamw_oracle.clip_grad_norm([(g * np.float32(0.5)).astype(np.float32) for g in gs], 5.0)
        assert abs(norm - want_norm) <= 3e-6 * want_norm
        assert (want_norm > 5.0) == (step == 1)
        ora.step(clipped)
        # the update must not have touched the gradients
        assert torch.equal(ps[0].grad, torch.from_numpy(gs[0]).cuda())
    for p, w in zip(ps, ora.p):
        assert np.abs(p.detach().cpu().numpy() - w).max() <= 3e-6 * max(1.0, np.abs(w).max())


@pytest.mark.parametrize('n', [4096 * 3, 5000, 37])
def test_adamw_writes_the_bf16_pack_copy(n):
    """dsrg_opt_tensor_t::pack: the update kernel also stores the new parameter rounded to bf16 (what attach_packs wires up)"""
    from dsrg_amd.optim import AdamW
    p = torch.nn.Parameter(torch.randn(n, device='cuda'))
    q = torch.nn.Parameter(torch.randn(64, device='cuda'))
    opt = AdamW([p, q], lr=1e-2)
    buf = torch.zeros(n + 8, dtype=torch.bfloat16, device='cuda')
    opt._pack_dst = {p.data_ptr(): buf.data_ptr() + 8}            # 8-byte aligned only
    opt._pack_gen += 1
    for _ in range(2):
        p.grad, q.grad = torch.randn_like(p), torch.randn_like(q)
        opt.clip_grad_norm(1.0)
        opt.step()
        assert torch.equal(buf[4:4 + n], p.detach().to(torch.bfloat16))
    assert float(buf[:4].abs().max()) == 0.0 and float(buf[4 + n:].abs().max()) == 0.0


def test_adamw_validation_and_capture():
    from dsrg_amd.optim import AdamW
    w = torch.nn.Parameter(torch.zeros(8, device='cuda'))
    for bad in (dict(lr=-1.0), dict(betas=(1.0, 0.9)), dict(betas=(0.9, 1.0)), dict(eps=-1.0)):
        with pytest.raises(ValueError):
            AdamW([w], **bad)
    h = torch.nn.Parameter(torch.zeros(8, device='cuda', dtype=torch.bfloat16))
    h.grad = torch.ones_like(h)
    with pytest.raises(Exception):
        AdamW([h]).step()
    # captured: the step replays with the lr the device word holds at replay time
    p = torch.nn.Parameter(torch.ones(5000, device='cuda'))
    q = torch.nn.Parameter(torch.ones(5000, device='cuda'))
    oa, ob = AdamW([p], lr=1e-2, correct_bias=False), AdamW([q], lr=1e-2, correct_bias=False)
    g = torch.full((5000,), 0.5, device='cuda')
    p.grad, q.grad = g.clone(), g.clone()
    for o in (oa, ob):                    # one eager step each: creates the moments (a capture may not)
        o.clip_grad_norm(1.0); o.step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            ob.clip_grad_norm(1.0)
            ob.step()
        for lr in (1e-2, 5e-3, 1e-3):
            for o in (oa, ob):
                o.param_groups[0]['lr'] = lr
            oa.clip_grad_norm(1.0); oa.step()
            ob.sync_hyper(); graph.replay()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert torch.equal(p.detach(), q.detach()) and float(p.detach()[0]) < 1.0
    assert int(oa.state[p]['step']) == 4 and int(ob.state[q]['step']) == 4
