"""not-gpu: the N>1 path on CPU with gloo, world_size 2 (spawned processes, 127.0.0.1 rendezvous).
  * episode-batch sharding: each rank runs its slice (the CPU oracle stands in for the device compute, the sharding
    and gathering code is the product's), the gathered result equals the unsharded one bit for bit;
  * DDP wrapper: gradients equal the mean over ranks, with and without the bf16 compression hook, with a parameter
    that is unused on one rank (find_unused_parameters semantics);
  * task-id broadcast and max-over-ranks timing reduction used by bench.py."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from dsrg_amd import distributed as D
    D.init_distributed('gloo')
    try:
        ret[rank] = fn(rank, world, D)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    [p.start() for p in ps]
    [p.join(300) for p in ps]
    assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    return dict(ret)


def _shard_case(rank, world, D):
    from dsrg_amd import synth
    from dsrg_amd.config import DSRGConfig
    from oracle import dsrg_oracle
    from tests import common
    cfg = DSRGConfig(num_l_layers=1, num_x_layers=1, num_pano_layers=1)
    sd = common.synth_sd('r2r_c1')
    B = 5                                           # odd: ranks get 3 + 2 episodes
    T = lambda b: common.to_t(b)
    lb = synth.language_batch(B, 24, 3, min_len=8)
    nb = synth.navigation_batch(B, 12, 38, 3)

    def run(l, n):
        l, n = T(l), T(n)
        txt = dsrg_oracle.forward(sd, cfg, 'language', l)
        n.update(txt_embeds=txt, txt_masks=l['txt_masks'])
        return dsrg_oracle.forward(sd, cfg, 'navigation', n)['fused_logits'].numpy()
    mine = run(D.shard_batch(lb, rank, world, B), D.shard_batch(nb, rank, world, B))
    parts = D.all_gather_objects(mine)
    full = run(lb, nb)
    got = np.concatenate(parts, 0)
    tid = D.broadcast_task_id(7 if rank == 0 else 99)
    tmax = D.reduce_max(1.0 + rank)
    return bool((got == full).all() or np.allclose(got, full, atol=1e-6, equal_nan=True)), tid, tmax, mine.shape[0]


def test_episode_sharding_gather_and_broadcast():
    r = _run(_shard_case)
    assert r[0][0] and r[1][0]
    assert r[0][1] == 7 and r[1][1] == 7
    assert r[0][2] == 2.0 and r[1][2] == 2.0
    assert (r[0][3], r[1][3]) == (3, 2)


def _ddp_case(rank, world, D):
    torch.manual_seed(0)
    out = {}
    for hook in (False, True):
        net = torch.nn.ModuleDict({'a': torch.nn.Linear(8, 8), 'b': torch.nn.Linear(8, 8)})
        ddp = D.wrap_model(net, bf16_grads=hook, bucket_cap_mb=1)
        x = torch.full((4, 8), float(rank + 1))
        # 'b' is used on rank 0 only: DDP must still reduce it (find_unused_parameters, as the reference needs)
        class Wrap(torch.nn.Module):
            def __init__(s, m): super().__init__(); s.m = m
            def forward(s, x, use_b): return s.m['a'](x).sum() + (s.m['b'](x).sum() if use_b else 0.0)
        ddp = D.wrap_model(Wrap(net), bf16_grads=hook, bucket_cap_mb=1)
        ddp(x, rank == 0).backward()
        out[hook] = {k: v.grad.clone().numpy() for k, v in net.named_parameters()}
    return out


def test_ddp_wrapper_grad_mean_and_bf16_hook():
    r = _run(_ddp_case)
    for hook in (False, True):
        g0, g1 = r[0][hook], r[1][hook]
        for k in g0:
            assert np.array_equal(g0[k], g1[k]), k          # ranks agree after the all-reduce
        # d/dW sum(Wx+b) = 4 rows * x ; mean over ranks of x = (1+2)/2 -> 4*1.5 = 6 ; 'b' only on rank 0 -> 4*1/2 = 2
        assert np.allclose(g0['a.weight'], 6.0, rtol=1e-2 if hook else 1e-6)
        assert np.allclose(g0['b.weight'], 2.0, rtol=1e-2 if hook else 1e-6)


def test_shard_slice_covers_everything():
    from dsrg_amd.distributed import shard_slice
    for n in (0, 1, 7, 64, 512):
        for w in (1, 2, 3, 8):
            s = [shard_slice(n, r, w) for r in range(w)]
            assert s[0][0] == 0 and s[-1][1] == n and all(s[i][1] == s[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in s) - min(b - a for a, b in s) <= 1
