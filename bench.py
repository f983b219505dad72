#!/usr/bin/env python3
"""bench.py — nav-policy forward steps/sec on MI355X (BASELINE.json metric, config[1]: R2R nav-step, bf16, bs=64).

One step = one 'panorama' + one 'navigation' call on a batch of 64 episodes (synthetic R2R shapes: L=80, 36 views,
3 objv2 objects/view, 20 graph nodes), inputs resident in HBM, RMF memory fed back step to step.  Every 15th step
(max_action_len, a new episode batch) the instruction encoding changes and the hoisted text K/V panel is recomputed
inside the timed region, as a rollout would.  N>1: one process per GPU (torchrun), independent replicas of 64
episodes each (weak scaling, no data-path collective); value = N*64*K / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5|c1|c4|ft] [--dtype bf16|fp32] [--eager]

--config c4 is BASELINE config[3]: one R2R pre-training optimisation step (MLM / MRC / SAP round-robin, forward +
backward + gradient clipping + AdamW, 24 trajectories per GPU, DDP gradient all-reduce over RCCL when N>1).
--config ft is one R2R fine-tuning optimisation step (two rollouts, one backward, clip 40, AdamW; agent_base.py:180-216).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (GPU_MAX_HW_QUEUES is left at HIP's default of 4: 8 bought 0.7 % with two concurrent chains, but with the three-chain
# schedule of this build any non-default value costs 35-110 %, profiles/r03t_ab_hw_queues.txt)

import numpy as np
import torch
import torch.distributed as dist

MFMA_PEAK_BF16 = 2.5e15   # dense bf16 MFMA FLOP/s, MI355X (MI355X_MICROARCH.md "Chip-level parameters")
MFMA_PEAK_F32 = 157.3e12
EPISODE_LEN = 15          # max_action_len of the reference rollout (agent.py:434)

CONFIGS = {   # name -> (reverie, B, L, G, O, No)
    'c1': (False, 8, 80, 20, 3, 0),
    'c2': (False, 64, 80, 20, 3, 0),
    'c3': (True, 64, 80, 20, 3, 20),
    'c5': (False, 64, 250, 100, 3, 0),
}


def T(b, dev):
    return {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in b.items()}


def nav_step_flops(L, V, O, G, Vp, og=False, hoisted_kv=True):
    """Algorithmic FLOPs of one episode-nav-step (SURVEY.md section 8d formula, 2*MAC)."""
    H, d, FF = 768, 128, 3072
    H2 = H * H

    def xlayer(N):
        f = 28 * N * H2 + 4 * N * L * H + 4 * N * N * H
        return f if hoisted_kv else f + 4 * L * H2
    nav = sum(4 * xlayer(N) for N in (G, Vp)) + 2 * (G + Vp + (Vp if og else 0)) * H2 + 4 * H2 + 6 * H2 + 6 * H2 \
        + 2 * (7 * G + 14 * Vp) * H
    n = V * O
    T_ = Vp - 2
    pano = 2 * V * H2 + 2 * (8 * n * d * d + 4 * n * d * FF + 4 * n * n * d) + 2 * (24 * T_ * H2 + 4 * T_ * T_ * H) + 2 * V * (H + d) * H
    return nav + pano


def build(cfg_name, dtype, dev):
    from dsrg_amd import synth
    from dsrg_amd.config import DSRGConfig
    reverie, B, L, G, O, No = CONFIGS[cfg_name]
    if reverie:
        from dsrg_amd.vilmodel_reverie import GlocalTextPathNavCMT
    else:
        from dsrg_amd.vilmodel import GlocalTextPathNavCMT
    cfg = DSRGConfig()
    model = GlocalTextPathNavCMT(cfg, compute_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes).items()}
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    lbs = [synth.language_batch(B, L, 100 + e) for e in range(2)]
    pb = synth.panorama_batch(B, 36, O, 100, reverie=reverie, No=No)
    Vp = pb['nav_types'].shape[1] + 2
    vlens = (36 + pb['obj_lens'] + 2) if reverie else None
    nb = synth.navigation_batch(B, G, Vp, 100, ncand=pb['ncand'], nav_types=pb['nav_types'], vp_lens=vlens,
                                reverie=reverie, all_valid=(cfg_name == 'c5'))
    return model, cfg, sd, lbs, pb, nb, (reverie, B, L, G, O, No, Vp)


def cpu_baseline(cfg, sd, lb, pb, nb, dims, budget_s=20.0):
    """The CPU oracle (a restatement of the reference's PyTorch path, oracle/dsrg_oracle.py) timed on the host cores
    on the same workload: bounded sample of nav-steps at the full batch size."""
    from oracle import dsrg_oracle
    reverie, B = dims[0], dims[1]
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    l = T(lb, 'cpu')
    with torch.no_grad():
        txt = dsrg_oracle.forward(sd, cfg, 'language', l, reverie)
        p, n = T(pb, 'cpu'), T(nb, 'cpu')
        n.update(txt_embeds=txt, txt_masks=l['txt_masks'])

        def step():
            pe, pm, pf = dsrg_oracle.forward(sd, cfg, 'panorama', p, reverie)
            n['vp_img_embeds'] = torch.cat([torch.zeros_like(pe[:, :2]), pe], 1)
            n['gmap_img_embeds'][:, 2] = pf
            return dsrg_oracle.forward(sd, cfg, 'navigation', n, reverie)
        step()
        t0, it = time.perf_counter(), 0
        while it < 3 or (time.perf_counter() - t0 < budget_s and it < 20):
            step()
            it += 1
        dt = time.perf_counter() - t0
    return {'value': B * it / dt, 'unit': 'episode-nav-steps/s', 'cores': cores, 'kind': 'port',
            'sample': '%d full nav-steps (panorama+navigation, B=%d, fp32, torch %s CPU, %d threads), %.1f s'
                      % (it, B, torch.__version__, cores, dt)}


def pmc_traffic(kernel_sig, grids, cfg='c2'):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_fetch.csv,
    *_pmc_write.csv; separate --pmc runs of this same bench, summarised by tools/pmc_summary.py).  FETCH_SIZE and
    WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so it is doubled (MI355X_MICROARCH.md,
    HBM section).  Averaged over the launches of one step, matched by kernel template + grid size.  None if absent."""
    import csv, glob
    here = os.path.dirname(os.path.abspath(__file__))
    tag = '' if cfg == 'c2' else cfg + '_'          # profiles/rNN_pmc_fetch.csv (c2) / rNN_c4_pmc_fetch.csv (pre-training step)
    import re
    pick = lambda kind: sorted(f for f in glob.glob(os.path.join(here, 'profiles', 'r*_%spmc_%s.csv' % (tag, kind)))
                               if re.fullmatch(r'r\d+[a-z]*_%spmc_%s\.csv' % (tag, kind), os.path.basename(f)))
    fetch, write = pick('fetch'), pick('write')
    if not fetch or not write:
        return None, None

    def table(path, counter):
        t = {}
        for r in csv.DictReader(open(path)):
            if r['Counter'] == counter and kernel_sig in r['Kernel']:
                t[int(r['GridSize'])] = float(r['AvgValue'])
        return t
    f, w = table(fetch[-1], 'FETCH_SIZE'), table(write[-1], 'WRITE_SIZE')
    tot = 0.0
    for g in grids:
        if g not in f or g not in w:
            return None, None
        tot += (2.0 * f[g] + w[g]) * 1024.0
    return tot / len(grids), os.path.basename(fetch[-1])


def profile_avg_us(kernel_sig, cfg='c2'):
    """Average in-step duration (us) of a kernel variant from the newest committed `rocprofv3 --kernel-trace --stats`
    summary of this bench (profiles/*_bench_c2_kernel_stats.csv: Name, Calls, TotalDurationNs, AverageNs ...), or None."""
    import csv, glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, 'profiles', 'r*_bench_%s_kernel_stats.csv' % cfg)))
    if not files:
        return None, None
    for r in csv.DictReader(open(files[-1])):
        if kernel_sig in r.get('Name', ''):
            return float(r['AverageNs']) * 1e-3, os.path.basename(files[-1])
    return None, os.path.basename(files[-1])


def gemm_roofline(step_fn, dtype, cfg='c2'):
    """Dominant kernel = the grouped NT GEMM.  Record every dsrg_gemm_nt launch of ONE eager nav-step and time, per
    kernel variant (tile, ring depth, output type), its launches as they run inside a step: all launches of the variant,
    in step order, each on its own operands (so A / W / residual come from beyond the XCD L2s, as in the step), captured
    in a HIP graph on the launch stream and replayed between HIP events.  achieved = algorithmic FLOPs of the dominant
    variant's launches / that time; the per-launch figure includes the ~1.3 us dependent-launch boundary.  `frac_isolated`
    is the same launches each replayed 20x back to back on hot operands (round 1's method, reads 30-50 % higher)."""
    from dsrg_amd import kernels
    kernels.GEMM_TRACE = []
    step_fn()
    trace, kernels.GEMM_TRACE = kernels.GEMM_TRACE, None
    torch.cuda.synchronize()
    by_tile = {}
    for rec in trace:
        d = by_tile.setdefault((rec['tile'], rec['odt'], rec.get('fuse', False), rec.get('trans', 0)), {'flops': 0.0, 'time': 0.0, 'hot': 0.0, 'launches': 0, 'grids': [], 'bytes': 0.0, 'recs': []})
        d['flops'] += rec['flops']; d['launches'] += 1; d['grids'].append(rec['wgs'] * 256); d['recs'].append(rec)
        esz, osz = (2 if rec['dt'] == 1 else 4), (4 if rec['odt'] == 0 else 2)
        d['bytes'] += sum((a.M * a.K + a.N * a.K) * esz + a.M * a.N * osz + (a.M * a.N * esz if a.residual else 0) for a in rec['arr'])
    st = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        for d in by_tile.values():
            for rec in d['recs']:
                kernels.gemm_replay(rec)
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for rec in d['recs']:
                    kernels.gemm_replay(rec)
            g.replay(); st.synchronize()
            rep = 20
            e0.record(st)
            for _ in range(rep):
                g.replay()
            e1.record(st)
            e1.synchronize()
            d['time'] = e0.elapsed_time(e1) * 1e-3 / rep
            for rec in d['recs']:                       # round 1's figure: each launch 20x back to back, hot operands
                kernels.gemm_replay(rec)
                e0.record(st)
                for _ in range(20):
                    kernels.gemm_replay(rec)
                e1.record(st)
                e1.synchronize()
                d['hot'] += e0.elapsed_time(e1) * 1e-3 / 20
    dom = max(by_tile, key=lambda k: by_tile[k]['time'])
    d = by_tile[dom]
    (bm, bn, nst), odt, fused, trans = dom
    peak = MFMA_PEAK_BF16 if dtype == torch.bfloat16 else MFMA_PEAK_F32
    tot_f = sum(v['flops'] for v in by_tile.values()); tot_t = sum(v['time'] for v in by_tile.values())
    sig = 'gemm_nt_kernel<%s, %s, %d, %d, %d, %d, %d, 2, %d>' % ('unsigned short' if dtype == torch.bfloat16 else 'float',
                                                                  'float' if odt == 0 else 'unsigned short', bm, bn, nst, int(fused),
                                                                  4 if bm == 256 else 2, trans)
    traffic, src = pmc_traffic(sig, d['grids'], cfg)
    prof_us, prof_src = profile_avg_us(sig, cfg)
    fpl = d['flops'] / d['launches']
    return {'bound': 'mfma', 'achieved': d['flops'] / d['time'] / 1e12, 'peak': peak / 1e12, 'unit': 'TFLOP/s',
            'frac': d['flops'] / d['time'] / peak, 'frac_isolated': d['flops'] / d['hot'] / peak,
            'frac_profile': (fpl / (prof_us * 1e-6) / peak) if prof_us else None, 'profile_avg_launch_us': prof_us, 'profile_source': prof_src,
            'traffic': traffic, 'traffic_unit': 'bytes/launch (2*FETCH_SIZE + WRITE_SIZE)',
            'traffic_source': src, 'algorithmic_bytes_per_launch': d['bytes'] / d['launches'],
            'algorithmic_flops_per_launch': fpl,
            'kernel': sig, 'ln_fused_epilogue': {0: 'none', 1: 'a_stats (consumer)', 2: 'r_stats / stats_out (producer)'}[int(fused)], 'operand_form': {0: 'NT', 1: 'TN (A stored [K,M])', 2: 'NN (dgrad)', 3: 'TN (wgrad)'}[trans], 'launches_per_step': d['launches'], 'avg_launch_us': d['time'] / d['launches'] * 1e6,
            'avg_launch_us_isolated': d['hot'] / d['launches'] * 1e6,
            'timing': 'HIP events around a graph of the variant\'s launches in step order, operands cold (as in the step)',
            'all_gemm_tflops': tot_f / tot_t / 1e12, 'all_gemm_launches_per_step': len(trace),
            'all_gemm_us_per_step': tot_t * 1e6,
            'variants': {'%dx%dx%d/%s%s%s' % (k[0] + ('f32' if k[1] == 0 else 'bf16', {0: '', 1: '/ln-a_stats', 2: '/ln-r_stats+stats_out'}[int(k[2])], {0: '', 1: '/tn', 2: '/nn-dgrad', 3: '/tn-wgrad'}[k[3]])): {'launches': v['launches'], 'us': v['time'] * 1e6, 'tflops': v['flops'] / v['time'] / 1e12}
                         for k, v in by_tile.items()}}


PT_TASKS = ('mlm', 'mrc', 'sap')      # mix of PT/config/r2r_pretrain.json, one task per optimisation step
PT_B, PT_L = 24, 80                   # trajectories per GPU, instruction length


def pretrain_main(a, dtype, dev, rank, world):
    """BASELINE config[3]: pre-training step = forward + backward (+ DDP all-reduce) + clip_grad_norm_(5.0) + AdamW, as
    PT/train_r2r.py:239-295.  Dropout is on (p=0.1, the reference's model config).  Weak scaling: 24 trajectories per GPU."""
    from dsrg_amd import synth, distributed as D
    from dsrg_amd.config import DSRGConfig
    from dsrg_amd.pretrain import GlocalTextPathCMTPreTraining
    cfg = DSRGConfig(use_lang2visn_attn=True, obj_feat_size=0, pretrain_tasks=PT_TASKS)
    model = GlocalTextPathCMTPreTraining(cfg, compute_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes).items()}
    model.load_state_dict(sd)
    model = model.to(dev).train()
    graphed = not a.eager            # the step as replayed HIP graphs (dsrg_amd.train_graph); --eager: the Python step under DDP
    ddp = model if graphed else D.wrap_model(model, dev, bf16_grads=(dtype == torch.bfloat16))
    decay = [p for n, p in model.named_parameters() if p.requires_grad and not any(t in n for t in ('bias', 'LayerNorm'))]
    nodecay = [p for n, p in model.named_parameters() if p.requires_grad and any(t in n for t in ('bias', 'LayerNorm'))]
    groups = [{'params': decay, 'weight_decay': 0.01}, {'params': nodecay, 'weight_decay': 0.0}]       # optim/misc.py:9-19
    if a.torch_optim:         # A/B: torch's fused AdamW + clip_grad_norm_ (multi-tensor ATen kernels)
        opt = torch.optim.AdamW(groups, lr=5e-5, betas=(0.9, 0.98), fused=True, capturable=graphed)
    else:                     # the reference's AdamW + clip in three HIP launches (dsrg_amd/optim.py, csrc/optim.hip)
        from dsrg_amd.optim import AdamW
        opt = AdamW(groups, lr=5e-5, betas=(0.9, 0.98))
    batches = {t: T(synth.pretrain_batch(PT_B, PT_L, 3, 200 + rank, t, min_steps=4, max_steps=7), dev) for t in PT_TASKS}
    last = {}
    params = [p for p in model.parameters() if p.requires_grad]      # walking the module tree costs ~1.4 ms of host time per step
    if graphed:
        from dsrg_amd.train_graph import GraphedTrainStep
        gs = GraphedTrainStep(model, opt, 5.0, params=params, bf16_grads=(dtype == torch.bfloat16))
        # one eager step + one capture per task (untimed); the timed steps hand the captured batches back: inputs resident
        # in HBM in the buffers the step reads, as in the eager bench
        static = {t: gs.static_batch(batches[t], t) for t in PT_TASKS}

    def step(i):
        task = PT_TASKS[i % len(PT_TASKS)]
        if world > 1:
            task = PT_TASKS[D.broadcast_task_id(PT_TASKS.index(task), dev)]
        if graphed:
            last[task] = gs.step(static[task], task)
            return
        dbg = os.environ.get('DSRG_BENCH_HANG_DUMP')
        t_ = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = ddp(batches[task], task, compute_loss=True).mean()
        if dbg:
            torch.cuda.synchronize(); print('rank %d step %d %s forward %.2f s' % (rank, i, task, time.perf_counter() - t_), file=sys.stderr, flush=True)
        loss.backward()
        if dbg:
            torch.cuda.synchronize(); print('rank %d step %d backward done %.2f s' % (rank, i, time.perf_counter() - t_), file=sys.stderr, flush=True)
        if a.torch_optim:
            torch.nn.utils.clip_grad_norm_(params, 5.0)
        else:
            opt.clip_grad_norm(5.0)
        opt.step()
        last[task] = loss.detach()

    # main thread + autograd thread on one core (before the first backward creates the engine thread): D.pin_step_threads.
    # The intra-op pool of the CPU baseline leg is created FIRST: threads born under the pin would inherit the one-core mask.
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.mm(torch.ones(256, 256), torch.ones(256, 256))
    prev_mask = None if a.no_pin else D.pin_step_threads(int(os.environ.get('LOCAL_RANK', rank)), int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prev_mask:
        os.sched_setaffinity(0, prev_mask)          # the CPU baseline below wants all its threads
    evidence = None
    if world > 1:
        dt, evidence = rank_evidence(dt, dev, world)
    assert all(torch.isfinite(v) for v in last.values()), 'non-finite pre-training loss'
    if rank == 0:
        nsteps = {t: int(batches[t]['traj_view_img_fts'].shape[0]) for t in PT_TASKS}
        out = {'metric': 'R2R pre-training optimisation steps (MLM+MRC+SAP round-robin): trajectories/s', 'value': world * PT_B * a.steps / dt,
               'unit': 'trajectories/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
               'config': {'workload': 'BASELINE config[3]: R2R pre-training step, forward+backward+clip+AdamW, %s' % a.dtype, 'name': 'c4',
                          'step': ('HIP graph replay per task (forward+backward+clip+AdamW captured once; %s)' % (
                                       'one graph' if world == 1 else 'two graphs around one flat gradient all-reduce')) if graphed else 'eager Python step',
                          'optimizer': 'torch.optim.AdamW(fused) + clip_grad_norm_' if a.torch_optim else 'dsrg_amd.optim.AdamW (reference AdamW semantics) + clip, 3 HIP launches',
                          'host_threads': 'main + autograd thread on their own free cores' if a.no_pin else 'main + autograd thread pinned to one core',
                          'trajectories_per_gpu': PT_B, 'txt_len': PT_L, 'trajectory_steps_total': nsteps, 'dropout': cfg.hidden_dropout_prob,
                          'parallelism': ('DDP x%d (RCCL all-reduce, bf16 gradient buckets)' % world if dist.get_backend() == 'nccl' else
                                          'DDP x%d (rehearsal: gloo, gradient buckets staged through host memory)' % world) if world > 1 else 'single GPU'},
               'final_losses': {t: float(v) for t, v in last.items()}}
        if evidence:
            out['ranks_seen'], out['backend'] = evidence['ranks_seen'], evidence['backend']
            out['per_rank_ms'] = [t / a.steps * 1e3 for t in evidence['per_rank_s']]
        if not a.no_roofline:
            # DDP hooks must not fire during the replayed step: run it on the bare module
            def one_step():
                opt.zero_grad(set_to_none=True)
                model(batches['sap'], 'sap', compute_loss=True).mean().backward()
            out['roofline'] = gemm_roofline(one_step, dtype, 'c4')
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = pretrain_cpu_baseline(cfg, sd)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pretrain_cpu_baseline(cfg, sd, budget_s=25.0):
    """CPU oracle (oracle/pretrain_oracle.py) forward+backward on a bounded sample: 4 trajectories per step."""
    from dsrg_amd import synth
    from oracle import pretrain_oracle
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd = {k: v.clone().float().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    Bc = 4
    bs = {t: T(synth.pretrain_batch(Bc, PT_L, 3, 200, t, min_steps=4, max_steps=7), 'cpu') for t in PT_TASKS}
    import copy
    c0 = copy.copy(cfg); c0.hidden_dropout_prob = 0.0; c0.attention_probs_dropout_prob = 0.0
    t0, it = time.perf_counter(), 0
    while it < len(PT_TASKS) or (time.perf_counter() - t0 < budget_s and it < 12):
        task = PT_TASKS[it % len(PT_TASKS)]
        pretrain_oracle.forward(sd, c0, bs[task], task, True).mean().backward()
        for v in sd.values():
            v.grad = None
        it += 1
    dt = time.perf_counter() - t0
    return {'value': Bc * it / dt, 'unit': 'trajectories/s', 'cores': cores, 'kind': 'port',
            'sample': '%d forward+backward steps of %d trajectories (no optimiser step, fp32, %d threads), %.1f s' % (it, Bc, cores, dt)}



FT_B, FT_T = 8, 7          # episodes per GPU (the reference fine-tunes REVERIE with bs 8, R2R with 4: scripts/run_*.sh); steps per rollout


class _RolloutLoss(torch.nn.Module):
    """One fine-tuning iteration's loss (finetune_src/r2r/agent_base.py:180-216, agent.py:396-578): two rollouts - teacher
    forcing weighted 0.2, sampling weighted 1.0 - each language -> T x (panorama -> navigation) with the RMF memory and the
    panorama embeddings carried inside the autograd graph, CrossEntropy(sum, ignore_index=-100) / B on the fused logits of
    every step.  One nn.Module so that DDP sees ONE forward per backward."""

    def __init__(self, model, feat_dropout=0.4):
        super().__init__()
        self.model, self.feat_dropout = model, feat_dropout

    def rollout(self, lb, pbs, nbs, targets):
        from dsrg_amd import functional as Fn
        m = self.model
        txt = m('language', lb)
        B = txt.shape[0]
        cls, loss = None, 0.0
        for pb, nb0, tg in zip(pbs, nbs, targets):
            pb = dict(pb)
            pb['view_img_fts'] = torch.nn.functional.dropout(pb['view_img_fts'], self.feat_dropout, self.training)   # VLNBert.drop_env (model.py:29)
            pe, pm, pf = m('panorama', pb)
            mem = torch.zeros(B, pe.shape[-1], device=pe.device, dtype=pe.dtype) if cls is None else cls
            nb = dict(nb0)
            nb['vp_img_embeds'] = torch.cat([torch.zeros_like(mem)[:, None], mem[:, None], pe], 1)                    # agent.py:231-237
            g = nb0['gmap_img_embeds']
            nb['gmap_img_embeds'] = torch.cat([g[:, :1], mem[:, None], pf[:, None], g[:, 3:]], 1)                     # agent.py:185-188, 443-447
            nb['txt_embeds'], nb['txt_masks'] = txt, lb['txt_masks']
            o = m('navigation', nb)
            cls = o['cls_embeds']
            loss = loss + Fn.cross_entropy_rows(o['fused_logits'], tg).sum() / B                                      # agent.py:502, agent_base.py:150
        return loss

    def forward(self, lb, pbs, nbs, targets):
        return 0.2 * self.rollout(lb, pbs, nbs, targets[0]) + 1.0 * self.rollout(lb, pbs, nbs, targets[1])


def finetune_batches(B, nsteps, L, G, dev, dtype, model, seed=300):
    """Synthetic rollout inputs, resident on the device: instruction, T panoramas, T graph-map states (fusion indices
    precomputed on the host as the device-resident graph map does), and teacher targets = a valid node."""
    from dsrg_amd import synth
    lb = T(synth.language_batch(B, L, seed), dev)
    pbs, nbs, tg = [], [], [[], []]
    for t in range(nsteps):
        pbn = synth.panorama_batch(B, 36, 3, seed + 10 * t)
        nbn = synth.navigation_batch(B, G, 38, seed + 10 * t, ncand=pbn['ncand'], nav_types=pbn['nav_types'])
        src, bw = model.fusion_indices(nbn['gmap_vpids'], nbn['gmap_visited_masks'], nbn['vp_cand_vpids'], G, 38, 2)
        pb, nb = T(pbn, dev), T(nbn, dev)
        nb['gmap_src'], nb['bw_mask'] = torch.from_numpy(src).to(dev), torch.from_numpy(bw).to(dev)
        nb['gmap_img_embeds'] = nb['gmap_img_embeds'].to(dtype)
        ok = nbn['gmap_masks'] & ~nbn['gmap_visited_masks']             # nodes whose fused logit is finite
        ok[:, 0] = True
        for r in range(2):
            rng = np.random.RandomState(seed + 97 * t + r)
            pick = np.array([rng.choice(np.nonzero(ok[i])[0]) for i in range(B)], np.int64)
            pick[rng.rand(B) < 0.15] = -100                              # ended episodes are ignored by the loss (agent.py:497-500)
            tg[r].append(torch.from_numpy(pick).to(dev))
        pbs.append(pb); nbs.append(nb)
    return lb, pbs, nbs, tg


def finetune_main(a, dtype, dev, rank, world):
    """`--config ft`: one fine-tuning optimisation step of the R2R agent's policy network = two rollouts (forward with
    autograd), ONE backward, clip_grad_norm_(40), AdamW - finetune_src/r2r/agent_base.py:180-216; dropout as in training."""
    from dsrg_amd import synth, distributed as D
    from dsrg_amd.config import DSRGConfig
    from dsrg_amd.vilmodel import GlocalTextPathNavCMT
    B, T, L, G = a.ft_batch, FT_T, 80, 20
    cfg = DSRGConfig()
    model = GlocalTextPathNavCMT(cfg, compute_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes).items()}
    model.load_state_dict(sd)
    model = model.to(dev).train()
    step_mod = _RolloutLoss(model).train()
    ddp = D.wrap_model(step_mod, dev, bf16_grads=(dtype == torch.bfloat16))
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-5, fused=True)
    lb, pbs, nbs, tg = finetune_batches(B, T, L, G, dev, dtype, model, 300 + rank)
    last = {}

    def step(i):
        opt.zero_grad(set_to_none=True)
        loss = ddp(lb, pbs, nbs, tg)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 40.0)
        opt.step()
        last['loss'] = loss.detach()

    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.mm(torch.ones(256, 256), torch.ones(256, 256))       # the intra-op pool exists before the pin below: its threads keep the full mask
    prev_mask = None if a.no_pin else D.pin_step_threads(int(os.environ.get('LOCAL_RANK', rank)), int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prev_mask:
        os.sched_setaffinity(0, prev_mask)
    evidence = None
    if world > 1:
        dt, evidence = rank_evidence(dt, dev, world)
    assert torch.isfinite(last['loss']), 'non-finite fine-tuning loss'
    if rank == 0:
        out = {'metric': 'R2R fine-tuning optimisation steps (2 rollouts of %d nav-steps, 1 backward): episodes/s' % T,
               'value': world * B * a.steps / dt, 'unit': 'episodes/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
               'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype,
               'data': 'synthetic',
               'config': {'workload': 'R2R fine-tuning step (finetune_src/r2r/agent_base.py:180-216): teacher + sample rollout, CE(sum)/B, '
                                      'one backward, clip 40, AdamW, %s' % a.dtype, 'name': 'ft', 'episodes_per_gpu': B,
                          'nav_steps_per_rollout': T, 'txt_len': L, 'graph_nodes': G, 'dropout': cfg.hidden_dropout_prob, 'feat_dropout': 0.4,
                          'parallelism': ('DDP x%d (RCCL all-reduce, bf16 gradient buckets)' % world if dist.get_backend() == 'nccl' else
                                          'DDP x%d (rehearsal: gloo, gradient buckets staged through host memory)' % world) if world > 1 else 'single GPU'},
               'nav_steps_per_s': world * B * 2 * T * a.steps / dt, 'final_loss': float(last['loss'])}
        if evidence:
            out['ranks_seen'], out['backend'] = evidence['ranks_seen'], evidence['backend']
            out['per_rank_ms'] = [t / a.steps * 1e3 for t in evidence['per_rank_s']]
        if not a.no_roofline:
            def one_step():
                opt.zero_grad(set_to_none=True)
                step_mod(lb, pbs, nbs, tg).backward()
            out['roofline'] = gemm_roofline(one_step, dtype, 'ft')
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = finetune_cpu_baseline(cfg, sd)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def finetune_cpu_baseline(cfg, sd, budget_s=25.0):
    """CPU oracle (oracle/dsrg_oracle.py, autograd on) on a bounded sample of the same step: 2 episodes, the same two
    rollouts of FT_T nav-steps + backward (no optimiser step)."""
    from dsrg_amd import synth
    from oracle import dsrg_oracle
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sdg = {k: v.clone().float().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    Bc, Tc = 2, FT_T
    from dsrg_amd.vilmodel import GlocalTextPathNavCMT
    lb, pbs, nbs, tg = finetune_batches(Bc, Tc, 80, 20, 'cpu', torch.float32, GlocalTextPathNavCMT, 300)

    class _O:            # the oracle behind the model's call signature
        training = False

        def __call__(self, mode, b):
            return dsrg_oracle.forward(sdg, cfg, mode, b, grad=True)
    import types
    step_mod = _RolloutLoss.__new__(_RolloutLoss)
    torch.nn.Module.__init__(step_mod)
    object.__setattr__(step_mod, 'model', _O())
    step_mod.feat_dropout = 0.0

    def ce(lg, t):      # the reference's criterion (agent_base.py:150)
        return torch.nn.functional.cross_entropy(lg, t, ignore_index=-100, reduction='none')
    import dsrg_amd.functional as Fn
    real = Fn.cross_entropy_rows
    Fn.cross_entropy_rows = ce
    try:
        t0, it = time.perf_counter(), 0
        while it < 2 or (time.perf_counter() - t0 < budget_s and it < 12):
            step_mod(lb, pbs, nbs, tg).backward()
            for v in sdg.values():
                v.grad = None
            it += 1
        dtc = time.perf_counter() - t0
    finally:
        Fn.cross_entropy_rows = real
    return {'value': Bc * it / dtc, 'unit': 'episodes/s', 'cores': cores, 'kind': 'port',
            'sample': '%d forward+backward iterations of %d episodes x 2 rollouts x %d nav-steps (no optimiser step, fp32, %d threads), %.1f s'
                      % (it, Bc, Tc, cores, dtc)}


def rank_evidence(dt, dev, world):
    """What the live backend saw (N > 1): ranks_seen = an all-reduce of ones, per_rank_ms = an all-gather of every rank's
    own timed-region wall time; returns (max over ranks, fields for the JSON line)."""
    on_dev = dist.get_backend() == 'nccl'
    d = dev if on_dev else 'cpu'
    ones = torch.ones(1, device=d, dtype=torch.float64)
    dist.all_reduce(ones)
    mine = torch.tensor([dt], device=d, dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    per = [float(t.item()) for t in every]
    return max(per), {'ranks_seen': int(round(float(ones.item()))), 'backend': dist.get_backend(), 'per_rank_s': per}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def spawn_ranks(n, timeout_s=None):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes of this script, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what `python -m torch.distributed.run --nproc-per-node N` would do;
    the reference launches its ranks the same way, pretrain_src/run_r2r.sh:8-15).  The parent never touches the GPU;
    rank 0's stdout (the one JSON line) is forwarded.  Fail fast: every child is polled; as soon as one exits non-zero
    the others are terminated (a rank that died before a barrier would otherwise leave its peers waiting in RCCL until the
    driver's time limit) and the parent exits non-zero; an overall time limit (DSRG_BENCH_SPAWN_TIMEOUT seconds, default
    1500) does the same.  Children are always fresh processes - nothing that touched a GPU is ever re-exec'ed."""
    import subprocess, tempfile
    if timeout_s is None:
        timeout_s = float(os.environ.get('DSRG_BENCH_SPAWN_TIMEOUT', '1500'))
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT') or str(_free_port()),
               LOCAL_WORLD_SIZE=str(n))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs = []
    out0 = tempfile.TemporaryFile()               # rank 0's stdout: a file, so that polling never blocks on a full pipe
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, start_new_session=True))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    t0, why = time.monotonic(), None
    try:
        while True:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                why = 'rank %d exited with code %d' % bad[0]
                break
            if all(rc == 0 for rc in rcs):
                break
            if time.monotonic() - t0 > timeout_s:
                why = 'no result after %.0f s' % timeout_s
                break
            time.sleep(0.05)
    finally:
        if why is not None or any(p.poll() is None for p in procs):
            stop_all()
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    if why is not None:
        raise SystemExit('bench.py: %s; remaining ranks stopped (exit codes %s)' % (why, [p.returncode for p in procs]))


def spawn_selftest(rank, world):
    """Launch plumbing only (tests/test_dist_gloo.py, no GPU needed): rendezvous over gloo, one all-reduce, rank 0 prints
    the line's launch fields."""
    dist.init_process_group('gloo')
    t = torch.ones(1)
    dist.all_reduce(t)
    assert int(t.item()) == world
    die = os.environ.get('DSRG_BENCH_SELFTEST_DIE_RANK')
    if die is not None and int(die) == rank:          # test hook: this rank dies mid-run, its peers wait in the next collective
        os._exit(17)
    if die is not None:
        time.sleep(1.0)
    if rank == 0:
        print(json.dumps({'selftest': True, 'n_gpus': world, 'ranks_seen': int(t.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=150)
    ap.add_argument('--warmup', type=int, default=15)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS) + ['c4', 'ft'])
    ap.add_argument('--ft-batch', type=int, default=FT_B, help='ft: episodes per GPU')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--eager', action='store_true', help='launch kernels from Python each step instead of replaying a HIP graph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--torch-optim', action='store_true', help="c4: torch.optim.AdamW(fused) + clip_grad_norm_ instead of dsrg_amd.optim.AdamW (A/B)")
    ap.add_argument('--no-pin', action='store_true', help='c4: do not pin the main / autograd threads to one core (A/B)')
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if 'RANK' not in os.environ and a.gpus > 1:
        return spawn_ranks(a.gpus)            # before anything touches the GPU: the children are fresh processes
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if os.environ.get('DSRG_BENCH_HANG_DUMP'):          # debugging aid: Python stacks of every rank after N seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['DSRG_BENCH_HANG_DUMP']), exit=True)
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (a.gpus, world))
    if os.environ.get('DSRG_BENCH_SPAWN_SELFTEST') == '1':
        return spawn_selftest(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the DSRG hot path has no CPU fallback')
    # rehearsal aid (one-GPU box): DSRG_BENCH_BACKEND=gloo runs the N ranks on the visible device(s) round-robin
    backend = os.environ.get('DSRG_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    dtype = torch.bfloat16 if a.dtype == 'bf16' else torch.float32

    if a.config == 'c4':
        return pretrain_main(a, dtype, dev, rank, world)
    if a.config == 'ft':
        return finetune_main(a, dtype, dev, rank, world)
    from dsrg_amd.runner import GraphedNavStep
    model, cfg, sd, lbs, pb, nb, dims = build(a.config, dtype, dev)
    reverie, B, L, G, O, No, Vp = dims
    with torch.no_grad():
        txts = []
        for lb in lbs:
            l = T(lb, dev)
            txts.append((model('language', l), l['txt_masks']))
    runner = GraphedNavStep(model, txts[0][0], txts[0][1], T(pb, dev), T(nb, dev), use_graph=not a.eager, reverie=reverie)

    def run(nsteps, start):
        for s in range(start, start + nsteps):
            if s % EPISODE_LEN == 0:
                runner.set_text(*txts[(s // EPISODE_LEN) % 2])
            runner.step()

    run(a.warmup, 0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps, a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    evidence = None
    if world > 1:
        dt, evidence = rank_evidence(dt, dev, world)
    fused = runner.out['fused_logits']
    assert torch.isfinite(fused).any(1).all(), 'bench produced no finite logits'

    if rank == 0:
        fl = nav_step_flops(L, 36, O, G, Vp, og=reverie, hoisted_kv=True) + 4 * L * 768 * 768 * 8 / EPISODE_LEN
        out = {
            'metric': 'nav-policy forward steps/sec (R2R shapes, bs=64) + % bf16 MFMA peak on GEMMs',
            'value': world * B * a.steps / dt, 'unit': 'episode-nav-steps/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': 'BASELINE config[1]: R2R nav-step inference (panorama + navigation), %s' % a.dtype
                       if a.config == 'c2' else 'config %s' % a.config,
                       'name': a.config, 'episodes_per_gpu': B, 'txt_len': L, 'views': 36, 'objv2_per_view': O,
                       'graph_nodes': G, 'vp_tokens': Vp, 'reverie': reverie, 'launch': 'eager' if a.eager else 'hipgraph',
                       'text_kv': 'hoisted: one GEMM per episode batch (every %d steps), inside the timed region' % EPISODE_LEN,
                       'parallelism': 'replicas x%d (episode batch sharding, no collective)' % world},
            'algorithmic_gflop_per_episode_step': fl / 1e9,
            'whole_step_tflops': world * B * a.steps * fl / dt / 1e12,
            'whole_step_frac_of_mfma_peak': B * a.steps * fl / dt / (MFMA_PEAK_BF16 if dtype == torch.bfloat16 else MFMA_PEAK_F32),
        }
        if evidence:
            out['ranks_seen'], out['backend'] = evidence['ranks_seen'], evidence['backend']
            out['per_rank_ms'] = [t / a.steps * 1e3 for t in evidence['per_rank_s']]
        # language mode runs once per episode, outside the nav-step metric (SURVEY 8d asks for it separately): eager launches
        l0 = T(lbs[0], dev)
        with torch.no_grad():
            model('language', l0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                model('language', l0)
            torch.cuda.synchronize()
        out['language_mode'] = {'ms_per_call': (time.perf_counter() - t1) / 5 * 1e3, 'episodes_per_call': B, 'launch': 'eager'}
        if not a.no_roofline:
            def one_step():                      # the launches of the schedule the graph replays (two-stream branches)
                model.split_when_eager = not a.eager
                try:
                    with torch.no_grad():
                        runner._body()
                finally:
                    model.split_when_eager = False
            out['roofline'] = gemm_roofline(one_step, dtype, a.config)
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, sd, lbs[0], pb, nb, dims)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
