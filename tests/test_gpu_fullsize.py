"""gpu: parity at BASELINE.json's FULL sizes (configs[1] R2R B=64, configs[4] L=250 / G=100), where the golden
fixtures (B=8) do not reach:

  * the HIP fp32 path against the CPU oracle on the whole 64-episode recurrent chain (<= 1e-3 on logits, same argmax);
  * size-independent properties of the path: episodes are independent (a sub-batch gives the same rows), padding
    behind the masks changes nothing, HIP-graph replay (two streams) equals eager launches bit for bit, bf16 agrees
    with fp32 wherever the fp32 top-2 margin is clear.
The oracle (oracle/dsrg_oracle.py) is pinned against the reference's own golden vectors in test_oracle_golden.py."""
import numpy as np
import pytest
import torch

from dsrg_amd import synth
from tests import common

pytestmark = pytest.mark.gpu


def _model(dtype, sd=None):
    from dsrg_amd.config import DSRGConfig
    from dsrg_amd.vilmodel import GlocalTextPathNavCMT
    sd = sd or common.synth_sd('r2r_c1')
    return GlocalTextPathNavCMT.from_pretrained(None, config=DSRGConfig(), state_dict=sd, compute_dtype=dtype).cuda().eval()


def _step_inputs(B, L, G, seed, dev='cuda', O=3):
    lb = common.to_t(synth.language_batch(B, L, seed), dev)
    pbn = synth.panorama_batch(B, 36, O, seed)
    nb = common.to_t(synth.navigation_batch(B, G, 38, seed, ncand=pbn['ncand'], nav_types=pbn['nav_types']), dev)
    return lb, common.to_t(pbn, dev), nb


def _one_step(m, lb, pb, nb):
    """language -> panorama -> navigation of one time step, glued as agent.py:434-472 does (first step: [MEM] = 0)."""
    with torch.no_grad():
        txt = m('language', lb)
        pe, pm, pf = m('panorama', pb)
        n = dict(nb)
        B = pe.shape[0]
        z = torch.zeros(B, 1, pe.shape[2], device=pe.device, dtype=pe.dtype)
        n['vp_img_embeds'] = torch.cat([z, z, pe], 1)
        g = n['gmap_img_embeds'].to(pe.dtype).clone()
        g[:, 1] = 0
        g[:, 2] = pf
        n['gmap_img_embeds'] = g
        n['txt_embeds'], n['txt_masks'] = txt, lb['txt_masks']
        return m('navigation', n)


def _slice(b, n):
    return {k: (v[:n] if isinstance(v, (torch.Tensor, list)) else v) for k, v in b.items()}


def _same(a, b, tol):
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    assert (fa == fb).all()
    return float((a[fa].float() - b[fb].float()).abs().max()) <= tol


@pytest.mark.parametrize('L,G,B,reverie,O', [(80, 20, 64, False, 3), (250, 100, 32, False, 3), (80, 20, 64, True, 3),
                                             (80, 20, 16, False, 20), (80, 20, 16, True, 20)])
def test_full_size_chain_matches_oracle(L, G, B, reverie, O):
    """configs[1] and configs[2] (REVERIE, 20 object tokens, OG head) at full batch and the configs[4] shapes: recurrent
    chain, HIP fp32 vs the CPU oracle.  O=20 is configs[2]'s stated upper end ("<= 20 objects/view": 720 object tokens per
    episode through the ASV encoder, vilmodel.py:697-714, vilmodel_reverie.py:689-759) at model level, R2R and REVERIE."""
    from dsrg_amd.config import DSRGConfig
    from oracle import dsrg_oracle
    tag = 'reverie_c3' if reverie else 'r2r_c1'
    sd = common.synth_sd(tag)
    if reverie:
        from dsrg_amd.vilmodel_reverie import GlocalTextPathNavCMT
        m = GlocalTextPathNavCMT.from_pretrained(None, config=DSRGConfig(), state_dict=sd, compute_dtype=torch.float32).cuda().eval()
    else:
        m = _model(torch.float32, sd)
    meta = [B, L, G, O, 42, int(reverie), 20 if reverie else 0]
    steps = 3 if (L == 80 and O == 3) else 2
    with torch.no_grad():
        out = common.run_chain(lambda mode, b: m(mode, b), meta, device='cuda', steps=steps)
        ref = common.run_chain(lambda mode, b: dsrg_oracle.forward(sd, DSRGConfig(), mode, b, reverie=reverie), meta, device='cpu', steps=steps)
    errs = common.compare(out, ref, tol=1e-3, logit_tol=1e-3, steps=steps)
    print('B=%d L=%d G=%d O=%d reverie=%s fp32 vs oracle' % (B, L, G, O, reverie), {k: '%.1e' % v for k, v in errs.items() if 'logits' in k})


@pytest.mark.parametrize('dtype,L,G,O', [(torch.float32, 80, 20, 3), (torch.bfloat16, 80, 20, 3), (torch.float32, 250, 100, 3),
                                         (torch.bfloat16, 250, 100, 3), (torch.bfloat16, 80, 20, 20)])
def test_episodes_are_independent_at_full_batch(dtype, L, G, O):
    """Rows of a 64-episode step == the same episodes run as a batch of 8 (different tile shapes and grids, same
    per-element arithmetic order): no cross-episode term anywhere on the path.  Also at configs[4]'s per-GPU share
    (64 episodes, 250 instruction tokens, 100 graph nodes) and at 20 objects per view in the product dtype."""
    m = _model(dtype)
    lb, pb, nb = _step_inputs(64, L, G, 777, O=O)
    full = _one_step(m, lb, pb, nb)
    part = _one_step(m, _slice(lb, 8), _slice(pb, 8), _slice(nb, 8))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    for k in ('global_logits', 'local_logits', 'fused_logits', 'cls_embeds'):
        assert _same(full[k][:8], part[k], tol), k
    assert (full['fused_logits'][:8].float().argmax(1) == part['fused_logits'].float().argmax(1)).all()


def test_padding_behind_masks_changes_nothing():
    """16 more (masked) instruction tokens and 6 more (masked) graph-map nodes: every valid logit is unchanged and the
    new nodes are -inf (extend_neg_masks / key-padding semantics, ops.py:24-43)."""
    m = _model(torch.float32)
    B, L, G = 64, 80, 20
    lb, pb, nb = _step_inputs(B, L, G, 911)
    base = _one_step(m, lb, pb, nb)
    lb2 = dict(lb)
    lb2['txt_ids'] = torch.cat([lb['txt_ids'], torch.full((B, 16), 7, device='cuda', dtype=lb['txt_ids'].dtype)], 1)
    lb2['txt_masks'] = torch.cat([lb['txt_masks'], torch.zeros(B, 16, device='cuda', dtype=torch.bool)], 1)
    nb2 = dict(nb)
    P = 6
    nb2['gmap_img_embeds'] = torch.cat([nb['gmap_img_embeds'], torch.randn(B, P, 768, device='cuda')], 1)
    nb2['gmap_step_ids'] = torch.cat([nb['gmap_step_ids'], torch.zeros(B, P, device='cuda', dtype=nb['gmap_step_ids'].dtype)], 1)
    nb2['gmap_pos_fts'] = torch.cat([nb['gmap_pos_fts'], torch.rand(B, P, 7, device='cuda')], 1)
    nb2['gmap_masks'] = torch.cat([nb['gmap_masks'], torch.zeros(B, P, device='cuda', dtype=torch.bool)], 1)
    nb2['gmap_visited_masks'] = torch.cat([nb['gmap_visited_masks'], torch.zeros(B, P, device='cuda', dtype=torch.bool)], 1)
    d = torch.zeros(B, G + P, G + P, device='cuda')
    d[:, :G, :G] = nb['gmap_pair_dists']
    d[:, G:, :] = 3.0
    d[:, :, G:] = 3.0
    nb2['gmap_pair_dists'] = d
    nb2['gmap_vpids'] = [list(v) + [None] * P for v in nb['gmap_vpids']]
    pad = _one_step(m, lb2, pb, nb2)
    assert _same(pad['global_logits'][:, :G], base['global_logits'], 2e-5)
    assert _same(pad['fused_logits'][:, :G], base['fused_logits'], 2e-5)
    assert _same(pad['local_logits'], base['local_logits'], 2e-5)
    assert torch.isinf(pad['fused_logits'][:, G:]).all() and (pad['fused_logits'][:, G:] < 0).all()
    assert (pad['fused_logits'].argmax(1) == base['fused_logits'].argmax(1)).all()


@pytest.mark.parametrize('dtype,L,G', [(torch.float32, 80, 20), (torch.bfloat16, 80, 20), (torch.bfloat16, 250, 100)])
def test_graph_replay_equals_eager_at_full_batch(dtype, L, G):
    """GraphedNavStep (HIP graph, forked streams, static buffers, hoisted text K/V) == eager launches, bit for bit,
    over 3 recurrent steps at B=64 (R2R shapes and the configs[4] shapes)."""
    from dsrg_amd.runner import GraphedNavStep
    m = _model(dtype)
    lb, pb, nb = _step_inputs(64, L, G, 313)
    with torch.no_grad():
        txt = m('language', lb)
    runs = []
    for use_graph in (True, False):
        r = GraphedNavStep(m, txt, lb['txt_masks'], pb, nb, use_graph=use_graph)
        r.set_text(txt, lb['txt_masks'])
        outs = []
        for _ in range(3):
            o = r.step()
            torch.cuda.synchronize()
            outs.append({k: o[k].clone() for k in ('fused_logits', 'local_logits', 'cls_embeds')})
        runs.append(outs)
    for a, b in zip(*runs):
        for k in a:
            assert torch.equal(torch.nan_to_num(a[k].float(), neginf=-1e30), torch.nan_to_num(b[k].float(), neginf=-1e30)), k


def test_graph_replay_equals_eager_reverie_three_chains():
    """REVERIE shapes (58 viewpoint tokens against 20 graph nodes): under capture the local branch runs as two half-batch
    chains beside the global one (vilmodel._x_layers_two_streams, split_big); the replayed step == eager grouped launches,
    bit for bit, object-grounding logits included."""
    from dsrg_amd.config import DSRGConfig
    from dsrg_amd.runner import GraphedNavStep
    from dsrg_amd.vilmodel_reverie import GlocalTextPathNavCMT
    B, L, G, No = 16, 80, 20, 20
    m = GlocalTextPathNavCMT.from_pretrained(None, config=DSRGConfig(), state_dict=common.synth_sd('reverie_c3'), compute_dtype=torch.bfloat16).cuda().eval()
    assert m.split_big > 1 and 58 >= m.split_ratio * G
    lb = common.to_t(synth.language_batch(B, L, 77), 'cuda')
    pbn = synth.panorama_batch(B, 36, 3, 77, reverie=True, No=No)
    nb = common.to_t(synth.navigation_batch(B, G, 36 + No + 2, 77, ncand=pbn['ncand'], nav_types=pbn['nav_types'], vp_lens=36 + pbn['obj_lens'] + 2,
                                            reverie=True), 'cuda')
    pb = common.to_t(pbn, 'cuda')
    with torch.no_grad():
        txt = m('language', lb)
    runs = []
    for use_graph in (True, False):
        r = GraphedNavStep(m, txt, lb['txt_masks'], pb, nb, use_graph=use_graph, reverie=True)
        r.set_text(txt, lb['txt_masks'])
        outs = []
        for _ in range(2):
            o = r.step()
            torch.cuda.synchronize()
            outs.append({k: o[k].clone() for k in ('fused_logits', 'local_logits', 'obj_logits', 'cls_embeds', 'vp_embeds')})
        runs.append(outs)
    for a, b in zip(*runs):
        for k in a:
            assert torch.equal(torch.nan_to_num(a[k].float(), neginf=-1e30), torch.nan_to_num(b[k].float(), neginf=-1e30)), k


def test_bf16_agrees_with_fp32_on_clear_margins_at_full_batch():
    """bf16 product path vs the fp32 parity path at B=64: logits within the stated 6e-2, same action wherever the fp32
    top-2 margin exceeds twice that."""
    lb, pb, nb = _step_inputs(64, 80, 20, 555)
    o32 = _one_step(_model(torch.float32), lb, pb, nb)
    o16 = _one_step(_model(torch.bfloat16), lb, pb, nb)
    assert _same(o16['fused_logits'], o32['fused_logits'], 6e-2)
    f = o32['fused_logits'].float().cpu().numpy()
    top2 = np.sort(np.where(np.isfinite(f), f, -1e30), 1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 0.12
    assert clear.sum() >= 32
    a16, a32 = o16['fused_logits'].float().argmax(1).cpu().numpy(), f.argmax(1)
    assert (a16[clear] == a32[clear]).all()


@pytest.mark.parametrize('task', ['mlm', 'mrc', 'sap'])
def test_full_size_pretraining_losses_match_oracle(task):
    """Pre-training twin at the bench's C4 shapes (24 trajectories of 4-7 steps, 80 instruction tokens): the per-sample
    loss vector of the fp32 HIP path against the CPU oracle on the same batch, within 1e-3 (dropout 0, forward only;
    the golden-vector tests in test_gpu_pretrain.py hold the gradients)."""
    from dsrg_amd.config import DSRGConfig
    from oracle import pretrain_oracle
    from tests.test_gpu_pretrain import _model as pt_model
    from tests.test_oracle_pt_golden import PT_CFG
    torch.set_num_threads(8)
    raw = synth.pretrain_batch(24, 80, 3, 77, task, min_steps=4, max_steps=7)
    m = pt_model(torch.float32)
    with torch.no_grad():
        mine = m(common.to_t(raw, 'cuda'), task, compute_loss=True).float().cpu().numpy()
    sd = common.synth_sd('r2r_pt')
    sd['bert.embeddings.word_embeddings.weight'] = sd['mlm_head.predictions.decoder.weight']   # tied, see the golden test
    with torch.no_grad():
        want = pretrain_oracle.forward(sd, DSRGConfig(**PT_CFG), common.to_t(raw), task, True).numpy()
    assert mine.shape == want.shape and np.isfinite(mine).all()
    assert np.abs(mine - want).max() <= 1e-3, (task, np.abs(mine - want).max())


@pytest.mark.parametrize('reverie', [False, True])
@pytest.mark.parametrize('B,L,G', [(1, 5, 3), (3, 33, 7), (65, 17, 13)])
def test_minimal_and_ragged_shapes_match_oracle(B, L, G, reverie):
    """The other end of the size range: one episode whose graph map is [stop, MEM, current node] and whose
    instruction is 5 tokens; odd sizes that leave a partial tile in every GEMM, attention and row kernel (B=65 is one
    row past a 64-row tile). Two recurrent steps, HIP fp32 vs the CPU oracle, same gates as the full-size chain."""
    from dsrg_amd.config import DSRGConfig
    from oracle import dsrg_oracle
    sd = common.synth_sd('reverie_c3' if reverie else 'r2r_c1')
    if reverie:
        from dsrg_amd.vilmodel_reverie import GlocalTextPathNavCMT
        m = GlocalTextPathNavCMT.from_pretrained(None, config=DSRGConfig(), state_dict=sd, compute_dtype=torch.float32).cuda().eval()
    else:
        m = _model(torch.float32, sd)
    meta = [B, L, G, 3, 11, int(reverie), 20 if reverie else 0]
    with torch.no_grad():
        out = common.run_chain(lambda mode, b: m(mode, b), meta, device='cuda', steps=2)
        ref = common.run_chain(lambda mode, b: dsrg_oracle.forward(sd, DSRGConfig(), mode, b, reverie=reverie), meta, device='cpu', steps=2)
    common.compare(out, ref, tol=1e-3, logit_tol=1e-3, steps=2)


def test_runner_current_node_slot_and_recapture_after_eval():
    """(1) The pooled panorama goes to the CURRENT node's graph-map slot (cur_gmap_idx), not to slot 2 (the start viewpoint
    after the first step: agent.py:165-187, 443-447); checked against direct model calls glued by hand.  (2) The captured
    graph bakes in addresses of the model's packed weights: model.eval() (which the reference agent calls before every
    test()) drops that pack; the runner must notice and re-capture instead of replaying freed memory."""
    from dsrg_amd.runner import GraphedNavStep
    m = _model(torch.bfloat16)
    B = 16
    lb, pb, nb = _step_inputs(B, 80, 20, 99)
    with torch.no_grad():
        txt = m('language', lb)
    cur = (torch.arange(B, device='cuda') % 5) + 2

    def by_hand():
        with torch.no_grad():
            pe, pm, pf = m('panorama', pb)
            n = dict(nb)
            z = torch.zeros(B, 1, 768, device='cuda', dtype=pe.dtype)
            n['vp_img_embeds'] = torch.cat([z, z, pe], 1)
            g = nb['gmap_img_embeds'].to(pe.dtype).clone()
            g[:, 1] = 0
            g[torch.arange(B, device='cuda'), cur] = pf
            n.update(gmap_img_embeds=g, txt_embeds=txt, txt_masks=lb['txt_masks'])
            return m('navigation', n)
    want = by_hand()
    r = GraphedNavStep(m, txt, lb['txt_masks'], pb, nb)
    r.set_text(txt, lb['txt_masks'])
    r.set_current_node(cur)
    got = {k: v.clone() for k, v in r.step().items() if isinstance(v, torch.Tensor)}
    for k in ('fused_logits', 'global_logits', 'cls_embeds'):
        assert torch.equal(torch.nan_to_num(got[k].float(), neginf=-1e30), torch.nan_to_num(want[k].float(), neginf=-1e30)), k
    g_old = r.graph
    m.eval()                                                   # invalidate_caches(): the pack the graph points into is dropped
    junk = [torch.full((1 << 22,), float('nan'), device='cuda') for _ in range(48)]      # reuse the freed blocks
    r.set_text(txt, lb['txt_masks'])
    r.set_current_node(cur)
    again = r.step()
    torch.cuda.synchronize()
    assert r.graph is not g_old                                # re-captured
    for k in ('fused_logits', 'global_logits', 'cls_embeds'):
        assert torch.equal(torch.nan_to_num(again[k].float(), neginf=-1e30), torch.nan_to_num(want[k].float(), neginf=-1e30)), k
    del junk


def test_split_runner_with_device_graphmap_matches_oracle_rollout():
    """GraphedNavStep(split=True) + BatchedGraphMap over a 4-step rollout against the CPU oracle model fed by the restated
    reference GraphMap + builders (oracle/agent_oracle.py) on the same synthetic walks: the static graph-map buffers are
    padded to 24 nodes behind the masks.  fp32; fused logits within 1e-3, same greedy actions, on every step."""
    from dsrg_amd.config import DSRGConfig
    from dsrg_amd.graphmap import BatchedGraphMap
    from dsrg_amd.runner import GraphedNavStep
    from oracle import dsrg_oracle, agent_oracle as AO
    torch.set_num_threads(16)
    B, T, V, L, seed, GS = 4, 4, 36, 40, 21, 24
    cfg = DSRGConfig()
    sd = common.synth_sd('r2r_c1')
    m = _model(torch.float32, dict(sd))
    obs, ended = synth.episode_walks(B, T, seed)
    lb = synth.language_batch(B, L, seed)
    t_cpu = dsrg_oracle.forward(sd, cfg, 'language', common.to_t(lb))
    lbg = common.to_t(lb, 'cuda')
    with torch.no_grad():
        t_gpu = m('language', lbg)
    gmaps = [AO.GraphMap(ob['viewpoint']) for ob in obs[0]]
    for i, ob in enumerate(obs[0]):
        gmaps[i].update_graph(ob)
    gm = BatchedGraphMap(B, 768, 'cuda', dtype=torch.float32, max_nodes=48, max_cand=8)
    gm.reset(obs[0])
    pb0 = synth.panorama_batch(B, V, 3, seed)
    nb0 = common.to_t(synth.navigation_batch(B, GS, V + 2, seed, ncand=pb0['ncand'], nav_types=pb0['nav_types']), 'cuda')
    r = GraphedNavStep(m, t_gpu, lbg['txt_masks'], common.to_t(pb0, 'cuda'), nb0, split=True)
    r.set_text(t_gpu, lbg['txt_masks'])
    mem_c = None
    for t in range(T):
        cand = [[c['viewpointId'] for c in ob['candidate']] for ob in obs[t]]
        pb = synth.panorama_batch(B, V, 3, seed + t)
        pb['nav_types'] = np.stack([np.asarray([1] * len(c) + [0] * (V - len(c)), np.int64) for c in cand])
        AO.step_begin(gmaps, obs[t], t, ended[t])
        pe, pm, pf = dsrg_oracle.forward(sd, cfg, 'panorama', common.to_t(pb))
        AO.update_embeds(gmaps, obs[t], cand, pe.numpy(), pf.numpy(), ended[t])
        nav = AO.nav_gmap_variable(obs[t], gmaps, None if mem_c is None else mem_c.numpy())
        nav.update(AO.nav_vp_variable(obs[t], gmaps, pe.numpy(), cand, [V] * B, pb['nav_types'], None if mem_c is None else mem_c.numpy()))
        nbc = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in nav.items()}
        nbc.update(txt_embeds=t_cpu, txt_masks=torch.from_numpy(lb['txt_masks']), vp_obj_masks=None)
        o_c = dsrg_oracle.forward(sd, cfg, 'navigation', nbc)
        mem_c = o_c['cls_embeds']
        # ---- HIP side: pano graph -> device graph map -> nav graph
        for k, v in common.to_t(pb, 'cuda').items():
            if isinstance(v, torch.Tensor) and k in r.pano:
                r.pano[k].copy_(v)
        gm.begin_step(obs[t], t, ended[t])
        pe_g, pm_g, pf_g = r.pano_step()
        gm.update_embeds(pe_g, pf_g, ended[t])
        ng = gm.nav_inputs(pe_g, r.pano['nav_types'], torch.full((B,), V, device='cuda'), None if t == 0 else r.out['cls_embeds'])
        r.load_nav(ng)
        o_g = r.nav_step()
        G = ng['gmap_masks'].shape[1]
        fc, fg = o_c['fused_logits'].numpy(), o_g['fused_logits'].float().cpu().numpy()
        assert np.isneginf(fg[:, G:]).all()
        fg = fg[:, :G]
        fin = np.isfinite(fc)
        assert (np.isfinite(fg) == fin).all() and np.abs(fg[fin] - fc[fin]).max() <= 1e-3, (t, np.abs(fg[fin] - fc[fin]).max())
        assert np.array_equal(fg.argmax(1), fc.argmax(1))
        assert int(gm.current_token_index(ng)[0]) >= 2
        if t + 1 < T:
            for i, ob in enumerate(obs[t + 1]):
                if not ended[t + 1][i]:
                    gmaps[i].update_graph(ob)
            gm.observe(obs[t + 1], ended[t + 1])
