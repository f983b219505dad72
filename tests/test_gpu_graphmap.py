"""gpu: the device-resident BatchedGraphMap (graphmap.hip through the C-ABI) against the fixture produced with the
reference's own GraphMap (tests/golden/graphmap.npz): 7 rollout steps of 6 episodes on synthetic connectivity graphs.
Integer / boolean outputs, graph distances and hop counts must be identical; the float32 sin / cos of the position
features may differ in the last bit (tolerance 1e-6); mean node embeddings 1e-6."""
import numpy as np
import pytest
import torch

from tests import common
from tests.test_oracle_graph_golden import replay

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_batched_graphmap_matches_reference_fixture(dtype):
    from dsrg_amd.graphmap import BatchedGraphMap
    from dsrg_amd.vilmodel import GlocalTextPathNavCMT
    fp32 = dtype == torch.float32

    def make(obs0):
        gm = BatchedGraphMap(len(obs0), 32, 'cuda', dtype=dtype, max_nodes=48, max_cand=8)
        gm.reset(obs0)
        return gm

    def step(gm, t, obs, ended, pano, fused, mem, cand, nav_types, gold, nobs, nended):
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        gm.begin_step(obs, t, ended)
        pe, pf = dev(pano).to(dtype), dev(fused).to(dtype)
        gm.update_embeds(pe, pf, ended)
        B, V = pano.shape[:2]
        nav = gm.nav_inputs(pe, dev(nav_types), torch.full((B,), V, device='cuda'), None if mem is None else dev(mem).to(dtype))
        g = lambda k: gold['s%d_%s' % (t, k)]
        G = g('gmap_step_ids').shape[1]
        assert nav['gmap_step_ids'].shape[1] == G
        for k in ('gmap_step_ids', 'gmap_visited_masks', 'gmap_masks', 'vp_masks', 'vp_nav_masks'):
            assert np.array_equal(nav[k].cpu().numpy(), g(k)), (t, k)
        assert np.array_equal(nav['no_vp_left'].cpu().numpy(), g('no_vp_left')) and np.array_equal(nav['gmap_len'].cpu().numpy(), g('gmap_len'))
        assert np.array_equal(nav['gmap_pair_dists'].cpu().numpy(), g('gmap_pair_dists')), t
        for k in ('gmap_pos_fts', 'vp_pos_fts'):
            mine, ref = nav[k].cpu().numpy(), g(k)
            assert np.array_equal(mine[..., 4:7], ref[..., 4:7]) and np.abs(mine - ref).max() <= 1e-6, (t, k, np.abs(mine - ref).max())
            if k == 'vp_pos_fts':
                assert np.array_equal(mine[..., 11:14], ref[..., 11:14])
        tol = 1e-6 if fp32 else 2e-2
        for k in ('gmap_img_embeds', 'vp_img_embeds'):
            assert np.abs(nav[k].float().cpu().numpy() - g(k)).max() <= tol, (t, k)
        # the device-built fusion indices equal the host matcher on the viewpoint-id strings
        src, bw = GlocalTextPathNavCMT.fusion_indices(nav['gmap_vpids'], nav['gmap_visited_masks'], nav['vp_cand_vpids'], G, V + 2, 2)
        assert np.array_equal(nav['gmap_src'].cpu().numpy(), src) and np.array_equal(nav['bw_mask'].cpu().numpy(), bw), t
        if nobs is not None:
            gm.observe(nobs, nended)
    replay(make, step)


def test_rollout_with_device_graphmap_matches_oracle_rollout():
    """The pieces compose: language -> per step [panorama -> device graph map -> navigation -> action tail] on the HIP side
    (fp32) against the CPU oracle model fed by the restated reference GraphMap + builders, on the same synthetic walks.
    Fused logits within 1e-3, same greedy actions, same stop probabilities."""
    from dsrg_amd import synth, functional as Fn
    from dsrg_amd.config import DSRGConfig
    from dsrg_amd.graphmap import BatchedGraphMap
    from dsrg_amd.vilmodel import GlocalTextPathNavCMT
    from oracle import dsrg_oracle, agent_oracle as AO
    torch.set_num_threads(16)
    B, T, V, L, seed = 4, 4, 36, 40, 21
    cfg = DSRGConfig()
    sd = common.synth_sd('r2r_c1')
    m = GlocalTextPathNavCMT.from_pretrained(None, config=cfg, state_dict=dict(sd), compute_dtype=torch.float32).cuda().eval()
    obs, ended = synth.episode_walks(B, T, seed)
    lb = synth.language_batch(B, L, seed)
    t_cpu = dsrg_oracle.forward(sd, cfg, 'language', common.to_t(lb))
    with torch.no_grad():
        t_gpu = m('language', common.to_t(lb, 'cuda'))
    gmaps = [AO.GraphMap(ob['viewpoint']) for ob in obs[0]]
    for i, ob in enumerate(obs[0]):
        gmaps[i].update_graph(ob)
    gm = BatchedGraphMap(B, 768, 'cuda', dtype=torch.float32, max_nodes=48, max_cand=8)
    gm.reset(obs[0])
    mem_c = mem_g = None
    for t in range(T):
        cand = [[c['viewpointId'] for c in ob['candidate']] for ob in obs[t]]
        pb = synth.panorama_batch(B, V, 3, seed + t)
        pb['nav_types'] = np.stack([np.asarray([1] * len(c) + [0] * (V - len(c)), np.int64) for c in cand])
        # ---- oracle side
        AO.step_begin(gmaps, obs[t], t, ended[t])
        pe, pm, pf = dsrg_oracle.forward(sd, cfg, 'panorama', common.to_t(pb))
        AO.update_embeds(gmaps, obs[t], cand, pe.numpy(), pf.numpy(), ended[t])
        nav = AO.nav_gmap_variable(obs[t], gmaps, None if mem_c is None else mem_c.numpy())
        nav.update(AO.nav_vp_variable(obs[t], gmaps, pe.numpy(), cand, [V] * B, pb['nav_types'], None if mem_c is None else mem_c.numpy()))
        nb = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in nav.items()}
        nb.update(txt_embeds=t_cpu, txt_masks=torch.from_numpy(lb['txt_masks']), vp_obj_masks=None)
        o_c = dsrg_oracle.forward(sd, cfg, 'navigation', nb)
        mem_c = o_c['cls_embeds']
        # ---- HIP side
        with torch.no_grad():
            gm.begin_step(obs[t], t, ended[t])
            pe_g, pm_g, pf_g = m('panorama', common.to_t(pb, 'cuda'))
            gm.update_embeds(pe_g, pf_g, ended[t])
            ng = gm.nav_inputs(pe_g, torch.from_numpy(pb['nav_types']).cuda(), torch.full((B,), V, device='cuda'), mem_g)
            ng.update(txt_embeds=t_gpu, txt_masks=torch.from_numpy(lb['txt_masks']).cuda())
            o_g = m('navigation', ng)
            mem_g = o_g['cls_embeds']
            tail = Fn.action_tail(o_g['fused_logits'])
        fc, fg = o_c['fused_logits'].numpy(), o_g['fused_logits'].cpu().numpy()
        fin = np.isfinite(fc)
        assert (np.isfinite(fg) == fin).all() and np.abs(fg[fin] - fc[fin]).max() <= 1e-3, (t, np.abs(fg[fin] - fc[fin]).max())
        assert np.array_equal(tail['argmax'].cpu().numpy(), fc.argmax(1))
        assert np.abs(tail['stop_prob'].cpu().numpy() - torch.softmax(o_c['fused_logits'], 1)[:, 0].numpy()).max() <= 1e-4
        if t + 1 < T:
            for i, ob in enumerate(obs[t + 1]):
                if not ended[t + 1][i]:
                    gmaps[i].update_graph(ob)
            gm.observe(obs[t + 1], ended[t + 1])
