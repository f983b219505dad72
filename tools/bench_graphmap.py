#!/usr/bin/env python3
"""Navigation-input assembly per rollout step: device-resident BatchedGraphMap vs the CPU restatement of
the reference's GraphMap + _nav_gmap_variable / _nav_vp_variable (oracle/agent_oracle.py), same synthetic rollouts.
B = 64 episodes, 15 steps (max_action_len), 36 views, hidden 768.  Prints one JSON line.

    python tools/bench_graphmap.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dsrg_amd import synth

B, T, V, H, SEED = 64, 15, 36, 768, 5


def main():
    from dsrg_amd.graphmap import BatchedGraphMap
    from oracle import agent_oracle as AO
    obs, ended = synth.episode_walks(B, T, SEED, world=80)
    ended[:] = False                                   # every episode runs all 15 steps: the worst case for both sides
    pano = torch.from_numpy(synth.uniform('bgm_pano', (B, V, H), SEED)).cuda().bfloat16()
    fused = torch.from_numpy(synth.uniform('bgm_fused', (B, H), SEED)).cuda().bfloat16()
    mem = torch.from_numpy(synth.uniform('bgm_mem', (B, H), SEED)).cuda().bfloat16()
    cand = [[[c['viewpointId'] for c in ob['candidate']] for ob in obs[t]] for t in range(T)]
    nav_types = [torch.from_numpy(np.stack([np.asarray([1] * len(c) + [0] * (V - len(c))) for c in cand[t]])).cuda() for t in range(T)]
    view_lens = torch.full((B,), V, device='cuda')

    def device_rollout():
        gm = BatchedGraphMap(B, H, 'cuda', dtype=torch.bfloat16, max_nodes=96, max_cand=8)
        gm.reset(obs[0])
        for t in range(T):
            gm.begin_step(obs[t], t, ended[t])
            gm.update_embeds(pano, fused, ended[t])
            nav = gm.nav_inputs(pano, nav_types[t], view_lens, None if t == 0 else mem)
            if t + 1 < T:
                gm.observe(obs[t + 1], ended[t + 1])
        torch.cuda.synchronize()
        return nav

    device_rollout()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        nav = device_rollout()
    dev_ms = (time.perf_counter() - t0) / (reps * T) * 1e3
    G = nav['gmap_pos_fts'].shape[1]

    pano_c, fused_c, mem_c = pano.float().cpu().numpy(), fused.float().cpu().numpy(), mem.float().cpu().numpy()
    nt_c = [x.cpu().numpy() for x in nav_types]

    def cpu_rollout():
        gmaps = [AO.GraphMap(ob['viewpoint']) for ob in obs[0]]
        for i, ob in enumerate(obs[0]):
            gmaps[i].update_graph(ob)
        for t in range(T):
            AO.step_begin(gmaps, obs[t], t, ended[t])
            AO.update_embeds(gmaps, obs[t], cand[t], pano_c, fused_c, ended[t])
            nav = AO.nav_gmap_variable(obs[t], gmaps, None if t == 0 else mem_c)
            nav.update(AO.nav_vp_variable(obs[t], gmaps, pano_c, cand[t], [V] * B, nt_c[t], None if t == 0 else mem_c))
            if t + 1 < T:
                for i, ob in enumerate(obs[t + 1]):
                    gmaps[i].update_graph(ob)
        return nav
    t0 = time.perf_counter()
    cpu_rollout()
    cpu_ms = (time.perf_counter() - t0) / T * 1e3
    print(json.dumps({'metric': 'nav-input assembly per rollout step (graph update + node embeddings + all graph-side inputs of the navigation call)',
                      'episodes': B, 'steps': T, 'graph_nodes_at_last_step': int(G) - 2, 'device_ms_per_step': dev_ms,
                      'device_note': 'BatchedGraphMap, 3 HIP launches + host id->slot bookkeeping + packed H2D of the observation, wall clock incl. host',
                      'cpu_ms_per_step': cpu_ms, 'cpu_note': 'oracle/agent_oracle.py (restated reference GraphMap + builders), 1 thread, numpy embeddings (no H2D)',
                      'speedup': cpu_ms / dev_ms}))


if __name__ == '__main__':
    main()
