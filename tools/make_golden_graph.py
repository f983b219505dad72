#!/usr/bin/env python3
"""Golden vectors for the navigation-input assembly, generated with the REFERENCE's own
finetune_src/models/graph_utils.GraphMap (imported read-only in the build container; numpy only).  The two padding
builders of agent.py cannot be imported (MatterSim ...); oracle/agent_oracle.py's transcription of them is applied on top
of the reference GraphMap.  Output: tests/golden/graphmap.npz = per step the nav inputs of B episodes.

    python tools/make_golden_graph.py
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference/finetune_src')
import numpy as np

from models import graph_utils as RG          # the reference
from dsrg_amd import synth
from oracle import agent_oracle as AO

B, T, H, V, SEED = 6, 7, 32, 36, 11


def embeds(t):
    return (synth.uniform('gm_pano%d' % t, (B, V, H), SEED), synth.uniform('gm_fused%d' % t, (B, H), SEED),
            synth.uniform('gm_mem%d' % t, (B, H), SEED))


def run(GraphMapCls):
    obs, ended = synth.episode_walks(B, T, SEED)
    gmaps = [GraphMapCls(ob['viewpoint']) for ob in obs[0]]
    for i, ob in enumerate(obs[0]):
        gmaps[i].update_graph(ob)
    out = {}
    for t in range(T):
        AO.step_begin(gmaps, obs[t], t, ended[t])
        pano, fused, mem = embeds(t)
        cand = [[c['viewpointId'] for c in ob['candidate']] for ob in obs[t]]
        AO.update_embeds(gmaps, obs[t], cand, pano, fused, ended[t])
        nav = AO.nav_gmap_variable(obs[t], gmaps, None if t == 0 else mem)
        nav_types = np.stack([np.asarray([1] * len(c) + [0] * (V - len(c))) for c in cand])
        nav.update(AO.nav_vp_variable(obs[t], gmaps, pano, cand, [V] * B, nav_types, None if t == 0 else mem))
        for k, v in nav.items():
            if isinstance(v, np.ndarray):
                out['s%d_%s' % (t, k)] = v
        out['s%d_no_vp_left' % t] = np.asarray(nav['no_vp_left'])
        out['s%d_gmap_len' % t] = np.asarray([len(x) for x in nav['gmap_vpids']])
        if t + 1 < T:
            for i, ob in enumerate(obs[t + 1]):
                if not ended[t + 1][i]:
                    gmaps[i].update_graph(ob)
    return out


if __name__ == '__main__':
    ref = run(RG.GraphMap)
    mine = run(AO.GraphMap)
    for k in ref:
        assert ref[k].shape == mine[k].shape and np.array_equal(ref[k], mine[k]), k
    ref['meta'] = np.asarray([B, T, H, V, SEED])
    path = os.path.join(ROOT, 'tests', 'golden', 'graphmap.npz')
    np.savez_compressed(path, **ref)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(ref), 'arrays; restated GraphMap identical to the reference one')
