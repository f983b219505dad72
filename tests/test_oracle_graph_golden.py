"""not-gpu: the restated GraphMap / FloydGraph and nav-input builders (oracle/agent_oracle.py) reproduce the fixture that
tools/make_golden_graph.py produced with the REFERENCE's graph_utils.GraphMap (tests/golden/graphmap.npz)."""
import numpy as np

from dsrg_amd import synth
from oracle import agent_oracle as AO
from tests import common


def replay(make_gmaps, per_step):
    gold = common.load_gold('graphmap')
    B, T, H, V, SEED = [int(x) for x in gold['meta']]
    obs, ended = synth.episode_walks(B, T, SEED)
    state = make_gmaps(obs[0])
    for t in range(T):
        pano, fused, mem = (synth.uniform('gm_pano%d' % t, (B, V, H), SEED), synth.uniform('gm_fused%d' % t, (B, H), SEED),
                            synth.uniform('gm_mem%d' % t, (B, H), SEED))
        cand = [[c['viewpointId'] for c in ob['candidate']] for ob in obs[t]]
        nav_types = np.stack([np.asarray([1] * len(c) + [0] * (V - len(c))) for c in cand])
        per_step(state, t, obs[t], ended[t], pano, fused, None if t == 0 else mem, cand, nav_types, gold,
                 obs[t + 1] if t + 1 < T else None, ended[t + 1] if t + 1 < T else None)
    return gold


def test_restated_graphmap_matches_reference_fixture():
    def make(obs0):
        g = [AO.GraphMap(ob['viewpoint']) for ob in obs0]
        for i, ob in enumerate(obs0):
            g[i].update_graph(ob)
        return g

    def step(gmaps, t, obs, ended, pano, fused, mem, cand, nav_types, gold, nobs, nended):
        AO.step_begin(gmaps, obs, t, ended)
        AO.update_embeds(gmaps, obs, cand, pano, fused, ended)
        nav = AO.nav_gmap_variable(obs, gmaps, mem)
        nav.update(AO.nav_vp_variable(obs, gmaps, pano, cand, [pano.shape[1]] * len(obs), nav_types, mem))
        for k, v in nav.items():
            if isinstance(v, np.ndarray):
                assert np.array_equal(v, gold['s%d_%s' % (t, k)]), (t, k)
        if nobs is not None:
            for i, ob in enumerate(nobs):
                if not nended[i]:
                    gmaps[i].update_graph(ob)
    replay(make, step)
